/* oracle/gms_oracle.h — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's set-intersection hot path (spcl/gms), used as the
 * checker for the HIP kernels and as bench.py's `cpu_baseline` leg.  Nothing under gms_amd/
 * (the product) may include, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function here against
 *   (a) the golden vectors in tests/golden/ (generated from the compiled reference by
 *       tools/make_golden.py through oracle/_ref/libgms_ref.so), and
 *   (b) the compiled reference itself whenever oracle/_ref/libgms_ref.so is present.
 *
 * All graphs are CSR: int64 offsets[n+1], int32 neigh[offsets[n]], rows sorted ascending,
 * duplicate- and loop-free, symmetric (what the reference loader produces:
 * gms/third_party/gapbs/builder.h:206-235,260-277).  File:line citations are relative to
 * /root/reference.
 */
#ifndef GMS_ORACLE_H
#define GMS_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- SortedSet algebra (gms/representations/sets/sorted_set_operations.h) ---- */
uint64_t gmso_intersect_count(const int32_t *a, size_t na, const int32_t *b, size_t nb); /* :44-71 */
size_t gmso_intersect(const int32_t *a, size_t na, const int32_t *b, size_t nb, int32_t *out); /* :36-42 */
size_t gmso_difference(const int32_t *a, size_t na, const int32_t *b, size_t nb, int32_t *out); /* :73-106 */
size_t gmso_union(const int32_t *a, size_t na, const int32_t *b, size_t nb, int32_t *out); /* :29-35 */
uint64_t gmso_union_count(const int32_t *a, size_t na, const int32_t *b, size_t nb); /* sorted_set.h:140-158 */
/* SortedSet(const T*, size_t): copy + sort (sorted_set.h:64-66); duplicates are kept, as there. */
void gmso_make_set(const int32_t *in, size_t n, int32_t *out);

/* ---- triangle counting (gms/algorithms/set_based/triangle_count) ---- */
/* Par::count_total, parallel/total.h:7-24: sum over u, v in N(u), u<v of |N(u) ∩ N(v)|; returns total/3.
 * raw_total (may be NULL) receives the un-divided sum so callers can check the reference's assert(total%3==0). */
uint64_t gmso_tc_total(int64_t n, const int64_t *off, const int32_t *neigh, int threads, uint64_t *raw_total);
/* The same loop restricted to u ≡ phase (mod stride): a bounded, unbiased sample of the workload for
 * the CPU baseline.  Returns the raw (undivided) partial sum; *edges = #intersect_count calls made,
 * *elements = Σ(d_u+d_v) over those calls. */
uint64_t gmso_tc_total_sample(int64_t n, const int64_t *off, const int32_t *neigh, int threads,
                              int64_t stride, int64_t phase, uint64_t *edges, uint64_t *elements);
/* Par::vertex_count2, parallel/vertex.h:14-27: counts[u] = Σ_{v in N(u)} |N(u) ∩ N(v)|. */
void gmso_tc_vertex_count2(int64_t n, const int64_t *off, const int32_t *neigh, int threads, int64_t *counts);
/* Par::vertex_count2_once, parallel/vertex.h:30-49 (u<v only, added to both ends; counts must be zeroed by the caller as there). */
void gmso_tc_vertex_count2_once(int64_t n, const int64_t *off, const int32_t *neigh, int threads, int64_t *counts);

/* ---- k-clique counting (gms/algorithms/set_based/k_clique_count/k_clique_count_set_based.h:5-31) ----
 * Returns the reference's value: k! * (#k-cliques) on a symmetric graph, arithmetic mod 2^64 like size_t. */
uint64_t gmso_kclique(int64_t n, const int64_t *off, const int32_t *neigh, int k, int threads);
/* KCliqueStar::Par::CliqueStar in count mode (k_clique_star_list/parallel/recursive.h:19-35, sequential/recursive.h:31-71): the number of
 * (k-clique, star) pairs and the total cardinality of the stars */
void gmso_kclique_star_count(int64_t n, const int64_t *off, const int32_t *neigh, int k, int threads, uint64_t *count_out, uint64_t *members_out);

/* ---- Bron–Kerbosch maximal-clique count ----
 * BkEppsteinPar::mceBench (parallel/eppsteinPAR.h:18-53) over BkTomita::expand/findPivot
 * (sequential/tomita.h:12-86) with -DBK_COUNT: returns BK_CLIQUE_COUNTER.  rank[] is in rank format. */
uint64_t gmso_bk_count(int64_t n, const int64_t *off, const int32_t *neigh, const int32_t *rank, int threads);
/* PpParallel::getDegreeOrdering<…, useRankFormat=true> (gms/algorithms/preprocessing/parallel/degree.h:16-62):
 * rank = position in ascending (degree, id) order. */
void gmso_degree_rank(int64_t n, const int64_t *off, int32_t *rank);

/* PpParallel::getDegeneracyOrderingApproxSGraph<boundary_function::averageDegree, useRankFormat=true>
 * (gms/algorithms/preprocessing/parallel/degeneracy_approx_set.h:14-86, boundary_function.h:14-23): rounds of
 * "remove every remaining vertex with remaining degree <= (unsigned)((1+eps) * mean remaining degree)", the others'
 * degrees drop by |N(v) ∩ X| (PULL style, :74-79).  rank[v] = position.  round_of / deg_at (may be NULL) receive the round a
 * vertex left in and its remaining degree at that moment.  Ties inside a round: (degree, id) — the reference leaves them to
 * __gnu_parallel::partition/sort; tests check its output is one of the orders this function's (round, degree) keys allow.
 * Returns the number of rounds. */
int32_t gmso_adg_rank(int64_t n, const int64_t *off, const int32_t *neigh, double epsilon, int32_t *rank, int32_t *round_of,
                      int32_t *deg_at);
/* PpParallel::triangleCountOrdering (preprocessing/parallel/triangle_count.h:11-30) over Par::vertex_count2_once:
 * ordering[i] = i-th vertex by increasing count, ties by id (std::sort leaves them unspecified). */
void gmso_tc_ordering(int64_t n, const int64_t *off, const int32_t *neigh, int threads, int32_t *ordering);

/* ---- vertex similarity (gms/algorithms/set_based/vertex_similarity/vertex_similarity.h:30-222) ----
 * metric: 0 Jaccard (sic: count/(|A|+|B|+count), :31-36), 1 Overlap (:66-68), 2 Adamic-Adar (:96-108), 3 Resource (:120-128),
 * 4 CommNeigh (:139-143), 5 TotalNeigh (:155-159), 6 PrefAtt (:171-174).  Double precision, reference evaluation order. */
double gmso_vertex_similarity(int metric, int64_t n, const int64_t *off, const int32_t *neigh, int32_t a, int32_t b);

/* Σ_{(u,v), u<v} (d_u + d_v): the element count behind SURVEY §8(d)'s B_alg. */
uint64_t gmso_tc_elements(int64_t n, const int64_t *off, const int32_t *neigh);
int gmso_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
