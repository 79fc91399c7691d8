/* oracle/gms_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY (see gms_oracle.h).
 * Plain C11 + OpenMP restatement of the reference's algorithms; written from the behaviour of the
 * cited reference functions, sharing no code with them.  Parity: PINNED (tests/test_oracle.py). */
#include "gms_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ set algebra */

/* sorted_set_operations.h:44-71 — scalar two-pointer merge, one step per comparison. */
uint64_t gmso_intersect_count(const int32_t *a, size_t na, const int32_t *b, size_t nb) {
    size_t i = 0, j = 0;
    uint64_t c = 0;
    while (i < na && j < nb) {
        int32_t x = a[i], y = b[j];
        if (x == y) { c++; i++; j++; }
        else if (x > y) j++;
        else i++;
    }
    return c;
}

/* sorted_set_operations.h:36-42 (std::set_intersection semantics). */
size_t gmso_intersect(const int32_t *a, size_t na, const int32_t *b, size_t nb, int32_t *out) {
    size_t i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) i++;
        else if (b[j] < a[i]) j++;
        else { out[k++] = a[i]; i++; j++; }
    }
    return k;
}

/* sorted_set_operations.h:73-106 — elements of a not in b, then the tail of a. */
size_t gmso_difference(const int32_t *a, size_t na, const int32_t *b, size_t nb, int32_t *out) {
    size_t i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        int32_t x = a[i], y = b[j];
        if (x == y) { i++; j++; }
        else if (x > y) j++;
        else { out[k++] = x; i++; }
    }
    while (i < na) out[k++] = a[i++];
    return k;
}

/* sorted_set_operations.h:29-35 (std::set_union semantics). */
size_t gmso_union(const int32_t *a, size_t na, const int32_t *b, size_t nb, int32_t *out) {
    size_t i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) out[k++] = a[i++];
        else if (b[j] < a[i]) out[k++] = b[j++];
        else { out[k++] = a[i]; i++; j++; }
    }
    while (i < na) out[k++] = a[i++];
    while (j < nb) out[k++] = b[j++];
    return k;
}

/* sorted_set.h:140-158 — |A ∪ B| by a counting merge. */
uint64_t gmso_union_count(const int32_t *a, size_t na, const int32_t *b, size_t nb) {
    return (uint64_t)na + (uint64_t)nb - gmso_intersect_count(a, na, b, nb);
}

static int cmp_i32(const void *p, const void *q) {
    int32_t x = *(const int32_t *)p, y = *(const int32_t *)q;
    return (x > y) - (x < y);
}
void gmso_make_set(const int32_t *in, size_t n, int32_t *out) {
    if (n) memcpy(out, in, n * sizeof(int32_t));
    qsort(out, n, sizeof(int32_t), cmp_i32);
}

/* ------------------------------------------------------------------ triangle counting */

static int pick_threads(int threads) {
#ifdef _OPENMP
    return threads > 0 ? threads : omp_get_max_threads();
#else
    (void)threads;
    return 1;
#endif
}
int gmso_max_threads(void) { return pick_threads(0); }

/* parallel/total.h:7-24 */
uint64_t gmso_tc_total(int64_t n, const int64_t *off, const int32_t *neigh, int threads, uint64_t *raw_total) {
    uint64_t total = 0;
    int nt = pick_threads(threads);
    (void)nt;
#pragma omp parallel for schedule(static, 17) reduction(+ : total) num_threads(nt)
    for (int64_t u = 0; u < n; u++) {
        const int32_t *nu = neigh + off[u];
        size_t du = (size_t)(off[u + 1] - off[u]);
        for (size_t i = 0; i < du; i++) {
            int32_t v = nu[i];
            if (u < v) total += gmso_intersect_count(nu, du, neigh + off[v], (size_t)(off[v + 1] - off[v]));
        }
    }
    if (raw_total) *raw_total = total;
    return total / 3;
}

uint64_t gmso_tc_total_sample(int64_t n, const int64_t *off, const int32_t *neigh, int threads,
                              int64_t stride, int64_t phase, uint64_t *edges, uint64_t *elements) {
    uint64_t total = 0, ne = 0, nel = 0;
    int nt = pick_threads(threads);
    (void)nt;
    int64_t cnt = phase < n ? (n - phase + stride - 1) / stride : 0;
#pragma omp parallel for schedule(static, 17) reduction(+ : total, ne, nel) num_threads(nt)
    for (int64_t s = 0; s < cnt; s++) {
        int64_t u = phase + s * stride;
        const int32_t *nu = neigh + off[u];
        size_t du = (size_t)(off[u + 1] - off[u]);
        for (size_t i = 0; i < du; i++) {
            int32_t v = nu[i];
            if (u < v) {
                size_t dv = (size_t)(off[v + 1] - off[v]);
                total += gmso_intersect_count(nu, du, neigh + off[v], dv);
                ne++;
                nel += du + dv;
            }
        }
    }
    if (edges) *edges = ne;
    if (elements) *elements = nel;
    return total;
}

/* parallel/vertex.h:14-27 */
void gmso_tc_vertex_count2(int64_t n, const int64_t *off, const int32_t *neigh, int threads, int64_t *counts) {
    int nt = pick_threads(threads);
    (void)nt;
#pragma omp parallel for schedule(static, 9) num_threads(nt)
    for (int64_t u = 0; u < n; u++) {
        const int32_t *nu = neigh + off[u];
        size_t du = (size_t)(off[u + 1] - off[u]);
        int64_t c = 0;
        for (size_t i = 0; i < du; i++) {
            int32_t v = nu[i];
            c += (int64_t)gmso_intersect_count(nu, du, neigh + off[v], (size_t)(off[v + 1] - off[v]));
        }
        counts[u] = c;
    }
}

/* parallel/vertex.h:30-49 */
void gmso_tc_vertex_count2_once(int64_t n, const int64_t *off, const int32_t *neigh, int threads, int64_t *counts) {
    int nt = pick_threads(threads);
    (void)nt;
#pragma omp parallel for schedule(dynamic, 9) num_threads(nt)
    for (int64_t u = 0; u < n; u++) {
        const int32_t *nu = neigh + off[u];
        size_t du = (size_t)(off[u + 1] - off[u]);
        int64_t c = 0;
        for (size_t i = 0; i < du; i++) {
            int32_t v = nu[i];
            if (u < v) {
                int64_t x = (int64_t)gmso_intersect_count(nu, du, neigh + off[v], (size_t)(off[v + 1] - off[v]));
                c += x;
#pragma omp atomic
                counts[v] += x;
            }
        }
#pragma omp atomic
        counts[u] += c;
    }
}

uint64_t gmso_tc_elements(int64_t n, const int64_t *off, const int32_t *neigh) {
    uint64_t s = 0;
#pragma omp parallel for reduction(+ : s) schedule(dynamic, 1024)
    for (int64_t u = 0; u < n; u++) {
        uint64_t du = (uint64_t)(off[u + 1] - off[u]);
        for (int64_t e = off[u]; e < off[u + 1]; e++) {
            int32_t v = neigh[e];
            if (u < v) s += du + (uint64_t)(off[v + 1] - off[v]);
        }
    }
    return s;
}

/* ------------------------------------------------------------------ k-clique counting */

/* k_clique_count_set_based.h:5-17.  `isect` has `ni` sorted elements. */
static uint64_t kclique_step(const int64_t *off, const int32_t *neigh, size_t k, const int32_t *isect, size_t ni) {
    if (k == 1) return (uint64_t)ni;
    uint64_t cur = 0;
    int32_t *buf = (int32_t *)malloc((ni ? ni : 1) * sizeof(int32_t));
    for (size_t i = 0; i < ni; i++) {
        int32_t vi = isect[i];
        size_t nc = gmso_intersect(isect, ni, neigh + off[vi], (size_t)(off[vi + 1] - off[vi]), buf);
        if (nc >= k - 2) cur += kclique_step(off, neigh, k - 1, buf, nc); /* size_t compare, as in the reference */
    }
    free(buf);
    return cur;
}

/* k_clique_count_set_based.h:19-31 */
uint64_t gmso_kclique(int64_t n, const int64_t *off, const int32_t *neigh, int k, int threads) {
    uint64_t total = 0;
    int nt = pick_threads(threads);
    (void)nt;
#pragma omp parallel for reduction(+ : total) schedule(dynamic, 64) num_threads(nt)
    for (int64_t u = 0; u < n; u++)
        total += kclique_step(off, neigh, (size_t)k - 1, neigh + off[u], (size_t)(off[u + 1] - off[u]));
    return total;
}

/* ------------------------------------------------------------------ k-clique-star (count mode)
 * GMS::KCliqueStar::Par::CliqueStar / Seq::RecursiveStepCliqueStar
 * (gms/algorithms/set_based/k_clique_star_list/parallel/recursive.h:19-35, sequential/recursive.h:31-71):
 * for every vertex u the recursion extends curClique = {u} by vertices vi of isect = N(u) ∩ … that are LARGER than every member
 * (`vi <= vj` -> skip), so each k-clique is reached once, in ascending order; at k == 0 the star of the clique is
 * ∩_{v in clique} (N(v) \ clique) and one (clique, star) pair is pushed.  Count mode: the number of pushes, and — what a listing would
 * carry — the total cardinality of the stars. */
static void kcstar_step(const int64_t *off, const int32_t *neigh, int k, int32_t *clique, int depth, const int32_t *isect, size_t ni,
                        uint64_t *count, uint64_t *members) {
    if (k == 0) {
        /* star = the common neighbours of all members outside the clique (sequential/recursive.h:44-49) */
        size_t cap = (size_t)(off[clique[0] + 1] - off[clique[0]]);
        int32_t *star = (int32_t *)malloc((cap + 1) * sizeof(int32_t)), *tmp = (int32_t *)malloc((cap + 1) * sizeof(int32_t));
        size_t ns = gmso_difference(neigh + off[clique[0]], cap, clique, (size_t)depth, star);  /* clique is ascending: a sorted set */
        for (int i = 1; i < depth; i++) {
            size_t di = (size_t)(off[clique[i] + 1] - off[clique[i]]);
            int32_t *t2 = (int32_t *)malloc((di + 1) * sizeof(int32_t));
            size_t nt = gmso_difference(neigh + off[clique[i]], di, clique, (size_t)depth, t2);
            ns = gmso_intersect(star, ns, t2, nt, tmp);
            memcpy(star, tmp, ns * sizeof(int32_t));
            free(t2);
        }
        *count += 1;
        *members += ns;
        free(star);
        free(tmp);
        return;
    }
    int32_t *cur = (int32_t *)malloc((ni + 1) * sizeof(int32_t));
    for (size_t i = 0; i < ni; i++) {
        const int32_t vi = isect[i];
        size_t nc = gmso_intersect(isect, ni, neigh + off[vi], (size_t)(off[vi + 1] - off[vi]), cur);
        if (vi <= clique[depth - 1]) continue;  /* curClique is built ascending: its last member is its largest */
        clique[depth] = vi;
        kcstar_step(off, neigh, k - 1, clique, depth + 1, cur, nc, count, members);
    }
    free(cur);
}
void gmso_kclique_star_count(int64_t n, const int64_t *off, const int32_t *neigh, int k, int threads, uint64_t *count_out, uint64_t *members_out) {
    uint64_t count = 0, members = 0;
    int nt = pick_threads(threads);
    (void)nt;
#pragma omp parallel for reduction(+ : count, members) schedule(dynamic, 64) num_threads(nt)
    for (int64_t u = 0; u < n; u++) {
        int32_t clique[64];
        clique[0] = (int32_t)u;
        kcstar_step(off, neigh, k - 1, clique, 1, neigh + off[u], (size_t)(off[u + 1] - off[u]), &count, &members);
    }
    *count_out = count;
    *members_out = members;
}

/* ------------------------------------------------------------------ Bron–Kerbosch */

typedef struct { const int64_t *off; const int32_t *neigh; uint64_t count; } bk_ctx;

#define ROW(c, x) ((c)->neigh + (c)->off[(x)])
#define DEG(c, x) ((size_t)((c)->off[(x) + 1] - (c)->off[(x)]))

/* tomita.h:12-40 — first argmax over cand, then fini, strict '>' */
static int32_t bk_pivot(const bk_ctx *c, const int32_t *cand, size_t nc, const int32_t *fini, size_t nf) {
    int32_t pivot = cand[0];
    /* NodeId maxDeg in the reference (int32); counts never exceed |cand| so no narrowing issue */
    int32_t maxdeg = (int32_t)gmso_intersect_count(cand, nc, ROW(c, pivot), DEG(c, pivot));
    for (size_t i = 1; i < nc; i++) {
        int32_t d = (int32_t)gmso_intersect_count(cand, nc, ROW(c, cand[i]), DEG(c, cand[i]));
        if (d > maxdeg) { pivot = cand[i]; maxdeg = d; }
    }
    for (size_t i = 0; i < nf; i++) {
        int32_t d = (int32_t)gmso_intersect_count(cand, nc, ROW(c, fini[i]), DEG(c, fini[i]));
        if (d > maxdeg) { pivot = fini[i]; maxdeg = d; }
    }
    return pivot;
}

static size_t sorted_insert(int32_t *s, size_t n, int32_t x) { /* sorted_set.h:127-138 */
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t m = (lo + hi) / 2; if (s[m] < x) lo = m + 1; else hi = m; }
    if (lo < n && s[lo] == x) return n;
    memmove(s + lo + 1, s + lo, (n - lo) * sizeof(int32_t));
    s[lo] = x;
    return n + 1;
}
static size_t sorted_remove(int32_t *s, size_t n, int32_t x) { /* sorted_set.h:207-216 */
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t m = (lo + hi) / 2; if (s[m] < x) lo = m + 1; else hi = m; }
    if (lo < n && s[lo] == x) { memmove(s + lo, s + lo + 1, (n - lo - 1) * sizeof(int32_t)); return n - 1; }
    return n;
}

/* tomita.h:51-86.  cand is mutated in place; fini must have capacity nf + nc. */
static void bk_expand(bk_ctx *c, int32_t *cand, size_t nc, int32_t *fini, size_t nf) {
    if (nc == 0) {
        if (nf == 0) c->count++;
        return;
    }
    int32_t pivot = bk_pivot(c, cand, nc, fini, nf);
    size_t cap_f = nf + nc;
    int32_t *mem = (int32_t *)malloc((nc + nc + cap_f) * sizeof(int32_t));
    int32_t *extu = mem, *cand_new = mem + nc, *fini_new = mem + 2 * nc;
    size_t ne = gmso_difference(cand, nc, ROW(c, pivot), DEG(c, pivot), extu);
    for (size_t i = 0; i < ne; i++) {
        int32_t q = extu[i];
        size_t ncn = gmso_intersect(cand, nc, ROW(c, q), DEG(c, q), cand_new);
        size_t nfn = gmso_intersect(fini, nf, ROW(c, q), DEG(c, q), fini_new);
        /* fini_new has room for nfn + ncn: nfn <= nf and ncn <= nc */
        bk_expand(c, cand_new, ncn, fini_new, nfn);
        nc = sorted_remove(cand, nc, q);
        nf = sorted_insert(fini, nf, q);
    }
    free(mem);
}

/* eppsteinPAR.h:18-53 */
uint64_t gmso_bk_count(int64_t n, const int64_t *off, const int32_t *neigh, const int32_t *rank, int threads) {
    uint64_t total = 0;
    int nt = pick_threads(threads);
    (void)nt;
#pragma omp parallel for schedule(dynamic) reduction(+ : total) num_threads(nt)
    for (int64_t v = 0; v < n; v++) {
        size_t d = (size_t)(off[v + 1] - off[v]);
        int32_t *cand = (int32_t *)malloc((2 * d + 1) * sizeof(int32_t));
        int32_t *fini = cand + d;
        size_t nc = 0, nf = 0;
        for (size_t i = 0; i < d; i++) { /* rows are sorted, so appending keeps cand/fini sorted */
            int32_t w = neigh[off[v] + (int64_t)i];
            if (rank[w] > rank[v]) cand[nc++] = w; else fini[nf++] = w;
        }
        /* fini needs capacity nf + nc: move it to its own buffer of that size */
        int32_t *fbuf = (int32_t *)malloc((nf + nc + 1) * sizeof(int32_t));
        memcpy(fbuf, fini, nf * sizeof(int32_t));
        bk_ctx c = {off, neigh, 0};
        bk_expand(&c, cand, nc, fbuf, nf);
        total += c.count;
        free(fbuf);
        free(cand);
    }
    return total;
}

typedef struct { int64_t deg; int32_t id; } deg_id;
static int cmp_deg_id(const void *p, const void *q) {
    const deg_id *a = (const deg_id *)p, *b = (const deg_id *)q;
    if (a->deg != b->deg) return a->deg < b->deg ? -1 : 1;
    return (a->id > b->id) - (a->id < b->id);
}
/* degree.h:16-62 — ascending (degree, id); rank[v] = position */
void gmso_degree_rank(int64_t n, const int64_t *off, int32_t *rank) {
    deg_id *t = (deg_id *)malloc((size_t)(n ? n : 1) * sizeof(deg_id));
    for (int64_t v = 0; v < n; v++) { t[v].deg = off[v + 1] - off[v]; t[v].id = (int32_t)v; }
    qsort(t, (size_t)n, sizeof(deg_id), cmp_deg_id);
    for (int64_t i = 0; i < n; i++) rank[t[i].id] = (int32_t)i;
    free(t);
}

/* ------------------------------------------------------------------ vertex similarity */

/* vertex_similarity.h:30-222 */
double gmso_vertex_similarity(int metric, int64_t n, const int64_t *off, const int32_t *neigh, int32_t a, int32_t b) {
    (void)n;
    const int32_t *ra = neigh + off[a], *rb = neigh + off[b];
    const size_t la = (size_t)(off[a + 1] - off[a]), lb = (size_t)(off[b + 1] - off[b]);
    if (metric == 2 || metric == 3) { /* iterate the intersection in ascending order, as the reference's range-for does */
        double sum = 0;
        size_t i = 0, j = 0;
        while (i < la && j < lb) {
            if (ra[i] < rb[j]) i++;
            else if (rb[j] < ra[i]) j++;
            else {
                double count = (double)(off[ra[i] + 1] - off[ra[i]]);
                sum += metric == 2 ? 1. / log(count) : 1.0 / count;
                i++; j++;
            }
        }
        return sum;
    }
    double count = (double)gmso_intersect_count(ra, la, rb, lb);
    switch (metric) {
        case 0: return (la == 0 && lb == 0) ? 1.0 : count / ((double)(la + lb) + count);
        case 1: return count / (double)(la < lb ? la : lb);
        case 4: return count;
        case 5: return (double)gmso_union_count(ra, la, rb, lb);
        default: return (double)(la * lb);
    }
}

/* ------------------------------------------------------------------ orderings that consume intersect_count */

typedef struct { int32_t deg; int32_t id; } adg_key;
static int cmp_degid_asc(const void *p, const void *q) {
    const adg_key *a = (const adg_key *)p, *b = (const adg_key *)q;
    if (a->deg != b->deg) return a->deg < b->deg ? -1 : 1;
    return a->id < b->id ? -1 : (a->id > b->id);
}

/* degeneracy_approx_set.h:14-86 with boundary_function::averageDegree (boundary_function.h:14-23). */
int32_t gmso_adg_rank(int64_t n, const int64_t *off, const int32_t *neigh, double epsilon, int32_t *rank, int32_t *round_of,
                      int32_t *deg_at) {
    int32_t *deg = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));   /* degreeCounter (:22) */
    int32_t *work = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));  /* vArray (:23), [start, n) = remaining */
    adg_key *batch = (adg_key *)malloc(sizeof(adg_key) * (size_t)(n > 0 ? n : 1));
    int32_t *x = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));     /* the batch as a sorted set X (:62) */
    for (int64_t i = 0; i < n; i++) { deg[i] = (int32_t)(off[i + 1] - off[i]); work[i] = (int32_t)i; }
    int64_t start = 0, counter = 0;
    int32_t rounds = 0;
    while (counter < n) {
        const int64_t remaining = n - start;
        double res = 0;  /* boundary_function.h:16-22: sum in double, (1+eps)*(res/size), truncated to unsigned (:39) */
        for (int64_t i = start; i < n; i++) res += deg[work[i]];
        const unsigned border = (unsigned)((1 + epsilon) * (res / (double)remaining));
        /* partition (:41-51): vertices with degree <= border to the front, then sort them by degree (:55-59); ties by id here */
        int64_t mid = 0;
        int64_t keep = n;  /* stable two-way split into batch[] and the tail of work[] */
        for (int64_t i = n - 1; i >= start; i--) {
            const int32_t v = work[i];
            if ((unsigned)deg[v] <= border) { batch[mid].deg = deg[v]; batch[mid].id = v; mid++; }
            else work[--keep] = v;
        }
        qsort(batch, (size_t)mid, sizeof(adg_key), cmp_degid_asc);
        for (int64_t i = 0; i < mid; i++) {
            const int32_t v = batch[i].id;
            rank[v] = (int32_t)(counter + i);  /* rank format (:68-69) */
            if (round_of) round_of[v] = rounds;
            if (deg_at) deg_at[v] = batch[i].deg;
            x[i] = v;
        }
        qsort(x, (size_t)mid, sizeof(int32_t), cmp_i32);  /* Set X(start_index, mid) (:62) */
        /* PULL update (:74-79): every remaining vertex loses |N(v) ∩ X| */
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t i = keep; i < n; i++) {
            const int32_t v = work[i];
            deg[v] -= (int32_t)gmso_intersect_count(neigh + off[v], (size_t)(off[v + 1] - off[v]), x, (size_t)mid);
        }
        start = keep;
        counter += mid;
        rounds++;
        (void)remaining;
    }
    free(deg); free(work); free(batch); free(x);
    return rounds;
}

typedef struct { int64_t cnt; int32_t id; } cnt_id;
static int cmp_cntid_asc(const void *p, const void *q) {
    const cnt_id *a = (const cnt_id *)p, *b = (const cnt_id *)q;
    if (a->cnt != b->cnt) return a->cnt < b->cnt ? -1 : 1;
    return a->id < b->id ? -1 : (a->id > b->id);
}
/* preprocessing/parallel/triangle_count.h:11-30 */
void gmso_tc_ordering(int64_t n, const int64_t *off, const int32_t *neigh, int threads, int32_t *ordering) {
    int64_t *counts = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
    cnt_id *keys = (cnt_id *)malloc(sizeof(cnt_id) * (size_t)(n > 0 ? n : 1));
    gmso_tc_vertex_count2_once(n, off, neigh, threads, counts);  /* CountFn default (:14) */
    for (int64_t u = 0; u < n; u++) { keys[u].cnt = counts[u]; keys[u].id = (int32_t)u; }
    qsort(keys, (size_t)n, sizeof(cnt_id), cmp_cntid_asc);
    for (int64_t i = 0; i < n; i++) ordering[i] = keys[i].id;
    free(counts); free(keys);
}
