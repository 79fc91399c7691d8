/* gmsx.h — C-ABI of the MI355X-native set-intersection / subgraph-enumeration backend for GMS.
 *
 * Plain C: opaque handles, raw pointers and sizes, int status codes.  No exceptions, no exit(),
 * no torch / STL types cross this boundary.  This is the boundary a GMS maintainer binds to; the
 * C++ adaptor that plugs it under the reference's template concept is include/gmsx_set_graph.hpp
 * and the reference-side glue is shown in INTEGRATION.md.
 *
 * Every entry point names the reference interface it replaces (paths relative to the spcl/gms
 * tree, i.e. /root/reference).
 *
 * Conventions
 *   - status: 0 = GMSX_OK, negative = error (gmsx_strerror()).  Outputs are written only on success.
 *   - host buffers stay caller-owned; the library copies what it keeps.
 *   - graphs are symmetric CSR: int64 offsets[n+1], int32 neigh[offsets[n]], each row sorted
 *     ascending, duplicate- and loop-free (exactly what the reference loader yields,
 *     gms/third_party/gapbs/builder.h:206-235); gmsx_csr_* builds such graphs, and
 *     gmsx_graph_upload() verifies the invariant on the device unless told not to.
 *   - one device per process (one process per GPU); handles are immutable after creation and may be
 *     used from one host thread at a time, like the reference's main-thread kernel invocation
 *     (gms/common/benchmark.h:111-118).
 */
#ifndef GMSX_H
#define GMSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMSX_VERSION 320

/* ---- status codes ---- */
enum {
    GMSX_OK = 0,
    GMSX_ERR_INVALID = -1,      /* bad argument (NULL, negative size, k out of range, …) */
    GMSX_ERR_NOMEM = -2,        /* host allocation failed */
    GMSX_ERR_IO = -3,           /* file could not be opened / parsed (reference: exit(-2)/(-6), reader.h:223-229,270-273) */
    GMSX_ERR_FORMAT = -4,       /* unknown suffix or malformed file (reference: exit(-3), reader.h:243-245) */
    GMSX_ERR_DIRECTED = -5,     /* a directed graph where the path needs an undirected one (reference: exit(100), cli/cli.h:166-171) */
    GMSX_ERR_NO_DEVICE = -6,    /* no HIP device / gmsx_init() not called / HIP runtime error */
    GMSX_ERR_DEVICE_MEM = -7,   /* hipMalloc failed */
    GMSX_ERR_NOT_CANONICAL = -8,/* CSR rows not sorted / not loop-free / not symmetric where required */
    GMSX_ERR_OVERFLOW = -9,     /* ids do not fit int32 (reference: exit(-31), generator.h:41-48) */
    GMSX_ERR_UNSUPPORTED = -10, /* valid request outside what this build implements */
    GMSX_ERR_KERNEL = -11,      /* a kernel launch or device synchronisation failed */
    GMSX_ERR_COMM = -12,        /* librccl missing or an RCCL call failed */
    GMSX_ERR_TIMEOUT = -13      /* a peer of the communicator did not arrive within GMSX_COMM_TIMEOUT_S seconds (gmsx_comm_init / gmsx_comm_allreduce_u64) */
};
const char *gmsx_strerror(int status);
int gmsx_version(void);

/* =====================================================================================
 * Host graph substrate  (replaces gms/third_party/gapbs: generator.h, builder.h, reader.h, writer.h)
 * ===================================================================================== */
typedef struct gmsx_csr gmsx_csr; /* host-resident CSR, replaces CSRGraph (gapbs/graph.h:93-374) */

enum { GMSX_GEN_KRONECKER = 0, GMSX_GEN_UNIFORM = 1 }; /* "-g kronecker|uniform" (cli/cli.h:111-114) */
enum {
    GMSX_RELABEL_NEVER = 0,
    GMSX_RELABEL_AUTO = 1,  /* relabel by decreasing degree iff WorthRelabelling (gapbs/benchmark.h:158-176) — what parse_and_load does (cli/cli.h:174-176) */
    GMSX_RELABEL_ALWAYS = 2
};

/* Generator::GenerateEL + Builder::MakeGraph (+ optional RelabelByDegree): bit-identical CSR to the
 * reference's "-g kronecker|uniform <scale> --deg <degree>" (generator.h:64-127, builder.h:1642-1660).
 * threads<=0: OpenMP default.  The output does not depend on the thread count. */
int gmsx_csr_generate(int generator, int scale, int degree, int relabel, int threads, gmsx_csr **out);

/* The same generator/builder pipeline with caller-chosen R-MAT quadrant probabilities (the reference hard-codes
 * A/B/C = .57/.19/.19, generator.h:82; with those values the result equals gmsx_csr_generate(KRONECKER)).  Used to
 * calibrate the skew of the Bron–Kerbosch workload. */
int gmsx_csr_generate_rmat(int scale, int degree, double a, double b, double c, int relabel, int threads, gmsx_csr **out);

/* Builder::MakeGraphFromEL + SquishGraph on a caller-supplied edge list (builder.h:279-298,237-251):
 * num_nodes<0 → max id + 1; symmetrize!=0 inserts both directions (the only mode the hot path accepts).
 * Rows come out sorted, de-duplicated and loop-free. */
int gmsx_csr_from_edges(int64_t num_nodes, int64_t num_edges, const int32_t *src, const int32_t *dst,
                        int symmetrize, int relabel, gmsx_csr **out);

/* Reader::ReadFile / ReadSerializedGraph by suffix (reader.h:220-250): ".el" text pairs (:49-56), ".wel" weighted pairs
 * (:58-66, weights dropped), ".gr" DIMACS (:68-84), ".graph" METIS (:86-142), ".mtx" Matrix Market coordinate (:146-218),
 * ".sg" binary CSR (:252-305); plus ".sgx", this library's mappable cache form of .sg (gmsx_csr_save_sgx).  Other suffixes → GMSX_ERR_FORMAT. */
int gmsx_csr_load(const char *path, int symmetrize, int relabel, gmsx_csr **out);
/* Writer::WriteSerializedGraph (writer.h:39-69). */
int gmsx_csr_save_sg(const gmsx_csr *g, const char *path);
/* The cache form of the same CSR, ".sgx": the .sg arrays behind a 64-byte header, each at a 64-byte aligned file offset (the reference's
 * .sg puts its int64 offsets at byte 17, so it cannot be used in place).  gmsx_csr_load(".sgx") MAPS the file (private, never written back)
 * instead of reading it: the ranks of a multi-GPU run that load one cache share one copy in the page cache — SURVEY §8(e)'s "replicate the
 * CSR" without N private host copies —, after the same validation as .sg (monotone offsets, ids in range).  gmsx_csr_is_mapped: 1 if the
 * handle's arrays are such a mapping. */
int gmsx_csr_save_sgx(const gmsx_csr *g, const char *path);
int gmsx_csr_is_mapped(const gmsx_csr *g);
/* Wrap (copy) caller arrays; validates monotone offsets and id range. */
int gmsx_csr_from_arrays(int64_t n, const int64_t *offsets, const int32_t *neigh, gmsx_csr **out);

int gmsx_csr_worth_relabelling(const gmsx_csr *g);                  /* gapbs/benchmark.h:158-176; 1/0 */
int gmsx_csr_relabel_by_degree(const gmsx_csr *g, gmsx_csr **out);  /* builder.h:1699-1733 */

int64_t gmsx_csr_num_nodes(const gmsx_csr *g);          /* CSRGraph::num_nodes */
int64_t gmsx_csr_num_edges(const gmsx_csr *g);          /* CSRGraph::num_edges  (= nnz/2, graph.h:186) */
int64_t gmsx_csr_num_edges_directed(const gmsx_csr *g); /* nnz */
const int64_t *gmsx_csr_offsets(const gmsx_csr *g);     /* n+1 entries; valid until gmsx_csr_free */
const int32_t *gmsx_csr_neighbors(const gmsx_csr *g);   /* nnz entries */
/* Σ_{(u,v)∈E,u<v}(d_u+d_v): element count behind the algorithmic-bytes figure. */
uint64_t gmsx_csr_merge_elements(const gmsx_csr *g);
/* FNV-1a-64 of the offsets / neighbour arrays (loader fingerprints). */
uint64_t gmsx_csr_fingerprint(const gmsx_csr *g, int which /*0 offsets, 1 neighbours*/);
void gmsx_csr_free(gmsx_csr *g);

/* Threads of the host substrate (generator, builder, relabelling: OpenMP, like the reference's loader).  n <= 0 restores
 * the runtime default.  Returns the previous maximum.  Launchers that export OMP_NUM_THREADS=1 to their workers
 * (torch.distributed.run does) would otherwise serialise graph generation: the rank that builds the graph calls this. */
int gmsx_set_host_threads(int n);

/* OPTIONS.  The library reads NO tuning from the environment (the one exception is GMSX_COMM_TIMEOUT_S below): limits, budgets, container
 * forms and diagnostics are set by explicit calls.  Every option changes HOW a result is computed — which kernel variant, how large a
 * chunk, what is printed to stderr — and never the result; the "wrong counts" A/B switches of the kernel sources exist only in
 * development builds (-DGMSX_DEV_HOOKS, tools/ab_lib.sh) and are absent from a library built by `make`.
 * name: one of the names below (without prefix), else GMSX_ERR_INVALID; value: a decimal number as a C string (< 32 characters), NULL = back
 * to the default.  Process-wide; set options before the calls they steer, not concurrently with them.  Names:
 *   diagnostics     TIMING (phase times and bin sizes on stderr), MEM_TRACE (one line per device allocation of a graph), BK_VERBOSE
 *   upload          SORT_CHUNK (vertices per row-sort range), UPLOAD_STAGED (0 = one hipMemcpyAsync per array), INIT_LAZY (1 = pin the staging
 *                   buffers at the first large upload instead of in gmsx_init)
 *   triangle count  TC_MEM_LIMIT_MB (budget of the containers: smaller = several passes), INLINE_LIMIT, TC_INLINE_FIRST, TC_TWO_SIDED,
 *                   TC_DELTA / TC_DELTA_PCT / TC_GAP12 / TC_HYBRID / TC_TAIL_DELTA (which container forms the stream rows may take),
 *                   TC_HOT_WINDOWS / TC_HOT_KB / TC_HOT_MIN (L2-window phases of the hub lists), TC_TEST_MAX_UNITS (pretend the task-list
 *                   fields are this narrow), TC_KEEP_ROWS (keep the per-vertex row descriptors for gmsx_tc_row_histogram), TC_OVERLAP,
 *                   TC_PERSIST, TC_ITEM_WGS (launch shapes)
 *   k-clique        KC_MAXD (widest pivot of the bit-matrix kernels), KC_SLAB_MB (budget of the global slabs), KC_STREAMS, KC_PIPE_ALL,
 *                   KC_STREAM_BUILD (BUILD variants), KC_REVERSE (0 = every member row streamed forward: no reverse rows; like KC_REV_* read when the lists of a graph are built, i.e. at its first k-clique call), KC_REV_MIN (edges a hub
 *                   receiver must get to take them, default 64), KC_REV_FACTOR (10 x how much cheaper in bytes the reverse side must be; default 0: every hub edge whose receiver qualifies), KC_REV_GW (8 / 16 lanes per record in the receivers' kernels),
 *                   KC_REV_TAIL (0 = tail members — rank id >= 65 535 — never hand their row to a receiver), KC_REV_TAIL_MIN (edges a tail receiver must get, default 256),
 *                   KC_TRI (0 = k = 4 keeps rectangular LDS matrices up to d+ = 1024 and the global slab beyond, instead of triangular ones up to 1472),
 *                   KC_MFMA (0 = the k = 4 count of the pivots of d+ > 512 by AND + popcount inside their BUILD kernels instead of on the matrix cores),
 *                   KC_POOL_MB (the pool those pivots' matrices pass through, default min(what the call needs, 6144): smaller = more, shorter chunks)
 *   Bron–Kerbosch   BK_MAXC (widest start vertex of the register-resident search), BK_ARENA_MB, BK_BUDGET / BK_BUDGET0 (nodes before a
 *                   search is re-split), BK_GROUPS, BK_SMALL_P, BK_SMALL_P_GROUPS, BK_RESUME_GRAB, BK_SPLIT_BUILD, BK_TINY_ROOTS,
 *                   BK_TINY_BESIDE (kernel variants) */
int gmsx_set_option(const char *name, const char *value);
void gmsx_reset_options(void);                          /* every option back to its default */
int gmsx_option_name(int index, const char **name);     /* enumerates the names: GMSX_ERR_INVALID past the last */

/* =====================================================================================
 * Device side
 * ===================================================================================== */
typedef struct gmsx_graph gmsx_graph; /* device-resident graph; replaces SetGraph<Set> (representations/graphs/set_graph.h:10-233) */

typedef struct gmsx_stats {
    double kernel_ms;          /* HIP-event time of the counting kernels of the last call (on the library's stream) */
    double setup_ms;           /* HIP-event time of per-call setup kernels (memsets, bin prep), not in kernel_ms */
    uint64_t units;            /* work units processed: intersect_count calls (edges) / root vertices */
    uint64_t alg_elements;     /* Σ(d_u+d_v) over the units of this call, 0 if n/a */
    uint64_t probes;           /* TC: id slots the oriented kernels probe (8 per list unit, 14 per byte-delta unit, 4 words per bitset
                                * unit, 4 / 6 per tail unit); BK: resume rounds; else 0 */
    int32_t launches;          /* number of kernel launches inside kernel_ms */
    int32_t reserved;
    uint64_t stream_bytes;     /* TC (oriented): algorithmic bytes of THIS formulation per call — every pivot's own containers once, plus
                                  for every oriented edge (u,v) the bytes of N+(v) in the form the kernels read: its stream rows (whole
                                  16-byte units of bitset / 16-bit list / byte-delta for the hub part, 32-bit ids / 16-bit delta for the
                                  tail part) or 4 bytes per inverted gather.  No cache is assumed: the traffic a pass would cause if
                                  nothing were ever re-used on chip.  k-clique: every pivot's own containers once + per member the containers of
                                  N+(v) the BUILD reads (bitset words or 16-bit list, tail ids; one 4-byte gather per pair for d+ <= 32) — or, for a
                                  hub member whose row the receivers' pass produced (round 6, kclique.hip: reverse rows), the 2 i bytes of the pivot's
                                  prefix below it + its 16-byte record + its row of ceil(i/32) words written and read back — + the
                                  slab matrices of pivots wider than 1024, written and read once.  Bron-Kerbosch: per start vertex the oriented
                                  rows of all its neighbours (what the builds walk) + Cadj | XT of the start vertices built in the arena, once
                                  + one Cadj row (c / 32 words) per search-tree node — the operand of cand.intersect(N(q)), tomita.h:51-70.
                                  0 for the other entry points. */
} gmsx_stats;

/* Bind this process to one HIP device.  device<0 → current device.  Calling it again with the same device is a no-op (and makes
 * the device current for the calling host thread); a different device → GMSX_ERR_UNSUPPORTED (one device per process). */
int gmsx_init(int device);
/* Use a caller-owned HIP stream (hipStream_t passed as void*) for all subsequent launches; NULL → library stream. */
int gmsx_set_stream(void *hip_stream);
int gmsx_device_info(char *name, size_t name_len, int *compute_units, int64_t *hbm_bytes);
/* Diagnostics: the rate (GB/s) a read-only in-order stream over a `bytes`-sized buffer (choose it far above the 256 MB Infinity Cache) reaches on
 * the bound device, 16-byte loads, `iterations` sweeps timed with HIP events — the measured ceiling bench.py quotes its roofline fractions
 * against beside the 8 TB/s specification peak.  No reference counterpart. */
int gmsx_hbm_read_probe(int64_t bytes, int iterations, double *gbps);

enum {
    GMSX_UPLOAD_DEFAULT = 0,
    GMSX_UPLOAD_TRUSTED = 1, /* skip the device-side check of the canonical-row invariant.  The check (default): every row strictly
                                ascending, ids in range, no self loops — exact — and the arc set equal to its transpose, decided by two
                                keyed 64-bit multiset hashes of both (keys drawn per process): an asymmetric input passes with
                                probability 2^-128, a symmetric one never fails.  With TRUSTED the caller GUARANTEES the invariant (only
                                out-of-range ids are still refused): on a row with a repeated neighbour, a self loop or a missing reverse
                                arc the counts are undefined — the triangle kernels, for one, maintain a pivot's LDS bitmap by XOR, so a
                                repeated id cancels itself */
    GMSX_UPLOAD_FOR_TC = 2   /* also build the triangle-count containers (stream rows, inline rows, task lists: ~5x the CSR) now, inside the
                                upload — what a triangle-count harness wants in its untimed "GraphExec buildTime" (common/benchmark.h:
                                105-109).  Without it they are built by gmsx_graph_prepare or by the first gmsx_tc_* call, and a k-clique /
                                Bron–Kerbosch user never pays for them (the reference's k-clique harness times its SGraph build,
                                k_clique_count_set_based.h:22) */
    /* bits 8..23 (test hook): if non-zero, the hub-container id range is [0, value) instead of [0, 65535), so that
       small graphs exercise the 32-bit tail containers; results never depend on it */
};
#define GMSX_UPLOAD_HUB_LIMIT(x) ((uint32_t)(x) << 8)
/* SetGraph::FromCGraph (set_graph.h:86-89,152-181): copies the CSR into HBM and builds the device-side
 * set representations (degree-oriented DAG rows; the analogue of the per-row SortedSet / RoaringSet
 * construction, sorted_set.h:64-66 / roaring_set.h:49-54).  Timed by callers as "GraphExec buildTime"
 * like the reference (common/benchmark.h:105-109). */
int gmsx_graph_upload(int64_t n, const int64_t *offsets, const int32_t *neigh, uint32_t flags, gmsx_graph **out);
int gmsx_graph_upload_csr(const gmsx_csr *g, uint32_t flags, gmsx_graph **out);
/* The same for ONE rank of a multi-GPU run: the CSR and the oriented containers are complete on every rank (any row may be
 * streamed; k-clique / Bron-Kerbosch shards of any (part, nparts) still work), but the triangle-count containers — task lists and inline
 * rows, ~4/5 of the device bytes — are built for the pivots of shard `part` of `nparts` only, so their memory and build time shrink with
 * the number of ranks.  gmsx_tc_partial on such a graph accepts exactly (part, nparts) (GMSX_TC_FULL any).  part = 0, nparts = 1 is
 * gmsx_graph_upload. */
int gmsx_graph_upload_shard(int64_t n, const int64_t *offsets, const int32_t *neigh, uint32_t flags, int part, int nparts, gmsx_graph **out);
int gmsx_graph_upload_csr_shard(const gmsx_csr *g, uint32_t flags, int part, int nparts, gmsx_graph **out);
/* Builds optional containers of an uploaded graph ahead of their first use (so that a caller can time or place the cost). */
enum { GMSX_PREPARE_TC = 1 /* the triangle-count containers, see GMSX_UPLOAD_FOR_TC */ };
int gmsx_graph_prepare(gmsx_graph *g, uint32_t what);
/* 0 = the triangle-count containers are not built yet; 1 = they hold every pivot (the normal case); k > 1 = they did not fit the device:
 * 1/k of the pivots is resident at a time and every gmsx_tc_* call walks k passes, rebuilding between them (slower, never refused). */
int gmsx_graph_tc_passes(const gmsx_graph *g);
int gmsx_graph_free(gmsx_graph *g);
int64_t gmsx_graph_num_nodes(const gmsx_graph *g);   /* SetGraph::num_nodes, set_graph.h:115-118 */
int64_t gmsx_graph_num_edges(const gmsx_graph *g);   /* undirected edges m */
int64_t gmsx_graph_device_bytes(const gmsx_graph *g);
int32_t gmsx_graph_max_out_degree(const gmsx_graph *g); /* max d+ of the degree-oriented DAG */

/* ---- triangle counting: gms/algorithms/set_based/triangle_count ---- */
enum {
    GMSX_TC_AUTO = 0,
    GMSX_TC_ORIENTED = 1, /* each edge intersects the degree-oriented rows N+(u) ∩ N+(v): every triangle met once */
    GMSX_TC_FULL = 2      /* the reference's formulation verbatim: every edge u<v intersects the FULL rows, total/3
                             (parallel/total.h:13-23); all m full-row intersect_count calls are executed on the device */
};
/* Par::count_total / Seq::count_total (parallel/total.h:7-24, sequential/total.h:7-23): number of triangles. */
int gmsx_tc_total(const gmsx_graph *g, int algo, uint64_t *triangles, gmsx_stats *stats);
/* Shard `part` of `nparts` (cost-balanced, disjoint, covering): returns the UN-divided partial sum so that
 * Σ_parts partial / gmsx_tc_divisor(algo) = triangles.  This is what each rank computes before the single
 * all-reduce of SURVEY §8(e) (the OpenMP reduction(+:total) of parallel/total.h:12). */
int gmsx_tc_partial(const gmsx_graph *g, int algo, int part, int nparts, uint64_t *partial, gmsx_stats *stats);
int gmsx_tc_divisor(int algo); /* 1 for ORIENTED/AUTO, 3 for FULL */
/* Diagnostics: the algorithmic stream bytes of one oriented pass (gmsx_stats.stream_bytes) split by what is read — out[0..2]: the hub
 * stream rows named by the work items' entries, by form (16-bit list, bitset, byte-delta); out[3..4]: the tail stream rows (32-bit
 * list, 16-bit delta); out[5]: the entries themselves; out[6]: the pivots' own containers, once per work item; out[7]: the part of
 * out[0] + out[3] that is inline rows (ids handed over by light pivots); out[8..10]: the light-pivot kernel — hub rows, tail rows of
 * the far light members it streams, and its own lists + descriptors; out[11..14]: counts — entries, inline entries, work items,
 * members streamed by the light-pivot kernel; out[15..20]: 0.  out[0..6] + out[8..10] = stream_bytes.  21 values, host. */
int gmsx_tc_stream_breakdown(const gmsx_graph *g, uint64_t *out21);
/* Diagnostics: the stream rows named by the work items' entries as a histogram over row length in 16-byte units — out[(cls*24 +
 * bin)*2] rows, [+1] units; cls 0..4 = hub rows as list / bitset / byte-delta, tail rows as list / delta; bin = 1 … 16 units exactly,
 * then 17-32, 33-64, … 1025+; out[240..243]: entries, inline entries, work items, bytes of the pivots' own containers;
 * out[248..251]: Σ over the oriented edges (u,v) of heavy then light pivots u of the stream units of v's rows and of min(units of
 * u's rows, units of v's rows); out[252]: the same sum for heavy u with u's rows cut at v's id (estimate) — out[248..252] are 0 unless the
 * graph was built with GMSX_TC_KEEP_ROWS=1 (the per-vertex row descriptors they read are freed after the build).  256 values, host. */
int gmsx_tc_row_histogram(const gmsx_graph *g, uint64_t *out256);
/* Diagnostics (round 5, the "stream a row once against several staged pivots" question): if the heavy pivots were processed in batches of
 * `batch` consecutive positions of the d+ order, each distinct stream row of a batch loaded once — out[0] entries of the hub work items,
 * out[1] the 16-byte units they stream per pass today, out[2] distinct (batch, row) pairs, out[3] the units a batched pass would stream
 * (every row once per batch, at the longest cut any pivot of the batch reads), out[4] hub work items.  batch = 1 shows what rows repeat
 * inside one pivot's own list (inline chunks never do).  8 values, host. */
int gmsx_tc_comembership(const gmsx_graph *g, int batch, uint64_t *out8);
/* Par::vertex_count2 / vertex_count2_once (parallel/vertex.h:14-49): counts[u] = Σ_{v∈N(u)} |N(u)∩N(v)| (= 2·triangles at u),
 * indexed by the vertex ids of the uploaded CSR.  Runs on the k = 3 bit-matrix kernels (one atomic per pivot member); pivots with
 * d+ > 8192 — none on RMAT up to scale 27 (max d+ = 3855) — run on the generic list recursion (one atomic per triangle, those pivots only). */
int gmsx_tc_vertex_count2(const gmsx_graph *g, int64_t *counts /* n, host */, gmsx_stats *stats);

/* ---- generic batched Set::intersect_count (sorted_set.h:176-182 / roaring_set.h:144-152) over graph rows:
 * out[i] = |N(u[i]) ∩ N(v[i])| on the full rows.  u, v, out are host arrays. */
int gmsx_intersect_count_batch(const gmsx_graph *g, int64_t n_pairs, const int32_t *u, const int32_t *v,
                               uint32_t *out, gmsx_stats *stats);
/* Set::intersect / Set::difference (sorted_set.h:160-197 -> sorted_set_operations.h:16-42, 73-99) MATERIALISED for a batch of vertex pairs —
 * result i = N(u[i]) ∩ N(v[i]) or N(u[i]) \ N(v[i]), ascending vertex ids of the uploaded CSR, CSR-shaped: out_offsets[n_pairs + 1] (host, always
 * filled) and out_ids (host).  out_ids == NULL is the sizing call (count pass + scan only); with out_ids the result must fit out_capacity ids, else
 * GMSX_ERR_INVALID (out_offsets[n_pairs] says how many it needs).  Two passes of one wave per pair (count, scan, fill): the base of the listing
 * consumers (BK without BK_COUNT, tomita.h:75-84; k-clique-star output, k_clique_star_list/parallel/output.h:14-68). */
enum { GMSX_SETOP_INTERSECT = 0, GMSX_SETOP_DIFFERENCE = 1 };
int gmsx_set_op_batch(const gmsx_graph *g, int op, int64_t n_pairs, const int32_t *u, const int32_t *v, int64_t *out_offsets /* n_pairs + 1 */,
                      int32_t *out_ids /* may be NULL */, int64_t out_capacity, gmsx_stats *stats);

/* ---- vertex similarity over graph rows: GMS::VertexSim::vertex_similarity<Metric> (gms/algorithms/set_based/vertex_similarity/
 * vertex_similarity.h:30-222), the scores behind the reference's link-prediction driver.  out[i] = metric(N(u[i]), N(v[i])) in
 * double precision, formulas as the reference writes them (its Jaccard divides by |A|+|B|+|A∩B|, :34-35).  The count-based
 * metrics are bit-identical to the reference; Adamic-Adar / resource-allocation sum their terms in a different order and use
 * the device log(): |rel. error| <= 1e-12. */
enum {
    GMSX_SIM_JACCARD = 0, GMSX_SIM_OVERLAP = 1, GMSX_SIM_ADAMIC_ADAR = 2, GMSX_SIM_RESOURCE = 3,
    GMSX_SIM_COMMON_NEIGHBORS = 4, GMSX_SIM_TOTAL_NEIGHBORS = 5, GMSX_SIM_PREF_ATTACHMENT = 6
};
int gmsx_vertex_similarity_batch(const gmsx_graph *g, int metric, int64_t n_pairs, const int32_t *u, const int32_t *v,
                                 double *out, gmsx_stats *stats);

/* ---- k-clique counting: CliqueCount (k_clique_count/k_clique_count_set_based.h:19-31).
 * *ordered_count = the reference's return value k!·C_k (mod 2^64, like size_t); *cliques = C_k (may be NULL).
 * k = 2 … 64.  The fast path — one bit-matrix per pivot — holds oriented out-degrees d+ <= 8192 for k <= 4, <= 4096 for
 * 5 <= k <= 10 (RMAT scale 26: max d+ = 3004); wider pivots and k > 10 run on a generic list recursion on the device (an order of
 * magnitude slower per pivot, same result), so no request fails for its size.  k > 64 → GMSX_ERR_UNSUPPORTED. */
int gmsx_kclique_count(const gmsx_graph *g, int k, uint64_t *ordered_count, uint64_t *cliques, gmsx_stats *stats);
int gmsx_kclique_partial(const gmsx_graph *g, int k, int part, int nparts, uint64_t *cliques_partial, gmsx_stats *stats);

/* ---- k-clique-star, COUNT mode: KCliqueStar::Par::CliqueStar<SGraph, OutputMode::Count>
 * (set_based/k_clique_star_list/parallel/recursive.h:19-35 over sequential/recursive.h:31-71).  The reference lists, for every k-clique
 * (reached once, members ascending), the clique and its STAR — the common neighbours of all its members outside it.
 * *stars = the number of (clique, star) pairs = C_k — what the reference prints as "total k-cliques" and what `output.size()` is;
 * *star_members = the total cardinality of the stars = (k+1)·C_{k+1}: every (k+1)-clique puts each of its members into the star of the
 * k-clique of the others (may be NULL: the (k+1)-clique pass is then skipped).  Both from the k-clique kernels above; the listing itself
 * is not produced on the device.  k = 1 … 63. */
int gmsx_kclique_star_count(const gmsx_graph *g, int k, uint64_t *stars, uint64_t *star_members, gmsx_stats *stats);

/* ---- Bron–Kerbosch maximal-clique count: BkEppsteinPar::mceBench with -DBK_COUNT
 * (maximal_clique_enum/parallel/eppsteinPAR.h:18-53 over sequential/tomita.h:12-86).
 * rank: n entries in rank format (host), or NULL.  The number of maximal cliques does not depend on the order the start
 * vertices split their neighbourhoods by (eppsteinPAR.h:39-45), so the device always splits by its own degree rank (the
 * order its containers are built for); a non-NULL `rank` is only VALIDATED — it must be a permutation of 0..n-1, else
 * GMSX_ERR_INVALID — the way the reference's drivers hand a rank vector from the preprocessing step to mceBench
 * (maximal_clique_enum_bron_kerbosch.cc:36-56).  gmsx_adg_rank() below is that preprocessing step on the device.
 * Width: start vertices with up to 16384 candidates (d+ in the device's degree rank; RMAT scale 27: max d+ = 3855) run on the
 * register-resident search kernels (1, 2, 4 or 8 words per lane, re-split across waves when a search outgrows its node budget);
 * wider ones run on a memory-resident search — one wave per start vertex, every level's sets in a global slab, an order of
 * magnitude slower per node, same count — so no graph is refused for the width of a neighbourhood. */
int gmsx_bk_count(const gmsx_graph *g, const int32_t *rank, uint64_t *maximal_cliques, gmsx_stats *stats);
int gmsx_bk_partial(const gmsx_graph *g, const int32_t *rank, int part, int nparts, uint64_t *partial, gmsx_stats *stats);

/* ---- vertex orderings that consume the path's operators ----
 * PpParallel::getDegeneracyOrderingApproxSGraph<boundary_function::averageDegree> (preprocessing/parallel/
 * degeneracy_approx_set.h:14-86, boundary_function.h:14-23): peel in rounds — every remaining vertex whose remaining degree is
 * <= (unsigned)((1+epsilon) * average remaining degree) leaves in this round; the degrees of the others drop by
 * |N(v) ∩ X| (the intersect_count of :77).  out[n] (host): rank_format != 0 → out[v] = position of v (what BK consumes,
 * eppsteinPAR.h:41), else out[i] = i-th vertex.  *rounds (may be NULL) = number of peeling rounds.
 * Ties: inside a round the vertices are ordered by (remaining degree, vertex id).  The reference leaves ties to
 * __gnu_parallel::partition / sort (:41-59), i.e. to the thread count; every order it can produce has the same rounds and the
 * same remaining-degree sequence as this one. */
int gmsx_adg_rank(const gmsx_graph *g, double epsilon, int rank_format, int32_t *out, int32_t *rounds, gmsx_stats *stats);
/* PpParallel::triangleCountOrdering (preprocessing/parallel/triangle_count.h:11-30): ordering[i] = i-th vertex by increasing
 * per-vertex count (the counts of gmsx_tc_vertex_count2); ties by vertex id (the reference's std::sort leaves them open).
 * (As shipped, the reference's default CountFn vertex_count2_once adds into an uninitialised pvector, triangle_count.h:19 +
 * parallel/vertex.h:42-46, so its own output depends on heap contents; this is the ordering by the well-defined counts, which
 * the reference produces with CountFn = Par::vertex_count2 — that instantiation is what the goldens were generated with.) */
int gmsx_tc_ordering(const gmsx_graph *g, int32_t *ordering /* n, host */, gmsx_stats *stats);

/* ---- the one collective of the path: the OpenMP reduction(+:total) of parallel/total.h:12,
 * k_clique_count_set_based.h:25 and the BK_CLIQUE_COUNTER atomic (tomita.h:76-77) across GPUs = ONE all-reduce of a u64 over
 * RCCL/xGMI.  One process per GPU; the 128-byte id is created on one rank (gmsx_comm_unique_id) and handed to the others by the
 * launcher (file, environment, MPI, torch.distributed store — the library does not care).  librccl is loaded on first use. */
typedef struct gmsx_comm gmsx_comm;
#define GMSX_COMM_ID_BYTES 128
int gmsx_comm_unique_id(void *id /* GMSX_COMM_ID_BYTES, out */);
/* ncclCommInitRank on the bound device, with a BOUNDED wait: the call runs on a helper thread and the caller waits at most GMSX_COMM_TIMEOUT_S
 * seconds (environment, default 180) for it — a peer that died or never started makes ncclCommInitRank wait for good.  GMSX_ERR_TIMEOUT then:
 * the helper thread stays parked inside RCCL, so the caller should report the failure and LEAVE THE PROCESS (its launcher — torch.distributed.run,
 * gmsx_driver's supervisor — takes the other ranks down).  The wait for the result of every gmsx_comm_allreduce_u64 is bounded the same way. */
int gmsx_comm_init(int rank, int nranks, const void *id /* GMSX_COMM_ID_BYTES */, gmsx_comm **out);
int gmsx_comm_allreduce_u64(gmsx_comm *c, uint64_t *value /* in: this rank's partial, out: the sum */); /* ncclAllReduce(count=1, ncclUint64, ncclSum) */
int gmsx_comm_rank(const gmsx_comm *c);
int gmsx_comm_size(const gmsx_comm *c);
int gmsx_comm_finalize(gmsx_comm *c);

#ifdef __cplusplus
}
#endif
#endif /* GMSX_H */
