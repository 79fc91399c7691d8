"""GPU parity tests of the HIP triangle-count path, called through the C-ABI (include/gmsx.h) and compared with the
oracle on the same inputs (bit-exact: integer counts), with the committed reference goldens, and — at sizes the
oracle cannot finish — with reference goldens / size-independent properties."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, edges_to_csr, host_graph, load_golden

pytestmark = pytest.mark.gpu
GRAPHS = load_golden("graphs.json")


def upload(gpu, csr, **kw):
    return gpu.DeviceGraph.from_csr(csr, **kw)


@pytest.mark.parametrize("key", [k for k, v in GRAPHS.items() if "triangles" in v and v["scale"] <= 18])
def test_tc_equals_reference_golden_and_oracle(gpu, oracle, key):
    rec = GRAPHS[key]
    csr = host_graph(gpu, rec["generator"], rec["scale"], rec["degree"], rec["relabel"])
    g = upload(gpu, csr)
    assert (g.num_nodes, g.num_edges) == (rec["n"], rec["m"])
    t, st = g.tc_total(stats=True)
    assert t == rec["triangles"]
    if rec["scale"] <= 14:
        assert t == oracle.tc_total(csr.offsets(), csr.neighbors())
    assert st["units"] == rec["m"] and st["alg_elements"] == csr.merge_elements() and st["kernel_ms"] > 0
    g.free()


@pytest.mark.parametrize("spec", [("kronecker", 9, 3), ("kronecker", 13, 40), ("uniform", 13, 30), ("kronecker", 15, 8), ("uniform", 8, 100)])
def test_tc_other_shapes_vs_oracle(gpu, oracle, spec):
    kind, scale, deg = spec
    for relabel in (True, False):
        csr = host_graph(gpu, kind, scale, deg, relabel)
        g = upload(gpu, csr)
        assert g.tc_total() == oracle.tc_total(csr.offsets(), csr.neighbors())
        assert g.tc_total(gpu.TC_ORIENTED) == g.tc_total(gpu.TC_AUTO)
        g.free()


def test_partials_cover_the_graph_exactly_once(gpu):
    csr = host_graph(gpu, "kronecker", 15)
    g = upload(gpu, csr)
    total = g.tc_total()
    for nparts in (1, 2, 3, 8):
        parts = [g.tc_partial(p, nparts, stats=True) for p in range(nparts)]
        assert sum(p[0] for p in parts) == total * gpu.lib().gmsx_tc_divisor(gpu.TC_AUTO)
        assert sum(p[1]["units"] for p in parts) == csr.num_edges
    g.free()


def test_reference_test_graphs_and_known_answers(gpu, oracle):
    for name, rec in load_golden("testgraphs.json").items():
        csr = gpu.HostCSR.load(os.path.join(GOLDEN, "testGraphs", name))
        g = upload(gpu, csr)
        assert g.tc_total() == rec["triangles"], name
        g.free()
    ka = load_golden("known_answers.json")
    for c in ka["kclique"] + ka["bk_random"]:
        csr = edges_to_csr(gpu, c["edges"], n=c.get("n", -1))
        g = upload(gpu, csr)
        assert g.tc_total() == c["triangles"]
        g.free()


def test_edge_cases(gpu):
    # empty graph (n = 1, no edges), isolated vertices, one edge, a star (no triangles), K_n, a long path
    for edges, n, expect in [([], -1, 0), ([(0, 1)], 5, 0), ([(0, i) for i in range(1, 300)], -1, 0),
                             ([(i, i + 1) for i in range(1000)], -1, 0)]:
        g = upload(gpu, edges_to_csr(gpu, edges, n=n))
        assert g.tc_total() == expect
        g.free()
    for k in (3, 4, 65, 130, 700):  # complete graphs: C(k,3); k=700 exercises the workgroup kernels with d+ up to 699
        iu = np.triu_indices(k, 1)
        csr = gpu.HostCSR.from_edges(iu[0].astype(np.int32), iu[1].astype(np.int32))
        g = upload(gpu, csr)
        assert g.tc_total() == k * (k - 1) * (k - 2) // 6
        assert g.max_out_degree == k - 1
        g.free()


def test_large_out_degree_tiles(gpu, oracle):
    """A dense block big enough that pivot rows exceed every table bin (d+ up to ~3000) plus random sparse noise."""
    rng = np.random.default_rng(3)
    k = 3000
    keep = rng.random((k, k)) < 0.35
    iu = np.triu_indices(k, 1)
    sel = keep[iu]
    src, dst = iu[0][sel].astype(np.int32), iu[1][sel].astype(np.int32)
    csr = gpu.HostCSR.from_edges(src, dst)
    g = upload(gpu, csr)
    a = np.zeros((k, k), dtype=np.float32)
    a[src, dst] = 1
    a[dst, src] = 1
    expect = int(round(np.trace(a @ a @ a))) // 6  # exact in fp32? no: use float64 for the trace
    expect = int(round(np.einsum("ij,ji->", (a.astype(np.float64) @ a), a))) // 6
    assert g.tc_total() == expect
    g.free()


def test_upload_rejects_non_canonical_input(gpu):
    off = np.array([0, 2, 3, 4], dtype=np.int64)
    bad_sorted = np.array([2, 1, 0, 0], dtype=np.int32)       # row 0 not ascending
    asym = np.array([1, 2, 0, 1], dtype=np.int32)             # 2 -> 1 but 1 -/-> 2
    loop = np.array([0, 1, 0, 0], dtype=np.int32)
    for arr in (bad_sorted, asym, loop):
        with pytest.raises(gpu.GmsxError) as ei:
            gpu.DeviceGraph.upload(off, arr)
        assert ei.value.status == gpu.ERR_NOT_CANONICAL
    with pytest.raises(gpu.GmsxError) as ei:
        gpu.DeviceGraph.upload(off, np.array([1, 7, 0, 0], dtype=np.int32))
    assert ei.value.status == gpu.ERR_NOT_CANONICAL
    # one arc of a real graph bent to a non-neighbour (rows stay ascending, loop-free, the arc count even): only the symmetry check —
    # the keyed multiset hashes of the arc set and of its transpose — can see it; the untouched graph passes
    csr = host_graph(gpu, "kronecker", 10)
    off, adj = csr.offsets().copy(), csr.neighbors().copy()
    gpu.DeviceGraph.upload(off, adj).free()
    bent = 0
    for u in range(len(off) - 1):
        for j in range(off[u], off[u + 1] - 1):
            if adj[j] + 1 < adj[j + 1] and adj[j] + 1 != u:
                bad = adj.copy()
                bad[j] += 1
                with pytest.raises(gpu.GmsxError) as ei:
                    gpu.DeviceGraph.upload(off, bad)
                assert ei.value.status == gpu.ERR_NOT_CANONICAL
                bent += 1
                break
        if bent == 20:
            break
    assert bent == 20


def test_relabel_invariance_and_golden_scale20(gpu):
    """Counts are label-invariant (the loader's relabel only shapes locality): raw and relabelled graphs agree,
    and scale 20 matches the reference golden 423 625 371."""
    a = host_graph(gpu, "kronecker", 16, relabel=False)
    b = host_graph(gpu, "kronecker", 16, relabel=True)
    ga, gb = upload(gpu, a), upload(gpu, b)
    assert ga.tc_total() == gb.tc_total() == GRAPHS["kronecker-16-16-relabel"]["triangles"]
    ga.free(), gb.free()
    csr = gpu.HostCSR.generate("kronecker", 20)
    g = upload(gpu, csr)
    assert g.tc_total() == GRAPHS["kronecker-20-16-relabel"]["triangles"]
    g.free()


@pytest.mark.parametrize("hub_limit", [1, 16, 300, 5000])
def test_tail_containers_and_tiling(gpu, oracle, hub_limit):
    """Shrinking the hub id range (upload test hook) pushes ids into the 32-bit tail containers: same counts.
    The dense block gives pivots with > 1024 tail entries, i.e. the tiled hash-set path."""
    flags = gpu.UPLOAD_DEFAULT | (hub_limit << 8)
    csr = host_graph(gpu, "kronecker", 13, 16, True)
    g = gpu.DeviceGraph.from_csr(csr, flags=flags)
    want = GRAPHS.get("kronecker-13-16-relabel", {}).get("triangles") or oracle.tc_total(csr.offsets(), csr.neighbors())
    assert g.tc_total() == want
    assert sum(g.tc_partial(p, 5) for p in range(5)) == want
    ordered = g.kclique_count(4)[0]
    assert ordered == oracle.kclique(csr.offsets(), csr.neighbors(), 4)
    g.free()
    k = 2600
    iu = np.triu_indices(k, 1)
    dense = gpu.HostCSR.from_edges(iu[0].astype(np.int32), iu[1].astype(np.int32))
    g = gpu.DeviceGraph.from_csr(dense, flags=flags)
    assert g.tc_total() == k * (k - 1) * (k - 2) // 6
    g.free()


def test_tail_bucket_overflow(gpu, oracle):
    """Light pivots whose tail members all hash to one bucket of the 64 x 4 bucket set (ids 4096 apart): the kernel must
    fall back to the open-addressing table and still count exactly.  A circulant graph keeps every degree equal, so
    rank ids follow vertex ids and the collisions are by construction."""
    n = 40000
    offs = [4096 * j for j in range(1, 7)] + [1, 4097]
    u = np.concatenate([np.arange(n, dtype=np.int64) for _ in offs])
    v = np.concatenate([(np.arange(n, dtype=np.int64) + d) % n for d in offs])
    csr = gpu.HostCSR.from_edges(u.astype(np.int32), v.astype(np.int32))
    want = oracle.tc_total(csr.offsets(), csr.neighbors())
    assert want > 0
    for hub_limit in (1, 0):
        g = gpu.DeviceGraph.from_csr(csr, flags=gpu.UPLOAD_DEFAULT | (hub_limit << 8))
        assert g.tc_total() == want, hub_limit
        assert sum(g.tc_partial(p, 3) for p in range(3)) == want
        g.free()


@pytest.mark.parametrize("hub_limit,inline_limit", [(16, 0), (16, 40), (16, 700), (16, 10 ** 6), (300, 301), (300, 4000), (0, 3000)])
def test_inline_limit(gpu, oracle, hub_limit, inline_limit):
    """Light pivots hand the edges to their heavy members and to members of rank id < inline_limit over as inline rows (the members
    below v copied next to v, scanned by v's work items); only the edges to far light members stay behind (k_tc_light).  Any limit gives the same
    counts — for the triangle kernels and for the k-clique / Bron–Kerbosch kernels that share the containers."""
    gpu.set_option("INLINE_LIMIT", str(inline_limit))
    try:
        for kind, scale, deg in (("kronecker", 13, 16), ("uniform", 12, 20)):
            csr = host_graph(gpu, kind, scale, deg, True)
            want = oracle.tc_total(csr.offsets(), csr.neighbors())
            g = gpu.DeviceGraph.from_csr(csr, flags=gpu.UPLOAD_DEFAULT | (hub_limit << 8))
            t, st = g.tc_total(stats=True)
            assert t == want and st["units"] == csr.num_edges
            parts = [g.tc_partial(p, 3, stats=True) for p in range(3)]
            assert sum(p[0] for p in parts) == want and sum(p[1]["units"] for p in parts) == csr.num_edges
            assert g.kclique_count(3)[1] == want
            if scale == 12:
                assert g.bk_count() == oracle.bk_count(csr.offsets(), csr.neighbors())
            g.free()
    finally:
        gpu.reset_options()


@pytest.mark.parametrize("knobs", [{"TC_TWO_SIDED": "0"}, {"TC_TWO_SIDED": "0", "INLINE_LIMIT": "0"},
                                   {"TC_OVERLAP": "1"}, {"TC_PERSIST": "0"}, {"TC_PERSIST": "0", "TC_OVERLAP": "1"},
                                   {"TC_ITEM_WGS": "1"}, {"TC_ITEM_WGS": "3"}, {"TC_GAP12": "0"}, {"TC_GAP12": "2"},
                                   {"TC_GAP12": "2", "TC_DELTA": "0"}, {"TC_HYBRID": "1"}, {"TC_HYBRID": "2"},
                                   {"TC_HYBRID": "2", "TC_DELTA": "0"},
                                   {"TC_INLINE_FIRST": "64"}, {"TC_INLINE_FIRST": "48"}, {"TC_INLINE_FIRST": "5"}, {"TC_INLINE_FIRST": "1", "TC_TWO_SIDED": "0"},
                                   {"TC_HOT_WINDOWS": "1", "TC_HOT_KB": "2", "TC_HOT_MIN": "4"},
                                   {"TC_HOT_WINDOWS": "3", "TC_HOT_KB": "1", "TC_HOT_MIN": "1"},
                                   {"TC_HOT_WINDOWS": "8", "TC_HOT_KB": "4", "TC_HOT_MIN": "16", "TC_PERSIST": "0"}])
def test_task_list_knobs(gpu, oracle, knobs):
    """Every oriented edge is counted at exactly one endpoint: at the pivot that keeps it (forward entry), at the member it was handed
    to because the member's row is the bigger one (reverse entry, cut at the member's id), or inside the member's inline rows.  With
    the hand-over off, the light-edge kernel on a side stream, the one-workgroup-per-item kernels of rounds 2-3 instead of the persistent
    one, one / three persistent workgroups per CU, or the sparse rows without / all in
    the 12-bit-gap form, heavy rows split into a prefix bitmap + the rest (GMSX_TC_HYBRID), the hub lists laid out and run in phases
    by the pool window their rows start in (GMSX_TC_HOT_*), or a heavy pivot handing more / fewer of its first members over inline
    (GMSX_TC_INLINE_FIRST), the count and the bookkeeping stay."""
    for k_opt, v_opt in knobs.items():
        gpu.set_option(k_opt, v_opt)
    try:
        for kind, scale, deg, hub_limit in (("kronecker", 14, 16, 0), ("kronecker", 13, 40, 64), ("uniform", 12, 120, 0), ("uniform", 12, 120, 500)):
            csr = host_graph(gpu, kind, scale, deg, True)
            want = oracle.tc_total(csr.offsets(), csr.neighbors())
            g = gpu.DeviceGraph.from_csr(csr, flags=gpu.UPLOAD_DEFAULT | (hub_limit << 8))
            t, st = g.tc_total(stats=True)
            assert t == want and st["units"] == csr.num_edges, (kind, scale, hub_limit)
            b = g.tc_stream_breakdown()
            assert sum(b[k] for k in gpu.DeviceGraph.BREAKDOWN_BYTES) == st["stream_bytes"]
            assert b["count_entries"] >= b["count_inline_entries"] and b["of_which_inline_rows"] <= b["hub_rows_list"] + b["tail_rows_list"]
            hist, extra = g.tc_row_histogram()
            assert int(extra[0]) == b["count_entries"] and int(extra[2]) == b["count_work_items"]
            assert 16 * int(hist[:3, :, 1].sum()) == b["hub_rows_list"] + b["hub_rows_bitset"] + b["hub_rows_delta"]
            parts = [g.tc_partial(p, 5, stats=True) for p in range(5)]
            assert sum(p[0] for p in parts) == want and sum(p[1]["units"] for p in parts) == csr.num_edges
            g.free()
    finally:
        gpu.reset_options()


@pytest.mark.parametrize("delta", ["0", "1", "2"])
def test_stream_row_forms(gpu, oracle, delta):
    """The heavy-pivot kernel reads every member's hub part as a 'stream row' in the cheapest of three forms (16-bit list, bitset,
    byte-delta with 255-escapes).  GMSX_TC_DELTA = 0 / 1 / 2 = never / when smaller / wherever possible: same counts, on graphs
    whose rows have small gaps (dense block), huge gaps (sparse uniform: escapes and one-id units) and both (RMAT)."""
    gpu.set_option("TC_DELTA", delta)
    gpu.set_option("TC_TAIL_DELTA", delta)   # the tail parts have the same choice: 32-bit list or 16-bit delta units
    try:
        for kind, scale, deg in (("kronecker", 14, 16), ("uniform", 13, 150), ("kronecker", 12, 64)):
            csr = host_graph(gpu, kind, scale, deg, True)
            want = oracle.tc_total(csr.offsets(), csr.neighbors())
            for hub_limit in (0, 2000, 40):  # 40: nearly everything lives in the tail containers
                g = gpu.DeviceGraph.from_csr(csr, flags=gpu.UPLOAD_DEFAULT | (hub_limit << 8))
                assert g.tc_total() == want, (kind, scale, hub_limit)
                assert sum(g.tc_partial(p, 3) for p in range(3)) == want
                g.free()
        k = 1500  # a dense random block: d+ up to ~500, gaps of 2-3 -> full delta units
        rng = np.random.default_rng(9)
        iu = np.triu_indices(k, 1)
        sel = rng.random(iu[0].size) < 0.4
        csr = gpu.HostCSR.from_edges(iu[0][sel].astype(np.int32), iu[1][sel].astype(np.int32))
        g = gpu.DeviceGraph.from_csr(csr)
        assert g.tc_total() == oracle.tc_total(csr.offsets(), csr.neighbors())
        g.free()
    finally:
        gpu.reset_options()


def test_shards_of_separate_processes_add_up(gpu):
    """Multi-GPU runs shard whole PIVOTS (shard_of(position in the d+ order)) and every rank uploads the graph itself.  The task lists are
    filled through atomic cursors — their order differs from process to process — so nothing may depend on it: three processes, one shard
    each, on a graph with reverse entries, inline rows and several items per hub, (a) on full uploads, (b) on SHARDED uploads
    (gmsx_graph_upload_csr_shard: only the rank's own task lists and inline rows are built).  Both sums equal the count, the bookkeeping
    units add up to m, a sharded graph refuses the other shards, its k-clique shards still work, and it is much smaller."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "tc_two_process_shards.py"), "19"], capture_output=True, text=True, cwd=root, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["ok"] and rec["total"] == rec["sum_of_process_shards"] == rec["sum_of_sharded_uploads"] and rec["units"] == rec["m"], rec
    assert max(rec["bytes_sharded"]) < 0.9 * rec["bytes_full"], rec  # a third of the task lists and inline rows each; the base layout and the stream rows are whole (scale 19: 0.85 — 0.75 before round 5 shrank what is sharded)


def test_rows_too_long_for_a_task_entry_are_refused_not_miscounted(gpu, oracle):
    """A task entry (6 bytes, device_graph.hpp TaskList) has 14 / 15 bits for the units of the stream row it names.  A graph with a longer row is
    REFUSED when the triangle-count containers are built (GMSX_ERR_UNSUPPORTED — not the memory fallback: a share of the pivots would have the same
    rows), the base layout stays usable, and the same graph counts right without the hook.  GMSX_TC_TEST_MAX_UNITS (test hook) narrows the fields."""
    csr = host_graph(gpu, "kronecker", 14)
    want = oracle.tc_total(csr.offsets(), csr.neighbors())
    gpu.set_option("TC_TEST_MAX_UNITS", "2")
    try:
        g = gpu.DeviceGraph.from_csr(csr)
        with pytest.raises(gpu.GmsxError) as ei:
            g.tc_total()
        assert ei.value.status == gpu.ERR_UNSUPPORTED
        assert g.kclique_count(3)[1] == want         # the base layout is untouched
        with pytest.raises(gpu.GmsxError):
            gpu.DeviceGraph.from_csr(csr, flags=gpu.UPLOAD_FOR_TC)
        gpu.set_option("TC_TEST_MAX_UNITS", "100000")
        assert g.tc_total() == want                  # the handle recovers once the rows fit
        g.free()
    finally:
        gpu.reset_options()


def test_containers_that_do_not_fit_fall_back_to_passes(gpu, oracle):
    """When the triangle-count containers of the whole graph do not fit the device, the library builds them for 1/k of the pivots at a time
    and walks k passes instead of returning GMSX_ERR_DEVICE_MEM.  GMSX_TC_MEM_LIMIT_MB (test hook) pretends the device is small: RMAT scale
    18 needs 83 MB of containers (round 5; passes at a budget of 96 / 64 / 40 / 24 / 16 MB: 1 / 2 / 4 / 16 / refused — tools/probes/tc_passes_probe.py); with a 64 MB budget the count, the units, the shard sums and the per-vertex path stay exact."""
    csr = host_graph(gpu, "kronecker", 18)
    want = oracle.tc_total(csr.offsets(), csr.neighbors())
    g0 = gpu.DeviceGraph.from_csr(csr, flags=gpu.UPLOAD_FOR_TC)
    assert g0.tc_passes == 1
    full_bytes = g0.device_bytes
    t0, st0 = g0.tc_total(stats=True)
    g0.free()
    gpu.set_option("TC_MEM_LIMIT_MB", "64")
    try:
        g = gpu.DeviceGraph.from_csr(csr)
        assert g.tc_passes == 0                      # lazy: nothing built yet
        t, st = g.tc_total(stats=True)
        assert g.tc_passes >= 2 and g.device_bytes < full_bytes
        assert t == t0 == want and st["units"] == csr.num_edges and st["stream_bytes"] > 0 and st["kernel_ms"] > 0
        assert g.tc_total() == want                  # walks the passes again
        assert sum(g.tc_partial(p, 3) for p in range(3)) == want   # a sharded call builds exactly its shard
        assert g.tc_total() == want
        # fewer shards than passes: a shard that cannot fit either is cut into nested sub-shards — slower, never refused —
        # and the handle stays usable for whole-graph calls afterwards
        gpu.set_option("TC_MEM_LIMIT_MB", "40")
        g2 = gpu.DeviceGraph.from_csr(csr)
        assert g2.tc_total() == want and g2.tc_passes >= 4
        parts = [g2.tc_partial(p, 2, stats=True) for p in range(2)]
        assert sum(p[0] for p in parts) == want and sum(p[1]["units"] for p in parts) == csr.num_edges
        assert g2.tc_partial(0, 1) == want and g2.tc_total() == want and g2.tc_passes >= 4
        g2.free()
        gpu.set_option("TC_MEM_LIMIT_MB", "64")
        assert g.kclique_count(3)[1] == want         # the base layout is untouched
        g.free()
        gpu.set_option("TC_MEM_LIMIT_MB", "1")     # nothing fits, not even 1/4096 of the pivots: the one refusal left
        g = gpu.DeviceGraph.from_csr(csr)
        with pytest.raises(gpu.GmsxError) as ei:
            g.tc_total()
        assert ei.value.status == gpu.ERR_DEVICE_MEM
        assert g.kclique_count(3)[1] == want
        g.free()
    finally:
        gpu.reset_options()


def test_random_small_graphs_all_paths(gpu, oracle):
    """Many small random graphs (sparse, dense, clustered, star-heavy) under hub limits that push ids into the tail containers and inline
    limits that move members between the inline rows and the light-pivot kernel: forward / reverse / cut / inline / first-member entries
    all occur, every count equals the oracle's, every shard split adds up."""
    rng = np.random.default_rng(20261003)
    try:
        for trial in range(36):
            n = int(rng.integers(40, 2500))
            kind = trial % 4
            if kind == 0:      # G(n, p) from sparse to fairly dense
                m = int(n * rng.uniform(2, 60))
                src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
            elif kind == 1:    # a dense core + a sparse periphery attached to it (heavy pivots with light members and vice versa)
                core = int(rng.integers(70, 300))
                iu = np.triu_indices(core, 1)
                keep = rng.random(iu[0].size) < rng.uniform(0.3, 0.9)
                ps, pd = rng.integers(core, n + core, 6 * n), rng.integers(0, core + n, 6 * n)
                src, dst = np.concatenate([iu[0][keep], ps]), np.concatenate([iu[1][keep], pd])
            elif kind == 2:    # preferential attachment flavour: targets drawn with a power-law bias
                m = int(n * rng.uniform(4, 30))
                src = rng.integers(0, n, m)
                dst = (n * rng.random(m) ** 3).astype(np.int64)
            else:              # several hubs seeing everything + random edges among the rest
                hubs = int(rng.integers(2, 40))
                hs = np.repeat(np.arange(hubs), n)
                hd = np.tile(np.arange(n), hubs)
                m = int(n * rng.uniform(1, 25))
                src, dst = np.concatenate([hs, rng.integers(0, n, m)]), np.concatenate([hd, rng.integers(0, n, m)])
            keep = src != dst
            csr = gpu.HostCSR.from_edges(src[keep].astype(np.int32), dst[keep].astype(np.int32))
            want = oracle.tc_total(csr.offsets(), csr.neighbors())
            hub_limit = int(rng.choice([0, 1, 7, 40, 300]))
            inline_limit = int(rng.choice([0, 10, 100, 1000, 10 ** 7]))
            gpu.set_option("INLINE_LIMIT", inline_limit)
            g = gpu.DeviceGraph.from_csr(csr, flags=gpu.UPLOAD_DEFAULT | (hub_limit << 8))
            t, st = g.tc_total(stats=True)
            assert t == want and st["units"] == csr.num_edges, (trial, n, kind, hub_limit, inline_limit)
            nparts = int(rng.integers(2, 6))
            assert sum(g.tc_partial(p, nparts) for p in range(nparts)) == want, (trial, nparts)
            g.free()
    finally:
        gpu.reset_options()
