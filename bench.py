#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json: edges-intersected/s (+ HBM roofline) of the triangle count on a
synthetic RMAT graph, through the gmsx C-ABI on N GPUs of one node.

  python bench.py                       # N=1, RMAT scale-26 ef-16 (the graph BASELINE.json's metric is quoted on), 5 steps,
                                        # 2 warm-ups; plus a parity record on scale 24 (configs[1], reference golden)
  python bench.py --scale 24            # configs[1] as the timed workload
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

At N=1 the same run appends the two other single-GPU perf configs of BASELINE.json as records of the JSON line:
  config2_kclique4_s22   k = 4 cliques, RMAT scale-22 ef-16 (configs[2]); count asserted against the compiled reference's golden
  config3_bk             Bron–Kerbosch maximal cliques on the com-Orkut-shaped RMAT (|E| = 117 M, configs[3]); likewise
each with kernel_ms (best of 3), roots/s, the lean upload time and beyond-L2 traffic from the same rocprofv3 PMC child mechanism.

A step = one full pass of the hot path over the graph: every rank counts its cost-balanced shard of the m undirected edges
(one intersect_count per edge) with the HIP kernels, then ONE 8-byte all-reduce over RCCL (native: gmsx_comm_allreduce_u64;
torch.distributed is only the launcher and the control plane) combines the partial counts — the device replacement of
Par::count_total (triangle_count/parallel/total.h:7-24).  The graph is resident in HBM before the timed region starts.
Rank 0 prints ONE JSON line.

Roofline record (all per launch = one rank's pass):
  achieved / frac   MEASURED traffic beyond the L2 (rocprofv3 PMC, separate passes, collected in this very run by short child processes
                    BEFORE the parent touches the GPU) / kernel time / 8 TB/s.  Reads = the L2's memory-side requests by size class,
                    32 n32 + 64 n64 + 128 n128 (TCC_EA0_RDREQ_{32B,64B,128B}_sum); FETCH_SIZE — every request tallied at 64 B — is collected
                    too and the ratio reported as `fetch_multiplier` (2.00 for these kernels: > 99 % of their requests are 128-byte ones);
                    writes = WRITE_SIZE.  The requests include Infinity-Cache hits, so this is beyond-L2 (MALL + HBM) bandwidth.
  algorithmic_bytes device-computed bytes of THIS formulation (oriented rows in the container form the kernels read, no
                    on-chip reuse assumed: gmsx_stats.stream_bytes); work_efficiency = traffic / algorithmic_bytes.
  reference_equivalent_GBps   B_alg / t with SURVEY §8(d)'s B_alg = 4*Σ(d_u+d_v) + 8(n+1) + 4*nnz — what the reference's
                    full-row merges would stream; NOT a hardware utilisation (the oriented kernels do ~20x less work).
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, "Chip-level parameters")
METRIC = "edges-intersected/sec + achieved HBM GB/s, RMAT-26 triangle count @1/2/4/8 GPU"  # BASELINE.json "metric", verbatim
# per workload: the sources its kernels and containers are built from (every PMC figure carries the hash of what it was measured on)
KERNEL_SOURCES = {
    "tc": ["gms_amd/csrc/hip/tc.hip", "gms_amd/csrc/hip/device_graph.hip", "gms_amd/csrc/hip/device_graph.hpp"],
    "kc4": ["gms_amd/csrc/hip/kclique.hip", "gms_amd/csrc/hip/kc4_mfma.hpp", "gms_amd/csrc/hip/device_graph.hip", "gms_amd/csrc/hip/device_graph.hpp"],
    "bk": ["gms_amd/csrc/hip/bk.hip", "gms_amd/csrc/hip/device_graph.hip", "gms_amd/csrc/hip/device_graph.hpp"],
}
KERNEL_SOURCES["kc26"] = KERNEL_SOURCES["kc4"]
SHARD_COUNTS = (1, 2, 4, 8)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def kernel_hash(workload="tc"):
    """Identity of a workload's kernels + containers: PMC numbers are only valid for the build they were measured on."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES[workload]:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def cpu_quota_cores():
    """CPU bandwidth limit of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited/unknown:
    the host may show 256 hardware threads while the container is throttled to a fraction of them."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        return None if quota <= 0 else round(quota / period, 2)
    except (OSError, ValueError):
        return None


def host_cores():
    quota = cpu_quota_cores()
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if quota:
        ncores = max(1, min(ncores, int(quota + 0.999)))
    return ncores


def host_memory():
    """RssAnon / RssFile / VmHWM of this process in bytes (/proc/self/status)."""
    out = {"RssAnon": 0, "RssFile": 0, "VmHWM": 0}
    try:
        with open("/proc/self/status") as f:
            for line in f:
                k = line.split(":")[0]
                if k in out:
                    out[k] = int(line.split()[1]) * 1024
    except (OSError, ValueError, IndexError):
        pass
    return out


def set_gomp_threads(n):
    os.environ["OMP_NUM_THREADS"] = str(n)  # for a libgomp that is not loaded yet
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def cpu_baseline(csr, seconds):
    """The oracle's OpenMP restatement of Par::count_total timed on a bounded, strided sample of the same graph."""
    quota = cpu_quota_cores()
    if quota and quota < (os.cpu_count() or 1) and "GMSX_CPU_THREADS" not in os.environ:
        # a throttled container: one thread per core of quota beats hundreds of threads sharing it (the OpenMP runtime of
        # the oracle and the compiled reference is libgomp, usually loaded long before by numpy / torch: set it at run time)
        set_gomp_threads(max(1, int(quota + 0.999)))
    elif "GMSX_CPU_THREADS" in os.environ:
        set_gomp_threads(int(os.environ["GMSX_CPU_THREADS"]))
    from oracle.bindings import Oracle
    O = Oracle()
    off, ng = csr.offsets(), csr.neighbors()
    cores = O.max_threads()
    total_elems = csr.merge_elements()
    m = csr.num_edges
    # calibrate on a ~1 s sample, then size the real sample for `seconds` (the rate depends heavily on the host)
    stride = max(1, int(round(total_elems / (0.1e9 * cores))))
    t0 = time.perf_counter()
    _, _, e0 = O.tc_total_sample(off, ng, stride, stride // 2)
    rate0 = e0 / max(time.perf_counter() - t0, 1e-6)
    stride = max(1, int(round(total_elems / max(seconds * rate0, 1.0))))
    for _ in range(3):  # the merge rate of a sparse sample differs from the calibration's: re-size until the sample is >= 60 % of the target
        phase = stride // 2  # vertices phase, phase+stride, … (a tiny sample is not handed the single biggest hub)
        t0 = time.perf_counter()
        raw, edges, elems = O.tc_total_sample(off, ng, stride, phase)
        dt = time.perf_counter() - t0
        if dt >= 0.6 * seconds or stride == 1:
            break
        stride = max(1, int(stride * dt / seconds))
    elems_per_s = elems / dt if dt > 0 else 0.0
    # whole-graph-equivalent rate: the merge cost is linear in merged ids, so edges/s = m / (Σ(d_u+d_v) / (ids/s))
    value = m / (total_elems / elems_per_s) if elems_per_s > 0 else 0.0
    return {
        "value": value, "unit": "edges/s", "cores": cores, "kind": "port",
        "sample": (f"oracle/gms_oracle.c gmso_tc_total_sample (OpenMP static,17 loop + scalar merge of "
                   f"parallel/total.h:12-19): vertices u = {phase} mod {stride} of the same graph, {edges} edges, "
                   f"{elems} merged ids in {dt:.2f} s; value = m / (total merged ids / sampled ids-per-second)"),
        "sample_seconds": dt, "sample_edges": edges, "sample_elements": elems,
        "algorithmic_GBps": 4.0 * elems_per_s / 1e9,
        "cpu_quota_cores": cpu_quota_cores(), "host_threads_visible": os.cpu_count(),
    }


def reference_baseline(scale, degree, headline_golden=None):
    """The COMPILED REFERENCE itself (oracle/_ref/libgms_ref.so: spcl/gms headers + CRoaring) timed on this host: Par::count_total over
    RoaringGraph — the reference's fastest flavour (triangle_count.cc:43) — on the largest graph of the family that runs in ≈15–30 s
    (the reference API only runs whole graphs), SortedSetGraph two scales below; plus the recorded whole-graph figure of the headline
    graph from tests/golden/graphs.json, labelled as such.  Skipped when the prebuilt library is absent/unloadable."""
    try:
        from oracle import bindings
        if not bindings.have_ref():
            return None
        R = bindings.Reference()
        out = {"kind": "reference", "threads": int(R.L.ref_omp_threads())}
        for name, kind, sc in (("RoaringGraph", 1, scale), ("SortedSetGraph", 0, max(scale - 2, 10))):
            g = R.generate("kronecker", sc, degree, relabel=True)
            m = R.L.ref_nnz(g) // 2
            # the reference harness times kernel(sgraph) alone — SetGraph::FromCGraph is its untimed "GraphExec buildTime"
            # (common/benchmark.h:105-116) —, so the shim clocks the two phases apart and the baseline is m / count_s
            if hasattr(R.L, "ref_tc_total_timed"):
                tri, build_s, count_s = R.tc_total_timed(g, kind)
            else:  # a prebuilt shim of an older round
                t0 = time.perf_counter()
                tri = R.tc_total(g, kind)
                build_s, count_s = 0.0, time.perf_counter() - t0
            R.free(g)
            out[name] = {"graph": f"RMAT scale-{sc} ef={degree} (reference loader)", "m": int(m), "setgraph_build_s": build_s, "count_s": count_s,
                         "edges_per_s": m / count_s, "edges_per_s_incl_setgraph_build": m / (build_s + count_s), "triangles": int(tri)}
        out["value"], out["unit"], out["cores"] = out["RoaringGraph"]["edges_per_s"], "edges/s", out["threads"]
        out["value_incl_setgraph_build"] = out["RoaringGraph"]["edges_per_s_incl_setgraph_build"]
        out["sample"] = ("Par::count_total<RoaringGraph> of the compiled reference on the whole " + out["RoaringGraph"]["graph"] +
                         ": m / seconds of kernel(sgraph) alone, as the reference harness times it (common/benchmark.h:111-116); the figure with "
                         "SetGraph::FromCGraph inside the clock is value_incl_setgraph_build")
        src = ((headline_golden or {}).get("sources") or {}).get("triangles", "")
        mt = re.search(r"(\d+) threads, (\d+) s", src)
        if headline_golden and mt:
            out["headline_graph_recorded"] = {"edges_per_s": headline_golden["m"] / float(mt.group(2)), "seconds": float(mt.group(2)), "threads": int(mt.group(1)),
                                              "where": "BUILD CONTAINER (8 vCPU), not this box: " + src}
        return out
    except Exception as e:  # noqa: BLE001 - a baseline must never break the bench line
        return {"error": repr(e)}


def golden_triangles(generator, scale, degree):
    try:
        with open(os.path.join(ROOT, "tests", "golden", "graphs.json")) as f:
            rec = json.load(f).get(f"{generator}-{scale}-{degree}-relabel")
        return (rec.get("triangles"), (rec.get("sources") or {}).get("triangles") or rec.get("source")) if rec else (None, None)
    except OSError:
        return None, None


def config1_check(capi, generator, scale, degree, algo, divisor):
    """BASELINE.json configs[1] (RMAT scale 24 on one GPU) as a parity record next to the headline: a few passes on that
    graph, the count asserted against the reference's golden (tests/golden/graphs.json, from the compiled reference)."""
    csr = capi.HostCSR.generate(generator, scale, degree, capi.RELABEL_AUTO)
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED | capi.UPLOAD_FOR_TC)
    ms = []
    for _ in range(4):
        partial, st = g.tc_partial(0, 1, algo, stats=True)
        ms.append(st["kernel_ms"])
    m = csr.num_edges
    g.free()
    tri = partial // divisor
    golden, _ = golden_triangles(generator, scale, degree)
    if golden is not None:
        assert tri == golden, f"PARITY FAILURE (configs[1]): {tri} != reference golden {golden}"
    best = min(ms[1:])
    return {"workload": f"triangle count, RMAT scale-{scale} ef={degree} (BASELINE.json configs[1])", "m": int(m), "triangles": int(tri),
            "parity": "== reference golden" if golden is not None else "no golden", "kernel_ms": best, "edges_per_s": m / (best * 1e-3)}



def golden_record(key):
    try:
        with open(os.path.join(ROOT, "tests", "golden", "graphs.json")) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def side_workload(capi, args, name, traffic, rank, cus, ceiling):
    """BASELINE.json configs[2] / configs[3] on this GPU: lean upload (no triangle-count containers), five timed calls (the best is reported, all are listed), the count asserted
    against the golden the COMPILED REFERENCE produced (tools/make_golden_big.py), traffic from the PMC child passes of this run."""
    w = WORKLOADS[name]
    t0 = time.perf_counter()
    csr = workload_csr(capi, args, name)
    t_load = time.perf_counter() - t0
    n, m, nnz = csr.num_nodes, csr.num_edges, csr.nnz
    t0 = time.perf_counter()
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_DEFAULT)  # validated on the device; DAG containers + bitsets only
    t_upload = time.perf_counter() - t0
    dev_bytes = g.device_bytes
    ms, values, st, setup_first = [], [], None, 0.0
    for it in range(5):
        v, st = run_workload(g, name)
        ms.append(st["kernel_ms"])
        values.append(v)
        if it == 0:
            setup_first = float(st.get("setup_ms") or 0.0)  # one-time container builds of the first call (outside kernel_ms, like the reference's SetGraph build)
    dev_bytes_after = g.device_bytes  # (k-clique: + the reverse-row lists and their arena, built by the first call — kclique.hip ensure_kc_reverse)
    g.free()
    assert len(set(values)) == 1, values
    rec_g = golden_record(w["golden_key"]) or {}
    golden = rec_g.get(w["golden_field"])
    if golden is not None:
        assert (rec_g.get("n"), rec_g.get("m")) == (n, m), (rec_g.get("n"), rec_g.get("m"), n, m)
        got = values[0] // 24 if w["golden_field"] == "kc4_true" else values[0]  # kc4_true = C_4 itself (the reference's kClist), kc4 = 4!·C_4 (its set-based CliqueCount)
        assert got == golden and (w["golden_field"] != "kc4_true" or values[0] == 24 * golden), f"PARITY FAILURE ({w['key']}): {got} != reference golden {golden}"
        parity = "== reference golden (%s)" % ((rec_g.get("sources") or {}).get(w["golden_field"], "tests/golden/graphs.json"))
    else:
        parity = "no reference golden in tests/golden/graphs.json (self-consistency only)"
    best = min(ms)
    out = {"workload": w["label"], "n": int(n), "m": int(m), "parity": parity, "kernel_ms": best, "kernel_ms_all": ms, "launches": st["launches"],
           "roots_per_s": n / (best * 1e-3), "upload_s_lean": t_upload, "graph_device_bytes": int(dev_bytes), "graph_device_bytes_after_first_call": int(dev_bytes_after),
           "first_call_setup_ms": setup_first, "load_or_generate_s": t_load,
           "one_shot_s_upload_plus_first_call": t_upload + (ms[0] + setup_first) * 1e-3}
    if name in ("kc4", "kc26"):
        out["ordered_count"] = int(values[0])
        out["cliques"] = int(values[0]) // 24
        out["cliques_per_s"] = out["cliques"] / (best * 1e-3)
        tri = rec_g.get("triangles")
        if tri is not None:
            # what the reference's recursion executes on the symmetric graph (k_clique_count_set_based.h:5-31): one `intersect` per
            # (u, v in N(u)) and one per (u, v, w in N(u)∩N(v)) = nnz + 6 T calls for k = 4
            out["reference_intersect_calls"] = int(nnz + 6 * tri)
            out["reference_intersect_calls_per_s"] = (nnz + 6 * tri) / (best * 1e-3)
    else:
        out["maximal_cliques"] = int(values[0])
        out["maximal_cliques_per_s"] = values[0] / (best * 1e-3)
        out["resume_rounds"] = int(st["probes"])
    trec = (traffic or {}).get("n1")
    alg = int(st.get("stream_bytes") or 0)
    out["roofline"] = make_roofline(alg, best * 1e-3, trec if (trec and "bytes" in trec) else None, cus, ceiling, extra={
        "algorithmic_bytes_are": ALG_BYTES_ARE[name], "kernel_hash": kernel_hash(name),
        "traffic_source": "rocprofv3 --pmc child passes of this run (memory-side read requests by size class + WRITE_SIZE, SQ_* and GRBM_GUI_ACTIVE: separate passes, one call each)"})
    log(rank, f"{w['key']}: {best:.2f} ms ({parity[:40]}…), upload {t_upload:.2f}s")
    return out


def vertex_count2_record(capi, args, rank):
    """SURVEY §8(f) row 1 on the configs[2] graph: Par::vertex_count2 (parallel/vertex.h:14-27) — counts[u] = Σ_{v∈N(u)} |N(u)∩N(v)| = 2 x the triangles at u,
    so Σ_u counts[u] = 6 T, asserted against the reference golden of that graph."""
    csr = workload_csr(capi, args, "kc4")
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_DEFAULT)
    ms, total = [], None
    for _ in range(3):
        c, st = g.tc_vertex_count2(stats=True)
        ms.append(st["kernel_ms"])
        total = int(c.sum(dtype="int64"))
    g.free()
    tri = (golden_record(WORKLOADS["kc4"]["golden_key"]) or {}).get("triangles")
    if tri is not None:
        assert total == 6 * tri, f"PARITY FAILURE (vertex_count2): sum {total} != 6 x {tri}"
    best = min(ms)
    log(rank, f"vertex_count2_s22: {best:.2f} ms")
    return {"workload": "per-vertex triangle counts (Par::vertex_count2), RMAT scale-22 ef=16", "n": int(csr.num_nodes), "m": int(csr.num_edges),
            "kernel_ms": best, "kernel_ms_all": ms, "sum_counts": total, "parity": "sum == 6 x the reference golden's triangles" if tri is not None else "no golden",
            "edges_per_s": csr.num_edges / (best * 1e-3), "launches": st["launches"]}


def tc_s27_record(capi, args, algo, divisor, rank, t_start):
    """BASELINE.json configs[4]'s graph (RMAT scale 27, 2.1 G edges) on ONE GPU: the whole pass, then each of the eight shards a rank of the 8-GPU run
    would count (gmsx_tc_partial(p, 8) on the full upload: the shard's own work items, compacted) — their times are what an 8-GPU step costs besides the
    8-byte all-reduce.  Count asserted against the reference golden (ref_tc_total_sliced, tests/golden/graphs.json)."""
    age = time.perf_counter() - t_start
    if age > args.big_budget_s:
        return {"skipped": f"the run was already {age:.0f} s old (--big-budget-s {args.big_budget_s:.0f}): generating the scale-27 graph on the host takes ~2 min more"}
    t0 = time.perf_counter()
    csr = capi.HostCSR.generate("kronecker", 27, 16, capi.RELABEL_AUTO)
    t_gen = time.perf_counter() - t0
    n, m = csr.num_nodes, csr.num_edges
    t0 = time.perf_counter()
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_DEFAULT)
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    g.prepare(capi.PREPARE_TC)
    t_build = time.perf_counter() - t0
    ms, whole, st = [], None, None
    for _ in range(3):
        whole, st = g.tc_partial(0, 1, algo, stats=True)
        ms.append(st["kernel_ms"])
    stream_bytes, dev_bytes = int(st["stream_bytes"]), int(g.device_bytes)
    shard_ms, shard_sum = [], 0
    for p in range(8):
        part, t = None, []
        for _ in range(2):
            part, sst = g.tc_partial(p, 8, algo, stats=True)
            t.append(sst["kernel_ms"])
        shard_ms.append(min(t))
        shard_sum += part
    g.free()
    tri = whole // divisor
    golden, src = golden_triangles("kronecker", 27, 16)
    assert shard_sum == whole, f"PARITY FAILURE (tc_s27): shards add up to {shard_sum}, the whole pass counted {whole}"
    if golden is not None:
        assert (n, m) == (134217728, 2111632322) and tri == golden, f"PARITY FAILURE (tc_s27): {tri} != reference golden {golden}"
    best = min(ms)
    log(rank, f"tc_s27: {best:.1f} ms whole, shards {min(shard_ms):.2f} … {max(shard_ms):.2f} ms, generate {t_gen:.0f}s")
    return {"workload": "triangle count, RMAT scale-27 ef=16 on ONE GPU + its eight shards one after the other (BASELINE.json configs[4]'s graph; the 8-GPU run itself is unmeasured: no node)",
            "n": int(n), "m": int(m), "triangles": int(tri), "parity": f"== reference golden ({src})" if golden is not None else "no golden",
            "kernel_ms": best, "kernel_ms_all": ms, "edges_per_s": m / (best * 1e-3), "shard_kernel_ms": shard_ms, "shard_ms_max": max(shard_ms), "shard_ms_min": min(shard_ms),
            "shards_sum_over_whole": sum(shard_ms) / best, "eight_gpu_step_estimate_edges_per_s": m / (max(shard_ms) * 1e-3),
            "eight_gpu_step_estimate_note": "m / the slowest shard's kernel time: what eight GPUs would reach if the 8-byte all-reduce were free — an estimate from one GPU, NOT a measurement",
            "algorithmic_bytes": stream_bytes, "algorithmic_GBps": stream_bytes / (best * 1e-3) / 1e9, "frac": stream_bytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "graph_device_bytes": dev_bytes, "generate_s": t_gen, "upload_s": t_up, "build_tc_containers_s": t_build}


# ---- rocprofv3 PMC passes, live --------------------------------------------------------------------------------------

def sg_cache_path(args):
    return os.path.join(args.cache_dir, f"{args.generator}-{args.scale}-{args.degree}.sgx")  # mapped, not read: the ranks / PMC children share one page-cache copy


# the single-GPU perf configs of BASELINE.json besides the triangle count (configs[2], configs[3])
WORKLOADS = {
    "kc4": dict(key="config2_kclique4_s22", gen=("kronecker", 22, 16), golden_key="kronecker-22-16-relabel", golden_field="kc4", kernels="k_kc",
                label="k=4 clique counting, RMAT scale-22 ef=16 (BASELINE.json configs[2])"),
    # the NORTH-STAR k-clique workload (BASELINE.json north_star: "bit-exact triangle and k-clique counts on RMAT scale-26"): k = 4 on the headline graph;
    # the golden is the reference's kClist count (tests/golden/graphs.json kc4_true: the set-based CliqueCount cannot reach this size)
    "kc26": dict(key="kclique4_s26", gen=("kronecker", 26, 16), golden_key="kronecker-26-16-relabel", golden_field="kc4_true", kernels="k_kc",
                 label="k=4 clique counting, RMAT scale-26 ef=16 (BASELINE.json north_star: k-clique counts on the headline graph)"),
    "bk": dict(key="config3_bk", gen=("rmat", 21, 56, 0.45, 0.22, 0.22), golden_key="rmat-21-56-a45-b22-c22", golden_field="bk", kernels="k_bk_",
               label="Bron-Kerbosch maximal cliques, com-Orkut-shaped RMAT scale-21 ef=56 A=.45 B=C=.22, |E|=117M (BASELINE.json configs[3])"),
}


# gmsx_stats.stream_bytes of the two side workloads (include/gmsx.h): computed on the device for the call, no cache assumed
ALG_BYTES_ARE = {
    "kc26": None,  # = kc4, set below
    "kc4": "per pivot (d+ >= 3) its own containers once + per member v the containers of N+(v) the BUILD reads (bitset words or 16-bit list, tail ids; one "
           "4-byte gather per pair for d+ <= 32) — or, for a hub member handed to its receiver (reverse rows), the 2 i bytes of the pivot's prefix below it + a "
           "16-byte record + the row's ceil(i/32) words written and read back — + the bit matrices of the pivots wider than 512 (k = 4: they pass through a pool "
           "in global memory to the matrix-core count, k_kc4_mfma), written and read once; the narrower pivots' COUNT runs on the bit-matrix in LDS",
    "bk": "per start vertex the oriented rows of all its neighbours (what the builds walk: candidates -> Cadj, in-neighbours -> XT) + Cadj | XT of the start "
          "vertices built in the arena, once + one Cadj row (c/32 words) per search-tree node — the operand of cand.intersect(N(q)) (tomita.h:51-70); the Xf / XT "
          "words a node reads and the saved levels are not counted",
}


ALG_BYTES_ARE["kc26"] = ALG_BYTES_ARE["kc4"]


def side_names(args):
    """The side records of an N = 1 run, in the order they run."""
    names = ["bk", "kc4"] if args.side else []
    if args.big and args.scale == 26 and args.generator == "kronecker" and args.degree == 16:
        names.append("kc26")  # k = 4 on the headline graph itself: shares its cache file
    return names


def workload_cache_path(args, name):
    gen = WORKLOADS[name]["gen"]
    return os.path.join(args.cache_dir, "-".join(str(x) for x in gen) + ".sgx")


def workload_csr(capi, args, name):
    """The host CSR of a side workload: from the .sgx cache of this box when present, else generated (and cached for the PMC children)."""
    path = workload_cache_path(args, name)
    if os.path.exists(path):
        try:
            return capi.HostCSR.load(path, relabel=capi.RELABEL_NEVER)
        except capi.GmsxError:
            pass
    gen = WORKLOADS[name]["gen"]
    csr = capi.HostCSR.generate(gen[0], gen[1], gen[2], capi.RELABEL_AUTO) if gen[0] != "rmat" else capi.HostCSR.generate_rmat(*gen[1:])
    try:
        os.makedirs(args.cache_dir, mode=0o700, exist_ok=True)
        fd, tmp = tempfile.mkstemp(prefix=".sg_", dir=args.cache_dir)
        os.close(fd)
        csr.save_sgx(tmp)
        os.replace(tmp, path)
    except (OSError, capi.GmsxError):
        pass
    return csr


def run_workload(g, name):
    if name in ("kc4", "kc26"):
        ordered, cliques, st = g.kclique_count(4, stats=True)
        return ordered, st
    total, st = g.bk_count(stats=True)
    return total, st


def pmc_child(args):
    """Runs under `rocprofv3 --pmc …` (spawned by the parent below).  Triangle count: loads the cached graph, uploads it and executes one
    pass per shard count — full graph, then shard 0 of 2, 4, 8 — printing which dispatches belong to which.  kc4 / bk: ONE call of the
    workload on its cached graph (lean upload)."""
    from gms_amd import capi
    capi.init(0)
    capi.set_host_threads(host_cores())
    if args.workload != "tc":
        csr = workload_csr(capi, args, args.workload)
        g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_DEFAULT)
        value, st = run_workload(g, args.workload)
        print("PMC_CHILD " + json.dumps([{"nparts": 1, "launches": st["launches"], "kernel_ms": st["kernel_ms"], "partial": value, "stream_bytes": 0}]), flush=True)
        g.free()
        return
    csr = capi.HostCSR.load(sg_cache_path(args), relabel=capi.RELABEL_NEVER)
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_DEFAULT | capi.UPLOAD_FOR_TC)  # a cache file is not trusted: validated on the device
    algo = {"auto": capi.TC_AUTO, "oriented": capi.TC_ORIENTED, "full": capi.TC_FULL}[args.algo]
    seq = []
    for nparts in SHARD_COUNTS:
        partial, st = g.tc_partial(0, nparts, algo, stats=True)
        seq.append({"nparts": nparts, "launches": st["launches"], "kernel_ms": st["kernel_ms"], "partial": partial,
                    "stream_bytes": st["stream_bytes"]})
    print("PMC_CHILD " + json.dumps(seq), flush=True)
    g.free()


def run_pmc_pass(args, counters, timeout, workload="tc"):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="gmsx_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    cmd = [rocprof, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "-o", "pmc", "--",
                                            sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", workload, "--scale", str(args.scale),
                                            "--degree", str(args.degree), "--generator", args.generator, "--algo", args.algo, "--cache-dir", args.cache_dir]
    prefix = "k_tc_" if workload == "tc" else WORKLOADS[workload]["kernels"]
    dur_key = "dur_ns:" + counters[0]
    counters = list(counters) + [dur_key]  # summed and carried like a counter
    try:
        env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
        r = subprocess.run(cmd, cwd=out, env=env, capture_output=True, text=True, timeout=timeout)
        line = next((l for l in r.stdout.splitlines() if l.startswith("PMC_CHILD ")), None)
        if r.returncode != 0 or line is None:
            return None, f"rc={r.returncode}: {(r.stderr or r.stdout)[-400:]}"
        seq = json.loads(line[len("PMC_CHILD "):])
        rows = []
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                rows += [x for x in csv.DictReader(fh) if prefix in x["Kernel_Name"] and "stats" not in x["Kernel_Name"] and "k_kcr_" not in x["Kernel_Name"]]  # (k_kcr_*: the one-off list build of a graph's first k-clique call)
        # dispatches in issue order -> passes (each pass = st["launches"] dispatches)
        by_dispatch = {}
        for x in rows:
            d = by_dispatch.setdefault(int(x["Dispatch_Id"]), {"kernel": x["Kernel_Name"].split("(")[0].replace("gmsx::", "").replace("void ", "")})
            d[x["Counter_Name"]] = d.get(x["Counter_Name"], 0.0) + float(x["Counter_Value"])
            try:  # the dispatch's own duration under this pass (the cycle base of the compute-side fractions comes from the same dispatches)
                d[dur_key] = float(x["End_Timestamp"]) - float(x["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
        order = [by_dispatch[k] for k in sorted(by_dispatch)]
        if workload != "tc":  # one call: every dispatch of the workload's kernels belongs to it (helper kernels have other prefixes)
            kernels = {}
            for d in order:
                k = kernels.setdefault(d["kernel"], {"dispatches": 0})
                k["dispatches"] += 1
                for c in counters:
                    k[c] = k.get(c, 0.0) + d.get(c, 0.0)
            return {"n1": {"counters": {c: sum(d.get(c, 0.0) for d in order) for c in counters}, "kernels": kernels,
                           "kernel_ms_under_pmc": seq[0]["kernel_ms"], "stream_bytes": 0, "result": seq[0]["partial"]}}, None
        if len(order) != sum(s["launches"] for s in seq):
            return None, f"dispatch count mismatch: {len(order)} PMC rows for {seq}"
        res, pos = {}, 0
        for s in seq:
            mine = order[pos:pos + s["launches"]]
            pos += s["launches"]
            res[f"n{s['nparts']}"] = {"counters": {c: sum(d.get(c, 0.0) for d in mine) for c in counters},
                                      "kernels": {d["kernel"]: {c: d.get(c, 0.0) for c in counters} for d in mine},
                                      "kernel_ms_under_pmc": s["kernel_ms"], "stream_bytes": s["stream_bytes"]}
        return res, None
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
        return None, repr(e)
    finally:
        shutil.rmtree(out, ignore_errors=True)


RDREQ = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
# the compute side of the roofline: what the SIMDs and the LDS arrays did while the kernels ran, from two more child passes
SQ_PASS = ["SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"]
GRBM_PASS = ["GRBM_GUI_ACTIVE"]
MFMA_PASS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA"]  # the matrix cores (round 6: the k = 4 count of the wide pivots, kc4_mfma.hpp); not collected for the triangle count
N_XCD = 8                      # MI355X: 8 XCDs x 32 CUs; GRBM_GUI_ACTIVE comes back summed over the XCDs (profiles/r05/s26_r5: 797 M per 48.5 ms launch = 8 x 2.05 GHz)
VALU_CYCLES_PER_INST = 3.0     # SIMD cycles per wave64 VALU instruction for the mix of these kernels: tools/probes/valu_rate.hip measures 2.2-2.5 for the
VALU_CYCLES_RANGE = (2.3, 4.3)  # add / and / shift class and 4.2-4.3 for v_bfe_u32 / v_lshl_or_b32 / v_mul_lo_u32 / v_bcnt_u32_b32 (profiles/r04/valu_rate.txt)


def compute_side(rec, cus):
    """VALU and LDS utilisation of the dispatches whose counter sums are in `rec` (one kernel, or a whole call).  Under the profiler the dispatches
    of a call run one after the other, so the cycle base is the sum of their own busy cycles.
      valu_frac = SQ_INSTS_VALU x cycles-per-instruction / (SIMDs x cycles)     (cycles = GRBM_GUI_ACTIVE / XCDs: one XCD clock's cycles while busy)
      lds_frac  = SQ_LDS_IDX_ACTIVE / (CUs x cycles)                            (cycles the LDS arrays were indexing; bank conflicts are part of it)"""
    gui, insts = rec.get("GRBM_GUI_ACTIVE"), rec.get("SQ_INSTS_VALU")
    if not gui or insts is None:
        return None
    cycles = gui / N_XCD
    out = {"valu_frac": insts * VALU_CYCLES_PER_INST / (4 * cus * cycles),
           "valu_frac_range": [insts * c / (4 * cus * cycles) for c in VALU_CYCLES_RANGE],
           "valu_instructions": insts, "busy_cycles_per_xcd": cycles}
    lds = rec.get("SQ_LDS_IDX_ACTIVE")
    if lds is not None:
        out["lds_frac"] = lds / (cus * cycles)
        out["lds_bank_conflict_share"] = (rec.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds) if lds else 0.0
    if rec.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:  # rocprofiler's MfmaUtil: busy cycles of the matrix pipes, summed over the SIMDs, / (SIMDs x cycles)
        out["mfma_frac"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * cus * cycles)
        out["mfma_instructions"] = rec.get("SQ_INSTS_MFMA")
    if rec.get("SQ_WAVE_CYCLES"):
        out["waves_waiting_share"] = rec.get("SQ_WAIT_ANY", 0.0) / rec["SQ_WAVE_CYCLES"]
    dur = rec.get("dur_ns:GRBM_GUI_ACTIVE")
    if dur:
        out["clock_GHz"] = cycles / dur
    return out


def make_roofline(alg_bytes, kernel_s, trec, cus, ceiling_gbs, extra=None):
    """The `roofline` object of one record.  achieved / frac follow the contract: ALGORITHMIC bytes of the formulation per launch / the launch's
    duration / the 8 TB/s specification peak; `traffic` = the bytes the L2 requested from memory (PMC, beyond-L2 = Infinity Cache + HBM), with its own
    rate and fraction beside it; valu_frac / lds_frac from the SQ / GRBM child passes; `bound` = whichever of the three fractions (memory by traffic,
    VALU, LDS) is highest."""
    ach = alg_bytes / kernel_s / 1e9 if alg_bytes else None
    traffic = trec["bytes"] if trec and "bytes" in trec else None
    r = {"bound": None, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic,
         "achieved_is": "algorithmic bytes of this formulation (gmsx_stats.stream_bytes: computed on the device for the call, no on-chip reuse assumed) / kernel time",
         "algorithmic_bytes": int(alg_bytes) if alg_bytes else None,
         "traffic_GBps": (traffic / kernel_s / 1e9) if traffic else None, "frac_traffic": (traffic / kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
         "work_efficiency_traffic_over_algorithmic": (traffic / alg_bytes) if (traffic and alg_bytes) else None,
         "memory_level": "beyond-L2 (Infinity Cache + HBM): the counters tally the L2's memory-side requests, MALL hits included; no DRAM-side counter separates them",
         "measured_stream_ceiling_GBps": ceiling_gbs,
         "frac_of_measured_stream_ceiling": (ach / ceiling_gbs) if (ach and ceiling_gbs) else None,
         "frac_traffic_of_measured_stream_ceiling": (traffic / kernel_s / 1e9 / ceiling_gbs) if (traffic and ceiling_gbs) else None,
         "measured_stream_ceiling_is": "gmsx_hbm_read_probe of this run: a read-only in-order sweep of an 8 GiB buffer (far beyond the 256 MB Infinity Cache), "
                                       "16-byte loads, HIP events — what HBM delivers to a pure stream on this box; a kernel whose beyond-L2 rate lies above it "
                                       "is being served from the Infinity Cache in part",
         "contract_bound": "hbm"}
    cs = compute_side(trec, cus) if trec else None
    fr = {"memory_by_traffic": r["frac_traffic"] if traffic else r["frac"]}
    if cs:
        r.update({k: cs[k] for k in ("valu_frac", "valu_frac_range", "lds_frac", "lds_bank_conflict_share", "waves_waiting_share", "clock_GHz", "mfma_frac", "mfma_instructions") if k in cs})
        if "mfma_frac" in cs:
            r["mfma_frac_is"] = "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x %d CUs x GRBM_GUI_ACTIVE / %d XCDs) — rocprofiler's MfmaUtil over the whole call" % (cus, N_XCD)
            fr["mfma"] = cs["mfma_frac"]
        r["valu_frac_is"] = ("SQ_INSTS_VALU x %.1f SIMD cycles per wave64 instruction (tools/probes/valu_rate.hip: %.1f for the add / and / shift class, %.1f for "
                             "v_bfe / v_lshl_or / v_mul_lo / v_bcnt; valu_frac_range = at those two) / (4 SIMDs x %d CUs x GRBM_GUI_ACTIVE / %d XCDs), "
                             "dispatches serialised by the profiler" % (VALU_CYCLES_PER_INST, VALU_CYCLES_RANGE[0], VALU_CYCLES_RANGE[1], cus, N_XCD))
        r["lds_frac_is"] = "SQ_LDS_IDX_ACTIVE / (%d CUs x GRBM_GUI_ACTIVE / %d XCDs)" % (cus, N_XCD)
        fr["valu"] = cs["valu_frac"]
        if "lds_frac" in cs:
            fr["lds"] = cs["lds_frac"]
    fr = {k: v for k, v in fr.items() if v is not None}
    if fr:
        top = max(fr, key=fr.get)
        r["bound"] = {"memory_by_traffic": "beyond-L2", "valu": "valu", "lds": "lds", "mfma": "mfma"}[top]
    r["fractions"] = fr
    if trec:
        r.update({"fetch_multiplier": trec.get("fetch_multiplier"), "multiplier_source": trec.get("multiplier_source"),
                  "kernel_ms_under_pmc": trec.get("kernel_ms_under_pmc"), "l2_hit_rate": trec.get("l2_hit_rate")})
        pk = {}
        for k, v in (trec.get("kernels") or {}).items():
            e = {"dispatches": v.get("dispatches", 1), "bytes": v.get("bytes")}
            kc = compute_side(v, cus)
            if kc:
                e.update({a: kc[a] for a in ("valu_frac", "lds_frac", "lds_bank_conflict_share", "waves_waiting_share", "clock_GHz", "mfma_frac") if a in kc})
                if v.get("dur_ns:GRBM_GUI_ACTIVE"):
                    e["ms_under_pmc"] = v["dur_ns:GRBM_GUI_ACTIVE"] * 1e-6
            pk[k] = e
        r["per_kernel"] = pk
    if extra:
        r.update(extra)
    return r


def measure_traffic(args, rank, workload="tc"):
    """Separate PMC passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2: never in one pass; never combined with tracing).
    Returns {"n1": {...}, "n2": …} or (None, reason)."""
    table, notes = {}, []
    # the memory-side read requests of the L2 by SIZE CLASS: FETCH_SIZE tallies every request at 64 B (MI355X_MICROARCH.md "HBM": exactly half the bytes of a
    # 16-byte-per-lane stream; "other access widths are uncalibrated") — 32 n32 + 64 n64 + 128 n128 is the byte count itself, per kernel, whatever the widths
    passes = ((RDREQ, ["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"]) if workload == "tc" else (RDREQ, ["FETCH_SIZE"], ["WRITE_SIZE"])) + (SQ_PASS, GRBM_PASS) + ((MFMA_PASS,) if workload.startswith("kc") else ())
    for counters in passes:
        t0 = time.perf_counter()
        res, err = run_pmc_pass(args, counters, timeout=args.pmc_timeout, workload=workload)
        log(rank, f"PMC pass {workload} {counters}: {'ok' if res else 'FAILED ' + str(err)} in {time.perf_counter() - t0:.1f}s")
        if res is None:
            notes.append(f"{counters}: {err}")
            if counters[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                return None, "; ".join(notes)
            continue  # (without the size classes the guide's x2 stands in, and the record says so)
        for key, rec in res.items():
            t = table.setdefault(key, {"kernels": {}, "stream_bytes": rec["stream_bytes"]})
            t.update(rec["counters"])
            if counters[0] == "FETCH_SIZE":  # the call the fetch bytes were counted on: its own kernel time travels with them
                t["kernel_ms_under_pmc"] = rec.get("kernel_ms_under_pmc")
            if "result" in rec:
                t["result"] = rec["result"]
            for k, c in rec["kernels"].items():
                t["kernels"].setdefault(k, {}).update(c)
    def read_bytes(rec):
        """(bytes, multiplier on FETCH_SIZE, where it comes from) of one record holding FETCH_SIZE (KB) and, if the pass ran, the request size classes"""
        n, n32, n64, n128 = (rec.get(c) for c in RDREQ)
        fetch = rec["FETCH_SIZE"] * 1024.0
        if n and n32 is not None and abs((n32 + n64 + n128) - n) <= 1e-3 * n + 8:
            b = 32.0 * n32 + 64.0 * n64 + 128.0 * n128
            return b, (b / fetch if fetch > 0 else None), ("TCC_EA0_RDREQ_{32B,64B,128B}_sum of this run: %.2f %% of the requests are 128-byte ones"
                                                          % (100.0 * n128 / n))
        return 2.0 * fetch, 2.0, "MI355X_MICROARCH.md: FETCH_SIZE tallies 128-byte requests at 64 (x2 for 16-byte-per-lane streams); size classes not collected"
    for key, t in table.items():
        # FETCH_SIZE / WRITE_SIZE are reported in KB
        rb, mult, src = read_bytes(t)
        t["read_bytes"], t["fetch_multiplier"], t["multiplier_source"] = rb, mult, src
        t["bytes"] = rb + t["WRITE_SIZE"] * 1024.0
        for k in t["kernels"].values():
            if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
                kb, km, _ = read_bytes(k)
                k["bytes"] = kb + k["WRITE_SIZE"] * 1024.0
                k["fetch_multiplier"] = km
        if "TCC_HIT_sum" in t and t["TCC_HIT_sum"] + t.get("TCC_MISS_sum", 0) > 0:
            t["l2_hit_rate"] = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
    return table, "; ".join(notes) if notes else None


def committed_traffic(khash, key, world):
    """profiles/hbm_traffic.json holds the last profiled numbers, keyed by the kernel hash they were measured on: stale entries are refused."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            doc = json.load(f)
        rec = doc.get("by_kernel_hash", {}).get(khash, {}).get(key, {}).get(f"n{world}")
        return rec
    except (OSError, ValueError):
        return None


T_START = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=26, help="RMAT scale: 26 = the graph BASELINE.json's metric is quoted on; 24 = configs[1]")
    ap.add_argument("--check-scale", type=int, default=24, help="N=1 only: extra untimed-setup pass on this scale, asserted against the "
                                                               "reference golden (configs[1]); 0 disables")
    ap.add_argument("--degree", type=int, default=16)
    ap.add_argument("--generator", default="kronecker")
    ap.add_argument("--algo", default="auto", choices=["auto", "oriented", "full"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="size of the CPU-baseline sample; 0 disables it")
    ap.add_argument("--ref-scale", type=int, default=22, help="scale of the graph the compiled reference is timed on; 0 disables")
    ap.add_argument("--cache-dir", default=os.environ.get("GMSX_CACHE") or os.path.join(tempfile.gettempdir(), f"gmsx_cache_{os.getuid()}"))
    ap.add_argument("--pmc", type=int, default=1, help="N=1: collect HBM traffic with rocprofv3 PMC child passes in this run (0 = use profiles/hbm_traffic.json)")
    ap.add_argument("--pmc-timeout", type=float, default=420.0)
    ap.add_argument("--dump-traffic", default="", help="merge this run's PMC traffic table (all shard counts) into the given JSON file "
                                                       "(the committed fallback profiles/hbm_traffic.json is produced this way)")
    ap.add_argument("--side", type=int, default=1, help="N=1: also run BASELINE configs[2] (k=4 cliques, scale 22) and configs[3] (Bron-Kerbosch, 117M-edge RMAT); 0 disables")
    ap.add_argument("--workload", default="tc", choices=["tc"] + sorted(WORKLOADS), help=argparse.SUPPRESS)
    ap.add_argument("--big", type=int, default=1, help="N=1: also the two north-star-size records — kclique4_s26 (k = 4 on the headline graph, count == the reference's kClist golden) "
                                                      "and tc_s27 (BASELINE configs[4]'s graph on ONE GPU: whole pass + its eight shards); 0 disables")
    ap.add_argument("--big-budget-s", type=float, default=330.0, help="tc_s27 (≈ 2 min of host-side graph generation) is skipped when the run is already older than this")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    from gms_amd import capi, dist
    rank, local_rank, world = dist.env_rank_world()
    khash = kernel_hash()
    gkey = f"{args.generator}-{args.scale}-{args.degree}/{args.algo}"

    # ---- synthetic input: the reference loader's "-g kronecker <scale> --deg <degree>" graph, bit-identical ------
    # host threads: the cores this container may really use (cgroup quota, else the affinity mask) — hundreds of OpenMP
    # threads on a 16-core quota make the generator 2x slower, and torch.distributed.run hands its workers OMP_NUM_THREADS=1
    ncores = host_cores()
    capi.set_host_threads(ncores)
    t0 = time.perf_counter()
    sg = sg_cache_path(args)
    csr = None
    if rank == 0:
        os.makedirs(args.cache_dir, mode=0o700, exist_ok=True)
        if os.path.exists(sg):  # a cache left by an earlier run on this box (N = 1, 2, 4, 8 back to back)
            try:
                csr = capi.HostCSR.load(sg, relabel=capi.RELABEL_NEVER)  # offsets / id range validated by the reader
            except capi.GmsxError as e:
                log(rank, f"cache {sg} unusable ({e}); regenerating")
        if csr is None:
            csr = capi.HostCSR.generate(args.generator, args.scale, args.degree, capi.RELABEL_AUTO)
            if os.environ.get("GMSX_NO_CACHE") != "1" or world > 1:
                try:
                    fd, tmp = tempfile.mkstemp(prefix=".sg_", dir=args.cache_dir)
                    os.close(fd)
                    csr.save_sgx(tmp)
                    os.replace(tmp, sg)
                    if world > 1:  # rank 0 drops its private copy and maps the cache like the other ranks: ONE copy of the CSR in host memory
                        csr = None
                        csr = capi.HostCSR.load(sg, relabel=capi.RELABEL_NEVER)
                except (OSError, capi.GmsxError) as e:  # a full or read-only cache directory is not an error of the benchmark
                    log(rank, f"cache not written: {e}")

    # ---- HBM traffic: PMC passes in child processes, before this process touches the GPU (N = 1) ----------------------------
    traffic_table, traffic_note, traffic_source = None, None, None
    if rank == 0 and world == 1 and args.pmc and os.path.exists(sg):
        traffic_table, traffic_note = measure_traffic(args, rank)
        if traffic_table:
            traffic_source = "rocprofv3 --pmc child passes of this run (memory-side read requests by size class + WRITE_SIZE, separate passes)"
            if args.dump_traffic:
                try:
                    with open(args.dump_traffic) as f:
                        doc = json.load(f)
                except (OSError, ValueError):
                    doc = {}
                doc["_comment"] = ("beyond-L2 bytes per launch of the triangle-count pass, memory-side read requests by size class (32 n32 + 64 n64 + 128 n128) + WRITE_SIZE from separate rocprofv3 --pmc "
                                   "passes (bench.py --dump-traffic), keyed by the hash of the kernel sources they were measured on (bench.kernel_hash); "
                                   "nK = shard 0 of K on one GPU.  bench.py measures live and uses this file only as a fallback, never across hashes.")
                doc.setdefault("by_kernel_hash", {}).setdefault(khash, {})[gkey] = traffic_table
                with open(args.dump_traffic, "w") as f:
                    json.dump(doc, f, indent=1)
            try:  # the shard figures serve the N = 2, 4, 8 runs that follow on this box
                with open(os.path.join(args.cache_dir, f"traffic_{khash}_{gkey.replace('/', '_')}.json"), "w") as f:
                    json.dump(traffic_table, f)
            except OSError:
                pass

    side_traffic = {}
    if rank == 0 and world == 1:
        for name in side_names(args):
            try:
                workload_csr(capi, args, name)  # generated on the host and cached for the children (and for the leg below)
                if args.pmc:
                    side_traffic[name], note = measure_traffic(args, rank, workload=name)
                    if note:
                        log(rank, f"PMC {name}: {note}")
            except Exception as e:  # noqa: BLE001 - a side record must never break the headline
                log(rank, f"side workload {name}: PMC setup failed: {e!r}")

    import torch
    # test hook for 1-GPU boxes: GMSX_SHARE_GPU=1 lets several ranks share cuda:0 (then over gloo, RCCL refuses duplicates)
    share = os.environ.get("GMSX_SHARE_GPU") == "1"
    rank, local_rank, world = dist.init_process_group(backend="gloo" if share else None)
    if os.environ.get("GMSX_BENCH_TEST_DIE_RANK") == str(rank) and world > 1:
        os._exit(5)  # test hook (tests/test_multi_gpu_readiness_gpu.py): this rank dies before the communicator id reaches it
    if world != args.gpus:
        log(rank, f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if share:
        local_rank %= max(torch.cuda.device_count(), 1)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: cuda:{local_rank} does not exist ({torch.cuda.device_count()} devices visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    t_i0 = time.perf_counter()
    capi.init(local_rank)  # binds the device, loads the library's code object, pins the upload's two staging buffers: once per process
    t_init = time.perf_counter() - t_i0
    capi.set_stream(torch.cuda.current_stream().cuda_stream)
    info = capi.device_info()
    ceiling = None
    if rank == 0 and not share:
        try:  # the box's own read-stream ceiling, before the graph takes the memory: 8 GiB swept 40 times, ~60 ms
            ceiling = capi.hbm_read_probe(8 << 30, 40)
            log(rank, f"read-only stream ceiling of this device: {ceiling:.0f} GB/s")
        except capi.GmsxError as e:
            log(rank, f"stream ceiling probe failed: {e}")
    algo = {"auto": capi.TC_AUTO, "oriented": capi.TC_ORIENTED, "full": capi.TC_FULL}[args.algo]
    divisor = capi.lib().gmsx_tc_divisor(algo)

    anon_delta = 0
    if world > 1:
        dist.barrier()  # rank 0 has written the cache
        if csr is None:
            capi.set_host_threads(max(1, ncores // world))  # every rank: its share of the cores for the host-side bookkeeping
            a0 = host_memory()["RssAnon"]
            csr = capi.HostCSR.load(sg, relabel=capi.RELABEL_NEVER)
            anon_delta = host_memory()["RssAnon"] - a0  # what holding the CSR adds to this rank's PRIVATE memory: ~0 for a mapped cache
    t_gen = time.perf_counter() - t0
    n, m, nnz = csr.num_nodes, csr.num_edges, csr.nnz
    elems = csr.merge_elements()
    t0 = time.perf_counter()
    # H2D + validation (sorted, loop-free, symmetric) + DAG containers + bitsets; for N > 1 a SHARDED upload: this rank's pivots' task lists
    # and inline rows only (gmsx_graph_upload_csr_shard), so the container build and its memory shrink with N
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_DEFAULT, shard=(rank, world) if world > 1 else None)
    torch.cuda.synchronize()
    t_upload_base = time.perf_counter() - t0
    base_bytes = g.device_bytes
    t0 = time.perf_counter()
    g.prepare(capi.PREPARE_TC)                                     # the triangle-count containers: stream rows, inline rows, task lists
    torch.cuda.synchronize()
    t_build_tc = time.perf_counter() - t0
    t_upload = t_upload_base + t_build_tc
    log(rank, f"{info['name']}: graph n={n} m={m} Σ(du+dv)={elems} generate/load(+PMC passes) {t_gen:.1f}s upload {t_upload_base:.2f}s "
              f"+ TC containers {t_build_tc:.2f}s max d+={g.max_out_degree} device bytes={g.device_bytes} (base {base_bytes})")

    # ---- the one collective: native RCCL communicator under the C-ABI; torch.distributed only carries the 128-byte id ------
    comm = None
    if world > 1 and not share:
        idt = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.Comm.unique_id()), dtype=torch.uint8))
        torch.distributed.broadcast(idt, 0)
        try:
            comm = capi.Comm.init(rank, world, bytes(idt.cpu().numpy().tobytes()))  # bounded: GMSX_COMM_TIMEOUT_S (a peer that never arrives)
        except capi.GmsxError as e:
            log(0, f"rank {rank}: {e}: leaving (the launcher takes the other ranks down)")
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(3)  # not sys.exit: after a timeout a helper thread is parked inside RCCL, and no exec — a fresh exit status is all the launcher needs
        assert (comm.rank, comm.size) == (rank, world), (comm.rank, comm.size, rank, world)
        collective = "gmsx_comm_allreduce_u64 = ncclAllReduce(count=1, ncclUint64, ncclSum) over RCCL (native, librccl), %d ranks" % comm.size
    elif world > 1:
        collective = "torch.distributed gloo all-reduce (GMSX_SHARE_GPU test hook: ranks share one GPU, RCCL refuses duplicates)"
    else:
        collective = "none (single rank)"

    def step():
        partial, st = g.tc_partial(rank, world, algo, stats=True)
        if comm is not None:
            total = comm.allreduce_u64(partial)
        else:
            total = dist.allreduce_count(partial, None if share else dev)
        return total, st

    t0 = time.perf_counter()
    _, st_first = step()  # the very first pass after the upload (cold caches, first launches): with the upload it is the one-shot cost
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    for _ in range(args.warmup - 1):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms, totals = [], []
    for _ in range(args.steps):
        total, st = step()
        kernel_ms.append(st["kernel_ms"])
        totals.append(total)
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = dist.allreduce_max(time.perf_counter() - t0, None if share else dev)

    assert len(set(totals)) == 1 and totals[0] % divisor == 0, totals
    triangles = totals[0] // divisor
    golden, golden_src = golden_triangles(args.generator, args.scale, args.degree)
    if golden is not None:
        assert triangles == golden, f"PARITY FAILURE: {triangles} != reference golden {golden}"
        parity = f"== reference golden ({golden_src or 'tests/golden/graphs.json'})"
    else:
        parity = "no reference golden at this size (self-consistency only: every step returned the same count)"

    ms_per_step = 1e3 * elapsed / args.steps
    value = m * args.steps / elapsed
    avg_kernel_ms = dist.allreduce_max(sum(kernel_ms) / len(kernel_ms), None if share else dev)
    t_kernel = avg_kernel_ms * 1e-3
    b_alg = 4 * elems + 8 * (n + 1) + 4 * nnz           # bytes the reference operator streams per pass
    stream_bytes = int(st["stream_bytes"])               # this rank's launch, oriented formulation, no reuse assumed

    trec = None
    if traffic_table and f"n{world}" in traffic_table:
        trec = traffic_table[f"n{world}"]
    else:
        for path in (os.path.join(args.cache_dir, f"traffic_{khash}_{gkey.replace('/', '_')}.json"),):
            try:
                with open(path) as f:
                    trec = json.load(f).get(f"n{world}")
                if trec is not None:
                    traffic_source = ("rocprofv3 --pmc passes of the N=1 run on this box, same kernel build: the bytes are those of SHARD 0 of %d measured "
                                      "on one GPU; kernel_ms is the max over this run's ranks" % world)
            except (OSError, ValueError):
                pass
        if trec is None:
            trec = committed_traffic(khash, gkey, world)
            if trec is not None:
                traffic_source = "profiles/hbm_traffic.json (measured on this kernel build: hash %s)" % khash
    traffic = trec["bytes"] if trec else None
    if traffic is None:
        traffic_source = "none for this kernel build (" + str(traffic_note or "no PMC pass, no cached table") + "): traffic is null"
    roofline = make_roofline(stream_bytes, t_kernel, trec, info["compute_units"], ceiling, extra={
        "traffic_source": traffic_source, "kernel_ms": avg_kernel_ms, "kernel_hash": khash,
        "kernel": "k_tc_items<false> (hub items) + k_tc_items<true> (tail items): one PERSISTENT launch per item queue — a pivot's row part in LDS, "
                  "the stream rows its task-list entries name streamed through it; ~95 % of the bytes and of the time — + k_tc_light (edges between two "
                  "light vertices, all-pairs in registers), one after the other on the launch stream; kernel_ms is the HIP-event wall time of the "
                  "whole pass; traffic and the compute-side counters are summed over the three (per_kernel splits them)",
        "algorithmic_GBps": stream_bytes / t_kernel / 1e9,
        "reference_equivalent_bytes": b_alg / world, "reference_equivalent_GBps": b_alg / world / t_kernel / 1e9,
        "reference_equivalent_note": "B_alg = 4*sum_{u<v}(d_u+d_v) + 8(n+1) + 4*nnz (SURVEY 8(d)): what the reference's full-row merges stream; "
                                     "not a hardware utilisation — the oriented kernels probe far fewer ids",
        "probes_per_launch": st["probes"], "graph_device_bytes": g.device_bytes})

    out = {
        "metric": METRIC if args.scale == 26 else METRIC.replace("RMAT-26", f"RMAT-{args.scale}"), "value": value, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32 ids / uint64 counts",
        "data": "synthetic",
        "config": {"workload": f"triangle count, RMAT scale-{args.scale} ef={args.degree} ({args.generator}, GAPBS generator "
                               f"seed 27491095, symmetrised, de-duplicated, relabelled by degree)",
                   "n": n, "m": m, "nnz": nnz, "algo": args.algo, "parallelism": f"edge-shard x{world} + 1 all-reduce(u64)",
                   "collective": collective, "triangles": triangles, "parity": parity, "device": info["name"]},
        "roofline": roofline,
        "setup_s": {"library_init": t_init, "generate_or_load_incl_pmc_passes": t_gen, "upload_and_build": t_upload, "upload_base": t_upload_base, "build_tc_containers": t_build_tc,
                    "first_pass": t_first},
        # what a caller that counts ONCE pays (the reference times only kernel(sgraph), common/benchmark.h:105-116; its SetGraph build is
        # likewise outside): host CSR -> H2D -> containers -> task lists -> first pass.  Never `value`.
        "upload_s": t_upload, "one_shot_edges_per_s": m / world / (t_upload + t_first),
        "one_shot_note": "m / (upload_base + build_tc_containers + first_pass) per rank; the upload builds the immutable device set graph once (the inline "
                         "rows materialise the members below v of every light pivot — the per-edge decisions — and are reused by every pass; every pass "
                         "still performs every membership probe)",
        "graph_device_bytes": {"base": int(base_bytes), "with_tc_containers": int(g.device_bytes)},
        # N > 1: every rank builds the triangle-count containers of ITS pivots only (gmsx_graph_upload_csr_shard); rank 0's figures above
        "upload": {"sharded": world > 1, "shard": [rank, world] if world > 1 else None},
    }
    # host memory of the ranks: the CSR every rank uploads from is a MAPPING of the one cache file for N > 1 (gmsx_csr_load of
    # ".sgx"), i.e. file pages shared through the page cache (RssFile), not N private copies (RssAnon)
    hm = host_memory()
    out["host_memory"] = {"csr_mapped": bool(csr.is_mapped), "csr_bytes": int(8 * (n + 1) + 4 * nnz),
                          "rss_anon_bytes_max_over_ranks": int(dist.allreduce_max(float(hm["RssAnon"]), None if share else dev)),
                          "rss_file_bytes_max_over_ranks": int(dist.allreduce_max(float(hm["RssFile"]), None if share else dev)),
                          "peak_rss_bytes_max_over_ranks": int(dist.allreduce_max(float(hm["VmHWM"]), None if share else dev)),
                          "csr_load_rss_anon_delta_max_over_ranks": int(dist.allreduce_max(float(anon_delta), None if share else dev)),
                          "rank0": hm, "note": "RssAnon = private memory of a rank; RssFile = mapped file pages (the shared CSR cache, libraries)"}
    if comm is not None:
        comm.finalize()
    if rank == 0 and world == 1 and args.check_scale > 0 and args.check_scale != args.scale:
        g.free()
        g = None
        out["config1_check"] = config1_check(capi, args.generator, args.check_scale, args.degree, algo, divisor)
    if rank == 0 and world == 1:
        if g is not None:
            g.free()
            g = None
        for name in side_names(args):
            try:
                out[WORKLOADS[name]["key"]] = side_workload(capi, args, name, side_traffic.get(name), rank, info["compute_units"], ceiling)
            except AssertionError:
                raise
            except Exception as e:  # noqa: BLE001
                out[WORKLOADS[name]["key"]] = {"error": repr(e)}
    if rank == 0 and world == 1 and args.side:
        try:
            out["vertex_count2_s22"] = vertex_count2_record(capi, args, rank)
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001
            out["vertex_count2_s22"] = {"error": repr(e)}
    cpu_cache = os.path.join(args.cache_dir, f"cpu_baseline_{args.generator}-{args.scale}-{args.degree}.json")
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        if g is not None:
            g.free()
        port = cpu_baseline(csr, args.cpu_seconds)
        ref = reference_baseline(args.ref_scale, args.degree, golden_record(f"{args.generator}-{args.scale}-{args.degree}-relabel")) if args.ref_scale > 0 else None
        if ref and "value" in ref:
            # the headline CPU figure is the reference's own FASTEST code on this host — the compiled reference, Par::count_total<RoaringGraph>, whole graph
            # of the stated scale (not the flattering one) —; the SortedSet-merge port on a sample of the headline graph sits beside it
            out["cpu_baseline"] = dict(ref, sortedset_port=port, roaring=ref)
        else:
            out["cpu_baseline"] = dict(port, roaring=ref)
        out["cpu_reference"] = ref  # same record under its round-2 name
        try:  # the N = 2, 4, 8 runs that follow on this box carry it (a CPU leg per rank count would measure the same thing again)
            with open(cpu_cache, "w") as f:
                json.dump(out["cpu_baseline"], f)
        except OSError:
            pass
    if rank == 0 and world == 1 and args.big:
        csr = None  # the headline graph's host copy goes before the scale-27 one is generated
        try:
            out["tc_s27"] = tc_s27_record(capi, args, algo, divisor, rank, T_START)
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001
            out["tc_s27"] = {"error": repr(e)}
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        pass
    elif rank == 0:
        out["cpu_baseline"] = None
        try:
            with open(cpu_cache) as f:
                out["cpu_baseline"] = dict(json.load(f), measured_by="the N=1 run of bench.py on this box (cached in --cache-dir); not re-timed at N>1")
        except (OSError, ValueError):
            pass
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()


if __name__ == "__main__":
    main()
