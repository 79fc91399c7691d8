"""GPU probe: triangle count on RMAT scale(s) — count vs the reference golden, best-of-N pass time, upload / container-build times.
usage: python tools/tc_probe.py 22 24 26 [--passes 5] [--shards 8]   (--shards N: also the kernel time of every shard of N on this one GPU)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi  # noqa: E402

opts = {}
args = []
it = iter(sys.argv[1:])
for a in it:
    if a.startswith("--"):
        opts[a] = int(next(it))
    else:
        args.append(a)
passes = opts.get("--passes", 5)
nshards = opts.get("--shards", 0)
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "graphs.json")) as f:
    GOLD = json.load(f)
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
for s in (int(a) for a in args):
    csr = capi.HostCSR.generate("kronecker", s)
    t0 = time.perf_counter()
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    g.prepare(capi.PREPARE_TC)
    t_tc = time.perf_counter() - t0
    ms = []
    for _ in range(passes):
        t, st = g.tc_total(stats=True)
        ms.append(round(st["kernel_ms"], 3))
    gold = GOLD.get("kronecker-%d-16-relabel" % s, {}).get("triangles")
    shards = sum(g.tc_partial(p, 3) for p in range(3))
    shard_ms = None
    if nshards > 1:
        parts = [min((g.tc_partial(p, nshards, stats=True) for _ in range(3)), key=lambda r: r[1]["kernel_ms"]) for p in range(nshards)]
        assert sum(p[0] for p in parts) == t
        shard_ms = [round(p[1]["kernel_ms"], 2) for p in parts]
    print(json.dumps({"scale": s, "m": csr.num_edges, "shard_kernel_ms": shard_ms, "triangles": t, "golden_ok": (t == gold) if gold is not None else None, "shards_ok": shards == t,
                      "units_ok": st["units"] == csr.num_edges, "kernel_ms": ms, "best_G_edges_per_s": round(csr.num_edges / min(ms) / 1e6, 3),
                      "upload_s": round(t_up, 3), "build_tc_s": round(t_tc, 3), "stream_GB": round(st["stream_bytes"] / 1e9, 2), "probes_G": round(st["probes"] / 1e9, 2),
                      "device_GB": round(g.device_bytes / 1e9, 2)}), flush=True)
    g.free()
    del csr
