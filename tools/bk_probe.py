"""GPU probe: Bron-Kerbosch on the BASELINE configs[3] graph (or `scale ef`): count vs golden, kernel time, rounds.
usage: bk_probe.py [scale ef] [--default-only]   (--default-only: skip the BK_SPLIT_BUILD = 1 / 0 comparison runs, e.g. under a profiler)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi
default_only = "--default-only" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
scale, ef = (int(argv[0]), int(argv[1])) if len(argv) > 1 else (21, 56)
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "graphs.json")))
csr = capi.HostCSR.generate_rmat(scale, ef, 0.45, 0.22, 0.22)
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
for knobs in (({},) if default_only else ({}, {"BK_SPLIT_BUILD": "1"}, {"BK_SPLIT_BUILD": "0"}, {})):
    capi.set_option("BK_SPLIT_BUILD", knobs.get("BK_SPLIT_BUILD"))
    ms = []
    for _ in range(3):
        total, st = g.bk_count(stats=True)
        ms.append(round(st["kernel_ms"], 1))
    gold = GOLD.get("rmat-%d-%d-a45-b22-c22" % (scale, ef), {}).get("bk")
    print(json.dumps({"graph": [scale, ef], "m": csr.num_edges, "knobs": knobs, "maximal_cliques": total, "golden_ok": (total == gold) if gold else None, "kernel_ms": ms,
                      "rounds": st["probes"], "launches": st["launches"]}), flush=True)
capi.set_option("BK_SPLIT_BUILD", None)
