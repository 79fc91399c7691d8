"""GPU probe: k-clique count on RMAT scale(s) — count vs the reference golden, best-of-3 kernel time, lean upload time.
usage: python tools/kc_probe.py 22 24 [--k 4] [--ab]     (--ab: every scale also with KC_REVERSE = 0, the forward-only BUILD of rounds 1-5)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi  # noqa: E402

k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 4
scales = [int(a) for a in sys.argv[1:] if a.isdigit() and (sys.argv[sys.argv.index(a) - 1] != "--k")]
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "graphs.json")))
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
for s in scales:
    csr = capi.HostCSR.generate("kronecker", s)
    for reverse in ((1, 0) if "--ab" in sys.argv else (1,)):
        capi.set_option("KC_REVERSE", None if reverse else 0)
        t0 = time.perf_counter()
        g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
        t_up = time.perf_counter() - t0
        ms, setup = [], []
        for _ in range(3):
            ordered, cliques, st = g.kclique_count(k, stats=True)
            ms.append(round(st["kernel_ms"], 2))
            setup.append(round(st["setup_ms"], 1))
        rec = GOLD.get("kronecker-%d-16-relabel" % s, {})
        gold = rec.get("kc%d" % k) if ("kc%d" % k) in rec else (24 * rec["kc4_true"] if k == 4 and "kc4_true" in rec else None)
        print(json.dumps({"scale": s, "k": k, "reverse_rows": bool(reverse), "ordered": ordered, "cliques": cliques, "golden_ok": (ordered == gold) if gold is not None else None,
                          "kernel_ms": ms, "setup_ms": setup, "stream_GB": round(st["stream_bytes"] / 1e9, 1), "upload_s": round(t_up, 3), "launches": st["launches"],
                          "device_GB": round(g.device_bytes / 1e9, 2)}), flush=True)
        g.free()
    capi.set_option("KC_REVERSE", None)
