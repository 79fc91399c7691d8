"""GPU probe: k-clique count on RMAT scale(s) — count vs the reference golden, best-of-3 kernel time, lean upload time.
usage: python tools/kc_probe.py 22 24 [--k 4] [--ab]     (--ab: every scale also with KC_MFMA = 0, the k = 4 count by AND + popcount inside the BUILD kernels, and
with KC_REVERSE = 0, the forward-only BUILD of rounds 1-5)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi  # noqa: E402

k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 4
scales = [int(a) for a in sys.argv[1:] if a.isdigit() and (sys.argv[sys.argv.index(a) - 1] not in ("--k", "--opts"))]
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "graphs.json")))
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
for s in scales:
    cache = "/tmp/gmsx_probe_cache/kronecker-%d-16.sgx" % s  # (a second probe process on the same box maps it instead of generating again)
    if os.path.exists(cache):
        csr = capi.HostCSR.load(cache, relabel=capi.RELABEL_NEVER)
    else:
        csr = capi.HostCSR.generate("kronecker", s)
        try:
            os.makedirs(os.path.dirname(cache), exist_ok=True)
            csr.save_sgx(cache + ".tmp")
            os.replace(cache + ".tmp", cache)
        except (OSError, capi.GmsxError):
            pass
    variants = [{}] + ([{"KC_MFMA": 0}, {"KC_REV_TAIL": 0}, {"KC_REVERSE": 0}] if "--ab" in sys.argv else [])
    if "--opts" in sys.argv:  # --opts "KC_REV_FACTOR=10;KC_REV_FACTOR=30,KC_REV_MIN=16": one more run per ';'-separated option set
        variants += [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[sys.argv.index("--opts") + 1].split(";")]
    for opts in variants:
        for kk, vv in opts.items():
            capi.set_option(kk, vv)
        reverse = str(opts.get("KC_REVERSE", 1)) != "0"
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
        print(json.dumps({"scale": s, "k": k, "reverse_rows": bool(reverse), "options": opts, "ordered": ordered, "cliques": cliques, "golden_ok": (ordered == gold) if gold is not None else None,
                          "kernel_ms": ms, "setup_ms": setup, "stream_GB": round(st["stream_bytes"] / 1e9, 1), "upload_s": round(t_up, 3), "launches": st["launches"],
                          "device_GB": round(g.device_bytes / 1e9, 2)}), flush=True)
        g.free()
        for kk in opts:
            capi.set_option(kk, None)  # (not reset_options: the GMSX_OPT_* of the environment stay)
