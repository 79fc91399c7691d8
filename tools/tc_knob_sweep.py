"""GPU probe: pass time of the triangle count at one RMAT scale under sets of build / launch knobs.
usage: python tools/tc_knob_sweep.py 26 "TC_GAP12=1" "TC_DELTA_PCT=100" "TC_GAP12=1,TC_DELTA_PCT=100" …   ("" = defaults)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi  # noqa: E402

scale = int(sys.argv[1])
configs = sys.argv[2:] or [""]
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
csr = capi.HostCSR.generate("kronecker", scale)
base = None
for cfg in [""] + [c for c in configs if c]:
    kv = dict(x.split("=", 1) for x in cfg.split(",") if x)
    for k, v in kv.items():
        capi.set_option(k, v)
    try:
        g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED | capi.UPLOAD_FOR_TC)
        ms = []
        for _ in range(5):
            t, st = g.tc_total(stats=True)
            ms.append(round(st["kernel_ms"], 2))
        base = t if base is None else base
        print(json.dumps({"scale": scale, "knobs": cfg or "(defaults)", "same_count": t == base, "kernel_ms": ms, "best": min(ms),
                          "stream_GB": round(st["stream_bytes"] / 1e9, 1), "probes_G": round(st["probes"] / 1e9, 1), "device_GB": round(g.device_bytes / 1e9, 1)}), flush=True)
        g.free()
    finally:
        capi.reset_options()
