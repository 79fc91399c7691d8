"""GPU probe: co-scheduling of the heavy- and light-pivot triangle kernels (tc.hip, GMSX_TC_OVERLAP / _HUB_WGS / _WAVE_WGS are read
once per process, so every configuration runs in its own child process on the cached graph).  usage: tc_overlap_sweep.py [scale]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from gms_amd import capi
    capi.init(0)
    csr = capi.HostCSR.load("/tmp/tc_overlap_%d.sg" % scale, relabel=capi.RELABEL_NEVER)
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
    ms = []
    for _ in range(5):
        tri, st = g.tc_total(stats=True)
        ms.append(st["kernel_ms"])
    print(json.dumps({"cfg": os.environ.get("CFG"), "triangles": tri, "kernel_ms": round(min(ms[1:]), 2), "all": [round(x, 1) for x in ms]}), flush=True)
    sys.exit(0)
from gms_amd import capi
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max": capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
path = "/tmp/tc_overlap_%d.sg" % scale
if not os.path.exists(path):
    capi.HostCSR.generate("kronecker", scale, 16).save_sg(path)
for cfg in [("0", 1, 2), ("1", 1, 2), ("1", 1, 1), ("1", 2, 2), ("1", 1, 3), ("1", 2, 3), ("1", 2, 1)]:
    env = dict(os.environ, GMSX_TC_OVERLAP=cfg[0], GMSX_TC_HUB_WGS=str(cfg[1]), GMSX_TC_WAVE_WGS=str(cfg[2]), CFG="overlap=%s hub=%d wave=%d" % cfg)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(scale), "child"], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-300:], flush=True)
