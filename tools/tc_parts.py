"""Strong-scaling preview on one GPU: kernel time of every shard p/N of the scale-24 triangle count, N = 1, 2, 4, 8."""
import sys, json, time
sys.path.insert(0, ".")
from gms_amd import capi
s = int(sys.argv[1]) if len(sys.argv) > 1 else 24
capi.init(0)
g = capi.DeviceGraph.from_csr(capi.HostCSR.generate("kronecker", s), flags=capi.UPLOAD_TRUSTED)
for n in (1, 2, 4, 8):
    for _ in range(2):
        ms, wall = [], []
        for p in range(n):
            t0 = time.perf_counter(); _, st = g.tc_partial(p, n, stats=True); wall.append(round(1e3 * (time.perf_counter() - t0), 2)); ms.append(round(st["kernel_ms"], 2))
    print(json.dumps({"nparts": n, "kernel_ms": ms, "wall_ms": wall, "max": max(ms), "ideal": round(sum(ms) / n, 2)}), flush=True)
