"""RMAT scale 27 (BASELINE.json configs[4] size; there sharded over 8 GPUs) on ONE MI355X: 2^31 generated edges, so every
edge index of the host substrate and every container offset on the device has to be 64-bit clean.  Guarded: needs
>= 200 GB of free host memory.  Cross-check: 8 shards vs total (no golden exists at this size)."""
import sys, time, json
sys.path.insert(0, ".")
avail = 0
for line in open("/proc/meminfo"):
    if line.startswith("MemAvailable"):
        avail = int(line.split()[1]) / 1e6
print(json.dumps({"host_mem_available_GB": round(avail, 1)}), flush=True)
if avail < 200:
    raise SystemExit("not enough host memory for scale 27: skipped")
from gms_amd import capi
capi.init(0)
t0 = time.time(); csr = capi.HostCSR.generate("kronecker", 27); t1 = time.time()
print(json.dumps({"scale": 27, "n": csr.num_nodes, "m": csr.num_edges, "gen_s": round(t1 - t0, 1)}), flush=True)
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED); t2 = time.time()
ms = []
for _ in range(3):
    tri, st = g.tc_total(stats=True); ms.append(round(st["kernel_ms"], 1))
parts = [g.tc_partial(p, 8, stats=True) for p in range(8)]
print(json.dumps({"scale": 27, "triangles": tri, "partials_ok": sum(p[0] for p in parts) == tri, "kernel_ms": ms, "shard_ms": [round(p[1]["kernel_ms"], 1) for p in parts],
                  "upload_s": round(t2 - t1, 2), "Gedges_s": round(csr.num_edges / (min(ms) / 1e3) / 1e9, 2), "max_dplus": g.max_out_degree,
                  "device_GB": round(g.device_bytes / 1e9, 2)}), flush=True)
try:
    o4, c4, st4 = g.kclique_count(4, stats=True)
    print(json.dumps({"scale": 27, "k": 4, "cliques": c4, "kernel_ms": round(st4["kernel_ms"], 1)}), flush=True)
except capi.GmsxError as e:
    print(json.dumps({"scale": 27, "k": 4, "error": str(e)}), flush=True)
