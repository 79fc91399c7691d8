"""Two PROCESSES, each with its own upload of the same graph, each computing one shard: the sum must be the count (the task lists and
inline rows must come out identical in every process).  usage: tc_two_process_shards.py <scale> [child part nparts]"""
import json, os, subprocess, sys
sys.path.insert(0, ".")
if len(sys.argv) > 2:
    from gms_amd import capi
    capi.init(0)
    g = capi.DeviceGraph.from_csr(capi.HostCSR.generate("kronecker", int(sys.argv[1])), flags=capi.UPLOAD_TRUSTED)
    part, st = g.tc_partial(int(sys.argv[2]), int(sys.argv[3]), stats=True)
    print(json.dumps({"partial": part, "units": st["units"]}))
    sys.exit(0)
scale, nparts = int(sys.argv[1]), 3
outs = [json.loads(subprocess.run([sys.executable, __file__, str(scale), str(p), str(nparts)], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
        for p in range(nparts)]
from gms_amd import capi
capi.init(0)
csr = capi.HostCSR.generate("kronecker", scale)
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
t = g.tc_total()
print(json.dumps({"scale": scale, "total": t, "sum_of_process_shards": sum(o["partial"] for o in outs), "units": sum(o["units"] for o in outs), "m": csr.num_edges,
                  "ok": t == sum(o["partial"] for o in outs) and csr.num_edges == sum(o["units"] for o in outs)}))
