"""Separate PROCESSES, each with its own upload of the same graph, each computing one shard: the sums must be the count.  Every child runs
the shard twice — on a full upload (`gmsx_tc_partial(part, nparts)`) and on a SHARDED upload (gmsx_graph_upload_csr_shard: only that
rank's task lists and inline rows exist) — and reports both partial counts, the bookkeeping units and the device bytes of both graphs.
usage: tc_two_process_shards.py <scale> [child part nparts]"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    from gms_amd import capi
    capi.init(0)
    csr = capi.HostCSR.generate("kronecker", int(sys.argv[1]))
    p, n = int(sys.argv[2]), int(sys.argv[3])
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
    part, st = g.tc_partial(p, n, stats=True)
    full_bytes = g.device_bytes
    g.free()
    gs = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED | capi.UPLOAD_FOR_TC, shard=(p, n))
    part_s, st_s = gs.tc_partial(p, n, stats=True)
    refused = False
    try:
        gs.tc_partial((p + 1) % n, n)
    except capi.GmsxError as e:
        refused = e.status == capi.ERR_INVALID
    sharded_bytes = gs.device_bytes  # (before the k-clique call: its reverse-row lists join the graph's device bytes on first use)
    kc = gs.kclique_partial(4, (p + 1) % n, n)  # the base containers of a sharded upload are complete: any k-clique shard works
    print(json.dumps({"partial": part, "units": st["units"], "partial_sharded_upload": part_s, "units_sharded_upload": st_s["units"],
                      "bytes_full": full_bytes, "bytes_sharded": sharded_bytes, "other_shard_refused": refused, "kc4_other_shard": kc}))
    sys.exit(0)
scale, nparts = int(sys.argv[1]), 3
outs = [json.loads(subprocess.run([sys.executable, __file__, str(scale), str(p), str(nparts)], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
        for p in range(nparts)]
from gms_amd import capi
capi.init(0)
csr = capi.HostCSR.generate("kronecker", scale)
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
t = g.tc_total()
kc = g.kclique_count(4)[1]
rec = {"scale": scale, "total": t, "sum_of_process_shards": sum(o["partial"] for o in outs), "units": sum(o["units"] for o in outs), "m": csr.num_edges,
       "sum_of_sharded_uploads": sum(o["partial_sharded_upload"] for o in outs), "units_sharded_uploads": sum(o["units_sharded_upload"] for o in outs),
       "bytes_full": outs[0]["bytes_full"], "bytes_sharded": [o["bytes_sharded"] for o in outs], "refused": all(o["other_shard_refused"] for o in outs),
       "kc4": kc, "kc4_sum_of_shards": sum(o["kc4_other_shard"] for o in outs)}
rec["ok"] = (t == rec["sum_of_process_shards"] == rec["sum_of_sharded_uploads"] and csr.num_edges == rec["units"] == rec["units_sharded_uploads"]
             and rec["refused"] and kc == rec["kc4_sum_of_shards"])
print(json.dumps(rec))
