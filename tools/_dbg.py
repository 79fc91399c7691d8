import sys, os
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gms_amd import capi
capi.init(0)
for kind, scale, deg in (("kronecker", 15, 8), ("kronecker", 16, 16)):
    for relabel in (True, False):
        csr = capi.HostCSR.generate(kind, scale, deg, relabel=int(relabel))
        for ts in ("1", "0"):
            os.environ["GMSX_TC_TWO_SIDED"] = ts
            g = capi.DeviceGraph.from_csr(csr)
            t, st = g.tc_total(stats=True)
            print(kind, scale, deg, relabel, "two_sided", ts, t, st["units"], csr.num_edges, g.max_out_degree)
            g.free()
