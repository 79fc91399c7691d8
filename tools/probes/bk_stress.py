"""GPU probe: Bron-Kerbosch on extreme structured graphs (cocktail party, K_{3 x 11}, K_450, K_513) with the default kernels, tiny budgets and the one-search-per-wave kernel."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.getcwd())
from gms_amd import capi
capi.init(0)
def graph(edges, n=None):
    e = np.array(edges, dtype=np.int32)
    return capi.HostCSR.from_edges(e[:, 0].copy(), e[:, 1].copy()) if n is None else capi.HostCSR.from_edges(e[:, 0].copy(), e[:, 1].copy(), num_nodes=n)
def complete_multipartite(parts):
    off = np.cumsum([0] + parts)
    ed = []
    for a in range(len(parts)):
        for b in range(a + 1, len(parts)):
            for x in range(off[a], off[a + 1]):
                for y in range(off[b], off[b + 1]):
                    ed.append((x, y))
    return graph(ed)
cases = [("cocktail party 18 pairs", complete_multipartite([2] * 18), 2 ** 18),
         ("K_{3 x 11}", complete_multipartite([3] * 11), 3 ** 11),
         ("K_450", complete_multipartite([1] * 450), 1),
         ("K_513", complete_multipartite([1] * 513), 1),
         ("K_{2 x 12} + K_{1 x 300}", None, None)]
for name, csr, want in cases:
    if csr is None:
        continue
    for knobs in ({}, {"BK_BUDGET": "64", "BK_BUDGET0": "64"}, {"BK_GROUPS": "0"}):
        capi.reset_options()
        for k, v in knobs.items():
            capi.set_option(k, v)
        g = capi.DeviceGraph.from_csr(csr)
        got, st = g.bk_count(stats=True)
        print(name, knobs, got, "OK" if got == want else "MISMATCH want %d" % want, round(st["kernel_ms"], 1), "ms", flush=True)
        g.free()
