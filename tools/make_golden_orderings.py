#!/usr/bin/env python3
"""Golden ORDERINGS from the compiled reference (oracle/_ref/libgms_ref.so) -> tests/golden/orderings.npz.  Build container only.

  adg_<key>    PpParallel::getDegeneracyOrderingApproxSGraph<averageDegree, rank format>(eps = 0.001) at `-t 1`
               (gms/algorithms/preprocessing/parallel/degeneracy_approx_set.h:14-86; one thread = sequential partition/sort,
               the only deterministic run of that code)
  tco_<key>    PpParallel::triangleCountOrdering (preprocessing/parallel/triangle_count.h:11-30), order format
  off_/adj_    the CSR of the reference loader for graphs the gmsx loader does not regenerate bit-identically by itself (files)
Ties are unspecified in both reference functions; tests/test_oracle.py checks the goldens against the oracle's (round, degree) /
(count) keys, tests/test_orderings_gpu.py checks the device against the oracle bit for bit.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.bindings import Reference  # noqa: E402

R = Reference()
out = {}
for kind, scale, deg in [("kronecker", 8, 16), ("kronecker", 10, 16), ("kronecker", 12, 16), ("kronecker", 14, 16), ("uniform", 12, 16),
                         ("kronecker", 12, 4)]:
    key = "%s_%d_%d" % (kind, scale, deg)
    g = R.generate(kind, scale, deg, True, threads=1)
    out["adg_" + key] = R.rank(g, 1)
    out["tco_" + key] = R.tc_ordering(g, 0)
    assert np.array_equal(np.sort(out["adg_" + key]), np.arange(R.num_nodes(g)))
    R.free(g)
tg = "/root/reference/testing/testGraphs"
for name in sorted(os.listdir(tg)):
    if name.endswith(".el"):
        g = R.load_file(os.path.join(tg, name), relabel=True)
        out["adg_file_" + name[:-3]] = R.rank(g, 1)
        out["tco_file_" + name[:-3]] = R.tc_ordering(g, 0)
        R.free(g)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "orderings.npz"), **out)
print("wrote", len(out), "arrays")
