"""CPU: the committed fallback traffic table (profiles/hbm_traffic.json) belongs to the kernel sources in the tree.  bench.py measures the
beyond-L2 traffic live (rocprofv3 --pmc child passes) and uses the table only when that is impossible — and never across kernel hashes, so
a table left behind by an older build would silently turn `roofline.traffic` into None for the N = 2, 4, 8 runs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_table_matches_kernel_sources():
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
        doc = json.load(f)
    khash = bench.kernel_hash()
    assert bench.committed_traffic("0" * 16, "kronecker-26-16/auto", 4) is None  # never served across kernel builds
    # (the live PMC passes of bench.py do not need the table; its N>1 fallback on a box without an N=1 run does — a stale table FAILS here, it is not skipped)
    assert khash in doc["by_kernel_hash"], (f"profiles/hbm_traffic.json holds no entry for the kernel sources in the tree ({khash}): re-run "
                                             "`python bench.py --dump-traffic profiles/hbm_traffic.json` on the GPU and commit it")
    rec = doc["by_kernel_hash"][khash]["kronecker-26-16/auto"]
    assert set(rec) >= {"n1", "n2", "n4", "n8"}
    # shards of a strong-scaling run: each about 1/N of the whole pass
    for n in (2, 4, 8):
        assert 0.8 / n < rec[f"n{n}"]["bytes"] / rec["n1"]["bytes"] < 1.3 / n
    assert bench.committed_traffic(khash, "kronecker-26-16/auto", 4)["bytes"] == rec["n4"]["bytes"]
