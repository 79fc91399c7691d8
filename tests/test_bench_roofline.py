"""CPU: the arithmetic of bench.py's roofline records on the counter values of profiles/r05/s26_r5/summary.txt — the hub kernel of the
triangle pass: 357 GB beyond the L2, 27.6 G VALU instructions, 19.5 G LDS-array cycles (11.3 G of them bank conflicts), GRBM_GUI_ACTIVE 797 M summed over
the 8 XCDs, 48.5 ms.  `achieved` / `frac` are ALGORITHMIC bytes over time over the 8 TB/s peak (the contract); the traffic, VALU and LDS fractions sit beside
them and `bound` names the highest."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_compute_side_and_bound():
    import bench
    hub = {"bytes": 357.0e9, "GRBM_GUI_ACTIVE": 797333066.0, "SQ_INSTS_VALU": 27553127078.0, "SQ_LDS_IDX_ACTIVE": 19544233512.0,
           "SQ_LDS_BANK_CONFLICT": 11319277462.0, "SQ_WAVE_CYCLES": 201201077306.0, "SQ_WAIT_ANY": 96.0e9, "dur_ns:GRBM_GUI_ACTIVE": 48.5e6}
    cs = bench.compute_side(hub, 256)
    assert abs(cs["lds_frac"] - 0.766) < 0.005 and abs(cs["valu_frac"] - 0.81) < 0.01 and abs(cs["clock_GHz"] - 2.055) < 0.01
    assert cs["valu_frac_range"][0] < cs["valu_frac"] < cs["valu_frac_range"][1] and abs(cs["lds_bank_conflict_share"] - 0.579) < 0.005
    trec = dict(hub, kernels={"k_tc_items<false>": dict(hub, dispatches=1)}, fetch_multiplier=2.0)
    r = bench.make_roofline(300.0e9, 48.5e-3, trec, 256, 6200.0)
    assert abs(r["achieved"] - 300e9 / 48.5e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12 and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac_traffic"] - 357e9 / 48.5e-3 / 1e9 / 8000.0) < 1e-9 and r["traffic"] == 357.0e9
    assert r["bound"] == "beyond-L2" and r["contract_bound"] == "hbm"          # memory by traffic 0.92 > VALU 0.81 > LDS 0.77
    assert r["frac_traffic_of_measured_stream_ceiling"] > 1.0                   # above what HBM streams: served by the Infinity Cache in part
    assert abs(r["frac_of_measured_stream_ceiling"] - r["achieved"] / 6200.0) < 1e-12
    assert r["per_kernel"]["k_tc_items<false>"]["valu_frac"] == cs["valu_frac"]
    # an issue-bound kernel: few bytes, many instructions
    bk = dict(hub, bytes=50.0e9)
    r2 = bench.make_roofline(20.0e9, 48.5e-3, dict(bk, kernels={}), 256, None)
    assert r2["bound"] == "valu" and r2["frac_of_measured_stream_ceiling"] is None
    # no PMC at all: the contract fields are still there, traffic is null
    r3 = bench.make_roofline(20.0e9, 48.5e-3, None, 256, 6200.0)
    assert r3["traffic"] is None and r3["frac"] is not None and r3["bound"] == "beyond-L2" and "valu_frac" not in r3
