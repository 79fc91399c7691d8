"""GPU: the N > 1 paths as far as a 1-GPU box can take them (SURVEY §8e — the OpenMP `reduction(+:total)` of
triangle_count/parallel/total.h:12 becomes one 8-byte all-reduce of per-rank partial counts).

* `bench.py --gpus 2` under `python -m torch.distributed.run --nproc-per-node 2`, started as a FRESH child process exactly as the driver
  starts it, both ranks sharing cuda:0 (GMSX_SHARE_GPU=1: gloo carries the one all-reduce, RCCL refuses two ranks on one device): the
  JSON line of an N = 2 run — sharded uploads, shard partials that add up to the reference golden, roofline / cpu_baseline records present.
* `gmsx_driver tc --gpus 2` on a box with ONE device: rank 1 has no GPU; the supervisor must take rank 0 down and return promptly with a
  non-zero status instead of leaving it in ncclCommInitRank forever."""
import glob
import json
import os
import socket
import subprocess
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_on_one_gpu_json_line(gpu, tmp_path):
    env = dict(os.environ, GMSX_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "20", "--steps", "3", "--warmup", "1", "--cache-dir",
           str(tmp_path / "cache")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                        # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "strong" and out["unit"] == "edges/s"
    assert out["config"]["triangles"] == 423625371                  # = the reference golden of scale 20 (bench.py asserts it too)
    assert "reference golden" in out["config"]["parity"]
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert "gloo" in out["config"]["collective"]                    # the share-one-GPU hook; on a real node: gmsx_comm_allreduce_u64 over RCCL
    assert isinstance(out.get("roofline"), dict) and out["roofline"]["bound"] in ("beyond-L2", "valu", "lds") and out["roofline"]["contract_bound"] == "hbm" and out["roofline"]["algorithmic_bytes"] > 0
    assert "cpu_baseline" in out                                     # None + a note at N > 1 without an N = 1 run on the box: never absent
    assert out["upload"]["sharded"] is True and out["upload"]["shard"] == [0, 2]


def test_bench_eight_ranks_on_one_gpu(gpu, tmp_path):
    """VERDICT r5 item 6 — first-contact readiness for an 8-GPU node, as far as one GPU goes: the N = 1 run of bench.py on the box (live PMC passes,
    shard traffic cached), then `bench.py --gpus 8` exactly as the driver launches it, eight ranks sharing cuda:0 (GMSX_SHARE_GPU=1, gloo carries the
    all-reduce).  One JSON line; eight sharded uploads whose partials add up to the reference golden; the ranks MAP the one cache file (no private CSR
    copies: RssAnon of the fattest rank stays below the CSR's size); the roofline of the N = 8 line comes from the traffic the N = 1 run measured for
    shard 0 of 8 on this kernel build.  UNMEASURED on real multi-GPU hardware: no 8-GPU node has been available to any round."""
    cache = str(tmp_path / "cache")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "22", "--steps", "2", "--warmup", "1", "--cache-dir", cache, "--side", "0",
                         "--cpu-seconds", "0", "--check-scale", "0", "--big", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert one["n_gpus"] == 1 and one["roofline"]["traffic"]
    env["GMSX_SHARE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--scale", "22", "--steps", "2", "--warmup", "1", "--cache-dir", cache]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["triangles"] == 2111140967   # the reference golden of scale 22
    assert out["upload"]["sharded"] is True and out["upload"]["shard"] == [0, 8]
    assert "gloo" in out["config"]["collective"]
    roof = out["roofline"]
    assert roof["traffic"] and "N=1 run on this box" in roof["traffic_source"], roof.get("traffic_source")
    assert 0.05 < roof["traffic"] / one["roofline"]["traffic"] < 0.25                                        # shard 0 of 8: about an eighth of the pass
    hm = out["host_memory"]
    # every rank holds the CSR as mapped FILE pages (one copy in the page cache for all eight), not as private memory
    assert hm["csr_mapped"] is True and hm["rss_file_bytes_max_over_ranks"] >= 0.9 * hm["csr_bytes"], hm
    assert hm["csr_load_rss_anon_delta_max_over_ranks"] < 0.1 * hm["csr_bytes"], hm
    assert hm["peak_rss_bytes_max_over_ranks"] > 0


def test_driver_two_ranks_one_device_fails_cleanly(gpu):
    exe = os.path.join(ROOT, "gms_amd", "lib", "gmsx_driver")
    before = set(glob.glob("/tmp/gmsx_driver_id_*"))
    t0 = time.time()
    r = subprocess.run([exe, "tc", "-g", "kronecker", "12", "-n", "1", "--gpus", "2"], capture_output=True, text=True, timeout=300)
    dt = time.time() - t0
    assert r.returncode != 0, r.stdout[-1000:]
    assert "a rank ended with status" in r.stderr, r.stderr[-1500:]
    assert dt < 120, dt                                              # not a hang: rank 0 was stopped, not left in a collective
    assert set(glob.glob("/tmp/gmsx_driver_id_*")) == before         # the id file is gone


def test_bench_rank_that_dies_before_the_id_broadcast_takes_the_job_down(gpu, tmp_path):
    """rank 1 of a two-rank `bench.py` dies right after the process group is up — before the communicator id is broadcast, before any
    collective.  Rank 0 must not be left waiting: the job ends non-zero within a minute, and nothing re-execs (every process that initialised the GPU
    either finishes or exits; the launcher reaps).  GMSX_COMM_TIMEOUT_S bounds both the library's communicator and the control plane's waits."""
    env = dict(os.environ, GMSX_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", GMSX_BENCH_TEST_DIE_RANK="1", GMSX_COMM_TIMEOUT_S="10")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "16", "--steps", "1", "--warmup", "1", "--cache-dir",
           str(tmp_path / "cache")]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    dt = time.time() - t0
    assert r.returncode != 0, r.stdout[-1000:]
    assert dt < 60, dt
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]   # no result line from a job that lost a rank
