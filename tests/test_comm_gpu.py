"""GPU: the native RCCL path under the C-ABI (gmsx_comm_*) and the C++ driver's --gpus mode, on the one GPU a test box has:
a 1-rank communicator runs the real ncclCommInitRank / ncclAllReduce(count=1, ncclUint64) code path."""
import os
import subprocess

import pytest

from conftest import ROOT, host_graph

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_allreduce(gpu):
    uid = gpu.Comm.unique_id()
    assert len(uid) == gpu.COMM_ID_BYTES and any(uid)
    c = gpu.Comm.init(0, 1, uid)
    assert (c.rank, c.size) == (0, 1)
    for v in (0, 1, 49175273487, (1 << 64) - 1):
        assert c.allreduce_u64(v) == v
    g = gpu.DeviceGraph.from_csr(host_graph(gpu, "kronecker", 12))
    assert c.allreduce_u64(g.tc_partial(0, 1)) == 483489
    g.free()
    c.finalize()


def test_driver_gpus_1_runs_through_rccl(gpu):
    exe = os.path.join(ROOT, "gms_amd", "lib", "gmsx_driver")
    for kernel, needle in (("tc", "triangles: 483489"), ("kclique", "total 4-cliques: 96513528"), ("bk", "The Number of maximal clique counted: 692903")):
        r = subprocess.run([exe, kernel, "-g", "kronecker", "12", "-n", "1", "-v", "--gpus", "1"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert needle in r.stdout and "RCCL communicator: 1 rank(s)" in r.stdout and " PASS " in r.stdout, r.stdout
