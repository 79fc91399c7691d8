"""CPU: compiles and runs tests/cpp/test_set_concept.cpp against include/gmsx_set_graph.hpp + libgmsx.so; when the
reference tree is present the reference's own algorithm templates are instantiated over gmsx::HipSetGraph too."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"


@pytest.mark.parametrize("with_ref", [False, True])
def test_set_concept(tmp_path, capi, with_ref):
    if with_ref and not os.path.isdir(os.path.join(REF, "gms")):
        pytest.skip("reference tree not present")
    exe = str(tmp_path / "t")
    cmd = ["g++", "-std=c++17", "-O1", "-fopenmp", "-w", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_set_concept.cpp"),
           "-L", os.path.join(ROOT, "gms_amd", "lib"), "-lgmsx", "-Wl,-rpath," + os.path.join(ROOT, "gms_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    if with_ref:
        cmd[1:1] = ["-DWITH_REFERENCE", "-DNOPAPIW", "-DBK_COUNT", "-I", REF]
        cmd += [os.path.join(ROOT, "oracle", "_ref", "roaring.o")]
    subprocess.run(cmd, check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "set concept ok" in out


def test_driver_builds_and_rejects_bad_usage(capi):
    exe = os.path.join(ROOT, "gms_amd", "lib", "gmsx_driver")
    assert os.path.exists(exe), "make -C gms_amd/csrc builds the driver"
    assert subprocess.run([exe]).returncode == 101                      # no graph given (cli/cli.h:131-133)
    assert subprocess.run([exe, "tc", "--bogus"]).returncode == 100     # unparsable flags (cli/cli.h:122-127)
    # --opt NAME=VALUE is gmsx_set_option: the library takes no tuning from the environment, and a name it does not know is refused, not ignored
    r = subprocess.run([exe, "tc", "-g", "kronecker", "6", "--opt", "NO_SUCH_OPTION=1"], capture_output=True, text=True)
    assert r.returncode == 100 and "not an option" in r.stderr
    r = subprocess.run([exe, "tc", "-g", "kronecker", "6", "--opt", "KC_REVERSE=0", "--opt", "TIMING=1"], capture_output=True, text=True)
    assert r.returncode != 100, r.stderr   # accepted: the run itself then needs a device (or fails loudly without one)


def _have_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="exercises the failure path of a GPU-less host")
def test_gpus_launcher_reaps_failed_ranks_instead_of_hanging(capi):
    """`--gpus N` forks its ranks and supervises them with waitpid(-1): a rank that ends non-zero (here: every rank, gmsx_init finds no HIP
    device) takes the others down instead of leaving them in a collective without a timeout.  The launcher returns the failing
    rank's status, promptly, removes its id file, and refuses to fork ranks under a profiler preload."""
    import glob
    import time
    exe = os.path.join(ROOT, "gms_amd", "lib", "gmsx_driver")
    before = set(glob.glob("/tmp/gmsx_driver_id_*"))
    t0 = time.time()
    r = subprocess.run([exe, "tc", "-g", "kronecker", "6", "--gpus", "3", "-n", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and time.time() - t0 < 60, (r.returncode, r.stderr)
    assert "a rank ended with status 3" in r.stderr  # (rank 0's own "no HIP device" line may or may not get out before it is stopped)
    assert set(glob.glob("/tmp/gmsx_driver_id_*")) == before
    r = subprocess.run([exe, "tc", "-g", "kronecker", "6", "--gpus", "2"], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, ROCP_TOOL_LIBRARIES="/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"))
    assert r.returncode == 6 and "profiler library is preloaded" in r.stderr
    # a user's own ROCPROF_* configuration variable is not a preload: the ranks are forked (and fail for lack of a device)
    r = subprocess.run([exe, "tc", "-g", "kronecker", "6", "--gpus", "2"], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, ROCPROF_OUTPUT_PATH="/tmp/x"))
    assert r.returncode == 3, (r.returncode, r.stderr)


def test_gpus_supervisor_passes_a_stop_signal_on(capi):
    """SIGTERM to the supervisor (a `timeout`, a scheduler cancel) stops the ranks too — they would otherwise stay behind, blocked in a
    collective without a timeout, holding their GPUs.  GMSX_DRIVER_TEST_HANG parks the ranks like a blocked collective would."""
    import signal
    import time
    exe = os.path.join(ROOT, "gms_amd", "lib", "gmsx_driver")
    p = subprocess.Popen([exe, "tc", "-g", "kronecker", "6", "--gpus", "2"], env=dict(os.environ, GMSX_DRIVER_TEST_HANG="1"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        time.sleep(1.0)
        kids = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
        assert len(kids) == 2, kids
        p.send_signal(signal.SIGTERM)
        _, err = p.communicate(timeout=20)
        assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err)
        assert "stopping the 2 rank(s)" in err
        time.sleep(0.2)
        for k in kids:
            assert not os.path.exists("/proc/%s" % k), "rank %s outlived the supervisor" % k
    finally:
        if p.poll() is None:
            p.kill()


@pytest.mark.gpu
def test_driver_runs_every_kernel_with_reference_output_lines(gpu):
    exe = os.path.join(ROOT, "gms_amd", "lib", "gmsx_driver")
    expect = {"tc": "triangles: 483489", "kclique": "total 4-cliques: 96513528", "bk": "The Number of maximal clique counted: 692903", "vertex": "@@@"}
    for kernel, needle in expect.items():
        out = subprocess.run([exe, kernel, "-g", "kronecker", "12", "--deg", "16", "-n", "2", "-v"], check=True, capture_output=True, text=True).stdout
        assert needle in out, out
        assert "Graph has 4096 nodes and 48386 undirected edges for degree: 11" in out
        assert "GraphExec buildTime:" in out and out.count("Trial Time:") == 2 and "Average Time:" in out
        marks = [l for l in out.splitlines() if l.startswith("@@@ ")]
        assert len(marks) == 2 and all(" PASS " in l for l in marks), out
