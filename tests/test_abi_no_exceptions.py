"""CPU: no exception crosses the C ABI.  Every exported function runs its body under gmsx::guard
(gms_amd/csrc/host/gmsx_internal.hpp): a failed allocation inside the C++ library comes back as a status code — the reference's
convention is exit codes, never unwinding through a caller (gapbs/reader.h:45,228) — instead of std::terminate taking the
host process (a cgo / JNI / ctypes caller cannot catch anything) down."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import ctypes, os, resource, sys
    sys.path.insert(0, %r)
    from gms_amd import capi
    L = capi.lib()
    L.gmsx_set_host_threads(2)
    h = ctypes.c_void_p()
    # warm-up: the OpenMP pool and the library's own state exist before the address space is capped
    assert L.gmsx_csr_generate(0, 10, 16, 1, 2, ctypes.byref(h)) == 0
    L.gmsx_csr_free(h)
    with open("/proc/self/statm") as f:
        vm_bytes = int(f.read().split()[0]) * os.sysconf("SC_PAGE_SIZE")
    resource.setrlimit(resource.RLIMIT_AS, (vm_bytes + (96 << 20), vm_bytes + (96 << 20)))
    h = ctypes.c_void_p()
    rc = L.gmsx_csr_generate(0, 23, 16, 1, 2, ctypes.byref(h))     # needs GBs: new[] / std::vector fail inside the loader
    print("generate", rc, flush=True)
    # a text file whose edge vectors outgrow the cap: std::vector growth throws inside the reader
    rc2 = L.gmsx_csr_load(%r.encode(), 1, 0, ctypes.byref(h))
    print("load", rc2, flush=True)
    print("alive", flush=True)
""")


def test_allocation_failure_is_a_status_code(tmp_path):
    el = tmp_path / "big.el"
    with open(el, "w") as f:  # 12 M edges = 96 MB of (u, v) int32 pairs once parsed, more while the vectors double
        for blk in range(120):
            f.write("".join("%d %d\n" % (i, i + 1) for i in range(blk * 100000, (blk + 1) * 100000)))
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, str(el))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])  # not SIGABRT from std::terminate
    out = dict(l.split()[:2] for l in r.stdout.splitlines() if l.startswith(("generate", "load")))
    assert int(out["generate"]) == -2, r.stdout       # GMSX_ERR_NOMEM
    assert int(out["load"]) in (-2, 0), r.stdout      # NOMEM under the cap (0 only if the box's allocator fitted it after all)
    assert "alive" in r.stdout
