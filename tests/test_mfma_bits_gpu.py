"""GPU tests of the two assumptions the matrix-core k = 4 count (gms_amd/csrc/hip/kc4_mfma.hpp) rests on, through the stand-alone probes under tools/probes:
bits expanded to i8 / fp4 (e2m1) elements multiply exactly on v_mfma_i32_32x32x32_i8 / v_mfma_scale_f32_32x32x64_f8f6f4 (scales 1.0), and the count kernel
itself — ragged widths in one pool, garbage in the rows and words the BUILD never writes, 2 x 2 and 4 x 4 tiles, teams of 1 … 32 workgroups — equals the
host's AND + popcount sum  Σ_{i>j, L_ij} |row_i ∩ row_j|  (k_clique_count_set_based.h:5-17 on a bit matrix).  The probes are compiled on the box with hipcc."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _build(tmp_path, name):
    exe = str(tmp_path / name)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-I" + os.path.join(ROOT, "gms_amd", "csrc", "hip"), os.path.join(ROOT, "tools", "probes", name + ".hip"), "-o", exe],
                   check=True, capture_output=True, timeout=600)
    return exe


def test_bit_products_are_exact_on_the_matrix_cores(gpu, tmp_path):
    out = subprocess.run([_build(tmp_path, "mfma_bits")], check=True, capture_output=True, text=True, timeout=300).stdout
    checks = [l for l in out.splitlines() if "wrong of 1024" in l]
    assert len(checks) == 3, out            # i8, fp4 with scale bytes 0x7f, fp4 with scale operand 0
    assert all(": 0 wrong of 1024" in l for l in checks), out


def test_count_kernel_equals_popcount_sum(gpu, tmp_path):
    exe = _build(tmp_path, "kc4_mfma_probe")
    # (d, density, matrices): below one tile, an odd number of column words, the K tails of 2 / 4 / 6 words, several blocks with teams
    out = subprocess.run([exe, "1", "0.5", "3", "33", "0.9", "5", "70", "0.5", "7", "200", "0.3", "40", "330", "0.6", "24", "600", "0.5", "64"],
                         check=True, capture_output=True, text=True, timeout=600).stdout
    sums = [l for l in out.splitlines() if " sum " in l and "noepi" not in l and "noexp" not in l]   # (the two debug variants are wrong by design)
    assert len(sums) == 6 * 12, out
    assert all(" OK " in l and "WRONG" not in l for l in sums), out
