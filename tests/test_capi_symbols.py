"""CPU: the C-ABI library loads and exports every function include/gmsx.h declares; without a GPU every device
entry point fails loudly instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gmsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gmsx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree(capi):
    names = declared_functions()
    assert len(names) >= 30
    L = ctypes.CDLL(capi.LIB_PATH)
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/gmsx.h but not exported by libgmsx.so"
    assert sorted(capi.SYMBOLS) == names
    assert capi.lib().gmsx_version() == 310
    assert capi.lib().gmsx_strerror(-6).decode().startswith("no HIP device")


def test_no_cpu_fallback_in_product():
    """The product package must not reach for the oracle: no file under gms_amd/ mentions it."""
    for base, _, files in os.walk(os.path.join(ROOT, "gms_amd")):
        if os.path.basename(base) in ("lib", "obj", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "gms_oracle" not in text and "libgms_ref" not in text and "oracle.bindings" not in text, f


def _have_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="only meaningful on a GPU-less host")
def test_device_calls_fail_loudly_without_gpu(capi):
    with pytest.raises(capi.GmsxError) as ei:
        capi.init(0)
    assert ei.value.status == capi.ERR_NO_DEVICE
    with pytest.raises(capi.GmsxError) as ei:
        capi.DeviceGraph.upload(np.array([0, 1, 2], dtype=np.int64), np.array([1, 0], dtype=np.int32))
    assert ei.value.status == capi.ERR_NO_DEVICE


@pytest.mark.skipif(_have_gpu(), reason="only meaningful on a GPU-less host")
def test_new_entry_points_fail_loudly_without_gpu(capi):
    """The RCCL communicator and the ordering kernels have no host path either."""
    with pytest.raises(capi.GmsxError) as ei:  # (creating an id is host-side bootstrap and may succeed; binding a rank needs the device)
        capi.Comm.init(0, 1, bytes(capi.COMM_ID_BYTES))
    assert ei.value.status in (capi.ERR_COMM, capi.ERR_NO_DEVICE)
