"""The C-ABI libraries load without a GPU and export every symbol the headers under include/ declare;
compute entry points fail loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+|zkcnn_[a-z0-9_]+)\s*\(", src)))


def test_hip_library_exports_every_declared_symbol(built):
    import zkcnn_amd
    lib = zkcnn_amd.hip_lib()
    names = _declared("zkcnn_hip.h")
    assert len(names) > 35
    for n in names:
        assert hasattr(lib, n), f"libzkcnn_hip.so lacks {n}"


def test_host_library_exports_every_declared_symbol(built):
    import zkcnn_amd
    lib = zkcnn_amd.host_lib()
    for n in _declared("zkcnn_api.h"):
        assert hasattr(lib, n), f"libzkcnn_host.so lacks {n}"


def test_oracle_exports_the_same_driver_api(oracle):
    for n in ("session_create", "session_prove", "session_destroy", "session_row"):
        assert hasattr(oracle.lib, "oracle_" + n)


def test_record_layouts_match_the_reference_structs():
    # reference src/circuit.h:15-33: uniGate {u32 g,u; u8 lu,sc} = 12 B, binGate {u32 g,u,v; u8 sc,l} = 16 B
    class Uni(ctypes.Structure):
        _fields_ = [("g", ctypes.c_uint32), ("u", ctypes.c_uint32), ("lu", ctypes.c_uint8), ("sc", ctypes.c_uint8), ("pad", ctypes.c_uint8 * 2)]

    class Bin(ctypes.Structure):
        _fields_ = [("g", ctypes.c_uint32), ("u", ctypes.c_uint32), ("v", ctypes.c_uint32), ("sc", ctypes.c_uint8), ("l", ctypes.c_uint8),
                    ("pad", ctypes.c_uint8 * 2)]
    assert ctypes.sizeof(Uni) == 12 and ctypes.sizeof(Bin) == 16


def test_product_path_fails_loudly_without_gpu(built):
    import torch
    import zkcnn_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        zkcnn_amd.HipContext(0)
    with pytest.raises(RuntimeError):
        zkcnn_amd.Session("custom:F4", (4, 4, 1), 1)
