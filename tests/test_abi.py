"""The C-ABI library loads here (no GPU) and exports every symbol include/ovr_hip.h declares; no compute is called."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ovr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ovr_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(ovr):
    decl = declared_symbols()
    assert len(decl) >= 30
    assert sorted(ovr._lib.SYMBOLS) == decl, "include/ovr_hip.h and the ctypes table list different entry points"


def test_library_exports_every_declared_symbol(ovr):
    lib = ovr._lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.ovr_hip_abi_version() == int(re.search(r"#define OVR_HIP_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "ovr_hip.h")).read()).group(1))


def test_stats_struct_layout_matches_header(ovr):
    # 5 x u64, 2 x f64, 2 x i32, 3 x f64, 3 x u64, 2 x i32, 3 x u64
    assert C.sizeof(ovr._lib.Stats) == 5 * 8 + 2 * 8 + 2 * 4 + 3 * 8 + 3 * 8 + 2 * 4 + 3 * 8


def test_no_cpu_fallback(ovr):
    """without an MI355X the product path must fail loudly, never fall back to the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device|HIP"):
        ovr.create_renderer("hip")
    with pytest.raises(RuntimeError, match="Could not find device_"):
        ovr.create_renderer("optix7")


def test_null_handles_are_rejected(ovr):
    lib = ovr._lib.load()
    assert lib.ovr_hip_commit(None) < 0
    assert b"null renderer" in lib.ovr_hip_last_error()
    assert lib.ovr_hip_render(None) < 0
    assert lib.ovr_hip_set_fbsize(None, 4, 4) < 0


def test_plugin_exports_the_reference_factory_symbol():
    """libdevice_hip.so (plugin/device_hip.cpp, built against the reference's headers) must export the C symbol the
    reference's factory looks up: ovr_create_renderer__hip (reference ovr/renderer.cpp:55-58, ObjectFactory.h:51)"""
    path = os.path.join(ROOT, "plugin", "libdevice_hip.so")
    if not os.path.exists(path):
        pytest.skip("plugin not built (needs the reference tree at build time)")
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    assert " T ovr_create_renderer__hip" in out
