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
    # 5 x u64, 2 x f64, 2 x i32, 3 x f64, 3 x u64, 2 x i32, 3 x u64, 3 x i32 + padding (ABI v6: skipping_kernels; v7: tuning; v8: replicas_building)
    assert C.sizeof(ovr._lib.Stats) == 5 * 8 + 2 * 8 + 2 * 4 + 3 * 8 + 3 * 8 + 2 * 4 + 3 * 8 + 4 * 4
    # the same fields, in the same order, as the header's struct
    import re
    hdr = open(os.path.join(ROOT, "include", "ovr_hip.h")).read()
    body = hdr[hdr.index("typedef struct ovr_hip_stats {"):hdr.index("} ovr_hip_stats;")]
    names = re.findall(r"^\s*(?:uint64_t|int32_t|double)\s+(\w+);", body, flags=re.M)
    assert names == [f[0] for f in ovr._lib.Stats._fields_]


def test_ctypes_structs_have_the_size_the_header_gives_them(ovr, tmp_path):
    """sizeof(ovr_hip_stats) / sizeof(ovr_hip_volume_info) as a C compiler sees include/ovr_hip.h == the ctypes structures' sizes, and the
    package refuses a library of another ABI version (ovr_hip_get_stats would write past a shorter ctypes buffer)"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include "ovr_hip.h"\nint main(void) { printf("%zu %zu %d\\n", sizeof(ovr_hip_stats), sizeof(ovr_hip_volume_info), OVR_HIP_ABI_VERSION); return 0; }\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "probe")])
    s_stats, s_info, abi = map(int, subprocess.check_output([str(tmp_path / "probe")], text=True).split())
    assert C.sizeof(ovr._lib.Stats) == s_stats
    assert C.sizeof(ovr._lib.VolumeInfo) == s_info
    assert ovr._lib.EXPECTED_ABI == abi == ovr._lib.load().ovr_hip_abi_version()
    # a library that reports another version is refused
    real = ovr._lib._lib
    try:
        ovr._lib._lib = None
        ovr._lib.EXPECTED_ABI = abi + 1
        with pytest.raises(RuntimeError, match="ABI version"):
            ovr._lib.load()
    finally:
        ovr._lib.EXPECTED_ABI = abi
        ovr._lib._lib = real


def test_addressing_mode_selection(ovr):
    """which addressing mode the kernels take (host arithmetic only): by the layout's size - and mode 3 (no LDS tables) whenever the per-axis
    tables would not fit in LDS next to the transfer function and the request queues, for EVERY base mode (a volume small in
    bytes with one very long axis used to be accepted and then fail at its first launch)"""
    lib = ovr._lib.load()

    def mode(dims, vt=400, layout=0, nc=1024, na=1024):
        return lib.ovr_hip_query_addressing_mode((C.c_int32 * 3)(*dims), vt, layout, nc, na)

    assert mode((256, 256, 256), 100) == 0              # C1: 16 MiB
    assert mode((1024, 1024, 1024)) == 1                # C3 general layout: 5.7 GB, < 2^32 stored voxels
    assert mode((1024, 1024, 1024), layout=3) == 2      # its quad replica: 2^32 floats
    assert mode((2048, 2048, 2048), 200) == 2           # C4: 11.4 G stored voxels
    assert mode((512, 512, 304), layout=3) == 0         # scene_lung's quad replica: 1.3 GB
    assert mode((40000, 8, 8)) == 3                     # 10 MB, but 160 KB of x table: computed offsets
    assert mode((8, 8, 30000), 100) == 3
    assert mode((20000, 8, 8)) == 0                     # 80 KB of tables still fit
    assert mode((20000, 8, 8), nc=4096, na=4096) == 3   # ... not next to an 80 KiB transfer function
    assert mode((64, 64, 64), 100, layout=1) < 0        # 8-bit volumes have no thin replicas ...
    assert mode((64, 64, 64), 100, layout=3) == 0       # ... but a quad replica, like 16-bit and float volumes
    assert mode((2048, 2048, 2048), 200, layout=3) == 2 # C4's would be 64 GiB
    assert mode((64, 64, 64), 201, layout=3) < 0        # signed types: general layout only
    # (round 4) the taps add the in-plane offsets X + Y in 32 bits: a layout whose z layer of macro blocks holds more than 2^32
    # elements would alias cells silently - a slab-shaped 8192 x 8192 x 64 u8 volume's quad replica (17 GB, inside the 40 % rule) did; such a
    # layout is not built, and the query says so
    assert mode((8192, 8192, 64), 100, layout=0) >= 0
    assert mode((8192, 8192, 64), 100, layout=3) < 0 and b"2^32" in lib.ovr_hip_last_error()
    assert mode((4095, 8190, 64), 100, layout=3) >= 0      # 128 x 256 macro blocks of 2^17 elements: exactly 2^32
    assert mode((20000, 20000, 8), 100, layout=0) < 0   # ... and the general layout beyond ~16 k x 16 k is refused by ovr_hip_set_volume


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


def test_march_kernels_keep_their_register_budget(tmp_path):
    """The march is fragile under the compiler's register allocation (compiler note): the pooled march must stay within
    the 3-waves-per-SIMD budget (<= 168 VGPRs - at 169 the skipping variant lost 15 %) and no march / shade kernel may spill to
    scratch.  Read from the code objects embedded in the built library (no GPU, no recompilation)."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "open-volume-renderer_amd", "libovr_hip.so")
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(lib) and os.path.exists(os.path.join(llvm, "llvm-objdump"))):
        pytest.skip("library or llvm tools not present")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp_path, capture_output=True, check=True)
    kernels = {}
    for f in sorted(os.listdir(tmp_path)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", f], cwd=tmp_path, capture_output=True, text=True, check=True).stdout
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split(":", 1)[1].strip()
                kernels[name] = {}
            elif name and (line.startswith(".vgpr_count:") or line.startswith(".private_segment_fixed_size:")):
                k, v = line.split(":")
                kernels[name][k.strip()] = int(v)
    march = {n: k for n, k in kernels.items() if "raymarch_kernel" in n or "shade_pool_kernel" in n}
    assert len(march) > 200, len(march)
    for n, k in march.items():
        # no scratch - but for the pinned skipping variants, where 3 spilled dwords in a cold path were measured 27 % faster than
        # giving up the third wave (transposed thin layout, C3 side view: march 0.345 vs 0.439 ms)
        pinned = re.search(r"raymarch_kernelILi\d+ELi\d+ELi[01]ELb1ELb1E", n) is not None
        # (round 4) the computed-offset mode (AM 3: a dimension of tens of thousands of voxels, no LDS tables) clamps its indices explicitly
        # since the layouts carry the clamp-to-edge copies; its in-place skipping march keeps 5 dwords in scratch - a cold path of a cold mode
        am3 = re.search(r"raymarch_kernelILi\d+ELi\d+ELi3ELb0ELb1E", n) is not None
        assert k[".private_segment_fixed_size"] <= (16 if pinned else 32 if am3 else 0), (n, k)
        # raymarch_kernel<VT, SHADE, AM, POOLED = true, SKIP, LDSB>: ...ILi<vt>ELi<shade>ELi<am>ELb1E...; the 64-bit addressing modes
        # (AM 2, 3) may take more (measured: no difference on C4, the only configuration that uses them)
        # ...ELb<pooled>ELb<skip>ELb<ldsb>ELb<deep>E: the deep variant (image shards) runs 6 instructions per round at 2 waves on purpose
        if "raymarch_kernel" in n and re.search(r"raymarch_kernelILi\d+ELi\d+ELi[01]ELb1ELb[01]ELb0ELb0E", n):
            assert k[".vgpr_count"] <= 168, (n, k)
