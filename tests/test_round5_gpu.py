"""Round 5: the host side of the C ABI - where an upload spends its time (ABI v10), a device group driven by one host thread per member, the group's refusal to
render after a member's failure, a capturing stream refused instead of corrupted."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import hip_frame, hip_setup, make_case, oracle_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_upload_times_are_split(ovr, oracle, hip_renderer_factory):
    """ovr_hip_get_upload_times: allocation / copies into the device / kernels of the last ovr_hip_set_volume.  A host array is copied
    (through the staging buffer), a device array is not; the parts do not exceed the whole"""
    import torch
    case = make_case(ovr, oracle, n=96, size=(64, 48))
    ren = hip_renderer_factory()
    hip_setup(ovr, ren, case)
    t = ren.upload_times()
    assert t["total_ms"] > 0 and t["kernels_ms"] > 0 and t["copy_ms"] > 0 and t["alloc_ms"] >= 0          # numpy volume: host memory
    assert t["alloc_ms"] + t["copy_ms"] + t["kernels_ms"] <= t["total_ms"] + 0.05
    vol = torch.as_tensor(case["vol"]).cuda()
    ren2 = hip_renderer_factory()
    hip_setup(ovr, ren2, dict(case, vol=vol))
    t2 = ren2.upload_times()
    assert t2["copy_ms"] == 0.0 and t2["kernels_ms"] > 0 and t2["total_ms"] >= t2["kernels_ms"]
    ren.render(); ren2.render()
    assert np.array_equal(hip_frame(ovr, ren)[0], hip_frame(ovr, ren2)[0])
    assert ren.group_host_times() == (0.0, 0.0, 0.0, 0.0)                                                # not a group


def test_a_capturing_stream_is_refused_not_corrupted(ovr, oracle, hip_renderer_factory):
    """a frame records timed events and hands counters to the host - it is not a unit for hipGraph capture.  On a caller's stream that is
    capturing, render_async fails with OVR_HIP_ESTATE; the capture survives, and so does the renderer"""
    import torch
    case = make_case(ovr, oracle, n=24, size=(48, 32))
    ren = hip_renderer_factory()
    hip_setup(ovr, ren, case)
    ren.render()
    ref = hip_frame(ovr, ren)[0]
    s = torch.cuda.Stream()
    ren.set_stream(s.cuda_stream)
    ren.render()
    assert np.array_equal(hip_frame(ovr, ren)[0], ref)
    x = torch.zeros(16, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        x += 1.0
        with pytest.raises(RuntimeError, match="capturing"):
            ren.render_async()
    g.replay()
    torch.cuda.synchronize()
    assert float(x[0]) == 1.0                       # the capture was left intact (replayed once)
    ren.render()                                    # ... and so was the renderer
    assert np.array_equal(hip_frame(ovr, ren)[0], ref)
    ren.set_stream(None)


@pytest.mark.parametrize("members", [2, 8])
def test_group_members_are_driven_by_their_own_threads(ovr, oracle, members):
    """one host thread per follower.  Frames and counters equal the one-device renderer's through camera moves (every member re-classifies its
    blocks and waits for its two words on its own thread), accumulation and swaps; the leader thread's time to launch and ship a frame is reported"""
    case = make_case(ovr, oracle, n=40, size=(160, 96), tf="bumps", cam="oblique")
    one = ovr.create_renderer("hip")
    grp = ovr.create_renderer("hip", 0, devices=[0] * members)
    for r in (one, grp):
        hip_setup(ovr, r, case, accumulate=True)
    eye, at, up = case["cam"]
    host = []
    for i in range(12):
        e = tuple(np.array(eye) + np.array([0.7 * i, -0.3 * i, 0.0]))
        for r in (one, grp):
            if i % 3 == 0:
                r.set_camera(e, at, up)
                r.commit()
            r.render()
            if i % 4 == 3:
                fb = ovr.FrameBufferData(); r.mapframe(fb); r.swap()
        a, b = one.stats(), grp.stats()
        assert (a.rays, a.samples, a.shaded_samples, a.shadow_samples, a.frame_index) == (b.rays, b.samples, b.shaded_samples, b.shadow_samples, b.frame_index), i
        assert all(np.array_equal(x, y) for x, y in zip(hip_frame(ovr, one), hip_frame(ovr, grp))), i
        host.append(grp.group_host_times())
    h = np.array(host)
    assert (h[:, 0] > 0).all() and (h[:, 2] > 0).all()
    assert np.median(h[:, 0] + h[:, 1]) < 2000.0     # microseconds (one card shared by all members; round 4: 225 us for the launches alone at 8 members)
    n, kind, _ = grp.group_info()
    assert n == members and kind == 1                # a device listed twice: peer copies
    one.close(); grp.close()


def test_a_group_refuses_to_render_in_mixed_state():
    """a commit that fails on ONE member leaves the members on different states - the group must not render frames whose tiles come from two
    configurations.  OVR_HIP_TEST_FAIL_MEMBER=k fails member k's commit of a 4242-pixel-wide framebuffer (a stand-in for an allocation failure)"""
    code = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r + "/tests", %r + "/oracle"]
import ovr_amd as ovr, oracle as O
from helpers import make_case, hip_setup, hip_frame
case = make_case(ovr, O, n=24, size=(64, 48))
one = ovr.create_renderer("hip"); grp = ovr.create_renderer("hip", 0, devices=[0, 0, 0])
for r in (one, grp): hip_setup(ovr, r, case)
one.render(); grp.render()
assert np.array_equal(hip_frame(ovr, one)[0], hip_frame(ovr, grp)[0])
grp.set_fbsize((4242, 16))
try:
    grp.commit(); print("NO ERROR")
except RuntimeError as e:
    print("commit:", e)
try:
    grp.render(); print("RENDERED")
except RuntimeError as e:
    print("render:", e)
grp.set_fbsize((64, 48)); grp.commit(); grp.render()          # a commit that succeeds everywhere and leaves the members in agreement heals the group
assert np.array_equal(hip_frame(ovr, one)[0], hip_frame(ovr, grp)[0])
print("HEALED")
''' % (ROOT, ROOT, ROOT)
    for k in ("0", "2"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OVR_HIP_TEST_FAIL_MEMBER=k, OVR_HIP_QUIET="1"), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        assert f"commit failed on member {k}" in out.stdout and "render: " in out.stdout and "not renderable" in out.stdout and "HEALED" in out.stdout, out.stdout
        assert "NO ERROR" not in out.stdout and "RENDERED" not in out.stdout


def test_the_default_bench_command_prints_one_compact_line():
    """VERDICT r4 #1, on the GPU: `python bench.py --steps 2 --warmup 1` (every leg of the default command) - exactly one `{` line on stdout, below 4 KB,
    with the dominant kernel's roofline and the CPU baseline; the full record is in the detail file"""
    from helpers import run_bench
    c, d, out = run_bench(["--steps", "2", "--warmup", "1"], timeout=1500)
    assert c["steps"] == 2 and c["warmup"] == 1 and c["n_gpus"] == 1 and c["value"] > 0 and c["config"]["name"] == "c3"
    r = c["roofline"]
    assert r["bound"] in ("hbm", "ta", "valu", "l1") and r["kernel"].startswith("raymarch_kernel") and r["kernel_ms"] > 0 and r["algorithmic_bytes_per_launch"] > 1e9
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["cores"] >= 1
    assert set(c["extra"]) == {"c4_one_gpu", "c5_one_gpu", "c3_shard_of_8", "c3_device_group_rehearsal"} and all("ms_per_step" in v for v in c["extra"].values()), c["extra"]
    assert c["variants"]["sampling_rate_4"]["ms_per_step"] > 5 * c["ms_per_step"]        # the scene files' own rate: what renderapp users get
    assert set(r["upload_ms"]) == {"total_ms", "alloc_ms", "copy_ms", "kernels_ms"}
    assert "views" in d["roofline"] and "kernels" in d["roofline"] and d["extra"]["c4_one_gpu"]["roofline"]["compulsory_floor_bytes"] > 2e10
    assert d["extra"]["c3_device_group_rehearsal"]["device_group"]["host_us_per_frame"] > 0


def test_group_on_distinct_devices_when_the_box_has_them(ovr, oracle):
    """every group test so far lists device 0 several times (one-GPU boxes) - the peer-copy branch.  On a box with >= 2 GPUs this one runs the
    group on DISTINCT devices, RCCL (one ncclGroupStart / End per frame) and peer copies: frames and counters must equal the one-device renderer's."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU visible: the distinct-device branches (RCCL send / recv, hipMemcpyPeerAsync) wait for a multi-GPU node")
    case = make_case(ovr, oracle, n=48, size=(256, 160), tf="bumps", cam="oblique")
    one = ovr.create_renderer("hip")
    hip_setup(ovr, one, case, accumulate=True)
    for _ in range(3):
        one.render()
    ref, st = hip_frame(ovr, one), one.stats()
    one.close()
    for gather in ("rccl", "copy"):
        os.environ["OVR_HIP_GATHER"] = gather
        try:
            grp = ovr.create_renderer("hip", 0, devices=list(range(min(n, 8))))
        finally:
            del os.environ["OVR_HIP_GATHER"]
        hip_setup(ovr, grp, case, accumulate=True)
        for _ in range(3):
            grp.render()
        got, sg = hip_frame(ovr, grp), grp.stats()
        assert grp.group_info()[1] == (2 if gather == "rccl" else 1)
        assert all(np.array_equal(a, b) for a, b in zip(ref, got)), gather
        assert (st.rays, st.samples, st.shaded_samples, st.shadow_samples) == (sg.rays, sg.samples, sg.shaded_samples, sg.shadow_samples), gather
        grp.close()


def test_group_workers_that_sleep_between_commands():
    """OVR_HIP_WORKER_SPIN_US=0: a member's host thread goes to sleep on its condition variable at once instead of spinning for its next command - the wake-up
    path a render loop rarely takes.  The state-machine fuzzer on a group of four: every frame the oracle's whole frame."""
    e = dict(os.environ, OVR_HIP_WORKER_SPIN_US="0", OVR_FUZZ_GROUP="4", OVR_HIP_QUIET="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_states.py"), "8", "77", "10"], capture_output=True, text=True, timeout=900, env=e)
    assert out.returncode == 0 and "8 episodes, 0 failed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_the_shade_order_never_enters_a_frame(ovr, oracle, hip_renderer_factory, monkeypatch):
    """Round 5, second half: the pooled pipeline's shade kernel takes its runs sorted by LIGHT BEAM (the march leaves a key per run and their histogram,
    shade_order_kernel sorts, the workgroups of an XCD start on their own list - PoolDesc in ovr_hip_kernels.h) instead of in creation order.  The order in
    which requests are shaded is no input of any result: frames, gradient layer and every counter equal the creation-order renderer's
    (OVR_HIP_SHADE_ORDER=0, read when a renderer is created) bit for bit - three beam widths (32 x 32, 64 x 64, 128 x 128 beams), several samples per pixel,
    accumulation, empty-space skipping, a camera move, an anisotropic grid, a 16-bit volume, sparse sampling; and the oracle's frame within the bar"""
    from helpers import compare
    cases = [dict(n=64, size=(160, 120), cam="oblique"), dict(n=48, size=(96, 64), cam="front", spp=2), dict(n=40, dims=(56, 40, 24), size=(96, 80), cam="oblique", spacing=(1.0, 0.5, 2.0), origin=(3.0, -2.0, 1.0)),
             dict(n=48, size=(96, 64), cam="oblique", dtype=np.uint16), dict(n=32, size=(64, 48), cam="oblique", tf="dense")]
    for kw in cases:
        case = make_case(ovr, oracle, **kw)
        frames = {}
        for tag, env in (("creation", {"OVR_HIP_SHADE_ORDER": "0"}), ("beam32", {}), ("beam4", {"OVR_HIP_SHADE_BEAM": "4"}), ("beam1", {"OVR_HIP_SHADE_BEAM": "1"})):
            for k in ("OVR_HIP_SHADE_ORDER", "OVR_HIP_SHADE_BEAM"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            ren = hip_renderer_factory()
            hip_setup(ovr, ren, case, accumulate=True, pipeline=2)
            got = []
            for step in range(3):
                ren.render()
                st = ren.stats()
                assert st.pipeline == 2
                got.append(hip_frame(ovr, ren) + (st.samples, st.shaded_samples, st.shadow_samples))
            ren.set_empty_space_skipping(True)
            ren.commit(); ren.render()
            got.append(hip_frame(ovr, ren) + (ren.stats().shaded_samples, ren.stats().skipped_shadow_samples))
            eye, at, up = case["cam"]
            ren.set_camera(ovr.Camera(tuple(np.array(eye) * 0.9 + 1.5), at, up, case["fovy"]))
            ren.commit(); ren.render()
            got.append(hip_frame(ovr, ren) + (ren.stats().shadow_samples,))
            frames[tag] = got
        ref = frames["creation"]
        for tag, got in frames.items():
            for a, b in zip(ref, got):
                assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), (kw, tag)
                assert a[2:] == b[2:], (kw, tag, a[2:], b[2:])
        if case["spp"] == 1:
            sc = oracle_scene(oracle, case)
            rgba, _, _ = sc.render()
            compare(oracle, frames["beam32"][0][0], rgba, name=f"shade order {kw}")


def test_row_loads_of_the_16_bit_layouts_are_bit_identical(ovr, oracle, hip_renderer_factory, monkeypatch):
    """Round 5: the texture addresser merges the lanes of a quad only for loads of 8 bytes or more (tools/ubench_align.hip), so 16-bit layouts beyond the caches'
    reach read the ALIGNED 8 bytes around a voxel pair and shift the pair out (addressing mode 4 / RowLoads in ovr_hip_device.h; by size - forced here on small
    volumes with OVR_HIP_ROW_LOADS).  The same voxels: the same frame, bit for bit - general layout and the thin replicas (front / side view), both pipelines, with
    empty-space skipping, u16 / i16 / u8 / i8 (the 8-bit layouts' pairs are 2-byte loads, which never merge); and under a more general addressing mode (element offsets, 64-bit z table)"""
    seen_layouts = {}
    for dtype in (np.uint16, np.int16, np.uint8, np.int8):
        for cam in ("oblique", "front", "side"):
            if np.dtype(dtype).itemsize == 1 and cam == "side":
                continue
            case = make_case(ovr, oracle, n=40, dims=(44, 40, 36), size=(96, 72), cam=cam, dtype=dtype)
            frames = {}
            for tag, env in (("dword", {"OVR_HIP_ROW_LOADS": "0"}), ("rows", {"OVR_HIP_ROW_LOADS": "1"}), ("rows_am1", {"OVR_HIP_ROW_LOADS": "1", "OVR_HIP_ADDRESSING": "1"}),
                             ("rows_am2", {"OVR_HIP_ROW_LOADS": "1", "OVR_HIP_ADDRESSING": "2"})):
                for k in ("OVR_HIP_ROW_LOADS", "OVR_HIP_ADDRESSING"):
                    monkeypatch.delenv(k, raising=False)
                if tag.startswith("rows_am") and (dtype is not np.uint16 or cam == "side"):
                    continue
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                got = []
                for pipeline in (1, 2):
                    ren = hip_renderer_factory()
                    ren.set_volume_layouts(2)            # every replica resident before the first frame: the layout rule picks the thin ones for the axis views
                    hip_setup(ovr, ren, case, pipeline=pipeline)
                    ren.render()
                    got.append(hip_frame(ovr, ren) + (ren.stats().samples, ren.stats().shadow_samples, ren.stats().layout))
                    ren.set_empty_space_skipping(True)
                    ren.commit(); ren.render()
                    got.append(hip_frame(ovr, ren) + (ren.stats().shaded_samples,))
                frames[tag] = got
            seen_layouts.setdefault(np.dtype(dtype).name, set()).update(g[4] for g in frames["rows"][0::2])
            for tag, got in frames.items():
                for a, b in zip(frames["dword"], got):
                    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), (dtype, cam, tag)
                    assert a[2:4] == b[2:4], (dtype, cam, tag)
    assert 0 in seen_layouts["uint16"] and seen_layouts["uint16"] & {1, 2}, seen_layouts   # the general layout and a thin replica (an axis view) were both read
