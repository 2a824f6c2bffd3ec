"""Production hygiene on the GPU: no device-memory leak over create / upload / render / destroy cycles, setters racing a render
thread (the reference's GUI thread sets, its render thread commits: SURVEY.md 8b "Threading"), and frames that stay finite and
stable over a long accumulation."""
import threading

import numpy as np
import pytest

from helpers import make_case, oracle_scene, hip_setup, hip_frame, compare

pytestmark = pytest.mark.gpu


def test_no_device_memory_leak(ovr, oracle):
    import torch
    torch.cuda.init()
    case = make_case(ovr, oracle, n=64, tf="bumps", cam="oblique", size=(320, 200), shading=2)

    def cycle():
        ren = hip_setup(ovr, ovr.create_renderer("hip"), case, accumulate=True)
        ren.set_empty_space_skipping(True)
        ren.commit()
        ren.render()
        ren.set_fbsize((200, 120))      # resize: framebuffers, pool bookkeeping and schedule are rebuilt
        ren.commit()
        ren.render()
        ren.mapframe_rgba8()
        ren.close()

    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 25 create/destroy cycles"


def test_setters_from_another_thread(ovr, oracle, hip_renderer_factory):
    case = make_case(ovr, oracle, n=40, tf="bumps", cam="oblique", size=(128, 96), shading=2)
    ren = hip_setup(ovr, hip_renderer_factory(), case, accumulate=True)
    cams = [ovr.synth.make_camera(c, 40) for c in ("oblique", "front", "inside")]
    tfs = [ovr.synth.make_tfn(t, 128) for t in ("bumps", "dense", "sparse")]
    stop = threading.Event()
    errors = []

    def gui():
        i = 0
        try:
            while not stop.is_set():
                ren.set_camera(ovr.Camera(*cams[i % 3], 60.0))
                ren.set_transfer_function(*tfs[i % 3])
                ren.set_focus((0.5, 0.5), 0.2 + 0.01 * (i % 7), 0.1)
                i += 1
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=gui)
    t.start()
    try:
        for _ in range(150):
            ren.commit()
            ren.render()
    finally:
        stop.set()
        t.join()
    assert not errors, errors
    # quiesce: a known state must render exactly as on a fresh renderer
    case["cam"] = tuple(cams[1])
    case["colors"], case["alphas"], case["vr"] = tfs[0]
    ren.set_camera(ovr.Camera(*cams[1], 60.0))
    ren.set_transfer_function(*tfs[0])
    ren.commit()
    ren.render()
    ref, _, cnt = oracle_scene(oracle, case).render(frames=1, accumulate=True)
    compare(oracle, hip_frame(ovr, ren)[0], ref, name="after racing setters")
    assert ren.stats().samples == cnt.samples and ren.stats().frame_index == 1


def test_long_accumulation_is_stable(ovr, oracle, hip_renderer_factory):
    case = make_case(ovr, oracle, n=32, tf="dense", cam="oblique", size=(96, 64), shading=2, spp=2)
    ren = hip_setup(ovr, hip_renderer_factory(), case, accumulate=True)
    for _ in range(400):
        ren.render()
    a = hip_frame(ovr, ren)[0].copy()
    for _ in range(100):
        ren.render()
    b = hip_frame(ovr, ren)[0]
    assert ren.stats().frame_index == 500 and np.isfinite(b).all()
    assert np.abs(a - b).max() < 0.02      # 2 jittered samples per pixel per frame: the running mean has converged
    assert b[..., 3].max() <= 1.0 + 1e-6
