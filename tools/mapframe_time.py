"""PCIe-inclusive frame hand-over: time of mapframe(HOST) (RGBA32F + gradient layer), RGBA32F only, and mapframe_rgba8(HOST)"""
import sys, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path.insert(0, _R)
import torch, ovr_amd as ovr
n, size = 256, (1920, 1080)
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), 'float32')
colors, alphas, vr = ovr.synth.make_tfn('sparse', 1024)
ren = ovr.create_renderer('hip')
ren.set_fbsize(size); ren.set_shading(2); ren.set_transfer_function(colors, alphas, vr)
ren.init(ovr.Scene(volume=vol, transfer_function=None), ovr.Camera(*ovr.synth.make_camera('oblique', n)))
ren.commit(); ren.render()
fb = ovr.FrameBufferData()
def t(f, k=20):
    f(); t0 = time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter() - t0) / k * 1e3
print('mapframe(HOST) rgba32f+grad: %.3f ms' % t(lambda: ren.mapframe(fb)))
print('mapframe_rgba8(HOST): %.3f ms' % t(lambda: ren.mapframe_rgba8()))
print('mapframe_rgba8(DEVICE): %.3f ms' % t(lambda: ren.mapframe_rgba8(device=True)))
