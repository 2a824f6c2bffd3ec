"""Per-rank render time of an image-sharded frame, all 'ranks' run one after the other on ONE card: how well does the
tile -> rank assignment balance the work?  usage: python tools/shard_balance.py [n] [tile ...]
OVR_SHARD_CONFIG=c4 | c5 rehearses BASELINE's 8-GPU configurations (2048^3 u16 at 1920x1080; 1024^3 f32 at 3840x2160 with the
blue-noise pixel jitter) instead of C3."""
import os
import sys
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import numpy as np, torch, ovr_amd as ovr
from test_full_size_gpu import _setup

_cfg = os.environ.get('OVR_SHARD_CONFIG', 'c3')
n = int(sys.argv[1]) if len(sys.argv) > 1 else (2048 if _cfg == 'c4' else 1024)
tiles = [int(a) for a in sys.argv[2:]] or [64, 32, 16]
size = (3840, 2160) if _cfg == 'c5' else (1920, 1080)
_dt = 'uint16' if _cfg == 'c4' else 'float32'
_np_dt = np.uint16 if _cfg == 'c4' else np.float32
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), _dt)
_setup0 = _setup


def _setup(ovr_, ren, vol_, n_, size_, shading, **kw):
    if _cfg == 'c5':  # blue-noise pixel jitter, one sample per pixel and frame (bench.py's c5)
        ren.set_noise_tile(ovr.synth.make_noise_tile(64))
        ren.set_pixel_jitter(ovr.JITTER_BLUE_NOISE)
    return _setup0(ovr_, ren, vol_, n_, size_, shading, dtype=_np_dt, **kw)


def kernel_ms(ren, frames=6):
    best = 1e9
    for _ in range(frames):
        ren.render()
        best = min(best, ren.stats().kernel_ms)
    return best


full = _setup(ovr, ovr.create_renderer('hip'), vol, n, size, 2, accumulate=True)
t_full = kernel_ms(full)
print(f'unsharded: {t_full:.3f} ms')
full.close()
for world in (() if os.environ.get('OVR_SHARD_PHASES_ONLY') == '1' else (2, 4, 8)):
    for tile in tiles:
        ms, work = [], []
        ren = ovr.create_renderer('hip')
        for rank in range(world):
            _setup(ovr, ren, vol, n, size, 2, accumulate=True, shard=(rank, world, tile, tile)) if rank == 0 else (ren.set_image_shard(rank, world, tile, tile), ren.commit())
            ms.append(kernel_ms(ren))
            st = ren.stats()
            work.append(st.samples + st.shadow_samples)
        ren.close()
        ms, work = np.array(ms), np.array(work, float)
        print(f'world {world} tile {tile:3d}: ms max {ms.max():.3f} mean {ms.mean():.3f} sum {ms.sum():.3f}  eff(render only) {t_full / (world * ms.max()):.3f}'
              f'  work max/mean {work.max() / work.mean():.3f}  ms: ' + ' '.join(f'{m:.2f}' for m in ms))

print('phases (march / shade / composite+reduce / total kernel ms), tile 16:')
for world in (1, 2, 4, 8):
    ren = ovr.create_renderer('hip')
    _setup(ovr, ren, vol, n, size, 2, accumulate=True, shard=(0, world, 16, 16))
    rows = []
    for _ in range(6):
        ren.render()
        st = ren.stats()
        rows.append((st.kernel_ms, st.march_ms, st.shade_ms, st.composite_ms, st.render_ms))
    k, m, s, c, r = min(rows)
    print(f'world {world}: march {m:.3f} shade {s:.3f} composite {c:.3f} kernel {k:.3f} blocking render() {r:.3f}  samples {st.samples} shadow {st.shadow_samples} chunks {st.pool_chunks}')
    ren.close()
