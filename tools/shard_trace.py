import sys
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import torch, ovr_amd as ovr
from test_full_size_gpu import _setup
world, tile = int(sys.argv[1]), int(sys.argv[2])
n, size = 1024, (1920, 1080)
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), 'float32')
ren = _setup(ovr, ovr.create_renderer('hip'), vol, n, size, 2, accumulate=True, shard=(0, world, tile, tile))
for _ in range(8):
    ren.render()
st = ren.stats()
print('march', st.march_ms, 'shade', st.shade_ms, 'comp', st.composite_ms)
