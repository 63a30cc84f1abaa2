"""Device augmentation throughput at the benchmark patch: loader-sized batch (B=2, basic_generator_patch_size) -> 48x192x192."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from multitalent_amd.training.data_augmentation.spatial import SpatialTransformDevice, MirrorTransformDevice, get_patch_size
from multitalent_amd.training.data_augmentation.color import default_3d_augmentation_params, GammaDevice
dev = torch.device('cuda:0')
p = default_3d_augmentation_params()
ps = (48, 192, 192)
bps = tuple(int(i) for i in get_patch_size(ps, p['rotation_x'], p['rotation_y'], p['rotation_z'], (0.85, 1.25)))
x = torch.randn((2, 1) + bps, device=dev); s = torch.randint(0, 5, (2, 1) + bps, device=dev).float()
st = SpatialTransformDevice(ps, angle_x=p['rotation_x'], angle_y=p['rotation_y'], angle_z=p['rotation_z'], scale=p['scale_range'],
                            p_rot_per_sample=1.0, p_scale_per_sample=1.0, border_cval_seg=-1)     # every sample resampled: worst case
np.random.seed(0)
st(x, s); torch.cuda.synchronize()
t = time.time(); n = 10
for _ in range(n):
    d, g = st(x, s)
torch.cuda.synchronize()
dt = (time.time() - t) / n
print('loader patch %s -> %s, B=2, order 3 data + per-label order 1 seg, every sample rotated+scaled: %.2f ms per batch (%.0f patches/s)' % (bps, ps, dt * 1e3, 2 / dt))
gm = GammaDevice(p_per_sample=1.0); mr = MirrorTransformDevice()
t = time.time()
for _ in range(n):
    d2 = gm(d.clone()); mr(d2, g.clone())
torch.cuda.synchronize()
print('gamma (retain stats) + mirror on every sample: %.2f ms per batch' % ((time.time() - t) / n * 1e3))
