"""Export post-processing: device time for a 47-channel probability volume resampled to the original grid and classified, next to
the CPU restatement of the reference path timed on a bounded sample (a few channels, extrapolated linearly to 47)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from multitalent_amd.inference.segmentation_export import resample_and_classify
from oracle.reference_ops import resample_probabilities

C, src, dst = 47, (320, 320, 320), (512, 512, 512)
dev = torch.device('cuda:0')
p = torch.rand((C,) + src, device=dev)
props = {'size_after_cropping': np.array(dst), 'original_size_of_raw_data': np.array(dst), 'crop_bbox': [[0, dst[i]] for i in range(3)],
         'original_spacing': np.array([1.0, 0.78, 0.78]), 'spacing_after_resampling': np.array([1.5, 1.5, 1.5])}
order = list(range(1, C + 1))
for sep, name in ((None, 'trilinear'), (True, 'separate z')):
    pr = dict(props)
    if sep:
        pr['original_spacing'] = np.array([5.0, 0.78, 0.78])
    resample_and_classify(p, pr, order); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3):
        out = resample_and_classify(p, pr, order)
    torch.cuda.synchronize()
    print('%s: %d x %s -> %s uint8 on the device: %.1f ms' % (name, C, src, dst, (time.time() - t) / 3 * 1e3))
nc = 2
x = p[:nc].cpu().numpy()
t = time.time()
resample_probabilities(x, dst, axis=None, do_separate_z=False)
dt = time.time() - t
print('CPU restatement (scipy zoom order 1, 1 core): %.1f s for %d channels -> %.0f s for %d channels (+ thresholds)' % (dt, nc, dt / nc * C, C))
