"""Device pre-processing time for a typical abdominal CT: 1 x 180 x 512 x 512 at (2.5, 0.8, 0.8) mm -> (1.5, 1.0, 1.0) mm."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from multitalent_amd.preprocessing.device_preprocessing import resample_and_normalize_ct
IP = {0: {'mean': 63.44, 'sd': 175.48, 'percentile_00_5': -927.0, 'percentile_99_5': 275.0}}
x = (torch.randn(1, 180, 512, 512, device='cuda') * 300).contiguous()
for sp0, name in (((2.5, 0.8, 0.8), 'separate z'), ((1.2, 0.8, 0.8), '3D cubic')):
    out = resample_and_normalize_ct(x, sp0, (1.5, 1.0, 1.0), IP); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        out = resample_and_normalize_ct(x, sp0, (1.5, 1.0, 1.0), IP)
    torch.cuda.synchronize()
    print('%s: %s -> %s in %.1f ms' % (name, tuple(x.shape), tuple(out.shape), (time.time() - t) / 5 * 1e3))
