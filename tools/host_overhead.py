"""Host-side cost of one training step (developer tool; GPU only): with a batch of 1 the GPU work is a few hundred
microseconds, so the steady-state step time is the time Python + the launch path need to enqueue a step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import data, models  # noqa: E402
from geniconet_amd.train import Trainer  # noqa: E402

R = 5
p = models.default_params('ico2ico', subdivisions=R)
tr = Trainer(p, torch.device('cuda', 0), seed=0)
for B in (1, 36):
    x, t = data.synthetic_batch(B, R, seed=1, device='cuda')
    x = x.contiguous(memory_format=torch.channels_last)
    for _ in range(5):
        tr.step(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(x, t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('batch %2d: enqueue %.2f ms/step, complete %.2f ms/step' % (B, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
