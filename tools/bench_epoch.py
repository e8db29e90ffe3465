"""Whole-epoch throughput through the device-resident dataset (developer tool; GPU only): train + validate over synthetic
I5 samples written to / read from .npz files, against the per-step rate bench.py reports.

  python tools/bench_epoch.py [N_TRAIN=288] [N_VAL=72] [BATCH=36]
"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import data, models, train  # noqa: E402

R = 5
n_trn, n_val, B = (int(a) for a in (sys.argv[1:4] + ['288', '72', '36'][len(sys.argv) - 1:]))
with tempfile.TemporaryDirectory() as d:
    t0 = time.time()
    for k in range(0, n_trn + n_val, 36):
        _, t = data.synthetic_batch(min(36, n_trn + n_val - k), R, seed=k, device='cuda')
        for j in range(t.shape[0]):
            data.save_sample(os.path.join(d, 'mesh%d.npz' % (k + j)), t[j].cpu().numpy())
    t1 = time.time()
    ds = data.IcoDataset(d, R, device='cuda')
    torch.cuda.synchronize()
    t2 = time.time()
print('wrote %d samples in %.1fs; listed, read and uploaded them in %.2fs (%.1f MB on the device)' % (
    len(ds), t1 - t0, t2 - t1, ds.targets.numel() * 4 / 1e6))
trn, val = ds.subset(range(n_trn)), ds.subset(range(n_trn, n_trn + n_val))
p = models.default_params('ico2ico', subdivisions=R)
tr = train.Trainer(p, 'cuda', seed=0)
train.fit(tr, trn, val, epochs=1, batch_size=B)                      # warm-up epoch (tables, allocator)
torch.cuda.synchronize()
t0 = time.time()
hist = train.fit(tr, trn, val, epochs=3, batch_size=B, first_epoch=2)
torch.cuda.synchronize()
dt = (time.time() - t0) / 3
print('epoch (train %d + validate %d meshes, batch %d): %.1f ms -> %.0f training meshes/s incl. validation; losses %s' % (
    n_trn, n_val, B, dt * 1e3, n_trn / dt, ['%.4f/%.4f' % (h[1], h[2]) for h in hist]))
