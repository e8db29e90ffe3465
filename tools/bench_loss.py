"""Time the loss of the training step in isolation (developer tool; GPU only): forward and forward+backward of the
P2P criterion on an I5 / batch-36 network output, HIP events, plus the optimiser step for comparison."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import data, models  # noqa: E402
from geniconet_amd.train import Trainer  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


R, B = 5, 36
MODEL = sys.argv[1] if len(sys.argv) > 1 else 'ico2ico'           # or ico2ico_vae (factors 0.6 / 0.2 / 0.2)
p = models.default_params(MODEL, subdivisions=R)
tr = Trainer(p, torch.device('cuda', 0), seed=0)
x, t = data.synthetic_batch(B, R, seed=1, device='cuda')
out = torch.randn(B, 3, 5 * 2 ** R, 2 ** (R + 1), device='cuda').contiguous(memory_format=torch.channels_last)
crit = tr.criterion
print('criterion:', type(crit).__name__, 'factors pos/nor/lap =', crit.factor_pos, crit.factor_nor, crit.factor_lap)


VAE = MODEL == 'ico2ico_vae'
mu = torch.randn(B, 512, 20, 8, device='cuda')


def arg(o):
    return (o, mu, mu) if VAE else o


def fwd():
    with torch.no_grad():
        return crit(arg(out), t)


def fwd_bwd():
    o = out.detach().requires_grad_()
    crit(arg(o), t).backward()


print('loss forward only      : %7.1f us' % timed(fwd))
print('loss forward + backward: %7.1f us' % timed(fwd_bwd))
if VAE:
    sys.exit(0)
tr.step(x.contiguous(memory_format=torch.channels_last), t)


def opt():
    tr.optimizer.step()
    tr.scheduler.step()


print('optimizer + scheduler  : %7.1f us   (host-bound if the GPU part is short)' % timed(opt))
