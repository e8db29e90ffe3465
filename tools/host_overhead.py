"""Host-side cost of one training step (developer tool; GPU only): with a batch of 1 the GPU work is a few hundred microseconds, so the
steady-state step time is the time Python + the launch path need to enqueue a step.

  python tools/host_overhead.py [--cores 2] [--ddp] [--graph]
--cores N : pin this process (and every thread it starts: autograd engine, DDP reducer) to N cores BEFORE anything touches the GPU and
            set OMP_NUM_THREADS=1 -- N = 2 is what a rank gets on the driver's box at 8 ranks (16 usable cores / 8).
--ddp     : ICN_FORCE_DDP=1 over an RCCL process group of one rank (everything N ranks would run on the host except the wire).
--graph   : the step as a HIP graph (Trainer(graph=True)); with --ddp the Trainer stays eager and says so.
"""
import argparse
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument('--cores', type=int, default=0)
ap.add_argument('--ddp', action='store_true')
ap.add_argument('--graph', action='store_true')
ap.add_argument('--steps', type=int, default=40)
a = ap.parse_args()
if a.cores > 0:
    avail = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, set(avail[:a.cores]))
    os.environ['OMP_NUM_THREADS'] = '1'
    os.environ['MKL_NUM_THREADS'] = '1'
if a.ddp:
    os.environ['ICN_FORCE_DDP'] = '1'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import data, models  # noqa: E402
from geniconet_amd.train import Trainer  # noqa: E402

if a.cores > 0:
    torch.set_num_threads(1)
if a.ddp:
    import torch.distributed as dist
    dist.init_process_group('nccl', rank=0, world_size=1)
R = 5
p = models.default_params('ico2ico', subdivisions=R)
tr = Trainer(p, torch.device('cuda', 0), seed=0, graph=a.graph)
tag = 'cores=%s ddp=%d graph=%d' % (a.cores or 'all', a.ddp, a.graph)
for B in (1, 36):
    x, t = data.synthetic_batch(B, R, seed=1, device='cuda')
    x = x.contiguous(memory_format=torch.channels_last)
    for _ in range(6):
        tr.step(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step(x, t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    why = tr.graph_usable() if a.graph else None
    print('%-28s batch %2d: enqueue %.2f ms/step, complete %.2f ms/step%s' % (
        tag, B, (t1 - t0) / a.steps * 1e3, (t2 - t0) / a.steps * 1e3, ('   [graph not used: %s]' % why) if why else ''), flush=True)
if a.ddp:
    dist.destroy_process_group()
