"""DDP wiring rehearsal on ONE GPU (developer tool): N ranks share cuda:0 and talk over gloo, as bench.py does under
ICN_BENCH_REHEARSAL=1, but with the problem size on the command line and a progress line per step.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         tools/dp_rehearsal.py R BATCH STEPS
"""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import data, models  # noqa: E402
from geniconet_amd.train import Trainer  # noqa: E402

R, B, STEPS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
faulthandler.dump_traceback_later(int(os.environ.get('WATCHDOG', 40)), exit=True, file=sys.stderr)
torch.cuda.set_device(0)
dist.init_process_group('gloo')
p = models.default_params('ico2ico', subdivisions=R)
tr = Trainer(p, torch.device('cuda', 0), seed=0)
x, t = data.synthetic_batch(B, R, seed=100 + rank, device='cuda')
x = x.contiguous(memory_format=torch.channels_last)
t0 = time.time()
for i in range(STEPS):
    loss = tr.step(x, t)
    torch.cuda.synchronize()
    print('[rank %d/%d] R=%d B=%d step %d done at %.2fs loss %.5f' % (rank, world, R, B, i, time.time() - t0, float(loss)), flush=True)
dist.barrier()
if rank == 0:
    print('REHEARSAL OK world=%d R=%d B=%d' % (world, R, B), flush=True)
dist.destroy_process_group()
