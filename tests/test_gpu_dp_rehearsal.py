"""Data-parallel wiring THROUGH THE HIP OPERATORS on one GPU: two ranks share cuda:0 and exchange gradients over gloo,
staged through the host (geniconet_amd/train.py: _host_staged_allreduce_hook) -- the only multi-rank execution of the
product path a one-GPU box allows (RCCL refuses two ranks on one device; the 8-GPU run is the driver's).  What is under
test is that DistributedDataParallel's bucket hooks fire for the custom autograd Functions (pair convolutions, fused
BN/ReLU, head, loss), that the averaged gradients equal the single-process gradients of the shards' mean, and that the
replicas stay in lock-step through Adam + CyclicLR.  3 processes touch the GPU (parent + 2 ranks; the box allows 6)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, PER_RANK = 3, 2


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(240, exit=True)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer
    p = models.default_params('ico2ico', subdivisions=R)
    p['ico2ico'].update(lr=1e-3, lr_base=1e-4, lr_max=1e-3)
    tr = Trainer(p, 'cuda:0', seed=100 + rank)            # deliberately different initial weights per rank
    x, t = data.synthetic_batch(PER_RANK * world, R, seed=5)
    xs = x[rank * PER_RANK:(rank + 1) * PER_RANK].cuda().contiguous(memory_format=torch.channels_last)
    ts = t[rank * PER_RANK:(rank + 1) * PER_RANK].cuda()
    w0 = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    tr.model.eval()                                       # BatchNorm on running statistics: shards and full batch agree
    tr.criterion(tr.net(xs), ts).backward()
    grads = {k: q.grad.detach().cpu().clone() for k, q in tr.model.named_parameters()}
    tr.model.train()                                      # training mode: fused BN / ReLU kernels, per-rank statistics
    losses = [float(tr.step(xs, ts)) for _ in range(3)]
    torch.cuda.synchronize()
    torch.save({'w0': w0, 'grads': grads, 'losses': losses,
                'w_end': {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_average_gradients_through_the_hip_operators(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(str(tmp_path / ('rank%d.pt' % r))) for r in range(world))
    for k in a['w0']:                                     # rank-0 broadcast although each rank seeded differently
        assert torch.equal(a['w0'][k], b['w0'][k]), k
    for k in a['grads']:                                  # one averaged gradient on both ranks
        assert torch.equal(a['grads'][k], b['grads'][k]), k
    # ... equal to the single-process gradient of the full global batch (mean loss over 2 equal shards = full-batch loss)
    from geniconet_amd import data, models
    from geniconet_amd.train import build_criterion
    p = models.default_params('ico2ico', subdivisions=R)
    net = models.ico2ico(p)
    net.load_state_dict(a['w0'])
    net = net.cuda().to(memory_format=torch.channels_last).eval()
    x, t = data.synthetic_batch(PER_RANK * world, R, seed=5)
    build_criterion(p, 'cuda')(net(x.cuda().contiguous(memory_format=torch.channels_last)), t.cuda()).backward()
    ref = {k: q.grad.cpu() for k, q in net.named_parameters()}
    floor = 1e-3 * max(float(g.norm()) for g in ref.values())
    for k, g in ref.items():
        err = float((g - a['grads'][k]).norm()) / max(float(g.norm()), floor)
        assert err < 1e-4, (k, err)
    for k in a['w_end']:                                  # lock-step replicas after 3 optimiser + scheduler steps
        if 'running' in k or 'num_batches' in k:
            continue                                      # BatchNorm statistics are per rank by design
        assert torch.equal(a['w_end'][k], b['w_end'][k]), k
    assert all(abs(v) < 1e9 for v in a['losses'] + b['losses'])


@pytest.mark.timeout(900)
@pytest.mark.parametrize('launcher', ['self', 'torch.distributed.run'])
def test_bench_launch_forms_rehearsal(launcher):
    """Both ways the driver may start the N > 1 bench, two ranks each:
      'self'  -- `python3 bench.py --gpus 2` with NO WORLD_SIZE in the environment (the form it uses for N = 1): bench.py must start
                 its ranks itself (fresh interpreters; the launcher never touches the GPU);
      'torch.distributed.run' -- `python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
                 --master-port P bench.py --gpus 2` (the form the contract names).
    Either way the ranks meet over the backend and rank 0 prints ONE JSON line that says how many met and who launched them.  On
    this one-GPU box the ranks share cuda:0 over gloo (ICN_BENCH_REHEARSAL=1); on an 8-GPU node the same entries run one rank per
    GPU over RCCL."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(ICN_BENCH_REHEARSAL='1', PYTHONFAULTHANDLER='1')
    bench = [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1']
    if launcher == 'self':
        cmd = [sys.executable] + bench
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port())] + bench
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['value'] > 0 and out['scaling'] == 'weak'
    d = out['distributed']
    assert d['n_ranks_seen'] == 2 and d['backend'] == 'gloo' and d['launcher'] == launcher and d['ddp'] is True
    assert d['bucket_mb'] > 0 and d['visible_devices'] >= 1
    assert d['replicas_identical_after_run'] is True          # different data per rank, identical weights: the gradients were averaged
    assert 0 < d['rank_ms_per_step']['min'] <= d['rank_ms_per_step']['max'] <= out['ms_per_step'] + 1e-3
    assert out['config']['global_batch'] == 72 and 'cpu_baseline' not in out and 'also' not in out
    assert out['roofline'] is not None and out['roofline']['kernel'].startswith('k_')
