"""Data-parallel path on CPU: world_size 2, gloo.  The product operators are GPU-only, so the replica is the CPU
oracle network injected into the product's Trainer; what is under test is the product's DP wiring
(geniconet_amd/train.py): rank-0 broadcast of the initial weights, per-rank shards, bucketed gradient averaging."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, PER_RANK = 3, 2


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer, build_criterion
    from oracle import models_ref
    p = models.default_params('ico2ico', subdivisions=R)
    p['ico2ico'].update(lr=1e-3, lr_base=1e-4, lr_max=1e-3)
    torch.manual_seed(100 + rank)                      # deliberately different initial weights per rank
    net = models_ref.ico2ico(R=R)
    tr = Trainer(p, 'cpu', model=net, criterion=build_criterion(p, 'cpu'), seed=100 + rank, channels_last=False)
    tr.model.eval()                                    # BatchNorm on running stats so shards and full batch agree
    x, t = data.synthetic_batch(PER_RANK * world, R, seed=5)
    xs, ts = x[rank * PER_RANK:(rank + 1) * PER_RANK], t[rank * PER_RANK:(rank + 1) * PER_RANK]
    w0 = {k: v.clone() for k, v in tr.model.state_dict().items()}
    loss = tr.criterion(tr.net(xs), ts)
    loss.backward()
    grads = {k: q.grad.clone() for k, q in tr.model.named_parameters()}
    losses = [float(tr.step(xs, ts)) for _ in range(3)]
    # the collective status check: one verdict for all ranks -- a failure reported by ONE rank raises on EVERY rank,
    # and a clean check passes everywhere; then only rank 0 writes the checkpoint and a stale '<name>.tmp<pid>' of a killed writer goes
    from geniconet_amd import train as icn_train
    tr.check_status_collective()
    real = tr.check_status

    def failing(synchronize=False):
        if rank == world - 1:
            raise RuntimeError('libicn: injected failure on rank %d' % rank)
    tr.check_status = failing
    try:
        tr.check_status_collective()
        raised = None
    except RuntimeError as e:
        raised = str(e)
    tr.check_status = real
    stale = os.path.join(out_dir, 'savedModel', 'ico2ico_E1.pt.tmp99999')
    if rank == 0:
        os.makedirs(os.path.dirname(stale), exist_ok=True)
        open(stale, 'w').write('half a checkpoint')
    dist.barrier()
    path = icn_train.save_checkpoint(tr, out_dir, 1, val_loss=0.0)
    torch.save({'w0': w0, 'grads': grads, 'losses': losses, 'w_end': tr.model.state_dict(), 'x': x, 't': t, 'raised': raised,
                'ckpt': path, 'stale_left': os.path.exists(stale)}, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradients_equal_single_process_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(str(tmp_path / ('rank%d.pt' % r))) for r in range(world))
    # identical initial weights (rank-0 broadcast) although each rank seeded differently
    for k in a['w0']:
        assert torch.equal(a['w0'][k], b['w0'][k]), k
    # averaged gradients are identical on both ranks ...
    for k in a['grads']:
        assert torch.equal(a['grads'][k], b['grads'][k]), k
    # ... and equal to the single-process gradient of the full global batch
    from geniconet_amd import models
    from geniconet_amd.train import build_criterion
    from oracle import models_ref
    p = models.default_params('ico2ico', subdivisions=R)
    net = models_ref.ico2ico(R=R)
    net.load_state_dict(a['w0'])
    net.eval()
    build_criterion(p, 'cpu')(net(a['x']), a['t']).backward()
    for k, q in net.named_parameters():
        err = float((q.grad - a['grads'][k]).norm() / q.grad.norm().clamp_min(1e-30))
        assert err < 1e-4, (k, err)
    # replicas stay in lock-step through optimiser + scheduler steps
    for k in a['w_end']:
        assert torch.equal(a['w_end'][k], b['w_end'][k]), k
    assert all(abs(x) < 1e9 for x in a['losses'])
    # collective status check: the failure injected on the LAST rank raised on both; the checkpoint was written by rank 0 only and the
    # stale temporary of a killed writer is gone
    assert a['raised'] and 'another rank' in a['raised'] and 'injected failure' in b['raised'], (a['raised'], b['raised'])
    assert a['ckpt'] and a['ckpt'].endswith('ico2ico_E1.pt') and b['ckpt'] is None
    assert not a['stale_left']


@pytest.mark.timeout(600)
def test_four_ranks_agree_and_finish(tmp_path):
    """world_size 4 (the scaling runs go to 8): same wiring; every rank ends with identical weights and the gradients on
    all ranks equal rank 0's.  Guards against collectives that only line up for two ranks."""
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(str(tmp_path / ('rank%d.pt' % r))) for r in range(world)]
    for o in outs[1:]:
        for k in outs[0]['w0']:
            assert torch.equal(outs[0]['w0'][k], o['w0'][k]), k
        for k in outs[0]['grads']:
            assert torch.equal(outs[0]['grads'][k], o['grads'][k]), k
        for k in outs[0]['w_end']:
            assert torch.equal(outs[0]['w_end'][k], o['w_end'][k]), k

