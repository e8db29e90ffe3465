"""The production communication path as far as ONE GPU can take it: an RCCL (backend 'nccl') process group of world size 1,
DistributedDataParallel around the product model exactly as geniconet_amd/train.py builds it for N > 1 (init_sync,
gradient_as_bucket_view, 10 MB buckets, torch's built-in C++ all-reduce comm hook, gradients written straight into the bucket
views from the third backward on), device barriers, destroy_process_group.

What this covers: communicator creation, DDP's reducer over the pair / upconv / BN / head / loss autograd Functions, the
bucket-view leases feeding icn_adam_step, the 'bucketed' join bookkeeping of the weight gradients' side stream, and
bit-identity with the plain trainer.  What it does NOT cover: any RCCL collective KERNEL -- with one rank the all-reduce
degenerates to buffer copies (the tracer of this very run shows no `nccl*` kernel: profiles/r03_ddp_vs_plain.txt, +17
copyBuffer / +20 fillBuffer per step), so a ring / tree kernel holding CUs beside the persistent conv kernels has never
executed here.  `test_two_ranks_over_rccl_average_gradients_and_stay_in_lock_step` below is that test; it needs two GPUs and
skips on the one-GPU boxes of this pool.

Checked: gradients and the weights / BatchNorm statistics after 6 optimiser + scheduler steps are BIT-IDENTICAL to the
trainer without DDP (a one-rank sum followed by a division by 1 must change nothing), at a small size and at the
BASELINE configs[1] size (I5, 36 meshes).  The worker is a freshly started interpreter (spawn): the process group is
created before anything else touches the GPU.  2 processes use the GPU (pytest + 1 worker)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, port, out_path):
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(420, exit=True)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    dist.init_process_group('nccl', device_id=device, rank=0, world_size=1)     # RCCL communicator, as bench.py creates it
    import importlib
    from geniconet_amd import _lib, data, models
    from geniconet_amd.train import Trainer
    # race detector (round 4): every weight-gradient launch on the side stream is preceded there by a ~1 ms spin kernel, so the side
    # stream lags the backward chain and a bucket whose all-reduce (here: its copies) is launched without the wait for the side
    # stream reads gradients that have not been written -- the bit-identity below then fails instead of passing by luck
    ic = importlib.import_module('geniconet_amd.ico_conv')
    real_side, lagged = ic._wgrad_stream, [0]

    def lagging(dev, dests, *tensors, **kw):
        side = real_side(dev, dests, *tensors, **kw)
        if side is not None:
            with torch.cuda.stream(side):
                torch.cuda._sleep(2_000_000)
            lagged[0] += 1
        return side
    ic._wgrad_stream = lagging
    report = {'backend': dist.get_backend(), 'cases': []}
    for name, R, B in (('ico2ico', 3, 3), ('ico2ico_vae', 3, 2), ('ico2ico', 5, 36)):
        p = models.default_params(name, subdivisions=R)
        p[name].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)
        plain = Trainer(p, device, seed=11, force_ddp=False)
        ddp = Trainer(p, device, seed=12, force_ddp=True)        # other initial weights: DDP is built first, then aligned
        assert isinstance(ddp.net, torch.nn.parallel.DistributedDataParallel) and plain.net is plain.model
        ddp.model.load_state_dict(plain.model.state_dict())
        x, t = data.synthetic_batch(B, R, seed=77, device=device)
        x = x.contiguous(memory_format=torch.channels_last)
        case = {'name': name, 'R': R, 'B': B, 'grad_mismatch': [], 'weight_mismatch': [], 'bucket_views': 0}
        # one backward: the gradient every parameter ends up with
        for tr in (plain, ddp):
            torch.manual_seed(5)                                # the VAE draws its noise from the default generator
            tr.optimizer.zero_grad()
            tr.criterion(tr.net(x), t).backward()
        gp = dict(plain.model.named_parameters())
        for k, q in ddp.model.named_parameters():
            if q.grad is None or gp[k].grad is None or not torch.equal(q.grad, gp[k].grad):
                case['grad_mismatch'].append(k)
        # after the first backward DDP has re-pointed .grad at views of its buckets (gradient_as_bucket_view)
        case['params'] = len(gp)
        # three full steps (the weights moved above by nothing: no optimiser step yet; BN statistics moved equally on both)
        from geniconet_amd import _gradbuf
        from geniconet_amd.ico_conv import wgrad_stream_counts
        nsteps = 6          # the reducer rebuilds its buckets after the first iteration; the weight gradients' side stream is armed
        #                     once the bucket views have been the same for two steps in a row
        for step in range(nsteps):
            for tr in (plain, ddp):
                torch.manual_seed(100 + step)
                before, wb = dict(_gradbuf.counts), dict(wgrad_stream_counts)
                loss = tr.step(x, t)
                if tr is ddp:
                    case['leases_step%d' % step] = (_gradbuf.counts['view'] - before['view'], _gradbuf.counts['new'] - before['new'])
                case['side_%s_step%d' % ('ddp' if tr is ddp else 'plain', step)] = (
                    wgrad_stream_counts['side'] - wb['side'], wgrad_stream_counts['joins'] - wb['joins'])
            dist.barrier(device_ids=[0])
        case['nsteps'] = nsteps
        case['buckets'] = len(set(ddp._bucket_of.values())) if ddp._bucket_of else 0
        sp, sd = plain.model.state_dict(), ddp.model.state_dict()
        for k in sp:
            if not torch.equal(sp[k], sd[k]):
                case['weight_mismatch'].append(k)
        case['finite'] = bool(torch.isfinite(loss))
        case['status'] = _lib.device_status(device)
        case['lagged'] = lagged[0]
        report['cases'].append(case)
        del plain, ddp
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    torch.save(report, out_path)


@pytest.mark.timeout(900)
def test_ddp_over_rccl_at_world_size_one_is_bit_identical_to_the_plain_trainer(tmp_path):
    out = str(tmp_path / 'report.pt')
    mp.spawn(_worker, args=(_free_port(), out), nprocs=1, join=True)
    rep = torch.load(out)
    assert rep['backend'] == 'nccl'
    assert [c['name'] for c in rep['cases']] == ['ico2ico', 'ico2ico_vae', 'ico2ico']
    for c in rep['cases']:
        assert c['grad_mismatch'] == [], (c['name'], c['R'], c['grad_mismatch'][:5])
        assert c['weight_mismatch'] == [], (c['name'], c['R'], c['weight_mismatch'][:5])
        assert c['finite'] and c['status'] == 0 and c['lagged'] >= 30, c       # (the side stream really lagged)
        # in the last step every gradient the package's backward kernels produce went straight into a bucket view (all
        # parameters but the 4 of the VAE latent heads' torch-native BatchNorms), none into a new tensor
        served, new = c['leases_step%d' % (c['nsteps'] - 1)]
        assert served >= c['params'] - 4 and new == 0, c
        # weight gradients beside the backward chain: the plain trainer joins once per backward pass; under DDP the side stream
        # is armed by the last step, with one wait per bucket (+ the one at the end of the pass) -- and the weights above are
        # bit-identical all the same
        side, joins = c['side_plain_step%d' % (c['nsteps'] - 1)]
        assert side >= 5 and joins == 1, c
        side, joins = c['side_ddp_step%d' % (c['nsteps'] - 1)]
        # (the first bucket to complete may hold nothing from the side stream yet -- no wait then --, every later one waits)
        assert c['buckets'] >= 2 and side >= 5 and c['buckets'] <= joins <= c['buckets'] + 1, c


# ---- N = 2 over RCCL: runs the day a box has two GPUs ---------------------------------------------------------------------------
def _worker2(rank, port, out_dir):
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(420, exit=True)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    torch.cuda.set_device(rank)                                   # one GPU per rank; the process group before any other GPU call
    device = torch.device('cuda', rank)
    dist.init_process_group('nccl', device_id=device, rank=rank, world_size=2)
    from geniconet_amd import _lib, data, models
    from geniconet_amd.train import Trainer
    R, B = 4, 6                                                   # per-rank batch B, global batch 2 B
    p = models.default_params('ico2ico', subdivisions=R)
    p['ico2ico'].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)
    tr = Trainer(p, device, seed=20 + rank)                       # differently seeded replicas: rank 0's weights must win
    x, t = data.synthetic_batch(2 * B, R, seed=5, device=device)
    x = x.contiguous(memory_format=torch.channels_last)
    xs, ts = x[rank * B:(rank + 1) * B], t[rank * B:(rank + 1) * B]
    rep = {'rank': rank, 'world': dist.get_world_size(), 'backend': dist.get_backend()}
    rep['w0'] = {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}
    # one backward through DDP: every rank ends with the mean over ranks of the per-rank gradients
    tr.optimizer.zero_grad()
    tr.criterion(tr.net(xs), ts).backward()
    rep['grad'] = {k: q.grad.detach().cpu().clone() for k, q in tr.model.named_parameters()}
    # the same gradient without DDP: the plain model on this rank's shard (BatchNorm statistics are per rank by design)
    tr.model.zero_grad()
    tr.criterion(tr.model(xs), ts).backward()
    rep['grad_local'] = {k: q.grad.detach().cpu().clone() for k, q in tr.model.named_parameters()}
    for step in range(4):                                        # lock-step: Adam + CyclicLR replicated
        tr.step(xs, ts)
    dist.barrier(device_ids=[rank])
    rep['w4'] = {k: v.detach().cpu() for k, v in tr.model.state_dict().items() if 'running' not in k and 'num_batches' not in k}
    rep['status'] = _lib.device_status(device)
    torch.save(rep, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier(device_ids=[rank])
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL refuses two ranks on one device)')
def test_two_ranks_over_rccl_average_gradients_and_stay_in_lock_step(tmp_path):
    """BASELINE configs[2] in miniature: 2 ranks, one GPU each, backend 'nccl' = RCCL, spawned fresh (process group before any
    other GPU call, no re-exec).  Rank 0's initial weights reach both ranks; after one backward both ranks hold the MEAN of the
    two per-rank gradients (each computed with its own BatchNorm statistics: DESIGN 6) -- compared with the per-rank
    gradients measured without DDP, to fp32 rounding of one addition; after four optimiser steps the replicas' weights are
    bit-identical.  This is the first test in which an RCCL collective kernel runs beside the persistent conv kernels and the
    weight gradients' side stream ('bucketed' mode)."""
    mp.spawn(_worker2, args=(_free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(str(tmp_path / ('rank%d.pt' % k))) for k in (0, 1))
    assert r0['backend'] == r1['backend'] == 'nccl' and r0['world'] == 2
    for k, v in r0['w0'].items():
        assert torch.equal(v, r1['w0'][k]), k                      # broadcast from rank 0
    for k in r0['grad']:
        assert torch.equal(r0['grad'][k], r1['grad'][k]), k        # one all-reduced result on both ranks
        want = 0.5 * (r0['grad_local'][k].double() + r1['grad_local'][k].double())
        err = float((r0['grad'][k].double() - want).norm()) / max(float(want.norm()), 1e-12)
        assert err < 1e-5, (k, err)
    for k, v in r0['w4'].items():
        assert torch.equal(v, r1['w4'][k]), k
    assert r0['status'] == 0 and r1['status'] == 0
