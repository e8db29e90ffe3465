"""The training step as ONE HIP graph (Trainer(graph=True) / ICN_GRAPH=1; SURVEY 7 step 7, the loop body of reference run.py:244-254):
forward + loss + backward (weight gradients on their side stream, forked from and joined to the capture stream inside the graph) + Adam,
with CyclicLR's learning rate and Adam's bias corrections as device scalars.  A replayed step must be the eager step, bit for bit."""
import numpy as np
import pytest
import torch

from test_gpu_training_parity import lagging_side_stream

pytestmark = pytest.mark.gpu


def _batches(R, B, n):
    from geniconet_amd import data
    out = []
    for k in range(n):
        x, t = data.synthetic_batch(B, R, seed=300 + k, device='cuda')
        out.append((x.contiguous(memory_format=torch.channels_last), t))
    return out


def _train(p, batches, steps, graph, seed=5):
    from geniconet_amd.train import Trainer
    tr = Trainer(p, 'cuda', seed=seed, graph=graph)
    losses = []
    for k in range(steps):
        x, t = batches[k % len(batches)]
        losses.append(float(tr.step(x, t)))              # (float(): read before the next replay overwrites the loss buffer)
    return tr, losses


@pytest.mark.parametrize('R,B', [(3, 3), (5, 4)])
def test_graph_replay_equals_the_eager_trainer_bit_for_bit(R, B):
    """7 steps over 3 different batches with a moving CyclicLR: parameters, BatchNorm statistics, Adam moments, step counts, learning
    rate and every loss equal the eager trainer's exactly; the graph was really replayed 5 times."""
    from geniconet_amd import models
    p = models.default_params('ico2ico', subdivisions=R)
    p['ico2ico'].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)
    batches = _batches(R, B, 3)
    eager, le = _train(p, batches, 7, False)
    graph, lg = _train(p, batches, 7, True)
    assert graph._g is not None and graph._g['n'] == 5 and eager._g is None
    assert le == lg, (le, lg)
    se, sg = eager.model.state_dict(), graph.model.state_dict()
    assert [k for k in se if not torch.equal(se[k], sg[k])] == []
    oe, og = eager.optimizer.state_dict()['state'], graph.optimizer.state_dict()['state']
    for i in oe:
        assert float(oe[i]['step']) == float(og[i]['step']) == 7.0
        assert torch.equal(oe[i]['exp_avg'], og[i]['exp_avg']) and torch.equal(oe[i]['exp_avg_sq'], og[i]['exp_avg_sq']), i
    assert eager.optimizer.param_groups[0]['lr'] == graph.optimizer.param_groups[0]['lr']
    assert eager.scheduler.last_epoch == graph.scheduler.last_epoch == 7
    from geniconet_amd import _lib
    assert _lib.device_status() == 0


def test_graph_capture_under_a_lagging_side_stream(monkeypatch):
    """The race detector inside the capture: every side-stream weight gradient sits behind a ~1 ms spin kernel IN THE GRAPH, so a
    replay whose join were missing would hand Adam unwritten gradients.  Bit-identical to the eager trainer on one stream."""
    from geniconet_amd import models
    from geniconet_amd.train import Trainer
    R, B = 3, 3
    p = models.default_params('ico2ico', subdivisions=R)
    p['ico2ico'].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)
    batches = _batches(R, B, 2)
    one = Trainer(p, 'cuda', seed=8, graph=False)
    one.overlap_weight_gradients = False
    for k in range(5):
        one.step(*batches[k % 2])
    lag = lagging_side_stream(monkeypatch)
    two, _ = _train(p, batches, 5, True, seed=8)
    assert lag['n'] >= 5 and two._g['n'] == 3, (lag, two._g['n'])
    s1, s2 = one.model.state_dict(), two.model.state_dict()
    assert [k for k in s1 if not torch.equal(s1[k], s2[k])] == []


def test_what_the_graph_path_does_not_cover_stays_eager():
    """The VAE (per-step noise, KL factor), keep_output and a per-step status check run the eager step and say why."""
    from geniconet_amd import models
    from geniconet_amd.train import Trainer
    R, B = 3, 2
    pv = models.default_params('ico2ico_vae', subdivisions=R)
    (x, t), = _batches(R, B, 1)
    tv = Trainer(pv, 'cuda', seed=1, graph=True)
    for _ in range(4):
        tv.step(x, t)
    assert tv._g is None and 'eager' in tv.graph_usable()
    pa = models.default_params('ico2ico', subdivisions=R)
    ta = Trainer(pa, 'cuda', seed=1, graph=True, check_device_status=True)
    for _ in range(4):
        ta.step(x, t)
    assert ta._g is None and ta.graph_usable() is not None
    tb = Trainer(pa, 'cuda', seed=1, graph=True)
    for _ in range(3):
        tb.step(x, t, keep_output=True)
    assert tb._g is None and tb.last_output is not None
    tb.step(x, t)
    assert tb._g is not None                                  # ... and the plain call is captured


def test_a_new_batch_shape_is_recaptured():
    from geniconet_amd import models
    from geniconet_amd.train import Trainer
    R = 3
    p = models.default_params('ico2ico', subdivisions=R)
    tr = Trainer(p, 'cuda', seed=2, graph=True)
    (x3, t3), = _batches(R, 3, 1)
    (x2, t2), = _batches(R, 2, 1)
    for _ in range(4):
        tr.step(x3, t3)
    g3 = tr._g
    assert g3 is not None and g3['n'] == 2
    for _ in range(2):                                        # the new shape's first two steps are eager again
        tr.step(x2, t2)
    assert tr._g is g3
    assert np.isfinite(float(tr.step(x2, t2))) and tr._g is not g3 and tr._g['shape'][0][0] == 2
