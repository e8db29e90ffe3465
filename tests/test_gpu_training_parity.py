"""GPU parity of the TRAINING path against the CPU oracle (round-2 additions; VERDICT r1 "close the parity-coverage holes"):
a multi-step trajectory (Adam + CyclicLR, BatchNorm running statistics), whole-VAE gradients with a fixed eps, eval-mode /
test-time forwards of the full models and of the four inference halves, the VAE loss (P2P + KLD) value and gradient, the
KLD / reparameterisation kernels, every Laplacian convention, full-size steps (determinism), and BatchNorm with
|mean| >> std.  Everything runs through the C ABI (libicn.so); the oracle is only the checker.
"""
import copy

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import loss_ref, models_ref
from relu_pattern import capture_relu_outputs, check_flips, gradient_errors, relu_pattern
from relu_pattern import worst as worst_of

pytestmark = pytest.mark.gpu
TOL = 1e-4
GRAD_TOL = 1e-3         # network gradients at a fixed activation pattern (relu_pattern.py); the contract's bound is 2e-3


def _product(ref, name, R):
    from geniconet_amd import models
    net = getattr(models, name)(models.default_params(name, subdivisions=R))
    net.load_state_dict(ref.state_dict(), strict=True)
    return net.cuda()


def _zero_true_gradient(key):
    """Conv biases in front of a train-mode BatchNorm: BN removes the mean, so their true gradient is exactly zero and both
    sides only hold rounding noise there (which Adam then normalises to full-size steps)."""
    return key.endswith('.bias') and ('.conv0' in key or '.conv1' in key or key in ('encoder.0.bias', 'mu.0.bias', 'logvar.0.bias'))


# ---- (a) training trajectory: reference run.py:244-254 over several batches -------------------------------------------------
def _fixed_noise(monkeypatch, R, B, steps):
    """The same reparameterisation noise on both sides: the product draws it with torch.randn_like as the reference does
    (models.py:91); every draw of step state['k'] (set by the test) gets that step's tensor, on any device, in any dtype."""
    n = 2 ** (R - 3)
    noise = [torch.randn(B, 512, 5 * n, 2 * n, generator=torch.Generator().manual_seed(70 + k)) for k in range(steps)]
    state = {'k': 0}

    def fixed_randn_like(t, **kw):
        return noise[state['k'] % steps].to(device=t.device, dtype=t.dtype)
    monkeypatch.setattr(torch, 'randn_like', fixed_randn_like)
    return state


def _ulp(t):
    """Spacing of fp32 numbers at |t| (elementwise)."""
    return torch.from_numpy(np.spacing(np.abs(t.numpy()).astype(np.float32)))


@pytest.mark.parametrize('name', ['ico2ico', 'ico2ico_vae'])
def test_training_steps_teacher_forced_against_the_oracle_trainer(name, monkeypatch):
    """Product Trainer on the GPU against the same Trainer class driving the oracle network on the CPU, TEACHER-FORCED: before
    every step the GPU side receives the CPU trainer's weights, BatchNorm statistics and Adam state, so each step is compared
    from identical state and nothing is amplified from step to step (Adam's first steps are ~ lr * sign(g): free-running
    trajectories drift apart at rounding-noise-level gradient elements, which is why the earlier free-running form of this
    test needed a 15 % bound and a list of exempt tensors).  Per step k = 0..3, a different batch each:
      1. forward + loss + backward on both sides: loss to 1e-4 relative, BatchNorm running statistics to 1e-4; EVERY parameter
         gradient to 1e-3 rel-L2 (half the contract's 2e-3; denominators floored at 1e-3 of the largest gradient norm: conv
         biases in front of a train-mode BatchNorm have a zero true gradient) against the float64 oracle evaluated at the GPU
         forward's own ReLU activation pattern -- tests/relu_pattern.py; one acceptance path, no calibration;
      2. the CPU gradients are copied into the GPU parameters' .grad, and Adam + CyclicLR step on both sides from IDENTICAL
         gradients, moments, step counts and learning rate.  The HIP Adam (icn_adam_step) must then reproduce torch's update on
         EVERY tensor, no exemptions: exp_avg and exp_avg_sq to 1e-6 rel-L2, every weight element within 2 ulp (of the
         larger of |w_old|, |w_new|) + 1e-6 of its update + 3e-6 lr (= 3e-10 here, a sixth of an ulp of a typical weight) of the
         CPU result (fp32 weights quantise an update of size lr at
         ulp(w) / lr ~ 1e-5, so ulps of w -- not a relative bound on the update -- are the sharp statement), and, against a float64 evaluation of Adam's formula, the applied update to 1e-4
         rel-L2 of the update itself (a wrong bias correction, a stale moment or a skipped tensor is off by >= 1e-1 there)."""
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer, build_criterion
    R, B, STEPS = 3, 3, 4
    p = models.default_params(name, subdivisions=R)
    p[name].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)          # the reference's 1e-9 .. 1e-3 cycle moves nothing in 4 steps
    torch.manual_seed(5)
    ref = getattr(models_ref, name)(R=R).train()
    gpu = Trainer(p, 'cuda', model=_product(ref, name, R))
    cpu = Trainer(p, 'cpu', model=ref, criterion=build_criterion(p, 'cpu'), channels_last=False)
    assert type(gpu.optimizer).__module__ == 'geniconet_amd.optim'
    noise_step = _fixed_noise(monkeypatch, R, B, STEPS) if name == 'ico2ico_vae' else {'k': 0}
    names = [k for k, _ in cpu.model.named_parameters()]
    pg, pc = dict(gpu.model.named_parameters()), dict(cpu.model.named_parameters())
    relu_outs, _hooks = capture_relu_outputs(gpu.model)
    n_relu = 1 + 2 * (6 if name == 'ico2ico' else 5)              # stem + two per residual block
    flips_seen = []
    for k in range(STEPS):
        relu_outs.clear()
        # -- teacher forcing: CPU state -> GPU (weights + BN buffers, Adam moments and step counts, scheduler position)
        gpu.model.load_state_dict(cpu.model.state_dict())
        if k > 0:
            for key in names:
                sc, sg = cpu.optimizer.state[pc[key]], gpu.optimizer.state[pg[key]]
                sg['exp_avg'].copy_(sc['exp_avg'])
                sg['exp_avg_sq'].copy_(sc['exp_avg_sq'])
                sg['step'].copy_(sc['step'])
        assert abs(gpu.optimizer.param_groups[0]['lr'] - cpu.optimizer.param_groups[0]['lr']) < 1e-15
        x, t = data.synthetic_batch(B, R, seed=40 + k)
        noise_step['k'] = k
        before_fwd = copy.deepcopy(cpu.model.state_dict())         # for the float64 arbiter below
        # -- 1. forward, loss, backward (run.py:244-250)
        outs = {}
        for tr, xx, tt in ((gpu, x.cuda().contiguous(memory_format=torch.channels_last), t.cuda()), (cpu, x, t)):
            tr.optimizer.zero_grad()
            loss = tr.criterion(tr.net(xx), tt)
            loss.backward()
            outs[tr.device.type] = float(loss.detach())
        assert abs(outs['cuda'] - outs['cpu']) <= 1e-4 * abs(outs['cpu']), (k, outs)
        # -- gradients: the float64 oracle evaluated AT THE GPU'S ACTIVATION PATTERN (tests/relu_pattern.py: the gradient is a
        #    discontinuous function of the forward pass, a ReLU pre-activation within fp32 rounding of zero flips it by
        #    1e-3 .. 1e-2; with the pattern fixed both sides are smooth functions of identical structure) -- every tensor
        #    to GRAD_TOL, one acceptance path.  Where the GPU's pattern differs from the oracle's own, the oracle's
        #    pre-activation must be < 1e-4 of its tensor's rms (the flip is a rounding event, not a forward error).
        ref64 = getattr(models_ref, name)(R=R).train()
        ref64.load_state_dict(before_fwd)
        ref64 = ref64.double()
        with relu_pattern(relu_outs) as pat:
            out64 = ref64(x.double())
        build_criterion(p, 'cpu').double()(out64, t.double()).backward()
        n_flips = check_flips(pat, n_relu)
        errs = gradient_errors(pg, dict(ref64.named_parameters()))
        assert set(errs) == set(names)
        assert worst_of(errs)[0] < GRAD_TOL, (k, n_flips, sorted(((v, kk) for kk, v in errs.items()), reverse=True)[:5])
        flips_seen.append(n_flips)
        if n_flips:
            # what the flips do to a comparison that ignores them (the unpatterned float64 oracle): reported, and capped --
            # never grossly off, and the CPU fp32 oracle is the same distance from float64 at the tensors it flips itself
            ref64u = getattr(models_ref, name)(R=R).train()
            ref64u.load_state_dict(before_fwd)
            ref64u = ref64u.double()
            build_criterion(p, 'cpu').double()(ref64u(x.double()), t.double()).backward()
            eu = gradient_errors(pg, dict(ref64u.named_parameters()))
            over = sorted(((v, kk) for kk, v in eu.items() if v >= GRAD_TOL), reverse=True)
            print('step %d: %d ReLU flips %s; %d tensors beyond %.0e against the unpatterned oracle: %s'
                  % (k, n_flips, pat.flips, len(over), GRAD_TOL, over[:6]))
            assert all(v < 3e-2 for v, _ in over), over
        sg, sc = gpu.model.state_dict(), cpu.model.state_dict()
        for key in sc:
            if 'num_batches_tracked' in key:
                assert int(sg[key]) == int(sc[key]) == k + 1, key
            elif 'running' in key:
                assert rel_l2(sg[key].cpu().numpy(), sc[key].numpy()) < 1e-4, (k, key)
        # -- 2. identical gradients in, Adam + CyclicLR on both sides (run.py:251-254)
        lr = cpu.optimizer.param_groups[0]['lr']
        before = {key: pc[key].detach().clone() for key in names}
        with torch.no_grad():
            for key in names:
                pg[key].grad.copy_(pc[key].grad)
        for tr in (gpu, cpu):
            tr.optimizer.step()
            tr.scheduler.step()
        assert abs(gpu.scheduler.get_last_lr()[0] - cpu.scheduler.get_last_lr()[0]) < 1e-15
        b1, b2, eps = 0.9, 0.999, 1e-8
        for key in names:
            sgk, sck = gpu.optimizer.state[pg[key]], cpu.optimizer.state[pc[key]]
            assert float(sgk['step']) == float(sck['step']) == k + 1, key
            assert rel_l2(sgk['exp_avg'].cpu().numpy(), sck['exp_avg'].numpy()) < 1e-6, (k, key)
            assert rel_l2(sgk['exp_avg_sq'].cpu().numpy(), sck['exp_avg_sq'].numpy()) < 1e-6, (k, key)
            wg, wc = pg[key].detach().cpu(), pc[key].detach()
            # 2 ulp at the larger of |w_old|, |w_new| (an element may cancel to ~0: its own ulp says nothing there) + 1e-6 of
            # the element's update + 3e-6 * lr (when g_t nearly cancels the running mean, m_t is a small difference of large
            # terms: its rounding error, hence the update's, scales with the terms -- a step of ~lr -- not with the small result)
            tol = 2.0 * _ulp(torch.maximum(before[key].abs(), wc.abs())) + 1e-6 * (wc - before[key]).abs() + 3e-6 * lr
            worst = float(((wg - wc).abs() / tol).max())
            assert worst <= 1.0, (k, key, worst)
            # float64 Adam from the CPU side's (identical) inputs
            m = sck['exp_avg'].double()
            v = sck['exp_avg_sq'].double()
            want = -(lr / (1 - b1 ** (k + 1))) * m / ((v / (1 - b2 ** (k + 1))).sqrt() + eps)
            got = wg.double() - before[key].double()
            quant = float(_ulp(before[key]).double().norm())                  # what fp32 storage of w can resolve
            assert float((got - want).norm()) <= 1e-4 * float(want.norm()) + quant, (k, key)


def test_free_running_trajectory_smoke():
    """Three free-running steps of both trainers from one state_dict (no teacher forcing): a smoke check only -- the losses stay
    together to 1e-3 and the BatchNorm statistics to 1e-2; the sharp per-step statements are in the teacher-forced test."""
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer, build_criterion
    name, R, B = 'ico2ico', 3, 3
    p = models.default_params(name, subdivisions=R)
    p[name].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)
    torch.manual_seed(5)
    ref = getattr(models_ref, name)(R=R).train()
    gpu = Trainer(p, 'cuda', model=_product(ref, name, R))
    cpu = Trainer(p, 'cpu', model=ref, criterion=build_criterion(p, 'cpu'), channels_last=False)
    for k in range(3):
        x, t = data.synthetic_batch(B, R, seed=40 + k)
        lg = float(gpu.step(x.cuda().contiguous(memory_format=torch.channels_last), t.cuda()))
        lc = float(cpu.step(x, t))
        assert abs(lg - lc) <= 1e-3 * abs(lc), (k, lg, lc)
    sg, sc = gpu.model.state_dict(), cpu.model.state_dict()
    for key, vc in sc.items():
        if 'running' in key:
            assert rel_l2(sg[key].cpu().numpy(), vc.numpy()) < 1e-2, key


# ---- (a2) checkpoints with the product model and the HIP Adam: reference run.py:330-372 --------------------------------------
@pytest.mark.parametrize('name', ['ico2ico', 'ico2ico_vae'])
def test_checkpoint_save_load_next_step_identical_on_the_device(name, tmp_path, monkeypatch):
    """save_checkpoint (run.py:330-340) after two steps of the product trainer on the GPU, load_checkpoint (run.py:342-372:
    tensors through the CPU, key filter + strict load, optimizer state restored) into a differently initialised product trainer,
    then the next step on both: the loss, every weight, every BatchNorm buffer and the HIP Adam's state must be BIT-identical
    to the trainer that never stopped.  The reference does not save the CyclicLR position (run.py:369-370 restores the
    optimizer only), so the test carries the scheduler's state_dict across by hand."""
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer, load_checkpoint, save_checkpoint
    R, B = 3, 3
    p = models.default_params(name, subdivisions=R)
    p[name].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)
    noise_step = _fixed_noise(monkeypatch, R, B, 3) if name == 'ico2ico_vae' else {'k': 0}
    batches = [data.synthetic_batch(B, R, seed=60 + k) for k in range(3)]
    batches = [(x.cuda().contiguous(memory_format=torch.channels_last), t.cuda()) for x, t in batches]
    a = Trainer(p, 'cuda', seed=3)
    for k, (x, t) in enumerate(batches[:2]):
        noise_step['k'] = k
        a.step(x, t)
    path = save_checkpoint(a, str(tmp_path), 2, val_loss=0.5)
    assert path is not None and path.endswith('%s_E2.pt' % name)
    sched = a.scheduler.state_dict()
    b = Trainer(p, 'cuda', seed=4)                                # other weights, fresh optimizer
    assert not torch.equal(next(iter(a.model.parameters())), next(iter(b.model.parameters())))
    ckpt = load_checkpoint(b.model, str(tmp_path), name, epoch=2, optimizer=b.optimizer)
    assert ckpt['epoch'] == 2 and ckpt['loss'] == 0.5
    b.scheduler.load_state_dict(sched)
    b.optimizer.param_groups[0]['lr'] = a.optimizer.param_groups[0]['lr']
    for key, v in a.model.state_dict().items():
        assert torch.equal(v, b.model.state_dict()[key]), key
    noise_step['k'] = 2                                           # both next steps draw the same noise
    la, lb = a.step(*batches[2]), b.step(*batches[2])
    assert torch.equal(la, lb)
    sa, sb = a.model.state_dict(), b.model.state_dict()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), key
    pa, pb = list(a.model.parameters()), list(b.model.parameters())
    for qa, qb in zip(pa, pb):
        ea, eb = a.optimizer.state[qa], b.optimizer.state[qb]
        assert float(ea['step']) == float(eb['step']) == 3
        assert torch.equal(ea['exp_avg'], eb['exp_avg']) and torch.equal(ea['exp_avg_sq'], eb['exp_avg_sq'])
    assert a.scheduler.get_last_lr() == b.scheduler.get_last_lr()


# ---- (b) whole-VAE gradients with a fixed eps ----------------------------------------------------------------------------------
def test_vae_forward_and_gradients_with_fixed_noise(monkeypatch):
    """ico2ico_vae.forward (encode -> reparameterise -> decode, reference models.py:89-97) and the gradient of the full VAE
    objective 0.6/0.2/0.2 + KLD (run.py:694-696, losses.py:137-142) with respect to every parameter, against
    oracle.models_ref.ico2ico_vae.forward(x, eps) + the torch formulation of the loss on the CPU."""
    from geniconet_amd import data, models
    from geniconet_amd.train import build_criterion
    R, B = 3, 3
    torch.manual_seed(21)
    ref = models_ref.ico2ico_vae(R=R).train()
    net = _product(ref, 'ico2ico_vae', R).train()
    p = models.default_params('ico2ico_vae', subdivisions=R)
    crit_g, crit_c = build_criterion(p, 'cuda'), build_criterion(p, 'cpu')
    x, t = data.synthetic_batch(B, R, seed=9)
    eps = torch.randn(B, 512, 5, 2, generator=torch.Generator().manual_seed(3))
    out_r = ref(x, eps)
    crit_c(out_r, t).backward()
    monkeypatch.setattr(torch, 'randn_like', lambda s, **kw: eps.to(s.device))
    out_g = net(x.cuda().contiguous(memory_format=torch.channels_last))
    loss_g = crit_g(out_g, t.cuda())
    loss_g.backward()
    for k, (a, b) in enumerate(zip(out_g, out_r)):
        assert rel_l2(a.detach().cpu().numpy(), b.detach().numpy()) < TOL, k
    assert abs(float(loss_g) - float(crit_c.loss)) <= 1e-4 * abs(float(crit_c.loss))
    gr = dict(ref.named_parameters())
    floor = 1e-3 * max(float(q.grad.norm()) for q in gr.values())
    errs = {k: float((q.grad.cpu() - gr[k].grad).norm()) / max(float(gr[k].grad.norm()), floor)
            for k, q in net.named_parameters()}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 2e-3, (worst, errs[worst])
    assert set(errs) == set(gr) and len(errs) == 2 * 17 + 2 * 19 + 2          # every conv / BN / head parameter has a gradient


# ---- (c) eval mode and the test-time path ----------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['ico2ico', 'ico2ico_vae'])
def test_eval_mode_forward_matches_the_oracle(name):
    """validate() (run.py:280-296) and app.py:1447-1454 run the model in eval mode: HIP convolutions + BatchNorm on running
    statistics.  The statistics are made non-trivial by two training-mode batches on the oracle first.  Full models and the
    four inference halves (reference models.py:234-252,302-340), the latter restored by the reference's key filter
    (run.py:360-367); also Trainer.evaluate's loss, and experiment_test's variant with BatchNorm left in training mode
    (run.py:516)."""
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer, build_criterion
    R, B = 3, 3
    torch.manual_seed(31)
    ref = getattr(models_ref, name)(R=R).train()
    with torch.no_grad():
        for k in range(2):
            ref(data.synthetic_batch(4, R, seed=60 + k)[0])
    ref.eval()
    net = _product(ref, name, R).eval()
    x, t = data.synthetic_batch(B, R, seed=8)
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    p = models.default_params(name, subdivisions=R)

    def restore(half):
        md = half.state_dict()
        half.load_state_dict({k: v for k, v in ref.state_dict().items() if k in md})
        return half.cuda().eval()

    with torch.no_grad():
        if name == 'ico2ico':
            want = ref(x)
            assert rel_l2(net(xg).cpu().numpy(), want.numpy()) < TOL
            enc, dec = restore(models.ico2enc(p)), restore(models.enc2ico(p))
            lat = enc(xg)
            assert rel_l2(lat.cpu().numpy(), ref.encoder(x).numpy()) < TOL
            assert rel_l2(dec(lat).cpu().numpy(), want.numpy()) < TOL
            crit = build_criterion(p, 'cpu')
            tr = Trainer(p, 'cuda', model=net)
            got = float(tr.evaluate(xg, t.cuda()))
            assert abs(got - float(crit(want, t))) <= 1e-4 * abs(got) and tr.model.training
            tr.model.eval()
        else:
            mu_r, lv_r = ref.encode(x)
            mu_g, lv_g = net.encode(xg)
            assert rel_l2(mu_g.cpu().numpy(), mu_r.numpy()) < TOL and rel_l2(lv_g.cpu().numpy(), lv_r.numpy()) < TOL
            z = torch.randn(B, 512, 5, 2, generator=torch.Generator().manual_seed(1))
            want = ref.decode(z)
            assert rel_l2(net.decode(z.cuda()).cpu().numpy(), want.numpy()) < TOL
            enc, dec = restore(models.ico2enc_vae(p)), restore(models.enc2ico_vae(p))
            mu_h, lv_h = enc(xg)
            assert rel_l2(mu_h.cpu().numpy(), mu_r.numpy()) < TOL and rel_l2(lv_h.cpu().numpy(), lv_r.numpy()) < TOL
            assert rel_l2(dec(z.cuda())[0].cpu().numpy(), want.numpy()) < TOL
        # experiment_test leaves the freshly built model in training mode under no_grad (run.py:516): batch statistics
        ref.train()
        net.train()
        if name == 'ico2ico':
            assert rel_l2(net(xg).cpu().numpy(), ref(x).numpy()) < TOL
        else:
            mu_r, _ = ref.encode(x)
            assert rel_l2(net.encode(xg)[0].cpu().numpy(), mu_r.numpy()) < TOL


@pytest.mark.parametrize('name', ['ico2ico', 'ico2ico_vae'])
def test_weight_gradients_on_the_second_stream_change_nothing(name):
    """the Trainer puts the weight-gradient launches on a second HIP stream and joins once per backward pass
    ('deferred').  Same kernels, same order per tensor: weights, BatchNorm statistics and Adam state after four steps are
    BIT-identical to a trainer that keeps everything on one stream, at a small size and at I5 / batch 36; also the 'eager'
    mode (join before every Function returns) through the operators directly, and the mode switch's argument checks."""
    from geniconet_amd import data, models
    from geniconet_amd.ico_conv import set_weight_gradient_stream, wgrad_stream_counts
    from geniconet_amd.train import Trainer
    dev = torch.device('cuda', 0)
    for R, B in ((3, 3), (5, 36)) if name == 'ico2ico' else ((3, 2),):
        p = models.default_params(name, subdivisions=R)
        p[name].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)
        two, one = Trainer(p, dev, seed=21), Trainer(p, dev, seed=22)
        one.model.load_state_dict(two.model.state_dict())
        assert two.overlap_weight_gradients and two._weight_gradient_mode()[0] == 'deferred'
        one.overlap_weight_gradients = False
        x, t = data.synthetic_batch(B, R, seed=70, device=dev)
        x = x.contiguous(memory_format=torch.channels_last)
        for step in range(4):
            for tr in (two, one):
                torch.manual_seed(300 + step)                 # the VAE draws its noise from the default generator
                before = dict(wgrad_stream_counts)
                loss = tr.step(x, t)
                d_side, d_join = wgrad_stream_counts['side'] - before['side'], wgrad_stream_counts['joins'] - before['joins']
                assert (d_side >= 5 and d_join == 1) if tr is two else (d_side == 0 and d_join == 0), (name, step, d_side, d_join)
        assert torch.isfinite(loss)
        s2, s1 = two.model.state_dict(), one.model.state_dict()
        assert [k for k in s1 if not torch.equal(s1[k], s2[k])] == []
        for q2, q1 in zip(two.model.parameters(), one.model.parameters()):
            assert torch.equal(two.optimizer.state[q2]['exp_avg_sq'], one.optimizer.state[q1]['exp_avg_sq'])
        del two, one
    # the operators by themselves: 'off' unless somebody opts in; 'eager' waits before each backward Function returns
    from geniconet_amd.ico_conv import ico_conv
    assert set_weight_gradient_stream('off')[0] == 'off'
    xs = torch.randn(2, 64, 20, 8, device='cuda', requires_grad=True)
    w = (torch.randn(64, 64, 7, device='cuda') / 21).requires_grad_(True)
    b = torch.randn(64, device='cuda', requires_grad=True)
    want = torch.autograd.grad(ico_conv(xs, w, b, 2, 1, 'average').square().sum(), (xs, w, b))
    for mode in ('eager', 'deferred'):
        set_weight_gradient_stream(mode)
        try:
            got = torch.autograd.grad(ico_conv(xs, w, b, 2, 1, 'average').square().sum(), (xs, w, b))
        finally:
            set_weight_gradient_stream('off')
        assert all(torch.equal(g, h) for g, h in zip(got, want)), mode
    with pytest.raises(ValueError):
        set_weight_gradient_stream('sideways')
    with pytest.raises(ValueError):
        set_weight_gradient_stream('bucketed')                # needs the parameter -> bucket map


def test_second_stream_guards_readers_it_cannot_see():
    """ico_conv._readers_can_wait on the device (VERDICT r3 / ADVICE r3): under 'deferred' a weight gradient that autograd
    would READ on the current stream before the join stays there.
      (1) gradient accumulation: two backward passes without zero_grad -- in the second every .grad exists, AccumulateGrad runs
          `grad += dw` on the current stream;
      (2) zero_grad(set_to_none=False);
      (3) a module applied twice in one graph: the engine adds the two weight gradients on the current stream.
    All three bit-identical to mode 'off', with the counters showing who went where."""
    from geniconet_amd import data, models
    from geniconet_amd.ico_conv import IcoConvS2S, set_weight_gradient_stream, wgrad_stream_counts
    from geniconet_amd.train import build_criterion
    R, B = 4, 8
    p = models.default_params('ico2ico', subdivisions=R)
    crit = build_criterion(p, 'cuda')
    xs = [data.synthetic_batch(B, R, seed=90 + k, device='cuda') for k in range(2)]
    xs = [(x.contiguous(memory_format=torch.channels_last), t) for x, t in xs]

    def run(mode, set_to_none):
        torch.manual_seed(7)
        net = models.ico2ico(p).cuda().to(memory_format=torch.channels_last).train()
        counts = []
        for k, (x, t) in enumerate(xs):
            if k == 1 and not set_to_none:
                net.zero_grad(set_to_none=False)              # (2): grads stay allocated (zeros)
            before = dict(wgrad_stream_counts)
            prev = set_weight_gradient_stream(mode)
            try:
                crit(net(x), t).backward()                    # (1): no zero_grad between the two passes when set_to_none
            finally:
                set_weight_gradient_stream(*prev)
            counts.append({c: wgrad_stream_counts[c] - before[c] for c in before})
        torch.cuda.synchronize()
        return {k: q.grad.clone() for k, q in net.named_parameters()}, counts

    for set_to_none in (True, False):
        want, _ = run('off', set_to_none)
        got, counts = run('deferred', set_to_none)
        assert counts[0]['side'] >= 5 and counts[0]['kept'] == 0 and counts[0]['joins'] == 1, counts      # first pass: grads are None
        assert counts[1]['side'] == 0 and counts[1]['kept'] >= 5, counts                                   # second: every .grad exists
        assert [k for k in want if not torch.equal(want[k], got[k])] == []
    # (3) one IcoConvS2S applied twice in a graph
    torch.manual_seed(8)
    conv = IcoConvS2S(64, 64, 1, True, 3, 'average').cuda()
    x = torch.randn(4, 64, 40, 16, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_()
    res = {}
    for mode in ('off', 'deferred'):
        conv.zero_grad()
        x.grad = None
        before = dict(wgrad_stream_counts)
        prev = set_weight_gradient_stream(mode)
        try:
            conv(torch.relu(conv(x))).square().mean().backward()
        finally:
            set_weight_gradient_stream(*prev)
        torch.cuda.synchronize()
        res[mode] = (conv.weight.grad.clone(), conv.bias.grad.clone(), x.grad.clone(),
                     {c: wgrad_stream_counts[c] - before[c] for c in before})
    assert res['deferred'][3]['side'] == 1 and res['deferred'][3]['kept'] == 1, res['deferred'][3]
    assert all(torch.equal(a, b) for a, b in zip(res['off'][:3], res['deferred'][:3]))


def lagging_side_stream(monkeypatch):
    """The race detector: every weight-gradient launch that goes to the side stream is preceded THERE by a ~1 ms spin kernel, so
    the side stream lags the backward chain by tens of milliseconds and any reader that does not wait for it reads memory the
    weight gradient has not written yet.  Returns the counter of injected lags ({'n': ...})."""
    import importlib
    ico_conv = importlib.import_module('geniconet_amd.ico_conv')        # (the package attribute of that name is the function)
    real = ico_conv._wgrad_stream
    lag = {'n': 0}

    def lagging(dev, dests, *tensors, **kw):
        side = real(dev, dests, *tensors, **kw)
        if side is not None:
            with torch.cuda.stream(side):
                torch.cuda._sleep(2_000_000)                  # ~1 ms at 2.1 - 2.4 GHz
            lag['n'] += 1
        return side
    monkeypatch.setattr(ico_conv, '_wgrad_stream', lagging)
    return lag


def test_a_backward_pass_that_raises_does_not_leave_the_next_one_unjoined(monkeypatch):
    """the mode set globally ('deferred', as ICN_WGRAD_STREAM=deferred does -- no Trainer, no try / finally around
    backward()), a backward pass that raises half-way (a tensor hook on an encoder activation), and then an ordinary pass with a
    LAGGING side stream followed by a read of the gradients on the current stream, which is what optimizer.step() does.  The
    aborted pass left weight gradients in flight and its end-of-pass callback never ran; the next pass must still join before
    anybody reads: gradients bit-identical to mode 'off'."""
    from geniconet_amd import data, models
    from geniconet_amd.ico_conv import set_weight_gradient_stream, wgrad_stream_counts
    from geniconet_amd.train import build_criterion
    R, B = 3, 3
    p = models.default_params('ico2ico', subdivisions=R)
    crit = build_criterion(p, 'cuda')
    x, t = data.synthetic_batch(B, R, seed=91, device='cuda')
    x = x.contiguous(memory_format=torch.channels_last)

    class Boom(RuntimeError):
        pass

    def run(mode, lagged):
        torch.manual_seed(17)
        net = models.ico2ico(p).cuda().to(memory_format=torch.channels_last).train()
        prev = set_weight_gradient_stream(mode)               # set once, globally: nobody switches around the passes below
        try:
            def boom(g):
                raise Boom('injected')
            h = net.enc.register_forward_hook(lambda m, a, o: o.register_hook(boom) and None)
            with pytest.raises(Boom):                         # the decoder's weight gradients are issued, then the pass dies
                crit(net(x), t).backward()
            h.remove()
            net.zero_grad()
            before = dict(wgrad_stream_counts)
            crit(net(x), t).backward()                        # the next pass
            grads = {k: q.grad.clone() for k, q in net.named_parameters()}      # a read on the current stream, like optimizer.step
            torch.cuda.synchronize()
            return grads, {c: wgrad_stream_counts[c] - before[c] for c in before}
        finally:
            set_weight_gradient_stream(*prev)
    want, _ = run('off', False)
    lag = lagging_side_stream(monkeypatch)
    got, counts = run('deferred', True)
    assert lag['n'] >= 8 and counts['side'] >= 5 and counts['joins'] >= 1, (lag, counts)
    assert [k for k in want if not torch.equal(want[k], got[k])] == []


def test_a_lagging_side_stream_is_waited_for(monkeypatch):
    """A race detector for the second stream: every weight-gradient launch that goes to the side stream is preceded there by a
    ~1 ms spin kernel, so the side stream lags the backward chain by tens of milliseconds and ANY reader that does not wait
    for it -- the optimiser after a missing end-of-pass join, AccumulateGrad on an existing .grad, the engine adding the two
    gradients of a shared weight -- reads memory the weight gradient has not written yet.  With the joins and the
    per-launch guard in place the results stay bit-identical to one stream: three trainer steps, a gradient-accumulation
    pass, a module applied twice."""
    from geniconet_amd import data, models
    from geniconet_amd.ico_conv import IcoConvS2S, set_weight_gradient_stream
    from geniconet_amd.train import Trainer, build_criterion
    lag = lagging_side_stream(monkeypatch)
    dev = torch.device('cuda', 0)
    R, B = 3, 3
    p = models.default_params('ico2ico', subdivisions=R)
    p['ico2ico'].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)
    two, one = Trainer(p, dev, seed=31), Trainer(p, dev, seed=32)
    one.model.load_state_dict(two.model.state_dict())
    one.overlap_weight_gradients = False
    x, t = data.synthetic_batch(B, R, seed=71, device=dev)
    x = x.contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        two.step(x, t)
        one.step(x, t)
    assert lag['n'] >= 15                                     # the lag was really injected (>= 5 launches per step)
    s2, s1 = two.model.state_dict(), one.model.state_dict()
    assert [k for k in s1 if not torch.equal(s1[k], s2[k])] == []
    # accumulation: a second backward on top of existing gradients, under 'deferred' and under 'off'
    crit = build_criterion(p, dev)
    grads = {}
    for mode, tr in (('deferred', two), ('off', one)):
        tr.model.zero_grad()
        for _ in range(2):
            prev = set_weight_gradient_stream(mode)
            try:
                crit(tr.model(x), t).backward()
            finally:
                set_weight_gradient_stream(*prev)
        torch.cuda.synchronize()
        grads[mode] = {k: q.grad.clone() for k, q in tr.model.named_parameters()}
    assert [k for k in grads['off'] if not torch.equal(grads['off'][k], grads['deferred'][k])] == []
    # a module applied twice
    torch.manual_seed(9)
    conv = IcoConvS2S(64, 64, 1, True, 3, 'average').cuda()
    xs = torch.randn(4, 64, 40, 16, device='cuda').contiguous(memory_format=torch.channels_last)
    res = {}
    for mode in ('off', 'deferred'):
        conv.zero_grad()
        prev = set_weight_gradient_stream(mode)
        try:
            conv(torch.relu(conv(xs))).square().mean().backward()
        finally:
            set_weight_gradient_stream(*prev)
        torch.cuda.synchronize()
        res[mode] = (conv.weight.grad.clone(), conv.bias.grad.clone())
    assert all(torch.equal(a, b) for a, b in zip(res['off'], res['deferred']))
    # torch.autograd.grad (nothing is accumulated into .grad): the end-of-pass callback joins before the results are handed out
    from geniconet_amd.ico_conv import ico_conv as conv_fn
    xg = xs.clone().requires_grad_()
    w = (torch.randn(64, 64, 7, device='cuda') / 21).requires_grad_()
    b = torch.randn(64, device='cuda', requires_grad=True)
    got = {}
    for mode in ('off', 'deferred'):
        prev = set_weight_gradient_stream(mode)
        try:
            got[mode] = torch.autograd.grad(conv_fn(xg, w, b, 3, 1, 'average').square().sum(), (xg, w, b))
        finally:
            set_weight_gradient_stream(*prev)
        got[mode] = [g.clone() for g in got[mode]]
    assert all(torch.equal(a, b_) for a, b_ in zip(got['off'], got['deferred']))


def test_validation_after_fused_training_steps_uses_the_current_running_statistics():
    """the fused training path updates running_mean / running_var through raw pointers (no version bump), so
    the eval-mode cache of [mean | 1/std] must be dropped by the training step itself.  The loop of run.py:479-487 -- step,
    validate, step, validate -- against the same model evaluated through torch's own BatchNorm modules (hooked BatchNorms take
    the module path), each time."""
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer
    R, B = 3, 4
    p = models.default_params('ico2ico', subdivisions=R)
    p['ico2ico'].update(lr=1e-3, lr_base=1e-3, lr_max=1e-2)
    tr = Trainer(p, 'cuda', seed=5)
    xv, tv = data.synthetic_batch(B, R, seed=7, device='cuda')
    xv = xv.contiguous(memory_format=torch.channels_last)
    seen = []
    for k in range(3):
        x, t = data.synthetic_batch(B, R, seed=20 + k, device='cuda')
        tr.step(x.contiguous(memory_format=torch.channels_last), t)
        fused_val = float(tr.evaluate(xv, tv))                                 # fused eval path (cached statistics vector)
        hooks = [m.register_forward_hook(lambda *a: None) for m in tr.model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        plain_val = float(tr.evaluate(xv, tv))                                 # torch's eval-mode BatchNorm modules
        for h in hooks:
            h.remove()
        assert abs(fused_val - plain_val) <= 1e-5 * abs(plain_val), (k, fused_val, plain_val)
        seen.append(fused_val)
    assert len(set(seen)) == 3                                                # the statistics (and weights) did move


def test_inference_batchnorm_runs_fused_on_the_running_statistics():
    """Eval mode without an autograd graph (serving, `--process test`): relu(bn(a) [+ bn(b)]) is ONE pass of icn_bn_relu_fwd on
    the running statistics.  Against torch's own eval-mode modules to 1e-6; the cached [mean | 1/std] vector follows the
    statistics when they change; with a graph being recorded the torch modules are taken (their backward is torch's)."""
    import torch.nn.functional as F
    from geniconet_amd import fused
    torch.manual_seed(3)
    for C in (64, 128, 512):
        bns = [torch.nn.BatchNorm2d(C).cuda() for _ in range(2)]
        for bn in bns:
            with torch.no_grad():
                bn.running_mean.normal_(0.3, 0.5)
                bn.running_var.uniform_(0.2, 3.0)
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0, 0.3)
            bn.eval()
        a = torch.randn(3, C, 20, 8, device='cuda').contiguous(memory_format=torch.channels_last)
        b = torch.randn(3, C, 20, 8, device='cuda').contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            assert fused.can_fuse_eval(a, *bns) and not fused.can_fuse(a, *bns)
            want1, want2 = F.relu(bns[0](a)), F.relu(bns[0](a) + bns[1](b))
            got1, got2 = fused.bn_relu_eval(a, bns[0]), fused.bn_add_relu_eval(a, bns[0], b, bns[1])
            assert got1.shape == want1.shape and rel_l2(got1.cpu().numpy(), want1.cpu().numpy()) < 1e-6
            assert rel_l2(got2.cpu().numpy(), want2.cpu().numpy()) < 1e-6
            # the statistics move (a training step, load_state_dict): the cached vector must follow
            bns[0].running_mean.add_(0.7)
            bns[0].running_var.mul_(1.9)
            assert rel_l2(fused.bn_relu_eval(a, bns[0]).cpu().numpy(), F.relu(bns[0](a)).cpu().numpy()) < 1e-6
            bns[0].load_state_dict({k: v.clone() for k, v in bns[1].state_dict().items()})
            assert rel_l2(fused.bn_relu_eval(a, bns[0]).cpu().numpy(), F.relu(bns[1](a)).cpu().numpy()) < 1e-6
        assert not fused.can_fuse_eval(a, *bns)                      # a graph would be recorded for the BatchNorm parameters
        for bn in bns:
            bn.requires_grad_(False)
        assert fused.can_fuse_eval(a, *bns) and not fused.can_fuse_eval(a.clone().requires_grad_(True), *bns)
        bns[0].train()
        assert not fused.can_fuse_eval(a, *bns)


# ---- (d) VAE loss on the device ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('factor_kl', [1.0, 0.81])
def test_p2pkld_loss_value_and_gradient_match_the_oracle(factor_kl):
    """P2PKLD_Loss.forward (reference losses.py:137-142) = P2P(0.6, 0.2, 0.2) + factor_kl * KLD (losses.py:105): value, the
    reported terms, and the gradients with respect to the reconstruction, mu and logvar, against oracle/loss_ref.py."""
    from geniconet_amd.losses import P2PKLD_Loss
    r, B = 3, 2
    g = torch.Generator().manual_seed(17)
    n = 2 ** r
    pred = torch.randn(B, 3, 5 * n, 2 * n, generator=g)
    target = torch.randn(B, 9, 10 * n * n + 2, generator=g)
    mu, lv = torch.randn(B, 512, 5, 2, generator=g), 0.5 * torch.randn(B, 512, 5, 2, generator=g)
    crit = P2PKLD_Loss(r, 0.6, 0.2, 0.2, factor_kl).cuda()
    cl = torch.channels_last
    xs = [pred.cuda().contiguous(memory_format=cl).requires_grad_(), mu.cuda().contiguous(memory_format=cl).requires_grad_(),
          lv.cuda().contiguous(memory_format=cl).requires_grad_()]
    loss = crit(tuple(xs), target.cuda())
    (2.0 * loss).backward()
    rec = loss_ref.p2p_loss(pred.numpy(), target.numpy(), r, 0.6, 0.2, 0.2)
    kl = loss_ref.kld(mu.numpy(), lv.numpy())
    assert abs(float(loss) - (rec + factor_kl * kl)) <= 2e-5 * abs(rec + factor_kl * kl)
    got = crit.get_last_losses()
    assert abs(got[0] - rec) <= 2e-5 * abs(rec) and abs(-got[3] - kl) <= 2e-5 * abs(kl) and abs(got[4] - float(loss)) < 1e-6
    gm, gl = loss_ref.kld_grad(mu.numpy(), lv.numpy())
    assert rel_l2(xs[0].grad.cpu().numpy(), 2.0 * loss_ref.p2p_grad(pred.numpy(), target.numpy(), r, 0.6, 0.2, 0.2)) < 5e-5
    assert rel_l2(xs[1].grad.cpu().numpy(), 2.0 * factor_kl * gm) < 1e-5
    assert rel_l2(xs[2].grad.cpu().numpy(), 2.0 * factor_kl * gl) < 1e-5


def test_kld_and_reparameterise_kernels_against_torch():
    """icn_kld_* and icn_reparam_* against the reference's torch expressions on the same device (losses.py:105,
    models.py:89-92): contiguous and channels_last inputs, value and gradients; the KLD sum is deterministic."""
    from geniconet_amd import fused, losses
    g = torch.Generator(device='cuda').manual_seed(4)
    for shape, cl in (((3, 512, 5, 2), True), ((36, 512, 20, 8), True), ((4, 8, 5, 2), False)):
        mu = torch.randn(shape, device='cuda', generator=g)
        lv = 0.7 * torch.randn(shape, device='cuda', generator=g)
        if cl:
            mu, lv = (t.contiguous(memory_format=torch.channels_last) for t in (mu, lv))
        a = [mu.clone().requires_grad_(), lv.clone().requires_grad_()]
        b = [mu.clone().requires_grad_(), lv.clone().requires_grad_()]
        k1 = losses.kld(*a)
        fm, fl = torch.flatten(b[0], 1), torch.flatten(b[1], 1)
        k2 = torch.mean(-0.5 * torch.mean(1 + fl - fm.pow(2) - fl.exp(), dim=1), dim=0)
        assert abs(float(k1) - float(k2)) <= 1e-5 * abs(float(k2))
        assert float(losses.kld(*a)) == float(k1)                               # fixed two-level sum
        (3 * k1).backward()
        (3 * k2).backward()
        for p_, q_ in zip(a, b):
            assert rel_l2(p_.grad.cpu().numpy(), q_.grad.cpu().numpy()) < 1e-5
        eps = torch.randn(shape, device='cuda', generator=g)
        gz = torch.randn(shape, device='cuda', generator=g)
        a = [mu.clone().requires_grad_(), lv.clone().requires_grad_()]
        b = [mu.clone().requires_grad_(), lv.clone().requires_grad_()]
        z1 = fused._ReparamFn.apply(a[0], a[1], eps)
        z2 = eps * torch.exp(0.5 * b[1]) + b[0]
        assert rel_l2(z1.detach().cpu().numpy(), z2.detach().cpu().numpy()) < 1e-6
        z1.backward(gz)
        z2.backward(gz)
        for p_, q_ in zip(a, b):
            assert rel_l2(p_.grad.cpu().numpy(), q_.grad.cpu().numpy()) < 1e-6
    torch.manual_seed(11)
    z_a = fused.reparameterize(mu, lv)
    torch.manual_seed(11)
    z_b = torch.randn_like(torch.exp(0.5 * lv)) * torch.exp(0.5 * lv) + mu      # the reference's expression, same generator
    assert rel_l2(z_a.cpu().numpy(), z_b.cpu().numpy()) < 1e-6


@pytest.mark.parametrize('mode', ['v-mean', 'sum-kv', 'kv-sum'])
def test_hip_loss_laplacian_conventions(mode):
    """The Laplacian convention is an option of the loss (upstream's is unknown): HIP value and gradient for the
    non-default conventions against the numpy oracle."""
    from geniconet_amd.losses import LAPLACIAN_MODES, P2P_Loss
    r, B = 2, 2
    g = torch.Generator().manual_seed(77)
    n = 2 ** r
    pred = torch.randn(B, 3, 5 * n, 2 * n, generator=g)
    target = torch.randn(B, 9, 10 * n * n + 2, generator=g)
    code = LAPLACIAN_MODES[mode]
    crit = P2P_Loss(r, 0.3, 0.2, 0.5, laplacian=mode).cuda()
    x = pred.cuda().requires_grad_()
    loss = crit(x, target.cuda())
    loss.backward()
    want = loss_ref.p2p_terms(pred.numpy(), target.numpy(), r, code)
    assert abs(float(crit.last_loss_lap) - want[2]) <= 2e-5 * want[2]
    assert abs(float(loss) - loss_ref.p2p_loss(pred.numpy(), target.numpy(), r, 0.3, 0.2, 0.5, code)) <= 2e-5 * abs(float(loss))
    assert rel_l2(x.grad.cpu().numpy(), loss_ref.p2p_grad(pred.numpy(), target.numpy(), r, 0.3, 0.2, 0.5, code)) < 5e-5


# ---- (e) full-size steps: size-independent checks -------------------------------------------------------------------------------
@pytest.mark.parametrize('lag', ['plain', 'lagging_side_stream'])
@pytest.mark.parametrize('cfg', [('ico2ico', 5, 36), ('ico2ico', 6, 8), ('ico2ico_vae', 5, 36)], ids=lambda c: '%s_I%d_b%d' % c)
def test_full_size_training_step_is_finite_and_deterministic(cfg, lag, monkeypatch):
    """BASELINE configs 2, 4 and 5 at their real sizes: two training steps through Trainer.step give a finite loss, finite
    weights, running statistics inside the data's range, and -- every reduction on the path being fixed-order (wgrad slabs,
    BatchNorm / loss / head two-level sums, no atomics) -- bit-identical results when repeated from the same state.
    'lagging_side_stream': the repetition runs with the race detector on (every side-stream launch behind a ~1 ms spin kernel),
    so the full-size steps, too, are bit-identical with a second stream that lags the chain by tens of milliseconds."""
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer
    name, R, B = cfg
    p = models.default_params(name, subdivisions=R)
    x, t = data.synthetic_batch(B, R, seed=1234, device='cuda')
    x = x.contiguous(memory_format=torch.channels_last)
    runs = []
    injected = None
    for rep in range(2):
        if rep == 1 and lag == 'lagging_side_stream':
            injected = lagging_side_stream(monkeypatch)
        tr = Trainer(p, 'cuda', seed=0)
        assert tr.overlap_weight_gradients
        torch.manual_seed(99)                                         # the VAE's noise
        losses_ = [tr.step(x, t) for _ in range(2)]
        runs.append(([float(v) for v in losses_], {k: v.clone() for k, v in tr.model.state_dict().items()}))
        del tr
    assert injected is None or injected['n'] >= 20, injected
    (la, sa), (lb, sb) = runs
    assert all(np.isfinite(la)) and la == lb, (la, lb)
    for k, v in sa.items():
        assert bool(torch.isfinite(v).all()), k
        assert torch.equal(v, sb[k]), k
    assert int(sa['encoder.1.num_batches_tracked']) == 2
    assert float(sa['encoder.1.running_var'].min()) > 0


# ---- (f) BatchNorm with |mean| >> std -------------------------------------------------------------------------------------------
@pytest.mark.parametrize('mean,std', [(50.0, 0.1), (-300.0, 1.0), (1e3, 1e-2)])
def test_fused_bn_with_large_mean_small_variance(mean, std):
    """Batch variance from shifted sums (icn_bn.hip): x = mean + std * randn must give the statistics, outputs and gradients
    of nn.BatchNorm2d (Welford) -- a plain E[x^2] - mean^2 in fp32 loses the variance here entirely."""
    from geniconet_amd import fused
    C = 64
    torch.manual_seed(2)
    bn, rf = torch.nn.BatchNorm2d(C).cuda().train(), torch.nn.BatchNorm2d(C).cuda().train()
    x64 = mean + std * torch.randn(4, C, 40, 16, dtype=torch.float64)
    x = x64.float().cuda().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(4, C, 40, 16, device='cuda')
    a1, a2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    y1 = fused.bn_relu(a1, bn)
    y2 = torch.relu(rf(a2))
    y1.backward(gy)
    y2.backward(gy)
    # float64 truth of the fp32-rounded input (what both implementations were given)
    xd = x.double().cpu()
    m64, v64 = xd.mean((0, 2, 3)), xd.var((0, 2, 3), unbiased=True)
    assert rel_l2(bn.running_mean.cpu().numpy(), (0.1 * m64).numpy()) < 1e-6
    # running_var = 0.9 * 1 + 0.1 * var in fp32: two ulps of 0.9 plus 1e-3 of the variance part
    rv_err = (bn.running_var.cpu().double() - (0.9 + 0.1 * v64)).abs()
    assert bool((rv_err <= 1.2e-7 + 1e-3 * 0.1 * v64).all()), float(rv_err.max())
    rv_err_torch = (rf.running_var.cpu().double() - (0.9 + 0.1 * v64)).abs()
    assert float(rv_err.max()) <= max(2 * float(rv_err_torch.max()), 2e-7)              # at least as good as the builtin
    yd = torch.relu((xd - m64[None, :, None, None]) / torch.sqrt(xd.var((0, 2, 3), unbiased=False) + 1e-5)[None, :, None, None])
    err_fused, err_torch = rel_l2(y1.detach().cpu().numpy(), yd.numpy()), rel_l2(y2.detach().cpu().numpy(), yd.numpy())
    # (x - mean) is exact in fp32, so what is left is the accuracy of mean and 1 / sqrt(var + eps)
    assert err_fused < max(2 * err_torch, 2e-4), (err_fused, err_torch)
    gd = _bn_relu_grad_f64(xd, gy.double().cpu(), m64, xd.var((0, 2, 3), unbiased=False))
    assert rel_l2(a1.grad.cpu().numpy(), gd.numpy()) < max(2 * rel_l2(a2.grad.cpu().numpy(), gd.numpy()), 1e-3)


def _bn_relu_grad_f64(x, gy, mean, var, eps=1e-5):
    """d relu(bn(x)) / dx in float64 (gamma = 1, beta = 0): the closed form the kernels implement."""
    inv = 1.0 / torch.sqrt(var + eps)
    xh = (x - mean[None, :, None, None]) * inv[None, :, None, None]
    g = gy * (xh > 0)
    m = x.numel() / x.shape[1]
    sg, sgx = g.sum((0, 2, 3)) / m, (g * xh).sum((0, 2, 3)) / m
    return inv[None, :, None, None] * (g - sg[None, :, None, None] - xh * sgx[None, :, None, None])


# ---- register-staged fallback kernels, in-process (icn_set_debug_flags) ---------------------------------------------------------
FALLBACK_CASES = [(2, 1, 64, 64, 2, 'average'), (3, 2, 128, 256, 2, 'average'), (4, 1, 128, 64, 2, 'average'),
                  (4, 1, 64, 320, 1, 'average'), (5, 1, 128, 128, 1, 'average'), (3, 1, 256, 128, 2, 'zeros')]


@pytest.fixture
def fallback_routing():
    from geniconet_amd import _lib
    old = _lib.lib().icn_set_debug_flags(48)              # 16: convs on k_gather_gemm, 32: weight gradients on k_wgrad
    yield
    _lib.lib().icn_set_debug_flags(old)


@pytest.mark.parametrize('case', FALLBACK_CASES, ids=lambda c: 'r%d_s%d_%dx%d_b%d_%s' % c)
def test_register_staged_fallback_kernels_stay_correct(case, fallback_routing):
    """k_gather_gemm and the non-DMA k_wgrad serve only tensors beyond 2 GiB and tiles with fewer than 4 K-steps, so the
    normal suite hardly reaches them: route every convolution to them (asserted through the profiling hooks) and repeat a
    few conv cases -- MFMA tiles of all shapes, stride 2, an odd width -- against the oracle."""
    from geniconet_amd import _lib
    from test_gpu_parity import conv_both
    _lib.profile_start(64)
    out = conv_both(*case, seed=11)
    used = {e['kernel'].split('<')[0] for e in _lib.profile_stop()}
    assert used == {'k_gather_gemm', 'k_wgrad'}, used
    for k, (got, want) in out.items():
        assert rel_l2(got.detach().cpu().numpy(), want.detach().numpy()) < TOL, k
