"""Host-side logic of the product (no GPU): module tree, state_dict keys, parameter counts, argument checks,
loud failure on CPU tensors, loss + data format against the numpy oracle, reference drop-in import."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import rel_l2
from geniconet_amd import data, geometry, losses, models
from geniconet_amd.ico_conv import IcoConvS2S, IcoUpsampleS2S
from oracle import loss_ref, models_ref

REFERENCE = '/root/reference'


def test_parameter_counts_and_keys():
    ae = models.ico2ico(models.default_params('ico2ico'))
    vae = models.ico2ico_vae(models.default_params('ico2ico_vae'))
    assert sum(p.numel() for p in ae.parameters()) == 4627715           # 7 taps per (Cin, Cout) pair
    assert sum(p.numel() for p in vae.parameters()) == 6004739
    keys = list(ae.state_dict())
    for k in ('encoder.0.weight', 'encoder.1.running_mean', 'encoder.3.conv00.weight', 'encoder.3.icobn00.weight',
              'encoder.5.conv10.bias', 'decoder.0.conv00.weight', 'decoder.2.icobn10.bias', 'enc2icoConv.0.weight'):
        assert k in keys
    vkeys = list(vae.state_dict())
    for k in ('mu.0.weight', 'mu.1.running_var', 'logvar.0.bias', 'final_layer.0.weight'):
        assert k in vkeys
    assert ae.encoder[0].weight.shape == (64, 3, 7) and vae.mu[0].weight.shape == (512, 256, 7)
    # the oracle's restated topology is key-compatible (parity tests load product weights into it)
    assert list(models_ref.ico2ico().state_dict()) == keys
    assert list(models_ref.ico2ico_vae().state_dict()) == vkeys
    # I6 configuration (BASELINE config 5) builds from the same factories
    ae6 = models.ico2ico(models.default_params('ico2ico', subdivisions=6))
    assert ae6.encoder[0].subdivisions == 6 and ae6.decoder[0].upsample00.subdivisions == 3
    assert sum(p.numel() for p in ae6.parameters()) == 4627715


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason='reference checkout only exists in the build container')
def test_reference_models_py_drops_in_unchanged():
    """The reference's own models.py imports `icocnn` (models.py:4-6); with this repo's icocnn package on the
    path it builds both networks with the same state_dict layout as our mirror."""
    sys.path.insert(0, REFERENCE)
    try:
        sys.modules.pop('models', None)
        ref = importlib.import_module('models')
        assert ref.__file__.startswith(REFERENCE)
        for name in ('ico2ico', 'ico2ico_vae'):
            p = models.default_params(name)
            a, b = getattr(ref, name)(p).state_dict(), getattr(models, name)(p).state_dict()
            assert list(a) == list(b)
            assert all(a[k].shape == b[k].shape for k in a)
    finally:
        sys.path.remove(REFERENCE)
        sys.modules.pop('models', None)


def test_operators_reject_cpu_tensors_loudly():
    conv = IcoConvS2S(3, 8, 1, True, 2, 'average')
    with pytest.raises(RuntimeError, match='HIP path'):
        conv(torch.zeros(1, 3, 20, 8))
    with pytest.raises(RuntimeError, match='HIP path'):
        IcoUpsampleS2S(3, 2, 'average')(torch.zeros(1, 3, 20, 8))


def test_constructor_argument_checks():
    with pytest.raises(ValueError):
        IcoConvS2S(3, 8, 3, True, 2, 'average')
    with pytest.raises(ValueError):
        IcoConvS2S(3, 8, 2, True, 0, 'average')
    with pytest.raises(ValueError):
        IcoConvS2S(3, 8, 1, True, 2, 'reflect')
    assert IcoConvS2S(3, 8, 1, False, 2).bias is None
    import icocnn
    assert icocnn.ico_conv.IcoConvS2S is IcoConvS2S
    assert icocnn.utils.ico_geometry.get_ico_faces(1).shape == (80, 3)     # run.py:28,144 access pattern


@pytest.mark.parametrize('r', [2, 3])
def test_losses_match_numpy_oracle(r):
    torch.manual_seed(1)
    n, N = 2 ** r, geometry.num_vertices(r)
    pred = torch.tanh(torch.randn(3, 3, 5 * n, 2 * n))
    _, tgt = data.synthetic_batch(3, r, seed=7)
    crit = losses.P2P_Loss(r, 0.6, 0.2, 0.2)
    loss = crit(pred, tgt)
    terms = loss_ref.p2p_terms(pred.numpy(), tgt.numpy(), r)
    mse, cos, lap, _, total = crit.get_last_losses()
    np.testing.assert_allclose([mse, cos, lap], terms, rtol=2e-5)
    np.testing.assert_allclose(total, 0.6 * terms[0] + 0.2 * terms[1] + 0.2 * terms[2], rtol=2e-5)
    assert abs(float(loss) - total) < 1e-7
    mu, lv = torch.randn(3, 8, 5, 2), torch.randn(3, 8, 5, 2) * 0.1
    kl = losses.P2PKLD_Loss(r, 0.6, 0.2, 0.2, 1.)
    out = kl((pred, mu, lv), tgt)
    np.testing.assert_allclose(float(out), total + loss_ref.kld(mu.numpy(), lv.numpy()), rtol=2e-5)
    kl.update_factor(25, 25, 0.9)
    assert abs(kl.get_factor() - 0.9) < 1e-12


def test_sample_format_round_trip(tmp_path):
    r = 2
    x, tgt = data.synthetic_batch(2, r, seed=3)
    assert x.shape == (2, 3, 20, 8) and tgt.shape == (2, 9, 162) and tgt.dtype == torch.float32
    assert float(tgt[:, :3].abs().max()) <= 0.95
    np.testing.assert_allclose(tgt[:, 3:6].norm(dim=1).numpy(), 1.0, atol=1e-5)
    # targets agree with the numpy oracle's normals / Laplacian
    f = geometry.get_ico_faces(r)
    v = tgt[0, :3].T.numpy().astype(np.float64)
    assert rel_l2(tgt[0, 3:6].T.numpy(), loss_ref.vertex_normals(v, f)) < 1e-5
    assert rel_l2(tgt[0, 6:9].T.numpy(), loss_ref.laplacian(v, f)) < 1e-4
    path = str(tmp_path / 'mesh.npz')
    data.save_sample(path, tgt[0].numpy())
    xi, lbl = data.load_sample(path, r)
    assert np.array_equal(lbl, tgt[0].numpy()) and np.array_equal(xi, x[0].numpy())
    assert list(np.load(path).keys()) == ['data']                          # generate.py:203 / data.py:66


def test_inference_halves_load_from_a_full_checkpoint_by_key_filter():
    """the encoder / decoder halves (reference models.py:234-252,302-340) are restored from a checkpoint of
    the full model exactly as the reference does it, run.py:360-367: keep the saved keys the half has, then a STRICT
    load_state_dict -- so every key of a half must exist in the full model under the same name."""
    import torch
    from geniconet_amd import models
    for name, enc_cls, dec_cls in (('ico2ico', models.ico2enc, models.enc2ico),
                                   ('ico2ico_vae', models.ico2enc_vae, models.enc2ico_vae)):
        p = models.default_params(name, subdivisions=3)               # the VAE's latent head strides down from level R - 2
        torch.manual_seed(1)
        full = getattr(models, name)(p)
        saved = {k: v.clone() for k, v in full.state_dict().items()}
        covered = set()
        for cls in (enc_cls, dec_cls):
            torch.manual_seed(2)                                       # different initial weights than the checkpoint
            half = cls(p)
            model_dict = half.state_dict()
            filtered = {k: v for k, v in saved.items() if k in model_dict}
            assert set(filtered) == set(model_dict), (name, cls.__name__, sorted(set(model_dict) - set(filtered))[:5])
            half.load_state_dict(filtered)                             # strict, like the reference
            for k, v in half.state_dict().items():
                assert torch.equal(v, saved[k]), k
            covered |= set(model_dict)
        assert covered == set(saved), (name, sorted(set(saved) - covered)[:5])   # nothing of the checkpoint is orphaned


def test_checkpoints_use_the_reference_format_and_resume_exactly(tmp_path):
    """run.py:317-372: {'model_state_dict', 'optimizer_state_dict', 'epoch', 'loss', 'misc'} at
    <logDir>/savedModel/<name>_E<epoch>.pt, best models as 'EB<int>', never overwritten, key-filtered strict load.
    Resuming (model + Adam state) continues bit-identically; run on the CPU restatement of the network."""
    import copy
    import torch
    from geniconet_amd import data, models, train
    from oracle import models_ref
    p = models.default_params('ico2ico', subdivisions=3)        # three stride-2 blocks: R >= 3
    for k in ('lr_base', 'lr_max'):                     # the reference does not checkpoint its scheduler; leave it out here
        p['ico2ico'].pop(k, None)

    def trainer(seed):
        torch.manual_seed(seed)
        net = models_ref.ico2ico(R=3).train()
        return train.Trainer(p, 'cpu', model=net, criterion=train.build_criterion(p, 'cpu'), channels_last=False)

    x, t = data.synthetic_batch(2, 3, seed=5)
    a = trainer(0)
    a.step(x, t)
    a.step(x, t)
    path = train.save_checkpoint(a, str(tmp_path), 'B3', val_loss=0.25, misc={'note': 1})
    assert path == str(tmp_path / 'savedModel' / 'ico2ico_EB3.pt')
    assert train.save_checkpoint(a, str(tmp_path), 'B3') is None                    # never overwrites (run.py:335-340)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert sorted(ck) == ['epoch', 'loss', 'misc', 'model_state_dict', 'optimizer_state_dict']
    assert ck['epoch'] == 3 and ck['loss'] == 0.25 and ck['misc'] == {'note': 1}
    train.save_checkpoint(a, str(tmp_path), 'B12')
    train.save_checkpoint(a, str(tmp_path), 7)                                      # a plain epoch file is not a "best"
    a.step(x, t)
    want = copy.deepcopy(a.model.state_dict())

    b = trainer(99)                                                                 # different weights, fresh Adam
    got = train.load_checkpoint(b.model, str(tmp_path), 'ico2ico', epoch=3, optimizer=None)
    assert got is None                                                              # 'E3' does not exist, only 'EB3'
    got = train.load_checkpoint(b.model, str(tmp_path), 'ico2ico', epoch=0, optimizer=b.optimizer)
    assert got['epoch'] == 12                                                       # newest best, natural order: B12 > B3
    b.step(x, t)
    for k, v in b.model.state_dict().items():
        assert torch.equal(v, want[k]), k

    half = models.ico2enc(p)                                                        # product encoder half, same keys
    assert train.load_checkpoint(half, str(tmp_path), 'ico2ico', epoch='B3') is not None
    assert torch.equal(half.state_dict()['encoder.0.weight'], ck['model_state_dict']['encoder.0.weight'])


def test_device_resident_dataset_and_epoch_loops(tmp_path):
    """SURVEY 8 f4 / run.py:233-329: samples listed in natural order (data.py:18), preloaded once, batched like the
    reference's DataLoader (shuffle per epoch, short last batch); validate = mean over batches of the total loss in eval
    mode; fit keeps '<name>_EB<epoch>.pt' checkpoints of the best validation losses.  CPU, oracle network."""
    import numpy as np
    import torch
    from geniconet_amd import data, models, train
    from oracle import models_ref
    R = 3
    x, t = data.synthetic_batch(7, R, seed=3)
    names = ['m1', 'm2', 'm10', 'm3', 'm11', 'm4', 'm20']
    d = tmp_path / 'trn'
    d.mkdir()
    for k, nme in enumerate(names):
        data.save_sample(str(d / (nme + '.npz')), t[k].numpy())
    (d / 'notes.txt').write_text('ignored')
    ds = data.IcoDataset(str(d), R)
    order = [names.index(n_) for n_ in ('m1', 'm2', 'm3', 'm4', 'm10', 'm11', 'm20')]
    assert [os.path.basename(f) for f in ds.files] == ['m1.npz', 'm2.npz', 'm3.npz', 'm4.npz', 'm10.npz', 'm11.npz', 'm20.npz']
    assert len(ds) == 7 and torch.equal(ds.targets, t[order])
    got = list(ds.batches(3))
    assert [b[0].shape[0] for b in got] == [3, 3, 1]                           # drop_last=False
    assert torch.equal(torch.cat([b[1] for b in got]), t[order])
    for img, lbl in got:                                                       # input = target[:3, :-2] on the grid
        ref_img, ref_lbl = data.load_sample(ds.files[0], R)
        assert img.shape[1:] == ref_img.shape and torch.equal(img, data.target_to_input(lbl, R))
    g = torch.Generator().manual_seed(1)
    seen = torch.cat([b[1] for b in ds.batches(3, shuffle=True, generator=g)])
    assert not torch.equal(seen, t[order])
    assert sorted(float(v) for v in seen[:, 0, 0]) == sorted(float(v) for v in t[:, 0, 0])   # a permutation of the samples

    p = models.default_params('ico2ico', subdivisions=R)
    torch.manual_seed(0)
    tr = train.Trainer(p, 'cpu', model=models_ref.ico2ico(R=R).train(), criterion=train.build_criterion(p, 'cpu'),
                       channels_last=False)
    trn, val = ds.subset([0, 1, 2, 3, 4]), ds.subset([5, 6])
    manual = np.mean([float(tr.evaluate(i_, l_)) for i_, l_ in val.batches(1)])
    assert abs(train.validate(tr, val, 1) - manual) < 1e-7 and tr.model.training
    hist = train.fit(tr, trn, val, epochs=3, batch_size=2, log_dir=str(tmp_path), seed=4)
    assert [h[0] for h in hist] == [1, 2, 3] and all(np.isfinite(h[1]) and np.isfinite(h[2]) for h in hist)
    best, saved = float('inf'), []
    for epoch, _, v in hist:
        if v <= best:
            best = v
            saved.append('ico2ico_EB%d.pt' % epoch)
    saved.append('ico2ico_E3.pt')                                              # the final saveModel (run.py:495-496)
    assert sorted(os.listdir(tmp_path / 'savedModel')) == sorted(saved) and len(saved) > 1
    ck = train.load_checkpoint(tr.model, str(tmp_path), 'ico2ico', epoch=0)
    assert abs(ck['loss'] - best) < 1e-12 and ck['misc'] is None              # misc only for the VAE losses


def test_fit_anneals_the_kl_factor_and_checkpoints_carry_the_vae_misc(tmp_path):
    """run.py:479-496 for the VAE: criterion.update_factor(epoch + 1, factor_step_size, factor_gamma) after every epoch
    (losses.py:116-118), misc = {'trn_mean', 'trn_logvar'} of the last training batch in every checkpoint (run.py:274-276;
    read by enc2ico_vae.createSample, models.py:329-332), a '<name>_E<epoch>.pt' every save_epoch_freq epochs and at the
    end.  CPU, oracle network."""
    import torch
    from geniconet_amd import data, models, train
    from oracle import models_ref
    R = 3
    p = models.default_params('ico2ico_vae', subdivisions=R)
    p['ico2ico_vae'].update(factor_step_size=2, factor_gamma=0.9, save_epoch_freq=2)
    x, t = data.synthetic_batch(5, R, seed=11)
    ds = data.IcoDataset.from_tensors(t, R)
    torch.manual_seed(0)
    tr = train.Trainer(p, 'cpu', model=models_ref.ico2ico_vae(R=R).train(), criterion=train.build_criterion(p, 'cpu'),
                       channels_last=False)
    assert tr.criterion.get_factor() == 1.0
    hist = train.fit(tr, ds.subset([0, 1, 2]), ds.subset([3, 4]), epochs=5, batch_size=2, log_dir=str(tmp_path), seed=1)
    assert len(hist) == 5
    assert abs(tr.criterion.get_factor() - 0.9 ** 2) < 1e-12                    # epochs 2 and 4
    files = sorted(os.listdir(tmp_path / 'savedModel'))
    for f in ('ico2ico_vae_E2.pt', 'ico2ico_vae_E4.pt', 'ico2ico_vae_E5.pt'):
        assert f in files, files
    ck = torch.load(str(tmp_path / 'savedModel' / 'ico2ico_vae_E5.pt'), map_location='cpu', weights_only=False)
    assert sorted(ck['misc']) == ['trn_logvar', 'trn_mean']
    n = 2 ** (R - 3)
    assert ck['misc']['trn_mean'].shape == (1, 512, 5 * n, 2 * n)              # last batch of 3 samples at batch size 2
    assert ck['misc']['trn_logvar'].shape == ck['misc']['trn_mean'].shape
    # the decoder half samples from it exactly as the reference does (models.py:329-332)
    half = models.enc2ico_vae(p)
    z = half.createSample(1, [ck['misc']])
    assert z.shape == ck['misc']['trn_mean'].shape and torch.isfinite(z).all()


@pytest.mark.parametrize('mode', ['mean-v', 'v-mean', 'sum-kv', 'kv-sum'])
def test_laplacian_convention_is_an_option_checked_against_the_dataset(tmp_path, mode):
    """Upstream's mesh.utils.compute_laplacian (generate.py:197 writes target rows 6:9 with it) is absent, so its sign and
    normalisation cannot be known offline: the loss takes the convention as an option, the dataset recovers it from the
    data and refuses a mismatch, and the CPU formulation agrees with the numpy oracle for every convention."""
    r = 2
    x, tgt = data.synthetic_batch(2, r, seed=5, laplacian=mode)
    found, errs = data.detect_laplacian_convention(tgt, r)
    assert found == mode and errs[mode] < 1e-5
    for k in range(2):
        data.save_sample(str(tmp_path / ('m%d.npz' % k)), tgt[k].numpy())
    ds = data.IcoDataset(str(tmp_path), r, laplacian=mode)
    assert ds.laplacian == mode and ds.subset([1]).laplacian == mode
    other = 'v-mean' if mode != 'v-mean' else 'mean-v'
    with pytest.raises(ValueError, match="they match '%s'" % mode):
        data.IcoDataset(str(tmp_path), r, laplacian=other)
    assert data.IcoDataset(str(tmp_path), r, laplacian=None).laplacian is None
    with pytest.raises(ValueError, match='laplacian must be one of'):
        losses.P2P_Loss(r, 1., 0., 0., laplacian='cotan')
    torch.manual_seed(2)
    n = 2 ** r
    pred = torch.tanh(torch.randn(2, 3, 5 * n, 2 * n))
    crit = losses.P2P_Loss(r, 0.6, 0.2, 0.2, laplacian=mode)
    crit(pred, tgt)
    code = losses.LAPLACIAN_MODES[mode]
    np.testing.assert_allclose(crit.get_last_losses()[:3], loss_ref.p2p_terms(pred.numpy(), tgt.numpy(), r, code), rtol=2e-5)
    p = models.default_params('ico2ico_vae', subdivisions=3)
    p['ico']['laplacian'] = mode
    from geniconet_amd import train
    assert train.build_criterion(p, 'cpu').laplacian == mode
