"""the reference's OWN run.py / losses.py / ico_utils.py / data.py run unchanged against this package.

Only in the build container (the reference checkout is not on the GPU box, and nothing of it is copied): the import shims of
geniconet_amd/shims stand in for the reference's absent dependencies, the repo's `icocnn` package for its operators.  There
is no GPU here and the product has no CPU path, so for the end-to-end run the two operator classes of `icocnn.ico_conv` are
swapped for the CPU oracle's (oracle/ico_ref.py, same constructor signature): what is under test is everything AROUND the
operators -- imports, the params plumbing, dataset layout, the loss through the shimmed mesh helpers, the epoch loop,
checkpoints -- i.e. that run.py reaches and completes train() / validate() / saveModel() with this package on the path.
"""
import os
import runpy
import sys

import numpy as np
import pytest
import torch

REFERENCE = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, 'run.py')),
                                reason='reference checkout only exists in the build container')
REF_MODULES = ('run', 'models', 'losses', 'data', 'ico_utils')


@pytest.fixture
def reference_env(monkeypatch):
    import geniconet_amd.shims as shims
    saved_path, saved_mods = list(sys.path), {k: sys.modules.get(k) for k in REF_MODULES + ('torch.utils.tensorboard',)}
    had_tb = hasattr(torch.utils, 'tensorboard')
    shims.install()
    sys.path.insert(0, REFERENCE)
    for k in REF_MODULES:
        sys.modules.pop(k, None)
    yield shims
    sys.path[:] = saved_path
    for k, v in saved_mods.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    if not had_tb and hasattr(torch.utils, 'tensorboard') and saved_mods['torch.utils.tensorboard'] is None:
        delattr(torch.utils, 'tensorboard')
    for k in ('natsort', 'python_utils', 'torch_utils', 'torchsummary', 'mesh', 'mesh.utils', 'kaolin', 'kaolin.metrics',
              'kaolin.metrics.trianglemesh'):
        sys.modules.pop(k, None)


def test_reference_modules_import_and_its_loss_agrees_with_ours(reference_env):
    """import run (which imports ico_utils / data / losses / models and the five absent third-party modules) succeeds from the
    reference checkout, and the reference's own Point2Point_Loss, running on the shimmed mesh helpers, gives the values of
    geniconet_amd.losses and of the numpy oracle."""
    import importlib
    from geniconet_amd import data, losses
    from oracle import loss_ref
    ref_run = importlib.import_module('run')
    assert ref_run.__file__.startswith(REFERENCE)
    ref_losses = sys.modules['losses']
    assert ref_losses.__file__.startswith(REFERENCE) and sys.modules['ico_utils'].__file__.startswith(REFERENCE)
    r = 2
    n = 2 ** r
    torch.manual_seed(3)
    pred = torch.tanh(torch.randn(2, 3, 5 * n, 2 * n))
    _, tgt = data.synthetic_batch(2, r, seed=4)
    theirs = ref_losses.P2P_Loss(r, 0.6, 0.2, 0.2)
    ours = losses.P2P_Loss(r, 0.6, 0.2, 0.2)
    lt, lo = theirs(pred, tgt), ours(pred, tgt)
    assert abs(float(lt) - float(lo)) <= 1e-6 * abs(float(lo))
    want = loss_ref.p2p_terms(pred.numpy(), tgt.numpy(), r)
    np.testing.assert_allclose(theirs.get_last_losses()[:3], want, rtol=2e-5)
    # the reference's grid -> vertex list (ico_utils.py:10-24) and ours
    from geniconet_amd.losses import grid_to_vertices
    assert torch.equal(sys.modules['ico_utils'].output2vertices(r, pred), grid_to_vertices(pred, r))
    # the reference's kaolin call resolves to our metric
    v, f = torch.rand(7, 3), torch.tensor([[0, 1, 2], [2, 3, 4], [4, 5, 6]])
    d = sys.modules['ico_utils'].computeDistance(v + 0.01, v, f, 'unused.off', mode='point2mesh')
    assert d is not None and float(d) >= 0


@pytest.mark.timeout(900)
def test_reference_run_py_trains_end_to_end_with_this_package(reference_env, tmp_path, monkeypatch):
    """`python run.py --model ico2ico --process train --quickLearn 3 ...` executed as __main__ from the reference checkout:
    dataset in the reference's ModelNet layout (<dataPth>/<class>/{train,test}/*.npz, data.py:22-35) written by
    geniconet_amd.data.save_sample, one epoch of train() + validate(), checkpoints by the reference's saveModel -- and the
    checkpoint it writes loads into this package's ico2ico by the reference's key filter."""
    import icocnn.ico_conv as ico_conv_mod
    from geniconet_amd import data, models, train
    from oracle import ico_ref
    monkeypatch.setattr(ico_conv_mod, 'IcoConvS2S', ico_ref.IcoConvS2S)          # CPU stand-ins for the HIP operators
    monkeypatch.setattr(ico_conv_mod, 'IcoUpsampleS2S', ico_ref.IcoUpsampleS2S)
    R = 5                                                                        # the reference hard-codes I5 (models.py:108)
    _, t = data.synthetic_batch(4, R, seed=1)
    for split, idx in (('train', [0, 1]), ('test', [1, 2, 3])):                 # run.py:80-93 logs 3 validation samples
        d = tmp_path / 'data' / 'chair' / split
        d.mkdir(parents=True)
        for k in idx:
            data.save_sample(str(d / ('chair_%04d_ahs_I5.npz' % k)), t[k].numpy())
    log_dir = tmp_path / 'log'
    argv = ['run.py', '--model', 'ico2ico', '--process', 'train', '--quickLearn', '3', '--batch_size', '2', '--train_epoch', '1',
            '--logDir', str(log_dir), '--dataPth', str(tmp_path / 'data')]
    monkeypatch.setattr(sys, 'argv', argv)
    monkeypatch.chdir(REFERENCE)
    torch.set_num_threads(8)
    torch.manual_seed(0)
    runpy.run_path(os.path.join(REFERENCE, 'run.py'), run_name='__main__')
    saved = sorted(os.listdir(log_dir / 'savedModel'))
    assert 'ico2ico_E1.pt' in saved and any(s.startswith('ico2ico_EB1') for s in saved), saved
    assert os.path.isfile(log_dir / 'params.json')                              # torch_utils.save_params (run.py:723)
    ck = torch.load(str(log_dir / 'savedModel' / 'ico2ico_E1.pt'), map_location='cpu', weights_only=False)
    assert sorted(ck) == ['epoch', 'loss', 'misc', 'model_state_dict', 'optimizer_state_dict'] and ck['epoch'] == 1
    ours = models.ico2ico(models.default_params('ico2ico', subdivisions=R))
    got = train.load_checkpoint(ours, str(log_dir), 'ico2ico', epoch=1)
    assert got is not None and got['epoch'] == 1
    for k, v in ours.state_dict().items():
        assert torch.equal(v, ck['model_state_dict'][k]), k

    # ... and `--process test` (run.py:499-536) on the model it has just saved: loadModel picks the newest '_EB' file, the
    # forward runs with BatchNorm in training mode (run.py:516), ico_utils.computeDistance writes the output mesh through
    # python_utils.writeOffMesh and takes the point-to-mesh distance through the kaolin stand-in, saveDistance the CSV.
    for k in REF_MODULES:
        sys.modules.pop(k, None)
    argv = ['run.py', '--model', 'ico2ico', '--process', 'test', '--data_instance', 'val', '--quickLearn', '1', '--batch_size', '1',
            '--test_mode', 'point2mesh', '--write_output_mesh', '--logDir', str(log_dir), '--dataPth', str(tmp_path / 'data')]
    monkeypatch.setattr(sys, 'argv', argv)
    runpy.run_path(os.path.join(REFERENCE, 'run.py'), run_name='__main__')
    out_root = log_dir / 'data' / 'ico2ico'
    csvs = [os.path.join(r_, f) for r_, _, fs in os.walk(out_root) for f in fs if f.endswith('.csv')]
    offs = [os.path.join(r_, f) for r_, _, fs in os.walk(out_root) for f in fs if f.endswith('.off')]
    assert len(csvs) == 1 and len(offs) == 1, (csvs, offs)
    rows = open(csvs[0]).read().strip().splitlines()
    assert rows[0] == 'Name,Distance' and len(rows) == 2
    dist = float(rows[1].split(',')[1])
    assert np.isfinite(dist) and 0 < dist < 4.0                                  # mean squared distance inside the unit ball
    import python_utils
    v, f = python_utils.read_off(offs[0])
    assert len(v) == 10242 and len(f) == 20480


def test_shim_helpers():
    """The small helpers on their own: natural sort, OFF round trip, free file names, row-normalised adjacency."""
    import geniconet_amd.shims as shims
    shims.install()
    import importlib
    natsort = importlib.import_module('natsort')
    assert natsort.natsorted(['m10', 'm9', 'M2']) == ['M2', 'm9', 'm10']
    pu = importlib.import_module('python_utils')
    mu = importlib.import_module('mesh.utils')
    from geniconet_amd import geometry
    f = torch.from_numpy(geometry.get_ico_faces(1))
    adj = mu.compute_adjacency_matrix_sparse(42, f).to_dense()
    assert torch.allclose(adj.sum(1), torch.ones(42)) and sorted(set((adj > 0).sum(1).tolist())) == [5, 6]
    v = torch.rand(2, 42, 3)
    from geniconet_amd.losses import compute_laplacian_batch
    nbr = torch.from_numpy(geometry.vertex_neighbours(1).copy())
    w = (nbr >= 0).float() / (nbr >= 0).sum(1, keepdim=True)
    assert torch.allclose(mu.compute_laplacian_batch(v, mu.compute_adjacency_matrix_sparse(42, f)),
                          compute_laplacian_batch(v, nbr.clamp_min(0), w), atol=1e-6)
    assert torch.allclose(mu.compute_laplacian(v[0], mu.compute_adjacency_matrix_sparse(42, f)),
                          compute_laplacian_batch(v[:1], nbr.clamp_min(0), w)[0], atol=1e-6)


def test_off_files_and_free_names(tmp_path):
    import geniconet_amd.shims as shims
    shims.install()
    import importlib
    pu = importlib.import_module('python_utils')
    v, f = np.random.rand(5, 3), np.array([[0, 1, 2], [2, 3, 4]])
    path = pu.writeOffMesh(str(tmp_path / 'a' / 'mesh'), torch.from_numpy(v), torch.from_numpy(f))
    v2, f2 = pu.read_off(path)
    assert np.allclose(v2, v, rtol=1e-6) and f2 == f.tolist()
    first = pu.get_new_name(str(tmp_path / 'train_ico2ico'), '.jpg')
    open(first, 'w').close()
    assert pu.get_new_name(str(tmp_path / 'train_ico2ico'), '.jpg') != first
