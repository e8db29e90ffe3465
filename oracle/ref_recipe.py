"""ORACLE (test infrastructure, NOT product code) -- the seeded recipes shared by oracle/make_ref_goldens.py (which runs the
reference's own Python in the build container) and the tests that hold the HIP path to the fixtures it wrote.

The 18.5 / 24 MB of model weights are too large to commit, so both sides re-create them from this recipe: numpy's PCG64
streams, keyed by the state_dict KEY (not by position), filled into whichever model is given -- the reference's
`models.ico2ico(params)` when the fixtures are made, `geniconet_amd.models.ico2ico(params)` on the GPU box.  The fixture
carries a sha256 over the tensors so that a recipe that drifted (another numpy, another key set) fails loudly instead of
comparing different networks.
"""
import hashlib
import zlib

import numpy as np

SEED = 20261004
SAMPLE_CAP = 8192           # a gradient tensor is stored whole up to this many elements, else as a strided sample of about that many


def _rng(*words):
    return np.random.default_rng([SEED] + [int(w) for w in words])


def _key_word(key):
    return zlib.crc32(key.encode())


def tensor_for(key, shape):
    """Value of state_dict entry `key` (float32 ndarray) -- None for entries the recipe leaves at their defaults
    (running statistics, num_batches_tracked)."""
    g = _rng(1, _key_word(key))
    leaf = key.rsplit('.', 1)[-1]
    if leaf in ('running_mean', 'running_var', 'num_batches_tracked'):
        return None
    if leaf == 'weight' and len(shape) == 3:                       # IcoConvS2S (Cout, Cin, 7): fan-in scaled
        b = 1.0 / np.sqrt(shape[1] * shape[2])
        return g.uniform(-b, b, size=shape).astype(np.float32)
    if leaf == 'weight' and len(shape) == 4:                       # Conv2d 1x1 head
        b = 1.0 / np.sqrt(shape[1])
        return g.uniform(-b, b, size=shape).astype(np.float32)
    if leaf == 'weight' and len(shape) == 1:                       # BatchNorm gamma: not the trivial 1
        return g.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if leaf == 'bias':                                             # conv / BatchNorm / head biases
        return (0.1 * g.standard_normal(size=shape)).astype(np.float32)
    raise KeyError('no recipe for %s %s' % (key, tuple(shape)))


def fill_model(model):
    """Overwrite every parameter of `model` by the recipe (in place); returns the sha256 over (key, bytes) in sorted key order."""
    import torch
    h = hashlib.sha256()
    sd = model.state_dict()
    with torch.no_grad():
        for key in sorted(sd):
            val = tensor_for(key, tuple(sd[key].shape))
            if val is None:
                continue
            sd[key].copy_(torch.from_numpy(val).reshape(sd[key].shape))
            h.update(key.encode())
            h.update(np.ascontiguousarray(val).tobytes())
    return h.hexdigest()


def noise(shape, stream):
    """Fixed N(0, 1) tensor (float32): the VAE's reparameterisation noise (stream 0) and anything else a case needs."""
    return _rng(2, stream).standard_normal(size=shape).astype(np.float32)


def projection_vector(key, numel):
    """Fixed unit-variance direction per tensor: <grad, vector> is stored beside the norm, so that every element of every
    gradient enters the comparison with a sign and a weight even where only a sample of the tensor is stored."""
    return _rng(3, _key_word(key)).standard_normal(size=numel)


def sample_index(numel):
    """Flat indices of the stored sample of a tensor with `numel` elements."""
    if numel <= SAMPLE_CAP:
        return np.arange(numel)
    step = -(-numel // SAMPLE_CAP)
    return np.arange(0, numel, step)


def shape_like_mesh(batch, r, stream):
    """A smooth closed surface in the chart layout + its (9, N) target rows, seeded: positions are a perturbed unit sphere
    sampled at the oracle's grid directions, rows 3:6 / 6:9 are seeded unit vectors / small vectors (the reference's loss
    only compares against them, losses.py:66-80, so they need not be the positions' true normals)."""
    from .make_goldens import synthetic_np
    pos = synthetic_np(batch, r, seed=SEED % 100000 + 17 * stream)                       # (B, N, 3) float64
    g = _rng(4, stream, r)
    nor = pos / np.linalg.norm(pos, axis=2, keepdims=True) + 0.2 * g.standard_normal(size=pos.shape)
    nor /= np.linalg.norm(nor, axis=2, keepdims=True)
    lap = 0.02 * g.standard_normal(size=pos.shape)
    return np.concatenate([pos, nor, lap], axis=2).transpose(0, 2, 1).astype(np.float32)  # (B, 9, N)
