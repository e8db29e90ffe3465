"""Encoder/decoder topology of GenIcoNet's `ico2ico` autoencoder and `ico2ico_vae` on the HIP operators.

Mirrors the module tree of the reference's models.py so that state_dict keys, parameter counts and forward
hooks are interchangeable with it (reference file:line in each docstring); the schedule is written once as
data (channels per level) instead of per-layer literals, and the top level R is a parameter so that the
BASELINE I6 configuration (subdivisions=6) builds from the same code.  The reference hard-codes R=5
(models.py:108,115,120,125,138,143,148).

`params` is the reference's nested dict: params['ico']['corner_mode'|'subdivisions'],
params[<model_name>]['model'] ('residualS2S'), params['model_name'] for the VAE (models.py:222-225,258-266).
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import fused
from .ico_conv import IcoConvS2S, IcoUpsampleS2S

# channels after the stem and after each Down block (AE: 3 blocks -> R-3; VAE encoder: 2 blocks -> R-2)
STEM_CHANNELS = 64
DOWN_CHANNELS = (128, 256, 256)
UP_CHANNELS = (256, 128, 64)
VAE_LATENT_CHANNELS = 512


class BasicIcoS2SDownBlock(nn.Module):
    """relu(bn01(conv01(relu(bn00(conv00_s2(x))))) + bn10(conv10_s2(x)))   -- reference models.py:22-40."""

    def __init__(self, in_features, out_features, bias, in_subdivisions, corner_mode):
        super().__init__()
        kw = dict(bias=bias, corner_mode=corner_mode)
        self.conv00 = IcoConvS2S(in_features, out_features, stride=2, subdivisions=in_subdivisions, **kw)
        self.icobn00 = nn.BatchNorm2d(out_features)
        self.conv01 = IcoConvS2S(out_features, out_features, stride=1, subdivisions=in_subdivisions - 1, **kw)
        self.icobn01 = nn.BatchNorm2d(out_features)
        self.conv10 = IcoConvS2S(in_features, out_features, stride=2, subdivisions=in_subdivisions, **kw)
        self.icobn10 = nn.BatchNorm2d(out_features)

    def forward(self, x):
        c00, c10 = fused.conv_pair(x, self.conv00, self.conv10)             # both branches read x: one launch per pass
        if fused.can_fuse(x, self.icobn00, self.icobn01, self.icobn10):    # same math, fused HIP BN / ReLU passes
            h = fused.bn_relu(c00, self.icobn00)
            return fused.bn_add_relu(self.conv01(h), self.icobn01, c10, self.icobn10)
        if fused.can_fuse_eval(x, self.icobn00, self.icobn01, self.icobn10):   # inference: running statistics, one pass each
            h = fused.bn_relu_eval(c00, self.icobn00)
            return fused.bn_add_relu_eval(self.conv01(h), self.icobn01, c10, self.icobn10)
        main = self.icobn01(self.conv01(F.relu(self.icobn00(c00))))
        skip = self.icobn10(c10)
        return F.relu(main + skip)


class BasicIcoS2SUpBlock(nn.Module):
    """relu(bn01(conv01(relu(bn00(conv00(up00(x)))))) + bn10(conv10(up10(x))))   -- reference models.py:42-62.
    upsample00/upsample10 are parameter-free and see the same input; both module names are kept for hooks."""

    def __init__(self, in_features, out_features, bias, in_subdivisions, corner_mode):
        super().__init__()
        kw = dict(stride=1, bias=bias, subdivisions=in_subdivisions + 1, corner_mode=corner_mode)
        self.upsample00 = IcoUpsampleS2S(in_features, in_subdivisions, corner_mode)
        self.conv00 = IcoConvS2S(in_features, out_features, **kw)
        self.icobn00 = nn.BatchNorm2d(out_features)
        self.conv01 = IcoConvS2S(out_features, out_features, **kw)
        self.icobn01 = nn.BatchNorm2d(out_features)
        self.upsample10 = IcoUpsampleS2S(in_features, in_subdivisions, corner_mode)
        self.conv10 = IcoConvS2S(in_features, out_features, **kw)
        self.icobn10 = nn.BatchNorm2d(out_features)

    def forward(self, x):
        # upsample00 and upsample10 are parameter-free and see the same input (reference models.py:59-60): the r -> r+1
        # upsample is shared, and upsample + the two convolutions run as one composite gather-GEMM over the coarse tensor
        # when the shape allows (fused.upconv_pair; module by module when someone hooked one of them).
        c00, c10 = fused.upconv_pair(x, self.upsample00, self.upsample10, self.conv00, self.conv10)
        if fused.can_fuse(x, self.icobn00, self.icobn01, self.icobn10):
            h = fused.bn_relu(c00, self.icobn00)
            return fused.bn_add_relu(self.conv01(h), self.icobn01, c10, self.icobn10)
        if fused.can_fuse_eval(x, self.icobn00, self.icobn01, self.icobn10):   # inference: running statistics, one pass each
            h = fused.bn_relu_eval(c00, self.icobn00)
            return fused.bn_add_relu_eval(self.conv01(h), self.icobn01, c10, self.icobn10)
        main = self.icobn01(self.conv01(F.relu(self.icobn00(c00))))
        skip = self.icobn10(c10)
        return F.relu(main + skip)


class IcoUpS2S(nn.Module):
    """upsample then stride-1 conv (reference models.py:9-20; unused by the shipped models)."""

    def __init__(self, in_features, out_features, bias=True, subdivisions=0, corner_mode='zeros'):
        super().__init__()
        self.up = IcoUpsampleS2S(in_features, subdivisions, corner_mode)
        self.conv = IcoConvS2S(in_features, out_features, 1, bias, subdivisions + 1, corner_mode=corner_mode)

    def forward(self, x):
        return self.conv(self.up(x))


def _check_model(model):
    if model != 'residualS2S':
        raise ValueError("only model='residualS2S' is built (reference models.py:102,135); got %r" % (model,))


class _Encoder(nn.Sequential):
    """nn.Sequential (same child names '0', '1', ... as the reference's encoder, models.py:103-127) whose stem
    conv -> BatchNorm2d -> ReLU runs BN + ReLU through the fused HIP kernel when possible."""

    def forward(self, x):
        mods = list(self)
        x = mods[0](x)
        relu = mods[2]
        hooked = relu._forward_hooks or relu._forward_pre_hooks or relu._backward_hooks
        if isinstance(relu, nn.ReLU) and not hooked and fused.can_fuse(x, mods[1]):
            x = fused.bn_relu(x, mods[1])
        elif isinstance(relu, nn.ReLU) and not hooked and isinstance(mods[1], nn.BatchNorm2d) and fused.can_fuse_eval(x, mods[1]):
            x = fused.bn_relu_eval(x, mods[1])
        else:
            x = relu(mods[1](x))
        for m in mods[3:]:
            x = m(x)
        return x


def _encoder(corner_mode, subdivisions, n_down):
    layers = [IcoConvS2S(3, STEM_CHANNELS, 1, True, subdivisions, corner_mode),
              nn.BatchNorm2d(STEM_CHANNELS), nn.ReLU(inplace=False)]
    cin = STEM_CHANNELS
    for k in range(n_down):
        layers.append(BasicIcoS2SDownBlock(cin, DOWN_CHANNELS[k], True, subdivisions - k, corner_mode))
        cin = DOWN_CHANNELS[k]
    return _Encoder(*layers)


def _decoder(corner_mode, subdivisions, latent_channels):
    blocks, cin = [], latent_channels
    for k, cout in enumerate(UP_CHANNELS):
        blocks.append(BasicIcoS2SUpBlock(cin, cout, True, subdivisions - 3 + k, corner_mode))
        cin = cout
    head = nn.Sequential(nn.Conv2d(cin, 3, kernel_size=(1, 1)), nn.Tanh())
    return nn.Sequential(*blocks), head


def createico2enc(corner_mode='average', model='simple', subdivisions=5):
    """stem 3->64 @R, Down 64->128->256->256 to level R-3   -- reference models.py:101-132."""
    _check_model(model)
    return _encoder(corner_mode, subdivisions, 3)


def createenc2ico(corner_mode='average', model='simple', subdivisions=5):
    """Up 256->256->128->64 from level R-3 to R, then 1x1 conv 64->3 + tanh   -- reference models.py:134-159."""
    _check_model(model)
    return _decoder(corner_mode, subdivisions, DOWN_CHANNELS[-1])


def createico2enc_vae(corner_mode='average', model='simple', subdivisions=5):
    """stem + two Down blocks to level R-2   -- reference models.py:162-188."""
    _check_model(model)
    return _encoder(corner_mode, subdivisions, 2)


def createenc2ico_vae(corner_mode='average', model='simple', subdivisions=5):
    """Up 512->256->128->64   -- reference models.py:190-216."""
    _check_model(model)
    return _decoder(corner_mode, subdivisions, VAE_LATENT_CHANNELS)


def _levels(params):
    return params['ico'].get('subdivisions', 5)


class ico2ico(nn.Module):
    """Autoencoder: encoder -> Identity hook point `enc` -> decoder -> head   -- reference models.py:219-232."""

    def __init__(self, params):
        super().__init__()
        mode, kind, R = params['ico']['corner_mode'], params['ico2ico']['model'], _levels(params)
        self.encoder = createico2enc(mode, kind, R)
        self.enc = nn.Identity()
        self.subdivisions = R
        self.decoder, self.enc2icoConv = createenc2ico(mode, kind, R)

    def forward(self, x):
        with fused.counters_deferred():                  # one num_batches_tracked bump for the 19 BatchNorms
            return fused.head(self.decoder(self.enc(self.encoder(x))), self.enc2icoConv)


class ico2enc(nn.Module):
    """Encoder half for inference (reference models.py:234-241)."""

    def __init__(self, params):
        super().__init__()
        self.encoder = createico2enc(params['ico']['corner_mode'], params['ico2ico']['model'], _levels(params))

    def forward(self, x):
        return self.encoder(x)


class enc2ico(nn.Module):
    """Decoder half for inference (reference models.py:243-252)."""

    def __init__(self, params):
        super().__init__()
        self.subdivisions = _levels(params)
        self.decoder, self.enc2icoConv = createenc2ico(params['ico']['corner_mode'], params['ico2ico']['model'],
                                                       self.subdivisions)

    def forward(self, x):
        return fused.head(self.decoder(x), self.enc2icoConv)


class VAE(nn.Module):
    """Base class: reparameterisation z = eps * exp(logvar / 2) + mu   -- reference models.py:75-97."""

    def __init__(self):
        super().__init__()
        self.encoder = self.mu = self.logvar = self.decoder = None

    def encode(self, input):
        raise NotImplementedError

    def decode(self, input):
        raise NotImplementedError

    def reparameterize(self, mu, logvar):
        return fused.reparameterize(mu, logvar)

    def forward(self, x):
        with fused.counters_deferred():
            mu, logvar = self.encode(x)
            return self.decode(self.reparameterize(mu, logvar)), mu, logvar


def _latent_head(corner_mode, subdivisions):
    """IcoConvS2S 256->512 stride 2 at level R-2, then BN(512)   -- reference models.py:268-286."""
    return nn.Sequential(IcoConvS2S(DOWN_CHANNELS[1], VAE_LATENT_CHANNELS, 2, True, subdivisions - 2, corner_mode),
                         nn.BatchNorm2d(VAE_LATENT_CHANNELS))


def _latent_heads(mu, logvar, h):
    """(mu(h), logvar(h)): the two latent heads are `Sequential(IcoConvS2S, BatchNorm2d)` over the same tensor (reference
    models.py:288-292), so their convolutions run as a pair (one launch per pass) when nobody hooked the pieces."""
    plain = all(isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[0], IcoConvS2S)
                and not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks) for m in (mu, logvar))
    if not plain:
        return mu(h), logvar(h)
    a, b = fused.conv_pair(h, mu[0], logvar[0])
    return mu[1](a), logvar[1](b)


class ico2ico_vae(VAE):
    """Variational autoencoder   -- reference models.py:254-300."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.model = params[params['model_name']]['model']
        mode, R = params['ico']['corner_mode'], _levels(params)
        self.encoder = createico2enc_vae(mode, self.model, R)
        self.mu = self.createMu()
        self.logvar = self.createLogvar()
        self.mu_hook = nn.Identity()
        self.logvar_hook = nn.Identity()
        self.reparameterize_hook = nn.Identity()
        self.subdivisions = R
        self.decoder, self.final_layer = createenc2ico_vae(mode, self.model, R)

    def createMu(self):
        return _latent_head(self.params['ico']['corner_mode'], _levels(self.params))

    def createLogvar(self):
        return _latent_head(self.params['ico']['corner_mode'], _levels(self.params))

    def encode(self, input):
        m, lv = _latent_heads(self.mu, self.logvar, self.encoder(input))
        return self.mu_hook(m), self.logvar_hook(lv)

    def decode(self, z):
        return fused.head(self.decoder(self.reparameterize_hook(z)), self.final_layer)


class ico2enc_vae(VAE):
    """VAE encoder half (reference models.py:302-319)."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.model = params[params['model_name']]['model']
        self.encoder = createico2enc_vae(params['ico']['corner_mode'], self.model, _levels(params))
        self.mu = ico2ico_vae.createMu(self)
        self.logvar = ico2ico_vae.createLogvar(self)

    def encode(self, input):
        return _latent_heads(self.mu, self.logvar, self.encoder(input))

    def forward(self, x):
        return self.encode(x)


class enc2ico_vae(VAE):
    """VAE decoder half (reference models.py:321-340)."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.model = params[params['model_name']]['model']
        self.subdivisions = _levels(params)
        self.decoder, self.final_layer = createenc2ico_vae(params['ico']['corner_mode'], self.model, self.subdivisions)

    def createSample(self, batch_size, misc):
        mean, logvar = misc[0]['trn_mean'], misc[0]['trn_logvar']
        return torch.add(mean, logvar * torch.randn(logvar.shape))

    def decode(self, z):
        return fused.head(self.decoder(z), self.final_layer)

    def forward(self, x):
        return self.decode(x), torch.tensor([]), torch.tensor([])


def default_params(model_name='ico2ico', subdivisions=5, corner_mode='average'):
    """The slice of run.py's params dict (run.py:616-697) the models and losses read."""
    p = {'model_name': model_name,
         'ico': {'corner_mode': corner_mode, 'subdivisions': subdivisions, 'width': 2 ** (subdivisions + 1)},
         'ico2ico': {'model': 'residualS2S', 'loss': 'p2p', 'lr': 1e-6, 'lr_base': 1e-9, 'lr_max': 1e-3},
         'ico2ico_vae': {'model': 'residualS2S', 'loss': 'p2pkld', 'lr': 1e-6, 'lr_base': 1e-9, 'lr_max': 1e-3,
                         'factor_step_size': 25, 'factor_gamma': 0.9}}
    if model_name == 'ico2ico':          # run.py:689-692
        p['ico'].update(factor_pos=1., factor_nor=0., factor_lap=0.)
    else:                                 # run.py:693-696
        p['ico'].update(factor_pos=0.6, factor_nor=0.2, factor_lap=0.2)
    return p
