"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference network topology.

Follows /root/reference/models.py: Down block :22-40, Up block :42-62, encoder :101-127, decoder :134-155,
ico2ico :219-232, VAE :75-97, ico2ico_vae :254-300 -- on the CPU operators of oracle/ico_ref.py.
Module names equal the reference's so that a product state_dict loads with strict=True.
PARITY UNPINNED vs upstream (see ico_ref.py).  Only tests/, smoke() and bench.py's cpu_baseline import this.
"""
import torch
from torch import nn
from torch.nn import functional as F

from .ico_ref import IcoConvS2S, IcoUpsampleS2S


class Down(nn.Module):                                    # models.py:22-40
    def __init__(self, cin, cout, r, mode):
        super().__init__()
        self.conv00 = IcoConvS2S(cin, cout, 2, True, r, mode)
        self.icobn00 = nn.BatchNorm2d(cout)
        self.conv01 = IcoConvS2S(cout, cout, 1, True, r - 1, mode)
        self.icobn01 = nn.BatchNorm2d(cout)
        self.conv10 = IcoConvS2S(cin, cout, 2, True, r, mode)
        self.icobn10 = nn.BatchNorm2d(cout)

    def forward(self, x):
        a = self.icobn01(self.conv01(F.relu(self.icobn00(self.conv00(x)))))
        return F.relu(a + self.icobn10(self.conv10(x)))


class Up(nn.Module):                                      # models.py:42-62
    def __init__(self, cin, cout, r, mode):
        super().__init__()
        self.upsample00 = IcoUpsampleS2S(cin, r, mode)
        self.conv00 = IcoConvS2S(cin, cout, 1, True, r + 1, mode)
        self.icobn00 = nn.BatchNorm2d(cout)
        self.conv01 = IcoConvS2S(cout, cout, 1, True, r + 1, mode)
        self.icobn01 = nn.BatchNorm2d(cout)
        self.upsample10 = IcoUpsampleS2S(cin, r, mode)
        self.conv10 = IcoConvS2S(cin, cout, 1, True, r + 1, mode)
        self.icobn10 = nn.BatchNorm2d(cout)

    def forward(self, x):
        a = self.icobn01(self.conv01(F.relu(self.icobn00(self.conv00(self.upsample00(x))))))
        return F.relu(a + self.icobn10(self.conv10(self.upsample10(x))))


def encoder(mode, R, n_down):                             # models.py:101-127 / :162-183
    ch = [64, 128, 256, 256]
    mods = [IcoConvS2S(3, 64, 1, True, R, mode), nn.BatchNorm2d(64), nn.ReLU()]
    mods += [Down(ch[k], ch[k + 1], R - k, mode) for k in range(n_down)]
    return nn.Sequential(*mods)


def decoder(mode, R, latent):                             # models.py:134-155 / :190-211
    ch = [latent, 256, 128, 64]
    ups = nn.Sequential(*[Up(ch[k], ch[k + 1], R - 3 + k, mode) for k in range(3)])
    return ups, nn.Sequential(nn.Conv2d(64, 3, kernel_size=1), nn.Tanh())


class ico2ico(nn.Module):                                 # models.py:219-232
    def __init__(self, R=5, mode='average'):
        super().__init__()
        self.encoder = encoder(mode, R, 3)
        self.enc = nn.Identity()
        self.decoder, self.enc2icoConv = decoder(mode, R, 256)

    def forward(self, x):
        return self.enc2icoConv(self.decoder(self.enc(self.encoder(x))))


class ico2ico_vae(nn.Module):                             # models.py:254-300, VAE.forward :94-97
    def __init__(self, R=5, mode='average'):
        super().__init__()
        self.encoder = encoder(mode, R, 2)
        self.mu = nn.Sequential(IcoConvS2S(256, 512, 2, True, R - 2, mode), nn.BatchNorm2d(512))
        self.logvar = nn.Sequential(IcoConvS2S(256, 512, 2, True, R - 2, mode), nn.BatchNorm2d(512))
        self.mu_hook, self.logvar_hook, self.reparameterize_hook = nn.Identity(), nn.Identity(), nn.Identity()
        self.decoder, self.final_layer = decoder(mode, R, 512)

    def encode(self, x):
        h = self.encoder(x)
        return self.mu(h), self.logvar(h)

    def decode(self, z):
        return self.final_layer(self.decoder(z))

    def forward(self, x, eps=None):
        mu, logvar = self.encode(x)
        std = torch.exp(0.5 * logvar)                     # models.py:89-92
        eps = torch.randn_like(std) if eps is None else eps
        return self.decode(eps * std + mu), mu, logvar
