"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the icosahedral operators.

PARITY UNPINNED vs upstream: the reference (hrdkjain/GenIcoNet) imports these operators from the
un-vendored, un-pinned sibling repo hrdkjain/IcosahedralCNN (reference models.py:4-6, README.md:18-24),
which is absent from /root/reference and cannot be fetched.  This file restates the operator the
reference's call sites describe ("5 charts of (2^r x 2^(r+1)) pixels stacked on rows, hex 1-ring conv
with inter-chart padding, stride 1/2, x2 icosahedral upsample") under the chart convention of
SURVEY.md App. A, and is pinned only by implementation-independent geometric known-answers
(tests/test_geometry_known_answers.py) -- never by upstream outputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What pins the interface (all citations into /root/reference):
  * constructor signatures / `subdivisions` semantics ........ models.py:9-62
  * tensor layout (B, C, 5*2^r, 2^(r+1)), charts on rows .... data.py:64-69, app.py:1506-1515
  * vertex order + pole neighbours ......................... ico_utils.py:10-24, losses.py:23-31,47-51
  * corner_mode in {'zeros','average'} ...................... models.py:11, run.py:683

Method (deliberately different from the product's table-driven HIP kernels): explicit pad-exchange
of every chart to (n+2, 2n+2) by slice copies (App. A.3), then a dense 3x3 `conv2d` whose (-1,-1) and
(+1,+1) corners are structurally zero.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# Tap order of the 7-vector (unpinned upstream; fixed here and in the product):
# t=0 centre, then the six hex neighbours counter-clockwise in lattice (a=row, b=col) coordinates.
TAPS = ((0, 0), (1, 0), (0, 1), (-1, 1), (-1, 0), (0, -1), (1, -1))


def pole_values(x5, corner_mode):
    """x5: (B, C, 5, n, 2n).  Returns (north, south), each (B, C): reference pole rule
    losses.py:49-51 / ico_utils.py:13-23 (mean of px[c*n, 0] resp. px[(c+1)*n-1, 2n-1])."""
    if corner_mode == 'average':
        return x5[:, :, :, 0, 0].mean(-1), x5[:, :, :, -1, -1].mean(-1)
    if corner_mode == 'zeros':
        z = x5.new_zeros(x5.shape[:2])
        return z, z
    raise ValueError('corner_mode must be zeros|average, got %r' % (corner_mode,))


def pad_charts(x, r, corner_mode):
    """Pad-exchange.  x: (B, C, 5n, 2n) -> (B, C, 5, n+2, 2n+2).
    Padded index (I, J) = (i+1, j+1); entries (0,0) and (n+1, 2n+1) are never read by the stencil."""
    n = 2 ** r
    B, C, H, W = x.shape
    assert H == 5 * n and W == 2 * n, (x.shape, r)
    x5 = x.reshape(B, C, 5, n, 2 * n)
    nxt = torch.roll(x5, -1, dims=2)   # nxt[c] = chart c+1
    prv = torch.roll(x5, 1, dims=2)    # prv[c] = chart c-1
    north, south = pole_values(x5, corner_mode)
    xp = x.new_zeros(B, C, 5, n + 2, 2 * n + 2)
    xp[..., 1:n + 1, 1:2 * n + 1] = x5
    # left column j=-1: i=0 -> N pole; i in [1,n] -> px(c+1; 0, i-1)
    xp[..., 1, 0] = north[:, :, None]
    xp[..., 2:n + 2, 0] = nxt[..., 0, 0:n]
    # bottom row i=n: j in [-1,n-1] -> px(c+1; 0, j+n); j in [n-1,2n-2] -> px(c+1; j+1-n, 2n-1); j=2n-1 -> S
    xp[..., n + 1, 0:n + 1] = nxt[..., 0, n - 1:2 * n]
    xp[..., n + 1, n:2 * n] = nxt[..., 0:n, 2 * n - 1]
    xp[..., n + 1, 2 * n] = south[:, :, None]
    # top row i=-1: j in [0,n-1] -> px(c-1; j, 0); j in [n,2n-1] -> px(c-1; n-1, j-n); j=2n -> px(c-1; n-1, n)
    xp[..., 0, 1:n + 1] = prv[..., 0:n, 0]
    xp[..., 0, n + 1:2 * n + 2] = prv[..., n - 1, 0:n + 1]
    # right column j=2n: i in [0,n-1] -> px(c-1; n-1, i+n)
    xp[..., 1:n + 1, 2 * n + 1] = prv[..., n - 1, n:2 * n]
    return xp


def hex_kernel(weight):
    """(Cout, Cin, 7) -> (Cout, Cin, 3, 3) with the (-1,-1) / (+1,+1) corners zero."""
    k = weight.new_zeros(weight.shape[0], weight.shape[1], 3, 3)
    for t, (di, dj) in enumerate(TAPS):
        k[:, :, di + 1, dj + 1] = weight[:, :, t]
    return k


def ico_conv(x, weight, bias, r, stride, corner_mode):
    """7-tap hex conv over the 5 charts.  x (B,Cin,5n,2n) at level r -> (B,Cout,5n',2n'), n'=n/stride.
    Stride 2 evaluates the stride-1 result at fine sites (2i, 2j+1) (App. A.4)."""
    n = 2 ** r
    B = x.shape[0]
    xp = pad_charts(x, r, corner_mode)                                  # (B,C,5,n+2,2n+2)
    xp = xp.permute(0, 2, 1, 3, 4).reshape(B * 5, x.shape[1], n + 2, 2 * n + 2)
    y = F.conv2d(xp, hex_kernel(weight), bias)                          # (B*5,Cout,n,2n)
    if stride == 2:
        y = y[:, :, 0::2, 1::2]
    elif stride != 1:
        raise ValueError('stride must be 1 or 2')
    Cout = weight.shape[0]
    y = y.reshape(B, 5, Cout, y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)
    return y.reshape(B, Cout, 5 * y.shape[3], y.shape[4])


def ico_upsample(x, r, corner_mode):
    """r -> r+1.  Coarse px (i,j) is fine px (2i, 2j+1); every other fine vertex is the midpoint of
    exactly one coarse edge and takes the mean of its two endpoints (App. A.4)."""
    n = 2 ** r
    B, C = x.shape[:2]
    xp = pad_charts(x, r, corner_mode)                                  # coarse (I,J) = (i+1, j+1)
    y = x.new_zeros(B, C, 5, 2 * n, 4 * n)
    c = xp[..., 1:n + 1, 1:2 * n + 1]
    y[..., 0::2, 1::2] = c
    y[..., 0::2, 0::2] = 0.5 * (c + xp[..., 1:n + 1, 0:2 * n])          # (2i,2j):   (i,j) & (i,j-1)
    y[..., 1::2, 1::2] = 0.5 * (c + xp[..., 2:n + 2, 1:2 * n + 1])      # (2i+1,2j+1): (i,j) & (i+1,j)
    y[..., 1::2, 0::2] = 0.5 * (c + xp[..., 2:n + 2, 0:2 * n])          # (2i+1,2j): (i,j) & (i+1,j-1)
    return y.reshape(B, C, 10 * n, 4 * n)


class IcoConvS2S(torch.nn.Module):
    """CPU oracle of icocnn.ico_conv.IcoConvS2S (call sites models.py:14,25-33,104-109)."""

    def __init__(self, in_features, out_features, stride=1, bias=True, subdivisions=0,
                 corner_mode='zeros'):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.stride, self.subdivisions, self.corner_mode = stride, subdivisions, corner_mode
        self.weight = torch.nn.Parameter(torch.empty(out_features, in_features, 7))
        self.bias = torch.nn.Parameter(torch.empty(out_features)) if bias else None
        bound = 1.0 / math.sqrt(7 * in_features)   # kaiming_uniform(a=sqrt(5)) on fan_in = 7*Cin
        torch.nn.init.uniform_(self.weight, -bound, bound)
        if bias:
            torch.nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return ico_conv(x, self.weight, self.bias, self.subdivisions, self.stride, self.corner_mode)


class IcoUpsampleS2S(torch.nn.Module):
    """CPU oracle of icocnn.ico_conv.IcoUpsampleS2S (call sites models.py:13,45,53)."""

    def __init__(self, in_features, subdivisions=0, corner_mode='zeros'):
        super().__init__()
        self.in_features, self.subdivisions, self.corner_mode = in_features, subdivisions, corner_mode

    def forward(self, x):
        return ico_upsample(x, self.subdivisions, self.corner_mode)


# ----------------------------------------------------------------------------------------------
# Index tables derived from the SAME pad_charts code (by padding an image of pixel ids).  Used by the
# tests to cross-check the product's independently built C++ tables.
# ----------------------------------------------------------------------------------------------
def id_image(r):
    n = 2 ** r
    return torch.arange(10 * n * n, dtype=torch.float64).reshape(1, 1, 5 * n, 2 * n)


def _pad_ids(r):
    """Padded id image (5, n+2, 2n+2) int64; poles are P and P+1; unused corners -1."""
    n = 2 ** r
    P = 10 * n * n
    ids = id_image(r)
    x5 = ids.reshape(1, 1, 5, n, 2 * n)
    nxt, prv = torch.roll(x5, -1, dims=2), torch.roll(x5, 1, dims=2)
    xp = torch.full((1, 1, 5, n + 2, 2 * n + 2), -1.0, dtype=torch.float64)
    xp[..., 1:n + 1, 1:2 * n + 1] = x5
    xp[..., 1, 0] = P
    xp[..., 2:n + 2, 0] = nxt[..., 0, 0:n]
    xp[..., n + 1, 0:n + 1] = nxt[..., 0, n - 1:2 * n]
    xp[..., n + 1, n:2 * n] = nxt[..., 0:n, 2 * n - 1]
    xp[..., n + 1, 2 * n] = P + 1
    xp[..., 0, 1:n + 1] = prv[..., 0:n, 0]
    xp[..., 0, n + 1:2 * n + 2] = prv[..., n - 1, 0:n + 1]
    xp[..., 1:n + 1, 2 * n + 1] = prv[..., n - 1, n:2 * n]
    return xp[0, 0].to(torch.int64)


def tap_table(r, stride=1):
    """(7, P_out) int64: id of the level-r pixel (or P / P+1 for the N / S pole) that tap t of output
    pixel p reads.  Output pixel order is row-major over the (5n', 2n') grid."""
    n = 2 ** r
    xp = _pad_ids(r)
    out = []
    for (di, dj) in TAPS:
        w = xp[:, 1 + di:1 + di + n, 1 + dj:1 + dj + 2 * n]            # (5, n, 2n)
        if stride == 2:
            w = w[:, 0::2, 1::2]
        out.append(w.reshape(-1))
    return torch.stack(out).numpy()


def upsample_table(r):
    """(2, P_fine) int64: the two coarse ids (level r; poles P, P+1) averaged into each level-(r+1) pixel
    (both equal at coarse sites)."""
    n = 2 ** r
    xp = _pad_ids(r)
    c = xp[:, 1:n + 1, 1:2 * n + 1]
    a = torch.zeros(5, 2 * n, 4 * n, dtype=torch.int64)
    b = torch.zeros_like(a)
    a[:, 0::2, 1::2], b[:, 0::2, 1::2] = c, c
    a[:, 0::2, 0::2], b[:, 0::2, 0::2] = c, xp[:, 1:n + 1, 0:2 * n]
    a[:, 1::2, 1::2], b[:, 1::2, 1::2] = c, xp[:, 2:n + 2, 1:2 * n + 1]
    a[:, 1::2, 0::2], b[:, 1::2, 0::2] = c, xp[:, 2:n + 2, 0:2 * n]
    return torch.stack([a.reshape(-1), b.reshape(-1)]).numpy()


def faces_from_lattice(r):
    """(20*4^r, 3) int64 faces in the reference vertex order (row-major grid, then N, S;
    ico_utils.py:20-23), built from the seam identifications of App. A.2 only (no padding code)."""
    n = 2 ** r
    P = 10 * n * n

    def owner(c, a, b):
        # canonical vertex id of lattice point (a,b), a in [0,n], b in [0,2n], of chart c
        while True:
            if a == 0 and b == 0:
                return P
            if a == n and b == 2 * n:
                return P + 1
            if b == 0:                       # (a,0)_c == (0,a)_{c+1}
                c, a, b = (c + 1) % 5, 0, a
            elif a == n:
                if b <= n:                   # (n,b)_c == (0,b+n)_{c+1}
                    c, a, b = (c + 1) % 5, 0, b + n
                else:                        # (n,b)_c == (b-n,2n)_{c+1}
                    c, a, b = (c + 1) % 5, b - n, 2 * n
            else:
                return (c * n + a) * 2 * n + (b - 1)

    f = []
    for c in range(5):
        for a in range(n):
            for b in range(2 * n):
                f.append((owner(c, a, b), owner(c, a + 1, b), owner(c, a, b + 1)))
                f.append((owner(c, a + 1, b), owner(c, a + 1, b + 1), owner(c, a, b + 1)))
    return np.asarray(f, dtype=np.int64)
