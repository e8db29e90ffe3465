"""The stream-K schedule of the persistent conv GEMM (csrc/icn_streamk.h, shared by the kernel and by icn_table_stream_k),
checked on the CPU over many launch shapes: every K-step of every tile is computed exactly once, no piece is shorter than the
minimum the kernel's pipeline needs, the work is balanced to a few steps, and the hand-off between workgroups has the
properties the kernel's wait logic relies on.  Ranges are K-steps since round 3 (ABI 4); they were whole
k-chunks of 7 steps before, which left every fourth workgroup of a typical launch with one unit (14 us) more than the rest."""
import itertools

import numpy as np
import pytest

from geniconet_amd import _lib

# tiles of real launches (I5 / batch 36: 1440, 2880, 5760, 360, 720; I6 / batch 8; ragged counts) x grids x
# (K-steps per tile, shortest piece): 7 taps x Cin / 32 = 14 ... 112, the one-tap dense GEMMs 8 ... 56, degenerate small ones
NTILES = [8, 9, 63, 360, 361, 720, 767, 768, 769, 1000, 1439, 1440, 1441, 1536, 2520, 2880, 5040, 5760, 10080, 20479]
GRIDS = [512, 768]
NK = [(1, 1), (2, 1), (3, 4), (7, 4), (8, 4), (14, 4), (16, 4), (28, 4), (56, 4), (112, 4)]


def _walks(ntiles, grid, nk, ku):
    t = _lib.table_stream_k(ntiles, grid, nk, ku)
    by_block = {}
    for b, tile, k0, k1 in t.tolist():
        by_block.setdefault(b, []).append((tile, k0, k1))
    return t, by_block


@pytest.mark.parametrize('grid', GRIDS)
@pytest.mark.parametrize('nk,ku', NK)
def test_every_k_step_of_every_tile_is_computed_exactly_once(grid, nk, ku):
    for ntiles in NTILES:
        t, _ = _walks(ntiles, grid, nk, ku)
        assert t[:, 1].min() >= 0 and t[:, 1].max() < ntiles
        assert (t[:, 2] >= 0).all() and (t[:, 3] > t[:, 2]).all() and (t[:, 3] <= nk).all()
        assert (t[:, 3] - t[:, 2] >= min(ku, nk)).all(), (ntiles, grid, nk, ku)     # no piece below the pipeline's minimum
        cover = np.zeros((ntiles, nk), dtype=np.int32)
        for _, tile, k0, k1 in t.tolist():
            cover[tile, k0:k1] += 1
        assert (cover == 1).all(), (ntiles, grid, nk, ku)
        assert (t[:, 1] % 8 == t[:, 0] % 8).all()              # a workgroup stays in its XCD residue class


@pytest.mark.parametrize('grid', GRIDS)
@pytest.mark.parametrize('nk,ku', NK)
def test_hand_off_goes_to_higher_workgroups_and_parked_pieces_come_first(grid, nk, ku):
    """What the kernel's wait logic assumes: a workgroup parks at most one piece, and that piece is the first segment of its
    split phase (right after its whole tiles); the pieces of a split tile lie in consecutive workgroups of one residue class
    in k order, so the finisher (last k-chunks) has the highest id and only ever waits for lower ids; nobody waits twice
    for the same partner; a piece never runs fewer than `ku` K-steps."""
    for ntiles in NTILES:
        t, by_block = _walks(ntiles, grid, nk, ku)
        pieces = {}
        for b, segs in by_block.items():
            parked = [i for i, (_, k0, k1) in enumerate(segs) if k1 < nk]
            assert len(parked) <= 1
            if parked:
                i = parked[0]
                assert all(k0 == 0 and k1 == nk for _, k0, k1 in segs[:i])          # only whole tiles before it ...
                assert all(k1 == nk for _, k0, k1 in segs[i + 1:])                  # ... and nothing parked after it
            for tile, k0, k1 in segs:
                if (k0, k1) != (0, nk):
                    pieces.setdefault(tile, []).append((k0, k1, b))
        for tile, ps in pieces.items():
            ps.sort()
            blocks = [b for _, _, b in ps]
            assert all(b % 8 == tile % 8 for b in blocks)
            # ascending ranks in k order; a rank in between may only be skipped if that workgroup's range is empty (degenerate
            # shapes: the kernel's finisher skips empty ranges)
            assert blocks == sorted(blocks) and len(set(blocks)) == len(blocks), (tile, ps)
            for skipped in set(range(blocks[0], blocks[-1], 8)) - set(blocks):
                assert all(k0 == 0 and k1 == nk for _, k0, k1 in by_block.get(skipped, [])), (tile, ps, skipped)
            assert ps[0][0] == 0 and ps[-1][1] == nk and all(a[1] == b[0] for a, b in zip(ps, ps[1:]))
            assert len(ps) <= 6
            assert all(k1 - k0 >= min(ku, nk) for k0, k1, _ in ps)


def test_work_is_balanced_to_a_few_steps_when_a_launch_is_split():
    """Division in K-steps: among the workgroups of one residue class that arrive k-th on their CU (ids [k, k + 1) * grid / occ:
    the plan gives earlier arrivals a few per cent more, icn_streamk.h) the shares differ by at most 1 step from the division
    itself plus 2 x (ku - 1) from boundaries kept off the zone next to a tile edge.  With whole k-chunks as units (rounds 1-2)
    the same launches differed by 7 steps.  Earlier arrivals never get less than later ones."""
    for ntiles, grid, nk in [(1440, 512, 28), (1440, 512, 56), (360, 768, 56), (2880, 512, 14), (5760, 768, 14), (1448, 512, 56),
                             (720, 512, 56), (1440, 512, 16)]:
        t, by_block = _walks(ntiles, grid, nk, 4)
        if (t[:, 3] - t[:, 2] == nk).all():
            continue                                            # whole rounds: nothing was split
        work = np.array([sum(k1 - k0 for _, k0, k1 in by_block.get(b, [])) for b in range(grid)])
        assert work.sum() == ntiles * nk
        occ, per_slot = grid // 256, grid // (grid // 256)
        means = []
        for k in range(occ):
            w = work[k * per_slot:(k + 1) * per_slot]
            for x in range(8):
                assert w[x::8].max() - w[x::8].min() <= 1 + 2 * 3, (ntiles, grid, nk, k, x, w[x::8].min(), w[x::8].max())
            means.append(w.mean())
        assert all(a >= b for a, b in zip(means, means[1:])), means
        assert means[0] <= 1.45 * means[-1], means


def test_whole_rounds_and_tiny_launches_are_left_alone():
    t, _ = _walks(1536, 512, 56, 4)                             # exactly 3 rounds
    assert (t[:, 2] == 0).all() and (t[:, 3] == 56).all()
    t, _ = _walks(40, 768, 1, 1)                                # fewer tiles than workgroups and nothing worth cutting
    assert (t[:, 2] == 0).all() and (t[:, 3] == 1).all() and len(t) == 40
    assert _lib.lib().icn_table_stream_k(100, 100, 4, 1, None, 0) < 0      # grid must be a multiple of 8
    assert _lib.lib().icn_table_stream_k(100, 512, 0, 4, None, 0) < 0      # nk, ku positive


# ---- tile lists of masked launches (stride-2 data gradients) ------------------------------------------------------------------
@pytest.mark.parametrize('r,B,bm,ntn,grid', [(4, 36, 64, 2, 768), (5, 36, 64, 1, 768), (3, 36, 64, 4, 768), (4, 36, 64, 1, 512),
                                             (2, 3, 64, 4, 768), (5, 8, 64, 2, 768), (4, 36, 128, 2, 512)])
def test_tile_lists_cover_every_tile_once_and_even_out_the_step_totals(r, B, bm, ntn, grid):
    """icn_table_tile_lists: every tile in exactly one workgroup's list, lists ascending, residue class kept (tile % 8 =
    workgroup % 8: the XCD / L2 association of the round-robin walk), and -- the point -- the workgroups' K-step totals, which
    differ by up to 2 : 1 per tile, are even within each arrival slot's share (the k-th arrivals on a CU get the plan's speed
    factor, icn_streamk.h), where the round-robin deal b, b + G, ... is off by tens of per cent."""
    off, ids = _lib.table_tile_lists(r, B, bm, ntn, grid)
    nt = len(ids)
    assert off[0] == 0 and off[-1] == nt and (np.diff(off) >= 0).all()
    assert sorted(ids.tolist()) == list(range(nt))
    cnt = np.diff(off)
    for b in range(grid):
        mine = ids[off[b]:off[b + 1]]
        assert (mine % 8 == b % 8).all() and (np.diff(mine) > 0).all()
    if nt >= 4 * grid:
        occ, per_slot = grid // 256, 256
        for k in range(occ):
            c = cnt[k * per_slot:(k + 1) * per_slot]
            assert c.max() - c.min() <= max(2, c.mean() * 0.6), (k, c.min(), c.max())   # tile counts differ (1- and 2-tap tiles) ...
        assert cnt[:256].mean() >= cnt[-256:].mean()                                    # ... and earlier arrivals never get less
