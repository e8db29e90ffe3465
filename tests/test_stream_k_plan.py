"""The stream-K schedule of the persistent conv GEMM (csrc/icn_streamk.h, shared by the kernel and by icn_table_stream_k),
checked on the CPU over many launch shapes: every k-chunk of every tile is computed exactly once, and the hand-off between
workgroups has the properties the kernel's wait logic relies on."""
import itertools

import numpy as np
import pytest

from geniconet_amd import _lib

# tiles of real launches (I5 / batch 36: 1440, 2880, 5760, 360, 720; I6 / batch 8; ragged counts) x grids x k-chunks per tile
NTILES = [8, 9, 63, 360, 361, 720, 767, 768, 769, 1000, 1439, 1440, 1441, 1536, 2520, 2880, 5040, 5760, 10080, 20479]
GRIDS = [512, 768]
NK = [(1, 1), (2, 1), (4, 1), (7, 1), (8, 1), (16, 1), (4, 4), (28, 4), (56, 4), (112, 4)]


def _walks(ntiles, grid, nk, ku):
    t = _lib.table_stream_k(ntiles, grid, nk, ku)
    by_block = {}
    for b, tile, k0, k1 in t.tolist():
        by_block.setdefault(b, []).append((tile, k0, k1))
    return t, by_block


@pytest.mark.parametrize('grid', GRIDS)
@pytest.mark.parametrize('nk,ku', NK)
def test_every_k_chunk_of_every_tile_is_computed_exactly_once(grid, nk, ku):
    for ntiles in NTILES:
        t, _ = _walks(ntiles, grid, nk, ku)
        assert t[:, 1].min() >= 0 and t[:, 1].max() < ntiles
        assert (t[:, 2] % ku == 0).all() and (t[:, 3] % ku == 0).all() and (t[:, 3] > t[:, 2]).all() and (t[:, 3] <= nk).all()
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
    for the same partner; a piece never runs fewer than `ku` k-chunks (>= 4 K-steps)."""
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
            assert blocks == list(range(blocks[0], blocks[0] + 8 * len(blocks), 8)), (tile, ps)   # consecutive ranks, k order
            assert ps[0][0] == 0 and ps[-1][1] == nk and all(a[1] == b[0] for a, b in zip(ps, ps[1:]))
            assert len(ps) <= 6


def test_work_is_balanced_to_within_one_unit_when_a_launch_is_split():
    for ntiles, grid, nk in [(1440, 512, 4), (1440, 512, 8), (360, 768, 8), (2880, 512, 2), (5760, 768, 2), (1441, 512, 8)]:
        t, by_block = _walks(ntiles, grid, nk, 1)
        if (t[:, 3] - t[:, 2] == nk).all():
            continue                                            # whole rounds: nothing was split
        work = np.array([sum(k1 - k0 for _, k0, k1 in by_block.get(b, [])) for b in range(grid)])
        per_class = [work[x::8] for x in range(8)]
        for w in per_class:
            assert w.max() - w.min() <= 1 + (nk if ntiles % 8 else 0), (ntiles, grid, nk, w.min(), w.max())
        assert work.sum() == ntiles * nk


def test_whole_rounds_and_tiny_launches_are_left_alone():
    t, _ = _walks(1536, 512, 8, 1)                              # exactly 3 rounds
    assert (t[:, 2] == 0).all() and (t[:, 3] == 8).all()
    t, _ = _walks(40, 768, 1, 1)                                # fewer tiles than workgroups and nothing worth cutting
    assert (t[:, 2] == 0).all() and (t[:, 3] == 1).all() and len(t) == 40
    assert _lib.lib().icn_table_stream_k(100, 100, 4, 1, None, 0) < 0      # grid must be a multiple of 8
    assert _lib.lib().icn_table_stream_k(100, 512, 6, 4, None, 0) < 0      # nk a multiple of ku
