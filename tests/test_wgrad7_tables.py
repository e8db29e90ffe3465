"""Patch tables of the all-taps weight-gradient kernel k_wgrad7 (csrc/icn_geometry.cpp: build_wgrad7), on the CPU:
the kernel reads tap t of output pixel p from row pos[p][t] of the staged union of its patch, so for every (patch, pixel, tap)
that row's code must be exactly the forward table's entry -- pole means included, 'nothing' as the guaranteed zero row."""
import numpy as np
import pytest

from geniconet_amd import _lib


def _dma_code(fwd):
    """Forward table -> DmaTable codes: pixels stay, IDX_POLE - k (-2, -3) become side-buffer slots in order of appearance
    (tap-major, pixel-minor: build_dma_table)."""
    code = fwd.copy()
    slot_of = {}
    for t in range(7):
        for p in np.nonzero(fwd[t] <= -2)[0]:
            v = int(fwd[t, p])
            code[t, p] = -2 - slot_of.setdefault(v, len(slot_of))
    return code


@pytest.mark.parametrize('mode', ['average', 'zeros'])
@pytest.mark.parametrize('r,stride', [(2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (3, 2), (4, 2), (5, 2), (6, 2)])
def test_union_rows_and_positions_reproduce_the_forward_table(r, stride, mode):
    tab = _lib.table_wgrad7(r, stride, mode)
    assert tab is not None
    rows, pos = tab
    fwd = _lib.table_conv_fwd(r, stride, mode)
    P = fwd.shape[1]
    U = rows.shape[1]
    assert U == (64 if stride == 1 else 112) and rows.shape[0] == P // 16 and pos.shape == (P // 16, 16, 8)
    assert (rows[:, U - 1] == -1).all()                                 # the zero row
    assert (pos % 256 == 0).all() and int(pos.max()) // 256 < U
    code = _dma_code(fwd)                                               # [7][P]
    got = rows[np.arange(P // 16)[:, None, None], pos[:, :, :7] // 256]  # [npatch][16][7]
    want = code.reshape(7, P // 16, 16).transpose(1, 2, 0)
    assert np.array_equal(got, want)
    # the union is what makes the kernel cheaper at stride 1: <= 53 distinct rows for 112 (pixel, tap) pairs (stride 2: the taps
    # of neighbouring outputs share only the odd columns of their own row: <= 97); each row listed once, pixels ascending
    used = (rows != -1).sum(1)
    assert used.max() <= (56 if stride == 1 else 100) and used.min() >= 16
    for q in (0, P // 32, P // 16 - 1):
        live = rows[q][rows[q] != -1]
        assert len(set(live.tolist())) == len(live)
        pix = live[live >= 0]
        assert (np.diff(pix) > 0).all()


def test_no_table_where_the_output_grid_is_not_made_of_whole_patches():
    """P_out % 16 != 0 (levels 0 and 1; stride 2 from level 2): the launch keeps the per-tap kernel."""
    assert _lib.table_wgrad7(0, 1, 'average') is None
    assert _lib.table_wgrad7(1, 1, 'average') is None
    assert _lib.table_wgrad7(2, 2, 'average') is None


def test_the_union_row_knob_is_one_definition_and_rejects_what_has_no_kernel():
    """ICN_W7_U (developer A/B knob, read once per process) selects the stride-1 table the kernel is launched with AND
    the one icn_table_wgrad7 shows; a value k_wgrad7 has no instantiation for is an error, not a silent fall-back."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from geniconet_amd import _lib\n"
            "t = _lib.table_wgrad7(4, 1, 'average'); print('U', t[0].shape[1], _lib.table_wgrad7(4, 2, 'average')[0].shape[1])\n" % root)
    for u, want in (('56', 'U 56 112'), ('64', 'U 64 112'), ('48', None), ('112', None)):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, ICN_W7_U=u), capture_output=True, text=True, timeout=300)
        if want:
            assert r.returncode == 0 and want in r.stdout, (u, r.stdout, r.stderr[-500:])
        else:
            assert r.returncode != 0 and 'ICN_W7_U must be 56 or 64' in r.stderr, (u, r.stderr[-500:])
