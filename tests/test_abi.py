"""The C-ABI shared library loads without a GPU and exports every symbol include/icn.h declares."""
import ctypes
import os
import re

from geniconet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'icn.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(icn_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 15
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), 'libicn.so does not export %s' % n
    assert sorted(_lib.SIGNATURES) == names, 'ctypes binding and include/icn.h disagree'


def test_abi_version_and_error_channel():
    L = _lib.lib()
    assert L.icn_abi_version() == _lib.ABI_VERSION
    assert L.icn_table_conv_fwd(99, 1, 1, None, 0) < 0
    assert b'subdivisions' in L.icn_last_error()


def test_workspace_query_is_host_only():
    L = _lib.lib()
    # AE I5/B36 largest layers: packed weights + the side buffer of the two pole means per mesh (fwd) and split-K slabs
    # (bwd-weight) are bounded
    # ... + the stream-K scratch (flag words and 512 partial 64x128 tiles, include/icn.h)
    sk = (1032 * 4 + 255) // 256 * 256 + 512 * 64 * 128 * 4
    # packed weights: the fp32 operand AND its bf16 x 3 image (6 bytes per weight, ABI 7) are both reserved, whatever the arithmetic mode
    assert L.icn_conv_workspace_bytes(_lib.OP_CONV_FWD, 36, 128, 64, 5, 1) == 7 * 128 * 64 * (4 + 6) + 36 * 2 * 128 * 4 + sk
    assert L.icn_conv_workspace_bytes(_lib.OP_CONV_FWD, 36, 3, 64, 5, 1) == 0
    assert 0 < L.icn_conv_workspace_bytes(_lib.OP_CONV_BWD_WEIGHT, 36, 256, 256, 3, 1) < 256 << 20
    assert L.icn_conv_workspace_bytes(7, 36, 256, 256, 3, 1) == 0


def test_adam_step_rejects_bad_arguments_before_touching_a_device():
    """icn_adam_step validates its host-side arguments first (include/icn.h); nothing here reaches a kernel launch."""
    import ctypes
    L = _lib.lib()
    one = (ctypes.c_void_p * 1)(0x1000)
    n1 = (ctypes.c_size_t * 1)(4)
    f1 = (ctypes.c_float * 1)(1e-3)
    assert L.icn_adam_step(0, None, None, None, None, None, None, None, 0.9, 0.999, 1e-8, 0.0, None) == 0   # nothing to do
    assert L.icn_adam_step(-1, one, one, one, one, n1, f1, f1, 0.9, 0.999, 1e-8, 0.0, None) != 0
    assert L.icn_adam_step(1, None, one, one, one, n1, f1, f1, 0.9, 0.999, 1e-8, 0.0, None) != 0
    null = (ctypes.c_void_p * 1)(None)
    assert L.icn_adam_step(1, one, null, one, one, n1, f1, f1, 0.9, 0.999, 1e-8, 0.0, None) != 0
    assert b'null tensor' in L.icn_last_error()
    assert L.icn_adam_step(1, one, one, one, one, n1, f1, f1, 1.0, 0.999, 1e-8, 0.0, None) != 0            # beta1 must be < 1
    assert L.icn_adam_step(1, one, one, one, one, n1, f1, f1, 0.9, 0.999, -1.0, 0.0, None) != 0            # eps >= 0
    assert b'betas' in L.icn_last_error()



def test_arithmetic_mode_and_build_flags_are_host_only():
    """ABI 7: icn_get_arith / icn_set_arith / icn_build_flags need no device; a bad mode is an error with a message."""
    L = _lib.lib()
    prev = L.icn_get_arith()
    assert prev in (0, 1)
    assert L.icn_set_arith(1) == prev and L.icn_get_arith() == 1
    assert L.icn_set_arith(0) == 1 and L.icn_get_arith() == 0
    assert L.icn_set_arith(2) == -1 and b'arithmetic mode' in L.icn_last_error() and L.icn_get_arith() == 0
    L.icn_set_arith(prev)
    assert L.icn_build_flags() & 0xffff == 0                       # the in-tree library is the product, not a pricing build
    import ctypes
    one, n1 = (ctypes.c_void_p * 1)(0x1000), (ctypes.c_size_t * 1)(4)
    assert L.icn_adam_step_dev(0, None, None, None, None, None, 0x1000, 0.9, 0.999, 1e-8, 0.0, None) == 0
    assert L.icn_adam_step_dev(1, one, one, one, one, n1, None, 0.9, 0.999, 1e-8, 0.0, None) != 0       # no scalar buffer
