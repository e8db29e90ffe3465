"""ctypes binding of libicn.so (C ABI declared in include/icn.h).

The library is the product: there is NO Python/CPU fallback.  If it is missing, loading fails loudly with
the build command; if a call fails, the library's own message is raised as RuntimeError.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ICN_LIB_PATH: developer override (tools/asan_host.sh runs the host-side tests against a sanitizer build of the library)
LIB_PATH = os.environ.get('ICN_LIB_PATH') or os.path.join(_HERE, 'libicn.so')

OP_CONV_FWD, OP_CONV_BWD_DATA, OP_CONV_BWD_WEIGHT = 0, 1, 2
CORNER_MODES = {'zeros': 0, 'average': 1}
ABI_VERSION = 7
LAP_MODES = {'mean-v': 0, 'v-mean': 1, 'sum-kv': 2, 'kv-sum': 3}   # ICN_LAP_* of include/icn.h

_c_float_p = ctypes.c_void_p      # device pointers travel as plain addresses
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_intp = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); mirrors include/icn.h one to one (checked by tests/test_abi.py)
SIGNATURES = {
    'icn_abi_version': (ctypes.c_int, []),
    'icn_last_error': (ctypes.c_char_p, []),
    'icn_prepare_conv': (ctypes.c_int, [ctypes.c_int] * 3),
    'icn_prepare_upsample': (ctypes.c_int, [ctypes.c_int] * 2),
    'icn_conv_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 6),
    'icn_conv_fwd': (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'icn_conv_bwd_data': (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'icn_conv_bwd_weight': (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'icn_conv_pair_supported': (ctypes.c_int, [ctypes.c_int] * 6),
    'icn_conv_pair_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 7),
    'icn_conv_pair_fwd': (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'icn_conv_pair_bwd_data': (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'icn_conv_pair_bwd_weight': (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'icn_upconv_supported': (ctypes.c_int, [ctypes.c_int] * 5),
    'icn_upconv_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 5),
    'icn_upconv_fwd': (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'icn_upconv_bwd_supported': (ctypes.c_int, [ctypes.c_int] * 6),
    'icn_upconv_bwd_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 5),
    'icn_upconv_bwd': (ctypes.c_int, [_c_float_p] * 10 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'icn_upconv_bwd_streams': (ctypes.c_int, [_c_float_p] * 10 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    'icn_upsample_fwd': (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    'icn_upsample_bwd': (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    'icn_bn_workspace_floats': (ctypes.c_size_t, [ctypes.c_int] * 2),
    'icn_bn_stats2': (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int] + ([ctypes.c_float] * 2 + [_c_float_p] * 3) * 2 + [_c_float_p, ctypes.c_void_p]),
    'icn_bn_stats': (ctypes.c_int, [_c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float] + [_c_float_p] * 4 + [ctypes.c_void_p]),
    'icn_bn_relu_fwd': (ctypes.c_int, [_c_float_p] * 9 + [ctypes.c_int] * 2 + [ctypes.c_void_p]),
    'icn_bn_relu_bwd': (ctypes.c_int, [_c_float_p] * 13 + [ctypes.c_int] * 2 + [_c_float_p] * 4 + [ctypes.c_void_p]),
    'icn_head_workspace_floats': (ctypes.c_size_t, [ctypes.c_int] * 2),
    'icn_head_fwd': (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    'icn_head_bwd': (ctypes.c_int, [_c_float_p] * 8 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    'icn_p2p_loss_workspace_floats': (ctypes.c_size_t, [ctypes.c_int] * 2),
    'icn_p2p_loss_fwd': (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_int] * 2 + [ctypes.c_float] * 3 + [ctypes.c_int] + [_c_float_p] * 2 + [ctypes.c_void_p]),
    'icn_p2p_loss_bwd_workspace_floats': (ctypes.c_size_t, [ctypes.c_int] * 2),
    'icn_p2p_loss_bwd': (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 2 + [ctypes.c_float] * 3 + [ctypes.c_int] + [_c_float_p] * 2 + [ctypes.c_void_p]),
    'icn_kld_workspace_floats': (ctypes.c_size_t, [ctypes.c_size_t]),
    'icn_kld_fwd': (ctypes.c_int, [_c_float_p] * 2 + [ctypes.c_size_t] + [_c_float_p] * 2 + [ctypes.c_void_p]),
    'icn_kld_bwd': (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_size_t] + [_c_float_p] * 2 + [ctypes.c_void_p]),
    'icn_reparam_fwd': (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_size_t] + [_c_float_p] + [ctypes.c_void_p]),
    'icn_reparam_bwd': (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_size_t] + [_c_float_p] * 2 + [ctypes.c_void_p]),
    'icn_adam_step': (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_double] * 4 + [ctypes.c_void_p]),
    'icn_adam_step_dev': (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_double] * 4 + [ctypes.c_void_p]),
    'icn_table_stream_k': (ctypes.c_long, [ctypes.c_int] * 4 + [_i32p, ctypes.c_size_t]),
    'icn_table_tile_lists': (ctypes.c_long, [ctypes.c_int] * 7 + [_i32p, ctypes.c_size_t]),
    'icn_set_debug_flags': (ctypes.c_int, [ctypes.c_int]),
    'icn_device_status': (ctypes.c_int, [ctypes.c_int]),
    'icn_get_arith': (ctypes.c_int, []),
    'icn_set_arith': (ctypes.c_int, [ctypes.c_int]),
    'icn_build_flags': (ctypes.c_uint, []),
    'icn_host_selfcheck': (ctypes.c_long, [ctypes.c_int] * 2),
    'icn_debug_trace': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    'icn_point_to_mesh': (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 4 + [_c_float_p] * 3 + [ctypes.c_void_p]),
    'icn_table_conv_fwd': (ctypes.c_long, [ctypes.c_int] * 3 + [_i32p, ctypes.c_size_t]),
    'icn_table_wgrad7': (ctypes.c_long, [ctypes.c_int] * 3 + [_i32p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint16), ctypes.c_size_t, _intp]),
    'icn_table_conv_bwd': (ctypes.c_long, [ctypes.c_int] * 3 + [_i32p, ctypes.c_size_t, _intp]),
    'icn_table_upsample': (ctypes.c_long, [ctypes.c_int] * 3 + [_i32p, _f32p, ctypes.c_size_t, _intp]),
    'icn_table_upsample_pairs': (ctypes.c_long, [ctypes.c_int, _i32p, ctypes.c_size_t]),
    'icn_table_faces': (ctypes.c_long, [ctypes.c_int, _i32p, ctypes.c_size_t]),
    'icn_table_upconv_bwd': (ctypes.c_long, [ctypes.c_int] * 2 + [_i32p, _f32p, ctypes.c_size_t, _intp]),
    'icn_table_upconv': (ctypes.c_long, [ctypes.c_int, ctypes.c_int, _i32p, ctypes.c_size_t, _f32p, ctypes.c_size_t, _intp]),
}



class ProfileEntry(ctypes.Structure):
    _fields_ = [('kernel', ctypes.c_char_p), ('launches', ctypes.c_long), ('total_ms', ctypes.c_double),
                ('total_flops', ctypes.c_double)]


SIGNATURES['icn_profile_start'] = (ctypes.c_int, [ctypes.c_int])
SIGNATURES['icn_profile_stop'] = (ctypes.c_int, [ctypes.POINTER(ProfileEntry), ctypes.c_int])
SIGNATURES['icn_profile_select'] = (ctypes.c_int, [ctypes.c_char_p])

_lib = None


def lib():
    """Load libicn.so once; raise with the build recipe if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'geniconet_amd: %s is missing -- the HIP extension is the product and has no fallback. '
                'Build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C geniconet_amd/csrc`).' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if handle.icn_abi_version() != ABI_VERSION:
            raise RuntimeError('geniconet_amd: libicn.so ABI %d != binding ABI %d; rebuild'
                               % (handle.icn_abi_version(), ABI_VERSION))
        refuse_pricing_build(handle.icn_build_flags(), LIB_PATH)
        _lib = handle
    return _lib


def refuse_pricing_build(flags, path, allow=None):
    """A library built with -DICN_EXP=<bits> (tools/build_exp.sh) leaves a feature of the convolution kernel out: RESULTS ARE WRONG
    by design, only launch times are read.  It loads through the ordinary ICN_LIB_PATH override, so it is refused here unless
    ICN_ALLOW_EXP=1 says the caller knows."""
    exp = flags & 0xffff
    allow = os.environ.get('ICN_ALLOW_EXP') == '1' if allow is None else allow
    if exp and not allow:
        raise RuntimeError('geniconet_amd: %s is a pricing build (ICN_EXP=%d: a feature of the convolution kernel is compiled out, '
                           'its RESULTS ARE WRONG by design); set ICN_ALLOW_EXP=1 to time it' % (path, exp))


def build_info():
    """What was loaded: path, whether ICN_LIB_PATH overrode the in-tree library, the build switches (icn_build_flags)."""
    f = lib().icn_build_flags()
    return dict(path=LIB_PATH, overridden=bool(os.environ.get('ICN_LIB_PATH')), exp=f & 0xffff, conv_waves_default=(f >> 16) & 0xff,
                chain_prio=(f >> 24) & 0xf)


ARITH_MODES = {'f32': 0, 'bf16x3': 1}


def get_arith():
    v = lib().icn_get_arith()
    if v < 0:
        check(-1, 'icn_get_arith')
    return {0: 'f32', 1: 'bf16x3'}[v]


def set_arith(mode):
    """Arithmetic of the channel-mixing contraction ('f32' exact MFMA, 'bf16x3' three-way bf16 split; include/icn.h); returns
    the previous mode."""
    v = lib().icn_set_arith(ARITH_MODES[mode])
    if v < 0:
        check(-1, 'icn_set_arith')
    return {0: 'f32', 1: 'bf16x3'}[v]


def source_sha256():
    """sha256 over the kernel / library sources under csrc/ and include/icn.h (names + contents, sorted): identifies the code a
    profile was measured on (tools/profile_summary.py stores it, bench.py refuses a profile of other sources)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, 'csrc', '*.hip')) + glob.glob(os.path.join(_HERE, 'csrc', '*.h'))
                   + glob.glob(os.path.join(_HERE, 'csrc', '*.cpp')) + [os.path.join(os.path.dirname(_HERE), 'include', 'icn.h')])
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0')
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed: %s' % (what, lib().icn_last_error().decode()))


STATUS_BITS = {1: "a stream-K workgroup of k_conv_dma_sk never received a partner's partial tile (the tile was set to NaN)"}


def device_status(device=None, clear=True, synchronize=True):
    """Bits of the device's asynchronous failure word (icn_device_status): 0 = no kernel reported a failure."""
    import torch
    with torch.cuda.device(device):
        if synchronize:
            torch.cuda.synchronize()
        v = lib().icn_device_status(1 if clear else 0)
    if v < 0:
        check(-1, 'icn_device_status')
    return v


def raise_on_device_status(device=None, synchronize=True):
    """Raise if a kernel reported a failure since the last check; `synchronize` first waits for the device (without it only
    the kernels that have completed are covered: use it right after a point that synchronised anyway)."""
    v = device_status(device, synchronize=synchronize)
    if v:
        raise RuntimeError('libicn: asynchronous kernel failure on %s: %s' % (
            device if device is not None else 'the current device',
            '; '.join(msg for bit, msg in STATUS_BITS.items() if v & bit) or 'status 0x%x' % v))


def corner_code(corner_mode):
    try:
        return CORNER_MODES[corner_mode]
    except KeyError:
        raise ValueError("corner_mode must be 'zeros' or 'average', got %r" % (corner_mode,))


# ---- host-side table introspection (no device needed) ------------------------------------------------------
def table_conv_fwd(r_in, stride, corner_mode):
    L, m = lib(), corner_code(corner_mode)
    n = L.icn_table_conv_fwd(r_in, stride, m, None, 0)
    if n < 0:
        check(-1, 'icn_table_conv_fwd')
    out = np.empty(n, dtype=np.int32)
    L.icn_table_conv_fwd(r_in, stride, m, out.ctypes.data_as(_i32p), n)
    return out.reshape(7, -1)


def table_wgrad7(r_in, stride, corner_mode):
    """Patch form of the forward table for the all-taps weight-gradient kernel: (rows [npatch][U], pos [npatch][16][8]) or
    None when the table does not exist for this level / stride."""
    L, m = lib(), corner_code(corner_mode)
    meta = (ctypes.c_int * 2)()
    n = L.icn_table_wgrad7(r_in, stride, m, None, 0, None, 0, meta)
    if n < 0:
        check(-1, 'icn_table_wgrad7')
    if n == 0:
        return None
    U, npatch = meta[0], meta[1]
    rows, pos = np.empty(n, dtype=np.int32), np.empty(npatch * 16 * 8, dtype=np.uint16)
    L.icn_table_wgrad7(r_in, stride, m, rows.ctypes.data_as(_i32p), n, pos.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), pos.size, meta)
    return rows.reshape(npatch, U), pos.reshape(npatch, 16, 8)


def table_conv_bwd(r_in, stride, corner_mode):
    L, m = lib(), corner_code(corner_mode)
    w = ctypes.c_int(0)
    n = L.icn_table_conv_bwd(r_in, stride, m, None, 0, ctypes.byref(w))
    if n < 0:
        check(-1, 'icn_table_conv_bwd')
    out = np.empty(n, dtype=np.int32)
    L.icn_table_conv_bwd(r_in, stride, m, out.ctypes.data_as(_i32p), n, ctypes.byref(w))
    return out.reshape(7, w.value, -1)


def table_upsample(r_in, corner_mode, transpose=False):
    L, m = lib(), corner_code(corner_mode)
    w = ctypes.c_int(0)
    n = L.icn_table_upsample(r_in, m, int(transpose), None, None, 0, ctypes.byref(w))
    if n < 0:
        check(-1, 'icn_table_upsample')
    idx, coef = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float32)
    L.icn_table_upsample(r_in, m, int(transpose), idx.ctypes.data_as(_i32p), coef.ctypes.data_as(_f32p), n, ctypes.byref(w))
    return idx.reshape(-1, w.value), coef.reshape(-1, w.value)


def table_faces(r):
    L = lib()
    n = L.icn_table_faces(r, None, 0)
    if n < 0:
        check(-1, 'icn_table_faces')
    out = np.empty(n, dtype=np.int32)
    L.icn_table_faces(r, out.ctypes.data_as(_i32p), n)
    return out.reshape(-1, 3)


def table_upsample_pairs(r_in):
    L = lib()
    n = L.icn_table_upsample_pairs(r_in, None, 0)
    if n < 0:
        check(-1, 'icn_table_upsample_pairs')
    out = np.empty(n, dtype=np.int32)
    L.icn_table_upsample_pairs(r_in, out.ctypes.data_as(_i32p), n)
    return out.reshape(2, -1)


def table_upconv(r_in, corner_mode):
    """Composite upsample + conv table (icn_table_upconv) as a dict of numpy arrays: seg (nseg, 3) = rows per sample / first
    list position / virtual-tap mask, pix (Pf,), code (NV, Pf), alpha (NV, 7), slot_idx / slot_coef (n_slots, E)."""
    L, m = lib(), corner_code(corner_mode)
    meta = (ctypes.c_int * 7)()
    n = L.icn_table_upconv(r_in, m, None, 0, None, 0, meta)
    if n < 0:
        check(-1, 'icn_table_upconv')
    Pf, Pc, n_slots, E, nseg, NV, nf = list(meta)
    ints, floats = np.empty(n, dtype=np.int32), np.empty(nf, dtype=np.float32)
    L.icn_table_upconv(r_in, m, ints.ctypes.data_as(_i32p), n, floats.ctypes.data_as(_f32p), nf, meta)
    o = 0
    seg = ints[o:o + 3 * nseg].reshape(nseg, 3); o += 3 * nseg
    pix = ints[o:o + Pf]; o += Pf
    code = ints[o:o + NV * Pf].reshape(NV, Pf); o += NV * Pf
    slot_idx = ints[o:o + n_slots * E].reshape(n_slots, E)
    return dict(Pf=Pf, Pc=Pc, seg=seg, pix=pix, code=code, alpha=floats[:NV * 7].reshape(NV, 7), slot_idx=slot_idx,
                slot_coef=floats[NV * 7:].reshape(n_slots, E))


def table_upconv_bwd(r_in, corner_mode):
    """ELL matrix dy -> g of the aggregated upsample + conv backward: (idx, coef), each (7 * Pc, width), row s * 7 + t."""
    L, m = lib(), corner_code(corner_mode)
    w = ctypes.c_int(0)
    n = L.icn_table_upconv_bwd(r_in, m, None, None, 0, ctypes.byref(w))
    if n < 0:
        check(-1, 'icn_table_upconv_bwd')
    idx, coef = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float32)
    L.icn_table_upconv_bwd(r_in, m, idx.ctypes.data_as(_i32p), coef.ctypes.data_as(_f32p), n, ctypes.byref(w))
    return idx.reshape(-1, w.value), coef.reshape(-1, w.value)


def table_stream_k(ntiles, grid, nk, ku=1):
    """Stream-K schedule (icn_table_stream_k) of ntiles tiles of nk K-steps, pieces >= ku steps: int32 array (n, 4) of rows
    (workgroup, tile, k0, k1) in walk order."""
    L = lib()
    n = L.icn_table_stream_k(ntiles, grid, nk, ku, None, 0)
    if n < 0:
        check(-1, 'icn_table_stream_k')
    out = np.empty(n, dtype=np.int32)
    L.icn_table_stream_k(ntiles, grid, nk, ku, out.ctypes.data_as(_i32p), n)
    return out.reshape(-1, 4)


def table_tile_lists(r_in, B, bm, ntn, grid, corner_mode='average'):
    """Tile lists of a stride-2 data-gradient launch (icn_table_tile_lists): (offsets [grid + 1], ids)."""
    L, m = lib(), corner_code(corner_mode)
    n = L.icn_table_tile_lists(r_in, 2, m, B, bm, ntn, grid, None, 0)
    if n < 0:
        check(-1, 'icn_table_tile_lists')
    out = np.empty(n, dtype=np.int32)
    L.icn_table_tile_lists(r_in, 2, m, B, bm, ntn, grid, out.ctypes.data_as(_i32p), n)
    return out[:grid + 1], out[grid + 1:]


def profile_start(max_launches=4096, only=None):
    """Start timing MFMA launches with HIP events; only='k_conv_dma<64, 64, false>' restricts it to one kernel (an event pair
    around every launch of a step costs about 2.5 % of it)."""
    check(lib().icn_profile_select(only.encode() if only else None), 'icn_profile_select')
    check(lib().icn_profile_start(max_launches), 'icn_profile_start')


def profile_stop():
    """-> list of dicts(kernel, launches, total_ms, total_flops) for the MFMA kernels launched since profile_start."""
    buf = (ProfileEntry * 48)()
    n = lib().icn_profile_stop(buf, 48)
    if n < 0:
        check(-1, 'icn_profile_stop')
    return [dict(kernel=buf[i].kernel.decode(), launches=buf[i].launches, total_ms=buf[i].total_ms,
                 total_flops=buf[i].total_flops) for i in range(n)]
