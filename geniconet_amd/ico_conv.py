"""`IcoConvS2S` / `IcoUpsampleS2S`: the nn.Module plugin surface of `icocnn.ico_conv`, on the HIP path.

Reference call sites (the only specification of the surface that survives; upstream icocnn is absent):
  IcoConvS2S(in_features, out_features, stride, bias, subdivisions, corner_mode)   models.py:14,25-33,104-109
  IcoUpsampleS2S(in_features, subdivisions, corner_mode)                            models.py:13,45,53
Tensors are (B, C, 5*2^r, 2^(r+1)) fp32 (data.py:64-69).  `subdivisions` is the level of the INPUT
(models.py:25-29).  Parameters: `weight` (Cout, Cin, 7), `bias` (Cout).

Every forward/backward goes through the C ABI of libicn.so (include/icn.h) on the caller's current HIP
stream.  There is no CPU implementation here: CPU tensors are rejected (the CPU restatement used for parity
is test infrastructure under oracle/).
"""
import math
import os

import torch

from . import _gradbuf, _lib


def _require_gpu(x, who):
    if not x.is_cuda:
        raise RuntimeError(
            '%s: got a %s tensor; this operator only runs on the MI355X HIP path (libicn.so). '
            'Move the module and its input to a ROCm device.' % (who, x.device))
    if x.dtype != torch.float32:
        raise TypeError('%s: fp32 only, got %s' % (who, x.dtype))


def _check_grid(x, r, who):
    n = 2 ** r
    if x.dim() != 4 or x.shape[2] != 5 * n or x.shape[3] != 2 * n:
        raise ValueError('%s: expected (B, C, %d, %d) at subdivisions=%d, got %s'
                         % (who, 5 * n, 2 * n, r, tuple(x.shape)))


_NO_UPCONV_BWD = os.environ.get('ICN_NO_UPCONV_BWD', '') == '1'    # developer switch: backward of the separate operators


# Weight gradients on a second HIP stream (round 3).  A layer's weight gradient depends on nothing that follows it in the backward
# pass and nothing in the pass depends on it, so it can run beside the chain data gradient -> BatchNorm backward -> next data
# gradient -> ...: its workgroups fill the tails of the chain's persistent launches and run under its HBM-bound passes
# (+2.4 - 3.1 % of a training step, measured).  What has to be certain is that nobody reads a gradient before it is complete:
#   'off'       everything on the current stream.  The default: safe for any caller.
#   'deferred'  weight gradients go to the side stream; the current stream waits for it once, when the whole backward pass has
#               run (autograd engine callback).  For callers that do not look at parameter gradients before backward()
#               returns: the Trainer without DistributedDataParallel sets it for the duration of its step.
#   'bucketed'  as 'deferred', and additionally the current stream waits for the side stream at the moment the LAST gradient
#               of a DistributedDataParallel bucket is handed to autograd -- i.e. just before the reducer launches that
#               bucket's all-reduce, the only reader of gradients inside a backward pass (the Trainer arms it with the
#               parameter -> bucket map it reads off the reducer's gradient views; tensor hooks on the parameters count).
#   'eager'     the current stream waits before every backward Function returns (a weight gradient then only overlaps its own
#               layer's data gradient).  Measured SLOWER than 'off' (two MFMA-bound launches side by side): kept for tests.
_MODES = ('off', 'eager', 'deferred', 'bucketed')
_wgrad_mode = [os.environ.get('ICN_WGRAD_STREAM', 'off')]
if _wgrad_mode[0] not in _MODES:
    raise ValueError('ICN_WGRAD_STREAM must be one of %s' % (_MODES,))
_side_streams = {}
_pending = [False]                                  # something went to the side stream since the current stream last waited for it
# Per-pass bookkeeping is keyed on the PASS (the autograd engine's graph-task id), never on a global flag: a pass that ends in an
# exception runs no end-of-pass callback, and whatever it leaves behind must not be mistaken for the next pass's state.
_seen = {}                                          # id(parameter) -> (pass id, one of its gradients of that pass is on the side stream)
_queued = set()                                     # passes whose end-of-pass callback has been queued
_known = set()                                      # passes seen (and not yet over): the bucket countdown is armed at a pass's first event
_last_pass = [None]                                 # the pass the previous event belonged to
_buckets = {'of': {}, 'size': {}, 'left': {}}       # id(param) -> bucket key; bucket key -> parameters in it / still to come
wgrad_stream_counts = {'side': 0, 'joins': 0, 'kept': 0}   # launches put on the side stream / waits issued / launches kept on
                                                          # the current stream because a reader could not wait (tests, diagnostics)


def set_weight_gradient_stream(mode, bucket_of=None):
    """Select the mode (see above); 'bucketed' needs bucket_of = {parameter: bucket key}.  Returns the previous (mode, map)."""
    if mode not in _MODES:
        raise ValueError('mode must be one of %s' % (_MODES,))
    prev = (_wgrad_mode[0], _buckets.get('map'))
    # A mode switch is a boundary between backward passes (the Trainer switches before and after every backward()): whatever a pass
    # left behind -- it may have ended in an exception, in which case the engine never ran the end-of-pass callback -- is settled
    # here, so that the next pass starts with a joined side stream and empty per-pass bookkeeping.
    if _pending[0]:
        _join_now()
    _forget_passes()
    if mode == 'bucketed':
        if not bucket_of:
            raise ValueError("mode 'bucketed' needs the parameter -> bucket map")
        if bucket_of is not _buckets.get('map'):
            _buckets['map'] = bucket_of
            _buckets['of'] = {id(p): k for p, k in bucket_of.items()}
            size = {}
            for k in bucket_of.values():
                size[k] = size.get(k, 0) + 1
            _buckets['size'] = size
        _buckets['left'] = dict(_buckets['size'])
    _wgrad_mode[0] = mode
    return prev


def _current_pass():
    """Id of the backward pass the engine is running on this thread (-1 outside one)."""
    return torch._C._current_graph_task_id()


def _forget_passes():
    _seen.clear()
    _queued.clear()
    _known.clear()
    _last_pass[0] = None


def _enter_pass():
    """Called by everything that acts inside a backward pass; returns the pass id.  An event of another pass than the previous
    event's is a pass boundary nobody announced: either the previous pass ended in an exception (OOM, KeyboardInterrupt; its
    end-of-pass callback never ran, the side stream may be un-joined) or passes are nested.  Either way the current stream waits
    for whatever is still on the side stream -- after that every earlier gradient is complete and only this pass's own launches
    need the bookkeeping, which is looked up by pass id, so a stale pass's entries are simply never matched.  The bucket
    countdown is (re)armed at a pass's first event."""
    tid = _current_pass()
    if tid != _last_pass[0]:
        if _pending[0]:
            _join_now()
        _last_pass[0] = tid
    if tid not in _known:
        if len(_known) > 16 or len(_seen) > 8192:      # leftovers of passes that raised: ids and ints only, drop them
            keep = {k: v for k, v in _seen.items() if v[0] == tid}
            _forget_passes()
            _seen.update(keep)
            _last_pass[0] = tid
        _known.add(tid)
        _buckets['left'] = dict(_buckets['size'])
    return tid


def _join_now():
    for dev_index, side in _side_streams.items():
        torch.cuda.current_stream(dev_index).wait_stream(side)
    _pending[0] = False
    wgrad_stream_counts['joins'] += 1


def _backward_pass_over(tid=None):
    """Autograd-engine callback at the end of backward pass `tid` (queued by the pass's first weight-gradient launch in a
    side-stream mode): the current stream waits for the side stream, the pass's bookkeeping is forgotten."""
    if _pending[0]:
        _join_now()
    if tid is None:
        _forget_passes()
    else:
        for k in [k for k, v in _seen.items() if v[0] == tid]:
            del _seen[k]
        _queued.discard(tid)
        _known.discard(tid)
        if _last_pass[0] == tid:
            _last_pass[0] = None
    _buckets['left'] = dict(_buckets['size'])


def _queue_end_of_pass(tid):
    torch.autograd.Variable._execution_engine.queue_callback(lambda: _backward_pass_over(tid))


def parameter_gradient_ready(param):
    """Tensor hook of a parameter (registered by the Trainer under DistributedDataParallel): its gradient is about to be
    accumulated, after which the reducer may launch the all-reduce of the parameter's bucket."""
    if _wgrad_mode[0] != 'bucketed':
        return
    _enter_pass()
    pending = _pending[0]                                        # something is on the side stream that nobody has waited for
    k = _buckets['of'].get(id(param))
    if k is None:                                                # not in the map: take no chances
        if pending:
            _join_now()
        return
    left = _buckets['left'][k] = _buckets['left'].get(k, 1) - 1  # (every gradient counts, also those that arrive before the
    if left <= 0 and pending:                                    #  first weight gradient of the pass went to the side stream)
        _join_now()


def _readers_can_wait(q, g):
    """True when nothing reads the gradient tensor `g` of parameter `q` on the CURRENT stream before the pass's join, i.e. when
    writing it on the side stream is safe.  Autograd and DistributedDataParallel do read it whenever it is not simply taken as
    `q.grad` (or found aliasing its bucket):
      * `q.grad` exists (gradient accumulation, `zero_grad(set_to_none=False)`): AccumulateGrad runs `q.grad += g`;
      * `q` already received a gradient in this pass (a module applied twice in one graph): the engine adds the two;
      * `q` is not a leaf: `g` flows on into other backward nodes;
      * somebody's tensor hook / post-accumulate hook on `q` looks at it ('deferred'; under 'bucketed' the Trainer's own
        countdown hooks are the ones registered);
      * 'bucketed': `g` is not the parameter's bucket view (the lease fell back to a new tensor), so the reducer copies it
        into the bucket -- on the current stream, and only the bucket's LAST gradient is preceded by a join."""
    if not isinstance(q, torch.Tensor) or not q.is_leaf or q.grad is not None or _seen.get(id(q), (None,))[0] == _current_pass():
        return False
    if _wgrad_mode[0] == 'bucketed':
        return _gradbuf.served_from_view(q, g)
    return not (q._backward_hooks or getattr(q, '_post_accumulate_grad_hooks', None))


def _wgrad_stream(dev, dests, *tensors, allow=True):
    """The side stream for a weight-gradient launch issued from inside a backward pass, ordered after everything the current
    stream has been given so far; `dests` = [(parameter, tensor its gradient is about to be written to)], `tensors` (allocated
    on the current stream) are kept alive for the side stream.  None -- the launch stays on the current stream -- in mode 'off'
    and whenever one of the destinations could be read on the current stream before the join (_readers_can_wait), or the caller
    does not want it there (`allow`); when an earlier gradient of one of these parameters is still on the side stream, the
    current stream waits for it first -- which is why callers that keep a launch on the current stream for their own reasons
    must still come through here."""
    if _wgrad_mode[0] == 'off':
        return None
    tid = _enter_pass()
    dests = [(q, g) for q, g in dests if g is not None]
    ok = allow and (_wgrad_mode[0] == 'eager' or (all(_readers_can_wait(q, g) for q, g in dests)
                                                  and len({id(q) for q, _ in dests}) == len(dests)))   # (one tensor as both weights)
    if _wgrad_mode[0] != 'eager' and tid not in _queued:
        _queue_end_of_pass(tid)                                                       # once per pass: join + forget
        _queued.add(tid)
    mine = {id(q): _seen[id(q)][1] for q, _ in dests if _seen.get(id(q), (None,))[0] == tid}   # earlier gradients of THIS pass
    again = any(mine.values())
    for q, _ in dests:
        _seen[id(q)] = (tid, mine.get(id(q), False) or ok)
    if not ok:
        if again and _pending[0]:
            _join_now()
        if allow:                                    # (not the caller's own choice: a reader could not have waited)
            wgrad_stream_counts['kept'] += 1
        return None
    side = _side_streams.get(dev.index)
    if side is None:
        side = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    for t in tensors + tuple(g for _, g in dests):
        if t is not None:
            t.record_stream(side)
    _pending[0] = True
    wgrad_stream_counts['side'] += 1
    return side


def _wgrad_done(dev, side):
    """End of a backward Function that put its weight gradient on `side`: in 'eager' mode the current stream waits for it."""
    if side is not None and _wgrad_mode[0] == 'eager':
        torch.cuda.current_stream(dev).wait_stream(side)
        _pending[0] = False
        wgrad_stream_counts['joins'] += 1


def _empty_batch(x, channels, n_out, *params):
    """An empty batch (B = 0): the empty output of the right shape, still attached to the graph -- x and the parameters get
    all-zero gradients, as torch's own conv2d gives them -- without a kernel launch (the C ABI rejects B < 1)."""
    keep = x.sum() * 0
    for p in params:
        if p is not None:
            keep = keep + p.sum() * 0
    return x.new_zeros((0, channels, 5 * n_out, 2 * n_out)) + keep


def _nhwc(x):
    """(B,C,H,W) logical -> contiguous (B,H,W,C) storage; free when x is already channels_last."""
    return x.permute(0, 2, 3, 1).contiguous()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


class _IcoConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, r, stride, mode):
        L = _lib.lib()
        B, Cin = x.shape[0], x.shape[1]
        Cout = weight.shape[0]
        n_out = 2 ** r // stride
        xp = _nhwc(x)
        w = weight.contiguous()
        b = bias.contiguous() if bias is not None else None
        y = torch.empty(B, 5 * n_out, 2 * n_out, Cout, dtype=torch.float32, device=x.device)
        ws_bytes = L.icn_conv_workspace_bytes(_lib.OP_CONV_FWD, B, Cin, Cout, r, stride)
        ws = _workspace(ws_bytes, x.device)
        with torch.cuda.device(x.device):
            rc = L.icn_conv_fwd(xp.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                                B, Cin, Cout, r, stride, mode, ws.data_ptr(), ws_bytes, _stream())
        _lib.check(rc, 'icn_conv_fwd')
        ctx.save_for_backward(xp, w)
        ctx.cfg = (B, Cin, Cout, r, stride, mode, bias is not None)
        ctx.params = (weight, bias)                      # where their gradients go (_gradbuf.lease)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        L = _lib.lib()
        xp, w = ctx.saved_tensors
        B, Cin, Cout, r, stride, mode, has_bias = ctx.cfg
        gyp = _nhwc(gy)
        dx = dw = db = side = None
        with torch.cuda.device(gyp.device):
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):     # first: it goes to the side stream
                dw = _gradbuf.lease(ctx.params[0], w.shape, w.device)
                db = _gradbuf.lease(ctx.params[1], (Cout,), w.device) if has_bias else None
                ws_bytes = L.icn_conv_workspace_bytes(_lib.OP_CONV_BWD_WEIGHT, B, Cin, Cout, r, stride)
                ws = _workspace(ws_bytes, gyp.device)
                # (a convolution whose input needs no gradient -- the stem -- is the last thing in the pass: its weight gradient
                #  stays on the current stream, which has nothing else left to do, instead of queueing behind the side stream's
                #  backlog: +1.2 %, 4346-4370 against 4298-4314 meshes/s; keeping the last one or two OTHER weight gradients on
                #  the current stream as well loses 1 - 1.5 %)
                side = _wgrad_stream(gyp.device, [(ctx.params[0], dw), (ctx.params[1], db)], xp, gyp, ws,
                                     allow=bool(ctx.needs_input_grad[0]))
                # (Round 6: for a while this launch first joined the side stream -- the stem's weight gradient came out different from
                #  run to run while the first residual block's bf16x3 pair gradient ran beside it.  The cause was in k_stem_wgrad's machine
                #  code, not in the streams: v_pk_fma_f32 with a cross-dword op_sel, see csrc/Makefile and profiles/r06_stem_wgrad_race.txt.)
                rc = L.icn_conv_bwd_weight(xp.data_ptr(), gyp.data_ptr(), dw.data_ptr(),
                                           db.data_ptr() if db is not None else None, B, Cin, Cout, r, stride, mode,
                                           ws.data_ptr(), ws_bytes, side.cuda_stream if side is not None else _stream())
                _lib.check(rc, 'icn_conv_bwd_weight')
            if ctx.needs_input_grad[0]:
                dxp = torch.empty_like(xp)
                ws_bytes = L.icn_conv_workspace_bytes(_lib.OP_CONV_BWD_DATA, B, Cin, Cout, r, stride)
                ws = _workspace(ws_bytes, gyp.device)
                rc = L.icn_conv_bwd_data(gyp.data_ptr(), w.data_ptr(), dxp.data_ptr(), B, Cin, Cout, r, stride, mode,
                                         ws.data_ptr(), ws_bytes, _stream())
                _lib.check(rc, 'icn_conv_bwd_data')
                dx = dxp.permute(0, 3, 1, 2)
            _wgrad_done(gyp.device, side)
        return dx, dw, db, None, None, None


class _IcoConvPairFn(torch.autograd.Function):
    """Two IcoConvS2S of the SAME input (the conv00 / conv10 branches of the reference's residual blocks,
    models.py:37-39,59-60) as one launch per pass -- icn_conv_pair_* in include/icn.h.  Same results as two
    _IcoConvFn calls (bwd-data: as their sum)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, r, stride, mode):
        L = _lib.lib()
        B, Cin = x.shape[0], x.shape[1]
        C0, C1 = w0.shape[0], w1.shape[0]
        n_out = 2 ** r // stride
        xp = _nhwc(x)
        w0c, w1c = w0.contiguous(), w1.contiguous()
        b0c = b0.contiguous() if b0 is not None else None
        b1c = b1.contiguous() if b1 is not None else None
        y0 = torch.empty(B, 5 * n_out, 2 * n_out, C0, dtype=torch.float32, device=x.device)
        y1 = torch.empty(B, 5 * n_out, 2 * n_out, C1, dtype=torch.float32, device=x.device)
        ws_bytes = L.icn_conv_pair_workspace_bytes(_lib.OP_CONV_FWD, B, Cin, C0, C1, r, stride)
        ws = _workspace(ws_bytes, x.device)
        with torch.cuda.device(x.device):
            rc = L.icn_conv_pair_fwd(xp.data_ptr(), w0c.data_ptr(), b0c.data_ptr() if b0c is not None else None,
                                     w1c.data_ptr(), b1c.data_ptr() if b1c is not None else None, y0.data_ptr(),
                                     y1.data_ptr(), B, Cin, C0, C1, r, stride, mode, ws.data_ptr(), ws_bytes, _stream())
        _lib.check(rc, 'icn_conv_pair_fwd')
        ctx.save_for_backward(xp, w0c, w1c)
        ctx.cfg = (B, Cin, C0, C1, r, stride, mode, b0 is not None)
        ctx.params = (w0, b0, w1, b1)
        return y0.permute(0, 3, 1, 2), y1.permute(0, 3, 1, 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy0, gy1):
        L = _lib.lib()
        xp, w0, w1 = ctx.saved_tensors
        B, Cin, C0, C1, r, stride, mode, has_bias = ctx.cfg
        g0, g1 = _nhwc(gy0), _nhwc(gy1)
        dx = dw0 = db0 = dw1 = db1 = side = None
        need = ctx.needs_input_grad
        with torch.cuda.device(g0.device):
            if need[1] or need[3] or (has_bias and (need[2] or need[4])):                # first: it goes to the side stream
                pw0, pb0, pw1, pb1 = ctx.params
                dw0, dw1 = _gradbuf.lease(pw0, w0.shape, w0.device), _gradbuf.lease(pw1, w1.shape, w1.device)
                if has_bias:
                    db0 = _gradbuf.lease(pb0, (C0,), w0.device)
                    db1 = _gradbuf.lease(pb1, (C1,), w1.device)
                ws_bytes = L.icn_conv_pair_workspace_bytes(_lib.OP_CONV_BWD_WEIGHT, B, Cin, C0, C1, r, stride)
                ws = _workspace(ws_bytes, g0.device)
                side = _wgrad_stream(g0.device, [(pw0, dw0), (pb0, db0), (pw1, dw1), (pb1, db1)], xp, g0, g1, ws)
                rc = L.icn_conv_pair_bwd_weight(xp.data_ptr(), g0.data_ptr(), g1.data_ptr(), dw0.data_ptr(),
                                                db0.data_ptr() if db0 is not None else None, dw1.data_ptr(),
                                                db1.data_ptr() if db1 is not None else None, B, Cin, C0, C1, r, stride, mode,
                                                ws.data_ptr(), ws_bytes, side.cuda_stream if side is not None else _stream())
                _lib.check(rc, 'icn_conv_pair_bwd_weight')
            if need[0]:
                dxp = torch.empty_like(xp)
                ws_bytes = L.icn_conv_pair_workspace_bytes(_lib.OP_CONV_BWD_DATA, B, Cin, C0, C1, r, stride)
                ws = _workspace(ws_bytes, g0.device)
                rc = L.icn_conv_pair_bwd_data(g0.data_ptr(), g1.data_ptr(), w0.data_ptr(), w1.data_ptr(), dxp.data_ptr(),
                                              B, Cin, C0, C1, r, stride, mode, ws.data_ptr(), ws_bytes, _stream())
                _lib.check(rc, 'icn_conv_pair_bwd_data')
                dx = dxp.permute(0, 3, 1, 2)
            _wgrad_done(g0.device, side)
        return dx, dw0, db0, dw1, db1, None, None, None


class _IcoUpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, mode):
        L = _lib.lib()
        B, C = x.shape[0], x.shape[1]
        n = 2 ** r
        xp = _nhwc(x)
        y = torch.empty(B, 10 * n, 4 * n, C, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = L.icn_upsample_fwd(xp.data_ptr(), y.data_ptr(), B, C, r, mode, _stream())
        _lib.check(rc, 'icn_upsample_fwd')
        ctx.cfg = (B, C, r, mode)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        L = _lib.lib()
        B, C, r, mode = ctx.cfg
        n = 2 ** r
        gyp = _nhwc(gy)
        dx = torch.empty(B, 5 * n, 2 * n, C, dtype=torch.float32, device=gy.device)
        with torch.cuda.device(gy.device):
            rc = L.icn_upsample_bwd(gyp.data_ptr(), dx.data_ptr(), B, C, r, mode, _stream())
        _lib.check(rc, 'icn_upsample_bwd')
        return dx.permute(0, 3, 1, 2), None, None


class _IcoUpConvPairFn(torch.autograd.Function):
    """(conv0(upsample(x)), conv1(upsample(x))) -- the head of the reference's decoder block (models.py:58-60) -- with the
    FORWARD computed from the coarse tensor by one composite gather-GEMM (icn_upconv_fwd in include/icn.h: 0.68 of the
    multiply-adds, no upsampled tensor) and the BACKWARD through one coarse-level aggregate of the output gradients
    (icn_upconv_bwd: both gradients as dense coarse-level contractions, a quarter of the multiply-adds).  With 'zeros' poles
    or shapes outside that path the backward is that of the separate operators: the upsample is recomputed from x, the pair's
    bwd-data / bwd-weight kernels run on it, and the upsample's transpose brings the gradient back to the coarse level.
    Same results as ico_conv_pair(ico_upsample(x), ...) up to fp32 rounding order."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, r, mode):
        L = _lib.lib()
        B, Cin = x.shape[0], x.shape[1]
        C0, C1 = w0.shape[0], w1.shape[0]
        nf = 2 ** (r + 1)
        xp = _nhwc(x)
        w0c, w1c = w0.contiguous(), w1.contiguous()
        b0c = b0.contiguous() if b0 is not None else None
        b1c = b1.contiguous() if b1 is not None else None
        y0 = torch.empty(B, 5 * nf, 2 * nf, C0, dtype=torch.float32, device=x.device)
        y1 = torch.empty(B, 5 * nf, 2 * nf, C1, dtype=torch.float32, device=x.device)
        ws_bytes = L.icn_upconv_workspace_bytes(B, Cin, C0, C1, r)
        ws = _workspace(ws_bytes, x.device)
        with torch.cuda.device(x.device):
            rc = L.icn_upconv_fwd(xp.data_ptr(), w0c.data_ptr(), b0c.data_ptr() if b0c is not None else None, w1c.data_ptr(),
                                  b1c.data_ptr() if b1c is not None else None, y0.data_ptr(), y1.data_ptr(), B, Cin, C0, C1, r,
                                  mode, ws.data_ptr(), ws_bytes, _stream())
        _lib.check(rc, 'icn_upconv_fwd')
        ctx.save_for_backward(xp, w0c, w1c)
        ctx.cfg = (B, Cin, C0, C1, r, mode, b0 is not None)
        ctx.params = (w0, b0, w1, b1)
        return y0.permute(0, 3, 1, 2), y1.permute(0, 3, 1, 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy0, gy1):
        L = _lib.lib()
        xp, w0, w1 = ctx.saved_tensors
        B, Cin, C0, C1, r, mode, has_bias = ctx.cfg
        rf, n, nf = r + 1, 2 ** r, 2 ** (r + 1)
        g0, g1 = _nhwc(gy0), _nhwc(gy1)
        dev = g0.device
        dx = dw0 = db0 = dw1 = db1 = None
        need = ctx.needs_input_grad
        want_w = need[1] or need[3] or (has_bias and (need[2] or need[4]))
        if not _NO_UPCONV_BWD and L.icn_upconv_bwd_supported(B, Cin, C0, C1, r, mode):
            # both gradients from one coarse-level aggregate of (gy0 | gy1): a quarter of the fine level's multiply-adds
            dxp = torch.empty(B, 5 * n, 2 * n, Cin, dtype=torch.float32, device=dev) if need[0] else None
            if want_w:
                pw0, pb0, pw1, pb1 = ctx.params
                dw0, dw1 = _gradbuf.lease(pw0, w0.shape, dev), _gradbuf.lease(pw1, w1.shape, dev)
                if has_bias:
                    db0 = _gradbuf.lease(pb0, (C0,), dev)
                    db1 = _gradbuf.lease(pb1, (C1,), dev)
            ws_bytes = L.icn_upconv_bwd_workspace_bytes(B, Cin, C0, C1, r)
            ws = _workspace(ws_bytes, dev)
            ptr = lambda t: t.data_ptr() if t is not None else None
            with torch.cuda.device(dev):
                # the weight gradients (after the aggregate pass, which stays on the current stream) on the side stream
                side = (_wgrad_stream(dev, [(pw0, dw0), (pb0, db0), (pw1, dw1), (pb1, db1)], xp, ws)
                        if want_w else None)
                _lib.check(L.icn_upconv_bwd_streams(xp.data_ptr(), g0.data_ptr(), g1.data_ptr(), w0.data_ptr(), w1.data_ptr(),
                                                    ptr(dxp), ptr(dw0), ptr(db0), ptr(dw1), ptr(db1), B, Cin, C0, C1, r, mode,
                                                    ws.data_ptr(), ws_bytes, _stream(),
                                                    side.cuda_stream if side is not None else None), 'icn_upconv_bwd_streams')
                _wgrad_done(dev, side)
            return (dxp.permute(0, 3, 1, 2) if dxp is not None else None), dw0, db0, dw1, db1, None, None
        with torch.cuda.device(dev):
            st = _stream()
            if need[0]:
                dup = torch.empty(B, 5 * nf, 2 * nf, Cin, dtype=torch.float32, device=dev)
                ws_bytes = L.icn_conv_pair_workspace_bytes(_lib.OP_CONV_BWD_DATA, B, Cin, C0, C1, rf, 1)
                ws = _workspace(ws_bytes, dev)
                _lib.check(L.icn_conv_pair_bwd_data(g0.data_ptr(), g1.data_ptr(), w0.data_ptr(), w1.data_ptr(), dup.data_ptr(), B,
                                                    Cin, C0, C1, rf, 1, mode, ws.data_ptr(), ws_bytes, st), 'icn_conv_pair_bwd_data')
                dxp = torch.empty(B, 5 * n, 2 * n, Cin, dtype=torch.float32, device=dev)
                _lib.check(L.icn_upsample_bwd(dup.data_ptr(), dxp.data_ptr(), B, Cin, r, mode, st), 'icn_upsample_bwd')
                del dup
                dx = dxp.permute(0, 3, 1, 2)
            if want_w:
                up = torch.empty(B, 5 * nf, 2 * nf, Cin, dtype=torch.float32, device=dev)
                _lib.check(L.icn_upsample_fwd(xp.data_ptr(), up.data_ptr(), B, Cin, r, mode, st), 'icn_upsample_fwd')
                dw0, dw1 = torch.empty_like(w0), torch.empty_like(w1)
                if has_bias:
                    db0 = torch.empty(C0, dtype=torch.float32, device=dev)
                    db1 = torch.empty(C1, dtype=torch.float32, device=dev)
                ws_bytes = L.icn_conv_pair_workspace_bytes(_lib.OP_CONV_BWD_WEIGHT, B, Cin, C0, C1, rf, 1)
                ws = _workspace(ws_bytes, dev)
                _lib.check(L.icn_conv_pair_bwd_weight(up.data_ptr(), g0.data_ptr(), g1.data_ptr(), dw0.data_ptr(),
                                                      db0.data_ptr() if db0 is not None else None, dw1.data_ptr(),
                                                      db1.data_ptr() if db1 is not None else None, B, Cin, C0, C1, rf, 1, mode,
                                                      ws.data_ptr(), ws_bytes, st), 'icn_conv_pair_bwd_weight')
        return dx, dw0, db0, dw1, db1, None, None


def ico_upconv_pair_supported(x, weight0, weight1, subdivisions):
    """True when (conv0(upsample(x)), conv1(upsample(x))) can take the composite forward (icn_upconv_fwd) and the pair
    backward kernels at the fine level; `subdivisions` is the level of x."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight0.dim() == 3 and weight1.dim() == 3):
        return False
    L = _lib.lib()
    B, Cin, C0, C1 = x.shape[0], x.shape[1], weight0.shape[0], weight1.shape[0]
    return bool(L.icn_upconv_supported(B, Cin, C0, C1, subdivisions)
                and L.icn_conv_pair_supported(B, Cin, C0, C1, subdivisions + 1, 1))


def ico_upconv_pair(x, weight0, bias0, weight1, bias1, subdivisions, corner_mode='zeros'):
    """ico_conv_pair(ico_upsample(x, subdivisions), ..., subdivisions + 1) with the forward computed from the coarse tensor."""
    _require_gpu(x, 'ico_upconv_pair')
    _check_grid(x, subdivisions, 'ico_upconv_pair')
    for w in (weight0, weight1):
        if w.dim() != 3 or w.shape[1] != x.shape[1] or w.shape[2] != 7:
            raise ValueError('ico_upconv_pair: weight must be (Cout, %d, 7), got %s' % (x.shape[1], tuple(w.shape)))
    if (bias0 is None) != (bias1 is None):
        raise ValueError('ico_upconv_pair: both convolutions carry a bias or neither does')
    if x.shape[0] == 0:
        n_out = 2 ** (subdivisions + 1)
        return _empty_batch(x, weight0.shape[0], n_out, weight0, bias0), _empty_batch(x, weight1.shape[0], n_out, weight1, bias1)
    return _IcoUpConvPairFn.apply(x, weight0, bias0, weight1, bias1, subdivisions, _lib.corner_code(corner_mode))


def ico_conv(x, weight, bias, subdivisions, stride=1, corner_mode='zeros'):
    """Functional form of IcoConvS2S.forward."""
    _require_gpu(x, 'ico_conv')
    _check_grid(x, subdivisions, 'ico_conv')
    if weight.dim() != 3 or weight.shape[1] != x.shape[1] or weight.shape[2] != 7:
        raise ValueError('ico_conv: weight must be (Cout, %d, 7), got %s' % (x.shape[1], tuple(weight.shape)))
    if stride not in (1, 2):
        raise ValueError('ico_conv: stride must be 1 or 2')
    if x.shape[0] == 0:
        return _empty_batch(x, weight.shape[0], 2 ** subdivisions // stride, weight, bias)
    return _IcoConvFn.apply(x, weight, bias, subdivisions, stride, _lib.corner_code(corner_mode))


def ico_conv_pair_supported(x, weight0, weight1, subdivisions, stride=1):
    """True when two convolutions of x with these weights can take the one-launch-per-pass pair path."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight0.dim() == 3 and weight1.dim() == 3):
        return False
    return bool(_lib.lib().icn_conv_pair_supported(x.shape[0], x.shape[1], weight0.shape[0], weight1.shape[0],
                                                    subdivisions, stride))


def ico_conv_pair(x, weight0, bias0, weight1, bias1, subdivisions, stride=1, corner_mode='zeros'):
    """(ico_conv(x, weight0, bias0, ...), ico_conv(x, weight1, bias1, ...)) in one launch per pass.  Raises when the
    shape is outside the pair path (ico_conv_pair_supported); the caller then uses two ico_conv calls."""
    _require_gpu(x, 'ico_conv_pair')
    _check_grid(x, subdivisions, 'ico_conv_pair')
    for w in (weight0, weight1):
        if w.dim() != 3 or w.shape[1] != x.shape[1] or w.shape[2] != 7:
            raise ValueError('ico_conv_pair: weight must be (Cout, %d, 7), got %s' % (x.shape[1], tuple(w.shape)))
    if (bias0 is None) != (bias1 is None):
        raise ValueError('ico_conv_pair: both convolutions carry a bias or neither does')
    if stride not in (1, 2):
        raise ValueError('ico_conv_pair: stride must be 1 or 2')
    if x.shape[0] == 0:
        n_out = 2 ** subdivisions // stride
        return _empty_batch(x, weight0.shape[0], n_out, weight0, bias0), _empty_batch(x, weight1.shape[0], n_out, weight1, bias1)
    return _IcoConvPairFn.apply(x, weight0, bias0, weight1, bias1, subdivisions, stride, _lib.corner_code(corner_mode))


def ico_upsample(x, subdivisions, corner_mode='zeros'):
    """Functional form of IcoUpsampleS2S.forward."""
    _require_gpu(x, 'ico_upsample')
    _check_grid(x, subdivisions, 'ico_upsample')
    if x.shape[0] == 0:
        return _empty_batch(x, x.shape[1], 2 ** (subdivisions + 1))
    return _IcoUpsampleFn.apply(x, subdivisions, _lib.corner_code(corner_mode))


class IcoConvS2S(torch.nn.Module):
    """7-tap hexagonal conv over the 5 icosahedral charts (scalar-to-scalar features), stride 1 or 2."""

    def __init__(self, in_features, out_features, stride=1, bias=True, subdivisions=0, corner_mode='zeros'):
        super().__init__()
        if stride not in (1, 2):
            raise ValueError('IcoConvS2S: stride must be 1 or 2')
        if stride == 2 and subdivisions < 1:
            raise ValueError('IcoConvS2S: stride 2 needs subdivisions >= 1')
        _lib.corner_code(corner_mode)
        self.in_features, self.out_features = in_features, out_features
        self.stride, self.subdivisions, self.corner_mode = stride, subdivisions, corner_mode
        self.weight = torch.nn.Parameter(torch.empty(out_features, in_features, 7))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        # upstream init is unknown; kaiming_uniform(a=sqrt(5)) on fan_in = 7*Cin, i.e. nn.Conv2d's default
        bound = 1.0 / math.sqrt(7 * self.in_features)
        torch.nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            torch.nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return ico_conv(x, self.weight, self.bias, self.subdivisions, self.stride, self.corner_mode)

    def extra_repr(self):
        return '%d, %d, stride=%d, subdivisions=%d, corner_mode=%s, bias=%s' % (
            self.in_features, self.out_features, self.stride, self.subdivisions, self.corner_mode, self.bias is not None)


class IcoUpsampleS2S(torch.nn.Module):
    """subdivisions -> subdivisions+1: copy at coarse sites, edge-midpoint mean elsewhere."""

    def __init__(self, in_features, subdivisions=0, corner_mode='zeros'):
        super().__init__()
        _lib.corner_code(corner_mode)
        self.in_features, self.subdivisions, self.corner_mode = in_features, subdivisions, corner_mode

    def forward(self, x):
        return ico_upsample(x, self.subdivisions, self.corner_mode)

    def extra_repr(self):
        return '%d, subdivisions=%d, corner_mode=%s' % (self.in_features, self.subdivisions, self.corner_mode)
