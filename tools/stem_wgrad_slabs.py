"""Developer tool (GPU; round 6, profiles/r06_stem_wgrad_race.txt): the slabs k_stem_wgrad writes INSIDE a full-size training step -- copied out by
hipMemcpyAsync right behind the launch: no allocation, no kernel in front of it -- against a clean repetition of the same launch from the same
dy after the step.  Prints which slabs / (tap, ci) rows / float4 components differ.  ICN_TREE=<checkout> runs another tree's package (the search
ran on a worktree of commit e573a08, whose library differed in every run); a join in front of the stem's launch, where a tree has one, is bypassed."""
import sys, os, importlib, ctypes
sys.path.insert(0, os.environ.get('ICN_TREE') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geniconet_amd import data, models, _lib
from geniconet_amd.train import Trainer
from geniconet_amd.ico_conv import set_weight_gradient_stream
ico = importlib.import_module('geniconet_amd.ico_conv')
R, B = 5, 36
p = models.default_params('ico2ico', subdivisions=R)
x, t = data.synthetic_batch(B, R, seed=1234, device='cuda')
x = x.contiguous(memory_format=torch.channels_last)
L = _lib.lib()
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpyAsync.restype = ctypes.c_int
stem_bytes = int(L.icn_conv_workspace_bytes(_lib.OP_CONV_BWD_WEIGHT, B, 3, 64, R, 1))
rows = B * 10 * (2 ** R) ** 2
saveA = torch.zeros(stem_bytes, dtype=torch.uint8, device='cuda')
saveDy = torch.zeros(rows * 64, dtype=torch.float32, device='cuda')
state = {'bypass': True, 'calls': 0, 'pending': False}

class Proxy:
    def __getattr__(self, name):
        f = getattr(L, name)
        if name != 'icn_conv_bwd_weight':
            return f
        def wrapped(xp, dy, dw, db, Bb, Cin, Cout, r, stride, mode, ws, ws_bytes, stream):
            rc = f(xp, dy, dw, db, Bb, Cin, Cout, r, stride, mode, ws, ws_bytes, stream)
            if Cin == 3 and state['bypass']:
                hip.hipMemcpyAsync(saveA.data_ptr(), ws, stem_bytes, 3, stream)
                hip.hipMemcpyAsync(saveDy.data_ptr(), dy, rows * 64 * 4, 3, stream)
                ico._pending[0] = state['pending']
                state['calls'] += 1
                state['mode'] = mode
            return rc
        return wrapped
proxy = Proxy()
_lib.lib = lambda: proxy
real_ws_stream = ico._wgrad_stream
def ws_wrap(dev, dests, *tensors, allow=True):
    side = real_ws_stream(dev, dests, *tensors, allow=allow)
    if not allow and state['bypass']:
        state['pending'] = ico._pending[0]
        ico._pending[0] = False                     # the product's join in front of the stem's weight gradient is skipped
    return side
ico._wgrad_stream = ws_wrap

slab = 21 * 64
S = 720
for rep in range(6):
    tr = Trainer(p, 'cuda', seed=0)
    stem = tr.model.encoder[0]
    out = tr.net(x)
    loss = tr.criterion(out, t)
    tr.optimizer.zero_grad()
    prev = set_weight_gradient_stream(*tr._weight_gradient_mode())
    loss.backward()
    set_weight_gradient_stream(*prev)
    torch.cuda.synchronize()
    dw = stem.weight.grad.clone()
    A = saveA.clone().view(torch.float32)
    # clean repetition: same x, the saved dy, nothing else on the device
    dw2 = torch.empty_like(stem.weight.grad); db2 = torch.empty(64, device='cuda')
    ws2 = torch.zeros(stem_bytes, dtype=torch.uint8, device='cuda')
    xp = x.permute(0, 2, 3, 1).contiguous()
    rc = L.icn_conv_bwd_weight(xp.data_ptr(), saveDy.data_ptr(), dw2.data_ptr(), db2.data_ptr(), B, 3, 64, R, 1, state['mode'], ws2.data_ptr(), stem_bytes,
                               torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Bv = ws2.view(torch.float32)
    n = S * slab
    d = (A[:n] != Bv[:n]).nonzero().flatten()
    print('rep %d: calls %d rc %d; dw (in step) == dw (clean) %s; differing slab floats: %d' % (rep, state['calls'], rc, torch.equal(dw, dw2), d.numel()), flush=True)
    if d.numel():
        idx = d.tolist()
        slabs = sorted({i // slab for i in idx})
        print('   slabs (blocks) touched: %d of %d: %s' % (len(slabs), S, slabs[:40]))
        ks = sorted({(i % slab) // 64 for i in idx}); cos = sorted({i % 64 % 4 for i in idx})
        print('   k rows %s ; co %% 4 %s' % (ks, cos))
        for i in idx[:12]:
            print('      slab %d k %d co %d: in step %.9g clean %.9g' % (i // slab, (i % slab) // 64, i % 64, float(A[i]), float(Bv[i])))
    dd = (dw != dw2).nonzero()
    if dd.numel():
        print('   dw differs in %d elements; first %s' % (dd.shape[0], dd[:6].tolist()))
    del tr
