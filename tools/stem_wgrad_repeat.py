"""Developer tool (GPU; round 6, profiles/r06_stem_wgrad_race.txt): five fresh trainers, one forward + backward each at I5 / batch 36, every gradient
compared bitwise with the first run's.  ICN_TREE=<checkout> runs another tree's package; BYPASS=1 (default) skips a join in front of the stem's
weight-gradient launch where a tree has one; COPY=1 adds tools/stem_wgrad_slabs.py's copies behind the launch."""
import sys, os, importlib, ctypes
sys.path.insert(0, os.environ.get('ICN_TREE') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geniconet_amd import data, models, _lib
from geniconet_amd.train import Trainer
from geniconet_amd.ico_conv import set_weight_gradient_stream
ico = importlib.import_module('geniconet_amd.ico_conv')
R, B = 5, 36
COPY = os.environ.get('COPY') == '1'
BYPASS = os.environ.get('BYPASS', '1') == '1'
p = models.default_params('ico2ico', subdivisions=R)
x, t = data.synthetic_batch(B, R, seed=1234, device='cuda')
x = x.contiguous(memory_format=torch.channels_last)
L = _lib.lib()
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
stem_bytes = int(L.icn_conv_workspace_bytes(_lib.OP_CONV_BWD_WEIGHT, B, 3, 64, R, 1))
rows = B * 10 * (2 ** R) ** 2
saveA = torch.zeros(stem_bytes, dtype=torch.uint8, device='cuda')
saveDy = torch.zeros(rows * 64, dtype=torch.float32, device='cuda')
state = {'pending': False, 'joins_skipped': 0}
class Proxy:
    def __getattr__(self, name):
        f = getattr(L, name)
        if name != 'icn_conv_bwd_weight':
            return f
        def wrapped(xp, dy, dw, db, Bb, Cin, Cout, r, stride, mode, ws, ws_bytes, stream):
            rc = f(xp, dy, dw, db, Bb, Cin, Cout, r, stride, mode, ws, ws_bytes, stream)
            if Cin == 3 and BYPASS:
                if COPY:
                    hip.hipMemcpyAsync(saveA.data_ptr(), ws, stem_bytes, 3, stream)
                    hip.hipMemcpyAsync(saveDy.data_ptr(), dy, rows * 64 * 4, 3, stream)
                ico._pending[0] = state['pending']
            return rc
        return wrapped
proxy = Proxy()
_lib.lib = lambda: proxy
real_ws_stream = ico._wgrad_stream
def ws_wrap(dev, dests, *tensors, allow=True):
    side = real_ws_stream(dev, dests, *tensors, allow=allow)
    if not allow and BYPASS:
        state['pending'] = ico._pending[0]
        if ico._pending[0]:
            state['joins_skipped'] += 1
        ico._pending[0] = False
    return side
ico._wgrad_stream = ws_wrap
runs = []
for rep in range(5):
    tr = Trainer(p, 'cuda', seed=0)
    out = tr.net(x)
    loss = tr.criterion(out, t)
    tr.optimizer.zero_grad()
    prev = set_weight_gradient_stream(*tr._weight_gradient_mode())
    loss.backward()
    set_weight_gradient_stream(*prev)
    torch.cuda.synchronize()
    runs.append({n: q.grad.clone() for n, q in tr.model.named_parameters()})
    del tr
print('bypass %s copy %s joins skipped %d joins made %d' % (BYPASS, COPY, state['joins_skipped'], ico.wgrad_stream_counts['joins']))
for i in range(1, len(runs)):
    bad = [(n, int((runs[i][n] != runs[0][n]).sum())) for n in runs[0] if not torch.equal(runs[i][n], runs[0][n])]
    print('run %d vs 0: %s' % (i, bad if bad else 'all identical'))
