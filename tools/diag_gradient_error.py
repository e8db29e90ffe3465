"""Where does a whole-network gradient mismatch against the float64 oracle come from?  (developer tool; GPU only)

Reproduces the teacher-forced test's state at a step of the subdivisions-3 VAE and prints, per parameter tensor, the GPU
gradient's distance from a float64 evaluation of the oracle (stream-K on and off) next to the CPU fp32 oracle's, then the same
for the gradients at the block boundaries.  Round 3 used it to establish that 1 - 4e-3 mismatches at single steps are ReLU
sign flips of pre-activations within fp32 rounding of zero (the error appears below one block, takes discrete values, and
changes with any change in rounding order, including torch's own device kernels under ICN_NO_FUSED_BN=1), not a kernel fault.
"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import models_ref
from geniconet_amd import _lib, data, models
from geniconet_amd.train import Trainer, build_criterion
name, R, B = 'ico2ico_vae', 3, 3
p = models.default_params(name, subdivisions=R)
p[name].update(lr=1e-4, lr_base=1e-4, lr_max=1e-3)
torch.manual_seed(5)
ref = getattr(models_ref, name)(R=R).train()
n = 2 ** (R - 3)
noise = [torch.randn(B, 512, 5 * n, 2 * n, generator=torch.Generator().manual_seed(70 + k)) for k in range(4)]
state = {'k': 0}
torch.randn_like = lambda t, **kw: noise[state['k']].to(device=t.device, dtype=t.dtype)
cpu = Trainer(p, 'cpu', model=ref, criterion=build_criterion(p, 'cpu'), channels_last=False)
x, t = data.synthetic_batch(B, R, seed=40)
cpu.step(x, t)
state['k'] = 1
x, t = data.synthetic_batch(B, R, seed=41)
before = copy.deepcopy(cpu.model.state_dict())
ref64 = getattr(models_ref, name)(R=R).train(); ref64.load_state_dict(before); ref64 = ref64.double()
build_criterion(p, 'cpu').double()(ref64(x.double()), t.double()).backward()
g64 = {kk: q.grad for kk, q in ref64.named_parameters()}
ref32 = getattr(models_ref, name)(R=R).train(); ref32.load_state_dict(before)
build_criterion(p, 'cpu')(ref32(x), t).backward()
g32 = {kk: q.grad for kk, q in ref32.named_parameters()}
for env in ({}, {'ICN_NO_FUSED_BN': '1'}, {'ICN_NO_UPCONV_BWD': '1'}, {'ICN_NO_PAIR': '1'}):
    pass
def run(flags):
    old = _lib.lib().icn_set_debug_flags(flags)
    net = getattr(models, name)(p); net.load_state_dict(before); net = net.cuda().to(memory_format=torch.channels_last).train()
    build_criterion(p, 'cuda')(net(x.cuda().contiguous(memory_format=torch.channels_last)), t.cuda()).backward()
    _lib.lib().icn_set_debug_flags(old)
    return {kk: q.grad.cpu().double() for kk, q in net.named_parameters()}
gg = run(0)
gn = run(128)
for kk in ('decoder.2.icobn01.bias', 'decoder.2.conv01.weight', 'decoder.2.icobn00.bias', 'decoder.2.conv00.weight', 'decoder.1.conv01.weight', 'encoder.0.weight'):
    d = float(g64[kk].norm())
    print('%-30s gpu %10.2e noSK %10.2e cpu32 %10.2e' % (kk, float((gg[kk] - g64[kk]).norm()) / d, float((gn[kk] - g64[kk]).norm()) / d, float((g32[kk].double() - g64[kk]).norm()) / d))
# ---- gradients of block outputs (module-level hooks on the blocks do not change the fused path)
def block_grads(net, xx, crit, tt, dev):
    keep = {}
    hs = []
    for nm, m in net.named_modules():
        if nm in ('decoder.0', 'decoder.1', 'decoder.2', 'encoder.3', 'encoder.4', 'reparameterize_hook', 'mu_hook', 'logvar_hook'):
            def f(mod, i, o, nm=nm):
                o.retain_grad(); keep[nm] = o
            hs.append(m.register_forward_hook(f))
    crit(net(xx), tt).backward()
    return {k: v.grad.detach().double().cpu() for k, v in keep.items()}
ref64b = getattr(models_ref, name)(R=R).train(); ref64b.load_state_dict(before); ref64b = ref64b.double()
b64 = block_grads(ref64b, x.double(), build_criterion(p, 'cpu').double(), t.double(), 'cpu')
for flags in (0, 128):
    old = _lib.lib().icn_set_debug_flags(flags)
    net = getattr(models, name)(p); net.load_state_dict(before); net = net.cuda().to(memory_format=torch.channels_last).train()
    bg = block_grads(net, x.cuda().contiguous(memory_format=torch.channels_last), build_criterion(p, 'cuda'), t.cuda(), 'cuda')
    _lib.lib().icn_set_debug_flags(old)
    print('flags', flags, {k: '%.2e' % (float((bg[k] - b64[k]).norm()) / float(b64[k].norm())) for k in b64 if k in bg})
