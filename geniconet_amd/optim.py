"""`torch.optim.Adam` (reference run.py:446) with the whole step as one HIP launch (`icn_adam_step`, csrc/icn_optim.hip).

Same constructor, same `state_dict` (per-parameter `step`, `exp_avg`, `exp_avg_sq`: checkpoints in the reference's format
load either way), same arithmetic as torch's `_single_tensor_adam`.  torch's own multi-tensor step is ~10 kernels of 70-odd
blocks each for this model's 4.6 M parameters (0.2 ms per step); one launch with 2048-element blocks is 0.03 ms.

Anything the kernel does not cover (amsgrad, maximize, capturable / differentiable, tensor learning rates, parameters that
are not dense fp32 on a ROCm device) goes to `torch.optim.Adam.step` unchanged; `ICN_NO_HIP_ADAM=1` forces that too.
"""
import ctypes
import os

import torch

from . import _lib


class Adam(torch.optim.Adam):
    def _hip_groups(self):
        """[(group, params with a gradient)] when every group can take the HIP step, else None."""
        if os.environ.get('ICN_NO_HIP_ADAM'):
            return None
        out = []
        for group in self.param_groups:
            if group.get('amsgrad') or group.get('maximize') or group.get('capturable') or group.get('differentiable'):
                return None
            if isinstance(group['lr'], torch.Tensor) or any(isinstance(b, torch.Tensor) for b in group['betas']):
                return None
            ps = [p for p in group['params'] if p.grad is not None]
            for p in ps:
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not g.is_sparse
                        and g.dtype == torch.float32 and g.device == p.device):
                    return None
                st = self.state.get(p)
                if st and (st['step'].is_cuda or not st['exp_avg'].is_contiguous() or not st['exp_avg_sq'].is_contiguous()):
                    return None        # e.g. a state_dict written by a capturable / fused optimiser: torch's step handles it
            out.append((group, ps))
        return out

    @torch.no_grad()
    def step(self, closure=None):
        # the closure first, as torch.optim.Adam.step does: eligibility and the list of parameters with gradients must come
        # from the gradients the closure leaves behind (zero_grad(set_to_none=True) + backward inside it), not from the ones
        # that existed before it ran
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = self._hip_groups()
        if groups is None:
            super().step()                    # the closure has run: not passed on
            return loss
        L = _lib.lib()
        for group, ps in groups:
            if not ps:
                continue
            beta1, beta2 = group['betas']
            grads, steps = [], []
            for p in ps:
                st = self.state[p]
                if len(st) == 0:           # as torch.optim.Adam._init_group: step on the host, moments like the parameter
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                steps.append(st['step'])
                grads.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
            by_dev = {}
            for i, p in enumerate(ps):
                by_dev.setdefault(p.device, []).append(i)
            torch._foreach_add_(steps, 1.0)
            lr = float(group['lr'])
            for dev, idx in by_dev.items():
                n = len(idx)
                vp = ctypes.c_void_p * n
                t = [float(steps[i]) for i in idx]
                ss = (ctypes.c_float * n)(*[lr / (1.0 - beta1 ** k) for k in t])
                bc = (ctypes.c_float * n)(*[(1.0 - beta2 ** k) ** 0.5 for k in t])
                numel = (ctypes.c_size_t * n)(*[ps[i].numel() for i in idx])
                pa = vp(*[ps[i].data_ptr() for i in idx])
                ga = vp(*[grads[i].data_ptr() for i in idx])
                ma = vp(*[self.state[ps[i]]['exp_avg'].data_ptr() for i in idx])
                va = vp(*[self.state[ps[i]]['exp_avg_sq'].data_ptr() for i in idx])
                with torch.cuda.device(dev):
                    rc = L.icn_adam_step(n, pa, ga, ma, va, numel, ss, bc, float(beta1), float(beta2), float(group['eps']),
                                         float(group['weight_decay']), torch.cuda.current_stream(dev).cuda_stream)
                _lib.check(rc, 'icn_adam_step')
        return loss

    # ---- the step as part of a HIP graph (Trainer, ICN_GRAPH=1) ---------------------------------------------------------------
    def graph_ready(self):
        """One parameter group the HIP step covers, every tensor on one device with one step count: what step_captured needs."""
        groups = self._hip_groups()
        if groups is None or len(groups) != 1 or not groups[0][1]:
            return False
        ps = groups[0][1]
        if len({p.device for p in ps}) != 1 or any(not p.grad.is_contiguous() for p in ps):
            return False
        steps = {float(self.state[p]['step']) if self.state.get(p) else 0.0 for p in ps}
        return len(steps) == 1

    @torch.no_grad()
    def step_captured(self, scalars_dev):
        """Launch the step with step_size / bc2_sqrt read from `scalars_dev` (2 floats on the device): called ONCE, inside the
        capture of the training step.  The step counts are NOT advanced here (host state): advance_host does that per replay."""
        (group, ps), = self._hip_groups()
        for p in ps:
            st = self.state[p]
            if len(st) == 0:
                st['step'] = torch.tensor(0.0, dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        n = len(ps)
        vp = ctypes.c_void_p * n
        beta1, beta2 = group['betas']
        numel = (ctypes.c_size_t * n)(*[p.numel() for p in ps])
        pa = vp(*[p.data_ptr() for p in ps])
        ga = vp(*[p.grad.data_ptr() for p in ps])
        ma = vp(*[self.state[p]['exp_avg'].data_ptr() for p in ps])
        va = vp(*[self.state[p]['exp_avg_sq'].data_ptr() for p in ps])
        dev = ps[0].device
        with torch.cuda.device(dev):
            rc = _lib.lib().icn_adam_step_dev(n, pa, ga, ma, va, numel, scalars_dev.data_ptr(), float(beta1), float(beta2),
                                              float(group['eps']), float(group['weight_decay']),
                                              torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, 'icn_adam_step_dev')
        self._captured = [self.state[p]['step'] for p in ps]

    def advance_host(self):
        """Per replay: step counts + 1 (host tensors, as step() does) -> (step_size, bc2_sqrt) as step() would have passed them."""
        group = self.param_groups[0]
        beta1, beta2 = group['betas']
        torch._foreach_add_(self._captured, 1.0)
        k = float(self._captured[0])
        return float(group['lr']) / (1.0 - beta1 ** k), (1.0 - beta2 ** k) ** 0.5
