"""Developer tool: which Python lines of the training step launch the small torch kernels (copies, fills, elementwise)?

  python tools/trace_small_ops.py [--config ae|vae]
Profiles one warm step with torch.profiler (with_stack) and prints every aten op that is not one of ours, with its call site."""
import argparse
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import data, models, train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='ae')
    ap.add_argument('--R', type=int, default=5)
    ap.add_argument('--batch', type=int, default=36)
    a = ap.parse_args()
    name = 'ico2ico' if a.config == 'ae' else 'ico2ico_vae'
    tr = train.Trainer(models.default_params(name, subdivisions=a.R), 'cuda', seed=0)
    img, lbl = data.synthetic_batch(a.batch, a.R, seed=1234, device='cuda')
    for _ in range(3):
        tr.step(img, lbl)
    torch.cuda.synchronize()
    from torch.autograd.profiler import record_function
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        with record_function('PHASE forward'):
            out = tr.net(img)
            loss = tr.criterion(out, lbl)
        with record_function('PHASE backward'):
            tr.optimizer.zero_grad()
            loss.backward()
        with record_function('PHASE optimizer'):
            tr.optimizer.step()
            if tr.scheduler is not None:
                tr.scheduler.step()
        torch.cuda.synchronize()
    sites = Counter()
    dev_us = Counter()
    for e in prof.events():
        if not e.kernels or not e.name.startswith('aten::'):
            continue
        chain, q = [], e
        while q is not None:
            chain.append(q.name)
            q = q.cpu_parent
        if any(k.name.startswith('void icn') or 'icn::' in k.name for k in e.kernels):
            continue
        key = (' < '.join(chain[:4]), str(e.input_shapes)[:80])
        sites[key] += 1
        dev_us[key] += sum(k.duration for k in e.kernels)
    for k, v in sorted(sites.items(), key=lambda kv: -dev_us[kv[0]]):
        print('%3d x %7.1f us  %s   %s' % (v, dev_us[k], k[0], k[1]))


if __name__ == '__main__':
    main()
