"""Forward-only throughput of the full models at I5 / batch 36 (developer tool; GPU only): eval mode (running statistics: the
serving path, one fused BatchNorm + ReLU pass per layer) and BatchNorm left in training mode under no_grad (what the
reference's `--process test` does, run.py:516).  Not the round's headline metric (bench.py measures training).

  python tools/bench_inference.py [--iters 30] [--batch 36] [--subdivisions 5]
  ICN_NO_FUSED_BN=1 python tools/bench_inference.py      # BatchNorm / ReLU through torch's modules, for comparison
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geniconet_amd import _lib, data, models  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--batch', type=int, default=36)
    ap.add_argument('--subdivisions', type=int, default=5)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    for name in ('ico2ico', 'ico2ico_vae'):
        net = getattr(models, name)(models.default_params(name, subdivisions=a.subdivisions)).to(dev)
        x, _ = data.synthetic_batch(a.batch, a.subdivisions, seed=1, device=dev)
        x = x.contiguous(memory_format=torch.channels_last)
        for mode in ('eval', 'train'):
            net.train(mode == 'train')
            with torch.no_grad():
                for _ in range(5):
                    net(x)
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(a.iters):
                    net(x)
                torch.cuda.synchronize()
            dt = (time.time() - t0) / a.iters
            print('%-12s forward only, BatchNorm in %-5s mode: %.3f ms per batch of %d = %.0f meshes/s'
                  % (name, mode, dt * 1e3, a.batch, a.batch / dt), flush=True)
        net.eval()
        _lib.profile_start(4000)
        with torch.no_grad():
            net(x)
        torch.cuda.synchronize()
        print('             MFMA kernels of one eval forward: %.3f ms' % sum(e['total_ms'] for e in _lib.profile_stop()))
        _lib.raise_on_device_status(dev)


if __name__ == '__main__':
    main()
