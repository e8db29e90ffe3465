"""`torch_utils` of the absent PythonFunctions checkout: the five functions the reference calls.
  selectDevice(torch.cuda) -> 'cuda' | 'cpu'                 run.py:713
  get_git_info(path) -> str                                  run.py:715-716
  save_params(params, logDir)                                run.py:723
  image_grad_flow(named_parameters, epoch, i, cdata) -> (1, 3, H, W) tensor     run.py:266-267 (TensorBoard 'Grad' image)
  tanh2sigmoid(x) -> (x + 1) / 2                             run.py:163
"""
import json
import os
import subprocess

import torch


def selectDevice(cuda_module):
    return 'cuda' if cuda_module.is_available() else 'cpu'


def get_git_info(path):
    try:
        out = subprocess.run(['git', '-C', path, 'rev-parse', 'HEAD'], capture_output=True, text=True, timeout=10)
        return out.stdout.strip() if out.returncode == 0 and out.stdout.strip() else 'unknown'
    except (OSError, subprocess.SubprocessError):
        return 'unknown'


def save_params(params, log_dir):
    """The run's parameter dictionary as JSON under <logDir>/params.json (values that are not JSON go through repr)."""
    os.makedirs(log_dir, exist_ok=True)
    path = os.path.join(log_dir, 'params.json')
    with open(path, 'w') as fh:
        json.dump(params, fh, indent=1, sort_keys=True, default=repr)
    return path


def tanh2sigmoid(x):
    return (x + 1.0) * 0.5


def image_grad_flow(named_parameters, epoch, i, cdata=None, height=64):
    """Gradient-flow picture for TensorBoard: one column per parameter tensor, bar height = mean |grad| relative to the
    largest (red) and max |grad| (blue).  Returns a float tensor (1, 3, height, n_tensors) in [0, 1]."""
    means, maxs = [], []
    for _, p in named_parameters:
        if p.requires_grad and p.grad is not None:
            g = p.grad.detach().abs()
            means.append(float(g.mean()))
            maxs.append(float(g.max()))
    n = max(len(means), 1)
    img = torch.ones(1, 3, height, n)
    top = max(maxs) if maxs and max(maxs) > 0 else 1.0
    for k in range(len(means)):
        hm = int(round((height - 1) * min(means[k] / top, 1.0)))
        hx = int(round((height - 1) * min(maxs[k] / top, 1.0)))
        img[0, :, height - 1 - hx:, k] = torch.tensor([0.6, 0.6, 1.0])[:, None]
        img[0, :, height - 1 - hm:, k] = torch.tensor([1.0, 0.3, 0.3])[:, None]
    return img
