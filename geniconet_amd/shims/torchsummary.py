"""The customised `torchsummary` of the absent PythonFunctions checkout, as the reference calls it (run.py:427-430,511-512):
  summ = summary_string(model, input_size=<shape without batch>, device=...)   -> (text, (total, trainable), cdata)
  save_summary(summ, image_path)                                               the summary next to `image_path` (as text)
  draw_graph(model, input_shape, device, path)                                 module tree as text (graphviz is not here)
`summ[2]` is handed to torch_utils.image_grad_flow as `cdata` (run.py:485): here the list of parameter names.
No forward pass is made: the summary lists the module tree and the parameter counts.
"""
import os


def summary_string(model, input_size=None, device=None, **kwargs):
    lines = ['%-48s %-28s %12s' % ('module', 'type', 'parameters'), '-' * 90]
    total = trainable = 0
    for name, m in model.named_modules():
        own = sum(p.numel() for p in m.parameters(recurse=False))
        if own or not list(m.children()):
            lines.append('%-48s %-28s %12d' % (name or '<root>', type(m).__name__, own))
    for p in model.parameters():
        total += p.numel()
        trainable += p.numel() if p.requires_grad else 0
    lines += ['-' * 90, 'input size (without batch): %s' % (str(tuple(input_size)) if input_size is not None else None),
              'total parameters: %d   trainable: %d' % (total, trainable)]
    return '\n'.join(lines), (total, trainable), [n for n, _ in model.named_parameters()]


def save_summary(summ, image_path):
    path = os.path.splitext(image_path)[0] + '_summary.txt'
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, 'w') as fh:
        fh.write(summ[0] + '\n')
    return path


def draw_graph(model, input_shape, device, path):
    out = path + '_graph.txt'
    d = os.path.dirname(out)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(out, 'w') as fh:
        fh.write(repr(model) + '\n')
    return out
