"""`python_utils` of the absent PythonFunctions checkout: the three functions the reference calls.
  get_new_name(path, ext)         run.py:428,512   a file name that does not exist yet
  writeOffMesh(path, v, f)        ico_utils.py:32  write a triangle mesh as OFF
  read_off(path) -> (v, f)        generate.py:167  read an OFF file
"""
import os

import numpy as np


def get_new_name(path, ext):
    """`path + ext`, or `path_<n> + ext` with the first free n when that exists."""
    cand, n = path + ext, 0
    while os.path.exists(cand):
        n += 1
        cand = '%s_%d%s' % (path, n, ext)
    return cand


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def writeOffMesh(path, vertices, faces):
    """OFF file at `path` ('.off' appended when missing); vertices (N, 3), faces (F, 3) -- tensors or arrays."""
    v, f = _to_numpy(vertices).reshape(-1, 3), _to_numpy(faces).reshape(-1, 3).astype(np.int64)
    if not str(path).endswith('.off'):
        path = str(path) + '.off'
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, 'w') as fh:
        fh.write('OFF\n%d %d 0\n' % (len(v), len(f)))
        for p in v:
            fh.write('%.8g %.8g %.8g\n' % tuple(p))
        for t in f:
            fh.write('3 %d %d %d\n' % tuple(t))
    return path


def read_off(path):
    """-> (vertices as a list of [x, y, z], faces as a list of index lists)."""
    with open(path) as fh:
        tokens = fh.read().split()
    if not tokens or not tokens[0].startswith('OFF'):
        raise ValueError('%s: not an OFF file' % path)
    pos = 1
    if tokens[0] != 'OFF':                                   # 'OFF123 456 0' without the line break (ModelNet quirk)
        tokens = [tokens[0][3:]] + tokens[1:]
        pos = 0
    nv, nf = int(tokens[pos]), int(tokens[pos + 1])
    pos += 3
    verts = [[float(tokens[pos + 3 * i + k]) for k in range(3)] for i in range(nv)]
    pos += 3 * nv
    faces = []
    for _ in range(nf):
        k = int(tokens[pos])
        faces.append([int(t) for t in tokens[pos + 1:pos + 1 + k]])
        pos += 1 + k
    return verts, faces
