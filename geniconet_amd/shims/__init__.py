"""Import shims that let the reference's own `run.py` (and `losses.py`, `ico_utils.py`, `data.py`) run UNCHANGED against
this package.

The reference imports, besides `icocnn` (this repo's drop-in package at the repo root), a second un-vendored sibling checkout
`../PythonFunctions` (`torch_utils`, `python_utils`, `torchsummary`, `mesh.utils`; run.py:22-26, losses.py:6-7,
ico_utils.py:7-8) plus `natsort`, `kaolin` and `torch.utils.tensorboard`, none of which exist offline.  The modules in this
directory restate the handful of functions the reference calls from them -- signatures from the call sites, behaviour from
the names and from how the results are used; each docstring cites the call site.  They are host-side orchestration helpers
(logging, file names, mesh I/O), not part of the hot path.

    import geniconet_amd.shims as shims; shims.install()      # then `import run` from the reference checkout works
or  python tools/run_reference.py /path/to/GenIcoNet --model ico2ico --process train --quickLearn 8 ...
"""
import importlib
import importlib.abc
import importlib.util
import logging
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))


class SummaryWriter:
    """No-op stand-in for torch.utils.tensorboard.SummaryWriter (run.py:413 and the log_* functions): every `add_*` call is
    accepted and counted, nothing is written.  Used only when the real tensorboard package is not installed."""

    def __init__(self, log_dir=None, **kwargs):
        self.log_dir = log_dir
        self.calls = {}
        if log_dir:
            os.makedirs(log_dir, exist_ok=True)

    def __getattr__(self, name):
        if not name.startswith('add_'):
            raise AttributeError(name)

        def record(*args, **kwargs):
            self.calls[name] = self.calls.get(name, 0) + 1
        return record

    def flush(self):
        pass

    def close(self):
        pass


SHIM_MODULES = ('natsort', 'kaolin', 'mesh', 'python_utils', 'torch_utils', 'torchsummary')
ACTIVE = []          # top-level stand-ins that were actually served (a real package of the same name wins, see _ShimFinder)


class _ShimFinder(importlib.abc.MetaPathFinder):
    """Serves a stand-in from this directory ONLY when the ordinary import machinery has failed to find the module: it sits
    at the END of sys.meta_path, behind the path-based finder, so a real installed `natsort` / `kaolin`, or a real
    PythonFunctions checkout that the reference appended to sys.path (run.py:22-23), is imported instead of being shadowed
    by a stand-in with different metrics or conventions.  Every stand-in that is used is recorded in ACTIVE and logged."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname not in SHIM_MODULES:
            return None                                   # submodules of a served package are found through its __path__
        pkg = os.path.join(_HERE, fullname, '__init__.py')
        if os.path.exists(pkg):
            spec = importlib.util.spec_from_file_location(fullname, pkg, submodule_search_locations=[os.path.dirname(pkg)])
        else:
            spec = importlib.util.spec_from_file_location(fullname, os.path.join(_HERE, fullname + '.py'))
        if spec is not None and fullname not in ACTIVE:
            ACTIVE.append(fullname)
            logging.getLogger('geniconet_amd.shims').warning(
                'using the stand-in for %r from %s (no real module of that name is importable)', fullname, _HERE)
        return spec


def install(tensorboard_stub=None):
    """Put the repo root (the `icocnn` drop-in) in front of sys.path and register the stand-ins as a LAST-RESORT finder (a
    real package of the same name is preferred, geniconet_amd.shims.ACTIVE lists the stand-ins in use); register the
    tensorboard stand-in when `torch.utils.tensorboard` cannot be imported (tensorboard_stub=True forces it, False forbids
    it).  Idempotent."""
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    if not any(isinstance(f, _ShimFinder) for f in sys.meta_path):
        sys.meta_path.append(_ShimFinder())
    import numpy as np
    if 'Inf' not in np.__dict__:                       # run.py:342,433,459 use np.Inf, which NumPy 2.0 removed
        np.Inf = np.inf
    # run.py:356 loads its checkpoints with a bare torch.load(); they hold 'loss' as a numpy scalar (np.average, run.py:311),
    # which torch >= 2.6 refuses under its weights_only default unless the numpy scalar types are allow-listed
    try:
        import torch
        safe = [np.dtype, type(np.dtype('float64')), type(np.dtype('float32')), type(np.dtype('int64'))]
        scalar = getattr(getattr(np, '_core', None), 'multiarray', None)
        if scalar is not None and hasattr(scalar, 'scalar'):
            safe.append(scalar.scalar)
        torch.serialization.add_safe_globals(safe)
    except (ImportError, AttributeError):
        pass
    if tensorboard_stub is False:
        return
    if tensorboard_stub is None:
        try:
            importlib.import_module('torch.utils.tensorboard')
            return
        except Exception:                                  # ImportError, or tensorboard's own version checks
            pass
    import torch.utils
    mod = types.ModuleType('torch.utils.tensorboard')
    mod.SummaryWriter = SummaryWriter
    mod.__doc__ = 'geniconet_amd.shims: no-op stand-in (tensorboard is not installed)'
    sys.modules['torch.utils.tensorboard'] = mod
    torch.utils.tensorboard = mod
