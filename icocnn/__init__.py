"""Drop-in for the `icocnn` package GenIcoNet imports from its sibling checkout ../IcosahedralCNN
(reference models.py:4-6, losses.py:4-5, run.py:27-28, generate.py:10-11), backed by the MI355X HIP path.

Put the directory that contains this package on sys.path (the reference appends '../IcosahedralCNN/'), e.g. by
symlinking it there.  `import icocnn` alone makes `icocnn.utils.ico_geometry` resolvable (run.py:28,144).
"""
from . import ico_conv, utils  # noqa: F401
from .utils import ico_geometry  # noqa: F401
