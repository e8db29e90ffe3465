"""Run the reference's own `run.py`, unchanged, on the MI355X path of this package.

    python tools/run_reference.py /path/to/GenIcoNet --model ico2ico --process train --dataPth <data> --logDir <log> ...

Everything after the checkout path goes to run.py as its command line.  What happens: the import shims for the reference's
absent dependencies are installed (geniconet_amd/shims: torch_utils, python_utils, torchsummary, mesh.utils, natsort, kaolin,
and a no-op torch.utils.tensorboard when tensorboard is missing), this repo's root (the `icocnn` drop-in package backed by
libicn.so) is put on sys.path, and run.py is executed as __main__ from inside its own directory (it appends relative sibling
paths and reads '.' for its git id, run.py:22,27,715).
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv):
    if len(argv) < 2 or not os.path.isfile(os.path.join(argv[1], 'run.py')):
        raise SystemExit(__doc__)
    ref = os.path.abspath(argv[1])
    import geniconet_amd.shims as shims
    shims.install()
    sys.path.insert(0, ref)
    os.chdir(ref)
    sys.argv = [os.path.join(ref, 'run.py')] + argv[2:]
    runpy.run_path(os.path.join(ref, 'run.py'), run_name='__main__')


if __name__ == '__main__':
    main(sys.argv)
