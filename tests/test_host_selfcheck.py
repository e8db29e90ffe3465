"""Host side of libicn without a GPU: icn_host_selfcheck (every table builder and launch planner of a level, device copies
skipped), plain and under AddressSanitizer + UndefinedBehaviorSanitizer (tools/asan_host.sh; ADVICE r2 asked for the host
code paths -- table builders, launch planning, argument checks -- to be run under sanitizers on the CPU build)."""
import glob
import os
import shutil
import subprocess

import pytest

from geniconet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('r', range(5))
def test_host_selfcheck_builds_every_table(r):
    L = _lib.lib()
    for mode in (0, 1):
        n = L.icn_host_selfcheck(r, mode)
        assert n > 0, L.icn_last_error()
    assert L.icn_host_selfcheck(9, 0) == -1 and b'icn_host_selfcheck' in L.icn_last_error()


@pytest.mark.timeout(900)
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    rt = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    if not rt or shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('hipcc / the clang ASan runtime are not installed')
    env = {k: v for k, v in os.environ.items() if k not in ('LD_PRELOAD', 'ICN_LIB_PATH')}
    res = subprocess.run([os.path.join(ROOT, 'tools', 'asan_host.sh'), str(tmp_path / 'build')], env=env, capture_output=True,
                         text=True, timeout=850)
    assert res.returncode == 0 and 'asan_host: no sanitizer report' in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]
