#!/bin/bash
# Host side of libicn under AddressSanitizer + UndefinedBehaviorSanitizer, on a machine WITHOUT a GPU (the
# one-GPU data-parallel rehearsal once ended in an unexplained SIGSEGV; this rules the library's host code in or out).
# Builds a sanitizer copy of the library (host code instrumented, device code not: GPU ASan is unavailable on this pool),
# then runs (a) icn_host_selfcheck for r = 0..5, both corner modes -- every table builder and launch planner the device paths
# use, with the device copies skipped -- and (b) the CPU tests that go through the C ABI's host-only entry points.
#   tools/asan_host.sh [build-dir]        exit code 0 = no sanitizer report
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/icn_asan}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan"
cd "$ROOT/geniconet_amd/csrc"
pids=()
for f in icn_api.cpp icn_geometry.cpp icn_kernels.hip icn_bn.hip icn_loss.hip icn_optim.hip; do
    $HIPCC $FLAGS -c "$f" -o "$OUT/${f%.*}.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC $FLAGS -shared -o "$OUT/libicn.so" "$OUT"/*.o
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$ROOT"
export ICN_LIB_PATH="$OUT/libicn.so" LD_PRELOAD="$RT"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python - <<'PY'
from geniconet_amd import _lib
L = _lib.lib()
assert 'asan' in open('/proc/self/maps').read(), 'sanitizer runtime not loaded'
import os
assert os.path.samefile(_lib.LIB_PATH, os.environ['ICN_LIB_PATH'])
for r in range(6):
    for mode in (0, 1):
        n = L.icn_host_selfcheck(r, mode)
        assert n > 0, (r, mode, L.icn_last_error())
        print('icn_host_selfcheck r=%d mode=%d: %d elements' % (r, mode, n), flush=True)
PY
python -m pytest tests/test_abi.py tests/test_tables_vs_oracle.py tests/test_upconv_tables.py tests/test_stream_k_plan.py \
    tests/test_geometry_known_answers.py -x -q -m "not gpu" -p no:cacheprovider
echo "asan_host: no sanitizer report"
