"""Condense gpurun_out/prof_<round>/ (rocprofv3 csv) into the tracked profiles/<round>_* files."""
import collections
import csv
import glob
import json
import os
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof_' + rnd)
dst = os.path.join(root, 'profiles')
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:70]


def newest_run(directory, suffix):
    """Files `<pid>_<suffix>` of the most recent rocprofv3 run under `directory`.  gpurun merges a box's output INTO the
    local gpurun_out/ and never removes anything, so a directory that was profiled twice holds both runs side by side;
    summarising across them would mix measurements of different code (it did, once)."""
    files = glob.glob(os.path.join(directory, '**', '*' + suffix), recursive=True)
    if not files:
        return []
    latest = max(files, key=os.path.getmtime)
    pid = os.path.basename(latest).split('_')[0]
    return [f for f in files if os.path.basename(f).split('_')[0] == pid and os.path.dirname(f) == os.path.dirname(latest)]


# ---- 1. kernel stats of the bench command (as it runs: weight gradients on a second stream beside the other launches, so
#         durations overlap and sum to more than the step), and of the same command with every kernel on one stream
def write_stats(sub, bench_json, out_name, note):
    stats = newest_run(os.path.join(src, sub), 'kernel_stats.csv')
    if not stats or not os.path.exists(os.path.join(src, bench_json)):
        return None
    bench = json.loads([l for l in open(os.path.join(src, bench_json)).read().strip().splitlines() if l.startswith('{')][-1])
    # (bench.py runs 4 survey steps between warm-up and timed region, and an overlapped run 2 more, untimed, behind it)
    steps_total = bench['steps'] + bench['warmup'] + 4 + (2 if (bench.get('roofline') or {}).get('weight_gradients_on_second_stream') else 0)
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, rnd + out_name), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps %d --warmup %d --no-cpu-baseline%s  (%d steps traced; '
                'bench line under the profiler: %.1f meshes/s, %.3f ms/step)\n' % (bench['steps'], bench['warmup'], note, steps_total,
                                                                               bench['value'], bench['ms_per_step']))
        f.write('kernel,calls,calls_per_step,total_ms,avg_us,min_us,max_us,percent\n')
        for r in rows:
            f.write('%s,%s,%.2f,%.3f,%.2f,%.2f,%.2f,%s\n' % (short(r['Name']).replace(',', ';'), r['Calls'], int(r['Calls']) / steps_total,
                                                           float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
                                                           float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage']))
    return stats


stats = write_stats('stats', 'bench_stats.json', '_bench_kernel_stats.csv', '')
write_stats('stats_one_stream', 'bench_stats_one_stream.json', '_bench_kernel_stats_one_stream.csv',
            '  [ICN_WGRAD_STREAM=off: every kernel on one stream]')
bench = json.loads([l for l in open(os.path.join(src, 'bench_stats.json')).read().strip().splitlines() if l.startswith('{')][-1])

# ---- 2. PMC passes: per-kernel mean counter value per dispatch
pmc = collections.defaultdict(dict)
dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(src, 'pmc_*/'))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest_run(d, 'counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in newest_run(d, 'kernel_trace.csv'):
        for r in csv.DictReader(open(f)):
            dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, cs in agg.items():
        for c, v in cs.items():
            pmc[k][c] = sum(v) / len(v)
            pmc[k]['dispatches_' + c] = len(v)
out = {}
for k, cs in pmc.items():
    if 'icn::' not in k:
        continue
    e = {'avg_us_under_pmc': round(sum(dur[k]) / len(dur[k]), 2)}
    e.update({c: v for c, v in cs.items()})
    # gfx950: FETCH_SIZE (KiB) under-reports wide coalesced reads by exactly 2x; WRITE_SIZE (KiB) is exact
    if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
        e['hbm_read_bytes_per_launch'] = cs['FETCH_SIZE'] * 1024 * 2
        e['hbm_write_bytes_per_launch'] = cs['WRITE_SIZE'] * 1024
        e['hbm_bytes_per_launch'] = e['hbm_read_bytes_per_launch'] + e['hbm_write_bytes_per_launch']
    if 'GRBM_GUI_ACTIVE' in cs:
        # GUI_ACTIVE cycles (summed over the 8 XCDs) / duration: a clock only for launches long enough that the cycles before the
        # first and after the last wave do not matter -- for a 10 us kernel the quotient comes out at 3 - 6 "GHz"
        e['clock_ghz_estimate'] = (round(cs['GRBM_GUI_ACTIVE'] / 8 / (e['avg_us_under_pmc'] * 1e-6) / 1e9, 3)
                                   if e['avg_us_under_pmc'] >= 100 else None)
    if 'TCC_HIT_sum' in cs:
        e['l2_hit_rate'] = round(cs['TCC_HIT_sum'] / (cs['TCC_HIT_sum'] + cs['TCC_MISS_sum']), 4)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs:
        e['mfma_pipe_busy_frac'] = round(cs['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (cs['GRBM_GUI_ACTIVE'] / 8), 4)
    out[k] = e
# HBM bytes of ONE training step from the counters: sum over kernels of bytes per launch x launches per step (the counter passes run
# `bench.py --steps 4 --warmup 2 --no-kernel-events` on one stream: no survey steps, so every step launches the same kernels)
try:
    pb = json.loads([l for l in open(os.path.join(src, 'pmc_FETCH_SIZE.json')).read().strip().splitlines() if l.startswith('{')][-1])
    pmc_steps = pb['steps'] + pb['warmup']
    total = 0.0
    for k, e in out.items():
        if 'hbm_bytes_per_launch' in e and 'dispatches_FETCH_SIZE' in e:
            e['calls_per_step'] = round(e['dispatches_FETCH_SIZE'] / pmc_steps, 3)
            total += e['hbm_bytes_per_launch'] * e['dispatches_FETCH_SIZE'] / pmc_steps
    out['_step_hbm_bytes'] = total
    out['_steps_under_pmc'] = pmc_steps
    out['_arith'] = 'bf16x3' if str(pb.get('arithmetic', '')).startswith('bf16x3') else 'f32'
except (OSError, IndexError, KeyError, ValueError) as e:
    print('no step total:', e)
# the kernel sources these counters were measured on (bench.py reports roofline.traffic only when the tree it runs from has the
# same hash: a kernel change without a re-profile must not report stale bytes)
sys.path.insert(0, root)
from geniconet_amd import _lib  # noqa: E402
sha_file = os.path.join(src, 'sources_sha256.txt')         # written on the box by profile_round.sh next to the raw counters
out['_kernel_sources_sha256'] = open(sha_file).read().strip() if os.path.exists(sha_file) else _lib.source_sha256()
json.dump(out, open(os.path.join(dst, rnd + '_pmc_per_kernel.json'), 'w'), indent=1, sort_keys=True)
json.dump(bench, open(os.path.join(dst, rnd + '_bench_under_rocprof.json'), 'w'), indent=1)
print('stats from', os.path.relpath(stats[0], root))
print('wrote', sorted(os.listdir(dst)))
