"""bench.py -- meshes/sec of the ico2ico training step (BASELINE.json metric) on N MI355X of one node.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: forward -> P2P loss -> zero_grad -> backward -> Adam -> CyclicLR
(reference run.py:244-254) on 36 synthetic I5 meshes per GPU already resident in HBM.  Weak scaling: per-GPU
batch fixed; gradients averaged by bucketed RCCL all-reduce overlapped with backward.
Rank 0 prints ONE JSON line, with
  roofline     : the dominant MFMA kernel, timed live with HIP events on its launch stream (achieved = algorithmic FLOPs per
                 launch / mean launch duration), vs 157.3 TFLOP/s dense fp32 MFMA.  The training step runs its weight
                 gradients on a second stream beside the other launches; a launch that shares the chip has no
                 roofline of its own, so the events are taken in the survey steps, which run on one stream, and the timed
                 region carries none (ICN_WGRAD_STREAM=off: one stream throughout, events over the timed region);
  cpu_baseline : the CPU restatement of the same step (oracle/) timed on this host's cores (N=1 only).
"""
import argparse
import faulthandler
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from geniconet_amd import _lib, data, models  # noqa: E402
from geniconet_amd.train import Trainer, force_ddp_requested  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, dense f32-input MFMA
# algorithmic work per mesh per TRAINING step (SURVEY.md 8d / BASELINE.md): 3 x forward conv FLOPs
TRAIN_GFLOP_PER_MESH = {('ico2ico', 5): 31.49, ('ico2ico_vae', 5): 35.46, ('ico2ico', 6): 125.96}
# fused-ideal HBM bytes per mesh per training step: every conv reads its input and writes its output once
TRAIN_MB_PER_MESH = {('ico2ico', 5): 175.9, ('ico2ico_vae', 5): 179.7, ('ico2ico', 6): 702.9}
PEAK_HBM_GBPS = 8000.0                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s
CONFIGS = {
    'ae': dict(model='ico2ico', R=5, batch=36, workload='ico2ico AE training, I5, batch 36 per GPU (BASELINE configs[1])'),
    'vae': dict(model='ico2ico_vae', R=5, batch=36, workload='ico2ico_vae training, I5, batch 36 per GPU (configs[3])'),
    'i6': dict(model='ico2ico', R=6, batch=8, workload='ico2ico AE training, I6, batch 8 per GPU (configs[4])'),
}


def usable_cores():
    """Cores this process may really use: affinity, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    """'model name' of /proc/cpuinfo (the CPU baseline states core count and CPU model)."""
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(cfg, budget_s=12.0):
    """The CPU restatement (oracle/) of the same training step on this host's cores: the only runnable 'reference
    CPU path' (upstream icocnn is absent; parity unpinned).  Headline: detect_anomaly off (faster, so conservative for the
    GPU/CPU ratio); `with_anomaly`: the same steps inside torch.autograd.detect_anomaly(), as the reference trains
    (run.py:237).  Bounded sample: the batch is sized from a 2-mesh calibration step so that 3 timed steps take about
    budget_s in each mode."""
    from geniconet_amd.train import build_criterion
    from oracle import models_ref
    cores = min(usable_cores(), int(os.environ.get('ICN_CPU_THREADS', 64)))
    torch.set_num_threads(cores)
    p = models.default_params(cfg['model'], subdivisions=cfg['R'])

    def trainer(anomaly=False):
        torch.manual_seed(0)
        net = getattr(models_ref, cfg['model'])(R=cfg['R']).train()
        return Trainer(p, 'cpu', model=net, criterion=build_criterion(p, 'cpu'), channels_last=False, anomaly=anomaly)
    tr = trainer()
    x, t = data.synthetic_batch(2, cfg['R'], seed=1234)
    tr.step(x, t)
    t0 = time.perf_counter()
    tr.step(x, t)
    per_mesh = (time.perf_counter() - t0) / 2
    steps = 3
    batch = int(max(2, min(cfg['batch'], budget_s / steps / max(per_mesh, 1e-4))))
    x, t = data.synthetic_batch(batch, cfg['R'], seed=1234)

    def timed(anomaly):
        import warnings
        tr = trainer(anomaly)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')                # detect_anomaly announces itself on every entry
            tr.step(x, t)                                   # warm-up at the sample's batch size
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(x, t)
            return batch * steps / (time.perf_counter() - t0)
    off, on = timed(False), timed(True)
    return {'value': round(off, 3), 'unit': 'meshes/s', 'cores': cores, 'cpu_model': cpu_model(), 'kind': 'port',
            'with_anomaly': round(on, 3),
            'sample': '%d timed training steps of batch %d at I%d after 1 warm-up (batch sized for ~%ds of CPU work per mode), '
                      'torch CPU restatement (oracle/), %d threads; value: detect_anomaly off, with_anomaly: on (run.py:237)'
                      % (steps, batch, cfg['R'], int(budget_s), cores)}


def main():
    faulthandler.enable()      # a native crash leaves every rank's Python stack on stderr
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='ae')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true', help='skip the per-launch HIP events (roofline = null)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py: no GPU visible; the HIP path has no CPU fallback')
    # Rehearsal switch for a 1-GPU box (never set by the driver): all ranks share device 0 and talk over gloo, which
    # exercises the same DDP wiring (bucketing, hooks on the custom autograd ops) without RCCL.
    rehearsal = os.environ.get('ICN_BENCH_REHEARSAL', '') == '1'
    if rehearsal:
        local = 0
        if world > 6:
            raise SystemExit('bench.py: the one-GPU rehearsal is limited to 6 ranks (process cap of a GPU box)')
        faulthandler.dump_traceback_later(int(os.environ.get('ICN_BENCH_WATCHDOG', 300)), exit=True)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    # ICN_FORCE_DDP=1 at N = 1 (never set by the driver): the production communication path on a one-GPU box -- an RCCL process
    # group of one rank, DistributedDataParallel around the model exactly as for N > 1, device barriers.  Everything N ranks
    # would run except the wire.
    forced = world == 1 and force_ddp_requested() and not rehearsal
    if forced:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group('nccl', device_id=device, rank=0, world_size=1)
    group = world > 1 or forced
    if world > 1:
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)      # RCCL over xGMI

    _lib.lib()
    p = models.default_params(cfg['model'], subdivisions=cfg['R'])
    tr = Trainer(p, device, seed=0)
    x, t = data.synthetic_batch(cfg['batch'], cfg['R'], seed=1234 + rank, device=device)
    x = x.contiguous(memory_format=torch.channels_last)

    def barrier():
        if group:
            if rehearsal:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    # HIP events around EVERY MFMA launch cost the step ~2.5 % (a marker packet between back-to-back kernels).  So: all
    # kernels are timed during the last warm-up steps (table of kernels, executed FLOPs per step, which kernel dominates),
    # and in the timed region only the dominant kernel carries events -- that is the `roofline` measurement.
    events = not args.no_kernel_events
    # (four survey steps -- 52 launches of the dominant kernel -- after the W warm-up steps the caller asked for and before the
    #  timed region, untimed like them: surveyed earlier, among the first steps of the process, the same launches take 3 - 4 %
    #  longer; and a roofline block must not depend on the caller asking for warm-up)
    n_survey = 4 if events else 0
    for _ in range(args.warmup):
        tr.step(x, t)
    survey = []
    overlapped = bool(getattr(tr, 'overlap_weight_gradients', False))    # weight gradients on a second stream beside the backward chain
    if n_survey:
        # The survey steps run every kernel on ONE stream: a launch's duration is a property of the kernel only when nothing
        # else shares the chip with it.  (In the timed region the weight gradients overlap the other launches; their
        # durations there are longer and sum to more than the step.)
        tr.overlap_weight_gradients = False
        torch.cuda.synchronize()
        _lib.profile_start(250 * n_survey)
        for _ in range(n_survey):
            tr.step(x, t)
        survey = _lib.profile_stop()
        tr.overlap_weight_gradients = overlapped
    dominant = max(survey, key=lambda e: e['total_ms'])['kernel'] if survey else None
    barrier()
    # One stream: the dominant kernel carries events through the timed region (the roofline measurement).  Overlapped: its
    # launches there share the chip with the weight gradients' -- no roofline quantity, and the marker packets around them
    # disturb the overlap (4163 ... 4334 meshes/s from run to run with them, 4269 ... 4322 without) -- so the timed region
    # carries no events, the kernel's own rate comes from the survey steps, and two more (untimed) overlapped steps after the
    # timed region record what its launches take when they share the chip.
    if dominant and not overlapped:
        _lib.profile_start(100 * args.steps, only=dominant)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step(x, t)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_stop() if dominant and not overlapped else []
    if dominant and overlapped:
        _lib.profile_start(200, only=dominant)
        for _ in range(2):
            tr.step(x, t)
        torch.cuda.synchronize()
        prof = _lib.profile_stop()
    el = torch.tensor([elapsed], device='cpu' if rehearsal else device, dtype=torch.float64)   # gloo: host tensors only
    if group:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el)
    final_loss = float(loss)
    _lib.raise_on_device_status(device)      # a kernel-side failure (lost stream-K partner) must not yield a bench line

    if rank == 0:
        meshes = cfg['batch'] * world * args.steps
        value = meshes / elapsed
        roofline = None
        if prof:
            dom = max(prof, key=lambda e: e['total_ms'])
            # HBM bytes per launch of that kernel from the latest committed PMC passes (tools/profile_round.sh:
            # separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH doubled per the gfx950 note)
            traffic, traffic_source = None, None
            try:
                import glob
                latest = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_per_kernel.json')))[-1]
                traffic = json.load(open(latest)).get('icn::' + dom['kernel'], {}).get('hbm_bytes_per_launch')
                if traffic is not None:
                    traffic_source = ('not measured by this run: looked up in the committed ' + os.path.relpath(latest, ROOT)
                                      + ' (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)')
            except (IndexError, OSError, ValueError):
                pass
            timed_ms = dom['total_ms'] / dom['launches']
            timed_tf = dom['total_flops'] / (dom['total_ms'] * 1e-3) / 1e12
            iso = next(e for e in survey if e['kernel'] == dom['kernel'])
            # overlapped run: the kernel's own rate comes from the survey steps (one stream); else from the timed region
            per_launch_ms = iso['total_ms'] / iso['launches'] if overlapped else timed_ms
            achieved = iso['total_flops'] / (iso['total_ms'] * 1e-3) / 1e12 if overlapped else timed_tf
            mfma_ms = sum(e['total_ms'] for e in survey)
            roofline = {
                'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                'kernel': dom['kernel'], 'launches_per_step': iso['launches'] / n_survey,
                'avg_launch_us': round(per_launch_ms * 1e3, 2),
                'measured': ('one stream: HIP events around the %d launches of this kernel in the %d survey steps, which run the weight '
                             'gradients on the main stream.  In the timed region the weight gradients run on a second stream beside the '
                             'other launches and no launch carries events; avg_launch_us_overlapped is what this kernel\'s launches take '
                             'there (2 extra untimed steps): a launch sharing the chip is not a roofline quantity, the step-level '
                             'figures below are' % (iso['launches'], n_survey)) if overlapped
                            else 'HIP events around every launch of this kernel in the timed region',
                'avg_launch_us_overlapped': round(timed_ms * 1e3, 2) if overlapped else None,
                'weight_gradients_on_second_stream': overlapped,
                'algorithmic_gflop_per_launch': round(dom['total_flops'] / dom['launches'] / 1e9, 3),
                # per-kernel figures count the FLOPs a launch executes (= algorithmic for ordinary convolutions; the composite
                # upsample+conv launches of the decoder execute 0.68 x / 0.25 x of the operators they replace), so frac <= 1
                'flops_counted': 'executed',
                # the other MFMA kernels: from the survey steps (events around every launch)
                'all_mfma_kernels': [{'kernel': e['kernel'], 'launches_per_step': e['launches'] / n_survey,
                                      'avg_launch_us': round(e['total_ms'] / e['launches'] * 1e3, 2),
                                      'tflops': round(e['total_flops'] / (e['total_ms'] * 1e-3) / 1e12, 2)} for e in survey],
                'mfma_kernels_ms_per_step': round(mfma_ms / n_survey, 3),
                'step_tflops': round(value * TRAIN_GFLOP_PER_MESH[(cfg['model'], cfg['R'])] / 1e3, 2),
                'step_frac_of_mfma_peak': round(value * TRAIN_GFLOP_PER_MESH[(cfg['model'], cfg['R'])] / 1e3
                                                / PEAK_FP32_MFMA_TFLOPS / world, 4),
                # step_tflops is ALGORITHMIC (3 x forward conv FLOPs of the reference's operator graph);
                # the MFMA launches of this implementation execute fewer (composite decoder blocks):
                'step_executed_tflops': round(sum(e['total_flops'] for e in survey) / n_survey / (elapsed / args.steps) / 1e12, 2),
                'step_executed_gflop': round(sum(e['total_flops'] for e in survey) / n_survey / 1e9, 1),
                # the HBM side of the same step (SURVEY 8d asks for both fractions; the binding one is MFMA)
                'achieved_hbm': round(value * TRAIN_MB_PER_MESH[(cfg['model'], cfg['R'])] / 1e3 / world, 1),
                'peak_hbm': PEAK_HBM_GBPS, 'unit_hbm': 'GB/s',
                'frac_hbm': round(value * TRAIN_MB_PER_MESH[(cfg['model'], cfg['R'])] / 1e3 / world / PEAK_HBM_GBPS, 4),
            }
        out = {
            'metric': 'meshes/sec training throughput, ico2ico I5 batch=36' if args.config == 'ae'
                      else 'meshes/sec training throughput, %s' % cfg['workload'],
            'value': round(value, 2), 'unit': 'meshes/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': cfg['workload'], 'per_gpu_batch': cfg['batch'], 'global_batch': cfg['batch'] * world,
                       'subdivisions': cfg['R'], 'parallelism': 'dp%d' % world + (' (DDP over RCCL forced at world size 1)' if forced else ''),
                       'final_loss': final_loss},
            'roofline': roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if group:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
