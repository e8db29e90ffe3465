"""bench.py -- meshes/sec of the ico2ico training step (BASELINE.json metric) on N MI355X of one node.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...        (no WORLD_SIZE in the environment: starts the N ranks itself, see launch_ranks)

A step = one pass of the hot path over one batch: forward -> P2P loss -> zero_grad -> backward -> Adam -> CyclicLR
(reference run.py:244-254) on 36 synthetic I5 meshes per GPU already resident in HBM.  Weak scaling: per-GPU
batch fixed; gradients averaged by bucketed RCCL all-reduce overlapped with backward.
Rank 0 prints ONE JSON line, with
  roofline     : the dominant MFMA kernel, timed live with HIP events on its launch stream (achieved = algorithmic FLOPs per
                 launch / mean launch duration), vs 157.3 TFLOP/s dense fp32 MFMA.  The training step runs its weight
                 gradients on a second stream beside the other launches; a launch that shares the chip has no
                 roofline of its own, so the events are taken in the survey steps, which run on one stream, and the timed
                 region carries none (ICN_WGRAD_STREAM=off: one stream throughout, events over the timed region);
  cpu_baseline : the CPU restatement of the same step (oracle/) timed on this host's cores (N=1 only);
  also         : (N=1 only) BASELINE configs 4 and 5 -- ico2ico_vae at I5 / batch 36, ico2ico at I6 / batch 8 -- timed in the same
                 process after the headline (model freed in between), same --steps / --warmup; the headline's fields, timed region
                 and metric are not affected.
"""
import argparse
import faulthandler
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# torch and the package are imported by run(), not here: the launcher path of `python bench.py --gpus N` (launch_ranks) must
# start its children before anything in this process could touch the GPU, and needs the standard library only.

PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, dense f32-input MFMA
PEAK_BF16_MFMA_TFLOPS = 2500.0         # same guide: dense bf16 MFMA (16 x the f32-input rate; the 5 PF headline figure is 2:1 sparsity)
B3_PRODUCTS = 6                        # bf16 MFMA products per fp32 product of the three-way split (csrc/icn_kernels.hip, ARITH = 1)
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
    import torch
    from geniconet_amd import data, models
    from geniconet_amd.train import Trainer, build_criterion
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


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process becomes a launcher.  It starts N fresh
    interpreters of this same file -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT (a free port) set, one
    rank per GPU -- exactly what `python -m torch.distributed.run --nproc-per-node N` would start, waits for them, and exits
    with the first non-zero code (the others are then stopped by PID).  The launcher never imports torch and never touches the
    GPU: the children are new processes, not re-execs of one that initialised HIP.  stdout / stderr are inherited, so rank 0's
    one JSON line is this command's one JSON line."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), ICN_BENCH_LAUNCHER='self')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this pool's hosts
        env.setdefault('OMP_NUM_THREADS', '1')                   # as torch.distributed.run sets it
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc, live = 0, list(procs)
    # A SIGTERM / SIGHUP to the launcher alone (a driver's timeout kills just this pid) must not leave N ranks holding their GPUs
    # inside an RCCL collective: the handlers only set the exit code; the loop below then stops the exact child PIDs.
    stop = {'sig': 0}

    def on_signal(signum, _frame):
        stop['sig'] = signum
    for sg in (signal.SIGTERM, signal.SIGHUP):
        signal.signal(sg, on_signal)
    try:
        while live and rc == 0:
            if stop['sig']:
                rc = 128 + stop['sig']
                sys.stderr.write('bench.py: launcher got signal %d; stopping %d rank(s)\n' % (stop['sig'], len(live)))
                break
            time.sleep(0.2)
            for q in list(live):
                code = q.poll()
                if code is not None:
                    live.remove(q)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 128 - code
                        sys.stderr.write('bench.py: rank %d (pid %d) exited with %d; stopping the other ranks\n' % (procs.index(q), q.pid, code))
    except KeyboardInterrupt:
        rc = 130
    for q in live:                                               # exact PIDs this launcher started, nothing by pattern
        q.send_signal(signal.SIGTERM)
    deadline = time.time() + 10
    for q in live:
        try:
            q.wait(max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            q.kill()
            q.wait()
    return rc


def measure(cfg, args, ctx, headline, arith=None):
    """W warm-up steps, (survey steps), then EXACTLY K timed steps of configuration `cfg` between barrier + synchronize on both
    sides; MAX over ranks.  Returns the fields of the bench line that depend on the measurement.  headline=False (the `also`
    entries) skips the two extra overlapped steps that only feed avg_launch_us_overlapped.  arith: 'f32' / 'bf16x3' for this
    measurement only (the library's mode is restored)."""
    import torch
    import torch.distributed as dist
    from geniconet_amd import _lib, data, models
    from geniconet_amd.train import Trainer
    device, rank, world, group, rehearsal, local = (ctx[k] for k in ('device', 'rank', 'world', 'group', 'rehearsal', 'local'))
    prev_arith = _lib.set_arith(arith) if arith else None
    try:
        return _measure(cfg, args, ctx, headline)
    finally:
        if prev_arith:
            _lib.set_arith(prev_arith)


def _measure(cfg, args, ctx, headline):
    import torch
    import torch.distributed as dist
    from geniconet_amd import _lib, data, models
    from geniconet_amd.train import Trainer
    device, rank, world, group, rehearsal, local = (ctx[k] for k in ('device', 'rank', 'world', 'group', 'rehearsal', 'local'))
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
    if dominant and overlapped and headline:
        _lib.profile_start(200, only=dominant)
        for _ in range(2):
            tr.step(x, t)
        torch.cuda.synchronize()
        prof = _lib.profile_stop()
    per_rank = [elapsed]
    if group:
        el = torch.tensor([elapsed], device='cpu' if rehearsal else device, dtype=torch.float64)   # gloo: host tensors only
        gathered = [torch.zeros_like(el) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, el)
        per_rank = [float(g) for g in gathered]
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el)
    final_loss = float(loss)
    _lib.raise_on_device_status(device)      # a kernel-side failure (lost stream-K partner) must not yield a bench line
    ddp = tr.net is not tr.model                                      # DistributedDataParallel around the model
    # Evidence that the gradient exchange really happened: the replicas start from rank 0's weights, see DIFFERENT data (seed
    # 1234 + rank) and must still hold bit-identical weights after W + K steps -- true only if every step's gradients were averaged
    # over all ranks.  One float64 checksum per rank, gathered.
    replicas_identical = None
    if group and ddp:
        cs = torch.stack([q.detach().double().sum() for q in tr.model.parameters()]).sum().reshape(1)
        cs = cs.cpu() if rehearsal else cs
        sums = [torch.zeros_like(cs) for _ in range(dist.get_world_size())]
        dist.all_gather(sums, cs)
        replicas_identical = bool(all(float(v) == float(sums[0]) for v in sums))
    del tr, x, t, loss
    import gc
    gc.collect()
    # (NO torch.cuda.empty_cache() here: handing the pool back to the driver between configurations made the caching allocator fall,
    #  one run in five, into a state in which every step of the NEXT configuration blocked the host in hipMalloc / hipFree -- 25 to
    #  83 ms per step instead of 8, enqueue-bound: tools/_exp/also_repro.py, round 5.  The three configurations together reserve 48 GB
    #  of the 288.)
    if rank != 0:
        return None

    key = (cfg['model'], cfg['R'])
    value = cfg['batch'] * world * args.steps / elapsed
    roofline = None
    if survey:
        dom_name = dominant
        iso = next(e for e in survey if e['kernel'] == dom_name)
        timed = next((e for e in prof if e['kernel'] == dom_name), None)
        timed_ms = timed['total_ms'] / timed['launches'] if timed else None
        # overlapped run: the kernel's own rate comes from the survey steps (one stream); else from the timed region
        src = iso if (overlapped or not timed) else timed
        per_launch_ms = src['total_ms'] / src['launches']
        equiv = src['total_flops'] / (src['total_ms'] * 1e-3) / 1e12       # fp32 multiply-adds of the operator per second
        # A launch on the three-way bf16 split executes B3_PRODUCTS bf16 MFMA products per fp32 product: its roofline is the bf16
        # MFMA peak and what it achieves there is the EXECUTED bf16 FLOP rate; the fp32-equivalent rate is reported beside it.
        split = is_split_kernel(dom_name)
        achieved = equiv * B3_PRODUCTS if split else equiv
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        traffic, traffic_source = committed_traffic(dom_name)
        # (the committed counter passes are runs of the HEADLINE configuration: no step total for the others)
        step_bytes, step_bytes_source = committed_step_traffic() if key == ('ico2ico', 5) else (None, 'the committed counter passes ran the headline configuration only')
        ideal_bytes = TRAIN_MB_PER_MESH[key] * 1e6 * cfg['batch']
        mfma_ms = sum(e['total_ms'] for e in survey)
        # executed FLOPs of one step by arithmetic: fp32-equivalent FLOPs of the split launches run as 6 x as many bf16 FLOPs
        ex_split = sum(e['total_flops'] for e in survey if is_split_kernel(e['kernel'])) / n_survey
        ex_exact = sum(e['total_flops'] for e in survey if not is_split_kernel(e['kernel'])) / n_survey
        step_s = elapsed / args.steps
        roofline = {
            'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
            'frac': round(achieved / peak, 4), 'traffic': traffic, 'traffic_source': traffic_source,
            'arithmetic': ('bf16x3 split: %d v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulate; achieved / peak count '
                           'EXECUTED bf16 FLOPs against the dense bf16 MFMA peak' % B3_PRODUCTS) if split
                          else 'exact fp32 MFMA (v_mfma_f32_32x32x2_f32)',
            # why frac stops where it does (not measured by this run: ladder knock-outs with the CU clock read inside the K-loop)
            'what_bounds_it': ('the power limit: in cycles the K-step keeps the matrix pipe 75 % busy, but the CUs run it at 1.62-1.75 GHz '
                               '(the peak is quoted at 2.4); K-steps are 86 % of a launch -- profiles/r06_ladder_b3_knockouts.txt') if split else None,
            'fp32_equivalent_tflops': round(equiv, 2),
            'fp32_equivalent_frac_of_fp32_mfma_peak': round(equiv / PEAK_FP32_MFMA_TFLOPS, 4),
            'kernel': dom_name, 'launches_per_step': iso['launches'] / n_survey,
            'avg_launch_us': round(per_launch_ms * 1e3, 2),
            'measured': ('one stream: HIP events around the %d launches of this kernel in the %d survey steps, which run the weight '
                         'gradients on the main stream.  In the timed region the weight gradients run on a second stream beside the '
                         'other launches and no launch carries events; avg_launch_us_overlapped is what this kernel\'s launches take '
                         'there (2 extra untimed steps): a launch sharing the chip is not a roofline quantity, the step-level '
                         'figures below are' % (iso['launches'], n_survey)) if overlapped
                        else 'HIP events around every launch of this kernel in the timed region',
            'avg_launch_us_overlapped': round(timed_ms * 1e3, 2) if (overlapped and timed) else None,
            'weight_gradients_on_second_stream': overlapped,
            'algorithmic_gflop_per_launch': round(iso['total_flops'] / iso['launches'] / 1e9, 3),
            # per-kernel figures count the FLOPs a launch executes (= algorithmic for ordinary convolutions; the composite
            # upsample+conv launches of the decoder execute 0.68 x / 0.25 x of the operators they replace), so frac <= 1
            'flops_counted': 'executed',
            # the other MFMA kernels: from the survey steps (events around every launch)
            'all_mfma_kernels': [{'kernel': e['kernel'], 'launches_per_step': e['launches'] / n_survey,
                                  'avg_launch_us': round(e['total_ms'] / e['launches'] * 1e3, 2),
                                  'tflops': round(e['total_flops'] / (e['total_ms'] * 1e-3) / 1e12, 2)} for e in survey],
            'mfma_kernels_ms_per_step': round(mfma_ms / n_survey, 3),
            'step_tflops': round(value * TRAIN_GFLOP_PER_MESH[key] / 1e3, 2),
            # algorithmic fp32 FLOPs of the reference's operator graph over the fp32 MFMA peak: an ALGORITHMIC ratio (this build executes
            # 57 % of those FLOPs, part of them as bf16 products), > 1 is possible and is not a roofline fraction -- the next one is
            'step_algorithmic_over_fp32_mfma_peak': round(value * TRAIN_GFLOP_PER_MESH[key] / 1e3 / PEAK_FP32_MFMA_TFLOPS / world, 4),
            # time the step's EXECUTED matrix work would take at the peaks (bf16 products at the bf16 peak, exact products at the fp32
            # peak) over the step's time: the whole-step MFMA roofline fraction, <= 1 by construction
            'step_frac_of_mfma_peak': round((ex_split * B3_PRODUCTS / (PEAK_BF16_MFMA_TFLOPS * 1e12) + ex_exact / (PEAK_FP32_MFMA_TFLOPS * 1e12)) / step_s, 4),
            'step_executed_bf16_tflops': round(ex_split * B3_PRODUCTS / step_s / 1e12, 2),
            'step_executed_exact_fp32_tflops': round(ex_exact / step_s / 1e12, 2),
            # HBM bytes of ONE step from the counters (sum over every kernel of bytes per launch x launches per step in the committed,
            # hash-matched PMC passes) against SURVEY 8(d)'s fused-ideal bytes: the wasted-traffic ratio
            'step_hbm_bytes_counters': step_bytes, 'step_hbm_bytes_fused_ideal': round(ideal_bytes),
            'step_hbm_traffic_over_ideal': round(step_bytes / ideal_bytes, 2) if step_bytes else None,
            'step_hbm_bytes_source': step_bytes_source,
            # step_tflops is ALGORITHMIC (3 x forward conv FLOPs of the reference's operator graph);
            # the MFMA launches of this implementation execute fewer (composite decoder blocks):
            'step_executed_tflops': round(sum(e['total_flops'] for e in survey) / n_survey / (elapsed / args.steps) / 1e12, 2),
            'step_executed_gflop': round(sum(e['total_flops'] for e in survey) / n_survey / 1e9, 1),
            # the HBM side of the same step (SURVEY 8d asks for both fractions; the binding one is MFMA)
            'achieved_hbm': round(value * TRAIN_MB_PER_MESH[key] / 1e3 / world, 1),
            'peak_hbm': PEAK_HBM_GBPS, 'unit_hbm': 'GB/s',
            'frac_hbm': round(value * TRAIN_MB_PER_MESH[key] / 1e3 / world / PEAK_HBM_GBPS, 4),
        }
    return {'value': round(value, 2), 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'final_loss': final_loss,
            'roofline': roofline, 'ddp': ddp, 'replicas_identical': replicas_identical,
            'rank_ms_per_step': {'min': round(min(per_rank) / args.steps * 1e3, 3), 'max': round(max(per_rank) / args.steps * 1e3, 3)}}


def is_split_kernel(name):
    """Kernels of the three-way bf16 split (the ARITH = 1 instantiations: k_conv_b3*, k_wgrad*<.., 1>)."""
    return name.startswith('k_conv_b3') or name.endswith(', 1>')


def _latest_pmc_summary():
    """(summary dict, relative path, reason-or-None): the newest committed profiles/r*_pmc_per_kernel.json, usable only when it was
    measured on THESE kernel sources with the in-tree library."""
    import glob
    from geniconet_amd import _lib
    if _lib.build_info()['overridden']:
        return None, None, 'not reported: ICN_LIB_PATH overrides the in-tree library, the committed counters belong to the in-tree build'
    try:
        latest = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_per_kernel.json')))[-1]
        summary = json.load(open(latest))
    except (IndexError, OSError, ValueError):
        return None, None, 'no committed profiles/r*_pmc_per_kernel.json'
    rel = os.path.relpath(latest, ROOT)
    have, want = summary.get('_kernel_sources_sha256'), _lib.source_sha256()
    if have != want:
        return None, rel, ('not reported: %s was measured on kernel sources %s, this tree has %s -- re-run tools/profile_round.sh'
                           % (rel, (have or 'unrecorded')[:16], want[:16]))
    return summary, rel, None


def committed_step_traffic():
    """HBM bytes of one training step of the headline configuration from the committed counter passes (tools/profile_summary.py:
    `_step_hbm_bytes` = sum over kernels of hbm_bytes_per_launch x calls per step, under the arithmetic recorded in `_arith`), or
    (None, reason) -- same rule as committed_traffic."""
    from geniconet_amd import _lib
    summary, rel, why = _latest_pmc_summary()
    if summary is None:
        return None, why
    if '_step_hbm_bytes' not in summary:
        return None, rel + ' carries no step total'
    if summary.get('_arith') and summary['_arith'] != _lib.get_arith():
        return None, 'not reported: %s was measured under arithmetic %s' % (rel, summary['_arith'])
    return round(summary['_step_hbm_bytes']), ('not measured by this run: committed ' + rel + ' (separate rocprofv3 --pmc FETCH_SIZE / '
                                                'WRITE_SIZE passes of `bench.py --steps 4 --warmup 2`, one stream)')


def committed_traffic(kernel):
    """HBM bytes per launch of `kernel` from the latest committed PMC passes (tools/profile_round.sh: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command, FETCH doubled per the gfx950 note) -- but only when that profile was taken on THESE
    kernel sources: tools/profile_summary.py records the sha256 of geniconet_amd/csrc/ in the summary, and a summary of other
    sources yields traffic = None with the reason (a kernel change without a re-profile must not report stale bytes)."""
    from geniconet_amd import _lib
    summary, rel, why = _latest_pmc_summary()
    if summary is None:
        return None, why
    want = _lib.source_sha256()
    traffic = summary.get('icn::' + kernel, {}).get('hbm_bytes_per_launch')
    if traffic is None:
        return None, 'kernel not in ' + rel
    return traffic, ('not measured by this run: looked up in the committed ' + rel + ' (separate rocprofv3 --pmc FETCH_SIZE / '
                     'WRITE_SIZE passes of this command on kernel sources ' + want[:16] + ')')


def run(args):
    import torch
    import torch.distributed as dist
    from geniconet_amd import _lib
    from geniconet_amd import train as icn_train
    cfg = CONFIGS[args.config]
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d in the environment (start it as `python bench.py --gpus N` with no '
                         'WORLD_SIZE set, or with torch.distributed.run --nproc-per-node N)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py: no GPU visible; the HIP path has no CPU fallback')
    # Rehearsal switch for a 1-GPU box (never set by the driver): all ranks share device 0 and talk over gloo, which
    # exercises the same DDP wiring (bucketing, hooks on the custom autograd ops) without RCCL.
    rehearsal = os.environ.get('ICN_BENCH_REHEARSAL', '') == '1'
    visible = torch.cuda.device_count()
    if rehearsal:
        local = 0
        if world > 6:
            raise SystemExit('bench.py: the one-GPU rehearsal is limited to 6 ranks (process cap of a GPU box)')
        faulthandler.dump_traceback_later(int(os.environ.get('ICN_BENCH_WATCHDOG', 300)), exit=True)
    elif local >= visible:
        raise SystemExit('bench.py: rank %d wants GPU %d but only %d device(s) are visible' % (rank, local, visible))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    # ICN_FORCE_DDP=1 at N = 1 (never set by the driver): the production communication path on a one-GPU box -- an RCCL process
    # group of one rank, DistributedDataParallel around the model exactly as for N > 1, device barriers.  Everything N ranks
    # would run except the wire.
    forced = world == 1 and icn_train.force_ddp_requested() and not rehearsal
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
    if args.arith:
        _lib.set_arith(args.arith)
    arith = _lib.get_arith()
    ctx = dict(device=device, rank=rank, world=world, group=group, rehearsal=rehearsal, local=local)
    head = measure(cfg, args, ctx, headline=True)

    out = None
    if rank == 0:
        out = {
            'metric': 'meshes/sec training throughput, ico2ico I5 batch=36' if args.config == 'ae'
                      else 'meshes/sec training throughput, %s' % cfg['workload'],
            'value': head['value'], 'unit': 'meshes/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            # tensors, accumulation and results are fp32 in both modes; 'bf16x3' forms every fp32 product of the stream-K convolution /
            # dense GEMM launches from three bf16 pieces per operand (24 significand bits, exact) -- fp32-grade: its error against
            # float64 is within 2 x the exact kernels' (tests/test_gpu_arith.py); the exact-fp32 run of the same step is in also[]
            'arithmetic': ('bf16x3 split (%d bf16 MFMA products per fp32 product), fp32 accumulate, in every convolution, data-gradient, '
                           'dense-GEMM and weight-gradient launch of the stream-K / masked classes; the three small launches outside them '
                           '(k_conv_dma<64, 64>, k_gather_gemm<64, 64>: 0.08 ms of the step) on exact fp32 MFMA' % B3_PRODUCTS) if arith == 'bf16x3'
                          else 'exact fp32 MFMA',
            'library': _lib.build_info(),
            'config': {'workload': cfg['workload'], 'per_gpu_batch': cfg['batch'], 'global_batch': cfg['batch'] * world,
                       'subdivisions': cfg['R'], 'parallelism': 'dp%d' % world + (' (DDP over RCCL forced at world size 1)' if forced else ''),
                       'final_loss': head['final_loss']},
            'roofline': head['roofline'],
            # what the ranks saw (lets a reader verify that N ranks really met over the backend named)
            'distributed': {'n_ranks_seen': dist.get_world_size() if group else 1,
                            'backend': dist.get_backend() if group else None,
                            'visible_devices': visible, 'ddp': head['ddp'],
                            'bucket_mb': icn_train.GRAD_BUCKET_MB if head['ddp'] else None,
                            'rank_ms_per_step': head['rank_ms_per_step'],
                            'replicas_identical_after_run': head['replicas_identical'],
                            'launcher': os.environ.get('ICN_BENCH_LAUNCHER') or ('torch.distributed.run' if 'TORCHELASTIC_RUN_ID' in os.environ
                                                                                  else 'none')},
        }
    # BASELINE configs 4 and 5 behind the headline, same process, same --steps / --warmup (N = 1; the scaling runs stay short)
    if world == 1 and not forced and args.config == 'ae' and not args.no_also:
        try:
            also = []
            if arith != 'f32':
                # the headline step on the exact fp32 kernels (ICN_ARITH=f32), same process, same steps
                m = measure(cfg, args, ctx, headline=False, arith='f32')
                r = m['roofline'] or {}
                also.append({'workload': cfg['workload'] + ' -- exact fp32 MFMA arithmetic (ICN_ARITH=f32)', 'value': m['value'], 'unit': 'meshes/s',
                             'ms_per_step': m['ms_per_step'],
                             'roofline': {'kernel': r.get('kernel'), 'frac': r.get('frac'), 'achieved': r.get('achieved'), 'peak': r.get('peak'),
                                          'avg_launch_us': r.get('avg_launch_us')},
                             'step_executed_tflops': r.get('step_executed_tflops'), 'step_tflops': r.get('step_tflops'),
                             'final_loss': m['final_loss']})
            for name in ('vae', 'i6'):
                c = CONFIGS[name]
                m = measure(c, args, ctx, headline=False)
                r = m['roofline'] or {}
                # Plausibility guard for these secondary entries only (the headline is timed exactly once): the MFMA kernels' one-stream
                # time per step (survey) is a floor of the step; a timed region far above it means something other than the device held
                # the steps up (round 5: the host, blocked in the allocator after an empty_cache() between configurations -- removed,
                # see measure()).  Such a region is measured a second time and BOTH are reported.
                # The entry's value is ALWAYS the first measurement (timed once, like the headline); a second one is reported beside
                # it, flagged, never substituted (a best-of-two for guarded entries only would bias them).
                runs, suspect = [m['ms_per_step']], False
                if r.get('mfma_kernels_ms_per_step') and m['ms_per_step'] > 1.8 * r['mfma_kernels_ms_per_step']:
                    suspect = True
                    runs.append(measure(c, args, ctx, headline=False)['ms_per_step'])
                also.append({'workload': c['workload'], 'value': m['value'], 'unit': 'meshes/s', 'ms_per_step': m['ms_per_step'],
                             'timed_regions_ms_per_step': runs, 'suspect_first_region': suspect,
                             'roofline': {'kernel': r.get('kernel'), 'frac': r.get('frac'), 'achieved': r.get('achieved'), 'peak': r.get('peak'),
                                          'avg_launch_us': r.get('avg_launch_us')},
                             'step_executed_tflops': r.get('step_executed_tflops'), 'step_tflops': r.get('step_tflops'),
                             'final_loss': m['final_loss']})
            out['also'] = also
        except Exception as e:                            # the secondary entries must never cost the headline its line
            import traceback
            traceback.print_exc()
            out['also'] = None
            out['also_error'] = '%s: %s' % (type(e).__name__, e)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if group:
        dist.destroy_process_group()


def main():
    faulthandler.enable()      # a native crash leaves every rank's Python stack on stderr
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='ae')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true', help='skip the per-launch HIP events (roofline = null)')
    ap.add_argument('--no-also', action='store_true', help='skip the vae / i6 configurations behind the headline (N = 1)')
    ap.add_argument('--arith', choices=['f32', 'bf16x3'], default=None,
                    help="arithmetic of the channel-mixing contraction (default: the library's, i.e. ICN_ARITH or bf16x3)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    run(args)


if __name__ == '__main__':
    main()
