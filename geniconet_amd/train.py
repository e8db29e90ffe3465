"""Training step of the hot path: forward -> loss -> zero_grad -> backward -> Adam -> CyclicLR, per batch.

Mirrors the reference's inner loop (run.py:240-254) and its optimiser/scheduler setup (run.py:436-454):
`P2P_Loss(subdivisions, 1, 0, 0)` / `P2PKLD_Loss(..., 0.6, 0.2, 0.2, factor_KL=1)`, `Adam(lr=1e-6)`,
`CyclicLR(base_lr=1e-9, max_lr=1e-3, cycle_momentum=False)` stepped every batch.  `detect_anomaly`
(run.py:237) is a debugging aid and is off by default here; pass anomaly=True to reproduce it.

Data parallelism (new; the reference is single-device): one process per GPU, full replica per rank, per-rank
batch and BatchNorm statistics (a rank reproduces the single-GPU computation on its shard), gradients
averaged by bucketed all-reduce over RCCL overlapped with backward (torch DistributedDataParallel), identical
initial weights by broadcast from rank 0, optimiser/scheduler replicated.
"""
import contextlib
import glob
import os
import re

import torch
import torch.distributed as dist

from . import _gradbuf, losses, models, optim
from .ico_conv import parameter_gradient_ready, set_weight_gradient_stream

# DistributedDataParallel's gradient buckets, decoder first: 18.5 MB (AE) / 24 MB (VAE) of fp32 gradients.  Every bucket boundary
# is a point where the weight gradients on the second stream have to be complete (ico_conv: 'bucketed'), so fewer buckets
# cost less there -- measured at world size 1 over RCCL, meshes/s: 5 MB 4160, 10 MB 4222, 25 MB 4276, no DDP 4322 -- and more
# all-reduce time is left uncovered behind the last gradient (estimated at N = 8 over xGMI: ~0.07 / 0.10 / 0.17 ms for the
# three sizes).  10 MB is the better trade on those numbers; ICN_BUCKET_MB overrides it for tuning on a real node.
GRAD_BUCKET_MB = float(os.environ.get('ICN_BUCKET_MB', '10'))


def force_ddp_requested():
    """ICN_FORCE_DDP=1: wrap the model in DistributedDataParallel whenever a process group exists, also at world size 1.
    With backend 'nccl' (= RCCL) that runs everything N ranks would run except the wire -- communicator creation, DDP's
    reducer over the custom autograd Functions with gradient_as_bucket_view, the bucket all-reduce kernels queued on RCCL's
    stream beside the persistent conv kernels -- which is how the production communication path is exercised on a one-GPU
    box (bench.py, tests/test_gpu_ddp_nccl.py)."""
    return os.environ.get('ICN_FORCE_DDP', '') == '1'


def _host_staged_allreduce_hook(world):
    """DDP communication hook for the one-GPU REHEARSAL only (several ranks share a device and talk over gloo, which has
    no RCCL): a bucket is copied to pinned host memory, averaged by gloo's CPU all-reduce and copied back.  gloo's own
    device-tensor path is avoided on purpose: with three or more processes on one device it took seconds to minutes per
    step and once ended in a SIGSEGV right after the rendezvous.  Production (one rank per GPU, backend
    'nccl' = RCCL) never takes this path: there DDP's built-in bucketed all-reduce runs on the device."""
    staging = {}

    def hook(_state, bucket):
        buf = bucket.buffer()
        key = (bucket.index(), buf.numel())
        host = staging.get(key)
        if host is None:
            host = staging[key] = torch.empty(buf.numel(), dtype=buf.dtype, pin_memory=True)
        host.copy_(buf)                                   # blocking copy: waits for the gradients on the current stream
        fut = dist.all_reduce(host, async_op=True).get_future()

        def finish(_f):
            with torch.cuda.device(buf.device):
                host.div_(world)
                buf.copy_(host)
            return buf
        return fut.then(finish)
    return hook



def build_model(params, device):
    cls = getattr(models, params['model_name'])
    return cls(params).to(device)


def build_criterion(params, device):
    """run.py:432-444."""
    ico, loss_kind = params['ico'], params[params['model_name']]['loss']
    args = (ico['subdivisions'], ico['factor_pos'], ico['factor_nor'], ico['factor_lap'])
    lap = ico.get('laplacian', 'mean-v')            # convention of the target's Laplacian rows (not a reference key)
    if loss_kind == 'p2p':
        crit = losses.P2P_Loss(*args, laplacian=lap)
    elif loss_kind == 'p2pkld':
        crit = losses.P2PKLD_Loss(*args, 1., laplacian=lap)
    else:
        raise ValueError('loss for %s model not specified' % params['model_name'])
    return crit.to(device)


class Trainer:
    def __init__(self, params, device, model=None, criterion=None, seed=0, anomaly=False, channels_last=True, force_ddp=None,
                 check_device_status=None, graph=None):
        self.params, self.device, self.anomaly = params, torch.device(device), anomaly
        # debug mode: after every step, synchronise and raise if a kernel reported a failure (icn_device_status)
        self.check_device_status = (os.environ.get('ICN_CHECK', '') == '1') if check_device_status is None else check_device_status
        cfg = params[params['model_name']]
        torch.manual_seed(seed)
        self.model = model if model is not None else build_model(params, self.device)
        if channels_last and self.device.type == 'cuda':
            # (B,C,H,W) tensors stored (B,H,W,C) = the (B, 5, n, 2n, C) chart layout of the kernels: BatchNorm,
            # ReLU and the 1x1 head then read/write that layout directly and no transposes are needed.
            self.model = self.model.to(memory_format=torch.channels_last)
        self.criterion = criterion if criterion is not None else build_criterion(params, self.device)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.net = self.model
        if force_ddp is None:
            force_ddp = force_ddp_requested()
        if self.world > 1 or (force_ddp and dist.is_available() and dist.is_initialized()):
            ids = [self.device.index] if self.device.type == 'cuda' else None
            # rehearsal on one GPU: ranks share a device over gloo -> stage every exchange through the host
            staged = self.device.type == 'cuda' and dist.get_backend() == 'gloo'
            if staged:
                self._broadcast_initial_state_via_host()
            self.net = torch.nn.parallel.DistributedDataParallel(
                self.model, device_ids=ids, bucket_cap_mb=GRAD_BUCKET_MB, gradient_as_bucket_view=True,
                broadcast_buffers=False, init_sync=not staged)
            if staged:
                self.net.register_comm_hook(None, _host_staged_allreduce_hook(self.world))
            else:
                # torch's built-in C++ all-reduce comm hook: the bucket is all-reduced and divided by the world size ONCE -- the
                # same averaging as the reducer's default path, which however scales every parameter's gradient separately as it
                # enters its bucket (one launch per parameter: 78 per step here, 0.36 ms of GPU time).  Together with gradients
                # that the backward kernels write straight into the bucket views (_gradbuf) the reducer launches nothing per
                # parameter.  Measured at world size 1 over RCCL (I5, 36 meshes; tools/_exp/ddp_host.py, ms per step): no DDP
                # 8.78, default reducer 9.06, built-in hook + views 8.92; the Python hook of torch.distributed.algorithms
                # costs more than it saves (9.11: its future chain is completed from the host).
                hook_type = getattr(dist, 'BuiltinCommHookType', None)
                if hook_type is not None and hasattr(self.net, '_register_builtin_comm_hook'):
                    self.net._register_builtin_comm_hook(hook_type.ALLREDUCE)
            self._grads_in_buckets = os.environ.get('ICN_NO_GRAD_VIEWS', '') != '1'
            if self.device.type == 'cuda' and not staged:
                # weight gradients beside the backward chain (ico_conv: 'bucketed'): every parameter reports when its gradient
                # is handed to autograd, so that the current stream can wait for the side stream just before a bucket's
                # all-reduce is launched
                for q in self.model.parameters():
                    q.register_hook(lambda g, q=q: parameter_gradient_ready(q))
                self._param_hooks = True
        ddp = self.net is not self.model
        # public switch: weight gradients on a second HIP stream beside the backward chain (ico_conv / DESIGN 4.2b); may be
        # turned off (and on again) between steps
        self.overlap_weight_gradients = (self.device.type == 'cuda' and os.environ.get('ICN_WGRAD_STREAM', '') != 'off'
                            and (not ddp or (self._grads_in_buckets and getattr(self, '_param_hooks', False))))
        self._bucket_of, self._bucket_sig = None, None
        self.optimizer = optim.Adam(self.model.parameters(), lr=cfg['lr'])    # run.py:446 (torch.optim.Adam, step on HIP)
        self.scheduler = None
        if 'lr_base' in cfg and 'lr_max' in cfg:                                               # run.py:448-450
            self.scheduler = torch.optim.lr_scheduler.CyclicLR(self.optimizer, cfg['lr_base'], cfg['lr_max'],
                                                               cycle_momentum=False)
        self.model.train()
        self.last_output = None
        if not hasattr(self, '_grads_in_buckets'):
            self._grads_in_buckets = False
        # The step as ONE HIP graph (the loop body of run.py:244-254): ICN_GRAPH=1 / graph=True.  Replayed from the
        # third step on (the first two run eagerly: tables, caches and the allocator settle); see _step_graph for what it covers.
        self.graph = (os.environ.get('ICN_GRAPH', '') == '1') if graph is None else bool(graph)
        self._g = None
        self._eager_steps = {}                             # (img shape, lbl shape) -> eager steps run on it

    def _broadcast_initial_state_via_host(self):
        """Rank 0's parameters and buffers to every rank through CPU tensors (what DDP's init_sync does on the device)."""
        with torch.no_grad():
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                host = t.detach().cpu()
                dist.broadcast(host, src=0)
                t.copy_(host)

    def _weight_gradient_mode(self):
        """Where this step's weight gradients run (ico_conv.set_weight_gradient_stream).  Without DistributedDataParallel nobody
        looks at a gradient before backward() returns: 'deferred'.  With it: 'bucketed' once the reducer's bucket views have
        been the same for two steps in a row (it rebuilds its buckets after the first iteration), 'off' until then."""
        if not self.overlap_weight_gradients or self.anomaly:            # (detect_anomaly reads every gradient inside the pass)
            return 'off', None
        if self.net is self.model:
            return 'deferred', None
        if self._bucket_of is None:
            return 'off', None
        return 'bucketed', self._bucket_of

    def _read_bucket_map(self):
        """After a backward under DistributedDataParallel: parameter -> bucket, read off the gradient views' storages."""
        of, sig = {}, []
        for q in self.model.parameters():
            g = q.grad
            if g is None:
                self._bucket_of, self._bucket_sig = None, None
                return
            key = g.untyped_storage().data_ptr()
            of[q] = key
            sig.append((key, g.data_ptr()))
        if sig == self._bucket_sig:
            if self._bucket_of is None:
                self._bucket_of = of                      # stable for two steps: arm
        else:
            self._bucket_of, self._bucket_sig = None, sig

    # ---- the step as a HIP graph ----------------------------------------------------------------------------------------------
    GRAPH_SCALAR_SLOTS = 8          # pinned {step_size, bc2_sqrt} slots: the host may run this many replays ahead of the device

    def graph_usable(self, img=None, keep_output=False):
        """Why the graph path cannot take this step (a string), or None.  Covered: the plain trainer on one device -- forward, loss,
        backward with the weight gradients on their side stream (forked from and joined to the capture stream inside the graph),
        Adam, with CyclicLR's learning rate and Adam's bias corrections as device scalars.  Not covered (stay eager, said here):
        DistributedDataParallel (its reducer's hooks and bucket rebuilds are host logic per step), detect_anomaly, models that draw
        random numbers per step (ico2ico_vae's reparameterisation) or whose loss carries a host-side factor that changes
        (P2PKLD_Loss.update_factor), keep_output, the per-step status check."""
        if self.device.type != 'cuda':
            return 'not on a ROCm device'
        if self.net is not self.model:
            return 'DistributedDataParallel stays eager'
        if self.anomaly or self.check_device_status or keep_output:
            return 'detect_anomaly / per-step status check / keep_output need the eager step'
        if self.params[self.params['model_name']]['loss'] != 'p2p':
            return 'a loss with per-step host state (KL factor) or a model with per-step random numbers stays eager'
        if not isinstance(self.optimizer, optim.Adam):
            return 'optimizer without a capturable step'
        return None

    def _capture(self, img, lbl):
        g = {'shape': (tuple(img.shape), tuple(lbl.shape))}
        g['img'] = torch.empty_like(img)                  # static inputs: every replay reads these addresses
        g['lbl'] = torch.empty_like(lbl)
        g['img'].copy_(img)
        g['lbl'].copy_(lbl)
        g['scal'] = torch.zeros(2, dtype=torch.float32, device=self.device)
        g['pinned'] = [torch.zeros(2, dtype=torch.float32).pin_memory() for _ in range(self.GRAPH_SCALAR_SLOTS)]
        g['events'] = [None] * self.GRAPH_SCALAR_SLOTS
        g['n'] = 0
        # Nothing may keep an earlier step's autograd graph alive: its AccumulateGrad nodes belong to the stream that step ran on, and
        # the engine would order the captured backward after that stream -- a dependency outside the capture (ends in a crash in
        # hipStreamEndCapture).  The Trainer's own reference is last_output (keep_output=True).
        self.last_output = None
        self.optimizer.zero_grad(set_to_none=True)         # gradients are allocated inside the capture: static addresses
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            output = self.net(g['img'])
            loss = self.criterion(output, g['lbl'])
            prev = set_weight_gradient_stream(*self._weight_gradient_mode())
            try:
                loss.backward()
            finally:
                set_weight_gradient_stream(*prev)
            if not self.optimizer.graph_ready():
                raise RuntimeError('Trainer: the optimizer state does not fit the captured Adam step')
            self.optimizer.step_captured(g['scal'])
            g['loss'] = loss.detach()
        g['graph'] = graph
        return g

    def _step_graph(self, img, lbl):
        g = self._g
        if g is None or g['shape'] != (tuple(img.shape), tuple(lbl.shape)):
            g = self._g = self._capture(img, lbl)
        slot = g['n'] % self.GRAPH_SCALAR_SLOTS
        g['n'] += 1
        if g['events'][slot] is not None:
            g['events'][slot].synchronize()               # (only when the host is GRAPH_SCALAR_SLOTS replays ahead)
        ss, bc = self.optimizer.advance_host()
        g['pinned'][slot][0] = ss
        g['pinned'][slot][1] = bc
        g['img'].copy_(img, non_blocking=True)
        g['lbl'].copy_(lbl, non_blocking=True)
        g['scal'].copy_(g['pinned'][slot], non_blocking=True)
        ev = g['events'][slot] = g['events'][slot] or torch.cuda.Event()
        ev.record()
        g['graph'].replay()
        if self.scheduler is not None:
            self.scheduler.step()
        return g['loss']

    def step(self, img, lbl, keep_output=False):
        """One batch of run.py:244-254.  Returns the loss tensor (no host sync).  keep_output: leave the model's output
        of this batch in self.last_output (the reference's train() keeps the last one for the VAE `misc`, run.py:274-276).
        With the graph path on, the returned tensor is the graph's loss buffer: read it before the next step."""
        if self.graph:
            # a shape's first two steps run eagerly (they build the launch tables, caches and workspaces of that shape -- host-side
            # allocations and copies that must not happen inside a capture)
            key = (tuple(img.shape), tuple(lbl.shape))
            if self._eager_steps.get(key, 0) >= 2 and self.graph_usable(img, keep_output) is None:
                return self._step_graph(img, lbl)
            self._eager_steps[key] = self._eager_steps.get(key, 0) + 1
        ctx = torch.autograd.detect_anomaly() if self.anomaly else contextlib.nullcontext()
        with ctx:
            output = self.net(img)
            if keep_output:
                self.last_output = output
            loss = self.criterion(output, lbl)
            self.optimizer.zero_grad()
            prev = set_weight_gradient_stream(*self._weight_gradient_mode())
            try:
                loss.backward()
            finally:
                set_weight_gradient_stream(*prev)
            if self._grads_in_buckets:
                # after the reducer's hooks every .grad is a view into its bucket: the next backward writes there directly
                _gradbuf.refresh(self.model.parameters())
                if self.overlap_weight_gradients:
                    self._read_bucket_map()
            self.optimizer.step()
            if self.scheduler is not None:
                self.scheduler.step()
        if self.check_device_status:
            self.check_status(synchronize=True)
        return loss.detach()

    def check_status(self, synchronize=False):
        """Raise if a kernel reported an asynchronous failure (icn_device_status: e.g. a lost stream-K partner, which otherwise
        only shows as NaNs that Adam then writes into every weight).  Without `synchronize` this only reads a word in pinned
        host memory, so it is called wherever the host has just synchronised anyway -- a loss converted to a float
        (validate, fit), a checkpoint written -- and covers every kernel that ran before that point.  ICN_CHECK=1 /
        check_device_status=True additionally synchronises and checks after every step."""
        if self.device.type == 'cuda':
            from . import _lib
            _lib.raise_on_device_status(self.device, synchronize=synchronize)

    def check_status_collective(self, synchronize=True):
        """check_status on EVERY rank with one verdict for all: each rank checks its own device, the flags are
        all-reduced (MAX), and every rank raises together -- a rank that raised alone would leave the others blocked in their next
        collective until the watchdog fires.  Call it on all ranks at the same point (fit(): once per epoch, before any
        checkpoint is written).  World size 1: plain check_status."""
        failure = None
        try:
            self.check_status(synchronize=synchronize)
        except RuntimeError as e:
            failure = e
        if self.world > 1:
            on_host = dist.get_backend() == 'gloo'
            flag = torch.tensor([1.0 if failure else 0.0], device='cpu' if on_host else self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if float(flag) and failure is None:
                failure = RuntimeError('libicn: another rank reported an asynchronous kernel failure')
        if failure is not None:
            raise failure

    @torch.no_grad()
    def evaluate(self, img, lbl):
        """Forward + loss with the module in eval mode (run.py:280-296), restoring train mode after."""
        was_training = self.model.training
        self.model.eval()
        try:
            return self.criterion(self.model(img), lbl)
        finally:
            self.model.train(was_training)


# ---- checkpoints in the reference's format (run.py:317-372) ---------------------------------------------------------------
def _natural_key(path):
    """natsort-like ordering (the reference sorts its 'EB<epoch>' files with natsort, run.py:322,346)."""
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', os.path.basename(path))]


def checkpoint_path(log_dir, model_name, epoch):
    """<logDir>/savedModel/<modelName>_E<epoch>.pt; `epoch` is an int or the reference's 'B<int>' for a best model
    (run.py:327,331)."""
    return os.path.join(log_dir, 'savedModel', '%s_E%s.pt' % (model_name, epoch))


def save_checkpoint(trainer, log_dir, epoch, val_loss=None, misc=None, model_name=None):
    """Write the dict the reference's saveModel writes (run.py:330-340): model_state_dict, optimizer_state_dict, epoch
    (the integer part of 'B<int>', ico_utils.py:67-71), loss, misc.  Like the reference it never overwrites: returns the
    path, or None when the file already exists.  Rank 0 only under DDP; the state dict is the bare model's (no 'module.'
    prefix)."""
    if trainer.world > 1 and dist.get_rank() != 0:
        return None
    name = model_name or trainer.params['model_name']
    path = checkpoint_path(log_dir, name, epoch)
    if os.path.exists(path):
        return None
    os.makedirs(os.path.dirname(path), exist_ok=True)
    # Check BEFORE anything reaches the disk: a kernel-side failure (a lost stream-K partner turns its tile into NaNs)
    # must raise here, not after a poisoned file has been written that a later --resume would load.  Under DDP this is rank 0's
    # own device only and a raise here would leave the other ranks walking into their next collective: callers there run
    # Trainer.check_status_collective() on ALL ranks first (fit() does, once per epoch before it saves), after which this check
    # cannot fail.
    trainer.check_status(synchronize=True)
    # The file appears under its name only when it is complete; leftovers of a writer that was killed mid-save ('<name>.pt.tmp<pid>'
    # of another pid) are removed here.
    for stale in glob.glob(path + '.tmp*'):
        try:
            os.remove(stale)
        except OSError:
            pass
    tmp = path + '.tmp%d' % os.getpid()
    try:
        torch.save({'model_state_dict': trainer.model.state_dict(), 'optimizer_state_dict': trainer.optimizer.state_dict(),
                    'epoch': epoch if isinstance(epoch, int) else int(str(epoch)[1:]), 'loss': val_loss, 'misc': misc}, tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return path


def load_checkpoint(model, log_dir, model_name, epoch=0, optimizer=None):
    """The reference's loadModel (run.py:342-372): epoch 0 = the newest '<name>_EB<int>.pt' best model, else
    '<name>_E<epoch>.pt'; tensors are mapped to the CPU first; only the saved keys the model has are kept and then loaded
    STRICTLY (so an encoder / decoder half restores from a full model's file); the optimizer state is restored when an
    optimizer is given.  Returns the checkpoint dict (epoch, loss, misc ...) or None when no file exists."""
    if epoch == 0:
        paths = sorted(glob.glob(os.path.join(log_dir, 'savedModel', model_name + '_EB*[0-9]*.pt')), key=_natural_key)
        path = paths[-1] if paths else checkpoint_path(log_dir, model_name, epoch)
    else:
        path = checkpoint_path(log_dir, model_name, epoch)
    if not os.path.exists(path):
        return None
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    have = model.state_dict()
    model.load_state_dict({k: v for k, v in ckpt['model_state_dict'].items() if k in have})
    if optimizer is not None:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return ckpt


# ---- epoch loops (run.py:233-316) over a device-resident dataset (data.IcoDataset) -----------------------------------------
def train_epoch(trainer, dataset, batch_size, shuffle=True, generator=None, misc=None):
    """One pass of run.py:233-278 (train()): model in train mode, one Trainer.step per batch.  Returns the per-batch losses
    as one device tensor (a single host sync when the caller reads it, instead of one per iteration).  For the VAE losses
    ('p2pkld') the reference returns misc = {'trn_mean', 'trn_logvar'} = mu / logvar of the LAST batch (run.py:274-276),
    which its checkpoints carry and enc2ico_vae.createSample reads (models.py:329-332): pass a dict as `misc` to receive
    them."""
    trainer.model.train()
    want_misc = misc is not None and trainer.params[trainer.params['model_name']]['loss'] == 'p2pkld'
    losses_ = []
    for img, lbl in dataset.batches(batch_size, shuffle, generator):
        losses_.append(trainer.step(img, lbl, keep_output=want_misc))
    if want_misc and trainer.last_output is not None:
        misc['trn_mean'], misc['trn_logvar'] = trainer.last_output[1].detach(), trainer.last_output[2].detach()
        trainer.last_output = None
    return torch.stack(losses_)


@torch.no_grad()
def validate(trainer, dataset, batch_size):
    """run.py:280-315 (validate()): eval mode, no grad, the unweighted mean over batches of the criterion's total loss."""
    loss = float(torch.stack([trainer.evaluate(img, lbl) for img, lbl in dataset.batches(batch_size)]).mean())
    trainer.check_status()                    # (the float() synchronised)
    return loss


def fit(trainer, trn, val, epochs, batch_size, log_dir=None, model_name=None, seed=0, first_epoch=1, best_loss=float('inf'),
        save_epoch_freq=None):
    """The reference's epoch loop (run.py:479-496): train, validate, keep the best models, anneal the KL factor.
      * '<name>_EB<epoch>.pt' is written whenever the validation loss does not exceed the best so far, and all but the
        newest six are deleted (saveBestModel, run.py:317-329);
      * '<name>_E<epoch>.pt' every `save_epoch_freq` epochs (run.py:488-489; default: the model's 'save_epoch_freq' entry
        of params when present) and once more after the last epoch (run.py:495-496; never overwrites);
      * every checkpoint carries misc = {'trn_mean', 'trn_logvar'} of the epoch's last training batch for the VAE losses
        (run.py:274-276), None for the auto-encoder;
      * after each epoch criterion.update_factor(epoch, factor_step_size, factor_gamma) when the model's params hold both
        keys (run.py:491-493: the VAE's KL factor x 0.9 every 25 epochs).
    `epoch` counts from 1 as in the reference's calls (its loop variable + 1).  Returns a list of
    (epoch, mean training loss, validation loss)."""
    name = model_name or trainer.params['model_name']
    cfg = trainer.params[trainer.params['model_name']]
    if save_epoch_freq is None:
        save_epoch_freq = cfg.get('save_epoch_freq', 0)
    gen = torch.Generator().manual_seed(seed)
    history = []
    misc, val_loss, epoch = None, None, first_epoch - 1
    for epoch in range(first_epoch, first_epoch + epochs):
        got = {}
        trn_loss = float(train_epoch(trainer, trn, batch_size, True, gen, misc=got).mean())
        trainer.check_status()                # (the float() synchronised: covers the epoch's kernels)
        misc = got or None
        val_loss = validate(trainer, val, batch_size)
        if trainer.world > 1:
            trainer.check_status_collective(synchronize=False)      # (validate's float() synchronised this rank's device)
        history.append((epoch, trn_loss, val_loss))
        if log_dir is not None and val_loss <= best_loss:
            if trainer.world == 1 or dist.get_rank() == 0:
                old = sorted(glob.glob(os.path.join(log_dir, 'savedModel', name + '_EB*[0-9]*.pt')), key=_natural_key)
                for path in old[:max(0, len(old) - 5)]:
                    os.remove(path)
            save_checkpoint(trainer, log_dir, 'B%d' % epoch, val_loss=val_loss, misc=misc, model_name=name)
            best_loss = val_loss
        if log_dir is not None and save_epoch_freq and epoch % save_epoch_freq == 0:
            save_checkpoint(trainer, log_dir, epoch, val_loss=val_loss, misc=misc, model_name=name)
        if 'factor_step_size' in cfg and 'factor_gamma' in cfg and hasattr(trainer.criterion, 'update_factor'):
            trainer.criterion.update_factor(epoch, cfg['factor_step_size'], cfg['factor_gamma'])
    if log_dir is not None and epochs > 0:
        save_checkpoint(trainer, log_dir, epoch, val_loss=val_loss, misc=misc, model_name=name)
    return history
