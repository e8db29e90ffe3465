"""The bookkeeping that decides when the current stream has to wait for the weight gradients' stream under
DistributedDataParallel (geniconet_amd/ico_conv.py, mode 'bucketed'), on the CPU with the stream wait replaced
by a recorder: a wait is due exactly when the LAST gradient of a bucket is handed to autograd and something has gone to the
side stream in this backward pass -- gradients that arrive before the first side launch count too."""
import importlib

import pytest
import torch

ic = importlib.import_module('geniconet_amd.ico_conv')


@pytest.fixture
def recorder(monkeypatch):
    calls = []
    def join():
        calls.append('join')
        ic._pending[0] = False
    monkeypatch.setattr(ic, '_join_now', join)
    prev = ic.set_weight_gradient_stream('off')
    yield calls
    ic._pending[0] = False
    ic.set_weight_gradient_stream(*prev) if prev[0] != 'bucketed' else ic.set_weight_gradient_stream('off')


def test_a_wait_is_issued_when_a_buckets_last_gradient_arrives(recorder):
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(7)]
    bucket_of = {ps[0]: 'A', ps[1]: 'A', ps[2]: 'A', ps[3]: 'B', ps[4]: 'B', ps[5]: 'C', ps[6]: 'C'}
    for _pass in range(2):                                    # the countdown is rearmed for every pass
        ic.set_weight_gradient_stream('bucketed', bucket_of)
        del recorder[:]
        ic.parameter_gradient_ready(ps[0])                    # head / BatchNorm gradients: nothing on the side stream yet
        ic.parameter_gradient_ready(ps[1])
        assert recorder == []
        ic._pending[0] = True                                 # the first weight gradient of the pass went to the side stream
        ic.parameter_gradient_ready(ps[3])
        assert recorder == []                                 # B is not complete
        ic.parameter_gradient_ready(ps[2])                    # A complete -- its first two gradients arrived before the side launch
        assert recorder == ['join']
        ic.parameter_gradient_ready(ps[4])                    # B complete, but nothing went to the side stream since the last wait
        assert recorder == ['join']
        ic._pending[0] = True                                 # another weight gradient
        ic.parameter_gradient_ready(ps[5])
        assert recorder == ['join']
        ic.parameter_gradient_ready(ps[6])                    # C complete
        assert recorder == ['join', 'join']
        ic._pending[0] = True                                 # the last layer's weight gradient
        ic._backward_pass_over()                              # end of the pass: the final wait, counters rearmed
        ic._backward_pass_over()                              # (queued once per side launch: the later ones find nothing to wait for)
        assert recorder == ['join'] * 3 and not ic._pending[0]
        assert ic._buckets['left'] == {'A': 3, 'B': 2, 'C': 2}


def test_a_bucket_that_completes_before_any_side_launch_needs_no_wait_and_unknown_parameters_take_no_chances(recorder):
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(3)]
    ic.set_weight_gradient_stream('bucketed', {ps[0]: 'A', ps[1]: 'B'})
    ic.parameter_gradient_ready(ps[0])                        # A complete, nothing pending
    assert recorder == []
    ic.parameter_gradient_ready(ps[2])                        # not in the map, nothing pending
    assert recorder == []
    ic._pending[0] = True
    ic.parameter_gradient_ready(ps[2])                        # not in the map, something pending: wait
    assert recorder == ['join']
    ic._pending[0] = True
    ic.parameter_gradient_ready(ps[1])
    assert recorder == ['join', 'join']


def test_modes_other_than_bucketed_ignore_the_hooks_and_the_switch_checks_its_arguments(recorder):
    p = torch.nn.Parameter(torch.zeros(1))
    for mode in ('off', 'deferred', 'eager'):
        ic.set_weight_gradient_stream(mode)
        del recorder[:]                                       # (the switch itself settles what the previous mode left pending)
        ic._pending[0] = True
        ic.parameter_gradient_ready(p)
        assert recorder == []
    ic._pending[0] = False
    with pytest.raises(ValueError):
        ic.set_weight_gradient_stream('bucketed')
    with pytest.raises(ValueError):
        ic.set_weight_gradient_stream('both')
    assert ic.set_weight_gradient_stream('off')[0] == 'eager'


def test_a_weight_gradient_stays_on_the_current_stream_when_somebody_could_read_it_before_the_join(recorder):
    """ico_conv._readers_can_wait: the side stream is only taken for a gradient that autograd simply adopts as `.grad` (or, under
    DistributedDataParallel, finds aliasing its bucket).  Anything that makes autograd or the reducer READ it on the current
    stream -- an existing .grad, a second use of the parameter in the same pass, a non-leaf weight, a tensor hook, a lease that
    was not served from the bucket view -- keeps the launch on the current stream."""
    from geniconet_amd import _gradbuf
    ic._seen.clear()
    q = torch.nn.Parameter(torch.zeros(4))
    g = torch.zeros(4)
    ic.set_weight_gradient_stream('deferred')
    assert ic._readers_can_wait(q, g)
    q.grad = torch.zeros(4)                                   # accumulation / zero_grad(set_to_none=False)
    assert not ic._readers_can_wait(q, g)
    q.grad = None
    ic._seen[id(q)] = (ic._current_pass(), True)             # already received a gradient in this pass
    assert not ic._readers_can_wait(q, g)
    ic._seen[id(q)] = (12345, True)                          # ... in ANOTHER pass (one that raised and left this behind): ignored
    assert ic._readers_can_wait(q, g)
    ic._seen.clear()
    assert not ic._readers_can_wait(q * 2, g)                 # non-leaf: the gradient flows on
    h = q.register_hook(lambda grad: None)                    # somebody looks at it
    assert not ic._readers_can_wait(q, g)
    h.remove()
    assert ic._readers_can_wait(q, g)
    # 'bucketed': only when the lease was served from the remembered bucket view
    ic.set_weight_gradient_stream('bucketed', {q: 'A'})
    assert not ic._readers_can_wait(q, g)                     # no view remembered: the reducer would copy on the current stream
    q.grad = torch.zeros(4)
    _gradbuf.refresh([q])
    q.grad = None
    leased = _gradbuf.lease(q, (4,), torch.device('cpu'))
    assert _gradbuf.served_from_view(q, leased) and ic._readers_can_wait(q, leased)
    again = _gradbuf.lease(q, (4,), torch.device('cpu'))      # second lease in one pass: a new tensor
    assert not _gradbuf.served_from_view(q, again) and not ic._readers_can_wait(q, again)
    ic._backward_pass_over()
    assert ic._seen == {} and not ic._queued and not ic._known


def test_a_mode_switch_settles_what_an_aborted_pass_left_behind(recorder):
    """A backward pass that raised never runs the engine's end-of-pass callback: the side stream would stay un-joined, the
    callback marked as queued and the per-pass parameter set stale.  set_weight_gradient_stream (called by the Trainer around
    every backward) joins and clears."""
    ic.set_weight_gradient_stream('deferred')
    q = torch.nn.Parameter(torch.zeros(2))
    ic._pending[0] = True
    ic._seen[id(q)] = (7, True)
    ic._queued.add(7)
    ic._known.add(7)
    ic.set_weight_gradient_stream('off')
    assert recorder == ['join'] and ic._seen == {} and not ic._queued and not ic._known and not ic._pending[0]


class _FakeStream:
    index = 0

    def wait_stream(self, other):
        pass


class _FakeGrad:
    def record_stream(self, s):
        pass


def test_an_aborted_pass_without_a_mode_switch_cannot_leave_the_next_pass_unjoined(recorder, monkeypatch):
    """the mode set once and globally (ICN_WGRAD_STREAM=deferred, or set_weight_gradient_stream without the Trainer's
    try / finally), and a backward() that raises half-way.  With one global 'callback queued' flag the next pass queued no
    callback, so nothing joined before optimizer.step().  The bookkeeping is keyed on the engine's pass id instead: the next
    pass (1) first waits for what the aborted one left on the side stream, (2) ignores its parameter set, (3) queues its own
    end-of-pass callback, whose join covers its own side launches."""
    side = _FakeStream()
    monkeypatch.setattr(ic, '_side_streams', {0: side})
    monkeypatch.setattr(ic.torch.cuda, 'current_stream', lambda dev=None: _FakeStream())
    queued = []
    monkeypatch.setattr(ic, '_queue_end_of_pass', lambda tid: queued.append(tid))
    now = {'tid': 41}
    monkeypatch.setattr(ic, '_current_pass', lambda: now['tid'])
    ic.set_weight_gradient_stream('deferred')
    dev = torch.device('cuda', 0)
    a, b = torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))
    assert ic._wgrad_stream(dev, [(a, _FakeGrad())]) is side                 # pass 41: a's gradient goes to the side stream ...
    assert queued == [41] and ic._pending[0] and recorder == []
    now['tid'] = 42                                                         # ... the pass raises: no callback, nothing joined
    assert ic._wgrad_stream(dev, [(b, _FakeGrad())]) is side                 # pass 42, first side launch:
    assert recorder == ['join']                                             # (1) the leftover is waited for first
    assert queued == [41, 42]                                               # (3) this pass queues its own callback
    assert ic._wgrad_stream(dev, [(a, _FakeGrad())]) is side                 # (2) a is "unseen" in pass 42 although pass 41 listed it
    assert ic._wgrad_stream(dev, [(a, _FakeGrad())]) is None                 # ... and its second gradient of THIS pass stays put,
    assert recorder == ['join', 'join']                                     #     after a wait for the first (still in flight)
    assert ic._pending[0] is False
    assert ic._wgrad_stream(dev, [(b, _FakeGrad())], allow=False) is None    # b again, kept by the caller's own choice (nothing in flight)
    ic._pending[0] = True
    ic._backward_pass_over(42)                                              # end of pass 42
    assert recorder == ['join'] * 3 and not ic._pending[0]
    assert all(v[0] != 42 for v in ic._seen.values()) and 42 not in ic._queued and 42 not in ic._known
    ic._backward_pass_over(41)                                              # (a late callback of a pass would find nothing to do)
    assert ic._seen == {} and recorder == ['join'] * 3
