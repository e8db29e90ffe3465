"""geniconet_amd._gradbuf: where a parameter's gradient is written (the remembered .grad of the previous backward -- under
DistributedDataParallel a view into a communication bucket -- so that the reducer has nothing to copy)."""
import torch

from geniconet_amd import _gradbuf


def test_lease_hands_out_the_remembered_view_once_per_backward():
    p = torch.nn.Parameter(torch.zeros(4, 3))
    bucket = torch.zeros(100)
    p.grad = bucket[10:22].view(4, 3)                       # as the reducer leaves it after a backward
    _gradbuf.refresh([p])
    p.grad = None                                           # optimizer.zero_grad(set_to_none=True)
    a = _gradbuf.lease(p, (4, 3), p.device)
    assert a.data_ptr() == bucket[10:22].data_ptr() and a is not getattr(p, '_icn_grad_view')   # same memory, a new tensor object
    a.fill_(7.0)
    assert float(bucket[10]) == 7.0 and float(bucket[9]) == 0.0 and float(bucket[22]) == 0.0
    b = _gradbuf.lease(p, (4, 3), p.device)                 # a second use of the parameter in the same backward: new memory
    assert b.data_ptr() != a.data_ptr()
    _gradbuf.refresh([p])                                   # .grad is None here: the old view is kept, the lease is returned
    c = _gradbuf.lease(p, (4, 3), p.device)
    assert c.data_ptr() == a.data_ptr()


def test_lease_falls_back_to_new_memory_when_the_view_does_not_fit():
    p = torch.nn.Parameter(torch.zeros(4, 3))
    p.grad = torch.zeros(4, 3)
    _gradbuf.refresh([p])
    keep = p.grad
    assert _gradbuf.lease(p, (4, 3), p.device).data_ptr() != keep.data_ptr()      # .grad still set (set_to_none=False): accumulate
    p.grad = None
    assert _gradbuf.lease(p, (3, 4), p.device).data_ptr() != keep.data_ptr()      # another shape
    assert _gradbuf.lease(p, (4, 3), p.device, dtype=torch.float64).dtype == torch.float64
    q = torch.zeros(4, 3, requires_grad=True)                                     # not a Parameter: never leased
    assert _gradbuf.lease(q, (4, 3), q.device).shape == (4, 3)
    assert _gradbuf.lease(None, (2,), torch.device('cpu')).shape == (2,)
    _gradbuf.forget([p])
    assert not hasattr(p, '_icn_grad_view')


def test_strided_lease_for_a_channels_last_weight():
    w = torch.nn.Parameter(torch.zeros(3, 64, 1, 1).contiguous(memory_format=torch.channels_last))
    g = _gradbuf.lease(w, w.shape, w.device, stride=w.stride())
    assert g.stride() == w.stride() and g.shape == w.shape
    w.grad = g
    _gradbuf.refresh([w])
    w.grad = None
    assert _gradbuf.lease(w, w.shape, w.device, stride=w.stride()).data_ptr() == g.data_ptr()
