"""rocBLAS / hipBLASLt sgemm (through torch.mm) on the dense coarse-level GEMM shapes of the decoder-block head, for comparison
with the LDS-DMA kernel's identity-gather launches (developer tool; GPU only)."""
import torch


def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


torch.backends.cuda.matmul.allow_tf32 = False
B = 36
for name, Pc, cin, c in (('up1', 160, 256, 512), ('up2', 640, 256, 256), ('up3', 2560, 128, 128)):
    M = B * Pc
    x = torch.randn(M, cin, device='cuda')
    wf = torch.randn(cin, 7 * c, device='cuda')
    g = torch.randn(M, 7 * c, device='cuda')
    wb = torch.randn(7 * c, cin, device='cuda')
    fl = 2.0 * M * cin * 7 * c
    t1 = timed(lambda: torch.mm(x, wf))
    t2 = timed(lambda: torch.mm(g, wb))
    t3 = timed(lambda: torch.mm(x.t(), g))
    print('%s M=%d Cin=%d 7C=%d | z=x*Wf %.1f us %.1f TF/s | dx=g*Wb %.1f us %.1f TF/s | dW=x^T*g %.1f us %.1f TF/s' % (
        name, M, cin, 7 * c, t1 * 1e6, fl / t1 / 1e12, t2 * 1e6, fl / t2 / 1e12, t3 * 1e6, fl / t3 / 1e12))
