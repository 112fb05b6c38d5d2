"""What the memory system gives plain streaming kernels on the named kernel's B = 4096 output (268 MB): torch's fill and copy kernels, back to back.
Outer bounds beside vc_prof_glu_copy_roof (bench.py named_kernel.copy_roof), which runs the named kernel's memory operations on the named kernel's grid
(measured: fill 6.96 TB/s, copy 5.45 TB/s at 268 MB; the control 3.5-3.8 TB/s, the kernel 2.8-2.9 TB/s)."""
import torch
dev = 'cuda:0'
for B in (256, 4096):
    y = torch.empty((B, 128, 128), device=dev); z = torch.randn_like(y)
    for name, f, byt in (('fill', lambda: y.fill_(1.0), y.numel() * 4), ('copy', lambda: y.copy_(z), 2 * y.numel() * 4)):
        for _ in range(200): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 400 if B == 256 else 60
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / n
        print('B=%-4d %s of the %6.1f MB output: %7.2f us back to back = %.2f TB/s (%s)' % (B, name, y.numel() * 4 / 1e6, t * 1e6, byt / t / 1e12, 'written' if name == 'fill' else 'read + written'))
