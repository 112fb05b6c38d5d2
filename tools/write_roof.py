"""What HBM allows a write-dominated stream of the named kernel's size: torch fill_ (pure write, 16-byte stores), copy_ (read + write) and a 16 %-read /
84 %-write mix (the kernel's own traffic mix: 12 KB in + 64 KB out per sample) on the tensors of B = 256 and B = 4096.  us per call back to back, TB/s."""
import torch
dev = 'cuda:0'
def t(f, n=200):
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (256, 4096, 16384):
    y = torch.empty((B, 128, 128), device=dev); y2 = torch.empty_like(y); x = torch.randn((B, 24, 128), device=dev); x2 = torch.empty_like(x)
    n = 200 if B <= 4096 else 50
    by = y.numel() * 4; bx = x.numel() * 4
    a = t(lambda: y.fill_(1.0), n); print('B=%5d fill   %6.1f MB  %8.2f us  %.2f TB/s written' % (B, by / 1e6, a, by / a / 1e6))
    a = t(lambda: y2.copy_(y), n); print('B=%5d copy   %6.1f MB  %8.2f us  %.2f TB/s (read + write)' % (B, 2 * by / 1e6, a, 2 * by / a / 1e6))
    a = t(lambda: (x2.copy_(x), y.fill_(1.0)), n); print('B=%5d x-copy + y-fill (two launches) %6.1f MB  %8.2f us  %.2f TB/s' % (B, (by + 2 * bx) / 1e6, a, (by + 2 * bx) / a / 1e6))
    a = t(lambda: torch.sum(y), n); print('B=%5d read   %6.1f MB  %8.2f us  %.2f TB/s read' % (B, by / 1e6, a, by / a / 1e6))
