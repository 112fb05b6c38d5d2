"""The north-star kernel (vc_gated_conv1d_glu_fwd) at several sizes, fp32 and bf16, HIP events around every call.  Run on the GPU box."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from voice_converter_cyclegan_amd._lib import lib, ptr, check, vp  # noqa: E402

import argparse  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument('--batches', default='', help='comma-separated batch sizes at T = 128 (default: the standard set incl. a long-utterance shape)')
ap.add_argument('--warm-ms', type=float, default=50.0, help='run the kernel this long before timing it (0: three calls, the rounds 1-3 protocol)')
args = ap.parse_args()
SHAPES = [(int(b), 128) for b in args.batches.split(',')] if args.batches else [(256, 128), (4096, 128), (8, 2048), (16, 128)]
dev = 'cuda:0'
for ver in (2,):
    for math in ('fp32', 'bf16'):
        for B, T in SHAPES:
            x = torch.randn((1, B, 24, T), device=dev)
            w = torch.randn((1, 15, 24, 256), device=dev) * 0.05
            b = torch.randn((1, 256), device=dev) * 0.1
            y = torch.empty((1, B, 128, T), device=dev)
            s = vp(torch.cuda.current_stream().cuda_stream)
            check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1}[math]))
            call = lambda: check(lib.vc_gated_conv1d_glu_fwd(ptr(x), 0, ptr(w), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, T, 256, 15, s))
            # the GPU drops its clocks while the host builds the tensors above: run the kernel for >= 50 ms before timing it, so
            # the figure is the kernel at sustained clocks (as inside a training loop), not at whatever the ramp had reached
            import time
            t_end = time.perf_counter() + args.warm_ms * 1e-3
            while True:
                for _ in range(50):
                    call()
                torch.cuda.synchronize()
                if time.perf_counter() >= t_end:
                    break
            lib.vc_prof_enable(1)
            reps = 20 if B <= 256 else 10
            for _ in range(reps):
                call()
            torch.cuda.synchronize()
            ms, n, fl = C.c_double(0), C.c_longlong(0), C.c_double(0)
            lib.vc_prof_collect(2, C.byref(ms), C.byref(n), C.byref(fl))
            lib.vc_prof_enable(0)
            check(lib.vc_set_math_mode(0))
            t = ms.value * 1e-3 / max(n.value, 1)
            algo = B * (24 * T + 128 * T) * 4 + (15 * 24 * 256 + 256) * 4
            flops = 2.0 * B * T * 256 * 15 * 24
            print('r4%s %-4s B=%-4d T=%-4d %8.2f us  HBM %6.3f of 8 TB/s  %7.1f TFLOP/s' % ('' if ver == 2 else '?', math, B, T, t * 1e6, algo / t / 8e12, flops / t / 1e12), flush=True)
