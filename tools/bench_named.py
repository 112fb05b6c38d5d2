"""The north-star kernel (vc_gated_conv1d_glu_fwd / _fwd_packed) at several sizes, fp32, bf16 and f16 (math mode 3).  Run on the GPU box.

Per (math, shape) three launches are timed with the same protocol (>= 50 ms of the same call first: sustained clocks):
  plain   vc_gated_conv1d_glu_fwd         filters in TF layout, converted by every workgroup
  packed  vc_gated_conv1d_glu_fwd_packed  filters as the pre-packed fragment image (vc_gated_conv1d_glu_prepack, outside the timed region)
  roof    vc_prof_glu_copy_roof           the same grid performing the kernel's memory operations only (bf16 image): the copy roof
and two clocks: `ev` = HIP events around every single call (launch + kernel + event overhead), `b2b` = N calls back to back between two
events / N (the average launch duration in a stream of launches: what a caller that does not synchronise per call sees)."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from voice_converter_cyclegan_amd._lib import lib, ptr, check, vp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batches', default='', help='comma-separated batch sizes at T = 128 (default: the standard set incl. a long-utterance shape)')
ap.add_argument('--warm-ms', type=float, default=50.0, help='run the kernel this long before timing it (0: three calls, the rounds 1-3 protocol)')
ap.add_argument('--maths', default='fp32,bf16')
ap.add_argument('--flags', type=int, default=0, help='vc_tune_glu switches for the timed launches')
ap.add_argument('--data', default='rand', help='rand | zero | one: operand fill (the chip\'s power management is data dependent)')
args = ap.parse_args()
SHAPES = [(int(b), 128) for b in args.batches.split(',')] if args.batches else [(256, 128), (4096, 128), (8, 2048), (16, 128)]
dev = 'cuda:0'


def timed(call, B):
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
    ev = ms.value * 1e-3 / max(n.value, 1)
    nb = 200 if B <= 256 else 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        call()
    e0.record()
    for _ in range(nb):
        call()
    e1.record()
    torch.cuda.synchronize()
    return ev, e0.elapsed_time(e1) * 1e-3 / nb


for math in args.maths.split(','):
    for B, T in SHAPES:
        x = torch.randn((1, B, 24, T), device=dev)
        w = torch.randn((1, 15, 24, 256), device=dev) * 0.05
        b = torch.randn((1, 256), device=dev) * 0.1
        if args.data != 'rand':
            x.fill_(0.0 if args.data == 'zero' else 1.0); w.fill_(0.0 if args.data == 'zero' else 1.0); b.zero_()
        y = torch.empty((1, B, 128, T), device=dev)
        s = vp(torch.cuda.current_stream().cuda_stream)
        lib.vc_tune_glu(args.flags)
        check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1, 'f16': 3}[math]))
        nb_img = lib.vc_gated_conv1d_glu_image_bytes(24, 15, 256)
        img = torch.empty((nb_img,), dtype=torch.uint8, device=dev)
        check(lib.vc_gated_conv1d_glu_prepack(ptr(w), 0, ptr(img), 0, 1, 24, 15, 256, s))
        check(lib.vc_set_math_mode(1))
        img16 = torch.empty((lib.vc_gated_conv1d_glu_image_bytes(24, 15, 256),), dtype=torch.uint8, device=dev)
        check(lib.vc_gated_conv1d_glu_prepack(ptr(w), 0, ptr(img16), 0, 1, 24, 15, 256, s))
        check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1, 'f16': 3}[math]))
        calls = [('plain ', lambda: check(lib.vc_gated_conv1d_glu_fwd(ptr(x), 0, ptr(w), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, T, 256, 15, s))),
                 ('packed', lambda: check(lib.vc_gated_conv1d_glu_fwd_packed(ptr(x), 0, ptr(img), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, T, 256, 15, s)))]
        if math in ('bf16', 'f16'):
            calls.append(('roof  ', lambda: check(lib.vc_prof_glu_copy_roof(ptr(x), ptr(img16), ptr(y), B, T, 256, s))))
        algo = B * (24 * T + 128 * T) * 4 + (15 * 24 * 256 + 256) * 4
        flops = 2.0 * B * T * 256 * 15 * 24
        for name, call in calls:
            ev, b2b = timed(call, B)
            print('r6%s %-4s %s B=%-4d T=%-4d  ev %8.2f us (HBM %5.3f, %7.1f TFLOP/s)   b2b %8.2f us (HBM %5.3f, %7.1f TFLOP/s)' % (
                '' if args.data == 'rand' else '[' + args.data + ']', math, name, B, T, ev * 1e6, algo / ev / 8e12, flops / ev / 1e12, b2b * 1e6, algo / b2b / 8e12, flops / b2b / 1e12), flush=True)
        check(lib.vc_set_math_mode(0))
