# Reads the phase stamps of a -DVC_GLU_TIMELINE build of csrc/vc_conv_glu.hip (tools/glu_timeline.patch): mean phase durations
# of workgroup (0,0,0) of vc_gated_conv1d_glu_fwd at B = 256 and B = 4096, fp32 and bf16.   python tools/glu_probe.py
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge  # noqa: E402,F401
import vcgan_pkg  # noqa: E402,F401
from voice_converter_cyclegan_amd._lib import lib, ptr, check, vp  # noqa: E402

fn = lib.vc_debug_glu_timeline
fn.argtypes = [C.c_void_p, C.c_int]
dev = 'cuda:0'
for math in ('bf16', 'fp32'):
    for B in (256, 4096):
        x = torch.randn((1, B, 24, 128), device=dev); w = torch.randn((1, 15, 24, 256), device=dev) * 0.05
        b = torch.randn((1, 256), device=dev) * 0.1; y = torch.empty((1, B, 128, 128), device=dev)
        s = vp(torch.cuda.current_stream().cuda_stream)
        check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1}[math]))
        call = lambda: check(lib.vc_gated_conv1d_glu_fwd(ptr(x), 0, ptr(w), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, 128, 256, 15, s))
        for _ in range(3):
            call()
        torch.cuda.synchronize(); fn(None, 1)
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        check(lib.vc_set_math_mode(0))
        out = (C.c_ulonglong * 8)(); fn(out, 0)
        n, tr = max(out[5], 1), max(out[6], 1)
        us = lambda c: c / 2290.0
        print('%s B=%d: launches %d, trips/launch %.1f' % (math, B, out[5], out[6] / n))
        print('   entry -> fragments + tile 0 in LDS   %8.0f cycles (%.2f us)' % (out[0] / n, us(out[0] / n)))
        print('   -> fragments in registers            %8.0f cycles (%.2f us)' % (out[1] / n, us(out[1] / n)))
        print('   per trip: start -> counted wait      %8.0f cycles (%.2f us)' % (out[2] / tr, us(out[2] / tr)))
        print('   per trip: wait -> barrier passed     %8.0f cycles (%.2f us)' % (out[3] / tr, us(out[3] / tr)))
        print('   loop end -> kernel end               %8.0f cycles (%.2f us)' % (out[4] / n, us(out[4] / n)), flush=True)
