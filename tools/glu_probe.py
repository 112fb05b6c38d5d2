# Reads the phase stamps of a -DVC_GLU_TIMELINE build of csrc/vc_conv_glu.hip (tools/glu_timeline.patch): mean phase durations of two
# workgroups (first / last of the grid), K set 0 and 1, of vc_gated_conv1d_glu_fwd[_packed], and the clock the kernel ran at
# (shader cycles / 100 MHz real-time ticks).   python tools/glu_probe.py
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
import itertools
CASES = [('bf16', 256, True, 0), ('bf16', 256, True, 1), ('bf16', 256, False, 0), ('bf16', 1024, True, 0), ('bf16', 1024, True, 1), ('bf16', 4096, True, 0),
         ('fp32', 256, True, 0), ('fp32', 256, False, 0)]
for math, B, packed, flags in CASES:
    for _once in (0,):
        for _once2 in (0,):
            x = torch.randn((1, B, 24, 128), device=dev); w = torch.randn((1, 15, 24, 256), device=dev) * 0.05
            b = torch.randn((1, 256), device=dev) * 0.1; y = torch.empty((1, B, 128, 128), device=dev)
            s = vp(torch.cuda.current_stream().cuda_stream)
            check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1}[math])); lib.vc_tune_glu(flags)
            img = torch.empty((lib.vc_gated_conv1d_glu_image_bytes(24, 15, 256),), dtype=torch.uint8, device=dev)
            check(lib.vc_gated_conv1d_glu_prepack(ptr(w), 0, ptr(img), 0, 1, 24, 15, 256, s))
            if packed:
                call = lambda: check(lib.vc_gated_conv1d_glu_fwd_packed(ptr(x), 0, ptr(img), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, 128, 256, 15, s))
            else:
                call = lambda: check(lib.vc_gated_conv1d_glu_fwd(ptr(x), 0, ptr(w), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, 128, 256, 15, s))
            for _ in range(200 if B <= 256 else 40):
                call()
            torch.cuda.synchronize(); fn(None, 1)
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            check(lib.vc_set_math_mode(0)); lib.vc_tune_glu(0)
            out = (C.c_ulonglong * 64)(); fn(out, 0)
            print('%s B=%d %s flags=%d' % (math, B, 'packed' if packed else 'plain', flags))
            for slot in range(4):
                o = out[16 * slot:16 * slot + 16]
                n, tr = max(o[5], 1), max(o[6], 1)
                ghz = o[7] / max(o[8], 1) / 10.0
                print('   wg %s set %d: clock %.2f GHz  kernel %6.0f cyc (%.2f us) | prologue %5.0f = issue %4.0f + landed %4.0f + staged %4.0f + barrier %4.0f | per trip (%4.1f): K loop %5.0f  hand-over out %4.0f  barrier %4.0f | tail %5.0f' % (
                    'first' if slot < 2 else 'last ', slot & 1, ghz, o[7] / n, o[8] / n * 0.01, o[0] / n, o[9] / n, o[10] / n, o[11] / n, o[12] / n, o[6] / n, o[1] / tr, o[2] / tr, o[3] / tr, o[4] / n), flush=True)
