"""Experiment (GPU box): does running the two generator groups as TWO launches on two streams, out of phase, hide the prologue /
epilogue of one under the K loop of the other?  A chain of N dependent-looking launches of one layer shape, three ways:
  joint   : groups = 2 in one launch, one stream (what the step does)
  split   : groups = 1 + 1 on two streams, started together
  shifted : the same, the second stream started half a launch late
Every launch is forced to the same tile configuration.  Prints us per layer-pair."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from voice_converter_cyclegan_amd._lib import lib, ptr, check, vp  # noqa: E402
from tests.step_shapes import layers  # noqa: E402

GL, DL = layers(24, 128)
B = 8
N = 40
dev = 'cuda:0'
for nm, mult, cfg in (('g_u2', 2, (64, 2, 1)), ('g_r_h1', 2, (64, 2, 1)), ('g_r_h2', 2, (64, 4, 2)), ('g_u1', 1, (64, 4, 1))):
    Cin, H, W, Cout, kh, kw, sh, sw = GL[nm]
    Bs = mult * B
    OH, OW = -(-H // sh), -(-W // sw)
    x = torch.randn((2, Bs, Cin, H, W), device=dev)
    w = torch.randn((2, kh, kw, Cin, Cout), device=dev) / float(np.sqrt(kh * kw * Cin))
    b = torch.randn((2, Cout), device=dev)
    y = torch.zeros((2, Bs, Cout, OH, OW), device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    check(lib.vc_tune_igemm_override(*cfg))

    def launch(g0, ng, st):
        check(lib.vc_conv2d_fwd(ptr(x[g0]), x[0].numel(), ptr(w[g0]), w[0].numel(), ptr(b[g0]), b[0].numel(), ptr(y[g0]), y[0].numel(),
                                ng, Bs, Cin, H, W, Cout, kh, kw, sh, sw, vp(st.cuda_stream)))

    def run(mode):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == 'joint':
            for _ in range(N):
                launch(0, 2, s1)
        else:
            if mode == 'shifted':
                launch(0, 1, s1)                     # a head start of one launch for stream 1 ...
                torch.cuda._sleep(20000)             # ... (host-side spacing is too coarse: let the GPU idle stream 2 ~10 us instead)
            for i in range(N):
                launch(0, 1, s1)
                launch(1, 1, s2)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6 / N

    for mode in ('joint', 'split', 'shifted'):
        for _ in range(3):
            run(mode)
        ts = sorted(run(mode) for _ in range(7))
        print('%-8s %-8s tile %dx%d/%d  %.1f us per layer pair (min %.1f)' % (nm, mode, cfg[0], cfg[1], cfg[2], ts[3], ts[0]), flush=True)
    check(lib.vc_tune_igemm_override(0, 0, 0))
