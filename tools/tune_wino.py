"""Tile configurations of the Winograd route's channel GEMMs (1 x 1, 6G groups, no split-K): forward and input gradient of the residual blocks'
two convolutions at the step's two batch sizes, every built (tile rows, K groups) forced in turn.  us per call, HIP events, 30 calls each
(the operator-level call includes the stand-alone transforms and the filter transform: a constant offset per shape)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from tests import gpu_util as U
from voice_converter_cyclegan_amd._lib import lib, check
G = 2
for Cin, Cout in ((512, 2048), (1024, 512)):
    for B in [int(b) for b in sys.argv[1:]] or [8, 16]:
        W = 32
        x = torch.randn((G, B, Cin, 1, W), device=U.DEV); w = torch.randn((G, 1, 3, Cin, Cout), device=U.DEV) * 0.02; b = torch.zeros((G, Cout), device=U.DEV)
        dy = torch.randn((G, B, Cout, 1, W), device=U.DEV)
        for op in ('fwd', 'dgrad'):
            res = []
            for bm, kg in ((0, 0), (64, 1), (64, 2), (64, 4), (128, 1), (128, 2)):
                check(lib.vc_tune_igemm_override(bm, kg, 1 if bm else 0))
                f = (lambda: U.conv_fwd(x, w, b, 1, 3, 1, 1)) if op == 'fwd' else (lambda: U.conv_dgrad(dy, w, 1, W, 1, 3, 1, 1))
                for _ in range(5): f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(30): f()
                e1.record(); torch.cuda.synchronize()
                res.append(('auto' if not bm else '%dx%d' % (bm, kg), e0.elapsed_time(e1) / 30 * 1e3))
            check(lib.vc_tune_igemm_override(0, 0, 0))
            print('%4d -> %4d B=%2d %5s: ' % (Cin, Cout, B, op) + '  '.join('%s %.1f' % r for r in res), flush=True)
