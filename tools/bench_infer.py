"""Whole-utterance inference throughput (model.test / convert.py:46): generator forward on [B, 24, T], weights and
input resident in HBM, HIP events around `reps` forwards.  usage: python tools/bench_infer.py [T=2048] [B=1] [math=fp32]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from voice_converter_cyclegan_amd.engine import CycleGANEngine
from voice_converter_cyclegan_amd._lib import lib, ptr, check, vp
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
math = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
eng = CycleGANEngine(24, seed=1); eng.set_math(math)
if os.environ.get('WINO') is not None:
    check(lib.vc_tune_winograd(int(os.environ['WINO'])))      # A/B of the fp32 Winograd route (default: on)
p = eng.plan(T)
x = torch.randn((B, 24, T), device='cuda'); y = torch.empty_like(x)
save = torch.empty(lib.vc_generator_save_floats(p, B), device='cuda')
s = vp(torch.cuda.current_stream().cuda_stream)
run = lambda: check(lib.vc_generator_forward(p, ptr(eng.params), 0, ptr(x), 0, ptr(y), 0, ptr(save), 1, B, s))
for _ in range(3): run()
torch.cuda.synchronize()
reps = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gf = 2.656 * B * T / 128        # GFLOP per application, linear in T
print('generator forward B=%d T=%d %s: %.3f ms  %.0f frames/s  %.1f TFLOP/s' % (B, T, math, ms, B * T / ms * 1e3, gf / ms))
