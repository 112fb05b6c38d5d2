# Reads the phase stamps of a -DVC_TIMELINE build of csrc/vc_conv_igemm.hip (tools/igemm_timeline.patch):
#   python tools/clk_probe.py timeline [batch]
# mean phase durations of workgroup (0,0,0) over all igemm launches of the step, in cycles and microseconds
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge  # noqa: E402,F401
import vcgan_pkg  # noqa: E402,F401
from voice_converter_cyclegan_amd.engine import CycleGANEngine  # noqa: E402
from voice_converter_cyclegan_amd._lib import lib  # noqa: E402

mode = 'timeline'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fn = lib.vc_debug_timeline
fn.argtypes = [C.c_void_p, C.c_int]
eng = CycleGANEngine(24, device='cuda:0', seed=1)
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda()
Bm = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda()
for _ in range(20):
    eng.train_step(A, Bm, 10.0, 5.0, 2e-4, 1e-4, sync=False)
torch.cuda.synchronize()
fn(None, 1)
for _ in range(200):
    eng.train_step(A, Bm, 10.0, 5.0, 2e-4, 1e-4, sync=False)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
fn(out, 0)
if mode == 'clk':
    print('B', B, 'cycles', out[0], 'ticks(100MHz)', out[1], 'clock GHz', out[0] / max(out[1], 1) * 0.1)
else:
    n = max(out[5], 1)
    names = ('entry -> tables + first loads issued', '-> first barrier passed (first MFMA)', '-> K loop done', '-> end of the epilogue')
    print('B', B, 'launches', out[5], 'mean K chunks per part', out[6] / n)
    for q, nm in enumerate(names):
        print('  %-40s %9.0f cycles  (%.2f us at 2.29 GHz)' % (nm, out[q] / n, out[q] / n / 2290.0))
