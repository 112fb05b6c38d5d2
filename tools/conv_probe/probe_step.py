"""Phase totals of conv_bf16_kernel over bf16 train steps (needs the build of tools/conv_probe/apply.py).  python tools/conv_probe/probe_step.py [batch] [serial]"""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.build()
from voice_converter_cyclegan_amd.engine import CycleGANEngine
from voice_converter_cyclegan_amd._lib import lib
lib.vc_probe_conv_bf16_phases.argtypes = [C.c_void_p, C.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = CycleGANEngine(24, seed=1); eng.set_math('bf16')
if 'serial' in sys.argv: eng.set_overlap(False)
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda(); Bm = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda()
for _ in range(3): eng.train_step(A, Bm, 10., 5., 2e-4, 1e-4, sync=False)
torch.cuda.synchronize()
buf = np.zeros(256, np.uint64); lib.vc_probe_conv_bf16_phases(buf.ctypes.data, 1)
import time
t0 = time.perf_counter(); N = 5
for _ in range(N): eng.train_step(A, Bm, 10., 5., 2e-4, 1e-4, sync=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
lib.vc_probe_conv_bf16_phases(buf.ctypes.data, 0)
q = buf.reshape(32, 8).astype(np.float64) / N
names = {0: '256/3', 1: '128/2', 2: '128/4', 3: '64/2', 4: '64/4'}
print('B=%d step %.2f ms; per step, WG-cycles summed (M = 1e6):' % (B, dt * 1e3))
tot = np.zeros(4)
for slot in list(range(10)) + list(range(16, 26)):
    pro, loop, epi, mf, wgs, launches = q[slot, :6]
    if wgs == 0: continue
    print('  %-6s role %d: %3.0f launches %6.0f WGs  prologue %7.1fM  loop %7.1fM  epilogue %7.1fM  | MFMA %7.1fM = %.2f of loop, %.2f of life; per WG: pro %5.0f loop %6.0f epi %6.0f' % (
        names[(slot % 16) // 2] + ('S' if slot >= 16 else ''), slot % 2, launches, wgs, pro / 1e6, loop / 1e6, epi / 1e6, mf / 1e6, mf / loop, mf / (pro + loop + epi), pro / wgs, loop / wgs, epi / wgs))
    tot += [pro, loop, epi, mf]
print('  all: prologue %.1fM loop %.1fM epilogue %.1fM MFMA %.1fM -> MFMA %.2f of life; if prologue+epilogue vanished: x%.2f' % (tot[0] / 1e6, tot[1] / 1e6, tot[2] / 1e6, tot[3] / 1e6, tot[3] / tot[:3].sum(), tot[:3].sum() / tot[1]))
