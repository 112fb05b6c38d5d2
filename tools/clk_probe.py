# needs a -DVC_CLKPROBE build of csrc/vc_conv_igemm.hip (tools/igemm_ablation.patch); prints the effective shader clock inside the igemm launches of the train step

import ctypes as C, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
import vcgan_pkg  # noqa: F401
from voice_converter_cyclegan_amd.engine import CycleGANEngine
from voice_converter_cyclegan_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = CycleGANEngine(24, device='cuda:0', seed=1)
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda(); Bm = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda()
for _ in range(20): eng.train_step(A, Bm, 10.0, 5.0, 2e-4, 1e-4, sync=False)
torch.cuda.synchronize()
lib.vc_debug_clk.argtypes = [C.c_void_p, C.c_int]
lib.vc_debug_clk(None, 1)
for _ in range(200): eng.train_step(A, Bm, 10.0, 5.0, 2e-4, 1e-4, sync=False)
torch.cuda.synchronize()
out = (C.c_ulonglong * 2)()
lib.vc_debug_clk(out, 0)
print('B', B, 'cycles', out[0], 'ticks(100MHz)', out[1], 'clock GHz', out[0] / max(out[1], 1) * 0.1)
