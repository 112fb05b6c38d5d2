"""One configuration of the fp32 step for rocprofv3 --kernel-trace --stats: WINO=0|1, serial launches.  python tools/prof_wino.py [batch]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
import vcgan_pkg  # noqa
from voice_converter_cyclegan_amd import engine as E
from voice_converter_cyclegan_amd._lib import lib, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = E.CycleGANEngine(24, seed=0)
rng = np.random.default_rng(0)
A = torch.as_tensor(rng.standard_normal((B, 24, 128)).astype(np.float32), device='cuda:0'); Bm = torch.as_tensor(rng.standard_normal((B, 24, 128)).astype(np.float32), device='cuda:0')
check(lib.vc_tune_winograd(int(os.environ.get('WINO', '1'))))
eng.set_overlap(False)
for _ in range(16):
    eng.train_step(A, Bm, 10.0, 5.0, 2e-4, 1e-4, sync=False)
torch.cuda.synchronize()
