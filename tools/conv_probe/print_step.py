"""One bf16 train step with VC_PROBE_PRINT=1: every conv_bf16_kernel launch's shape, configuration and time on stderr (needs the build of tools/conv_probe/apply.py)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.build()
from voice_converter_cyclegan_amd.engine import CycleGANEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = CycleGANEngine(24, seed=1); eng.set_math('bf16'); eng.set_overlap(False)
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda(); Bm = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda()
for _ in range(3): eng.train_step(A, Bm, 10., 5., 2e-4, 1e-4, sync=False)
torch.cuda.synchronize(); sys.stderr.write('PPSTART\n'); sys.stderr.flush()
eng.train_step(A, Bm, 10., 5., 2e-4, 1e-4, sync=False); torch.cuda.synchronize()
