import sys, time; sys.path.insert(0, '/root/repo')
import torch, numpy as np
import __graft_entry__ as ge; ge.build()
from voice_converter_cyclegan_amd.engine import CycleGANEngine
eng = CycleGANEngine(24, seed=1)
for B in (1, 8):
    A = torch.randn(B, 24, 128, device='cuda'); Bm = torch.randn(B, 24, 128, device='cuda')
    for _ in range(5): eng.train_step(A, Bm, 10., 5., 2e-4, 1e-4, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): eng.train_step(A, Bm, 10., 5., 2e-4, 1e-4, sync=False)
    t_host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 20
    print('B=%d host enqueue %.2f ms/step, wall %.2f ms/step' % (B, t_host * 1e3, t_all * 1e3))
