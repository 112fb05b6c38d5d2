"""A/B of one switch on the fp32 train step: ms per step, interleaved.  HOOK=vc_tune_winograd (default) | vc_tune_layer_updates.
python tools/ab_wino.py [batch ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
import vcgan_pkg  # noqa
from voice_converter_cyclegan_amd import engine as E
from voice_converter_cyclegan_amd._lib import lib, check
HOOK = getattr(lib, os.environ.get('HOOK', 'vc_tune_winograd'))
for B in [int(a) for a in sys.argv[1:]] or [8]:
    eng = E.CycleGANEngine(24, seed=0)
    rng = np.random.default_rng(0)
    A = torch.as_tensor(rng.standard_normal((B, 24, 128)).astype(np.float32), device='cuda:0'); Bm = torch.as_tensor(rng.standard_normal((B, 24, 128)).astype(np.float32), device='cuda:0')
    def run(n):
        for _ in range(n):
            eng.train_step(A, Bm, 10.0, 5.0, 2e-4, 1e-4, sync=False)
    if os.environ.get('OVERLAP'):
        run(1); check(lib.vc_plan_set_overlap(eng.plan(128), int(os.environ['OVERLAP'])))
    res = {0: [], 1: []}
    for rep in range(3):
        for w in (1, 0):
            check(HOOK(w))
            n = 30 if B > 2 else 100
            run(n // 2); torch.cuda.synchronize(); t0 = time.perf_counter(); run(n); torch.cuda.synchronize()
            res[w].append((time.perf_counter() - t0) / n * 1e3)
    check(HOOK(1))
    print('overlap %s fp32 B=%d: %s on %s ms   off %s ms' % (os.environ.get('OVERLAP', 'default'), B, os.environ.get('HOOK', 'vc_tune_winograd'),
                                                             ' '.join('%.3f' % x for x in res[1]), ' '.join('%.3f' % x for x in res[0])), flush=True)
    eng.close()
