"""Per-launch efficiency of the MFMA kernels inside one train step (HIP events around every launch)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from voice_converter_cyclegan_amd.engine import CycleGANEngine
from voice_converter_cyclegan_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = CycleGANEngine(24, seed=1)
if 'serial' in sys.argv[2:]: eng.set_overlap(False)      # per-launch timings without co-running chains
for a in sys.argv[2:]:
    if a in CycleGANEngine.MATH_MODES: eng.set_math(a)
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda(); Bm = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).cuda()
for _ in range(3): eng.train_step(A, Bm, 10., 5., 2e-4, 1e-4, sync=False)
torch.cuda.synchronize(); CycleGANEngine.profiling(True)
eng.train_step(A, Bm, 10., 5., 2e-4, 1e-4, sync=False); torch.cuda.synchronize()
for kind, nm in ((0, 'igemm'), (1, 'wgrad')):
    ms = (C.c_double * 512)(); fl = (C.c_double * 512)()
    n = lib.vc_prof_records(kind, ms, fl, 512)
    rows = [(ms[i] * 1e3, fl[i] / 1e9) for i in range(n)]
    tot_t = sum(r[0] for r in rows); tot_f = sum(r[1] for r in rows)
    print('== %s: %d launches, %.2f ms, %.1f GFLOP, %.1f TF avg' % (nm, n, tot_t / 1e3, tot_f, tot_f / tot_t * 1e3 if tot_t else 0))
    # time lost vs running at 95 TF
    lost = sorted(((t - f * 10.0, t, f, i) for i, (t, f) in enumerate(rows)), reverse=True)   # us lost vs 100 TFLOP/s (f GFLOP / 100 TF = f*10 us)
    for l_, t, f, i in lost[:int(os.environ.get('TOPN', '22'))]:
        print('   #%3d  %7.1f us  %7.2f GFLOP  %5.1f TF   lost vs 100TF: %6.1f us' % (i, t, f, f / t * 1e3 if t else 0, l_))
    print('   total lost vs 100 TF: %.2f ms' % (sum(x[0] for x in lost) / 1e3))
CycleGANEngine.profiling(False)
