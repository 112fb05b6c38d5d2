"""Bitwise stress of the whole train step (model.py:110-125) in deterministic mode: from the same parameters / Adam state / inputs every
repetition must give the same losses, generations, gradients and updated parameters, bit for bit.  A missing dependency between the
step's five streams would show up here as a rare mismatch.   python tools/stress_step.py [repetitions] [batch] [math] [det|default]
default mode (atomics: summation order varies from run to run, and an L1 element within rounding of a tie may flip its sign): repetitions agree
to ~1e-6 .. 1e-3; a race that corrupts a tile moves whole tensors by O(1) -- the check is losses 1e-4, generations 1e-3, gradients 2e-2 (rel. L2)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.build()
from voice_converter_cyclegan_amd.engine import CycleGANEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
math = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
det = (sys.argv[4] if len(sys.argv) > 4 else 'det') == 'det'
dev = 'cuda:0'
eng = CycleGANEngine(num_features=24, device=dev, seed=42)
eng.set_math(math); eng.set_deterministic(det)
rng = np.random.default_rng(7)
A = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).to(dev); Bm = torch.from_numpy(rng.standard_normal((B, 24, 128)).astype(np.float32)).to(dev)
gA = torch.empty((B, 24, 128), device=dev); gB = torch.empty_like(gA)
p0, m0, v0 = eng.params.clone(), eng.m.clone(), eng.v.clone()
b1, b2 = list(eng.b1p), list(eng.b2p)
ref = None; bad = 0; worst = 0.0
for i in range(N + 1):
    eng.params.copy_(p0); eng.m.copy_(m0); eng.v.copy_(v0); eng.b1p, eng.b2p = list(b1), list(b2)
    eng.train_step(A, Bm, 10.0, 5.0, 2e-4, 1e-4, sync=False, gen_A=gA, gen_B=gB)
    torch.cuda.synchronize()
    cur = (eng.losses.clone(), gA.clone(), gB.clone(), eng.grads.clone(), eng.params.clone())
    if ref is None:
        ref = cur; continue
    if det:
        d = [bool(torch.equal(a, b)) for a, b in zip(cur, ref)]
    else:
        rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        tol = (1e-4, 1e-3, 1e-3, 2e-2, 1e-3) if math == 'fp32' else (2e-3, 2e-2, 2e-2, 1e-1, 1e-2)
        e = [rel(a, b) for a, b in zip(cur, ref)]
        worst = max(worst, *[x / t for x, t in zip(e, tol)]) if i > 1 else max(x / t for x, t in zip(e, tol))
        d = [x <= t for x, t in zip(e, tol)]
    if not all(d):
        bad += 1
        if bad <= 5:
            print('   repetition %d differs: losses %s gen_A %s gen_B %s grads %s params %s; grads differing elements %d' % (i, *d, int((cur[3] != ref[3]).sum().item())), flush=True)
print('%s B=%d %s: %d repetitions, %d differ from the first%s' % (math, B, 'deterministic (bitwise)' if det else 'default mode (tolerances)', N, bad, '' if det else '; worst error / tolerance %.3g' % worst))
