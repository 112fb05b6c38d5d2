"""Names a stale read / ordering defect of the train step (generator_B2A gradients 2e-3 apart on the first lambda_identity == 0
step after a lambda_identity > 0 step, bf16 mode).  Run on the GPU box:  python tools/stale_read_probe.py [fp32|bf16] [B]
  A  reproduce: step(5), then three steps(0); gradients AND the whole step workspace of consecutive steps compared region by region
     (vc_testing_describe_step names the regions): the first region in schedule order that differs is where the defect enters
  B  which of the two is wrong: both against a serialised, device-synchronised steady-state step(0)
  C  poison: workspace and plan-internal buffers filled with NaN between steps; NaN in a gradient = a read of something this step never wrote
  D  de-confound round 5's experiments: generation outputs requested x device-wide synchronisation
  E  overlap mechanisms one at a time (vc_plan_set_overlap(8 * bits))
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from tests.test_step_gpu import _setup, _is_noise_bias
from voice_converter_cyclegan_amd._lib import lib, check
import ctypes as C

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
O, P, A, Bm, eng = _setup(B, gap=False)
eng.set_math(mode)
check(lib.vc_tune_conv_bf16_override(128, 4, 1)); check(lib.vc_tune_igemm_override(64, 1, 1)); check(lib.vc_tune_wgrad_override(64, 1, 1))
gA = torch.empty((B, 24, 128), device='cuda:0'); gB = torch.empty_like(gA)
A_d, B_d = eng._to_dev(A), eng._to_dev(Bm)


def step(lam, gen=True, devsync=False, snap=False):
    if gen:
        p = eng.step_grads(A_d, B_d, 10.0, lam, gen_A=gA)
    else:
        p = eng.step_grads(A_d, B_d, 10.0, lam)
    if devsync:
        torch.cuda.synchronize()
    g = eng.tf_gradients()
    return (g, eng._ws.clone() if snap else None, p)


def worst(a, b, k=3, scope=None):
    w = sorted(((float(np.linalg.norm(a[n].astype(np.float64) - b[n].astype(np.float64)) / (np.linalg.norm(b[n].astype(np.float64)) + 1e-30)), n)
                for n in b if not _is_noise_bias(n) and (scope is None or n.startswith(scope))), reverse=True)[:k]
    return ' '.join('%s %.2e' % (n, e) for e, n in w)


def regions(p, Bbwd):
    n = lib.vc_testing_describe_step(p, B, Bbwd, None, 0)
    buf = C.create_string_buffer(int(n))
    lib.vc_testing_describe_step(p, B, Bbwd, buf, n)
    out = []
    for line in buf.value.decode().splitlines():
        name, off, ln = line.split()
        out.append((name, int(off), int(ln)))
    return out


def ws_diff(a, b, regs, tag, limit=40):
    ai, bi = a.view(torch.int32), b.view(torch.int32)
    neq = (ai != bi)
    cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=a.device), neq.to(torch.int64).cumsum(0)])
    print('--- workspace regions that differ: %s (total %d of %d floats)' % (tag, int(cs[-1]), a.numel()))
    shown = 0
    for name, off, ln in regs:
        if ln <= 0 or off + ln > a.numel():
            continue
        c = int(cs[off + ln] - cs[off])
        if c and shown < limit:
            x, y = a[off:off + ln], b[off:off + ln]
            d = (x - y).abs()
            fin = torch.isfinite(d)
            md = float(d[fin].max()) if bool(fin.any()) else float('nan')
            print('   %-28s off %11d len %10d  differing %9d  max|d| %.3e  max|ref| %.3e' % (name, off, ln, c, md, float(y[torch.isfinite(y)].abs().max()) if bool(torch.isfinite(y).any()) else float('nan')))
            shown += 1


def nan_report(ws, regs, tag, limit=60):
    bad = torch.isnan(ws)
    cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=ws.device), bad.to(torch.int64).cumsum(0)])
    print('--- NaN per workspace region after the step: %s' % tag)
    shown = 0
    for name, off, ln in regs:
        if ln <= 0 or off + ln > ws.numel():
            continue
        c = int(cs[off + ln] - cs[off])
        if c and shown < limit:
            print('   %-28s len %10d  NaN %10d%s' % (name, ln, c, '  (all)' if c == ln else ''))
            shown += 1


print('=== mode %s  B %d' % (mode, B), flush=True)
# ---------------------------------------------------------------- A
g5, _, p = step(5.0)
g5b, _, _ = step(5.0)
print('A  step(5) vs step(5):           ', worst(g5b, g5))
s1, w1, _ = step(0.0, snap=True)
s2, w2, _ = step(0.0, snap=True)
s3, w3, _ = step(0.0, snap=True)
print('A  first step(0) vs second:      ', worst(s1, s2))
print('A  second step(0) vs third:      ', worst(s2, s3))
for sc in ('generator_A2B', 'generator_B2A', 'discriminator_A', 'discriminator_B'):
    print('A    first vs second, %-16s' % sc, worst(s1, s2, 2, sc))
regs = regions(p, B)
ws_diff(w1, w2, regs, 'first step(0) vs second step(0)')
ws_diff(w2, w3, regs, 'second step(0) vs third step(0)')
# ---------------------------------------------------------------- B
eng.set_overlap(False)
for _ in range(2):
    r, wr, _ = step(0.0, devsync=True, snap=True)
eng.set_overlap(True)
print('B  first step(0) vs serialised reference: ', worst(s1, r))
print('B  second step(0) vs serialised reference:', worst(s2, r))
ws_diff(w1, wr, regs, 'first step(0) vs serialised reference', limit=25)
ws_diff(w2, wr, regs, 'second step(0) vs serialised reference', limit=25)
del w1, w2, w3, wr
# ---------------------------------------------------------------- C
for lam_prev, lam in ((5.0, 0.0), (0.0, 5.0), (5.0, 5.0), (0.0, 0.0)):
    step(lam_prev); torch.cuda.synchronize()
    eng._ws.view(torch.int32).fill_(-1)                       # 0xffffffff: NaN as fp32, NaN as bf16
    check(lib.vc_testing_poison_internal(p, eng._stream(), 0xff))
    torch.cuda.synchronize()
    g, _, _ = step(lam, devsync=True)
    bad = [(n, int(np.isnan(v).sum()), v.size) for n, v in g.items() if np.isnan(v).any()]
    print('C  poison, step(%g) then step(%g): gradient tensors with NaN: %d' % (lam_prev, lam, len(bad)), bad[:12])
    print('C    losses', eng.losses.cpu().numpy())
    if bad:
        nan_report(eng._ws, regions(p, B if lam == 0.0 else 2 * B), 'step(%g) after poison' % lam)
# ---------------------------------------------------------------- D
for gen in (True, False):
    for ds in (False, True):
        step(5.0, gen, ds); a, _, _ = step(0.0, gen, ds); b, _, _ = step(0.0, gen, ds)
        print('D  generation outputs %-5s device sync %-5s: first step(0) vs second: %s' % (gen, ds, worst(a, b, 2)))
# ---------------------------------------------------------------- E
for bits in (0, 1, 2, 4, 3, 5, 6, 7):
    check(lib.vc_plan_set_overlap(p, 8 * bits if bits else 0))
    step(5.0); a, _, _ = step(0.0); b, _, _ = step(0.0)
    print('E  overlap bits %d: first step(0) vs second: %s' % (bits, worst(a, b, 2)))
check(lib.vc_plan_set_overlap(p, 1))
