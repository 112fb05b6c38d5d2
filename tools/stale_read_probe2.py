"""Catches a deviating bf16 step in the act: N identical steps, the whole workspace of each compared region by region against step 1's
(tools/stale_read_probe.py showed: the 2e-3 alternative in a generator's gradient chain appears sporadically between IDENTICAL steps of
either schedule).  python tools/stale_read_probe2.py [lam] [N] [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from tests.test_step_gpu import _setup, _is_noise_bias
from voice_converter_cyclegan_amd._lib import lib, check
import ctypes as C
lam = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
O, P, A, Bm, eng = _setup(B, gap=False)
eng.set_math('bf16')
if 'auto' not in sys.argv:
    check(lib.vc_tune_conv_bf16_override(128, 4, 1)); check(lib.vc_tune_igemm_override(64, 1, 1)); check(lib.vc_tune_wgrad_override(64, 1, 1))
A_d, B_d = eng._to_dev(A), eng._to_dev(Bm)


def regions(p, Bbwd):
    n = lib.vc_testing_describe_step(p, B, Bbwd, None, 0)
    buf = C.create_string_buffer(int(n))
    lib.vc_testing_describe_step(p, B, Bbwd, buf, n)
    return [(l.split()[0], int(l.split()[1]), int(l.split()[2])) for l in buf.value.decode().splitlines()]


def ws_diff(a, b, regs, limit=80):
    neq = (a.view(torch.int32) != b.view(torch.int32))
    cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=a.device), neq.to(torch.int64).cumsum(0)])
    shown = 0
    for name, off, ln in regs:
        if ln <= 0 or off + ln > a.numel():
            continue
        c = int(cs[off + ln] - cs[off])
        if c and shown < limit:
            x, y = a[off:off + ln], b[off:off + ln]
            d = (x - y).abs(); fin = torch.isfinite(d)
            md = float(d[fin].max()) if bool(fin.any()) else float('nan')
            mr = float(y[torch.isfinite(y)].abs().max()) if bool(torch.isfinite(y).any()) else float('nan')
            first = int(torch.nonzero(neq[off:off + ln])[0])
            print('   %-28s off %11d len %10d  differing %9d  first@%d  max|d| %.3e  max|ref| %.3e' % (name, off, ln, c, first, md, mr))
            shown += 1


p = eng.step_grads(A_d, B_d, 10.0, lam); torch.cuda.synchronize()
regs = regions(p, B if lam == 0.0 else 2 * B)
eng.step_grads(A_d, B_d, 10.0, lam); torch.cuda.synchronize()
ref_ws = eng._ws.clone(); ref_g = eng.grads.clone()
shown = 0
for i in range(N):
    eng.step_grads(A_d, B_d, 10.0, lam); torch.cuda.synchronize()
    rel = float((eng.grads - ref_g).norm() / ref_g.norm())
    nd = int((eng._ws.view(torch.int32) != ref_ws.view(torch.int32)).sum())
    print('step %3d: flat gradient rel-L2 vs step 1 %.3e   workspace floats differing %d' % (i + 2, rel, nd), flush=True)
    if rel > 1e-4 and shown < 3:
        ws_diff(eng._ws, ref_ws, regs); shown += 1
