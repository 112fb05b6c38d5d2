"""Reads the per-workgroup entry / exit stamps of a -DVC_TIMELINE -DVC_WGSPAN build of csrc/vc_conv_igemm.hip
(tools/igemm_timeline.patch) for single launches of some layers of the batch-8 step:  python tools/wg_span.py
Prints, per launch: workgroups, kernel span (last exit - first entry), spread of the entries (dispatch), of the exits, and the
lifetime quantiles -- what part of a launch's duration is NOT the lifetime of a typical workgroup."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from tests import gpu_util as U  # noqa: E402
from voice_converter_cyclegan_amd._lib import lib, check  # noqa: E402

fn = lib.vc_debug_wgspan
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
GHZ = 2.29e3        # cycles per us (tools/clk_probe.py)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
from tests.step_shapes import layers  # noqa: E402
GL, DL = layers(24, 128)
LAY = dict(GL); LAY.update(DL)
CASES = [(nm, mult * B) + tuple(LAY[nm]) + (op,) for nm, mult, op in (
    ('g_u1', 2, 'fwd'), ('g_u1', 2, 'dgrad'), ('g_r_h1', 2, 'fwd'), ('g_r_h1', 1, 'dgrad'), ('g_d2', 1, 'fwd'), ('g_u2', 2, 'fwd'),
    ('d_d1', 2, 'fwd'), ('d_d3', 2, 'fwd'), ('d_d3', 2, 'dgrad'))]
rng = np.random.default_rng(0)
out4 = (C.c_int * 4)()
for name, Bs, Cin, H, W, Cout, kh, kw, sh, sw, op in CASES:
    G = 2
    x = U.dev(rng.standard_normal((G, Bs, Cin, H, W)).astype(np.float32))
    w = U.dev((rng.standard_normal((G, kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)).astype(np.float32))
    b = U.dev(rng.standard_normal((G, Cout)).astype(np.float32))
    y = U.conv_fwd(x, w, b, kh, kw, sh, sw)
    dy = torch.randn_like(y)
    f = (lambda: U.conv_fwd(x, w, b, kh, kw, sh, sw)) if op == 'fwd' else (lambda: U.conv_dgrad(dy, w, H, W, kh, kw, sh, sw))
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    check(lib.vc_tune_last_igemm_config(out4))
    buf = np.zeros(4 * 8192 + 4, np.uint64)
    assert fn(buf.ctypes.data) == 0
    gx, gy, gz = (int(v) for v in buf[4 * 8192:4 * 8192 + 3])
    n = min(gx * gy * gz, 8192)
    r = buf[:4 * n].reshape(n, 4)
    t0, t1 = r[:, 0].astype(np.int64) / 100.0, r[:, 1].astype(np.int64) / 100.0          # us (100 MHz)
    base = t0.min()
    life = t1 - t0
    hw, xcc = r[:, 3] & np.uint64(0xffffffff), (r[:, 3] >> np.uint64(32)) & np.uint64(0xf)
    cu = ((hw >> np.uint64(8)) & np.uint64(0xf)) | (((hw >> np.uint64(12)) & np.uint64(0x1)) << np.uint64(4)) | (((hw >> np.uint64(13)) & np.uint64(0x7)) << np.uint64(5)) | (xcc << np.uint64(8))
    ids, cnt = np.unique(cu, return_counts=True)
    per_cu = dict(zip(ids.tolist(), cnt.tolist()))
    load = np.array([per_cu[int(c)] for c in cu])
    q = lambda a, p: float(np.percentile(a, p))
    print('%-8s %-5s tile %dx%d/%d grid %dx%dx%d = %d workgroups on %d CUs (per CU: %s) | kernel span %.1f us | entries: p50 %.1f p90 %.1f max %.1f us | '
          'lifetime p10 %.1f p50 %.1f p90 %.1f max %.1f us | mean lifetime by workgroups sharing the CU: %s' % (
              name, op, out4[0], out4[1], out4[2], gx, gy, gz, n, len(ids),
              ', '.join('%d x %d' % (int((cnt == k).sum()), k) for k in sorted(set(cnt.tolist()))),
              t1.max() - base, q(t0 - base, 50), q(t0 - base, 90), (t0 - base).max(), q(life, 10), q(life, 50), q(life, 90), life.max(),
              ', '.join('%d: %.1f' % (k, life[load == k].mean()) for k in sorted(set(load.tolist())))), flush=True)
