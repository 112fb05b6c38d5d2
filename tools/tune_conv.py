"""Times one conv geometry (fwd / dgrad / wgrad) through the C-ABI with HIP events.
usage: python tools/tune_conv.py NAME [reps]   NAME in the table below; env knobs:
VC_IGEMM_TILE=64|128, VC_IGEMM_SPLITK=n, VC_IGEMM_LDS_PAD=bytes."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from tests import gpu_util as U  # noqa: E402
from voice_converter_cyclegan_amd._lib import lib  # noqa: E402

# name: (G, B, Cin, H, W, Cout, kh, kw, sh, sw)
SHAPES = {
    'g_h1': (2, 16, 24, 1, 128, 256, 1, 15, 1, 1), 'g_d1': (2, 16, 128, 1, 128, 512, 1, 5, 1, 2),
    'g_d2': (2, 16, 256, 1, 64, 1024, 1, 5, 1, 2), 'g_r1': (2, 16, 512, 1, 32, 2048, 1, 3, 1, 1),
    'g_r2': (2, 16, 1024, 1, 32, 512, 1, 3, 1, 1), 'g_u1': (2, 16, 512, 1, 32, 2048, 1, 5, 1, 1),
    'g_u2': (2, 16, 512, 1, 64, 1024, 1, 5, 1, 1), 'g_o1': (2, 16, 256, 1, 128, 24, 1, 15, 1, 1),
    'g_r1c': (2, 8, 512, 1, 32, 2048, 1, 3, 1, 1), 'g_r2c': (2, 8, 1024, 1, 32, 512, 1, 3, 1, 1),
    'd_h1': (2, 16, 1, 24, 128, 256, 3, 3, 1, 2), 'd_d1': (2, 16, 128, 24, 64, 512, 3, 3, 2, 2),
    'd_d2': (2, 16, 256, 12, 32, 1024, 3, 3, 2, 2), 'd_d3': (2, 16, 512, 6, 16, 2048, 6, 3, 1, 2),
    'g_h1_b256': (1, 256, 24, 1, 128, 256, 1, 15, 1, 1),
    'r1_b4': (2, 4, 512, 1, 32, 2048, 1, 3, 1, 1),
    'r1_b8': (2, 8, 512, 1, 32, 2048, 1, 3, 1, 1),
    'r1_b12': (2, 12, 512, 1, 32, 2048, 1, 3, 1, 1),
    'r1_b16': (2, 16, 512, 1, 32, 2048, 1, 3, 1, 1),
    'r1_b20': (2, 20, 512, 1, 32, 2048, 1, 3, 1, 1),
    'r1_b24': (2, 24, 512, 1, 32, 2048, 1, 3, 1, 1),
    'r1_b32': (2, 32, 512, 1, 32, 2048, 1, 3, 1, 1),
    'r1_b48': (2, 48, 512, 1, 32, 2048, 1, 3, 1, 1),
    'r1_b64': (2, 64, 512, 1, 32, 2048, 1, 3, 1, 1),
    'r1_b128': (2, 128, 512, 1, 32, 2048, 1, 3, 1, 1),

}


def run(name, reps=20, modes=('fwd', 'dgrad', 'wgrad')):
    G, B, Cin, H, W, Cout, kh, kw, sh, sw = SHAPES[name]
    rng = np.random.default_rng(0)
    x = U.dev(rng.standard_normal((G, B, Cin, H, W)).astype(np.float32))
    w = U.dev((rng.standard_normal((G, kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)).astype(np.float32))
    b = U.dev(rng.standard_normal((G, Cout)).astype(np.float32))
    y = U.conv_fwd(x, w, b, kh, kw, sh, sw)
    dy = torch.randn_like(y)
    flops = 2.0 * G * B * y.shape[3] * y.shape[4] * Cout * kh * kw * Cin
    out = {}
    for mode in modes:
        f = {'fwd': lambda: U.conv_fwd(x, w, b, kh, kw, sh, sw),
             'dgrad': lambda: U.conv_dgrad(dy, w, H, W, kh, kw, sh, sw),
             'wgrad': lambda: U.conv_wgrad(x, dy, kh, kw, sh, sw)}[mode]
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        lib.vc_prof_enable(1)
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        ms, n, fl = C.c_double(0), C.c_longlong(0), C.c_double(0)
        lib.vc_prof_collect(1 if mode == 'wgrad' else 0, C.byref(ms), C.byref(n), C.byref(fl))
        lib.vc_prof_enable(0)
        us = 1e3 * ms.value / reps
        out[mode] = (us, flops / (us * 1e-6) / 1e12)
    return flops, out


if __name__ == '__main__':
    names = sys.argv[1].split(',') if len(sys.argv) > 1 else list(SHAPES)
    modes = tuple(sys.argv[2].split(',')) if len(sys.argv) > 2 else ('fwd', 'dgrad', 'wgrad')
    knobs = {k: v for k, v in os.environ.items() if k.startswith('VC_')}
    for nm in names:
        fl, out = run(nm, modes=modes)
        print('%-10s %6.2f GFLOP ideal %6.1f us | ' % (nm, fl / 1e9, fl / 157.3e12 * 1e6) +
              '  '.join('%s %7.1f us %5.1f TF' % (m, *out[m]) for m in out) + '  ' + str(knobs), flush=True)
