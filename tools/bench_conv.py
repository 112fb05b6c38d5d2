"""One convolution shape through the C-ABI, repeated: forward, input gradient, filter gradient (for rocprofv3 --pmc /
--kernel-trace runs on a single kernel).  python tools/bench_conv.py MODE Cin H W Cout kh kw sh sw B [reps]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from tests import gpu_util as U
from voice_converter_cyclegan_amd._lib import lib, check
mode = sys.argv[1]
Cin, H, W, Cout, kh, kw, sh, sw, B = map(int, sys.argv[2:11])
reps = int(sys.argv[11]) if len(sys.argv) > 11 else 10
what = sys.argv[12] if len(sys.argv) > 12 else 'fwd,dgrad,wgrad'
check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1, 'bf16x3': 2}[mode]))
rng = np.random.default_rng(0)
G = 2
x = U.dev(rng.standard_normal((G, B, Cin, H, W)).astype(np.float32))
w = U.dev((rng.standard_normal((G, kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)).astype(np.float32))
b = U.dev(np.zeros((G, Cout), np.float32))
y = U.conv_fwd(x, w, b, kh, kw, sh, sw)
dy = torch.randn_like(y)
flop = 2.0 * G * B * y.shape[3] * y.shape[4] * Cout * kh * kw * Cin
for name, fn in (('fwd', lambda: U.conv_fwd(x, w, b, kh, kw, sh, sw)), ('dgrad', lambda: U.conv_dgrad(dy, w, H, W, kh, kw, sh, sw)),
                 ('wgrad', lambda: U.conv_wgrad(x, dy, kh, kw, sh, sw))):
    if name not in what: continue
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print('%s %s: %.1f us per call (packs and allocation included), %.1f GFLOP -> %.1f TFLOP/s' % (mode, name, us, flop / 1e9, flop / us / 1e6))
