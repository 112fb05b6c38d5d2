"""A/B on the GPU box: one convolution launch shape timed with the configuration the tables choose ('auto') and with the SAME
configuration forced through vc_tune_igemm_override, alternating -- the two must agree; a gap is a measurement-order artefact."""
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
from tools.tune_igemm import time_call  # noqa: E402

G, Bs, Cin, H, W, Cout, kh, kw, sh, sw = 2, 16, 512, 1, 32, 1024, 1, 5, 1, 1       # g_u1_2B at batch 8
rng = np.random.default_rng(0)
x = U.dev(rng.standard_normal((G, Bs, Cin, H, W)).astype(np.float32))
w = U.dev((rng.standard_normal((G, kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)).astype(np.float32))
b = U.dev(rng.standard_normal((G, Cout)).astype(np.float32))
f = lambda: U.conv_fwd(x, w, b, kh, kw, sh, sw)
out4 = (C.c_int * 4)()
f(); check(lib.vc_tune_last_igemm_config(out4)); cfg = tuple(out4)
print('table choice', cfg)
for rnd in range(4):
    check(lib.vc_tune_igemm_override(0, 0, 0)); a, _ = time_call(f, 0, 20)
    check(lib.vc_tune_igemm_override(cfg[0], cfg[1], cfg[2])); fo, _ = time_call(f, 0, 20)
    check(lib.vc_tune_igemm_override(0, 0, 0)); f(); check(lib.vc_tune_last_igemm_config(out4))
    print('round %d: auto %.1f us, forced %.1f us (auto config again: %s)' % (rnd, a, fo, tuple(out4)))
