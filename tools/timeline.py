"""Per-workgroup timeline of one igemm_kl launch (debug build with -DVC_DBG_TIMELINE): s_memtime at entry, after the
prologue, after the K loop, at exit.  usage: VC_EXTRA_FLAGS=-DVC_DBG_TIMELINE (build), then python tools/timeline.py NAME [fwd|dgrad]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.tune_conv as tc
from voice_converter_cyclegan_amd._lib import lib
name = sys.argv[1]; mode = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
fl, out = tc.run(name, reps=5, modes=(mode,))
print(name, mode, '%.1f us' % out[mode][0])
n = 4 * 8192
buf = (C.c_ulonglong * n)()
lib.vc_dbg_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.vc_dbg_timeline(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
t = t[t[:, 3] > 0]
t0 = t[:, 0].min()
rel = (t - t0) / float(os.environ.get("TICKS_PER_US", "100"))       # s_memtime ticks -> us (100 MHz reference clock by default)
print('workgroups', len(t))
for k, nm in enumerate(('entry', 'after prologue', 'after K loop', 'exit')):
    print('%-15s min %7.2f  median %7.2f  max %7.2f us' % (nm, rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
d = np.diff(rel, axis=1)
for k, nm in enumerate(('prologue', 'K loop', 'epilogue')):
    print('%-15s min %7.2f  median %7.2f  max %7.2f us' % (nm, d[:, k].min(), np.median(d[:, k]), d[:, k].max()))
