"""Per-workgroup phases of single conv_bf16_kernel launches (needs the build of tools/conv_probe/apply.py).  python tools/conv_probe/probe_conv.py [samples per group]"""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.build()
from tests import gpu_util as U
from voice_converter_cyclegan_amd._lib import lib, check
lib.vc_probe_conv_bf16_read.argtypes = [C.c_void_p]; lib.vc_probe_conv_bf16_read.restype = C.c_int
check(lib.vc_set_math_mode(1))
rng = np.random.default_rng(0)
G = 2
def run(name, Cin, H, W, Cout, kh, kw, sh, sw, B, what):
    x = U.dev(rng.standard_normal((G, B, Cin, H, W)).astype(np.float32))
    w = U.dev((rng.standard_normal((G, kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)).astype(np.float32))
    b = U.dev(np.zeros((G, Cout), np.float32))
    y = U.conv_fwd(x, w, b, kh, kw, sh, sw); dy = torch.randn_like(y)
    fn = {'fwd': lambda: U.conv_fwd(x, w, b, kh, kw, sh, sw), 'dgrad': lambda: U.conv_dgrad(dy, w, H, W, kh, kw, sh, sw), 'wgrad': lambda: U.conv_wgrad(x, dy, kh, kw, sh, sw)}[what]
    for _ in range(30): fn()
    torch.cuda.synchronize()
    buf = np.zeros(8 * 4096 + 8, np.uint64)
    assert lib.vc_probe_conv_bf16_read(buf.ctypes.data) == 0
    gx, gz = int(buf[8 * 4096]), int(buf[8 * 4096 + 1]); n = min(gx * gz, 4096)
    q = buf[:8 * n].reshape(n, 8).astype(np.int64)
    r0, r1 = q[:, 0], q[:, 1]
    span = (r1.max() - r0.min()) / 100.0
    life = (r1 - r0) / 100.0
    pro, loop, epi, cn = q[:, 2], q[:, 3], q[:, 4], q[:, 5]
    cyc = (pro + loop + epi); ghz = cyc / np.maximum(life, 1e-9) / 1e3
    cfg = int(q[0, 6]); BM = cfg // 1000; spec = (cfg % 1000) >= 100; cfg -= 100 if spec else 0
    mf = cn * (BM // 64) * 2 * 4 * 32      # MFMA cycles per wave in the loop: chunks x (MI x 2 tiles x 4 steps) x 32
    flop = 2.0 * G * B * y.shape[3] * y.shape[4] * Cout * kh * kw * Cin
    start = (r0 - r0.min()) / 100.0
    print(('SPEC ' if spec else '     ') + '%-28s %s cfg BM=%d NST=%d grid %d x %d  launch span %.1f us (%.0f TFLOP/s)  WG life med %.1f us (min %.1f max %.1f)  clock %.2f GHz' % (name, what, BM, cfg % 1000, gx, gz, span, flop / span / 1e6, np.median(life), life.min(), life.max(), np.median(ghz)))
    print('    cycles med: prologue %d  K loop %d (%d chunks, MFMA %d = %.2f of the loop)  epilogue %d ; WG starts: %.1f us spread (p50 %.1f p90 %.1f)' % (
        np.median(pro), np.median(loop), int(np.median(cn)), int(np.median(mf)), np.median(mf / np.maximum(loop, 1)), np.median(epi), start.max(), np.median(start), np.percentile(start, 90)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
run('res conv 512->2048 k3 T32', 512, 1, 32, 2048, 1, 3, 1, 1, B, 'fwd')
run('res conv 512->2048 k3 T32', 512, 1, 32, 2048, 1, 3, 1, 1, B, 'dgrad')
run('res conv 512->2048 k3 T32', 512, 1, 32, 2048, 1, 3, 1, 1, B, 'wgrad')
run('res conv 1024->512 k3 T32', 1024, 1, 32, 512, 1, 3, 1, 1, B, 'fwd')
run('down 256->1024 k5 s2 T128', 256, 1, 128, 1024, 1, 5, 1, 2, B, 'fwd')
run('D d3 512->1024(x2) 6x3 s1x2', 512, 6, 32, 2048, 6, 3, 1, 2, 128, 'fwd')
