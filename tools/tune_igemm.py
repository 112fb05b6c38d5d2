"""Measures every implicit-GEMM launch shape of the train step (conv forward + input gradient of every Cin > 1 layer at
the batch sizes the step uses) under each built configuration (tile rows x K groups x split-K), and the filter
gradients (tile columns x K groups x split-K), through the C-ABI with HIP events (vc_prof_*).  Run on the GPU box:

    python tools/tune_igemm.py [--batch 8] [--features 24 --frames 128] [--sweep] [--emit gpurun_out/tuned_b8]

Without --sweep: one line per (layer, op) with the automatic configuration (what the step will run).
With --sweep   : every configuration, the best ones per shape, and (--emit PREFIX) the entries of csrc/vc_tuned_igemm.h
                 (PREFIX_igemm.h) and csrc/vc_tuned_wgrad.h (PREFIX_wgrad.h) for this batch / geometry.

Filter gradients are measured as the step launches them: plain stores (vc_tune_wgrad_store), the generator's layers
over 3B samples (first hop 2B + cycle hop B reduce in ONE launch; a single application of 3B samples has the same chunk
count), the discriminator's over 2B.  Split-K needs a zeroed dW: its zero-fill is added at 3 TB/s.
"""
import argparse
import time
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

CONFIGS = [(bm, kg, sk) for bm in (128, 64) for kg in (1, 2, 4) for sk in (1, 2, 4, 8, 16) if not (bm == 128 and kg == 4)]
CONFIGS_BF16 = [(bm, nst, sk) for bm in (128, 64) for nst in (2, 4) for sk in (1, 2, 4, 8)] + [(256, 3, sk) for sk in (1, 2, 4)]      # conv_bf16_kernel<tile rows, ring stages> x split-K


def layers(F, T):
    """name: (Cin, H, W, Cout, kh, kw, sh, sw) of module.py:148-213 at F MCEPs x T frames"""
    g = {'g_d1': (128, 1, T, 512, 1, 5, 1, 2), 'g_d2': (256, 1, T // 2, 1024, 1, 5, 1, 2),
         'g_r_h1': (512, 1, T // 4, 2048, 1, 3, 1, 1), 'g_r_h2': (1024, 1, T // 4, 512, 1, 3, 1, 1),
         'g_u1': (512, 1, T // 4, 2048, 1, 5, 1, 1), 'g_u2': (512, 1, T // 2, 1024, 1, 5, 1, 1),
         'g_o1': (256, 1, T, F, 1, 15, 1, 1), 'g_h1': (F, 1, T, 256, 1, 15, 1, 1)}
    H1, H2 = -(-F // 2), -(-(-(-F // 2)) // 2)
    d = {'d_d1': (128, F, T // 2, 512, 3, 3, 2, 2), 'd_d2': (256, H1, T // 4, 1024, 3, 3, 2, 2), 'd_d3': (512, H2, T // 8, 2048, 6, 3, 1, 2)}
    return g, d


def time_call(f, kind, reps):
    t_end = time.perf_counter() + 0.03          # >= 30 ms of the launch itself before timing it: the clocks drop while the host prepares a shape (tools/ab_auto_forced.py: 10 ms left the first measurement of a shape 10 % slow)
    while time.perf_counter() < t_end:
        for _ in range(5):
            f()
        torch.cuda.synchronize()
    lib.vc_prof_enable(1)
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    ms, n, fl = C.c_double(0), C.c_longlong(0), C.c_double(0)
    lib.vc_prof_collect(kind, C.byref(ms), C.byref(n), C.byref(fl))
    lib.vc_prof_enable(0)
    return 1e3 * ms.value / reps, int(n.value // reps)      # us per call, launches per call


def _same(n, k, s):
    out = -(-n // s); total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def _pad_rows(ntaps, per_tap):
    rows = ntaps * (-(-per_tap // 32) * 32)
    return -(-rows // 64) * 64


def launch_keys(op, Cin, H, W, Cout, kh, kw, sh, sw, Bs):
    """(a_trans, M, Ptot, krows) of every implicit-GEMM launch the op makes (vc_ops.hip: conv_fwd / conv_dgrad)"""
    OH, pt = _same(H, kh, sh); OW, pl = _same(W, kw, sw)
    if op == 'fwd':
        return [(0, Cout, Bs * OH * OW, _pad_rows(kh * kw, Cin))]
    keys = []
    for ph in range(sh):
        for pw in range(sw):
            Hc, Wc = -(-(H - ph) // sh), -(-(W - pw) // sw)
            nt = sum(1 for i in range(kh) for j in range(kw) if (ph + pt - i) % sh == 0 and (pw + pl - j) % sw == 0)
            if Hc > 0 and Wc > 0 and nt:
                keys.append((1, Cin, Bs * Hc * Wc, _pad_rows(nt, Cout)))
    return keys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--features', type=int, default=24)
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--sweep', action='store_true')
    ap.add_argument('--emit', default='')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default='')
    ap.add_argument('--math', choices=['fp32', 'bf16'], default='fp32', help='bf16: sweep conv_bf16_kernel (the 16-bit data path) instead')
    args = ap.parse_args()
    bf16 = args.math == 'bf16'
    check(lib.vc_set_math_mode(1 if bf16 else 0))
    B = args.batch
    GL, DL = layers(args.features, args.frames)
    rng = np.random.default_rng(0)
    rows, table, wtable, btable = [], [], [], []
    # (name, geometry, samples per group, ops)
    shapes = []
    for nm, g in GL.items():
        shapes.append((nm, g, B, ('fwd', 'dgrad')))              # cycle hop
        shapes.append((nm + '_2B', g, 2 * B, ('fwd', 'dgrad')))   # first hop
        shapes.append((nm + '_3B', g, 3 * B, ('wgrad',)))         # both applications in one filter-gradient launch
    for nm, g in DL.items():
        shapes.append((nm, g, B, ('dgrad',)))                     # generator-loss backward through D's inputs
        shapes.append((nm + '_2B', g, 2 * B, ('fwd', 'dgrad', 'wgrad')))
    for nm, (Cin, H, W, Cout, kh, kw, sh, sw), Bs, oplist in shapes:
        if args.only and not any(o in nm for o in args.only.split(',')):
            continue
        G = 2
        x = U.dev(rng.standard_normal((G, Bs, Cin, H, W)).astype(np.float32))
        w = U.dev((rng.standard_normal((G, kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)).astype(np.float32))
        b = U.dev(rng.standard_normal((G, Cout)).astype(np.float32))
        y = U.conv_fwd(x, w, b, kh, kw, sh, sw)
        dy = torch.randn_like(y)
        zmode = os.environ.get('VC_TUNE_DATA', '')      # timing experiments: 'zero' = all-zero operands, 'zw' = zero filter only, 'one' = constant 1
        if zmode == 'zero':
            x.zero_(); w.zero_(); dy.zero_()
        elif zmode == 'zw':
            w.zero_()
        elif zmode == 'zx':
            x.zero_(); dy.zero_()
        elif zmode == 'one':
            x.fill_(1.0); w.fill_(1.0); dy.fill_(1.0)
        OH, OW = y.shape[3], y.shape[4]
        flops = 2.0 * G * Bs * OH * OW * Cout * kh * kw * Cin
        dw = torch.zeros((G, kh, kw, Cin, Cout), device=U.DEV)

        def wgrad_call():
            check(lib.vc_conv2d_wgrad(U.ptr(x), x[0].numel(), U.ptr(dy), dy[0].numel(), U.ptr(dw), dw[0].numel(),
                                      G, Bs, Cin, H, W, Cout, kh, kw, sh, sw, U.stream()), 'conv_wgrad')
        ops = {'fwd': (lambda: U.conv_fwd(x, w, b, kh, kw, sh, sw), 0), 'dgrad': (lambda: U.conv_dgrad(dy, w, H, W, kh, kw, sh, sw), 0),
               'wgrad': (wgrad_call, 1)}
        for op in oplist:
            f, kind = ops[op]
            if nm.startswith('g_h1') and op == 'fwd' and args.features == 24:
                continue                                    # weight-stationary kernel, not the implicit GEMM
            check(lib.vc_tune_igemm_override(0, 0, 0)); check(lib.vc_tune_wgrad_override(0, 0, 0))
            check(lib.vc_tune_wgrad_store(1 if op == 'wgrad' else 0))
            if bf16:
                check(lib.vc_tune_conv_bf16_override(0, 0, 0))
                kb = (C.c_int * 256)()
                lib.vc_tune_conv_bf16_keys(kb, 64); f(); torch.cuda.synchronize()
                nk = lib.vc_tune_conv_bf16_keys(kb, 64)
                bkeys = sorted({tuple(kb[4 * i:4 * i + 4]) for i in range(min(nk, 64))})
            us, nl = time_call(f, kind, args.reps)
            line = '%-10s %-5s %6.2f GFLOP ideal %6.1f us | auto %7.1f us %6.1f TF (%d launch)' % (
                nm, op, flops / 1e9, flops / 157.3e12 * 1e6, us, flops / us / 1e6, nl)
            if args.sweep and bf16:
                res = []
                zero_us = 4.0 * (dw.numel() if op == 'wgrad' else 0) / 3e6
                for bm, nst, sk in CONFIGS_BF16:
                    check(lib.vc_tune_conv_bf16_override(bm, nst, sk))
                    try:
                        u, _ = time_call(f, kind, args.reps)
                    except Exception:
                        continue
                    res.append((u + (zero_us if sk > 1 else 0.0), bm, nst, sk))
                check(lib.vc_tune_conv_bf16_override(0, 0, 0))
                res.sort()
                line += ' | best ' + '  '.join('%dx%d/%d %.1f' % (bm, nst, sk, u) for u, bm, nst, sk in res[:4])
                if res and bkeys:
                    btable.append((nm, op, res[0], us, bkeys))
            elif args.sweep:
                res = []
                zero_us = 4.0 * dw.numel() / 3e6              # zero-fill of dW at 3 TB/s, paid by split-K filter gradients
                for bm, kg, sk in CONFIGS:
                    if bm == 128 and (Cout if op != 'dgrad' else Cin) < 128:
                        continue
                    check((lib.vc_tune_wgrad_override if op == 'wgrad' else lib.vc_tune_igemm_override)(bm, kg, sk))
                    try:
                        u, _ = time_call(f, kind, args.reps)
                    except Exception:                       # a configuration the launch refuses
                        continue
                    res.append((u + (zero_us if op == 'wgrad' and sk > 1 else 0.0), bm, kg, sk))
                check(lib.vc_tune_igemm_override(0, 0, 0)); check(lib.vc_tune_wgrad_override(0, 0, 0))
                # the tables' choice once more, now in the same thermal / clock state as the forced configurations (the first
                # measurement of a shape follows tens of ms of host work and reads up to 10 % slow: tools/ab_auto_forced.py)
                us = min(us, time_call(f, kind, args.reps)[0])
                line = '%-10s %-5s %6.2f GFLOP ideal %6.1f us | auto %7.1f us %6.1f TF (%d launch)' % (
                    nm, op, flops / 1e9, flops / 157.3e12 * 1e6, us, flops / us / 1e6, nl)
                res.sort()
                line += ' | best ' + '  '.join('%dx%d/%d %.1f' % (bm, kg, sk, u) for u, bm, kg, sk in res[:4])
                if res and op == 'wgrad':
                    ktiles = _pad_rows(kh * kw, Cin) // 64
                    nchunks = -(-(Bs * OH * OW) // 32) if not nm.endswith('_3B') else (-(-(2 * B * OH * OW) // 32) + -(-(B * OH * OW) // 32))
                    wtable.append((nm, res[0], us, (ktiles, Cout, nchunks)))
                elif res:
                    table.append((nm, op, res[0], us, launch_keys(op, Cin, H, W, Cout, kh, kw, sh, sw, Bs)))
            check(lib.vc_tune_wgrad_store(0))
            print(line, flush=True)
            rows.append((nm, op, flops, us))
    tot = {}
    for nm, op, fl, us in rows:
        t = tot.setdefault(op, [0.0, 0.0]); t[0] += fl; t[1] += us
    for op, (fl, us) in tot.items():
        print('sum %-5s %7.1f GFLOP %8.1f us  %6.1f TF' % (op, fl / 1e9, us, fl / us / 1e6))
    if args.emit and table:
        with open(args.emit + '_igemm.h', 'w') as f:
            f.write('    // ---- batch %d per GPU, %dx%d MCEP\n' % (B, args.features, args.frames))
            for nm, op, (u, bm, kg, sk), auto, keys in table:
                f.write('    // %s %s: %.1f us (automatic choice before this sweep: %.1f us)\n' % (nm, op, u, auto))
                for at, M, P, kr in keys:
                    f.write('    {%d, %d, %d, %d, 2, %d, %d, %d},\n' % (at, M, P, kr, bm, kg, sk))
    if args.emit and btable:
        with open(args.emit + '_bf16.h', 'w') as f:
            f.write('    // ---- batch %d per GPU, %dx%d MCEP\n' % (B, args.features, args.frames))
            for nm, op, (u, bm, nst, sk), auto, keys in btable:
                f.write('    // %s %s: %.1f us (rule of conv_bf16_config: %.1f us)\n' % (nm, op, u, auto))
                for M, P, ko, gr in keys:
                    f.write('    {%d, %d, %d, %d, %d, %d, %d},\n' % (M, P, ko, gr, bm, nst, sk))
    if args.emit and wtable:
        with open(args.emit + '_wgrad.h', 'w') as f:
            f.write('    // ---- batch %d per GPU, %dx%d MCEP\n' % (B, args.features, args.frames))
            for nm, (u, bn, kg, sk), auto, (ktiles, N, nchunks) in wtable:
                f.write('    // %s: %.1f us (automatic choice before this sweep: %.1f us)\n' % (nm, u, auto))
                f.write('    {%d, %d, %d, 2, 1, %d, %d, %d},\n' % (ktiles, N, nchunks, bn, kg, sk))


if __name__ == '__main__':
    main()
