"""The convolution launches of one train step (reference model.py:110-125 over module.py:148-213) as SHAPES: which layer,
which operator, how many samples per group -- and the keys under which csrc/vc_tuned_igemm.h / vc_tuned_wgrad.h list their
measured configurations.  Shared by tools/tune_igemm.py (which measures them), tests/test_tuned_tables_cpu.py (every benched
shape must hit its measured row) and tests/test_tuned_tables_gpu.py (every such launch against the oracle).  No torch, no GPU.
"""


def layers(F, T):
    """name: (Cin, H, W, Cout, kh, kw, sh, sw) of module.py:148-213 at F MCEPs x T frames (value + gate convs stored as one)"""
    g = {'g_d1': (128, 1, T, 512, 1, 5, 1, 2), 'g_d2': (256, 1, T // 2, 1024, 1, 5, 1, 2),
         'g_r_h1': (512, 1, T // 4, 2048, 1, 3, 1, 1), 'g_r_h2': (1024, 1, T // 4, 512, 1, 3, 1, 1),
         'g_u1': (512, 1, T // 4, 2048, 1, 5, 1, 1), 'g_u2': (512, 1, T // 2, 1024, 1, 5, 1, 1),
         'g_o1': (256, 1, T, F, 1, 15, 1, 1), 'g_h1': (F, 1, T, 256, 1, 15, 1, 1)}
    H1, H2 = -(-F // 2), -(-(-(-F // 2)) // 2)
    d = {'d_d1': (128, F, T // 2, 512, 3, 3, 2, 2), 'd_d2': (256, H1, T // 4, 1024, 3, 3, 2, 2), 'd_d3': (512, H2, T // 8, 2048, 6, 3, 1, 2)}
    return g, d


def same_pad(n, k, s):
    """TF 'SAME': output size and low-side padding (module.py:32,54 padding='same')"""
    out = -(-n // s); total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def pad_rows(ntaps, per_tap):
    """K rows of a table: every tap padded to whole 32-row chunks, the table to whole 64-row tiles (vc_tables.hip: upload_table)"""
    rows = ntaps * (-(-per_tap // 32) * 32)
    return -(-rows // 64) * 64


def launch_keys(op, Cin, H, W, Cout, kh, kw, sh, sw, Bs):
    """(a_trans, M, Ptot, krows) of every implicit-GEMM launch the op makes (vc_ops.hip: conv_fwd / conv_dgrad)"""
    OH, pt = same_pad(H, kh, sh); OW, pl = same_pad(W, kw, sw)
    if op == 'fwd':
        return [(0, Cout, Bs * OH * OW, pad_rows(kh * kw, Cin))]
    keys = []
    for ph in range(sh):
        for pw in range(sw):
            Hc, Wc = -(-(H - ph) // sh), -(-(W - pw) // sw)
            nt = sum(1 for i in range(kh) for j in range(kw) if (ph + pt - i) % sh == 0 and (pw + pl - j) % sw == 0)
            if Hc > 0 and Wc > 0 and nt:
                keys.append((1, Cin, Bs * Hc * Wc, pad_rows(nt, Cout)))
    return keys


def wgrad_key(Cin, H, W, Cout, kh, kw, sh, sw, batches):
    """(K tiles, Cout, 32-pixel chunks) of the filter-gradient launch that reduces the applications in `batches` (vc_ops.hip:
    conv_wgrad; the generator's first hop 2B and cycle hop B go out as ONE launch)"""
    OH, _ = same_pad(H, kh, sh); OW, _ = same_pad(W, kw, sw)
    return (pad_rows(kh * kw, Cin) // 64, Cout, sum(-(-(b * OH * OW) // 32) for b in batches))


def step_launches(F, T, B, lambda_identity_zero=False):
    """[(name, op, geometry, samples per group [, ...])] of one step at batch B: the forward and input-gradient launches of the
    first hop (2B), the cycle hop (B), the discriminators (2B forward and discriminator-loss backward, B for the generator-
    loss backward), and the filter gradients ((2B, B) for generator layers, (2B,) for discriminator layers).  The generator's
    h1 forward (24 MCEPs: gated_conv1d_glu_kernel) and the discriminator's Cin = 1 layer are not implicit-GEMM launches."""
    GL, DL = layers(F, T)
    Bf = B if lambda_identity_zero else 2 * B        # train.py:98-99: the identity half is not propagated back once lambda_identity = 0
    out = []
    for nm, g in GL.items():
        fwd_igemm = not (nm == 'g_h1' and F == 24)
        if fwd_igemm:
            out.append((nm, 'fwd', g, (B,))); out.append((nm + '_2B', 'fwd', g, (2 * B,)))
        out.append((nm, 'dgrad', g, (B,)))
        if nm != 'g_h1':                              # the first hop's input needs no gradient
            out.append((nm + '_2B', 'dgrad', g, (Bf,)))
        out.append((nm + '_3B', 'wgrad', g, (Bf, B)))
    for nm, g in DL.items():
        out.append((nm, 'dgrad', g, (B,)))
        out.append((nm + '_2B', 'fwd', g, (2 * B,)))
        out.append((nm + '_2B', 'dgrad', g, (2 * B,)))
        out.append((nm + '_2B', 'wgrad', g, (2 * B,)))
    return out
