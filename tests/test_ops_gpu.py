"""-m gpu: operator-level parity of the HIP kernels (through the C-ABI) against the
numpy oracle (oracle/cyclegan_oracle.py), fp32, tolerance 1e-4 relative to the
tensor's max magnitude (north_star asks 1e-3; exact-fp32 MFMA leaves ~1e-6)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _oracle():
    from oracle import cyclegan_oracle as O
    return O


# (Cin, H, W, Cout, kh, kw, sh, sw, B) -- every conv geometry of both networks plus ragged cases
CONV_CASES = [
    (24, 1, 128, 256, 1, 15, 1, 1, 2),     # G.h1 (value+gate)
    (128, 1, 128, 512, 1, 5, 1, 2, 2),     # G.d1
    (256, 1, 64, 1024, 1, 5, 1, 2, 2),     # G.d2
    (512, 1, 32, 2048, 1, 3, 1, 1, 3),     # G.rN.h1
    (1024, 1, 32, 512, 1, 3, 1, 1, 2),     # G.rN.h2
    (512, 1, 32, 2048, 1, 5, 1, 1, 1),     # G.u1
    (512, 1, 64, 1024, 1, 5, 1, 1, 2),     # G.u2
    (256, 1, 128, 24, 1, 15, 1, 1, 2),     # G.o1
    (1, 24, 128, 256, 3, 3, 1, 2, 2),      # D.h1
    (128, 24, 64, 512, 3, 3, 2, 2, 1),     # D.d1
    (256, 12, 32, 1024, 3, 3, 2, 2, 2),    # D.d2
    (512, 6, 16, 2048, 6, 3, 1, 2, 2),     # D.d3
    (36, 1, 20, 64, 1, 15, 1, 1, 3),       # 36-dim MCEP, ragged T
    (32, 9, 10, 64, 3, 3, 2, 2, 2),        # odd H (TF SAME pads differ by parity)
    (64, 1, 132, 128, 1, 5, 1, 2, 1),      # T=132
    (16, 5, 7, 32, 6, 3, 1, 2, 2),         # odd everything
    (128, 1, 8, 512, 1, 5, 1, 2, 3),       # tiny T (T=8 utterance): W smaller than the kernel
    (256, 1, 4, 1024, 1, 5, 1, 2, 3),
    (512, 1, 2, 2048, 1, 3, 1, 1, 3),
    (512, 1, 2, 2048, 1, 5, 1, 1, 3),
    (256, 1, 8, 24, 1, 15, 1, 1, 3),
    (24, 1, 8, 256, 1, 15, 1, 1, 3),
    (512, 1, 32, 2048, 1, 3, 1, 1, 8),     # residual h1 at the B=8 step's sizes (cycle hop)
    (1024, 1, 32, 512, 1, 3, 1, 1, 16),    # residual h2, first hop (2B)
    (512, 1, 32, 2048, 1, 5, 1, 1, 4),     # up-sample 1 (k = 5)
    (512, 1, 64, 1024, 1, 5, 1, 1, 8),     # up-sample 2 (T = 64)
    (64, 1, 128, 48, 1, 3, 1, 1, 2),       # T = 128, narrow output
    (32, 1, 16, 80, 1, 5, 1, 1, 4),        # T = 16: one pixel tile per sample
    (1, 9, 10, 64, 3, 3, 2, 2, 3),         # Cin = 1 -> vector-ALU kernels (c1_*): strided both ways, odd H
    (1, 1, 40, 40, 1, 5, 1, 1, 2),         # Cin = 1, 1-D, Cout not a multiple of 32
    (1, 5, 6, 8, 2, 2, 1, 1, 1),           # Cin = 1, even kernel (SAME pads only right/bottom)
    # the shapes bench.py TIMES (timed paths must be tested paths): the named north-star kernel, G.h1 at B=256
    # (module.py:161-162; wsconv1d_kernel walks 8 tiles per workgroup), and an 8 x 2048-frame inference batch (16 tiles)
    (24, 1, 128, 256, 1, 15, 1, 1, 256),
    (24, 1, 2048, 256, 1, 15, 1, 1, 8),
    (24, 1, 132, 256, 1, 15, 1, 1, 5),     # ragged last tile of every sample (T % 128 != 0), odd tile count
    (36, 1, 256, 256, 1, 15, 1, 1, 2),     # config 5's h1 (36 MCEPs): taps of 36 rows padded to 64 in the K table
    (256, 1, 256, 36, 1, 15, 1, 1, 2),     # config 5's o1
    (1024, 1, 32, 512, 1, 3, 1, 1, 8),     # residual h2, cycle hop: K groups inside the workgroup + split-K
    (256, 1, 128, 24, 1, 15, 1, 1, 8),     # o1 at B = 8: 24-row taps (padded to 32) in the input-gradient table
    # rowskip (WgradArgs::rowskip): D.d3's 6 tap rows on 6 image rows, SAME pads (2, 3) -- the filter gradient enumerates its
    # pixels (row, sample, column) and walks only the output rows a K tile's taps reach
    (512, 6, 16, 2048, 6, 3, 1, 2, 8),     # the B = 8 step's size: 64 pixels per output row = one tile per row
    (512, 9, 32, 2048, 6, 3, 1, 2, 3),     # 36 MCEPs (9 rows), odd batch: 48 pixels per row, tiles straddle rows
    (64, 6, 16, 128, 6, 3, 1, 2, 16),      # 128 pixels per row: two tiles per row
]


def _np_conv(O, x_cl, w, b, kh, sh, sw):
    if kh == 1 and x_cl.shape[1] == 1:
        y, cache = O.conv1d_fwd(x_cl[:, 0], w[0], b, sw)
        return y[:, None], cache, True
    y, cache = O.conv2d_fwd(x_cl, w, b, (sh, sw))
    return y, cache, False


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(built_lib, case):
    _conv_check(case)


@pytest.mark.parametrize('on', [1, 0])
@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[4] == 6] + [(16, 5, 7, 32, 6, 3, 1, 2, 2)])
def test_conv_rowskip_modes(built_lib, case, on):
    """WgradArgs::rowskip (filter gradient: launch pixels enumerated (row, sample, column); a K tile reduces only over the
    output rows its taps reach): on by default (vc_ops.hip: rowskip_pays); both settings are forced here (vc_tune_rowskip) on the
    6-row-kernel geometries it applies to (module.py:208, D.d3)."""
    from voice_converter_cyclegan_amd._lib import lib, check
    check(lib.vc_tune_rowskip(on))
    try:
        _conv_check(case)
    finally:
        check(lib.vc_tune_rowskip(1))


# bf16 MFMA modes (vc_set_math_mode): 'bf16' is checked against the oracle with the SAME operand rounding (so only the
# fp32 accumulation order differs -> same tolerance as fp32); 'bf16x3' against the unrounded oracle (2^-16 per product).
MATH_CASES = [CONV_CASES[i] for i in (0, 1, 3, 4, 6, 7, 8, 9, 11, 13, 15, 16, 22, 23, 28)]


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('case', MATH_CASES)
def test_conv_bf16_modes(built_lib, case, mode):
    """operator-level calls carry no plan: they run in the calling thread's default mode (vc_set_math_mode)"""
    from voice_converter_cyclegan_amd._lib import lib, check
    O = _oracle()
    check(lib.vc_set_math_mode({'bf16': 1, 'bf16x3': 2}[mode]))
    O.set_operand_rounding('bf16' if mode == 'bf16' else None)
    try:
        assert lib.vc_get_math_mode() == {'bf16': 1, 'bf16x3': 2}[mode]
        _conv_check(case)
    finally:
        O.set_operand_rounding(None)
        check(lib.vc_set_math_mode(0))
    assert lib.vc_set_math_mode(4) != 0 and lib.vc_get_math_mode() == 0      # unknown mode is refused (3: f16, the gated conv1d + GLU forward only)


@pytest.mark.parametrize('cfg', [(128, 2, 1), (128, 4, 1), (64, 2, 1), (64, 4, 1), (128, 4, 2), (64, 2, 4), (256, 3, 1), (256, 3, 2),
                                 # the one-workgroup-per-CU configurations once more in their 256-thread form (production runs them
                                 # wave-specialised: 4 multiplying + 4 copying waves; vc_tune_conv_bf16_spec(0) is the A/B switch)
                                 (128, 4, 1, 'plain'), (64, 4, 2, 'plain'), (256, 3, 1, 'plain')])
@pytest.mark.parametrize('case', [(128, 1, 128, 512, 1, 5, 1, 2, 4),      # G.d1: stride 2 -> two input-gradient parity classes
                                  (512, 1, 32, 2048, 1, 3, 1, 1, 8),      # residual h1
                                  (256, 12, 32, 1024, 3, 3, 2, 2, 2),     # D.d2: 2-D, four parity classes
                                  (64, 1, 132, 192, 1, 5, 1, 1, 3)])      # ragged: M and the pixel count are not tile multiples
def test_conv_bf16_data_path_every_configuration(built_lib, case, cfg):
    """conv_bf16_kernel<tile rows, LDS ring stages> x split-K (vc_conv_bf16.hip): every built configuration against the
    oracle with bf16 operand rounding (module.py:32-40, 54-62 and their input gradients).  The automatic choice only
    reaches (128, 2) and (64, *) at batch >= 32 -- the sizes bench.py --math bf16 --batch 64 times -- so they are forced
    here (vc_tune_conv_bf16_override) on sizes the oracle finishes in seconds."""
    from voice_converter_cyclegan_amd._lib import lib, check
    O = _oracle()
    check(lib.vc_set_math_mode(1))
    check(lib.vc_tune_conv_bf16_override(*cfg[:3]))
    check(lib.vc_tune_conv_bf16_spec(0 if len(cfg) > 3 else 1))
    O.set_operand_rounding('bf16')
    try:
        _conv_check(case)
    finally:
        O.set_operand_rounding(None)
        check(lib.vc_tune_conv_bf16_override(0, 0, 0))
        check(lib.vc_tune_conv_bf16_spec(-1))
        check(lib.vc_set_math_mode(0))
    assert lib.vc_tune_conv_bf16_override(96, 2, 1) != 0          # not a built configuration


FP32_CFG_CASES = [(128, 1, 128, 512, 1, 5, 1, 2, 4),      # G.d1: stride 2 -> two input-gradient parity classes (fused into one launch)
                  (512, 1, 32, 2048, 1, 3, 1, 1, 8),      # residual h1
                  (256, 12, 32, 1024, 3, 3, 2, 2, 2),     # D.d2: 2-D, four parity classes
                  (64, 1, 132, 192, 1, 5, 1, 1, 3)]       # ragged: M and the pixel count are not tile multiples


@pytest.mark.parametrize('cfg', [(64, 1, 1), (64, 2, 1), (64, 4, 1), (128, 1, 1), (128, 2, 1),
                                 (64, 1, 4), (64, 2, 2), (64, 4, 2), (128, 1, 2), (128, 2, 4)])
@pytest.mark.parametrize('case', FP32_CFG_CASES)
def test_conv_fp32_every_configuration(built_lib, case, cfg):
    """igemm_kernel<tile rows, transposed?, K groups> x split-K (vc_conv_igemm.hip) and wgrad_kernel<tile columns, K groups> x
    split-K (vc_conv_wgrad.hip): EVERY built fp32 configuration against the oracle (module.py:32-40, 54-62 and their input
    / filter gradients) at 1e-4.  The B = 8 step takes its configurations from the measured table (vc_tuned_igemm.h); other
    batches (bench.py --batch 32 / 64 / 256, 36x256) go by the cost model, which reaches 128-row tiles and other K-group /
    split-K combinations -- they are forced here (vc_tune_igemm_override / vc_tune_wgrad_override) on sizes the oracle
    finishes in seconds, so no timed instantiation is an untested one."""
    from voice_converter_cyclegan_amd._lib import lib, check
    check(lib.vc_tune_igemm_override(*cfg))
    check(lib.vc_tune_wgrad_override(*cfg))
    try:
        _conv_check(case)
    finally:
        check(lib.vc_tune_igemm_override(0, 0, 0))
        check(lib.vc_tune_wgrad_override(0, 0, 0))
    assert lib.vc_tune_igemm_override(96, 2, 1) != 0 and lib.vc_tune_wgrad_override(128, 4, 1) != 0      # not built


@pytest.mark.parametrize('case', [CONV_CASES[i] for i in (1, 3, 8, 10, 11, 15, 27, 28)])
def test_conv_deterministic_mode(built_lib, case):
    """vc_set_deterministic: no split-K across workgroups, gather-form Cin = 1 input gradient, one block per Cin = 1 filter-
    gradient element -- same oracle tolerance, and two calls on the same inputs are BITWISE equal (model.py:110-125 run twice
    on one feed gives one answer in the reference's own deterministic CPU kernels)."""
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, check
    check(lib.vc_set_deterministic(1))
    try:
        assert lib.vc_get_deterministic() == 1
        _conv_check(case)
        Cin, H, W, Cout, kh, kw, sh, sw, B = case
        rng = np.random.default_rng(5)
        x = U.dev(rng.standard_normal((2, B, Cin, H, W)).astype(np.float32))
        w = U.dev((rng.standard_normal((2, kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)).astype(np.float32))
        b = U.dev(rng.uniform(-0.5, 0.5, (2, Cout)).astype(np.float32))
        y1 = U.conv_fwd(x, w, b, kh, kw, sh, sw); y2 = U.conv_fwd(x, w, b, kh, kw, sh, sw)
        assert torch.equal(y1, y2)
        dy = torch.randn_like(y1)
        assert torch.equal(U.conv_dgrad(dy, w, H, W, kh, kw, sh, sw), U.conv_dgrad(dy, w, H, W, kh, kw, sh, sw))
        assert torch.equal(U.conv_wgrad(x, dy, kh, kw, sh, sw), U.conv_wgrad(x, dy, kh, kw, sh, sw))
    finally:
        check(lib.vc_set_deterministic(0))


def test_bf16_mode_is_really_bf16(built_lib):
    """The bf16 mode must differ from fp32 by bf16-sized errors (guards against a silent fp32 fallback)."""
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, check
    rng = np.random.default_rng(3)
    x = U.dev(rng.standard_normal((1, 2, 512, 1, 32)).astype(np.float32))
    w = U.dev((rng.standard_normal((1, 1, 3, 512, 2048)) / 40).astype(np.float32))
    b = U.dev(np.zeros((1, 2048), np.float32))
    y32 = U.conv_fwd(x, w, b, 1, 3, 1, 1).cpu().numpy()
    errs = {}
    for mode in (1, 2):
        check(lib.vc_set_math_mode(mode))
        try:
            errs[mode] = U.rel_err(U.conv_fwd(x, w, b, 1, 3, 1, 1).cpu().numpy(), y32)
        finally:
            check(lib.vc_set_math_mode(0))
    assert 1e-4 < errs[1] < 2e-2, errs
    assert errs[2] < 1e-4, errs


def _conv_check(case):
    from tests import gpu_util as U
    O = _oracle()
    Cin, H, W, Cout, kh, kw, sh, sw, B = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    G = 2
    xs = rng.standard_normal((G, B, H, W, Cin)).astype(np.float32)          # channels-last for the oracle
    ws = (rng.standard_normal((G, kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)).astype(np.float32)
    bs = rng.uniform(-0.5, 0.5, (G, Cout)).astype(np.float32)
    ys, dxs, dws, dys = [], [], [], []
    for g in range(G):
        y, cache, one_d = _np_conv(O, xs[g], ws[g], bs[g], kh, sh, sw)
        dy = rng.standard_normal(y.shape).astype(np.float32)
        if one_d:
            dx, dw, _ = O.conv1d_bwd(dy[:, 0], ws[g][0], cache)
            dx, dw = dx[:, None], dw[None]
        else:
            dx, dw, _ = O.conv2d_bwd(dy, ws[g], cache)
        ys.append(y); dxs.append(dx); dws.append(dw); dys.append(dy)
    to_cf = lambda a: np.ascontiguousarray(np.transpose(np.stack(a), (0, 1, 4, 2, 3)))   # [G,B,C,H,W]
    x_d, w_d, b_d = U.dev(to_cf(list(xs))), U.dev(ws), U.dev(bs)
    y_d = U.conv_fwd(x_d, w_d, b_d, kh, kw, sh, sw)
    e = U.rel_err(y_d.cpu().numpy(), to_cf(ys))
    assert e < TOL, 'fwd rel err %g' % e
    dy_d = U.dev(to_cf(dys))
    dx_d = U.conv_dgrad(dy_d, w_d, H, W, kh, kw, sh, sw)
    e = U.rel_err(dx_d.cpu().numpy(), to_cf(dxs))
    assert e < TOL, 'dgrad rel err %g' % e
    # addend + accumulate semantics
    add = rng.standard_normal(to_cf(dxs).shape).astype(np.float32)
    init = rng.standard_normal(to_cf(dxs).shape).astype(np.float32)
    dx2 = U.conv_dgrad(dy_d, w_d, H, W, kh, kw, sh, sw, addend=U.dev(add), init=U.dev(init))
    e = U.rel_err(dx2.cpu().numpy(), to_cf(dxs) + add + init)
    assert e < TOL, 'dgrad(+addend,+accumulate) rel err %g' % e
    dw_d = U.conv_wgrad(x_d, dy_d, kh, kw, sh, sw)
    e = U.rel_err(dw_d.cpu().numpy(), np.stack(dws))
    assert e < TOL, 'wgrad rel err %g' % e
    # the operator ALWAYS accumulates: a second call doubles the result (atomics path on a non-zero destination)
    check_acc = U.conv_wgrad(x_d, dy_d, kh, kw, sh, sw, init=dw_d)
    e = U.rel_err(check_acc.cpu().numpy(), 2 * np.stack(dws))
    assert e < TOL, 'wgrad accumulate rel err %g' % e


@pytest.mark.parametrize('packed', [False, True])
@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'f16'])
@pytest.mark.parametrize('B,T', [(2, 128), (256, 128), (3, 132), (8, 2048), (1, 4), (5, 260)])
def test_gated_conv1d_glu_fused(built_lib, B, T, mode, packed):
    """vc_gated_conv1d_glu_fwd / vc_gated_conv1d_glu_prepack + vc_gated_conv1d_glu_fwd_packed: the generator's h1 block
    (module.py:161-163) in one kernel, against conv1d + conv1d + gated_linear_layer of the oracle.  (256, 128) is the shape
    bench.py times as the north-star kernel; (8, 2048) an inference batch; T = 132 / 260 / 4: ragged last tile (a partial
    32-pixel block), tile smaller than the kernel.  Mode f16 (math mode 3, IEEE-half operands, fp32 accumulate) is the 16-bit kernel
    that HOLDS the north star's "fp32 parity within 1e-3": asserted below against the UNROUNDED oracle."""
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, ptr, check
    O = _oracle()
    rng = np.random.default_rng(B * 7 + T)
    G, C = 2, 128
    x = rng.standard_normal((G, B, T, 24)).astype(np.float32)
    w = (rng.standard_normal((G, 15, 24, 2 * C)) / np.sqrt(360)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (G, 2 * C)).astype(np.float32)
    check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1, 'f16': 3}[mode]))
    O.set_operand_rounding(mode if mode != 'fp32' else None)
    try:
        ref = []
        for g in range(G):
            v, _ = O.conv1d_fwd(x[g], w[g][:, :, :C], b[g][:C], 1)
            gt, _ = O.conv1d_fwd(x[g], w[g][:, :, C:], b[g][C:], 1)
            ref.append(np.transpose(O.glu_fwd(v, gt)[0], (0, 2, 1)))
        x_d = U.dev(np.ascontiguousarray(np.transpose(x, (0, 1, 3, 2))))
        w_d, b_d = U.dev(w), U.dev(b)
        out = torch.full((G, B, C, T), float('nan'), device=U.DEV)
        if packed:
            nb = lib.vc_gated_conv1d_glu_image_bytes(24, 15, 2 * C)
            assert nb > 0 and nb % 16 == 0
            img = torch.zeros((G, nb), dtype=torch.uint8, device=U.DEV)
            check(lib.vc_gated_conv1d_glu_prepack(ptr(w_d), w_d[0].numel(), ptr(img), nb, G, 24, 15, 2 * C, U.stream()))
            check(lib.vc_gated_conv1d_glu_fwd_packed(ptr(x_d), x_d[0].numel(), ptr(img), nb, ptr(b_d), 2 * C,
                                                     ptr(out), out[0].numel(), G, B, 24, T, 2 * C, 15, U.stream()))
        else:
            check(lib.vc_gated_conv1d_glu_fwd(ptr(x_d), x_d[0].numel(), ptr(w_d), w_d[0].numel(), ptr(b_d), 2 * C,
                                              ptr(out), out[0].numel(), G, B, 24, T, 2 * C, 15, U.stream()))
        e = U.rel_err(out.cpu().numpy(), np.stack(ref))
    finally:
        O.set_operand_rounding(None)
        check(lib.vc_set_math_mode(0))
    assert e < (TOL if mode == 'fp32' else 2e-4), 'rel err %g' % e
    if mode != 'fp32':
        # ... and against the UNROUNDED oracle: what bf16 operands cost against the reference's fp32 arithmetic.  K = 360
        # products of two RNE-rounded operands: measured 1.9e-3 - 2.6e-3 of the tensor's maximum (2.4e-3 rel-L2) -- ABOVE the north
        # star's 1e-3, so the bf16 timings of this kernel are reported beside the north-star figure, never as it; the
        # bound asserted here is the arithmetic's own (a regression to double rounding or bf16 accumulation would land at > 5e-3).
        exact = []
        for g in range(G):
            v, _ = O.conv1d_fwd(x[g], w[g][:, :, :C], b[g][:C], 1)
            gt, _ = O.conv1d_fwd(x[g], w[g][:, :, C:], b[g][C:], 1)
            exact.append(np.transpose(O.glu_fwd(v, gt)[0], (0, 2, 1)))
        e32 = U.rel_err(out.cpu().numpy(), np.stack(exact))
        print('gated conv1d + GLU, %s operands vs the fp32 oracle: B=%d T=%d packed=%s  max-rel %.3e' % (mode, B, T, packed, e32))
        if mode == 'bf16':
            assert 2e-4 < e32 < 5e-3, 'bf16 operands vs the unrounded oracle: %g' % e32
        else:
            # IEEE half: 10-bit mantissa, an eighth of bf16's operand error -- INSIDE the north star's 1e-3 against fp32 arithmetic
            assert 2e-5 < e32 < 1e-3, 'f16 operands vs the unrounded oracle: %g' % e32


def test_f16_mode_is_one_operator_only(built_lib):
    """Math mode 3 (IEEE-half operands) is built for the gated conv1d + GLU forward; every other entry point REFUSES it with an error code
    instead of computing in some other arithmetic (no silent fall-back), and so does plan creation.  module.py:161-163."""
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, check, VcError
    import ctypes as C
    x = torch.randn((1, 1, 64, 1, 32), device=U.DEV); w = torch.randn((1, 1, 3, 64, 64), device=U.DEV); b = torch.zeros((1, 64), device=U.DEV)
    check(lib.vc_set_math_mode(3))
    try:
        assert lib.vc_get_math_mode() == 3
        with pytest.raises(VcError, match='math mode 3'):
            U.conv_fwd(x, w, b, 1, 3, 1, 1)
        with pytest.raises(VcError, match='math mode 3'):
            U.conv_dgrad(torch.randn((1, 1, 64, 1, 32), device=U.DEV), w, 1, 32, 1, 3, 1, 1)
        with pytest.raises(VcError, match='math mode 3'):
            U.conv_wgrad(x, torch.randn((1, 1, 64, 1, 32), device=U.DEV), 1, 3, 1, 1)
        plan = C.c_void_p()
        assert lib.vc_plan_create(24, 128, C.byref(plan)) != 0 and b'math mode 3' in lib.vc_last_error()
    finally:
        check(lib.vc_set_math_mode(0))
    assert lib.vc_set_math_mode(4) != 0
    # a bf16 and an f16 fragment image have the same size and layout: the library remembers which mode packed an image and refuses the other
    from voice_converter_cyclegan_amd._lib import ptr
    G, B, T = 1, 2, 128
    xg = torch.randn((G, B, 24, T), device=U.DEV); wg = torch.randn((G, 15, 24, 256), device=U.DEV) * 0.05; bg = torch.zeros((G, 256), device=U.DEV)
    out = torch.empty((G, B, 128, T), device=U.DEV)
    try:
        check(lib.vc_set_math_mode(1))
        nb = lib.vc_gated_conv1d_glu_image_bytes(24, 15, 256)
        img = torch.zeros((G, nb), dtype=torch.uint8, device=U.DEV)
        check(lib.vc_gated_conv1d_glu_prepack(ptr(wg), wg[0].numel(), ptr(img), nb, G, 24, 15, 256, U.stream()))
        call = lambda: lib.vc_gated_conv1d_glu_fwd_packed(ptr(xg), xg[0].numel(), ptr(img), nb, ptr(bg), 256, ptr(out), out[0].numel(), G, B, 24, T, 256, 15, U.stream())
        assert call() == 0
        check(lib.vc_set_math_mode(3))
        assert lib.vc_gated_conv1d_glu_image_bytes(24, 15, 256) == nb
        assert call() != 0 and b'built' in lib.vc_last_error() and b'mode 1' in lib.vc_last_error()
        check(lib.vc_gated_conv1d_glu_prepack(ptr(wg), wg[0].numel(), ptr(img), nb, G, 24, 15, 256, U.stream()))     # re-packed in the new mode: accepted
        assert call() == 0
        check(lib.vc_set_math_mode(1))
        assert call() != 0
    finally:
        check(lib.vc_set_math_mode(0))


@pytest.mark.parametrize('mode', [1, 3])
@pytest.mark.parametrize('packed', [False, True])
def test_gated_conv1d_glu_is_bitwise_repeatable(built_lib, packed, mode):
    """The launch is deterministic (no atomics, fixed summation order): 3000 launches of the 16-bit kernel (bf16; f16 = math mode 3, the
    same schedule) at B = 1024 must each equal the
    first one bit for bit.  This is the shape and path on which round 5 found a timing-dependent failure that an oracle comparison of
    single launches cannot see reliably (0.1 - 0.4 % of the launches: frames of the second tile staged from register copies made while
    their loads were in flight; csrc/vc_conv_glu.hip prologue, tests/test_glu_isa_cpu.py) -- with 3000 launches a failure rate of
    0.1 % is caught with probability 0.95.  module.py:161-163."""
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, ptr, check
    torch.manual_seed(5)
    G, B, T = 1, 1024, 128
    x = torch.randn((G, B, 24, T), device=U.DEV); w = torch.randn((G, 15, 24, 256), device=U.DEV) * 0.05; b = torch.randn((G, 256), device=U.DEV) * 0.1
    y0 = torch.empty((G, B, 128, T), device=U.DEV); y = torch.empty_like(y0)
    check(lib.vc_set_math_mode(mode))
    try:
        nb = lib.vc_gated_conv1d_glu_image_bytes(24, 15, 256)
        img = torch.zeros((G, nb), dtype=torch.uint8, device=U.DEV)
        check(lib.vc_gated_conv1d_glu_prepack(ptr(w), w[0].numel(), ptr(img), nb, G, 24, 15, 256, U.stream()))

        def call(out):
            if packed:
                check(lib.vc_gated_conv1d_glu_fwd_packed(ptr(x), x[0].numel(), ptr(img), nb, ptr(b), 256, ptr(out), out[0].numel(), G, B, 24, T, 256, 15, U.stream()))
            else:
                check(lib.vc_gated_conv1d_glu_fwd(ptr(x), x[0].numel(), ptr(w), w[0].numel(), ptr(b), 256, ptr(out), out[0].numel(), G, B, 24, T, 256, 15, U.stream()))
        call(y0)
        bad = torch.zeros((), dtype=torch.int64, device=U.DEV)
        for _ in range(3000):
            y.fill_(float('nan'))
            call(y)
            bad += (y != y0).any().to(torch.int64)
        assert int(bad.item()) == 0, '%d of 3000 launches differ from the first' % int(bad.item())
    finally:
        check(lib.vc_set_math_mode(0))


def test_conv_group_stride_negative(built_lib):
    """group 1 placed BEFORE group 0 in memory (signed group strides)."""
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, ptr, check
    O = _oracle()
    rng = np.random.default_rng(7)
    B, Cin, T, Cout, k = 2, 32, 16, 64, 3
    x = rng.standard_normal((2, B, T, Cin)).astype(np.float32)
    w = rng.standard_normal((2, 1, k, Cin, Cout)).astype(np.float32) * 0.1
    b = rng.standard_normal((2, Cout)).astype(np.float32)
    x_d = U.dev(np.transpose(x, (0, 1, 3, 2))[:, :, :, None, :])
    w_d, b_d = U.dev(w), U.dev(b)
    y_d = torch.zeros((2, B, Cout, 1, T), device=U.DEV)
    n = x_d[0].numel()
    check(lib.vc_conv2d_fwd(U._lib.vp(x_d.data_ptr() + 4 * n), -n, ptr(w_d), w_d[0].numel(), ptr(b_d), Cout,
                            ptr(y_d), y_d[0].numel(), 2, B, Cin, 1, T, Cout, 1, k, 1, 1, U.stream()))
    for g in range(2):   # output group g was computed from input group 1-g with weights g
        ref, _ = O.conv1d_fwd(x[1 - g], w[g][0], b[g], 1)
        assert U.rel_err(y_d[g, :, :, 0].cpu().numpy(), np.transpose(ref, (0, 2, 1))) < TOL


# (mode, ps, C, L, B)
NORM_CASES = [
    (0, 0, 256, 64, 2), (0, 0, 1024, 32, 3), (0, 1, 512, 64, 2), (0, 1, 256, 128, 2),
    (0, 0, 256, 384, 2), (0, 0, 512, 96, 2), (0, 0, 1024, 48, 3), (0, 0, 16, 1000, 1),
    (1, 0, 128, 128, 2), (1, 0, 128, 1536, 2), (2, 0, 512, 32, 3), (2, 0, 64, 33, 2), (0, 1, 8, 6, 2),
    (1, 0, 64, 4608, 2), (0, 0, 32, 2304, 2), (2, 0, 16, 3000, 1), (0, 1, 8, 4100, 2), (1, 0, 8, 2049, 1),   # rows > 2048: streaming kernels
    (0, 0, 512, 2, 3), (0, 0, 256, 4, 3), (2, 0, 512, 2, 3), (0, 1, 512, 4, 3), (0, 1, 256, 8, 3), (1, 0, 128, 8, 3),
]


@pytest.mark.parametrize('case', [NORM_CASES[i] for i in (0, 2, 4, 8, 10, 13, 16, 18, 22)])
def test_instnorm_glu_deterministic_mode(built_lib, case):
    """the backward's gamma / beta / conv-bias sums reduced over the batch in a fixed order (norm_bwd_kernel<.., DET>): same
    oracle tolerance (module.py:9-20 and its gradient); bitwise repeatability is asserted at step level"""
    from voice_converter_cyclegan_amd._lib import lib, check
    check(lib.vc_set_deterministic(1))
    try:
        test_instnorm_glu(built_lib, case)
    finally:
        check(lib.vc_set_deterministic(0))


@pytest.mark.parametrize('case', NORM_CASES)
def test_instnorm_glu(built_lib, case):
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, ptr, check
    O = _oracle()
    mode, ps, C, L, B = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    G = 2
    nb = 1 if mode == 2 else 2
    Cy, Ly = (2 * C, L // 2) if ps else (C, L)          # conv channels per branch / conv length
    y = rng.standard_normal((G, B, nb * Cy, Ly)).astype(np.float32) * 2 + 0.3     # [G,B,C,T] device layout
    gam = rng.uniform(0.5, 1.5, (G, nb * C)).astype(np.float32)
    bet = rng.uniform(-0.3, 0.3, (G, nb * C)).astype(np.float32)
    res = rng.standard_normal((G, B, C, L)).astype(np.float32)
    dout = rng.standard_normal((G, B, C, L)).astype(np.float32)
    outs, dys, dgs, dbs, dbias = [], [], [], [], []
    for g in range(G):
        br_out, caches = [], []
        for br in range(nb):
            h = np.transpose(y[g][:, br * Cy:(br + 1) * Cy], (0, 2, 1))           # [B,T,Cy] channels-last
            if ps:
                h = O.pixel_shuffle_fwd(h)
            if mode != 1:
                h, cch = O.instance_norm_fwd(h, gam[g][br * C:(br + 1) * C], bet[g][br * C:(br + 1) * C])
                caches.append(cch)
            br_out.append(h)
        do = np.transpose(dout[g], (0, 2, 1))
        if mode == 2:
            out = np.transpose(res[g], (0, 2, 1)) + br_out[0]
            dbr = [do]
        else:
            out, gc = O.glu_fwd(br_out[0], br_out[1])
            dbr = list(O.glu_bwd(do, gc))
        dyg, dgg, dbg, dbi = [], [], [], []
        for br in range(nb):
            d = dbr[br]
            if mode != 1:
                d, dga, dbe = O.instance_norm_bwd(d, caches[br])
                dgg.append(dga); dbg.append(dbe)
            if ps:
                d = O.pixel_shuffle_bwd(d)
            dyg.append(np.transpose(d, (0, 2, 1)))
            dbi.append(d.sum(axis=(0, 1)))
        outs.append(np.transpose(out, (0, 2, 1))); dys.append(np.concatenate(dyg, axis=1))
        dgs.append(np.concatenate(dgg) if dgg else None); dbs.append(np.concatenate(dbg) if dbg else None)
        dbias.append(np.concatenate(dbi))
    y_d, gam_d, bet_d, res_d, do_d = map(U.dev, (y, gam, bet, res, dout))
    out_d = torch.full((G, B, C, L), float('nan'), device=U.DEV)
    st_d = torch.zeros((G, B, nb * C, 2), device=U.DEV)
    check(lib.vc_instnorm_glu_fwd(mode, ps, ptr(y_d), y_d[0].numel(), ptr(res_d), res_d[0].numel(),
                                  ptr(gam_d), ptr(bet_d), nb * C, ptr(out_d), out_d[0].numel(),
                                  ptr(st_d), st_d[0].numel(), G, B, C, L, U.stream()))
    e = U.rel_err(out_d.cpu().numpy(), np.stack(outs))
    assert e < TOL, 'fwd rel err %g' % e
    dy_d = torch.full_like(y_d, float('nan'))
    dg_d = torch.zeros_like(gam_d); db_d = torch.zeros_like(bet_d)
    dbias_d = torch.zeros((G, nb * Cy), device=U.DEV)
    check(lib.vc_instnorm_glu_bwd(mode, ps, ptr(y_d), y_d[0].numel(), ptr(do_d), do_d[0].numel(),
                                  ptr(gam_d), ptr(bet_d), nb * C, ptr(st_d), st_d[0].numel(),
                                  ptr(dy_d), dy_d[0].numel(), ptr(dg_d), ptr(db_d), nb * C,
                                  ptr(dbias_d), nb * Cy, G, B, C, L, U.stream()))
    e = U.rel_err(dy_d.cpu().numpy(), np.stack(dys))
    assert e < TOL, 'dy rel err %g' % e
    if mode != 1:
        assert U.rel_err(dg_d.cpu().numpy(), np.stack(dgs)) < TOL
        assert U.rel_err(db_d.cpu().numpy(), np.stack(dbs)) < TOL
        if ps:
            # the norm pools conv channels c and c+C, so each conv bias keeps a real gradient
            assert U.rel_err(dbias_d.cpu().numpy(), np.stack(dbias)) < TOL
        else:
            # conv bias directly under an instance norm: exactly-zero gradient, rounding noise on both sides
            scale = np.abs(np.stack(dys)).sum(axis=(1, 3)).max()
            assert np.abs(dbias_d.cpu().numpy()).max() < 2e-4 * scale
    else:
        assert U.rel_err(dbias_d.cpu().numpy(), np.stack(dbias)) < TOL


@pytest.mark.parametrize('B,C,HW', [(2, 1024, 48), (3, 64, 144), (1, 1024, 70)])
def test_disc_head(built_lib, B, C, HW):
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, ptr, check
    O = _oracle()
    rng = np.random.default_rng(B * 1000 + C + HW)
    G = 2
    x = rng.standard_normal((G, B, C, HW)).astype(np.float32)
    w = (rng.standard_normal((G, C)) / np.sqrt(C)).astype(np.float32)
    b = rng.standard_normal((G, 1)).astype(np.float32)
    dout = rng.standard_normal((G, B, HW)).astype(np.float32)
    par = np.zeros((G, C + 4), np.float32); par[:, :C] = w; par[:, C] = b[:, 0]
    x_d, par_d, do_d = U.dev(x), U.dev(par), U.dev(dout)
    o_d = torch.zeros((G, B, HW), device=U.DEV)
    check(lib.vc_disc_head_fwd(ptr(x_d), x_d[0].numel(), ptr(par_d), U._lib.vp(par_d.data_ptr() + 4 * C), C + 4,
                               ptr(o_d), B * HW, G, B, C, HW, U.stream()))
    z = np.einsum('gbcp,gc->gbp', x.astype(np.float64), w.astype(np.float64)) + b[:, :, None]
    o = O.sigmoid(z)
    assert U.rel_err(o_d.cpu().numpy(), o) < TOL
    dx_d = torch.zeros_like(x_d); dpar_d = torch.zeros_like(par_d)
    check(lib.vc_disc_head_bwd(ptr(x_d), x_d[0].numel(), ptr(par_d), C + 4, ptr(o_d), B * HW, ptr(do_d), B * HW,
                               ptr(dx_d), dx_d[0].numel(), ptr(dpar_d), U._lib.vp(dpar_d.data_ptr() + 4 * C), C + 4,
                               G, B, C, HW, U.stream()))
    dz = dout * o * (1 - o)
    assert U.rel_err(dx_d.cpu().numpy(), dz[:, :, None, :] * w[:, None, :, None]) < TOL
    assert U.rel_err(dpar_d.cpu().numpy()[:, :C], np.einsum('gbp,gbcp->gc', dz, x)) < TOL
    assert U.rel_err(dpar_d.cpu().numpy()[:, C], dz.sum(axis=(1, 2))) < TOL


def test_bias_grad_and_adam(built_lib):
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, ptr, check
    O = _oracle()
    rng = np.random.default_rng(3)
    dy = rng.standard_normal((2, 3, 24, 128)).astype(np.float32)
    dy_d = U.dev(dy); db_d = torch.zeros((2, 24), device=U.DEV)
    check(lib.vc_bias_grad(ptr(dy_d), dy_d[0].numel(), ptr(db_d), 24, 2, 3, 24, 128, U.stream()))
    assert U.rel_err(db_d.cpu().numpy(), dy.sum(axis=(1, 3))) < TOL
    # Adam, TF form, 3 steps, odd length (tail path)
    n = 4 * 1000 + 3
    P = {'generator_x': rng.standard_normal(n).astype(np.float32)}
    opt = O.AdamTF(['generator_x'], P)
    p_d = U.dev(np.concatenate([P['generator_x'], [0.0]]))
    m_d = torch.zeros_like(p_d); v_d = torch.zeros_like(p_d)
    for step in range(3):
        g = (rng.standard_normal(n) * (10.0 ** rng.integers(-6, 1, n))).astype(np.float32)
        alpha = float(opt.alpha(2e-4))
        g_d = U.dev(np.concatenate([g, [0.0]]))
        check(lib.vc_adam_tf_update(ptr(p_d), ptr(m_d), ptr(v_d), ptr(g_d), n, alpha, 0.5, 0.999, 1e-8, 1.0, U.stream()))
        opt.apply(P, {'generator_x': g}, 2e-4)
        got = p_d.cpu().numpy()[:n]
        assert np.abs(got - P['generator_x']).max() < 2e-7, (step, np.abs(got - P['generator_x']).max())


# ---- Winograd F(4,3) route (csrc/vc_conv_wino.hip): forward and input gradient of 3-tap stride-1 1-D convolutions in fp32 mode
WINO_CASES = [
    # Cin, W, Cout, B
    (512, 32, 2048, 8),        # residual h1, cycle hop of the batch-8 step: 64 pixels per transform position = the route's threshold
    (1024, 32, 512, 16),       # residual h2, first hop
    (64, 8, 64, 32),           # two tiles a row: both are edge tiles
    (128, 64, 96, 4),          # 36 x 256 geometry's frame count (16 tiles a row), Cout not a multiple of 64
    (96, 12, 64, 24),          # three tiles a row (W = 12), Cin = 96
    (128, 64, 128, 4),         # filter gradient: two k octets a row (16 tiles)
    (64, 32, 64, 12),          # filter gradient: 12 iterations per transform position (prefetch depth 4 instead of 8)
    # F(4,5): the up-sampling blocks' 5-tap convolutions (module.py:115-133), eight transform positions
    (512, 32, 2048, 8, 5),     # up-sample 1 at the B = 8 step's cycle hop
    (512, 64, 1024, 4, 5),     # up-sample 2 (64 frames: two k octets a row)
    (64, 8, 96, 32, 5),        # two tiles a row: every tile touches both edges' padding
    (96, 12, 64, 24, 5),       # three tiles a row
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', WINO_CASES)
def test_winograd_route_against_the_oracle_and_the_direct_kernel(built_lib, case):
    """module.py:66-83 / 115-133 (conv1d, kernel 3 or 5, stride 1, SAME): the Winograd route (production in fp32 mode) against the oracle at the operator
    tolerance (forced for operator-level calls: vc_tune_winograd(2)) and against the direct implicit GEMM (vc_tune_winograd(0)) at 2e-5 of the tensor's maximum -- the route changes the summation
    order and adds the rounding of its transforms (2-3e-6 measured against fp64), nothing else; addend / accumulate semantics of the input
    gradient included.  The launch counter proves which route ran."""
    from tests import gpu_util as U
    lib, check = built_lib.lib, built_lib.check
    O = _oracle()
    Cin, W, Cout, B = case[:4]
    K = case[4] if len(case) > 4 else 3
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    G = 2
    xs = rng.standard_normal((G, B, W, Cin)).astype(np.float32)
    ws = (rng.standard_normal((G, 1, K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    bs = rng.uniform(-0.5, 0.5, (G, Cout)).astype(np.float32)
    ys, dxs, dys, dws = [], [], [], []
    for g in range(G):
        y, cache = O.conv1d_fwd(xs[g], ws[g][0], bs[g], 1)
        dy = rng.standard_normal(y.shape).astype(np.float32)
        dx, dw, _ = O.conv1d_bwd(dy, ws[g][0], cache)
        ys.append(y); dxs.append(dx); dys.append(dy); dws.append(dw[None])
    # the fused filter-gradient kernel: whole k octets, 64 x 64 channel tiles, iterations in multiples of 4 (else the direct kernel)
    wg = (W // 4) % 8 == 0 and Cin % 64 == 0 and Cout % 64 == 0 and (B * (W // 32)) % 4 == 0
    cf = lambda a: np.ascontiguousarray(np.transpose(np.stack(a), (0, 1, 3, 2)))[:, :, :, None, :]      # [G, B, C, 1, W]
    x_d, w_d, b_d, dy_d = U.dev(cf(list(xs))), U.dev(ws), U.dev(bs), U.dev(cf(dys))
    add = U.dev(rng.standard_normal(cf(dxs).shape).astype(np.float32)); init = U.dev(rng.standard_normal(cf(dxs).shape).astype(np.float32))
    out = {}
    try:
        for on in (2, 0):
            check(lib.vc_tune_winograd(on)); lib.vc_testing_wino_gemms(1)
            y = U.conv_fwd(x_d, w_d, b_d, 1, K, 1, 1)
            dx = U.conv_dgrad(dy_d, w_d, 1, W, 1, K, 1, 1)
            dx2 = U.conv_dgrad(dy_d, w_d, 1, W, 1, K, 1, 1, addend=add, init=init)
            dw = U.conv_wgrad(x_d, dy_d, 1, K, 1, 1)
            dw2 = U.conv_wgrad(x_d, dy_d, 1, K, 1, 1, init=dw)              # the operator accumulates: twice the gradient
            assert lib.vc_testing_wino_gemms(1) == ((3 + (2 if wg else 0)) if on else 0)      # forward, two input gradients, two filter gradients
            out[on] = (y.cpu().numpy(), dx.cpu().numpy(), dx2.cpu().numpy(), dw.cpu().numpy(), dw2.cpu().numpy())
            assert U.rel_err(out[on][0], cf(ys)) < TOL and U.rel_err(out[on][1], cf(dxs)) < TOL
            assert U.rel_err(out[on][2], cf(dxs) + add.cpu().numpy() + init.cpu().numpy()) < TOL
            assert U.rel_err(out[on][3], np.stack(dws)) < TOL and U.rel_err(out[on][4], 2 * np.stack(dws)) < TOL
    finally:
        check(lib.vc_tune_winograd(1))
    for a, b in zip(out[2], out[0]):
        e = U.rel_err(a, b)
        print('winograd vs direct: %.2e' % e)
        assert e < 2e-5


@pytest.mark.gpu
def test_winograd_route_leaves_other_geometries_alone(built_lib):
    """stride 2, 7 taps, 2-D, a width that is not a multiple of 4, bf16 mode, fewer than 64 pixels per transform position: the direct kernels"""
    from tests import gpu_util as U
    lib, check = built_lib.lib, built_lib.check
    check(lib.vc_tune_winograd(2))
    lib.vc_testing_wino_gemms(1)
    for Cin, H, W, Cout, kh, kw, sh, sw, B in [(64, 1, 32, 64, 1, 3, 1, 2, 8), (64, 1, 32, 64, 1, 7, 1, 1, 8), (64, 4, 32, 64, 3, 3, 1, 1, 8),
                                                (64, 1, 30, 64, 1, 3, 1, 1, 16), (64, 1, 32, 64, 1, 3, 1, 1, 4)]:
        x = torch.randn((2, B, Cin, H, W), device=U.DEV); w = torch.randn((2, kh, kw, Cin, Cout), device=U.DEV) * 0.05; b = torch.zeros((2, Cout), device=U.DEV)
        U.conv_fwd(x, w, b, kh, kw, sh, sw)
    assert lib.vc_testing_wino_gemms(1) == 0
    check(lib.vc_set_math_mode(1))
    try:
        x = torch.randn((2, 16, 64, 1, 32), device=U.DEV); w = torch.randn((2, 1, 3, 64, 64), device=U.DEV) * 0.05; b = torch.zeros((2, 64), device=U.DEV)
        U.conv_fwd(x, w, b, 1, 3, 1, 1)
    finally:
        check(lib.vc_set_math_mode(0)); check(lib.vc_tune_winograd(1))
    assert lib.vc_testing_wino_gemms(1) == 0
    # production mode: an operator-level call carries no transformed filters and stays on the direct kernel (its filter gradient needs none)
    x = torch.randn((2, 16, 64, 1, 32), device=U.DEV); w = torch.randn((2, 1, 3, 64, 64), device=U.DEV) * 0.05; b = torch.zeros((2, 64), device=U.DEV)
    U.conv_fwd(x, w, b, 1, 3, 1, 1)
    assert lib.vc_testing_wino_gemms(1) == 0


# ---- F(3,6) along H (csrc/vc_conv_wino.hip: winoh_kernel): the discriminator's last block, kernel 6 x 3, strides 1 x 2 (module.py:208)
@pytest.mark.gpu
@pytest.mark.parametrize('case', [(512, 6, 16, 2048, 8), (512, 9, 32, 2048, 3), (64, 6, 16, 128, 16), (64, 12, 8, 96, 4)])
def test_winograd_along_h_for_the_discriminators_last_block(built_lib, case):
    """forward and input gradient (+ addend / accumulate) against the oracle (1e-4) and against the direct kernel (2e-5); 6 rows (24 MCEPs),
    9 rows (36 MCEPs: three tiles), 12 rows; the launch counter proves which route ran"""
    from tests import gpu_util as U
    lib, check = built_lib.lib, built_lib.check
    O = _oracle()
    Cin, H, W, Cout, B = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    G = 2
    xs = rng.standard_normal((G, B, H, W, Cin)).astype(np.float32)
    ws = (rng.standard_normal((G, 6, 3, Cin, Cout)) / np.sqrt(18 * Cin)).astype(np.float32)
    bs = rng.uniform(-0.5, 0.5, (G, Cout)).astype(np.float32)
    ys, dxs, dys = [], [], []
    for g in range(G):
        y, cache = O.conv2d_fwd(xs[g], ws[g], bs[g], (1, 2))
        dy = rng.standard_normal(y.shape).astype(np.float32)
        dx, _, _ = O.conv2d_bwd(dy, ws[g], cache)
        ys.append(y); dxs.append(dx); dys.append(dy)
    cf = lambda a: np.ascontiguousarray(np.transpose(np.stack(a), (0, 1, 4, 2, 3)))
    x_d, w_d, b_d, dy_d = U.dev(cf(list(xs))), U.dev(ws), U.dev(bs), U.dev(cf(dys))
    add = U.dev(rng.standard_normal(cf(dxs).shape).astype(np.float32)); init = U.dev(rng.standard_normal(cf(dxs).shape).astype(np.float32))
    out = {}
    try:
        for on in (2, 0):
            check(lib.vc_tune_winograd(on)); lib.vc_testing_wino_gemms(1)
            y = U.conv_fwd(x_d, w_d, b_d, 6, 3, 1, 2)
            dx = U.conv_dgrad(dy_d, w_d, H, W, 6, 3, 1, 2)
            dx2 = U.conv_dgrad(dy_d, w_d, H, W, 6, 3, 1, 2, addend=add, init=init)
            assert lib.vc_testing_wino_gemms(1) == (3 if on else 0)
            out[on] = (y.cpu().numpy(), dx.cpu().numpy(), dx2.cpu().numpy())
            assert U.rel_err(out[on][0], cf(ys)) < TOL and U.rel_err(out[on][1], cf(dxs)) < TOL
            assert U.rel_err(out[on][2], cf(dxs) + add.cpu().numpy() + init.cpu().numpy()) < TOL
    finally:
        check(lib.vc_tune_winograd(1))
    for a, b in zip(out[2], out[0]):
        e = U.rel_err(a, b)
        print('winograd along H vs direct: %.2e' % e)
        assert e < 2e-5
