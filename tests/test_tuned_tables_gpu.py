"""-m gpu: every row of the measured configuration tables (csrc/vc_tuned_igemm.h, csrc/vc_tuned_wgrad.h) that the other
configurations bench.py reports run with -- fp32 batch 32 (BASELINE configs[3]'s per-GPU share), 36x256 batch 64
(configs[4]'s share) and batch 1 (configs[0], the reference's own mini_batch_size) -- executed AT ITS EXACT SHAPE through the C-ABI and held to the oracle at 1e-4 (reference
module.py:22-64 conv1d_layer / conv2d_layer, their input and filter gradients by autodiff of model.py:107-108).  The
launch must report that it took an exact table row (vc_tune_last_*_config source 1), so the configuration checked IS the
one the step at that batch runs.  Batches of 64-192 samples are beyond the numpy oracle in test time, and unnecessarily so:
convolution and its input gradient are per-sample (two random samples of every group are compared), the filter gradient
is a sum over the whole batch but separable in (cin, cout) (the FULL reduction is compared on a random 4 x 8 channel block).
The batch-8 rows are covered by test_conv_fwd_dgrad_wgrad / the step tests at B = 8."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.step_shapes import step_launches, wgrad_key

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _cl(t):
    """[n, C, H, W] device tensor -> channels-last numpy for the oracle"""
    return np.ascontiguousarray(t.permute(0, 2, 3, 1).cpu().numpy())


def _np_conv_fwd(O, x_cl, w, b, kh, sh, sw):
    if b is None:
        b = np.zeros(w.shape[-1], np.float32)
    if kh == 1 and x_cl.shape[1] == 1:
        y, cache = O.conv1d_fwd(x_cl[:, 0], w[0], b, sw)
        return y[:, None], cache, True
    y, cache = O.conv2d_fwd(x_cl, w, b, (sh, sw))
    return y, cache, False


def _np_conv_bwd(O, dy_cl, w, cache, one_d, need_dx=True):
    if one_d:
        dx, dw, _ = O.conv1d_bwd(dy_cl[:, 0], w[0], cache, need_dx)
        return (dx[:, None] if need_dx else None), dw[None]
    dx, dw, _ = O.conv2d_bwd(dy_cl, w, cache, need_dx)
    return dx, dw


@pytest.mark.parametrize('F,T,B', [(24, 128, 32), (36, 256, 64), (24, 128, 1), (24, 128, 2), (24, 128, 4), (24, 128, 16), (24, 128, 64)])
def test_every_benched_table_row_against_the_oracle(built_lib, F, T, B):
    from oracle import cyclegan_oracle as O
    from tests import gpu_util as U
    from voice_converter_cyclegan_amd._lib import lib, check
    out4 = (C.c_int * 4)()
    gen = torch.Generator(device=U.DEV); gen.manual_seed(F * 1000 + B)
    rng = np.random.default_rng(B)
    G = 2
    done = 0
    for name, op, (Cin, H, W, Cout, kh, kw, sh, sw), batches in step_launches(F, T, B):
        Bs = sum(batches)                        # a filter gradient over (2B, B) has the chunk count of one application of 3B samples
        x = torch.randn((G, Bs, Cin, H, W), device=U.DEV, generator=gen)
        w = torch.randn((G, kh, kw, Cin, Cout), device=U.DEV, generator=gen) / float(np.sqrt(kh * kw * Cin))
        b = torch.rand((G, Cout), device=U.DEV, generator=gen) - 0.5
        OH, OW = -(-H // sh), -(-W // sw)
        pick = sorted(rng.choice(Bs, size=min(2, Bs), replace=False).tolist())
        # fp32 production sends the residual blocks' launches (3 taps, stride 1) down the Winograd route from 64 pixels per transform position
        # on (csrc/vc_conv_wino.hip): those run HERE as production runs them (parity against the oracle; their GEMMs are launches the
        # cost model configures, no measured row; vc_tune_winograd(2): without a step's cached filters) AND once more on the direct kernel,
        # whose measured row operator-level calls, inference and the control path use
        routes = (2, 0) if ((kh == 1 and kw in (3, 5) and sh == 1 and sw == 1) or (kh == 6 and sh == 1)) else (1,)
        if op == 'fwd':
            for route in routes:
                check(lib.vc_tune_winograd(route)); lib.vc_testing_wino_gemms(1)
                try:
                    y = U.conv_fwd(x, w, b, kh, kw, sh, sw)
                finally:
                    check(lib.vc_tune_winograd(1))
                wino = lib.vc_testing_wino_gemms(1) > 0
                check(lib.vc_tune_last_igemm_config(out4))
                assert wino or out4[3] == 1, (name, op, list(out4))
                for g in range(G):
                    ref, _, _ = _np_conv_fwd(O, _cl(x[g, pick]), w[g].cpu().numpy(), b[g].cpu().numpy(), kh, sh, sw)
                    e = U.rel_err(_cl(y[g, pick]), ref)
                    assert e < TOL, '%s fwd (winograd %d, tile %d, K groups %d, split-K %d) rel err %g' % (name, wino, out4[0], out4[1], out4[2], e)
                if not wino:
                    break
        elif op == 'dgrad':
            dy = torch.randn((G, Bs, Cout, OH, OW), device=U.DEV, generator=gen)
            for route in routes:
                check(lib.vc_tune_winograd(route)); lib.vc_testing_wino_gemms(1)
                try:
                    dx = U.conv_dgrad(dy, w, H, W, kh, kw, sh, sw)
                finally:
                    check(lib.vc_tune_winograd(1))
                wino = lib.vc_testing_wino_gemms(1) > 0
                check(lib.vc_tune_last_igemm_config(out4))
                assert wino or out4[3] == 1, (name, op, list(out4))
                for g in range(G):
                    wg = w[g].cpu().numpy()
                    _, cache, one_d = _np_conv_fwd(O, _cl(x[g, pick]), wg, None, kh, sh, sw)
                    ref, _ = _np_conv_bwd(O, _cl(dy[g, pick]), wg, cache, one_d)
                    e = U.rel_err(_cl(dx[g, pick]), ref)
                    assert e < TOL, '%s dgrad (winograd %d, tile %d, K groups %d, split-K %d) rel err %g' % (name, wino, out4[0], out4[1], out4[2], e)
                if not wino:
                    break
        else:
            dy = torch.randn((G, Bs, Cout, OH, OW), device=U.DEV, generator=gen)
            kt, N, nch = wgrad_key(Cin, H, W, Cout, kh, kw, sh, sw, batches)
            check(lib.vc_tune_query_wgrad(kt, N, nch, G, 1, out4))
            # the step's form: plain stores into an untouched dW -- or, for rows that split K, atomics into a zeroed one
            init = torch.zeros((G, kh, kw, Cin, Cout), device=U.DEV) if out4[2] > 1 else torch.full((G, kh, kw, Cin, Cout), float('nan'), device=U.DEV)
            ci = sorted(rng.choice(Cin, size=min(4, Cin), replace=False).tolist())
            co = sorted(rng.choice(Cout, size=8, replace=False).tolist())
            for route in routes:
                check(lib.vc_tune_winograd(route)); lib.vc_testing_wino_gemms(1)
                check(lib.vc_tune_wgrad_store(1))
                try:
                    dw = U.conv_wgrad(x, dy, kh, kw, sh, sw, init=init)
                finally:
                    check(lib.vc_tune_wgrad_store(0)); check(lib.vc_tune_winograd(1))
                wino = lib.vc_testing_wino_gemms(1) > 0
                check(lib.vc_tune_last_wgrad_config(out4))
                assert wino or out4[3] == 1, (name, op, list(out4))
                for g in range(G):
                    xs, dys = _cl(x[g][:, ci]), _cl(dy[g][:, co])
                    wsub = np.zeros((kh, kw, len(ci), len(co)), np.float32)
                    _, cache, one_d = _np_conv_fwd(O, xs, wsub, None, kh, sh, sw)
                    _, ref = _np_conv_bwd(O, dys, wsub, cache, one_d, need_dx=False)
                    got = dw[g][:, :, ci][:, :, :, co].cpu().numpy()
                    e = U.rel_err(got, ref)
                    assert e < TOL, '%s wgrad (winograd %d, tile %d, K groups %d, split-K %d) rel err %g' % (name, wino, out4[0], out4[1], out4[2], e)
                if not wino:
                    break
        done += 1
        del x, w, b
    assert done >= 40
