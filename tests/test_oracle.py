"""CPU: pins for the oracle itself (oracle/cyclegan_oracle.py).  The reference cannot run
here (PARITY UNPINNED, see the oracle header), so the oracle is validated on its own:
finite differences in fp64, an independent torch-autograd restatement, the loss-composition
identity recorded in the reference's own train_log, and committed golden vectors."""
import os

import numpy as np
import pytest

from oracle import cyclegan_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _probe(a, i):
    r = np.random.default_rng(1000 + i).standard_normal(a.size)
    return float(a.reshape(-1).astype(np.float64) @ r)


def test_same_padding_rule():
    # S1: TF SAME, extra pad right/bottom
    assert O.same_pad(128, 15, 1) == (128, 7, 7)
    assert O.same_pad(128, 5, 2) == (64, 1, 2)
    assert O.same_pad(24, 3, 2) == (12, 0, 1)
    assert O.same_pad(6, 6, 1) == (6, 2, 3)
    assert O.same_pad(9, 3, 2) == (5, 1, 1)
    assert O.same_pad(7, 5, 2) == (4, 2, 2)


def test_parameter_counts_match_survey():
    P = O.init_params(seed=0)
    n = lambda s: sum(v.size for k, v in P.items() if k.startswith(s + '/'))
    assert n('generator_A2B') == n('generator_B2A') == 38055704
    assert n('discriminator_A') == n('discriminator_B') == 21837825
    assert sum(1 for k in P if k.startswith('generator_A2B/')) == 110
    assert sum(1 for k in P if k.startswith('discriminator_A/')) == 30


def test_pixel_shuffle_is_row_major_reshape():
    x = np.arange(2 * 3 * 8, dtype=np.float64).reshape(2, 3, 8)
    y = O.pixel_shuffle_fwd(x)
    for n in range(2):
        for w in range(3):
            for j in range(2):
                for oc in range(4):
                    assert y[n, 2 * w + j, oc] == x[n, w, j * 4 + oc]
    assert np.array_equal(O.pixel_shuffle_bwd(y), x)


def _fd(f, x, eps=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=['multi_index'])
    for _ in it:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps; fp = f()
        x[i] = old - eps; fm = f()
        x[i] = old
        g[i] = (fp - fm) / (2 * eps)
    return g


def test_finite_differences_shrunk_networks():
    """central differences (fp64) on width/32 copies of both players pin the hand-written backward."""
    P = O.init_params(seed=5, dtype=np.float64, randomize_affine=True, wd=32)
    rng = np.random.default_rng(6)
    A, B = rng.standard_normal((2, 24, 16)), rng.standard_normal((2, 24, 16))   # T/4 = 4 (T/4 = 2 is ill-conditioned)
    Sg, Gg, gA, gB = O.generator_losses_and_grads(P, A, B, 10.0, 5.0)
    Sd, Gd = O.discriminator_losses_and_grads(P, A, B, gA, gB)
    fg = lambda: O.generator_losses_and_grads(P, A, B, 10.0, 5.0, want_grads=False)[0]['generator_loss']
    fd_ = lambda: O.discriminator_losses_and_grads(P, A, B, gA, gB, want_grads=False)[0]['discriminator_loss']
    picks = ['generator_A2B/o1_conv/bias', 'generator_B2A/h1_conv_gates/bias', 'generator_A2B/InstanceNorm_3/gamma',
             'generator_B2A/InstanceNorm_22/beta', 'generator_A2B/upsample1d_block2_h1_gates/bias',
             'generator_B2A/residual1d_block4_h2_conv/bias']
    for n in picks:
        num = _fd(fg, P[n])
        assert np.abs(num - Gg[n]).max() < 2e-5 * max(1.0, np.abs(num).max()), n
    # a slice of conv kernels (full tensors are too many evaluations)
    for n, sl in (('generator_A2B/downsample1d_block1_h1_conv/kernel', (slice(0, 5), slice(0, 2), slice(0, 2))),
                  ('generator_B2A/upsample1d_block1_h1_conv/kernel', (slice(1, 3), slice(0, 2), slice(3, 6))),
                  ('generator_A2B/h1_conv/kernel', (slice(6, 9), slice(5, 7), slice(0, 2)))):
        w = P[n]
        view = w[sl]
        num = _fd(fg, view)
        assert np.abs(num - Gg[n][sl]).max() < 2e-5 * max(1.0, np.abs(num).max()), n
    for n in ('discriminator_A/dense/kernel', 'discriminator_B/dense/bias', 'discriminator_A/InstanceNorm_5/gamma',
              'discriminator_B/h1_conv/bias'):
        num = _fd(fd_, P[n])
        assert np.abs(num - Gd[n].reshape(P[n].shape)).max() < 2e-5 * max(1.0, np.abs(num).max()), n
    n, sl = 'discriminator_A/downsample2d_block3_h1_gates/kernel', (slice(0, 6), slice(0, 3), slice(0, 1), slice(0, 2))
    num = _fd(fd_, P[n][sl])
    assert np.abs(num - Gd[n][sl]).max() < 2e-5 * max(1.0, np.abs(num).max())


def test_two_restatements_agree_fp64():
    from oracle import torch_ref as TR
    P = O.init_params(seed=1, dtype=np.float64, randomize_affine=True, wd=16)
    rng = np.random.default_rng(0)
    A, B = rng.standard_normal((2, 24, 16)), rng.standard_normal((2, 24, 16))
    Sg, Gg, gA, gB = O.generator_losses_and_grads(P, A, B, 10.0, 5.0)
    Sd, Gd = O.discriminator_losses_and_grads(P, A, B, gA, gB)
    S2, G2, gA2, gB2 = TR.losses_and_grads(P, A, B, 10.0, 5.0)
    for k, v in {**Sg, **Sd}.items():
        assert abs(v - S2[k]) < 1e-12 * max(1.0, abs(v)), k
    assert np.abs(gA - gA2).max() < 1e-12 and np.abs(gB - gB2).max() < 1e-12
    G1 = {**Gg, **Gd}
    for n, g2 in G2.items():
        g1 = G1[n].reshape(g2.shape)
        tol = 1e-9 * max(np.abs(g2).max(), 1e-12) + 1e-14      # IN-shadowed biases are ~1e-16 on both sides
        assert np.abs(g1 - g2).max() <= tol, (n, np.abs(g1 - g2).max())


def test_full_width_fp32_against_torch():
    """the real architecture, fp32, one sample, short T: numpy im2col path vs torch conv path."""
    from oracle import torch_ref as TR
    P = O.init_params(seed=2, randomize_affine=True)
    rng = np.random.default_rng(3)
    A = rng.standard_normal((1, 24, 16)).astype(np.float32); B = rng.standard_normal((1, 24, 16)).astype(np.float32)
    Sg, Gg, gA, gB = O.generator_losses_and_grads(P, A, B, 10.0, 5.0)
    Sd, Gd = O.discriminator_losses_and_grads(P, A, B, gA, gB)
    S2, G2, gA2, gB2 = TR.losses_and_grads(P, A, B, 10.0, 5.0)
    for k, v in {**Sg, **Sd}.items():
        assert abs(v - S2[k]) < 1e-4 * max(1.0, abs(v)), (k, v, S2[k])
    assert np.abs(gA - gA2).max() < 1e-4 * np.abs(gA2).max()
    G1 = {**Gg, **Gd}
    for n in ('generator_A2B/o1_conv/kernel', 'generator_B2A/residual1d_block3_h1_conv/kernel',
              'discriminator_A/downsample2d_block3_h1_conv/kernel', 'generator_A2B/InstanceNorm_24/gamma'):
        a, b = G1[n].reshape(G2[n].shape).astype(np.float64), G2[n].astype(np.float64)
        assert np.linalg.norm(a - b) < 2e-3 * np.linalg.norm(b), n


def test_loss_composition_pinned_by_reference_train_log():
    """The only numbers the reference publishes about this path: at step 167 the logged generator_loss
    equals g_A2B + g_B2A + 10*cycle + 5*identity (model.py:72 with train.py:25-26)."""
    z = np.load(os.path.join(GOLD, 'train_log_pins.npz'))
    recomposed = (z['generator_loss_A2B'] + z['generator_loss_B2A'] + z['lambda_cycle'] * z['cycle_loss']
                  + z['lambda_identity'] * z['identity_loss'])
    assert abs(float(recomposed) - float(z['generator_loss'])) < 1e-5 * float(z['generator_loss'])
    # and the oracle composes its own loss the same way
    P = O.init_params(seed=9, dtype=np.float64, wd=32)
    rng = np.random.default_rng(1)
    A, B = rng.standard_normal((1, 24, 8)), rng.standard_normal((1, 24, 8))
    S = O.generator_losses_and_grads(P, A, B, 10.0, 5.0, want_grads=False)[0]
    assert abs(S['generator_loss'] - (S['generator_loss_A2B'] + S['generator_loss_B2A'] + 10 * S['cycle_loss']
                                      + 5 * S['identity_loss'])) < 1e-12


def test_adam_is_tf_form_not_torch_form():
    """S7: eps is outside the bias correction; t starts at 1; beta powers are parameter-dtype variables."""
    P = {'generator_w': np.array([1.0, -2.0, 0.5], np.float64)}
    opt = O.AdamTF(['generator_w'], P)
    g = np.array([0.1, -1e-9, 3.0])
    p0 = P['generator_w'].copy()
    opt.apply(P, {'generator_w': g}, 2e-4)
    m, v = 0.5 * g, 0.001 * g * g
    alpha = 2e-4 * np.sqrt(1 - 0.999) / (1 - 0.5)
    assert np.allclose(P['generator_w'], p0 - alpha * m / (np.sqrt(v) + 1e-8), rtol=1e-12, atol=0)
    assert opt.b1p == 0.25 and abs(opt.b2p - 0.999 ** 2) < 1e-15


def test_golden_shrunk_fp64():
    z = np.load(os.path.join(GOLD, 'golden_shrunk_f64.npz'))
    P = O.init_params(seed=3, dtype=np.float64, randomize_affine=True, wd=16)
    Sg, Gg, gA, gB = O.generator_losses_and_grads(P, z['A'], z['B'], 10.0, 5.0)
    Sd, Gd = O.discriminator_losses_and_grads(P, z['A'], z['B'], gA, gB)
    for k, v in {**Sg, **Sd}.items():
        assert abs(v - float(z['loss/' + k])) < 1e-10, k
    assert np.abs(gA - z['gen_A']).max() < 1e-10
    G = {**Gg, **Gd}
    for i, n in enumerate(z['grad_names']):
        n = str(n)
        assert abs(np.linalg.norm(G[n]) - z['grad_l2'][i]) <= 1e-8 * z['grad_l2'][i] + 1e-13, n
        assert abs(_probe(G[n], i) - z['grad_proj'][i]) <= 1e-7 * z['grad_l2'][i] * np.sqrt(G[n].size) + 1e-13, n


@pytest.mark.timeout(600)
def test_golden_full_fp32_step0():
    """real architecture, fp32, B=1: regenerates step 0 of the committed golden run (a few seconds)."""
    z = np.load(os.path.join(GOLD, 'golden_full_f32.npz'))
    P = O.init_params(seed=42, randomize_affine=True)
    A, B = O.synthetic_batch(1, 24, 128)
    orc = O.CycleGANOracle(P)
    orc.train(A, B, 10.0, 5.0, 2e-4, 1e-4)
    got = np.array([orc.last[k] for k in ('cycle_loss', 'identity_loss', 'generator_loss_A2B', 'generator_loss_B2A',
                                         'generator_loss', 'discriminator_loss_A', 'discriminator_loss_B',
                                         'discriminator_loss')])
    assert np.abs(got - z['step0/losses']).max() < 1e-4 * np.abs(z['step0/losses']).max()   # BLAS thread count may differ
    assert np.abs(orc.last_fakes[0] - z['gen_A']).max() < 1e-4 * np.abs(z['gen_A']).max()


@pytest.mark.timeout(600)
@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, 'golden_tf_f32.npz')),
                    reason='no TensorFlow-made fixture: run tests/golden/make_tf_golden.py under TensorFlow 1.x (parity unpinned until then)')
def test_oracle_against_tensorflow_reference():
    """THE pin: the oracle against numbers produced by the reference graph itself under TensorFlow 1.x
    (tests/golden/make_tf_golden.py; same schema as golden_full_f32.npz).  1e-3, the north_star tolerance."""
    z = np.load(os.path.join(GOLD, 'golden_tf_f32.npz'))
    P = O.init_params(seed=42, randomize_affine=True)
    A, B = O.synthetic_batch(1, 24, 128)
    orc = O.CycleGANOracle({k: v.copy() for k, v in P.items()})
    keys = ('cycle_loss', 'identity_loss', 'generator_loss_A2B', 'generator_loss_B2A', 'generator_loss',
            'discriminator_loss_A', 'discriminator_loss_B', 'discriminator_loss')
    for step in range(3):
        orc.train(A, B, 10.0, 5.0, 2e-4, 1e-4)
        got = np.array([orc.last[k] for k in keys])
        ref = z['step%d/losses' % step]
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), (step, got, ref)
        if step == 0:
            assert np.abs(orc.last_fakes[0] - z['gen_A']).max() < 1e-3 * np.abs(z['gen_A']).max()
            assert np.abs(orc.last_fakes[1] - z['gen_B']).max() < 1e-3 * np.abs(z['gen_B']).max()
            G = dict(orc.last_grads[0]); G.update(orc.last_grads[1])
            for i, n in enumerate(z['grad_names']):
                n = str(n)
                l2 = float(np.linalg.norm(G[n].astype(np.float64)))
                if z['grad_l2'][i] > 1e-6 * max(z['grad_l2']):      # conv biases under an instance norm: exactly-zero gradient
                    assert abs(l2 - z['grad_l2'][i]) <= 1e-3 * z['grad_l2'][i], (n, l2, float(z['grad_l2'][i]))
                    assert abs(_probe(G[n], i) - z['grad_proj'][i]) <= 3e-3 * z['grad_l2'][i], n
    for i, n in enumerate(z['grad_names']):                          # parameters after three TF-Adam steps
        n = str(n)
        assert abs(np.linalg.norm(orc.P[n].astype(np.float64)) - z['param_l2'][i]) <= 1e-4 * z['param_l2'][i] + 1e-6, n
        if 'param_proj' in z.files:
            # random projection of the parameters after three TF-Adam steps: three moves of <= lr = 2e-4 per element; a flipped
            # near-zero gradient is 2 lr on one element, so the bound is in units of lr * sqrt(size)
            tol = 3e-3 * z['param_l2'][i] + 0.1 * 2e-4 * np.sqrt(orc.P[n].size)
            assert abs(_probe(orc.P[n], i) - z['param_proj'][i]) <= tol, n


def test_bf16_operand_rounding_is_round_to_nearest_even():
    """oracle.round_bf16 (the BASELINE config-3 operand rounding) against torch's float32 -> bfloat16 cast."""
    import torch
    from oracle import cyclegan_oracle as O
    x = np.random.default_rng(0).standard_normal(200000).astype(np.float32) * np.float32(3.7)
    x[:4] = [0.0, -0.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8]          # exact ties: to even
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(O.round_bf16(x), want)
    assert O.round_bf16(x.astype(np.float64)).dtype == np.float64
    O.set_operand_rounding('bf16')
    try:
        y, _ = O.conv1d_fwd(x[:2 * 16 * 8].reshape(2, 16, 8), x[1000:1000 + 3 * 8 * 4].reshape(3, 8, 4), np.zeros(4, np.float32), 1)
    finally:
        O.set_operand_rounding(None)
    y32, _ = O.conv1d_fwd(x[:2 * 16 * 8].reshape(2, 16, 8), x[1000:1000 + 3 * 8 * 4].reshape(3, 8, 4), np.zeros(4, np.float32), 1)
    assert 1e-4 < np.abs(y - y32).max() / np.abs(y32).max() < 3e-2
    with pytest.raises(ValueError):
        O.set_operand_rounding('fp8')


def test_f16_operand_rounding_is_ieee_half_round_to_nearest_even():
    """oracle operand rounding 'f16' (math mode 3 of the HIP path: the gated conv1d + GLU forward of module.py:161-163 with IEEE-half operands) against
    torch's float32 -> float16 cast: ties to even, gradual underflow, overflow to inf -- what v_cvt_pk_f16_f32 does in the default rounding mode."""
    import torch
    from oracle import cyclegan_oracle as O
    x = np.random.default_rng(1).standard_normal(200000).astype(np.float32) * np.float32(3.7)
    x[:8] = [0.0, -0.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 1e-8, 3e-5, 65519.0, 70000.0]      # ties, subnormals, the last finite value, overflow
    want = torch.from_numpy(x).to(torch.float16).to(torch.float32).numpy()
    O.set_operand_rounding('f16')
    try:
        got = O._rop(x)
        assert got.dtype == np.float32 and np.array_equal(got, want)
        assert got[2] == 1.0 and got[3] == np.float32(1.0 + 2.0 ** -9) and np.isinf(got[7]) and got[6] == 65504.0
        assert O._rop(x.astype(np.float64)).dtype == np.float64
        y, _ = O.conv1d_fwd(x[8:8 + 2 * 16 * 8].reshape(2, 16, 8), x[1000:1000 + 3 * 8 * 4].reshape(3, 8, 4), np.zeros(4, np.float32), 1)
    finally:
        O.set_operand_rounding(None)
    y32, _ = O.conv1d_fwd(x[8:8 + 2 * 16 * 8].reshape(2, 16, 8), x[1000:1000 + 3 * 8 * 4].reshape(3, 8, 4), np.zeros(4, np.float32), 1)
    e = np.abs(y - y32).max() / np.abs(y32).max()
    assert 1e-5 < e < 2e-3, e                # an eighth of bf16's operand error
