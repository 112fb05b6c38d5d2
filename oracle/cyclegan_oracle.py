"""CPU oracle for the CycleGAN-VC train step -- TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy, the arithmetic that the reference executes
through TensorFlow 1.x for the hot path `model.CycleGAN.train()`:

  * network topology ............ /root/reference/module.py:3-213
  * dataflow + loss algebra ..... /root/reference/model.py:44-90
  * optimisers .................. /root/reference/model.py:103-108
  * two-phase step semantics .... /root/reference/model.py:110-125
  * loss primitives ............. /root/reference/utils.py:6-12

PARITY UNPINNED: TensorFlow is an un-vendored third-party dependency of the
reference (README pins TF 1.8, the Dockerfile TF 1.14) and is not installable
in the build container; the reference ships no tests, golden vectors or
checkpoints.  The TF-library semantics this oracle follows (rules S1-S8 of
SURVEY.md section 8c) are therefore restated from the TF 1.x documentation and
source as published, and validated only on their own: finite-difference
gradients in float64 (tests/test_oracle.py) and an independent second
restatement on torch-CPU autograd (oracle/torch_ref.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product path never does.

Layout follows TensorFlow (channels-last): 1-D activations are [B, T, C],
2-D activations are [B, H, W, C]; conv kernels are [k, Cin, Cout] /
[kh, kw, Cin, Cout].  This is deliberately NOT the layout of the HIP path
([B, C, T] / [B, C, H, W]), so layout mistakes cannot cancel out.

Parameters live in a flat dict keyed by the TF variable names the reference
graph would create, e.g.
`generator_A2B/residual1d_block3_h1_gates/kernel`, `discriminator_A/dense/bias`,
`generator_B2A/InstanceNorm_7/gamma`.
"""
import numpy as np

IN_EPS = 1e-6          # module.py:11
ADAM_BETA1 = 0.5       # model.py:107-108
ADAM_BETA2 = 0.999     # tf.train.AdamOptimizer default
ADAM_EPS = 1e-8        # tf.train.AdamOptimizer default


# ----------------------------------------------------------------------------
# S1: TF 'SAME' padding (extra padding goes right / bottom)
# ----------------------------------------------------------------------------
def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    lo = total // 2
    return out, lo, total - lo


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------
# Matmul-operand rounding for BASELINE config 3 ("bf16 MFMA, fp32 accumulate,
# fp32 instance-norm statistics").  The reference itself is float32 everywhere
# (model.py:35-42); this switch restates what the bf16 mode of the HIP path
# defines: BOTH operands of every conv contraction (forward, input gradient,
# filter gradient) are rounded to bfloat16, round-to-nearest-even, products and
# sums stay in the working dtype.  None (default) = no rounding.  'f16' restates
# math mode 3 of the HIP path (IEEE half operands, one operator: the gated
# conv1d + GLU forward of module.py:161-163): numpy's float16 cast is
# round-to-nearest-even with gradual underflow, as v_cvt_pk_f16_f32 is.
# ----------------------------------------------------------------------------
OPERAND_ROUNDING = None


def set_operand_rounding(kind):
    global OPERAND_ROUNDING
    if kind not in (None, 'bf16', 'f16'):
        raise ValueError('operand rounding must be None, "bf16" or "f16"')
    OPERAND_ROUNDING = kind


def round_bf16(x):
    """float -> nearest bfloat16 (ties to even), returned in x's dtype."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    u = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    return u.view(np.float32).astype(x.dtype, copy=False)


def _rop(x):
    if OPERAND_ROUNDING == 'f16':
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(x.dtype)
    return round_bf16(x) if OPERAND_ROUNDING == 'bf16' else x


# ----------------------------------------------------------------------------
# conv1d_layer (module.py:22-42): tf.layers.conv1d(padding='same') + bias.
# S2: cross-correlation, kernel [k, Cin, Cout].
# ----------------------------------------------------------------------------
def conv1d_fwd(x, w, b, stride):
    B, T, Cin = x.shape
    k, _, Cout = w.shape
    To, pl, pr = same_pad(T, k, stride)
    xp = np.pad(x, ((0, 0), (pl, pr), (0, 0)))
    idx = np.arange(To)[:, None] * stride + np.arange(k)[None, :]
    cols = _rop(xp[:, idx, :].reshape(B * To, k * Cin))
    y = cols @ _rop(w.reshape(k * Cin, Cout)) + b
    return y.reshape(B, To, Cout), (cols, x.shape, stride, pl, pr)


def conv1d_bwd(dy, w, cache, need_dx=True):
    cols, xshape, stride, pl, pr = cache
    B, T, Cin = xshape
    k, _, Cout = w.shape
    To = dy.shape[1]
    dy2 = dy.reshape(B * To, Cout)
    dyr = _rop(dy2)                                   # cols were rounded in the forward (rounding is idempotent)
    dw = (cols.T @ dyr).reshape(k, Cin, Cout)
    db = dy2.sum(axis=0)
    dx = None
    if need_dx:
        dcols = (dyr @ _rop(w.reshape(k * Cin, Cout)).T).reshape(B, To, k, Cin)
        dxp = np.zeros((B, T + pl + pr, Cin), dtype=dy.dtype)
        for tap in range(k):
            dxp[:, tap:tap + (To - 1) * stride + 1:stride, :] += dcols[:, :, tap, :]
        dx = dxp[:, pl:pl + T, :]
    return dx, dw, db


# ----------------------------------------------------------------------------
# conv2d_layer (module.py:44-64): tf.layers.conv2d(padding='same') + bias.
# ----------------------------------------------------------------------------
def conv2d_fwd(x, w, b, strides):
    B, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    sh, sw = strides
    Ho, pt, pb = same_pad(H, kh, sh)
    Wo, pl, pr = same_pad(W, kw, sw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    cols = np.empty((B, Ho, Wo, kh, kw, Cin), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + (Ho - 1) * sh + 1:sh,
                                        j:j + (Wo - 1) * sw + 1:sw, :]
    cols = _rop(cols.reshape(B * Ho * Wo, kh * kw * Cin))
    y = cols @ _rop(w.reshape(kh * kw * Cin, Cout)) + b
    return y.reshape(B, Ho, Wo, Cout), (cols, x.shape, strides, (pt, pb, pl, pr))


def conv2d_bwd(dy, w, cache, need_dx=True):
    cols, xshape, strides, pads = cache
    B, H, W, Cin = xshape
    kh, kw, _, Cout = w.shape
    sh, sw = strides
    pt, pb, pl, pr = pads
    _, Ho, Wo, _ = dy.shape
    dy2 = dy.reshape(B * Ho * Wo, Cout)
    dyr = _rop(dy2)
    dw = (cols.T @ dyr).reshape(kh, kw, Cin, Cout)
    db = dy2.sum(axis=0)
    dx = None
    if need_dx:
        dcols = (dyr @ _rop(w.reshape(kh * kw * Cin, Cout)).T).reshape(B, Ho, Wo, kh, kw, Cin)
        dxp = np.zeros((B, H + pt + pb, W + pl + pr, Cin), dtype=dy.dtype)
        for i in range(kh):
            for j in range(kw):
                dxp[:, i:i + (Ho - 1) * sh + 1:sh, j:j + (Wo - 1) * sw + 1:sw, :] += \
                    dcols[:, :, :, i, j, :]
        dx = dxp[:, pt:pt + H, pl:pl + W, :]
    return dx, dw, db


# ----------------------------------------------------------------------------
# instance_norm_layer (module.py:9-20): tf.contrib.layers.instance_norm,
# epsilon=1e-6, center+scale, NHWC.  S3: moments over every axis except batch
# and channel, biased variance, tf.nn.batch_normalization op order
# (inv = rsqrt(var+eps)*gamma ; y = x*inv + (beta - mean*inv)).
# ----------------------------------------------------------------------------
def instance_norm_fwd(x, gamma, beta):
    axes = tuple(range(1, x.ndim - 1))
    mean = x.mean(axis=axes, keepdims=True)
    var = np.square(x - mean).mean(axis=axes, keepdims=True)
    rstd = 1.0 / np.sqrt(var + x.dtype.type(IN_EPS))
    inv = rstd * gamma
    y = x * inv + (beta - mean * inv)
    xhat = (x - mean) * rstd
    return y, (xhat, rstd, gamma, axes)


def instance_norm_bwd(dy, cache):
    xhat, rstd, gamma, axes = cache
    red = tuple(range(0, dy.ndim - 1))
    dgamma = (dy * xhat).sum(axis=red)
    dbeta = dy.sum(axis=red)
    dxhat = dy * gamma
    m1 = dxhat.mean(axis=axes, keepdims=True)
    m2 = (dxhat * xhat).mean(axis=axes, keepdims=True)
    dx = rstd * (dxhat - m1 - xhat * m2)
    return dx, dgamma, dbeta


# ----------------------------------------------------------------------------
# gated_linear_layer (module.py:3-7): inputs * sigmoid(gates)
# ----------------------------------------------------------------------------
def glu_fwd(a, g):
    s = sigmoid(g)
    return a * s, (a, s)


def glu_bwd(dy, cache):
    a, s = cache
    return dy * s, dy * a * s * (1.0 - s)


# ----------------------------------------------------------------------------
# pixel_shuffler (module.py:135-146): S4 row-major reshape [n,w,c]->[n,2w,c/2]
# ----------------------------------------------------------------------------
def pixel_shuffle_fwd(x, r=2):
    n, w, c = x.shape
    return x.reshape(n, w * r, c // r)


def pixel_shuffle_bwd(dy, r=2):
    n, ow, oc = dy.shape
    return dy.reshape(n, ow // r, oc * r)


# ----------------------------------------------------------------------------
# Parameter book-keeping: TF variable names in creation order.
# ----------------------------------------------------------------------------
def generator_layout(num_features=24, wd=1):
    """[(tf_var_suffix, shape)] in TF creation order for one generator scope
    (module.py:161-182).  InstanceNorm scopes are auto-numbered in creation
    order because instance_norm_layer ignores its `name` (module.py:13-18).
    wd > 1 divides every hidden width (test-only shrunk nets for finite
    differences); wd=1 is the reference architecture."""
    L = []
    n_in = [0]
    c128, c256, c512, c1024 = 128 // wd, 256 // wd, 512 // wd, 1024 // wd

    def conv(name, k, cin, cout):
        L.append((name + '/kernel', (k, cin, cout)))
        L.append((name + '/bias', (cout,)))

    def inorm(c):
        nm = 'InstanceNorm' if n_in[0] == 0 else 'InstanceNorm_%d' % n_in[0]
        n_in[0] += 1
        L.append((nm + '/beta', (c,)))
        L.append((nm + '/gamma', (c,)))

    conv('h1_conv', 15, num_features, c128)
    conv('h1_conv_gates', 15, num_features, c128)
    for blk, cin, cout in (('downsample1d_block1_', c128, c256), ('downsample1d_block2_', c256, c512)):
        conv(blk + 'h1_conv', 5, cin, cout); inorm(cout)
        conv(blk + 'h1_gates', 5, cin, cout); inorm(cout)
    for i in range(1, 7):
        blk = 'residual1d_block%d_' % i
        conv(blk + 'h1_conv', 3, c512, c1024); inorm(c1024)
        conv(blk + 'h1_gates', 3, c512, c1024); inorm(c1024)
        conv(blk + 'h2_conv', 3, c1024, c512); inorm(c512)
    for blk, cin, cout in (('upsample1d_block1_', c512, c1024), ('upsample1d_block2_', c512, c512)):
        conv(blk + 'h1_conv', 5, cin, cout); inorm(cout // 2)
        conv(blk + 'h1_gates', 5, cin, cout); inorm(cout // 2)
    # module.py:182 hard-codes filters=24; SURVEY 7 "hard parts": config 5 needs
    # filters=num_features, identical at the reference's only value (24).
    conv('o1_conv', 15, c256, num_features)
    return L


def discriminator_layout(wd=1):
    """[(tf_var_suffix, shape)] for one discriminator scope (module.py:201-211)."""
    L = []
    n_in = [0]
    c128, c256, c512, c1024 = 128 // wd, 256 // wd, 512 // wd, 1024 // wd

    def conv(name, kh, kw, cin, cout):
        L.append((name + '/kernel', (kh, kw, cin, cout)))
        L.append((name + '/bias', (cout,)))

    def inorm(c):
        nm = 'InstanceNorm' if n_in[0] == 0 else 'InstanceNorm_%d' % n_in[0]
        n_in[0] += 1
        L.append((nm + '/beta', (c,)))
        L.append((nm + '/gamma', (c,)))

    conv('h1_conv', 3, 3, 1, c128)
    conv('h1_conv_gates', 3, 3, 1, c128)
    for blk, kh, kw, cin, cout in (('downsample2d_block1_', 3, 3, c128, c256),
                                   ('downsample2d_block2_', 3, 3, c256, c512),
                                   ('downsample2d_block3_', 6, 3, c512, c1024)):
        conv(blk + 'h1_conv', kh, kw, cin, cout); inorm(cout)
        conv(blk + 'h1_gates', kh, kw, cin, cout); inorm(cout)
    L.append(('dense/kernel', (c1024, 1)))
    L.append(('dense/bias', (1,)))
    return L


SCOPES_G = ('generator_A2B', 'generator_B2A')
SCOPES_D = ('discriminator_A', 'discriminator_B')


def init_params(seed=42, num_features=24, dtype=np.float32, randomize_affine=False, wd=1):
    """tf.layers defaults: glorot-uniform kernels (limit = sqrt(6/(fan_in+fan_out)),
    fan = receptive_field * channels), zero biases; instance_norm: beta=0,
    gamma=1.  randomize_affine=True perturbs bias/beta/gamma so parity tests
    exercise those paths."""
    rng = np.random.default_rng(seed)
    P = {}
    for scope, layout in ([(s, generator_layout(num_features, wd)) for s in SCOPES_G] +
                          [(s, discriminator_layout(wd)) for s in SCOPES_D]):
        for name, shape in layout:
            full = scope + '/' + name
            if name.endswith('/kernel'):
                rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
                fan_in, fan_out = rf * shape[-2], rf * shape[-1]
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                P[full] = rng.uniform(-lim, lim, size=shape).astype(dtype)
            elif name.endswith('/gamma'):
                P[full] = (rng.uniform(0.5, 1.5, size=shape) if randomize_affine
                           else np.ones(shape)).astype(dtype)
            else:  # bias, beta
                P[full] = (rng.uniform(-0.1, 0.1, size=shape) if randomize_affine
                           else np.zeros(shape)).astype(dtype)
    return P


def synthetic_batch(batch, num_features=24, n_frames=128, dtype=np.float32):
    """z-scored MCEPs are unit Gaussian per coefficient."""
    A = np.random.default_rng(1234).standard_normal((batch, num_features, n_frames)).astype(dtype)
    Bm = np.random.default_rng(5678).standard_normal((batch, num_features, n_frames)).astype(dtype)
    return A, Bm


# ----------------------------------------------------------------------------
# Gated blocks
# ----------------------------------------------------------------------------
class _IN:
    """Hands out InstanceNorm{,_1,...} names in creation order."""
    def __init__(self, scope):
        self.scope, self.n = scope, 0

    def next(self):
        nm = 'InstanceNorm' if self.n == 0 else 'InstanceNorm_%d' % self.n
        self.n += 1
        return self.scope + '/' + nm


def _conv(P, name, x, stride, two_d=False):
    w, b = P[name + '/kernel'], P[name + '/bias']
    return (conv2d_fwd if two_d else conv1d_fwd)(x, w, b, stride)


def _gated_block_fwd(P, scope, inn, x, conv_name, gate_name, stride, two_d=False,
                     shuffle=False, norm=True):
    """downsample1d_block / downsample2d_block / upsample1d_block / the h1 pair
    (module.py:85-133, 161-163, 201-203): conv and conv_gates on the same input,
    [pixel shuffle], [instance norm], GLU."""
    c = {}
    h, c['conv'] = _conv(P, scope + '/' + conv_name, x, stride, two_d)
    if shuffle:
        h = pixel_shuffle_fwd(h)
    if norm:
        c['in_name'] = inn.next()
        h, c['in'] = instance_norm_fwd(h, P[c['in_name'] + '/gamma'], P[c['in_name'] + '/beta'])
    g, c['conv_g'] = _conv(P, scope + '/' + gate_name, x, stride, two_d)
    if shuffle:
        g = pixel_shuffle_fwd(g)
    if norm:
        c['in_g_name'] = inn.next()
        g, c['in_g'] = instance_norm_fwd(g, P[c['in_g_name'] + '/gamma'], P[c['in_g_name'] + '/beta'])
    y, c['glu'] = glu_fwd(h, g)
    c.update(conv_name=scope + '/' + conv_name, gate_name=scope + '/' + gate_name,
             two_d=two_d, shuffle=shuffle, norm=norm)
    return y, c


def _acc(G, name, val):
    if G is None:
        return
    G[name] = G[name] + val if name in G else val


def _gated_block_bwd(P, G, dy, c, need_dx=True):
    dh, dg = glu_bwd(dy, c['glu'])
    if c['norm']:
        dh, dga, dbe = instance_norm_bwd(dh, c['in'])
        _acc(G, c['in_name'] + '/gamma', dga); _acc(G, c['in_name'] + '/beta', dbe)
        dg, dga, dbe = instance_norm_bwd(dg, c['in_g'])
        _acc(G, c['in_g_name'] + '/gamma', dga); _acc(G, c['in_g_name'] + '/beta', dbe)
    if c['shuffle']:
        dh, dg = pixel_shuffle_bwd(dh), pixel_shuffle_bwd(dg)
    bwd = conv2d_bwd if c['two_d'] else conv1d_bwd
    dx1, dw, db = bwd(dh, P[c['conv_name'] + '/kernel'], c['conv'], need_dx)
    _acc(G, c['conv_name'] + '/kernel', dw); _acc(G, c['conv_name'] + '/bias', db)
    dx2, dw, db = bwd(dg, P[c['gate_name'] + '/kernel'], c['conv_g'], need_dx)
    _acc(G, c['gate_name'] + '/kernel', dw); _acc(G, c['gate_name'] + '/bias', db)
    return (dx1 + dx2) if need_dx else None


# ----------------------------------------------------------------------------
# generator_gatedcnn (module.py:148-185)
# ----------------------------------------------------------------------------
def generator_fwd(P, scope, inputs):
    """inputs [B, num_features, T] -> [B, num_features, T] (+ cache)."""
    inn = _IN(scope)
    C = {}
    x = np.transpose(inputs, (0, 2, 1))                                   # :152
    x, C['h1'] = _gated_block_fwd(P, scope, inn, x, 'h1_conv', 'h1_conv_gates', 1, norm=False)
    x, C['d1'] = _gated_block_fwd(P, scope, inn, x, 'downsample1d_block1_h1_conv',
                                  'downsample1d_block1_h1_gates', 2)      # :166
    x, C['d2'] = _gated_block_fwd(P, scope, inn, x, 'downsample1d_block2_h1_conv',
                                  'downsample1d_block2_h1_gates', 2)      # :167
    for i in range(1, 7):                                                 # :170-175
        blk = 'residual1d_block%d_' % i
        h, c1 = _gated_block_fwd(P, scope, inn, x, blk + 'h1_conv', blk + 'h1_gates', 1)
        h2, c2 = _conv(P, scope + '/' + blk + 'h2_conv', h, 1)
        nm = inn.next()
        h2n, c3 = instance_norm_fwd(h2, P[nm + '/gamma'], P[nm + '/beta'])
        x = x + h2n                                                       # :81
        C['r%d' % i] = (c1, c2, c3, nm, scope + '/' + blk + 'h2_conv')
    x, C['u1'] = _gated_block_fwd(P, scope, inn, x, 'upsample1d_block1_h1_conv',
                                  'upsample1d_block1_h1_gates', 1, shuffle=True)   # :178
    x, C['u2'] = _gated_block_fwd(P, scope, inn, x, 'upsample1d_block2_h1_conv',
                                  'upsample1d_block2_h1_gates', 1, shuffle=True)   # :179
    o, C['o1'] = _conv(P, scope + '/o1_conv', x, 1)                       # :182
    C['scope'] = scope
    return np.transpose(o, (0, 2, 1)), C                                  # :183


def generator_bwd(P, G, dout, C, need_dx=True):
    """dout [B, F, T]; accumulates parameter grads into dict G (may be None);
    returns d inputs [B, F, T] (or None)."""
    scope = C['scope']
    d = np.transpose(dout, (0, 2, 1))
    d, dw, db = conv1d_bwd(d, P[scope + '/o1_conv/kernel'], C['o1'])
    _acc(G, scope + '/o1_conv/kernel', dw); _acc(G, scope + '/o1_conv/bias', db)
    d = _gated_block_bwd(P, G, d, C['u2'])
    d = _gated_block_bwd(P, G, d, C['u1'])
    for i in range(6, 0, -1):
        c1, c2, c3, nm, h2name = C['r%d' % i]
        dh2, dga, dbe = instance_norm_bwd(d, c3)
        _acc(G, nm + '/gamma', dga); _acc(G, nm + '/beta', dbe)
        dh, dw, db = conv1d_bwd(dh2, P[h2name + '/kernel'], c2)
        _acc(G, h2name + '/kernel', dw); _acc(G, h2name + '/bias', db)
        d = d + _gated_block_bwd(P, G, dh, c1)
    d = _gated_block_bwd(P, G, d, C['d2'])
    d = _gated_block_bwd(P, G, d, C['d1'])
    d = _gated_block_bwd(P, G, d, C['h1'], need_dx=need_dx)
    return np.transpose(d, (0, 2, 1)) if need_dx else None


# ----------------------------------------------------------------------------
# discriminator (module.py:188-213)
# ----------------------------------------------------------------------------
def discriminator_fwd(P, scope, inputs):
    """inputs [B, F, T] -> sigmoid patch map [B, H', W', 1] (+ cache)."""
    inn = _IN(scope)
    C = {'scope': scope}
    x = inputs[..., None]                                                 # :192
    x, C['h1'] = _gated_block_fwd(P, scope, inn, x, 'h1_conv', 'h1_conv_gates', (1, 2),
                                  two_d=True, norm=False)
    x, C['d1'] = _gated_block_fwd(P, scope, inn, x, 'downsample2d_block1_h1_conv',
                                  'downsample2d_block1_h1_gates', (2, 2), two_d=True)
    x, C['d2'] = _gated_block_fwd(P, scope, inn, x, 'downsample2d_block2_h1_conv',
                                  'downsample2d_block2_h1_gates', (2, 2), two_d=True)
    x, C['d3'] = _gated_block_fwd(P, scope, inn, x, 'downsample2d_block3_h1_conv',
                                  'downsample2d_block3_h1_gates', (1, 2), two_d=True)
    z = x @ P[scope + '/dense/kernel'] + P[scope + '/dense/bias']         # :211 (S5)
    o = sigmoid(z)
    C['dense'] = (x, o)
    return o, C


def discriminator_bwd(P, G, dout, C, need_dx=True):
    scope = C['scope']
    x, o = C['dense']
    dz = dout * o * (1.0 - o)
    _acc(G, scope + '/dense/kernel', (x.reshape(-1, x.shape[-1]).T @ dz.reshape(-1, 1)))
    _acc(G, scope + '/dense/bias', dz.sum().reshape(1))
    d = dz @ P[scope + '/dense/kernel'].T
    d = _gated_block_bwd(P, G, d, C['d3'])
    d = _gated_block_bwd(P, G, d, C['d2'])
    d = _gated_block_bwd(P, G, d, C['d1'])
    d = _gated_block_bwd(P, G, d, C['h1'], need_dx=need_dx)
    return d[..., 0] if need_dx else None


# ----------------------------------------------------------------------------
# utils.py:6-12 (S6: means over every element; d|x|/dx = sign(x), sign(0)=0)
# ----------------------------------------------------------------------------
def l1_loss(y, y_hat):
    return np.mean(np.abs(y - y_hat))


def l2_loss(y, y_hat):
    return np.mean(np.square(y - y_hat))


# ----------------------------------------------------------------------------
# Loss graph + gradients: model.py:44-95
# ----------------------------------------------------------------------------
def generator_losses_and_grads(P, A, Bm, lambda_cycle, lambda_identity, want_grads=True):
    """G sub-step graph (model.py:44-72).  Returns (scalars dict, grads dict over
    generator variables, generation_A, generation_B)."""
    dt = A.dtype.type
    gen_B, c_genB = generator_fwd(P, 'generator_A2B', A)        # :44
    cyc_A, c_cycA = generator_fwd(P, 'generator_B2A', gen_B)    # :45
    gen_A, c_genA = generator_fwd(P, 'generator_B2A', Bm)       # :47
    cyc_B, c_cycB = generator_fwd(P, 'generator_A2B', gen_A)    # :48
    id_A, c_idA = generator_fwd(P, 'generator_B2A', A)          # :50
    id_B, c_idB = generator_fwd(P, 'generator_A2B', Bm)         # :51
    dA_fake, c_dA = discriminator_fwd(P, 'discriminator_A', gen_A)   # :53
    dB_fake, c_dB = discriminator_fwd(P, 'discriminator_B', gen_B)   # :54

    cycle_loss = l1_loss(A, cyc_A) + l1_loss(Bm, cyc_B)              # :57
    identity_loss = l1_loss(A, id_A) + l1_loss(Bm, id_B)             # :60
    g_A2B = l2_loss(np.ones_like(dB_fake), dB_fake)                  # :68
    g_B2A = l2_loss(np.ones_like(dA_fake), dA_fake)                  # :69
    g_loss = g_A2B + g_B2A + dt(lambda_cycle) * cycle_loss + dt(lambda_identity) * identity_loss  # :72
    S = dict(cycle_loss=cycle_loss, identity_loss=identity_loss, generator_loss_A2B=g_A2B,
             generator_loss_B2A=g_B2A, generator_loss=g_loss)
    # test aid: distance of the closest L1 element to a tie.  d|x|/dx = sign(x) (S6) is discontinuous there, so an
    # implementation whose forward differs by less than fp32 rounding may legitimately take the other sign of that one
    # element; parity tests pick inputs whose gap is far above that noise (tests/test_step_gpu.py).
    dists = []
    if lambda_cycle != 0:
        dists += [np.abs(cyc_A - A).ravel(), np.abs(cyc_B - Bm).ravel()]
    if lambda_identity != 0:
        dists += [np.abs(id_A - A).ravel(), np.abs(id_B - Bm).ravel()]
    generator_losses_and_grads.last_l1_dist = np.concatenate(dists) if dists else np.zeros(0)   # every |y_hat - y|
    generator_losses_and_grads.last_l1_gap = float(generator_losses_and_grads.last_l1_dist.min()) if dists else float('inf')
    if not want_grads:
        return S, None, gen_A, gen_B

    G = {}
    n_x = dt(A.size)
    n_d = dt(dA_fake.size)
    # adversarial terms: gradient flows through D's *inputs* only (var_list = generator_vars, :95,:108)
    d_genA = discriminator_bwd(P, None, -2.0 * (1.0 - dA_fake) / n_d, c_dA)
    d_genB = discriminator_bwd(P, None, -2.0 * (1.0 - dB_fake) / n_d, c_dB)
    # cycle terms
    d_genB = d_genB + generator_bwd(P, G, dt(lambda_cycle) * np.sign(cyc_A - A) / n_x, c_cycA)
    d_genA = d_genA + generator_bwd(P, G, dt(lambda_cycle) * np.sign(cyc_B - Bm) / n_x, c_cycB)
    # first-hop generators
    generator_bwd(P, G, d_genB, c_genB, need_dx=False)
    generator_bwd(P, G, d_genA, c_genA, need_dx=False)
    # identity terms
    generator_bwd(P, G, dt(lambda_identity) * np.sign(id_A - A) / n_x, c_idA, need_dx=False)
    generator_bwd(P, G, dt(lambda_identity) * np.sign(id_B - Bm) / n_x, c_idB, need_dx=False)
    return S, G, gen_A, gen_B


def discriminator_losses_and_grads(P, A, Bm, A_fake, B_fake, want_grads=True):
    """D sub-step graph (model.py:75-90)."""
    dt = A.dtype.type
    dA_real, cAr = discriminator_fwd(P, 'discriminator_A', A)         # :75
    dB_real, cBr = discriminator_fwd(P, 'discriminator_B', Bm)        # :76
    dA_fake, cAf = discriminator_fwd(P, 'discriminator_A', A_fake)    # :77
    dB_fake, cBf = discriminator_fwd(P, 'discriminator_B', B_fake)    # :78
    lAr = l2_loss(np.ones_like(dA_real), dA_real)                     # :81
    lAf = l2_loss(np.zeros_like(dA_fake), dA_fake)                    # :82
    d_loss_A = (lAr + lAf) / dt(2)                                    # :83
    lBr = l2_loss(np.ones_like(dB_real), dB_real)                     # :85
    lBf = l2_loss(np.zeros_like(dB_fake), dB_fake)                    # :86
    d_loss_B = (lBr + lBf) / dt(2)                                    # :87
    d_loss = d_loss_A + d_loss_B                                      # :90
    S = dict(discriminator_loss_A=d_loss_A, discriminator_loss_B=d_loss_B, discriminator_loss=d_loss)
    if not want_grads:
        return S, None
    G = {}
    n = dt(dA_real.size)
    discriminator_bwd(P, G, -(1.0 - dA_real) / n, cAr, need_dx=False)
    discriminator_bwd(P, G, dA_fake / n, cAf, need_dx=False)
    discriminator_bwd(P, G, -(1.0 - dB_real) / n, cBr, need_dx=False)
    discriminator_bwd(P, G, dB_fake / n, cBf, need_dx=False)
    return S, G


# ----------------------------------------------------------------------------
# S7: tf.train.AdamOptimizer(lr, beta1=0.5) -- TF form, float32 beta powers.
# ----------------------------------------------------------------------------
class AdamTF:
    """ApplyAdam: alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1);
    v += (g*g-v)(1-b2); var -= alpha*m/(sqrt(v)+eps).  beta powers are
    variables of the parameter dtype, multiplied once per step after the update."""

    def __init__(self, names, P):
        dt = next(iter(P.values())).dtype.type
        self.dt = dt
        self.m = {n: np.zeros_like(P[n]) for n in names}
        self.v = {n: np.zeros_like(P[n]) for n in names}
        self.b1p, self.b2p = dt(ADAM_BETA1), dt(ADAM_BETA2)

    def alpha(self, lr):
        dt = self.dt
        return dt(lr) * np.sqrt(dt(1) - self.b2p) / (dt(1) - self.b1p)

    def apply(self, P, G, lr):
        dt = self.dt
        a = self.alpha(lr)
        b1, b2, eps = dt(ADAM_BETA1), dt(ADAM_BETA2), dt(ADAM_EPS)
        for n in self.m:
            g = G[n].reshape(P[n].shape).astype(P[n].dtype)
            self.m[n] += (g - self.m[n]) * (dt(1) - b1)
            self.v[n] += (g * g - self.v[n]) * (dt(1) - b2)
            P[n] -= (self.m[n] * a) / (np.sqrt(self.v[n]) + eps)
        self.b1p = dt(self.b1p * b1)
        self.b2p = dt(self.b2p * b2)


class CycleGANOracle:
    """model.CycleGAN restated (train mode only; no session/saver/summaries)."""

    def __init__(self, params):
        self.P = params
        self.g_names = [n for n in params if 'generator' in n]        # model.py:95
        self.d_names = [n for n in params if 'discriminator' in n]    # model.py:94
        self.opt_g = AdamTF(self.g_names, params)
        self.opt_d = AdamTF(self.d_names, params)
        self.last = {}

    def train(self, input_A, input_B, lambda_cycle, lambda_identity,
              generator_learning_rate, discriminator_learning_rate):
        """model.py:110-125.  Both sub-steps see the *pre-update* weights of the
        other player: the G run fetches generation_A/B before its own update is
        visible, and the D run consumes those fetched arrays."""
        dt = next(iter(self.P.values())).dtype
        A, Bm = np.asarray(input_A, dtype=dt), np.asarray(input_B, dtype=dt)
        Sg, Gg, gen_A, gen_B = generator_losses_and_grads(self.P, A, Bm, lambda_cycle, lambda_identity)
        self.opt_g.apply(self.P, Gg, generator_learning_rate)          # :112-114
        Sd, Gd = discriminator_losses_and_grads(self.P, A, Bm, gen_A, gen_B)
        self.opt_d.apply(self.P, Gd, discriminator_learning_rate)      # :118-119
        self.last = dict(Sg); self.last.update(Sd)
        self.last_grads = (Gg, Gd)
        self.last_fakes = (gen_A, gen_B)
        self.last_l1_gap = generator_losses_and_grads.last_l1_gap
        self.last_l1_dist = generator_losses_and_grads.last_l1_dist
        return Sg['generator_loss'], Sd['discriminator_loss']

    def test(self, inputs, direction):                                  # model.py:128-137
        dt = next(iter(self.P.values())).dtype
        x = np.asarray(inputs, dtype=dt)
        if direction == 'A2B':
            return generator_fwd(self.P, 'generator_A2B', x)[0]
        elif direction == 'B2A':
            return generator_fwd(self.P, 'generator_B2A', x)[0]
        raise Exception('Conversion direction must be specified.')
