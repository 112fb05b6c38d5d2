"""Second, independent CPU restatement of the reference graph on torch-CPU
autograd -- TEST INFRASTRUCTURE ONLY (see oracle/cyclegan_oracle.py header;
PARITY UNPINNED for the same reason).

Purpose: cross-validate the numpy oracle's hand-written backward passes and
its im2col convolutions with a differently-built implementation (channels-first
F.conv1d/F.conv2d with explicit TF-'SAME' F.pad, autograd for every gradient).
Follows /root/reference/module.py:3-213 and model.py:44-90.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .cyclegan_oracle import same_pad, IN_EPS


def _conv1d(P, name, x, stride):           # x [B, C, T]; TF kernel [k, Cin, Cout]
    w = P[name + '/kernel'].permute(2, 1, 0)
    _, pl, pr = same_pad(x.shape[-1], w.shape[-1], stride)
    return F.conv1d(F.pad(x, (pl, pr)), w, P[name + '/bias'], stride=stride)


def _conv2d(P, name, x, strides):          # x [B, C, H, W]; TF kernel [kh, kw, Cin, Cout]
    w = P[name + '/kernel'].permute(3, 2, 0, 1)
    _, pt, pb = same_pad(x.shape[-2], w.shape[-2], strides[0])
    _, pl, pr = same_pad(x.shape[-1], w.shape[-1], strides[1])
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, P[name + '/bias'], stride=strides)


class _IN:
    def __init__(self, P, scope):
        self.P, self.scope, self.n = P, scope, 0

    def __call__(self, x):                 # channels-first; stats over all trailing dims
        nm = self.scope + ('/InstanceNorm' if self.n == 0 else '/InstanceNorm_%d' % self.n)
        self.n += 1
        dims = tuple(range(2, x.dim()))
        mean = x.mean(dim=dims, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=dims, keepdim=True)
        shp = (1, -1) + (1,) * (x.dim() - 2)
        return (x - mean) * torch.rsqrt(var + IN_EPS) * self.P[nm + '/gamma'].view(shp) + \
            self.P[nm + '/beta'].view(shp)


def _ps(x):   # channels-first view of tf.reshape([n,w,c]->[n,2w,c/2]) done in channels-last
    n, c, w = x.shape
    return x.permute(0, 2, 1).reshape(n, 2 * w, c // 2).permute(0, 2, 1)


def generator(P, scope, x):                # [B, F, T] -> [B, F, T]
    inn = _IN(P, scope)
    s = scope + '/'
    h = _conv1d(P, s + 'h1_conv', x, 1) * torch.sigmoid(_conv1d(P, s + 'h1_conv_gates', x, 1))
    for blk in ('downsample1d_block1_', 'downsample1d_block2_'):
        a = inn(_conv1d(P, s + blk + 'h1_conv', h, 2))
        g = inn(_conv1d(P, s + blk + 'h1_gates', h, 2))
        h = a * torch.sigmoid(g)
    for i in range(1, 7):
        blk = 'residual1d_block%d_' % i
        a = inn(_conv1d(P, s + blk + 'h1_conv', h, 1))
        g = inn(_conv1d(P, s + blk + 'h1_gates', h, 1))
        h = h + inn(_conv1d(P, s + blk + 'h2_conv', a * torch.sigmoid(g), 1))
    for blk in ('upsample1d_block1_', 'upsample1d_block2_'):
        a = inn(_ps(_conv1d(P, s + blk + 'h1_conv', h, 1)))
        g = inn(_ps(_conv1d(P, s + blk + 'h1_gates', h, 1)))
        h = a * torch.sigmoid(g)
    return _conv1d(P, s + 'o1_conv', h, 1)


def discriminator(P, scope, x):            # [B, F, T] -> [B, H', W', 1]
    inn = _IN(P, scope)
    s = scope + '/'
    x = x.unsqueeze(1)
    h = _conv2d(P, s + 'h1_conv', x, (1, 2)) * torch.sigmoid(_conv2d(P, s + 'h1_conv_gates', x, (1, 2)))
    for blk, st in (('downsample2d_block1_', (2, 2)), ('downsample2d_block2_', (2, 2)),
                    ('downsample2d_block3_', (1, 2))):
        a = inn(_conv2d(P, s + blk + 'h1_conv', h, st))
        g = inn(_conv2d(P, s + blk + 'h1_gates', h, st))
        h = a * torch.sigmoid(g)
    z = h.permute(0, 2, 3, 1) @ P[s + 'dense/kernel'] + P[s + 'dense/bias']
    return torch.sigmoid(z)


def losses_and_grads(P_np, A, Bm, lambda_cycle, lambda_identity):
    """Returns (scalars, grads dict over all variables) with the reference's
    variable partition: generator grads from generator_loss, discriminator
    grads from discriminator_loss evaluated on the fetched (detached) fakes."""
    tdt = torch.float64 if A.dtype == np.float64 else torch.float32
    P = {k: torch.tensor(v, dtype=tdt, requires_grad=True) for k, v in P_np.items()}
    A, Bm = torch.tensor(A, dtype=tdt), torch.tensor(Bm, dtype=tdt)
    gen_B = generator(P, 'generator_A2B', A); cyc_A = generator(P, 'generator_B2A', gen_B)
    gen_A = generator(P, 'generator_B2A', Bm); cyc_B = generator(P, 'generator_A2B', gen_A)
    id_A = generator(P, 'generator_B2A', A); id_B = generator(P, 'generator_A2B', Bm)
    dA_f = discriminator(P, 'discriminator_A', gen_A); dB_f = discriminator(P, 'discriminator_B', gen_B)
    cyc = (A - cyc_A).abs().mean() + (Bm - cyc_B).abs().mean()
    idl = (A - id_A).abs().mean() + (Bm - id_B).abs().mean()
    gA2B = ((1 - dB_f) ** 2).mean(); gB2A = ((1 - dA_f) ** 2).mean()
    g_loss = gA2B + gB2A + lambda_cycle * cyc + lambda_identity * idl
    g_names = [n for n in P if 'generator' in n]
    g_grads = torch.autograd.grad(g_loss, [P[n] for n in g_names])
    fA, fB = gen_A.detach(), gen_B.detach()
    dlA = (((1 - discriminator(P, 'discriminator_A', A)) ** 2).mean() +
           (discriminator(P, 'discriminator_A', fA) ** 2).mean()) / 2
    dlB = (((1 - discriminator(P, 'discriminator_B', Bm)) ** 2).mean() +
           (discriminator(P, 'discriminator_B', fB) ** 2).mean()) / 2
    d_loss = dlA + dlB
    d_names = [n for n in P if 'discriminator' in n]
    d_grads = torch.autograd.grad(d_loss, [P[n] for n in d_names])
    S = dict(cycle_loss=cyc.item(), identity_loss=idl.item(), generator_loss_A2B=gA2B.item(),
             generator_loss_B2A=gB2A.item(), generator_loss=g_loss.item(),
             discriminator_loss_A=dlA.item(), discriminator_loss_B=dlB.item(),
             discriminator_loss=d_loss.item())
    G = {n: g.numpy() for n, g in zip(g_names, g_grads)}
    G.update({n: g.numpy() for n, g in zip(d_names, d_grads)})
    return S, G, gen_A.detach().numpy(), gen_B.detach().numpy()


class TorchTrainer:
    """model.CycleGAN.train (model.py:110-125) on torch-CPU (oneDNN / MKL convolutions, autograd, all host threads) with
    persistent parameters and TF-form Adam (model.py:107-108; rule S7) -- the fair CPU baseline bench.py times beside the
    numpy oracle.  Same arithmetic as losses_and_grads + cyclegan_oracle.AdamTF (tests/test_oracle.py compares them)."""

    def __init__(self, P_np):
        self.P = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in P_np.items()}
        self.g_names = [n for n in self.P if 'generator' in n]
        self.d_names = [n for n in self.P if 'discriminator' in n]
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.b1p = {'g': np.float32(0.5), 'd': np.float32(0.5)}
        self.b2p = {'g': np.float32(0.999), 'd': np.float32(0.999)}

    def _adam(self, names, grads, lr, k):
        one = np.float32(1)
        alpha = float(np.float32(lr) * np.sqrt(one - self.b2p[k]) / (one - self.b1p[k]))
        with torch.no_grad():
            for n, g in zip(names, grads):
                m, v = self.m[n], self.v[n]
                m.add_(g - m, alpha=0.5)
                v.add_(g * g - v, alpha=0.001)
                self.P[n].sub_(m * alpha / (v.sqrt() + 1e-8))
        self.b1p[k] = np.float32(self.b1p[k] * np.float32(0.5)); self.b2p[k] = np.float32(self.b2p[k] * np.float32(0.999))

    def train(self, A, Bm, lambda_cycle, lambda_identity, lr_g, lr_d):
        P = self.P
        A, Bm = torch.as_tensor(A, dtype=torch.float32), torch.as_tensor(Bm, dtype=torch.float32)
        gen_B = generator(P, 'generator_A2B', A); cyc_A = generator(P, 'generator_B2A', gen_B)
        gen_A = generator(P, 'generator_B2A', Bm); cyc_B = generator(P, 'generator_A2B', gen_A)
        id_A = generator(P, 'generator_B2A', A); id_B = generator(P, 'generator_A2B', Bm)
        dA_f = discriminator(P, 'discriminator_A', gen_A); dB_f = discriminator(P, 'discriminator_B', gen_B)
        cyc = (A - cyc_A).abs().mean() + (Bm - cyc_B).abs().mean()
        idl = (A - id_A).abs().mean() + (Bm - id_B).abs().mean()
        g_loss = ((1 - dB_f) ** 2).mean() + ((1 - dA_f) ** 2).mean() + lambda_cycle * cyc + lambda_identity * idl
        g_grads = torch.autograd.grad(g_loss, [P[n] for n in self.g_names])
        fA, fB = gen_A.detach(), gen_B.detach()
        self._adam(self.g_names, g_grads, lr_g, 'g')                       # model.py:112-114
        dlA = (((1 - discriminator(P, 'discriminator_A', A)) ** 2).mean() + (discriminator(P, 'discriminator_A', fA) ** 2).mean()) / 2
        dlB = (((1 - discriminator(P, 'discriminator_B', Bm)) ** 2).mean() + (discriminator(P, 'discriminator_B', fB) ** 2).mean()) / 2
        d_loss = dlA + dlB
        d_grads = torch.autograd.grad(d_loss, [P[n] for n in self.d_names])
        self._adam(self.d_names, d_grads, lr_d, 'd')                       # model.py:118-119
        return float(g_loss.detach()), float(d_loss.detach())
