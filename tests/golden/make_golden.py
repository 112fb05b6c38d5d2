"""Generates tests/golden/*.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference itself cannot be executed (no
TensorFlow in the container), so these vectors pin the ORACLE (regression)
and give the GPU tests an oracle-free comparison target; they are not reference outputs.

 golden_shrunk_f64.npz : width/16 networks, float64, B=2, T=16 -- losses + every gradient.
 golden_full_f32.npz   : the real architecture, float32, B=1, 24x128, seed 42 weights with
                         randomised bias/beta/gamma -- 8 losses, generation_A/B, per-tensor
                         gradient L2 norms, and losses of train steps 1..3.
 train_log_pins.npz    : five numbers copied from the reference's TensorBoard export
                         (train_log/*.csv row 2, step 167) that pin model.py:72's loss weights.
 golden_f32_b32.npz / golden_f32_36x256_b8.npz / golden_bf16_b64_lam0.npz : the same record at the other sizes
                         bench.py reports (`make_golden.py b32|36x256|b64lam0`); lam0 = lambda_cycle = lambda_identity = 0.
 golden_f32_b256.npz   : the same at BASELINE configs[3]'s global batch (256: the single-GPU end of the strong-scaling curve), generations by the
                         first 4 samples + norm / projection; b32 / b256 also hold two COMPLETE gradient tensors (one generator, one
                         discriminator filter) and every 64th element of the largest one (`make_golden.py b32|b256`).
 golden_f32_b64.npz    : one train step (model.py:110-125) at the batch bench.py times as BASELINE configs[2]
 golden_bf16_b64.npz     (B=64, 24x128), fp32 and with bf16 operand rounding (O.set_operand_rounding): the 8 logged
                         scalars, generation_A/B, per-tensor gradient L2 + projection, and the cumulative
                         distribution of the L1 terms' |y_hat - y| (how many utils.l1_loss elements sit within a
                         given distance of a sign flip -- the test derives its gradient tolerance from it).
                         `python tests/golden/make_golden.py b64` (minutes of CPU; not part of the default run).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cyclegan_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def probe(a, i):
    r = np.random.default_rng(1000 + i).standard_normal(a.size)
    return float(a.reshape(-1).astype(np.float64) @ r)


def shrunk():
    P = O.init_params(seed=3, dtype=np.float64, randomize_affine=True, wd=16)
    rng = np.random.default_rng(4)
    A, B = rng.standard_normal((2, 24, 16)), rng.standard_normal((2, 24, 16))
    Sg, Gg, gA, gB = O.generator_losses_and_grads(P, A, B, 10.0, 5.0)
    Sd, Gd = O.discriminator_losses_and_grads(P, A, B, gA, gB)
    out = {'A': A, 'B': B, 'gen_A': gA, 'gen_B': gB}
    out.update({'loss/' + k: np.float64(v) for k, v in {**Sg, **Sd}.items()})
    # every gradient tensor is pinned by two numbers: its L2 norm and its projection on a fixed
    # pseudo-random direction (keeps the fixture small; any wrong element moves both)
    names = sorted({**Gg, **Gd})
    out['grad_names'] = np.array(names)
    out['grad_l2'] = np.array([np.linalg.norm({**Gg, **Gd}[n]) for n in names])
    out['grad_proj'] = np.array([probe({**Gg, **Gd}[n], i) for i, n in enumerate(names)])
    np.savez_compressed(os.path.join(HERE, 'golden_shrunk_f64.npz'), **out)


def full():
    P = O.init_params(seed=42, randomize_affine=True)
    A, B = O.synthetic_batch(1, 24, 128)
    orc = O.CycleGANOracle({k: v.copy() for k, v in P.items()})
    out = {}
    for step in range(3):
        g, d = orc.train(A, B, 10.0, 5.0, 2e-4, 1e-4)
        out['step%d/losses' % step] = np.array([orc.last[k] for k in (
            'cycle_loss', 'identity_loss', 'generator_loss_A2B', 'generator_loss_B2A', 'generator_loss',
            'discriminator_loss_A', 'discriminator_loss_B', 'discriminator_loss')], np.float32)
        if step == 0:
            out['gen_A'], out['gen_B'] = orc.last_fakes
            Gg, Gd = orc.last_grads
            names = sorted({**Gg, **Gd})
            out['grad_names'] = np.array(names)
            out['grad_l2'] = np.array([np.linalg.norm({**Gg, **Gd}[n].astype(np.float64)) for n in names])
            out['grad_proj'] = np.array([probe({**Gg, **Gd}[n], i) for i, n in enumerate(names)])
    np.savez_compressed(os.path.join(HERE, 'golden_full_f32.npz'), **out)


LOSS_KEYS = ('cycle_loss', 'identity_loss', 'generator_loss_A2B', 'generator_loss_B2A', 'generator_loss',
             'discriminator_loss_A', 'discriminator_loss_B', 'discriminator_loss')
L1_CDF_X = np.logspace(-7, -1, 49)          # distances of an L1 element from a tie at which the CDF is recorded


# complete gradient tensors kept in the large-batch records (everything else is pinned by two numbers per tensor): one generator
# and one discriminator filter gradient element by element, the largest one (D.d3, 9.4 M floats) by every 64th element
FULL_TENSORS = ('generator_A2B/o1_conv/kernel', 'discriminator_A/downsample2d_block1_h1_conv/kernel')
STRIDED_TENSORS = (('discriminator_B/downsample2d_block3_h1_conv/kernel', 64),)


def step_golden(batch, rounding, F=24, T=128, lam=(10.0, 5.0), name=None, keep_samples=0, keep_full=False):
    """One step (model.py:110-125) at `batch` x F x T with the weights / batch generator of full(); rounding None or 'bf16';
    lam = (lambda_cycle, lambda_identity)."""
    P = O.init_params(seed=42, num_features=F, randomize_affine=True)
    A, B = O.synthetic_batch(batch, F, T)
    O.set_operand_rounding(rounding)
    try:
        orc = O.CycleGANOracle({k: v.copy() for k, v in P.items()})
        orc.train(A, B, lam[0], lam[1], 2e-4, 1e-4)
    finally:
        O.set_operand_rounding(None)
    out = {'losses': np.array([orc.last[k] for k in LOSS_KEYS], np.float32)}
    gA, gB = orc.last_fakes
    if keep_samples:      # large batches: the first samples verbatim, the whole tensors by L2 norm + random projection
        out['gen_l2'] = np.array([np.linalg.norm(gA.astype(np.float64)), np.linalg.norm(gB.astype(np.float64))])
        out['gen_proj'] = np.array([probe(gA, 777), probe(gB, 778)])
        gA, gB = gA[:keep_samples], gB[:keep_samples]
    out['gen_A'], out['gen_B'] = gA, gB
    Gg, Gd = orc.last_grads
    G = {**Gg, **Gd}
    names = sorted(G)
    out['grad_names'] = np.array(names)
    out['grad_l2'] = np.array([np.linalg.norm(G[n].astype(np.float64)) for n in names])
    out['grad_proj'] = np.array([probe(G[n], i) for i, n in enumerate(names)])
    if keep_full:
        for n in FULL_TENSORS:
            out['full/' + n] = G[n].astype(np.float32)
        for n, st in STRIDED_TENSORS:
            out['strided%d/' % st + n] = G[n].astype(np.float32).reshape(-1)[::st].copy()
    d = np.sort(orc.last_l1_dist.astype(np.float64))
    out['l1_cdf_x'] = L1_CDF_X
    out['l1_cdf_y'] = np.searchsorted(d, L1_CDF_X, side='right') / float(d.size)
    out['l1_count'] = np.int64(d.size)
    np.savez_compressed(os.path.join(HERE, name or 'golden_%s_b64.npz' % ('bf16' if rounding else 'f32')), **out)


def full_b64(rounding):
    step_golden(64, rounding)


def train_log_pins():
    # /root/reference/train_log/run_20180619-011402-tag-generator_summaries_*.csv, line 2 (step 167)
    np.savez(os.path.join(HERE, 'train_log_pins.npz'), step=167, cycle_loss=1.444779872894287,
             identity_loss=1.3221745491027832, generator_loss_A2B=0.8852359652519226,
             generator_loss_B2A=0.9709569811820984, generator_loss=22.91486358642578,
             lambda_cycle=10.0, lambda_identity=5.0)


if __name__ == '__main__':
    if sys.argv[1:] == ['b64']:
        full_b64(None); full_b64('bf16')
    elif sys.argv[1:] == ['b32']:              # BASELINE configs[3]'s per-GPU share (bench.py extra.configs): fp32, B=32, 24x128
        step_golden(32, None, name='golden_f32_b32.npz', keep_full=True)
    elif sys.argv[1:] == ['b256']:             # BASELINE configs[3]: global batch 256 -- the N = 1 end of the strong-scaling curve (~8 min of CPU)
        step_golden(256, None, name='golden_f32_b256.npz', keep_samples=4, keep_full=True)
    elif sys.argv[1:] == ['36x256']:           # BASELINE configs[4]'s geometry at a batch the oracle finishes in minutes
        step_golden(8, None, F=36, T=256, name='golden_f32_36x256_b8.npz')
    elif sys.argv[1:] == ['36x256b64']:        # ... and at configs[4]'s per-GPU share itself (the size bench.py times; ~4 min of CPU)
        step_golden(64, None, F=36, T=256, name='golden_f32_36x256_b64.npz', keep_samples=4)
    elif sys.argv[1:] == ['b64lam0']:          # bf16 B=64 with the L1 terms switched off: a smooth objective, tight gradient bound
        step_golden(64, 'bf16', lam=(0.0, 0.0), name='golden_bf16_b64_lam0.npz')
    else:
        shrunk(); full(); train_log_pins()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
