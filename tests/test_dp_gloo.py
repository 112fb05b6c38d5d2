"""CPU, world_size 2, gloo: the data-parallel gradient exchange used for N>1 GPUs
(voice_converter_cyclegan_amd.engine.GradSync: SUM all-reduce + 1/world folded into Adam).
Each rank computes oracle gradients on its shard of the minibatch; after the exchange the
scaled gradient must equal the single-process full-batch gradient, and both
ranks must hold bit-identical parameters after the TF-Adam update."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _flatten(G, names):
    return np.concatenate([G[n].reshape(-1) for n in names])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import vcgan_pkg  # noqa: F401
    from voice_converter_cyclegan_amd.engine import GradSync
    from oracle import cyclegan_oracle as O
    P = O.init_params(seed=8, dtype=np.float64, randomize_affine=True, wd=16)
    rng = np.random.default_rng(9)
    A, B = rng.standard_normal((4, 24, 16)), rng.standard_normal((4, 24, 16))
    sh = slice(rank * 2, rank * 2 + 2)                                  # contiguous shard per rank
    Sg, Gg, gA, gB = O.generator_losses_and_grads(P, A[sh], B[sh], 10.0, 5.0)
    Sd, Gd = O.discriminator_losses_and_grads(P, A[sh], B[sh], gA, gB)
    names = sorted({**Gg, **Gd})
    flat = torch.from_numpy(_flatten({**Gg, **Gd}, names))
    sync = GradSync()
    assert sync.world == world
    nG = sum(Gg[n].size for n in names if 'generator' in n)
    # same two-segment exchange as CycleGANEngine.exchange_grads
    w1 = sync.all_reduce(flat[nG:], async_op=True)
    w2 = sync.all_reduce(flat[:nG], async_op=True)
    w1.wait(); w2.wait()
    avg = flat.numpy() * sync.scale
    # TF-Adam on the exchanged gradient (grad_scale = 1/world folded in, as vc_adam_tf_update does)
    opt = O.AdamTF(names, P)
    off = 0
    G = {}
    for n in names:
        G[n] = avg[off:off + P[n].size].reshape(P[n].shape); off += P[n].size
    opt.apply(P, G, 2e-4)
    losses = torch.tensor([Sg['generator_loss'], Sd['discriminator_loss']], dtype=torch.float64)
    sync.all_reduce(losses)
    q.put((rank, avg, _flatten(P, names), (losses * sync.scale).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_full_batch():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import cyclegan_oracle as O
    P = O.init_params(seed=8, dtype=np.float64, randomize_affine=True, wd=16)
    rng = np.random.default_rng(9)
    A, B = rng.standard_normal((4, 24, 16)), rng.standard_normal((4, 24, 16))
    Sg, Gg, gA, gB = O.generator_losses_and_grads(P, A, B, 10.0, 5.0)
    Sd, Gd = O.discriminator_losses_and_grads(P, A, B, gA, gB)
    names = sorted({**Gg, **Gd})
    full = _flatten({**Gg, **Gd}, names)
    assert np.array_equal(res[0][1], res[1][1]), 'ranks disagree on the exchanged gradient'
    assert np.array_equal(res[0][2], res[1][2]), 'replicas diverged after Adam'
    assert np.abs(res[0][1] - full).max() < 1e-12 * np.abs(full).max() + 1e-15
    assert abs(res[0][3][0] - Sg['generator_loss']) < 1e-12 and abs(res[0][3][1] - Sd['discriminator_loss']) < 1e-12
