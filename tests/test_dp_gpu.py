"""-m gpu: the N>1 path of the engine for real -- two ranks, one process each, sharing the single GPU of the test
box through the gloo backend (RCCL refuses two ranks on one device; the code path in the engine is identical:
event-gated, bucketed, asynchronous all-reduce of the flat gradient + 1/world folded into Adam).
Each rank runs a B=1 shard; the exchanged gradient must equal the single-process B=2 gradient and the replicas
must stay bit-identical after the update."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import __graft_entry__ as ge
    ge.build()
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    from oracle import cyclegan_oracle as O
    # a tie-free batch (tests/test_step_gpu.py::_setup): no L1 element of the oracle's forward within 5e-5 of a sign flip, so the
    # full-batch engine and the two shard engines cannot legitimately disagree on a sign(.) of utils.l1_loss
    for seed in range(13, 13 + 64):
        P = O.init_params(seed=seed, randomize_affine=True)
        rng = np.random.default_rng(seed + 1)
        A = rng.standard_normal((world, 24, 128)).astype(np.float32); B = rng.standard_normal((world, 24, 128)).astype(np.float32)
        O.generator_losses_and_grads(P, A, B, 10.0, 5.0, want_grads=False)
        if O.generator_losses_and_grads.last_l1_gap >= 5e-5:
            break
    else:
        raise AssertionError('no tie-free seed')
    ref = CycleGANEngine(24, device='cuda:0', data_parallel=False)
    ref.load_tf_variables(P)
    ref.step_grads(A, B, 10.0, 5.0)
    ref_g = ref.grads.clone(); ref_l = ref.losses.clone()
    eng = CycleGANEngine(24, device='cuda:0')
    assert eng.sync.world == world
    eng.load_tf_variables(P)
    segs = eng.grad_segments(eng.plan(128))
    covered = sorted(r for _, _, rs in segs for r in rs)
    assert covered[0][0] == 0 and covered[-1][1] == eng.n_flat and all(a[1] == b[0] for a, b in zip(covered, covered[1:])), covered
    p = eng.step_grads(A[rank:rank + 1], B[rank:rank + 1], 10.0, 5.0)
    eng.exchange_grads(p)
    torch.cuda.synchronize()
    rel = float(((eng.grads * eng.sync.scale - ref_g).double().norm() / ref_g.double().norm()))
    g, d = eng.train_step(A[rank:rank + 1], B[rank:rank + 1], 10.0, 5.0, 2e-4, 1e-4)      # full path incl. Adam + loss averaging
    torch.cuda.synchronize()
    # the same step with the exchange as ONE all-reduce after the backward (vc_plan_set_exchange_mode / engine.set_exchange_mode)
    one = CycleGANEngine(24, device='cuda:0')
    one.load_tf_variables(P)
    one.set_exchange_mode('single')
    assert one.grad_segments(one.plan(128)) == [('all', 0, [(0, one.n_flat)])]
    g1, d1 = one.train_step(A[rank:rank + 1], B[rank:rank + 1], 10.0, 5.0, 2e-4, 1e-4)
    torch.cuda.synchronize()
    # (default mode: split-K atomics, so the two engines' shard gradients agree to ~1e-6, and Adam's first step is ~lr sign(g))
    moved = float(((one.params - eng.params).abs() > 0.05 * 2e-4).double().mean())
    np.save(os.path.join(outdir, 'r%d.npy' % rank),
            np.array([rel, float(eng.params.double().sum()), float(eng.params.double().abs().sum()), g, d,
                      float(ref_l[4]), float(ref_l[7]), float(one.params.double().sum()), float(one.params.double().abs().sum()), g1, d1, moved]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_gpu(built_lib, tmp_path):
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    r0, r1 = (np.load(os.path.join(str(tmp_path), 'r%d.npy' % r)) for r in range(2))
    assert r0[0] < 1e-3 and r1[0] < 1e-3, (r0[0], r1[0])          # exchanged gradient == full-batch gradient (rel-L2; tie-free batch)
    assert r0[1] == r1[1] and r0[2] == r1[2], 'replicas diverged after the update'
    assert abs(r0[3] - r0[5]) < 1e-3 * abs(r0[5]) and abs(r0[4] - r0[6]) < 1e-3 * max(abs(r0[6]), 1e-3)   # losses = global means
    assert r0[3] == r1[3] and r0[4] == r1[4]
    # single-collective mode: replicas identical too, same losses, and (up to weights whose gradient is within rounding of zero) the same update
    assert r0[7] == r1[7] and r0[8] == r1[8], 'replicas diverged after the single-collective update'
    assert abs(r0[9] - r0[3]) < 1e-4 * abs(r0[3]) and abs(r0[10] - r0[4]) < 1e-4 * max(abs(r0[4]), 1e-3), (r0[9], r0[3])
    assert r0[11] < 1e-3, r0[11]


def _nccl_worker(port, outdir):
    """the production configuration of GradSync: backend 'nccl' (= RCCL), native communicator"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    import __graft_entry__ as ge
    ge.build()
    from voice_converter_cyclegan_amd.engine import CycleGANEngine, GradSync
    from voice_converter_cyclegan_amd._lib import lib
    s = GradSync(None, enabled=True, device=torch.device('cuda:0'))
    np.save(os.path.join(outdir, 'nccl.npy'), np.array([s.world, 0 if s.comm is None else 1]))
    dist.destroy_process_group()


def test_rccl_through_the_c_abi_single_rank(built_lib, tmp_path):
    """vc_comm_unique_id / vc_comm_init (ncclCommInitRank) / vc_allreduce_f32 (ncclAllReduce) / vc_step_exchange_and_update
    with a communicator / vc_comm_destroy on a 1-rank RCCL communicator: the only RCCL configuration a 1-GPU box offers
    (RCCL refuses two ranks on one device).  With a communicator the native exchange ISSUES its collectives whatever the
    rank count -- ncclAllReduce of the discriminator half, then ncclGroupStart / 2 x ncclAllReduce / ncclGroupEnd per generator
    bucket, each on the update stream behind vc_step_wait_d_grads / vc_step_wait_g_bucket -- and a SUM over one rank is the
    identity: in deterministic mode the parameters after the update must equal the no-communicator update BIT FOR BIT."""
    import ctypes as C
    from voice_converter_cyclegan_amd import _lib
    from voice_converter_cyclegan_amd._lib import lib, ptr, check
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    ident = (C.c_ubyte * 128)()
    check(lib.vc_comm_unique_id(ident), 'vc_comm_unique_id')
    assert any(ident), 'ncclGetUniqueId returned zeros'
    torch.cuda.set_device(0)
    h = _lib.vp()
    check(lib.vc_comm_init(0, 1, ident, C.byref(h)), 'vc_comm_init')
    assert lib.vc_comm_world(h) == 1
    t = torch.arange(1000, dtype=torch.float32, device='cuda:0') * 0.5
    want = t.clone()
    check(lib.vc_allreduce_f32(h, ptr(t), t.numel(), _lib.vp(torch.cuda.current_stream().cuda_stream)), 'vc_allreduce_f32')
    torch.cuda.synchronize()
    assert torch.equal(t, want)
    # the step's exchange + update through the communicator == without it
    rng = np.random.default_rng(5)
    A = rng.standard_normal((2, 24, 128)).astype(np.float32); B = rng.standard_normal((2, 24, 128)).astype(np.float32)
    res = []
    n0 = lib.vc_comm_collectives(h)
    for comm in (h, None):
        eng = CycleGANEngine(24, device='cuda:0', seed=9, data_parallel=False)
        eng.set_deterministic(True)
        eng.sync.comm = comm                      # as GradSync sets it up on the 'nccl' backend
        for _ in range(2):
            eng.train_step(A, B, 10.0, 5.0, 2e-4, 1e-4)
        torch.cuda.synchronize()
        res.append((eng.params.clone(), eng.m.clone(), eng.v.clone()))
        eng.sync.comm = None
    nb = lib.vc_step_num_g_buckets(eng.plan(128))
    assert lib.vc_comm_collectives(h) - n0 == 2 * (1 + 2 * nb), 'the native exchange did not issue its collectives'
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1])), 'update through a 1-rank communicator differs from the plain update'
    assert torch.isfinite(res[0][0]).all()
    # BASELINE.json's literal form: ONE ncclAllReduce over the flat gradient per step, same parameters bit for bit
    n1 = lib.vc_comm_collectives(h)
    eng = CycleGANEngine(24, device='cuda:0', seed=9, data_parallel=False)
    eng.set_deterministic(True); eng.set_exchange_mode('single')
    eng.sync.comm = h
    for _ in range(2):
        eng.train_step(A, B, 10.0, 5.0, 2e-4, 1e-4)
    torch.cuda.synchronize()
    eng.sync.comm = None
    assert lib.vc_comm_collectives(h) - n1 == 2, 'single mode: one collective per step'
    assert torch.equal(eng.params, res[0][0]) and torch.equal(eng.m, res[0][1]) and torch.equal(eng.v, res[0][2])
    check(lib.vc_comm_destroy(h), 'vc_comm_destroy')
    # GradSync on the nccl backend builds the native communicator (world 1: no collective is needed, none is created)
    ctx = mp.get_context('spawn')
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), str(tmp_path)))
    p.start(); p.join(300)
    assert p.exitcode == 0
    assert np.load(os.path.join(str(tmp_path), 'nccl.npy'))[0] == 1
