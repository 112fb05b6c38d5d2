#!/usr/bin/env python
"""bench.py -- MCEP frames/sec of the CycleGAN-VC train step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one call of the hot path behind model.CycleGAN.train (reference
model.py:110-125): forward of the 6 generator + 6 discriminator applications,
fused cycle/identity/LSGAN losses, every gradient, the gradient all-reduce (N>1)
and both Adam updates, on one synthetic [B, 24, 128] minibatch per domain that is
already resident in HBM.  Workload = BASELINE.json configs[1]: fp32, batch 8 per GPU.
Default: weak scaling, every rank runs batch 8, ranks all-reduce the 479 MB fp32 gradient (native RCCL communicator).
--global-batch G: strong scaling (BASELINE configs[3]: --global-batch 256), every rank runs G / N samples.

Prints ONE JSON line on rank 0 (see --help of the driver contract in the task):
  value          whole-job frames/s = n_gpus * B * n_frames * K / max-over-ranks time
  roofline       dominant kernel (implicit-GEMM conv fwd+dgrad on v_mfma_f32_32x32x2_f32):
                 algorithmic FLOPs / HIP-event time, against the 157.3 TFLOP/s fp32 MFMA peak
  cpu_baseline   the numpy oracle (a port of the reference arithmetic; TF1 itself cannot
                 run here) timed on this host's cores on a bounded sample (B=1)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: Peak BF16 MFMA, dense
MATH_TEXT = {'fp32': 'fp32', 'bf16': 'bf16 conv operands in HBM and LDS (packed 16-bit data path), fp32 accumulate, fp32 conv outputs / instance-norm statistics / Adam',
             'bf16x3': 'fp32 emulated as 3 bf16 MFMAs per product (hi/lo operand split)'}
HBM_PEAK_GBS = 8000.0


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return ', '.join('%s x%d' % (i.get('internal_api', '?'), i.get('num_threads', 0)) for i in threadpool_info()) or 'none'
    except Exception:
        return 'unknown'


def cpu_baseline(n_steps=10):
    """oracle/cyclegan_oracle.py on the host cores: B=1, 24x128 (BASELINE configs[0]: 10 steps), 1 warm-up + n timed."""
    import numpy as np
    from oracle import cyclegan_oracle as O
    P = O.init_params(seed=42)
    A, Bm = O.synthetic_batch(1, 24, 128)
    orc = O.CycleGANOracle(P)
    orc.train(A, Bm, 10.0, 5.0, 2e-4, 1e-4)
    ts = []
    for _ in range(n_steps):
        t = time.perf_counter()
        orc.train(A, Bm, 10.0, 5.0, 2e-4, 1e-4)
        ts.append(time.perf_counter() - t)
    med = float(np.median(ts))
    return dict(value=128.0 / med, unit='frames/s', cores=os.cpu_count(), kind='port',
                sample='numpy oracle (oracle/cyclegan_oracle.py), B=1 x 24x128, 1 warm-up + %d timed full '
                       'train steps, median %.2f s/step; the GEMMs run in numpy\'s BLAS (%s), the rest single-threaded; '
                       'TF1 reference not runnable (no tensorflow)' % (n_steps, med, _blas_threads()))


def _pmc_traffic(B, F, T, world, math):
    """HBM-side bytes per igemm launch from the committed PMC summary of the same configuration (None if not profiled)"""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r02_pmc_traffic.json')) as f:
            d = json.load(f)
        return d.get('%s_b%d_%dx%d' % (math, B, F, T)) if world == 1 else None
    except Exception:
        return None


def named_kernel_h1(lib, dev, B=256, reps=20, peak=FP32_MFMA_PEAK_TFLOPS, math='fp32'):
    """The north star's named kernel: the generator's gated conv1d block (module.py:161-163: h1 value + gate convolutions +
    GLU) forward at B=256, as ONE kernel (vc_gated_conv1d_glu_fwd).  SURVEY 8d: algorithmic bytes 20.29 MB (frames in,
    GLU output out, weights once), algorithmic FLOPs 6.04 G; exact fp32 caps the HBM fraction at ~6.6 %."""
    import numpy as np
    import torch
    from voice_converter_cyclegan_amd._lib import ptr, check, vp
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    x = torch.randn((1, B, 24, 128), device=dev)
    w = torch.randn((1, 15, 24, 256), device=dev) * 0.05
    b = torch.randn((1, 256), device=dev) * 0.1
    y = torch.empty((1, B, 128, 128), device=dev)
    s = vp(torch.cuda.current_stream().cuda_stream)
    check(lib.vc_set_math_mode({'fp32': 0, 'bf16': 1}[math]))
    wsp = torch.empty(lib.vc_gated_conv1d_glu_workspace_floats(1, 256), device=dev)
    call = lambda: check(lib.vc_gated_conv1d_glu_fwd(ptr(x), 0, ptr(w), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, 128, 256, 15, ptr(wsp), wsp.numel(), s))
    try:
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        CycleGANEngine.profiling(True)
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        ms, n, fl = C.c_double(0), C.c_longlong(0), C.c_double(0)
        lib.vc_prof_collect(2, C.byref(ms), C.byref(n), C.byref(fl))
        CycleGANEngine.profiling(False)
    finally:
        check(lib.vc_set_math_mode(0))
    t = ms.value * 1e-3 / max(n.value, 1)
    algo = B * (24 * 128 + 128 * 128) * 4 + (15 * 24 * 256 + 256) * 4         # 20.29 MB at B=256; also what the kernel moves
    flops = 2.0 * B * 128 * 256 * 15 * 24
    return {'kernel': 'gated_conv1d_glu_kernel: G.h1 value + gate conv1d + GLU forward in one kernel, B=%d, %s' % (B, math),
            'us': t * 1e6, 'algorithmic_bytes': algo,
            'hbm': {'achieved_GBs': algo / t / 1e9, 'peak_GBs': HBM_PEAK_GBS, 'frac': algo / t / 1e9 / HBM_PEAK_GBS},
            'mfma': {'achieved_TFLOPs': flops / t / 1e12, 'peak_TFLOPs': peak, 'frac': flops / t / 1e12 / peak},
            'note': ('exact-fp32 MFMA makes this kernel FLOP-bound: 38.4 us at the fp32 peak vs 2.5 us of HBM time' if math == 'fp32'
                     else 'bf16 MFMA operands (rounded once on their way into LDS), fp32 tensors in HBM, fp32 accumulate')}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='per-GPU minibatch (BASELINE configs[1])')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong scaling: total minibatch split over the ranks (BASELINE configs[3]: 256); overrides --batch')
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--features', type=int, default=24)
    ap.add_argument('--math', choices=['fp32', 'bf16', 'bf16x3'], default='fp32',
                    help='arithmetic of the conv kernels; fp32 (default) is the reference precision and the headline, '
                         'bf16 is BASELINE configs[2] (bf16 conv operands in HBM, fp32 accumulate/outputs/statistics)')
    ap.add_argument('--graph', action='store_true', help='replay the step as one hipGraph (vc_train_step_graph) instead of launching it')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--serial', action='store_true',
                    help='serialise the step\'s two independent chains on one stream for the WHOLE run (the condition the '
                         'roofline leg always uses; for rocprof summaries that must agree with it)')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)'
                  % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if rank == 0:
        ge.build()
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL
        dist.barrier()
    ge.build()
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    from voice_converter_cyclegan_amd._lib import lib

    dev = 'cuda:%d' % local_rank
    eng = CycleGANEngine(num_features=args.features, device=dev, seed=42)
    if args.serial:
        eng.set_overlap(False)
    if args.graph:
        eng.use_graph = True
    eng.set_math(args.math)
    peak = {'fp32': FP32_MFMA_PEAK_TFLOPS, 'bf16': BF16_MFMA_PEAK_TFLOPS, 'bf16x3': BF16_MFMA_PEAK_TFLOPS / 3}[args.math]
    if args.global_batch:
        if args.global_batch % world:
            sys.exit('--global-batch must be a multiple of the number of GPUs')
        args.batch = args.global_batch // world
    B, F, T = args.batch, args.features, args.frames
    rng = np.random.default_rng(1234 + rank)
    A = torch.from_numpy(rng.standard_normal((B, F, T)).astype(np.float32)).to(dev)
    Bm = torch.from_numpy(rng.standard_normal((B, F, T)).astype(np.float32)).to(dev)
    hp = (10.0, 5.0, 2e-4, 1e-4)        # train.py:17-26

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.train_step(A, Bm, *hp, sync=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step(A, Bm, *hp, sync=False)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = eng.losses.cpu().numpy()
    assert np.isfinite(losses).all(), losses

    out = None
    if rank == 0:
        frames = world * B * T * args.steps
        out = {
            'metric': 'MCEP frames/sec per train step (G+D fwd/bwd)',
            'value': frames / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'strong' if args.global_batch else 'weak', 'vs_baseline': None, 'dtype': {'fp32': 'f32'}.get(args.math, args.math), 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[%d]: full CycleGAN-VC train step (2G+2D fwd/bwd + losses + '
                                   'TF-Adam), %s, batch %d per GPU, %dx%d MCEP, random-init weights'
                                   % (4 if (F, T) == (36, 256) else 3 if args.global_batch else 1 if args.math == 'fp32' else 2,
                                      MATH_TEXT[args.math], B, F, T),
                       'global_batch': world * B, 'frames_per_step_per_domain': world * B * T,
                       'parallelism': 'dp%d' % world,
                       'launch': 'one hipGraph replay per step (vc_train_step_graph)' if (eng.use_graph and world == 1 and not args.serial)
                       else 'launch by launch',
                       'gradient_exchange': ('native RCCL communicator (vc_comm_init / vc_step_exchange_and_update), %d ranks'
                                             % lib.vc_comm_world(eng.sync.comm)) if eng.sync.comm is not None
                       else ('none (single GPU)' if world == 1 else 'torch.distributed'),
                       # the reference graph is 91.41 GFLOP per sample; the fused step legitimately skips the two
                       # D(fake) forwards of the D sub-step (D weights are unchanged by the G update, model.py:112-119) and the
                       # input gradients nobody consumes: what it EXECUTES is filled in below from the per-launch records
                       'reference_graph_gflop_per_step': 91.41 * world * B if (F, T) == (24, 128) else None},
        }

    # ---- roofline leg: the same steps again with HIP events around every implicit-GEMM launch
    if not args.no_roofline:
        barrier()
        # per-kernel timing: serialise the step's chains so a launch is never time-sharing the chip with a launch of the
        # other chain (the headline `value` above runs them on two streams)
        eng.set_overlap(False)
        for _ in range(2):
            eng.train_step(A, Bm, *hp, sync=False)
        torch.cuda.synchronize()
        CycleGANEngine.profiling(True)
        n_prof = 3
        for _ in range(n_prof):
            eng.train_step(A, Bm, *hp, sync=False)
        torch.cuda.synchronize()
        stats = {}
        knames = ((0, 'igemm_kernel (conv fwd + dgrad, every Cin > 1 layer)'), (1, 'wgrad_kernel'), (2, 'wsconv1d_kernel'))
        if args.math == 'bf16':       # 16-bit data path (csrc/vc_conv_bf16.hip); launches too small for its tiles stay on igemm / wgrad_kernel
            knames = ((0, 'conv_bf16_kernel (conv fwd + dgrad; + igemm_kernel for the few launches below its tile size)'),
                      (1, 'filter gradient: conv_bf16_kernel as a GEMM over the pixel axis on im2col / pixel-octet packed operands (+ wgrad_kernel for o1)'), (2, 'wsconv1d_kernel'))
        for kind, name in knames:
            ms, n, fl = C.c_double(0), C.c_longlong(0), C.c_double(0)
            lib.vc_prof_collect(kind, C.byref(ms), C.byref(n), C.byref(fl))
            stats[kind] = (name, ms.value, n.value, fl.value)
        CycleGANEngine.profiling(False)
        eng.set_overlap(not args.serial)
        if rank == 0:
            name, ms, n, fl = stats[0]
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            out['roofline'] = {
                'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                'frac': ach / peak,
                # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
                # (profiles/r01_pmc_hbm_traffic_b8.txt; FETCH_SIZE doubled per MI355X_MICROARCH.md, L3 hits included)
                'traffic': _pmc_traffic(B, F, T, world, args.math),
                'traffic_unit': 'bytes/launch (offline rocprofv3 PMC passes of this command, profiles/r02_pmc_traffic.json; '
                                'FETCH_SIZE doubled per MI355X_MICROARCH.md; upper bound: Infinity-Cache hits counted)',
                'kernel': name, 'launches_per_step': n // n_prof,
                'condition': 'step chains serialised on one stream (no co-running launches); see --serial',
                'avg_launch_us': 1e3 * ms / max(n, 1), 'algorithmic_gflop_per_step': fl / n_prof / 1e9,
                'kernel_ms_per_step': ms / n_prof,
            }
            name, ms, n, fl = stats[1]
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            out['roofline_wgrad'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak,
                                     'unit': 'TFLOP/s', 'frac': ach / peak, 'kernel': name,
                                     'launches_per_step': n // n_prof, 'kernel_ms_per_step': ms / n_prof}
            # FLOPs the step executes: every MFMA launch carries its algorithmic FLOPs; the Cin = 1 layer (D.h1, vector ALU)
            # is added analytically: forward over 2B samples per D, filter gradient over 2B, input gradient over B
            c1 = 2.0 * 9 * 256 * (F * (T // 2)) * 2 * (2 * B + 2 * B + B)      # 9 taps x 256 filters x F x T/2 pixels, 2 discriminators
            executed = (stats[0][3] + stats[1][3] + stats[2][3]) / n_prof + c1
            out['config']['executed_gflop_per_step'] = world * executed / 1e9
            out['step_flop_frac_of_fp32_peak'] = executed / (out['ms_per_step'] * 1e-3) / (FP32_MFMA_PEAK_TFLOPS * 1e12)

    # ---- the kernel the north_star names: gated-conv1d forward (G.h1: 24 -> 2x128, k=15) at B=256
    if rank == 0 and not args.no_roofline:
        out['named_kernel'] = named_kernel_h1(lib, dev, peak=FP32_MFMA_PEAK_TFLOPS, math='fp32')
        # the same kernel with bf16 MFMA operands (BASELINE configs[2] arithmetic): the HBM-roofline view the north star asks for
        r = named_kernel_h1(lib, dev, peak=BF16_MFMA_PEAK_TFLOPS, math='bf16')
        out['named_kernel']['bf16_mode'] = {'us': r['us'], 'hbm_frac': r['hbm']['frac'], 'achieved_TFLOPs': r['mfma']['achieved_TFLOPs']}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
