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


def _cpu_leg(kind, B, threads, budget_s):
    """one leg of cpu_baseline, run in its own process (see there): prints one JSON line"""
    import numpy as np
    from oracle import cyclegan_oracle as O
    if kind == 'torch':
        import torch
        from oracle.torch_ref import TorchTrainer
        torch.set_num_threads(threads)
        tt = TorchTrainer(O.init_params(seed=42))
        step = lambda A, Bm: tt.train(A, Bm, 10.0, 5.0, 2e-4, 1e-4)
        name = 'torch-CPU (oracle/torch_ref.py TorchTrainer: oneDNN / MKL convolutions + autograd, %d threads)' % torch.get_num_threads()
    else:
        orc = O.CycleGANOracle(O.init_params(seed=42))
        step = lambda A, Bm: orc.train(A, Bm, 10.0, 5.0, 2e-4, 1e-4)
        name = 'numpy oracle (oracle/cyclegan_oracle.py; GEMMs in numpy\'s BLAS: %s, the rest single-threaded)' % _blas_threads()
    A, Bm = O.synthetic_batch(B, 24, 128)
    t = time.perf_counter(); step(A, Bm); warm = time.perf_counter() - t          # warm-up (also sizes the sample)
    n = int(max(5, min(10, budget_s / max(warm, 1e-3))))                          # median of >= 5 timed steps
    ts = []
    for _ in range(n):
        t = time.perf_counter(); step(A, Bm); ts.append(time.perf_counter() - t)
    med = float(np.median(ts))
    print(json.dumps({'fps': B * 128.0 / med, 'threads': threads,
                      'text': '%s B=%d: 1 warm-up + %d timed full train steps, median %.3f s/step = %.1f frames/s' % (name, B, n, med, B * 128.0 / med)}))


def cpu_baseline(leg_timeout_s=150):
    """The reference arithmetic on the host cores (TF1 itself cannot run here), timed on a bounded sample of the
    workload: oracle/torch_ref.py (torch-CPU; SURVEY 8d's stand-in) at B=1 (BASELINE configs[0]) and at the benched B=8,
    with 32 threads and with all host threads (many-core hosts can lose more time synchronising 256 threads over
    convolutions this small than they gain), and the numpy oracle at B=1.  Every leg runs in its own process under a
    timeout (a leg that does not finish is reported as such): one warm-up step, then 5..10 timed steps (as many as fit the
    leg's budget: 25 s at B=8, 15 s at B=1), the MEDIAN is reported.  `value` is the fastest leg in frames/s.  The
    all-threads leg of earlier rounds is gone: with 256 threads a B=8 step did not finish in 75 s (oneDNN synchronising 256
    threads over convolutions this small); 32 threads is what the stated `cores` means."""
    import subprocess
    cores = os.cpu_count()
    legs, notes = [], []
    plan = [('torch', 1, min(cores, 32), 15.0), ('torch', 8, min(cores, 32), 25.0), ('numpy', 1, cores, 15.0)]
    for kind, B, threads, budget_s in plan:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-leg', '%s,%d,%d,%g' % (kind, B, threads, budget_s)],
                               capture_output=True, text=True, timeout=leg_timeout_s)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            legs.append((d['fps'], d['threads'], d['text']))
            print('[bench] cpu baseline: ' + d['text'], file=sys.stderr, flush=True)
        except Exception as e:
            notes.append('%s B=%d with %d threads: did not finish within %d s (%s)' % (kind, B, threads, leg_timeout_s, type(e).__name__))
            print('[bench] cpu baseline: ' + notes[-1], file=sys.stderr, flush=True)
    if not legs:
        return dict(value=None, unit='frames/s', cores=cores, kind='port', sample='; '.join(notes))
    best = max(legs)
    return dict(value=best[0], unit='frames/s', cores=best[1], host_cores=cores, kind='port',
                sample='24x128 MCEP, fp32; ' + '; '.join([l[2] for l in legs] + notes) + '; TF1 reference not runnable (no tensorflow)')


PMC_FILES = ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json')


def _pmc_traffic(B, F, T, world, math, role='igemm'):
    """HBM-side bytes per launch of one roofline entry's kernels (role 'igemm': convolutions + input gradients; 'wgrad': filter
    gradients) from the committed PMC summary of the same configuration (tools/profile_round.sh -> tools/summarize_pmc.py):
    (bytes per launch, launches per step the figure averages over, file) or (None, None, None) if not profiled."""
    if world != 1:
        return None, None, None
    key = '%s_b%d_%dx%d' % (math, B, F, T)
    for fn in PMC_FILES:
        try:
            with open(os.path.join(ROOT, 'profiles', fn)) as f:
                d = json.load(f).get(key)
        except Exception:
            continue
        if isinstance(d, dict) and role in d:
            return d[role]['bytes_per_launch'], d[role]['launches_per_step'], 'profiles/' + fn
        if isinstance(d, (int, float)) and role == 'igemm' and math != 'bf16':     # round-3 file: one figure per configuration
            return d, None, 'profiles/' + fn                                       # (its bf16 figure mixed both roles: not used)
    return None, None, None


def named_kernel_h1(lib, dev, B=256, reps=20, peak=FP32_MFMA_PEAK_TFLOPS, math='fp32', variant='packed', data='rand'):
    """The north star's named kernel: the generator's gated conv1d block (module.py:161-163: h1 value + gate convolutions +
    GLU) forward at B=256, as ONE kernel.  SURVEY 8d: algorithmic bytes 20.29 MB (frames in, GLU output out, weights once),
    algorithmic FLOPs 6.04 G; exact fp32 caps the HBM fraction at ~6.6 %.
      variant 'packed': vc_gated_conv1d_glu_fwd_packed (filters as the pre-packed fragment image, built outside the timed region by
                        vc_gated_conv1d_glu_prepack: once per weight update);  'plain': vc_gated_conv1d_glu_fwd (TF-layout filters,
                        converted by every workgroup);  'roof': vc_prof_glu_copy_roof -- the same grid performing the kernel's
                        memory operations only (the measured roof of a launch of this shape).
      two clocks: us = HIP events around every single call (launch + kernel + event overhead, the round 1-4 protocol);
                  us_back_to_back = N calls between two events / N (average launch duration in a stream of launches)."""
    import torch
    from voice_converter_cyclegan_amd._lib import ptr, check, vp
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    x = torch.randn((1, B, 24, 128), device=dev)
    w = torch.randn((1, 15, 24, 256), device=dev) * 0.05
    b = torch.randn((1, 256), device=dev) * 0.1
    if data == 'zero':
        x.zero_(); w.zero_(); b.zero_()
    y = torch.empty((1, B, 128, 128), device=dev)
    s = vp(torch.cuda.current_stream().cuda_stream)
    MODES = {'fp32': 0, 'bf16': 1, 'f16': 3}      # f16: IEEE-half operands (math mode 3, this operator only): the 16-bit kernel inside the north star's 1e-3
    check(lib.vc_set_math_mode(1 if variant == 'roof' else MODES[math]))
    img = torch.empty((lib.vc_gated_conv1d_glu_image_bytes(24, 15, 256),), dtype=torch.uint8, device=dev)
    check(lib.vc_gated_conv1d_glu_prepack(ptr(w), 0, ptr(img), 0, 1, 24, 15, 256, s))
    check(lib.vc_set_math_mode(MODES[math]))
    if variant == 'packed':
        call = lambda: check(lib.vc_gated_conv1d_glu_fwd_packed(ptr(x), 0, ptr(img), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, 128, 256, 15, s))
    elif variant == 'plain':
        call = lambda: check(lib.vc_gated_conv1d_glu_fwd(ptr(x), 0, ptr(w), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, 128, 256, 15, s))
    else:
        call = lambda: check(lib.vc_prof_glu_copy_roof(ptr(x), ptr(img), ptr(y), B, 128, 256, s))
    try:
        # sustained clocks first: the GPU has idled while the host built the tensors (see tools/bench_named.py)
        t_end = time.perf_counter() + 0.05
        while time.perf_counter() < t_end:
            for _ in range(50):
                call()
            torch.cuda.synchronize()
        CycleGANEngine.profiling(True)
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        ms, n, fl = C.c_double(0), C.c_longlong(0), C.c_double(0)
        lib.vc_prof_collect(2, C.byref(ms), C.byref(n), C.byref(fl))
        CycleGANEngine.profiling(False)
        nb = 200 if B <= 256 else 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(20):
            call()
        e0.record()
        for _ in range(nb):
            call()
        e1.record()
        torch.cuda.synchronize()
        t_b2b = e0.elapsed_time(e1) * 1e-3 / nb
    finally:
        check(lib.vc_set_math_mode(0))
    t = ms.value * 1e-3 / max(n.value, 1)
    err = None
    if math != 'fp32' and variant == 'packed' and data == 'rand':
        # what the 16-bit operands cost against the exact-fp32 kernel on the same inputs (that kernel is held to 1e-4 of the oracle by
        # tests/test_ops_gpu.py::test_gated_conv1d_glu_fused): max |diff| / max |ref| and relative L2.  The north star couples its roofline
        # target to "fp32 parity within 1e-3": a figure above that means only the exact-fp32 entry answers it.
        y_bf = y.clone()
        check(lib.vc_set_math_mode(0))
        check(lib.vc_gated_conv1d_glu_fwd(ptr(x), 0, ptr(w), 0, ptr(b), 0, ptr(y), 0, 1, B, 24, 128, 256, 15, s))
        torch.cuda.synchronize()
        err = {'max_rel': float((y_bf - y).abs().max() / y.abs().max()), 'rel_l2': float((y_bf - y).norm() / y.norm())}
        err['within_north_star_1e-3'] = bool(err['max_rel'] < 1e-3 and err['rel_l2'] < 1e-3)
    algo = B * (24 * 128 + 128 * 128) * 4 + (15 * 24 * 256 + 256) * 4         # 20.29 MB at B=256; also what the kernel moves
    flops = 2.0 * B * 128 * 256 * 15 * 24
    return {'kernel': 'gated_conv1d_glu_kernel: G.h1 value + gate conv1d + GLU forward in one kernel, B=%d, %s, %s' % (B, math, variant),
            'us': t * 1e6, 'us_back_to_back': t_b2b * 1e6, 'algorithmic_bytes': algo,
            'hbm': {'achieved_GBs': algo / t / 1e9, 'peak_GBs': HBM_PEAK_GBS, 'frac': algo / t / 1e9 / HBM_PEAK_GBS,
                    'frac_back_to_back': algo / t_b2b / 1e9 / HBM_PEAK_GBS},
            'mfma': {'achieved_TFLOPs': flops / t / 1e12, 'peak_TFLOPs': peak, 'frac': flops / t / 1e12 / peak,
                     'frac_back_to_back': flops / t_b2b / 1e12 / peak},
            'err_vs_fp32': err,
            'note': ('exact-fp32 MFMA makes this kernel FLOP-bound: 38.4 us at the fp32 peak vs 2.5 us of HBM time' if math == 'fp32'
                     else '%s MFMA operands (rounded once, RNE, on their way into LDS), fp32 tensors in HBM, fp32 accumulate' % math)}


def named_kernel_report(lib, dev):
    """BASELINE north_star: the gated-conv1d forward at B = 256 (and B = 4096: the asymptote), exact fp32 and bf16 operands, each against
    (a) the 8 TB/s / MFMA roofs the north star names and (b) the MEASURED copy roof: a streaming kernel on the same grid that performs
    the same reads and the same stores and nothing else."""
    nk = named_kernel_h1(lib, dev, peak=FP32_MFMA_PEAK_TFLOPS, math='fp32', variant='packed')
    nk['plain'] = {k: named_kernel_h1(lib, dev, peak=FP32_MFMA_PEAK_TFLOPS, math='fp32', variant='plain')[k] for k in ('us', 'us_back_to_back')}
    bf = {}
    for B, reps in ((256, 20), (4096, 10)):
        r = named_kernel_h1(lib, dev, B=B, reps=reps, peak=BF16_MFMA_PEAK_TFLOPS, math='bf16', variant='packed')
        pl = named_kernel_h1(lib, dev, B=B, reps=reps, peak=BF16_MFMA_PEAK_TFLOPS, math='bf16', variant='plain')
        rf = named_kernel_h1(lib, dev, B=B, reps=reps, peak=BF16_MFMA_PEAK_TFLOPS, math='bf16', variant='roof')
        z = named_kernel_h1(lib, dev, B=B, reps=reps, peak=BF16_MFMA_PEAK_TFLOPS, math='bf16', variant='packed', data='zero')
        bf[B] = {'us': r['us'], 'us_back_to_back': r['us_back_to_back'], 'hbm_frac': r['hbm']['frac'], 'hbm_frac_back_to_back': r['hbm']['frac_back_to_back'],
                 'achieved_TFLOPs': r['mfma']['achieved_TFLOPs'], 'algorithmic_bytes': r['algorithmic_bytes'], 'err_vs_fp32': r['err_vs_fp32'],
                 'plain_us': pl['us'], 'plain_us_back_to_back': pl['us_back_to_back'],
                 'copy_roof': {'us': rf['us'], 'us_back_to_back': rf['us_back_to_back'], 'GBs_back_to_back': rf['algorithmic_bytes'] / rf['us_back_to_back'] / 1e3,
                               'kernel_frac_of_copy_roof': rf['us'] / r['us'], 'kernel_frac_of_copy_roof_back_to_back': rf['us_back_to_back'] / r['us_back_to_back'],
                               'what': 'vc_prof_glu_copy_roof: same grid (256 workgroups x 512 threads), every workgroup reads its slab\'s 46 KB fragment image and its '
                                       'tiles\' frames (16-byte loads, eight tiles in flight) and writes its 32 x 128 output block per tile with the kernel\'s own dword '
                                       'stores (two full 128-byte lines per instruction); no arithmetic, no LDS'},
                 'zero_operands_us_back_to_back': z['us_back_to_back']}
    nk['bf16_mode'] = dict(bf[256]); nk['bf16_mode']['B4096'] = bf[4096]
    # IEEE-half operands: the same kernel with another conversion and another MFMA -- the 16-bit form whose error against the exact-fp32 kernel
    # is INSIDE the 1e-3 the north star couples its HBM target to (bf16's 2.5e-3 is not); its copy roof is the bf16 entry's (same memory operations)
    hf = {}
    for B, reps in ((256, 20), (4096, 10)):
        r = named_kernel_h1(lib, dev, B=B, reps=reps, peak=BF16_MFMA_PEAK_TFLOPS, math='f16', variant='packed')
        hf[B] = {'us': r['us'], 'us_back_to_back': r['us_back_to_back'], 'hbm_frac': r['hbm']['frac'], 'hbm_frac_back_to_back': r['hbm']['frac_back_to_back'],
                 'achieved_TFLOPs': r['mfma']['achieved_TFLOPs'], 'algorithmic_bytes': r['algorithmic_bytes'], 'err_vs_fp32': r['err_vs_fp32'],
                 'kernel_frac_of_copy_roof': bf[B]['copy_roof']['us'] / r['us'],
                 'kernel_frac_of_copy_roof_back_to_back': bf[B]['copy_roof']['us_back_to_back'] / r['us_back_to_back']}
    nk['f16_mode'] = dict(hf[256]); nk['f16_mode']['B4096'] = hf[4096]
    nk['f16_mode']['what'] = ('math mode 3: frames and filters rounded to IEEE half (10-bit mantissa, RNE, v_cvt_pk_f16_f32), v_mfma_f32_32x32x16_f16, fp32 accumulate, '
                              'fp32 tensors in HBM; z-scored MCEP frames and glorot filters lie well inside half\'s range.  This operator only')
    nk['timed'] = ('one vc_gated_conv1d_glu_fwd_packed call = one kernel (plain_*: vc_gated_conv1d_glu_fwd, filters converted in the kernel); us = HIP events '
                   'around every call on the call\'s stream, us_back_to_back = 200 (40) calls between two events / N; each after 50 ms of the same call '
                   '(sustained clocks).  zero_operands_*: the same launch on all-zero operands -- the gap to the headline is the chip\'s data-dependent '
                   'clock management (in-kernel probe: 1.7-2.1 GHz on random bf16 operands, 2.4 GHz fp32), not the kernel')
    return nk


def time_steps(eng, A, Bm, hp, steps, warmup, barrier):
    for _ in range(warmup):
        eng.train_step(A, Bm, *hp, sync=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(A, Bm, *hp, sync=False)
    barrier()
    return time.perf_counter() - t0


def roofline_leg(eng, lib, A, Bm, hp, math, peak, world, serial, n_prof=3):
    """the same steps again with HIP events around every MFMA launch (vc_prof_*), the step's chains serialised on one stream so
    a launch never time-shares the chip with a launch of another chain (the headline `value` runs them on five streams)"""
    import torch
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    B, F, T = A.shape
    eng.set_overlap(False)
    for _ in range(2):
        eng.train_step(A, Bm, *hp, sync=False)
    torch.cuda.synchronize()
    CycleGANEngine.profiling(True)
    lib.vc_testing_wino_saved_flops(0, 1); lib.vc_testing_wino_saved_flops(1, 1)
    for _ in range(n_prof):
        eng.train_step(A, Bm, *hp, sync=False)
    torch.cuda.synchronize()
    saved = [lib.vc_testing_wino_saved_flops(0, 1), lib.vc_testing_wino_saved_flops(1, 1)]    # fp32: work the Winograd route did not execute
    stats = {}
    knames = ((0, 'igemm_kernel (conv fwd + dgrad, every Cin > 1 layer)'), (1, 'wgrad_kernel'), (2, 'wsconv1d_kernel'))
    if math == 'bf16':       # 16-bit data path (csrc/vc_conv_bf16.hip); launches too small for its tiles stay on igemm / wgrad_kernel
        knames = ((0, 'conv_bf16_kernel (conv fwd + dgrad; + igemm_kernel for the few launches below its tile size)'),
                  (1, 'filter gradient: conv_bf16_kernel as a GEMM over the pixel axis on im2col / pixel-octet packed operands (+ wgrad_kernel for o1)'), (2, 'wsconv1d_kernel'))
    for kind, name in knames:
        ms, n, fl = C.c_double(0), C.c_longlong(0), C.c_double(0)
        lib.vc_prof_collect(kind, C.byref(ms), C.byref(n), C.byref(fl))
        stats[kind] = (name, ms.value, n.value, fl.value)
    CycleGANEngine.profiling(False)
    eng.set_overlap(not serial)
    # `achieved` / `frac` price a launch's time in ALGORITHMIC FLOPs -- SURVEY 8d's per-unit figure (the direct-form convolution count,
    # 91.41 GFLOP per sample and step) x the units the launches process -- as the bench contract defines the roofline entry.  Where a layer
    # runs in a Winograd transform domain (fp32: residual and up-sampling blocks, D.d3) the MFMA pipe EXECUTES fewer multiply-adds than that
    # for the same result: `executed` reports those (what the MFMA roof physically bounds); without such layers (bf16, batch 1) the two agree.
    name, ms, n, fl = stats[0]
    exe = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    ach = (fl + saved[0]) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    tr, tr_n, tr_file = _pmc_traffic(B, F, T, world, math, 'igemm')
    out = {'roofline': {
        'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
        'flops': 'algorithmic = direct-form convolution FLOPs of these launches (SURVEY 8d), %.1f GFLOP per step' % ((fl + saved[0]) / n_prof / 1e9),
        'executed': {'achieved': exe, 'frac': exe / peak, 'gflop_per_step': fl / n_prof / 1e9,
                     'what': 'multiply-adds the MFMA pipe executed in the same time (Winograd-domain layers execute 0.4-0.67 of the direct form)'},
        # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        # (FETCH_SIZE doubled per MI355X_MICROARCH.md, Infinity-Cache hits included), over THIS entry's launches
        'traffic': tr,
        'traffic_unit': 'bytes/launch, mean over the %s launches per step of this entry\'s kernels (offline rocprofv3 PMC passes of this '
                        'command, %s; FETCH_SIZE doubled per MI355X_MICROARCH.md; upper bound: Infinity-Cache hits counted)'
                        % ('%g' % tr_n if tr_n else '?', tr_file),
        'kernel': name, 'launches_per_step': n // n_prof,
        'condition': 'step chains serialised on one stream (no co-running launches); see --serial',
        'avg_launch_us': 1e3 * ms / max(n, 1), 'algorithmic_gflop_per_step': (fl + saved[0]) / n_prof / 1e9,
        'kernel_ms_per_step': ms / n_prof}}
    if saved[0] > 0 or saved[1] > 0:
        # The residual / up-sampling blocks' and D.d3's convolutions run in a Winograd transform domain (csrc/vc_conv_wino.hip: F(4,3), F(4,5),
        # F(3,6) along H): 0.4-0.67 of the direct form's multiplies for the same result
        out['roofline']['winograd'] = {
            'what': 'forward + input gradient of the residual blocks (F(4,3)), up-sampling blocks (F(4,5)) and D.d3 (F(3,6) along H) as transform-domain GEMMs (grouped launches of igemm_kernel); direct form = what the round-5 kernels executed for the same layers',
            'direct_form_gflop_per_step': (fl + saved[0]) / n_prof / 1e9,
            'direct_equivalent_tflops': (fl + saved[0]) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
            'direct_equivalent_frac': (fl + saved[0]) / (ms * 1e-3) / 1e12 / peak if ms > 0 else 0.0}
    name, ms, n, fl = stats[1]
    exe = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    ach = (fl + saved[1]) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    out['roofline_wgrad'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'kernel': name,
                             'executed': {'achieved': exe, 'frac': exe / peak, 'gflop_per_step': fl / n_prof / 1e9},
                             'launches_per_step': n // n_prof, 'kernel_ms_per_step': ms / n_prof,
                             'traffic': _pmc_traffic(B, F, T, world, math, 'wgrad')[0]}
    if saved[1] > 0:
        out['roofline_wgrad']['kernel'] = 'wgrad_kernel + wino_wgrad_kernel (residual and up-sampling blocks: Winograd-domain filter gradient, taps folded in registers)'
        out['roofline_wgrad']['winograd'] = {'direct_form_gflop_per_step': (fl + saved[1]) / n_prof / 1e9,
                                             'direct_equivalent_tflops': (fl + saved[1]) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                                             'direct_equivalent_frac': (fl + saved[1]) / (ms * 1e-3) / 1e12 / peak if ms > 0 else 0.0}
    # FLOPs the step executes: every MFMA launch carries its algorithmic FLOPs; the Cin = 1 layer (D.h1, vector ALU)
    # is added analytically: forward over 2B samples per D, filter gradient over 2B, input gradient over B
    c1 = 2.0 * 9 * 256 * (F * (T // 2)) * 2 * (2 * B + 2 * B + B)      # 9 taps x 256 filters x F x T/2 pixels, 2 discriminators
    out['executed_flop_per_step'] = (stats[0][3] + stats[1][3] + stats[2][3]) / n_prof + c1
    return out


def dp_selftest(eng, lib, A, Bm, hp, steps, warmup):
    """What one GPU can say about the data-parallel step: the step time with the gradient exchange of
    vc_step_exchange_and_update actually on the update stream, against the plain single-GPU step.
      rccl_1rank : a real RCCL communicator of ONE rank (RCCL refuses two ranks on one device): the collectives are issued
                   (vc_comm_collectives counts them) -- what RCCL launches for a 1-rank in-place SUM is RCCL's business;
      standin    : the exchange replaced by stand-in traffic on the same stream behind the same events: every gradient
                   segment is copied out and back (2 reads + 2 writes of the segment = the HBM traffic of a ring
                   all-reduce's reduce-scatter + all-gather on one rank), so the 479 MB exchange competes with the generator
                   backward for HBM and CUs the way the real one does.  NOT a measurement of xGMI -- the link time
                   (2 (N-1)/N x 479 MB / per-link bandwidth) comes on top where it is not hidden.
    Reported: ms/step of each leg, the exposed (not overlapped) exchange + update time from events, the segment sizes."""
    import ctypes as C
    import torch
    from voice_converter_cyclegan_amd import _lib
    from voice_converter_cyclegan_amd._lib import check
    out = {}
    p = eng.plan(A.shape[2])
    segs = eng.grad_segments(p)
    out['segments_MB'] = [round(sum(hi - lo for lo, hi in r) * 4 / 1e6, 1) for _, _, r in segs]
    eng.measure_exposed = True

    def leg():
        dt = time_steps(eng, A, Bm, hp, steps, warmup, torch.cuda.synchronize)
        return 1e3 * dt / steps, eng.exposed_update_ms()
    ms, ex = leg()
    out['plain'] = {'ms_per_step': ms, 'exposed_update_ms': ex}
    ident = (C.c_ubyte * 128)()
    h = _lib.vp()
    try:
        check(lib.vc_comm_unique_id(ident), 'vc_comm_unique_id')
        check(lib.vc_comm_init(0, 1, ident, C.byref(h)), 'vc_comm_init')
        n0 = lib.vc_comm_collectives(h)
        eng.sync.comm = h
        ms, ex = leg()
        out['rccl_1rank'] = {'ms_per_step': ms, 'exposed_update_ms': ex,
                             'collectives_per_step': (lib.vc_comm_collectives(h) - n0) / float(steps + warmup)}
    except Exception as e:      # no RCCL on this box: say so, keep the stand-in leg
        out['rccl_1rank'] = {'error': str(e)}
    finally:
        eng.sync.comm = None
        if h:
            lib.vc_comm_destroy(h)
    scratch = torch.empty(max(hi - lo for _, _, r in segs for lo, hi in r), device=A.device)

    def standin(g):
        scratch[:g.numel()].copy_(g)
        g.copy_(scratch[:g.numel()])
    eng.exchange_standin = standin
    ms, ex = leg()
    eng.exchange_standin = None
    eng.measure_exposed = False
    out['standin'] = {'ms_per_step': ms, 'exposed_update_ms': ex}
    out['slowdown_standin_vs_plain'] = out['standin']['ms_per_step'] / out['plain']['ms_per_step']
    out['note'] = ('one GPU: rccl_1rank issues the real collectives on a 1-rank communicator; standin moves each gradient segment '
                   'out and back on the update stream (ring all-reduce HBM traffic without the links)')
    return out


def workload_text(math, B, F, T, global_batch):
    idx = 4 if (F, T) == (36, 256) else 3 if global_batch else 2 if math != 'fp32' else 0 if B == 1 else 1
    return ('BASELINE configs[%d]: full CycleGAN-VC train step (2G+2D fwd/bwd + losses + TF-Adam), %s, batch %d per GPU, '
            '%dx%d MCEP, random-init weights' % (idx, MATH_TEXT[math], B, F, T))


def inference_report(lib, dev):
    """the conversion path (CycleGAN.test, model.py:128-137; convert.py:46 feeds one whole utterance [1, 24, T] per call): generator
    forward through an inference plan, weights and frames resident in HBM, HIP events around `reps` calls on the call's stream.  T = 2208
    frames = 11 s of speech at convert.py's 5 ms frame period (a multiple of 4 as preprocess.py pads it); B = 8 the same as a batch."""
    import torch
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    from voice_converter_cyclegan_amd._lib import ptr, check, vp
    out = []
    for math in ('fp32', 'bf16'):
        eng = CycleGANEngine(num_features=24, device=dev, seed=42, data_parallel=False)
        eng.set_math(math)
        for B, T in ((1, 2208), (8, 2208)):
            p = eng._infer_plans.get(T)
            x = torch.randn((B, 24, T), device=dev); y = torch.empty_like(x)
            save = torch.empty(lib.vc_generator_save_floats(p, B), device=dev)
            s = vp(torch.cuda.current_stream().cuda_stream)
            run = lambda: check(lib.vc_generator_forward(p, ptr(eng.params), 0, ptr(x), 0, ptr(y), 0, ptr(save), 1, B, s))
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            reps = 30
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            assert bool(torch.isfinite(y).all())
            gf = 2.656 * B * T / 128        # GFLOP per generator application, linear in T
            out.append({'workload': 'generator forward (vc_generator_forward, inference plan), %s, B=%d, 24x%d' % (math, B, T), 'dtype': {'fp32': 'f32'}.get(math, math),
                        'ms': ms, 'frames_per_s': B * T / ms * 1e3, 'TFLOPs': gf / ms,
                        'x_real_time': B * T * 5.0 / ms})        # convert.py: frame_period = 5.0 ms
        eng.close()
        del eng
        torch.cuda.empty_cache()
    return out


def extra_config_own_process(B, steps, warmup, note):
    """one more BASELINE configuration measured by THIS script in a process of its own (`bench.py --batch B ...`), the parent idle meanwhile.
    For batch 1 only: the step is ~210 launches in 2.8 ms on five streams, and inside the process that has already run the headline, the named
    kernel and the data-parallel legs (dozens of streams created and destroyed, a larger heap) the same engine measures 0.2 ms slower than in a
    fresh process (2.97-3.07 vs 2.75-2.80 ms, round 6: neither the data-parallel leg nor the named-kernel leg alone accounts for it) -- a training
    run at the reference's batch size IS a fresh process.  The larger configurations measure the same either way and stay in-process."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--batch', str(B), '--steps', str(steps), '--warmup', str(warmup),
           '--no-cpu-baseline', '--no-extra', '--no-named']
    print('[bench] extra configuration in its own process: %s' % ' '.join(cmd[2:]), file=sys.stderr, flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError('bench.py --batch %d in its own process failed (rc=%d): %s' % (B, r.returncode, r.stderr[-2000:]))
    d = json.loads(r.stdout.strip().splitlines()[-1])
    out = {'workload': d['config']['workload'] + ' -- ' + note, 'dtype': d['dtype'], 'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'],
           'steps': d['steps'], 'warmup': d['warmup'], 'process': 'own process (python bench.py --batch %d --steps %d --warmup %d)' % (B, steps, warmup),
           'executed_gflop_per_step': d['config'].get('executed_gflop_per_step'), 'step_flop_frac_of_fp32_peak': d.get('step_flop_frac_of_fp32_peak')}
    for k in ('roofline', 'roofline_wgrad'):
        if k in d:
            out[k] = d[k]
    return out


def _config0(lib, dev):
    """BASELINE configs[0] in a process of its own; if that process cannot be started or fails (the measurement must not cost the bench its one JSON
    line), the same configuration measured in this process, and the entry says which it was."""
    note = 'configs[0]: batch 1, the reference\'s own mini_batch_size (train.py:16); 40 warm-up steps = 0.1 s: a latency-bound step reads the clock ramp after the engine set-up'
    try:
        return extra_config_own_process(1, 60, 40, note)
    except Exception as e:      # noqa: BLE001 -- any failure of the child process
        print('[bench] configs[0] in its own process failed (%s): measuring it in this process' % (str(e)[:300],), file=sys.stderr, flush=True)
        out = extra_config(lib, dev, 'fp32', 1, 24, 128, 60, 40, note)
        out['process'] = 'this process (the own-process run failed)'
        return out


def extra_config(lib, dev, math, B, F, T, steps, warmup, note, constant_operands=False):
    """one more BASELINE configuration on this GPU, measured like the headline: wall time of `steps` steps + its own roofline leg.
    constant_operands: every parameter and every input 0 (activations and weight gradients stay 0, only biases move) -- the
    same launches on zero-entropy data (a DIAGNOSTIC of the
    chip's data-dependent power management, profiles/HISTORY.md 4.1 / profiles/r03_data_dependent_power.txt; never a performance figure)"""
    import numpy as np
    import torch
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    print('[bench] extra configuration: %s B=%d %dx%d' % (math, B, F, T), file=sys.stderr, flush=True)
    eng = CycleGANEngine(num_features=F, device=dev, seed=42, data_parallel=False)
    eng.set_math(math)
    rng = np.random.default_rng(99)
    A = torch.from_numpy(rng.standard_normal((B, F, T)).astype(np.float32)).to(dev)
    Bm = torch.from_numpy(rng.standard_normal((B, F, T)).astype(np.float32)).to(dev)
    hp = (10.0, 5.0, 2e-4, 1e-4)
    if constant_operands:
        eng.params.zero_(); A.zero_(); Bm.zero_()
    dt = time_steps(eng, A, Bm, hp, steps, warmup, torch.cuda.synchronize)
    assert np.isfinite(eng.losses.cpu().numpy()).all()
    peak = {'fp32': FP32_MFMA_PEAK_TFLOPS, 'bf16': BF16_MFMA_PEAK_TFLOPS}[math]
    r = roofline_leg(eng, lib, A, Bm, hp, math, peak, 1, False)
    executed = r.pop('executed_flop_per_step')
    out = {'workload': workload_text(math, B, F, T, 0) + ' -- ' + note, 'dtype': {'fp32': 'f32'}.get(math, math),
           'value': B * T * steps / dt, 'unit': 'frames/s', 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': warmup,
           'executed_gflop_per_step': executed / 1e9,
           'step_flop_frac_of_fp32_peak': executed / (dt / steps) / (FP32_MFMA_PEAK_TFLOPS * 1e12)}
    out.update(r)
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--exchange', choices=['bucketed', 'single', 'both'], default='both',
                    help='N > 1: gradient exchange of the headline (bucketed: 5 segments under the backward, the default; single: ONE all-reduce '
                         'after it, BASELINE.json\'s literal form); both = headline bucketed + the same steps again in single mode under config.exchange_modes')
    ap.add_argument('--batch', type=int, default=8, help='per-GPU minibatch (BASELINE configs[1])')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong scaling: total minibatch split over the ranks (BASELINE configs[3]: 256); overrides --batch')
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--features', type=int, default=24)
    ap.add_argument('--math', choices=['fp32', 'bf16', 'bf16x3'], default='fp32',
                    help='arithmetic of the conv kernels; fp32 (default) is the reference precision and the headline, '
                         'bf16 is BASELINE configs[2] (bf16 conv operands in HBM, fp32 accumulate/outputs/statistics)')
    ap.add_argument('--graph', action='store_true', help='replay the step as one hipGraph (vc_train_step_graph) instead of launching it')
    ap.add_argument('--deterministic', action='store_true', help='vc_plan_set_deterministic: fixed summation order (bitwise reproducible steps)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-leg', default='', help=argparse.SUPPRESS)      # internal: one leg of cpu_baseline in its own process
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the other BASELINE configurations (bf16 B=64, fp32 B=32 = configs[3]\'s share of an 8-GPU node, 36x256 '
                         'B=64) that a default single-GPU run measures after the headline and reports under extra.configs')
    ap.add_argument('--no-named', action='store_true', help=argparse.SUPPRESS)      # internal: roofline leg without the named-kernel report
    ap.add_argument('--dp-selftest', action='store_true',
                    help='single GPU: time the step with the gradient exchange really on the update stream (1-rank RCCL communicator; '
                         'stand-in traffic) and report it under extra.dp_overlap; part of the default single-GPU run')
    ap.add_argument('--serial', action='store_true',
                    help='serialise the step\'s two independent chains on one stream for the WHOLE run (the condition the '
                         'roofline leg always uses; for rocprof summaries that must agree with it)')
    ap.add_argument('--print-key', action='store_true', help=argparse.SUPPRESS)   # internal: the configuration's key in profiles/*_pmc_traffic.json
    ap.add_argument('--launch-selftest', action='store_true',
                    help='exercise ONLY the launch path: spawn the ranks if needed, build the process group (RCCL on GPUs, gloo without), '
                         'all-reduce one number, print {"launch_selftest": ...} and exit (tests/test_dp_gloo.py drives it on CPU)')
    args = ap.parse_args()
    if args.print_key:
        print('%s_b%d_%dx%d' % (args.math, args.batch, args.features, args.frames))
        return 0
    if args.cpu_leg:
        kind, B_, th, bud = args.cpu_leg.split(',')
        return _cpu_leg(kind, int(B_), int(th), float(bud))
    # The contract is ONE JSON line on stdout.  Libraries underneath print there too (RCCL's version banner at communicator
    # creation, on every rank): from here on file descriptor 1 IS stderr, and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    default_workload = (args.batch, args.global_batch, args.frames, args.features, args.math, args.graph, args.serial, args.deterministic) == \
                       (8, 0, 128, 24, 'fp32', False, False, False)

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: become the launcher.  One rank per GPU through torch.distributed.run
        # on this node (the command the driver itself documents); the ranks re-enter this file with RANK / WORLD_SIZE set, rank 0
        # prints the JSON line on the inherited stdout.  (this used to exit 2 at the argument check.)
        import socket
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        os.dup2(json_fd, 1)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print('[bench] --gpus %d without a launcher: ' % args.gpus + ' '.join(cmd), file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but the launcher set WORLD_SIZE=%d: pass the same number to both' % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    have_gpu = torch.cuda.is_available()
    if have_gpu and torch.cuda.device_count() < world and not args.launch_selftest:
        # device selection, not argument parsing, is where a too-large N fails -- and it says so
        print('bench.py rank %d: %d ranks on this node but only %d GPU(s) visible (LOCAL_RANK %d has no device): one process per GPU is the '
              'only layout this step supports (DESIGN.md 9)' % (rank, world, torch.cuda.device_count(), local_rank), file=sys.stderr, flush=True)
        sys.exit(3)
    if not have_gpu and not args.launch_selftest:
        print('bench.py: no GPU visible (the hot path has no CPU fallback)', file=sys.stderr, flush=True)
        sys.exit(3)
    if rank == 0:
        ge.build()
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if have_gpu and torch.cuda.device_count() >= world:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL
        else:
            dist.init_process_group('gloo')                                                # --launch-selftest without GPUs
        dist.barrier()
    if args.launch_selftest:
        backend = dist.get_backend() if world > 1 else 'none'
        t = torch.ones(1, device=('cuda:%d' % local_rank) if backend == 'nccl' else 'cpu') * (rank + 1)
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            os.write(json_fd, (json.dumps({'launch_selftest': True, 'n_gpus': world, 'backend': backend, 'sum_of_rank_plus_1': float(t.item()),
                                           'launched_by': 'torch.distributed.run' if 'TORCHELASTIC_RUN_ID' in os.environ or world > 1 else 'python'}) + '\n').encode())
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0
    ge.build()
    from voice_converter_cyclegan_amd.engine import CycleGANEngine
    from voice_converter_cyclegan_amd._lib import lib

    assert lib.vc_thread_is_product_run() == 1, 'a test / tuning switch is active: the bench runs production kernel configurations only'
    dev = 'cuda:%d' % local_rank
    eng = CycleGANEngine(num_features=args.features, device=dev, seed=42)
    if args.serial:
        eng.set_overlap(False)
    if args.graph:
        eng.use_graph = True
    eng.set_math(args.math)
    eng.set_deterministic(args.deterministic)
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

    eng.measure_exposed = world > 1              # N > 1: how much of exchange + update is NOT hidden under the backward
    eng.set_exchange_mode('single' if args.exchange == 'single' else 'bucketed')
    dt = time_steps(eng, A, Bm, hp, args.steps, args.warmup, barrier)
    exposed_ms = eng.exposed_update_ms() if world > 1 else None
    exchange_modes = None
    if world > 1:                                # the A/B the north star asks for: the same steps with the other exchange form
        headline_mode = 'single' if args.exchange == 'single' else 'bucketed'
        exchange_modes = {headline_mode: {'ms_per_step': 1e3 * dt / args.steps, 'exposed_update_ms': exposed_ms, 'headline': True}}
        if args.exchange == 'both':
            eng.set_exchange_mode('single')
            dt2 = time_steps(eng, A, Bm, hp, args.steps, args.warmup, barrier)
            ex2 = eng.exposed_update_ms()
            t2 = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            exchange_modes['single'] = {'ms_per_step': 1e3 * float(t2.item()) / args.steps, 'exposed_update_ms': ex2, 'headline': False}
            eng.set_exchange_mode('bucketed')
    eng.measure_exposed = False
    if world > 1:                                # every rank says what it runs with, so a SCALE line explains itself
        p_ = eng.plan(T)
        print('[bench] rank %d/%d: %.3f ms/step, exposed exchange+update %.3f ms, communicator %s, gradient segments (MB) %s'
              % (rank, world, 1e3 * dt / args.steps, exposed_ms or -1.0,
                 ('native RCCL, %d ranks' % lib.vc_comm_world(eng.sync.comm)) if eng.sync.comm is not None else 'torch.distributed',
                 [round(sum(hi - lo for lo, hi in r) * 4 / 1e6, 1) for _, _, r in eng.grad_segments(p_)]), file=sys.stderr, flush=True)
    if rank == 0:
        print('[bench] headline: %.3f ms/step' % (1e3 * dt / args.steps), file=sys.stderr, flush=True)
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
            'config': {'workload': workload_text(args.math, B, F, T, args.global_batch),
                       'global_batch': world * B, 'frames_per_step_per_domain': world * B * T,
                       'parallelism': 'dp%d' % world,
                       'rccl_ranks': int(lib.vc_comm_world(eng.sync.comm)) if eng.sync.comm is not None else (1 if world == 1 else None),
                       'build_id': lib.vc_build_id().decode(),
                       'launch': 'one hipGraph replay per step (vc_train_step_graph)' if (eng.use_graph and world == 1 and not args.serial)
                       else 'launch by launch',
                       'deterministic': bool(args.deterministic),
                       'gradient_exchange': ('native RCCL communicator (vc_comm_init / vc_step_exchange_and_update), %d ranks'
                                             % lib.vc_comm_world(eng.sync.comm)) if eng.sync.comm is not None
                       else ('none (single GPU)' if world == 1 else 'torch.distributed'),
                       # the reference graph is 91.41 GFLOP per sample; the fused step legitimately skips the two
                       # D(fake) forwards of the D sub-step (D weights are unchanged by the G update, model.py:112-119) and the
                       # input gradients nobody consumes: what it EXECUTES is filled in below from the per-launch records
                       'reference_graph_gflop_per_step': 91.41 * world * B if (F, T) == (24, 128) else None},
        }
        if world > 1:
            out['config']['exposed_exchange_update_ms_rank0'] = exposed_ms
            out['config']['gradient_bytes_per_step'] = 4 * int(eng.n_flat)
            exchange_modes[next(k for k, v in exchange_modes.items() if v['headline'])]['ms_per_step'] = out['ms_per_step']     # max over ranks
            out['config']['exchange_modes'] = exchange_modes      # ms_per_step (max over ranks) and rank 0's exposed exchange + update per mode

    # ---- roofline leg: the same steps again with HIP events around every implicit-GEMM launch
    if not args.no_roofline:
        barrier()
        r = roofline_leg(eng, lib, A, Bm, hp, args.math, peak, world, args.serial)
        if rank == 0:
            executed = r.pop('executed_flop_per_step')
            out.update(r)
            out['config']['executed_gflop_per_step'] = world * executed / 1e9
            out['step_flop_frac_of_fp32_peak'] = executed / (out['ms_per_step'] * 1e-3) / (FP32_MFMA_PEAK_TFLOPS * 1e12)
            if out['config'].get('reference_graph_gflop_per_step'):
                # SURVEY 8d's whole-step ceiling: 91.41 GFLOP per sample (direct form) at the fp32 peak = 220 k frames/s per GPU at 24 x 128
                out['step_algorithmic_frac_of_fp32_peak'] = (out['config']['reference_graph_gflop_per_step'] * 1e9 / world / (out['ms_per_step'] * 1e-3)
                                                             / (FP32_MFMA_PEAK_TFLOPS * 1e12))

    # ---- the kernel the north_star names: gated-conv1d forward (G.h1: 24 -> 2x128, k=15) at B=256 (and B=4096: the asymptote)
    if rank == 0 and not args.no_roofline and not args.no_named:
        out['named_kernel'] = named_kernel_report(lib, dev)

    dp = None
    if rank == 0 and world == 1 and (args.dp_selftest or (default_workload and not args.no_extra and not args.no_roofline)):
        print('[bench] data-parallel self-test on one GPU', file=sys.stderr, flush=True)
        dp = dp_selftest(eng, lib, A, Bm, hp, args.steps, args.warmup)
        if args.dp_selftest:
            out.setdefault('extra', {})['dp_overlap'] = dp
    # ---- the other BASELINE configurations this GPU can hold, each with its own roofline (default single-GPU run only)
    if rank == 0 and world == 1 and default_workload and not args.no_extra and not args.no_roofline:
        eng.close()
        torch.cuda.empty_cache()
        out['extra'] = {'configs': [
            _config0(lib, dev),
            extra_config(lib, dev, 'bf16', 64, 24, 128, 10, 3, 'configs[2]: batch 64, bf16'),
            extra_config(lib, dev, 'fp32', 32, 24, 128, 10, 3, 'configs[3]: the per-GPU share (32) of global batch 256 on an 8-GPU node, measured on one GPU without the exchange'),
            extra_config(lib, dev, 'fp32', 256, 24, 128, 3, 1, 'configs[3] on ONE GPU: global batch 256, the N = 1 end of the strong-scaling curve (`--gpus 1 --global-batch 256`; golden-backed: tests/golden/golden_f32_b256.npz)'),
            extra_config(lib, dev, 'fp32', 64, 36, 256, 5, 2, 'configs[4]: the per-GPU share (64) of global batch 512 at 36x256, measured on one GPU without the exchange'),
        ]}
        out['extra']['inference'] = inference_report(lib, dev)
        # diagnostic, not a result: the headline configuration's launches on constant operands (data-dependent power management)
        out['extra']['dp_overlap'] = dp
        out['extra']['constant_operands_diagnostic'] = extra_config(
            lib, dev, 'fp32', 8, 24, 128, 20, 5, 'DIAGNOSTIC: configs[1] with every parameter and every input 0 (same launches, '
            'zero-entropy operands); compare its roofline.frac with the headline\'s -- the difference is the chip\'s power management, not the kernels',
            constant_operands=True)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
