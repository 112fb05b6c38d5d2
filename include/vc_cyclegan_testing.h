/* vc_cyclegan_testing.h -- tuning, test and profiling hooks of libvcgan_hip.so.
 *
 * NOT part of the product API (include/vc_cyclegan.h): nothing in model.py / module.py / engine.py's training or
 * inference path calls these.  Users: tests/ (force every built kernel configuration against the oracle, check that the
 * benched shapes hit their measured table rows), tools/tune_igemm.py (regenerates csrc/vc_tuned_*.h) and bench.py's
 * roofline leg (per-launch HIP-event timing).  All state is per calling thread.
 */
#ifndef VC_CYCLEGAN_TESTING_H
#define VC_CYCLEGAN_TESTING_H
#ifdef __cplusplus
extern "C" {
#endif

/* Tuning / test hooks (tools/tune_igemm.py regenerates csrc/vc_tuned_igemm.h with them; tests force every built kernel
 * instantiation against the oracle).  Each forces ONE configuration for the following launches of the CALLING THREAD
 * (thread-local state); (0, 0, 0) restores the automatic choice.  Not for production use.
 *   igemm    : conv forward / input gradient, fp32 storage: tile rows (64 / 128), K groups per workgroup (1, 2, 4), split-K
 *   wgrad    : filter gradient, fp32 storage: tile columns (64 / 128), K groups (1, 2, 4), split-K
 *   conv_bf16: the 16-bit data path of the bf16 mode (csrc/vc_conv_bf16.hip): tile rows (64 / 128), LDS ring stages (2 / 4),
 *              split-K (1..16) */
int  vc_tune_igemm_override(int tile_rows, int k_groups, int splitk);
int  vc_tune_wgrad_override(int tile_cols, int k_groups, int splitk);
int  vc_tune_rowskip(int wgrad_on);     /* filter gradients of 2-D layers whose taps reach far into the SAME padding (D.d3) enumerate pixels row-major and
                                         * skip the output rows a K tile's taps never reach (csrc/vc_ops.hip: rowskip_pays); default on; thread-local */
int  vc_tune_wgrad_store(int on);       /* tuning only: while on, vc_conv2d_wgrad OVERWRITES dw (the train step's plain-store form) */
int  vc_tune_conv_bf16_override(int tile_rows, int stages, int splitk);
/* conv_bf16_kernel's one-workgroup-per-CU configurations (256 x 3, 128 x 4, 64 x 4) for the calling thread's following launches:
 * 1 = wave-specialised (512 threads: 4 multiplying + 4 copying waves; production), 0 = the 256-thread form, -1 = the process default. */
int  vc_tune_conv_bf16_spec(int on);
/* bf16 data path, calling thread's following backward passes: 1 (production) = the norm backward writes dy as pixel octets into the layer's
 * stacked buffer (and drops the fp32 copy inside a train step), 0 = it writes fp32 only and every filter gradient packs its operand itself,
 * 2 = octets written and the fp32 copy kept (separates the two halves of the change in control experiments). */
int  vc_tune_bf16_pixel_octets(int on);
/* D.h1's input gradient (Cin = 1, csrc/vc_conv_small.hip) for the calling thread's following launches: 1 = the gather kernel of deterministic
 * mode (fixed summation order) without the rest of that mode, 0 = the atomic scatter (production).  The scatter's summation order is the one
 * run-to-run noise source of a split-K-free step that PROPAGATES: it feeds the generators' output gradients. */
int  vc_tune_c1_dgrad_gather(int on);
/* Single-GPU parameter update (vc_step_exchange_and_update without a communicator) for the calling thread's following steps: 0 (production) =
 * the five-segment form of the multi-GPU exchange; 1 = ApplyAdam layer by layer in the order the backward finishes them, behind per-layer events
 * the backward records (44 launches and ~150 event operations more per step).  Measured both ways, round 6: in a fresh process 2.75 against
 * 2.80 ms at batch 1 and 7.02 against 7.07 ms at batch 8, but 3.34 against 3.01 ms at batch 1 once the host is the slower side (the same
 * process after a batch-8 run, which is where bench.py times configs[0]) and 6.72 against 6.67 ms at batch 8 with the Winograd route on -- so
 * it is a switch, not the default.  Also the timing control of the round-6 wrong-lane-group finding. */
int  vc_tune_layer_updates(int on);
/* fp32 mode, calling thread's following convolutions.  1 (production): inside a train step, forward and input gradient of the stride-1 1-D
 * convolutions with 3 or 5 taps whose width is a multiple of 4 (residual and up-sampling blocks, module.py:66-133: Winograd F(4,3) / F(4,5)) and of
 * the discriminators' 6 x 3 block (module.py:208: F(3,6) along H) run as GEMMs in the transform domain on filters transformed once per step, and
 * the 1-D layers' filter gradients in the fused transform-domain kernel (csrc/vc_conv_wino.hip); calls without cached filters (operator / network
 * level, inference) stay on the direct kernels.  2: those calls take the route too, transforming the filters per call (the parity tests).
 * 0: the direct implicit GEMM everywhere (control of the parity tests and of the A/B timing). */
int  vc_tune_winograd(int on);
/* Winograd-domain GEMM launches (one per convolution forward / input gradient on the route) the calling thread has issued; reset != 0 clears. */
long long vc_testing_wino_gemms(int reset);
/* Multiply-add work (FLOP) those launches did NOT execute compared with the direct form, since the last reset: kind 0 = forward + input gradient,
 * 1 = filter gradient.  bench.py prints the kernels' executed FLOP/s (its MFMA roofline) and, beside it, the direct-equivalent rate. */
double vc_testing_wino_saved_flops(int kind, int reset);
/* Bit mask of the switches of this header that are NOT at their production value in the calling thread (0 = a product run; the product
 * header's vc_thread_is_product_run() is `== 0` of this).  bits: 1 igemm, 2 wgrad, 4 conv_bf16 override, 8 conv_bf16_spec, 16 pixel octets,
 * 32 rowskip, 64 wgrad_store, 128 glu, 256 c1 gather, 512 layer updates, 1024 winograd. */
int  vc_testing_hooks_active(void);
/* Which configuration did the calling thread's last conv forward / input-gradient launch (igemm) or filter-gradient launch
 * (wgrad) run with, and where did it come from?  out4 = {tile rows | columns, K groups, split-K, source}; source: 0 cost
 * model, 1 exact row of the measured table (csrc/vc_tuned_igemm.h / vc_tuned_wgrad.h), 2 the same layer's row measured at
 * the nearest size, 3 forced by vc_tune_*_override, 4 the bf16-operand rule. */
int  vc_tune_last_igemm_config(int* out4);
int  vc_tune_last_wgrad_config(int* out4);
/* The same answer for a launch SHAPE, without launching (host only; works without a GPU): the table keys are
 * {transposed filter?, M, launch pixels, K rows, groups} and {K tiles of 64 rows, Cout, 32-pixel chunks, groups, plain-store?}. */
int  vc_tune_query_igemm(int a_trans, int M, int Ptot, int krows, int groups, int* out4);
int  vc_tune_query_wgrad(int ktiles, int N, int nchunks, int groups, int store, int* out4);
int  vc_tune_conv_bf16_keys(int* keys4, int max_keys);   /* tuning only: shape keys {M, Ptot, K octets, groups} of this thread's conv_bf16 launches since the last call */
/* Opt-in per-launch timing of the implicit-GEMM kernels (bench.py roofline leg).
 * While enabled every igemm/wgrad launch is bracketed by hipEvents on its stream. */
int  vc_prof_enable(int on);
/* Collect: sum of kernel ms, launch count and algorithmic FLOPs since enable/reset,
 * kind 0 = igemm (conv fwd + dgrad), 1 = wgrad, 2 = wsconv1d (generator h1 forward).  Synchronises the recorded events. */
int  vc_prof_collect(int kind, double* ms, long long* launches, double* flops);
/* Per-launch records (in launch order) since enable: fills ms[i], flops[i] for i < max_n; returns the record count. */
int  vc_prof_records(int kind, double* ms, double* flops, int max_n);

/* How many captured step graphs does the plan hold (vc_train_step_graph's cache)?  Tests: N identical graph steps must leave ONE
 * (a key built after the capture's own masking never matched, every step re-captured). */
int  vc_plan_num_step_graphs(const struct vc_plan* p);
/* Tuning switches of the gated conv1d + GLU kernel for the calling thread's following launches (0 = production).  bit 0: do NOT raise
 * the issue priority of the younger K set's waves (s_setprio 1 is the production setting).  bit 2: vc_prof_glu_copy_roof writes its
 * blocks with 16-byte stores instead of the kernel's dword stores (the kernel itself ignores the bit). */
int  vc_tune_glu(int flags);
/* Measurement control for the north-star kernel (bench.py named_kernel.copy_roof): ONE launch on vc_gated_conv1d_glu_fwd_packed's grid
 * (G = 1) that performs that kernel's memory operations only -- every workgroup reads its slab's bf16 fragment image and its tiles'
 * frames (eight tiles in flight) and writes its output rows with the same dword stores -- and no arithmetic.  Recorded under profiling kind 2. */
int  vc_prof_glu_copy_roof(const float* x, const void* image_bf16, float* out, int B, int T, int Cout2, void* stream);

/* Stale-read probes (tests; profiles/r06_stale_read_probe.txt): vc_testing_describe_step writes "name offset floats" lines for every buffer of the
 * train step inside the caller's workspace (the backward scratch areas as each of the four backward passes lays them out) into buf and
 * returns the bytes needed; vc_testing_poison_internal fills the plan-internal buffers (packed-operand workspaces of the plan's streams
 * and of `stream`, the bf16 filter images) with `byte` (0xff = NaN in fp32 and bf16) after a device-wide synchronisation. */
struct vc_plan;
long long vc_testing_describe_step(const struct vc_plan* p, int B, int Bbwd, char* buf, long long cap);
int  vc_testing_poison_internal(const struct vc_plan* p, void* stream, int byte);

#ifdef __cplusplus
}
#endif
#endif
