/* vc_cyclegan.h -- C-ABI of libvcgan_hip.so: the MI355X (gfx950) CycleGAN-VC
 * train-step hot path.
 *
 * The reference (leimao/Voice-Converter-CycleGAN) has no FFI: its hot path is a
 * TensorFlow-1 graph built by module.py and driven by model.CycleGAN.train().
 * Each entry point below therefore cites the reference *Python/TF call site* it
 * replaces (file:line relative to the reference tree).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - Every data pointer is a DEVICE pointer owned by the caller (fp32 unless
 *     noted).  `stream` is a hipStream_t passed as void*.  Calls enqueue work on
 *     that stream and return; they never synchronise.  Steady-state calls never
 *     allocate.  What IS allocated, once, on first use: the small geometry tables of
 *     a convolution geometry (or ahead of time by vc_plan_create); in bf16 mode the
 *     packed-operand workspace of the stream a convolution is issued on (operator-
 *     AND plan-level calls; grown with hipMalloc when a larger operand arrives, see
 *     vc_release_workspaces) and a plan's filter images (first bf16 train step).
 *   - Return value: 0 = ok, <0 = argument error (VC_E_*), >0 = hipError_t.
 *     vc_last_error() returns a human-readable message for the calling thread.
 *   - Layout is time-contiguous: 1-D activations [G][B][C][T], 2-D activations
 *     [G][B][C][H][W] (W = time).  G is the "group" dimension: both generators
 *     (or both discriminators) run in lock-step inside one launch; a tensor's
 *     group stride is a signed element count so group 1 may sit before group 0.
 *   - Conv kernels keep the TensorFlow layout [kh][kw][Cin][Cout] (Cout
 *     contiguous).  The value and gate convolutions of a gated block
 *     (module.py:85-98) are stored as ONE kernel with Cout = 2C: filters
 *     [0,C) are `*_h1_conv`, [C,2C) are `*_h1_gates`; same for bias/gamma/beta.
 */
#ifndef VC_CYCLEGAN_H
#define VC_CYCLEGAN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_E_ARG      (-1)   /* bad argument / unsupported geometry */
#define VC_E_NOMEM    (-2)   /* workspace too small */
#define VC_E_STATE    (-3)   /* bad handle */

typedef struct vc_plan vc_plan;   /* opaque: geometry tables + workspace map for one (F, T) */

/* ---- library ------------------------------------------------------------- */
int         vc_version(void);          /* ABI revision: 116 (bumped whenever an entry point changes or is added) */
int         vc_thread_is_product_run(void);   /* 1 when every test / tuning switch of vc_cyclegan_testing.h is at its production value in the
                                               * calling thread (model.CycleGAN.train and bench.py refuse to run otherwise: a forced tile
                                               * configuration left behind by a test must never be trained or benched with) */
/* The content hash of the sources + compiler flags this library was built from (16 hex digits; build.py stamps it, _lib.py
 * compares it with the sources in the tree at import and refuses to load on a mismatch; bench.py prints it). */
const char* vc_build_id(void);
const char* vc_last_error(void);
/* CRC-32C (Castagnoli) of `n` host bytes, continuing from `crc` (0 to start): the checksum inside TensorFlow's
 * checkpoint bundles and event files, which CycleGAN.save()/the summary writer emit (model.py:21,26-29).  Host only. */
uint32_t vc_crc32c(const void* data, uint64_t n, uint32_t crc);
/* Arithmetic of the MFMA kernels (conv forward / input gradient / filter gradient):
 *   0 fp32   (default) v_mfma_f32_32x32x2_f32 -- exact fp32 products; the reference's precision (model.py:35-42 float32)
 *   1 bf16   operands rounded to bf16 (round-to-nearest-even) on the way into v_mfma_f32_32x32x16_bf16, fp32 accumulate;
 *            tensors in HBM, instance-norm statistics, losses and Adam state stay fp32 (BASELINE config 3)
 *   2 bf16x3 operands split hi+lo into two bf16, three MFMAs (hi*hi + hi*lo + lo*hi): ~2^-16 relative per product
 *   3 f16    operands rounded to IEEE half (round-to-nearest-even, 10-bit mantissa) into v_mfma_f32_32x32x16_f16, fp32 accumulate: the
 *            bf16 kernel's rate at an eighth of its operand error (3e-4 of the output's maximum against fp32 arithmetic, where bf16
 *            gives 2.5e-3).  THREAD mode only and ONE operator only: vc_gated_conv1d_glu_fwd / _prepack / _fwd_packed (z-scored frames
 *            and glorot filters lie well inside half's range).  Every other operator, vc_plan_create and vc_plan_set_math_mode
 *            return VC_E_ARG in / for this mode -- nothing computes in a different arithmetic than the one asked for.
 * The mode is a property of a PLAN (vc_plan_set_math_mode): every launch made through a plan uses the plan's mode,
 * so two plans in one process never interfere.  vc_set_math_mode sets the calling THREAD's default: the mode of the
 * operator-level calls below (which carry no plan) and the initial mode of plans that thread creates.
 * Returns VC_E_ARG on an unknown mode. */
int  vc_set_math_mode(int mode);
int  vc_get_math_mode(void);
/* bf16 mode keeps packed bf16 copies of the convolution operands in one workspace per stream, grown on first use and
 * reused afterwards (steady state never allocates; vc_plan_destroy frees those of the plan's own streams).  This frees
 * them all (synchronises the device); they regrow on demand.  Step graphs captured in bf16 mode (vc_train_step_graph) hold
 * their addresses: the graph cache is keyed on a workspace generation, so a graph captured before a release / regrowth is
 * never replayed again (the step is re-captured). */
int  vc_release_workspaces(void);
/* Deterministic mode.  By default a few reductions meet in scheduling order (split-K partial tiles and bias / gamma /
 * beta sums are combined with fp32 atomics), so two runs on identical inputs agree to rounding, not bitwise -- and
 * d|x|/dx = sign(x) (utils.py:6-8) can turn one such rounding into a flipped gradient element.  With the mode on, every
 * launch reduces in a fixed order: K is split only inside a workgroup, per-channel parameter gradients are summed over
 * the batch inside one lane group, the Cin = 1 input gradient is a gather.  Two runs are then BITWISE equal (also a
 * hipGraph replay vs the same step launched kernel by kernel).  Slower on small batches (no split-K to fill the chip).
 * Like the math mode it is a property of a PLAN (vc_plan_set_deterministic); vc_set_deterministic sets the calling
 * thread's default: the mode of operator-level calls and the initial mode of plans that thread creates. */
int  vc_set_deterministic(int on);
int  vc_get_deterministic(void);
/* Tuning, test and profiling hooks (vc_tune_*, vc_prof_*): include/vc_cyclegan_testing.h -- not part of the product API. */

/* ---- parameter layout (replaces tf.get_variable bookkeeping, module.py:154-159,
 *      194-199; variable names of SURVEY.md section 8a row 13) ------------------ */
typedef struct {
    char    name[96];     /* TF variable name below the scope, e.g. "h1_conv/kernel" */
    int64_t offset;       /* float offset of element [..,0,0] inside one network's block */
    int32_t ndim;
    int32_t shape[4];     /* TF shape */
    int32_t ld;           /* stride (floats) between consecutive rows of the last dim */
} vc_param_rec;
/* kind 0 = generator_gatedcnn, 1 = discriminator.  Returns number of records
 * (writes at most max_recs); *n_floats = padded size of one network's block. */
int vc_param_layout(int kind, int num_features, vc_param_rec* recs, int max_recs, int64_t* n_floats);

/* ---- operator level (used by the network level; exported for parity tests) ---- */

/* tf.layers.conv1d / tf.layers.conv2d, padding='same' (module.py:32-40, 54-62).
 * x [G][B][Cin][H][W] -> y [G][B][Cout][OH][OW], 1-D: H = kh = sh = 1.
 * w [kh][kw][Cin][Cout], bias [Cout] or NULL.  *_gs are signed group strides. */
int vc_conv2d_fwd(const float* x, int64_t x_gs, const float* w, int64_t w_gs,
                  const float* bias, int64_t b_gs, float* y, int64_t y_gs,
                  int G, int B, int Cin, int H, int W, int Cout,
                  int kh, int kw, int sh, int sw, void* stream);
/* Conv2DBackpropInput of the same op (autodiff of model.py:107-108).
 * dy [G][B][Cout][OH][OW] -> dx [G][B][Cin][H][W].  addend (same shape as dx,
 * may be NULL) is added (residual skip, module.py:81).  accumulate != 0 adds
 * into the existing contents of dx instead of overwriting. */
int vc_conv2d_dgrad(const float* dy, int64_t dy_gs, const float* w, int64_t w_gs,
                    float* dx, int64_t dx_gs, const float* addend, int64_t add_gs, int accumulate,
                    int G, int B, int Cin, int H, int W, int Cout,
                    int kh, int kw, int sh, int sw, void* stream);
/* Conv2DBackpropFilter: dw += sum over batch and pixels (ALWAYS accumulates). */
int vc_conv2d_wgrad(const float* x, int64_t x_gs, const float* dy, int64_t dy_gs,
                    float* dw, int64_t dw_gs,
                    int G, int B, int Cin, int H, int W, int Cout,
                    int kh, int kw, int sh, int sw, void* stream);
/* The generator's first block in ONE kernel (module.py:161-163: h1 = conv1d, h1_gates = conv1d, h1_glu = gated_linear_layer,
 * no instance norm): out [G][B][Cout2/2][T] = (conv_v(x) + b_v) * sigmoid(conv_g(x) + b_g), stride 1, SAME.  w [k][Cin][Cout2]
 * with the value filters in [0, Cout2/2) and the gate filters in [Cout2/2, Cout2), bias likewise.  Neither raw convolution
 * is written: HBM sees the frames once and the activation once (the north-star kernel; inference plans use it, the train
 * step keeps value and gate for the backward).  Built for Cin = 24, k = 15 (24 MCEPs), Cout2 % 64 == 0, T % 4 == 0.
 * Math mode of the calling thread: fp32 (exact; also taken in mode 2), f16 (mode 3) or bf16 (operands rounded once: the filters when a workgroup builds its
 * register-resident fragments straight from w, the frames when they are staged).  ONE launch, no workspace.  Tensors of
 * one group must stay below 2 GiB (32-bit buffer offsets). */
int vc_gated_conv1d_glu_fwd(const float* x, int64_t x_gs, const float* w, int64_t w_gs,
                            const float* bias, int64_t b_gs, float* out, int64_t out_gs,
                            int G, int B, int Cin, int T, int Cout2, int k, void* stream);
/* The same block with the filters pre-packed (weights that do not change between calls: CycleGAN.test / convert.py, model.py:128-137,
 * or once per weight update).  vc_gated_conv1d_glu_prepack writes, per group, the MFMA fragment image of w for the calling
 * thread's math mode (bf16 / f16: rounded once, 16-bit; fp32: exact) -- vc_gated_conv1d_glu_image_bytes(Cin, k, Cout2) bytes, 16-byte
 * aligned, group stride in bytes; -1: geometry not built.  vc_gated_conv1d_glu_fwd_packed is vc_gated_conv1d_glu_fwd reading
 * that image instead of w: a workgroup fetches its filter fragments with 16-byte coalesced loads straight into registers.
 * The image must have been built in the math mode of the call: the bf16 and the f16 image of the same filters differ only in how their 16-bit
 * fields are read, so the library remembers which mode packed the image at an address (its last 64 prepack calls, host side) and returns
 * VC_E_ARG for a packed call in another mode; an image it has never seen is taken on trust. */
int64_t vc_gated_conv1d_glu_image_bytes(int Cin, int k, int Cout2);
int vc_gated_conv1d_glu_prepack(const float* w, int64_t w_gs, void* image, int64_t image_gs_bytes,
                                int G, int Cin, int k, int Cout2, void* stream);
int vc_gated_conv1d_glu_fwd_packed(const float* x, int64_t x_gs, const void* image, int64_t image_gs_bytes,
                                   const float* bias, int64_t b_gs, float* out, int64_t out_gs,
                                   int G, int B, int Cin, int T, int Cout2, int k, void* stream);
/* BiasAddGrad: db[c] += sum_{b,l} dy[g][b][c][l]  (dy [G][B][C][L]). */
int vc_bias_grad(const float* dy, int64_t dy_gs, float* db, int64_t db_gs,
                 int G, int B, int C, int L, void* stream);

/* instance_norm_layer + gated_linear_layer + pixel_shuffler + residual add
 * (module.py:3-20, 81, 135-146), fused.
 *   mode 0: out = IN(y_val) * sigmoid(IN(y_gate))      y [G][B][2*Cy][Ly]
 *   mode 1: out = y_val * sigmoid(y_gate)   (h1 pair, module.py:161-163, 201-203)
 *   mode 2: out = res + IN(y)               (residual1d_block tail, module.py:79-81)
 * ps != 0 applies tf.reshape pixel shuffle (x2) to each branch BEFORE the norm:
 *   logical row (c, l) of length L = 2*Ly reads conv channel (l&1)*C + c at l>>1.
 * C = output channels, L = logical row length (T, or H*W).  gamma/beta hold the
 * value branch in [0,C) and the gate branch in [C,2C).  stats [G][B][nb*C][2]
 * receives (mean, rstd) for the backward pass (unused in mode 1). */
int vc_instnorm_glu_fwd(int mode, int ps, const float* y, int64_t y_gs,
                        const float* res, int64_t res_gs,
                        const float* gamma, const float* beta, int64_t par_gs,
                        float* out, int64_t out_gs, float* stats, int64_t st_gs,
                        int G, int B, int C, int L, void* stream);
/* Backward of the above.  dout [G][B][C][L]; writes dy in the raw conv layout of y
 * (out of place), and ACCUMULATES dgamma/dbeta (same layout as gamma/beta; NULL in
 * mode 1) and dbias (conv bias gradient, [nb*Cy]) with atomics. */
int vc_instnorm_glu_bwd(int mode, int ps, const float* y, int64_t y_gs,
                        const float* dout, int64_t do_gs,
                        const float* gamma, const float* beta, int64_t par_gs,
                        const float* stats, int64_t st_gs,
                        float* dy, int64_t dy_gs,
                        float* dgamma, float* dbeta, int64_t dpar_gs,
                        float* dbias, int64_t dbias_gs,
                        int G, int B, int C, int L, void* stream);

/* tf.layers.dense(units=1, activation=sigmoid) on the channel axis (module.py:211).
 * x [G][B][C][HW] -> o [G][B][HW]. */
int vc_disc_head_fwd(const float* x, int64_t x_gs, const float* w, const float* b, int64_t par_gs,
                     float* o, int64_t o_gs, int G, int B, int C, int HW, void* stream);
/* dout,o [G][B][HW] -> dx [G][B][C][HW] (NULL to skip); dw [C], db [1] accumulate (NULL to skip). */
int vc_disc_head_bwd(const float* x, int64_t x_gs, const float* w, int64_t par_gs,
                     const float* o, int64_t o_gs, const float* dout, int64_t do_gs,
                     float* dx, int64_t dx_gs, float* dw, float* db, int64_t dpar_gs,
                     int G, int B, int C, int HW, void* stream);

/* The crop of preprocess.sample_train_data (preprocess.py:226-233) on GPU-resident utterances ("next" row 8f-1):
 * utterance n is a row-major [F][len[n]] block at src + off[n]; out[n][f][t] = src[off[n] + f*len[n] + start[n] + t].
 * off / len / start are DEVICE arrays of N entries. */
int vc_crop_batch(const float* src, const int64_t* off, const int32_t* len, const int32_t* start, float* out,
                  int N, int F, int n_frames, void* stream);

/* utils.l1_loss / utils.l2_loss (utils.py:6-12): *loss = mean(|y - y_hat|) resp. mean((y - y_hat)^2) over n
 * elements; d_yhat (NULL to skip) receives d loss / d y_hat (sign(0) = 0). */
int vc_l1_loss(const float* y, const float* y_hat, int64_t n, float* loss, float* d_yhat, void* stream);
int vc_l2_loss(const float* y, const float* y_hat, int64_t n, float* loss, float* d_yhat, void* stream);

/* tf.train.AdamOptimizer(lr, beta1=0.5).minimize -> ApplyAdam (model.py:107-108):
 *   m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= alpha*m/(sqrt(v)+eps)
 * alpha = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller in fp32 (TF keeps
 * the beta powers as fp32 variables).  grad_scale multiplies g first (1/world
 * after a summing all-reduce). */
int vc_adam_tf_update(float* p, float* m, float* v, const float* g, int64_t n,
                      float alpha, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* ---- network level --------------------------------------------------------- */
/* Geometry plan for num_features x n_frames inputs (n_frames % 4 == 0): geometry tables of every layer of both networks
 * (forward, input gradient, filter gradient), streams and events of the step schedule.  vc_plan_create_inference builds
 * only what vc_generator_forward needs (CycleGAN.test, model.py:128-137: a new utterance length per call) -- the other
 * network-level entry points refuse such a plan.  vc_plan_destroy gives the tables back (geometries are shared between
 * plans and freed with their last user). */
int  vc_plan_create(int num_features, int n_frames, vc_plan** out);
int  vc_plan_create_inference(int num_features, int n_frames, vc_plan** out);
void vc_plan_destroy(vc_plan* p);
int  vc_plan_set_math_mode(vc_plan* p, int mode);     /* see vc_set_math_mode */
int  vc_plan_get_math_mode(const vc_plan* p);
int  vc_plan_set_deterministic(vc_plan* p, int on);   /* see vc_set_deterministic */
int  vc_plan_get_deterministic(const vc_plan* p);
/* Workspace sizes in floats for batch B per group. */
int64_t vc_generator_save_floats(const vc_plan* p, int B);       /* activations kept for backward */
int64_t vc_discriminator_save_floats(const vc_plan* p, int B);
int64_t vc_scratch_floats(const vc_plan* p, int B);              /* backward scratch (max over G and D); larger in bf16 mode (the filter-gradient GEMMs'
                                                                 * pixel-octet operands live in it): query again after vc_plan_set_math_mode */

/* module.generator_gatedcnn (module.py:148-185) for G networks in lock-step.
 * params: base of group 0's parameter block, par_gs = signed stride to group 1's.
 * x [G][B][F][T] (group stride x_gs) -> y [G][B][F][T].  save: vc_generator_save_floats(p,B)*G floats. */
int vc_generator_forward(const vc_plan* p, const float* params, int64_t par_gs,
                         const float* x, int64_t x_gs, float* y, int64_t y_gs,
                         float* save, int G, int B, void* stream);
/* Backward of the same application.  dy [G][B][F][T]; gradients ACCUMULATE into
 * dparams (layout == params; NULL skips all weight gradients); dx (NULL to skip)
 * receives d/dx, added to its contents if dx_accumulate.  scratch: scratch_floats >= vc_scratch_floats(p,B)*G floats, queried
 * AFTER the plan's math mode was set (bf16 lays more into it); a smaller scratch is refused with VC_E_NOMEM (ABI 113: the size
 * used to be implied, and a scratch sized before vc_plan_set_math_mode(bf16) was overrun silently). */
int vc_generator_backward(const vc_plan* p, const float* params, int64_t par_gs,
                          const float* x, int64_t x_gs, const float* dy, int64_t dy_gs,
                          const float* save, float* dparams, int64_t dpar_gs,
                          float* dx, int64_t dx_gs, int dx_accumulate,
                          float* scratch, int64_t scratch_floats, int G, int B, void* stream);
/* module.discriminator (module.py:188-213).  x [G][B][F][T] -> o [G][B][H'*W']. */
int vc_discriminator_out_elems(const vc_plan* p);   /* H'*W' per sample (48 for 24x128) */
int vc_discriminator_forward(const vc_plan* p, const float* params, int64_t par_gs,
                             const float* x, int64_t x_gs, float* o, int64_t o_gs,
                             float* save, int G, int B, void* stream);
int vc_discriminator_backward(const vc_plan* p, const float* params, int64_t par_gs,
                              const float* x, int64_t x_gs, const float* dout, int64_t do_gs,
                              const float* save, float* dparams, int64_t dpar_gs,
                              float* dx, int64_t dx_gs, int dx_accumulate,
                              float* scratch, int64_t scratch_floats, int G, int B, void* stream);

/* ---- step level: model.CycleGAN.train (model.py:110-125) minus the update ---- */
/* Offsets (floats) of the networks inside the flat parameter / gradient / Adam buffers:
 * [generator_A2B | generator_B2A | discriminator_A | discriminator_B]. */
int64_t vc_flat_param_floats(const vc_plan* p);
int64_t vc_flat_offset(const vc_plan* p, int net /*0..3*/);
int64_t vc_step_workspace_floats(const vc_plan* p, int B);       /* depends on the plan's math mode (bf16 is larger): query after vc_plan_set_math_mode;
                                                                 * the step entry points refuse a workspace that is too small */
/* One fused forward+backward of the whole CycleGAN graph (model.py:44-90):
 * inputs A,B [B][F][T]; params/grads flat buffers (16-byte aligned); every element of grads is (over)written: filter
 * gradients by plain stores of their single launch, everything that is accumulated (biases, gamma/beta, padding) is
 * zeroed first by one multi-buffer launch -- the caller never has to clear grads.
 * losses[8] (device) = cycle_loss, identity_loss, generator_loss_A2B,
 * generator_loss_B2A, generator_loss, discriminator_loss_A, discriminator_loss_B,
 * discriminator_loss (model.py:155-167).  gen_A/gen_B [B][F][T] receive
 * generation_A / generation_B (model.py:112).  The Adam updates (vc_adam_tf_update)
 * and, for N>1 ranks, the gradient all-reduce are issued by the caller.
 * Events recorded inside let the caller overlap both with the backward: vc_step_wait_d_grads / vc_step_wait_g_bucket,
 * or simply call vc_step_exchange_and_update. */
int vc_cyclegan_step_grads(const vc_plan* p, const float* params, float* grads,
                           const float* A, const float* B, int batch,
                           float lambda_cycle, float lambda_identity,
                           float* losses, float* gen_A, float* gen_B,
                           float* workspace, int64_t workspace_floats, void* stream);
/* The step runs its independent chains (cycle-hop forward || D forward; generator_loss backward || discriminator_loss
 * backward) on `stream` and an internal side stream, joined before it returns.  enabled = 0 serialises them on `stream`
 * (used when timing individual kernels, so launches do not share the chip).  Default: enabled. */
int vc_plan_set_overlap(vc_plan* p, int enabled);
/* How vc_step_exchange_and_update exchanges the gradients of a data-parallel step (N > 1 ranks; with one rank it only orders the updates):
 *   VC_EXCHANGE_BUCKETED (default): 5 segments in the order they become final -- the discriminator half, then four generator buckets in
 *     reverse layer order (9 ncclAllReduce calls, the generators' two ranges of a bucket in one group) -- each all-reduced and updated on
 *     the update stream while the rest of the backward still runs;
 *   VC_EXCHANGE_SINGLE: BASELINE.json's literal form -- ONE ncclAllReduce over the whole flat gradient (479 MB) after the backward, then
 *     both updates.  Same arithmetic (SUM, 1/world folded into Adam); the A/B partner on multi-GPU hardware (bench.py --exchange). */
#define VC_EXCHANGE_BUCKETED 0
#define VC_EXCHANGE_SINGLE   1
int vc_plan_set_exchange_mode(vc_plan* p, int mode);
int vc_plan_get_exchange_mode(const vc_plan* p);

/* Makes `other_stream` wait until the discriminator half of `grads` from the most recent
 * vc_cyclegan_step_grads on this plan is final (it is produced before the generator backward) AND the step has
 * finished reading the discriminator weights -- after it the caller may all-reduce that half and update those
 * parameters in place while the generator backward is still running. */
int vc_step_wait_d_grads(const vc_plan* p, void* other_stream);

/* Generator gradients become final bucket by bucket (layers in reverse order) during the last backward of
 * vc_cyclegan_step_grads.  Bucket k covers floats [lo, hi) of EACH generator's block (add vc_flat_offset(p, 0|1));
 * vc_step_wait_g_bucket makes `other_stream` wait until bucket k of the most recent step is final and the step no
 * longer reads the weights of those layers (same guarantee as above: all-reduce + in-place update may start).  Buckets
 * are numbered in the order they complete and together tile [0, generator block size). */
int vc_step_num_g_buckets(const vc_plan* p);
int vc_step_g_bucket(const vc_plan* p, int k, int64_t* lo, int64_t* hi);
int vc_step_wait_g_bucket(const vc_plan* p, int k, void* other_stream);

/* The WHOLE train step (model.py:110-125: vc_cyclegan_step_grads + both ApplyAdam updates, single GPU) as one hipGraph
 * replay.  The reference trains at mini_batch_size = 1 (train.py:16), where the step is ~330 small launches and the host
 * is the bottleneck.  The first call with a given (batch, lambda_identity == 0, buffer set) captures and instantiates the
 * graph (cached in the plan, at most 8); later calls write {lambda_cycle, lambda_identity, alpha_g, alpha_d} to a device
 * buffer and replay.  Work is ordered after what `stream` holds at the call and `stream` waits for it.  All buffers as in
 * vc_cyclegan_step_grads / vc_adam_tf_update; they must stay alive (and at the same addresses) between replays. */
int vc_train_step_graph(vc_plan* p, float* params, float* grads, float* m, float* v,
                        const float* A, const float* B, int batch,
                        float lambda_cycle, float lambda_identity, float alpha_g, float alpha_d,
                        float beta1, float beta2, float eps,
                        float* losses, float* gen_A, float* gen_B,
                        float* workspace, int64_t workspace_floats, void* stream);

/* ---- data parallel: the step's one gradient exchange (the reference itself is single-process, model.py:22) ----
 * One process per GPU, replicas identical, global batch split contiguously over the ranks.  RCCL is bound at run time
 * (dlopen librccl.so.1); single-GPU users never need it. */
typedef struct vc_comm vc_comm;
/* ncclGetUniqueId: rank 0 fills 128 bytes and ships them to the other ranks by any means (file, socket, MPI, a
 * torch.distributed broadcast). */
int vc_comm_unique_id(void* id128);
/* ncclCommInitRank on the calling thread's current HIP device; collective over all `world` ranks. */
int vc_comm_init(int rank, int world, const void* id128, vc_comm** out);
int vc_comm_destroy(vc_comm* c);
int vc_comm_world(const vc_comm* c);                   /* ranks RCCL reports for this communicator (1 for NULL) */
int64_t vc_comm_collectives(const vc_comm* c);         /* ncclAllReduce calls enqueued through this communicator so far (diagnostics / tests) */
/* in-place SUM all-reduce of n floats on `stream` (ncclAllReduce, fp32) */
int vc_allreduce_f32(const vc_comm* c, float* buf, int64_t n, void* stream);
/* After vc_cyclegan_step_grads: gradient exchange + both ApplyAdam updates (model.py:107-108 / 112-119), segment by
 * segment on `update_stream` as soon as a segment's gradients are final and its weights no longer read (the events of
 * vc_step_wait_*): SUM all-reduce over `comm` (NULL: single GPU, no collective; a communicator of ANY size, one rank
 * included, gets the collectives: 1 for the discriminator half + a group of 2 per generator bucket), then Adam with
 * grad_scale 1/world.  alpha_g / alpha_d as in vc_adam_tf_update, one per optimizer.  The caller makes its main stream wait for `update_stream`
 * before the next step. */
int vc_step_exchange_and_update(const vc_plan* p, const vc_comm* comm, float* params, float* grads, float* m, float* v,
                                float alpha_g, float alpha_d, float beta1, float beta2, float eps, void* update_stream);

#ifdef __cplusplus
}
#endif
#endif
