// Internal launch descriptors shared by the device TUs (vc_conv_igemm.hip, vc_conv_wgrad.hip, vc_conv_small.hip,
// vc_elementwise.hip) and the host orchestration (vc_host.h: vc_tables.hip, vc_ops.hip, vc_net.hip, vc_step.hip; vc_comm.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include <vector>

// Arithmetic of the MFMA kernels (vc_plan_set_math_mode / vc_set_math_mode):
//   0  fp32   v_mfma_f32_32x32x2_f32, exact fp32 products (default; the parity configuration)
//   1  bf16   operands rounded to bf16 (RNE) on their way into v_mfma_f32_32x32x16_bf16, fp32 accumulate
//   2  bf16x3 every operand split x = hi + lo (two bf16), products hi*hi + hi*lo + lo*hi: ~2^-16 relative per product
//   3  f16    operands rounded to IEEE half (RNE, 10-bit mantissa) into v_mfma_f32_32x32x16_f16, fp32 accumulate: the bf16 kernel's rate at
//             an eighth of its operand error.  Thread mode only, and built for ONE operator: the gated conv1d + GLU forward
//             (vc_conv_glu.hip; z-scored frames and glorot filters sit well inside half's range).  Every other entry point refuses it.
#define VC_MATH_FP32 0
#define VC_MATH_BF16 1
#define VC_MATH_BF16X3 2
#define VC_MATH_F16 3

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 8 k-values of one operand row -> one bf16 MFMA fragment (and the residual fragment for the split mode)
template <int MATH> __device__ __forceinline__ void to_frag(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (__bf16)v[e];
        if constexpr (MATH == VC_MATH_BF16X3) lo[e] = (__bf16)(v[e] - (float)hi[e]);
    }
}

// x / d and x % d for 0 <= x < 2^24 with inv = 1.0f / d: float estimate plus one correction step (a 32-bit integer
// division costs ~40 instructions on gfx950; this is 8).
__device__ __forceinline__ int fast_divmod(int x, int d, float inv, int* rem) {
    int q = (int)((float)x * inv);
    int r = x - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
    *rem = r;
    return q;
}

// One entry per GEMM-K row of an implicit-GEMM convolution (filter-gradient kernel).
//   c     : channel of the gathered tensor this K row reads
//   dh,dw : offset (in gathered-tensor pixels) added to the launch pixel (i*si, j*sj)
//   a_off : float offset of this K row inside the weight tensor, -1 = padding row
struct __attribute__((aligned(16))) KEntry { int c, dh, dw, a_off; };
struct Tap { int dh, dw; };
#define VC_OOB 0xffffffffu     // buffer-load offset beyond num_records: the hardware returns 0
#define VC_KC 32               // K rows per chunk; every chunk of a table lies inside ONE filter tap (taps are padded to 32 rows)

// Y[g][b][m][P(i,j)] (+)= sum_kk A[a_off(kk) + m*a_ms] * X[g][b][c(kk)][i*si+dh(kk)][j*sj+dw(kk)]
// The K rows are described by three scalar-cache tables (structure of arrays, one scalar load serves 4-8 rows):
//   xsoff[kk] : byte offset (c*xHW + dh*xW + dw)*4 + xbias of the row inside the gathered tensor, >= 0.  It travels in the
//               buffer load's SGPR offset (excluded from the hardware range check); the lane's own pixel offset travels
//               in the VGPR offset, which is VC_OOB when the chunk's tap falls outside the tensor for that pixel.
//   rtap[kk/4]: 1 << tap of the aligned run of 4 rows (0 = padding run; runs never straddle taps or padding)
//   aoff[kk]  : byte offset of the row inside the weight tensor (padding rows: 0 -- any finite value, their X is 0)
struct IgemmArgs {
    const float* A; const float* X; float* Y; const float* bias; const float* addend;
    const int* xsoff; const unsigned* rtap; const unsigned* aoff;
    const Tap* taps; int ntaps;                   // <= 32 taps
    int xbias;                                    // bytes the X base pointer is moved down so that every xsoff >= 0
    unsigned x_bytes, a_bytes;                    // buffer sizes (per group) for the hardware range check
    long a_gs, x_gs, y_gs, bias_gs, add_gs;      // signed group strides (floats)
    long x_bs, y_bs;                              // batch strides (floats)
    int krows;                                    // K rows of the table (a multiple of 32)
    int M, a_ms;                                  // output rows, stride of m in A
    int Hl, Wl;                                   // bounds of the gathered tensor
    int xHW, xW;                                  // channel / row stride of the gathered tensor
    int si, sj;                                   // pixel stride launch grid -> gathered tensor
    int NPB, OWc, Ptot;                           // launch pixel grid: per sample, width, total
    float inv_npb, inv_ow;                        // 1/NPB, 1/OWc (fast_divmod)
    int yHW, yW;                                  // channel / row stride of the output tensor
    int so_h, so_w, o_h0, o_w0;                   // launch pixel -> output pixel
    int vec4;                                     // 4 consecutive launch pixels are 16 contiguous, aligned output bytes
    int splitk, accumulate, mtiles, ntiles, groups;
    int bm, kg;                                   // tile rows (64 / 128) and K groups inside a workgroup (1, 2, 4)
    // fuse != 0 (forward of a gated block with instance norm, module.py:85-98: 64-row tiles, no split-K, a (sample, channel)
    // row of NPB = 32 or 64 pixels inside the 64-pixel tile): a tile holds 32 value channels and the matching 32 gate
    // channels, and the epilogue writes the raw pair (for the backward), the statistics and act = IN(value) * sigmoid(IN(gate))
    int fuse;
    const float* gamma; const float* beta; float* stats; float* act;
    long par_gs, st_gs, act_gs;
};

// Up to 4 launches of the same tile configuration fused into one grid (blockIdx.y): the stride-parity classes of an
// input gradient, which individually underfill the chip.
struct IgemmMulti { IgemmArgs c[4]; int n; };

// dW[a_off(kk) + co] (+)= sum over segments, samples and pixels of X[g][b][c(kk)][i*si+dh][j*sj+dw] * dY[g][b][co][i][j].
// A segment is one application of the layer (the generator runs twice per step on different batches, model.py:44-51);
// both reduce into the same filter gradient, so they go out as ONE launch whose reduction axis is their concatenation.
struct WgradSeg { const float* X; const float* dY; long x_gs, dy_gs; unsigned x_bytes, dy_bytes; int Ptot, nchunks;
                  int RW; float inv_rw; };       // rowskip: pixels per output row of this application (B * OWc)
struct WgradArgs {
    WgradSeg seg[2]; int nseg;
    float* dW; const KEntry* ktab; const int* xsoff; int xbias; long dw_gs; long x_bs, dy_bs;   // xsoff / xbias: as IgemmArgs (forward table)
    float inv_ow, inv_npb;
    int N;                                        // Cout
    int Hl, Wl, xHW, xW, si, sj;
    int NPB, OWc;
    int nchunks;                                  // all segments
    int splitk, ktiles, ntiles, groups;
    int store;                                    // dW holds no earlier contribution and split-K is 1: plain stores
    int rowskip;                                  // 2-D layers whose kernel height reaches far into the SAME padding (D.d3: 6 taps on 6 rows): pixels enumerated
                                                  // (row, sample, column); a K tile reduces only over the output rows its two taps reach
};

// The 16-bit data path of the bf16 mode (vc_conv_bf16.hip): Y[g][b][m][P] (+)= sum_k Ap[g][m][k] * Xp-gather[k][P].
//   X      packed gathered tensor  [g][b][C/8][Hl*Wl][8] bf16 (16-byte units: 8 channels of one pixel)
//   A      packed filter image     [g][koct][M][8] bf16, k in the order of the octet table, padding octets 0
//   xsoff  per K octet: byte offset ((c/8)*Hl*Wl + dh*xW + dw)*16 + xbias of the unit (SGPR offset of the load)
//   otap   per K octet: 1 << tap (0 = padding octet: the load is sent out of range and returns 0)
// Pixel grid / output fields as IgemmArgs.  Strides of A and X in BYTES, of Y / bias / addend in floats.
struct ConvBf16Args {
    const void* A; const void* X; float* Y; const float* bias; const float* addend;
    const int* xsoff; const unsigned* otap; const Tap* taps; int ntaps; int xbias;
    unsigned x_bytes, a_bytes;
    long a_gs, x_gs, x_bs;                        // bytes
    long y_gs, bias_gs, add_gs, y_bs;             // floats
    int koct;                                     // K octets of the table (a multiple of 8 = whole 64-row chunks)
    int M;
    int Hl, Wl, xW, si, sj;
    int NPB, OWc, Ptot; float inv_npb, inv_ow;
    int yHW, yW, so_h, so_w, o_h0, o_w0, vec4;
    int splitk, accumulate, mtiles, ntiles, groups;
    int bm, nst;                                  // tile rows (128 / 64), LDS ring depth (2 / 4)
};
void conv_bf16_config(ConvBf16Args& a);           // fills bm / nst / mtiles / ntiles / splitk
int launch_conv_bf16(const ConvBf16Args& a, hipStream_t s, double flops, int prof_kind = 0);
// filter-gradient operands: Xcol[g][po][k][8] (im2col of the packed activations, 8 output pixels to a unit) and
// dYq[g][po][co][8]; po0 = first pixel octet of this application inside the (two-application) buffers
int launch_im2col_bf16(const void* xp, long xp_gs, void* xcol, long xcol_gs, const Tap* taps, int ntaps, int G, int B, int Cin,
                       int Hl, int Wl, int OH, int OW, int si, int sj, int po0, hipStream_t s);
int launch_pack_pixel_octets_bf16(const float* dy, long dy_gs, void* out, long out_gs_bytes, int G, int B, int Cout, int NPB, int po0, hipStream_t s);
int launch_pack_act_bf16(const float* x, long x_gs, void* xp, long xp_gs_bytes, int G, int B, int C, int HW, hipStream_t s);
// Ap[g][k/8][m][k%8] = bf16(W[g][wsrc[k] + m * a_ms]), 0 where wsrc[k] < 0; kp = 8 * koct
int launch_pack_filter_bf16(const float* w, long w_gs, void* ap, long ap_gs_bytes, const int* wsrc, int a_ms, int G, int M, int kp, hipStream_t s);
// many filter images in one launch: job = one image (both groups); ap_gs in bf16 elements; blk0 / nkb are filled by the launcher
struct PackFilterJob { const float* w; unsigned short* ap; const int* wsrc; long w_gs, ap_gs; int a_ms, M, kp, blk0, nkb; };
#define VC_PACK_JOBS 48
struct PackFilterJobs { PackFilterJob j[VC_PACK_JOBS]; int n; };
int launch_pack_filter_multi(const std::vector<PackFilterJob>& jobs, int G, hipStream_t s);

// Single-input-channel convolution (the discriminator's first layer, module.py:201-202: Cin = 1, 3x3 -> K = 9): a
// bandwidth job, done on the vector ALU instead of padding K = 9 up to an MFMA tile.  kh*kw <= 9.
struct C1Args {
    const float* x; const float* w; const float* bias; const float* addend; const float* dy;
    float* y; float* dx; float* dw; float* db;
    float* act; long act_gs;                                      // forward, optional: GLU of the gated pair [G][B][Cout/2][P]
    long x_gs, w_gs, b_gs, y_gs, dx_gs, add_gs, dw_gs, db_gs;     // signed group strides (floats); y_gs also strides dy
    int B, H, W, Cout, kh, kw, sh, sw, OH, OW, pt, pl, G;
    int math;                                               // VC_MATH_*: bf16 mode rounds both operands like the MFMA kernels do
    int det;                                                // deterministic mode: gather-form input gradient, one block per filter-gradient element
};
int launch_c1_fwd(const C1Args& a, hipStream_t s);
int launch_c1_dgrad(const C1Args& a, int accumulate, hipStream_t s);   // dx (+)= ...; atomics into dx (zeroed here unless accumulate)
int launch_c1_wgrad(const C1Args& a, hipStream_t s);                   // dw += ... (atomics)

// Weight-stationary 1-D convolution forward for few input channels (the generator's h1 layer, module.py:161-162:
// 24 -> 2x128 channels, k = 15, stride 1): K = Cin*k = 360 is too short a loop for the tiled implicit GEMM (12 chunks
// per workgroup, mostly prologue/epilogue).  Here a workgroup keeps a 32-row filter slab in LDS and streams pixel tiles
// through it; the k-fold window expansion happens in LDS, so HBM sees every input frame once per slab.
struct WsConvArgs {
    const float* x; const float* w; const float* bias; float* y;
    long x_gs, w_gs, b_gs, y_gs;          // signed group strides (floats)
    int B, T, Cout, G;
    float* raw; long raw_gs;              // gated_conv1d_glu only, optional: the raw (value, gate) pair [G][B][Cout][T] for a backward pass
    int flags;                            // gated_conv1d_glu only: tuning switches (vc_tune_glu; 0 in production)
    const void* wimg; long wimg_gs;       // gated_conv1d_glu only, optional: the filters as a pre-packed fragment image (group stride in BYTES); w unused then
};
int launch_wsconv1d(const WsConvArgs& a, int cin, int kw, int math, hipStream_t s, double flops);   // returns -2 when the geometry is not built
// the whole gated block in one kernel: y [G][B][Cout/2][T] = (conv_v + b_v) * sigmoid(conv_g + b_g); -2 when not built
int launch_gated_conv1d_glu(const WsConvArgs& a, int cin, int kw, int math, hipStream_t s, double flops);
// the fragment image of a gated filter pair w [kw][cin][cout] for `math` (fp32 / bf16): bytes per group (-1: geometry not built), builder
long gated_conv1d_glu_image_bytes(int cin, int kw, int cout, int math);
int launch_gated_conv1d_glu_prepack(const float* w, long w_gs, void* img, long img_gs_bytes, int G, int cin, int kw, int cout, int math, hipStream_t s);
// measurement control: the launch of the gated kernel with its memory operations only (a.wimg = a bf16 image, a.y written)
int launch_glu_copy_roof(const WsConvArgs& a, int cin, int kw, hipStream_t s);

// Winograd F(4,3) / F(4,5) transforms (vc_conv_wino.hip): 1-D, 3 or 5 taps, stride 1, W = 4 nt; taps + 3 transform positions
struct WinoFilter { const float* w; float* U; long w_gs; int Cin, Cout, taps; }; // U [G][taps + 3][Cin][Cout], group stride (taps + 3) Cin Cout (taps 6: F(3,6) along H, 8 planes, Cin = kw x the layer's)
int launch_wino_filters(const std::vector<WinoFilter>& f, int G, hipStream_t s);
int launch_wino_input(const float* x, long x_gs, float* V, int G, int B, int C, int nt, hipStream_t s, int taps = 3);
int launch_wino_output(const float* M, const float* bias, long b_gs, float* y, long y_gs, int G, int B, int C, int nt, hipStream_t s, int taps = 3);
int launch_wino_dy(const float* dy, long dy_gs, float* dM, int G, int B, int C, int nt, hipStream_t s, int taps = 3);
int launch_winoh(int mode, const float* src, long src_gs, float* dst, long dst_gs, const float* bias, long b_gs, const float* addend, long add_gs,
                 int accumulate, int G, int B, int C, int nt, int W, hipStream_t s);      // F(3,6) along H (D.d3): see vc_conv_wino.hip
struct WinoPrepJob { const float* src; long src_gs; float* dst; int B, C, kind; };   // kind 0: x -> V = B^T x, 1: dy -> dM = A dy
int launch_wino_prep(const WinoPrepJob* jobs, int n, int G, int nt, hipStream_t s, int taps = 3);
int launch_wino_wgrad(const float* const* V, const float* const* dM, const int* B, int napp, float* dW, long dw_gs, int store,
                      int G, int Cin, int Cout, int nt, hipStream_t s, double flops, int taps = 3);
int launch_wino_dx(const float* dV, float* dx, long dx_gs, const float* addend, long add_gs, int accumulate, int G, int B, int C, int nt, hipStream_t s, int taps = 3);

struct NormArgs {
    const float* y; const float* res; const float* gamma; const float* beta;
    const float* dout; const float* stats_in;
    float* out; float* stats; float* dy; float* dgamma; float* dbeta; float* dbias;
    void* outp; long outp_gs;             // optional: the output (forward) / dy (backward) once more as bf16 channel octets [b][channels / 8][positions][8] (group stride in BYTES)
    void* outq; long outq_gs; int outq_po0;   // optional (backward): dy once more as bf16 PIXEL octets [outq_po0 + pixel octet][conv channel][8] -- the filter-gradient
                                          // GEMM's operand, this application's part of the layer's stacked buffer (group stride in BYTES)
    long y_gs, res_gs, par_gs, out_gs, st_gs, do_gs, dy_gs, dpar_gs, dbias_gs;
    int G, B, C, L, ps;
    int det;                              // backward: reduce dgamma / dbeta / dbias over the batch inside one lane group (fixed order, no atomics)
    // fp32 mode, Winograd route (vc_conv_wino.hip; transform-domain tensors are six planes [p][b][channel][L / 4], group stride 6 planes):
    // the transforms next to a norm launch run inside it (launch_norm_* honour these for rows of up to 128 positions without pixel shuffle, not
    // in deterministic mode: ask norm_takes_wino first).  Forward: the raw conv output ARRIVES as M planes (win; + the conv bias wbias) and is
    // written to yw (= y, kept for the backward); the activation leaves once more as V planes (wout) for the next layer's GEMMs.  Backward:
    // dout ARRIVES as dV planes (dwin; + dadd, the skip path) and is written to dsave when somebody else needs it; dy leaves once more as dM
    // planes (dwout) for this layer's input-gradient GEMMs.
    const float* win; long win_gs; const float* wbias; long wbias_gs; float* yw;
    float* wout; long wout_gs;
    const float* dwin; long dwin_gs; const float* dadd; long dadd_gs; float* dsave; long dsave_gs;
    float* dwout; long dwout_gs;
};
bool norm_takes_wino(const NormArgs& a, bool fwd);   // launch_norm_fwd / launch_norm_bwd would honour the Winograd fields for this launch

// picks tile rows, K groups and split-K for a launch (fills bm/kg/splitk/mtiles/ntiles)
void igemm_config(IgemmArgs& a, int math);
int launch_igemm(const IgemmArgs& a, bool a_trans, int math, hipStream_t s, double flops);
int launch_igemm_multi(const IgemmArgs* cls, int n, bool a_trans, int math, hipStream_t s, double flops);   // all configured alike
int wgrad_splitk(const WgradArgs& a);            // the split-K factor launch_wgrad will use (1 = plain stores possible)
int launch_wgrad(const WgradArgs& a, int math, hipStream_t s, double flops);
int launch_zero_list(const std::vector<std::pair<float*, long>>& bufs, hipStream_t s);
struct CopyItem { float* dst; const float* src; long n; };
int launch_copy_list(const std::vector<CopyItem>& items, hipStream_t s);      // dst[i][0..n) = src[i][0..n), one launch per 8 items
int launch_add(float* dst, const float* src, long n, hipStream_t s);   // dst += src
int launch_norm_fwd(int mode, const NormArgs& a, hipStream_t s);
bool norm_fwd_emits_packed(const NormArgs& a);     // NormArgs::outp is honoured for this launch (else: leave it null)
bool norm_bwd_emits_packed(const NormArgs& a);     // ... by launch_norm_bwd: dy as channel octets in the convolution's layout
bool norm_bwd_emits_pixel_octets(const NormArgs& a);   // NormArgs::outq is honoured by launch_norm_bwd (the convolution's pixel count per sample is a multiple of 8, ...)
int launch_norm_bwd(int mode, const NormArgs& a, hipStream_t s);
int launch_bias_grad(const float* dy, long dy_gs, float* db, long db_gs, int G, int B, int C, int L, hipStream_t s);
int launch_head_fwd(const float* x, long x_gs, const float* w, const float* b, long par_gs,
                    float* o, long o_gs, int G, int B, int C, int HW, hipStream_t s);
int launch_head_bwd(const float* x, long x_gs, const float* w, long par_gs, const float* o, long o_gs,
                    const float* dout, long do_gs, float* dx, long dx_gs, float* dw, float* db,
                    long dpar_gs, int G, int B, int C, int HW, hipStream_t s);
int launch_adam(float* p, float* m, float* v, const float* g, long n, float alpha, float b1,
                float b2, float eps, float gscale, hipStream_t s, const float* alpha_dev = nullptr);   // alpha_dev: read the step size from device memory
int launch_set4(float* dst, float a, float b, float c, float d, hipStream_t s);

struct LossArgs {
    const float* A; const float* Bm;      // real [B][F][T]
    const float* out1; long out1_gs;      // [2][2B][F][T]: g0 = A2B([A;B]) = [gen_B; id_B], g1 = B2A([B;A]) = [gen_A; id_A]
    const float* cyc;  long cyc_gs;       // [2][B][F][T]:  g0 = A2B(gen_A) = cycle_B,      g1 = B2A(gen_B) = cycle_A
    const float* dout; long dout_gs;      // [2][2B][HW]:   g0 = D_A([gen_A; A]),            g1 = D_B([gen_B; B])
    float* d_out1; long d_out1_gs;        // identity halves written, generation halves zeroed
    float* d_cyc;  long d_cyc_gs;
    float* d_dout_D; long d_dout_D_gs;    // [2][2B][HW] seeds of discriminator_loss
    float* d_dout_G; long d_dout_G_gs;    // [2][B][HW]  seeds of generator_loss_{B2A,A2B}
    float* acc;                           // VC_LOSS_BLOCKS x 10 per-block partial sums
    float* losses;                        // 8 scalars
    int B, FT, HW; float lambda_cycle, lambda_identity;
    const float* hp;                      // non-null: {lambda_cycle, lambda_identity} are read from device memory (captured steps)
};
#define VC_LOSS_BLOCKS 256
int launch_loss(const LossArgs& a, hipStream_t s);
int launch_loss_finalize(const LossArgs& a, hipStream_t s);
int launch_crop_batch(const float* src, const long* off, const int* len, const int* start, float* out,
                      int N, int F, int n_frames, hipStream_t s);
int launch_mean_loss(const float* y, const float* yh, long n, int squared, float* loss, float* d_yhat, hipStream_t s);

// per-launch timing records (vc_prof_*): kind 0 igemm, 1 wgrad, 2 wsconv
struct ProfScope {
    bool on; hipEvent_t a, b; double flops; int kind; hipStream_t s;
    ProfScope(int kind, double flops, hipStream_t st);
    ~ProfScope();
};

// Deterministic mode (vc_plan_set_deterministic / vc_set_deterministic): every reduction of the launches made while it is
// active has a fixed order -- no split-K across workgroups (K groups inside the workgroup only), no atomics that meet
// another contribution in scheduling order (bias / gamma / beta / Cin = 1 kernels reduce over the batch inside one
// thread).  Thread-local: a plan-level entry point activates its plan's setting for the launches it issues (DetScope),
// operator-level calls use the calling thread's default.
bool vc_det_active();
struct DetScope { int saved; explicit DetScope(int det); ~DetScope(); };
// test / tuning hooks, thread-local (the calling thread's following launches): forced tile configuration of the fp32
// filter-gradient kernel (tile columns 64 / 128, K groups 1 / 2 / 4, split-K); 0 = automatic
void vc_set_error(const char* fmt, ...);
#define VC_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    vc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    return (int)_e; } } while (0)

// ---- test / tuning switches (include/vc_cyclegan_testing.h).  ALL of them live in one thread-local object owned by vc_testing.hip, the only
// translation unit that writes it; the product translation units read it through vc_hooks().  vc_testing_hooks_active() tells a caller
// (bench.py, model.CycleGAN.train) that the calling thread runs with every switch at its production value.
struct TestHooks {
    int igemm_bm = 0, igemm_kg = 0, igemm_sk = 0;      // vc_tune_igemm_override
    int wgrad_bn = 0, wgrad_kg = 0, wgrad_sk = 0;      // vc_tune_wgrad_override
    int bf16_bm = 0, bf16_nst = 0, bf16_sk = 0;        // vc_tune_conv_bf16_override
    int bf16_spec = -1;                                // vc_tune_conv_bf16_spec (-1: the process default)
    int pixel_octets = 1;                              // vc_tune_bf16_pixel_octets
    int rowskip_wgrad = 1;                             // vc_tune_rowskip
    int wgrad_store = 0;                               // vc_tune_wgrad_store
    int glu_flags = 0;                                 // vc_tune_glu
    int c1_gather = 0;                                 // vc_tune_c1_dgrad_gather
    int layer_updates = 0;                             // vc_tune_layer_updates
    int winograd = 1;                                  // vc_tune_winograd
};
const TestHooks& vc_hooks();
void vc_count_wino_gemm(int kind, double saved_flops);   // vc_testing_wino_gemms / _saved_flops: one more Winograd-domain launch of the calling thread (kind 0 fwd / dgrad, 1 wgrad)
