// Internal launch descriptors shared by vc_kernels.hip (device code + launchers)
// and vc_api.hip (host orchestration + C-ABI).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One entry per GEMM-K row of an implicit-GEMM convolution.
//   c     : channel of the gathered tensor this K row reads
//   dh,dw : offset (in gathered-tensor pixels) added to the launch pixel (i*si, j*sj)
//   a_off : float offset of this K row inside the weight tensor, -1 = padding row
struct __attribute__((aligned(16))) KEntry { int c, dh, dw, a_off; };

// Y[g][b][m][P(i,j)] (+)= sum_kk A[a_off(kk) + m*a_ms] * X[g][b][c(kk)][i*si+dh(kk)][j*sj+dw(kk)]
// Gather-table entry of one GEMM-K row, read through the scalar cache:
//   soff   : byte offset (c*xHW + dh*xW + dw)*4 added to the lane's pixel byte offset
//   tapbit : 1 << tap index; 0 marks a padding row.  A lane keeps a bitmask of the taps that fall inside the
//            gathered tensor for ITS pixel, so the per-element bounds check is one AND.
struct __attribute__((aligned(8))) XEntry { int soff; unsigned tapbit; };
struct Tap { int dh, dw; };
#define VC_OOB 0xffffffffu     // buffer-load offset beyond num_records: the hardware returns 0

// Y[g][b][m][P(i,j)] (+)= sum_kk A[a_off(kk) + m*a_ms] * X[g][b][c(kk)][i*si+dh(kk)][j*sj+dw(kk)]
struct IgemmArgs {
    const float* A; const float* X; float* Y; const float* bias; const float* addend;
    const XEntry* xtab;                           // nchunks*32 entries
    const Tap* taps; int ntaps;                   // <= 32 taps
    const unsigned* atab;                         // transposed-A mode: byte offset of each K row (0x80000000 = padding)
    unsigned a_ks_bytes;                          // plain-A mode: byte stride between K rows
    unsigned x_bytes, a_bytes;                    // buffer sizes (per group) for the hardware range check
    long a_gs, x_gs, y_gs, bias_gs, add_gs;      // signed group strides (floats)
    long x_bs, y_bs;                              // batch strides (floats)
    int krows;                                    // K rows of the table (padded to a multiple of 64)
    int kc;                                       // K chunk chosen by igemm_config (32 or 64)
    int M, a_ms;                                  // output rows, stride of m in A
    int Hl, Wl;                                   // bounds of the gathered tensor
    int xHW, xW;                                  // channel / row stride of the gathered tensor
    int si, sj;                                   // pixel stride launch grid -> gathered tensor
    int NPB, OWc, Ptot;                           // launch pixel grid: per sample, width, total
    int yHW, yW;                                  // channel / row stride of the output tensor
    int so_h, so_w, o_h0, o_w0;                   // launch pixel -> output pixel
    int splitk, accumulate, mtiles, ntiles, groups, big;
    int kl_ok;                                    // the k-contiguous-LDS kernel may serve this launch (transposed A: K rows come in aligned runs of 4)
};

// Up to 4 launches of the same tile configuration fused into one grid (blockIdx.y): the stride-parity classes of an
// input gradient, which individually underfill the chip.
struct IgemmMulti { IgemmArgs c[4]; int n; };

// dW[a_off(kk) + co] += sum_{b,i,j} X[g][b][c(kk)][i*si+dh][j*sj+dw] * dY[g][b][co][i][j]
struct WgradArgs {
    const float* X; const float* dY; float* dW; const KEntry* ktab;
    long x_gs, dy_gs, dw_gs; long x_bs, dy_bs;
    unsigned x_bytes, dy_bytes;
    float inv_ow;                                 // 1/OWc for the pixel decode
    int N;                                        // Cout
    int Hl, Wl, xHW, xW, si, sj;
    int NPB, OWc, Ptot;
    int nchunks;                                  // ceil(Ptot/32)
    int splitk, ktiles, ntiles, groups;
    int first_touch;                              // dW holds no earlier contribution (it is zero): store instead of add
};

// Single-input-channel convolution (the discriminator's first layer, module.py:201-202: Cin = 1, 3x3 -> K = 9): a
// bandwidth job, done on the vector ALU instead of padding K = 9 up to an MFMA tile.  kh*kw <= 9.
struct C1Args {
    const float* x; const float* w; const float* bias; const float* addend; const float* dy;
    float* y; float* dx; float* dw;
    long x_gs, w_gs, b_gs, y_gs, dx_gs, add_gs, dw_gs;     // signed group strides (floats); y_gs also strides dy
    int B, H, W, Cout, kh, kw, sh, sw, OH, OW, pt, pl, G;
    int math;                                               // VC_MATH_*: bf16 mode rounds both operands like the MFMA kernels do
};
int launch_c1_fwd(const C1Args& a, hipStream_t s);
int launch_c1_dgrad(const C1Args& a, int accumulate, hipStream_t s);   // dx (+)= ...; atomics into dx (zeroed here unless accumulate)
int launch_c1_wgrad(const C1Args& a, hipStream_t s);                   // dw += ... (atomics)
int vc_math_mode();

// Weight-stationary 1-D convolution forward for few input channels (the generator's h1 layer, module.py:161-162:
// 24 -> 2x128 channels, k = 15, stride 1): K = Cin*k = 360 is too short a loop for the tiled implicit GEMM (12 chunks
// per workgroup, mostly prologue/epilogue).  Here a workgroup keeps a 32-row filter slab in LDS and streams pixel tiles
// through it; the k-fold window expansion happens in LDS, so HBM sees every input frame once per slab.
struct WsConvArgs {
    const float* x; const float* w; const float* bias; float* y;
    long x_gs, w_gs, b_gs, y_gs;          // signed group strides (floats)
    int B, T, Cout, G;
};
int launch_wsconv1d(const WsConvArgs& a, int cin, int kw, hipStream_t s, double flops);   // returns -2 when the geometry is not built

struct NormArgs {
    const float* y; const float* res; const float* gamma; const float* beta;
    const float* dout; const float* stats_in;
    float* out; float* stats; float* dy; float* dgamma; float* dbeta; float* dbias;
    long y_gs, res_gs, par_gs, out_gs, st_gs, do_gs, dy_gs, dpar_gs, dbias_gs;
    int G, B, C, L, ps;
};

void igemm_config(IgemmArgs& a);   // picks tile size + split-K, fills mtiles/ntiles/splitk/big
int launch_igemm(const IgemmArgs& a, bool a_trans, hipStream_t s, double flops);
int launch_igemm_multi(const IgemmArgs* cls, int n, bool a_trans, hipStream_t s, double flops);   // all configured alike
int launch_wgrad(const WgradArgs& a, hipStream_t s, double flops);
#include <utility>
#include <vector>
int launch_zero_list(const std::vector<std::pair<float*, long>>& bufs, hipStream_t s);
int launch_norm_fwd(int mode, const NormArgs& a, hipStream_t s);
int launch_norm_bwd(int mode, const NormArgs& a, hipStream_t s);
int launch_bias_grad(const float* dy, long dy_gs, float* db, long db_gs, int G, int B, int C, int L, hipStream_t s);
int launch_head_fwd(const float* x, long x_gs, const float* w, const float* b, long par_gs,
                    float* o, long o_gs, int G, int B, int C, int HW, hipStream_t s);
int launch_head_bwd(const float* x, long x_gs, const float* w, long par_gs, const float* o, long o_gs,
                    const float* dout, long do_gs, float* dx, long dx_gs, float* dw, float* db,
                    long dpar_gs, int G, int B, int C, int HW, hipStream_t s);
int launch_adam(float* p, float* m, float* v, const float* g, long n, float alpha, float b1,
                float b2, float eps, float gscale, hipStream_t s);

struct LossArgs {
    const float* A; const float* Bm;      // real [B][F][T]
    const float* out1; long out1_gs;      // [2][2B][F][T]: g0 = A2B([A;B]) = [gen_B; id_B], g1 = B2A([B;A]) = [gen_A; id_A]
    const float* cyc;  long cyc_gs;       // [2][B][F][T]:  g0 = A2B(gen_A) = cycle_B,      g1 = B2A(gen_B) = cycle_A
    const float* dout; long dout_gs;      // [2][2B][HW]:   g0 = D_A([gen_A; A]),            g1 = D_B([gen_B; B])
    float* d_out1; long d_out1_gs;        // identity halves written, generation halves zeroed
    float* d_cyc;  long d_cyc_gs;
    float* d_dout_D; long d_dout_D_gs;    // [2][2B][HW] seeds of discriminator_loss
    float* d_dout_G; long d_dout_G_gs;    // [2][B][HW]  seeds of generator_loss_{B2A,A2B}
    float* acc;                           // 10 fp32 partial sums (zeroed by the launcher)
    float* losses;                        // 8 scalars
    int B, FT, HW; float lambda_cycle, lambda_identity;
};
int launch_loss(const LossArgs& a, hipStream_t s);
int launch_crop_batch(const float* src, const long* off, const int* len, const int* start, float* out,
                      int N, int F, int n_frames, hipStream_t s);
int launch_mean_loss(const float* y, const float* yh, long n, int squared, float* loss, float* d_yhat, hipStream_t s);

void vc_set_error(const char* fmt, ...);
#define VC_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    vc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    return (int)_e; } } while (0)
