// vc_conv_wino.hip -- Toom-Cook / Winograd transforms around the convolutions that have stride 1 and >= 3 taps along an axis (fp32 mode):
//   F(4,3)  the generators' residual blocks   (reference module.py:66-83:   conv1d, 3 taps)          6 multiplies per output quad instead of 12
//   F(4,5)  their up-sampling blocks          (module.py:115-133:           conv1d, 5 taps)          8 instead of 20
//   F(3,6)  the discriminators' last block    (module.py:208: 6 x 3 kernel, strides 1 x 2) ALONG H   8 row planes per 3 output rows, each a 1 x 3 stride-2 conv
//
//     y[4i .. 4i+3] = A^T [ (G w) . (B^T x[4i-1 .. 4i+4]) ]       (Lavin & Gray 2016; points 0, +-1, +-2, (+-1/2,) inf; matrices in vc_wino.h)
//
// The convolution becomes taps + 3 (8) independent channel GEMMs over a quarter (third) of the pixels each, launched as ONE grouped launch on the
// implicit-GEMM kernel (vc_conv_igemm.hip) -- exact fp32 products, fp32 accumulation; the transforms add the rounding of a few additions (against
// the fp64 direct convolution at the layers' sizes: 1.2-3e-6 of the output's maximum, the direct fp32 kernel 0.2-1.2e-6; the operator tests hold
// both to 1e-4).  The input gradient is the adjoint of the same three steps (dx = B [ U^T (A dy) ]) on the same transformed filters U = G w; the
// 1-D layers' filter gradient is wino_wgrad_kernel below (dU[p] per position, folded into the taps in registers); D.d3's stays direct.
//
// Layouts (1-D, W = 4 nt):  x [g][b][C][W];  V = B^T x [g][p][b][C][nt];  U [g][p][Cin][Cout];  M = U V [g][p][b][Cout][nt].
// Along H (D.d3, H = 3 nt): x [g][b][C][H][W];  V [g][p][b][C][nt][W];  U [g][p][kw][Cin][Cout];  M [g][p][b][Cout][nt][OW].
#include "vc_internal.h"
#include "vc_device_util.h"
#include "vc_wino.h"
#include <algorithm>

struct WinoArgs {
    const float* src; float* dst; const float* bias; const float* addend;
    long src_gs, dst_gs, bias_gs, add_gs;      // group strides (floats); for the transformed side: stride of one g (6 planes inside)
    long plane;                                // floats of one transform plane: B * C * nt
    int G, B, C, nt, accumulate;
    int rows;                                  // G * B * C
    int taps;                                  // 3: F(4,3), six planes; 5: F(4,5), eight planes
};

// a thread = one tile of one (g, b, c) row; consecutive threads = consecutive tiles, then consecutive channels
__device__ __forceinline__ bool wino_row(const WinoArgs& p, int& g, long& row_in_g, int& i) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t / p.nt; i = (int)(t - row * p.nt);
    if (row >= p.rows) return false;
    const long per_g = (long)p.B * p.C;
    g = (int)(row / per_g); row_in_g = row - g * per_g;
    return true;
}

// V[p] = B^T d, d = x[4i-1 .. 4i+4] (zero beyond the row)
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoArgs p) {
    int g, i; long r;
    if (!wino_row(p, g, r, i)) return;
    const float* x = p.src + (long)g * p.src_gs + r * (4 * p.nt) + 4 * i;
    const f32x4 c = *reinterpret_cast<const f32x4*>(x);
    float* v = p.dst + (long)g * p.dst_gs + r * p.nt + i;
    if (p.taps == 5) {
        const float d[8] = {i > 0 ? x[-2] : 0.f, i > 0 ? x[-1] : 0.f, c[0], c[1], c[2], c[3], i + 1 < p.nt ? x[4] : 0.f, i + 1 < p.nt ? x[5] : 0.f};
        wino5_store_input(d, v, p.plane);
    } else wino_store_input(c, i > 0 ? x[-1] : 0.f, i + 1 < p.nt ? x[4] : 0.f, v, p.plane);
}

// y[4i .. 4i+3] = A^T M + bias
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoArgs p) {
    int g, i; long r;
    if (!wino_row(p, g, r, i)) return;
    const int c = (int)(r % p.C);
    const float b = p.bias ? p.bias[(long)g * p.bias_gs + c] : 0.f;
    const float* m = p.src + (long)g * p.src_gs + r * p.nt + i;
    *reinterpret_cast<f32x4*>(p.dst + (long)g * p.dst_gs + r * (4 * p.nt) + 4 * i) = p.taps == 5 ? wino5_load_output(m, p.plane, b) : wino_load_output(m, p.plane, b);
}

// dM[p] = A dy[4i .. 4i+3]
__global__ __launch_bounds__(256) void wino_dy_kernel(const WinoArgs p) {
    int g, i; long r;
    if (!wino_row(p, g, r, i)) return;
    const f32x4 d = *reinterpret_cast<const f32x4*>(p.src + (long)g * p.src_gs + r * (4 * p.nt) + 4 * i);
    float* m = p.dst + (long)g * p.dst_gs + r * p.nt + i;
    if (p.taps == 5) wino5_store_dy(d, m, p.plane); else wino_store_dy(d, m, p.plane);
}

// dx[4i .. 4i+3] (+)= B dV (+ addend: the skip path of a residual block head)
__global__ __launch_bounds__(256) void wino_dx_kernel(const WinoArgs p) {
    int g, i; long r;
    if (!wino_row(p, g, r, i)) return;
    const float* v = p.src + (long)g * p.src_gs + r * p.nt + i;
    f32x4 dx = p.taps == 5 ? wino5_load_dx(v, p.plane, i > 0, i + 1 < p.nt) : wino_load_dx(v, p.plane, i > 0, i + 1 < p.nt);
    const long o = r * (4 * p.nt) + 4 * i;
    if (p.addend) dx += *reinterpret_cast<const f32x4*>(p.addend + (long)g * p.add_gs + o);
    float* dst = p.dst + (long)g * p.dst_gs + o;
    if (p.accumulate) dx += *reinterpret_cast<const f32x4*>(dst);
    *reinterpret_cast<f32x4*>(dst) = dx;
}

// U[p][cin][cout] = G w[tap][cin][cout]; a thread = 4 consecutive output channels of one input channel
struct WinoFilterJob { const float* w; float* U; long w_gs, U_gs; int n4, taps; };      // n4 = Cin * Cout / 4
#define VC_WINO_JOBS 16
struct WinoFilterJobs { WinoFilterJob j[VC_WINO_JOBS]; int n, G; unsigned blk0[VC_WINO_JOBS + 1]; };
__global__ __launch_bounds__(256) void wino_filter_kernel(const WinoFilterJobs jj) {
    int k = 0;
    while (k + 1 < jj.n && blockIdx.x >= jj.blk0[k + 1]) ++k;
    const WinoFilterJob& j = jj.j[k];
    const int t = (int)(blockIdx.x - jj.blk0[k]) * 256 + threadIdx.x;
    if (t >= j.n4) return;
    const int g = blockIdx.y;
    const long plane = (long)j.n4 * 4;
    const float* w = j.w + (long)g * j.w_gs + (long)t * 4;
    if (j.taps == 6) {        // F(3,6) along H: a plane = one row tap's [kw][Cin][Cout]; U[p] = sum_dh G6[p][dh] w[dh]
        f32x4 wk[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) wk[k] = *reinterpret_cast<const f32x4*>(w + k * plane);
        float* u6 = j.U + (long)g * j.U_gs + (long)t * 4;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 6; ++k) if (kG6[p][k] != 0.f) a += wk[k] * kG6[p][k];
            *reinterpret_cast<f32x4*>(u6 + p * plane) = a;
        }
        return;
    }
    if (j.taps == 5) {        // U[p] = sum_k G5[p][k] w[k]
        f32x4 wk[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) wk[k] = *reinterpret_cast<const f32x4*>(w + k * plane);
        float* u5 = j.U + (long)g * j.U_gs + (long)t * 4;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 5; ++k) if (kG5[p][k] != 0.f) a += wk[k] * kG5[p][k];
            *reinterpret_cast<f32x4*>(u5 + p * plane) = a;
        }
        return;
    }
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(w), g1 = *reinterpret_cast<const f32x4*>(w + plane), g2 = *reinterpret_cast<const f32x4*>(w + 2 * plane);
    float* u = j.U + (long)g * j.U_gs + (long)t * 4;
    const f32x4 s = (g0 + g2) * (-1.f / 6.f), q = g1 * (1.f / 6.f);
    const f32x4 e = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), h = g1 * (1.f / 12.f);
    *reinterpret_cast<f32x4*>(u) = g0 * 0.25f;
    *reinterpret_cast<f32x4*>(u + plane) = s - q;
    *reinterpret_cast<f32x4*>(u + 2 * plane) = s + q;
    *reinterpret_cast<f32x4*>(u + 3 * plane) = e + h;
    *reinterpret_cast<f32x4*>(u + 4 * plane) = e - h;
    *reinterpret_cast<f32x4*>(u + 5 * plane) = g2;
}

// up to four tile transforms in one launch (blockIdx.y = job): the operands of a filter gradient in the Winograd domain -- both applications'
// inputs (B^T x) and output gradients (A dy)
struct WinoPrep { WinoArgs a[4]; int kind[4]; int n; };      // kind 0: input transform, 1: dy transform
__global__ __launch_bounds__(256) void wino_prep_kernel(const WinoPrep pp) {
    const WinoArgs& p = pp.a[blockIdx.y];
    int g, i; long r;
    if (!wino_row(p, g, r, i)) return;
    const float* x = p.src + (long)g * p.src_gs + r * (4 * p.nt) + 4 * i;
    const f32x4 c = *reinterpret_cast<const f32x4*>(x);
    float* d = p.dst + (long)g * p.dst_gs + r * p.nt + i;
    if (p.taps == 5) {
        if (pp.kind[blockIdx.y] == 0) {
            const float dd[8] = {i > 0 ? x[-2] : 0.f, i > 0 ? x[-1] : 0.f, c[0], c[1], c[2], c[3], i + 1 < p.nt ? x[4] : 0.f, i + 1 < p.nt ? x[5] : 0.f};
            wino5_store_input(dd, d, p.plane);
        } else wino5_store_dy(c, d, p.plane);
    } else if (pp.kind[blockIdx.y] == 0) wino_store_input(c, i > 0 ? x[-1] : 0.f, i + 1 < p.nt ? x[4] : 0.f, d, p.plane);
    else wino_store_dy(c, d, p.plane);
}

// Filter gradient in the Winograd domain: dU[p] = sum over tiles of V[p] dM[p]^T (six channel GEMMs over a quarter of the pixels each), reduced
// to the three taps in registers: dw[k] = sum_p G[p][k] dU[p].  One wave = a 32 x 32 (cin x cout) tile of all three taps; no LDS: the operands
// are [p][b][channel][tile] with 8 k tiles of a row contiguous, so a lane's 16-byte load IS its four k values of four MFMA steps (lanes 0-31
// take tiles 0-3 of the octet, lanes 32-63 tiles 4-7 -- any k order is right as long as both operands use the same one), and a wave's load is
// one contiguous 1 KB run.  Loads run VC_WW_AHEAD (8, or 4 when the iteration count is no multiple of 8) iterations ahead of their multiplies.  Fixed summation order (no atomics, no split-K).
struct WinoWgradArgs {
    const float* V[2]; const float* dM[2]; long V_gs[2], dM_gs[2]; int B[2]; int napp;
    float* dW; long dw_gs; int Cin, Cout, nt, store;
};
__device__ const float kG3dev[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6.f, -1.f / 6.f, -1.f / 6.f}, {-1.f / 6.f, 1.f / 6.f, -1.f / 6.f},
                                        {1.f / 24.f, 1.f / 12.f, 1.f / 6.f}, {1.f / 24.f, -1.f / 12.f, 1.f / 6.f}, {0.f, 0.f, 1.f}};
__device__ const float kG5dev[8][5] = {{1.f, 0.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f, 1.f}, {1.f, 2.f, 4.f, 8.f, 16.f}, {1.f, -2.f, 4.f, -8.f, 16.f},
                                        {1.f, 0.5f, 0.25f, 0.125f, 0.0625f}, {1.f, -0.5f, 0.25f, -0.125f, 0.0625f}, {0.f, 0.f, 0.f, 0.f, 1.f}};
template <int VC_WW_AHEAD, int TAPS>
__global__ __launch_bounds__(256) void wino_wgrad_kernel(const WinoWgradArgs a) {
    constexpr int NP = TAPS + 3;                                     // transform positions
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    const int ctiles = a.Cout >> 6;                                  // 64 x 64 per workgroup: waves 2 x 2
    const int tm = blockIdx.x / ctiles, tn = blockIdx.x - tm * ctiles;
    const int cin0 = tm * 64 + (wave >> 1) * 32, cout0 = tn * 64 + (wave & 1) * 32;
    const int g = blockIdx.y;
    const int oct = a.nt >> 3;                                       // k octets (8 tiles) per row
    f32x16 acc[TAPS], T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        T[r] = 0.f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc[k][r] = 0.f;
    }
    // fold the finished transform position p into the taps: acc[k] += G[p][k] T
    auto fold = [&](int p) {
        float gk[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) gk[k] = TAPS == 3 ? kG3dev[p][k] : kG5dev[p][k];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int k = 0; k < TAPS; ++k) acc[k][r] += gk[k] * T[r];
            T[r] = 0.f;
        }
    };
    for (int app = 0; app < a.napp; ++app) {
        // one application: iterations = (p, sample, octet) with p slowest -- the operands' own memory order, so both pointers walk by constant
        // steps straight through the six transform positions; B oct is a multiple of VC_WW_AHEAD (the launcher checks)
        const int per_p = a.B[app] * oct;
        const float* pa = a.V[app] + (long)g * a.V_gs[app] + (long)(cin0 + l31) * a.nt + h * 4;
        const float* pb = a.dM[app] + (long)g * a.dM_gs[app] + (long)(cout0 + l31) * a.nt + h * 4;
        const long ja = (long)a.Cin * a.nt - 8 * (oct - 1), jb = (long)a.Cout * a.nt - 8 * (oct - 1);   // step from a row's last octet to the next sample's first
        int o = 0;
        auto advance = [&]() { if (++o == oct) { o = 0; pa += ja; pb += jb; } else { pa += 8; pb += 8; } };
        f32x4 av[VC_WW_AHEAD], bv[VC_WW_AHEAD];
#pragma unroll
        for (int d = 0; d < VC_WW_AHEAD; ++d) { av[d] = *reinterpret_cast<const f32x4*>(pa); bv[d] = *reinterpret_cast<const f32x4*>(pb); advance(); }
        for (int p = 0; p < NP; ++p) {
            const int nc = p < NP - 1 ? per_p : per_p - VC_WW_AHEAD;       // (the last VC_WW_AHEAD iterations of the application have nothing to fetch: below)
            for (int c = 0; c < nc; c += VC_WW_AHEAD) {
#pragma unroll
                for (int d = 0; d < VC_WW_AHEAD; ++d) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) T = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d][e], bv[d][e], T, 0, 0, 0);
                    av[d] = *reinterpret_cast<const f32x4*>(pa); bv[d] = *reinterpret_cast<const f32x4*>(pb); advance();
                }
            }
            if (p < NP - 1) fold(p);
        }
#pragma unroll
        for (int d = 0; d < VC_WW_AHEAD; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e) T = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d][e], bv[d][e], T, 0, 0, 0);
        fold(NP - 1);
    }
    float* dw = a.dW + (long)g * a.dw_gs;
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cin = cin0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float* q = dw + ((long)k * a.Cin + cin) * a.Cout + cout0 + l31;
            *q = a.store ? acc[k][r] : *q + acc[k][r];
        }
}

// ---- F(3,6) along H (vc_wino.h).  x [g][b][C][H][W], H = 3 nt; transform-domain tensors [g][p][b][C][nt][W] (eight planes).
// A thread = 4 consecutive columns of one tile of one (g, b, c) image.
struct WinoHArgs {
    const float* src; float* dst; const float* bias; const float* addend;
    long src_gs, dst_gs, bias_gs, add_gs, plane;      // plane = B * C * nt * W floats
    int G, B, C, nt, W, accumulate, mode;             // mode 0: x -> V, 1: M -> y (+ bias), 2: dy -> dM, 3: dV -> dx
    long items;                                       // mode 0..2: G * B * C * nt * W / 4; mode 3: G * B * C * H * W / 4
};
__global__ __launch_bounds__(256) void winoh_kernel(const WinoHArgs p) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.items) return;
    const int W4 = p.W >> 2, H = 3 * p.nt;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (p.mode == 3) {          // dx row r = sum over the tiles that read it: q = r - 3 t + 2 in 0 .. 7
        const int wq = (int)(t % W4); long u = t / W4;
        const int r = (int)(u % H); u /= H;                    // u = (g, b, c) image
        const long per_g = (long)p.B * p.C; const int g = (int)(u / per_g); const long img = u - g * per_g;
        const float* v = p.src + (long)g * p.src_gs + img * p.nt * p.W + 4 * wq;
        f32x4 a = zero;
        for (int tt = 0; tt < p.nt; ++tt) {
            const int q = r - 3 * tt + 2;
            if (q < 0 || q > 7) continue;
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                float c = 0.f;
#pragma unroll
                for (int qq = 0; qq < 8; ++qq) if (qq == q) c = kBT5[pp][qq];
                if (c != 0.f) a += *reinterpret_cast<const f32x4*>(v + pp * p.plane + (long)tt * p.W) * c;
            }
        }
        const long o = (img * H + r) * p.W + 4 * wq;
        if (p.addend) a += *reinterpret_cast<const f32x4*>(p.addend + (long)g * p.add_gs + o);
        float* d = p.dst + (long)g * p.dst_gs + o;
        if (p.accumulate) a += *reinterpret_cast<const f32x4*>(d);
        *reinterpret_cast<f32x4*>(d) = a;
        return;
    }
    const int wq = (int)(t % W4); long u = t / W4;
    const int tt = (int)(u % p.nt); u /= p.nt;
    const long per_g = (long)p.B * p.C; const int g = (int)(u / per_g); const long img = u - g * per_g;
    const long tile = (img * p.nt + tt) * p.W + 4 * wq;          // offset inside one transform plane
    if (p.mode == 0) {          // V[p] = B^T x[3t-2 .. 3t+5]
        const float* x = p.src + (long)g * p.src_gs + img * H * p.W + 4 * wq;
        f32x4 d[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int r = 3 * tt - 2 + q; d[q] = (r >= 0 && r < H) ? *reinterpret_cast<const f32x4*>(x + (long)r * p.W) : zero; }
        float* v = p.dst + (long)g * p.dst_gs + tile;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            f32x4 a = zero;
#pragma unroll
            for (int q = 0; q < 8; ++q) if (kBT5[pp][q] != 0.f) a += d[q] * kBT5[pp][q];
            *reinterpret_cast<f32x4*>(v + pp * p.plane) = a;
        }
    } else if (p.mode == 1) {   // y[3t + m] = A^T M + bias
        const float* m = p.src + (long)g * p.src_gs + tile;
        f32x4 mm[8];
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) mm[pp] = *reinterpret_cast<const f32x4*>(m + pp * p.plane);
        const int c = (int)(img % p.C);
        const float b = p.bias ? p.bias[(long)g * p.bias_gs + c] : 0.f;
        float* y = p.dst + (long)g * p.dst_gs + img * H * p.W + 4 * wq;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            f32x4 a = {b, b, b, b};
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) if (kAT6[e][pp] != 0.f) a += mm[pp] * kAT6[e][pp];
            *reinterpret_cast<f32x4*>(y + (long)(3 * tt + e) * p.W) = a;
        }
    } else {                    // dM[p] = A dy[3t .. 3t+2]
        const float* y = p.src + (long)g * p.src_gs + img * H * p.W + 4 * wq;
        f32x4 d[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) d[e] = *reinterpret_cast<const f32x4*>(y + (long)(3 * tt + e) * p.W);
        float* m = p.dst + (long)g * p.dst_gs + tile;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            f32x4 a = zero;
#pragma unroll
            for (int e = 0; e < 3; ++e) if (kAT6[e][pp] != 0.f) a += d[e] * kAT6[e][pp];
            *reinterpret_cast<f32x4*>(m + pp * p.plane) = a;
        }
    }
}

static int launch_wino(void (*k)(WinoArgs), WinoArgs a, hipStream_t s) {
    a.rows = a.G * a.B * a.C;
    a.plane = (long)a.B * a.C * a.nt;
    const long threads = (long)a.rows * a.nt;
    if (threads <= 0) return 0;
    k<<<dim3((unsigned)((threads + 255) / 256)), 256, 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}


// x [G][B][C][4 nt] -> V [G][6][B][C][nt]
int launch_wino_input(const float* x, long x_gs, float* V, int G, int B, int C, int nt, hipStream_t s, int taps) {
    WinoArgs a{}; a.taps = taps; a.src = x; a.src_gs = x_gs; a.dst = V; a.dst_gs = (long)(taps + 3) * B * C * nt; a.G = G; a.B = B; a.C = C; a.nt = nt;
    return launch_wino(wino_input_kernel, a, s);
}
// M [G][6][B][C][nt] -> y [G][B][C][4 nt] (+ bias)
int launch_wino_output(const float* M, const float* bias, long b_gs, float* y, long y_gs, int G, int B, int C, int nt, hipStream_t s, int taps) {
    WinoArgs a{}; a.taps = taps; a.src = M; a.src_gs = (long)(taps + 3) * B * C * nt; a.dst = y; a.dst_gs = y_gs; a.bias = bias; a.bias_gs = b_gs; a.G = G; a.B = B; a.C = C; a.nt = nt;
    return launch_wino(wino_output_kernel, a, s);
}
// dy [G][B][C][4 nt] -> dM [G][6][B][C][nt]
int launch_wino_dy(const float* dy, long dy_gs, float* dM, int G, int B, int C, int nt, hipStream_t s, int taps) {
    WinoArgs a{}; a.taps = taps; a.src = dy; a.src_gs = dy_gs; a.dst = dM; a.dst_gs = (long)(taps + 3) * B * C * nt; a.G = G; a.B = B; a.C = C; a.nt = nt;
    return launch_wino(wino_dy_kernel, a, s);
}
// dV [G][6][B][C][nt] -> dx [G][B][C][4 nt] = (accumulate ? dx : 0) + addend + B dV
int launch_wino_dx(const float* dV, float* dx, long dx_gs, const float* addend, long add_gs, int accumulate, int G, int B, int C, int nt, hipStream_t s, int taps) {
    WinoArgs a{}; a.taps = taps; a.src = dV; a.src_gs = (long)(taps + 3) * B * C * nt; a.dst = dx; a.dst_gs = dx_gs; a.addend = addend; a.add_gs = add_gs; a.accumulate = accumulate;
    a.G = G; a.B = B; a.C = C; a.nt = nt;
    return launch_wino(wino_dx_kernel, a, s);
}
// filters w [G][taps][Cin][Cout] -> U [G][taps + 3][Cin][Cout], several layers per launch
int launch_wino_filters(const std::vector<WinoFilter>& f, int G, hipStream_t s) {
    for (size_t k0 = 0; k0 < f.size(); k0 += VC_WINO_JOBS) {
        WinoFilterJobs jj{}; jj.G = G; unsigned blk = 0;
        for (size_t k = k0; k < f.size() && k < k0 + VC_WINO_JOBS; ++k) {
            const long n = (long)f[k].Cin * f[k].Cout;
            if (n % 4 != 0 || n / 4 >= (1L << 30)) { vc_set_error("winograd filter transform: Cin * Cout must be a multiple of 4"); return -1; }
            WinoFilterJob& j = jj.j[jj.n]; j.w = f[k].w; j.U = f[k].U; j.w_gs = f[k].w_gs; j.U_gs = (long)(f[k].taps == 6 ? 8 : f[k].taps + 3) * n; j.n4 = (int)(n / 4); j.taps = f[k].taps;
            jj.blk0[jj.n++] = blk; blk += (unsigned)((j.n4 + 255) / 256);
        }
        jj.blk0[jj.n] = blk;
        wino_filter_kernel<<<dim3(blk, (unsigned)G), 256, 0, s>>>(jj);
        VC_HIP(hipGetLastError());
    }
    return 0;
}

// the four operands of a Winograd-domain filter gradient in one launch: napp applications' x -> V and dy -> dM (all [G][6][B][C][nt])
int launch_wino_prep(const WinoPrepJob* jobs, int n, int G, int nt, hipStream_t s, int taps) {
    WinoPrep pp{}; pp.n = n; long maxthreads = 0;
    for (int k = 0; k < n; ++k) {
        WinoArgs& a = pp.a[k]; a = WinoArgs{};
        a.taps = taps; a.src = jobs[k].src; a.src_gs = jobs[k].src_gs; a.dst = jobs[k].dst; a.dst_gs = (long)(taps + 3) * jobs[k].B * jobs[k].C * nt;
        a.G = G; a.B = jobs[k].B; a.C = jobs[k].C; a.nt = nt; a.rows = G * jobs[k].B * jobs[k].C; a.plane = (long)jobs[k].B * jobs[k].C * nt;
        pp.kind[k] = jobs[k].kind;
        maxthreads = std::max(maxthreads, (long)a.rows * nt);
    }
    if (maxthreads <= 0) return 0;
    wino_prep_kernel<<<dim3((unsigned)((maxthreads + 255) / 256), (unsigned)n), 256, 0, s>>>(pp);
    VC_HIP(hipGetLastError());
    return 0;
}
// dW [G][3][Cin][Cout] (+)= the filter gradient of napp applications from their transform-domain operands
int launch_wino_wgrad(const float* const* V, const float* const* dM, const int* B, int napp, float* dW, long dw_gs, int store,
                      int G, int Cin, int Cout, int nt, hipStream_t s, double flops, int taps) {
    WinoWgradArgs a{};
    for (int k = 0; k < napp; ++k) { a.V[k] = V[k]; a.dM[k] = dM[k]; a.B[k] = B[k]; a.V_gs[k] = (long)(taps + 3) * B[k] * Cin * nt; a.dM_gs[k] = (long)(taps + 3) * B[k] * Cout * nt; }
    a.napp = napp; a.dW = dW; a.dw_gs = dw_gs; a.Cin = Cin; a.Cout = Cout; a.nt = nt; a.store = store;
    ProfScope prof(1, flops, s);
    bool by8 = true;
    for (int k = 0; k < napp; ++k) {
        if ((B[k] * (nt / 8)) % 4 != 0) { vc_set_error("winograd filter gradient: samples x k octets must be a multiple of 4"); return -1; }
        by8 = by8 && (B[k] * (nt / 8)) % 8 == 0;
    }
    const dim3 grid((unsigned)((Cin / 64) * (Cout / 64)), (unsigned)G);
    if (taps == 5) { if (by8) wino_wgrad_kernel<8, 5><<<grid, 256, 0, s>>>(a); else wino_wgrad_kernel<4, 5><<<grid, 256, 0, s>>>(a); }
    else if (by8) wino_wgrad_kernel<8, 3><<<grid, 256, 0, s>>>(a); else wino_wgrad_kernel<4, 3><<<grid, 256, 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}

// F(3,6) along H: mode 0 x [G][B][C][3 nt][W] -> V [G][8][B][C][nt][W]; 1 M -> y (+ bias); 2 dy -> dM; 3 dV -> dx = (accumulate ? dx : 0) + addend + B dV
int launch_winoh(int mode, const float* src, long src_gs, float* dst, long dst_gs, const float* bias, long b_gs, const float* addend, long add_gs,
                 int accumulate, int G, int B, int C, int nt, int W, hipStream_t s) {
    WinoHArgs a{}; a.mode = mode; a.src = src; a.dst = dst; a.bias = bias; a.bias_gs = b_gs; a.addend = addend; a.add_gs = add_gs; a.accumulate = accumulate;
    a.G = G; a.B = B; a.C = C; a.nt = nt; a.W = W; a.plane = (long)B * C * nt * W;
    const long planes_gs = 8 * a.plane;
    a.src_gs = (mode == 1 || mode == 3) ? planes_gs : src_gs;
    a.dst_gs = (mode == 0 || mode == 2) ? planes_gs : dst_gs;
    a.items = (long)G * B * C * (mode == 3 ? 3 * nt : nt) * (W / 4);
    if (a.items <= 0) return 0;
    winoh_kernel<<<dim3((unsigned)((a.items + 255) / 256)), 256, 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}
