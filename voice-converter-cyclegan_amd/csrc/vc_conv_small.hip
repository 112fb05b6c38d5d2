// vc_conv_small.hip -- the two first-layer convolutions of the reference networks, which are not GEMM-shaped:
//  wsconv1d_kernel : generator h1 as a plain convolution (module.py:161-162: 24 -> 2x128 channels, k = 15; the fused conv + GLU form is csrc/vc_conv_glu.hip): weight-stationary, window expansion in LDS
//  c1_*_kernel     : discriminator h1 (module.py:201-202: Cin = 1, 3x3 -> K = 9) on the vector ALU (HBM-bound)
#include "vc_internal.h"
#include "vc_device_util.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

// =====================================================================================
// wsconv1d_kernel: weight-stationary 1-D convolution forward (WsConvArgs), stride 1, TF-SAME, odd kernel width.
// Workgroup = 4 waves; LDS holds a 32-row filter slab Ws[m][k] (k contiguous, k = tap*CIN + c, zero-padded to a multiple
// of 16) for the whole kernel and one 128-pixel input tile Xs[pos][c] (c contiguous) at a time.  Wave w owns pixels
// 32w..32w+31 of the tile.  Operand fetch as in igemm_kl_kernel: per 8 k one ds_read_b128 for A and one for B (lane half h
// takes k = 8q + 4h .. +3), 4 MFMAs; a run of 4 consecutive k is 4 channels of one tap, i.e. 16 contiguous bytes of
// Xs row (pixel + tap), so the window expansion costs no instructions.  No split-K, no atomics, no zeroing.
// =====================================================================================
template <int CIN, int KW, int MATH>
__global__ __launch_bounds__(256, 2) void wsconv1d_kernel(const WsConvArgs a) {
    static_assert(CIN % 4 == 0 && (KW & 1), "runs of 4 channels; odd kernel width");
    constexpr int K = CIN * KW, KR = ((K + 15) / 16) * 16, NQ = KR / 8;
    constexpr int LDW = KR + 4;                       // row stride of the slab: 8 consecutive rows start 52 banks apart (conflict-free b128)
    constexpr int BN = 128, PADL = (KW - 1) / 2;
    constexpr int XR = BN + KW + 1;                   // rows 0 .. BN+KW-2 hold input frames; up to 2 more are touched by the zero-weight tail runs
    constexpr int CS = CIN + 4;                       // row stride of the input tile (conflict-free b128 across consecutive pixels)
    constexpr int NXF = (XR * CIN + 255) / 256;       // input floats staged per thread and tile
    __shared__ __attribute__((aligned(16))) float Ws[32 * LDW];
    __shared__ __attribute__((aligned(16))) float Xs[XR * CS];
    constexpr int LDY = BN + 4;
    __shared__ __attribute__((aligned(16))) float Ys[32 * LDY];   // output tile staged for 16-byte stores (rows of 512 B)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int g = blockIdx.z, slab = blockIdx.y;
    const int tps = (a.T + BN - 1) / BN, ntiles = a.B * tps;

    // ---- filter slab, once: Ws[m][k] = w[k][slab*32 + m].  Loads in batches of 8 before the LDS stores (issued one by one
    //      behind their own store, 46 dependent round trips cost 14 us per workgroup at B=256).
    {
        const float* w = a.w + (long)g * a.w_gs + slab * 32;
        const int m = tid & 31, mvalid = slab * 32 + m < a.Cout;
        constexpr int NW = (32 * KR + 255) / 256;       // (k, m) pairs per thread
#pragma unroll
        for (int i0 = 0; i0 < NW; i0 += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = (tid >> 5) + 8 * (i0 + i);
                const bool ok = (i0 + i < NW) && k < K && mvalid;
                v[i] = w[ok ? (long)k * a.Cout + m : 0];
                v[i] = ok ? v[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = (tid >> 5) + 8 * (i0 + i);
                if (i0 + i < NW && k < KR) Ws[m * LDW + k] = v[i];
            }
        }
    }
    // bias of the 16 rows this lane writes, once per workgroup
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = slab * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        bv[r] = (a.bias && m < a.Cout) ? a.bias[(long)g * a.b_gs + m] : 0.f;
    }
    const float* xg = a.x + (long)g * a.x_gs;
    float xr[NXF];
    unsigned xok = 0;                                // bit i: xr[i] is a frame of the tile (else padding -> 0); applied in store_tile
    auto load_tile = [&](int tile) {                 // input frames t0-PADL .. of all channels -> registers (lanes along time: coalesced)
        const int b = tile / tps, t0 = (tile - b * tps) * BN - PADL;
        const float* xb = xg + (long)b * CIN * a.T;
        xok = 0;
#pragma unroll
        for (int i = 0; i < NXF; ++i) {
            const int idx = tid + 256 * i, c = idx / XR, pos = idx - c * XR, t = t0 + pos;
            const bool ok = idx < XR * CIN && (unsigned)t < (unsigned)a.T;
            xr[i] = xb[ok ? c * a.T + t : 0];        // untouched until store_tile: the loads stay in flight during the MFMAs
            xok |= (unsigned)ok << i;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NXF; ++i) {
            const int idx = tid + 256 * i, c = idx / XR, pos = idx - c * XR;
            if (idx < XR * CIN) Xs[pos * CS + c] = ((xok >> i) & 1u) ? xr[i] : 0.f;
        }
    };
    const f32x4* ap = reinterpret_cast<const f32x4*>(Ws + l31 * LDW + 4 * h);
    const float* xrow = Xs + (wid * 32 + l31) * CS;
    const f32x4* bp4 = reinterpret_cast<const f32x4*>(xrow + 4 * h);     // lane half 1 reads the next run: +4 channels ...
    const f32x4* bp8 = reinterpret_cast<const f32x4*>(xrow + 8 * h);     // ... or, when that wraps into the next tap, + CS + 4 - CIN = 8 floats
    static_assert(CS + 4 - CIN == 8, "tap wrap offset");

    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                              // the previous tile's readers are done (first trip: the slab writes are ordered below)
        store_tile();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);   // next tile's frames are in flight during the MFMAs

        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        // Fragment ring, RD groups of 8 k deep: the reads of group q + RD are issued before the MFMAs of group q and pinned
        // there (left alone the compiler issues each read right before its use and exposes the LDS latency every 8 MFMAs:
        // 59 % -> 7x % MFMA pipe utilisation).
        constexpr int RD = 3, RS = RD + 2;          // ring slots: group q - 1 stays live while q + RD is loaded (bf16 pairs groups)
        f32x4 ra[RS], rb[RS];
        auto rd = [&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            if constexpr (Q < NQ) {
                constexpr int k0 = 8 * Q, tap = k0 / CIN, c0 = k0 % CIN;     // first k of lane half 0: channels c0..c0+3 of one tap
                constexpr bool wrap = c0 + 4 >= CIN;                         // lane half 1's run starts in the next tap
                constexpr int boff = (tap * CS + c0) / 4;                    // in f32x4 units (CS, c0 multiples of 4)
                ra[Q % RS] = ap[2 * Q];
                rb[Q % RS] = wrap ? bp8[boff] : bp4[boff];
            }
        };
        auto mm = [&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            const f32x4 av = ra[Q % RS], bv = rb[Q % RS];
            if constexpr (MATH == VC_MATH_FP32) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[e & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[e & 1], 0, 0, 0);
            } else if constexpr (Q & 1) {
                // bf16 modes: two consecutive groups (16 k; lane half h holds 8 of them, the same for A and B) make one 32x32x16 MFMA
                const f32x4 a0 = ra[(Q - 1) % RS], b0 = rb[(Q - 1) % RS];
                float va[8] = {a0[0], a0[1], a0[2], a0[3], av[0], av[1], av[2], av[3]};
                float vb[8] = {b0[0], b0[1], b0[2], b0[3], bv[0], bv[1], bv[2], bv[3]};
                bf16x8 ah, al, bh, bl;
                to_frag<MATH>(va, ah, al); to_frag<MATH>(vb, bh, bl);
                constexpr int bank = (Q >> 1) & 1;
                if constexpr (MATH == VC_MATH_BF16X3) {
                    acc[bank] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[bank], 0, 0, 0);
                    acc[bank] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[bank], 0, 0, 0);
                }
                acc[bank] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[bank], 0, 0, 0);
            }
        };
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (rd(std::integral_constant<int, Q>{}), ...); }(std::make_integer_sequence<int, RD>{});
        [&]<int... Q>(std::integer_sequence<int, Q...>) {
            ((rd(std::integral_constant<int, Q + RD>{}), __builtin_amdgcn_sched_barrier(0),
              mm(std::integral_constant<int, Q>{}), __builtin_amdgcn_sched_barrier(0)), ...);
        }(std::make_integer_sequence<int, NQ>{});
        // ---- epilogue: + bias, through LDS so that a lane stores 16 contiguous bytes and a wave instruction two whole
        //      512-byte rows (scalar stores straight from the accumulators: 16 instructions of 2 x 128 B per wave, +8 us at B=256)
        const f32x16 accs = acc[0] + acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = (r & 3) + 8 * (r >> 2) + 4 * h;
            Ys[ml * LDY + wid * 32 + l31] = accs[r] + bv[r];
        }
        __syncthreads();
        {
            const int b = tile / tps, t = (tile - b * tps) * BN + (tid & 31) * 4;
            float* y = a.y + (long)g * a.y_gs + (long)b * a.Cout * a.T + t;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ml = (tid >> 5) + 8 * i, m = slab * 32 + ml;
                const f32x4 v = *reinterpret_cast<const f32x4*>(Ys + ml * LDY + (tid & 31) * 4);
                if (m < a.Cout && t < a.T) *reinterpret_cast<f32x4*>(y + (long)m * a.T) = v;
            }
        }
    }
}

// =====================================================================================
// Cin = 1 convolution on the vector ALU (C1Args).  Lanes run along output pixels (coalesced y / dy rows); filter taps
// and biases are wave-uniform scalar loads.
// =====================================================================================
__device__ __forceinline__ float c1_round(float v, int math) { return math == VC_MATH_BF16 ? (float)(__bf16)v : v; }

// window of output pixel (i, j): the kh*kw (<= 9) input values, zero outside the image / beyond the last tap
__device__ __forceinline__ void c1_window(const C1Args& a, const float* x, int i, int j, bool pv, float (&xv)[9]) {
    const int nt = a.kh * a.kw;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int di = t / a.kw, dj = t - di * a.kw;
        const int h = i * a.sh + di - a.pt, w = j * a.sw + dj - a.pl;
        const bool ok = pv && t < nt && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
        const float v = x[ok ? h * a.W + w : 0];                 // unconditional load, select afterwards: no divergence
        xv[t] = ok ? c1_round(v, a.math) : 0.f;
    }
}

// y[g][b][co][p] = bias[co] + sum_t w[t][co] * window_t(p);  block = 256 pixels x 64 channels, filter slice in LDS.
// With a.act set the layer is a gated pair (module.py:201-203: value filters [0, C), gate filters [C, 2C), C = Cout / 2):
// a block then holds 32 value channels and the matching 32 gate channels, and writes act = value * sigmoid(gate) next to
// the raw pair (kept for the backward pass) -- the GLU costs no second pass over the raw tensor.
__global__ __launch_bounds__(256) void c1_fwd_kernel(const C1Args a) {
    touch_kernarg(a);
    __shared__ __attribute__((aligned(16))) float ws[10 * 64];       // [t][c] + bias row
    const bool glu = a.act != nullptr;
    const int C = a.Cout >> 1;
    const int nch = glu ? (C + 31) / 32 : (a.Cout + 63) / 64;
    const int g = blockIdx.z / nch, cc = blockIdx.z - g * nch, b = blockIdx.y;
    const int P = a.OH * a.OW, p = blockIdx.x * 256 + threadIdx.x, nt = a.kh * a.kw;
    const float* wt = a.w + (long)g * a.w_gs;
    for (int idx = threadIdx.x; idx < 640; idx += 256) {
        const int t = idx >> 6, c = idx & 63;
        const int co = glu ? (c < 32 ? 0 : C) + cc * 32 + (c & 31) : cc * 64 + c;
        const bool cv = glu ? cc * 32 + (c & 31) < C : co < a.Cout;
        float v = 0.f;
        if (cv) { if (t < nt) v = c1_round(wt[t * a.Cout + co], a.math); else if (t == 9 && a.bias) v = a.bias[(long)g * a.b_gs + co]; }
        ws[idx] = v;
    }
    __syncthreads();
    const bool pv = p < P;
    const int i = pv ? p / a.OW : 0, j = pv ? p - i * a.OW : 0;
    float xv[9];
    c1_window(a, a.x + (long)g * a.x_gs + (long)b * a.H * a.W, i, j, pv, xv);
    if (!pv) return;
    float* y = a.y + (long)g * a.y_gs + (long)b * a.Cout * P + p;
    if (glu) {
        float* o = a.act + (long)g * a.act_gs + (long)b * C * P + p;
#pragma unroll 2
        for (int c4 = 0; c4 < 8; ++c4) {
            f32x4 av = *reinterpret_cast<const f32x4*>(ws + 9 * 64 + c4 * 4), ag = *reinterpret_cast<const f32x4*>(ws + 9 * 64 + 32 + c4 * 4);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                av += xv[t] * *reinterpret_cast<const f32x4*>(ws + t * 64 + c4 * 4);
                ag += xv[t] * *reinterpret_cast<const f32x4*>(ws + t * 64 + 32 + c4 * 4);
            }
            const int c = cc * 32 + c4 * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < C) { y[(long)(c + e) * P] = av[e]; y[(long)(C + c + e) * P] = ag[e]; o[(long)(c + e) * P] = av[e] * sigmoidf_(ag[e]); }
        }
        return;
    }
#pragma unroll 4
    for (int c4 = 0; c4 < 16; ++c4) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + 9 * 64 + c4 * 4);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += xv[t] * *reinterpret_cast<const f32x4*>(ws + t * 64 + c4 * 4);
        const int co = cc * 64 + c4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (co + e < a.Cout) y[(long)(co + e) * P] = acc[e];
    }
}

// input gradient as a scatter from dy: z[t] = sum_co w[t][co] * dy[co][p] for the thread's pixel p (dy read once,
// coalesced), then dx[window_t(p)] += z[t] (atomics; dx zeroed by the launcher unless it accumulates)
__global__ __launch_bounds__(256) void c1_dgrad_kernel(const C1Args a) {
    touch_kernarg(a);
    __shared__ __attribute__((aligned(16))) float ws[64 * 12];       // [c][t], t padded to 12
    const int nch = (a.Cout + 63) / 64;
    const int g = blockIdx.z / nch, cc = blockIdx.z - g * nch, b = blockIdx.y;
    const int P = a.OH * a.OW, p = blockIdx.x * 256 + threadIdx.x, nt = a.kh * a.kw;
    const float* wt = a.w + (long)g * a.w_gs;
    for (int idx = threadIdx.x; idx < 768; idx += 256) {
        const int c = idx / 12, t = idx - c * 12, co = cc * 64 + c;
        ws[idx] = (t < nt && co < a.Cout) ? c1_round(wt[t * a.Cout + co], a.math) : 0.f;
    }
    __syncthreads();
    if (p >= P) return;
    const float* dy = a.dy + (long)g * a.y_gs + ((long)b * a.Cout + cc * 64) * P + p;
    const int nc = min(64, a.Cout - cc * 64);
    float z[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) z[t] = 0.f;
#pragma unroll 8
    for (int c = 0; c < nc; ++c) {
        const float d = c1_round(dy[(long)c * P], a.math);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws + c * 12), w1 = *reinterpret_cast<const f32x4*>(ws + c * 12 + 4);
        const float w8 = ws[c * 12 + 8];
#pragma unroll
        for (int t = 0; t < 4; ++t) { z[t] = fmaf(d, w0[t], z[t]); z[4 + t] = fmaf(d, w1[t], z[4 + t]); }
        z[8] = fmaf(d, w8, z[8]);
    }
    const int i = p / a.OW, j = p - i * a.OW;
    float* dx = a.dx + (long)g * a.dx_gs + (long)b * a.H * a.W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int di = t / a.kw, dj = t - di * a.kw;
        const int h = i * a.sh + di - a.pt, w = j * a.sw + dj - a.pl;
        if (t < nt && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) atomicAdd(dx + h * a.W + w, z[t]);
    }
}

// Deterministic mode: the same input gradient as a GATHER -- one thread per dx element walks the taps and filters in a fixed
// order (dy is re-read kh*kw times from L2; this mode trades speed for bitwise reproducibility).  dx = (accumulate ? dx : 0) + sum.
__global__ __launch_bounds__(256) void c1_dgrad_gather_kernel(const C1Args a, int accumulate) {
    const int g = blockIdx.z, b = blockIdx.y;
    const int HW = a.H * a.W, q = blockIdx.x * 256 + threadIdx.x, P = a.OH * a.OW, nt = a.kh * a.kw;
    if (q >= HW) return;
    const int h = q / a.W, w = q - h * a.W;
    const float* wt = a.w + (long)g * a.w_gs;
    const float* dy = a.dy + (long)g * a.y_gs + (long)b * a.Cout * P;
    float acc = 0.f;
    for (int t = 0; t < nt; ++t) {
        const int di = t / a.kw, dj = t - di * a.kw;
        const int ih = h + a.pt - di, iw = w + a.pl - dj;
        if (ih < 0 || iw < 0 || ih % a.sh || iw % a.sw) continue;
        const int i = ih / a.sh, j = iw / a.sw;
        if (i >= a.OH || j >= a.OW) continue;
        const float* d = dy + i * a.OW + j;
        float z = 0.f;
        for (int co = 0; co < a.Cout; ++co) z = fmaf(c1_round(d[(long)co * P], a.math), c1_round(wt[t * a.Cout + co], a.math), z);
        acc += z;
    }
    float* dx = a.dx + (long)g * a.dx_gs + (long)b * HW + q;
    *dx = (accumulate ? *dx : 0.f) + acc;
}

// dw[g][t][co] += sum_{b,p} window_t(p) * dy[g][b][co][p].  A block owns 8 channels and C1_BPB samples; its 4 waves take a
// quarter of the pixels each, gather the window of a pixel once for the 8 channels, reduce with wave shuffles and LDS,
// and issue ONE atomic per (tap, channel) and block (atomics on one cache line serialise: per-wave atomics cost 80 us).
// (deterministic mode: one block walks ALL samples in order, so every dw element receives exactly one contribution)
__global__ __launch_bounds__(256) void c1_wgrad_kernel(const C1Args a) {
    touch_kernarg(a);
    const int C1_BPB = a.det ? a.B : 1;
    __shared__ float red[4][72];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = blockIdx.z, co0 = blockIdx.x * 8;
    const int P = a.OH * a.OW, nt = a.kh * a.kw;
    const int per = ((P + 4 * 64 - 1) / (4 * 64)) * 64, p0 = wid * per, p1 = min(P, p0 + per);
    float acc[8][9];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    const int b1 = min(a.B, (int)(blockIdx.y + 1) * C1_BPB);
    for (int b = blockIdx.y * C1_BPB; b < b1; ++b) {
        const float* x = a.x + (long)g * a.x_gs + (long)b * a.H * a.W;
        const float* dy = a.dy + (long)g * a.y_gs + (long)b * a.Cout * P;
        // two pixels per lane and trip: 16 dy loads + 2 windows in flight before the first use (the kernel is latency-bound:
        // ~64 KB per CU must be in flight to cover HBM latency); loads are unconditional (clamped), masked afterwards
        for (int p = p0 + lane; p < p1; p += 128) {
            const int q = p + 64; const bool qv = q < p1; const int qc = qv ? q : p;
            float d0[8], d1[8], xv0[9], xv1[9];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const long row = (long)min(co0 + c, a.Cout - 1) * P;
                d0[c] = dy[row + p]; d1[c] = dy[row + qc];
            }
            { const int i = p / a.OW, j = p - i * a.OW; c1_window(a, x, i, j, true, xv0); }
            { const int i = qc / a.OW, j = qc - i * a.OW; c1_window(a, x, i, j, qv, xv1); }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float e0 = c1_round(d0[c], a.math), e1 = c1_round(d1[c], a.math);
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(xv1[t], e1, fmaf(xv0[t], e0, acc[c][t]));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = wave_sum_dpp(acc[c][t]);
            if (lane == 0) red[wid][c * 9 + t] = v;
        }
    __syncthreads();
    if (threadIdx.x < 72) {
        const int c = threadIdx.x / 9, t = threadIdx.x - c * 9;
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (t < nt && co0 + c < a.Cout) atomicAdd(a.dw + (long)g * a.dw_gs + t * a.Cout + co0 + c, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------
int launch_c1_fwd(const C1Args& a, hipStream_t s) {
    const int P = a.OH * a.OW, nch = a.act ? (a.Cout / 2 + 31) / 32 : (a.Cout + 63) / 64;
    if (a.act && (a.Cout & 1)) { vc_set_error("c1_fwd: a gated pair needs an even filter count"); return -1; }
    c1_fwd_kernel<<<dim3((P + 255) / 256, a.B, a.G * nch), 256, 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_c1_dgrad(const C1Args& a, int accumulate, hipStream_t s) {
    const int P = a.OH * a.OW, nch = (a.Cout + 63) / 64;
    if (a.det || vc_hooks().c1_gather) {     // (tests: the gather form without the rest of deterministic mode -- the scatter's atomics are the one
                                              //  summation-order source that PROPAGATES into the generators' gradient chains)
        c1_dgrad_gather_kernel<<<dim3((a.H * a.W + 255) / 256, a.B, a.G), 256, 0, s>>>(a, accumulate);
        VC_HIP(hipGetLastError());
        return 0;
    }
    if (!accumulate) {          // the scatter accumulates with atomics: one launch zeroes dx of every group
        bool al = true;
        for (int g = 0; g < a.G; ++g) al = al && ((uintptr_t)(a.dx + (long)g * a.dx_gs) % 16 == 0);
        if (al) {
            std::vector<std::pair<float*, long>> zl;
            for (int g = 0; g < a.G; ++g) zl.push_back({a.dx + (long)g * a.dx_gs, (long)a.B * a.H * a.W});
            const int r = launch_zero_list(zl, s); if (r) return r;
        } else
            for (int g = 0; g < a.G; ++g) VC_HIP(hipMemsetAsync(a.dx + (long)g * a.dx_gs, 0, (size_t)a.B * a.H * a.W * sizeof(float), s));
    }
    c1_dgrad_kernel<<<dim3((P + 255) / 256, a.B, a.G * nch), 256, 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_c1_wgrad(const C1Args& a, hipStream_t s) {
    c1_wgrad_kernel<<<dim3((a.Cout + 7) / 8, a.det ? 1 : a.B, a.G), 256, 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_wsconv1d(const WsConvArgs& a, int cin, int kw, int math, hipStream_t s, double flops) {
    if (cin != 24 || kw != 15) return -2;            // the geometries this file instantiates (generator h1 at 24 MCEPs)
    const int slabs = (a.Cout + 31) / 32, ntiles = a.B * ((a.T + 127) / 128);
    int parts = 512 / (slabs * a.G); if (parts < 1) parts = 1; if (parts > ntiles) parts = ntiles;   // 2 workgroups per CU, persistent over the tiles
    dim3 grid(parts, slabs, a.G), block(256);
    ProfScope ps(2, flops, s);                       // its own timing class: bench.py's roofline leg is about igemm_kl_kernel
    if (math == VC_MATH_BF16) wsconv1d_kernel<24, 15, VC_MATH_BF16><<<grid, block, 0, s>>>(a);
    else if (math == VC_MATH_BF16X3) wsconv1d_kernel<24, 15, VC_MATH_BF16X3><<<grid, block, 0, s>>>(a);
    else wsconv1d_kernel<24, 15, VC_MATH_FP32><<<grid, block, 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}
