// vc_conv_small.hip -- the two first-layer convolutions of the reference networks, which are not GEMM-shaped:
//  wsconv1d_kernel : generator h1 (module.py:161-162: 24 -> 2x128 channels, k = 15): weight-stationary, window expansion in LDS
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
    if (a.det) {
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

// =====================================================================================
// gated_conv1d_glu_kernel: the generator's h1 block in ONE kernel (module.py:161-163):
//     out = (conv_value(x) + b_v) * sigmoid(conv_gate(x) + b_g)
// value and gate filters live in one tensor w[k][2C] (filters [0,C) value, [C,2C) gate).  Weight-stationary like
// wsconv1d_kernel: a workgroup keeps the 32 value rows AND the matching 32 gate rows of its channel slab in LDS for its
// whole life and streams pixel tiles through them; a wave owns 32 pixels of the tile and accumulates the value tile and
// the gate tile side by side (one B fragment feeds two MFMAs), so the GLU happens in registers and HBM sees
// 12 KB of frames in and 64 KB of activations out per sample -- SURVEY 8d's algorithmic bytes.
//   MATH fp32 : operands fp32 in LDS, v_mfma_f32_32x32x2_f32 (exact)            4 waves, 128-pixel tiles
//   MATH bf16 : operands rounded to bf16 ONCE on their way into LDS (RNE),       8 waves, 256-pixel tiles
//               fragments are plain 16-byte LDS reads into v_mfma_f32_32x32x16_bf16, fp32 accumulate
// The window expansion is addressing: a run of R consecutive k (R = 4 fp32 / 8 bf16) is R channels of one tap, i.e.
// contiguous bytes of Xs row (pixel + tap).
// =====================================================================================
// INKW = true (default): the workgroup builds its filter slab in LDS straight from the fp32 weights w[k][Cout] (lanes along the
// 64 rows: two coalesced 128-byte runs per k; a lane converts R consecutive k of its row and stores them with one 16-byte
// LDS write) -- no separate packing launch, nothing between the caller's weights and this kernel.  INKW = false: the slab is
// a straight copy of the image pack_gated_weights_kernel prepared (kept for A/B measurements: VC_GLU_PACKED=1).
// NWV_ = waves per workgroup (bf16: 8 = one 512-thread workgroup per CU working on two tiles at a time; 4 = TWO 256-thread
// workgroups per CU, one tile each, whose load / multiply / store phases drift apart and overlap).
// RAW = the raw (value, gate) pair is stored too (training).  A template parameter, and the stores are buffer stores whose
// out-of-range lanes are dropped by the hardware range check, so that EVERY trip issues the same number of store
// instructions: the wait for the next tile's frames (issued before this tile's stores) can then be a counted
// s_waitcnt vmcnt(#stores) -- with stores under branches the compiler has to wait for vmcnt(0), i.e. for the stores' own
// completion in L2, once per tile (gfx950 has one in-order counter for loads and stores).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int CIN, int KW, bool BF16, bool INKW, int NWV_, bool RAW>
__global__ __launch_bounds__(64 * NWV_, (BF16 && NWV_ == 4) ? 2 : (NWV_ == 8 ? 2 : 1)) void gated_conv1d_glu_v1_kernel(const WsConvArgs a, const void* __restrict__ packed) {
    typedef typename std::conditional<BF16, __bf16, float>::type T;
    constexpr int R = BF16 ? 8 : 4;                        // k run per lane half and step
    static_assert(CIN % R == 0 && (KW & 1), "runs of R channels inside one tap; odd kernel width");
    constexpr int NWV = NWV_, TP = 128, NSUB = NWV / 4;   // waves, pixels per tile, tiles in flight
    constexpr int K = CIN * KW, KR = ((K + 2 * R - 1) / (2 * R)) * (2 * R), NG = KR / (2 * R);   // steps of 2R k (one run per lane half)
    constexpr int LDW = KR + (BF16 ? 8 : 4);               // slab row stride (elements): conflict-free 16-byte reads across 32 rows
    constexpr int PADL = (KW - 1) / 2;
    constexpr int XR = TP + KW + 1;                        // input positions of a tile (+ the rows the zero-weight tail runs touch)
    constexpr int CS = BF16 ? CIN + 16 : CIN + 4;          // Xs row stride (elements): 80 B / 112 B -> conflict-free 16-byte reads across pixels
    constexpr int NXF = (XR * CIN + 255) / 256;            // a 4-wave half of the workgroup stages its own tile
    constexpr int LDS_S = 36;                              // per-wave 32 x 32 staging tile (row stride)
    __shared__ __attribute__((aligned(16))) T Ws[64 * LDW];
    __shared__ __attribute__((aligned(16))) T Xs[NSUB * XR * CS];
    __shared__ __attribute__((aligned(16))) float Ss[NWV * 32 * LDS_S];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = wid >> 2, w4 = wid & 3, stid = tid & 255;      // which of the NSUB tiles in flight this wave works on
    const int l31 = lane & 31, h = lane >> 5;
    const int g = blockIdx.z, slab = blockIdx.y;
    const int C = a.Cout >> 1;
    const int tps = (a.T + TP - 1) / TP, ntiles = a.B * tps;

    float bv[16], bg[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = slab * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool ok = a.bias && c < C;
        bv[r] = ok ? a.bias[(long)g * a.b_gs + c] : 0.f;
        bg[r] = ok ? a.bias[(long)g * a.b_gs + C + c] : 0.f;
    }
    const float* xg = a.x + (long)g * a.x_gs;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.y + (long)g * a.y_gs), 0, (int)((unsigned)a.B * (unsigned)C * (unsigned)a.T * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rawrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RAW ? a.raw + (long)g * a.raw_gs : a.y), 0, (int)((unsigned)a.B * (unsigned)a.Cout * (unsigned)a.T * 4u), 0x00020000);
    T* Xt = Xs + sub * XR * CS;
    float xr[NXF];
    unsigned xok = 0;                                // bit i: xr[i] is a frame of the tile (else padding -> 0)   [INKW = false only]
    // ---- frame staging.  Per-thread constants, once: element i of a thread is (channel c_i, position pos_i) of EVERY tile;
    //      its byte offset inside the group's tensor is tile base + goff[i], its LDS slot lds[i].  A tile then costs one add,
    //      one range compare and one select per element, and the load itself is a BUFFER load whose offset is sent out of
    //      range for padding positions (hardware returns 0): no 64-bit address arithmetic, no masks, no selects at store time.
    //      (The PMC pass of the first version of this kernel showed ~600 vector-ALU instructions per tile and wave -- address
    //      arithmetic and an IEEE division per GLU output -- against 46 MFMAs: VALU-bound at 0.32 MFMA-pipe busy.)
    int goff[NXF], gpos[NXF]; unsigned lds[NXF];
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 xrs_;                                       // buffer descriptor of this group's frames, in SGPRs (inline asm operand)
    if constexpr (INKW) {
#pragma unroll
        for (int i = 0; i < NXF; ++i) {
            const int idx = stid + 256 * i, c = idx / XR, pos = idx - c * XR;
            const bool in = idx < XR * CIN;
            goff[i] = (c * a.T + pos) * 4;
            gpos[i] = in ? pos : 0x40000000;          // out of the staging area: always fails the range check below
            lds[i] = (unsigned)((in ? pos * CS + c : CS - 1) * (int)sizeof(T));   // dummy slot: a pad column no fragment reads
        }
        const unsigned long long pa = (unsigned long long)xg;
        xrs_[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pa);
        xrs_[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(pa >> 32) & 0xffff);
        xrs_[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)a.B * CIN * (unsigned)a.T * 4u));
        xrs_[3] = 0x00020000;
    }
    // The loads are issued and NOT touched until store_tile, one whole tile of MFMAs later: masking the value right behind
    // its load (as the round-2 kernel did) puts an s_waitcnt between the prefetch and the multiply loop, i.e. a full HBM
    // round trip (~2 us) in front of every tile.
    auto load_tile = [&](int tile) {                 // input frames t0-PADL .. of all channels -> registers (lanes along time: coalesced)
        const bool tv = tile < ntiles;
        const int b = tv ? tile / tps : 0, t0 = (tile - b * tps) * TP - PADL;
        if constexpr (INKW) {
            // Issued behind the compiler's back (it would otherwise wait for them with s_waitcnt vmcnt(0) -- it cannot count the
            // younger stores across the loop's back edge -- and that also waits for the previous tile's output stores to
            // COMPLETE, once per tile).  wait_tile() below is the matching, counted wait.
            const int base = (b * CIN * a.T + t0) * 4, tlim = tv ? a.T : 0;
#pragma unroll
            for (int i = 0; i < NXF; ++i) {
                const unsigned off = ((unsigned)(t0 + gpos[i]) < (unsigned)tlim) ? (unsigned)(base + goff[i]) : VC_OOB;
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(xr[i]) : "v"(off), "s"(xrs_) : "memory");
            }
        } else {
            const float* xb = xg + (long)b * CIN * a.T;
            xok = 0;
#pragma unroll
            for (int i = 0; i < NXF; ++i) {
                const int idx = stid + 256 * i, c = idx / XR, pos = idx - c * XR, t = t0 + pos;
                const bool ok = tv && idx < XR * CIN && (unsigned)t < (unsigned)a.T;
                xr[i] = xb[ok ? c * a.T + t : 0];
                xok |= (unsigned)ok << i;
            }
        }
    };
    // The frames of the tile about to be staged were requested one trip ago, BEFORE that trip's NST output stores: the vector
    // memory counter retires in order, so "at most NST operations outstanding" means the frames have arrived while the
    // stores may still be on their way.  (The compiler sees xr[] pass through the asm, so no use can move above the wait.)
    // NST = buffer stores per trip (store_plane: 4 per plane, all unconditional): 12 with RAW, else 4
    auto wait_tile = [&](bool stores_behind) {
        if constexpr (INKW) {
            static_assert(NXF == 14, "operand list below");
#define VC_XR_OPS "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]), \
                  "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13])
            if (stores_behind) { if constexpr (RAW) asm volatile("s_waitcnt vmcnt(12)" : VC_XR_OPS :: "memory"); else asm volatile("s_waitcnt vmcnt(4)" : VC_XR_OPS :: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" : VC_XR_OPS :: "memory");
#undef VC_XR_OPS
        }
    };
    auto store_tile = [&]() {
        if constexpr (INKW) {
            char* xb_ = reinterpret_cast<char*>(Xt);
#pragma unroll
            for (int i = 0; i < NXF; ++i) *reinterpret_cast<T*>(xb_ + lds[i]) = (T)xr[i];
        } else {
#pragma unroll
            for (int i = 0; i < NXF; ++i) {
                const int idx = stid + 256 * i, c = idx / XR, pos = idx - c * XR;
                if (idx < XR * CIN) Xt[pos * CS + c] = (T)(((xok >> i) & 1u) ? xr[i] : 0.f);
            }
        }
    };
    typedef typename std::conditional<BF16, bf16x8, f32x4>::type frag_t;
    static_assert(LDW % R == 0, "a slab row is a whole number of R-element runs");
    const frag_t* av = reinterpret_cast<const frag_t*>(Ws + l31 * LDW + R * h);
    const frag_t* ag = reinterpret_cast<const frag_t*>(Ws + (32 + l31) * LDW + R * h);
    const T* xrow = Xt + (w4 * 32 + l31) * CS;
    float* sw = Ss + wid * 32 * LDS_S;

    // persistent over the tiles: trip it handles tiles (blockIdx.x + it * gridDim.x) * NSUB + sub
    int t0i = blockIdx.x * NSUB;
    if (t0i < ntiles) load_tile(t0i + sub);          // the first tile's frames travel while the filter slab is copied
    // ---- filter slab, once: rows 0-31 = value filters 32*slab.., rows 32-63 = the matching gate filters
    if constexpr (INKW) {
        // lane = slab row m (channel ch), wave = run of R consecutive k: R coalesced loads (a wave reads the 32 value and the
        // 32 gate filters of one k: two 128-byte runs), one conversion, one 16-byte LDS store (conflict-free: 8 consecutive
        // rows are 8 different bank quads).  Two runs are loaded before the first is stored (latency overlap).
        constexpr int NRUN = LDW / R;                       // runs per row, zero tail included
        constexpr int NIT = (NRUN + NWV - 1) / NWV;        // runs per wave
        const int m = lane, ch = (m < 32 ? 0 : C - 32) + slab * 32 + m;
        const bool valid = slab * 32 + (m & 31) < C;
        const float* wg = a.w + (long)g * a.w_gs + (valid ? ch : 0);
        frag_t* dst = reinterpret_cast<frag_t*>(Ws + m * LDW);
        // every load is issued before the first store: ONE memory round trip for the whole slab (a loop of load-then-store
        // pays the L2 / HBM latency once per iteration: ~1 us each, the bulk of the round-2 kernel's fixed cost)
        float v[NIT][R];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int e = 0; e < R; ++e) {
                const int k = (wid + it * NWV) * R + e;
                const bool ok = valid && k < K;
                v[it][e] = wg[ok ? (long)k * a.Cout : 0];
                v[it][e] = ok ? v[it][e] : 0.f;
            }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (wid + it * NWV < NRUN) {
                frag_t f;
#pragma unroll
                for (int e = 0; e < R; ++e) f[e] = (T)v[it][e];
                dst[wid + it * NWV] = f;
            }
        }
    } else {
        // already in the LDS image (element type, row stride, zero tail) by pack_gated_weights_kernel: a straight 16-byte copy
        constexpr int NV16 = 64 * LDW * (int)sizeof(T) / 16;
        const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(packed) + ((long)g * gridDim.y + slab) * 64 * LDW);
        f32x4* dst = reinterpret_cast<f32x4*>(Ws);
        for (int i = tid; i < NV16; i += 64 * NWV) dst[i] = src[i];
    }
    static_assert(!INKW || NSUB == 1, "the counted wait assumes every trip of a wave stores its tile");
    bool first_trip = true;
    for (; t0i < ntiles; t0i += gridDim.x * NSUB) {
        const int tile = t0i + sub;
        __syncthreads();                              // the previous tile's readers are done (first trip: orders the slab writes)
        wait_tile(!first_trip); first_trip = false;
        store_tile();
        __syncthreads();
        load_tile(t0i + gridDim.x * NSUB + sub);      // next tiles' frames are in flight during the MFMAs (past the last tile: masked, harmless)
        if (tile >= ntiles) continue;                 // odd tile count: this half idles (barriers are above)

        // one accumulator per branch, started at the bias: value and gate MFMAs alternate, so two MFMAs (>= the dependent-
        // accumulate latency) lie between updates of the same registers; no bank sum, no bias add in the epilogue
        f32x16 accv, accg;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accv[r] = bv[r]; accg[r] = bg[r]; }
        // fragment ring RD steps deep, reads pinned ahead of the MFMAs that use them (LDS latency stays covered)
        constexpr int RD = 3, RS = RD + 1;
        frag_t fa[RS], fg[RS], fb[RS];
        auto rd = [&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            if constexpr (Q < NG) {
                // lane half h reads run 2Q + h: tap = run / (CIN / R), channel offset (run % (CIN / R)) * R
                constexpr int RPT = CIN / R;
                constexpr int r0 = 2 * Q, r1 = 2 * Q + 1;
                constexpr int o0 = (r0 / RPT) * CS + (r0 % RPT) * R, o1 = (r1 / RPT) * CS + (r1 % RPT) * R;
                fa[Q % RS] = av[2 * Q];
                fg[Q % RS] = ag[2 * Q];
                fb[Q % RS] = *reinterpret_cast<const frag_t*>(xrow + (h ? o1 : o0));
            }
        };
        auto mm = [&](auto QC) {
            constexpr int Q = decltype(QC)::value;
            const frag_t a_ = fa[Q % RS], g_ = fg[Q % RS], b_ = fb[Q % RS];
            if constexpr (BF16) {
                accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, accv, 0, 0, 0);
                accg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g_, b_, accg, 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    accv = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[e], b_[e], accv, 0, 0, 0);
                    accg = __builtin_amdgcn_mfma_f32_32x32x2f32(g_[e], b_[e], accg, 0, 0, 0);
                }
            }
        };
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (rd(std::integral_constant<int, Q>{}), ...); }(std::make_integer_sequence<int, RD>{});
        [&]<int... Q>(std::integer_sequence<int, Q...>) {
            ((rd(std::integral_constant<int, Q + RD>{}), __builtin_amdgcn_sched_barrier(0),
              mm(std::integral_constant<int, Q>{}), __builtin_amdgcn_sched_barrier(0)), ...);
        }(std::make_integer_sequence<int, NG>{});

        // ---- epilogue in registers: + bias, GLU; through a per-wave LDS tile so that a lane stores 16 contiguous bytes
        const f32x16 sv = accv, sg = accg;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = (r & 3) + 8 * (r >> 2) + 4 * h;
            // sigmoid with the hardware reciprocal (1 ulp) instead of an IEEE division: ~5 instead of ~15 instructions per output
            sw[ml * LDS_S + l31] = sv[r] * __builtin_amdgcn_rcpf(1.f + __expf(-sg[r]));
        }
        // wave-private tile: no workgroup barrier, the LDS queue is in order within a wave
        const int b = tile / tps, t = (tile - b * tps) * TP + w4 * 32 + (lane & 7) * 4;
        // the staged 32 x 32 tile -> rows (channels) of a [B][nch][T] tensor, 16 bytes per lane; lanes outside the tensor get an
        // out-of-range offset (dropped by the buffer range check): the instruction count does not depend on the data
        auto store_plane = [&](__amdgpu_buffer_rsrc_t rs, int nch, int ch0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ml = (lane >> 3) + 8 * i, c = slab * 32 + ml;
                const f32x4 v = *reinterpret_cast<const f32x4*>(sw + ml * LDS_S + (lane & 7) * 4);
                const unsigned off = (c < C && t < a.T) ? (unsigned)((((long)b * nch + ch0 + c) * a.T + t) * 4) : VC_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)off, 0, 0);
            }
        };
        store_plane(yrs, C, 0);
        if constexpr (RAW) {                          // training: the raw value / gate pair is kept for the backward pass
#pragma unroll
            for (int r = 0; r < 16; ++r) sw[((r & 3) + 8 * (r >> 2) + 4 * h) * LDS_S + l31] = sv[r];
            store_plane(rawrs, 2 * C, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) sw[((r & 3) + 8 * (r >> 2) + 4 * h) * LDS_S + l31] = sg[r];
            store_plane(rawrs, 2 * C, C);
        }
    }
}

// packed[g][slab][m][k] = T(w[g][k][channel(slab, m)]) in the fused kernel's LDS image: row stride LDW, rows 0-31 the value
// filters of the slab, rows 32-63 its gate filters, k >= K zero.  Weights change once per optimizer step (or never, at
// inference); the image is rebuilt per call here (a 3 us launch), kept out of the kernel's own time by ProfScope.
template <int CIN, int KW, bool BF16>
__global__ __launch_bounds__(256) void pack_gated_weights_kernel(const float* __restrict__ w, long w_gs, int Cout, void* __restrict__ packed) {
    typedef typename std::conditional<BF16, __bf16, float>::type T;
    constexpr int R = BF16 ? 8 : 4, K = CIN * KW, KR = ((K + 2 * R - 1) / (2 * R)) * (2 * R), LDW = KR + (BF16 ? 8 : 4);
    const int slab = blockIdx.x, g = blockIdx.y, C = Cout >> 1;
    T* dst = reinterpret_cast<T*>(packed) + ((long)g * gridDim.x + slab) * 64 * LDW;
    const float* wg = w + (long)g * w_gs;
    // lanes along the 64 rows (channels: coalesced reads), waves along k
    const int m = threadIdx.x & 63, ch = (m < 32 ? 0 : C - 32) + slab * 32 + m;
    const bool valid = slab * 32 + (m & 31) < C;
    for (int k = (threadIdx.x >> 6) + 4 * blockIdx.z; k < LDW; k += 4 * gridDim.z)     // k rows spread over gridDim.z blocks: ~6 loads per thread
        dst[m * LDW + k] = (T)((k < K && valid) ? wg[(long)k * Cout + ch] : 0.f);
}

int64_t gated_conv1d_glu_workspace_floats(int G, int cout2) {
    const int slabs = cout2 / 64;
    return (int64_t)G * slabs * 64 * (368 + 8);           // the larger of the two images (fp32: 64 x 364 floats; bf16: 64 x 376 halves)
}

// y [G][B][Cout/2][T] = GLU of the gated pair; returns -2 when the geometry is not built
int launch_gated_conv1d_glu_v1(const WsConvArgs& a, int cin, int kw, int math, float* workspace, hipStream_t s, double flops) {
    if (cin != 24 || kw != 15 || (a.Cout & 63) || (a.T & 3)) return -2;
    if ((long)a.B * a.Cout * a.T * 4 >= 0xfff00000L) return -2;      // 32-bit buffer offsets of the output stores
    const int C = a.Cout / 2, slabs = C / 32;
    const bool bf = math == VC_MATH_BF16;
    static const bool packed_image = getenv("VC_GLU_PACKED") != nullptr;      // A/B: the round-2 two-launch form
    const bool wide = packed_image;                   // the round-2 form: one 8-wave workgroup per CU working on two tiles at a time
    const int nsub = (bf && wide) ? 2 : 1, ntiles = a.B * ((a.T + 127) / 128), trips = (ntiles + nsub - 1) / nsub;
    const int wgs = (bf && !wide) ? 512 : 256;        // resident workgroups on the chip, persistent over the tiles
    int parts = wgs / (slabs * a.G); if (parts < 1) parts = 1; if (parts > trips) parts = trips;
    dim3 grid(parts, slabs, a.G);
    ProfScope ps(2, flops, s);                       // everything a call launches
#define VC_GLU(BF, INKW, NWV, P) { if (a.raw) gated_conv1d_glu_v1_kernel<24, 15, BF, INKW, NWV, true><<<grid, dim3(64 * NWV), 0, s>>>(a, P); \
                                  else gated_conv1d_glu_v1_kernel<24, 15, BF, INKW, NWV, false><<<grid, dim3(64 * NWV), 0, s>>>(a, P); }
    if (packed_image) {
        if (bf) pack_gated_weights_kernel<24, 15, true><<<dim3(slabs, a.G, 16), 256, 0, s>>>(a.w, a.w_gs, a.Cout, workspace);
        else pack_gated_weights_kernel<24, 15, false><<<dim3(slabs, a.G, 16), 256, 0, s>>>(a.w, a.w_gs, a.Cout, workspace);
        if (bf) VC_GLU(true, false, 8, workspace) else VC_GLU(false, false, 4, workspace)
    } else if (bf) VC_GLU(true, true, 4, nullptr)
    else VC_GLU(false, true, 4, nullptr)
#undef VC_GLU
    VC_HIP(hipGetLastError());
    return 0;
}
