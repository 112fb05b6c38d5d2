// vc_elementwise.hip -- the non-GEMM kernels of the CycleGAN-VC hot path (all HBM- or latency-bound):
//  norm_*_kernel   : instance_norm_layer + gated_linear_layer (+ pixel_shuffler, + residual add),
//                    DPP / wavefront-shuffle reductions (reference module.py:3-20,81,135-146)
//  head_*          : tf.layers.dense(1, sigmoid) (module.py:211)
//  loss_kernel     : utils.py:6-12 + loss algebra model.py:57-90, fused with the seed gradients
//  adam_kernel     : ApplyAdam (model.py:107-108)
//  crop_batch      : preprocess.sample_train_data's crop (preprocess.py:226-233)
#include "vc_internal.h"
#include "vc_device_util.h"
#include "vc_wino.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

// =====================================================================================
// instance norm / GLU / pixel shuffle / residual, one (g,b,c) row per LPR lanes,
// whole row in registers, wavefront shuffles for the per-channel reductions.
// =====================================================================================
#define IN_EPS 1e-6f

// sum over the LPR (32 or 64) consecutive lanes that share one (sample, channel) row; every lane gets the total.
// Rows of 16 on DPP, the last one or two steps through ds_bpermute (was 5-6 bpermutes per sum: the norm kernels are
// latency-bound and issue 4-8 of these per thread).
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
    static_assert(LPR == 32 || LPR == 64, "a row group is 32 or 64 lanes");
    v = row16_sum(v);
#pragma unroll
    for (int o = 16; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// float offset of logical element (branch br, channel c, position l) of sample b in the raw conv output
template <int NB> __device__ __forceinline__ long yoff(const NormArgs& p, int b, int br, int c, int l) {
    if (p.ps) {
        const int T = p.L >> 1;
        return ((long)b * (NB * 2 * p.C) + br * 2 * p.C + (l & 1) * p.C + c) * T + (l >> 1);
    }
    return ((long)b * (NB * p.C) + br * p.C + c) * p.L + l;
}

template <int MODE, int LPR, int NE>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const NormArgs p) {
    touch_kernarg(p);
    constexpr int NB = (MODE == 2) ? 1 : 2;
    const int unit = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    const int total = p.G * p.B * p.C;
    if (unit >= total) return;
    const int g = unit / (p.B * p.C); const int rem = unit - g * (p.B * p.C);
    const int b = rem / p.C, c = rem - b * p.C;
    const float* y = p.y + (long)g * p.y_gs;
    float* out = p.out + (long)g * p.out_gs + ((long)b * p.C + c) * p.L;
    const float invL = 1.f / (float)p.L;

    float v[NE], gt[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int l = sub + e * LPR;
        const bool ok = l < p.L;
        v[e] = ok ? y[yoff<NB>(p, b, 0, c, l)] : 0.f;
        if (NB == 2) gt[e] = ok ? y[yoff<NB>(p, b, 1, c, l)] : 0.f;
    }
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < NE; ++e) { const int l = sub + e * LPR; if (l < p.L) out[l] = v[e] * sigmoidf_(gt[e]); }
        return;
    }
    const float* gam = p.gamma + (long)g * p.par_gs;
    const float* bet = p.beta + (long)g * p.par_gs;
    float* st = p.stats + (long)g * p.st_gs + ((long)b * NB * p.C) * 2;

    float s = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) s += v[e];
    const float mu = group_sum<LPR>(s) * invL;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) { const float d = (sub + e * LPR < p.L) ? v[e] - mu : 0.f; q += d * d; }
    const float rs = rsqrtf(group_sum<LPR>(q) * invL + IN_EPS);
    const float ga = gam[c], be = bet[c];
    if (sub == 0) { st[2 * c] = mu; st[2 * c + 1] = rs; }

    if (MODE == 2) {
        const float* res = p.res + (long)g * p.res_gs + ((long)b * p.C + c) * p.L;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int l = sub + e * LPR;
            if (l < p.L) out[l] = res[l] + ((v[e] - mu) * rs * ga + be);
        }
        return;
    }
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) s2 += gt[e];
    const float mu2 = group_sum<LPR>(s2) * invL;
    float q2 = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) { const float d = (sub + e * LPR < p.L) ? gt[e] - mu2 : 0.f; q2 += d * d; }
    const float rs2 = rsqrtf(group_sum<LPR>(q2) * invL + IN_EPS);
    const float ga2 = gam[p.C + c], be2 = bet[p.C + c];
    if (sub == 0) { st[2 * (p.C + c)] = mu2; st[2 * (p.C + c) + 1] = rs2; }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int l = sub + e * LPR;
        if (l < p.L) out[l] = ((v[e] - mu) * rs * ga + be) * sigmoidf_((gt[e] - mu2) * rs2 * ga2 + be2);
    }
}

// DET (deterministic mode): a lane group owns channel c of EVERY sample and walks the batch in order, so dgamma / dbeta /
// dbias are reduced in registers in a fixed order and added with one plain read-modify-write per element -- no atomics.
template <int MODE, int LPR, int NE, bool DET>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const NormArgs p) {
    touch_kernarg(p);
    constexpr int NB = (MODE == 2) ? 1 : 2;
    const int unit = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    const int rows = DET ? p.C : p.B * p.C;
    if (unit >= p.G * rows) return;
    const int g = unit / rows; const int rem = unit - g * rows;
    const int b0 = DET ? 0 : rem / p.C, c = DET ? rem : rem - b0 * p.C;
    const int b1 = DET ? p.B : b0 + 1;
    const float* y = p.y + (long)g * p.y_gs;
    float* dy = p.dy + (long)g * p.dy_gs;
    const float invL = 1.f / (float)p.L;
    const bool odd = sub & 1;   // LPR is even: every element this lane owns has parity sub&1
    float* dbias = p.dbias ? p.dbias + (long)g * p.dbias_gs : nullptr;
    // conv channels whose bias gradient this row contributes to
    const int cb0 = c;                                  // branch 0, even l (or all l)
    const int cstride_par = p.ps ? p.C : 0;             // odd l -> +C when shuffled
    const int brstride = p.ps ? 2 * p.C : p.C;          // branch 1 offset in conv channels
    float* dgam = (MODE != 1 && p.dgamma) ? p.dgamma + (long)g * p.dpar_gs : nullptr;
    float* dbet = (MODE != 1 && p.dbeta) ? p.dbeta + (long)g * p.dpar_gs : nullptr;
    // reduction slots (DET: summed over the batch in registers, flushed below; otherwise one atomic per row)
    float* slot[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (MODE == 1) { if (dbias) { slot[0] = dbias + c; slot[1] = dbias + p.C + c; } }
    else if (MODE == 2) { if (dgam) { slot[0] = dgam + c; slot[1] = dbet + c; } if (dbias) slot[2] = dbias + c; }
    else {
        if (dbias) {
            slot[0] = dbias + cb0; slot[1] = dbias + brstride + cb0;
            if (p.ps) { slot[2] = dbias + cstride_par + cb0; slot[3] = dbias + brstride + cstride_par + cb0; }
        }
        if (dgam) { slot[4] = dgam + c; slot[5] = dbet + c; slot[6] = dgam + p.C + c; slot[7] = dbet + p.C + c; }
    }
    float racc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto reduce = [&](auto KC, float val) {
        constexpr int K = decltype(KC)::value;
        if constexpr (DET) racc[K] += val; else { if (sub == 0 && slot[K]) atomicAdd(slot[K], val); }
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>; using K4 = std::integral_constant<int, 4>; using K5 = std::integral_constant<int, 5>;
    using K6 = std::integral_constant<int, 6>; using K7 = std::integral_constant<int, 7>;

    for (int b = b0; b < b1; ++b) {
        const float* dout = p.dout + (long)g * p.do_gs + ((long)b * p.C + c) * p.L;
        float v[NE], gt[NE], dO[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int l = sub + e * LPR;
            const bool ok = l < p.L;
            v[e] = ok ? y[yoff<NB>(p, b, 0, c, l)] : 0.f;
            if (NB == 2) gt[e] = ok ? y[yoff<NB>(p, b, 1, c, l)] : 0.f;
            dO[e] = ok ? dout[l] : 0.f;
        }
        if (MODE == 1) {
            float sv = 0.f, sg = 0.f;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int l = sub + e * LPR;
                const float s = sigmoidf_(gt[e]);
                const float dv = dO[e] * s, dg = dO[e] * v[e] * s * (1.f - s);
                if (l < p.L) { dy[yoff<NB>(p, b, 0, c, l)] = dv; dy[yoff<NB>(p, b, 1, c, l)] = dg; }
                sv += dv; sg += dg;
            }
            if (dbias) { reduce(K0{}, group_sum<LPR>(sv)); reduce(K1{}, group_sum<LPR>(sg)); }
            continue;
        }
        const float* gam = p.gamma + (long)g * p.par_gs;
        const float* bet = p.beta + (long)g * p.par_gs;
        const float* st = p.stats_in + (long)g * p.st_gs + ((long)b * NB * p.C) * 2;
        const float mu = st[2 * c], rs = st[2 * c + 1], ga = gam[c], be = bet[c];

        if (MODE == 2) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < NE; ++e) { const float xh = (v[e] - mu) * rs; s1 += dO[e]; s2 += dO[e] * xh; }
            s1 = group_sum<LPR>(s1); s2 = group_sum<LPR>(s2);
            float sb = 0.f;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int l = sub + e * LPR;
                const float xh = (v[e] - mu) * rs;
                const float d = ga * rs * (dO[e] - s1 * invL - xh * s2 * invL);
                if (l < p.L) { dy[yoff<NB>(p, b, 0, c, l)] = d; sb += d; }
            }
            sb = group_sum<LPR>(sb);
            reduce(K0{}, s2); reduce(K1{}, s1); reduce(K2{}, sb);
            continue;
        }
        const float mu2 = st[2 * (p.C + c)], rs2 = st[2 * (p.C + c) + 1], ga2 = gam[p.C + c], be2 = bet[p.C + c];
        float s1v = 0.f, s2v = 0.f, s1g = 0.f, s2g = 0.f;
        float dnv[NE], dng[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float xv = (v[e] - mu) * rs, xg = (gt[e] - mu2) * rs2;
            const float nv = ga * xv + be, ng = ga2 * xg + be2;
            const float s = sigmoidf_(ng);
            dnv[e] = dO[e] * s; dng[e] = dO[e] * nv * s * (1.f - s);
            s1v += dnv[e]; s2v += dnv[e] * xv; s1g += dng[e]; s2g += dng[e] * xg;
        }
        s1v = group_sum<LPR>(s1v); s2v = group_sum<LPR>(s2v);
        s1g = group_sum<LPR>(s1g); s2g = group_sum<LPR>(s2g);
        float sbv = 0.f, sbg = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int l = sub + e * LPR;
            const float xv = (v[e] - mu) * rs, xg = (gt[e] - mu2) * rs2;
            const float dv = ga * rs * (dnv[e] - s1v * invL - xv * s2v * invL);
            const float dg = ga2 * rs2 * (dng[e] - s1g * invL - xg * s2g * invL);
            if (l < p.L) { dy[yoff<NB>(p, b, 0, c, l)] = dv; dy[yoff<NB>(p, b, 1, c, l)] = dg; sbv += dv; sbg += dg; }
        }
        if (dbias) {
            // conv-channel sums; with pixel shuffle even/odd positions belong to different conv channels
            reduce(K0{}, group_sum<LPR>((p.ps && odd) ? 0.f : sbv)); reduce(K1{}, group_sum<LPR>((p.ps && odd) ? 0.f : sbg));
            if (p.ps) { reduce(K2{}, group_sum<LPR>(odd ? sbv : 0.f)); reduce(K3{}, group_sum<LPR>(odd ? sbg : 0.f)); }
        }
        if (dgam) { reduce(K4{}, s2v); reduce(K5{}, s1v); reduce(K6{}, s2g); reduce(K7{}, s1g); }
    }
    if constexpr (DET) {
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) if (slot[k]) *slot[k] += racc[k];       // the only writer of this element in this launch
        }
    }
}

// ---- 16-byte variants for rows of up to 512 positions (every norm launch of the step except D.h1's 1536-pixel GLU): a lane
// owns NV groups of 4 consecutive positions (one f32x4 per tensor and group; with pixel shuffle two f32x2: even positions
// come from conv channel c, odd ones from channel C + c), LPR = 8 / 16 / 32 lanes a row, up to 32 rows a block.  Same
// arithmetic as norm_{fwd,bwd}_kernel, four times the bytes in flight per lane: the scalar forms leave the HBM pipe half
// empty (2.0-2.9 TB/s in profiles/r02_bf16_b64_hbm_gbs_per_kernel.txt) and sit on the critical path between two convolutions
// (measured, whole step: bf16 batch 64 14.98 -> 13.87 ms, fp32 batch 8 7.73 -> 7.53 ms with the first version: L <= 128, no shuffle).
template <int LPR> __device__ __forceinline__ float rowv_sum(float v) {
    v += dpp_read<0xB1>(v); v += dpp_read<0x4E>(v); v += dpp_read<0x141>(v);      // sums of 8 lanes
    if constexpr (LPR >= 16) v += dpp_read<0x140>(v);                             // row of 16
    if constexpr (LPR >= 32) v += __shfl_xor(v, 16, 64);
    return v;
}
__device__ __forceinline__ float hsum4(const f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// 4 consecutive logical positions starting at pos4 of (sample b, branch br, channel c) in the raw conv layout
template <int NB, bool PS> struct RowV4 {
    long even, odd; int C, L;      // float offsets of the row (PS: of conv channels c and C + c, rows of L / 2)
    __device__ RowV4(const NormArgs& p, int b, int br, int c) : C(p.C), L(p.L) {
        if (PS) { const int T = p.L >> 1; even = ((long)b * NB * 2 * p.C + br * 2 * p.C + c) * T; odd = even + (long)p.C * T; }
        else { even = ((long)b * NB * p.C + br * p.C + c) * p.L; odd = 0; }
    }
    __device__ f32x4 load(const float* base, int pos4) const {
        if (PS) {
            const f32x2 e = *reinterpret_cast<const f32x2*>(base + even + (pos4 >> 1)), o = *reinterpret_cast<const f32x2*>(base + odd + (pos4 >> 1));
            return f32x4{e[0], o[0], e[1], o[1]};
        }
        return *reinterpret_cast<const f32x4*>(base + even + pos4);
    }
    __device__ void store(float* base, int pos4, const f32x4 v) const {
        if (PS) {
            *reinterpret_cast<f32x2*>(base + even + (pos4 >> 1)) = f32x2{v[0], v[2]};
            *reinterpret_cast<f32x2*>(base + odd + (pos4 >> 1)) = f32x2{v[1], v[3]};
        } else *reinterpret_cast<f32x4*>(base + even + pos4) = v;
    }
};

// The bf16 mode's convolutions read their gathered operand as channel octets, [b][C/8][position][8] bf16 (16-byte units: 8 channels of
// one position; csrc/vc_conv_bf16.hip).  Round 4 wrote every activation as fp32 and re-read it in a separate launch to build that
// form (pack_act_bf16_kernel: 77 launches and 1.06 ms of the batch-64 step, 6 bytes of HBM traffic per element).  The kernels that
// PRODUCE a tensor now emit it: a block's 256 / LPR rows are consecutive channels of one sample -- whole octets --, each lane drops
// its 4 NV values as bf16 (RNE, the conversion the pack kernel did) into an LDS tile [octet][position][8], and the block writes the
// tile out as 16-byte units, lanes along the positions.  `rowv` = the lane's row exists; rows of a block are all or nothing per octet.
// One PLANE of a block: its 256 / LPR rows as consecutive channels of a channel-octet tensor with Lc positions a row.  SHUF = false:
// a lane's value j holds positions 4 (sub + j LPR) .. + 3.  SHUF = true (the gradient of a pixel-shuffled layer, written in the
// convolution's own layout): the plane is the even (which = 0) or odd (which = 1) positions -- conv channel c or C + c --, a lane's
// value j holds positions 2 (sub + j LPR), + 1 of it in elements which and which + 2.
template <int LPR, int NV, bool SHUF>
__device__ __forceinline__ void emit_channel_octets(const f32x4 (&o)[NV], const bool (&ok)[NV], bool rowv, int which, int Lc,
                                                    u32x4* dst /* unit of (sample, first octet of the block, position 0) */, int octs_valid) {
    constexpr int R = 256 / LPR, OCTS = R / 8, LMAX = LPR * 4 * NV;
    __shared__ __attribute__((aligned(16))) unsigned short tile[OCTS * LMAX * 8];
    const int r = threadIdx.x / LPR, sub = threadIdx.x % LPR;
    __syncthreads();                                        // an earlier plane of this block has left the tile
    if (rowv) {
#pragma unroll
        for (int j = 0; j < NV; ++j) if (ok[j]) {
            if constexpr (SHUF) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    tile[((r >> 3) * LMAX + (sub + j * LPR) * 2 + k) * 8 + (r & 7)] = __builtin_bit_cast(unsigned short, (__bf16)o[j][which + 2 * k]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    tile[((r >> 3) * LMAX + (sub + j * LPR) * 4 + e) * 8 + (r & 7)] = __builtin_bit_cast(unsigned short, (__bf16)o[j][e]);
            }
        }
    }
    __syncthreads();
    for (int u = threadIdx.x; u < octs_valid * Lc; u += 256) {
        const int oc = u / Lc, pos = u - oc * Lc;
        dst[(long)oc * Lc + pos] = *reinterpret_cast<const u32x4*>(tile + (oc * LMAX + pos) * 8);
    }
}

// The filter-gradient GEMM of the bf16 mode reduces over pixels and reads dY as PIXEL octets, dYq[pixel octet][conv channel][8] bf16 (16-byte
// units: 8 consecutive pixels of one channel; csrc/vc_ops.hip: conv_wgrad).  Round 4 / early round 5 re-read the fp32 dY in a pack launch per
// filter gradient (37 launches, 0.47 ms of the batch-64 step, 6 bytes of traffic per element); the kernel that produces dY writes this form
// too: a block's rows are consecutive channels of one sample, a lane's values are 4 (shuffled: 2) consecutive pixels of its row -- half (a
// quarter) of a unit -- dropped as bf16 (RNE, the pack kernel's conversion) into an LDS tile [octet][row][8]; the block then writes, per
// pixel octet, its rows' units side by side: 256 / LPR x 16 contiguous bytes.  Same plane / `which` conventions as emit_channel_octets.
template <int LPR, int NV, bool SHUF>
__device__ __forceinline__ void emit_pixel_octets(const f32x4 (&o)[NV], const bool (&ok)[NV], bool rowv, int which, int Lc,
                                                  u32x4* dst /* unit of (the sample's pixel octet 0, first channel of the block's plane) */,
                                                  long po_stride /* units between consecutive pixel octets = the convolution's channel count */, int rows_valid) {
    constexpr int R = 256 / LPR, LMAX = LPR * 4 * NV;
    __shared__ __attribute__((aligned(16))) unsigned short tile[(LMAX / 8) * R * 8];
    const int r = threadIdx.x / LPR, sub = threadIdx.x % LPR;
    __syncthreads();                                        // an earlier plane of this block has left the tile
    if (rowv) {
#pragma unroll
        for (int j = 0; j < NV; ++j) if (ok[j]) {
            if constexpr (SHUF) {
                const int pos = (sub + j * LPR) * 2;
                unsigned short* q = tile + ((pos >> 3) * R + r) * 8 + (pos & 7);
                q[0] = __builtin_bit_cast(unsigned short, (__bf16)o[j][which]); q[1] = __builtin_bit_cast(unsigned short, (__bf16)o[j][which + 2]);
            } else {
                const int pos = (sub + j * LPR) * 4;
                unsigned short* q = tile + ((pos >> 3) * R + r) * 8 + (pos & 7);
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = __builtin_bit_cast(unsigned short, (__bf16)o[j][e]);
            }
        }
    }
    __syncthreads();
    const int npo = Lc >> 3;
    for (int u = threadIdx.x; u < npo * rows_valid; u += 256) {
        const int po = u / rows_valid, rr = u - po * rows_valid;
        dst[(long)po * po_stride + rr] = *reinterpret_cast<const u32x4*>(tile + (po * R + rr) * 8);
    }
}

template <int MODE, int LPR, int NV, bool PS>
__global__ __launch_bounds__(256) void norm_fwd_v4_kernel(const NormArgs p) {
    touch_kernarg(p);
    constexpr int NB = (MODE == 2) ? 1 : 2;
    const int unit0 = blockIdx.x * (256 / LPR);
    const int unit = unit0 + threadIdx.x / LPR, sub = threadIdx.x % LPR;
    const int total = p.G * p.B * p.C;
    const bool rowv = unit < total;
    if (!rowv && !p.outp) return;                           // (with a packed output every thread meets the block's barrier)
    const int uu = rowv ? unit : 0;
    const int g = uu / (p.B * p.C); const int rem = uu - g * (p.B * p.C);
    const int b = rem / p.C, c = rem - b * p.C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float* y = p.y + (long)g * p.y_gs;
    float* out = p.out + (long)g * p.out_gs + ((long)b * p.C + c) * p.L;
    const float invL = 1.f / (float)p.L;
    const RowV4<NB, PS> r0(p, b, 0, c), r1(p, b, NB - 1, c);
    f32x4 v[NV], gt[NV], o[NV]; bool ok[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int pos4 = (sub + j * LPR) * 4;
        ok[j] = rowv && pos4 < p.L;
        o[j] = zero;
        if constexpr (NV == 1 && !PS) {
            if (p.win) {       // Winograd route: this row's quad = A^T of its six transform-domain values + the conv bias; the raw value is kept for the backward
                const int nt = p.L >> 2;
                const long plane = (long)p.B * NB * p.C * nt;
                const float* m = p.win + (long)g * p.win_gs + ((long)b * NB * p.C + c) * nt + sub;
                const float* wb = p.wbias + (long)g * p.wbias_gs;
                float* yw = p.yw + (long)g * p.y_gs;
                v[j] = zero; gt[j] = zero;
                if (ok[j]) {
                    v[j] = wino_load_output(m, plane, wb[c]); r0.store(yw, pos4, v[j]);
                    if (NB == 2) { gt[j] = wino_load_output(m + (long)p.C * nt, plane, wb[p.C + c]); r1.store(yw, pos4, gt[j]); }
                }
                continue;
            }
        }
        v[j] = ok[j] ? r0.load(y, pos4) : zero;
        gt[j] = (NB == 2 && ok[j]) ? r1.load(y, pos4) : zero;
    }
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[j][e] = v[j][e] * sigmoidf_(gt[j][e]);
    } else {
        const float* gam = p.gamma + (long)g * p.par_gs;
        const float* bet = p.beta + (long)g * p.par_gs;
        float* st = p.stats + (long)g * p.st_gs + ((long)b * NB * p.C) * 2;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += hsum4(v[j]);
        const float mu = rowv_sum<LPR>(s) * invL;
        float q = 0.f; f32x4 d[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { d[j] = ok[j] ? v[j] - mu : zero; q += hsum4(d[j] * d[j]); }
        const float rs = rsqrtf(rowv_sum<LPR>(q) * invL + IN_EPS);
        const float ga = gam[c], be = bet[c];
        if (sub == 0 && rowv) { st[2 * c] = mu; st[2 * c + 1] = rs; }
        if (MODE == 2) {
            const float* res = p.res + (long)g * p.res_gs + ((long)b * p.C + c) * p.L;
#pragma unroll
            for (int j = 0; j < NV; ++j) if (ok[j]) o[j] = *reinterpret_cast<const f32x4*>(res + (sub + j * LPR) * 4) + (d[j] * rs * ga + be);
        } else {
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) s2 += hsum4(gt[j]);
            const float mu2 = rowv_sum<LPR>(s2) * invL;
            float q2 = 0.f; f32x4 d2[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) { d2[j] = ok[j] ? gt[j] - mu2 : zero; q2 += hsum4(d2[j] * d2[j]); }
            const float rs2 = rsqrtf(rowv_sum<LPR>(q2) * invL + IN_EPS);
            const float ga2 = gam[p.C + c], be2 = bet[p.C + c];
            if (sub == 0 && rowv) { st[2 * (p.C + c)] = mu2; st[2 * (p.C + c) + 1] = rs2; }
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[j][e] = (d[j][e] * rs * ga + be) * sigmoidf_(d2[j][e] * rs2 * ga2 + be2);
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) if (ok[j]) *reinterpret_cast<f32x4*>(out + (sub + j * LPR) * 4) = o[j];
    if constexpr (NV == 1 && !PS) {
        if (p.wout) {          // the activation once more in the next layer's Winograd domain: V = B^T (x[4i-1], quad, x[4i+4]), neighbours from the lanes beside
            const int nt = p.L >> 2;
            const float left = __shfl_up(o[0][3], 1, LPR), right = __shfl_down(o[0][0], 1, LPR);
            if (ok[0]) wino_store_input(o[0], sub > 0 ? left : 0.f, sub + 1 < nt ? right : 0.f,
                                        p.wout + (long)g * p.wout_gs + ((long)b * p.C + c) * nt + sub, (long)p.B * p.C * nt);
        }
    }
    if (p.outp) {
        // the block's rows: units unit0 .. unit0 + 256 / LPR - 1 = consecutive channels of ONE (group, sample) (the launcher checks C % (256 / LPR) == 0)
        const int u0 = unit0 < total ? unit0 : 0;
        const int g0 = u0 / (p.B * p.C); const int rem0 = u0 - g0 * (p.B * p.C);
        const int b0 = rem0 / p.C, c0 = rem0 - b0 * p.C;
        const int left = (total - unit0) / 8;
        u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.outp) + (long)g0 * p.outp_gs) + ((long)b0 * (p.C / 8) + c0 / 8) * p.L;
        emit_channel_octets<LPR, NV, false>(o, ok, rowv, 0, p.L, dst, left < (256 / LPR) / 8 ? (left > 0 ? left : 0) : (256 / LPR) / 8);
    }
}

template <int MODE, int LPR, int NV, bool PS>
__global__ __launch_bounds__(256) void norm_bwd_v4_kernel(const NormArgs p) {
    touch_kernarg(p);
    constexpr int NB = (MODE == 2) ? 1 : 2;
    const int unit0 = blockIdx.x * (256 / LPR);
    const int unit = unit0 + threadIdx.x / LPR, sub = threadIdx.x % LPR;
    const int total = p.G * p.B * p.C;
    const bool rowv = unit < total;
    if (!rowv && !p.outp && !p.outq) return;                // (with a packed output every thread meets the block's barriers)
    const int uu = rowv ? unit : 0;
    const int g = uu / (p.B * p.C); const int rem = uu - g * (p.B * p.C);
    const int b = rem / p.C, c = rem - b * p.C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float* y = p.y + (long)g * p.y_gs;
    const bool dy_on = p.dy != nullptr;                      // bf16 data path inside a train step: both readers of dy take the octet forms, the fp32 copy is not written
    float* dy = p.dy + (long)g * p.dy_gs;
    const float* dout = p.dout + (long)g * p.do_gs + ((long)b * p.C + c) * p.L;
    // dy once more as bf16 channel octets in the convolution's layout [b][Cout / 8][OH OW][8] (what its input gradient gathers): branch br
    // = conv channels [br Cc, (br + 1) Cc), Cc = C (2 C when shuffled: even positions channel c, odd ones C + c)
    auto emit_dy = [&](const f32x4 (&val)[NV], const bool (&okv)[NV], int br) {
        const int Cc = PS ? 2 * p.C : p.C, Lc = PS ? p.L >> 1 : p.L, Cout8 = NB * Cc / 8;
        const int u0 = unit0 < total ? unit0 : 0;
        const int g0 = u0 / (p.B * p.C); const int rem0 = u0 - g0 * (p.B * p.C);
        const int b0 = rem0 / p.C, c0 = rem0 - b0 * p.C;
        const int left = (total - unit0) / 8, nv = left < (256 / LPR) / 8 ? (left > 0 ? left : 0) : (256 / LPR) / 8;
#pragma unroll
        for (int which = 0; which < (PS ? 2 : 1); ++which) {
            const int chb = br * Cc + which * p.C + c0;
            if (p.outp) {
                u32x4* base = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.outp) + (long)g0 * p.outp_gs);
                emit_channel_octets<LPR, NV, PS>(val, okv, rowv, which, Lc, base + ((long)b0 * Cout8 + chb / 8) * Lc, nv);
            }
            if (p.outq) {       // [outq_po0 + the sample's first pixel octet + po][conv channel][8]
                u32x4* base = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.outq) + (long)g0 * p.outq_gs);
                const int rows = total - unit0 < 256 / LPR ? (total > unit0 ? total - unit0 : 0) : 256 / LPR;
                emit_pixel_octets<LPR, NV, PS>(val, okv, rowv, which, Lc, base + ((long)p.outq_po0 + (long)b0 * (Lc >> 3)) * (NB * Cc) + chb, (long)NB * Cc, rows);
            }
        }
    };
    const float invL = 1.f / (float)p.L;
    const RowV4<NB, PS> r0(p, b, 0, c), r1(p, b, NB - 1, c);
    float* dbias = p.dbias ? p.dbias + (long)g * p.dbias_gs : nullptr;
    // conv channels whose bias gradient this row feeds: branch br, even positions (or all) -> cb(br), odd (shuffled) -> + C
    const int brs = PS ? 2 * p.C : p.C;
    f32x4 v[NV], gt[NV], dO[NV]; bool ok[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int pos4 = (sub + j * LPR) * 4;
        ok[j] = rowv && pos4 < p.L;
        v[j] = ok[j] ? r0.load(y, pos4) : zero;
        gt[j] = (NB == 2 && ok[j]) ? r1.load(y, pos4) : zero;
        if constexpr (NV == 1 && !PS) {
            if (p.dwin) {      // Winograd route: the gradient of this activation arrives as the next layer's dV planes: dO = B dV (+ the skip path)
                const int nt = p.L >> 2;
                dO[j] = zero;
                if (ok[j]) {
                    dO[j] = wino_load_dx(p.dwin + (long)g * p.dwin_gs + ((long)b * p.C + c) * nt + sub, (long)p.B * p.C * nt, sub > 0, sub + 1 < nt);
                    const long o = ((long)b * p.C + c) * p.L + pos4;
                    if (p.dadd) dO[j] += *reinterpret_cast<const f32x4*>(p.dadd + (long)g * p.dadd_gs + o);
                    if (p.dsave) *reinterpret_cast<f32x4*>(p.dsave + (long)g * p.dsave_gs + o) = dO[j];
                }
                continue;
            }
        }
        dO[j] = ok[j] ? *reinterpret_cast<const f32x4*>(dout + pos4) : zero;
    }
    // dy once more in this layer's Winograd domain (dM = A dy), for its input-gradient GEMMs: branch br = conv channels [br C, (br + 1) C)
    auto emit_dm = [&](const f32x4 (&val)[NV], int br) {
        if constexpr (NV == 1 && !PS) {
            if (p.dwout && ok[0]) {
                const int nt = p.L >> 2;
                wino_store_dy(val[0], p.dwout + (long)g * p.dwout_gs + ((long)b * NB * p.C + br * p.C + c) * nt + sub, (long)p.B * NB * p.C * nt);
            }
        }
    };
    auto bias_sums = [&](const f32x4 (&dv)[NV], int br) {          // conv-bias gradient of one branch (even / odd positions apart when shuffled)
        float se = 0.f, so = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) { se += PS ? dv[j][0] + dv[j][2] : hsum4(dv[j]); so += PS ? dv[j][1] + dv[j][3] : 0.f; }
        se = rowv_sum<LPR>(se); if (PS) so = rowv_sum<LPR>(so);
        if (sub == 0 && rowv) { atomicAdd(dbias + br * brs + c, se); if (PS) atomicAdd(dbias + br * brs + p.C + c, so); }
    };
    if (MODE == 1) {
        f32x4 dv[NV], dg[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float sg = sigmoidf_(gt[j][e]); dv[j][e] = dO[j][e] * sg; dg[j][e] = dO[j][e] * v[j][e] * sg * (1.f - sg); }
            if (ok[j] && dy_on) { r0.store(dy, (sub + j * LPR) * 4, dv[j]); r1.store(dy, (sub + j * LPR) * 4, dg[j]); }
        }
        if (dbias) { bias_sums(dv, 0); bias_sums(dg, 1); }
        if (p.outp || p.outq) { emit_dy(dv, ok, 0); emit_dy(dg, ok, 1); }
        return;
    }
    const float* gam = p.gamma + (long)g * p.par_gs;
    const float* bet = p.beta + (long)g * p.par_gs;
    const float* st = p.stats_in + (long)g * p.st_gs + ((long)b * NB * p.C) * 2;
    float* dgam = p.dgamma ? p.dgamma + (long)g * p.dpar_gs : nullptr;
    float* dbet = p.dbeta ? p.dbeta + (long)g * p.dpar_gs : nullptr;
    const float mu = st[2 * c], rs = st[2 * c + 1], ga = gam[c], be = bet[c];
    f32x4 xv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) xv[j] = ok[j] ? (v[j] - mu) * rs : zero;
    if (MODE == 2) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) { s1 += hsum4(dO[j]); s2 += hsum4(dO[j] * xv[j]); }
        s1 = rowv_sum<LPR>(s1); s2 = rowv_sum<LPR>(s2);
        f32x4 d[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            d[j] = ok[j] ? (dO[j] - s1 * invL - xv[j] * (s2 * invL)) * (ga * rs) : zero;
            if (ok[j] && dy_on) r0.store(dy, (sub + j * LPR) * 4, d[j]);
        }
        if (dbias) bias_sums(d, 0);
        if (sub == 0 && dgam && rowv) { atomicAdd(dgam + c, s2); atomicAdd(dbet + c, s1); }
        emit_dm(d, 0);
        if (p.outp || p.outq) emit_dy(d, ok, 0);
        return;
    }
    const float mu2 = st[2 * (p.C + c)], rs2 = st[2 * (p.C + c) + 1], ga2 = gam[p.C + c], be2 = bet[p.C + c];
    f32x4 xg[NV], dnv[NV], dng[NV];
    float s1v = 0.f, s2v = 0.f, s1g = 0.f, s2g = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        xg[j] = ok[j] ? (gt[j] - mu2) * rs2 : zero;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float nv = ga * xv[j][e] + be, ng = ga2 * xg[j][e] + be2;
            const float sg = sigmoidf_(ng);
            dnv[j][e] = dO[j][e] * sg; dng[j][e] = dO[j][e] * nv * sg * (1.f - sg);
        }
        s1v += hsum4(dnv[j]); s2v += hsum4(dnv[j] * xv[j]); s1g += hsum4(dng[j]); s2g += hsum4(dng[j] * xg[j]);
    }
    s1v = rowv_sum<LPR>(s1v); s2v = rowv_sum<LPR>(s2v); s1g = rowv_sum<LPR>(s1g); s2g = rowv_sum<LPR>(s2g);
    f32x4 dv[NV], dg[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        dv[j] = ok[j] ? (dnv[j] - s1v * invL - xv[j] * (s2v * invL)) * (ga * rs) : zero;
        dg[j] = ok[j] ? (dng[j] - s1g * invL - xg[j] * (s2g * invL)) * (ga2 * rs2) : zero;
        if (ok[j] && dy_on) { r0.store(dy, (sub + j * LPR) * 4, dv[j]); r1.store(dy, (sub + j * LPR) * 4, dg[j]); }
    }
    if (dbias) { bias_sums(dv, 0); bias_sums(dg, 1); }
    if (sub == 0 && dgam && rowv) {
        atomicAdd(dgam + c, s2v); atomicAdd(dbet + c, s1v);
        atomicAdd(dgam + p.C + c, s2g); atomicAdd(dbet + p.C + c, s1g);
    }
    emit_dm(dv, 0); emit_dm(dg, 1);
    if (p.outp || p.outq) { emit_dy(dv, ok, 0); emit_dy(dg, ok, 1); }
}

// GLU backward without norm (module.py:201-203 / 161-163: the two h1 blocks) for rows of ANY length that is a multiple of 4
// (D.h1: 1536 pixels at 24x128, 4608 at 36x256): no row statistics, so the row is streamed, 32 lanes x 16 bytes a trip
__global__ __launch_bounds__(256) void glu_bwd_v4_kernel(const NormArgs p) {
    touch_kernarg(p);
    constexpr int LPR = 32;
    const int unit = blockIdx.x * (256 / LPR) + threadIdx.x / LPR, sub = threadIdx.x % LPR;
    if (unit >= p.G * p.B * p.C) return;
    const int g = unit / (p.B * p.C); const int rem = unit - g * (p.B * p.C);
    const int b = rem / p.C, c = rem - b * p.C;
    const long rv = ((long)b * 2 * p.C + c) * p.L, rg = rv + (long)p.C * p.L;
    const float* y = p.y + (long)g * p.y_gs;
    float* dy = p.dy + (long)g * p.dy_gs;
    const float* dout = p.dout + (long)g * p.do_gs + ((long)b * p.C + c) * p.L;
    float sv = 0.f, sg = 0.f;
#pragma unroll 2
    for (int pos4 = sub * 4; pos4 < p.L; pos4 += LPR * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + rv + pos4), gt = *reinterpret_cast<const f32x4*>(y + rg + pos4);
        const f32x4 dO = *reinterpret_cast<const f32x4*>(dout + pos4);
        f32x4 dv, dg;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float s = sigmoidf_(gt[e]); dv[e] = dO[e] * s; dg[e] = dO[e] * v[e] * s * (1.f - s); }
        *reinterpret_cast<f32x4*>(dy + rv + pos4) = dv; *reinterpret_cast<f32x4*>(dy + rg + pos4) = dg;
        sv += hsum4(dv); sg += hsum4(dg);
    }
    if (p.dbias) {
        float* dbias = p.dbias + (long)g * p.dbias_gs;
        sv = rowv_sum<LPR>(sv); sg = rowv_sum<LPR>(sg);
        if (sub == 0) { atomicAdd(dbias + c, sv); atomicAdd(dbias + p.C + c, sg); }
    }
}

// ---- streaming variants for rows longer than 2048 (36x256 discriminator, long utterances): one 256-thread block
// per (g, b, c) row, the row is re-read from L2 for each pass instead of living in registers.
__device__ __forceinline__ float block_sum_256s(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <int MODE>
__global__ __launch_bounds__(256) void norm_fwd_stream_kernel(const NormArgs p) {
    touch_kernarg(p);
    constexpr int NB = (MODE == 2) ? 1 : 2;
    __shared__ float red[4];
    const int unit = blockIdx.x;
    const int g = unit / (p.B * p.C); const int rem = unit - g * (p.B * p.C);
    const int b = rem / p.C, c = rem - b * p.C;
    const float* y = p.y + (long)g * p.y_gs;
    float* out = p.out + (long)g * p.out_gs + ((long)b * p.C + c) * p.L;
    const float invL = 1.f / (float)p.L;
    if (MODE == 1) {
        for (int l = threadIdx.x; l < p.L; l += 256) out[l] = y[yoff<NB>(p, b, 0, c, l)] * sigmoidf_(y[yoff<NB>(p, b, 1, c, l)]);
        return;
    }
    const float* gam = p.gamma + (long)g * p.par_gs;
    const float* bet = p.beta + (long)g * p.par_gs;
    float* st = p.stats + (long)g * p.st_gs + ((long)b * NB * p.C) * 2;
    float mu[NB], rs[NB];
#pragma unroll
    for (int br = 0; br < NB; ++br) {
        float s = 0.f;
        for (int l = threadIdx.x; l < p.L; l += 256) s += y[yoff<NB>(p, b, br, c, l)];
        mu[br] = block_sum_256s(s, red) * invL;
        float q = 0.f;
        for (int l = threadIdx.x; l < p.L; l += 256) { const float d = y[yoff<NB>(p, b, br, c, l)] - mu[br]; q += d * d; }
        rs[br] = rsqrtf(block_sum_256s(q, red) * invL + IN_EPS);
        if (threadIdx.x == 0) { st[2 * (br * p.C + c)] = mu[br]; st[2 * (br * p.C + c) + 1] = rs[br]; }
    }
    const float ga = gam[c], be = bet[c];
    if (MODE == 2) {
        const float* res = p.res + (long)g * p.res_gs + ((long)b * p.C + c) * p.L;
        for (int l = threadIdx.x; l < p.L; l += 256) out[l] = res[l] + ((y[yoff<NB>(p, b, 0, c, l)] - mu[0]) * rs[0] * ga + be);
    } else {
        const float ga2 = gam[p.C + c], be2 = bet[p.C + c];
        for (int l = threadIdx.x; l < p.L; l += 256)
            out[l] = ((y[yoff<NB>(p, b, 0, c, l)] - mu[0]) * rs[0] * ga + be) *
                     sigmoidf_((y[yoff<NB>(p, b, NB - 1, c, l)] - mu[NB - 1]) * rs[NB - 1] * ga2 + be2);
    }
}

template <int MODE, bool DET>
__global__ __launch_bounds__(256) void norm_bwd_stream_kernel(const NormArgs p) {
    touch_kernarg(p);
    constexpr int NB = (MODE == 2) ? 1 : 2;
    __shared__ float red[4];
    const int unit = blockIdx.x;
    const int rows = DET ? p.C : p.B * p.C;          // DET: a block owns channel c of every sample, walked in order
    const int g = unit / rows; const int rem = unit - g * rows;
    const int b0 = DET ? 0 : rem / p.C, c = DET ? rem : rem - b0 * p.C;
    const int b1 = DET ? p.B : b0 + 1;
    const float* y = p.y + (long)g * p.y_gs;
    float* dy = p.dy + (long)g * p.dy_gs;
    const float invL = 1.f / (float)p.L;
    float* dbias = p.dbias ? p.dbias + (long)g * p.dbias_gs : nullptr;
    const int cstride_par = p.ps ? p.C : 0, brstride = p.ps ? 2 * p.C : p.C;
    float* dgam = (MODE != 1 && p.dgamma) ? p.dgamma + (long)g * p.dpar_gs : nullptr;
    float* dbet = (MODE != 1 && p.dbeta) ? p.dbeta + (long)g * p.dpar_gs : nullptr;
    float tg[NB], tb[NB], te[NB], to[NB];            // DET: batch totals of dgamma, dbeta, dbias (even / odd positions)
#pragma unroll
    for (int br = 0; br < NB; ++br) tg[br] = tb[br] = te[br] = to[br] = 0.f;
    for (int b = b0; b < b1; ++b) {
        const float* dout = p.dout + (long)g * p.do_gs + ((long)b * p.C + c) * p.L;
        // conv-channel bias sums: [branch][parity]; without pixel shuffle only parity 0 is used
        float sb[NB][2];
#pragma unroll
        for (int br = 0; br < NB; ++br) sb[br][0] = sb[br][1] = 0.f;
        if (MODE == 1) {
            for (int l = threadIdx.x; l < p.L; l += 256) {
                const float v = y[yoff<NB>(p, b, 0, c, l)], s = sigmoidf_(y[yoff<NB>(p, b, 1, c, l)]);
                const float dv = dout[l] * s, dg = dout[l] * v * s * (1.f - s);
                dy[yoff<NB>(p, b, 0, c, l)] = dv; dy[yoff<NB>(p, b, 1, c, l)] = dg;
                sb[0][0] += dv; sb[NB - 1][0] += (NB == 2) ? dg : 0.f;
            }
        } else {
            const float* gam = p.gamma + (long)g * p.par_gs;
            const float* bet = p.beta + (long)g * p.par_gs;
            const float* st = p.stats_in + (long)g * p.st_gs + ((long)b * NB * p.C) * 2;
            float mu[NB], rs[NB], ga[NB], be[NB], s1[NB], s2[NB];
#pragma unroll
            for (int br = 0; br < NB; ++br) {
                mu[br] = st[2 * (br * p.C + c)]; rs[br] = st[2 * (br * p.C + c) + 1];
                ga[br] = gam[br * p.C + c]; be[br] = bet[br * p.C + c]; s1[br] = s2[br] = 0.f;
            }
            auto dn = [&](int l, float* d) {     // gradient w.r.t. the normalised value(s) at position l
                const float dO = dout[l];
                if (MODE == 2) { d[0] = dO; return; }
                const float nv = ga[0] * (y[yoff<NB>(p, b, 0, c, l)] - mu[0]) * rs[0] + be[0];
                const float ng = ga[NB - 1] * (y[yoff<NB>(p, b, NB - 1, c, l)] - mu[NB - 1]) * rs[NB - 1] + be[NB - 1];
                const float s = sigmoidf_(ng);
                d[0] = dO * s; d[NB - 1] = dO * nv * s * (1.f - s);
            };
            for (int l = threadIdx.x; l < p.L; l += 256) {
                float d[NB]; dn(l, d);
#pragma unroll
                for (int br = 0; br < NB; ++br) { s1[br] += d[br]; s2[br] += d[br] * (y[yoff<NB>(p, b, br, c, l)] - mu[br]) * rs[br]; }
            }
#pragma unroll
            for (int br = 0; br < NB; ++br) { s1[br] = block_sum_256s(s1[br], red); s2[br] = block_sum_256s(s2[br], red); }
            for (int l = threadIdx.x; l < p.L; l += 256) {
                float d[NB]; dn(l, d);
#pragma unroll
                for (int br = 0; br < NB; ++br) {
                    const float xh = (y[yoff<NB>(p, b, br, c, l)] - mu[br]) * rs[br];
                    const float v = ga[br] * rs[br] * (d[br] - s1[br] * invL - xh * s2[br] * invL);
                    dy[yoff<NB>(p, b, br, c, l)] = v;
                    sb[br][p.ps ? (l & 1) : 0] += v;
                }
            }
            if (dgam) {
#pragma unroll
                for (int br = 0; br < NB; ++br) {
                    if constexpr (DET) { tg[br] += s2[br]; tb[br] += s1[br]; }
                    else if (threadIdx.x == 0) { atomicAdd(dgam + br * p.C + c, s2[br]); atomicAdd(dbet + br * p.C + c, s1[br]); }
                }
            }
        }
        if (dbias) {
#pragma unroll
            for (int br = 0; br < NB; ++br) {
                const float e = block_sum_256s(sb[br][0], red), o = block_sum_256s(sb[br][1], red);
                if constexpr (DET) { te[br] += e; to[br] += o; }
                else if (threadIdx.x == 0) {
                    atomicAdd(dbias + br * brstride + c, e);
                    if (p.ps) atomicAdd(dbias + br * brstride + cstride_par + c, o);
                }
            }
        }
    }
    if constexpr (DET) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int br = 0; br < NB; ++br) {
                if (dgam) { dgam[br * p.C + c] += tg[br]; dbet[br * p.C + c] += tb[br]; }
                if (dbias) { dbias[br * brstride + c] += te[br]; if (p.ps) dbias[br * brstride + cstride_par + c] += to[br]; }
            }
        }
    }
}

__global__ __launch_bounds__(256) void bias_grad_kernel(const float* dy, long dy_gs, float* db, long db_gs,
                                                        int B, int C, int L) {
    __shared__ float red[4];
    const int c = blockIdx.x, g = blockIdx.y;
    const float* d = dy + (long)g * dy_gs;
    float s = 0.f;
    const int n = B * L;
    for (int i = threadIdx.x; i < n; i += 256) { const int b = i / L, l = i - b * L; s += d[((long)b * C + c) * L + l]; }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(db + (long)g * db_gs + c, s);
}

// 16 waves a block: lanes along 64 pixels, the waves split the channels, 8 loads in flight per lane (the round-2 form -- 4
// waves, one dependent load per trip -- took 63 us for 6 MB at the B = 8 step, on the critical path between D.d3 and the losses)
#define VC_HEAD_WAVES 16
__global__ __launch_bounds__(64 * VC_HEAD_WAVES) void head_fwd_kernel(const float* x, long x_gs, const float* w, const float* bias,
                                                                    long par_gs, float* o, long o_gs, int C, int HW) {
    __shared__ float red[VC_HEAD_WAVES][64];
    const int g = blockIdx.z, b = blockIdx.y;
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), cq = threadIdx.x >> 6;
    const float* xb = x + (long)g * x_gs + (long)b * C * HW;
    const float* wg = w + (long)g * par_gs;
    float s = 0.f;
    if (px < HW) {
        int c = cq;
#pragma unroll 1
        for (; c + 7 * VC_HEAD_WAVES < C; c += 8 * VC_HEAD_WAVES) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = xb[(long)(c + u * VC_HEAD_WAVES) * HW + px];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u] * wg[c + u * VC_HEAD_WAVES];
        }
        for (; c < C; c += VC_HEAD_WAVES) s += xb[(long)c * HW + px] * wg[c];
    }
    red[cq][threadIdx.x & 63] = s;
    __syncthreads();
    if (cq == 0 && px < HW) {
        float z = bias[(long)g * par_gs];
#pragma unroll
        for (int q = 0; q < VC_HEAD_WAVES; ++q) z += red[q][px & 63];
        o[(long)g * o_gs + (long)b * HW + px] = sigmoidf_(z);
    }
}

// dx[b][c][p] = dz[b][p] * w[c], dz = dout*o*(1-o): a wave per (sample, channel) row, lanes along the pixels (32-bit index
// arithmetic only: the flat 64-bit div / mod of the round-2 form cost more than the 3 MB it wrote)
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* w, long par_gs, const float* o, long o_gs,
                                                          const float* dout, long do_gs, float* dx, long dx_gs, int B, int C, int HW) {
    const int g = blockIdx.y, lane = threadIdx.x & 63;
    const int rows = B * C;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const int b = row / C, c = row - b * C;
        const float wc = w[(long)g * par_gs + c];
        for (int pxl = lane; pxl < HW; pxl += 64) {
            const float ov = o[(long)g * o_gs + (long)b * HW + pxl];
            const float dz = dout[(long)g * do_gs + (long)b * HW + pxl] * ov * (1.f - ov);
            dx[(long)g * dx_gs + (long)row * HW + pxl] = dz * wc;
        }
    }
}
// one wave per channel: dw[c] += sum_{b,p} dz*x ; the wave of channel 0 also does db
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const float* x, long x_gs, const float* o, long o_gs, const float* dout,
                                                          long do_gs, float* dw, float* db, long dpar_gs, int B, int C, int HW) {
    const int g = blockIdx.y;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    const int n = B * HW;
    float s = 0.f, sb = 0.f;
    for (int i = lane; i < n; i += 64) {
        const int b = i / HW, pxl = i - b * HW;
        const float ov = o[(long)g * o_gs + i];
        const float dz = dout[(long)g * do_gs + i] * ov * (1.f - ov);
        s += dz * x[(long)g * x_gs + ((long)b * C + c) * HW + pxl];
        sb += dz;
    }
    s = wave_sum(s);
    if (lane == 0) atomicAdd(dw + (long)g * dpar_gs + c, s);
    if (c == 0) { sb = wave_sum(sb); if (lane == 0) atomicAdd(db + (long)g * dpar_gs, sb); }
}

// The Adam slots m and v are pure streams (read once, written once per step, 2 x 479 MB): non-temporal loads / stores
// keep them from evicting the weights and gradients the rest of the step re-reads from L2 / Infinity Cache.  Two float4
// groups per thread and trip: twice the bytes in flight.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, long n4, long n, float alpha_arg,
                                                   float b1, float b2, float eps, float gs, const float* __restrict__ alpha_dev) {
    const float alpha = alpha_dev ? *alpha_dev : alpha_arg;      // captured steps (hipGraph) read the step size from device memory
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    auto ld = [](const f32x4* q) { return __builtin_nontemporal_load(q); };
    auto st = [](f32x4* q, const f32x4 x) { __builtin_nontemporal_store(x, q); };
    auto upd = [&](f32x4& P, f32x4& M, f32x4& V, const f32x4 G) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = G[e] * gs;
            M[e] += (gg - M[e]) * omb1; V[e] += (gg * gg - V[e]) * omb2;
            P[e] -= (M[e] * alpha) / (sqrtf(V[e]) + eps);
        }
    };
    f32x4* p4 = reinterpret_cast<f32x4*>(p); f32x4* m4 = reinterpret_cast<f32x4*>(m); f32x4* v4 = reinterpret_cast<f32x4*>(v);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        f32x4 P0 = p4[i], M0 = ld(m4 + i), V0 = ld(v4 + i), P1 = p4[i + stride], M1 = ld(m4 + i + stride), V1 = ld(v4 + i + stride);
        const f32x4 G0 = g4[i], G1 = g4[i + stride];
        upd(P0, M0, V0, G0); upd(P1, M1, V1, G1);
        p4[i] = P0; st(m4 + i, M0); st(v4 + i, V0); p4[i + stride] = P1; st(m4 + i + stride, M1); st(v4 + i + stride, V1);
    }
    if (i < n4) {
        f32x4 P0 = p4[i], M0 = ld(m4 + i), V0 = ld(v4 + i);
        upd(P0, M0, V0, g4[i]);
        p4[i] = P0; st(m4 + i, M0); st(v4 + i, V0);
    }
    if (blockIdx.x == 0) {
        for (long k = n4 * 4 + threadIdx.x; k < n; k += 256) {
            const float gg = g[k] * gs;
            float mm = m[k], vv = v[k];
            mm += (gg - mm) * omb1; vv += (gg * gg - vv) * omb2;
            p[k] -= (mm * alpha) / (sqrtf(vv) + eps);
            m[k] = mm; v[k] = vv;
        }
    }
}

__device__ __forceinline__ float signf_(float x) { return (float)((x > 0.f) - (x < 0.f)); }

// utils.py:6-12 + model.py:57-90, plus the gradients of every mean w.r.t. its input.
__global__ __launch_bounds__(256) void loss_kernel(const LossArgs a) {
    touch_kernarg(a);
    __shared__ float red[4];
    float s[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) s[k] = 0.f;
    const long n1 = (long)a.B * a.FT, n2 = (long)a.B * a.HW;
    const float inv1 = 1.f / (float)n1, inv2 = 1.f / (float)n2;
    const float lambda_cycle = a.hp ? a.hp[0] : a.lambda_cycle, lambda_identity = a.hp ? a.hp[1] : a.lambda_identity;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n1; i += stride) {
        const float A = a.A[i], Bm = a.Bm[i];
        const float cycB = a.cyc[i], cycA = a.cyc[a.cyc_gs + i];
        const float idB = a.out1[n1 + i], idA = a.out1[a.out1_gs + n1 + i];
        s[0] += fabsf(A - cycA); s[1] += fabsf(Bm - cycB); s[2] += fabsf(A - idA); s[3] += fabsf(Bm - idB);
        a.d_cyc[i] = lambda_cycle * signf_(cycB - Bm) * inv1;
        a.d_cyc[a.d_cyc_gs + i] = lambda_cycle * signf_(cycA - A) * inv1;
        a.d_out1[n1 + i] = lambda_identity * signf_(idB - Bm) * inv1;
        a.d_out1[a.d_out1_gs + n1 + i] = lambda_identity * signf_(idA - A) * inv1;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
        const float dAf = a.dout[i], dAr = a.dout[n2 + i];
        const float dBf = a.dout[a.dout_gs + i], dBr = a.dout[a.dout_gs + n2 + i];
        s[4] += (1.f - dBf) * (1.f - dBf); s[5] += (1.f - dAf) * (1.f - dAf);
        s[6] += (1.f - dAr) * (1.f - dAr); s[7] += dAf * dAf;
        s[8] += (1.f - dBr) * (1.f - dBr); s[9] += dBf * dBf;
        a.d_dout_G[i] = -2.f * (1.f - dAf) * inv2;
        a.d_dout_G[a.d_dout_G_gs + i] = -2.f * (1.f - dBf) * inv2;
        a.d_dout_D[i] = dAf * inv2;                 a.d_dout_D[n2 + i] = -(1.f - dAr) * inv2;
        a.d_dout_D[a.d_dout_D_gs + i] = dBf * inv2; a.d_dout_D[a.d_dout_D_gs + n2 + i] = -(1.f - dBr) * inv2;
    }
    // per-block partial sums, combined in block order by loss_finalize_kernel: the 8 scalars are bitwise reproducible
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float t = block_sum_256(s[k], red);
        if (threadIdx.x == 0) a.acc[blockIdx.x * 10 + k] = t;
    }
}
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* part, int nblk, float* L, float n1, float n2, float lc, float li, const float* hp) {
    __shared__ float red[4];
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = block_sum_256((int)threadIdx.x < nblk ? part[threadIdx.x * 10 + k] : 0.f, red);
    if (threadIdx.x == 0) {
        if (hp) { lc = hp[0]; li = hp[1]; }
        const float cyc = acc[0] / n1 + acc[1] / n1;
        const float idl = acc[2] / n1 + acc[3] / n1;
        const float gA2B = acc[4] / n2, gB2A = acc[5] / n2;
        const float dA = (acc[6] / n2 + acc[7] / n2) * 0.5f, dB = (acc[8] / n2 + acc[9] / n2) * 0.5f;
        L[0] = cyc; L[1] = idl; L[2] = gA2B; L[3] = gB2A;
        L[4] = gA2B + gB2A + lc * cyc + li * idl;
        L[5] = dA; L[6] = dB; L[7] = dA + dB;
    }
}

// sample_train_data's crop (preprocess.py:226-233): out[n][f][t] = src[off[n] + f*len[n] + start[n] + t]
__global__ __launch_bounds__(256) void crop_batch_kernel(const float* src, const long* off, const int* len, const int* start,
                                                         float* out, int F, int n_frames) {
    const int n = blockIdx.y;
    const float* s = src + off[n] + start[n];
    const int L = len[n], tot = F * n_frames;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tot; i += gridDim.x * 256) {
        const int f = i / n_frames, t = i - f * n_frames;
        out[(long)n * tot + i] = s[(long)f * L + t];
    }
}
int launch_crop_batch(const float* src, const long* off, const int* len, const int* start, float* out,
                      int N, int F, int n_frames, hipStream_t s) {
    int bx = (F * n_frames + 255) / 256; if (bx > 64) bx = 64;
    crop_batch_kernel<<<dim3(bx, N), 256, 0, s>>>(src, off, len, start, out, F, n_frames);
    VC_HIP(hipGetLastError());
    return 0;
}

// utils.l1_loss / utils.l2_loss (utils.py:6-12) as stand-alone ops for the plug-in path:
// loss = mean(|y - y_hat|) or mean((y - y_hat)^2); d_yhat = d loss / d y_hat.
__global__ __launch_bounds__(256) void mean_loss_kernel(const float* y, const float* yh, long n, int squared,
                                                        float* loss, float* d_yhat) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv = 1.f / (float)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = y[i] - yh[i];
        if (squared) { s += d * d; if (d_yhat) d_yhat[i] = -2.f * d * inv; }
        else { s += fabsf(d); if (d_yhat) d_yhat[i] = -signf_(d) * inv; }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(loss, s * inv);
}

// =====================================================================================
// launchers
// =====================================================================================
// ---- opt-in per-launch timing of the MFMA kernels (bench.py roofline leg): HIP events on the launch's own stream
namespace {
struct ProfRec { hipEvent_t a, b; double flops; int kind; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
}  // namespace
ProfScope::ProfScope(int kind_, double flops_, hipStream_t st) : on(g_prof_on), a(nullptr), b(nullptr), flops(flops_), kind(kind_), s(st) {
    if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
}
ProfScope::~ProfScope() {
    if (on) { (void)hipEventRecord(b, s); std::lock_guard<std::mutex> l(g_prof_mu); g_prof.push_back(ProfRec{a, b, flops, kind}); }
}

extern "C" int vc_prof_enable(int on) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = on != 0;
    return 0;
}
extern "C" int vc_prof_collect(int kind, double* ms, long long* launches, double* flops) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    double t = 0, f = 0; long long n = 0;
    for (auto& r : g_prof) {
        if (r.kind != kind) continue;
        VC_HIP(hipEventSynchronize(r.b));
        float e = 0; VC_HIP(hipEventElapsedTime(&e, r.a, r.b));
        t += e; f += r.flops; ++n;
    }
    if (ms) *ms = t; if (launches) *launches = n; if (flops) *flops = f;
    return 0;
}

// one launch zeroes a whole list of buffers (split-K destinations of a network pass)
struct ZeroList { float* p[48]; long n[48]; int count; };
__global__ __launch_bounds__(256) void zero_list_kernel(const ZeroList z) {
    float* p = z.p[blockIdx.y]; const long n = z.n[blockIdx.y];
    const long n4 = n >> 2;
    f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) reinterpret_cast<f32x4*>(p)[i] = zero;
    if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) p[i] = 0.f;
}
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] += src[i];
}
int launch_add(float* dst, const float* src, long n, hipStream_t s) {
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    add_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(dst, src, n);
    VC_HIP(hipGetLastError());
    return 0;
}
__global__ void set4_kernel(float* dst, float a, float b, float c, float d) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; }
int launch_set4(float* dst, float a, float b, float c, float d, hipStream_t s) {
    set4_kernel<<<1, 1, 0, s>>>(dst, a, b, c, d);
    VC_HIP(hipGetLastError());
    return 0;
}
// one launch copies a whole list of (dst, src, floats) chunks -- the step's input assembly (was 4 + 4 + 2 hipMemcpyAsync per step)
struct CopyList { float* d[8]; const float* s[8]; long n[8]; int count; };
__global__ __launch_bounds__(256) void copy_list_kernel(const CopyList z) {
    float* d = z.d[blockIdx.y]; const float* s = z.s[blockIdx.y]; const long n = z.n[blockIdx.y];
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(s)[i];
    if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) d[i] = s[i];
}
int launch_copy_list(const std::vector<CopyItem>& items, hipStream_t s) {
    for (size_t i0 = 0; i0 < items.size(); i0 += 8) {
        CopyList z{}; long mx = 0; bool al = true;
        z.count = (int)std::min<size_t>(8, items.size() - i0);
        for (int i = 0; i < z.count; ++i) {
            z.d[i] = items[i0 + i].dst; z.s[i] = items[i0 + i].src; z.n[i] = items[i0 + i].n; mx = std::max(mx, z.n[i]);
            al = al && ((uintptr_t)z.d[i] % 16 == 0) && ((uintptr_t)z.s[i] % 16 == 0);
        }
        if (!al) {      // unaligned buffers: plain copies
            for (int i = 0; i < z.count; ++i) VC_HIP(hipMemcpyAsync(z.d[i], z.s[i], (size_t)z.n[i] * sizeof(float), hipMemcpyDeviceToDevice, s));
            continue;
        }
        long bx = (mx / 4 + 255) / 256; if (bx > 128) bx = 128; if (bx < 1) bx = 1;
        copy_list_kernel<<<dim3((unsigned)bx, z.count), 256, 0, s>>>(z);
    }
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_zero_list(const std::vector<std::pair<float*, long>>& bufs, hipStream_t s) {
    for (size_t i0 = 0; i0 < bufs.size(); i0 += 48) {
        ZeroList z{}; long mx = 0;
        z.count = (int)std::min<size_t>(48, bufs.size() - i0);
        for (int i = 0; i < z.count; ++i) {
            z.p[i] = bufs[i0 + i].first; z.n[i] = bufs[i0 + i].second; mx = std::max(mx, z.n[i]);
            if ((uintptr_t)z.p[i] % 16) { vc_set_error("zero list: buffer not 16-byte aligned"); return -1; }
        }
        long bx = (mx / 4 + 255) / 256; if (bx > 256) bx = 256; if (bx < 1) bx = 1;
        zero_list_kernel<<<dim3((unsigned)bx, z.count), 256, 0, s>>>(z);
    }
    VC_HIP(hipGetLastError());
    return 0;
}

extern "C" int vc_prof_records(int kind, double* ms, double* flops, int max_n) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    int n = 0;
    for (auto& r : g_prof) {
        if (r.kind != kind) continue;
        if (n < max_n) {
            VC_HIP(hipEventSynchronize(r.b));
            float e = 0; VC_HIP(hipEventElapsedTime(&e, r.a, r.b));
            ms[n] = e; flops[n] = r.flops;
        }
        ++n;
    }
    return n;
}

template <int MODE, bool FWD>
static int norm_dispatch(const NormArgs& a, hipStream_t s) {
    const bool det = !FWD && a.det;                    // deterministic backward: one lane group per (group, channel), batch walked in order
    const int units = a.G * (det ? 1 : a.B) * a.C;
    dim3 block(256);
    {
        // rows of up to 512 positions, 16-byte aligned tensors: the f32x4 kernels
        auto al = [](const void* q, long gs) { return ((uintptr_t)q & 15) == 0 && (gs & 3) == 0; };
        bool ok = !det && (a.L & 3) == 0 && al(a.y, a.y_gs);
        if (FWD) ok = ok && al(a.out, a.out_gs) && (MODE != 2 || al(a.res, a.res_gs));
        else ok = ok && al(a.dy, a.dy_gs) && (a.dwin || al(a.dout, a.do_gs));
        if ((a.win || a.wout || a.dwin || a.dwout) && !(MODE != 1 && norm_takes_wino(a, FWD))) {
            vc_set_error("norm forward / backward: Winograd-domain operands were passed to a launch that does not apply the transforms (ask norm_takes_wino first)");
            return -1;
        }
        if (!FWD && !a.dy && !(a.outp || a.outq)) { vc_set_error("norm backward: dy may only be left out when an octet form of it is written"); return -1; }
        if (!FWD && a.outq && !(ok && a.L <= 512 && a.C % 32 == 0 && ((a.ps ? a.L >> 1 : a.L) & 7) == 0 && al(a.outq, a.outq_gs / 4))) {
            vc_set_error("norm backward: pixel octets were requested for a launch that does not produce them (ask norm_bwd_emits_pixel_octets first)");
            return -1;
        }
        if (a.outp && !(ok && a.L <= 512 && a.C % 32 == 0 && al(a.outp, a.outp_gs / 4))) {
            vc_set_error("norm forward / backward: a packed bf16 output was requested for a launch the 16-byte kernels do not serve (ask norm_fwd_emits_packed / norm_bwd_emits_packed first)");
            return -1;
        }
        if (ok && MODE == 1 && !FWD && !a.ps && a.L > 512) {          // long GLU rows (D.h1): streamed
            glu_bwd_v4_kernel<<<dim3((units + 7) / 8), block, 0, s>>>(a);
            VC_HIP(hipGetLastError());
            return 0;
        }
        ok = ok && a.L <= 512;
        if (ok) {
#define VC_NORM_V4P(LPR, NV, PS) { dim3 grid((units + (256 / LPR) - 1) / (256 / LPR)); \
            if (FWD) norm_fwd_v4_kernel<MODE, LPR, NV, PS><<<grid, block, 0, s>>>(a); else norm_bwd_v4_kernel<MODE, LPR, NV, PS><<<grid, block, 0, s>>>(a); }
#define VC_NORM_V4(LPR, NV) { if (a.ps) VC_NORM_V4P(LPR, NV, true) else VC_NORM_V4P(LPR, NV, false) }
            if (a.L <= 32) VC_NORM_V4(8, 1) else if (a.L <= 64) VC_NORM_V4(16, 1) else if (a.L <= 128) VC_NORM_V4(32, 1)
            else if (a.L <= 256) VC_NORM_V4(32, 2) else if (a.L <= 384) VC_NORM_V4(32, 3) else VC_NORM_V4(32, 4)
#undef VC_NORM_V4
#undef VC_NORM_V4P
            VC_HIP(hipGetLastError());
            return 0;
        }
    }
#define VC_NORM_CASE(LPR, NE) { dim3 grid((units + (256 / LPR) - 1) / (256 / LPR)); \
        if (FWD) norm_fwd_kernel<MODE, LPR, NE><<<grid, block, 0, s>>>(a); \
        else if (det) norm_bwd_kernel<MODE, LPR, NE, true><<<grid, block, 0, s>>>(a); \
        else     norm_bwd_kernel<MODE, LPR, NE, false><<<grid, block, 0, s>>>(a); }
    if (a.L <= 32) VC_NORM_CASE(32, 1)
    else if (a.L <= 64) VC_NORM_CASE(64, 1)
    else if (a.L <= 128) VC_NORM_CASE(64, 2)
    else if (a.L <= 256) VC_NORM_CASE(64, 4)
    else if (a.L <= 512) VC_NORM_CASE(64, 8)
    else if (a.L <= 1024) VC_NORM_CASE(64, 16)
    else if (a.L <= 2048) VC_NORM_CASE(64, 32)
    else {   // streaming: one block per row
        if (FWD) norm_fwd_stream_kernel<MODE><<<dim3(units), block, 0, s>>>(a);
        else if (det) norm_bwd_stream_kernel<MODE, true><<<dim3(units), block, 0, s>>>(a);
        else     norm_bwd_stream_kernel<MODE, false><<<dim3(units), block, 0, s>>>(a);
    }
#undef VC_NORM_CASE
    VC_HIP(hipGetLastError());
    return 0;
}
// would launch_norm_fwd / launch_norm_bwd apply the Winograd transforms of NormArgs (win / wout, dwin / dwout) inside this launch?  The 16-byte
// kernels with one quad per lane: rows of 8 .. 128 positions in multiples of 4, no pixel shuffle, gated-with-norm or residual-tail blocks, not the
// deterministic backward (it walks the batch inside a lane group)
bool norm_takes_wino(const NormArgs& a, bool fwd) {
    auto al = [](const void* q, long gs) { return ((uintptr_t)q & 15) == 0 && (gs & 3) == 0; };
    bool ok = (a.L & 3) == 0 && a.L >= 8 && a.L <= 128 && !a.ps && al(a.y, a.y_gs);
    if (fwd) ok = ok && al(a.out, a.out_gs) && (!a.res || al(a.res, a.res_gs));
    else ok = ok && !a.det && al(a.dy, a.dy_gs) && (a.dwin || al(a.dout, a.do_gs));
    return ok;
}
// would launch_norm_fwd honour NormArgs::outp for this launch?  (rows of up to 512 positions in multiples of 4, 16-byte aligned tensors,
// channel counts whose blocks of 8 / 16 / 32 rows hold whole octets of one sample)
bool norm_fwd_emits_packed(const NormArgs& a) {
    auto al = [](const void* q, long gs) { return ((uintptr_t)q & 15) == 0 && (gs & 3) == 0; };
    return (a.L & 3) == 0 && a.L <= 512 && a.C % 32 == 0 && al(a.y, a.y_gs) && al(a.out, a.out_gs) && (!a.res || al(a.res, a.res_gs));
}
bool norm_bwd_emits_packed(const NormArgs& a) {             // same question for the backward's dy (not in deterministic mode: its kernels walk the batch)
    auto al = [](const void* q, long gs) { return ((uintptr_t)q & 15) == 0 && (gs & 3) == 0; };
    return !a.det && (a.L & 3) == 0 && a.L <= 512 && a.C % 32 == 0 && al(a.y, a.y_gs) && al(a.dy, a.dy_gs) && al(a.dout, a.do_gs);
}
bool norm_bwd_emits_pixel_octets(const NormArgs& a) {      // ... and as pixel octets: the convolution's pixels per sample come in whole octets
    return norm_bwd_emits_packed(a) && ((a.ps ? a.L >> 1 : a.L) & 7) == 0 && !(a.L > 512);
}
int launch_norm_fwd(int mode, const NormArgs& a, hipStream_t s) {
    if (a.ps && (a.L & 1)) { vc_set_error("pixel shuffle needs even L"); return -1; }
    switch (mode) {
        case 0: return norm_dispatch<0, true>(a, s);
        case 1: return norm_dispatch<1, true>(a, s);
        case 2: return norm_dispatch<2, true>(a, s);
    }
    vc_set_error("bad norm mode %d", mode); return -1;
}
int launch_norm_bwd(int mode, const NormArgs& a, hipStream_t s) {
    if (a.ps && (a.L & 1)) { vc_set_error("pixel shuffle needs even L"); return -1; }
    switch (mode) {
        case 0: return norm_dispatch<0, false>(a, s);
        case 1: return norm_dispatch<1, false>(a, s);
        case 2: return norm_dispatch<2, false>(a, s);
    }
    vc_set_error("bad norm mode %d", mode); return -1;
}

int launch_bias_grad(const float* dy, long dy_gs, float* db, long db_gs, int G, int B, int C, int L, hipStream_t s) {
    bias_grad_kernel<<<dim3(C, G), 256, 0, s>>>(dy, dy_gs, db, db_gs, B, C, L);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_head_fwd(const float* x, long x_gs, const float* w, const float* b, long par_gs,
                    float* o, long o_gs, int G, int B, int C, int HW, hipStream_t s) {
    head_fwd_kernel<<<dim3((HW + 63) / 64, B, G), 64 * VC_HEAD_WAVES, 0, s>>>(x, x_gs, w, b, par_gs, o, o_gs, C, HW);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_head_bwd(const float* x, long x_gs, const float* w, long par_gs, const float* o, long o_gs,
                    const float* dout, long do_gs, float* dx, long dx_gs, float* dw, float* db,
                    long dpar_gs, int G, int B, int C, int HW, hipStream_t s) {
    if (dx) {
        const long rows = (long)B * C;
        const int blocks = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
        head_bwd_dx_kernel<<<dim3(blocks, G), 256, 0, s>>>(w, par_gs, o, o_gs, dout, do_gs, dx, dx_gs, B, C, HW);
    }
    if (dw) head_bwd_dw_kernel<<<dim3((C + 3) / 4, G), 256, 0, s>>>(x, x_gs, o, o_gs, dout, do_gs, dw, db, dpar_gs, B, C, HW);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_adam(float* p, float* m, float* v, const float* g, long n, float alpha, float b1,
                float b2, float eps, float gscale, hipStream_t s, const float* alpha_dev) {
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    adam_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(p, m, v, g, n4, n, alpha, b1, b2, eps, gscale, alpha_dev);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_mean_loss(const float* y, const float* yh, long n, int squared, float* loss, float* d_yhat, hipStream_t s) {
    VC_HIP(hipMemsetAsync(loss, 0, sizeof(float), s));
    long blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
    if (vc_det_active()) blocks = 1;              // one contribution to *loss: reproducible
    mean_loss_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(y, yh, n, squared, loss, d_yhat);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_loss(const LossArgs& a, hipStream_t s) {
    const long n1 = (long)a.B * a.FT;
    long blocks = (n1 + 255) / 256; if (blocks > VC_LOSS_BLOCKS) blocks = VC_LOSS_BLOCKS; if (blocks < 1) blocks = 1;
    loss_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}
// the 8 logged scalars from the per-block partial sums.  Nothing of the backward needs them: the step issues this launch on its side
// stream behind the discriminator_loss backward instead of between the loss kernel and the first backward kernels of both chains
int launch_loss_finalize(const LossArgs& a, hipStream_t s) {
    const long n1 = (long)a.B * a.FT;
    long blocks = (n1 + 255) / 256; if (blocks > VC_LOSS_BLOCKS) blocks = VC_LOSS_BLOCKS; if (blocks < 1) blocks = 1;
    loss_finalize_kernel<<<1, 256, 0, s>>>(a.acc, (int)blocks, a.losses, (float)n1, (float)((long)a.B * a.HW),
                                           a.lambda_cycle, a.lambda_identity, a.hp);
    VC_HIP(hipGetLastError());
    return 0;
}
