// vc_conv_bf16.hip -- the bf16 arithmetic of BASELINE configs[2] (vc_plan_set_math_mode(VC_MATH_BF16)) as a 16-bit data
// path: convolution forward (reference module.py:32-40, 54-62), input gradient and filter gradient (autodiff of
// model.py:107-108) on v_mfma_f32_32x32x16_bf16 with BOTH operands stored as bf16, 8 K values to a 16-byte unit, fp32
// accumulate and fp32 output.  Same arithmetic as the operand-rounding mode of vc_conv_igemm.hip (every operand rounded to
// bf16 once, RNE; fp32 products and sums), so the oracle's set_operand_rounding('bf16') describes it exactly.
//
// Layouts (built by the pack kernels below, consumed by conv_bf16_kernel):
//   gathered operand  Xp[g][b][c/8][pixel][8]   one 16-byte unit = 8 consecutive channels of one pixel.  GEMM-K runs
//                     (channel octet, tap): a K octet is ONE unit, so 64 consecutive pixels of an octet are 1 KB of
//                     global memory AND the MFMA B fragments of 64 lanes.
//   filter image      Ap[g][k/8][m][8]          the same shape with the output channel in the pixel's place, K in the order
//                     of the launch's table (padding octets 0): forward and input gradient read it the same way.
// Both operands therefore move global -> LDS with buffer_load_dwordx4 ... lds (lane-linear 1 KB copies: no staging
// registers, no ds_write pass), the LDS stage [octet][row][16 B] is read back with ds_read_b128 at consecutive 16-byte
// units per lane (conflict-free without padding), and nothing is transposed anywhere.  Lanes of padding taps are sent out
// of the buffer's range: the hardware returns, and copies, zeros.
//
// Workgroup: 256 threads = 2 x 2 waves, tile 128 or 64 (m) x 128 (pixels), K chunk 64, a ring of 2 or 4 LDS stages of
// 32 KB; a wave owns 64 x 64 = 2 x 2 MFMA tiles (16 MFMAs of 32 cycles per chunk against 16 ds_read_b128 and 8 LDS-direct
// copies).  The filter gradient is the same kernel on operands repacked 8 PIXELS to a unit (im2col_bf16_kernel,
// pack_pixel_octets_bf16_kernel; launch set-up in vc_ops.hip: conv_wgrad).
#include "vc_internal.h"
#include "vc_device_util.h"
#include <array>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef const __attribute__((address_space(4))) int* c_int_p;
typedef const __attribute__((address_space(4))) unsigned* c_uint_p;
typedef const __attribute__((address_space(4))) Tap* c_tap_p;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 1 KB per wave, global -> LDS without registers: lane l's 16 bytes land at lds + 16 l (buffer_load_dwordx4 ... lds).
// voff beyond the buffer's range copies zeros.  (The host pass of hipcc does not know the builtin: device pass only.)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void lds_copy16(__amdgpu_buffer_rsrc_t rs, lds_ptr_t lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, voff, soff, 0, 0);
#endif
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo), b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

// ------------------------------------------------------------------------------------------------------------------
// pack kernels
// ------------------------------------------------------------------------------------------------------------------
// planar fp32 [g][b][C][HW] -> Xp[g][b][C/8][HW][8] bf16.  Thread = (pixel, channel octet): 8 reads 4 B (lanes along the
// pixels: coalesced), one 16-byte store.
__global__ __launch_bounds__(256) void pack_act_bf16_kernel(const float* __restrict__ x, long x_gs, u32x4* __restrict__ xp, long xp_gs16,
                                                            int C8, int HW, int B) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= HW) return;
    const int cb = blockIdx.y, b = blockIdx.z % B, g = blockIdx.z / B;
    const float* src = x + (long)g * x_gs + ((long)b * C8 * 8 + (long)cb * 8) * HW + px;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[(long)e * HW];
    u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    xp[(long)g * xp_gs16 + ((long)b * C8 + cb) * HW + px] = o;
}

// filter image: Ap[g][k/8][m][k%8] = bf16(W[g][wsrc[k] + m * a_ms]) (0 where wsrc[k] < 0), 64 x 64 tiles through LDS.
//   a_ms == 1 (forward: TF kernels are [tap][cin][cout], m = cout): lanes along m when reading;
//   a_ms != 1 (input gradient: m = cin, consecutive k = consecutive cout): lanes along k when reading.
// Writes are 16 bytes per lane, lanes along m.
__global__ __launch_bounds__(256) void pack_filter_bf16_kernel(const float* __restrict__ w, long w_gs, unsigned short* __restrict__ ap, long ap_gs,
                                                               const int* __restrict__ wsrc, int a_ms, int M, int kp) {
    __shared__ unsigned short tile[64][64 + 2];
    const int k0 = blockIdx.x * 64, m0 = blockIdx.y * 64, g = blockIdx.z, tid = threadIdx.x;
    const float* W = w + (long)g * w_gs;
    if (a_ms == 1) {
        const int m = tid & 63;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int kk = (tid >> 6) + 4 * i, src = wsrc[k0 + kk];
            const float v = (src >= 0 && m0 + m < M) ? W[(long)src + m0 + m] : 0.f;
            tile[kk][m] = __builtin_bit_cast(unsigned short, (__bf16)v);
        }
    } else {
        const int kk = tid & 63, src = wsrc[k0 + kk];
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int m = (tid >> 6) + 4 * i;
            const float v = (src >= 0 && m0 + m < M) ? W[(long)src + (long)(m0 + m) * a_ms] : 0.f;
            tile[kk][m] = __builtin_bit_cast(unsigned short, (__bf16)v);
        }
    }
    __syncthreads();
    const int m = tid & 63;                      // lanes along m: 1 KB of one image octet per wave instruction
    if (m0 + m < M) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oct = (tid >> 6) + 4 * i;
            unsigned q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = (unsigned)tile[oct * 8 + 2 * e][m] | ((unsigned)tile[oct * 8 + 2 * e + 1][m] << 16);
            u32x4 o = {q[0], q[1], q[2], q[3]};
            *reinterpret_cast<u32x4*>(ap + (long)g * ap_gs + ((long)(k0 / 8 + oct) * M + m0 + m) * 8) = o;
        }
    }
}

// The same for MANY filter images in one launch (refresh_filter_images: every layer's forward image and its input-gradient
// images, once per train step -- 49 launches of a few microseconds each in round 3, 0.8 ms on the step's critical path, most
// of it launch latency).  blockIdx.x walks the concatenated (k block, m block) grids of the jobs.
__global__ __launch_bounds__(256) void pack_filter_multi_kernel(const PackFilterJobs jobs) {
    touch_kernarg(jobs);
    __shared__ unsigned short tile[64][64 + 2];
    int ji = 0;
    for (int q = 1; q < jobs.n; ++q) if ((int)blockIdx.x >= jobs.j[q].blk0) ji = q;      // uniform scan of <= 48 entries
    const PackFilterJob& J = jobs.j[ji];
    const int lb = blockIdx.x - J.blk0, kb = lb % J.nkb, mb = lb / J.nkb;
    const int k0 = kb * 64, m0 = mb * 64, g = blockIdx.z, tid = threadIdx.x, M = J.M, a_ms = J.a_ms;
    const float* W = J.w + (long)g * J.w_gs;
    const int* wsrc = J.wsrc;
    if (a_ms == 1) {
        const int m = tid & 63;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int kk = (tid >> 6) + 4 * i, src = wsrc[k0 + kk];
            const float v = (src >= 0 && m0 + m < M) ? W[(long)src + m0 + m] : 0.f;
            tile[kk][m] = __builtin_bit_cast(unsigned short, (__bf16)v);
        }
    } else {
        const int kk = tid & 63, src = wsrc[k0 + kk];
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int m = (tid >> 6) + 4 * i;
            const float v = (src >= 0 && m0 + m < M) ? W[(long)src + (long)(m0 + m) * a_ms] : 0.f;
            tile[kk][m] = __builtin_bit_cast(unsigned short, (__bf16)v);
        }
    }
    __syncthreads();
    const int m = tid & 63;
    if (m0 + m < M) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oct = (tid >> 6) + 4 * i;
            unsigned q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = (unsigned)tile[oct * 8 + 2 * e][m] | ((unsigned)tile[oct * 8 + 2 * e + 1][m] << 16);
            u32x4 o = {q[0], q[1], q[2], q[3]};
            *reinterpret_cast<u32x4*>(J.ap + (long)g * J.ap_gs + ((long)(k0 / 8 + oct) * M + m0 + m) * 8) = o;
        }
    }
}
int launch_pack_filter_multi(const std::vector<PackFilterJob>& jobs, int G, hipStream_t s) {
    for (size_t at = 0; at < jobs.size(); at += VC_PACK_JOBS) {
        PackFilterJobs pj{};
        int blk = 0;
        pj.n = (int)std::min<size_t>(VC_PACK_JOBS, jobs.size() - at);
        for (int i = 0; i < pj.n; ++i) {
            pj.j[i] = jobs[at + i];
            pj.j[i].nkb = pj.j[i].kp / 64; pj.j[i].blk0 = blk;
            blk += pj.j[i].nkb * ((pj.j[i].M + 63) / 64);
        }
        if (blk == 0) continue;
        pack_filter_multi_kernel<<<dim3(blk, 1, G), 256, 0, s>>>(pj);
        VC_HIP(hipGetLastError());
    }
    return 0;
}

// ---- filter gradient (autodiff of model.py:107-108 through module.py:32-40, 54-62) --------------------------------------
//   dW[k][co] = sum over (application, sample, pixel) of Xg[k][P] * dY[co][P],   k = (tap, cin) in TF's own order
// is a plain GEMM over the pixel axis, so it runs on conv_bf16_kernel itself: rows = K rows, "pixels" = Cout, K octets =
// pixel octets (vc_ops.hip: conv_wgrad).  What it needs is both operands with 8 consecutive PIXELS to a unit:
// im2col in bf16: Xcol[g][po][k][8] = the 8 consecutive output pixels 8 po .. 8 po + 7 of K row k = tap * Cin + c, read
// from the channel-octet packed activations.  A thread owns (channel octet, tap, pixel octet): it loads the 8 gathered
// units (8 pixels x 8 channels, zeros where the tap leaves the image), transposes them in registers (v_perm_b32) and
// stores 8 units = 128 contiguous bytes; lanes run along the channel octets, so a wave writes 8 KB in one piece.
struct Im2colArgs {
    const void* xp; void* xcol; long xp_gs, xcol_gs;      // bytes
    const Tap* taps; int ntaps;
    int C8, Hl, Wl, OW, NPB, si, sj, B, K;                  // K = ntaps * 8 * C8
    int po0;                                                // first pixel octet of this application inside xcol
    float inv_npb, inv_ow;
};
__global__ __launch_bounds__(256) void im2col_bf16_kernel(const Im2colArgs a) {
    touch_kernarg(a);
    const int cb = blockIdx.x * 64 + threadIdx.x;
    const int tap = blockIdx.y % a.ntaps, g = blockIdx.y / a.ntaps;
    const int po = blockIdx.z * 4 + threadIdx.y;
    if (cb >= a.C8 || po * 8 >= a.B * a.NPB) return;
    int r, j0; const int b = fast_divmod(po * 8, a.NPB, a.inv_npb, &r); const int i = fast_divmod(r, a.OW, a.inv_ow, &j0);
    const int ii = i * a.si + a.taps[tap].dh;
    const u32x4* src = reinterpret_cast<const u32x4*>((const char*)a.xp + (long)g * a.xp_gs) + ((long)b * a.C8 + cb) * a.Hl * a.Wl + (long)ii * a.Wl;
    u32x4 u[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int jj = (j0 + e) * a.sj + a.taps[tap].dw;
        const bool ok = (unsigned)ii < (unsigned)a.Hl && (unsigned)jj < (unsigned)a.Wl;
        const u32x4 z = {0u, 0u, 0u, 0u};
        u[e] = ok ? src[jj] : z;
    }
    u32x4* dst = reinterpret_cast<u32x4*>((char*)a.xcol + (long)g * a.xcol_gs) + (long)(a.po0 + po) * a.K + (long)tap * a.C8 * 8 + cb * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {                // channel c of the octet: halfword c of every unit -> one unit of 8 pixels
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o[q] = __builtin_amdgcn_perm(u[2 * q + 1][c >> 1], u[2 * q][c >> 1], (c & 1) ? 0x07060302u : 0x05040100u);
        dst[c] = o;
    }
}
// dYq[g][po][co][8] = bf16 of the 8 consecutive pixels 8 po .. + 7 of channel co (fp32 planar [g][b][Cout][NPB] in).
// Lanes along co: 1 KB stores; the 32-byte reads of neighbouring pixel octets share their cache lines.
__global__ __launch_bounds__(256) void pack_pixel_octets_bf16_kernel(const float* __restrict__ dy, long dy_gs, u32x4* __restrict__ out, long out_gs16,
                                                                      int Cout, int NPB, int npo, int po0, float inv_npb) {
    const int co = blockIdx.x * 64 + threadIdx.x, po = blockIdx.y * 4 + threadIdx.y, g = blockIdx.z;
    if (co >= Cout || po >= npo) return;
    int r; const int b = fast_divmod(po * 8, NPB, inv_npb, &r);
    const f32x4* src = reinterpret_cast<const f32x4*>(dy + (long)g * dy_gs + ((long)b * Cout + co) * NPB + r);
    const f32x4 v0 = src[0], v1 = src[1];
    u32x4 o = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
    out[(long)g * out_gs16 + (long)(po0 + po) * Cout + co] = o;
}
int launch_im2col_bf16(const void* xp, long xp_gs, void* xcol, long xcol_gs, const Tap* taps, int ntaps, int G, int B, int Cin,
                       int Hl, int Wl, int OH, int OW, int si, int sj, int po0, hipStream_t s) {
    Im2colArgs a{};
    a.xp = xp; a.xcol = xcol; a.xp_gs = xp_gs; a.xcol_gs = xcol_gs; a.taps = taps; a.ntaps = ntaps;
    a.C8 = Cin / 8; a.Hl = Hl; a.Wl = Wl; a.OW = OW; a.NPB = OH * OW; a.si = si; a.sj = sj; a.B = B; a.K = ntaps * Cin; a.po0 = po0;
    a.inv_npb = 1.0f / (float)a.NPB; a.inv_ow = 1.0f / (float)OW;
    const int npo = B * a.NPB / 8;
    if (ntaps * G > 65535 || (npo + 3) / 4 > 65535) { vc_set_error("im2col_bf16: grid too large"); return -1; }
    im2col_bf16_kernel<<<dim3((a.C8 + 63) / 64, ntaps * G, (npo + 3) / 4), dim3(64, 4), 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_pack_pixel_octets_bf16(const float* dy, long dy_gs, void* out, long out_gs_bytes, int G, int B, int Cout, int NPB, int po0, hipStream_t s) {
    const int npo = B * NPB / 8;
    if (NPB % 8 || dy_gs % 4 || ((uintptr_t)dy & 15) || (npo + 3) / 4 > 65535) { vc_set_error("pack_pixel_octets_bf16: NPB %% 8 != 0, misaligned tensor or grid too large"); return -1; }
    pack_pixel_octets_bf16_kernel<<<dim3((Cout + 63) / 64, (npo + 3) / 4, G), dim3(64, 4), 0, s>>>(dy, dy_gs, (u32x4*)out, out_gs_bytes / 16, Cout, NPB, npo, po0, 1.0f / (float)NPB);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_pack_act_bf16(const float* x, long x_gs, void* xp, long xp_gs_bytes, int G, int B, int C, int HW, hipStream_t s) {
    if (C % 8 || (long)B * G > 65535 || C / 8 > 65535) { vc_set_error("pack_act_bf16: C %% 8 != 0 or batch too large"); return -1; }
    dim3 grid((HW + 255) / 256, C / 8, B * G);
    pack_act_bf16_kernel<<<grid, 256, 0, s>>>(x, x_gs, (u32x4*)xp, xp_gs_bytes / 16, C / 8, HW, B);
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_pack_filter_bf16(const float* w, long w_gs, void* ap, long ap_gs_bytes, const int* wsrc, int a_ms, int G, int M, int kp, hipStream_t s) {
    dim3 grid(kp / 64, (M + 63) / 64, G);
    pack_filter_bf16_kernel<<<grid, 256, 0, s>>>(w, w_gs, (unsigned short*)ap, ap_gs_bytes / 2, wsrc, a_ms, M, kp);
    VC_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// the convolution
// ------------------------------------------------------------------------------------------------------------------
// BM = 256 (round 4): a wave owns 4 x 2 MFMA tiles (128 rows x 64 pixels, 128 accumulator registers): 6 fragment reads feed 8
// MFMAs (4 feed 4 at BM = 128), a chunk is 32 MFMAs per wave instead of 16 -- the per-chunk barrier, table fetch and LDS
// round trip weigh half as much -- on a 3-stage ring (48 KB per stage), one workgroup per CU.
// ROLE changes no code: 0 = convolution / input gradient, 1 = the filter-gradient GEMM -- two kernel names, so that rocprofv3
// statistics and PMC counters of a step separate the two uses
// SPEC (round 5): 512 threads = 4 multiplying waves + 4 copying waves, one of each per SIMD.  An LDS-direct copy holds the issuing
// wave for 60 - 180 cycles (the CU's address path moves 64 B / clock and four waves queue on it); with one wave per SIMD doing both,
// the matrix pipe idled behind every copy: measured 0.62 of the K loop busy at BM = 256 (12 copies against 32 MFMAs per chunk), 0.28 - 0.48
// on the smaller tiles.  The copying waves take all of them (the same pieces, the same order, the same counted wait before the
// chunk's barrier); the multiplying waves issue only ds_read_b128 and MFMA between two barriers.
template <int BM, int NST, int ROLE, bool SPEC = false> __global__ __launch_bounds__(SPEC ? 512 : 256, (!SPEC && NST <= 2) ? 2 : 1)
void conv_bf16_kernel(const ConvBf16Args p) {
    touch_kernarg(p);
    constexpr int NT = SPEC ? 512 : 256;
    constexpr int BN = 128, MI = BM / 64;                 // a wave owns MI x 2 MFMA tiles: rows [wm * BM/2, +BM/2), pixels [wn * 64, +64)
    constexpr int ASZ = 8 * BM * 16, XSZ = 8 * BN * 16, STAGE = ASZ + XSZ;      // one K chunk: [octet][row][16 B] for both operands
    constexpr int LDY = BN + 4;
    static_assert(BM * LDY * 4 <= 3 * STAGE, "the epilogue stages the tile inside the operand buffers");
    __shared__ __attribute__((aligned(16))) char smem[BM * LDY * 4 > NST * STAGE ? BM * LDY * 4 : NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wv = wv8 & 3;                               // multiplying wave wv and copying wave wv (SPEC: waves 4 - 7) share this index
    const bool copier = SPEC && wv8 >= 4;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;

    int mt, nt;
    {   // XCD-aware walk, as vc_conv_igemm.hip: every XCD gets one contiguous range of the m-major tile order
        const int total = p.mtiles * p.ntiles, per = total >> 3, tile = blockIdx.x;
        const int idx = (tile < (per << 3)) ? (tile & 7) * per + (tile >> 3) : tile;
        mt = idx / p.ntiles; nt = idx - mt * p.ntiles;
    }
    const int g = blockIdx.z / p.splitk, sp = blockIdx.z - g * p.splitk;
    const int nchunks = p.koct >> 3;
    const int cps = (nchunks + p.splitk - 1) / p.splitk;
    const int c0 = sp * cps, cn = min(cps, nchunks - c0);
    if (cn <= 0) return;

    // ---- global -> LDS copies (global_load_lds_dwordx4: lane l of the instruction lands at base + 16 l, no register
    //      staging, no ds_write): a wave moves octets 2 wv and 2 wv + 1 of every chunk, rows [0, 64) and [64, 128) of both
    //      operands = 8 copies of 1 KB per chunk.  The lane's two pixels / two filter rows:
    unsigned xoffb[2], tapmask[2] = {0u, 0u};
    {
        // one vector load brings the whole tap table (lane t holds tap t; at most 32 taps), the loop reads a tap back with
        // v_readlane -- a scalar load per tap and pixel half was a chain of up to 36 dependent scalar round trips (D.d3: 18 taps)
        int vdh = 0, vdw = 0;
        if (lane < p.ntaps) { const Tap tp = p.taps[lane]; vdh = tp.dh; vdw = tp.dw; }
        int li0[2], lj0[2]; bool pv[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int P = nt * BN + hf * 64 + lane;
            pv[hf] = P < p.Ptot;
            int xb = 0, xi = 0, xj = 0;
            if (pv[hf]) { int r; xb = fast_divmod(P, p.NPB, p.inv_npb, &r); xi = fast_divmod(r, p.OWc, p.inv_ow, &xj); }
            li0[hf] = xi * p.si; lj0[hf] = xj * p.sj;
            xoffb[hf] = (unsigned)((long)xb * p.x_bs + ((long)li0[hf] * p.xW + lj0[hf]) * 16);
        }
        const unsigned Hl = (unsigned)p.Hl, Wl = (unsigned)p.Wl;
        const int ntaps = p.ntaps;
        for (int t = 0; t < ntaps; ++t) {
            const int dh = __builtin_amdgcn_readlane(vdh, t), dw = __builtin_amdgcn_readlane(vdw, t);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const unsigned ok = (unsigned)pv[hf] & (unsigned)((unsigned)(li0[hf] + dh) < Hl) & (unsigned)((unsigned)(lj0[hf] + dw) < Wl);
                tapmask[hf] |= ok << t;
            }
        }
    }
    // buffer addressing: the lane's pixel (or VC_OOB for a padding tap: the hardware range check returns -- and copies --
    // zeros) in the VGPR offset, the K octet in the SGPR offset; no per-copy address arithmetic on the vector ALU
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.X + (long)g * p.x_gs - p.xbias), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + (long)g * p.a_gs), 0, (int)p.a_bytes, 0x00020000);
    unsigned avoff[MI];         // filter image [octet][M][16 B]; rows beyond M are clamped (multiplied, never stored)
#pragma unroll
    for (int hf = 0; hf < MI; ++hf) avoff[hf] = (unsigned)min(mt * BM + hf * 64 + lane, p.M - 1) * 16u;
    const unsigned a_os = (unsigned)p.M * 16u;          // bytes between consecutive octets of the image
    c_int_p xsoff = (c_int_p)p.xsoff;
    c_uint_p otap = (c_uint_p)p.otap;
    // table entries of the chunk the NEXT issue() copies (fetched one iteration ahead: a scalar load is a few hundred
    // cycles, and issue() sits between the barrier and the iteration's first MFMA)
    int tso[2]; unsigned ttb[2];
    auto fetch_tab = [&](int c) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { tso[q] = xsoff[c * 8 + 2 * wv + q]; ttb[q] = otap[c * 8 + 2 * wv + q]; }
    };
    // chunk c of the K range (table entries already in tso / ttb) -> LDS stage buf, as GL = 2 (2 + MI) copies in a fixed order;
    // issue_part<PART> sends out the PART-th quarter of them.  In the K loop a quarter rides behind the MFMAs of each of the
    // four steps: an LDS-direct copy costs ~60 cycles of issue, and with one workgroup per CU (one wave per SIMD) nothing
    // else covers GL of them in a row at the top of the iteration (round 3 issued them there: 0.5 - 0.7 us per chunk with
    // the matrix pipe idle, which is why two workgroups per CU measured faster whenever they fitted).
    constexpr int GL = 4 + 2 * MI;
    auto copy_one = [&](int c, int buf, int i) {          // i-th copy of the chunk: q = octet of the wave's pair, then X halves, then A quarters
        char* As = smem + buf * STAGE; char* Xs = As + ASZ;
        const int q = i / (2 + MI), k = i - q * (2 + MI);
        const int oq = 2 * wv + q, o = c * 8 + oq;
        if (k < 2) lds_copy16(xrs, (lds_ptr_t)(Xs + (oq * BN + k * 64) * 16), (tapmask[k < 2 ? k : 0] & ttb[q]) ? xoffb[k < 2 ? k : 0] : VC_OOB, (unsigned)tso[q]);
        else lds_copy16(ars, (lds_ptr_t)(As + (oq * BM + (k - 2) * 64) * 16), avoff[k >= 2 ? k - 2 : 0], (unsigned)o * a_os);
    };
    auto issue_part = [&](int c, int buf, auto PC) {
        constexpr int PART = decltype(PC)::value, I0 = GL * PART / 4, I1 = GL * (PART + 1) / 4;
#pragma unroll
        for (int i = I0; i < I1; ++i) copy_one(c, buf, i);
    };
    auto issue = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < GL; ++i) copy_one(c, buf, i);
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // NST LDS stages in a ring: while chunk c is multiplied, chunks c + 1 .. c + NST - 1 are streaming in.  A chunk costs
    // ~0.25 us of MFMA work and an L2 / HBM round trip is 0.5 - 2 us, so launches with one workgroup per CU run NST = 4
    // (three chunks in flight); launches that fill every CU twice over run NST = 2 with two workgroups per CU.
    // Top of an iteration: wait until THIS wave's copies of chunk c have landed (a counted vmcnt: the younger chunks stay
    // in flight across the barrier), then the barrier tells every wave that chunk c is complete and that chunk c - 1, the
    // previous tenant of the stage refilled next, has been multiplied.  Past the end of the K range the ring keeps
    // copying the last chunk into stages nobody reads, so the count is the same in every iteration.
    auto wait_landed = [&]() {
        if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr ((NST - 2) * GL == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if constexpr ((NST - 2) * GL == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr ((NST - 2) * GL == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr ((NST - 2) * GL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr ((NST - 2) * GL == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else static_assert(NST == 2, "add the counted wait for this ring depth");
    };
    // one K chunk of MFMAs out of stage st; REFILL: a quarter of the refill of stage stn rides behind each step (the 256-thread form)
    auto multiply_chunk = [&](int st, int stn, int cnx, int c, auto REFILL) {
        const char* As = smem + st * STAGE + ((wm * (BM / 2) + l31) + h * BM) * 16;
        const char* Xs = smem + st * STAGE + ASZ + ((wn * 64 + l31) + h * BN) * 16;
        // MFMA step s multiplies octets 2 s (lanes 0-31) and 2 s + 1 (lanes 32-63); the fragments of step s + 1 are
        // requested before the MFMAs of step s are issued, so an LDS round trip hides behind the MFMAs
        bf16x8 fa[2][MI], fb[2][2];
        auto frags = [&](int s, int set) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(As + s * 2 * BM * 16 + i * 32 * 16);
            fb[set][0] = *reinterpret_cast<const bf16x8*>(Xs + s * 2 * BN * 16); fb[set][1] = *reinterpret_cast<const bf16x8*>(Xs + s * 2 * BN * 16 + 32 * 16);
        };
        frags(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) frags(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);      // keep the reads of step s + 1 ahead of the MFMAs of step s
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][1], acc[i][1], 0, 0, 0);
            }
            if constexpr (decltype(REFILL)::value) {
                // a quarter of the refill of stage stn behind this step's MFMAs (the table entries were fetched an iteration ago)
                if (s == 0) issue_part(cnx, stn, std::integral_constant<int, 0>{});
                else if (s == 1) issue_part(cnx, stn, std::integral_constant<int, 1>{});
                else if (s == 2) issue_part(cnx, stn, std::integral_constant<int, 2>{});
                else { issue_part(cnx, stn, std::integral_constant<int, 3>{}); fetch_tab(c0 + min(c + NST, cn - 1)); }   // next used a barrier + a step of MFMAs from here
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (!SPEC || copier) {
#pragma unroll
        for (int k = 0; k < NST - 1; ++k) { fetch_tab(c0 + min(k, cn - 1)); issue(c0 + min(k, cn - 1), k); }
        fetch_tab(c0 + min(NST - 1, cn - 1));
    }
    if constexpr (!SPEC) {
        int st = 0;                                             // stage of chunk c
        for (int c = 0; c < cn; ++c) {
            wait_landed();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int stn = st == 0 ? NST - 1 : st - 1;         // (c + NST - 1) % NST: the stage chunk c - 1 has just left
            multiply_chunk(st, stn, c0 + min(c + NST - 1, cn - 1), c, std::true_type{});
            st = st + 1 == NST ? 0 : st + 1;
        }
    } else if (copier) {
        // every wave of the workgroup passes the same cn barriers: copying waves retire chunk c's pieces, meet the multiplying waves
        // (which have finished chunk c - 1), then refill the stage chunk c - 1 has left
        int st = 0;
        for (int c = 0; c < cn; ++c) {
            wait_landed();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int stn = st == 0 ? NST - 1 : st - 1;
            issue(c0 + min(c + NST - 1, cn - 1), stn);
            fetch_tab(c0 + min(c + NST, cn - 1));
            st = st + 1 == NST ? 0 : st + 1;
        }
    } else {
        __builtin_amdgcn_s_setprio(1);                          // the SIMD's other wave only issues copies: MFMAs first
        int st = 0;
        for (int c = 0; c < cn; ++c) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            multiply_chunk(st, 0, 0, c, std::false_type{});
            st = st + 1 == NST ? 0 : st + 1;
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();                                        // (drains the ring's trailing copies: vmcnt(0))

    // ---- epilogue: tile through LDS, then bias (+ addend) and 16-byte stores, or one pixel per lane for atomics /
    //      strided parity classes (the two paths of vc_conv_igemm.hip)
    float* Ys = reinterpret_cast<float*>(smem);
    if (!copier) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ys[(wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDY + wn * 64 + j * 32 + l31] = acc[i][j][r];
    }
    __syncthreads();
    const bool atomic = p.splitk > 1 || p.accumulate;
    const bool first = sp == 0;
    float* Yg = p.Y + (long)g * p.y_gs;
    const float* Bg = (p.bias && first) ? p.bias + (long)g * p.bias_gs : nullptr;
    const float* Dg = (p.addend && first) ? p.addend + (long)g * p.add_gs : nullptr;
    if (p.vec4 && !atomic) {
#pragma unroll 4
        for (int i = 0; i < BM * BN / 4 / NT; ++i) {
            const int idx = tid + i * NT, ml = idx >> 5, n4 = (idx & 31) * 4;
            const int m = mt * BM + ml, P0 = nt * BN + n4;
            if (m >= p.M || P0 >= p.Ptot) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(Ys + ml * LDY + n4);
            if (Bg) v += Bg[m];
            int r0, j; const int b = fast_divmod(P0, p.NPB, p.inv_npb, &r0); const int ii = fast_divmod(r0, p.OWc, p.inv_ow, &j);
            const long o = (long)b * p.y_bs + (long)m * p.yHW + (long)(ii * p.so_h + p.o_h0) * p.yW + j + p.o_w0;
            if (Dg) v += *reinterpret_cast<const f32x4*>(Dg + o);
            *reinterpret_cast<f32x4*>(Yg + o) = v;
        }
    } else {
        const int n = tid & 127, Pn = nt * BN + n;
        long obase = 0;
        if (Pn < p.Ptot) {
            int r0, j; const int b = fast_divmod(Pn, p.NPB, p.inv_npb, &r0); const int ii = fast_divmod(r0, p.OWc, p.inv_ow, &j);
            obase = (long)b * p.y_bs + (long)(ii * p.so_h + p.o_h0) * p.yW + (j * p.so_w + p.o_w0);
        }
#pragma unroll 4
        for (int i = 0; i < BM * BN / NT; ++i) {
            const int ml = (tid >> 7) + i * (NT / 128), m = mt * BM + ml;
            if (m >= p.M || Pn >= p.Ptot) continue;
            float v = Ys[ml * LDY + n];
            if (Bg) v += Bg[m];
            const long o = obase + (long)m * p.yHW;
            if (Dg) v += Dg[o];
            if (atomic) atomicAdd(Yg + o, v); else Yg[o] = v;
        }
    }
}

// Tile rows and LDS ring depth by how many tiles the launch has.  Split-K across workgroups costs fp32 atomics on the whole output
// plus its zero-fill (~1 TB/s of read-modify-write: 17 MB of output is ~20 us, as long as a 48-chunk K loop): only
// launches that would leave most of the chip idle split.
// measured choices (tools/tune_igemm.py --math bf16 --sweep on MI355X), keyed by the launch shape; everything else by the rule below
struct TunedBf16 { int M, Ptot, koct, groups, bm, nst, sk; };
#include "vc_tuned_bf16.h"
void conv_bf16_config(ConvBf16Args& a) {
    a.ntiles = (a.Ptot + 127) / 128;
    // 256-row tiles (one workgroup per CU, 128 x 64 per wave) when they fill the chip in whole rounds: at least 3/4 of the
    // last round's CUs busy.  VC_BF16_WIDE=0 switches the rule off (A/B measurements).
    static const bool wide_on = !(getenv("VC_BF16_WIDE") && atoi(getenv("VC_BF16_WIDE")) == 0);
    if (!vc_hooks().bf16_bm && wide_on && a.M >= 256 && a.M % 256 == 0 && a.koct >= 64) {
        const long t256 = (long)(a.M / 256) * a.ntiles * a.groups, rounds = (t256 + 255) / 256;
        if (t256 * 4 >= rounds * 256 * 3) { a.bm = 256; a.nst = 3; a.mtiles = a.M / 256; a.splitk = 1; return; }
    }
    if (!vc_hooks().bf16_bm)
        for (const TunedBf16& t : kTunedBf16)
            if (t.M == a.M && t.Ptot == a.Ptot && t.koct == a.koct && t.groups == a.groups) {
                a.bm = t.bm; a.nst = t.nst; a.mtiles = (a.M + a.bm - 1) / a.bm;
                a.splitk = vc_det_active() ? 1 : t.sk;
                return;
            }
    const long t128 = (long)((a.M + 127) / 128) * a.ntiles * a.groups;
    // measured (tools/bench_conv.py + vc_tune_conv_bf16_override per shape, then whole steps at batch 8 - 64 on one box):
    //   more than one 128-row tile per CU : 128 rows, two resident workgroups with two stages each (1007 TFLOP/s on D.d3)
    //   about one per CU                  : 128 rows, one workgroup per CU on the 4-stage ring
    //   fewer                             : 64-row tiles double the workgroups, 4-stage ring (-11 % step time at batch 8)
    if (t128 > 256 && a.M > 64) { a.bm = 128; a.nst = 2; }
    else if (t128 > 128 && a.M > 64) { a.bm = 128; a.nst = 4; }
    else { a.bm = 64; a.nst = 4; }
    if (vc_hooks().bf16_bm) { a.bm = (vc_hooks().bf16_bm == 256 && a.M <= 128) ? 128 : vc_hooks().bf16_bm; a.nst = a.bm == vc_hooks().bf16_bm ? vc_hooks().bf16_nst : 4; }
    a.mtiles = (a.M + a.bm - 1) / a.bm;
    const long tiles = (long)a.mtiles * a.ntiles * a.groups;
    const int nchunks = a.koct / 8;
    int sk = 1;
    while (tiles * sk < 128 && nchunks / (sk * 2) >= 8 && sk < 16) sk *= 2;
    if (vc_hooks().bf16_bm) sk = std::min(vc_hooks().bf16_sk, std::max(1, nchunks / 2));
    if (vc_det_active()) sk = 1;               // deterministic mode: no split-K atomics
    a.splitk = sk;
}

// tuning aid (tools/tune_igemm.py): the shape keys {M, Ptot, K octets, groups} of the calling thread's launches since the last read
static thread_local std::vector<std::array<int, 4>> t_bf16_keys;
extern "C" int vc_tune_conv_bf16_keys(int* out, int max_keys) {
    const int n = (int)t_bf16_keys.size();
    for (int i = 0; i < n && i < max_keys; ++i) for (int k = 0; k < 4; ++k) out[4 * i + k] = t_bf16_keys[i][k];
    t_bf16_keys.clear();
    return n;
}
int launch_conv_bf16(const ConvBf16Args& a, hipStream_t s, double flops, int prof_kind) {
    if (t_bf16_keys.size() < 64) t_bf16_keys.push_back({a.M, a.Ptot, a.koct, a.groups});
    if (a.splitk > 1 && !a.accumulate) { vc_set_error("launch_conv_bf16: split-K requires accumulate semantics"); return -1; }
    dim3 grid(a.mtiles * a.ntiles, 1, a.groups * a.splitk);
    ProfScope ps(prof_kind, flops, s);
#define VC_BF16_LAUNCH(BM, NST) do { if (prof_kind == 1) conv_bf16_kernel<BM, NST, 1><<<grid, 256, 0, s>>>(a); \
                                     else conv_bf16_kernel<BM, NST, 0><<<grid, 256, 0, s>>>(a); } while (0)
    // One workgroup per CU (256-row tiles; the 4-stage ring of the smaller tiles) runs wave-specialised: 4 multiplying + 4 copying waves
    // (K loop 0.62 -> 0.80 busy at 256 rows).  Two workgroups per CU (NST = 2) already have a second wave per SIMD to cover a copy's issue
    // time, and the split would cost them the second resident workgroup: measured slower, not built.  VC_BF16_SPEC=0 (process) /
    // vc_tune_conv_bf16_spec(0) (calling thread): the 256-thread form everywhere (A/B measurements, tests).
#define VC_BF16_LAUNCH_SPEC(BM, NST) do { if (prof_kind == 1) conv_bf16_kernel<BM, NST, 1, true><<<grid, 512, 0, s>>>(a); \
                                          else conv_bf16_kernel<BM, NST, 0, true><<<grid, 512, 0, s>>>(a); } while (0)
    static const bool spec_default = !(getenv("VC_BF16_SPEC") && atoi(getenv("VC_BF16_SPEC")) == 0);
    const bool spec = vc_hooks().bf16_spec < 0 ? spec_default : vc_hooks().bf16_spec != 0;
    if (spec && a.bm == 256) VC_BF16_LAUNCH_SPEC(256, 3);
    else if (spec && a.bm == 128 && a.nst == 4) VC_BF16_LAUNCH_SPEC(128, 4);
    else if (spec && a.bm == 64 && a.nst == 4) VC_BF16_LAUNCH_SPEC(64, 4);
    else if (a.bm == 256) VC_BF16_LAUNCH(256, 3);
    else if (a.bm == 64) { if (a.nst == 4) VC_BF16_LAUNCH(64, 4); else VC_BF16_LAUNCH(64, 2); }
    else { if (a.nst == 4) VC_BF16_LAUNCH(128, 4); else VC_BF16_LAUNCH(128, 2); }
#undef VC_BF16_LAUNCH_SPEC
#undef VC_BF16_LAUNCH
    VC_HIP(hipGetLastError());
    return 0;
}
