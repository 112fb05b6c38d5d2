// vc_conv_wgrad.hip -- filter gradient of tf.layers.conv1d/conv2d (autodiff of reference model.py:107-108 through
// module.py:32-40, 54-62) on v_mfma_f32_32x32x2_f32:
//
//   dW[kk][co] = sum over (segment, sample, pixel) of Xg[kk][P] * dY[co][P]
//
// rows = K rows of the forward table (64 per tile: two 32-row chunks, each inside ONE filter tap), cols = Cout (BN per
// tile), reduction = pixels in chunks of 32.  A *segment* is one application of the layer: the generator's weights are
// used twice per step (first hop on 2B samples, cycle hop on B samples, model.py:44-51, `reuse=True`), and both
// applications reduce into the same dW, so they are ONE launch over the concatenated pixel axis -- one write of dW,
// no atomics, no zero-filled destination.
//
// Loader without address arithmetic: a lane owns pixel (lane & 31) of the chunk and the row pairs 2*(wave + 4q) + (lane >> 5):
// the pair base rides in the buffer load's SGPR offset (loop-invariant), the lane's pixel offset (+ one channel for the
// upper lane half) in the VGPR offset, VC_OOB where the tap leaves the image for that pixel (hardware range check -> 0).
#include "vc_internal.h"
#include "vc_device_util.h"
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>

typedef const __attribute__((address_space(4))) int* c_int_p;

template <int BN, int KG, int MATH>
__global__ __launch_bounds__(256 * KG, KG == 4 ? 4 : 2) void wgrad_kernel(const WgradArgs p) {
    touch_kernarg(p);
    // LDS tiles are [row][pixel] with a 36-float row stride: the loader writes whole pixel runs (conflict-free), and a
    // lane reads its 16 k values (pixels 16h..16h+15 of its row; MFMA step u pairs pixels u and 16+u) as 4 ds_read_b128
    constexpr int BM = 64, KC = VC_KC, LDT = KC + 4;
    constexpr int WN = BN / 2, NJ = WN / 32;
    constexpr int ASZ = BM * LDT, BSZ = BN * LDT;
    constexpr int LDY = BN + 4, REG = 2 * (ASZ + BSZ), NT = 256 * KG;
    static_assert(BM * LDY <= REG, "epilogue staging fits the tile buffers");
    // KG K groups of 4 waves, each with its own pair of tile buffers and its own range of the pixel axis; their partial
    // tiles are summed through LDS in the epilogue (split-K inside the workgroup: no atomics)
    __shared__ __attribute__((aligned(16))) float smem[KG * REG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int kg = __builtin_amdgcn_readfirstlane(tid >> 8);
    float* As = smem + kg * REG;
    float* Bs = As + 2 * ASZ;
    const int wid = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int nt = blockIdx.x % p.ntiles, kt = blockIdx.x / p.ntiles;
    const int g = blockIdx.z / p.splitk, sp = blockIdx.z - g * p.splitk;
    // every split-K part runs `cps` chunks (an even count: the loop body is a pair of chunks with static register sets);
    // chunks past the last segment load zeros
    // ---- loop invariants of the K rows and columns this wave loads
    const KEntry e0 = p.ktab[kt * BM], e1 = p.ktab[kt * BM + 32];          // the two taps of this k-tile (wave-uniform)
    const int dh0 = e0.a_off >= 0 ? e0.dh : 0x4000, dw0 = e0.dw;           // padding chunk: dh = 16384 always fails the bounds check
    const int dh1 = e1.a_off >= 0 ? e1.dh : 0x4000, dw1 = e1.dw;
    // The reduction axis: the chunks of segment 0, then those of segment 1.  rowskip (pixels enumerated (row, sample,
    // column)): only the output rows i whose input row i*si + dh lies inside the image for one of the tile's two taps -- the
    // others would add zeros (a quarter of D.d3's filter-gradient work): per segment a contiguous chunk range [slo, slo + snv).
    int slo0 = 0, slo1 = 0, snv0 = p.seg[0].nchunks, snv1 = p.nseg > 1 ? p.seg[1].nchunks : 0;
    if (p.rowskip) {
        const int OHc = p.NPB / p.OWc;
        int ilo = OHc, ihi = -1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int dh = k ? dh1 : dh0;
            if (dh == 0x4000) continue;
            // 0 <= i*si + dh <= Hl-1
            const int a = dh >= 0 ? 0 : (-dh + p.si - 1) / p.si, b = (p.Hl - 1 - dh) >= 0 ? (p.Hl - 1 - dh) / p.si : -1;
            ilo = min(ilo, max(a, 0)); ihi = max(ihi, min(b, OHc - 1));
        }
        if (ihi < ilo) ilo = ihi = 0;                                      // no row reaches the image (table padding): reduce row 0, i.e. zeros
        const int plo0 = ilo * p.seg[0].RW, phi0 = min((ihi + 1) * p.seg[0].RW, p.seg[0].Ptot);
        slo0 = plo0 / KC; snv0 = (phi0 + KC - 1) / KC - slo0;
        if (p.nseg > 1) {
            const int plo1 = ilo * p.seg[1].RW, phi1 = min((ihi + 1) * p.seg[1].RW, p.seg[1].Ptot);
            slo1 = plo1 / KC; snv1 = (phi1 + KC - 1) / KC - slo1;
        }
    }
    const int nchunks = snv0 + snv1;
    const int parts = p.splitk * KG;
    const int cps = (((nchunks + parts - 1) / parts) + 1) & ~1;
    if (sp * KG * cps >= nchunks) return;                 // whole workgroup beyond the pixel axis (uniform)
    const int c0 = (sp * KG + kg) * cps;
    c_int_p xso = (c_int_p)p.xsoff;
    const int xbias = p.xbias;
    int xs[8];                                                             // SGPR offsets of the 8 row pairs (rows 2*(wid+4q), +1 in the upper lane half)
#pragma unroll
    for (int q = 0; q < 8; ++q) xs[q] = xso[kt * BM + 2 * (wid + 4 * q)];
    constexpr int NY = BN / 8;
    int ys[NY];                                                            // same for the dY rows (columns co of the tile); rows >= N clamp to the last pair
#pragma unroll
    for (int q = 0; q < NY; ++q) ys[q] = min(nt * BN + 2 * (wid + 4 * q), p.N - 2) * p.NPB * 4;
    const unsigned xhalf = (unsigned)h * (unsigned)p.xHW * 4u, yhalf = (unsigned)h * (unsigned)p.NPB * 4u;

    // ---- segment / pixel cursor of the LOADER (runs two chunks ahead of the multiply)
    int lc = c0;                                         // chunk (of the concatenated axis) the next load() fetches
    int seg = (p.nseg > 1 && lc >= snv0) ? 1 : 0;
    int segbase = seg ? snv0 : 0, segend = seg ? nchunks : snv0, segP = p.seg[seg].Ptot;
    int segRW = p.seg[seg].RW; float seg_inv_rw = p.seg[seg].inv_rw;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(p.seg[seg].X + (long)g * p.seg[seg].x_gs) - xbias), 0, (int)p.seg[seg].x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.seg[seg].dY + (long)g * p.seg[seg].dy_gs), 0, (int)p.seg[seg].dy_bytes, 0x00020000);
    int pP = ((seg ? slo1 : slo0) + lc - segbase) * KC + l31;      // pixel of this lane inside the segment
    int pr, pb = fast_divmod(pP, p.NPB, p.inv_npb, &pr);

    f32x16 acc[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nj][r] = 0.f;

    // Loader cut into items [pixel decode + X loads | dY loads | X stores | dY stores]; one slice runs in the shadow
    // of every MFMA group.  Two register sets: loads run two chunks ahead.
    constexpr int NLOAD = 8 + NY, NITEMS = 2 * NLOAD;
    float xr[2][8], dr[2][NY];
    unsigned vx0 = VC_OOB, vx1 = VC_OOB, vy = VC_OOB;
    auto decode = [&]() {
        if (lc == segend && seg + 1 < p.nseg) {          // uniform: the loader crosses into the next segment
            seg = 1; segbase = segend; segend = nchunks; segP = p.seg[1].Ptot; segRW = p.seg[1].RW; seg_inv_rw = p.seg[1].inv_rw;
            xrs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)(p.seg[1].X + (long)g * p.seg[1].x_gs) - xbias), 0, (int)p.seg[1].x_bytes, 0x00020000);
            yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.seg[1].dY + (long)g * p.seg[1].dy_gs), 0, (int)p.seg[1].dy_bytes, 0x00020000);
            pP = slo1 * KC + l31; pb = fast_divmod(pP, p.NPB, p.inv_npb, &pr);
        }
        const bool pv = pP < segP && lc < segend;        // chunks past the last segment (loader run-ahead) load zeros
        int i, j;
        if (p.rowskip) {                                  // (row, sample, column) enumeration; pr = pixel inside the sample
            int rem; i = fast_divmod(pP, segRW, seg_inv_rw, &rem); pb = fast_divmod(rem, p.OWc, p.inv_ow, &j); pr = i * p.OWc + j;
        } else i = fast_divmod(pr, p.OWc, p.inv_ow, &j);
        const int li0 = i * p.si, lj0 = j * p.sj;
        const unsigned xoffb = (unsigned)(((long)pb * p.x_bs + (long)li0 * p.xW + lj0) * 4) + xhalf;
        const bool ok0 = pv & ((unsigned)(li0 + dh0) < (unsigned)p.Hl) & ((unsigned)(lj0 + dw0) < (unsigned)p.Wl);
        const bool ok1 = pv & ((unsigned)(li0 + dh1) < (unsigned)p.Hl) & ((unsigned)(lj0 + dw1) < (unsigned)p.Wl);
        vx0 = ok0 ? xoffb : VC_OOB; vx1 = ok1 ? xoffb : VC_OOB;
        vy = pv ? (unsigned)(((long)pb * p.dy_bs + pr) * 4) + yhalf : VC_OOB;
        ++lc; pP += KC;
        if (p.rowskip) {}                                                                                    // decoded from pP above
        else if (p.NPB >= KC) { pr += KC; const bool w = pr >= p.NPB; pr = w ? pr - p.NPB : pr; pb += (int)w; }   // at most one wrap
        else pb = fast_divmod(pP, p.NPB, p.inv_npb, &pr);                                                  // tiny images (uniform branch)
    };
    auto item = [&](auto IC, auto SLC, auto SSC, int sbuf) {
        constexpr int I = decltype(IC)::value, SL = decltype(SLC)::value, SS = decltype(SSC)::value;
        if constexpr (I < 8) {
            if constexpr (I == 0) decode();
            xr[SL][I] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, I < 4 ? vx0 : vx1, xs[I], 0));
        } else if constexpr (I < NLOAD) {
            constexpr int q = I - 8;
            dr[SL][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, vy, ys[q], 0));
        } else if constexpr (I < NLOAD + 8) {
            constexpr int q = I - NLOAD;
            As[sbuf * ASZ + (2 * (wid + 4 * q) + h) * LDT + l31] = xr[SS][q];
        } else {
            constexpr int q = I - NLOAD - 8;
            Bs[sbuf * BSZ + (2 * (wid + 4 * q) + h) * LDT + l31] = dr[SS][q];
        }
    };
    auto items = [&](auto I0C, auto CNT, auto SLC, auto SSC, int sbuf) {
        constexpr int I0 = decltype(I0C)::value, N = decltype(CNT)::value;
        [&]<int... Ix>(std::integer_sequence<int, Ix...>) {
            ((I0 + Ix < NITEMS ? item(std::integral_constant<int, (I0 + Ix < NITEMS ? I0 + Ix : 0)>{}, SLC, SSC, sbuf)
                               : (void)0), ...);
        }(std::make_integer_sequence<int, N>{});
    };
    constexpr int NSLOT = KC / 2;
    constexpr int IPS = (NITEMS + NSLOT - 1) / NSLOT;
    auto chunk = [&](auto SLC, auto SSC, int buf) {
        const f32x4* as = reinterpret_cast<const f32x4*>(As + buf * ASZ + (wm * 32 + l31) * LDT + 16 * h);
        const f32x4* bs = reinterpret_cast<const f32x4*>(Bs + buf * BSZ + (wn * WN + l31) * LDT + 16 * h);
        f32x4 a4[4], b4[NJ][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a4[j] = as[j];
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) b4[nj][j] = bs[nj * 32 * LDT / 4 + j];
        }
        __builtin_amdgcn_sched_barrier(0);
        auto slot = [&](auto SC) {
            constexpr int u = decltype(SC)::value;      // MFMA step u multiplies pixels u (lanes 0-31) and 16+u (lanes 32-63)
            if constexpr (MATH == VC_MATH_FP32) {
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
                    acc[nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u / 4][u % 4], b4[nj][u / 4][u % 4], acc[nj], 0, 0, 0);
            } else if constexpr ((u & 7) == 7) {
                // bf16 modes: pixels 16h + 8*(u/8) .. +7 of this lane's row (same set for X and dY) -> one 32x32x16 MFMA
                constexpr int j0 = (u >> 3) * 2;
                bf16x8 ah, al, bh, bl;
                {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = a4[j0 + e / 4][e % 4];
                    to_frag<MATH>(v, ah, al);
                }
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = b4[nj][j0 + e / 4][e % 4];
                    to_frag<MATH>(v, bh, bl);
                    if constexpr (MATH == VC_MATH_BF16X3) {
                        acc[nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nj], 0, 0, 0);
                        acc[nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nj], 0, 0, 0);
                    }
                    acc[nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nj], 0, 0, 0);
                }
            }
            items(std::integral_constant<int, u * IPS>{}, std::integral_constant<int, IPS>{}, SLC, SSC, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        };
        [&]<int... Sx>(std::integer_sequence<int, Sx...>) { (slot(std::integral_constant<int, Sx>{}), ...); }
            (std::make_integer_sequence<int, NSLOT>{});
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using ZERO = std::integral_constant<int, 0>;
    using LD0 = std::integral_constant<int, NLOAD>;
    // the loader never branches on the range: it runs up to 3 chunks past the part -- inside the range of the NEXT split
    // that data is real but lands in LDS that is never multiplied; past the last segment it is zeros
    items(ZERO{}, LD0{}, S0{}, S0{}, 0);        // chunk c0   -> set 0
    items(ZERO{}, LD0{}, S1{}, S1{}, 0);        // chunk c0+1 -> set 1
    items(LD0{}, LD0{}, S0{}, S0{}, 0);         // set 0 -> LDS[0]
    __syncthreads();
    for (int c = 0; c < cps; c += 2) {
        chunk(S0{}, S1{}, 0);                   // multiply chunk c, load c+2 -> set 0, store set 1 (c+1) -> LDS[1]
        __syncthreads();
        chunk(S1{}, S0{}, 1);                   // multiply chunk c+1, load c+3 -> set 1, store set 0 (c+2) -> LDS[0]
        __syncthreads();
    }

    // ---- epilogue through LDS: every K group parks its tile; all threads sum the groups.  Plain 16-byte stores (4
    //      consecutive Cout values of one K row per lane) when this launch is the only contribution of the step to dW,
    //      fire-and-forget fp32 atomics (one value per lane, lanes along Cout) otherwise.
    {
        float* Ys = smem + kg * REG;
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ys[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDY + wn * WN + nj * 32 + l31] = acc[nj][r];
    }
    __syncthreads();
    float* Wg = p.dW + (long)g * p.dw_gs;
    if (p.store) {
        constexpr int NV = BM * BN / 4 / NT;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NT, ml = idx / (BN / 4), n4 = (idx % (BN / 4)) * 4;
            const int aoff = p.ktab[kt * BM + ml].a_off, co = nt * BN + n4;
            if (aoff < 0 || co >= p.N) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + ml * LDY + n4);
#pragma unroll
            for (int k2 = 1; k2 < KG; ++k2) v += *reinterpret_cast<const f32x4*>(smem + k2 * REG + ml * LDY + n4);
            *reinterpret_cast<f32x4*>(Wg + aoff + co) = v;          // Cout % 4 == 0: whole runs are in range and aligned
        }
    } else {
        constexpr int NE = BM * BN / NT;
#pragma unroll 4
        for (int i = 0; i < NE; ++i) {
            const int idx = tid + i * NT, ml = idx / BN, n = idx % BN;
            const int aoff = p.ktab[kt * BM + ml].a_off, co = nt * BN + n;
            if (aoff < 0 || co >= p.N) continue;
            float v = smem[ml * LDY + n];
#pragma unroll
            for (int k2 = 1; k2 < KG; ++k2) v += smem[k2 * REG + ml * LDY + n];
            atomicAdd(Wg + aoff + co, v);
        }
    }
}

// Tile width (Cout columns), K groups and split-K: the cost model of igemm_config (vc_conv_igemm.hip) with this kernel's
// chunk (32 pixels x (64 + BN) rows; 32 or 16 MFMAs per wave).  us per chunk of one wave among k on its SIMD.
struct WgradCfg { int bn, kg, sk; int src = 0; };   // src: 0 cost model, 1 exact measured row, 2 neighbour row, 3 forced
// measured choices (tools/tune_igemm.py --sweep on MI355X), keyed by the launch shape; everything else goes by the cost model
struct TunedWgrad { int ktiles, N, nchunks, groups, store, bn, kg, sk; };
#include "vc_tuned_wgrad.h"
static WgradCfg wgrad_config(const WgradArgs& a) {
    const bool det = vc_det_active();          // deterministic mode: the pixel axis is split inside the workgroup only
    if (vc_hooks().wgrad_bn) {
        const int bn = (vc_hooks().wgrad_bn == 128 && a.N < 128) ? 64 : vc_hooks().wgrad_bn;
        return WgradCfg{bn, vc_hooks().wgrad_kg, det ? 1 : std::min(vc_hooks().wgrad_sk, std::max(1, a.nchunks / (2 * vc_hooks().wgrad_kg))), 3};
    }
    for (const TunedWgrad& t : kTunedWgrad)
        if (t.ktiles == a.ktiles && t.N == a.N && t.nchunks == a.nchunks && t.groups == a.groups && t.store == a.store && !(det && t.sk > 1))
            return WgradCfg{t.bn, t.kg, t.sk, 1};
    {
        // no exact entry: the same layer (K tiles, Cout, groups) measured at another reduction length, nearest in log(chunks)
        // within a factor of 4, lends its configuration (see igemm_config)
        const TunedWgrad* nb = nullptr; double bd = 1e30;
        for (const TunedWgrad& t : kTunedWgrad)
            if (t.ktiles == a.ktiles && t.N == a.N && t.groups == a.groups && t.store == a.store && t.nchunks > 0 && !(det && t.sk > 1)) {
                const double d = std::fabs(std::log((double)t.nchunks / (double)a.nchunks));
                if (d < bd) { bd = d; nb = &t; }
            }
        if (nb && bd <= std::log(4.0) + 1e-9) {
            int sk = nb->sk;
            while (sk > 1) {
                const int parts = nb->kg * sk, cps = (((a.nchunks + parts - 1) / parts) + 1) & ~1;
                if (cps * (parts - 1) < a.nchunks) break;
                sk /= 2;
            }
            return WgradCfg{nb->bn, nb->kg, sk, 2};
        }
    }
    WgradCfg best{128, 1, 1, 0}; double bestc = 1e30;
    const double out_bytes = 4.0 * 64.0 * a.ktiles * (double)a.N * a.groups;
    for (int bn : {128, 64}) {
        if (bn == 128 && a.N < 128) continue;
        const long tiles = (long)a.ktiles * ((a.N + bn - 1) / bn) * a.groups;
        for (int kg : {1, 2, 4}) {
            if (bn == 128 && kg == 4) continue;                              // LDS: 55 KB per group
            for (int sk = 1; sk <= (det ? 1 : 16); sk *= 2) {
                const int parts = kg * sk;
                const int cps = (((a.nchunks + parts - 1) / parts) + 1) & ~1;
                if (parts > 1 && cps * (parts - 1) >= a.nchunks) continue;   // a part with no work
                const long wgs = tiles * sk;
                const int res = bn == 128 ? (kg == 1 ? 2 : 1) : (kg == 1 ? 4 : kg == 2 ? 2 : 1);
                const long n_cu = (wgs + 255) / 256;
                const long conc = std::min<long>(res, n_cu);
                const double rounds = (double)((n_cu + conc - 1) / conc);
                const double k = (double)conc * kg;                          // waves per SIMD
                const double t_chunk = bn == 128 ? 0.3 + 1.25 * k : 0.25 + 0.72 * k;
                const double c = 5.0 + rounds * (cps * t_chunk + 1.5) + ((sk > 1 || !a.store) ? 6.0 + out_bytes * sk / 0.4e6 : 0.0);
                if (c < bestc) { bestc = c; best = WgradCfg{bn, kg, sk, 0}; }
            }
        }
    }
    return best;
}
int wgrad_splitk(const WgradArgs& a) { return wgrad_config(a).sk; }
static thread_local int g_last_wgrad[4] = {0, 0, 0, 0};
extern "C" int vc_tune_last_wgrad_config(int* out4) { for (int i = 0; i < 4; ++i) out4[i] = g_last_wgrad[i]; return 0; }
// host-only: the configuration a filter-gradient launch of this shape would get (no launch; runs without a GPU)
extern "C" int vc_tune_query_wgrad(int ktiles, int N, int nchunks, int groups, int store, int* out4) {
    if (ktiles <= 0 || N <= 0 || nchunks <= 0 || groups <= 0 || !out4) { vc_set_error("vc_tune_query_wgrad: bad shape"); return -1; }
    WgradArgs a{}; a.ktiles = ktiles; a.N = N; a.nchunks = nchunks; a.groups = groups; a.store = store ? 1 : 0;
    const WgradCfg c = wgrad_config(a);
    out4[0] = c.bn; out4[1] = c.kg; out4[2] = c.sk; out4[3] = c.src;
    return 0;
}

int launch_wgrad(const WgradArgs& a0, int math, hipStream_t s, double flops) {
    WgradArgs a = a0;
    const WgradCfg c = wgrad_config(a);
    g_last_wgrad[0] = c.bn; g_last_wgrad[1] = c.kg; g_last_wgrad[2] = c.sk; g_last_wgrad[3] = c.src;
    a.ntiles = (a.N + c.bn - 1) / c.bn;
    a.splitk = c.sk;
    if (a.splitk > 1) a.store = 0;
    dim3 grid(a.ktiles * a.ntiles, 1, a.groups * a.splitk), block(256 * c.kg);
    ProfScope ps(1, flops, s);
#define VC_WG(BN, KG, MATH) wgrad_kernel<BN, KG, MATH><<<grid, block, 0, s>>>(a);
#define VC_WGRAD_MATH(MATH) \
    switch (c.bn * 8 + c.kg) { \
        case 128 * 8 + 1: VC_WG(128, 1, MATH) break; \
        case 128 * 8 + 2: VC_WG(128, 2, MATH) break; \
        case 64 * 8 + 1: VC_WG(64, 1, MATH) break; \
        case 64 * 8 + 2: VC_WG(64, 2, MATH) break; \
        default: VC_WG(64, 4, MATH) break; \
    }
    if (math == VC_MATH_BF16) VC_WGRAD_MATH(VC_MATH_BF16)
    else if (math == VC_MATH_BF16X3) VC_WGRAD_MATH(VC_MATH_BF16X3)
    else VC_WGRAD_MATH(VC_MATH_FP32)
#undef VC_WGRAD_MATH
#undef VC_WG
    VC_HIP(hipGetLastError());
    return 0;
}
