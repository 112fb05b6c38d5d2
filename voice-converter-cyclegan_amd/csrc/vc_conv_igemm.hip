// vc_conv_igemm.hip -- table-driven implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
// Serves tf.layers.conv1d/conv2d forward (reference module.py:32-40, 54-62) and their input gradients (autodiff of
// model.py:107-108), 1-D and 2-D, TF-SAME asymmetric padding, every stride-parity class of a transposed convolution.
//
//   D[m][n] = sum_k A[k][m] * Xg[k][n],  m = output channel, n = launch pixel (time-contiguous), K = (tap, channel).
//
// Workgroup = KG K-groups x (BM/32 x 2) waves; a K group multiplies its own K range into a BM x 64 tile (every wave owns
// one 32x32 MFMA tile), the groups' partial tiles are summed through LDS in the epilogue (split-K inside the workgroup:
// no atomics, no zero-filled destination, 16-byte stores).  LDS tiles are k-contiguous ([row][32 + 4]): a lane fetches
// its 16 k values of a chunk with 4 ds_read_b128 per operand (k = 8q + 4h .. + 3; MFMA step 4q + e pairs k = 8q + e in
// lanes 0-31 with k = 8q + 4 + e in lanes 32-63, the same for A and B).
//
// What the K loop costs besides its 16 MFMAs decides this kernel (an f32-input MFMA runs on the vector ALU's own FMA
// lanes: nothing co-issues with it, every other instruction is pipe time).  So the loader carries no address arithmetic:
//   * every 32-row chunk lies inside ONE filter tap (tables pad taps to 32 rows) and a lane loads aligned runs of 4 rows,
//     so the bounds check is one AND per run: the VGPR offset of a run's loads is the lane's pixel offset, or VC_OOB
//     (hardware range check -> 0) when the tap leaves the image for that pixel or the run is table padding;
//   * the K row's offset rides in the buffer load's SGPR offset, fetched from scalar-cache tables one chunk ahead;
//   * loader instructions are issued one slice behind every MFMA (sched_barrier-pinned), global loads two chunks ahead.
//
// FUSE = true (forward, 64-row tiles): the instance norm + GLU that follow a gated convolution pair (module.py:85-98) run
// in the epilogue -- see the FUSE branch below; every other launch runs the FUSE = false instantiation, unchanged.
#include "vc_internal.h"
#include <cstddef>
#include "vc_device_util.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

typedef const __attribute__((address_space(4))) int* c_int_p;
typedef const __attribute__((address_space(4))) unsigned* c_uint_p;

// launch pixel -> (sample, row, column) of the launch grid.  (Written on values: a form with two code paths assigning through
// the references made the compiler park b and i in SCRATCH memory behind a computed offset -- 12 bytes of private memory per
// lane and a store / load round trip through the memory system in every workgroup's prologue and again in its epilogue.)
__device__ __forceinline__ void px_decode(const IgemmArgs& p, int P, int& b, int& i, int& j) {
    int r;
    b = fast_divmod(P, p.NPB, p.inv_npb, &r);
    i = fast_divmod(r, p.OWc, p.inv_ow, &j);
}

template <int BM, bool AT, int KG, int MATH, bool FUSE = false>
__global__ __launch_bounds__(64 * (BM / 32) * 2 * KG, (64 * (BM / 32) * 2 * KG >= 1024) ? 4 : 2) void igemm_kernel(const IgemmMulti mm) {
    static_assert(!FUSE || (BM == 64 && !AT), "the fused instance-norm epilogue exists for forward launches with 64-row tiles");
    const IgemmArgs& p = mm.c[blockIdx.y];                    // blockIdx.y: which fused launch (parity class)
    static_assert(offsetof(IgemmMulti, c) == 0, "touch_kernarg below addresses c[blockIdx.y] from the kernarg segment's base");
    touch_kernarg(p, blockIdx.y * (unsigned)sizeof(IgemmArgs));
    if (blockIdx.x >= (unsigned)(p.mtiles * p.ntiles) || blockIdx.z >= (unsigned)(p.groups * p.splitk)) return;
    constexpr int BN = 64, KC = VC_KC, LDK = KC + 4;
    constexpr int WMG = BM / 32, WNG = 2, GT = 64 * WMG * WNG, NT = GT * KG;   // GT: threads of one K group
    constexpr int ASZ = BM * LDK, XSZ = BN * LDK, REG = 2 * (ASZ + XSZ);       // floats of one K group's LDS region
    constexpr int LDY = BN + 4;
    static_assert(BM * LDY <= REG, "the epilogue stages a group's tile inside its own region");
    __shared__ __attribute__((aligned(16))) float smem[KG * REG];

    const int tid = threadIdx.x, lane = tid & 63;
    const int kg = __builtin_amdgcn_readfirstlane(tid / GT);
    const int gtid = tid % GT, gw = __builtin_amdgcn_readfirstlane(gtid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = gw / WNG, wn = gw % WNG;
    float* As = smem + kg * REG;
    float* Xs = As + 2 * ASZ;

    // XCD-aware tile order (speed only; any order is correct): block b runs on XCD b % 8 with a private L2.  Walk the
    // tiles m-major (all pixel tiles of one weight slab are consecutive) and give every XCD one CONTIGUOUS range of that
    // walk, so a weight slab is fetched into as few L2s as possible.
    int mt, nt;
    {
        const int total = p.mtiles * p.ntiles, per = total >> 3, tile = blockIdx.x;
        const int idx = (tile < (per << 3)) ? (tile & 7) * per + (tile >> 3) : tile;   // the last total % 8 tiles keep their place
        mt = idx / p.ntiles; nt = idx - mt * p.ntiles;
    }
    // K range: split-K parts across workgroups (blockIdx.z) times K groups inside the workgroup; every part runs `cps`
    // chunks (an even count: the loop body is a pair of chunks with static register sets and no exit in between), parts
    // that reach past the table multiply its all-zero padding chunk.
    const int g = blockIdx.z / p.splitk, sp = blockIdx.z - g * p.splitk;
    const int nchunks = p.krows / KC;
    const int parts = p.splitk * KG;
    const int cps = (((nchunks + parts - 1) / parts) + 1) & ~1;
    if (sp * KG * cps >= nchunks) return;                     // whole workgroup beyond the K range (uniform)
    const int c0 = (sp * KG + kg) * cps;

    // ---- gathered-operand loader: a lane owns pixel n = lane of the tile and XL consecutive K rows of every chunk
    constexpr int XL = (KC * BN) / GT;                        // 4 (8-wave group) or 8 (4-wave group)
    static_assert(XL % 4 == 0, "whole 16-byte k runs per thread");
    const int xrow0 = gw * XL;
    // The launch's fixed cost is a chain of dependent memory round trips (kernel arguments -> tap table -> K tables ->
    // first operands: the round-3 timeline probe put 4.0 us between a workgroup's entry and its first operand load).  The
    // tap table (one vector load: lane t holds tap t; at most 32 taps) and the first chunk's K-table entries depend on the
    // kernel arguments only, so both are requested here, side by side; the weight loads of chunk c0 -- which need the table
    // but not the taps -- go out before the tap loop, the gathered operand's loads behind it.
    int vdh = 0, vdw = 0;
    if (lane < p.ntaps) { const Tap tp = p.taps[lane]; vdh = tp.dh; vdw = tp.dw; }
    unsigned xoffb = 0, tapmask = 0;                          // the lane's pixel offset and tap-validity mask: filled in behind the first table fetch
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(p.X + (long)g * p.x_gs) - p.xbias), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)g * p.a_gs), 0, (int)p.a_bytes, 0x00020000);
    c_int_p xsoff = (c_int_p)p.xsoff;
    c_uint_p rtap = (c_uint_p)p.rtap;
    c_uint_p aoff = (c_uint_p)p.aoff;

    // ---- weight loader.  Rows m >= M are clamped to row M-1 (loaded, multiplied, never stored): no out-of-range reads.
    //  plain : thread = (m = gtid % BM, k quad gtid / BM + q * QPP): 4 loads of 4 B (rows 4*quad .. +3, lanes along m:
    //          coalesced), one 16-B LDS store.  The quad is wave-uniform, so the 4 row offsets are one scalar load.
    //  transp: thread = (k quad gtid % 8, m = gtid / 8 + q * MPP): one 16-B load along k (a chunk's 32 rows are contiguous
    //          in the weight tensor: same tap, consecutive channels), one 16-B LDS store.  A run that reaches past the last
    //          channel of a row (taps padded to 32 rows) reads the next row's weights (finite; their X operand is 0), past
    //          the end of the tensor 0 (hardware range check).
    constexpr int QPP = GT / BM, NQ = (KC / 4) / QPP;         // plain: quads per pass, passes
    constexpr int MPP = GT / 8, NMP = BM / MPP;               // transposed: m rows per pass, passes
    const int am = AT ? gtid / 8 : gtid % BM;
    const int aq = AT ? gtid % 8 : __builtin_amdgcn_readfirstlane(gtid / BM);
    unsigned avoff[AT ? NMP : 1];
    if constexpr (AT) {
#pragma unroll
        for (int q = 0; q < NMP; ++q) avoff[q] = (unsigned)(min(mt * BM + am + q * MPP, p.M - 1) * p.a_ms) * 4u + (unsigned)aq * 16u;
    } else {
        // FUSE: tile rows 0-31 are value channels 32 mt .., rows 32-63 the matching gate channels C + 32 mt ..
        const int ch = FUSE ? (am < 32 ? 0 : (p.M >> 1) - 32) + mt * 32 + am : mt * BM + am;
        avoff[0] = (unsigned)min(ch, p.M - 1) * 4u;
    }

    constexpr int NAL = AT ? NMP : NQ * 4, NAS = AT ? NMP : NQ;
    float xr[2][XL];                 // two register sets: global loads run two chunks ahead of the multiply
    float at[2][AT ? 1 : NQ * 4];
    f32x4 af[2][AT ? NMP : 1];
    int xs[XL];                      // scalar-cache table entries of the chunk the NEXT load() fetches
    unsigned as[NAL];
    unsigned tb[XL / 4];
    // table reads: one byte offset per chunk (scalar), constant displacements per entry -> s_load with immediate offsets
    typedef const __attribute__((address_space(4))) char* c_char_p;
    auto fetch_tab = [&](int c) {
        const unsigned cb = (unsigned)min(c, nchunks) * (unsigned)(KC * 4);   // chunks past the table: its all-zero padding chunk
#pragma unroll
        for (int q = 0; q < XL; ++q) xs[q] = *(c_int_p)((c_char_p)xsoff + (cb + (unsigned)(xrow0 + q) * 4u));
#pragma unroll
        for (int q = 0; q < XL / 4; ++q) tb[q] = *(c_uint_p)((c_char_p)rtap + ((cb >> 2) + (unsigned)(xrow0 + 4 * q)));
        if constexpr (AT) {
#pragma unroll
            for (int q = 0; q < NAL; ++q) as[q] = *(c_uint_p)((c_char_p)aoff + cb);
        } else {
#pragma unroll
            for (int q = 0; q < NAL; ++q) as[q] = *(c_uint_p)((c_char_p)aoff + (cb + (unsigned)(4 * (aq + (q >> 2) * QPP) + (q & 3)) * 4u));
        }
    };

    constexpr int NACC = 2;          // two accumulator banks: the dependency distance of the accumulate chain is two MFMAs
    f32x16 accb[NACC];
#pragma unroll
    for (int bk = 0; bk < NACC; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[bk][r] = 0.f;

    // loader items: [X loads | A loads | X stores | A stores]
    constexpr int NLOAD = XL + NAL, NSTORE = XL / 4 + NAS, NITEMS = NLOAD + NSTORE;
    unsigned vx = VC_OOB;
    auto item = [&](auto IC, auto SLC, auto SSC, int sbuf) {
        constexpr int I = decltype(IC)::value, SL = decltype(SLC)::value, SS = decltype(SSC)::value;
        if constexpr (I < XL) {
            if constexpr (I % 4 == 0) vx = (tapmask & tb[I / 4]) ? xoffb : VC_OOB;
            xr[SL][I] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vx, xs[I], 0));
        } else if constexpr (I < NLOAD) {
            constexpr int q = I - XL;
            if constexpr (AT)
                af[SL][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff[q], as[q], 0));
            else
                at[SL][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, avoff[0], as[q], 0));
        } else if constexpr (I < NLOAD + XL / 4) {
            constexpr int q = I - NLOAD;
            f32x4 v = {xr[SS][4 * q], xr[SS][4 * q + 1], xr[SS][4 * q + 2], xr[SS][4 * q + 3]};
            *reinterpret_cast<f32x4*>(Xs + sbuf * XSZ + lane * LDK + xrow0 + 4 * q) = v;
        } else {
            constexpr int q = I - NLOAD - XL / 4;
            if constexpr (AT) *reinterpret_cast<f32x4*>(As + sbuf * ASZ + (am + q * MPP) * LDK + aq * 4) = af[SS][q];
            else {
                f32x4 v = {at[SS][4 * q], at[SS][4 * q + 1], at[SS][4 * q + 2], at[SS][4 * q + 3]};
                *reinterpret_cast<f32x4*>(As + sbuf * ASZ + am * LDK + (aq + q * QPP) * 4) = v;
            }
        }
    };
    // items [I0, I0+N) -- expanded at compile time so every register index stays static
    auto items = [&](auto I0C, auto CNT, auto SLC, auto SSC, int sbuf) {
        constexpr int I0 = decltype(I0C)::value, N = decltype(CNT)::value;
        [&]<int... Ix>(std::integer_sequence<int, Ix...>) {
            ((I0 + Ix < NITEMS ? item(std::integral_constant<int, (I0 + Ix < NITEMS ? I0 + Ix : 0)>{}, SLC, SSC, sbuf)
                               : (void)0), ...);
        }(std::make_integer_sequence<int, N>{});
    };
    // One K chunk: 16 MFMA slots on LDS[buf]; behind every slot one slice of the loader: global loads of chunk `cl` into
    // register set SL, LDS stores of set SS into LDS[buf^1].
    constexpr int NSLOT = KC / 2;
    constexpr int IPS = (NITEMS + NSLOT - 1) / NSLOT;
    auto chunk = [&](auto SLC, auto SSC, int buf, int cl) {
        const f32x4* asp = reinterpret_cast<const f32x4*>(As + buf * ASZ + (wm * 32 + l31) * LDK + 4 * h);
        const f32x4* xsp = reinterpret_cast<const f32x4*>(Xs + buf * XSZ + (wn * 32 + l31) * LDK + 4 * h);
        f32x4 a4[4], b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a4[q] = asp[2 * q]; b4[q] = xsp[2 * q]; }
        __builtin_amdgcn_sched_barrier(0);
        auto slot = [&](auto SC) {
            constexpr int sidx = decltype(SC)::value;
            if constexpr (MATH == VC_MATH_FP32) {
                accb[sidx % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[sidx / 4][sidx % 4], b4[sidx / 4][sidx % 4], accb[sidx % NACC], 0, 0, 0);
            } else if constexpr ((sidx & 7) == 7) {
                // bf16 modes: the 8 k-steps just passed (the SAME k set for A and B in either lane half) become one 32x32x16 MFMA
                constexpr int q0 = (sidx >> 3) * 2, bank = (sidx >> 3) % NACC;
                bf16x8 ah, al, bh, bl;
                float va[8], vb[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { va[e] = a4[q0 + e / 4][e % 4]; vb[e] = b4[q0 + e / 4][e % 4]; }
                to_frag<MATH>(va, ah, al); to_frag<MATH>(vb, bh, bl);
                if constexpr (MATH == VC_MATH_BF16X3) {
                    accb[bank] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, accb[bank], 0, 0, 0);
                    accb[bank] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, accb[bank], 0, 0, 0);
                }
                accb[bank] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, accb[bank], 0, 0, 0);
            }
            items(std::integral_constant<int, sidx * IPS>{}, std::integral_constant<int, IPS>{}, SLC, SSC, buf ^ 1);
            if constexpr (sidx * IPS < NLOAD && (sidx + 1) * IPS >= NLOAD) {
                // Scalar loads return out of order and share lgkmcnt with the LDS: once one is in flight, the next wait for an
                // LDS fragment read can only be lgkmcnt(0), i.e. a full scalar-memory round trip in front of an MFMA.  The
                // chunk's 8 fragment reads were issued ~8 MFMAs ago: retire them first (free), then fetch the tables.
                __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
                fetch_tab(cl + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        [&]<int... Sx>(std::integer_sequence<int, Sx...>) { (slot(std::integral_constant<int, Sx>{}), ...); }
            (std::make_integer_sequence<int, NSLOT>{});
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using ZERO = std::integral_constant<int, 0>;
    using LD0 = std::integral_constant<int, NLOAD>;
    using ST0 = std::integral_constant<int, NSTORE>;

    // the loader never branches: it runs up to 3 chunks past the part's range (LDS it then fills is never read)
    fetch_tab(c0);
    using XLC = std::integral_constant<int, XL>;
    using ALC = std::integral_constant<int, NAL>;
    items(XLC{}, ALC{}, S0{}, S0{}, 0);                       // weight loads of chunk c0: tables only
    {
        const int P = nt * BN + lane;
        const bool pv = P < p.Ptot;
        int xb = 0, xi = 0, xj = 0;
        if (pv) px_decode(p, P, xb, xi, xj);
        const int li0 = xi * p.si, lj0 = xj * p.sj;
        xoffb = (unsigned)(((long)xb * p.x_bs + (long)li0 * p.xW + lj0) * 4);
        // the loop reads a tap back with v_readlane: no scalar-memory round trip per tap
        const unsigned Hl = (unsigned)p.Hl, Wl = (unsigned)p.Wl;
        const int ntaps = p.ntaps;
        for (int t = 0; t < ntaps; ++t) {                // branch-free body: no kernel-argument reload, no exec juggling per tap
            const int dh = __builtin_amdgcn_readlane(vdh, t), dw = __builtin_amdgcn_readlane(vdw, t);
            const unsigned ok = (unsigned)pv & (unsigned)((unsigned)(li0 + dh) < Hl) & (unsigned)((unsigned)(lj0 + dw) < Wl);
            tapmask |= ok << t;
        }
    }
    items(ZERO{}, XLC{}, S0{}, S0{}, 0);                      // gathered-operand loads of chunk c0
    fetch_tab(c0 + 1);
    items(ZERO{}, LD0{}, S1{}, S1{}, 0);
    fetch_tab(c0 + 2);
    items(LD0{}, ST0{}, S0{}, S0{}, 0);
    __syncthreads();
    for (int c = 0; c < cps; c += 2) {
        chunk(S0{}, S1{}, 0, c0 + c + 2);   // multiply LDS[0] = chunk c; load c+2 -> set 0; store set 1 (c+1) -> LDS[1]
        __syncthreads();
        chunk(S1{}, S0{}, 1, c0 + c + 3);
        __syncthreads();
    }

    // ---- epilogue: every K group parks its tile in its own LDS region; all threads then sum the groups, add
    //      bias (+ addend) and store 4 consecutive pixels per lane (16 bytes when they are contiguous in the output).
    // Its kernel arguments first, ALL of them side by side and pinned in SGPRs: left to itself the compiler re-reads each field
    // from the kernarg segment where it is first used, behind its own s_waitcnt and usually a branch -- the round-4 ISA reading
    // found a dozen such scalar round trips in a row between the last MFMA and the first store (the round-3 probe: 4.7 us).
    // Fetched here they travel while the accumulators are parked, and so do the bias / addend values of the vec4 path.
#define VC_PIN_S(x) asm volatile("" : "+s"(x))
    float* e_Y = p.Y; const float* e_bias = p.bias; const float* e_add = p.addend;
    long e_ygs = p.y_gs, e_bgs = p.bias_gs, e_ags = p.add_gs, e_ybs = p.y_bs;
    int e_M = p.M, e_Ptot = p.Ptot, e_yHW = p.yHW, e_yW = p.yW, e_soh = p.so_h, e_sow = p.so_w, e_oh0 = p.o_h0, e_ow0 = p.o_w0;
    int e_vec4 = p.vec4, e_sk = p.splitk, e_acc = p.accumulate, e_NPB = p.NPB, e_OWc = p.OWc;
    float e_inpb = p.inv_npb, e_iow = p.inv_ow;
    VC_PIN_S(e_Y); VC_PIN_S(e_bias); VC_PIN_S(e_add); VC_PIN_S(e_ygs); VC_PIN_S(e_bgs); VC_PIN_S(e_ags); VC_PIN_S(e_ybs);
    VC_PIN_S(e_M); VC_PIN_S(e_Ptot); VC_PIN_S(e_yHW); VC_PIN_S(e_yW); VC_PIN_S(e_soh); VC_PIN_S(e_sow); VC_PIN_S(e_oh0); VC_PIN_S(e_ow0);
    VC_PIN_S(e_vec4); VC_PIN_S(e_sk); VC_PIN_S(e_acc); VC_PIN_S(e_NPB); VC_PIN_S(e_OWc); VC_PIN_S(e_inpb); VC_PIN_S(e_iow);
    auto decode = [&](int P, int& b, int& i, int& j) { int r; b = fast_divmod(P, e_NPB, e_inpb, &r); i = fast_divmod(r, e_OWc, e_iow, &j); };
    const bool atomic = e_sk > 1 || e_acc;
    const bool first = sp == 0;
    float* Yg = e_Y + (long)g * e_ygs;
    const float* Bg = (e_bias && first) ? e_bias + (long)g * e_bgs : nullptr;
    const float* Dg = (e_add && first) ? e_add + (long)g * e_ags : nullptr;
    const bool path_vec4 = !FUSE && e_vec4 && !atomic;
    // vec4 path: output offsets, bias and addend of this thread's NV quads, requested BEFORE the tile is parked
    constexpr int NV4 = BM * BN / 4 / NT;
    long vo[NV4]; float vb[NV4]; f32x4 vd[NV4]; bool vok[NV4];
    if (path_vec4) {
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int idx = tid + i * NT, ml = idx >> 4, n4 = (idx & 15) * 4;
            const int m = mt * BM + ml, P0 = nt * BN + n4;
            vok[i] = m < e_M && P0 < e_Ptot;
            vo[i] = 0; vb[i] = 0.f; vd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (vok[i]) {
                int b, ii, j; decode(P0, b, ii, j);
                vo[i] = (long)b * e_ybs + (long)m * e_yHW + (long)(ii * e_soh + e_oh0) * e_yW + j + e_ow0;
                if (Bg) vb[i] = Bg[m];
                if (Dg) vd[i] = *reinterpret_cast<const f32x4*>(Dg + vo[i]);
            }
        }
    }
    {
        const f32x16 acc = accb[0] + accb[1];
        float* Ys = smem + kg * REG;          // the tile loop ended on a barrier: the region is free
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Ys[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDY + wn * 32 + l31] = acc[r];
    }
    __syncthreads();
    if constexpr (FUSE) {
        // ---- instance norm + GLU in the epilogue (module.py:85-98 as one kernel with the convolution pair).  1-D, vec4
        //      layout, no split-K (checked by the launcher).  Phase A: sum the K groups, add the bias, store the raw pair and
        //      leave the sums in LDS region 0; phase B: 16 lanes own one (value row, gate row) pair of the tile, 4 pixels
        //      each: moments over the sample's NPB pixels by DPP, then the normalised, gated output.
        const int C = e_M >> 1;
        const float* gam = p.gamma + (long)g * p.par_gs;
        const float* bet = p.beta + (long)g * p.par_gs;
        float* actg = p.act + (long)g * p.act_gs;
        float* stg = p.stats + (long)g * p.st_gs;
        VC_PIN_S(gam); VC_PIN_S(bet); VC_PIN_S(actg); VC_PIN_S(stg);
        constexpr int NV = BM * BN / 4 / NT;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NT, ml = idx >> 4, n4 = (idx & 15) * 4;
            const int ch = (ml < 32 ? 0 : C - 32) + mt * 32 + ml, P0 = nt * BN + n4;
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + ml * LDY + n4);
#pragma unroll
            for (int k2 = 1; k2 < KG; ++k2) v += *reinterpret_cast<const f32x4*>(smem + k2 * REG + ml * LDY + n4);
            if (Bg) v += Bg[ch];
            *reinterpret_cast<f32x4*>(smem + ml * LDY + n4) = v;      // this cell is read and written by this thread only
            if (P0 < e_Ptot) {
                int r0; const int b = fast_divmod(P0, e_NPB, e_inpb, &r0);
                *reinterpret_cast<f32x4*>(Yg + (long)b * e_ybs + (long)ch * e_yHW + r0) = v;
            }
        }
        __syncthreads();
        constexpr int PT = NT < 512 ? NT : 512;
        if (tid < PT) {
            const float invL = e_inpb;
            const bool whole = e_NPB == 64;               // 16 lanes = one sample, or two samples of 32 pixels
            auto rsum = [&](float x) {
                x += dpp_read<0xB1>(x); x += dpp_read<0x4E>(x); x += dpp_read<0x141>(x);      // sums of 8 lanes
                if (whole) x += dpp_read<0x140>(x);
                return x;
            };
#pragma unroll
            for (int pass = 0; pass < 512 / PT; ++pass) {
                const int pr = (tid >> 4) + pass * (PT / 16), n4 = (tid & 15) * 4;
                const int c = mt * 32 + pr, P0 = nt * BN + n4;
                const bool pv = P0 < e_Ptot;
                // the affine parameters are requested before the reductions that do not need them
                const float ga = gam[c], be = bet[c], ga2 = gam[C + c], be2 = bet[C + c];
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + pr * LDY + n4);
                const f32x4 gt = *reinterpret_cast<const f32x4*>(smem + (pr + 32) * LDY + n4);
                const float mu = rsum(pv ? v[0] + v[1] + v[2] + v[3] : 0.f) * invL;
                const float mu2 = rsum(pv ? gt[0] + gt[1] + gt[2] + gt[3] : 0.f) * invL;
                const f32x4 dv = v - mu, dg = gt - mu2;
                const float rs = rsqrtf(rsum(pv ? dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2] + dv[3] * dv[3] : 0.f) * invL + 1e-6f);
                const float rs2 = rsqrtf(rsum(pv ? dg[0] * dg[0] + dg[1] * dg[1] + dg[2] * dg[2] + dg[3] * dg[3] : 0.f) * invL + 1e-6f);
                if (!pv) continue;
                int r0; const int b = fast_divmod(P0, e_NPB, e_inpb, &r0);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (dv[e] * rs * ga + be) * sigmoidf_(dg[e] * rs2 * ga2 + be2);
                *reinterpret_cast<f32x4*>(actg + ((long)b * C + c) * e_NPB + r0) = o;
                if (r0 == 0) {                            // one lane per (sample, channel pair)
                    float* st = stg + ((long)b * 2 * C) * 2;
                    st[2 * c] = mu; st[2 * c + 1] = rs; st[2 * (C + c)] = mu2; st[2 * (C + c) + 1] = rs2;
                }
            }
        }
    } else
    if (path_vec4) {
        // plain stores, 4 consecutive pixels (16 contiguous bytes) per lane: a wave instruction writes 4 rows x 256 B
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int idx = tid + i * NT, ml = idx >> 4, n4 = (idx & 15) * 4;
            if (!vok[i]) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + ml * LDY + n4);
#pragma unroll
            for (int k2 = 1; k2 < KG; ++k2) v += *reinterpret_cast<const f32x4*>(smem + k2 * REG + ml * LDY + n4);
            v += vb[i]; v += vd[i];
            *reinterpret_cast<f32x4*>(Yg + vo[i]) = v;
        }
    } else {
        // one pixel per lane, lanes along the pixels of one output row: atomics (split-K across workgroups, accumulate)
        // and strided stores (stride-parity classes of an input gradient) touch whole 256-byte runs per instruction
        constexpr int NE = BM * BN / NT;
        const int n = tid & 63, Pn = nt * BN + n;
        long obase = 0;
        if (Pn < e_Ptot) {
            int b, ii, j; decode(Pn, b, ii, j);
            obase = (long)b * e_ybs + (long)(ii * e_soh + e_oh0) * e_yW + (j * e_sow + e_ow0);
        }
#pragma unroll 4
        for (int i = 0; i < NE; ++i) {
            const int ml = (tid >> 6) + i * (NT / 64), m = mt * BM + ml;
            if (m >= e_M || Pn >= e_Ptot) continue;
            float v = smem[ml * LDY + n];
#pragma unroll
            for (int k2 = 1; k2 < KG; ++k2) v += smem[k2 * REG + ml * LDY + n];
            if (Bg) v += Bg[m];
            const long o = obase + (long)m * e_yHW;
            if (Dg) v += Dg[o];
            if (atomic) atomicAdd(Yg + o, v); else Yg[o] = v;
        }
    }
#undef VC_PIN_S
}

// ------------------------------------------------------------------------------------------------------------------
// Tile / K-group / split-K choice.  A workgroup costs its K chunks plus a fixed prologue + epilogue (`ovh` chunks); a
// launch runs in ceil(workgroups / resident workgroups) rounds.  K groups inside the workgroup are free of atomics and
// of the zero-fill launch, so they are preferred over split-K across workgroups at equal cost.
// ------------------------------------------------------------------------------------------------------------------
static const int kCUs = 256;
static int resident_per_cu(int bm, int kg) {            // LDS: 55 KB (128 rows) / 37 KB (64 rows) per K group, 160 KB per CU; 32 waves per CU
    const int waves = (bm / 32) * 2 * kg;
    const int by_lds = 160 / ((bm == 128 ? 55 : 37) * kg), by_waves = 32 / waves;
    return std::max(1, std::min(std::min(by_lds, by_waves), bm == 128 ? 2 : 4));
}
// measured choices (tools/tune_igemm.py on MI355X), keyed by the launch shape; everything else goes by the cost model
struct TunedCfg { int at, M, Ptot, krows, groups, bm, kg, sk; };
#include "vc_tuned_igemm.h"

// where the calling thread's last igemm_config took its answer from (tests: a benched shape must hit its measured row)
enum { CFG_MODEL = 0, CFG_EXACT = 1, CFG_NEIGHBOUR = 2, CFG_FORCED = 3, CFG_BF16_RULE = 4 };
static thread_local int g_last_igemm[4] = {0, 0, 0, 0};
static void igemm_config_impl(IgemmArgs& a, int math, int& src);
void igemm_config(IgemmArgs& a, int math) {
    int src = CFG_MODEL;
    igemm_config_impl(a, math, src);
    g_last_igemm[0] = a.bm; g_last_igemm[1] = a.kg; g_last_igemm[2] = a.splitk; g_last_igemm[3] = src;
}
extern "C" int vc_tune_last_igemm_config(int* out4) { for (int i = 0; i < 4; ++i) out4[i] = g_last_igemm[i]; return 0; }
// host-only: the configuration a launch of this shape would get (no launch; runs without a GPU)
extern "C" int vc_tune_query_igemm(int a_trans, int M, int Ptot, int krows, int groups, int* out4) {
    if (M <= 0 || Ptot <= 0 || krows <= 0 || krows % VC_KC || groups <= 0 || !out4) { vc_set_error("vc_tune_query_igemm: bad shape"); return -1; }
    IgemmArgs a{}; a.M = M; a.Ptot = Ptot; a.krows = krows; a.groups = groups; a.a_ms = a_trans ? 2 : 1;
    int src = CFG_MODEL;
    igemm_config_impl(a, VC_MATH_FP32, src);
    out4[0] = a.bm; out4[1] = a.kg; out4[2] = a.splitk; out4[3] = src;
    return 0;
}
static void igemm_config_impl(IgemmArgs& a, int math, int& src) {
    const int nchunks = a.krows / VC_KC;
    const bool det = vc_det_active();          // deterministic mode: K is split inside the workgroup only (fixed summation order)
    a.mtiles = 0;
    if (math != VC_MATH_FP32 && !vc_hooks().igemm_bm) {
        // bf16 operand modes: the MFMAs are 16x shorter, the launch is bound by the loader -> the tile with the least loader
        // work per MFMA (128 rows), no K groups, split-K only to fill the chip (the round-1 rule, measured for this mode)
        a.bm = a.M >= 512 ? 128 : 64; a.kg = 1;
        const long tiles = (long)((a.M + a.bm - 1) / a.bm) * ((a.Ptot + 63) / 64) * a.groups;
        int sk = 1;
        while (tiles * sk < 512 && nchunks / (sk * 2) >= 4 && sk < 16) sk *= 2;
        a.splitk = det ? 1 : sk;
        a.mtiles = (a.M + a.bm - 1) / a.bm; a.ntiles = (a.Ptot + 63) / 64;
        src = CFG_BF16_RULE;
        return;
    }
    if (vc_hooks().igemm_bm) { src = CFG_FORCED; a.bm = vc_hooks().igemm_bm; a.kg = vc_hooks().igemm_kg; a.splitk = det ? 1 : std::min(vc_hooks().igemm_sk, std::max(1, nchunks / (2 * a.kg))); }
    else {
        const int at = a.a_ms != 1 ? 1 : 0;
        for (const TunedCfg& t : kTunedIgemm)
            if (t.at == at && t.M == a.M && t.Ptot == a.Ptot && t.krows == a.krows && t.groups == a.groups && !(det && t.sk > 1)) { a.bm = t.bm; a.kg = t.kg; a.splitk = t.sk; a.mtiles = 1; src = CFG_EXACT; break; }
    }
    if (!vc_hooks().igemm_bm && !a.mtiles) {
        // No exact entry: the same LAYER (same filter-side shape: transposed?, M, K rows, groups) measured at another pixel
        // count.  The nearest one in log(pixels) within a factor of 4 lends its tile rows and K groups; split-K halves /
        // doubles with the pixel count (more pixel tiles -> less need to split K).  Makes the measured tables serve every
        // batch size instead of the three benched ones (the cost model below remains for layers never measured).
        const int at = a.a_ms != 1 ? 1 : 0;
        const TunedCfg* nb = nullptr; double bd = 1e30;
        for (const TunedCfg& t : kTunedIgemm)
            if (t.at == at && t.M == a.M && t.krows == a.krows && t.groups == a.groups && !(det && t.sk > 1)) {
                const double d = std::fabs(std::log((double)t.Ptot / (double)a.Ptot));
                if (d < bd) { bd = d; nb = &t; }
            }
        if (nb && bd <= std::log(4.0) + 1e-9) {
            a.bm = nb->bm; a.kg = nb->kg;
            int sk = nb->sk;
            for (double r = (double)a.Ptot / nb->Ptot; r >= 1.99 && sk > 1; r *= 0.5) sk /= 2;
            if (!det) for (double r = (double)a.Ptot / nb->Ptot; r <= 0.501 && sk < 16; r *= 2.0) sk *= 2;
            while (sk > 1) {          // every part must have work
                const int parts = a.kg * sk, cps = (((nchunks + parts - 1) / parts) + 1) & ~1;
                if (cps * (parts - 1) < nchunks) break;
                sk /= 2;
            }
            a.splitk = sk; a.mtiles = 1; src = CFG_NEIGHBOUR;
        }
    }
    if (vc_hooks().igemm_bm || a.mtiles) { a.mtiles = (a.M + a.bm - 1) / a.bm; a.ntiles = (a.Ptot + 63) / 64; return; }
    // Cost model in microseconds, fitted to the sweep of tools/tune_igemm.py (profiles/r02_tune_sweep_b8.txt): a workgroup
    // runs `cps` chunks at a pace set by the k waves that share its SIMDs (0.2 + 0.62 k us per chunk: a lone wave
    // exposes its LDS and barrier latency, four saturate the MFMA pipe at ~65 %), plus a fixed prologue + epilogue;
    // workgroups beyond what is resident run in further rounds; split-K across workgroups pays for its fp32 atomics
    // (~0.4 TB/s of read-modify-write) and the zero-fill.  64-row tiles win everywhere at batch <= 16 (two K groups
    // share nothing but the epilogue, so they cost no more loader work than two workgroups would).
    double best = 1e30;
    const double out_bytes = 4.0 * a.M * (double)a.Ptot * a.groups;
    for (int bm : {64, 128}) {
        if (bm == 128 && a.M < 512) continue;
        const long tiles = (long)((a.M + bm - 1) / bm) * ((a.Ptot + 63) / 64) * a.groups;
        for (int kg : {1, 2, 4}) {
            if (bm == 128 && kg == 4) continue;
            for (int sk = 1; sk <= (det ? 1 : 16); sk *= 2) {
                const int parts = kg * sk;
                const int cps = (((nchunks + parts - 1) / parts) + 1) & ~1;
                if (parts > 1 && cps * (parts - 1) >= nchunks) continue;      // a part with no work
                const long wgs = tiles * sk;
                const double w_wg = (bm / 32) * 2 * kg / 4.0;                  // waves per SIMD of one workgroup
                const long n_cu = (wgs + kCUs - 1) / kCUs;
                const long conc = std::min<long>(resident_per_cu(bm, kg), n_cu);
                const double rounds = (double)((n_cu + conc - 1) / conc);
                const double t_chunk = (0.2 + 0.62 * (double)conc * w_wg) * (bm == 128 ? 1.05 : 1.0);
                const double c = 5.0 + rounds * (cps * t_chunk + 1.0) + (sk > 1 ? 6.0 + out_bytes * sk / 0.4e6 : 0.0);
                if (c < best) { best = c; a.bm = bm; a.kg = kg; a.splitk = sk; }
            }
        }
    }
    a.mtiles = (a.M + a.bm - 1) / a.bm; a.ntiles = (a.Ptot + 63) / 64;
}

int launch_igemm_multi(const IgemmArgs* cls, int n, bool a_trans, int math, hipStream_t s, double flops) {
    if (n < 1 || n > 4) { vc_set_error("launch_igemm_multi: 1..4 launches can be fused"); return -1; }
    IgemmMulti m{}; m.n = n;
    int tiles = 0, z = 0;
    for (int i = 0; i < n; ++i) {
        const IgemmArgs& a = cls[i];
        if (a.splitk > 1 && !a.accumulate) { vc_set_error("launch_igemm: split-K requires accumulate semantics"); return -1; }
        if (a.bm != cls[0].bm || a.kg != cls[0].kg || a.groups != cls[0].groups) { vc_set_error("launch_igemm_multi: mixed tile configurations"); return -1; }
        m.c[i] = a; tiles = std::max(tiles, a.mtiles * a.ntiles); z = std::max(z, a.groups * a.splitk);
    }
    const int bm = cls[0].bm, kg = cls[0].kg;
    const bool fuse = cls[0].fuse != 0;
    if (fuse) {
        const IgemmArgs& a = cls[0];
        if (n != 1 || a_trans || bm != 64 || a.splitk != 1 || a.accumulate || !a.vec4 || (a.NPB != 32 && a.NPB != 64) || (a.M & 63) ||
            a.so_w != 1 || a.o_w0 != 0 || a.yHW != a.NPB) {
            vc_set_error("launch_igemm: this launch cannot carry the fused instance-norm epilogue"); return -1;
        }
    }
    dim3 grid(tiles, n, z), block(64 * (bm / 32) * 2 * kg);
    ProfScope ps(0, flops, s);
#define VC_IG(BM, KG, MATH) { if (a_trans) igemm_kernel<BM, true, KG, MATH><<<grid, block, 0, s>>>(m); \
                              else if constexpr (BM == 64) { if (fuse) igemm_kernel<BM, false, KG, MATH, true><<<grid, block, 0, s>>>(m); \
                                                             else igemm_kernel<BM, false, KG, MATH><<<grid, block, 0, s>>>(m); } \
                              else igemm_kernel<BM, false, KG, MATH><<<grid, block, 0, s>>>(m); }
#define VC_IG_MATH(MATH) \
    switch (bm * 8 + kg) { \
        case 128 * 8 + 1: VC_IG(128, 1, MATH) break; \
        case 128 * 8 + 2: VC_IG(128, 2, MATH) break; \
        case 64 * 8 + 1: VC_IG(64, 1, MATH) break; \
        case 64 * 8 + 2: VC_IG(64, 2, MATH) break; \
        case 64 * 8 + 4: VC_IG(64, 4, MATH) break; \
        default: vc_set_error("launch_igemm: tile %d x K groups %d is not built", bm, kg); return -1; \
    }
    if (math == VC_MATH_BF16) VC_IG_MATH(VC_MATH_BF16)
    else if (math == VC_MATH_BF16X3) VC_IG_MATH(VC_MATH_BF16X3)
    else VC_IG_MATH(VC_MATH_FP32)
#undef VC_IG_MATH
#undef VC_IG
    VC_HIP(hipGetLastError());
    return 0;
}
int launch_igemm(const IgemmArgs& a, bool a_trans, int math, hipStream_t s, double flops) { return launch_igemm_multi(&a, 1, a_trans, math, s, flops); }
