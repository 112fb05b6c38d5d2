// vc_conv_glu.hip -- gated_conv1d_glu_kernel: the generator's h1 block (module.py:161-163) in ONE kernel,
//     out = (conv_value(x) + b_v) * sigmoid(conv_gate(x) + b_g),        x [B][24][T] fp32  ->  out [B][C][T] fp32
// the kernel BASELINE.json's north_star names ("gated-conv1d forward at B=256").  HBM sees the frames once and the
// activation once (12 KB in + 64 KB out per sample).
//
// Design (round 4).  D[pixel][channel] = sum_k X[pixel][k] * W[k][channel], k = tap * 24 + cin:
//   * the FILTERS are the MFMA's B operand and live in REGISTERS for the whole life of the workgroup: a lane owns channel
//     (lane & 31) of the slab's 32 value and 32 gate channels and keeps its 8-k (bf16) / 4-k (fp32) fragments of every K
//     step -- 184 / 360 VGPRs, hence one 4-wave workgroup per CU with the 512-register budget (launch bounds (256, 1)).
//     Round 3 kept them in LDS and re-read them for every tile: 3 ds_read_b128 per MFMA pair put the LDS pipe beside
//     the MFMA pipe on the critical path; now the K loop issues ONE ds_read_b128 (the frames) per MFMA pair.
//   * the FRAMES are the A operand: a 128-pixel tile (+ halo) sits in LDS pixel-major ([pixel][24 cin], no row padding:
//     run q of 8 (4) consecutive k of the K axis starts q * 16 bytes after the pixel's row start, because consecutive
//     taps are consecutive rows), converted to bf16 ONCE and written with one 16-byte ds_write per (pixel, channel octet)
//     -- round 3 wrote 2 bytes at a time.  48-byte rows make both the writes and the fragment reads conflict-free.
//   * pixels are the MFMA's M axis, so a lane ends up with 4 CONSECUTIVE pixels of one channel in 4 consecutive
//     accumulator registers: bias + GLU happen in registers and go out as 16-byte stores with no LDS staging at all.
//   * software pipeline over the workgroup's tiles (one barrier per tile): while tile i is multiplied, the frames of
//     tile i+2 are in flight (inline-asm buffer loads retired by a COUNTED s_waitcnt: gfx950 has one in-order counter for
//     loads and stores, and the compiler would wait vmcnt(0), i.e. for the previous tile's output stores to complete),
//     the frames of tile i+1 are converted and written to the other LDS slot, and -- bf16 -- the GLU + stores of tile
//     i-1 are interleaved with the MFMAs (second accumulator set).  Every trip issues the same number of loads and
//     stores (out-of-tensor lanes carry an out-of-range buffer offset), which is what makes the wait countable.
//   MATH fp32 : v_mfma_f32_32x32x2_f32, exact;  MATH bf16 : operands rounded RNE once, v_mfma_f32_32x32x16_bf16.
#include "vc_internal.h"
#include "vc_device_util.h"
#include <type_traits>
#include <utility>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define VC_OOB_ST 0x80000000u      // store offset beyond every buffer this kernel accepts (< 2 GiB); adding an immediate cannot wrap

template <int CIN, int KW, bool BF16, bool RAW>
__global__ __launch_bounds__(256, 1) void gated_conv1d_glu_kernel(const WsConvArgs a) {
    touch_kernarg(a);
    typedef typename std::conditional<BF16, __bf16, float>::type T;
    typedef typename std::conditional<BF16, bf16x8, f32x4>::type frag_t;
    constexpr int R = BF16 ? 8 : 4;                         // k values per lane and K step (one 16-byte fragment)
    constexpr int RPT = CIN / R;                            // runs per tap
    static_assert(CIN % R == 0 && (BF16 || RPT % 2 == 0), "runs of R channels inside one tap; fp32: both lane halves in the same tap");
    constexpr int NRUN = KW * RPT, NU = (NRUN + 1) / 2;     // K steps: lane half h multiplies run 2u + h
    constexpr int K = CIN * KW;
    constexpr int TP = 128, PADL = (KW - 1) / 2;
    constexpr int XR = 144;                                 // staged rows: TP + KW - 1 frames + the row the zero-weight pad run touches
    static_assert(TP + KW <= XR && (KW & 1), "tile + halo fits the staged rows; odd kernel width");
    // Row stride in elements.  bf16: 24 (48 B) -- run q of a pixel's K axis then starts 16 q bytes after its row.  fp32: 28
    // (112 B, conflict-free 16-byte reads across pixels); run q = (tap, quad o) starts at tap * 28 + 4 o floats.
    constexpr int CS = BF16 ? CIN : CIN + 4;
    constexpr int SLOT = (XR + 1) * CS;                     // + one dummy row for the threads that have no staging task
    constexpr int NTASK = XR * RPT, NTPT = (NTASK + 255) / 256, NXL = NTPT * R;   // staging tasks (pixel row, run), per thread, loads per thread
    static_assert(NXL == 16, "operand lists of the counted waits below");
    constexpr int NST = RAW ? 12 : 4;                       // 16-byte output stores per tile and lane
    constexpr bool OVERLAP = BF16;                          // GLU + stores of tile i-1 inside the MFMA loop of tile i
    __shared__ __attribute__((aligned(16))) T Xs[2 * SLOT];
    __shared__ __attribute__((aligned(16))) frag_t Wf[2 * NU * 64];
    // per-wave 32 x 32 fp32 transposition tiles for the output planes (row = channel, stride 36 floats): the accumulators
    // hold 4 consecutive pixels of ONE channel per lane; stored straight from there an instruction touched 32 channel rows
    // with 32 bytes each -- 512 partial-line transactions per tile and CU, measured 2.0 us per tile against 0.64 us of MFMA
    // time.  Through the tile an instruction writes 8 rows x 128 contiguous bytes.
    constexpr int LDS_S = 36, NSTT = (BF16 && RAW) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float Ss[4 * NSTT * 32 * LDS_S];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int g = blockIdx.z, slab = blockIdx.y;
    const int C = a.Cout >> 1;
    const int tps = (a.T + TP - 1) / TP, ntiles = a.B * tps;
    const int T4 = a.T * 4;
    const int c = slab * 32 + l31;                          // this lane's channel (column of the MFMA tile)
    const float bvl = a.bias ? a.bias[(long)g * a.b_gs + c] : 0.f;
    const float bgl = a.bias ? a.bias[(long)g * a.b_gs + C + c] : 0.f;

    // ---- frame staging, per-thread constants: task j of a thread is (row r, run o) of EVERY tile
    unsigned goff[NTPT], ldsb[NTPT]; int rpos[NTPT];
#pragma unroll
    for (int j = 0; j < NTPT; ++j) {
        const int i = tid + 256 * j, o = i / XR, r = i - o * XR;
        const bool in = i < NTASK;
        goff[j] = (unsigned)((o * R * a.T + r) * 4);
        rpos[j] = in ? r : 0x40000000;                      // no task: always fails the range check below
        ldsb[j] = (unsigned)((in ? r * CS + o * R : XR * CS) * (int)sizeof(T));
    }
    i32x4 xrs_;                                             // buffer descriptor of this group's frames (SGPRs; inline asm operand)
    {
        const unsigned long long pa = (unsigned long long)(a.x + (long)g * a.x_gs);
        xrs_[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pa);
        xrs_[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(pa >> 32) & 0xffff);
        xrs_[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)a.B * CIN * (unsigned)a.T * 4u));
        xrs_[3] = 0x00020000;
    }
    float xr0[NXL], xr1[NXL];                               // two register sets of frames in flight
    // frames of `tile` -> a register set: R channels x NTPT rows per thread, lanes along time (coalesced); the channel's offset
    // rides in the load's SGPR offset.  Positions outside the sample (SAME padding) and tiles past the end get an out-of-
    // range VGPR offset: the hardware returns 0.  Always NXL load instructions.
    // tile -> (sample, tile inside the sample) without a division per trip: a cursor that advances by the workgroup's tile
    // step (tstep = qs * tps + rs, decomposed once)
    struct Cursor { int tile, b, tt; };
    const int tstep_ = gridDim.x, qs = tstep_ / tps, rs = tstep_ - qs * tps;
    auto cursor_at = [&](int tile) { Cursor c; c.tile = tile; c.b = tile / tps; c.tt = tile - c.b * tps; return c; };
    auto advance = [&](Cursor& c) {                       // branch-free: one select per trip
        c.tile += tstep_; c.b += qs; c.tt += rs;
        const bool w = c.tt >= tps;
        c.tt = w ? c.tt - tps : c.tt; c.b += (int)w;
    };
    auto retreat = [&](Cursor c) {                        // the cursor one step BEFORE c (prologue only): advance(retreat(c)) == c
        c.tile -= tstep_; c.b -= qs; c.tt -= rs;
        if (c.tt < 0) { c.tt += tps; --c.b; }
        return c;
    };
    auto issue_loads = [&](float (&x)[NXL], const Cursor& cu) {
        const bool tv = cu.tile < ntiles;
        const int b = tv ? cu.b : 0, t0 = (tv ? cu.tt : 0) * TP - PADL;
        const int base = (b * CIN * a.T + t0) * 4, tlim = tv ? a.T : 0;
#pragma unroll
        for (int j = 0; j < NTPT; ++j) {
            const unsigned off = ((unsigned)(t0 + rpos[j]) < (unsigned)tlim) ? (unsigned)base + goff[j] : VC_OOB;
#pragma unroll
            for (int e = 0; e < R; ++e)
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(x[j * R + e]) : "v"(off), "s"(xrs_), "s"(e * T4) : "memory");
        }
    };
    // The counter retires in order: "at most N younger operations outstanding" means the register set has arrived.  The
    // compiler sees the registers pass through the asm, so no use can move above the wait.
#define VC_XR_OPS "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), \
                  "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
    auto wait_all = [&](float (&x)[NXL]) { asm volatile("s_waitcnt vmcnt(0)" : VC_XR_OPS :: "memory"); };
    // younger operations at the point of use: the next-but-one tile's NXL loads + two tiles' output stores
    static_assert(16 + 2 * NST == 24 || 16 + 2 * NST == 40, "the two counted waits below");
    auto wait_counted = [&](float (&x)[NXL]) {
        if constexpr (NST == 4) asm volatile("s_waitcnt vmcnt(24)" : VC_XR_OPS :: "memory");
        else asm volatile("s_waitcnt vmcnt(40)" : VC_XR_OPS :: "memory");
    };
#undef VC_XR_OPS
    // task j of a register set -> LDS slot: R values converted once, one 16-byte write
    auto stage_task = [&](const float (&x)[NXL], auto JC, T* slot) {
        constexpr int j = decltype(JC)::value;
        frag_t f;
#pragma unroll
        for (int e = 0; e < R; ++e) f[e] = (T)x[j * R + e];
        *reinterpret_cast<frag_t*>(reinterpret_cast<char*>(slot) + ldsb[j]) = f;
    };

    // ---- prologue: first tile's frames travel while the filter fragments are built
    const int tile0 = blockIdx.x, tstep = gridDim.x;
    Cursor lcur = cursor_at(tile0);                      // next tile whose frames are requested
    issue_loads(xr0, lcur); advance(lcur);
    {
        // Fragment unit U = 2u + vg (K step u; value / gate): lane (channel l31, half h) holds W[k = R (2u + h) + e][channel],
        // e < R.  Wave w builds units w, w + 4, ..: R coalesced loads each (32 channels = 128 bytes per k row), all issued
        // before the first conversion (ONE memory round trip), one conflict-free 16-byte LDS write per unit.
        constexpr int NB = (2 * NU + 3) / 4;
        const float* wg_ = a.w + (long)g * a.w_gs;
        float v[NB][R];
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int U = w4 + 4 * it, u = U >> 1, vg = U & 1;
            const int ch = (vg ? C : 0) + slab * 32 + l31;
#pragma unroll
            for (int e = 0; e < R; ++e) {
                const int k = R * (2 * u + h) + e;
                const bool ok = U < 2 * NU && k < K;
                v[it][e] = wg_[ok ? (long)k * a.Cout + ch : 0];
                v[it][e] = ok ? v[it][e] : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int U = w4 + 4 * it;
            if (U < 2 * NU) {
                frag_t f;
#pragma unroll
                for (int e = 0; e < R; ++e) f[e] = (T)v[it][e];
                Wf[U * 64 + lane] = f;
            }
        }
    }
    wait_all(xr0);
    [&]<int... J>(std::integer_sequence<int, J...>) { (stage_task(xr0, std::integral_constant<int, J>{}, Xs), ...); }
        (std::make_integer_sequence<int, NTPT>{});
    issue_loads(xr1, lcur); advance(lcur);

    // output addressing: an accumulator quad (4 consecutive pixels of the lane's channel) goes into the wave's transposition
    // tile; flushing a plane, lane l stores pixels 4 (l & 7) .. + 3 of rows (l >> 3) + 8 i, i < 4: 16 bytes per lane
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.y + (long)g * a.y_gs), 0, (int)((unsigned)a.B * (unsigned)C * (unsigned)a.T * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rawrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RAW ? a.raw + (long)g * a.raw_gs : a.y), 0, (int)((unsigned)a.B * (unsigned)a.Cout * (unsigned)a.T * 4u), 0x00020000);
    const int CT4 = C * T4;
    // where the wave's 32 x 32 output tile of `tile` goes: the lane's byte offsets for flush row 0 (tile < 0 or past the end, or
    // pixels past the end of the sample: invalid -> an out-of-range offset, dropped by the hardware)
    struct OutPos { bool ok; unsigned yb, rb; };
    auto out_pos = [&](const Cursor& cu) {
        OutPos p;
        const bool tv = cu.tile >= 0 && cu.tile < ntiles;
        const int b = tv ? cu.b : 0;
        const int t = (tv ? cu.tt : 0) * TP + w4 * 32 + 4 * (lane & 7), row = slab * 32 + (lane >> 3);
        p.ok = tv && t < a.T;
        p.yb = (unsigned)(((b * C + row) * a.T + t) * 4);
        p.rb = (unsigned)(((b * 2 * C + row) * a.T + t) * 4);
        return p;
    };
    float* const sw_y = Ss + (w4 * NSTT) * 32 * LDS_S;                      // GLU output plane
    float* const sw_r = Ss + (w4 * NSTT + (NSTT - 1)) * 32 * LDS_S;         // raw value / gate planes, one after the other
    auto put_quad = [&](float* sw, const f32x16& v, int q) {                // quad q: pixels 8 q + 4 h .. + 3 of channel row l31
        const f32x4 x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        *reinterpret_cast<f32x4*>(sw + l31 * LDS_S + 8 * q + 4 * h) = x;
    };
    // A plane leaves in two halves: 4 reads of the tile (lane l: pixels 4 (l & 7) .. + 3 of rows (l >> 3) + 8 i), then -- a K
    // step later in the bf16 schedule, so that the LDS round trip hides behind MFMAs -- 4 stores, always issued.  Wave-private
    // tile and the LDS queue is in order within a wave: no barrier between the quads' writes and these reads, nor between
    // these reads and the next plane's writes.
    auto read_plane = [&](const float* sw, f32x4 (&r)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const f32x4*>(sw + ((lane >> 3) + 8 * i) * LDS_S + 4 * (lane & 7));
    };
    auto store_plane = [&](const f32x4 (&r)[4], __amdgpu_buffer_rsrc_t rs, unsigned base, bool ok, int plane_off) {
        const unsigned off = ok ? base : VC_OOB_ST;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r[i]), rs, (int)off, plane_off + i * 8 * T4, 0);
    };
    auto flush_plane = [&](const float* sw, __amdgpu_buffer_rsrc_t rs, unsigned base, bool ok, int plane_off) {
        f32x4 r[4];
        read_plane(sw, r); store_plane(r, rs, base, ok, plane_off);
    };
    // value * sigmoid(gate) = value * rcp(1 + exp2(-log2(e) gate)): hardware exp2 and reciprocal (1 ulp each).  One wave per
    // SIMD means nobody covers a DEPENDENT chain (mul -> exp -> add -> rcp -> mul: the in-order wave cannot issue the next MFMA
    // pair from behind it), so the chain is cut into STAGES applied to four values at a time: four independent instructions
    // per stage, one stage per K step (bf16), i.e. a stage's inputs are two MFMAs old when it runs.
    auto glu_stage = [&](const f32x16& sv, const f32x16& sg, f32x4& t, f32x16& o, int q, int st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (st == 0) t[j] = sg[4 * q + j] * -1.44269504f;
            else if (st == 1) t[j] = __builtin_amdgcn_exp2f(t[j]);
            else if (st == 2) t[j] = 1.f + t[j];
            else if (st == 3) t[j] = __builtin_amdgcn_rcpf(t[j]);
            else o[4 * q + j] = sv[4 * q + j] * t[j];
        }
    };
    auto glu_all = [&](const f32x16& sv, const f32x16& sg, f32x16& o) {     // outside the MFMA loop: the scheduler interleaves the 16 chains
        f32x4 t[4];
#pragma unroll
        for (int st = 0; st < 5; ++st)
#pragma unroll
            for (int q = 0; q < 4; ++q) glu_stage(sv, sg, t[q], o, q, st);
    };
    // a whole tile's planes, one after the other (fp32 trips, the last tile, the dummy tile): NST stores
    auto store_tile = [&](const f32x16& sv, const f32x16& sg, const f32x16& o, const OutPos& p) {
        if constexpr (RAW) {                                // training: the raw value / gate pair is kept for the backward pass
#pragma unroll
            for (int q = 0; q < 4; ++q) put_quad(sw_r, sv, q);
            flush_plane(sw_r, rawrs, p.rb, p.ok, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) put_quad(sw_r, sg, q);
            flush_plane(sw_r, rawrs, p.rb, p.ok, CT4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) put_quad(sw_y, o, q);
        flush_plane(sw_y, yrs, p.yb, p.ok, 0);
    };
    // dummy "tile -2" stores: with them the first trip sees the same operation count behind its frames as every other
    {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        store_tile(z, z, z, out_pos(retreat(cursor_at(tile0))));      // tile0 - tstep < 0: every lane out of range
    }
    __syncthreads();                                        // filter fragments and tile 0 are in LDS
    // bf16: value and gate fragments in registers (184 of the 512).  fp32: 360 fragment registers + accumulators + frames in
    // flight do not fit, and a register allocator under pressure parks the in-flight frame registers in AGPRs -- which copies
    // them BEFORE their loads have landed (tests/test_glu_isa_cpu.py caught exactly that on the first fp32 build).  fp32 is
    // bound by its 64-cycle MFMAs (8 per K step), so it re-reads both fragments from LDS behind them: two 16-byte reads per
    // 512 cycles of matrix work, nowhere near the LDS pipe's capacity, and ~130 registers in all.
    constexpr bool W_REGS = BF16;
    frag_t wv[W_REGS ? NU : 1], wgt[W_REGS ? NU : 1];
    if constexpr (W_REGS) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            wv[u] = Wf[(2 * u) * 64 + lane];
            wgt[u] = Wf[(2 * u + 1) * 64 + lane];
            // Pin the gate fragments in ACCUMULATOR registers (gfx950: one 512-entry file; an MFMA reads its B operand from
            // either half).  Left alone the allocator keeps all 184 fragment registers in the VGPR half, runs out, "spills"
            // a third of them to AGPRs and copies each back (4 v_accvgpr_mov) in front of every use -- 8 extra issue slots
            // per MFMA pair.  Defined through an "a" constraint they are AGPR values from the start and are used in place.
            asm volatile("" : "+a"(wgt[u]));
        }
    }

    // the bias as two constant accumulator tuples (a lane's 16 outputs are 16 pixels of ONE channel), kept in the AGPR half
    f32x16 bvv, bgv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { bvv[r] = bvl; bgv[r] = bgl; }
    asm volatile("" : "+a"(bvv)); asm volatile("" : "+a"(bgv));
    Cursor scur = retreat(cursor_at(tile0));             // tile whose planes the next trip stores (the first trip: a dummy in front of tile0)
    f32x16 accv[2], accg[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { accv[0][r] = accv[1][r] = 0.f; accg[0][r] = accg[1][r] = 0.f; }

    // One trip: multiply tile `tile` (LDS slot P) into accumulator set P (bf16) / 0 (fp32); frames of tile + 2 steps -> register
    // set P; frames of tile + 1 step (register set P ^ 1) -> LDS slot P ^ 1; GLU + stores of the previous tile.
    auto trip = [&](auto PC, int tile) {
        constexpr int P = decltype(PC)::value, AC = OVERLAP ? P : 0, AP = OVERLAP ? P ^ 1 : 0;
        float (&xp)[NXL] = P ? xr1 : xr0;                   // set that receives tile + 2 steps
        float (&xq)[NXL] = P ? xr0 : xr1;                   // set that holds tile + 1 step
        const T* xrow = Xs + P * SLOT + (w4 * 32 + l31) * CS + h * R;
        T* nslot = Xs + (P ^ 1) * SLOT;
        const OutPos pp = out_pos(scur); advance(scur);   // the previous tile's output position
        f32x16 o;
        if constexpr (!OVERLAP) {
            // fp32: the K loop is 360 MFMAs of 64 cycles; the previous tile's epilogue in front of it is < 2 % of a trip
            issue_loads(xp, lcur); advance(lcur);
            glu_all(accv[0], accg[0], o);
            store_tile(accv[0], accg[0], o, pp);
        }
        // fragment ring RD steps deep, reads pinned ahead of the MFMAs that use them
        constexpr int RD = 3, RS = RD + 1;
        frag_t fx[RS], fwv[W_REGS ? 1 : RS], fwg[W_REGS ? 1 : RS];
        auto rd = [&](auto UC) {
            constexpr int u = decltype(UC)::value;
            if constexpr (u < NU) {
                constexpr int off = BF16 ? 2 * R * u : (u / (RPT / 2)) * CS + (u % (RPT / 2)) * 2 * R;
                fx[u % RS] = *reinterpret_cast<const frag_t*>(xrow + off);
                if constexpr (!W_REGS) { fwv[u % RS] = Wf[(2 * u) * 64 + lane]; fwg[u % RS] = Wf[(2 * u + 1) * 64 + lane]; }
            }
        };
        auto mm = [&](auto UC) {
            constexpr int u = decltype(UC)::value;
            const frag_t x_ = fx[u % RS];
            if constexpr (BF16) {
                // step 0 starts the accumulators at the bias: the constant tuples are its C operand (no 32 moves per tile)
                accv[AC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x_, wv[u], u == 0 ? bvv : accv[AC], 0, 0, 0);
                accg[AC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x_, wgt[u], u == 0 ? bgv : accg[AC], 0, 0, 0);
            } else {
                const frag_t v_ = fwv[u % RS], g_ = fwg[u % RS];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    accv[AC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x_[e], v_[e], (u == 0 && e == 0) ? bvv : accv[AC], 0, 0, 0);
                    accg[AC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x_[e], g_[e], (u == 0 && e == 0) ? bgv : accg[AC], 0, 0, 0);
                }
            }
        };
        // bf16: what rides behind the MFMAs of step u.  Step 0: the next-but-one tile's loads; steps 0-3 (training): the previous
        // tile's raw value / gate planes through the transposition tile; steps 0-19: stage (u % 5) of its GLU quad (u / 5), the
        // finished quad into the tile; step 20: the GLU plane back out of the tile, step 21: its stores; steps 21-22: the next
        // tile's frames -> LDS.
        f32x4 gt, pl[4];
        auto side = [&](auto UC) {
            constexpr int u = decltype(UC)::value;
            if constexpr (OVERLAP) {
                if constexpr (u == 0) { issue_loads(xp, lcur); advance(lcur); }
                if constexpr (RAW) {
                    if constexpr (u == 0) { put_quad(sw_r, accv[AP], 0); put_quad(sw_r, accv[AP], 1); put_quad(sw_r, accv[AP], 2); put_quad(sw_r, accv[AP], 3); }
                    if constexpr (u == 1) read_plane(sw_r, pl);
                    if constexpr (u == 2) { store_plane(pl, rawrs, pp.rb, pp.ok, 0);
                                            put_quad(sw_r, accg[AP], 0); put_quad(sw_r, accg[AP], 1); put_quad(sw_r, accg[AP], 2); put_quad(sw_r, accg[AP], 3); }
                    if constexpr (u == 3) read_plane(sw_r, pl);
                    if constexpr (u == 5) store_plane(pl, rawrs, pp.rb, pp.ok, CT4);
                }
                if constexpr (u < 20) {
                    glu_stage(accv[AP], accg[AP], gt, o, u / 5, u % 5);
                    if constexpr (u % 5 == 4) put_quad(sw_y, o, u / 5);
                }
                if constexpr (u == 20) read_plane(sw_y, pl);
                if constexpr (u == 21) {
                    store_plane(pl, yrs, pp.yb, pp.ok, 0);
                    wait_counted(xq);
                    stage_task(xq, std::integral_constant<int, 0>{}, nslot);
                }
                if constexpr (u == 22) stage_task(xq, std::integral_constant<int, 1>{}, nslot);
            }
        };
        [&]<int... U>(std::integer_sequence<int, U...>) { (rd(std::integral_constant<int, U>{}), ...); }(std::make_integer_sequence<int, RD>{});
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((rd(std::integral_constant<int, U + RD>{}), __builtin_amdgcn_sched_barrier(0),
              mm(std::integral_constant<int, U>{}), side(std::integral_constant<int, U>{}), __builtin_amdgcn_sched_barrier(0)), ...);
        }(std::make_integer_sequence<int, NU>{});
        if constexpr (!OVERLAP) {
            wait_counted(xq);
            [&]<int... J>(std::integer_sequence<int, J...>) { (stage_task(xq, std::integral_constant<int, J>{}, nslot), ...); }
                (std::make_integer_sequence<int, NTPT>{});
        }
        __syncthreads();                                    // slot P ^ 1 is complete; slot P may be overwritten by the next trip
    };
    static_assert(!OVERLAP || (NU >= 23 && NTPT == 2), "the bf16 side schedule places its work on steps 0-22");

    int tile = tile0;
    for (;;) {
        if (tile >= ntiles) break;
        trip(std::integral_constant<int, 0>{}, tile); tile += tstep;
        if (tile >= ntiles) break;
        trip(std::integral_constant<int, 1>{}, tile); tile += tstep;
    }
    // ---- the last tile's GLU + stores (its accumulators: set ((trips - 1) & 1) in bf16 mode, set 0 in fp32)
    {
        const int ptile = tile - tstep;
        const int trips = (ptile - tile0) / tstep + 1;
        const OutPos pp = out_pos(scur);
        f32x16 o;
        if (OVERLAP && !(trips & 1)) {
            glu_all(accv[1], accg[1], o);
            store_tile(accv[1], accg[1], o, pp);
        } else {
            glu_all(accv[0], accg[0], o);
            store_tile(accv[0], accg[0], o, pp);
        }
    }
}

// y [G][B][Cout/2][T] = GLU of the gated pair; returns -2 when the geometry is not built
int launch_gated_conv1d_glu(const WsConvArgs& a, int cin, int kw, int math, hipStream_t s, double flops) {
    if (cin != 24 || kw != 15 || (a.Cout & 63) || (a.T & 3)) return -2;
    if ((long)a.B * a.Cout * a.T * 4 >= 0x7ff00000L) return -2;      // 32-bit buffer offsets; VC_OOB_ST must lie beyond every tensor
    const int C = a.Cout / 2, slabs = C / 32;
    const int ntiles = a.B * ((a.T + 127) / 128);
    int parts = 256 / (slabs * a.G); if (parts < 1) parts = 1; if (parts > ntiles) parts = ntiles;   // one workgroup per CU, persistent over its tiles
    dim3 grid(parts, slabs, a.G);
    ProfScope ps(2, flops, s);
#define VC_GLU(BF) { if (a.raw) gated_conv1d_glu_kernel<24, 15, BF, true><<<grid, dim3(256), 0, s>>>(a); \
                     else gated_conv1d_glu_kernel<24, 15, BF, false><<<grid, dim3(256), 0, s>>>(a); }
    if (math == VC_MATH_BF16) VC_GLU(true) else VC_GLU(false)
#undef VC_GLU
    VC_HIP(hipGetLastError());
    return 0;
}
