// vc_conv_glu.hip -- gated_conv1d_glu_kernel: the generator's h1 block (module.py:161-163) in ONE kernel,
//     out = (conv_value(x) + b_v) * sigmoid(conv_gate(x) + b_g),        x [B][24][T] fp32  ->  out [B][C][T] fp32
// the kernel BASELINE.json's north_star names ("gated-conv1d forward at B=256").  HBM sees the frames once and the
// activation once (12 KB in + 64 KB out per sample).
//
// Design (round 5).  D[channel][pixel] = sum_k W[channel][k] * X[k][pixel], k = tap * 24 + cin, per workgroup a slab of 32
// value + 32 gate channels and a stream of 128-pixel tiles:
//   * EIGHT waves, two per SIMD, split the K axis (round 4: four waves, one per SIMD, 184 filter registers each; counters:
//     ~6.5 other instructions between two MFMAs of the lone in-order wave, matrix pipe 38 % busy).  Wave (pb, S): pixel
//     block pb of the tile, K steps of set S (11 / 12 of the 23 bf16 steps).  The FILTERS of its K steps are the MFMA's A
//     operand and live in REGISTERS for the life of the workgroup (88 / 96 of the wave's 256, the gate fragments and 5 / 8 value
//     fragments in the AGPR half); the FRAMES are the B operand,
//     one ds_read_b128 per MFMA pair, out of a pixel-major LDS tile ([pixel][24 cin], 48-byte rows: run q of 8 consecutive
//     k starts 16 q bytes after the pixel's row start because consecutive taps are consecutive rows; writes and fragment
//     reads conflict-free).  Operand reuse is the round-4 kernel's: every frame fragment and every filter fragment is read
//     once per tile and CU.
//   * the two K sets of a pixel block meet through LDS once per tile: each wave hands over the half of its partial sums
//     that the partner finishes (16 registers out, 16 in, register index = channel group, so the split needs no shuffle)
//     -- the igemm kernel's K-group scheme.  fp32: same structure, filters in LDS (360 fragment registers do not fit),
//     exact v_mfma_f32_32x32x2_f32; bf16: operands rounded RNE once, v_mfma_f32_32x32x16_bf16.
//   * channels are the MFMA's M axis: a lane ends up with ONE pixel of 16 channels, value and gate of a channel in the same
//     register index -> bias (the first MFMA's C operand) + GLU happen in registers, and register i of the 32 lanes of a
//     half-wave is 128 contiguous bytes of one channel row: outputs leave as dword stores of two full 128-byte lines per
//     instruction.  (Round 4 had pixels on M: 16-byte stores touched 32 rows x 32 bytes, cured with an LDS transposition
//     of every plane; that LDS traffic is gone.)
//   * filters arrive PRE-PACKED (vc_gated_conv1d_glu_prepack: the fragment image [slab][K step][value|gate][lane][16 B],
//     built once per weight update): a workgroup copies its slab's image global -> LDS once, without registers (LDS-DMA), while
//     tile 0's frames travel, and every wave reads the fragments of its K steps out of LDS into registers.  Round 4 re-read
//     92 KB of fp32 filters per workgroup with 4-byte loads and converted them: 4.7 + 0.7 us in front of the first MFMA.  The
//     unpacked entry point keeps that path (PRE = false) for callers whose weights change before every call (the train step).
//   * software pipeline over the workgroup's tiles (one barrier per tile): while tile i is multiplied, the frames of
//     tile i+2 are in flight (inline-asm buffer loads retired by a COUNTED s_waitcnt: gfx950 has one in-order counter for
//     loads and stores, and the compiler would wait vmcnt(0), i.e. for the previous tile's output stores to complete),
//     the frames of tile i+1 are converted and written to the other LDS slot, and -- bf16 -- the hand-over, GLU and stores of
//     the wave's half of tile i-1 ride between the MFMAs.  Every trip issues the same number of loads and stores (out-of-tensor
//     lanes carry an out-of-range buffer offset), which is what makes the wait countable.
//   * the compiler knows nothing about registers whose inline-asm loads are in flight, so it must never get the chance to copy
//     them: inside the loop the in-flight set only crosses trip boundaries (tests/test_glu_isa_cpu.py reads every instruction
//     there), and the PROLOGUE retires the first two tiles' loads with one wait before control reaches the loop's entry -- where
//     the compiler does place register copies.  (Round 5 shipped a build without that for a few hours: 0.1 - 0.4 % of the launches
//     of the un-packed bf16 kernel at B >= 512 computed their second tile from copies of registers that had not landed.)
#include "vc_internal.h"
#include "vc_device_util.h"
#include <type_traits>
#include <utility>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define VC_OOB_ST 0x80000000u      // store offset beyond every buffer this kernel accepts (< 2 GiB); adding an immediate cannot wrap

// element e of a fragment from an fp32 value.  IEEE half goes pair-wise through a vector conversion: gfx950 has v_cvt_pk_f16_f32 (two RNE
// conversions + the pack in one instruction), which the compiler takes for <2 x float> -> <2 x half> but not always for two scalar casts
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
template <typename T, int R, typename F, typename X> __device__ __forceinline__ void glu_cvt(F& f, X&& src) {      // f[e] = (T)src(e), e < R
    if constexpr (std::is_same<T, _Float16>::value) {
#pragma unroll
        for (int e = 0; e < R; e += 2) {
            const f16x2 h = __builtin_convertvector(f32x2_{src(e), src(e + 1)}, f16x2);
            f[e] = h[0]; f[e + 1] = h[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < R; ++e) f[e] = (T)src(e);
    }
}

// geometry of the fragment image, shared by the kernel, the pre-pack kernel and the host
template <int CIN, int KW, bool BF16> struct GluGeo {
    static constexpr int R = BF16 ? 8 : 4;                 // k values per lane and K step (one 16-byte fragment)
    static constexpr int RPT = CIN / R;                    // runs per tap
    static constexpr int NRUN = KW * RPT, NU = (NRUN + 1) / 2;   // K steps: lane half h multiplies run 2u + h
    static constexpr int NU0 = NU / 2, NU1 = NU - NU0;           // K steps of wave set 0 / 1 (set 0, which also carries the bias tuples, gets the smaller share)
    static constexpr int K = CIN * KW;
    static constexpr long SLAB_BYTES = 2L * NU * 64 * 16;  // fragment unit U = 2u + vg: 64 lanes x 16 bytes
};

// image[g][slab][U = 2u + vg][lane] (16 bytes): lane (channel l31 of the slab, half h) holds W[k = R (2u + h) + e][channel], e < R
template <int CIN, int KW, bool BF16, bool HALF = false>      // HALF (with BF16): the 16-bit form is IEEE half instead of bfloat16
__global__ __launch_bounds__(256) void glu_prepack_kernel(const float* __restrict__ w, long w_gs, char* __restrict__ img, long img_gs, int Cout) {
    typedef GluGeo<CIN, KW, BF16> GG;
    const int C = Cout >> 1, slabs = C / 32, g = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= slabs * 2 * GG::NU * 64) return;
    const int lane = i & 63, U = (i >> 6) % (2 * GG::NU), slab = (i >> 6) / (2 * GG::NU);
    const int u = U >> 1, vg = U & 1, l31 = lane & 31, h = lane >> 5;
    const int ch = (vg ? C : 0) + slab * 32 + l31;
    const float* wg_ = w + (long)g * w_gs;
    float v[GG::R];
#pragma unroll
    for (int e = 0; e < GG::R; ++e) {
        const int k = GG::R * (2 * u + h) + e;
        v[e] = k < GG::K ? wg_[(long)k * Cout + ch] : 0.f;
    }
    char* dst = img + (long)g * img_gs + (long)i * 16;
    if constexpr (BF16 && HALF) {
        f16x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (_Float16)v[e];
        *reinterpret_cast<f16x8*>(dst) = f;
    } else if constexpr (BF16) {
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x8*>(dst) = f;
    } else {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    }
}

// LDS map of the kernel (one allocation, shared by the two bodies)
template <int CIN, int KW, bool BF16, bool PRE> struct GluLds {
    typedef GluGeo<CIN, KW, BF16> GG;
    static constexpr int CS = BF16 ? CIN : CIN + 4, SLOT = (144 + 1) * CS, ESZ = BF16 ? 2 : 4;
    static constexpr bool W_LDS = true;                    // the slab's fragment image always passes through (bf16) or stays in (fp32) LDS
    static constexpr int NXB = BF16 ? 2 : 1;
    static constexpr int WF_OFF = (2 * SLOT * ESZ + 15) & ~15;
    static constexpr int EX_OFF = WF_OFF + (W_LDS ? 2 * GG::NU * 64 * 16 : 0);
    static constexpr int BYTES = EX_OFF + NXB * 8 * 4 * 64 * 16;
};

// the kernel's body for the waves of K set S (a wave-uniform branch at the kernel's entry picks one: the set's K steps, which half of
// the accumulators it hands over and which half it finishes are compile-time constants, and no register is live across a join
// of the two).  Both bodies execute the same sequence of workgroup barriers.
template <int CIN, int KW, bool BF16, bool RAW, bool PRE, bool HALF, int S>
__device__ __forceinline__ void gated_conv1d_glu_body(const WsConvArgs& a, char* lds) {
    typedef GluGeo<CIN, KW, BF16> GG;
    static_assert(BF16 || !HALF, "HALF selects the element type of the 16-bit form");
    typedef typename std::conditional<BF16, typename std::conditional<HALF, _Float16, __bf16>::type, float>::type T;
    typedef typename std::conditional<BF16, typename std::conditional<HALF, f16x8, bf16x8>::type, f32x4>::type frag_t;
    constexpr int R = GG::R, RPT = GG::RPT, NU = GG::NU, K = GG::K;
    static_assert(CIN % R == 0 && (BF16 || RPT % 2 == 0), "runs of R channels inside one tap; fp32: both lane halves in the same tap");
    constexpr int TP = 128, PADL = (KW - 1) / 2;
    constexpr int XR = 144;                                 // staged rows: TP + KW - 1 frames + the row the zero-weight pad run touches
    static_assert(TP + KW <= XR && (KW & 1), "tile + halo fits the staged rows; odd kernel width");
    // Row stride in elements.  bf16: 24 (48 B) -- run q of a pixel's K axis then starts 16 q bytes after its row.  fp32: 28
    // (112 B, conflict-free 16-byte reads across pixels); run q = (tap, quad o) starts at tap * 28 + 4 o floats.
    constexpr int CS = BF16 ? CIN : CIN + 4;
    constexpr int SLOT = (XR + 1) * CS;                     // + one dummy row for the threads that have no staging task
    constexpr int NTASK = XR * RPT, NTPT = (NTASK + 511) / 512, NXL = NTPT * R;   // staging tasks (pixel row, run), per thread, loads per thread
    static_assert(NXL == 8, "operand lists of the counted waits below");
    constexpr int NST = RAW ? 24 : 8;                       // dword output stores per tile and lane
    constexpr bool OVERLAP = BF16;                          // GLU + stores of tile i-1 between the MFMAs of tile i
    constexpr bool W_LDS = true;                            // fragment image in LDS: fp32 reads it in the K loop, bf16 passes through it on its way into registers
    constexpr int NXB = BF16 ? 2 : 1;                       // exchange buffers (fp32: one, behind a second barrier per 10-us trip)
    typedef GluLds<CIN, KW, BF16, PRE> L;
    static_assert(L::SLOT == SLOT && L::NXB == NXB && L::W_LDS == W_LDS, "LDS map");
    T* const Xs = reinterpret_cast<T*>(lds);                                   // [2][SLOT]
    frag_t* const Wf = reinterpret_cast<frag_t*>(lds + L::WF_OFF);             // [2 NU][64] (if W_LDS)
    f32x4* const Ex = reinterpret_cast<f32x4*>(lds + L::EX_OFF);               // [buffer][sending wave][quad][lane]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = w8 & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int g = blockIdx.z, slab = blockIdx.y;
    const int C = a.Cout >> 1;
    const int tps = (a.T + TP - 1) / TP, ntiles = a.B * tps;
    const int T4 = a.T * 4;

    // ---- frame staging, per-thread constants: task j of a thread is (row r, run o) of EVERY tile
    unsigned goff[NTPT], ldsb[NTPT]; int rpos[NTPT];
#pragma unroll
    for (int j = 0; j < NTPT; ++j) {
        const int i = tid + 512 * j, o = i / XR, r = i - o * XR;
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
    // frames of `tile` -> a register set: R channels x NTPT rows per thread, lanes along time (coalesced); the channel's offset
    // rides in the load's SGPR offset.  Positions outside the sample (SAME padding) and tiles past the end get an out-of-
    // range VGPR offset: the hardware returns 0.  Always NXL load instructions.
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
#define VC_XR_OPS "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
    // prologue: everything issued so far has landed; BOTH register sets pass through the asm (see below)
    auto wait_all2 = [&](float (&x)[NXL], float (&z)[NXL]) {
        asm volatile("s_waitcnt vmcnt(0)" : VC_XR_OPS, "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]) :: "memory");
    };
    // younger operations at the point of use: the next-but-one tile's NXL loads + two tiles' output stores
    static_assert(NXL + 2 * NST == 24 || NXL + 2 * NST == 56, "the two counted waits below");
    auto wait_counted = [&](float (&x)[NXL]) {
        if constexpr (NST == 8) asm volatile("s_waitcnt vmcnt(24)" : VC_XR_OPS :: "memory");
        else asm volatile("s_waitcnt vmcnt(56)" : VC_XR_OPS :: "memory");
    };
#undef VC_XR_OPS
    // task j of a register set -> LDS slot: R values converted once, one 16-byte write
    auto stage_task = [&](const float (&x)[NXL], auto JC, T* slot) {
        constexpr int j = decltype(JC)::value;
        frag_t f;
        glu_cvt<T, R>(f, [&](int e) { return x[j * R + e]; });
        *reinterpret_cast<frag_t*>(reinterpret_cast<char*>(slot) + ldsb[j]) = f;
    };
    auto stage_all = [&](const float (&x)[NXL], T* slot) {
        [&]<int... J>(std::integer_sequence<int, J...>) { (stage_task(x, std::integral_constant<int, J>{}, slot), ...); }
            (std::make_integer_sequence<int, NTPT>{});
    };

    // ---- prologue: the first TWO tiles' frames travel while the filter fragments arrive, and BOTH are retired by the one wait below.
    // Nothing may be in flight when control leaves the prologue: at the loop's entry the compiler is free to shuffle registers (it
    // did: v_mov copies of the second set's registers in front of the first trip's wait in every bf16 instantiation), and a copy of a
    // register whose load has not landed is a copy of garbage -- 0.1 - 0.4 % of the launches of the un-packed bf16 kernel at
    // B >= 512 staged such frames for their second tile (round 5, tools/stress_glu.py; the ISA test only policed the loop body).
    // Inside the loop the in-flight set crosses trip boundaries only, where tests/test_glu_isa_cpu.py reads every instruction.
    const int tile0 = blockIdx.x, tstep = gridDim.x;
    Cursor lcur = cursor_at(tile0);                      // next tile whose frames are requested
    issue_loads(xr0, lcur); advance(lcur);
    issue_loads(xr1, lcur); advance(lcur);
    [[maybe_unused]] const char* const wimg = PRE ? (const char*)a.wimg + (long)g * a.wimg_gs + (long)slab * GG::SLAB_BYTES : nullptr;
    // bf16: the wave's filter fragments, value and gate, of its K steps (read out of the LDS image behind the prologue's barrier)
    constexpr bool W_REGS = BF16;
    constexpr int NUS = S ? GG::NU1 : GG::NU0, U0 = S ? GG::NU0 : 0;
    frag_t wv[W_REGS ? NUS : 1], wgt[W_REGS ? NUS : 1];
    constexpr int NVV = S ? NUS - 8 : NUS - 5;                            // value fragments [NVV, NUS) live in the accumulator half as well (set 0: 5 of them, which fills its 128 AGPRs beside gate fragments, accumulators and
                                                            // bias tuples; set 1: 8): the VGPR half (128 of the
                                                            // wave's 256) also holds the frames in flight, the fragment ring and the previous tile's GLU
    // the bias as the first MFMA's C operand: register i of a lane is channel row 8 (i / 4) + 4 h + i % 4.  Set 1 starts from 0.
    f32x16 bvv, bgv;
#pragma unroll
    for (int i = 0; i < 16; ++i) { bvv[i] = 0.f; bgv[i] = 0.f; }
    if constexpr (S == 0) {
        if (a.bias) {                                       // one uniform branch around all 32 loads
            const float* bp = a.bias + (long)g * a.b_gs + slab * 32 + 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) { bvv[i] = bp[8 * (i >> 2) + (i & 3)]; bgv[i] = bp[C + 8 * (i >> 2) + (i & 3)]; }
        }
    }

    if constexpr (PRE) {
        // the slab's pre-packed image -> LDS as it lies, ONCE per workgroup, global -> LDS without registers (buffer_load_dwordx4 ... lds:
        // lane l's 16 bytes land at the wave-uniform LDS address + 16 l; wave w copies units w, w + 8, ..).  Round 5's first build
        // let every wave fetch its own fragments straight into registers: the four waves of a K set fetch the same 24 KB, 188 KB per
        // CU through the vector L1 -- the probe put 3.0 - 5.5 k cycles on just ISSUING those loads.
#if defined(__HIP_DEVICE_COMPILE__)
        const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)wimg, 0, (int)GG::SLAB_BYTES, 0x00020000);
        constexpr int NB = (2 * NU + 7) / 8;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int U = w8 + 8 * it;
            if (U < 2 * NU)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)(Wf + U * 64), 16, (unsigned)lane * 16u, U * 1024, 0, 0);
        }
#endif
    }
    if constexpr (W_LDS && !PRE) {
        // Fragment unit U = 2u + vg straight from the TF-layout filters: wave w builds units w, w + 8, ..: R coalesced loads
        // each (32 channels = 128 bytes per k row), all issued before the first conversion (ONE memory round trip), one
        // conflict-free 16-byte LDS write per unit.
        constexpr int NB = (2 * NU + 7) / 8;
        const float* wg_ = a.w + (long)g * a.w_gs;
        float v[NB][R];
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int U = w8 + 8 * it, u = U >> 1, vg = U & 1;
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
            const int U = w8 + 8 * it;
            if (U < 2 * NU) {
                frag_t f;
                glu_cvt<T, R>(f, [&](int e) { return v[it][e]; });
                Wf[U * 64 + lane] = f;
            }
        }
    }
    wait_all2(xr0, xr1);
    // the bias tuples are consumed HERE (everything issued so far has landed)
    if constexpr (S == 0) { asm volatile("" : "+a"(bvv)); asm volatile("" : "+a"(bgv)); }
    stage_all(xr0, Xs);
    // output addressing: accumulator register i of a lane is pixel l31 of channel row 8 (i / 4) + 4 h + i % 4 of the slab: the
    // lane's byte offset carries (sample, slab, 4 h, pixel), the register's row rides in the store's SGPR offset
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.y + (long)g * a.y_gs), 0, (int)((unsigned)a.B * (unsigned)C * (unsigned)a.T * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rawrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RAW ? a.raw + (long)g * a.raw_gs : a.y), 0, (int)((unsigned)a.B * (unsigned)a.Cout * (unsigned)a.T * 4u), 0x00020000);
    const int CT4 = C * T4;
    // where the wave's 32 x 32 output tile of `tile` goes (tile < 0 or past the end, or pixels past the end of the sample:
    // invalid -> an out-of-range offset, dropped by the hardware)
    struct OutPos { unsigned yb, rb; };
    // lane part (row 4 h of the slab, pixel pb * 32 + l31 of the tile) + uniform part (sample, tile) of the byte offset: the same lane
    // constant serves the GLU output and the raw pair (their rows differ by the uniform sample stride only)
    const int lane_t = pb * 32 + l31;
    const unsigned lane_off = (unsigned)(((slab * 32 + 4 * h) * a.T + lane_t) * 4);
    auto out_pos = [&](const Cursor& cu) {
        OutPos p;
        const bool tv = cu.tile >= 0 && cu.tile < ntiles;
        const int b = tv ? cu.b : 0, t0 = (tv ? cu.tt : 0) * TP;
        const bool ok = tv && t0 + lane_t < a.T;
        p.yb = ok ? lane_off + (unsigned)((b * C * a.T + t0) * 4) : VC_OOB_ST;
        p.rb = ok ? lane_off + (unsigned)((b * 2 * C * a.T + t0) * 4) : VC_OOB_ST;
        return p;
    };
    // dummy "tile -2" stores: with them the first trip sees the same operation count behind its frames as every other
#pragma unroll
    for (int i = 0; i < NST; ++i) __builtin_amdgcn_raw_buffer_store_b32(0u, yrs, (int)VC_OOB_ST, 0, 0);
    __syncthreads();                                        // fragment image (if any) and tile 0 are in LDS

    // value * sigmoid(gate) = value * rcp(1 + exp2(-log2(e) gate)): hardware exp2 and reciprocal (1 ulp each), cut into STAGES
    // of four independent instructions (a dependent chain would stall the in-order wave and the MFMAs behind it)
    auto glu_stage = [&](const f32x4& sv, const f32x4& sg, f32x4& t, int st) {     // the result replaces t
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (st == 0) t[j] = sg[j] * -1.44269504f;
            else if (st == 1) t[j] = __builtin_amdgcn_exp2f(t[j]);
            else if (st == 2) t[j] = 1.f + t[j];
            else if (st == 3) t[j] = __builtin_amdgcn_rcpf(t[j]);
            else t[j] = sv[j] * t[j];
        }
    };

    {
        constexpr int OWN = S ? 2 : 0, SEND = S ? 0 : 2;   // first accumulator quad (of 4) the wave finishes / hands over
        // store rows of the wave's 8 own registers: register 4 q + r of quad OWN + q
        const int srow0 = (8 * OWN) * T4, srow1 = (8 * (OWN + 1)) * T4;

        if constexpr (W_REGS) {
#pragma unroll
            for (int j = 0; j < NUS; ++j) {
                wv[j] = Wf[(2 * (U0 + j)) * 64 + lane];
                wgt[j] = Wf[(2 * (U0 + j) + 1) * 64 + lane];
            }
            // Pin the gate fragments in ACCUMULATOR registers (gfx950: one register file; an MFMA reads its A operand from either
            // half).  Left alone the allocator keeps every fragment in the VGPR half, runs out, parks some in AGPRs and
            // copies each back in front of every use.  Defined through an "a" constraint they are used in place.
#pragma unroll
            for (int j = 0; j < NUS; ++j) {
                asm volatile("" : "+a"(wgt[j]));
                if (j >= NVV) asm volatile("" : "+a"(wv[j]));
            }
        }
        f32x16 zero16;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
        Cursor scur = retreat(cursor_at(tile0));         // tile whose outputs the next trip stores (the first trip: a dummy in front of tile0)
        f32x16 accv, accg;
        // the wave's own half of the previous tile's partial sums (copied out of the accumulators behind the K loop); the partner's
        // contribution waits in the exchange buffer and is added inside the NEXT trip, between its MFMAs
        f32x4 ownv[2], owng[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { accv[r] = 0.f; accg[r] = 0.f; }
#pragma unroll
        for (int q = 0; q < 2; ++q) { ownv[q] = f32x4{0.f, 0.f, 0.f, 0.f}; owng[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // The second-dispatched half of the workgroup (waves 4-7 = K set 1) loses every issue arbitration against its older partner
        // on the same SIMD (measured: set 0 finished its K loop after 1430 cycles and then idled 1240 at the barrier while set 1
        // needed 2150): raise it once, no per-segment flips (MI355X_MICROARCH.md, two waves per SIMD, item 4).
        if constexpr (S == 1) { if (!(a.flags & 1)) __builtin_amdgcn_s_setprio(1); }   // vc_tune_glu bit 0 switches it off (A/B: 1 % at B = 256)

        auto store_quad = [&](const f32x4& v, __amdgpu_buffer_rsrc_t rs, unsigned off, int srow) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xr = v[r];                      // (a bit_cast of the vector-element lvalue itself reads element 0: clang)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xr), rs, (int)off, srow + r * T4, 0);
            }
        };
        auto quad = [](const f32x16& v, int q) { return f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; };
        // exchange buffer of trip parity P: [sending wave][quad: value 0, value 1, gate 0, gate 1][lane]
        auto ex_of = [&](int P, int wave) { return Ex + (((NXB > 1 ? P : 0) * 8 + wave) * 4) * 64 + lane; };
        // previous tile, finished outside the MFMA loop (fp32 trips, the last tile): partner's half + own half, GLU, NST stores
        auto finish_tile = [&](const f32x4* ex_in, const OutPos& p) {
            f32x4 fv[2], fg[2], t[2];
            fv[0] = ownv[0] + ex_in[0]; fv[1] = ownv[1] + ex_in[64]; fg[0] = owng[0] + ex_in[128]; fg[1] = owng[1] + ex_in[192];
#pragma unroll
            for (int st = 0; st < 5; ++st)
#pragma unroll
                for (int q = 0; q < 2; ++q) glu_stage(fv[q], fg[q], t[q], st);
            if constexpr (RAW) {                            // training: the raw value / gate pair is kept for the backward pass
                store_quad(fv[0], rawrs, p.rb, srow0); store_quad(fv[1], rawrs, p.rb, srow1);
                store_quad(fg[0], rawrs, p.rb, srow0 + CT4); store_quad(fg[1], rawrs, p.rb, srow1 + CT4);
            }
            store_quad(t[0], yrs, p.yb, srow0); store_quad(t[1], yrs, p.yb, srow1);
        };

        // One trip: multiply tile `tile` (LDS slot P); frames of tile + 2 steps -> register set P; frames of tile + 1 step
        // (register set P ^ 1) -> LDS slot P ^ 1; the previous tile's hand-over, GLU and stores; this tile's hand-over out.
        auto trip = [&](auto PC) {
            constexpr int P = decltype(PC)::value;
            float (&xp)[NXL] = P ? xr1 : xr0;               // set that receives tile + 2 steps
            float (&xq)[NXL] = P ? xr0 : xr1;               // set that holds tile + 1 step
            const T* xrow = Xs + P * SLOT + (pb * 32 + l31) * CS + h * R;
            T* nslot = Xs + (P ^ 1) * SLOT;
            const OutPos pp = out_pos(scur); advance(scur);   // the previous tile's output position
            const f32x4* ex_in = ex_of(P ^ 1, w8 ^ 4);       // what the partner handed over behind the previous trip's K loop (first trip: never
                                                            // written -- whatever it holds goes to an out-of-range store offset)
            f32x4 xin[2], fv[2], fg[2], gt[2];
            if constexpr (!OVERLAP) {
                // fp32: the K loop is 180 MFMAs of 64 cycles per wave; the previous tile's epilogue in front of it is < 2 % of a trip
                issue_loads(xp, lcur); advance(lcur);
                finish_tile(ex_in, pp);
                __syncthreads();                            // single exchange buffer: every wave has read the previous hand-over before this trip's overwrites it
            }
            // fragment ring RD steps deep, reads pinned ahead of the MFMAs that use them
            constexpr int RD = BF16 ? 2 : 3, RS = RD + 1;   // (bf16: two steps ahead are 4 MFMAs = 128+ cycles of cover; a third costs the 4 registers that tip the VGPR half into scratch)
            frag_t fx[RS], fwv[W_REGS ? 1 : RS], fwg[W_REGS ? 1 : RS];
            auto rd = [&](auto JC) {
                constexpr int j = decltype(JC)::value;
                if constexpr (j < NUS) {
                    constexpr int u = U0 + j;
                    constexpr int off = BF16 ? 2 * R * u : (u / (RPT / 2)) * CS + (u % (RPT / 2)) * 2 * R;
                    fx[j % RS] = *reinterpret_cast<const frag_t*>(xrow + off);
                    if constexpr (!W_REGS) { fwv[j % RS] = Wf[(2 * u) * 64 + lane]; fwg[j % RS] = Wf[(2 * u + 1) * 64 + lane]; }
                }
            };
            // step 0 starts the accumulators at the bias (set 0) / zero (set 1): a constant tuple is its C operand
            auto mm = [&](auto JC, auto HC) {                // HC: 0 = the value MFMA(s) of step j, 1 = the gate MFMA(s)
                constexpr int j = decltype(JC)::value, half = decltype(HC)::value;
                const frag_t x_ = fx[j % RS];
                if constexpr (BF16 && HALF) {
                    if constexpr (half == 0) accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[j], x_, j == 0 ? (S ? zero16 : bvv) : accv, 0, 0, 0);
                    else accg = __builtin_amdgcn_mfma_f32_32x32x16_f16(wgt[j], x_, j == 0 ? (S ? zero16 : bgv) : accg, 0, 0, 0);
                } else if constexpr (BF16) {
                    if constexpr (half == 0) accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[j], x_, j == 0 ? (S ? zero16 : bvv) : accv, 0, 0, 0);
                    else accg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wgt[j], x_, j == 0 ? (S ? zero16 : bgv) : accg, 0, 0, 0);
                } else if constexpr (half == 0) {
                    const frag_t v_ = fwv[j % RS], g_ = fwg[j % RS];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        accv = __builtin_amdgcn_mfma_f32_32x32x2f32(v_[e], x_[e], (j == 0 && e == 0) ? (S ? zero16 : bvv) : accv, 0, 0, 0);
                        accg = __builtin_amdgcn_mfma_f32_32x32x2f32(g_[e], x_[e], (j == 0 && e == 0) ? (S ? zero16 : bgv) : accg, 0, 0, 0);
                    }
                }
            };
            // bf16: what rides behind the value MFMA (half 0) and the gate MFMA (half 1) of step j -- four instructions or so each, so
            // that neither wave of the SIMD holds the issue port for long.  Step 0: the next-but-one tile's loads; 1-3: the partner's
            // half of the previous tile out of the exchange buffer + the own half (value, then gate, through the same registers);
            // 4-8: GLU stage j - 4 (quad 0 | quad 1); 4-5 (training): raw value / gate stores; 9: output stores; 10: the next
            // tile's frames -> LDS.
            auto side = [&](auto JC, auto HC) {
                constexpr int j = decltype(JC)::value, half = decltype(HC)::value;
                if constexpr (OVERLAP) {
                    if constexpr (j == 0 && half == 0) { issue_loads(xp, lcur); advance(lcur); }
                    if constexpr (j == 1 && half == 0) { xin[0] = ex_in[0]; xin[1] = ex_in[64]; }
                    if constexpr (j == 2 && half == 0) { fv[0] = ownv[0] + xin[0]; fv[1] = ownv[1] + xin[1]; }
                    if constexpr (j == 2 && half == 1) { xin[0] = ex_in[128]; xin[1] = ex_in[192]; }
                    if constexpr (j == 3 && half == 1) { fg[0] = owng[0] + xin[0]; fg[1] = owng[1] + xin[1]; }
                    if constexpr (RAW) {
                        if constexpr (j == 4) store_quad(fv[half], rawrs, pp.rb, half ? srow1 : srow0);
                        if constexpr (j == 5) store_quad(fg[half], rawrs, pp.rb, (half ? srow1 : srow0) + CT4);
                    }
                    if constexpr (j >= 4 && j <= 8) glu_stage(fv[half], fg[half], gt[half], j - 4);
                    if constexpr (j == 9) store_quad(gt[half], yrs, pp.yb, half ? srow1 : srow0);
                    if constexpr (j == 10 && half == 0) { wait_counted(xq); stage_all(xq, nslot); }
                }
            };
            [&]<int... J>(std::integer_sequence<int, J...>) { (rd(std::integral_constant<int, J>{}), ...); }(std::make_integer_sequence<int, RD>{});
            [&]<int... J>(std::integer_sequence<int, J...>) {
                ((rd(std::integral_constant<int, J + RD>{}), __builtin_amdgcn_sched_barrier(0),
                  mm(std::integral_constant<int, J>{}, std::integral_constant<int, 0>{}), side(std::integral_constant<int, J>{}, std::integral_constant<int, 0>{}),
                  __builtin_amdgcn_sched_barrier(0),
                  mm(std::integral_constant<int, J>{}, std::integral_constant<int, 1>{}), side(std::integral_constant<int, J>{}, std::integral_constant<int, 1>{}),
                  __builtin_amdgcn_sched_barrier(0)), ...);
            }(std::make_integer_sequence<int, NUS>{});
            if constexpr (!OVERLAP) {
                wait_counted(xq);
                stage_all(xq, nslot);
            }
            // hand-over: the half of the partial sums the partner finishes goes to LDS, the own half out of the accumulators (the next
            // trip's first MFMAs overwrite them)
            f32x4* ex_out = ex_of(P, w8);
            ex_out[0] = quad(accv, SEND); ex_out[64] = quad(accv, SEND + 1);
            ex_out[128] = quad(accg, SEND); ex_out[192] = quad(accg, SEND + 1);
            ownv[0] = quad(accv, OWN); ownv[1] = quad(accv, OWN + 1); owng[0] = quad(accg, OWN); owng[1] = quad(accg, OWN + 1);
            __syncthreads();                                // hand-over written; slot P ^ 1 is complete; slot P may be overwritten by the next trip
        };
        static_assert(!OVERLAP || NUS >= 11, "the bf16 side schedule places its work on steps 0-10");

        int tile = tile0, par = 0;                          // par: parity of the NEXT trip = which exchange buffer the last trip did not use
        for (;;) {
            if (tile >= ntiles) break;
            trip(std::integral_constant<int, 0>{}); tile += tstep; par = 1;
            if (tile >= ntiles) break;
            trip(std::integral_constant<int, 1>{}); tile += tstep; par = 0;
        }
        // ---- the last tile: the partner's half lies in the buffer of the last trip's parity
        finish_tile(ex_of(par ^ 1, w8 ^ 4), out_pos(scur));
        if constexpr (S == 1) __builtin_amdgcn_s_setprio(0);
    }
}

template <int CIN, int KW, bool BF16, bool RAW, bool PRE, bool HALF>
__global__ __launch_bounds__(512, 1) void gated_conv1d_glu_kernel(const WsConvArgs a) {
    touch_kernarg(a);
    __shared__ __attribute__((aligned(16))) char lds[GluLds<CIN, KW, BF16, PRE>::BYTES];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) gated_conv1d_glu_body<CIN, KW, BF16, RAW, PRE, HALF, 0>(a, lds);
    else gated_conv1d_glu_body<CIN, KW, BF16, RAW, PRE, HALF, 1>(a, lds);
}

// same grid, same frame and fragment-image reads, same output stores, no arithmetic: what the memory system alone allows a launch of
// this shape (bench.py named_kernel.copy_roof; include/vc_cyclegan_testing.h: vc_prof_glu_copy_roof).  A streaming kernel: the loads of
// eight tiles are in flight while a tile's stores are issued, the stored value does not depend on the tile's loads (they are summed into
// a value that leaves once, at the end).
__global__ __launch_bounds__(512, 1) void glu_copy_roof_kernel(const WsConvArgs a) {
    constexpr int CIN = 24, TP = 128;
    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6, pb = w8 & 3, kset = w8 >> 2, l31 = lane & 31, h = lane >> 5;
    const int g = blockIdx.z, slab = blockIdx.y, C = a.Cout >> 1;
    const int tps = (a.T + TP - 1) / TP, ntiles = a.B * tps;
    const float* x = a.x + (long)g * a.x_gs;
    float* y = a.y + (long)g * a.y_gs;
    float acc = 0.f;
    f32x4 iv[6];
    {   // the slab's fragment image: 46 KB (bf16 form), 16 bytes per lane and unit, all loads in flight together
        const f32x4* img = reinterpret_cast<const f32x4*>((const char*)a.wimg + (long)g * a.wimg_gs + (long)slab * GluGeo<24, 15, true>::SLAB_BYTES);
#pragma unroll
        for (int i = 0; i < 6; ++i) iv[i] = img[(tid + 512 * i) < 46 * 64 ? tid + 512 * i : 0];
    }
    // a tile's frames: 24 channels x 128 frames = 768 x 16 bytes: threads < 384 carry two 16-byte loads each
    auto load_tile = [&](int tile, f32x4 (&v)[2]) {
        const int tcl = tile < ntiles ? tile : 0;
        const int b = tcl / tps, tt = tcl - b * tps;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 384 * j, c = i >> 5, t = tt * TP + 4 * (i & 31);
            const bool ok = tid < 384 && tile < ntiles && t < a.T;
            v[j] = ok ? *reinterpret_cast<const f32x4*>(x + ((long)b * CIN + c) * a.T + t) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // DEPTH tiles of frames in flight: a tile's loads are consumed DEPTH trips after they were issued (the first version of this control consumed
    // them one trip later and measured memory LATENCY per trip -- 1.6 us x 64 trips at B = 4096, by coincidence the kernel's own time)
    constexpr int DEPTH = 8;
    f32x4 buf[DEPTH][2];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) load_tile(blockIdx.x + k * (int)gridDim.x, buf[k]);
    auto store_tile = [&](int tile) {
        const int b = tile / tps, tt = tile - b * tps;
        const int t = tt * TP + pb * 32 + l31;
        if (a.flags & 4) {                                  // vc_tune_glu bit 2: the same 16 KB block as 16-byte stores (what an LDS-transposed epilogue would issue)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = tid + 512 * k, row = slab * 32 + (idx >> 5), tq = tt * TP + 4 * (idx & 31);
                if (tq < a.T) *reinterpret_cast<f32x4*>(y + ((long)b * C + row) * a.T + tq) = f32x4{1.f, 1.f, 1.f, 1.f};
            }
        } else if (t < a.T) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = slab * 32 + 8 * (2 * kset + (i >> 2)) + 4 * h + (i & 3);
                y[((long)b * C + row) * a.T + t] = 1.f;
            }
        }
    };
    for (int tile = blockIdx.x; tile < ntiles; ) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            if (tile < ntiles) {
                store_tile(tile);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc += buf[k][j][0] + buf[k][j][1] + buf[k][j][2] + buf[k][j][3];
                load_tile(tile + DEPTH * (int)gridDim.x, buf[k]);
                tile += gridDim.x;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc += iv[i][0] + iv[i][1] + iv[i][2] + iv[i][3];
    if (acc == 1.2345e-30f) y[0] = acc;                     // keeps the loads alive
}

long gated_conv1d_glu_image_bytes(int cin, int kw, int cout, int math) {
    if (cin != 24 || kw != 15 || (cout & 63)) return -1;
    const long slabs = cout / 64;
    return slabs * (math == VC_MATH_BF16 || math == VC_MATH_F16 ? GluGeo<24, 15, true>::SLAB_BYTES : GluGeo<24, 15, false>::SLAB_BYTES);
}

int launch_gated_conv1d_glu_prepack(const float* w, long w_gs, void* img, long img_gs_bytes, int G, int cin, int kw, int cout, int math, hipStream_t s) {
    if (gated_conv1d_glu_image_bytes(cin, kw, cout, math) < 0) return -2;
    const int slabs = cout / 64;
    if (math == VC_MATH_BF16 || math == VC_MATH_F16) {
        const int n = slabs * 2 * GluGeo<24, 15, true>::NU * 64;
        if (math == VC_MATH_F16) glu_prepack_kernel<24, 15, true, true><<<dim3((n + 255) / 256, G), dim3(256), 0, s>>>(w, w_gs, (char*)img, img_gs_bytes, cout);
        else glu_prepack_kernel<24, 15, true><<<dim3((n + 255) / 256, G), dim3(256), 0, s>>>(w, w_gs, (char*)img, img_gs_bytes, cout);
    } else {
        const int n = slabs * 2 * GluGeo<24, 15, false>::NU * 64;
        glu_prepack_kernel<24, 15, false><<<dim3((n + 255) / 256, G), dim3(256), 0, s>>>(w, w_gs, (char*)img, img_gs_bytes, cout);
    }
    VC_HIP(hipGetLastError());
    return 0;
}

static bool glu_grid(const WsConvArgs& a, int cin, int kw, dim3* grid) {
    if (cin != 24 || kw != 15 || (a.Cout & 63) || (a.T & 3)) return false;
    if ((long)a.B * a.Cout * a.T * 4 >= 0x7ff00000L) return false;   // 32-bit buffer offsets; VC_OOB_ST must lie beyond every tensor
    const int C = a.Cout / 2, slabs = C / 32;
    const int ntiles = a.B * ((a.T + 127) / 128);
    int parts = 256 / (slabs * a.G); if (parts < 1) parts = 1; if (parts > ntiles) parts = ntiles;   // one workgroup per CU, persistent over its tiles
    *grid = dim3(parts, slabs, a.G);
    return true;
}

// y [G][B][Cout/2][T] = GLU of the gated pair; a.wimg != null: the filters as a pre-packed fragment image of `math`; returns -2
// when the geometry is not built
int glu_tune_flags();       // vc_ops.hip: the calling thread's vc_tune_glu switches
int launch_gated_conv1d_glu(const WsConvArgs& a_, int cin, int kw, int math, hipStream_t s, double flops) {
    dim3 grid;
    if (!glu_grid(a_, cin, kw, &grid)) return -2;
    WsConvArgs a = a_; a.flags = glu_tune_flags();
    ProfScope ps(2, flops, s);
#define VC_GLU2(BF, RW, HF) { if (a.wimg) gated_conv1d_glu_kernel<24, 15, BF, RW, true, HF><<<grid, dim3(512), 0, s>>>(a); \
                              else gated_conv1d_glu_kernel<24, 15, BF, RW, false, HF><<<grid, dim3(512), 0, s>>>(a); }
#define VC_GLU(BF, HF) { if (a.raw) VC_GLU2(BF, true, HF) else VC_GLU2(BF, false, HF) }
    if (math == VC_MATH_F16) VC_GLU(true, true) else if (math == VC_MATH_BF16) VC_GLU(true, false) else VC_GLU(false, false)
#undef VC_GLU
#undef VC_GLU2
    VC_HIP(hipGetLastError());
    return 0;
}

int launch_glu_copy_roof(const WsConvArgs& a, int cin, int kw, hipStream_t s) {
    dim3 grid;
    if (!glu_grid(a, cin, kw, &grid) || !a.wimg) return -2;
    WsConvArgs b = a; b.flags = glu_tune_flags();
    ProfScope ps(2, 0.0, s);
    glu_copy_roof_kernel<<<grid, dim3(512), 0, s>>>(b);
    VC_HIP(hipGetLastError());
    return 0;
}
