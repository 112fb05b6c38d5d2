// vc_ops.hip -- convolution operators and the operator-level C-ABI (host side of libvcgan_hip.so, part 2 of 4: vc_host.h)
#include "vc_host.h"
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------
// conv operators
// ------------------------------------------------------------------------------------

// Cin = 1 with at most 9 taps (the discriminator's h1 layer) goes to the vector-ALU kernels
bool c1_path(const Geom& g) { return g.Cin == 1 && g.kh * g.kw <= 9; }
// 1-D, stride 1, 24 input channels, 15 taps (the generator's h1 layer at 24 MCEPs): weight-stationary kernel
bool wsconv_path(const Geom& g) {
    return g.H == 1 && g.kh == 1 && g.sh == 1 && g.sw == 1 && g.Cin == 24 && g.kw == 15 && g.W % 4 == 0;   // 16-byte output stores
}
C1Args c1_args(const Geom& g, int G, int B, int math) {
    C1Args c{};
    c.B = B; c.H = g.H; c.W = g.W; c.Cout = g.Cout; c.kh = g.kh; c.kw = g.kw; c.sh = g.sh; c.sw = g.sw;
    c.OH = g.OH; c.OW = g.OW; c.pt = g.pt; c.pl = g.pl; c.G = G; c.math = math; c.det = vc_det_active() ? 1 : 0;
    return c;
}
int check_conv_geom(const Geom& g, int B) {
    VC_ARG(g.Cout % 4 == 0, "conv: Cout must be a multiple of 4");
    VC_ARG(c1_path(g) || g.Cin % 4 == 0, "conv: Cin must be a multiple of 4 (or 1 with at most 9 taps)");
    VC_ARG((long)B * g.Cin * g.H * g.W * 4 < 0xfff00000L && (long)B * g.Cout * g.OH * g.OW * 4 < 0xfff00000L,
           "conv: a tensor of one group exceeds 4 GiB (32-bit buffer offsets)");
    VC_ARG((long)B * g.OH * g.OW < (1L << 24) && (long)B * g.H * g.W < (1L << 24), "conv: more than 2^24 pixels per group");
    return 0;
}
static void fill_pixel_grid(IgemmArgs& a, int B, int NPB, int OWc) {
    a.NPB = NPB; a.OWc = OWc; a.Ptot = B * NPB; a.inv_npb = 1.0f / (float)NPB; a.inv_ow = 1.0f / (float)OWc;
}
// Filter gradient of a 2-D layer whose kernel height reaches far into the SAME padding: does walking only the output rows a K
// tile's taps reach pay?  OHc output rows at row stride si over a gathered tensor of Hl rows: the share of (output row, tap)
// pairs whose tap row lies in the padding.  D.d3 (6 tap rows on 6 image rows, pads (2, 3)): 25 %; the 3x3 stride-2 layers:
// 3 %; 1-D layers: one row, never.  Measured (profiles/r03_pmc_d3_rowskip.txt): D.d3's filter gradient at the B = 8 step
// 545 -> 459 us (its reduction axis is the pixel axis, the skipped rows are simply not read).  The same enumeration for the
// forward / input gradient was built and measured in round 3 (-3 % .. +7 %: pixel tiles of one weight slab then walk different
// parts of it and lose its sharing through the XCD's L2) and removed in round 4.  vc_tune_rowskip switches it for the tests.
static bool rowskip_pays(const Table& t, int OHc, int si, int Hl) {
    if (!vc_hooks().rowskip_wgrad || OHc < 2 || t.ntaps < 2 || t.cpt <= 0) return false;
    long valid = 0;
    for (int i = 0; i < OHc; ++i)
        for (int tp = 0; tp < t.ntaps; ++tp) valid += (i * si + t.tdh[tp] >= 0 && i * si + t.tdh[tp] < Hl) ? 1 : 0;
    return valid * 100 < 92L * OHc * t.ntaps;
}
// implicit-GEMM launches (forward / input gradient) walk the table in channel-block-major K order (upload_table)
static void set_korder(IgemmArgs& a, const Table& t) {
    if (t.xsoff_p) { a.xsoff = t.xsoff_p; a.rtap = t.rtap_p; a.aoff = t.aoff_p; }
}
static void set_vec4(IgemmArgs& a) {
    a.vec4 = a.so_w == 1 && a.OWc % 4 == 0 && a.o_w0 % 4 == 0 && a.yW % 4 == 0 && a.yHW % 4 == 0 && a.y_bs % 4 == 0 &&
             a.y_gs % 4 == 0 && aligned16(a.Y) && (!a.addend || (aligned16(a.addend) && a.add_gs % 4 == 0));
}

// ------------------------------------------------------------------------------------
// 16-bit data path of the bf16 mode: per-stream workspace for the packed operands, launch set-up
// ------------------------------------------------------------------------------------
// Packed operands live in a workspace owned by the stream the convolution is issued on (launches of one stream are
// ordered, so consecutive convolutions may reuse it; concurrent chains of the step run on different streams).  It grows
// on demand (hipFree synchronises the device, so no launch still reads the old block) and is never shrunk.
struct PackSpace { char* p = nullptr; size_t cap = 0; };
static std::mutex g_ws_mu;
static std::map<hipStream_t, PackSpace> g_ws;
// Captured step graphs hold workspace addresses: every stream carries a generation, bumped whenever ITS block is freed or
// moved, and a graph's cache key embeds the generations of the streams its plan launches on -- so destroying an unrelated plan
// (CycleGAN.test evicts inference plans all the time) does not force the training plan to re-capture.
static std::map<hipStream_t, long> g_ws_gen;
static long g_ws_gen_all = 0;                     // vc_release_workspaces: every block went away
long ws_generation(std::initializer_list<hipStream_t> streams) {
    std::lock_guard<std::mutex> l(g_ws_mu);
    long sum = g_ws_gen_all;
    for (hipStream_t s : streams) { auto it = g_ws_gen.find(s); if (it != g_ws_gen.end()) sum += it->second; }
    return sum;
}
static int pack_space(hipStream_t s, size_t bytes, char** out) {
    std::lock_guard<std::mutex> l(g_ws_mu);
    PackSpace& w = g_ws[s];
    if (w.cap < bytes) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            vc_set_error("bf16 data path: the packed-operand workspace would have to grow during stream capture; run one eager step of this batch first");
            return VC_E_STATE;
        }
        if (w.p) { VC_HIP(hipFree(w.p)); ++g_ws_gen[s]; }
        w.p = nullptr; w.cap = 0;
        const size_t cap = bytes + bytes / 4 + (1u << 20);
        VC_HIP(hipMalloc((void**)&w.p, cap));
        w.cap = cap;
    }
    *out = w.p;
    return 0;
}
// gives the packed-operand workspaces of every stream back to the driver (they regrow on the next bf16 convolution)
extern "C" int vc_release_workspaces(void) {
    VC_HIP(hipDeviceSynchronize());              // nothing in flight may still read them
    std::lock_guard<std::mutex> l(g_ws_mu);
    for (auto& kv : g_ws) if (kv.second.p) (void)hipFree(kv.second.p);
    g_ws.clear();
    ++g_ws_gen_all;
    return 0;
}
// a plan's streams go away with it: free their workspaces (a recycled stream handle must not inherit a stale block)
void drop_pack_spaces(std::initializer_list<hipStream_t> streams) {
    std::lock_guard<std::mutex> l(g_ws_mu);
    for (hipStream_t s : streams) {
        auto it = g_ws.find(s);
        if (it == g_ws.end()) continue;
        if (it->second.p) (void)hipFree(it->second.p);      // hipFree waits for the device
        g_ws.erase(it);
        g_ws_gen.erase(s);                        // the stream is going away with its plan: nobody else's key mentions it
    }
}
// testing: fill the packed-operand workspaces of these streams with a byte pattern (0xff: NaN as fp32 and as bf16), so a launch that reads a
// unit its own producer did not write in THIS step shows up as NaN instead of as last step's (identical-looking) values
int poison_pack_spaces(std::initializer_list<hipStream_t> streams, int byte) {
    VC_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> l(g_ws_mu);
    for (hipStream_t s : streams) {
        auto it = g_ws.find(s);
        if (it != g_ws.end() && it->second.p) VC_HIP(hipMemset(it->second.p, byte, it->second.cap));
    }
    return 0;
}
// the path serves launches that fill its 128 x 128 tiles; everything else stays on the table-driven fp32-storage kernels
static bool bf16_path(const Table& t, int math, int M, long Ptot) { return math == VC_MATH_BF16 && t.koct > 0 && M >= 64 && Ptot >= 64; }
// the forward of this geometry at batch B runs on the 16-bit data path (the decision conv_fwd takes)
bool conv_fwd_packs_input(const Geom& g, int B, int math) {
    if (math != VC_MATH_BF16 || c1_path(g) || wsconv_path(g)) return false;
    Table t; if (get_fwd_table(g, &t, nullptr)) return false;
    return bf16_path(t, math, g.Cout, (long)B * g.OH * g.OW);
}
// the decision conv_dgrad takes: every stride-parity class of the input gradient has an octet table and the launch fills the tiles
bool conv_dgrad_packs_input(const Geom& g, int B, int math) {
    if (math != VC_MATH_BF16 || c1_path(g) || g.Cin % 4) return false;
    Table t0; if (get_dgrad_table(g, 0, 0, &t0)) return false;
    if (!bf16_path(t0, math, g.Cin, (long)B * ((g.H + g.sh - 1) / g.sh) * ((g.W + g.sw - 1) / g.sw))) return false;
    for (int ph = 0; ph < g.sh; ++ph) for (int pw = 0; pw < g.sw; ++pw) {
        Table t; if (get_dgrad_table(g, ph, pw, &t) || t.koct <= 0) return false;
    }
    return true;
}
static void bf16_set_vec4(ConvBf16Args& a) {
    a.vec4 = a.so_w == 1 && a.OWc % 4 == 0 && a.o_w0 % 4 == 0 && a.yW % 4 == 0 && a.yHW % 4 == 0 && a.y_bs % 4 == 0 &&
             a.y_gs % 4 == 0 && aligned16(a.Y) && (!a.addend || (aligned16(a.addend) && a.add_gs % 4 == 0));
}

// ------------------------------------------------------------------------------------
// Winograd F(4,3) route (csrc/vc_conv_wino.hip): forward and input gradient of the 3-tap stride-1 1-D convolutions in fp32 mode
// ------------------------------------------------------------------------------------
// B: the batch of the launch -- each of the six GEMMs sees B W / 4 pixels, and below one 64-pixel tile the direct kernel wins (measured at
// batch 1: 3.4 -> 5.1 ms per step with the route forced on; a threshold of 128 pixels instead of 64 measured the same at batch 8 and 16)
bool wino_eligible(const Geom& g, int math, int B) {
    return math == VC_MATH_FP32 && vc_hooks().winograd != 0 && g.kh == 1 && (g.kw == 3 || g.kw == 5) && g.sh == 1 && g.sw == 1 && g.H == 1 &&
           g.W % 4 == 0 && g.W >= 8 && g.Cin % 32 == 0 && g.Cout % 32 == 0 && g.Cin >= 64 && g.Cout >= 64 && (long)B * (g.W / 4) >= 64;
}
// the filter gradient of the same layers: the fused Winograd-domain kernel (vc_conv_wino.hip: wino_wgrad_kernel) wants whole k octets
// (W / 4 a multiple of 8) and 64 x 64 channel tiles; Bmax = the largest application of the launch
static bool wino_wgrad_eligible(const Geom& g, int math, const WgradApp* apps, int napps) {
    int Bmax = 0;
    for (int k = 0; k < napps; ++k) { Bmax = std::max(Bmax, apps[k].B); if ((apps[k].B * (g.W / 32)) % 4 != 0) return false; }   // the kernel's prefetch depth
    return wino_eligible(g, math, Bmax) && (g.W / 4) % 8 == 0 && g.Cin % 64 == 0 && g.Cout % 64 == 0;
}
// (row, row tap) pairs of a 6-tap SAME convolution along H that do not fall into the padding: what the direct kernel multiplies per column tap
static double winoh_valid_pairs(const Geom& g) {
    int n = 0;
    for (int i = 0; i < g.OH; ++i) for (int d = 0; d < g.kh; ++d) { const int r = i + d - g.pt; n += (r >= 0 && r < g.H) ? 1 : 0; }
    return (double)n;
}
// F(3,6) along H for the discriminator's last block (kernel 6 x kw, stride 1 along H; csrc/vc_conv_wino.hip: winoh_kernel): forward and input
// gradient as eight (1 x kw, stride 1 x sw) convolutions of transformed row planes.  B: each plane's GEMM sees B (H / 3) OW pixels.
bool winoh_eligible(const Geom& g, int math, int B) {
    return math == VC_MATH_FP32 && vc_hooks().winograd != 0 && g.kh == 6 && g.sh == 1 && g.H % 3 == 0 && g.H >= 6 && g.OH == g.H && g.pt == 2 &&
           g.W % 4 == 0 && g.OW % 4 == 0 && g.Cin % 32 == 0 && g.Cout % 32 == 0 && g.Cin >= 64 && g.Cout >= 64 && (long)B * (g.H / 3) * g.OW >= 64;
}
bool wino_route(const Geom& g, int math, int B) { return wino_eligible(g, math, B) || winoh_eligible(g, math, B); }
static Geom winoh_gemm_geom(const Geom& g) { return make_geom(g.Cin, g.H / 3, g.W, g.Cout, 1, g.kw, 1, g.sw); }
static int winoh_fwd(const float* x, long x_gs, const float* w, long w_gs, const float* bias, long b_gs, float* y, long y_gs,
                     int G, int B, const Geom& g, int math, hipStream_t s, const float* U) {
    const int nt = g.H / 3; const Geom g1 = winoh_gemm_geom(g);
    const long inner = (long)g.kw * g.Cin * g.Cout;
    const size_t nV = (size_t)8 * G * B * g.Cin * nt * g.W, nM = (size_t)8 * G * B * g.Cout * nt * g.OW, nU = U ? 0 : (size_t)8 * G * inner;
    char* wsp; VC_CHECK(pack_space(s, (nV + nM + nU) * sizeof(float), &wsp));
    float* V = (float*)wsp; float* M = V + nV; float* Ub = M + nM;
    if (!U) { VC_CHECK(launch_wino_filters({WinoFilter{w, Ub, w_gs, g.kw * g.Cin, g.Cout, 6}}, G, s)); U = Ub; }
    VC_CHECK(launch_winoh(0, x, x_gs, V, 0, nullptr, 0, nullptr, 0, 0, G, B, g.Cin, nt, g.W, s));
    {
        DetScope no_split(1);
        vc_count_wino_gemm(0, 2.0 * G * B * g.OW * (double)g.Cout * g.Cin * g.kw * (winoh_valid_pairs(g) - 8.0 * nt));
        VC_CHECK(conv_fwd(V, (long)B * g.Cin * nt * g.W, U, inner, nullptr, 0, M, (long)B * g.Cout * nt * g.OW, 8 * G, B, g1, math, s));
    }
    return launch_winoh(1, M, 0, y, y_gs, bias, b_gs, nullptr, 0, 0, G, B, g.Cout, nt, g.OW, s);
}
static int winoh_dgrad(const float* dy, long dy_gs, const float* w, long w_gs, float* dx, long dx_gs, const float* addend, long add_gs,
                       int accumulate, int G, int B, const Geom& g, int math, hipStream_t s, const float* U) {
    const int nt = g.H / 3; const Geom g1 = winoh_gemm_geom(g);
    const long inner = (long)g.kw * g.Cin * g.Cout;
    const size_t nM = (size_t)8 * G * B * g.Cout * nt * g.OW, nV = (size_t)8 * G * B * g.Cin * nt * g.W, nU = U ? 0 : (size_t)8 * G * inner;
    char* wsp; VC_CHECK(pack_space(s, (nV + nM + nU) * sizeof(float), &wsp));
    float* dM = (float*)wsp; float* dV = dM + nM; float* Ub = dV + nV;
    if (!U) { VC_CHECK(launch_wino_filters({WinoFilter{w, Ub, w_gs, g.kw * g.Cin, g.Cout, 6}}, G, s)); U = Ub; }
    VC_CHECK(launch_winoh(2, dy, dy_gs, dM, 0, nullptr, 0, nullptr, 0, 0, G, B, g.Cout, nt, g.OW, s));
    {
        DetScope no_split(1);
        vc_count_wino_gemm(0, 2.0 * G * B * g.OW * (double)g.Cout * g.Cin * g.kw * (winoh_valid_pairs(g) - 8.0 * nt));
        VC_CHECK(conv_dgrad(dM, (long)B * g.Cout * nt * g.OW, U, inner, dV, (long)B * g.Cin * nt * g.W, nullptr, 0, 0, 8 * G, B, g1, math, s));
    }
    return launch_winoh(3, dV, 0, dx, dx_gs, nullptr, 0, addend, add_gs, accumulate, G, B, g.Cin, nt, g.W, s);
}
// transform-domain scratch (floats) one forward / input-gradient call of a routed layer takes from its stream's block when the caller supplies the filters
size_t wino_scratch_floats(const Geom& g, int G, int B) {
    if (g.kh == 6) { const int nt = g.H / 3; return (size_t)8 * G * B * ((size_t)g.Cin * nt * g.W + (size_t)g.Cout * nt * g.OW); }
    return (size_t)(g.kw + 3) * G * B * ((size_t)g.Cin + g.Cout) * (g.W / 4);
}
int wino_workspace(hipStream_t s, size_t floats, float** out) {
    char* q; VC_CHECK(pack_space(s, floats * sizeof(float), &q));
    *out = (float*)q;
    return 0;
}
// the six channel GEMMs of one layer: M[p] = U[p] V[p] (forward) / dV[p] = U[p]^T dM[p] (input gradient), p = transform position, as a 1 x 1
// convolution with 6 G groups.  K is split inside the workgroup only: the launches have 6 x the groups of the direct form to fill the chip with,
// and split-K across workgroups would cost a memset of the transform-domain destination per launch.
int wino_gemm_fwd(const float* V, const float* U, float* M, int G, int B, const Geom& g, int math, hipStream_t s) {
    const int nt = g.W / 4, P = g.kw + 3; const Geom g1 = wino_gemm_geom(g);
    DetScope no_split(1);
    vc_count_wino_gemm(0, 2.0 * G * B * g.W * (double)g.Cout * g.Cin * (g.kw - 0.25 * P));      // direct: kw taps x W pixels; here: kw + 3 positions x W / 4
    return conv_fwd(V, (long)B * g.Cin * nt, U, (long)g.Cin * g.Cout, nullptr, 0, M, (long)B * g.Cout * nt, P * G, B, g1, math, s);
}
int wino_gemm_dgrad(const float* dM, const float* U, float* dV, int G, int B, const Geom& g, int math, hipStream_t s) {
    const int nt = g.W / 4, P = g.kw + 3; const Geom g1 = wino_gemm_geom(g);
    DetScope no_split(1);
    vc_count_wino_gemm(0, 2.0 * G * B * g.W * (double)g.Cout * g.Cin * (g.kw - 0.25 * P));
    return conv_dgrad(dM, (long)B * g.Cout * nt, U, (long)g.Cin * g.Cout, dV, (long)B * g.Cin * nt, nullptr, 0, 0, P * G, B, g1, math, s);
}
Geom wino_gemm_geom(const Geom& g) { return make_geom(g.Cin, 1, g.W / 4, g.Cout, 1, 1, 1, 1); }
// U: the layer's transformed filters [G][6][Cin][Cout] when the caller holds them (inside a train step: built once per step), else built here
static int wino_fwd(const float* x, long x_gs, const float* w, long w_gs, const float* bias, long b_gs, float* y, long y_gs,
                    int G, int B, const Geom& g, int math, hipStream_t s, const float* U) {
    const int nt = g.W / 4, P = g.kw + 3;
    const size_t nV = (size_t)P * G * B * g.Cin * nt, nM = (size_t)P * G * B * g.Cout * nt, nU = U ? 0 : (size_t)P * G * g.Cin * g.Cout;
    char* wsp; VC_CHECK(pack_space(s, (nV + nM + nU) * sizeof(float), &wsp));
    float* V = (float*)wsp; float* M = V + nV; float* Ub = M + nM;
    if (!U) { VC_CHECK(launch_wino_filters({WinoFilter{w, Ub, w_gs, g.Cin, g.Cout, g.kw}}, G, s)); U = Ub; }
    VC_CHECK(launch_wino_input(x, x_gs, V, G, B, g.Cin, nt, s, g.kw));
    VC_CHECK(wino_gemm_fwd(V, U, M, G, B, g, math, s));
    return launch_wino_output(M, bias, b_gs, y, y_gs, G, B, g.Cout, nt, s, g.kw);
}
static int wino_dgrad(const float* dy, long dy_gs, const float* w, long w_gs, float* dx, long dx_gs, const float* addend, long add_gs,
                      int accumulate, int G, int B, const Geom& g, int math, hipStream_t s, const float* U) {
    const int nt = g.W / 4, P = g.kw + 3;
    const size_t nM = (size_t)P * G * B * g.Cout * nt, nV = (size_t)P * G * B * g.Cin * nt, nU = U ? 0 : (size_t)P * G * g.Cin * g.Cout;
    char* wsp; VC_CHECK(pack_space(s, (nV + nM + nU) * sizeof(float), &wsp));
    float* dM = (float*)wsp; float* dV = dM + nM; float* Ub = dV + nV;
    if (!U) { VC_CHECK(launch_wino_filters({WinoFilter{w, Ub, w_gs, g.Cin, g.Cout, g.kw}}, G, s)); U = Ub; }
    VC_CHECK(launch_wino_dy(dy, dy_gs, dM, G, B, g.Cout, nt, s, g.kw));
    VC_CHECK(wino_gemm_dgrad(dM, U, dV, G, B, g, math, s));
    return launch_wino_dx(dV, dx, dx_gs, addend, add_gs, accumulate, G, B, g.Cin, nt, s, g.kw);
}

// query != nullptr: do not launch, only report whether the launch would use split-K atomics (needs a zeroed y)
int conv_fwd(const float* x, long x_gs, const float* w, long w_gs, const float* bias, long b_gs,
                    float* y, long y_gs, int G, int B, const Geom& g, int math, hipStream_t s,
                    bool prezeroed, bool* query, const char* packed_w, char* xp_keep, long xp_keep_gs, FuseNorm* fuse, bool x_prepacked) {
    // fuse: the caller would like instance norm + GLU done in the epilogue; fuse->done tells whether this launch did it
    // xp_keep: where the packed bf16 copy of x is left for the layer's filter gradient (16-bit data path; group stride in bytes)
    VC_CHECK(check_conv_geom(g, B));
    // Winograd routes, fp32: taken on the caller's transformed filters (inside a train step: built once per step, used four times).  A call
    // without them (operator / network level, inference) would transform the filters itself -- measured slower than the direct kernels for a
    // forward-only pass (B = 8 x 128 frames 0.56 vs 0.46 ms, one 2048-frame utterance 0.68 vs 0.60) -- and takes the route only when a test
    // forces it (vc_tune_winograd(2)).
    const bool wino_ok = packed_w != nullptr || vc_hooks().winograd == 2;
    if (wino_ok && wino_eligible(g, math, B)) {      // residual / up-sampling convolutions: taps + 3 GEMMs in the transform domain; every output is stored once
        if (query) { *query = false; return 0; }
        return wino_fwd(x, x_gs, w, w_gs, bias, b_gs, y, y_gs, G, B, g, math, s, (const float*)packed_w);
    }
    if (wino_ok && winoh_eligible(g, math, B)) {     // D.d3: F(3,6) along H
        if (query) { *query = false; return 0; }
        return winoh_fwd(x, x_gs, w, w_gs, bias, b_gs, y, y_gs, G, B, g, math, s, (const float*)packed_w);
    }
    if (c1_path(g)) {      // Cin = 1 (D.h1): vector-ALU kernel, plain stores, never needs a zeroed destination
        if (query) { *query = false; return 0; }
        C1Args c = c1_args(g, G, B, math); c.x = x; c.x_gs = x_gs; c.w = w; c.w_gs = w_gs; c.bias = bias; c.b_gs = b_gs; c.y = y; c.y_gs = y_gs;
        return launch_c1_fwd(c, s);
    }
    if (wsconv_path(g)) {  // few input channels, 1-D, stride 1 (G.h1): weight-stationary kernel, plain stores
        if (query) { *query = false; return 0; }
        WsConvArgs c{}; c.x = x; c.w = w; c.bias = bias; c.y = y; c.x_gs = x_gs; c.w_gs = w_gs; c.b_gs = b_gs; c.y_gs = y_gs;
        c.B = B; c.T = g.W; c.Cout = g.Cout; c.G = G;
        const int r = launch_wsconv1d(c, g.Cin, g.kw, math, s, 2.0 * G * B * g.W * (double)g.Cout * g.kw * g.Cin);
        if (r != -2) return r;
    }
    Table t; VC_CHECK(get_fwd_table(g, &t));
    if (bf16_path(t, math, g.Cout, (long)B * g.OH * g.OW)) {
        ConvBf16Args a{};
        a.Y = y; a.bias = bias; a.addend = nullptr;
        a.xsoff = t.oxs; a.otap = t.otap; a.taps = t.taps; a.ntaps = t.ntaps; a.xbias = t.oxbias;
        const size_t xp_g = align256((size_t)B * g.Cin * g.H * g.W * 2), ap_g = align256((size_t)g.Cout * t.koct * 16);
        a.x_bytes = (unsigned)((long)B * g.Cin * g.H * g.W * 2 + t.oxbias); a.a_bytes = (unsigned)((long)g.Cout * t.koct * 16);
        a.a_gs = (long)ap_g; a.x_gs = (long)xp_g; a.x_bs = (long)g.Cin * g.H * g.W * 2;
        a.y_gs = y_gs; a.bias_gs = b_gs; a.add_gs = 0; a.y_bs = (long)g.Cout * g.OH * g.OW;
        a.koct = t.koct; a.M = g.Cout; a.Hl = g.H; a.Wl = g.W; a.xW = g.W; a.si = g.sh; a.sj = g.sw;
        a.NPB = g.OH * g.OW; a.OWc = g.OW; a.Ptot = B * a.NPB; a.inv_npb = 1.0f / (float)a.NPB; a.inv_ow = 1.0f / (float)g.OW;
        a.yHW = g.OH * g.OW; a.yW = g.OW; a.so_h = 1; a.so_w = 1; a.o_h0 = 0; a.o_w0 = 0;
        a.accumulate = 0; a.groups = G;
        bf16_set_vec4(a);
        conv_bf16_config(a);
        if (query) { *query = a.splitk > 1; return 0; }
        char* wsp = nullptr;
        const size_t need = (xp_keep ? 0 : xp_g * G) + (packed_w ? 0 : ap_g * G);
        if (need) VC_CHECK(pack_space(s, need, &wsp));
        char* xp = xp_keep ? xp_keep : wsp;
        if (xp_keep) a.x_gs = xp_keep_gs;
        a.X = xp; a.A = packed_w ? packed_w : wsp + (xp_keep ? 0 : xp_g * G);
        if (!(x_prepacked && xp_keep)) VC_CHECK(launch_pack_act_bf16(x, x_gs, xp, a.x_gs, G, B, g.Cin, g.H * g.W, s));
        if (!packed_w) VC_CHECK(launch_pack_filter_bf16(w, w_gs, const_cast<void*>(a.A), (long)ap_g, t.wsrc, 1, G, g.Cout, t.koct * 8, s));
        if (a.splitk > 1) {
            if (!prezeroed)
                for (int gi = 0; gi < G; ++gi) VC_HIP(hipMemsetAsync(y + (long)gi * y_gs, 0, (size_t)B * a.y_bs * sizeof(float), s));
            a.accumulate = 1;
        }
        return launch_conv_bf16(a, s, 2.0 * G * B * a.NPB * (double)g.Cout * t.valid);
    }
    IgemmArgs a{};
    a.A = w; a.X = x; a.Y = y; a.bias = bias; a.addend = nullptr;
    a.xsoff = t.xsoff; a.rtap = t.rtap; a.aoff = t.aoff; a.taps = t.taps; a.ntaps = t.ntaps; a.xbias = t.xbias;
    // buffer sizes for the hardware range check: measured on gfx950, the check covers VGPR offset + SGPR offset, so the
    // gathered tensor's size includes the table bias that rides in every SGPR offset
    a.x_bytes = (unsigned)((long)B * g.Cin * g.H * g.W * 4 + t.xbias); a.a_bytes = (unsigned)std::min<long>((long)t.valid * g.Cout * 4, 0xffffff00L);
    a.a_gs = w_gs; a.x_gs = x_gs; a.y_gs = y_gs; a.bias_gs = b_gs; a.add_gs = 0;
    a.x_bs = (long)g.Cin * g.H * g.W; a.y_bs = (long)g.Cout * g.OH * g.OW;
    a.krows = t.rows; a.M = g.Cout; a.a_ms = 1;
    a.Hl = g.H; a.Wl = g.W; a.xHW = g.H * g.W; a.xW = g.W; a.si = g.sh; a.sj = g.sw;
    fill_pixel_grid(a, B, g.OH * g.OW, g.OW);
    a.yHW = g.OH * g.OW; a.yW = g.OW; a.so_h = 1; a.so_w = 1; a.o_h0 = 0; a.o_w0 = 0;
    a.accumulate = 0; a.groups = G;
    set_vec4(a);
    set_korder(a, t);
    igemm_config(a, math);
    if (query) { *query = a.splitk > 1; return 0; }
    if (fuse && a.bm == 64 && a.splitk == 1 && a.vec4 && g.H == 1 && g.kh == 1 && (a.NPB == 32 || a.NPB == 64) && g.Cout % 64 == 0) {
        a.fuse = 1; a.gamma = fuse->gamma; a.beta = fuse->beta; a.par_gs = fuse->par_gs; a.stats = fuse->stats; a.st_gs = fuse->st_gs;
        a.act = fuse->act; a.act_gs = fuse->act_gs; fuse->done = true;
    }
    if (a.splitk > 1) {          // atomics need a zeroed destination
        if (!prezeroed)
            for (int gi = 0; gi < G; ++gi) VC_HIP(hipMemsetAsync(y + (long)gi * y_gs, 0, (size_t)B * a.y_bs * sizeof(float), s));
        a.accumulate = 1;
    }
    return launch_igemm(a, false, math, s, 2.0 * G * B * a.NPB * (double)g.Cout * t.valid);
}

int conv_dgrad(const float* dy, long dy_gs, const float* w, long w_gs, float* dx, long dx_gs,
                      const float* addend, long add_gs, int accumulate, int G, int B, const Geom& g, int math, hipStream_t s,
                      bool prezeroed, bool* query, const char* const* packed_w, const char* dy_packed, long dy_packed_gs) {
    VC_CHECK(check_conv_geom(g, B));
    const bool wino_ok = packed_w != nullptr || vc_hooks().winograd == 2;      // see conv_fwd
    if (wino_ok && wino_eligible(g, math, B)) {
        if (query) { *query = false; return 0; }      // the back transform stores (or accumulates into) every element of dx itself
        VC_ARG(dy, "conv input gradient (Winograd route): the fp32 dy was left unwritten");
        return wino_dgrad(dy, dy_gs, w, w_gs, dx, dx_gs, addend, add_gs, accumulate, G, B, g, math, s, packed_w ? (const float*)packed_w[0] : nullptr);
    }
    if (wino_ok && winoh_eligible(g, math, B)) {
        if (query) { *query = false; return 0; }
        VC_ARG(dy, "conv input gradient (Winograd route): the fp32 dy was left unwritten");
        return winoh_dgrad(dy, dy_gs, w, w_gs, dx, dx_gs, addend, add_gs, accumulate, G, B, g, math, s, packed_w ? (const float*)packed_w[0] : nullptr);
    }
    if (c1_path(g)) {
        if (query) { *query = false; return 0; }      // zeroes dx itself when it has to
        C1Args c = c1_args(g, G, B, math); c.dy = dy; c.y_gs = dy_gs; c.w = w; c.w_gs = w_gs; c.dx = dx; c.dx_gs = dx_gs;
        const long n = (long)B * g.H * g.W;
        if (addend)           // the kernel scatters with atomics: seed dx with the addend first
            for (int gi = 0; gi < G; ++gi) {
                if (accumulate) VC_CHECK(launch_add(dx + (long)gi * dx_gs, addend + (long)gi * add_gs, n, s));
                else VC_HIP(hipMemcpyAsync(dx + (long)gi * dx_gs, addend + (long)gi * add_gs, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
        return launch_c1_dgrad(c, accumulate || prezeroed || addend != nullptr, s);
    }
    VC_ARG(g.Cin % 4 == 0, "conv input gradient: Cin must be a multiple of 4");
    // one launch per stride-parity class (ph, pw): inside a class the transposed conv is a dense stride-1 gather
    {
        Table t0; VC_CHECK(get_dgrad_table(g, 0, 0, &t0));
        bool native = bf16_path(t0, math, g.Cin, (long)B * ((g.H + g.sh - 1) / g.sh) * ((g.W + g.sw - 1) / g.sw));
        for (int ph = 0; ph < g.sh && native; ++ph) for (int pw = 0; pw < g.sw && native; ++pw) {   // a class without taps: table-driven path
            Table t; VC_CHECK(get_dgrad_table(g, ph, pw, &t));
            native = t.koct > 0;
        }
        if (native) {
            // 16-bit data path: dY is packed once, every class has its own filter image (its taps only)
            ConvBf16Args cl[4]; double flb[4]; const int* wsrc[4]; int nc = 0; bool split = false;
            const size_t xp_g = align256((size_t)B * g.Cout * g.OH * g.OW * 2);
            size_t ap_off[4], ap_tot = 0;
            for (int ph = 0; ph < g.sh; ++ph) for (int pw = 0; pw < g.sw; ++pw) {
                const int Hc = (g.H - ph + g.sh - 1) / g.sh, Wc = (g.W - pw + g.sw - 1) / g.sw;
                // prebuilt images (refresh_filter_images) are indexed by (ph, pw) without gaps: an empty class would shift them
                VC_ARG(!(packed_w && (Hc <= 0 || Wc <= 0)), "conv input gradient: an empty stride-parity class with prebuilt filter images");
                if (Hc <= 0 || Wc <= 0) continue;
                VC_ARG(nc < 4, "conv input gradient: more than 4 stride-parity classes");
                Table t; VC_CHECK(get_dgrad_table(g, ph, pw, &t));
                VC_ARG(t.koct > 0, "conv input gradient: inconsistent octet tables");
                ConvBf16Args a{};
                a.Y = dx; a.bias = nullptr; a.addend = addend;
                a.xsoff = t.oxs; a.otap = t.otap; a.taps = t.taps; a.ntaps = t.ntaps; a.xbias = t.oxbias;
                a.x_bytes = (unsigned)((long)B * g.Cout * g.OH * g.OW * 2 + t.oxbias); a.a_bytes = (unsigned)((long)g.Cin * t.koct * 16);
                a.a_gs = (long)align256((size_t)g.Cin * t.koct * 16); a.x_gs = (long)xp_g; a.x_bs = (long)g.Cout * g.OH * g.OW * 2;
                a.y_gs = dx_gs; a.bias_gs = 0; a.add_gs = add_gs; a.y_bs = (long)g.Cin * g.H * g.W;
                a.koct = t.koct; a.M = g.Cin; a.Hl = g.OH; a.Wl = g.OW; a.xW = g.OW; a.si = 1; a.sj = 1;
                a.NPB = Hc * Wc; a.OWc = Wc; a.Ptot = B * a.NPB; a.inv_npb = 1.0f / (float)a.NPB; a.inv_ow = 1.0f / (float)Wc;
                a.yHW = g.H * g.W; a.yW = g.W; a.so_h = g.sh; a.so_w = g.sw; a.o_h0 = ph; a.o_w0 = pw;
                a.accumulate = accumulate; a.groups = G;
                bf16_set_vec4(a);
                conv_bf16_config(a);
                split = split || a.splitk > 1;
                ap_off[nc] = ap_tot; ap_tot += (size_t)a.a_gs * G;
                wsrc[nc] = t.wsrc; flb[nc] = 2.0 * G * B * a.NPB * (double)g.Cin * t.valid; cl[nc++] = a;
            }
            if (query) { *query = split; return 0; }
            char* wsp; VC_CHECK(pack_space(s, xp_g * G + (packed_w ? 0 : ap_tot), &wsp));
            if (!dy_packed) VC_CHECK(launch_pack_act_bf16(dy, dy_gs, wsp, (long)xp_g, G, B, g.Cout, g.OH * g.OW, s));
            if (split && !accumulate) {
                if (!prezeroed)
                    for (int gi = 0; gi < G; ++gi)
                        VC_HIP(hipMemsetAsync(dx + (long)gi * dx_gs, 0, (size_t)B * g.Cin * g.H * g.W * sizeof(float), s));
                for (int i = 0; i < nc; ++i) cl[i].accumulate = 1;
            }
            for (int i = 0; i < nc; ++i) {
                cl[i].X = dy_packed ? dy_packed : wsp; if (dy_packed) cl[i].x_gs = dy_packed_gs;
                cl[i].A = packed_w ? packed_w[i] : wsp + xp_g * G + ap_off[i];
                if (!packed_w) VC_CHECK(launch_pack_filter_bf16(w, w_gs, wsp + xp_g * G + ap_off[i], cl[i].a_gs, wsrc[i], g.Cout, G, g.Cin, cl[i].koct * 8, s));
                VC_CHECK(launch_conv_bf16(cl[i], s, flb[i]));
            }
            return 0;
        }
    }
    VC_ARG(query || dy, "conv input gradient: the fp32 dy was left unwritten but this launch does not run on the 16-bit data path");
    IgemmArgs cls[4]; double fl[4]; int ncls = 0;
    bool any_split = false;
    for (int ph = 0; ph < g.sh; ++ph) for (int pw = 0; pw < g.sw; ++pw) {
        const int Hc = (g.H - ph + g.sh - 1) / g.sh, Wc = (g.W - pw + g.sw - 1) / g.sw;
        if (Hc <= 0 || Wc <= 0) continue;
        VC_ARG(ncls < 4, "conv input gradient: more than 4 stride-parity classes");
        Table t; VC_CHECK(get_dgrad_table(g, ph, pw, &t));
        IgemmArgs a{};
        a.A = w; a.X = dy; a.Y = dx; a.bias = nullptr; a.addend = addend;
        a.xsoff = t.xsoff; a.rtap = t.rtap; a.aoff = t.aoff; a.taps = t.taps; a.ntaps = t.ntaps; a.xbias = t.xbias;
        a.x_bytes = (unsigned)((long)B * g.Cout * g.OH * g.OW * 4 + t.xbias);
        a.a_bytes = (unsigned)((long)g.kh * g.kw * g.Cin * g.Cout * 4);   // whole tensor: a 16-byte run past its end reads 0
        a.a_gs = w_gs; a.x_gs = dy_gs; a.y_gs = dx_gs; a.bias_gs = 0; a.add_gs = add_gs;
        a.x_bs = (long)g.Cout * g.OH * g.OW; a.y_bs = (long)g.Cin * g.H * g.W;
        a.krows = t.rows; a.M = g.Cin; a.a_ms = g.Cout;
        a.Hl = g.OH; a.Wl = g.OW; a.xHW = g.OH * g.OW; a.xW = g.OW; a.si = 1; a.sj = 1;
        fill_pixel_grid(a, B, Hc * Wc, Wc);
        a.yHW = g.H * g.W; a.yW = g.W; a.so_h = g.sh; a.so_w = g.sw; a.o_h0 = ph; a.o_w0 = pw;
        a.accumulate = accumulate; a.groups = G;
        set_vec4(a);
        set_korder(a, t);
        igemm_config(a, math);
        any_split = any_split || a.splitk > 1;
        fl[ncls] = 2.0 * G * B * a.NPB * (double)g.Cin * t.valid; cls[ncls++] = a;
    }
    if (query) { *query = any_split; return 0; }
    // the classes interleave inside dx: if ANY of them needs atomics, zero dx once up front and let all accumulate
    if (any_split && !accumulate) {
        if (!prezeroed)
            for (int gi = 0; gi < G; ++gi)
                VC_HIP(hipMemsetAsync(dx + (long)gi * dx_gs, 0, (size_t)B * g.Cin * g.H * g.W * sizeof(float), s));
        for (int i = 0; i < ncls; ++i) cls[i].accumulate = 1;
    }
    // short-K parity classes go out as ONE launch (each alone underfills the chip); long-K ones (D.d3: 288 chunks) are
    // L2-heavy and measured faster back to back
    int max_chunks = 0; bool same = true;
    for (int i = 0; i < ncls; ++i) {
        max_chunks = std::max(max_chunks, cls[i].krows / VC_KC);
        same = same && cls[i].bm == cls[0].bm && cls[i].kg == cls[0].kg;
    }
    if (ncls > 1 && max_chunks <= 128 && same) {
        double fsum = 0; for (int i = 0; i < ncls; ++i) fsum += fl[i];
        return launch_igemm_multi(cls, ncls, true, math, s, fsum);
    }
    for (int i = 0; i < ncls; ++i) VC_CHECK(launch_igemm(cls[i], true, math, s, fl[i]));
    return 0;
}

// Octet table of a plain GEMM on conv_bf16_kernel (the filter gradient): K octet o is unit row o of an operand laid out
// [octet][N][8]; padding octets up to a whole chunk.  Cached per (N, octets); a few KB each.
struct GemmTab { int* xs; unsigned* ot; Tap* taps; int koct; };
static int get_gemm_table(int N, int noct, GemmTab* out) {
    static std::mutex mu; static std::map<std::pair<int, int>, GemmTab> cache;
    std::lock_guard<std::mutex> l(mu);
    auto it = cache.find({N, noct});
    if (it == cache.end()) {
        VC_ARG((long)noct * N * 16 < 0x7fffff00L, "GEMM operand too large for 32-bit byte offsets");
        std::vector<int> xs; std::vector<unsigned> ot;
        for (int o = 0; o < noct; ++o) { xs.push_back(o * N * 16); ot.push_back(1u); }
        while (xs.size() % 8) { xs.push_back(0); ot.push_back(0u); }
        GemmTab t{}; t.koct = (int)xs.size();
        for (int i = 0; i < 8; ++i) { xs.push_back(0); ot.push_back(0u); }
        std::vector<Tap> tp{Tap{0, 0}};
        VC_CHECK(to_device(xs, &t.xs)); VC_CHECK(to_device(ot, &t.ot)); VC_CHECK(to_device(tp, &t.taps));
        it = cache.emplace(std::make_pair(N, noct), t).first;
    }
    *out = it->second;
    return 0;
}

// dW (+)= sum over the applications; store: dW holds nothing yet (plain stores when the launch does not split K)
// query != nullptr: only report whether the launch would accumulate with atomics (needs a zeroed dW)
bool conv_wgrad_reads_pixel_octets(const Geom& g, long pixels, int math) {
    if (math != VC_MATH_BF16 || c1_path(g)) return false;
    Table t; if (get_fwd_table(g, &t, nullptr)) return false;
    return bf16_path(t, math, g.Cout, pixels) && (g.OH * g.OW) % 8 == 0 && g.OW % 8 == 0;
}
int conv_wgrad(const WgradApp* apps, int napps, float* dw, long dw_gs, int G, const Geom& g, int math, hipStream_t s,
                      bool store, bool* query, const WgradPacked* pq) {
    if (c1_path(g)) {
        if (query) { *query = true; return 0; }
        for (int k = 0; k < napps; ++k) {
            VC_CHECK(check_conv_geom(g, apps[k].B));
            VC_ARG(apps[k].dy, "conv filter gradient (Cin = 1): an application's fp32 dy was left unwritten");
            C1Args c = c1_args(g, G, apps[k].B, math); c.x = apps[k].x; c.x_gs = apps[k].x_gs; c.dy = apps[k].dy; c.y_gs = apps[k].dy_gs;
            c.dw = dw; c.dw_gs = dw_gs;
            VC_CHECK(launch_c1_wgrad(c, s));
        }
        return 0;
    }
    VC_ARG(napps >= 1 && napps <= 2, "conv filter gradient: 1 or 2 applications per launch");
    if (wino_wgrad_eligible(g, math, apps, napps)) {
        // residual-block convolutions, fp32: both applications' operands go to the transform domain in one launch (V = B^T x, dM = A dy), the fused
        // kernel multiplies per transform position and folds the six results into the three taps in registers; a tile has ONE writer
        if (query) { *query = !store; return 0; }
        const int nt = g.W / 4;
        size_t nV[2] = {0, 0}, nM[2] = {0, 0}, tot = 0;
        for (int k = 0; k < napps; ++k) {
            VC_CHECK(check_conv_geom(g, apps[k].B));
            VC_ARG(apps[k].dy && apps[k].x, "conv filter gradient (Winograd route): an application's fp32 operands are missing");
            nV[k] = (size_t)(g.kw + 3) * G * apps[k].B * g.Cin * nt; nM[k] = (size_t)(g.kw + 3) * G * apps[k].B * g.Cout * nt; tot += nV[k] + nM[k];
        }
        float* wsp; VC_CHECK(wino_workspace(s, tot, &wsp));
        WinoPrepJob jobs[4]; const float* V[2]; const float* dM[2]; int Bs[2]; int nj = 0; double flops = 0;
        for (int k = 0; k < napps; ++k) {
            float* v = wsp; wsp += nV[k]; float* m = wsp; wsp += nM[k];
            jobs[nj++] = WinoPrepJob{apps[k].x, apps[k].x_gs, v, apps[k].B, g.Cin, 0};
            jobs[nj++] = WinoPrepJob{apps[k].dy, apps[k].dy_gs, m, apps[k].B, g.Cout, 1};
            V[k] = v; dM[k] = m; Bs[k] = apps[k].B;
            flops += 2.0 * (g.kw + 3) * G * apps[k].B * nt * (double)g.Cin * g.Cout;
        }
        VC_CHECK(launch_wino_prep(jobs, nj, G, nt, s, g.kw));
        vc_count_wino_gemm(1, flops * (4.0 * g.kw / (g.kw + 3) - 1.0));      // the direct form multiplies kw x W against (kw + 3) x W / 4
        return launch_wino_wgrad(V, dM, Bs, napps, dw, dw_gs, store ? 1 : 0, G, g.Cin, g.Cout, nt, s, flops, g.kw);
    }
    Table t; VC_CHECK(get_fwd_table(g, &t));
    {
        long ptot = 0; for (int k = 0; k < napps; ++k) ptot += (long)apps[k].B * g.OH * g.OW;
        if (bf16_path(t, math, g.Cout, ptot) && (g.OH * g.OW) % 8 == 0 && g.OW % 8 == 0) {
            // 16-bit data path: the filter gradient as a GEMM over the pixel axis on conv_bf16_kernel itself,
            //   dW[k][co] = sum_P Xcol[k][P] * dY[co][P],
            // with both operands repacked 8 pixels to a unit (im2col of the packed activations; dY by pixel octets): rows =
            // K rows in TF's own (tap, cin) order, "pixels" = Cout, K octets = pixel octets of both applications back to back.
            const int NPB = g.OH * g.OW, K = t.valid;
            int po0[2], npo = 0;
            for (int k = 0; k < napps; ++k) { VC_CHECK(check_conv_geom(g, apps[k].B)); po0[k] = npo; npo += apps[k].B * NPB / 8; }
            GemmTab gt; VC_CHECK(get_gemm_table(g.Cout, npo, &gt));
            ConvBf16Args a{};
            a.Y = dw; a.bias = nullptr; a.addend = nullptr;
            a.xsoff = gt.xs; a.otap = gt.ot; a.taps = gt.taps; a.ntaps = 1; a.xbias = 0;
            const size_t xcol_g = align256((size_t)npo * K * 16), dyq_g = align256((size_t)npo * g.Cout * 16);
            VC_ARG(xcol_g < 0xfff00000UL && dyq_g < 0xfff00000UL, "conv filter gradient: repacked operands of one group exceed 4 GiB");
            a.x_bytes = (unsigned)((size_t)npo * g.Cout * 16); a.a_bytes = (unsigned)((size_t)npo * K * 16);
            a.a_gs = (long)xcol_g; a.x_gs = (long)dyq_g; a.x_bs = 0;
            a.y_gs = dw_gs; a.bias_gs = 0; a.add_gs = 0; a.y_bs = 0;
            a.koct = gt.koct; a.M = K; a.Hl = 1; a.Wl = g.Cout; a.xW = g.Cout; a.si = 1; a.sj = 1;
            a.NPB = g.Cout; a.OWc = g.Cout; a.Ptot = g.Cout; a.inv_npb = 1.0f / (float)g.Cout; a.inv_ow = a.inv_npb;
            a.yHW = g.Cout; a.yW = g.Cout; a.so_h = 1; a.so_w = 1; a.o_h0 = 0; a.o_w0 = 0;
            a.accumulate = store ? 0 : 1; a.groups = G;
            bf16_set_vec4(a);
            conv_bf16_config(a);
            if (query) { *query = !store || a.splitk > 1; return 0; }
            if (a.splitk > 1) a.accumulate = 1;
            size_t xp_g[2] = {0, 0}, xp_tot = 0;
            for (int k = 0; k < napps; ++k) if (!apps[k].xp) { xp_g[k] = align256((size_t)apps[k].B * g.Cin * g.H * g.W * 2); xp_tot += xp_g[k] * G; }
            // the stacked dY: the layer's own buffer when its norm backward wrote (part of) it, else this launch's pack space
            const bool stacked = pq && pq->dyq && pq->cap >= (size_t)npo * g.Cout * 16 && ((uintptr_t)pq->dyq & 15) == 0 && pq->dyq_gs % 16 == 0;
            char* wsp; VC_CHECK(pack_space(s, (xcol_g + (stacked ? 0 : dyq_g)) * G + xp_tot, &wsp));
            char* xcol = wsp; char* dyq = stacked ? pq->dyq : wsp + xcol_g * G; char* xtmp = wsp + (xcol_g + (stacked ? 0 : dyq_g)) * G;
            const long dyq_gs = stacked ? pq->dyq_gs : (long)dyq_g;
            a.x_gs = dyq_gs;
            double flops = 0;
            for (int k = 0; k < napps; ++k) {
                const char* xp = apps[k].xp; long xp_gs = apps[k].xp_gs;
                if (!xp) {
                    VC_CHECK(launch_pack_act_bf16(apps[k].x, apps[k].x_gs, xtmp, (long)xp_g[k], G, apps[k].B, g.Cin, g.H * g.W, s));
                    xp = xtmp; xp_gs = (long)xp_g[k]; xtmp += xp_g[k] * G;
                }
                VC_CHECK(launch_im2col_bf16(xp, xp_gs, xcol, (long)xcol_g, t.taps, t.ntaps, G, apps[k].B, g.Cin, g.H, g.W, g.OH, g.OW, g.sh, g.sw, po0[k], s));
                if (!(stacked && (pq->ready >> k & 1))) {
                    VC_ARG(apps[k].dy, "conv filter gradient: an application's fp32 dy was left unwritten but its pixel octets are not in the stacked buffer");
                    VC_CHECK(launch_pack_pixel_octets_bf16(apps[k].dy, apps[k].dy_gs, dyq, dyq_gs, G, apps[k].B, g.Cout, NPB, po0[k], s));
                }
                flops += 2.0 * G * apps[k].B * NPB * (double)g.Cout * t.valid;
            }
            a.A = xcol; a.X = dyq;
            return launch_conv_bf16(a, s, flops, /*prof_kind=*/1);
        }
    }
    WgradArgs a{};
    a.dW = dw; a.ktab = t.dev; a.xsoff = t.xsoff; a.xbias = t.xbias; a.dw_gs = dw_gs;
    a.x_bs = (long)g.Cin * g.H * g.W; a.dy_bs = (long)g.Cout * g.OH * g.OW;
    a.N = g.Cout; a.Hl = g.H; a.Wl = g.W; a.xHW = g.H * g.W; a.xW = g.W; a.si = g.sh; a.sj = g.sw;
    a.NPB = g.OH * g.OW; a.OWc = g.OW; a.inv_ow = 1.0f / (float)g.OW; a.inv_npb = 1.0f / (float)a.NPB;
    a.ktiles = t.rows / 64; a.groups = G; a.nseg = napps; a.nchunks = 0;
    double flops = 0;
    for (int k = 0; k < napps; ++k) {
        VC_CHECK(check_conv_geom(g, apps[k].B));
        VC_ARG(query || apps[k].dy, "conv filter gradient: an application's fp32 dy was left unwritten but the launch runs on the fp32-storage kernel");
        WgradSeg& sg = a.seg[k];
        sg.X = apps[k].x; sg.dY = apps[k].dy; sg.x_gs = apps[k].x_gs; sg.dy_gs = apps[k].dy_gs;
        sg.x_bytes = (unsigned)((long)apps[k].B * a.x_bs * 4 + t.xbias); sg.dy_bytes = (unsigned)((long)apps[k].B * a.dy_bs * 4);
        sg.Ptot = apps[k].B * a.NPB; sg.nchunks = (sg.Ptot + VC_KC - 1) / VC_KC;
        sg.RW = apps[k].B * a.OWc; sg.inv_rw = 1.0f / (float)sg.RW;
        a.nchunks += sg.nchunks;
        flops += 2.0 * G * apps[k].B * a.NPB * (double)g.Cout * t.valid;
    }
    a.store = store ? 1 : 0;
    a.rowskip = rowskip_pays(t, g.OH, g.sh, g.H) ? 1 : 0;
    if (query) { *query = !store || wgrad_splitk(a) > 1; return 0; }
    return launch_wgrad(a, math, s, flops);
}

// mode 3 (IEEE-half operands) exists for the gated conv1d + GLU forward only: every other operator refuses it instead of computing in another arithmetic
#define VC_NO_F16(name) VC_ARG(vc_thread_math() != VC_MATH_F16, name ": math mode 3 (f16 operands) is built for vc_gated_conv1d_glu_fwd / _prepack / _fwd_packed only")
static int glu_math() { const int m = vc_thread_math(); return m == VC_MATH_BF16 || m == VC_MATH_F16 ? m : VC_MATH_FP32; }   // bf16x3: the exact fp32 kernel
extern "C" int vc_conv2d_fwd(const float* x, int64_t x_gs, const float* w, int64_t w_gs,
                             const float* bias, int64_t b_gs, float* y, int64_t y_gs,
                             int G, int B, int Cin, int H, int W, int Cout,
                             int kh, int kw, int sh, int sw, void* stream) {
    VC_ARG(x && w && y && G > 0 && B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "vc_conv2d_fwd: bad args");
    VC_ARG(Cout % 4 == 0, "vc_conv2d_fwd: Cout must be a multiple of 4");
    VC_NO_F16("vc_conv2d_fwd");
    return conv_fwd(x, x_gs, w, w_gs, bias, b_gs, y, y_gs, G, B, make_geom(Cin, H, W, Cout, kh, kw, sh, sw), vc_thread_math(), (hipStream_t)stream);
}
extern "C" int vc_conv2d_dgrad(const float* dy, int64_t dy_gs, const float* w, int64_t w_gs,
                               float* dx, int64_t dx_gs, const float* addend, int64_t add_gs, int accumulate,
                               int G, int B, int Cin, int H, int W, int Cout,
                               int kh, int kw, int sh, int sw, void* stream) {
    VC_ARG(dy && w && dx && G > 0 && B > 0, "vc_conv2d_dgrad: bad args");
    VC_NO_F16("vc_conv2d_dgrad");
    return conv_dgrad(dy, dy_gs, w, w_gs, dx, dx_gs, addend, add_gs, accumulate, G, B,
                      make_geom(Cin, H, W, Cout, kh, kw, sh, sw), vc_thread_math(), (hipStream_t)stream);
}
// tuning hook (tools/tune_igemm.py): operator-level filter gradients OVERWRITE dW (the step's plain-store form) while on
extern "C" int vc_conv2d_wgrad(const float* x, int64_t x_gs, const float* dy, int64_t dy_gs,
                               float* dw, int64_t dw_gs, int G, int B, int Cin, int H, int W, int Cout,
                               int kh, int kw, int sh, int sw, void* stream) {
    VC_ARG(x && dy && dw && G > 0 && B > 0, "vc_conv2d_wgrad: bad args");
    VC_NO_F16("vc_conv2d_wgrad");
    const WgradApp app{x, x_gs, dy, dy_gs, B};
    return conv_wgrad(&app, 1, dw, dw_gs, G, make_geom(Cin, H, W, Cout, kh, kw, sh, sw), vc_thread_math(), (hipStream_t)stream, /*store=*/vc_hooks().wgrad_store != 0);
}
extern "C" int vc_gated_conv1d_glu_fwd(const float* x, int64_t x_gs, const float* w, int64_t w_gs,
                                       const float* bias, int64_t b_gs, float* out, int64_t out_gs,
                                       int G, int B, int Cin, int T, int Cout2, int k, void* stream) {
    VC_ARG(x && w && out && G > 0 && B > 0 && Cin > 0 && T > 0 && Cout2 > 0 && k > 0, "vc_gated_conv1d_glu_fwd: bad args");
    VC_ARG(aligned16(out) && out_gs % 4 == 0, "vc_gated_conv1d_glu_fwd: out must be 16-byte aligned");
    WsConvArgs c{}; c.x = x; c.w = w; c.bias = bias; c.y = out; c.x_gs = x_gs; c.w_gs = w_gs; c.b_gs = b_gs; c.y_gs = out_gs;
    c.B = B; c.T = T; c.Cout = Cout2; c.G = G;
    const int r = launch_gated_conv1d_glu(c, Cin, k, glu_math(), (hipStream_t)stream, 2.0 * G * B * T * (double)Cout2 * k * Cin);
    VC_ARG(r != -2, "vc_gated_conv1d_glu_fwd: built for Cin = 24, k = 15, Cout2 %% 64 == 0, T %% 4 == 0, tensors < 2 GiB (the generator's h1 block)");
    return r;
}
int glu_tune_flags() { return vc_hooks().glu_flags; }
// The bf16 and the IEEE-half fragment image of the same filters have the same size and layout and differ only in how 16 bits are read: a
// call in the other mode would multiply garbage silently.  The library remembers which mode built the image at an address (the last
// VC_IMG_LOG prepack calls; host side, no device traffic) and refuses a packed call whose image it knows to be of another mode.  An image it
// has never seen (copied, or built by another process) passes: the contract in the header stays "built in the mode of the call".
#define VC_IMG_LOG 64
static std::mutex g_img_mu;
static struct { const void* p; int math; } g_img_log[VC_IMG_LOG];
static int g_img_next = 0;
static void glu_image_note(const void* img, int math) {
    std::lock_guard<std::mutex> l(g_img_mu);
    for (auto& e : g_img_log) if (e.p == img) { e.math = math; return; }
    g_img_log[g_img_next] = {img, math}; g_img_next = (g_img_next + 1) % VC_IMG_LOG;
}
static int glu_image_mode(const void* img) {       // -1: unknown
    std::lock_guard<std::mutex> l(g_img_mu);
    for (const auto& e : g_img_log) if (e.p == img && img) return e.math;
    return -1;
}
extern "C" int64_t vc_gated_conv1d_glu_image_bytes(int Cin, int k, int Cout2) {
    return gated_conv1d_glu_image_bytes(Cin, k, Cout2, glu_math());
}
extern "C" int vc_gated_conv1d_glu_prepack(const float* w, int64_t w_gs, void* image, int64_t image_gs_bytes,
                                           int G, int Cin, int k, int Cout2, void* stream) {
    VC_ARG(w && image && G > 0, "vc_gated_conv1d_glu_prepack: bad args");
    VC_ARG(aligned16(image) && image_gs_bytes % 16 == 0, "vc_gated_conv1d_glu_prepack: image must be 16-byte aligned");
    const int math = glu_math();
    const int r = launch_gated_conv1d_glu_prepack(w, w_gs, image, image_gs_bytes, G, Cin, k, Cout2, math, (hipStream_t)stream);
    VC_ARG(r != -2, "vc_gated_conv1d_glu_prepack: built for Cin = 24, k = 15, Cout2 %% 64 == 0");
    if (r == 0) glu_image_note(image, math);
    return r;
}
extern "C" int vc_gated_conv1d_glu_fwd_packed(const float* x, int64_t x_gs, const void* image, int64_t image_gs_bytes,
                                              const float* bias, int64_t b_gs, float* out, int64_t out_gs,
                                              int G, int B, int Cin, int T, int Cout2, int k, void* stream) {
    VC_ARG(x && image && out && G > 0 && B > 0 && Cin > 0 && T > 0 && Cout2 > 0 && k > 0, "vc_gated_conv1d_glu_fwd_packed: bad args");
    VC_ARG(aligned16(image) && image_gs_bytes % 16 == 0, "vc_gated_conv1d_glu_fwd_packed: image must be 16-byte aligned");
    WsConvArgs c{}; c.x = x; c.wimg = image; c.wimg_gs = image_gs_bytes; c.bias = bias; c.y = out; c.x_gs = x_gs; c.b_gs = b_gs; c.y_gs = out_gs;
    c.B = B; c.T = T; c.Cout = Cout2; c.G = G;
    const int math = glu_math();
    {
        const int built = glu_image_mode(image);
        VC_ARG(built < 0 || built == math, "vc_gated_conv1d_glu_fwd_packed: this image was built by vc_gated_conv1d_glu_prepack in math mode %d, the calling thread's mode is %d "
                                           "(a fragment image is read in the arithmetic it was packed for)", built, math);
    }
    const int r = launch_gated_conv1d_glu(c, Cin, k, math, (hipStream_t)stream, 2.0 * G * B * T * (double)Cout2 * k * Cin);
    VC_ARG(r != -2, "vc_gated_conv1d_glu_fwd_packed: built for Cin = 24, k = 15, Cout2 %% 64 == 0, T %% 4 == 0, tensors < 2 GiB (the generator's h1 block)");
    return r;
}
extern "C" int vc_prof_glu_copy_roof(const float* x, const void* image_bf16, float* out, int B, int T, int Cout2, void* stream) {
    VC_ARG(x && image_bf16 && out, "vc_prof_glu_copy_roof: bad args");
    WsConvArgs c{}; c.x = x; c.wimg = image_bf16; c.y = out; c.B = B; c.T = T; c.Cout = Cout2; c.G = 1;
    const int r = launch_glu_copy_roof(c, 24, 15, (hipStream_t)stream);
    VC_ARG(r != -2, "vc_prof_glu_copy_roof: geometry not built");
    return r;
}
extern "C" int vc_bias_grad(const float* dy, int64_t dy_gs, float* db, int64_t db_gs,
                            int G, int B, int C, int L, void* stream) {
    VC_ARG(dy && db, "vc_bias_grad: bad args");
    return launch_bias_grad(dy, dy_gs, db, db_gs, G, B, C, L, (hipStream_t)stream);
}

extern "C" int vc_instnorm_glu_fwd(int mode, int ps, const float* y, int64_t y_gs,
                                   const float* res, int64_t res_gs,
                                   const float* gamma, const float* beta, int64_t par_gs,
                                   float* out, int64_t out_gs, float* stats, int64_t st_gs,
                                   int G, int B, int C, int L, void* stream) {
    VC_ARG(y && out && G > 0 && B > 0 && C > 0 && L > 0, "vc_instnorm_glu_fwd: bad args");
    VC_ARG(mode == 1 || (gamma && beta && stats), "vc_instnorm_glu_fwd: gamma/beta/stats required");
    VC_ARG(mode != 2 || res, "vc_instnorm_glu_fwd: mode 2 needs res");
    NormArgs a{};
    a.y = y; a.res = res; a.gamma = gamma; a.beta = beta; a.out = out; a.stats = stats;
    a.y_gs = y_gs; a.res_gs = res_gs; a.par_gs = par_gs; a.out_gs = out_gs; a.st_gs = st_gs;
    a.G = G; a.B = B; a.C = C; a.L = L; a.ps = ps;
    return launch_norm_fwd(mode, a, (hipStream_t)stream);
}
extern "C" int vc_instnorm_glu_bwd(int mode, int ps, const float* y, int64_t y_gs,
                                   const float* dout, int64_t do_gs,
                                   const float* gamma, const float* beta, int64_t par_gs,
                                   const float* stats, int64_t st_gs, float* dy, int64_t dy_gs,
                                   float* dgamma, float* dbeta, int64_t dpar_gs,
                                   float* dbias, int64_t dbias_gs,
                                   int G, int B, int C, int L, void* stream) {
    VC_ARG(y && dout && dy && G > 0 && B > 0 && C > 0 && L > 0, "vc_instnorm_glu_bwd: bad args");
    VC_ARG(mode == 1 || (gamma && beta && stats), "vc_instnorm_glu_bwd: gamma/beta/stats required");
    NormArgs a{};
    a.y = y; a.dout = dout; a.gamma = gamma; a.beta = beta; a.stats_in = stats; a.dy = dy;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias;
    a.y_gs = y_gs; a.do_gs = do_gs; a.par_gs = par_gs; a.st_gs = st_gs; a.dy_gs = dy_gs;
    a.dpar_gs = dpar_gs; a.dbias_gs = dbias_gs;
    a.G = G; a.B = B; a.C = C; a.L = L; a.ps = ps; a.det = vc_det_active() ? 1 : 0;
    return launch_norm_bwd(mode, a, (hipStream_t)stream);
}
extern "C" int vc_disc_head_fwd(const float* x, int64_t x_gs, const float* w, const float* b, int64_t par_gs,
                                float* o, int64_t o_gs, int G, int B, int C, int HW, void* stream) {
    VC_ARG(x && w && b && o, "vc_disc_head_fwd: bad args");
    return launch_head_fwd(x, x_gs, w, b, par_gs, o, o_gs, G, B, C, HW, (hipStream_t)stream);
}
extern "C" int vc_disc_head_bwd(const float* x, int64_t x_gs, const float* w, int64_t par_gs,
                                const float* o, int64_t o_gs, const float* dout, int64_t do_gs,
                                float* dx, int64_t dx_gs, float* dw, float* db, int64_t dpar_gs,
                                int G, int B, int C, int HW, void* stream) {
    VC_ARG(x && w && o && dout, "vc_disc_head_bwd: bad args");
    return launch_head_bwd(x, x_gs, w, par_gs, o, o_gs, dout, do_gs, dx, dx_gs, dw, db, dpar_gs, G, B, C, HW, (hipStream_t)stream);
}
extern "C" int vc_crop_batch(const float* src, const int64_t* off, const int32_t* len, const int32_t* start, float* out,
                             int N, int F, int n_frames, void* stream) {
    VC_ARG(src && off && len && start && out && N > 0 && F > 0 && n_frames > 0, "vc_crop_batch: bad args");
    return launch_crop_batch(src, (const long*)off, len, start, out, N, F, n_frames, (hipStream_t)stream);
}
extern "C" int vc_l1_loss(const float* y, const float* y_hat, int64_t n, float* loss, float* d_yhat, void* stream) {
    VC_ARG(y && y_hat && loss && n > 0, "vc_l1_loss: bad args");
    return launch_mean_loss(y, y_hat, n, 0, loss, d_yhat, (hipStream_t)stream);
}
extern "C" int vc_l2_loss(const float* y, const float* y_hat, int64_t n, float* loss, float* d_yhat, void* stream) {
    VC_ARG(y && y_hat && loss && n > 0, "vc_l2_loss: bad args");
    return launch_mean_loss(y, y_hat, n, 1, loss, d_yhat, (hipStream_t)stream);
}
extern "C" int vc_adam_tf_update(float* p, float* m, float* v, const float* g, int64_t n,
                                 float alpha, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    VC_ARG(p && m && v && g && n > 0, "vc_adam_tf_update: bad args");
    VC_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)g % 16 == 0),
           "vc_adam_tf_update: buffers must be 16-byte aligned");
    return launch_adam(p, m, v, g, n, alpha, beta1, beta2, eps, grad_scale, (hipStream_t)stream);
}
