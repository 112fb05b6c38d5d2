// vc_api.hip -- host side of libvcgan_hip.so: geometry tables, parameter layout, the
// generator / discriminator forward+backward schedules and the fused CycleGAN step.
// Everything exported here is declared (with reference citations) in include/vc_cyclegan.h.
#include "vc_internal.h"
#include "../../include/vc_cyclegan.h"
#include "../../include/vc_cyclegan_testing.h"
#include <algorithm>
#include <array>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vc_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* vc_last_error(void) { return g_err; }
extern "C" int vc_version(void) { return 100; }
// CRC-32C (Castagnoli), slicing-by-8: the checksum of TensorFlow's checkpoint bundles and event files
// (model.py:21,26-29 tf.train.Saver / tf.summary.FileWriter); host-only utility for the checkpoint / summary writers.
extern "C" uint32_t vc_crc32c(const void* data, uint64_t n, uint32_t crc) {
    static uint32_t T[8][256];
    static const bool init = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xff];
        return true;
    }();
    (void)init;
    const unsigned char* p = (const unsigned char*)data;
    crc = ~crc;
    while (n && ((uintptr_t)p & 7)) { crc = (crc >> 8) ^ T[0][(crc ^ *p++) & 0xff]; --n; }
    while (n >= 8) {
        uint64_t v; memcpy(&v, p, 8);
        v ^= crc;
        crc = T[7][v & 0xff] ^ T[6][(v >> 8) & 0xff] ^ T[5][(v >> 16) & 0xff] ^ T[4][(v >> 24) & 0xff] ^
              T[3][(v >> 32) & 0xff] ^ T[2][(v >> 40) & 0xff] ^ T[1][(v >> 48) & 0xff] ^ T[0][(v >> 56) & 0xff];
        p += 8; n -= 8;
    }
    while (n--) crc = (crc >> 8) ^ T[0][(crc ^ *p++) & 0xff];
    return ~crc;
}

#define VC_CHECK(x) do { int _r = (x); if (_r != 0) return _r; } while (0)
#define VC_ARG(cond, ...) do { if (!(cond)) { vc_set_error(__VA_ARGS__); return VC_E_ARG; } } while (0)

// ------------------------------------------------------------------------------------
// geometry: TF 'SAME' padding (extra pad goes right/bottom)
// ------------------------------------------------------------------------------------
struct Geom { int Cin, H, W, Cout, kh, kw, sh, sw, OH, OW, pt, pl; };
static void same_pad(int n, int k, int s, int* out, int* lo) {
    *out = (n + s - 1) / s;
    int total = (*out - 1) * s + k - n; if (total < 0) total = 0;
    *lo = total / 2;
}
static Geom make_geom(int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw) {
    Geom g{Cin, H, W, Cout, kh, kw, sh, sw, 0, 0, 0, 0};
    same_pad(H, kh, sh, &g.OH, &g.pt);
    same_pad(W, kw, sw, &g.OW, &g.pl);
    return g;
}

// ------------------------------------------------------------------------------------
// K tables (device resident, cached per geometry, reference-counted by the plans that use them)
//
// A table lists the GEMM-K rows of one convolution (forward / filter gradient) or of one stride-parity class of its
// input gradient, tap by tap; every tap is padded to a multiple of 32 rows, so a 32-row chunk never straddles taps
// (vc_internal.h: IgemmArgs).  Padding rows: tap bit 0 for their run of 4 (the gathered operand reads 0), weight
// offset 0 (any finite weight), a_off -1 (no filter-gradient row).
// ------------------------------------------------------------------------------------
struct HostRow { int c, dh, dw, a_off, tap; };
struct Table { KEntry* dev = nullptr; int* xsoff = nullptr; unsigned* rtap = nullptr; unsigned* aoff = nullptr; Tap* taps = nullptr;
               int ntaps = 0, rows = 0, valid = 0, xbias = 0, refs = 0;
               int cpt = 0; signed char tdh[32] = {};          // K chunks per filter tap; row offset of every tap (filter-gradient rowskip)
               // the same chunks in channel-block-major, tap-minor order for the implicit-GEMM launches (upload_table); null
               // when the table has one tap or no uniform tap size
               int* xsoff_p = nullptr; unsigned* rtap_p = nullptr; unsigned* aoff_p = nullptr;
               // the same K rows by octets of 8 channels, for the 16-bit data path (vc_internal.h: ConvBf16Args); koct == 0:
               // the channel count is not a multiple of 8 and the path is not available for this geometry
               int* oxs = nullptr; unsigned* otap = nullptr; int* wsrc = nullptr; int koct = 0, oxbias = 0; };
typedef std::array<int, 14> TabKey;
static std::mutex g_tab_mu;
static std::map<TabKey, Table> g_tabs;

template <class T> static int to_device(const std::vector<T>& h, T** d) {
    VC_HIP(hipMalloc((void**)d, h.size() * sizeof(T)));
    VC_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}
static void free_table(Table& t) {
    (void)hipFree(t.dev); (void)hipFree(t.xsoff); (void)hipFree(t.rtap); (void)hipFree(t.aoff); (void)hipFree(t.taps);
    (void)hipFree(t.xsoff_p); (void)hipFree(t.rtap_p); (void)hipFree(t.aoff_p);
    (void)hipFree(t.oxs); (void)hipFree(t.otap); (void)hipFree(t.wsrc);
    t = Table{};
}
// rows_per_tap[t] consecutive entries of `rows` belong to tap t.  xH, xW: dims of the tensor the K rows gather from
// (forward: the input; input gradient: dY).
static int upload_table(const std::vector<HostRow>& rows, const std::vector<int>& rows_per_tap, const std::vector<Tap>& taps,
                        int xH, int xW, Table* t) {
    if (taps.size() > 32) { vc_set_error("conv kernel has %d taps; at most 32 supported", (int)taps.size()); return VC_E_ARG; }
    long min_soff = 0;
    for (const HostRow& r : rows) min_soff = std::min(min_soff, ((long)r.c * xH * xW + (long)r.dh * xW + r.dw) * 4);
    const long xbias = -min_soff;
    std::vector<KEntry> ke; std::vector<int> xs; std::vector<unsigned> rt, ao;
    size_t at = 0;
    for (size_t tp = 0; tp < rows_per_tap.size(); ++tp) {
        const int n = rows_per_tap[tp];
        if (n % 4) { vc_set_error("conv: channel count %d is not a multiple of 4", n); return VC_E_ARG; }
        const HostRow& r0 = rows[at];
        const long soff0 = ((long)r0.c * xH * xW + (long)r0.dh * xW + r0.dw) * 4 + xbias;
        for (int k = 0; k < n; ++k) {
            const HostRow& r = rows[at + k];
            const long soff = ((long)r.c * xH * xW + (long)r.dh * xW + r.dw) * 4 + xbias;
            if (soff > 0x7fffff00L || (long)r.a_off * 4 > 0x7fffff00L) { vc_set_error("tensor too large for 32-bit byte offsets"); return VC_E_ARG; }
            ke.push_back(KEntry{r.c, r.dh, r.dw, r.a_off}); xs.push_back((int)soff); ao.push_back((unsigned)r.a_off * 4u);
            if (k % 4 == 0) rt.push_back(1u << r.tap);
        }
        for (int k = n; k % VC_KC; ++k) {          // pad the tap to whole chunks: addressable (row 0 of the tap), never contributing
            ke.push_back(KEntry{r0.c, r0.dh, r0.dw, -1}); xs.push_back((int)soff0); ao.push_back(0u);
            if (k % 4 == 0) rt.push_back(0u);
        }
        at += n;
    }
    t->valid = (int)rows.size(); t->ntaps = (int)taps.size(); t->xbias = (int)xbias;
    t->cpt = rows_per_tap.empty() ? 0 : (rows_per_tap[0] + VC_KC - 1) / VC_KC;
    for (size_t tp = 0; tp < taps.size(); ++tp) {
        if (rows_per_tap[tp] != rows_per_tap[0] || taps[tp].dh < -100 || taps[tp].dh > 100) t->cpt = 0;   // no uniform tap size: rowskip unavailable
        t->tdh[tp] = (signed char)taps[tp].dh;
    }
    // Second K order for the implicit-GEMM launches (forward / input gradient), whole chunks permuted: channel block major,
    // tap minor.  The taps of one channel block gather the same lines of the tensor shifted by a pixel or a row; next to
    // each other, kh*kw - 1 of every kh*kw fetches hit the L1 / L2 instead of re-streaming the whole gathered tensor once
    // per tap through every XCD (round-3 PMC: 133 MB of L2-miss traffic per launch against ~25 MB of operands).  Any
    // order of whole chunks is correct: a chunk stays inside one tap and contiguous in the filter.  The filter gradient
    // keeps the tap-major arrays (its rowskip chunk ranges are per tap).
    std::vector<int> xs_p; std::vector<unsigned> rt_p, ao_p;
    if (t->cpt > 0 && taps.size() > 1) {
        const int cpt = t->cpt, nt = (int)taps.size();
        for (int cb = 0; cb < cpt; ++cb)
            for (int tp = 0; tp < nt; ++tp) {
                const size_t c = (size_t)tp * cpt + cb;
                xs_p.insert(xs_p.end(), xs.begin() + c * VC_KC, xs.begin() + (c + 1) * VC_KC);
                ao_p.insert(ao_p.end(), ao.begin() + c * VC_KC, ao.begin() + (c + 1) * VC_KC);
                rt_p.insert(rt_p.end(), rt.begin() + c * (VC_KC / 4), rt.begin() + (c + 1) * (VC_KC / 4));
            }
    }
    const bool perm = !xs_p.empty();
    auto pad_chunk = [&]() {
        for (int k = 0; k < VC_KC; ++k) {
            ke.push_back(KEntry{0, 0, 0, -1}); xs.push_back((int)xbias); ao.push_back(0u); if (k % 4 == 0) rt.push_back(0u);
            if (perm) { xs_p.push_back((int)xbias); ao_p.push_back(0u); if (k % 4 == 0) rt_p.push_back(0u); }
        }
    };
    while (ke.size() % 64 || ke.empty()) pad_chunk();          // the filter-gradient kernel walks 64-row tiles
    t->rows = (int)ke.size();
    for (int i = 0; i < 4; ++i) pad_chunk();                   // the loaders run ahead of the K range without branching
    VC_CHECK(to_device(ke, &t->dev)); VC_CHECK(to_device(xs, &t->xsoff)); VC_CHECK(to_device(rt, &t->rtap)); VC_CHECK(to_device(ao, &t->aoff));
    if (perm) { VC_CHECK(to_device(xs_p, &t->xsoff_p)); VC_CHECK(to_device(rt_p, &t->rtap_p)); VC_CHECK(to_device(ao_p, &t->aoff_p)); }
    std::vector<Tap> tp = taps; if (tp.empty()) tp.push_back(Tap{0, 0});
    VC_CHECK(to_device(tp, &t->taps));
    // octet view: unit offsets inside the packed tensor [C/8][xH*xW][8] (16 bytes per unit), filter source offset per K row
    bool oct = !rows.empty();
    for (int n : rows_per_tap) oct = oct && n % 8 == 0;
    if (oct) {
        long omin = 0;
        for (const HostRow& r : rows) omin = std::min(omin, ((long)(r.c / 8) * xH * xW + (long)r.dh * xW + r.dw) * 16);
        // K order of this view: channel octet major, tap minor.  The taps of one channel octet read the same 16-byte
        // units shifted by a pixel, so putting them next to each other turns kh*kw - 1 of every kh*kw fetches of the
        // gathered tensor into cache hits (tap-major order re-streamed the whole tensor once per tap: measured 43 % L2
        // misses, the launch bound by fabric traffic).  Any K order is correct: the filter image is built in the same one.
        std::vector<int> oxs, wsrc; std::vector<unsigned> otap;
        std::vector<size_t> first;                            // first row of every octet, in tap-major table order
        for (size_t k = 0; k < rows.size(); k += 8) first.push_back(k);
        std::stable_sort(first.begin(), first.end(), [&](size_t a, size_t b) { return rows[a].c < rows[b].c; });
        for (size_t k : first) {
            const HostRow& r = rows[k];
            for (int e = 0; e < 8; ++e) wsrc.push_back(rows[k + e].a_off);
            const long so = ((long)(r.c / 8) * xH * xW + (long)r.dh * xW + r.dw) * 16 - omin;
            if (so > 0x7fffff00L) { oct = false; break; }
            oxs.push_back((int)so); otap.push_back(1u << r.tap);
        }
        if (oct) {
            auto pad_oct = [&]() { oxs.push_back((int)-omin); otap.push_back(0u); for (int e = 0; e < 8; ++e) wsrc.push_back(-1); };
            while (oxs.size() % 8) pad_oct();
            t->koct = (int)oxs.size(); t->oxbias = (int)-omin;
            for (int i = 0; i < 40; ++i) pad_oct();            // the loaders run up to four chunks past their K range
            VC_CHECK(to_device(oxs, &t->oxs)); VC_CHECK(to_device(otap, &t->otap)); VC_CHECK(to_device(wsrc, &t->wsrc));
        }
    }
    return 0;
}
static TabKey fwd_key(const Geom& g) { return TabKey{0, g.Cin, g.Cout, g.kh, g.kw, g.sh, g.sw, g.pt, g.pl, 0, 0, g.H, g.W, 0}; }
static TabKey dgrad_key(const Geom& g, int ph, int pw) { return TabKey{1, g.Cin, g.Cout, g.kh, g.kw, g.sh, g.sw, g.pt, g.pl, ph, pw, g.OH, g.OW, 0}; }
// forward / wgrad table: row kk = ((i*kw + j)*Cin + ci), tap = i*kw + j
static int get_fwd_table(const Geom& g, Table* out, std::vector<TabKey>* owner = nullptr) {
    const TabKey key = fwd_key(g);
    std::lock_guard<std::mutex> l(g_tab_mu);
    auto it = g_tabs.find(key);
    if (it == g_tabs.end()) {
        std::vector<HostRow> rows; std::vector<Tap> taps; std::vector<int> rpt;
        for (int i = 0; i < g.kh; ++i) for (int j = 0; j < g.kw; ++j) {
            taps.push_back(Tap{i - g.pt, j - g.pl}); rpt.push_back(g.Cin);
            for (int ci = 0; ci < g.Cin; ++ci)
                rows.push_back(HostRow{ci, i - g.pt, j - g.pl, ((i * g.kw + j) * g.Cin + ci) * g.Cout, i * g.kw + j});
        }
        Table t; VC_CHECK(upload_table(rows, rpt, taps, g.H, g.W, &t));
        it = g_tabs.emplace(key, t).first;
    }
    if (owner) { ++it->second.refs; owner->push_back(key); }
    *out = it->second;
    return 0;
}
// input-gradient table of parity class (ph, pw): rows (tap in class, co), gathered tensor = dY [Cout][OH][OW]
static int get_dgrad_table(const Geom& g, int ph, int pw, Table* out, std::vector<TabKey>* owner = nullptr) {
    const TabKey key = dgrad_key(g, ph, pw);
    std::lock_guard<std::mutex> l(g_tab_mu);
    auto it = g_tabs.find(key);
    if (it == g_tabs.end()) {
        std::vector<HostRow> rows; std::vector<Tap> taps; std::vector<int> rpt;
        for (int i = 0; i < g.kh; ++i) {
            const int nh = ph + g.pt - i;
            if (((nh % g.sh) + g.sh) % g.sh) continue;
            for (int j = 0; j < g.kw; ++j) {
                const int nw = pw + g.pl - j;
                if (((nw % g.sw) + g.sw) % g.sw) continue;
                const int dh = nh / g.sh, dw = nw / g.sw;   // exact
                const int tap = (int)taps.size();
                taps.push_back(Tap{dh, dw}); rpt.push_back(g.Cout);
                for (int co = 0; co < g.Cout; ++co)
                    rows.push_back(HostRow{co, dh, dw, ((i * g.kw + j) * g.Cin) * g.Cout + co, tap});
            }
        }
        Table t; VC_CHECK(upload_table(rows, rpt, taps, g.OH, g.OW, &t));
        it = g_tabs.emplace(key, t).first;
    }
    if (owner) { ++it->second.refs; owner->push_back(key); }
    *out = it->second;
    return 0;
}
// a plan gives its tables back; the last owner frees them (hipFree waits for the device, so no launch still reads them)
static void release_tables(std::vector<TabKey>& keys) {
    std::lock_guard<std::mutex> l(g_tab_mu);
    for (const TabKey& k : keys) {
        auto it = g_tabs.find(k);
        if (it != g_tabs.end() && --it->second.refs <= 0) { free_table(it->second); g_tabs.erase(it); }
    }
    keys.clear();
}

// arithmetic of the MFMA kernels for calls that carry no plan (operator level): per calling thread
static thread_local int t_math = VC_MATH_FP32;
extern "C" int vc_set_math_mode(int mode) {
    if (mode < VC_MATH_FP32 || mode > VC_MATH_BF16X3) { vc_set_error("vc_set_math_mode: mode %d (0 fp32, 1 bf16, 2 bf16x3)", mode); return VC_E_ARG; }
    t_math = mode;
    return 0;
}
extern "C" int vc_get_math_mode(void) { return t_math; }

// deterministic mode (vc_internal.h): the calling thread's default, and the setting of the plan whose launches are being issued
static thread_local int t_det = 0, t_det_scope = -1;
bool vc_det_active() { return (t_det_scope >= 0 ? t_det_scope : t_det) != 0; }
DetScope::DetScope(int det) : saved(t_det_scope) { t_det_scope = det ? 1 : 0; }
DetScope::~DetScope() { t_det_scope = saved; }
extern "C" int vc_set_deterministic(int on) { t_det = on ? 1 : 0; return 0; }
extern "C" int vc_get_deterministic(void) { return t_det; }

// ------------------------------------------------------------------------------------
// conv operators
// ------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Cin = 1 with at most 9 taps (the discriminator's h1 layer) goes to the vector-ALU kernels
static bool c1_path(const Geom& g) { return g.Cin == 1 && g.kh * g.kw <= 9; }
// 1-D, stride 1, 24 input channels, 15 taps (the generator's h1 layer at 24 MCEPs): weight-stationary kernel
static bool wsconv_path(const Geom& g) {
    return g.H == 1 && g.kh == 1 && g.sh == 1 && g.sw == 1 && g.Cin == 24 && g.kw == 15 && g.W % 4 == 0;   // 16-byte output stores
}
static C1Args c1_args(const Geom& g, int G, int B, int math) {
    C1Args c{};
    c.B = B; c.H = g.H; c.W = g.W; c.Cout = g.Cout; c.kh = g.kh; c.kw = g.kw; c.sh = g.sh; c.sw = g.sw;
    c.OH = g.OH; c.OW = g.OW; c.pt = g.pt; c.pl = g.pl; c.G = G; c.math = math; c.det = vc_det_active() ? 1 : 0;
    return c;
}
static int check_conv_geom(const Geom& g, int B) {
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
static thread_local int t_rowskip_wgrad = 1;
extern "C" int vc_tune_rowskip(int wgrad_on) { t_rowskip_wgrad = wgrad_on ? 1 : 0; return 0; }
static bool rowskip_pays(const Table& t, int OHc, int si, int Hl) {
    if (!t_rowskip_wgrad || OHc < 2 || t.ntaps < 2 || t.cpt <= 0) return false;
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
static long ws_generation(std::initializer_list<hipStream_t> streams) {
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
static void drop_pack_spaces(std::initializer_list<hipStream_t> streams) {
    std::lock_guard<std::mutex> l(g_ws_mu);
    for (hipStream_t s : streams) {
        auto it = g_ws.find(s);
        if (it == g_ws.end()) continue;
        if (it->second.p) (void)hipFree(it->second.p);      // hipFree waits for the device
        g_ws.erase(it);
        g_ws_gen.erase(s);                        // the stream is going away with its plan: nobody else's key mentions it
    }
}
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// the path serves launches that fill its 128 x 128 tiles; everything else stays on the table-driven fp32-storage kernels
static bool bf16_path(const Table& t, int math, int M, long Ptot) { return math == VC_MATH_BF16 && t.koct > 0 && M >= 64 && Ptot >= 64; }
// the forward of this geometry at batch B runs on the 16-bit data path (the decision conv_fwd takes)
static bool conv_fwd_packs_input(const Geom& g, int B, int math) {
    if (math != VC_MATH_BF16 || c1_path(g) || wsconv_path(g)) return false;
    Table t; if (get_fwd_table(g, &t, nullptr)) return false;
    return bf16_path(t, math, g.Cout, (long)B * g.OH * g.OW);
}
static void bf16_set_vec4(ConvBf16Args& a) {
    a.vec4 = a.so_w == 1 && a.OWc % 4 == 0 && a.o_w0 % 4 == 0 && a.yW % 4 == 0 && a.yHW % 4 == 0 && a.y_bs % 4 == 0 &&
             a.y_gs % 4 == 0 && aligned16(a.Y) && (!a.addend || (aligned16(a.addend) && a.add_gs % 4 == 0));
}

// What a forward convolution needs to finish a gated block with instance norm in its own epilogue (IgemmArgs::fuse)
struct FuseNorm { const float* gamma; const float* beta; long par_gs; float* stats; long st_gs; float* act; long act_gs; bool done; };
// query != nullptr: do not launch, only report whether the launch would use split-K atomics (needs a zeroed y)
static int conv_fwd(const float* x, long x_gs, const float* w, long w_gs, const float* bias, long b_gs,
                    float* y, long y_gs, int G, int B, const Geom& g, int math, hipStream_t s,
                    bool prezeroed = false, bool* query = nullptr, const char* packed_w = nullptr,
                    char* xp_keep = nullptr, long xp_keep_gs = 0, FuseNorm* fuse = nullptr) {
    // fuse: the caller would like instance norm + GLU done in the epilogue; fuse->done tells whether this launch did it
    // xp_keep: where the packed bf16 copy of x is left for the layer's filter gradient (16-bit data path; group stride in bytes)
    VC_CHECK(check_conv_geom(g, B));
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
        VC_CHECK(launch_pack_act_bf16(x, x_gs, xp, a.x_gs, G, B, g.Cin, g.H * g.W, s));
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

static int conv_dgrad(const float* dy, long dy_gs, const float* w, long w_gs, float* dx, long dx_gs,
                      const float* addend, long add_gs, int accumulate, int G, int B, const Geom& g, int math, hipStream_t s,
                      bool prezeroed = false, bool* query = nullptr, const char* const* packed_w = nullptr) {
    VC_CHECK(check_conv_geom(g, B));
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
            VC_CHECK(launch_pack_act_bf16(dy, dy_gs, wsp, (long)xp_g, G, B, g.Cout, g.OH * g.OW, s));
            if (split && !accumulate) {
                if (!prezeroed)
                    for (int gi = 0; gi < G; ++gi)
                        VC_HIP(hipMemsetAsync(dx + (long)gi * dx_gs, 0, (size_t)B * g.Cin * g.H * g.W * sizeof(float), s));
                for (int i = 0; i < nc; ++i) cl[i].accumulate = 1;
            }
            for (int i = 0; i < nc; ++i) {
                cl[i].X = wsp; cl[i].A = packed_w ? packed_w[i] : wsp + xp_g * G + ap_off[i];
                if (!packed_w) VC_CHECK(launch_pack_filter_bf16(w, w_gs, wsp + xp_g * G + ap_off[i], cl[i].a_gs, wsrc[i], g.Cout, G, g.Cin, cl[i].koct * 8, s));
                VC_CHECK(launch_conv_bf16(cl[i], s, flb[i]));
            }
            return 0;
        }
    }
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

// One application of a layer as seen by the filter gradient: its input, the gradient of its raw output, its batch.
struct WgradApp { const float* x; long x_gs; const float* dy; long dy_gs; int B;
                  const char* xp = nullptr; long xp_gs = 0; };   // xp: x already packed for the 16-bit data path (group stride in bytes)
// dW (+)= sum over the applications; store: dW holds nothing yet (plain stores when the launch does not split K)
// query != nullptr: only report whether the launch would accumulate with atomics (needs a zeroed dW)
static int conv_wgrad(const WgradApp* apps, int napps, float* dw, long dw_gs, int G, const Geom& g, int math, hipStream_t s,
                      bool store, bool* query = nullptr) {
    if (c1_path(g)) {
        if (query) { *query = true; return 0; }
        for (int k = 0; k < napps; ++k) {
            VC_CHECK(check_conv_geom(g, apps[k].B));
            C1Args c = c1_args(g, G, apps[k].B, math); c.x = apps[k].x; c.x_gs = apps[k].x_gs; c.dy = apps[k].dy; c.y_gs = apps[k].dy_gs;
            c.dw = dw; c.dw_gs = dw_gs;
            VC_CHECK(launch_c1_wgrad(c, s));
        }
        return 0;
    }
    VC_ARG(napps >= 1 && napps <= 2, "conv filter gradient: 1 or 2 applications per launch");
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
            char* wsp; VC_CHECK(pack_space(s, (xcol_g + dyq_g) * G + xp_tot, &wsp));
            char* xcol = wsp; char* dyq = wsp + xcol_g * G; char* xtmp = dyq + dyq_g * G;
            double flops = 0;
            for (int k = 0; k < napps; ++k) {
                const char* xp = apps[k].xp; long xp_gs = apps[k].xp_gs;
                if (!xp) {
                    VC_CHECK(launch_pack_act_bf16(apps[k].x, apps[k].x_gs, xtmp, (long)xp_g[k], G, apps[k].B, g.Cin, g.H * g.W, s));
                    xp = xtmp; xp_gs = (long)xp_g[k]; xtmp += xp_g[k] * G;
                }
                VC_CHECK(launch_im2col_bf16(xp, xp_gs, xcol, (long)xcol_g, t.taps, t.ntaps, G, apps[k].B, g.Cin, g.H, g.W, g.OH, g.OW, g.sh, g.sw, po0[k], s));
                VC_CHECK(launch_pack_pixel_octets_bf16(apps[k].dy, apps[k].dy_gs, dyq, (long)dyq_g, G, apps[k].B, g.Cout, NPB, po0[k], s));
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

extern "C" int vc_conv2d_fwd(const float* x, int64_t x_gs, const float* w, int64_t w_gs,
                             const float* bias, int64_t b_gs, float* y, int64_t y_gs,
                             int G, int B, int Cin, int H, int W, int Cout,
                             int kh, int kw, int sh, int sw, void* stream) {
    VC_ARG(x && w && y && G > 0 && B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "vc_conv2d_fwd: bad args");
    VC_ARG(Cout % 4 == 0, "vc_conv2d_fwd: Cout must be a multiple of 4");
    return conv_fwd(x, x_gs, w, w_gs, bias, b_gs, y, y_gs, G, B, make_geom(Cin, H, W, Cout, kh, kw, sh, sw), t_math, (hipStream_t)stream);
}
extern "C" int vc_conv2d_dgrad(const float* dy, int64_t dy_gs, const float* w, int64_t w_gs,
                               float* dx, int64_t dx_gs, const float* addend, int64_t add_gs, int accumulate,
                               int G, int B, int Cin, int H, int W, int Cout,
                               int kh, int kw, int sh, int sw, void* stream) {
    VC_ARG(dy && w && dx && G > 0 && B > 0, "vc_conv2d_dgrad: bad args");
    return conv_dgrad(dy, dy_gs, w, w_gs, dx, dx_gs, addend, add_gs, accumulate, G, B,
                      make_geom(Cin, H, W, Cout, kh, kw, sh, sw), t_math, (hipStream_t)stream);
}
// tuning hook (tools/tune_igemm.py): operator-level filter gradients OVERWRITE dW (the step's plain-store form) while on
static thread_local int t_wgrad_store = 0;
extern "C" int vc_tune_wgrad_store(int on) { t_wgrad_store = on ? 1 : 0; return 0; }
extern "C" int vc_conv2d_wgrad(const float* x, int64_t x_gs, const float* dy, int64_t dy_gs,
                               float* dw, int64_t dw_gs, int G, int B, int Cin, int H, int W, int Cout,
                               int kh, int kw, int sh, int sw, void* stream) {
    VC_ARG(x && dy && dw && G > 0 && B > 0, "vc_conv2d_wgrad: bad args");
    const WgradApp app{x, x_gs, dy, dy_gs, B};
    return conv_wgrad(&app, 1, dw, dw_gs, G, make_geom(Cin, H, W, Cout, kh, kw, sh, sw), t_math, (hipStream_t)stream, /*store=*/t_wgrad_store != 0);
}
extern "C" int vc_gated_conv1d_glu_fwd(const float* x, int64_t x_gs, const float* w, int64_t w_gs,
                                       const float* bias, int64_t b_gs, float* out, int64_t out_gs,
                                       int G, int B, int Cin, int T, int Cout2, int k, void* stream) {
    VC_ARG(x && w && out && G > 0 && B > 0 && Cin > 0 && T > 0 && Cout2 > 0 && k > 0, "vc_gated_conv1d_glu_fwd: bad args");
    VC_ARG(aligned16(out) && out_gs % 4 == 0, "vc_gated_conv1d_glu_fwd: out must be 16-byte aligned");
    WsConvArgs c{}; c.x = x; c.w = w; c.bias = bias; c.y = out; c.x_gs = x_gs; c.w_gs = w_gs; c.b_gs = b_gs; c.y_gs = out_gs;
    c.B = B; c.T = T; c.Cout = Cout2; c.G = G;
    const int r = launch_gated_conv1d_glu(c, Cin, k, t_math, (hipStream_t)stream, 2.0 * G * B * T * (double)Cout2 * k * Cin);
    VC_ARG(r != -2, "vc_gated_conv1d_glu_fwd: built for Cin = 24, k = 15, Cout2 %% 64 == 0, T %% 4 == 0, tensors < 2 GiB (the generator's h1 block)");
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

// ------------------------------------------------------------------------------------
// network descriptions + parameter layout
// ------------------------------------------------------------------------------------
enum { M_GATED_IN = 0, M_GATED = 1, M_IN_RES = 2, M_PLAIN = 3 };
struct Layer {
    int kh, kw, cin, cout, sh, sw;   // cout = stored conv width (2C for gated pairs)
    int mode, ps;
    int C;                           // channels per norm/GLU branch after (optional) shuffle
    long w, b, gamma, beta;          // float offsets inside one network block
    std::string n_conv, n_gate;      // TF scope names
    int in_idx;                      // index of first InstanceNorm scope (creation order)
};
struct NetDesc { std::vector<Layer> layers; long head_w = -1, head_b = -1; int head_c = 0; long total = 0; };

static long align4(long x) { return (x + 3) & ~3L; }

static void add_layer(NetDesc& n, int& in_ctr, const char* conv, const char* gate, int kh, int kw, int cin,
                      int cout, int sh, int sw, int mode, int ps) {
    Layer L{};
    L.kh = kh; L.kw = kw; L.cin = cin; L.cout = cout; L.sh = sh; L.sw = sw; L.mode = mode; L.ps = ps;
    L.n_conv = conv; L.n_gate = gate ? gate : "";
    L.C = (mode == M_GATED_IN || mode == M_GATED) ? (ps ? cout / 4 : cout / 2) : cout;
    L.w = n.total; n.total = align4(n.total + (long)kh * kw * cin * cout);
    L.b = n.total; n.total = align4(n.total + cout);
    L.gamma = L.beta = -1; L.in_idx = -1;
    if (mode == M_GATED_IN || mode == M_IN_RES) {
        const int nb = (mode == M_GATED_IN) ? 2 : 1;
        L.in_idx = in_ctr; in_ctr += nb;
        L.gamma = n.total; n.total = align4(n.total + nb * L.C);
        L.beta = n.total; n.total = align4(n.total + nb * L.C);
    }
    n.layers.push_back(L);
}

static NetDesc make_generator(int F) {   // module.py:161-182
    NetDesc n; int in = 0; char a[64], b[64];
    add_layer(n, in, "h1_conv", "h1_conv_gates", 1, 15, F, 256, 1, 1, M_GATED, 0);
    add_layer(n, in, "downsample1d_block1_h1_conv", "downsample1d_block1_h1_gates", 1, 5, 128, 512, 1, 2, M_GATED_IN, 0);
    add_layer(n, in, "downsample1d_block2_h1_conv", "downsample1d_block2_h1_gates", 1, 5, 256, 1024, 1, 2, M_GATED_IN, 0);
    for (int i = 1; i <= 6; ++i) {
        snprintf(a, sizeof a, "residual1d_block%d_h1_conv", i); snprintf(b, sizeof b, "residual1d_block%d_h1_gates", i);
        add_layer(n, in, a, b, 1, 3, 512, 2048, 1, 1, M_GATED_IN, 0);
        snprintf(a, sizeof a, "residual1d_block%d_h2_conv", i);
        add_layer(n, in, a, nullptr, 1, 3, 1024, 512, 1, 1, M_IN_RES, 0);
    }
    add_layer(n, in, "upsample1d_block1_h1_conv", "upsample1d_block1_h1_gates", 1, 5, 512, 2048, 1, 1, M_GATED_IN, 1);
    add_layer(n, in, "upsample1d_block2_h1_conv", "upsample1d_block2_h1_gates", 1, 5, 512, 1024, 1, 1, M_GATED_IN, 1);
    add_layer(n, in, "o1_conv", nullptr, 1, 15, 256, F, 1, 1, M_PLAIN, 0);   // module.py:182 (24 == num_features)
    return n;
}
static NetDesc make_discriminator() {    // module.py:201-211
    NetDesc n; int in = 0;
    add_layer(n, in, "h1_conv", "h1_conv_gates", 3, 3, 1, 256, 1, 2, M_GATED, 0);
    add_layer(n, in, "downsample2d_block1_h1_conv", "downsample2d_block1_h1_gates", 3, 3, 128, 512, 2, 2, M_GATED_IN, 0);
    add_layer(n, in, "downsample2d_block2_h1_conv", "downsample2d_block2_h1_gates", 3, 3, 256, 1024, 2, 2, M_GATED_IN, 0);
    add_layer(n, in, "downsample2d_block3_h1_conv", "downsample2d_block3_h1_gates", 6, 3, 512, 2048, 1, 2, M_GATED_IN, 0);
    n.head_c = 1024;
    n.head_w = n.total; n.total = align4(n.total + 1024);
    n.head_b = n.total; n.total = align4(n.total + 1);
    return n;
}

static void put_rec(std::vector<vc_param_rec>& R, const std::string& name, long off, int nd, const int* shp, int ld) {
    vc_param_rec r{}; snprintf(r.name, sizeof r.name, "%s", name.c_str());
    r.offset = off; r.ndim = nd; for (int i = 0; i < nd; ++i) r.shape[i] = shp[i]; r.ld = ld;
    R.push_back(r);
}
static std::string in_name(int idx) { return idx == 0 ? "InstanceNorm" : "InstanceNorm_" + std::to_string(idx); }

extern "C" int vc_param_layout(int kind, int num_features, vc_param_rec* recs, int max_recs, int64_t* n_floats) {
    VC_ARG(kind == 0 || kind == 1, "vc_param_layout: kind must be 0 or 1");
    VC_ARG(kind == 1 || (num_features > 0 && num_features % 4 == 0), "vc_param_layout: num_features must be a positive multiple of 4");
    const NetDesc n = kind == 0 ? make_generator(num_features) : make_discriminator();
    std::vector<vc_param_rec> R;
    for (const Layer& L : n.layers) {
        const bool gated = L.mode == M_GATED || L.mode == M_GATED_IN;
        const int Cb = gated ? L.cout / 2 : L.cout;            // filters per TF conv
        int shp[4]; int nd;
        if (kind == 0) { nd = 3; shp[0] = L.kw; shp[1] = L.cin; shp[2] = Cb; }
        else { nd = 4; shp[0] = L.kh; shp[1] = L.kw; shp[2] = L.cin; shp[3] = Cb; }
        int s1[1] = {Cb};
        put_rec(R, L.n_conv + "/kernel", L.w, nd, shp, L.cout);
        put_rec(R, L.n_conv + "/bias", L.b, 1, s1, 1);
        int sc[1] = {L.C};
        if (L.in_idx >= 0) {
            put_rec(R, in_name(L.in_idx) + "/beta", L.beta, 1, sc, 1);
            put_rec(R, in_name(L.in_idx) + "/gamma", L.gamma, 1, sc, 1);
        }
        if (gated) {
            put_rec(R, L.n_gate + "/kernel", L.w + Cb, nd, shp, L.cout);
            put_rec(R, L.n_gate + "/bias", L.b + Cb, 1, s1, 1);
            if (L.in_idx >= 0) {
                put_rec(R, in_name(L.in_idx + 1) + "/beta", L.beta + L.C, 1, sc, 1);
                put_rec(R, in_name(L.in_idx + 1) + "/gamma", L.gamma + L.C, 1, sc, 1);
            }
        }
    }
    if (kind == 1) {
        int s2[2] = {n.head_c, 1}; int s1[1] = {1};
        put_rec(R, "dense/kernel", n.head_w, 2, s2, 1);
        put_rec(R, "dense/bias", n.head_b, 1, s1, 1);
    }
    if (n_floats) *n_floats = n.total;
    for (int i = 0; i < (int)R.size() && i < max_recs; ++i) recs[i] = R[i];
    return (int)R.size();
}

// ------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------
struct LayerGeo { Geom g; int inL; long raw_ps, act_ps; };   // per-sample sizes (floats) of raw conv output / activation
struct vc_plan {
    int F, T;
    NetDesc G, D;
    std::vector<LayerGeo> gg, dg;
    int d_hw;                     // discriminator output pixels per sample
    int math = VC_MATH_FP32;      // arithmetic of the MFMA kernels for every launch made through this plan
    int det = 0;                  // deterministic mode (vc_plan_set_deterministic): fixed summation order in every launch of this plan
    bool inference = false;       // generator-forward tables only
    std::vector<TabKey> tables;   // geometry tables this plan holds a reference on
    hipEvent_t d_ready = nullptr;
    // side stream for the chains of the step that do not depend on each other (fork / join with events)
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // filter gradients are leaves of the backward chain: they run on their own stream behind one event per layer
    hipStream_t wstream = nullptr; hipEvent_t ev_layer[24] = {}; hipEvent_t ev_wjoin = nullptr;
    hipStream_t wstream_d = nullptr; hipEvent_t ev_layer_d[8] = {}; hipEvent_t ev_wjoin_d = nullptr;   // same for the discriminator chain
    int use_side = 7;             // bit 0: independent chains on the side stream; bit 1 / 2: generator / discriminator filter gradients on their own streams
    // generator gradient buckets: layers [first, last] of each generator become final (in reverse layer order) during
    // the last backward of the step; an event per bucket lets the caller overlap their all-reduce with the rest of it
    static constexpr int kBuckets = 4;
    int bucket_first[kBuckets] = {15, 9, 3, 0};          // {u1,u2,o1}, {r4..r6}, {r1..r3}, {h1,d1,d2}
    hipEvent_t g_ready[kBuckets] = {nullptr, nullptr, nullptr, nullptr};
    // ... and the moment the step has issued its last READ of the same layers' weights (their input gradients run on
    // the main stream, the filter gradients on the wgrad stream): after both, the caller may update them in place
    hipEvent_t g_free[kBuckets] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t d_free = nullptr;                          // last read of the discriminator weights (generator_loss path)
    // captured train steps (vc_train_step_graph): one executable graph per (batch, schedule, buffer set)
    struct StepGraph { std::array<const void*, 12> key; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<StepGraph> graphs;
    hipStream_t gstream = nullptr, ustream = nullptr;     // capture / replay stream, update (exchange + Adam) stream
    hipEvent_t ev_gin = nullptr, ev_gout = nullptr, ev_ujoin = nullptr;
    float* hp_dev = nullptr;                              // {lambda_cycle, lambda_identity, alpha_g, alpha_d} of the step being replayed
    // bf16 data path: filter images of every layer (forward + one per input-gradient parity class), rebuilt from the
    // fp32 weights once at the start of a train step instead of before every launch (a step uses each image 2-3 times)
    struct LayerImages { const char* fwd = nullptr; const char* dg[4] = {nullptr, nullptr, nullptr, nullptr}; bool has_dg = false; };
    mutable std::vector<LayerImages> img_g, img_d;
    mutable char* img_block = nullptr;
};

static void plan_geos(const NetDesc& n, int C0, int H0, int W0, std::vector<LayerGeo>& out) {
    int C = C0, H = H0, W = W0;
    for (const Layer& L : n.layers) {
        LayerGeo lg{};
        lg.g = make_geom(C, H, W, L.cout, L.kh, L.kw, L.sh, L.sw);
        lg.raw_ps = (long)L.cout * lg.g.OH * lg.g.OW;
        const int nb = (L.mode == M_GATED || L.mode == M_GATED_IN) ? 2 : 1;
        lg.act_ps = lg.raw_ps / nb;
        out.push_back(lg);
        if (L.mode == M_IN_RES) { C = L.cout; }           // residual tail: same shape as block input
        else if (L.mode == M_PLAIN) { C = L.cout; }
        else { C = L.C; }
        H = lg.g.OH; W = lg.g.OW * (L.ps ? 2 : 1);
    }
}

static int plan_create(int num_features, int n_frames, bool inference, vc_plan** out) {
    VC_ARG(out, "vc_plan_create: out is NULL");
    VC_ARG(num_features > 0 && num_features % 4 == 0, "vc_plan_create: num_features must be a positive multiple of 4");
    VC_ARG(n_frames > 0 && n_frames % 4 == 0, "vc_plan_create: n_frames must be a positive multiple of 4 (preprocess.py:146-156)");
    vc_plan* p = new vc_plan();
    p->F = num_features; p->T = n_frames; p->inference = inference; p->math = t_math; p->det = t_det;
    p->G = make_generator(num_features); p->D = make_discriminator();
    plan_geos(p->G, num_features, 1, n_frames, p->gg);
    plan_geos(p->D, 1, num_features, n_frames, p->dg);
    p->d_hw = p->dg.back().g.OH * p->dg.back().g.OW;
    // build every geometry table now so steady-state calls never allocate (an inference plan: generator forward only)
    int r = 0;
    for (auto* v : {&p->gg, &p->dg}) {
        if (inference && v == &p->dg) break;
        for (auto& l : *v) {
            if (c1_path(l.g)) continue;
            Table t;
            if (!(wsconv_path(l.g) && inference)) r = r ? r : get_fwd_table(l.g, &t, &p->tables);
            if (inference) continue;
            for (int ph = 0; ph < l.g.sh && !r; ++ph) for (int pw = 0; pw < l.g.sw && !r; ++pw) r = get_dgrad_table(l.g, ph, pw, &t, &p->tables);
        }
    }
    if (r) { release_tables(p->tables); delete p; return r; }
    auto mk_stream = [&](hipStream_t* st) { return hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
    hipError_t e = hipEventCreateWithFlags(&p->d_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = mk_stream(&p->side);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = mk_stream(&p->wstream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_wjoin, hipEventDisableTiming);
    for (int k = 0; k < 24 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&p->ev_layer[k], hipEventDisableTiming);
    if (e == hipSuccess) e = mk_stream(&p->wstream_d);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_wjoin_d, hipEventDisableTiming);
    for (int k = 0; k < 8 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&p->ev_layer_d[k], hipEventDisableTiming);
    for (int k = 0; k < vc_plan::kBuckets && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&p->g_ready[k], hipEventDisableTiming);
    for (int k = 0; k < vc_plan::kBuckets && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&p->g_free[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->d_free, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->gstream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->ustream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_gin, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_gout, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_ujoin, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&p->hp_dev, 4 * sizeof(float));
    if (e != hipSuccess) { vc_set_error("hipEventCreate failed: %s", hipGetErrorString(e)); vc_plan_destroy(p); return (int)e; }
    *out = p;
    return 0;
}
extern "C" int vc_plan_create(int num_features, int n_frames, vc_plan** out) { return plan_create(num_features, n_frames, false, out); }
extern "C" int vc_plan_create_inference(int num_features, int n_frames, vc_plan** out) { return plan_create(num_features, n_frames, true, out); }
extern "C" void vc_plan_destroy(vc_plan* p) {
    if (!p) return;
    drop_pack_spaces({p->side, p->wstream, p->wstream_d, p->gstream, p->ustream});
    if (p->d_ready) (void)hipEventDestroy(p->d_ready);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->side) (void)hipStreamDestroy(p->side);
    if (p->wstream) (void)hipStreamDestroy(p->wstream);
    if (p->ev_wjoin) (void)hipEventDestroy(p->ev_wjoin);
    for (hipEvent_t ev : p->ev_layer) if (ev) (void)hipEventDestroy(ev);
    if (p->wstream_d) (void)hipStreamDestroy(p->wstream_d);
    if (p->ev_wjoin_d) (void)hipEventDestroy(p->ev_wjoin_d);
    for (hipEvent_t ev : p->ev_layer_d) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->g_ready) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->g_free) if (ev) (void)hipEventDestroy(ev);
    if (p->d_free) (void)hipEventDestroy(p->d_free);
    for (auto& g : p->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    if (p->gstream) (void)hipStreamDestroy(p->gstream);
    if (p->ustream) (void)hipStreamDestroy(p->ustream);
    if (p->ev_gin) (void)hipEventDestroy(p->ev_gin);
    if (p->ev_gout) (void)hipEventDestroy(p->ev_gout);
    if (p->ev_ujoin) (void)hipEventDestroy(p->ev_ujoin);
    if (p->hp_dev) (void)hipFree(p->hp_dev);
    if (p->img_block) (void)hipFree(p->img_block);
    release_tables(p->tables);
    delete p;
}
extern "C" int vc_plan_set_math_mode(vc_plan* p, int mode) {
    VC_ARG(p, "vc_plan_set_math_mode: bad plan");
    VC_ARG(mode >= VC_MATH_FP32 && mode <= VC_MATH_BF16X3, "vc_plan_set_math_mode: mode %d (0 fp32, 1 bf16, 2 bf16x3)", mode);
    p->math = mode;
    return 0;
}
extern "C" int vc_plan_get_math_mode(const vc_plan* p) { return p ? p->math : VC_E_STATE; }
extern "C" int vc_plan_set_deterministic(vc_plan* p, int on) {
    VC_ARG(p, "vc_plan_set_deterministic: bad plan");
    p->det = on ? 1 : 0;
    return 0;
}
extern "C" int vc_plan_get_deterministic(const vc_plan* p) { return p ? p->det : VC_E_STATE; }

static long align64(long x) { return (x + 63) & ~63L; }
// save map: for every layer raw output, activation, stats -- offsets (floats) inside one group's save area
// actp: the activation once more as bf16 in channel-octet packed layout [B][C/8][L][8] (what the 16-bit data path of the
// bf16 mode reads, vc_conv_bf16.hip): written when the next layer's forward packs it, read again by that layer's filter
// gradient.  Half the size of act; unused in the other modes.
struct SaveMap { std::vector<long> raw, act, st, actp; long total; };
static SaveMap save_map(const NetDesc& n, const std::vector<LayerGeo>& geo, int B) {
    SaveMap m; long o = 0;
    for (size_t i = 0; i < n.layers.size(); ++i) {
        const Layer& L = n.layers[i];
        const bool last_plain = L.mode == M_PLAIN;
        m.raw.push_back(o); if (!last_plain) o = align64(o + geo[i].raw_ps * B);
        m.act.push_back(o); if (!last_plain) o = align64(o + geo[i].act_ps * B);
        m.st.push_back(o);
        if (L.mode == M_GATED_IN) o = align64(o + (long)B * 2 * L.C * 2);
        else if (L.mode == M_IN_RES) o = align64(o + (long)B * L.C * 2);
    }
    for (size_t i = 0; i < n.layers.size(); ++i) {
        m.actp.push_back(o);
        if (n.layers[i].mode != M_PLAIN) o = align64(o + (geo[i].act_ps * B + 1) / 2);
    }
    m.total = o;
    return m;
}
extern "C" int64_t vc_generator_save_floats(const vc_plan* p, int B) { return p ? save_map(p->G, p->gg, B).total : 0; }
extern "C" int64_t vc_discriminator_save_floats(const vc_plan* p, int B) { return p ? save_map(p->D, p->dg, B).total : 0; }
// backward scratch of one group: for every layer i [dY_i: gradient w.r.t. its raw conv output][dA_i: gradient w.r.t. its
// activation].  Per-layer buffers (not a rotating pair) so a filter gradient on the side stream can still read dY_i
// while the chain has moved on.
static long scratch_floats_net(const std::vector<LayerGeo>& geo, int B) {
    long o = 0;
    for (const LayerGeo& l : geo) o += align64(l.raw_ps * B) + align64(l.act_ps * B);
    return o;
}
extern "C" int64_t vc_scratch_floats(const vc_plan* p, int B) {
    if (!p) return 0;
    return std::max(scratch_floats_net(p->gg, B), scratch_floats_net(p->dg, B));
}
extern "C" int vc_discriminator_out_elems(const vc_plan* p) { return p ? p->d_hw : 0; }

// ------------------------------------------------------------------------------------
// forward / backward schedules (shared by generator and discriminator)
// ------------------------------------------------------------------------------------
// An earlier application of the same network whose filter gradients were deferred (the cycle hop): the first-hop
// backward reduces both applications into dW with ONE launch per layer (WgradArgs segments).
struct DeferredApp { const float* x; long x_gs; const float* save; long save_gs; const SaveMap* sm;
                     const float* scratch; long scr_gs; const float* dtop; long dtop_gs; int B; };
enum { W_NONE = 0, W_SELF = 1 };
struct NetCtx {
    const NetDesc* n; const std::vector<LayerGeo>* geo;
    const float* params; long par_gs;
    float* dparams; long dpar_gs;          // may be null
    float* save; long save_gs; SaveMap sm; // layout batch = Bl
    int G, B, Bl, math;
    hipStream_t s;
    hipStream_t ws = nullptr; const hipEvent_t* ev_layer = nullptr; hipEvent_t ev_wjoin = nullptr;   // wgrad stream (null: same stream)
    const std::vector<vc_plan::LayerImages>* images = nullptr;   // bf16 data path: this step's prebuilt filter images
    bool keep_packed = false;              // bf16 data path, inside a train step: forward leaves the packed inputs in the save area (SaveMap::actp)
    bool fuse_glu = false;                 // forward only: gated blocks without instance norm go out as one conv + GLU kernel
    int wmode = W_SELF;                    // W_NONE: leave the conv filter gradients to a later application (DeferredApp)
    bool wgrad_store = false;              // this backward's filter gradients are the step's only contribution to dparams' conv kernels
    const DeferredApp* deferred = nullptr; // reduce this earlier application in the same launches
    const int* ev_first = nullptr; const hipEvent_t* ev = nullptr; int n_ev = 0;   // record ev[k] once layer ev_first[k] is done (backward)
    const hipEvent_t* ev_free = nullptr;                                            // ... and ev_free[k] after its input gradient is issued
};
// scratch map of one group: dY[i], dA[i] per layer (dA[i] = gradient w.r.t. the activation (output) of layer i)
static void scratch_map(const std::vector<LayerGeo>& geo, int B, long* dY, long* dA) {
    long o = 0;
    for (size_t i = 0; i < geo.size(); ++i) { dY[i] = o; o += align64(geo[i].raw_ps * B); dA[i] = o; o += align64(geo[i].act_ps * B); }
}

static int net_forward(NetCtx& c, const float* x, long x_gs, float* y_plain, long y_plain_gs) {
    // one launch zeroes every raw conv output of this pass whose convolution will reduce over split-K with atomics
    bool plain_zeroed = true;
    {
        std::vector<std::pair<float*, long>> zl;
        for (size_t i = 0; i < c.n->layers.size(); ++i) {
            const Layer& L = c.n->layers[i];
            // the network's output (the plain last layer writes it directly) rides in the same list when it is 16-byte aligned
            float* dst = (L.mode == M_PLAIN) ? y_plain : c.save + c.sm.raw[i];
            const long dst_gs = (L.mode == M_PLAIN) ? y_plain_gs : c.save_gs;
            // the query sees the REAL destination: whatever the configuration derives from it (alignment, vec4) is what the
            // launch below will derive too, so "prezeroed" can never describe a different split-K decision than the launch takes
            bool split = false;
            VC_CHECK(conv_fwd(nullptr, 0, nullptr, 0, nullptr, 0, dst, dst_gs, c.G, c.B, (*c.geo)[i].g, c.math, c.s, false, &split));
            if (!split) continue;
            if (L.mode == M_PLAIN && (!aligned16(y_plain) || (y_plain_gs & 3))) { plain_zeroed = false; continue; }
            for (int g = 0; g < c.G; ++g) zl.push_back({dst + (long)g * dst_gs, (*c.geo)[i].raw_ps * c.B});
        }
        if (!zl.empty()) VC_CHECK(launch_zero_list(zl, c.s));
    }
    const float* in = x; long in_gs = x_gs;
    const float* blk_in = nullptr; long blk_in_gs = 0;
    for (size_t i = 0; i < c.n->layers.size(); ++i) {
        const Layer& L = c.n->layers[i]; const LayerGeo& lg = (*c.geo)[i];
        float* raw = (L.mode == M_PLAIN) ? y_plain : c.save + c.sm.raw[i];
        const long raw_gs = (L.mode == M_PLAIN) ? y_plain_gs : c.save_gs;
        if (L.mode == M_GATED_IN && i + 1 < c.n->layers.size() && c.n->layers[i + 1].mode == M_IN_RES) { blk_in = in; blk_in_gs = in_gs; }
        if (L.mode == M_GATED && wsconv_path(lg.g) && c.math != VC_MATH_BF16X3) {
            // gated block without instance norm (G.h1, module.py:161-163): convolution pair + GLU in one kernel.  Forward-only
            // passes (CycleGAN.test) never need the raw value / gate tensors again; a training pass keeps them for backward.
            WsConvArgs f{}; f.x = in; f.w = c.params + L.w; f.bias = c.params + L.b; f.y = c.save + c.sm.act[i];
            f.x_gs = in_gs; f.w_gs = c.par_gs; f.b_gs = c.par_gs; f.y_gs = c.save_gs; f.B = c.B; f.T = lg.g.W; f.Cout = L.cout; f.G = c.G;
            if (!c.fuse_glu) { f.raw = raw; f.raw_gs = raw_gs; }
            const int r = launch_gated_conv1d_glu(f, lg.g.Cin, lg.g.kw, c.math, c.s, 2.0 * c.G * c.B * lg.g.W * (double)L.cout * lg.g.kw * lg.g.Cin);
            if (r != -2) { VC_CHECK(r); in = f.y; in_gs = c.save_gs; continue; }
        }
        if (L.mode == M_GATED && c1_path(lg.g)) {
            // same for the single-input-channel pair (D.h1, module.py:201-203): GLU in the vector-ALU kernel's epilogue
            VC_CHECK(check_conv_geom(lg.g, c.B));
            C1Args f = c1_args(lg.g, c.G, c.B, c.math);
            f.x = in; f.x_gs = in_gs; f.w = c.params + L.w; f.w_gs = c.par_gs; f.bias = c.params + L.b; f.b_gs = c.par_gs;
            f.y = raw; f.y_gs = raw_gs; f.act = c.save + c.sm.act[i]; f.act_gs = c.save_gs;
            VC_CHECK(launch_c1_fwd(f, c.s));
            in = f.act; in_gs = c.save_gs; continue;
        }
        // gated block with instance norm and no pixel shuffle: the convolution may finish the block in its epilogue
        FuseNorm fn{};
        if (L.mode == M_GATED_IN && !L.ps) {
            fn.gamma = c.params + L.gamma; fn.beta = c.params + L.beta; fn.par_gs = c.par_gs;
            fn.stats = c.save + c.sm.st[i]; fn.st_gs = c.save_gs; fn.act = c.save + c.sm.act[i]; fn.act_gs = c.save_gs;
        }
        VC_CHECK(conv_fwd(in, in_gs, c.params + L.w, c.par_gs, c.params + L.b, c.par_gs, raw, raw_gs, c.G, c.B, lg.g, c.math, c.s,
                          /*prezeroed=*/L.mode != M_PLAIN || plain_zeroed, nullptr, c.images ? (*c.images)[i].fwd : nullptr,
                          (c.keep_packed && i > 0) ? (char*)(c.save + c.sm.actp[i - 1]) : nullptr, c.save_gs * 4,
                          fn.act ? &fn : nullptr));
        if (L.mode == M_PLAIN) break;
        if (fn.done) { in = fn.act; in_gs = c.save_gs; continue; }
        NormArgs a{};
        a.y = raw; a.y_gs = raw_gs; a.out = c.save + c.sm.act[i]; a.out_gs = c.save_gs;
        a.G = c.G; a.B = c.B; a.C = L.C; a.ps = L.ps;
        a.L = lg.g.OH * lg.g.OW * (L.ps ? 2 : 1);
        if (L.mode != M_GATED) {
            a.gamma = c.params + L.gamma; a.beta = c.params + L.beta; a.par_gs = c.par_gs;
            a.stats = c.save + c.sm.st[i]; a.st_gs = c.save_gs;
        }
        if (L.mode == M_IN_RES) { a.res = blk_in; a.res_gs = blk_in_gs; }
        VC_CHECK(launch_norm_fwd(L.mode, a, c.s));
        in = a.out; in_gs = c.save_gs;
    }
    return 0;
}

// dtop: gradient w.r.t. the network's last activation (for the generator: d o1 output, i.e. dy of the
// plain layer; for the discriminator: gradient w.r.t. the last GLU output, produced by the head).
static int net_backward(NetCtx& c, const float* x, long x_gs, const float* dtop, long dtop_gs,
                        float* dx, long dx_gs, int dx_acc, float* scratch, long scr_gs) {
    const int nl = (int)c.n->layers.size();
    VC_ARG(nl <= 24, "net_backward: more than 24 layers");
    long dYo[24], dAo[24];
    scratch_map(*c.geo, c.B, dYo, dAo);
    long dYd[24], dAd[24];
    if (c.deferred) scratch_map(*c.geo, c.deferred->B, dYd, dAd);
    hipStream_t ws = c.ws ? c.ws : c.s;       // the plan sizes its event pools for its own networks (18 / 4 layers)
    // one launch zeroes every dA that a split-K input-gradient will accumulate into
    {
        std::vector<std::pair<float*, long>> zl;
        for (int i = 1; i < nl; ++i) {
            bool split = false;
            VC_CHECK(conv_dgrad(nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, c.G, c.B, (*c.geo)[i].g, c.math, c.s, false, &split));
            if (split) for (int g = 0; g < c.G; ++g) zl.push_back({scratch + dAo[i - 1] + (long)g * scr_gs, (*c.geo)[i - 1].act_ps * c.B});
        }
        if (!zl.empty()) VC_CHECK(launch_zero_list(zl, c.s));
    }
    const float* dcur = dtop; long dcur_gs = dtop_gs;   // gradient w.r.t. the output activation of layer i
    const float* skip = nullptr; long skip_gs = 0;      // gradient flowing around a residual block (module.py:81)
    for (int i = nl - 1; i >= 0; --i) {
        const Layer& L = c.n->layers[i]; const LayerGeo& lg = (*c.geo)[i];
        const float* in = (i == 0) ? x : c.save + c.sm.act[i - 1];
        const long in_gs = (i == 0) ? x_gs : c.save_gs;
        const float* dyp; long dy_gs;
        if (L.mode == M_PLAIN) {
            dyp = dcur; dy_gs = dcur_gs;
            if (c.dparams) {
                if (ws != c.s) { VC_HIP(hipEventRecord(c.ev_layer[i], c.s)); VC_HIP(hipStreamWaitEvent(ws, c.ev_layer[i], 0)); }
                VC_CHECK(launch_bias_grad(dyp, dy_gs, c.dparams + L.b, c.dpar_gs, c.G, c.B, L.cout, lg.g.OH * lg.g.OW, ws));
            }
        } else {
            if (L.mode == M_IN_RES) { skip = dcur; skip_gs = dcur_gs; }
            NormArgs a{};
            a.y = c.save + c.sm.raw[i]; a.y_gs = c.save_gs; a.dout = dcur; a.do_gs = dcur_gs;
            a.dy = scratch + dYo[i]; a.dy_gs = scr_gs;
            a.G = c.G; a.B = c.B; a.C = L.C; a.ps = L.ps; a.L = lg.g.OH * lg.g.OW * (L.ps ? 2 : 1); a.det = vc_det_active() ? 1 : 0;
            if (L.mode != M_GATED) {
                a.gamma = c.params + L.gamma; a.beta = c.params + L.beta; a.par_gs = c.par_gs;
                a.stats_in = c.save + c.sm.st[i]; a.st_gs = c.save_gs;
                if (c.dparams) { a.dgamma = c.dparams + L.gamma; a.dbeta = c.dparams + L.beta; a.dpar_gs = c.dpar_gs; }
            }
            if (c.dparams) { a.dbias = c.dparams + L.b; a.dbias_gs = c.dpar_gs; }
            VC_CHECK(launch_norm_bwd(L.mode, a, c.s));
            dyp = scratch + dYo[i]; dy_gs = scr_gs;
            if (c.dparams && c.wmode != W_NONE && ws != c.s) { VC_HIP(hipEventRecord(c.ev_layer[i], c.s)); VC_HIP(hipStreamWaitEvent(ws, c.ev_layer[i], 0)); }
        }
        if (c.dparams && c.wmode != W_NONE) {
            WgradApp apps[2]; int na = 0;
            if (c.deferred) {
                const DeferredApp& d = *c.deferred;
                apps[na++] = WgradApp{i == 0 ? d.x : d.save + d.sm->act[i - 1], i == 0 ? d.x_gs : d.save_gs,
                                      L.mode == M_PLAIN ? d.dtop : d.scratch + dYd[i], L.mode == M_PLAIN ? d.dtop_gs : d.scr_gs, d.B};
                if (c.keep_packed && i > 0 && conv_fwd_packs_input(lg.g, d.B, c.math)) {   // d.B: the batch of that forward
                    apps[na - 1].xp = (const char*)(d.save + d.sm->actp[i - 1]); apps[na - 1].xp_gs = d.save_gs * 4;
                }
            }
            apps[na++] = WgradApp{in, in_gs, dyp, dy_gs, c.B};
            if (c.keep_packed && i > 0 && conv_fwd_packs_input(lg.g, c.Bl, c.math)) {       // c.Bl: the batch the forward ran (and laid out) with
                apps[na - 1].xp = (const char*)(c.save + c.sm.actp[i - 1]); apps[na - 1].xp_gs = c.save_gs * 4;
            }
            VC_CHECK(conv_wgrad(apps, na, c.dparams + L.w, c.dpar_gs, c.G, lg.g, c.math, ws, c.wgrad_store));
        }
        for (int k = 0; k < c.n_ev; ++k) if (c.ev_first[k] == i) VC_HIP(hipEventRecord(c.ev[k], ws));
        if (i == 0) {
            if (dx) VC_CHECK(conv_dgrad(dyp, dy_gs, c.params + L.w, c.par_gs, dx, dx_gs, nullptr, 0, dx_acc, c.G, c.B, lg.g, c.math, c.s,
                                        false, nullptr, (c.images && (*c.images)[i].has_dg) ? (*c.images)[i].dg : nullptr));
            for (int k = 0; k < c.n_ev; ++k) if (c.ev_free && c.ev_first[k] == i) VC_HIP(hipEventRecord(c.ev_free[k], c.s));
            break;        // falls through to the join below
        }
        // gradient w.r.t. this layer's input = activation of layer i-1; the head of a residual block adds the skip
        const bool block_head = (L.mode == M_GATED_IN && i + 1 < nl && c.n->layers[i + 1].mode == M_IN_RES);
        VC_CHECK(conv_dgrad(dyp, dy_gs, c.params + L.w, c.par_gs, scratch + dAo[i - 1], scr_gs,
                            block_head ? skip : nullptr, block_head ? skip_gs : 0, 0, c.G, c.B, lg.g, c.math, c.s, /*prezeroed=*/true,
                            nullptr, (c.images && (*c.images)[i].has_dg) ? (*c.images)[i].dg : nullptr));
        if (block_head) skip = nullptr;
        for (int k = 0; k < c.n_ev; ++k) if (c.ev_free && c.ev_first[k] == i) VC_HIP(hipEventRecord(c.ev_free[k], c.s));
        dcur = scratch + dAo[i - 1]; dcur_gs = scr_gs;
    }
    if (ws != c.s) { VC_HIP(hipEventRecord(c.ev_wjoin, ws)); VC_HIP(hipStreamWaitEvent(c.s, c.ev_wjoin, 0)); }   // join
    return 0;
}

static NetCtx gen_ctx(const vc_plan* p, const float* params, long par_gs, float* dparams, long dpar_gs,
                      float* save, int G, int B, int Bl, hipStream_t s) {
    NetCtx c{};
    c.n = &p->G; c.geo = &p->gg; c.params = params; c.par_gs = par_gs; c.dparams = dparams; c.dpar_gs = dpar_gs;
    c.sm = save_map(p->G, p->gg, Bl); c.save = save; c.save_gs = c.sm.total; c.G = G; c.B = B; c.Bl = Bl; c.s = s; c.math = p->math;
    return c;
}
static NetCtx disc_ctx(const vc_plan* p, const float* params, long par_gs, float* dparams, long dpar_gs,
                       float* save, int G, int B, int Bl, hipStream_t s) {
    NetCtx c{};
    c.n = &p->D; c.geo = &p->dg; c.params = params; c.par_gs = par_gs; c.dparams = dparams; c.dpar_gs = dpar_gs;
    c.sm = save_map(p->D, p->dg, Bl); c.save = save; c.save_gs = c.sm.total; c.G = G; c.B = B; c.Bl = Bl; c.s = s; c.math = p->math;
    return c;
}

// The step's prebuilt filter images (refresh_filter_images) describe `params` of the running step, two groups: only the
// step itself hands them to its network passes (StepImages scope); operator- and network-level calls pack per launch.
static thread_local const vc_plan* t_step_images = nullptr;
struct StepImages { StepImages(const vc_plan* p) { t_step_images = p; } ~StepImages() { t_step_images = nullptr; } };
static const std::vector<vc_plan::LayerImages>* step_images(const vc_plan* p, bool gen, int G) {
    if (t_step_images != p || G != 2 || p->math != VC_MATH_BF16) return nullptr;
    const auto& v = gen ? p->img_g : p->img_d;
    return v.empty() ? nullptr : &v;
}
static int gen_forward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                       float* y, long y_gs, float* save, int G, int B, hipStream_t s) {
    NetCtx c = gen_ctx(p, params, par_gs, nullptr, 0, save, G, B, B, s);
    c.images = step_images(p, true, G); c.keep_packed = c.images != nullptr;
    c.fuse_glu = p->inference;
    return net_forward(c, x, x_gs, y, y_gs);
}
struct BwdOpts { bool record_buckets = false, wgrad_store = false; int wmode = W_SELF; const DeferredApp* deferred = nullptr; };
static int gen_backward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                        const float* dy, long dy_gs, float* save, int Bl, float* dparams, long dpar_gs,
                        float* dx, long dx_gs, int dx_acc, float* scratch, long scr_gs, int G, int B, hipStream_t s,
                        const BwdOpts& o = BwdOpts()) {
    VC_ARG(!p->inference, "this plan was created for inference only (vc_plan_create_inference)");
    NetCtx c = gen_ctx(p, params, par_gs, dparams, dpar_gs, save, G, B, Bl, s);
    c.wgrad_store = o.wgrad_store; c.wmode = o.wmode; c.deferred = o.deferred;
    c.images = step_images(p, true, G); c.keep_packed = c.images != nullptr;
    if ((p->use_side & 2) && dparams && o.wmode != W_NONE) { c.ws = p->wstream; c.ev_layer = p->ev_layer; c.ev_wjoin = p->ev_wjoin; }
    if (o.record_buckets) { c.ev_first = p->bucket_first; c.ev = p->g_ready; c.ev_free = p->g_free; c.n_ev = vc_plan::kBuckets; }
    return net_backward(c, x, x_gs, dy, dy_gs, dx, dx_gs, dx_acc, scratch, scr_gs);
}
static int disc_forward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                        float* o, long o_gs, float* save, int G, int B, hipStream_t s) {
    VC_ARG(!p->inference, "this plan was created for inference only (vc_plan_create_inference)");
    NetCtx c = disc_ctx(p, params, par_gs, nullptr, 0, save, G, B, B, s);
    c.images = step_images(p, false, G); c.keep_packed = c.images != nullptr;
    VC_CHECK(net_forward(c, x, x_gs, nullptr, 0));
    const int last = (int)p->D.layers.size() - 1;
    return launch_head_fwd(save + c.sm.act[last], c.save_gs, params + p->D.head_w, params + p->D.head_b, par_gs,
                           o, o_gs, G, B, p->D.head_c, p->d_hw, s);
}
// o: the forward outputs [G][Bl][HW] (first B samples are used)
static int disc_backward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                         const float* o, long o_gs, const float* dout, long do_gs, float* save, int Bl,
                         float* dparams, long dpar_gs, float* dx, long dx_gs, int dx_acc,
                         float* scratch, long scr_gs, int G, int B, hipStream_t s, bool wgrad_store = false) {
    VC_ARG(!p->inference, "this plan was created for inference only (vc_plan_create_inference)");
    NetCtx c = disc_ctx(p, params, par_gs, dparams, dpar_gs, save, G, B, Bl, s);
    c.wgrad_store = wgrad_store;
    c.images = step_images(p, false, G); c.keep_packed = c.images != nullptr;
    if ((p->use_side & 4) && dparams) { c.ws = p->wstream_d; c.ev_layer = p->ev_layer_d; c.ev_wjoin = p->ev_wjoin_d; }
    const int last = (int)p->D.layers.size() - 1;
    // the head's input gradient goes where net_backward's scratch map puts dA[last]
    long dYo[24], dAo[24];
    scratch_map(p->dg, B, dYo, dAo);
    float* dtop = scratch + dAo[last];
    VC_CHECK(launch_head_bwd(save + c.sm.act[last], c.save_gs, params + p->D.head_w, par_gs, o, o_gs, dout, do_gs,
                             dtop, scr_gs, dparams ? dparams + p->D.head_w : nullptr,
                             dparams ? dparams + p->D.head_b : nullptr, dpar_gs, G, B, p->D.head_c, p->d_hw, s));
    return net_backward(c, x, x_gs, dtop, scr_gs, dx, dx_gs, dx_acc, scratch, scr_gs);
}

extern "C" int vc_generator_forward(const vc_plan* p, const float* params, int64_t par_gs,
                                    const float* x, int64_t x_gs, float* y, int64_t y_gs,
                                    float* save, int G, int B, void* stream) {
    VC_ARG(p && params && x && y && save && G > 0 && B > 0, "vc_generator_forward: bad args");
    DetScope det_scope(p->det);
    return gen_forward(p, params, par_gs, x, x_gs, y, y_gs, save, G, B, (hipStream_t)stream);
}
extern "C" int vc_generator_backward(const vc_plan* p, const float* params, int64_t par_gs,
                                     const float* x, int64_t x_gs, const float* dy, int64_t dy_gs,
                                     const float* save, float* dparams, int64_t dpar_gs,
                                     float* dx, int64_t dx_gs, int dx_accumulate,
                                     float* scratch, int G, int B, void* stream) {
    VC_ARG(p && params && x && dy && save && scratch && G > 0 && B > 0, "vc_generator_backward: bad args");
    DetScope det_scope(p->det);
    return gen_backward(p, params, par_gs, x, x_gs, dy, dy_gs, const_cast<float*>(save), B, dparams, dpar_gs,
                        dx, dx_gs, dx_accumulate, scratch, vc_scratch_floats(p, B), G, B, (hipStream_t)stream);
}
extern "C" int vc_discriminator_forward(const vc_plan* p, const float* params, int64_t par_gs,
                                        const float* x, int64_t x_gs, float* o, int64_t o_gs,
                                        float* save, int G, int B, void* stream) {
    VC_ARG(p && params && x && o && save && G > 0 && B > 0, "vc_discriminator_forward: bad args");
    DetScope det_scope(p->det);
    return disc_forward(p, params, par_gs, x, x_gs, o, o_gs, save, G, B, (hipStream_t)stream);
}
extern "C" int vc_discriminator_backward(const vc_plan* p, const float* params, int64_t par_gs,
                                         const float* x, int64_t x_gs, const float* dout, int64_t do_gs,
                                         const float* save, float* dparams, int64_t dpar_gs,
                                         float* dx, int64_t dx_gs, int dx_accumulate,
                                         float* scratch, int G, int B, void* stream) {
    VC_ARG(p && params && x && dout && save && scratch && G > 0 && B > 0, "vc_discriminator_backward: bad args");
    DetScope det_scope(p->det);
    // the sigmoid outputs are recomputed from the saved last activation (cheap) into the tail of scratch
    hipStream_t s = (hipStream_t)stream;
    SaveMap sm = save_map(p->D, p->dg, B);
    const int last = (int)p->D.layers.size() - 1;
    // the dY region of the scratch is free until the first norm backward writes it; o only has to survive the head
    // kernels (stream-ordered)
    float* o = scratch;
    const long o_gs = (long)B * p->d_hw;
    VC_CHECK(launch_head_fwd(save + sm.act[last], sm.total, params + p->D.head_w, params + p->D.head_b, par_gs,
                             o, o_gs, G, B, p->D.head_c, p->d_hw, s));
    return disc_backward(p, params, par_gs, x, x_gs, o, o_gs, dout, do_gs, const_cast<float*>(save), B, dparams, dpar_gs,
                         dx, dx_gs, dx_accumulate, scratch, vc_scratch_floats(p, B), G, B, s);
}

// ------------------------------------------------------------------------------------
// fused step (model.py:44-90 forward + every gradient; Adam is a separate call)
// ------------------------------------------------------------------------------------
extern "C" int64_t vc_flat_param_floats(const vc_plan* p) { return p ? 2 * p->G.total + 2 * p->D.total : 0; }
extern "C" int64_t vc_flat_offset(const vc_plan* p, int net) {
    if (!p) return -1;
    switch (net) { case 0: return 0; case 1: return p->G.total; case 2: return 2 * p->G.total; case 3: return 2 * p->G.total + p->D.total; }
    return -1;
}
struct StepMap { long X1, O1, CY, DX, DO, dO1, dCY, dDO_D, dDO_G, S1, S2, SD, SCR, SCRC, SCR2, ACC, total;
                 long ft2, ft1, hw2, hw1, s1_gs, s2_gs, sd_gs, scr_gs, scrc_gs; };
static StepMap step_map(const vc_plan* p, int B) {
    StepMap m{}; long o = 0;
    const long FT = (long)p->F * p->T;
    m.ft2 = 2 * B * FT; m.ft1 = B * FT; m.hw2 = 2L * B * p->d_hw; m.hw1 = (long)B * p->d_hw;
    auto take = [&](long n) { long r = o; o = align64(o + 2 * align64(n)); return r; };   // two groups each
    m.X1 = take(m.ft2); m.O1 = take(m.ft2); m.CY = take(m.ft1); m.DX = take(m.ft2); m.DO = take(m.hw2);
    m.dO1 = take(m.ft2); m.dCY = take(m.ft1); m.dDO_D = take(m.hw2); m.dDO_G = take(m.hw1);
    m.s1_gs = vc_generator_save_floats(p, 2 * B); m.S1 = take(m.s1_gs);
    m.s2_gs = vc_generator_save_floats(p, B);     m.S2 = take(m.s2_gs);
    m.sd_gs = vc_discriminator_save_floats(p, 2 * B); m.SD = take(m.sd_gs);
    m.scr_gs = vc_scratch_floats(p, 2 * B); m.SCR = take(m.scr_gs);   // backward calls with batch B use a smaller group stride inside it
    m.scrc_gs = scratch_floats_net(p->gg, B); m.SCRC = take(m.scrc_gs);   // cycle-hop backward: its dY buffers stay live until the
                                                                          // first-hop backward has reduced them into the filter gradients
    m.SCR2 = take(m.scr_gs);                      // scratch of the discriminator_loss backward (runs on the side stream)
    m.ACC = o; o += align64(VC_LOSS_BLOCKS * 10);     // per-block partial sums of the loss kernel
    m.total = o;
    return m;
}
extern "C" int64_t vc_step_workspace_floats(const vc_plan* p, int B) { return p ? step_map(p, B).total : 0; }

// Gradient ranges that are accumulated with atomics during the step (conv biases, instance-norm gamma/beta, the dense
// head, the Cin = 1 kernel, filter gradients whose launch splits K) -- together with the padding between tensors --
// are zeroed by ONE multi-buffer launch.  Every other filter gradient is written with plain stores by its single
// (two-application) launch, so the 479 MB gradient memset of a naive step is not needed.
static int zero_step_grads(const vc_plan* p, float* grads, int B, int Bbwd, hipStream_t s) {
    std::vector<std::pair<float*, long>> zl;
    const long PG = p->G.total, PD = p->D.total;
    for (int net = 0; net < 4; ++net) {
        const bool gen = net < 2;
        const NetDesc& n = gen ? p->G : p->D;
        const std::vector<LayerGeo>& geo = gen ? p->gg : p->dg;
        float* base = grads + (gen ? net * PG : 2 * PG + (net - 2) * PD);
        for (size_t i = 0; i < n.layers.size(); ++i) {
            const Layer& L = n.layers[i];
            const long next_w = (i + 1 < n.layers.size()) ? n.layers[i + 1].w : (n.head_w >= 0 ? n.head_w : n.total);
            long lo = L.b;
            // does the filter gradient of this layer accumulate?  (same decision conv_wgrad will take)
            WgradApp apps[2] = {{nullptr, 0, nullptr, 0, gen ? B : 2 * B}, {nullptr, 0, nullptr, 0, Bbwd}};
            bool acc = true;
            VC_CHECK(conv_wgrad(apps, gen ? 2 : 1, nullptr, 0, 2, geo[i].g, p->math, s, /*store=*/true, &acc));
            if (acc) lo = L.w;
            zl.push_back({base + lo, next_w - lo});
        }
        if (n.head_w >= 0) zl.push_back({base + n.head_w, n.total - n.head_w});
    }
    return launch_zero_list(zl, s);
}

// hp != nullptr: lambda_cycle / lambda_identity are read from hp[0..1] on the device (a captured step is replayed with new
// values); the host copies still decide the schedule (lambda_identity == 0 skips the identity half).
// bf16 data path: build every layer's filter images from the step's fp32 weights (both generators / both discriminators
// as the two groups of one launch).  Runs on the step's main stream before anything forks from it.
static int refresh_filter_images(const vc_plan* p, const float* params, hipStream_t s) {
    const long PG = p->G.total, PD = p->D.total;
    struct Job { const float* w; long w_gs; const int* wsrc; int a_ms, M, kp; size_t off, gs; };
    std::vector<Job> jobs; size_t tot = 256;        // offsets are never 0, so 0 / nullptr can mean 'no image'
    const bool first = p->img_g.empty();
    std::vector<vc_plan::LayerImages> ig(p->G.layers.size()), id(p->D.layers.size());
    for (int net = 0; net < 2; ++net) {
        const NetDesc& n = net ? p->D : p->G; const auto& geo = net ? p->dg : p->gg; auto& img = net ? id : ig;
        const float* base = net ? params + 2 * PG : params; const long gs = net ? PD : PG;
        for (size_t i = 0; i < n.layers.size(); ++i) {
            const Geom& g = geo[i].g;
            if (c1_path(g)) continue;
            Table t; VC_CHECK(get_fwd_table(g, &t));
            if (t.koct > 0 && g.Cout >= 64) {
                const size_t a = align256((size_t)g.Cout * t.koct * 16);
                jobs.push_back(Job{base + n.layers[i].w, gs, t.wsrc, 1, g.Cout, t.koct * 8, tot, a});
                img[i].fwd = (const char*)tot; tot += 2 * a;
            }
            bool all = g.Cin >= 64; size_t off[4]; Job cj[4]; int nc = 0;
            for (int ph = 0; ph < g.sh && all; ++ph) for (int pw = 0; pw < g.sw && all; ++pw) {
                Table td; VC_CHECK(get_dgrad_table(g, ph, pw, &td));
                if (td.koct <= 0) { all = false; break; }
                const size_t a = align256((size_t)g.Cin * td.koct * 16);
                cj[nc] = Job{base + n.layers[i].w, gs, td.wsrc, g.Cout, g.Cin, td.koct * 8, tot, a}; off[nc++] = tot; tot += 2 * a;
            }
            if (all && nc) { for (int k = 0; k < nc; ++k) { jobs.push_back(cj[k]); img[i].dg[k] = (const char*)off[k]; } img[i].has_dg = true; }
            else if (nc) tot = off[0];            // give the partial reservation back
        }
    }
    if (first) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            vc_set_error("bf16 data path: run one eager step before capturing (the filter images are allocated on first use)");
            return VC_E_STATE;
        }
        VC_HIP(hipMalloc((void**)&p->img_block, tot + 256));
        for (auto* v : {&ig, &id})
            for (auto& li : *v) {
                if (li.fwd) li.fwd = p->img_block + (size_t)li.fwd;
                for (int k = 0; k < 4; ++k) if (li.dg[k]) li.dg[k] = p->img_block + (size_t)li.dg[k];
            }
        p->img_g = ig; p->img_d = id;
    }
    std::vector<PackFilterJob> pj;
    for (const Job& j : jobs)
        pj.push_back(PackFilterJob{j.w, (unsigned short*)(p->img_block + j.off), j.wsrc, j.w_gs, (long)j.gs / 2, j.a_ms, j.M, j.kp, 0, 0});
    return launch_pack_filter_multi(pj, 2, s);
}

static int step_grads_impl(const vc_plan* p, const float* params, float* grads,
                           const float* A, const float* Bm, int batch,
                           float lambda_cycle, float lambda_identity, const float* hp,
                           float* losses, float* gen_A, float* gen_B,
                           float* ws, int64_t ws_floats, void* stream) {
    VC_ARG(p && params && grads && A && Bm && losses && ws && batch > 0, "vc_cyclegan_step_grads: bad args");
    VC_ARG(!p->inference, "this plan was created for inference only (vc_plan_create_inference)");
    VC_ARG(aligned16(grads) && aligned16(ws), "vc_cyclegan_step_grads: grads and workspace must be 16-byte aligned");
    DetScope det_scope(p->det);
    hipStream_t s = (hipStream_t)stream;
    const int B = batch;
    const StepMap m = step_map(p, B);
    if (ws_floats < m.total) { vc_set_error("workspace too small: %lld < %lld floats", (long long)ws_floats, (long long)m.total); return VC_E_NOMEM; }
    auto gs = [](long n) { return align64(n); };     // group stride of a two-group region
    const long PG = p->G.total, PD = p->D.total;
    const long ft1 = m.ft1;
    float *X1 = ws + m.X1, *O1 = ws + m.O1, *CY = ws + m.CY, *DX = ws + m.DX, *DO = ws + m.DO;
    float *dO1 = ws + m.dO1, *dCY = ws + m.dCY, *dDO_D = ws + m.dDO_D, *dDO_G = ws + m.dDO_G;
    const long X1g = gs(m.ft2), O1g = gs(m.ft2), CYg = gs(m.ft1), DXg = gs(m.ft2), DOg = gs(m.hw2);
    const long dO1g = gs(m.ft2), dCYg = gs(m.ft1), dDDg = gs(m.hw2), dDGg = gs(m.hw1);
    float *S1 = ws + m.S1, *S2 = ws + m.S2, *SD = ws + m.SD, *SCR = ws + m.SCR, *SCRC = ws + m.SCRC;
    const long scr_gs = align64(m.scr_gs), scrc_gs = align64(m.scrc_gs);
    // the save/scratch areas were sized with take(): group stride = align64(size)
    // (save_map totals are already multiples of 64, so align64 is the identity there)
    // With lambda_identity == 0 (train.py:98-99 after 10 000 iterations) the identity half carries an exactly-zero
    // seed gradient: only the generation half (the first B samples of every saved tensor) is propagated.
    const int Bbwd = (lambda_identity == 0.0f) ? B : 2 * B;

    hipStream_t s2 = (p->use_side & 1) ? p->side : s;
    if (p->math == VC_MATH_BF16) VC_CHECK(refresh_filter_images(p, params, s));
    StepImages scope(p->math == VC_MATH_BF16 ? p : nullptr);
    // 1. first-hop inputs: g0 = A2B on [A;B], g1 = B2A on [B;A]   (model.py:44,51 / 47,50)
    VC_CHECK(launch_copy_list({{X1, A, ft1}, {X1 + ft1, Bm, ft1}, {X1 + X1g, Bm, ft1}, {X1 + X1g + ft1, A, ft1}}, s));
    // 0. zero the accumulated parts of the flat gradient on the side stream, under the first-hop forward (nothing touches
    //    them before step 7; the fork/join pairs below order it before every consumer).  The fork comes AFTER the first
    //    work on `stream`: under hipGraph capture, an event recorded on a stream that has captured nothing yet does not
    //    pull the waiting stream into the capture (ROCm 7.2; the replayed graph then crashes at hipStreamEndCapture).
    if (s2 != s) { VC_HIP(hipEventRecord(p->ev_fork, s)); VC_HIP(hipStreamWaitEvent(s2, p->ev_fork, 0)); }
    VC_CHECK(zero_step_grads(p, grads, B, Bbwd, s2));
    // 2. O1[g0] = [gen_B; id_B], O1[g1] = [gen_A; id_A]
    VC_CHECK(gen_forward(p, params, PG, X1, X1g, O1, O1g, S1, 2, 2 * B, s));
    // Two pairs of chains are independent and run side by side (main stream s / side stream s2, fork+join by events):
    //   forward : [3. cycle hop G forward]            ||  [4-5. D inputs + D forward]
    //   backward: [8-10. generator_loss backward]     ||  [7. discriminator_loss backward]
    auto fork = [&]() -> int { if (s2 != s) { VC_HIP(hipEventRecord(p->ev_fork, s)); VC_HIP(hipStreamWaitEvent(s2, p->ev_fork, 0)); } return 0; };
    auto join = [&]() -> int { if (s2 != s) { VC_HIP(hipEventRecord(p->ev_join, s2)); VC_HIP(hipStreamWaitEvent(s, p->ev_join, 0)); } return 0; };
    VC_CHECK(fork());
    // 3. cycle hop: g0 = A2B(gen_A) = cycle_B, g1 = B2A(gen_B) = cycle_A   (model.py:45,48)
    VC_CHECK(gen_forward(p, params, PG, O1 + O1g, -O1g, CY, CYg, S2, 2, B, s));
    // 4. discriminator inputs: g0 = D_A on [gen_A; A], g1 = D_B on [gen_B; B]   (model.py:53-54,75-78); the caller's copies of
    //    generation_A / generation_B (model.py:112) ride in the same launch, on the side stream (joined before the step returns)
    {
        std::vector<CopyItem> items{{DX, O1 + O1g, ft1}, {DX + ft1, A, ft1}, {DX + DXg, O1, ft1}, {DX + DXg + ft1, Bm, ft1}};
        if (gen_A) items.push_back({gen_A, O1 + O1g, ft1});
        if (gen_B) items.push_back({gen_B, O1, ft1});
        VC_CHECK(launch_copy_list(items, s2));
    }
    // 5. D forward (the fake half doubles as D(fake) of the D sub-step: D weights are unchanged by the G update)
    VC_CHECK(disc_forward(p, params + 2 * PG, PD, DX, DXg, DO, DOg, SD, 2, 2 * B, s2));
    VC_CHECK(join());
    // 6. losses + seed gradients
    LossArgs la{};
    la.A = A; la.Bm = Bm; la.out1 = O1; la.out1_gs = O1g; la.cyc = CY; la.cyc_gs = CYg; la.dout = DO; la.dout_gs = DOg;
    la.d_out1 = dO1; la.d_out1_gs = dO1g; la.d_cyc = dCY; la.d_cyc_gs = dCYg;
    la.d_dout_D = dDO_D; la.d_dout_D_gs = dDDg; la.d_dout_G = dDO_G; la.d_dout_G_gs = dDGg;
    la.acc = ws + m.ACC; la.losses = losses; la.B = B; la.FT = p->F * p->T; la.HW = p->d_hw;
    la.lambda_cycle = lambda_cycle; la.lambda_identity = lambda_identity; la.hp = hp;
    VC_CHECK(launch_loss(la, s));
    VC_CHECK(fork());
    // 7. discriminator_loss backward over [fake; real] -> D gradients (model.py:94,107)   [side stream, own scratch]
    VC_CHECK(disc_backward(p, params + 2 * PG, PD, DX, DXg, DO, DOg, dDO_D, dDDg, SD, 2 * B,
                           grads + 2 * PG, PD, nullptr, 0, 0, ws + m.SCR2, scr_gs, 2, 2 * B, s2, /*wgrad_store=*/true));
    VC_HIP(hipEventRecord(p->d_ready, s2));
    // 8. adversarial generator terms: gradient through D's inputs only (var_list = generator_vars, model.py:95,108)
    //    g0 = D_A -> d gen_A = dO1[g1][0:B]; g1 = D_B -> d gen_B = dO1[g0][0:B]
    VC_CHECK(disc_backward(p, params + 2 * PG, PD, DX, DXg, DO, DOg, dDO_G, dDGg, SD, 2 * B,
                           nullptr, 0, dO1 + dO1g, -dO1g, 0, SCR, scr_gs, 2, B, s));
    VC_HIP(hipEventRecord(p->d_free, s));     // last read of the discriminator weights in this step
    // 9. cycle hop backward: g0 = A2B on gen_A -> adds into d gen_A; g1 = B2A on gen_B -> d gen_B.  Its filter
    //    gradients are NOT launched here: step 10 reduces both applications of every layer in one launch.
    BwdOpts oc; oc.wmode = W_NONE;
    VC_CHECK(gen_backward(p, params, PG, O1 + O1g, -O1g, dCY, dCYg, S2, B, grads, PG,
                          dO1 + dO1g, -dO1g, 1, SCRC, scrc_gs, 2, B, s, oc));
    // 10. first-hop backward over [generation; identity] (+ the deferred cycle-hop filter gradients)
    const SaveMap sm2 = save_map(p->G, p->gg, B);
    const DeferredApp cyc{O1 + O1g, -O1g, S2, sm2.total, &sm2, SCRC, scrc_gs, dCY, dCYg, B};
    BwdOpts o1; o1.record_buckets = true; o1.wgrad_store = true; o1.deferred = &cyc;
    VC_CHECK(gen_backward(p, params, PG, X1, X1g, dO1, dO1g, S1, 2 * B, grads, PG,
                          nullptr, 0, 0, SCR, scr_gs, 2, Bbwd, s, o1));
    VC_CHECK(join());      // the caller's next work on `stream` (Adam) sees the discriminator gradients
    return 0;
}

extern "C" int vc_cyclegan_step_grads(const vc_plan* p, const float* params, float* grads,
                                      const float* A, const float* Bm, int batch,
                                      float lambda_cycle, float lambda_identity,
                                      float* losses, float* gen_A, float* gen_B,
                                      float* ws, int64_t ws_floats, void* stream) {
    return step_grads_impl(p, params, grads, A, Bm, batch, lambda_cycle, lambda_identity, nullptr, losses, gen_A, gen_B, ws, ws_floats, stream);
}

// Gradient exchange + ApplyAdam per segment (see vc_step_exchange_and_update in vc_comm.hip); hp: step sizes from device memory
int step_update_impl(const vc_plan* p, const vc_comm* comm, float* params, float* grads, float* m, float* v,
                     float alpha_g, float alpha_d, const float* hp, float beta1, float beta2, float eps, hipStream_t us);

// The whole train step (model.py:110-125: forward, every gradient, both ApplyAdam updates) as ONE hipGraph launch.
// The reference trains at mini_batch_size = 1 (train.py:16), where a step is ~330 small launches and the host is the
// bottleneck; the captured graph replays them with one call.  Values that change from step to step (lambda_cycle,
// lambda_identity, both Adam step sizes) live in a 4-float device buffer written just before the replay; what changes the
// SCHEDULE (batch, lambda_identity == 0, any pointer) selects a different cached graph.  Single GPU (no collective).
extern "C" int vc_train_step_graph(vc_plan* p, float* params, float* grads, float* m, float* v,
                                   const float* A, const float* Bm, int batch,
                                   float lambda_cycle, float lambda_identity, float alpha_g, float alpha_d,
                                   float beta1, float beta2, float eps,
                                   float* losses, float* gen_A, float* gen_B,
                                   float* ws, int64_t ws_floats, void* stream) {
    VC_ARG(p && params && grads && m && v && A && Bm && losses && ws && batch > 0, "vc_train_step_graph: bad args");
    VC_ARG(!p->inference, "this plan was created for inference only (vc_plan_create_inference)");
    hipStream_t user = (hipStream_t)stream;
    auto make_key = [&]() {
        return std::array<const void*, 12>{params, grads, m, v, A, Bm, losses, gen_A, gen_B, ws,
                                           (const void*)(intptr_t)(batch * 16 + (lambda_identity == 0.0f ? 1 : 0) * 8 + (p->use_side & 7)),
                                           (const void*)(intptr_t)(p->math + 16 * p->det +
                                                                   64 * ws_generation({p->gstream, p->side, p->wstream, p->wstream_d, p->ustream}))};
    };
    std::array<const void*, 12> key = make_key();
    vc_plan::StepGraph* sg = nullptr;
    for (auto& g : p->graphs) if (g.key == key) { sg = &g; break; }
    if (!sg) {
        // ROCm 7.2: a stream forked from an already-forked stream (the discriminator's filter-gradient stream hangs off
        // the side stream) crashes hipStreamEndCapture; the captured schedule keeps those launches on the side stream
        const int overlap = p->use_side;
        p->use_side &= ~4;
        if (p->math == VC_MATH_BF16) {
            // the 16-bit data path keeps packed operands in per-stream workspaces and per-plan filter images that are
            // allocated on first use -- not allowed while capturing: run the schedule once, eagerly, on the capture streams
            // (it writes only gradients, losses and the generation outputs, all of which the replay overwrites)
            int r0 = step_grads_impl(p, params, grads, A, Bm, batch, lambda_cycle, lambda_identity, nullptr, losses, gen_A, gen_B,
                                     ws, ws_floats, p->gstream);
            const hipError_t e0 = hipStreamSynchronize(p->gstream);
            if (r0 || e0 != hipSuccess) { p->use_side = overlap; if (r0) return r0; VC_HIP(e0); }
        }
        key = make_key();                       // the eager run may have grown a workspace: the graph captured below holds the NEW addresses
        VC_HIP(hipStreamBeginCapture(p->gstream, hipStreamCaptureModeThreadLocal));
        int r = step_grads_impl(p, params, grads, A, Bm, batch, lambda_cycle, lambda_identity, p->hp_dev, losses, gen_A, gen_B,
                                ws, ws_floats, p->gstream);
        if (!r) r = step_update_impl(p, nullptr, params, grads, m, v, 0.f, 0.f, p->hp_dev, beta1, beta2, eps, p->ustream);
        if (!r) { hipError_t e = hipEventRecord(p->ev_ujoin, p->ustream); if (e == hipSuccess) e = hipStreamWaitEvent(p->gstream, p->ev_ujoin, 0); r = (int)e; }
        hipGraph_t graph = nullptr;
        const hipError_t ec = hipStreamEndCapture(p->gstream, &graph);
        p->use_side = overlap;
        if (r) { if (graph) (void)hipGraphDestroy(graph); return r; }
        VC_HIP(ec);
        hipGraphExec_t exec = nullptr;
        VC_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        if (p->graphs.size() >= 8) {            // few schedules exist in practice (lambda_identity on / off); bound the cache
            (void)hipGraphExecDestroy(p->graphs.front().exec); (void)hipGraphDestroy(p->graphs.front().graph);
            p->graphs.erase(p->graphs.begin());
        }
        p->graphs.push_back(vc_plan::StepGraph{key, graph, exec});
        sg = &p->graphs.back();
    }
    // order: caller's stream -> graph stream -> caller's stream
    VC_HIP(hipEventRecord(p->ev_gin, user));
    VC_HIP(hipStreamWaitEvent(p->gstream, p->ev_gin, 0));
    VC_CHECK(launch_set4(p->hp_dev, lambda_cycle, lambda_identity, alpha_g, alpha_d, p->gstream));
    VC_HIP(hipGraphLaunch(sg->exec, p->gstream));
    VC_HIP(hipEventRecord(p->ev_gout, p->gstream));
    VC_HIP(hipStreamWaitEvent(user, p->ev_gout, 0));
    return 0;
}

extern "C" int vc_plan_set_overlap(vc_plan* p, int enabled) {
    VC_ARG(p, "vc_plan_set_overlap: bad plan");
    p->use_side = enabled == 1 ? 7 : ((enabled >> 3) & 7);   // 0 / 1: the documented switch; 8 * bits selects single mechanisms (debugging)
    return 0;
}
extern "C" int vc_step_num_g_buckets(const vc_plan* p) { return p ? vc_plan::kBuckets : 0; }
extern "C" int vc_step_g_bucket(const vc_plan* p, int k, int64_t* lo, int64_t* hi) {
    VC_ARG(p && k >= 0 && k < vc_plan::kBuckets && lo && hi, "vc_step_g_bucket: bad args");
    const int first = p->bucket_first[k];
    const int last = (k == 0 ? (int)p->G.layers.size() : p->bucket_first[k - 1]) - 1;
    *lo = p->G.layers[first].w;
    *hi = (last + 1 < (int)p->G.layers.size()) ? p->G.layers[last + 1].w : p->G.total;
    return 0;
}
extern "C" int vc_step_wait_g_bucket(const vc_plan* p, int k, void* other_stream) {
    VC_ARG(p && k >= 0 && k < vc_plan::kBuckets, "vc_step_wait_g_bucket: bad args");
    VC_HIP(hipStreamWaitEvent((hipStream_t)other_stream, p->g_ready[k], 0));
    VC_HIP(hipStreamWaitEvent((hipStream_t)other_stream, p->g_free[k], 0));
    return 0;
}
extern "C" int vc_step_wait_d_grads(const vc_plan* p, void* other_stream) {
    VC_ARG(p, "vc_step_wait_d_grads: bad plan");
    VC_HIP(hipStreamWaitEvent((hipStream_t)other_stream, p->d_ready, 0));
    VC_HIP(hipStreamWaitEvent((hipStream_t)other_stream, p->d_free, 0));
    return 0;
}
