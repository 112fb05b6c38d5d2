// vc_tables.hip -- errors, CRC-32C, TF-SAME geometry, K tables, thread modes (host side of libvcgan_hip.so, part 1 of 4: vc_host.h)
#include "vc_host.h"
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vc_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* vc_last_error(void) { return g_err; }
// 110: vc_gated_conv1d_glu_prepack / _fwd_packed / _image_bytes, vc_build_id (round 5; round 4 changed vc_gated_conv1d_glu_fwd's
// arity without a bump -- ADVICE r4).  _lib.py asserts the version it was written against.
// 111: vc_tune_conv_bf16_spec (testing header; round 5: the wave-specialised bf16 convolutions and their 256-thread control)
// 112: vc_tune_bf16_pixel_octets (testing header: dy as pixel octets from the norm backward, and the pack-launch control)
extern "C" int vc_version(void) { return 116; }
#ifndef VC_BUILD_ID
#define VC_BUILD_ID "unstamped"
#endif
extern "C" const char* vc_build_id(void) { return VC_BUILD_ID; }
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

// ------------------------------------------------------------------------------------
// geometry: TF 'SAME' padding (extra pad goes right/bottom)
// ------------------------------------------------------------------------------------
static void same_pad(int n, int k, int s, int* out, int* lo) {
    *out = (n + s - 1) / s;
    int total = (*out - 1) * s + k - n; if (total < 0) total = 0;
    *lo = total / 2;
}
Geom make_geom(int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw) {
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
static std::mutex g_tab_mu;
static std::map<TabKey, Table> g_tabs;

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
int get_fwd_table(const Geom& g, Table* out, std::vector<TabKey>* owner) {
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
int get_dgrad_table(const Geom& g, int ph, int pw, Table* out, std::vector<TabKey>* owner) {
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
void release_tables(std::vector<TabKey>& keys) {
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
    if (mode < VC_MATH_FP32 || mode > VC_MATH_F16) { vc_set_error("vc_set_math_mode: mode %d (0 fp32, 1 bf16, 2 bf16x3, 3 f16: gated conv1d + GLU forward only)", mode); return VC_E_ARG; }
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

int vc_thread_math() { return t_math; }
int vc_thread_det() { return t_det; }
