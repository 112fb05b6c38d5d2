// vc_net.hip -- network descriptions, parameter layout, plans, forward / backward schedules (part 3 of 4: vc_host.h)
#include "vc_host.h"
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------
// network descriptions + parameter layout
// ------------------------------------------------------------------------------------
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
    VC_ARG(vc_thread_math() != VC_MATH_F16, "vc_plan_create: the calling thread's math mode 3 (f16 operands) is built for the gated conv1d + GLU forward only; a plan takes 0, 1 or 2");
    vc_plan* p = new vc_plan();
    p->F = num_features; p->T = n_frames; p->inference = inference; p->math = vc_thread_math(); p->det = vc_thread_det();
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
            if (!inference && winoh_eligible(l.g, VC_MATH_FP32, 1 << 20)) {      // D.d3: the (1 x kw, stride 1 x sw) tables of the F(3,6)-along-H planes
                const Geom g1 = make_geom(l.g.Cin, l.g.H / 3, l.g.W, l.g.Cout, 1, l.g.kw, 1, l.g.sw);
                r = r ? r : get_fwd_table(g1, &t, &p->tables);
                for (int pw = 0; pw < g1.sw && !r; ++pw) r = get_dgrad_table(g1, 0, pw, &t, &p->tables);
            }
            if (!inference && v == &p->gg && wino_eligible(l.g, VC_MATH_FP32, 1 << 20)) {      // the 1 x 1 tables of the Winograd route's channel GEMMs
                const Geom g1 = wino_gemm_geom(l.g);
                r = r ? r : get_fwd_table(g1, &t, &p->tables);
                r = r ? r : get_dgrad_table(g1, 0, 0, &t, &p->tables);
            }
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
    for (int k = 0; k < 24 && e == hipSuccess; ++k) { e = hipEventCreateWithFlags(&p->gl_ready[k], hipEventDisableTiming); if (e == hipSuccess) e = hipEventCreateWithFlags(&p->gl_free[k], hipEventDisableTiming); }
    for (int k = 0; k < 8 && e == hipSuccess; ++k) {
        e = hipEventCreateWithFlags(&p->dl_ready[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->dl_free_d[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->dl_free_g[k], hipEventDisableTiming);
    }
    for (int k = 0; k < vc_plan::kWinoChunks && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&p->ev_wino[k], hipEventDisableTiming);
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
    for (auto* arr : {p->gl_ready, p->gl_free}) for (int k = 0; k < 24; ++k) if (arr[k]) (void)hipEventDestroy(arr[k]);
    for (auto* arr : {p->dl_ready, p->dl_free_d, p->dl_free_g}) for (int k = 0; k < 8; ++k) if (arr[k]) (void)hipEventDestroy(arr[k]);
    for (hipEvent_t ev : p->ev_wino) if (ev) (void)hipEventDestroy(ev);
    if (p->wino_block) (void)hipFree(p->wino_block);
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

// save map: for every layer raw output, activation, stats -- offsets (floats) inside one group's save area
// actp: the activation once more as bf16 in channel-octet packed layout [B][C/8][L][8] (what the 16-bit data path of the
// bf16 mode reads, vc_conv_bf16.hip): written when the next layer's forward packs it, read again by that layer's filter
// gradient.  Half the size of act; unused in the other modes.
SaveMap save_map(const NetDesc& n, const std::vector<LayerGeo>& geo, int B) {
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
// + one region for the bf16 channel-octet copy of the dY in flight (bf16 data path: norm backward emits it, the layer's input gradient
// reads it, the next layer's norm backward overwrites it -- all on one stream): half the floats of the largest dY
static long scratch_dyp_floats(const std::vector<LayerGeo>& geo, int B) {
    long m = 0;
    for (const LayerGeo& l : geo) m = std::max(m, align64((l.raw_ps * B + 1) / 2));
    return m;
}
// + per layer the filter-gradient GEMM's dY operand (bf16 pixel octets, both applications of the layer stacked: the cycle-hop pass writes
// its B samples' part here, the first-hop pass adds up to 2 B more and reduces: room for 3 B), written by the norm backward of each pass
// (bf16 mode only: the fp32 workspace does not carry them)
static long scratch_dyq_floats(const LayerGeo& l, int B) { return align64((3 * l.raw_ps * B + 1) / 2); }
long scratch_floats_net(const std::vector<LayerGeo>& geo, int B, int math) {
    long o = 0;
    for (const LayerGeo& l : geo) o += align64(l.raw_ps * B) + align64(l.act_ps * B) + (math == VC_MATH_BF16 ? scratch_dyq_floats(l, B) : 0);
    return o + scratch_dyp_floats(geo, B);
}
extern "C" int64_t vc_scratch_floats(const vc_plan* p, int B) {
    if (!p) return 0;
    return std::max(scratch_floats_net(p->gg, B, p->math), scratch_floats_net(p->dg, B, p->math));
}
extern "C" int vc_discriminator_out_elems(const vc_plan* p) { return p ? p->d_hw : 0; }

// ------------------------------------------------------------------------------------
// forward / backward schedules (shared by generator and discriminator)
// ------------------------------------------------------------------------------------
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
    unsigned* q_ready_out = nullptr;       // W_NONE passes report which layers' dY they left as pixel octets (BwdOpts::q_ready_out)
    unsigned* dy_dropped_out = nullptr; bool drop_dy = false;   // BwdOpts::drop_fp32_dy
    const hipEvent_t* wino_ev = nullptr; const int* wino_chunk = nullptr;   // fp32, inside a train step: `images` holds the Winograd-domain filters; layer i's are valid once wino_ev[wino_chunk[i]] has passed
    bool prezeroed = false;                // the split-K dA buffers of this pass were zeroed ahead of time (backward_prezero)
    int wmode = W_SELF;                    // W_NONE: leave the conv filter gradients to a later application (DeferredApp)
    bool wgrad_store = false;              // this backward's filter gradients are the step's only contribution to dparams' conv kernels
    const DeferredApp* deferred = nullptr; // reduce this earlier application in the same launches
    const int* ev_first = nullptr; const hipEvent_t* ev = nullptr; int n_ev = 0;   // record ev[k] once layer ev_first[k] is done (backward)
    const hipEvent_t* ev_free = nullptr;                                            // ... and ev_free[k] after its input gradient is issued
    const hipEvent_t* lay_ready = nullptr; const hipEvent_t* lay_free = nullptr;    // the same per layer (backward): [i] after layer i's filter gradient / input gradient is issued
};
// scratch map of one group: dY[i], dA[i] per layer (dA[i] = gradient w.r.t. the activation (output) of layer i)
static long scratch_map(const std::vector<LayerGeo>& geo, int B, long* dY, long* dA, long* dYq = nullptr) {      // returns the offset of the packed-dY region
    long o = 0;
    for (size_t i = 0; i < geo.size(); ++i) { dY[i] = o; o += align64(geo[i].raw_ps * B); dA[i] = o; o += align64(geo[i].act_ps * B); }
    if (dYq) {                                           // bf16 mode; behind the channel-octet region: the layers' stacked pixel-octet buffers
        long q = o + scratch_dyp_floats(geo, B);
        for (size_t i = 0; i < geo.size(); ++i) { dYq[i] = q; q += scratch_dyq_floats(geo[i], B); }
    }
    return o;
}

// testing (vc_testing_describe_step): the named sub-regions of a save area / a backward scratch, as "name offset floats" lines
void describe_save(std::string& out, const char* tag, const NetDesc& n, const std::vector<LayerGeo>& geo, int B, long base, long gs, int G) {
    const SaveMap m = save_map(n, geo, B);
    char line[256];
    for (int g = 0; g < G; ++g)
        for (size_t i = 0; i < n.layers.size(); ++i) {
            const Layer& L = n.layers[i];
            const long st = L.mode == M_GATED_IN ? (long)B * 2 * L.C * 2 : L.mode == M_IN_RES ? (long)B * L.C * 2 : 0;
            const bool lp = L.mode == M_PLAIN;
            snprintf(line, sizeof line, "%s.g%d.L%zu.raw %ld %ld\n", tag, g, i, base + g * gs + m.raw[i], lp ? 0 : geo[i].raw_ps * B); out += line;
            snprintf(line, sizeof line, "%s.g%d.L%zu.act %ld %ld\n", tag, g, i, base + g * gs + m.act[i], lp ? 0 : geo[i].act_ps * B); out += line;
            snprintf(line, sizeof line, "%s.g%d.L%zu.st %ld %ld\n", tag, g, i, base + g * gs + m.st[i], st); out += line;
            snprintf(line, sizeof line, "%s.g%d.L%zu.actp %ld %ld\n", tag, g, i, base + g * gs + m.actp[i], lp ? 0 : (geo[i].act_ps * B + 1) / 2); out += line;
        }
}
void describe_scratch(std::string& out, const char* tag, const std::vector<LayerGeo>& geo, int B, int math, long base, long gs, int G) {
    long dY[24], dA[24], dYq[24];
    const bool q = math == VC_MATH_BF16;
    const long dyp = scratch_map(geo, B, dY, dA, q ? dYq : nullptr);
    char line[256];
    for (int g = 0; g < G; ++g) {
        for (size_t i = 0; i < geo.size(); ++i) {
            snprintf(line, sizeof line, "%s.g%d.L%zu.dY %ld %ld\n", tag, g, i, base + g * gs + dY[i], geo[i].raw_ps * B); out += line;
            snprintf(line, sizeof line, "%s.g%d.L%zu.dA %ld %ld\n", tag, g, i, base + g * gs + dA[i], geo[i].act_ps * B); out += line;
            if (q) { snprintf(line, sizeof line, "%s.g%d.L%zu.dYq %ld %ld\n", tag, g, i, base + g * gs + dYq[i], scratch_dyq_floats(geo[i], B)); out += line; }
        }
        snprintf(line, sizeof line, "%s.g%d.dYp %ld %ld\n", tag, g, base + g * gs + dyp, scratch_dyp_floats(geo, B)); out += line;
    }
}

bool layer_events_on() { return vc_hooks().layer_updates != 0; }
static int net_forward(NetCtx& c, const float* x, long x_gs, float* y_plain, long y_plain_gs) {
    unsigned wino_waited_mask = 0;
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
    // fp32 inside a train step: the 3-tap layers (residual blocks) run in the Winograd domain (vc_conv_wino.hip) on the step's transformed filters;
    // the transforms ride in the instance-norm launches on either side of the GEMMs (NormArgs::win / wout), stand-alone launches where they cannot
    auto wino_cached = [&](size_t i) {       // the step holds this layer's transformed filters and the launch is large enough for the route
        return c.images && c.math == VC_MATH_FP32 && (*c.images)[i].fwd && wino_route((*c.geo)[i].g, c.math, c.B);
    };
    auto wino_layer = [&](size_t i) { return wino_cached(i) && (*c.geo)[i].g.kh == 1 && (*c.geo)[i].g.kw == 3; };     // ... and its transforms can ride in the norm launches
    auto wino_wait = [&](size_t i) -> int {    // this step's filter transforms (side stream), chunk by chunk
        if (c.wino_ev && !(wino_waited_mask >> c.wino_chunk[i] & 1u)) { VC_HIP(hipStreamWaitEvent(c.s, c.wino_ev[c.wino_chunk[i]], 0)); wino_waited_mask |= 1u << c.wino_chunk[i]; }
        return 0;
    };
    float* wV = nullptr; float* wM = nullptr; bool v_ready = false;
    {
        size_t nV = 0, nM = 0;
        size_t other = 0;      // the routed layers whose transforms stay stand-alone take their scratch inside conv_fwd, from the same per-stream block:
                               // ask for the pass's largest need NOW, so that no later request can move the block under wV / wM
        for (size_t i = 0; i < c.n->layers.size(); ++i) {
            const Geom& g = (*c.geo)[i].g;
            if (wino_layer(i)) { nV = std::max(nV, (size_t)6 * c.G * c.B * g.Cin * (g.W / 4)); nM = std::max(nM, (size_t)6 * c.G * c.B * g.Cout * (g.W / 4)); }
            else if (wino_cached(i)) other = std::max(other, wino_scratch_floats(g, c.G, c.B));
        }
        if (nV || other) { VC_CHECK(wino_workspace(c.s, std::max(nV + nM, other), &wV)); wM = wV + nV; if (!nV) wV = wM = nullptr; }
    }
    const float* in = x; long in_gs = x_gs;
    const float* blk_in = nullptr; long blk_in_gs = 0;
    bool in_packed = false;               // bf16 data path: layer i's input already lies in actp[i - 1] as channel octets (its producer emitted them)
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
        const bool wino = wV && wino_layer(i) && L.mode != M_PLAIN;
        if (wino) {
            VC_CHECK(wino_wait(i));
            if (!v_ready) VC_CHECK(launch_wino_input(in, in_gs, wV, c.G, c.B, lg.g.Cin, lg.g.W / 4, c.s));
            v_ready = false;
            VC_CHECK(wino_gemm_fwd(wV, (const float*)(*c.images)[i].fwd, wM, c.G, c.B, lg.g, c.math, c.s));
        } else {
        const bool wino_u = wino_cached(i);      // 5-tap layers (pixel-shuffled up-sampling blocks): the route with stand-alone transforms, on the step's filters
        if (wino_u) VC_CHECK(wino_wait(i));
        VC_CHECK(conv_fwd(in, in_gs, c.params + L.w, c.par_gs, c.params + L.b, c.par_gs, raw, raw_gs, c.G, c.B, lg.g, c.math, c.s,
                          /*prezeroed=*/L.mode != M_PLAIN || plain_zeroed, nullptr, (c.images && (c.math == VC_MATH_BF16 || wino_u)) ? (*c.images)[i].fwd : nullptr,
                          (c.keep_packed && i > 0) ? (char*)(c.save + c.sm.actp[i - 1]) : nullptr, c.save_gs * 4,
                          fn.act ? &fn : nullptr, in_packed));
        }
        in_packed = false;
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
        if (wino) {            // the raw conv output still lies in the transform domain
            if (L.mode != M_GATED && norm_takes_wino(a, true)) {
                a.win = wM; a.win_gs = 6L * c.B * lg.g.Cout * (lg.g.W / 4); a.wbias = c.params + L.b; a.wbias_gs = c.par_gs; a.yw = raw;
            } else VC_CHECK(launch_wino_output(wM, c.params + L.b, c.par_gs, raw, raw_gs, c.G, c.B, lg.g.Cout, lg.g.W / 4, c.s));
        }
        if (wV && i + 1 < c.n->layers.size() && wino_layer(i + 1) && L.mode != M_GATED && norm_takes_wino(a, true)) {
            a.wout = wV; a.wout_gs = 6L * c.B * L.C * (a.L / 4);      // the next layer's GEMMs read this activation as V = B^T x
            v_ready = true;
        }
        // the next layer's convolution reads this activation as bf16 channel octets out of actp[i]: let the kernel that produces it write
        // them (round 4: a pack launch per convolution re-read the fp32 tensor)
        if (c.keep_packed && i + 1 < c.n->layers.size() && conv_fwd_packs_input((*c.geo)[i + 1].g, c.B, c.math) && norm_fwd_emits_packed(a)) {
            a.outp = (char*)(c.save + c.sm.actp[i]); a.outp_gs = c.save_gs * 4;
            in_packed = true;
        }
        VC_CHECK(launch_norm_fwd(L.mode, a, c.s));
        in = a.out; in_gs = c.save_gs;
    }
    return 0;
}

// dtop: gradient w.r.t. the network's last activation (for the generator: d o1 output, i.e. dy of the
// plain layer; for the discriminator: gradient w.r.t. the last GLU output, produced by the head).
// vc_hooks().pixel_octets (tests): 0 = the norm backward leaves dy in fp32 only for the filter gradient (which then packs its pixel octets
// itself: the round-4 path), 2 = octets written and the fp32 copy kept
// one launch zeroes every dA of a backward pass that a split-K input gradient will accumulate into.  Part of the pass itself, or -- for the
// passes whose scratch nobody else uses earlier in the step -- issued ahead of time on the side stream (BwdOpts::prezeroed: the launch sat
// alone on the chip between the loss and the first backward kernels of both chains, profiles/r06_timeline_b8.txt)
static int zero_split_dA(const NetCtx& c, float* scratch, long scr_gs) {
    const int nl = (int)c.n->layers.size();
    VC_ARG(nl <= 24, "net_backward: more than 24 layers");
    long dYo[24], dAo[24];
    scratch_map(*c.geo, c.B, dYo, dAo, nullptr);
    std::vector<std::pair<float*, long>> zl;
    for (int i = 1; i < nl; ++i) {
        bool split = false;
        VC_CHECK(conv_dgrad(nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, c.G, c.B, (*c.geo)[i].g, c.math, c.s, false, &split));
        if (split) for (int g = 0; g < c.G; ++g) zl.push_back({scratch + dAo[i - 1] + (long)g * scr_gs, (*c.geo)[i - 1].act_ps * c.B});
    }
    return zl.empty() ? 0 : launch_zero_list(zl, c.s);
}
static int net_backward(NetCtx& c, const float* x, long x_gs, const float* dtop, long dtop_gs,
                        float* dx, long dx_gs, int dx_acc, float* scratch, long scr_gs) {
    const int nl = (int)c.n->layers.size();
    VC_ARG(nl <= 24, "net_backward: more than 24 layers");
    long dYo[24], dAo[24], dYqo[24];
    const bool qmap = c.math == VC_MATH_BF16;            // the scratch was sized for the plan's mode (vc_scratch_floats / vc_step_workspace_floats)
    const long dYp_off = scratch_map(*c.geo, c.B, dYo, dAo, qmap ? dYqo : nullptr);
    long dYd[24], dAd[24], dYqd[24];
    if (c.deferred) scratch_map(*c.geo, c.deferred->B, dYd, dAd, qmap ? dYqd : nullptr);
    unsigned q_ready = 0, dy_dropped = 0;
    hipStream_t ws = c.ws ? c.ws : c.s;       // the plan sizes its event pools for its own networks (18 / 4 layers)
    if (!c.prezeroed) VC_CHECK(zero_split_dA(c, scratch, scr_gs));
    const float* dcur = dtop; long dcur_gs = dtop_gs;   // gradient w.r.t. the output activation of layer i
    const float* skip = nullptr; long skip_gs = 0;      // gradient flowing around a residual block (module.py:81)
    // fp32 inside a train step: input gradients of the 3-tap layers in the Winograd domain (see net_forward): dM = A dy leaves the layer's norm
    // backward, dV = U^T dM is the six GEMMs, and dO = B dV (+ skip) is rebuilt by the NEXT norm backward on its way in (`pend`)
    auto wino_cached = [&](int i) {
        return c.images && c.math == VC_MATH_FP32 && (*c.images)[i].has_dg && wino_route((*c.geo)[i].g, c.math, c.B);
    };
    auto wino_layer = [&](int i) { return wino_cached(i) && (*c.geo)[i].g.kh == 1 && (*c.geo)[i].g.kw == 3; };
    auto norm_fusable = [&](int i) {      // layer i's norm backward can apply the transforms (the launcher checks alignment once more)
        const Layer& Li = c.n->layers[i]; const LayerGeo& li = (*c.geo)[i];
        const int Ln = li.g.OH * li.g.OW;
        return (Li.mode == M_GATED_IN || Li.mode == M_IN_RES) && !Li.ps && !vc_det_active() && (Ln & 3) == 0 && Ln >= 8 && Ln <= 128;
    };
    float* wdM = nullptr; float* wdV = nullptr;
    {
        size_t nV = 0, nM = 0;
        size_t other = 0;      // (see net_forward: the pass's largest scratch need is requested up front)
        for (int i = 0; i < nl; ++i) {
            const Geom& g = (*c.geo)[i].g;
            if (i > 0 && wino_layer(i)) { nV = std::max(nV, (size_t)6 * c.G * c.B * g.Cin * (g.W / 4)); nM = std::max(nM, (size_t)6 * c.G * c.B * g.Cout * (g.W / 4)); }
            else if (wino_cached(i) && (i > 0 || dx)) other = std::max(other, wino_scratch_floats(g, c.G, c.B));
        }
        if (nV || other) { VC_CHECK(wino_workspace(c.s, std::max(nV + nM, other), &wdM)); wdV = wdM + nM; if (!nV) wdM = wdV = nullptr; }
    }
    struct { const float* dV; long dV_gs; const float* add; long add_gs; bool on; } pend{nullptr, 0, nullptr, 0, false};
    for (int i = nl - 1; i >= 0; --i) {
        const Layer& L = c.n->layers[i]; const LayerGeo& lg = (*c.geo)[i];
        const float* in = (i == 0) ? x : c.save + c.sm.act[i - 1];
        const long in_gs = (i == 0) ? x_gs : c.save_gs;
        const float* dyp; long dy_gs;
        const char* dy_packed = nullptr;                 // dy of this layer as bf16 channel octets in the scratch's packed region, if its producer emitted them
        const bool wino = wdM && i > 0 && wino_layer(i);
        bool dm_ready = false;
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
            if (pend.on) {     // the gradient of this activation lies in the transform domain; the block's skip gradient (= this dO) is kept for the block head
                a.dwin = pend.dV; a.dwin_gs = pend.dV_gs; a.dadd = pend.add; a.dadd_gs = pend.add_gs;
                if (L.mode == M_IN_RES) { a.dsave = const_cast<float*>(dcur); a.dsave_gs = dcur_gs; }
                pend.on = false;
            }
            if (wino && norm_fusable(i)) { a.dwout = wdM; a.dwout_gs = 6L * c.B * lg.g.Cout * (lg.g.W / 4); dm_ready = true; }
            // bf16 data path: this layer's input gradient gathers dy as bf16 channel octets -- the kernel that produces dy writes them
            // (round 4: a pack launch per input gradient re-read the fp32 tensor).  i == 0 without dx: nobody reads them.
            if ((i > 0 || dx) && conv_dgrad_packs_input(lg.g, c.B, c.math) && norm_bwd_emits_packed(a)) {
                a.outp = (char*)(scratch + dYp_off); a.outp_gs = scr_gs * 4;
                dy_packed = (const char*)a.outp;
            }
            // ... and its filter gradient (the bf16 GEMM over pixels) reads dy as pixel octets out of the layer's stacked buffer: this
            // pass's own scratch, or the deferred application's when that pass opened the buffer (its part sits in front)
            if (c.dparams && vc_hooks().pixel_octets && norm_bwd_emits_pixel_octets(a) && conv_wgrad_reads_pixel_octets(lg.g, (long)c.B * lg.g.OH * lg.g.OW, c.math) &&
                (!c.deferred || c.B <= 2 * c.deferred->B)) {
                if (c.deferred) { a.outq = (char*)const_cast<float*>(c.deferred->scratch + dYqd[i]); a.outq_gs = c.deferred->scr_gs * 4; a.outq_po0 = c.deferred->B * lg.g.OH * lg.g.OW / 8; }
                else { a.outq = (char*)(scratch + dYqo[i]); a.outq_gs = scr_gs * 4; a.outq_po0 = 0; }
                q_ready |= 1u << i;
            }
            // both readers of this dy take the octet forms (or do not exist): its fp32 copy -- the largest stream of the kernel -- stays unwritten
            if (c.drop_dy && vc_hooks().pixel_octets != 2 && ((i == 0 && !dx) || dy_packed) && (!c.dparams || (q_ready >> i & 1u)) && (dy_packed || a.outq)) { a.dy = nullptr; dy_dropped |= 1u << i; }
            VC_CHECK(launch_norm_bwd(L.mode, a, c.s));
            dyp = a.dy ? scratch + dYo[i] : nullptr; dy_gs = scr_gs;
            if (c.dparams && c.wmode != W_NONE && ws != c.s) { VC_HIP(hipEventRecord(c.ev_layer[i], c.s)); VC_HIP(hipStreamWaitEvent(ws, c.ev_layer[i], 0)); }
        }
        if (c.dparams && c.wmode != W_NONE) {
            WgradApp apps[2]; int na = 0;
            if (c.deferred) {
                const DeferredApp& d = *c.deferred;
                apps[na++] = WgradApp{i == 0 ? d.x : d.save + d.sm->act[i - 1], i == 0 ? d.x_gs : d.save_gs,
                                      L.mode == M_PLAIN ? d.dtop : (d.dy_dropped >> i & 1u) ? nullptr : d.scratch + dYd[i], L.mode == M_PLAIN ? d.dtop_gs : d.scr_gs, d.B};
                if (c.keep_packed && i > 0 && conv_fwd_packs_input(lg.g, d.B, c.math)) {   // d.B: the batch of that forward
                    apps[na - 1].xp = (const char*)(d.save + d.sm->actp[i - 1]); apps[na - 1].xp_gs = d.save_gs * 4;
                }
            }
            apps[na++] = WgradApp{in, in_gs, dyp, dy_gs, c.B};
            if (c.keep_packed && i > 0 && conv_fwd_packs_input(lg.g, c.Bl, c.math)) {       // c.Bl: the batch the forward ran (and laid out) with
                apps[na - 1].xp = (const char*)(c.save + c.sm.actp[i - 1]); apps[na - 1].xp_gs = c.save_gs * 4;
            }
            WgradPacked pq{nullptr, 0, 0, 0};
            if (!qmap) {}
            else if (c.deferred) {
                pq.dyq = (char*)const_cast<float*>(c.deferred->scratch + dYqd[i]); pq.dyq_gs = c.deferred->scr_gs * 4;
                pq.cap = (size_t)scratch_dyq_floats(lg, c.deferred->B) * 4;
                pq.ready = (c.deferred->q_ready >> i & 1u) | ((q_ready >> i & 1u) << 1);
            } else { pq.dyq = (char*)(scratch + dYqo[i]); pq.dyq_gs = scr_gs * 4; pq.cap = (size_t)scratch_dyq_floats(lg, c.B) * 4; pq.ready = q_ready >> i & 1u; }
            VC_CHECK(conv_wgrad(apps, na, c.dparams + L.w, c.dpar_gs, c.G, lg.g, c.math, ws, c.wgrad_store, nullptr, qmap ? &pq : nullptr));
        }
        for (int k = 0; k < c.n_ev; ++k) if (c.ev_first[k] == i) VC_HIP(hipEventRecord(c.ev[k], ws));
        if (c.lay_ready && layer_events_on()) VC_HIP(hipEventRecord(c.lay_ready[i], ws));
        if (i == 0) {
            if (dx) VC_CHECK(conv_dgrad(dyp, dy_gs, c.params + L.w, c.par_gs, dx, dx_gs, nullptr, 0, dx_acc, c.G, c.B, lg.g, c.math, c.s,
                                        false, nullptr, (c.images && c.math == VC_MATH_BF16 && (*c.images)[i].has_dg) ? (*c.images)[i].dg : nullptr, dy_packed, scr_gs * 4));
            for (int k = 0; k < c.n_ev; ++k) if (c.ev_free && c.ev_first[k] == i) VC_HIP(hipEventRecord(c.ev_free[k], c.s));
            if (c.lay_free && layer_events_on()) VC_HIP(hipEventRecord(c.lay_free[i], c.s));
            break;        // falls through to the join below
        }
        // gradient w.r.t. this layer's input = activation of layer i-1; the head of a residual block adds the skip
        const bool block_head = (L.mode == M_GATED_IN && i + 1 < nl && c.n->layers[i + 1].mode == M_IN_RES);
        if (wino) {
            const int nt = lg.g.W / 4;
            if (!dm_ready) VC_CHECK(launch_wino_dy(dyp, dy_gs, wdM, c.G, c.B, lg.g.Cout, nt, c.s));
            VC_CHECK(wino_gemm_dgrad(wdM, (const float*)(*c.images)[i].dg[0], wdV, c.G, c.B, lg.g, c.math, c.s));
            if (norm_fusable(i - 1)) { pend.dV = wdV; pend.dV_gs = 6L * c.B * lg.g.Cin * nt; pend.add = block_head ? skip : nullptr; pend.add_gs = block_head ? skip_gs : 0; pend.on = true; }
            else VC_CHECK(launch_wino_dx(wdV, scratch + dAo[i - 1], scr_gs, block_head ? skip : nullptr, block_head ? skip_gs : 0, 0, c.G, c.B, lg.g.Cin, nt, c.s));
        } else
        VC_CHECK(conv_dgrad(dyp, dy_gs, c.params + L.w, c.par_gs, scratch + dAo[i - 1], scr_gs,
                            block_head ? skip : nullptr, block_head ? skip_gs : 0, 0, c.G, c.B, lg.g, c.math, c.s, /*prezeroed=*/true,
                            nullptr, (c.images && (c.math == VC_MATH_BF16 || wino_cached(i)) && (*c.images)[i].has_dg) ? (*c.images)[i].dg : nullptr, dy_packed, scr_gs * 4));
        if (block_head) skip = nullptr;
        for (int k = 0; k < c.n_ev; ++k) if (c.ev_free && c.ev_first[k] == i) VC_HIP(hipEventRecord(c.ev_free[k], c.s));
        if (c.lay_free && layer_events_on()) VC_HIP(hipEventRecord(c.lay_free[i], c.s));
        dcur = scratch + dAo[i - 1]; dcur_gs = scr_gs;
    }
    if (ws != c.s) { VC_HIP(hipEventRecord(c.ev_wjoin, ws)); VC_HIP(hipStreamWaitEvent(c.s, c.ev_wjoin, 0)); }   // join
    if (c.q_ready_out) *c.q_ready_out = q_ready;
    if (c.dy_dropped_out) *c.dy_dropped_out = dy_dropped;
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
StepImages::StepImages(const vc_plan* p) { t_step_images = p; }
StepImages::~StepImages() { t_step_images = nullptr; }
static const std::vector<vc_plan::LayerImages>* step_images(const vc_plan* p, bool gen, int G) {
    if (t_step_images != p || G != 2) return nullptr;
    if (p->math == VC_MATH_FP32) {
        const auto& v = gen ? p->wino_img : p->wino_img_d;
        return (p->wino_active && !v.empty() && vc_hooks().winograd != 0) ? &v : nullptr;
    }
    if (p->math != VC_MATH_BF16) return nullptr;
    const auto& v = gen ? p->img_g : p->img_d;
    return v.empty() ? nullptr : &v;
}
// fp32 mode, head of a train step: U = G w for every generator layer on the Winograd route, one launch on stream s (then ev_wino)
int refresh_wino_filters(const vc_plan* p, const float* params, int B, hipStream_t s, bool* active) {
    *active = false; p->wino_active = false;
    if (p->math != VC_MATH_FP32 || vc_hooks().winograd == 0 || p->inference) return 0;
    // The block's layout covers every layer the route can EVER take in this plan (geometry only); a step transforms the layers its own batch
    // takes it for (B = the largest batch of its passes: a layer eligible in a smaller pass is eligible here).  Round 6, found by the
    // deterministic-mode test: a layout built from the first step's batch was too small for a later, larger one.
    const long PG = p->G.total, PD = p->D.total;
    struct Lay { int net, i; size_t off; WinoFilter f; };
    std::vector<Lay> lays; size_t tot = 0;
    for (int net = 0; net < 2; ++net) {
        const NetDesc& n = net ? p->D : p->G; const auto& geo = net ? p->dg : p->gg;
        const float* base = net ? params + 2 * PG : params; const long gs = net ? PD : PG;
        for (size_t i = 0; i < n.layers.size(); ++i) {
            const Geom& g = geo[i].g;
            const bool w1 = wino_eligible(g, p->math, 1 << 20), wh = !w1 && winoh_eligible(g, p->math, 1 << 20);
            if (!w1 && !wh) continue;
            const int planes = wh ? 8 : g.kw + 3;
            const long inner = wh ? (long)g.kw * g.Cin * g.Cout : (long)g.Cin * g.Cout;
            lays.push_back(Lay{net, (int)i, tot, WinoFilter{base + n.layers[i].w, nullptr, gs, wh ? g.kw * g.Cin : g.Cin, g.Cout, wh ? 6 : g.kw}});
            tot += (size_t)2 * planes * inner;
        }
    }
    if (lays.empty()) return 0;
    auto takes = [&](const Lay& l) { return wino_route((l.net ? p->dg : p->gg)[l.i].g, p->math, B); };
    bool any = false;
    for (const Lay& l : lays) any = any || takes(l);
    if (!any) return 0;
    if (!p->wino_block) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            vc_set_error("Winograd route: run one eager step before capturing (the transformed filters are allocated on first use)");
            return VC_E_STATE;
        }
        VC_HIP(hipMalloc((void**)&p->wino_block, tot * sizeof(float)));
    }
    if (p->wino_img.empty()) {
        // in layer order (generators, then discriminators), kWinoChunks launches with an event behind each: the first-hop forward reaches its
        // first residual block ~70 us into the step and waits for that block's filters only (one launch: 0.12 ms ahead of the first GEMM)
        p->wino_img.assign(p->G.layers.size(), vc_plan::LayerImages{}); p->wino_img_d.assign(p->D.layers.size(), vc_plan::LayerImages{});
        p->wino_chunk.assign(p->G.layers.size(), -1); p->wino_chunk_d.assign(p->D.layers.size(), -1);
        const int per = ((int)lays.size() + vc_plan::kWinoChunks - 1) / vc_plan::kWinoChunks;
        int k = 0;
        for (const Lay& l : lays) {
            vc_plan::LayerImages& li = (l.net ? p->wino_img_d : p->wino_img)[l.i];
            li.fwd = (const char*)(p->wino_block + l.off); li.dg[0] = li.fwd; li.has_dg = true;
            (l.net ? p->wino_chunk_d : p->wino_chunk)[l.i] = k++ / per;
        }
    }
    for (int ck = 0; ck < vc_plan::kWinoChunks; ++ck) {
        std::vector<WinoFilter> jobs;
        for (const Lay& l : lays)
            if ((l.net ? p->wino_chunk_d : p->wino_chunk)[l.i] == ck && takes(l)) { WinoFilter f = l.f; f.U = p->wino_block + l.off; jobs.push_back(f); }
        if (!jobs.empty()) VC_CHECK(launch_wino_filters(jobs, 2, s));
        VC_HIP(hipEventRecord(p->ev_wino[ck], s));
    }
    *active = true; p->wino_active = true; p->wino_stream = s;
    return 0;
}
int gen_forward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                       float* y, long y_gs, float* save, int G, int B, hipStream_t s) {
    NetCtx c = gen_ctx(p, params, par_gs, nullptr, 0, save, G, B, B, s);
    c.images = step_images(p, true, G); c.keep_packed = c.images != nullptr && p->math == VC_MATH_BF16;
    if (c.images && p->math == VC_MATH_FP32 && s != p->wino_stream) { c.wino_ev = p->ev_wino; c.wino_chunk = p->wino_chunk.data(); }
    c.fuse_glu = p->inference;
    return net_forward(c, x, x_gs, y, y_gs);
}
int gen_backward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                        const float* dy, long dy_gs, float* save, int Bl, float* dparams, long dpar_gs,
                        float* dx, long dx_gs, int dx_acc, float* scratch, long scr_gs, int G, int B, hipStream_t s,
                        const BwdOpts& o) {
    VC_ARG(!p->inference, "this plan was created for inference only (vc_plan_create_inference)");
    NetCtx c = gen_ctx(p, params, par_gs, dparams, dpar_gs, save, G, B, Bl, s);
    c.prezeroed = o.prezeroed;
    c.wgrad_store = o.wgrad_store; c.wmode = o.wmode; c.deferred = o.deferred; c.q_ready_out = o.q_ready_out; c.dy_dropped_out = o.dy_dropped_out; c.drop_dy = o.drop_fp32_dy;
    c.images = step_images(p, true, G); c.keep_packed = c.images != nullptr && p->math == VC_MATH_BF16;
    if (c.images && p->math == VC_MATH_FP32 && s != p->wino_stream) { c.wino_ev = p->ev_wino; c.wino_chunk = p->wino_chunk.data(); }
    if ((p->use_side & 2) && dparams && o.wmode != W_NONE) { c.ws = p->wstream; c.ev_layer = p->ev_layer; c.ev_wjoin = p->ev_wjoin; }
    if (o.record_buckets) { c.ev_first = p->bucket_first; c.ev = p->g_ready; c.ev_free = p->g_free; c.n_ev = vc_plan::kBuckets; c.lay_ready = p->gl_ready; c.lay_free = p->gl_free; }
    return net_backward(c, x, x_gs, dy, dy_gs, dx, dx_gs, dx_acc, scratch, scr_gs);
}
// the zero launch of a later gen_backward / disc_backward at batch B on this scratch, issued now on stream s (BwdOpts::prezeroed)
int backward_prezero(const vc_plan* p, bool gen, float* scratch, long scr_gs, int G, int B, hipStream_t s) {
    NetCtx c = gen ? gen_ctx(p, nullptr, 0, nullptr, 0, nullptr, G, B, B, s) : disc_ctx(p, nullptr, 0, nullptr, 0, nullptr, G, B, B, s);
    return zero_split_dA(c, scratch, scr_gs);
}
int disc_forward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                        float* o, long o_gs, float* save, int G, int B, hipStream_t s) {
    VC_ARG(!p->inference, "this plan was created for inference only (vc_plan_create_inference)");
    NetCtx c = disc_ctx(p, params, par_gs, nullptr, 0, save, G, B, B, s);
    c.images = step_images(p, false, G); c.keep_packed = c.images != nullptr && p->math == VC_MATH_BF16;
    // (same stream as the rebuild: ordered already -- and under hipGraph capture a wait on an event the waiting stream itself recorded crashed
    // hipStreamEndCapture on ROCm 7.2, found by the graph test when D.d3 joined the route)
    if (c.images && p->math == VC_MATH_FP32 && s != p->wino_stream) { c.wino_ev = p->ev_wino; c.wino_chunk = p->wino_chunk_d.data(); }
    VC_CHECK(net_forward(c, x, x_gs, nullptr, 0));
    const int last = (int)p->D.layers.size() - 1;
    return launch_head_fwd(save + c.sm.act[last], c.save_gs, params + p->D.head_w, params + p->D.head_b, par_gs,
                           o, o_gs, G, B, p->D.head_c, p->d_hw, s);
}
// o: the forward outputs [G][Bl][HW] (first B samples are used)
int disc_backward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                         const float* o, long o_gs, const float* dout, long do_gs, float* save, int Bl,
                         float* dparams, long dpar_gs, float* dx, long dx_gs, int dx_acc,
                         float* scratch, long scr_gs, int G, int B, hipStream_t s, bool wgrad_store, bool drop_fp32_dy, bool record_layers, bool prezeroed) {
    VC_ARG(!p->inference, "this plan was created for inference only (vc_plan_create_inference)");
    NetCtx c = disc_ctx(p, params, par_gs, dparams, dpar_gs, save, G, B, Bl, s);
    c.wgrad_store = wgrad_store; c.drop_dy = drop_fp32_dy;
    c.images = step_images(p, false, G); c.keep_packed = c.images != nullptr && p->math == VC_MATH_BF16;
    // (same stream as the rebuild: ordered already -- and under hipGraph capture a wait on an event the waiting stream itself recorded crashed
    // hipStreamEndCapture on ROCm 7.2, found by the graph test when D.d3 joined the route)
    if (c.images && p->math == VC_MATH_FP32 && s != p->wino_stream) { c.wino_ev = p->ev_wino; c.wino_chunk = p->wino_chunk_d.data(); }
    c.prezeroed = prezeroed;
    if (record_layers) { if (dparams) { c.lay_ready = p->dl_ready; c.lay_free = p->dl_free_d; } else c.lay_free = p->dl_free_g; }
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
                                     float* scratch, int64_t scratch_floats, int G, int B, void* stream) {
    VC_ARG(p && params && x && dy && save && scratch && G > 0 && B > 0, "vc_generator_backward: bad args");
    if (scratch_floats < vc_scratch_floats(p, B) * G) {
        vc_set_error("vc_generator_backward: scratch too small: %lld < %lld floats (vc_scratch_floats depends on the plan's math mode)",
                     (long long)scratch_floats, (long long)(vc_scratch_floats(p, B) * G));
        return VC_E_NOMEM;
    }
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
                                         float* scratch, int64_t scratch_floats, int G, int B, void* stream) {
    VC_ARG(p && params && x && dout && save && scratch && G > 0 && B > 0, "vc_discriminator_backward: bad args");
    if (scratch_floats < vc_scratch_floats(p, B) * G) {
        vc_set_error("vc_discriminator_backward: scratch too small: %lld < %lld floats (vc_scratch_floats depends on the plan's math mode)",
                     (long long)scratch_floats, (long long)(vc_scratch_floats(p, B) * G));
        return VC_E_NOMEM;
    }
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
