// vc_step.hip -- the fused CycleGAN-VC train step, step graphs, gradient buckets (part 4 of 4: vc_host.h)
#include "vc_host.h"
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
    m.scrc_gs = scratch_floats_net(p->gg, B, p->math); m.SCRC = take(m.scrc_gs);   // cycle-hop backward: its dY buffers stay live until the
                                                                          // first-hop backward has reduced them into the filter gradients
    m.SCR2 = take(m.scr_gs);                      // scratch of the discriminator_loss backward (runs on the side stream)
    m.ACC = o; o += align64(VC_LOSS_BLOCKS * 10);     // per-block partial sums of the loss kernel
    m.total = o;
    return m;
}
// testing: where every buffer of the step lies inside the caller's workspace ("name offset floats" per line; the scratch areas as
// each pass lays them out), and a poison for the plan-internal buffers -- the round-6 stale-read probe named its buffers with them (profiles/r06_stale_read_probe.txt; tests/test_step_gpu.py poisons between steps)
void describe_save(std::string& out, const char* tag, const NetDesc& n, const std::vector<LayerGeo>& geo, int B, long base, long gs, int G);
void describe_scratch(std::string& out, const char* tag, const std::vector<LayerGeo>& geo, int B, int math, long base, long gs, int G);
extern "C" long long vc_testing_describe_step(const vc_plan* p, int B, int Bbwd, char* buf, long long cap) {
    if (!p || B <= 0) return -1;
    const StepMap m = step_map(p, B);
    std::string out; char line[256];
    auto two = [&](const char* name, long off, long n) {
        for (int g = 0; g < 2; ++g) { snprintf(line, sizeof line, "%s.g%d %ld %ld\n", name, g, off + g * align64(n), n); out += line; }
    };
    two("X1", m.X1, m.ft2); two("O1", m.O1, m.ft2); two("CY", m.CY, m.ft1); two("DX", m.DX, m.ft2); two("DO", m.DO, m.hw2);
    two("dO1", m.dO1, m.ft2); two("dCY", m.dCY, m.ft1); two("dDO_D", m.dDO_D, m.hw2); two("dDO_G", m.dDO_G, m.hw1);
    describe_save(out, "S1", p->G, p->gg, 2 * B, m.S1, align64(m.s1_gs), 2);
    describe_save(out, "S2", p->G, p->gg, B, m.S2, align64(m.s2_gs), 2);
    describe_save(out, "SD", p->D, p->dg, 2 * B, m.SD, align64(m.sd_gs), 2);
    describe_scratch(out, "SCR.step8", p->dg, B, p->math, m.SCR, align64(m.scr_gs), 2);
    describe_scratch(out, "SCR.step10", p->gg, Bbwd, p->math, m.SCR, align64(m.scr_gs), 2);
    describe_scratch(out, "SCRC.step9", p->gg, B, p->math, m.SCRC, align64(m.scrc_gs), 2);
    describe_scratch(out, "SCR2.step7", p->dg, 2 * B, p->math, m.SCR2, align64(m.scr_gs), 2);
    snprintf(line, sizeof line, "ACC %ld %d\ntotal %ld 0\n", m.ACC, VC_LOSS_BLOCKS * 10, m.total); out += line;
    if (buf && cap > 0) { const size_t n = std::min<size_t>(out.size(), (size_t)cap - 1); memcpy(buf, out.data(), n); buf[n] = 0; }
    return (long long)out.size() + 1;
}
extern "C" int vc_testing_poison_internal(const vc_plan* p, void* stream, int byte) {
    VC_ARG(p, "vc_testing_poison_internal: bad plan");
    VC_CHECK(poison_pack_spaces({(hipStream_t)stream, p->side, p->wstream, p->wstream_d, p->gstream, p->ustream}, byte));
    // (the filter images are rebuilt from the weights at the head of every step: poisoning them checks exactly that)
    if (p->img_block) {
        size_t sz = 0; hipDeviceptr_t base = nullptr;
        if (hipMemGetAddressRange(&base, &sz, (hipDeviceptr_t)p->img_block) == hipSuccess && sz) VC_HIP(hipMemset(p->img_block, byte, sz));
    }
    return 0;
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
// as the two groups of one launch).  Two launches: the generators' forward images on the step's main stream (the first-hop forward
// needs them at once), everything else -- the generators' input-gradient images and all of the discriminators' -- is handed back in
// `later` and goes out on the side stream under the first-hop forward (round 4: one launch of 0.52 ms at batch 64 with nothing beside it).
static int refresh_filter_images(const vc_plan* p, const float* params, hipStream_t s, std::vector<PackFilterJob>* later) {
    const long PG = p->G.total, PD = p->D.total;
    struct Job { const float* w; long w_gs; const int* wsrc; int a_ms, M, kp; size_t off, gs; bool now; };
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
                jobs.push_back(Job{base + n.layers[i].w, gs, t.wsrc, 1, g.Cout, t.koct * 8, tot, a, net == 0});
                img[i].fwd = (const char*)tot; tot += 2 * a;
            }
            bool all = g.Cin >= 64; size_t off[4]; Job cj[4]; int nc = 0;
            for (int ph = 0; ph < g.sh && all; ++ph) for (int pw = 0; pw < g.sw && all; ++pw) {
                Table td; VC_CHECK(get_dgrad_table(g, ph, pw, &td));
                if (td.koct <= 0) { all = false; break; }
                const size_t a = align256((size_t)g.Cin * td.koct * 16);
                cj[nc] = Job{base + n.layers[i].w, gs, td.wsrc, g.Cout, g.Cin, td.koct * 8, tot, a, false}; off[nc++] = tot; tot += 2 * a;
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
        (j.now ? pj : *later).push_back(PackFilterJob{j.w, (unsigned short*)(p->img_block + j.off), j.wsrc, j.w_gs, (long)j.gs / 2, j.a_ms, j.M, j.kp, 0, 0});
    return pj.empty() ? 0 : launch_pack_filter_multi(pj, 2, s);
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
    std::vector<PackFilterJob> later_images;
    if (p->math == VC_MATH_BF16) VC_CHECK(refresh_filter_images(p, params, s, &later_images));
    bool wino = false;                      // fp32: the generators' 3-tap layers run in the Winograd domain on filters transformed once per step
    if (p->math == VC_MATH_FP32 && s2 == s) VC_CHECK(refresh_wino_filters(p, params, 2 * B, s, &wino));
    const bool wino_side = p->math == VC_MATH_FP32 && s2 != s;
    StepImages scope((p->math == VC_MATH_BF16 || p->math == VC_MATH_FP32) ? p : nullptr);
    // 1. first-hop inputs: g0 = A2B on [A;B], g1 = B2A on [B;A]   (model.py:44,51 / 47,50)
    VC_CHECK(launch_copy_list({{X1, A, ft1}, {X1 + ft1, Bm, ft1}, {X1 + X1g, Bm, ft1}, {X1 + X1g + ft1, A, ft1}}, s));
    // 0. zero the accumulated parts of the flat gradient on the side stream, under the first-hop forward (nothing touches
    //    them before step 7; the fork/join pairs below order it before every consumer).  The fork comes AFTER the first
    //    work on `stream`: under hipGraph capture, an event recorded on a stream that has captured nothing yet does not
    //    pull the waiting stream into the capture (ROCm 7.2; the replayed graph then crashes at hipStreamEndCapture).
    if (s2 != s) { VC_HIP(hipEventRecord(p->ev_fork, s)); VC_HIP(hipStreamWaitEvent(s2, p->ev_fork, 0)); }
    // the filter images nothing reads before the discriminator forward (side stream, in order behind this launch) and the backward
    // passes (behind the join after the discriminator forward)
    if (wino_side) VC_CHECK(refresh_wino_filters(p, params, 2 * B, s2, &wino));      // first on the side stream: the first-hop forward reaches its residual blocks ~0.1 ms in
    if (!later_images.empty()) VC_CHECK(launch_pack_filter_multi(later_images, 2, s2));
    VC_CHECK(zero_step_grads(p, grads, B, Bbwd, s2));
    // ... and the split-K destinations of the three backward passes whose scratch is theirs alone until they run (step 10 shares SCR with step 8)
    const bool pz = s2 != s;
    if (pz) {
        VC_CHECK(backward_prezero(p, false, ws + m.SCR2, align64(m.scr_gs), 2, 2 * B, s2));
        VC_CHECK(backward_prezero(p, false, ws + m.SCR, align64(m.scr_gs), 2, B, s2));
        VC_CHECK(backward_prezero(p, true, ws + m.SCRC, align64(m.scrc_gs), 2, B, s2));
    }
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
                           grads + 2 * PG, PD, nullptr, 0, 0, ws + m.SCR2, scr_gs, 2, 2 * B, s2, /*wgrad_store=*/true, /*drop_fp32_dy=*/true, /*record_layers=*/true, /*prezeroed=*/pz));
    VC_CHECK(launch_loss_finalize(la, s2));    // the 8 scalars: off the backward's critical path (joined with the side stream before the step returns)
    VC_HIP(hipEventRecord(p->d_ready, s2));
    // 8. adversarial generator terms: gradient through D's inputs only (var_list = generator_vars, model.py:95,108)
    //    g0 = D_A -> d gen_A = dO1[g1][0:B]; g1 = D_B -> d gen_B = dO1[g0][0:B]
    VC_CHECK(disc_backward(p, params + 2 * PG, PD, DX, DXg, DO, DOg, dDO_G, dDGg, SD, 2 * B,
                           nullptr, 0, dO1 + dO1g, -dO1g, 0, SCR, scr_gs, 2, B, s, /*wgrad_store=*/false, /*drop_fp32_dy=*/true, /*record_layers=*/true, /*prezeroed=*/pz));
    VC_HIP(hipEventRecord(p->d_free, s));     // last read of the discriminator weights in this step
    // 9. cycle hop backward: g0 = A2B on gen_A -> adds into d gen_A; g1 = B2A on gen_B -> d gen_B.  Its filter
    //    gradients are NOT launched here: step 10 reduces both applications of every layer in one launch.
    unsigned cyc_q_ready = 0, cyc_dy_dropped = 0;   // layers whose cycle-hop dY already lies in SCRC's stacked pixel-octet buffers (bf16 data path)
    BwdOpts oc; oc.wmode = W_NONE; oc.q_ready_out = &cyc_q_ready; oc.dy_dropped_out = &cyc_dy_dropped;
    oc.prezeroed = pz;
    oc.drop_fp32_dy = true;                 // this step's first-hop pass reduces at most 2 B more samples: it always takes the stacked buffers
    VC_CHECK(gen_backward(p, params, PG, O1 + O1g, -O1g, dCY, dCYg, S2, B, grads, PG,
                          dO1 + dO1g, -dO1g, 1, SCRC, scrc_gs, 2, B, s, oc));
    // 10. first-hop backward over [generation; identity] (+ the deferred cycle-hop filter gradients)
    const SaveMap sm2 = save_map(p->G, p->gg, B);
    const DeferredApp cyc{O1 + O1g, -O1g, S2, sm2.total, &sm2, SCRC, scrc_gs, dCY, dCYg, B, cyc_q_ready, cyc_dy_dropped};
    BwdOpts o1; o1.record_buckets = true; o1.wgrad_store = true; o1.deferred = &cyc; o1.drop_fp32_dy = true;
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
    // the key carries the CALLER's overlap setting: the capture below masks use_side, and a key built from the masked value would never
    // match the next call's lookup (every graph step re-captured)
    const int overlap_key = p->use_side & 7;
    auto make_key = [&]() {
        return std::array<const void*, 12>{params, grads, m, v, A, Bm, losses, gen_A, gen_B, ws,
                                           (const void*)(intptr_t)(batch * 16 + (lambda_identity == 0.0f ? 1 : 0) * 8 + overlap_key),
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
        if (p->math == VC_MATH_BF16 || p->math == VC_MATH_FP32) {
            // (fp32: the Winograd route's transformed filters and transform-domain scratch, same rule)
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

extern "C" int vc_plan_num_step_graphs(const vc_plan* p) { return p ? (int)p->graphs.size() : -1; }
extern "C" int vc_plan_set_overlap(vc_plan* p, int enabled) {
    VC_ARG(p, "vc_plan_set_overlap: bad plan");
    p->use_side = enabled == 1 ? 7 : ((enabled >> 3) & 7);   // 0 / 1: the documented switch; 8 * bits selects single mechanisms (debugging)
    return 0;
}
extern "C" int vc_plan_set_exchange_mode(vc_plan* p, int mode) {
    VC_ARG(p && (mode == VC_EXCHANGE_BUCKETED || mode == VC_EXCHANGE_SINGLE), "vc_plan_set_exchange_mode: plan, mode 0 (bucketed) or 1 (single)");
    p->exchange_mode = mode;
    return 0;
}
extern "C" int vc_plan_get_exchange_mode(const vc_plan* p) { return p ? p->exchange_mode : VC_E_STATE; }
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
// single-GPU update: one segment per LAYER, in the order the backward finishes them (vc_comm.hip: step_update_impl without a communicator)
int step_num_layer_segments(const vc_plan* p) { return (int)(p->D.layers.size() + p->G.layers.size()); }
int step_layer_segment(const vc_plan* p, int k, int64_t* lo0, int64_t* lo1, int64_t* n, int* disc, hipStream_t us) {
    const int nD = (int)p->D.layers.size(), nG = (int)p->G.layers.size();
    const bool d = k < nD;
    const int i = d ? nD - 1 - k : nG - 1 - (k - nD);              // backward order: D.d3 .. D.h1, then G.o1 .. G.h1
    const NetDesc& net = d ? p->D : p->G;
    const long lo = net.layers[i].w, hi = (i + 1 < (int)net.layers.size()) ? net.layers[i + 1].w : net.total;   // D's last layer takes the dense head with it
    *lo0 = vc_flat_offset(p, d ? 2 : 0) + lo; *lo1 = vc_flat_offset(p, d ? 3 : 1) + lo; *n = hi - lo; *disc = d ? 1 : 0;
    if (d) {
        VC_HIP(hipStreamWaitEvent(us, p->dl_ready[i], 0)); VC_HIP(hipStreamWaitEvent(us, p->dl_free_d[i], 0)); VC_HIP(hipStreamWaitEvent(us, p->dl_free_g[i], 0));
    } else {
        VC_HIP(hipStreamWaitEvent(us, p->gl_ready[i], 0)); VC_HIP(hipStreamWaitEvent(us, p->gl_free[i], 0));
    }
    return 0;
}
extern "C" int vc_step_wait_d_grads(const vc_plan* p, void* other_stream) {
    VC_ARG(p, "vc_step_wait_d_grads: bad plan");
    VC_HIP(hipStreamWaitEvent((hipStream_t)other_stream, p->d_ready, 0));
    VC_HIP(hipStreamWaitEvent((hipStream_t)other_stream, p->d_free, 0));
    return 0;
}
