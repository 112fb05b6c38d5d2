// vc_host.h -- host-side declarations shared by the four host translation units of libvcgan_hip.so:
//   vc_tables.hip  errors, CRC-32C, TF-SAME geometry, the K tables of every convolution, thread math / deterministic modes
//   vc_ops.hip     convolution operators (forward, input gradient, filter gradient; fp32 and 16-bit data path) + operator C-ABI
//   vc_net.hip     network descriptions (module.py:148-213), parameter layout, plans, forward / backward schedules
//   vc_step.hip    the fused train step (model.py:44-125), step graphs, gradient buckets
#pragma once
#include "vc_internal.h"
#include "../../include/vc_cyclegan.h"
#include "../../include/vc_cyclegan_testing.h"
#include <array>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#define VC_CHECK(x) do { int _r = (x); if (_r != 0) return _r; } while (0)
#define VC_ARG(cond, ...) do { if (!(cond)) { vc_set_error(__VA_ARGS__); return VC_E_ARG; } } while (0)

// ---- geometry and K tables (vc_tables.hip)
struct Geom { int Cin, H, W, Cout, kh, kw, sh, sw, OH, OW, pt, pl; };
Geom make_geom(int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw);
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
template <class T> static int to_device(const std::vector<T>& h, T** d) {
    VC_HIP(hipMalloc((void**)d, h.size() * sizeof(T)));
    VC_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}
int get_fwd_table(const Geom& g, Table* out, std::vector<TabKey>* owner = nullptr);
int get_dgrad_table(const Geom& g, int ph, int pw, Table* out, std::vector<TabKey>* owner = nullptr);
void release_tables(std::vector<TabKey>& keys);
int vc_thread_math();                 // the calling thread's default math mode (vc_set_math_mode)
int vc_thread_det();                  // ... and deterministic mode (vc_set_deterministic)
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static inline long align64(long x) { return (x + 63) & ~63L; }
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- convolution operators (vc_ops.hip)
bool c1_path(const Geom& g);          // Cin = 1 with at most 9 taps (D.h1): vector-ALU kernels
bool wsconv_path(const Geom& g);      // 1-D, stride 1, 24 input channels, 15 taps (G.h1): weight-stationary kernels
C1Args c1_args(const Geom& g, int G, int B, int math);
int check_conv_geom(const Geom& g, int B);
bool conv_fwd_packs_input(const Geom& g, int B, int math);
// What a forward convolution needs to finish a gated block with instance norm in its own epilogue (IgemmArgs::fuse)
struct FuseNorm { const float* gamma; const float* beta; long par_gs; float* stats; long st_gs; float* act; long act_gs; bool done; };
// One application of a layer as seen by the filter gradient: its input, the gradient of its raw output, its batch.
struct WgradApp { const float* x; long x_gs; const float* dy; long dy_gs; int B;
                  const char* xp = nullptr; long xp_gs = 0; };   // xp: x already packed for the 16-bit data path (group stride in bytes)
int conv_fwd(const float* x, long x_gs, const float* w, long w_gs, const float* bias, long b_gs,
             float* y, long y_gs, int G, int B, const Geom& g, int math, hipStream_t s,
             bool prezeroed = false, bool* query = nullptr, const char* packed_w = nullptr,
             char* xp_keep = nullptr, long xp_keep_gs = 0, FuseNorm* fuse = nullptr, bool x_prepacked = false);   // x_prepacked: xp_keep already holds x's channel octets (its producer wrote them)
int conv_dgrad(const float* dy, long dy_gs, const float* w, long w_gs, float* dx, long dx_gs,
               const float* addend, long add_gs, int accumulate, int G, int B, const Geom& g, int math, hipStream_t s,
               bool prezeroed = false, bool* query = nullptr, const char* const* packed_w = nullptr,
               const char* dy_packed = nullptr, long dy_packed_gs = 0);   // dy_packed: dy already as bf16 channel octets (its producer wrote them; group stride in bytes)
bool wino_eligible(const Geom& g, int math, int B);                        // fp32: forward / input gradient of this geometry take the Winograd F(4,3) route (vc_ops.hip)
bool winoh_eligible(const Geom& g, int math, int B);                       // ... or the F(3,6)-along-H route (D.d3)
bool wino_route(const Geom& g, int math, int B);                           // either
Geom wino_gemm_geom(const Geom& g);                                   // the 1 x 1 geometry of its six channel GEMMs
size_t wino_scratch_floats(const Geom& g, int G, int B);               // what one routed forward / input-gradient call takes from that block
int wino_workspace(hipStream_t s, size_t floats, float** out);        // transform-domain scratch of the launches issued on stream s (grows on demand, reused launch to launch)
int wino_gemm_fwd(const float* V, const float* U, float* M, int G, int B, const Geom& g, int math, hipStream_t s);
int wino_gemm_dgrad(const float* dM, const float* U, float* dV, int G, int B, const Geom& g, int math, hipStream_t s);
bool conv_dgrad_packs_input(const Geom& g, int B, int math);          // the input gradient of this geometry at batch B runs on the 16-bit data path
// bf16 data path: the layer's stacked pixel-octet dY [all applications' pixel octets][Cout][8] bf16 (group stride in bytes; room for every
// application of the launch) and which applications' parts their norm backward has already written (bit k = application k): the
// launch reads it in place and packs only the others
struct WgradPacked { char* dyq; long dyq_gs; size_t cap; unsigned ready; };   // cap: bytes of one group's buffer
int conv_wgrad(const WgradApp* apps, int napps, float* dw, long dw_gs, int G, const Geom& g, int math, hipStream_t s,
               bool store, bool* query = nullptr, const WgradPacked* pq = nullptr);
bool conv_wgrad_reads_pixel_octets(const Geom& g, long pixels, int math);   // the filter gradient of this geometry runs as the bf16 GEMM over `pixels` (>= one application's)
void drop_pack_spaces(std::initializer_list<hipStream_t> streams);
long ws_generation(std::initializer_list<hipStream_t> streams);
int poison_pack_spaces(std::initializer_list<hipStream_t> streams, int byte);

// ---- networks, plans, schedules (vc_net.hip)
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
    int exchange_mode = 0;        // vc_plan_set_exchange_mode: 0 bucketed (5 segments under the backward), 1 single (one all-reduce after it)
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
    // the same two moments PER LAYER (vc_tune_layer_updates(1): the single-GPU update runs ApplyAdam layer by layer in backward order, each
    // layer as soon as its gradient is final and its weights are no longer read; off by default, see include/vc_cyclegan_testing.h).
    // Discriminator weights have two readers: the discriminator_loss backward (side stream) and the generator_loss backward (main).
    hipEvent_t gl_ready[24] = {}, gl_free[24] = {}, dl_ready[8] = {}, dl_free_d[8] = {}, dl_free_g[8] = {};
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
    // fp32 mode: the Winograd-domain filters U = G w of the generators' 3-tap layers (vc_conv_wino.hip), both generators as the two groups,
    // rebuilt from the weights at the head of every train step on the side stream (a step uses each four times: two forward passes, two
    // input gradients); fwd = dg[0] = the layer's U.
    mutable std::vector<LayerImages> wino_img, wino_img_d;     // generators' layers; discriminators' (D.d3: F(3,6) along H)
    mutable float* wino_block = nullptr;
    mutable bool wino_active = false;      // the running step rebuilt them (its batch is large enough for the route)
    mutable hipStream_t wino_stream = nullptr;   // ... on this stream (a pass on the same stream needs no event wait)
    static constexpr int kWinoChunks = 3;  // the rebuild goes out as this many launches in layer order, an event behind each: the forward waits for a layer's chunk only
    hipEvent_t ev_wino[kWinoChunks] = {nullptr, nullptr, nullptr};
    mutable std::vector<int> wino_chunk, wino_chunk_d;   // per generator / discriminator layer: which launch carries its filters (-1: none)
};
struct SaveMap { std::vector<long> raw, act, st, actp; long total; };
SaveMap save_map(const NetDesc& n, const std::vector<LayerGeo>& geo, int B);
long scratch_floats_net(const std::vector<LayerGeo>& geo, int B, int math);
// An earlier application of the same network whose filter gradients were deferred (the cycle hop): the first-hop
// backward reduces both applications into dW with ONE launch per layer (WgradArgs segments).
struct DeferredApp { const float* x; long x_gs; const float* save; long save_gs; const SaveMap* sm;
                     const float* scratch; long scr_gs; const float* dtop; long dtop_gs; int B;
                     unsigned q_ready = 0, dy_dropped = 0; };   // bit i: layer i's dY of that application already lies as pixel octets in ITS scratch's stacked
                                                                // buffer / its fp32 copy was never written (BwdOpts::drop_fp32_dy)
enum { W_NONE = 0, W_SELF = 1 };
struct BwdOpts { bool record_buckets = false, wgrad_store = false; int wmode = W_SELF; const DeferredApp* deferred = nullptr;
                 // W_NONE passes report which layers' dY went out as pixel octets / were never written as fp32 (for the DeferredApp of the
                 // pass that reduces them)
                 unsigned* q_ready_out = nullptr; unsigned* dy_dropped_out = nullptr;
                 // the caller guarantees that every reader of a layer's dy takes the octet forms whenever the norm backward wrote them
                 // (inside a train step: yes): the fp32 copy is then not written at all
                 bool drop_fp32_dy = false;
                 bool prezeroed = false; };   // backward_prezero ran for this pass (same scratch, batch, groups) earlier in stream order
int gen_forward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                float* y, long y_gs, float* save, int G, int B, hipStream_t s);
int gen_backward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                 const float* dy, long dy_gs, float* save, int Bl, float* dparams, long dpar_gs,
                 float* dx, long dx_gs, int dx_acc, float* scratch, long scr_gs, int G, int B, hipStream_t s,
                 const BwdOpts& o = BwdOpts());
int disc_forward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                 float* o, long o_gs, float* save, int G, int B, hipStream_t s);
int disc_backward(const vc_plan* p, const float* params, long par_gs, const float* x, long x_gs,
                  const float* o, long o_gs, const float* dout, long do_gs, float* save, int Bl,
                  float* dparams, long dpar_gs, float* dx, long dx_gs, int dx_acc,
                  float* scratch, long scr_gs, int G, int B, hipStream_t s, bool wgrad_store = false, bool drop_fp32_dy = false,
                  bool record_layers = false, bool prezeroed = false);
int backward_prezero(const vc_plan* p, bool gen, float* scratch, long scr_gs, int G, int B, hipStream_t s);   // record_layers (train step): the plan's per-layer events dl_ready / dl_free_d (dparams) or dl_free_g
// The step's prebuilt filter images describe `params` of the running step: only the step hands them to its network passes
struct StepImages { explicit StepImages(const vc_plan* p); ~StepImages(); };
int refresh_wino_filters(const vc_plan* p, const float* params, int B, hipStream_t s, bool* active);   // B: the largest batch of the step's generator passes   // vc_net.hip; *active: the plan's passes will take them
