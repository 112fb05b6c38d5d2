// vc_comm.hip -- the ONE collective of the data-parallel train step: a SUM all-reduce of the flat fp32 gradient over
// RCCL (xGMI), issued per gradient segment as the native schedule finishes it, followed by ApplyAdam on that segment.
// The reference is single-process (model.py:22 `tf.Session()`); SURVEY 8e: plain data parallel over the minibatch,
// replicas identical, gradient of the global mean = mean of the shard gradients.
//
// RCCL is bound at run time with dlopen("librccl.so.1"): the library then shares whatever RCCL the process already
// loaded (PyTorch's), and a single-GPU user never needs RCCL at all.
#include "vc_internal.h"
#include "../../include/vc_cyclegan.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>

#define VC_ARG(cond, ...) do { if (!(cond)) { vc_set_error(__VA_ARGS__); return VC_E_ARG; } } while (0)

namespace {
typedef struct { char internal[128]; } rcclUniqueId;            // ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128
typedef void* rcclComm_t;
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(rcclComm_t, int*) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
const Rccl* rccl() {
    std::call_once(g_rccl_once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        Rccl r; r.h = h;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
        if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GroupStart && r.GroupEnd) g_rccl = r;
    });
    return g_rccl.h ? &g_rccl : nullptr;
}
const int kFloat32 = 7, kSum = 0;                                // ncclFloat32, ncclSum (rccl.h)
}  // namespace

struct vc_comm { rcclComm_t comm; int rank, world; mutable long n_allreduce = 0; };   // n_allreduce: collectives enqueued so far

#define VC_RCCL(expr) do { int _e = (expr); if (_e != 0) { \
    vc_set_error("%s failed: %s", #expr, R->GetErrorString ? R->GetErrorString(_e) : "rccl error"); return 1000 + _e; } } while (0)

extern "C" int vc_comm_unique_id(void* id128) {
    VC_ARG(id128, "vc_comm_unique_id: id is NULL");
    const Rccl* R = rccl();
    if (!R) { const char* why = dlerror(); vc_set_error("RCCL (librccl.so.1) could not be loaded: %s", why ? why : "no such library"); return VC_E_ARG; }
    rcclUniqueId id;
    VC_RCCL(R->GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return 0;
}
extern "C" int vc_comm_init(int rank, int world, const void* id128, vc_comm** out) {
    VC_ARG(out && id128 && world >= 1 && rank >= 0 && rank < world, "vc_comm_init: bad args");
    const Rccl* R = rccl();
    VC_ARG(R, "RCCL (librccl.so.1) could not be loaded");
    rcclUniqueId id; memcpy(&id, id128, sizeof id);
    rcclComm_t c = nullptr;
    VC_RCCL(R->CommInitRank(&c, world, id, rank));               // on the calling thread's current HIP device
    *out = new vc_comm{c, rank, world, 0};
    return 0;
}
extern "C" int vc_comm_destroy(vc_comm* c) {
    if (!c) return 0;
    const Rccl* R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
    return 0;
}
extern "C" int vc_comm_world(const vc_comm* c) {
    if (!c) return 1;
    const Rccl* R = rccl();
    int n = c->world;
    if (R && R->CommCount) (void)R->CommCount(c->comm, &n);       // what RCCL itself saw
    return n;
}
extern "C" int64_t vc_comm_collectives(const vc_comm* c) { return c ? c->n_allreduce : 0; }
extern "C" int vc_allreduce_f32(const vc_comm* c, float* buf, int64_t n, void* stream) {
    VC_ARG(c && buf && n > 0, "vc_allreduce_f32: bad args");
    const Rccl* R = rccl();
    VC_ARG(R, "RCCL not loaded");
    VC_RCCL(R->AllReduce(buf, buf, (size_t)n, kFloat32, kSum, c->comm, (hipStream_t)stream));
    ++c->n_allreduce;
    return 0;
}

// Gradient exchange + ApplyAdam (model.py:107-108), segment by segment on `update_stream`, each segment as soon as the
// step's own events say its gradients are final AND its weights are no longer read: the discriminator half first (it is
// final before the generator backward starts), then the generator buckets in reverse layer order.  Each segment is
// all-reduced (SUM over the ranks of `comm`; comm == NULL: single GPU, no collective) and then updated in place with the
// 1/world factor folded into Adam.  All but the last small bucket ride under the rest of the backward.
bool layer_events_on();                                    // vc_net.hip
int step_num_layer_segments(const vc_plan* p);             // vc_step.hip
int step_layer_segment(const vc_plan* p, int k, int64_t* lo0, int64_t* lo1, int64_t* n, int* disc, hipStream_t us);
int step_update_impl(const vc_plan* p, const vc_comm* comm, float* params, float* grads, float* m, float* v,
                     float alpha_g, float alpha_d, const float* hp, float beta1, float beta2, float eps, hipStream_t us) {
    VC_ARG(p && params && grads && m && v, "vc_step_exchange_and_update: bad args");
    // a communicator means "exchange": also on a single-rank communicator the collectives are issued (RCCL reduces over one
    // rank: the identity), so the path a multi-GPU job takes is the one the one-GPU tests execute
    const Rccl* R = comm ? rccl() : nullptr;
    VC_ARG(!comm || R, "RCCL not loaded");
    const float gscale = comm ? 1.0f / (float)comm->world : 1.0f;
    const int64_t g0 = vc_flat_offset(p, 0), g1 = vc_flat_offset(p, 1), d0 = vc_flat_offset(p, 2), total = vc_flat_param_floats(p);
    auto seg = [&](int64_t lo, int64_t hi, bool disc) -> int {        // hp: {.., .., alpha_g, alpha_d} on the device (captured steps)
        return launch_adam(params + lo, m + lo, v + lo, grads + lo, hi - lo, disc ? alpha_d : alpha_g, beta1, beta2, eps, gscale, us,
                           hp ? hp + (disc ? 3 : 2) : nullptr);
    };
    if (vc_plan_get_exchange_mode(p) == VC_EXCHANGE_SINGLE) {
        // the north star's literal form: ONE all-reduce of the whole flat gradient (479 MB) once the backward has finished, then both
        // updates.  Nothing of the exchange is hidden under the backward: the A/B partner of the bucketed form below.
        { int r = vc_step_wait_d_grads(p, us); if (r) return r; }
        for (int k = 0; k < vc_step_num_g_buckets(p); ++k) { int r = vc_step_wait_g_bucket(p, k, us); if (r) return r; }
        if (R) { VC_RCCL(R->AllReduce(grads, grads, (size_t)total, kFloat32, kSum, comm->comm, us)); ++comm->n_allreduce; }
        { int r = seg(g0, d0, false); if (r) return r; }
        return seg(d0, total, true);
    }
    if (!comm && layer_events_on()) {
        // one GPU, nothing to exchange: ApplyAdam layer by layer in the order the backward finishes them (D.d3 .. D.h1, G.o1 .. G.h1),
        // each layer of both networks of a kind as soon as its gradient is final and its weights are no longer read.  Same arithmetic
        // as the segmented form (Adam is elementwise); what is left after the backward has ended is G.h1's 0.4 MB instead of a bucket.
        for (int k = 0; k < step_num_layer_segments(p); ++k) {
            int64_t lo0 = 0, lo1 = 0, n = 0; int disc = 0;
            { int r = step_layer_segment(p, k, &lo0, &lo1, &n, &disc, us); if (r) return r; }
            { int r = seg(lo0, lo0 + n, disc != 0); if (r) return r; }
            { int r = seg(lo1, lo1 + n, disc != 0); if (r) return r; }
        }
        return 0;
    }
    // discriminator half: one contiguous range
    { int r = vc_step_wait_d_grads(p, us); if (r) return r; }
    if (R) { VC_RCCL(R->AllReduce(grads + d0, grads + d0, (size_t)(total - d0), kFloat32, kSum, comm->comm, us)); ++comm->n_allreduce; }
    { int r = seg(d0, total, true); if (r) return r; }
    // generator buckets: the same layer range of both generators, one collective group
    for (int k = 0; k < vc_step_num_g_buckets(p); ++k) {
        int64_t lo = 0, hi = 0;
        { int r = vc_step_g_bucket(p, k, &lo, &hi); if (r) return r; }
        { int r = vc_step_wait_g_bucket(p, k, us); if (r) return r; }
        if (R) {
            VC_RCCL(R->GroupStart());
            int e = R->AllReduce(grads + g0 + lo, grads + g0 + lo, (size_t)(hi - lo), kFloat32, kSum, comm->comm, us);
            if (!e) e = R->AllReduce(grads + g1 + lo, grads + g1 + lo, (size_t)(hi - lo), kFloat32, kSum, comm->comm, us);
            const int e2 = R->GroupEnd();                                // always closed, also after a failed enqueue
            if (!e && !e2) comm->n_allreduce += 2;
            if (e || e2) { vc_set_error("ncclAllReduce (generator bucket %d) failed: %s", k, R->GetErrorString ? R->GetErrorString(e ? e : e2) : "rccl error"); return 1000 + (e ? e : e2); }
        }
        { int r = seg(g0 + lo, g0 + hi, false); if (r) return r; }
        { int r = seg(g1 + lo, g1 + hi, false); if (r) return r; }
    }
    return 0;
}
extern "C" int vc_step_exchange_and_update(const vc_plan* p, const vc_comm* comm, float* params, float* grads, float* m, float* v,
                                           float alpha_g, float alpha_d, float beta1, float beta2, float eps, void* update_stream) {
    return step_update_impl(p, comm, params, grads, m, v, alpha_g, alpha_d, nullptr, beta1, beta2, eps, (hipStream_t)update_stream);
}
