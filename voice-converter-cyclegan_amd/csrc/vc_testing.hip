// vc_testing.hip -- the test / tuning switches of include/vc_cyclegan_testing.h: one thread-local object, written only here.
// Product code never calls a setter (tests/test_abi_cpu.py checks the Python side; the C side has no caller outside this file);
// bench.py and model.CycleGAN.train refuse to run while vc_testing_hooks_active() != 0.
#include "vc_host.h"

static thread_local TestHooks t_hooks;
const TestHooks& vc_hooks() { return t_hooks; }

extern "C" int vc_tune_igemm_override(int bm, int kg, int splitk) {
    const bool ok = (bm == 0 && kg == 0 && splitk == 0) ||
                    ((bm == 64 || bm == 128) && (kg == 1 || kg == 2 || (kg == 4 && bm == 64)) && splitk >= 1 && splitk <= 64);
    if (!ok) { vc_set_error("vc_tune_igemm_override: tile %d, K groups %d, split-K %d is not a built configuration", bm, kg, splitk); return -1; }
    t_hooks.igemm_bm = bm; t_hooks.igemm_kg = kg; t_hooks.igemm_sk = splitk;
    return 0;
}
extern "C" int vc_tune_wgrad_override(int bn, int kg, int splitk) {
    const bool ok = (bn == 0 && kg == 0 && splitk == 0) ||
                    ((bn == 64 || bn == 128) && (kg == 1 || kg == 2 || (kg == 4 && bn == 64)) && splitk >= 1 && splitk <= 64);
    if (!ok) { vc_set_error("vc_tune_wgrad_override: tile columns %d, K groups %d, split-K %d is not a built configuration", bn, kg, splitk); return -1; }
    t_hooks.wgrad_bn = bn; t_hooks.wgrad_kg = kg; t_hooks.wgrad_sk = splitk;
    return 0;
}
extern "C" int vc_tune_conv_bf16_override(int tile_rows, int stages, int splitk) {
    const bool ok = (tile_rows == 0 && stages == 0 && splitk == 0) ||
                    ((((tile_rows == 64 || tile_rows == 128) && (stages == 2 || stages == 4)) || (tile_rows == 256 && stages == 3)) && splitk >= 1 && splitk <= 16);
    if (!ok) { vc_set_error("vc_tune_conv_bf16_override: tile rows %d, stages %d, split-K %d is not a built configuration", tile_rows, stages, splitk); return -1; }
    t_hooks.bf16_bm = tile_rows; t_hooks.bf16_nst = stages; t_hooks.bf16_sk = splitk;
    return 0;
}
extern "C" int vc_tune_conv_bf16_spec(int on) { t_hooks.bf16_spec = on < 0 ? -1 : (on ? 1 : 0); return 0; }
extern "C" int vc_tune_bf16_pixel_octets(int on) { t_hooks.pixel_octets = on; return 0; }   // 2: emit, but keep writing the fp32 copy
extern "C" int vc_tune_rowskip(int wgrad_on) { t_hooks.rowskip_wgrad = wgrad_on ? 1 : 0; return 0; }
extern "C" int vc_tune_wgrad_store(int on) { t_hooks.wgrad_store = on ? 1 : 0; return 0; }
extern "C" int vc_tune_glu(int flags) { t_hooks.glu_flags = flags; return 0; }
extern "C" int vc_tune_c1_dgrad_gather(int on) { t_hooks.c1_gather = on ? 1 : 0; return 0; }
extern "C" int vc_tune_layer_updates(int on) { t_hooks.layer_updates = on ? 1 : 0; return 0; }
extern "C" int vc_tune_winograd(int on) { t_hooks.winograd = on < 0 ? 0 : on > 2 ? 2 : on; return 0; }
static thread_local long long t_wino_gemms = 0;
static thread_local double t_wino_saved[2] = {0.0, 0.0};
void vc_count_wino_gemm(int kind, double saved_flops) { ++t_wino_gemms; t_wino_saved[kind ? 1 : 0] += saved_flops; }
// multiply-add work the Winograd route did NOT execute (direct form minus transform-domain GEMMs) since the last reset: kind 0 forward + input
// gradient launches, 1 filter gradients.  bench.py reports the kernels' executed FLOP/s (the MFMA roofline) and, beside it, the direct-equivalent rate.
extern "C" double vc_testing_wino_saved_flops(int kind, int reset) { const double v = t_wino_saved[kind ? 1 : 0]; if (reset) t_wino_saved[kind ? 1 : 0] = 0.0; return v; }
extern "C" long long vc_testing_wino_gemms(int reset) { const long long n = t_wino_gemms; if (reset) t_wino_gemms = 0; return n; }
extern "C" int vc_testing_hooks_active(void);
extern "C" int vc_thread_is_product_run(void) { return vc_testing_hooks_active() == 0; }
// bit per switch that is NOT at its production value in the calling thread (0: a product run)
extern "C" int vc_testing_hooks_active(void) {
    const TestHooks d{};
    const TestHooks& h = t_hooks;
    return (h.igemm_bm != d.igemm_bm ? 1 : 0) | (h.wgrad_bn != d.wgrad_bn ? 2 : 0) | (h.bf16_bm != d.bf16_bm ? 4 : 0) | (h.bf16_spec != d.bf16_spec ? 8 : 0) |
           (h.pixel_octets != d.pixel_octets ? 16 : 0) | (h.rowskip_wgrad != d.rowskip_wgrad ? 32 : 0) | (h.wgrad_store != d.wgrad_store ? 64 : 0) |
           (h.glu_flags != d.glu_flags ? 128 : 0) | (h.c1_gather != d.c1_gather ? 256 : 0) | (h.layer_updates != d.layer_updates ? 512 : 0) | (h.winograd != d.winograd ? 1024 : 0);
}
