// Measured configurations of conv_bf16_kernel (the 16-bit data path of the bf16 mode): tools/tune_igemm.py --math bf16 --batch 64
// --sweep on MI355X (round 3; profiles/r03_tune_sweep_bf16_b64.txt), BASELINE configs[2]'s launch shapes (forward, input gradient,
// and the filter gradient as a GEMM over the pixel axis).  Only shapes where the sweep beat the tile-count rule of
// conv_bf16_config by more than the run-to-run spread (4 %) are listed.  {M, Ptot, K octets, groups, tile rows, LDS ring stages,
// split-K}.
static const TunedBf16 kTunedBf16[] = {
    // ---- batch 64 per GPU, 24x128 MCEP
    // g_d1_3B wgrad: 46.0 us (rule of conv_bf16_config: 54.9 us)
    {640, 512, 1536, 2, 128, 4, 4},
    // g_d2 fwd: 23.6 us (rule of conv_bf16_config: 26.4 us)
    {1024, 2048, 160, 2, 64, 2, 1},
    // g_d2_3B wgrad: 60.8 us (rule of conv_bf16_config: 70.4 us)
    // g_r_h1 dgrad: 46.7 us (rule of conv_bf16_config: 53.6 us)
    // g_r_h1_2B dgrad: 67.1 us (rule of conv_bf16_config: 72.5 us)
    // g_r_h2 dgrad: 24.1 us (rule of conv_bf16_config: 25.6 us)
    // g_r_h2_2B fwd: 39.7 us (rule of conv_bf16_config: 43.9 us)
    // g_r_h2_3B wgrad: 60.4 us (rule of conv_bf16_config: 70.6 us)
    // g_u1 dgrad: 63.9 us (rule of conv_bf16_config: 81.4 us)
    // g_u1_2B dgrad: 101.4 us (rule of conv_bf16_config: 110.2 us)
    // g_u2 dgrad: 56.1 us (rule of conv_bf16_config: 59.7 us)
    // g_u2_3B wgrad: 162.5 us (rule of conv_bf16_config: 175.1 us)
    {2560, 1024, 1536, 2, 256, 3, 1},        // round 5: 125.1 us
    // d_d1_2B wgrad: 223.4 us (rule of conv_bf16_config: 362.8 us)
    {1152, 512, 6144, 2, 256, 3, 4},         // round 5: 183.0 us
    // d_d2 dgrad: 139.0 us (rule of conv_bf16_config: 153.6 us)
    // d_d3 dgrad: 298.8 us (rule of conv_bf16_config: 396.3 us)
    // d_d3_2B fwd: 425.4 us (rule of conv_bf16_config: 480.3 us)
    {2048, 6144, 1152, 2, 128, 2, 1},
    // ---- round 5 re-sweep with the wave-specialised 256-row tiles (batch 64): shapes the tile-count rule sends to smaller tiles
    // g_u1_3B wgrad: 150.1 us (rule: 162.4 us)
    // d_d2_2B wgrad: 155.5 us (rule: 186.9 us)
    {2304, 1024, 1536, 2, 256, 3, 1},
    // ---- round 5, second sweep: the 4-stage ring (one workgroup per CU) runs wave-specialised too; 128-row tiles on it beat the
    // two-workgroup configurations on the M = 512 / 256 shapes (tools/tune_igemm.py --math bf16 --batch 64 --sweep; us: new (before))
    {1280, 1024, 768, 2, 128, 4, 1},         // g_d2_3B wgrad 50.0 us (was 55.5)
    {512, 2048, 768, 2, 128, 4, 2},         // g_r_h1 dgrad 41.7 us (47.7)
    {512, 4096, 768, 2, 128, 4, 1},         // g_r_h1_2B dgrad 55.6 us (67.7)
    {1024, 2048, 192, 2, 128, 4, 1},         // g_r_h2 dgrad 21.8 us (23.3)
    {512, 4096, 384, 2, 128, 4, 1},         // g_r_h2_2B fwd 34.4 us (40.5)
    {3072, 512, 768, 2, 128, 4, 1},         // g_r_h2_3B wgrad 48.0 us (62.2)
    {512, 2048, 1280, 2, 128, 4, 2},         // g_u1 dgrad 56.4 us (64.7)
    {512, 4096, 1280, 2, 128, 4, 1},         // g_u1_2B dgrad 84.3 us (102.9)
    {2560, 2048, 768, 2, 128, 4, 1},         // g_u1_3B wgrad 151.7 us (158.6)
    {512, 4096, 640, 2, 128, 4, 1},         // g_u2 dgrad 47.2 us (56.0)
    {256, 6144, 128, 2, 128, 4, 1},         // d_d2 dgrad 111.8 us (134.2)
    {256, 6144, 256, 2, 128, 4, 1},
    {256, 6144, 512, 2, 128, 4, 1},
    {512, 3072, 1536, 2, 128, 4, 1},         // d_d3 dgrad 256.5 us (280.8)
    {512, 3072, 3072, 2, 128, 4, 1},
};
