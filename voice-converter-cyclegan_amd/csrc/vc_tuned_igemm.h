// Measured implicit-GEMM configurations: tools/tune_igemm.py --sweep on MI355X (round 2, batch 8 per GPU: the launch
// shapes of BASELINE configs[1]; profiles/r02_tune_sweep_b8.txt).  {a_trans, M, Ptot, krows, groups, tile rows, K groups,
// split-K}.  Shapes not listed go by the cost model in igemm_config.
static const TunedCfg kTunedIgemm[] = {
    // g_d1 fwd: 20.2 us (cost model: 21.8 us)
    {0, 512, 512, 640, 2, 64, 2, 2},
    // g_d1 dgrad: 17.8 us (cost model: 21.0 us)
    {1, 128, 512, 1024, 2, 64, 2, 4},
    {1, 128, 512, 1536, 2, 64, 2, 4},
    // g_d1_2B fwd: 21.7 us (cost model: 23.2 us)
    {0, 512, 1024, 640, 2, 64, 2, 1},
    // g_d1_2B dgrad: 23.1 us (cost model: 31.6 us)
    {1, 128, 1024, 1024, 2, 64, 2, 2},
    {1, 128, 1024, 1536, 2, 64, 2, 2},
    // g_d2 fwd: 24.8 us (cost model: 32.9 us)
    {0, 1024, 256, 1280, 2, 64, 2, 2},
    // g_d2 dgrad: 23.9 us (cost model: 32.2 us)
    {1, 256, 256, 2048, 2, 64, 4, 4},
    {1, 256, 256, 3072, 2, 64, 4, 4},
    // g_d2_2B fwd: 32.6 us (cost model: 35.3 us)
    {0, 1024, 512, 1280, 2, 64, 4, 1},
    // g_d2_2B dgrad: 36.7 us (cost model: 35.8 us)
    {1, 256, 512, 2048, 2, 64, 4, 2},
    {1, 256, 512, 3072, 2, 64, 4, 2},
    // g_r_h1 fwd: 35.7 us (cost model: 39.3 us)
    {0, 2048, 256, 1536, 2, 64, 4, 1},
    // g_r_h1 dgrad: 38.1 us (cost model: 36.2 us)
    {1, 512, 256, 6144, 2, 64, 4, 4},
    // g_r_h1_2B fwd: 59.8 us (cost model: 66.4 us)
    {0, 2048, 512, 1536, 2, 64, 2, 1},
    // g_r_h1_2B dgrad: 60.0 us (cost model: 60.0 us)
    {1, 512, 512, 6144, 2, 64, 4, 2},
    // g_r_h2 fwd: 25.8 us (cost model: 27.4 us)
    {0, 512, 256, 3072, 2, 64, 4, 4},
    // g_r_h2 dgrad: 23.0 us (cost model: 31.6 us)
    {1, 1024, 256, 1536, 2, 64, 2, 2},
    // g_r_h2_2B fwd: 37.9 us (cost model: 41.3 us)
    {0, 512, 512, 3072, 2, 64, 4, 2},
    // g_r_h2_2B dgrad: 32.8 us (cost model: 33.4 us)
    {1, 1024, 512, 1536, 2, 64, 4, 1},
    // g_u1 fwd: 52.0 us (cost model: 58.7 us)
    {0, 2048, 256, 2560, 2, 64, 4, 1},
    // g_u1 dgrad: 55.7 us (cost model: 52.5 us)
    {1, 512, 256, 10240, 2, 64, 2, 8},
    // g_u1_2B fwd: 91.3 us (cost model: 103.1 us)
    {0, 2048, 512, 2560, 2, 64, 2, 1},
    // g_u1_2B dgrad: 93.9 us (cost model: 93.3 us)
    {1, 512, 512, 10240, 2, 64, 2, 4},
    // g_u2 fwd: 51.6 us (cost model: 58.9 us)
    {0, 1024, 512, 2560, 2, 64, 4, 1},
    // g_u2 dgrad: 51.6 us (cost model: 51.4 us)
    {1, 512, 512, 5120, 2, 64, 2, 4},
    // g_u2_2B fwd: 90.4 us (cost model: 104.1 us)
    {0, 1024, 1024, 2560, 2, 64, 2, 1},
    // g_u2_2B dgrad: 88.1 us (cost model: 93.8 us)
    {1, 512, 1024, 5120, 2, 64, 2, 2},
    // g_o1 fwd: 20.3 us (cost model: 19.8 us)
    {0, 24, 1024, 3840, 2, 64, 2, 8},
    // g_o1 dgrad: 15.7 us (cost model: 18.0 us)
    {1, 256, 1024, 512, 2, 64, 2, 2},
    // g_o1_2B fwd: 28.7 us (cost model: 29.4 us)
    {0, 24, 2048, 3840, 2, 64, 2, 8},
    // g_o1_2B dgrad: 18.5 us (cost model: 18.4 us)
    {1, 256, 2048, 512, 2, 64, 4, 1},
    // g_h1 dgrad: 18.8 us (cost model: 18.6 us)
    {1, 24, 1024, 3840, 2, 64, 2, 8},
    // g_h1_2B dgrad: 27.0 us (cost model: 27.7 us)
    {1, 24, 2048, 3840, 2, 64, 4, 4},
    // d_d1 fwd: 72.1 us (cost model: 78.3 us)
    {0, 512, 3072, 1152, 2, 64, 1, 1},
    // d_d1 dgrad: 76.4 us (cost model: 73.7 us)
    {1, 128, 3072, 2048, 2, 64, 1, 1},
    {1, 128, 3072, 1024, 2, 64, 1, 1},
    {1, 128, 3072, 512, 2, 64, 1, 1},
    // d_d1_2B fwd: 135.1 us (cost model: 155.1 us)
    {0, 512, 6144, 1152, 2, 64, 1, 1},
    // d_d1_2B dgrad: 132.3 us (cost model: 128.0 us)
    {1, 128, 6144, 2048, 2, 64, 1, 1},
    {1, 128, 6144, 1024, 2, 64, 1, 1},
    {1, 128, 6144, 512, 2, 64, 1, 1},
    // d_d2 fwd: 73.2 us (cost model: 90.8 us)
    {0, 1024, 768, 2304, 2, 64, 1, 2},
    // d_d2 dgrad: 76.8 us (cost model: 78.8 us)
    {1, 256, 768, 4096, 2, 64, 1, 2},
    {1, 256, 768, 2048, 2, 64, 1, 2},
    {1, 256, 768, 1024, 2, 64, 1, 2},
    // d_d2_2B fwd: 126.0 us (cost model: 143.3 us)
    {0, 1024, 1536, 2304, 2, 64, 1, 1},
    // d_d2_2B dgrad: 135.4 us (cost model: 135.1 us)
    {1, 256, 1536, 4096, 2, 64, 2, 1},
    {1, 256, 1536, 2048, 2, 64, 2, 1},
    {1, 256, 1536, 1024, 2, 64, 2, 1},
    // d_d3 fwd: 237.7 us (cost model: 308.0 us)
    {0, 2048, 384, 9216, 2, 64, 1, 4},
    // d_d3 dgrad: 254.4 us (cost model: 301.3 us)
    {1, 512, 384, 24576, 2, 64, 1, 8},
    {1, 512, 384, 12288, 2, 64, 1, 8},
    // d_d3_2B fwd: 458.1 us (cost model: 583.2 us)
    {0, 2048, 768, 9216, 2, 64, 1, 4},
    // d_d3_2B dgrad: 455.5 us (cost model: 468.1 us)
    {1, 512, 768, 24576, 2, 128, 1, 8},
    {1, 512, 768, 12288, 2, 128, 1, 8},
};
