// Measured implicit-GEMM configurations: tools/tune_igemm.py --sweep on MI355X (round 3; profiles/r03_tune_sweep_*.txt), for the
// launch shapes of the configurations bench.py reports: batch 8 (BASELINE configs[1], the headline), batch 32 (configs[3]'s
// per-GPU share of global batch 256 on 8 GPUs), 36x256 at batch 64 (configs[4]'s share) and batch 1 (configs[0]; round 4).  {a_trans, M, Ptot, krows,
// groups, tile rows, K groups, split-K}.  Shapes not listed go by the cost model in igemm_config.
// Keys {a_trans, M, Ptot, krows, groups} are UNIQUE (tests/test_tuned_tables_cpu.py): sections measured at different batch sizes produce the
// same launch shape now and then (a 2B launch at batch 4 is a 1B launch at batch 8); igemm_config takes the first match, so the later
// duplicates -- 196 rows, 65 of them with another winner from another box -- were dead and have been removed.
static const TunedCfg kTunedIgemm[] = {
    // ---- batch 8 per GPU, 24x128 MCEP (re-measured on the round-4 kernels, profiles/r04_tune_sweep_*.txt)
    // g_d1 fwd: 19.0 us (automatic choice before this sweep: 19.3 us)
    {0, 512, 512, 640, 2, 64, 2, 2},
    // g_d1 dgrad: 17.0 us (automatic choice before this sweep: 16.9 us)
    {1, 128, 512, 1024, 2, 64, 2, 4},
    {1, 128, 512, 1536, 2, 64, 2, 4},
    // g_d1_2B fwd: 20.1 us (automatic choice before this sweep: 20.1 us)
    {0, 512, 1024, 640, 2, 64, 2, 1},
    // g_d1_2B dgrad: 21.6 us (automatic choice before this sweep: 22.2 us)
    {1, 128, 1024, 1024, 2, 64, 2, 2},
    {1, 128, 1024, 1536, 2, 64, 2, 2},
    // g_d2 fwd: 23.6 us (automatic choice before this sweep: 24.3 us)
    {0, 1024, 256, 1280, 2, 64, 2, 2},
    // g_d2 dgrad: 22.3 us (automatic choice before this sweep: 22.3 us)
    {1, 256, 256, 2048, 2, 64, 2, 4},
    {1, 256, 256, 3072, 2, 64, 2, 4},
    // g_d2_2B fwd: 30.2 us (automatic choice before this sweep: 30.0 us)
    {0, 1024, 512, 1280, 2, 64, 4, 1},
    // g_d2_2B dgrad: 34.8 us (automatic choice before this sweep: 34.7 us)
    {1, 256, 512, 2048, 2, 64, 4, 2},
    {1, 256, 512, 3072, 2, 64, 4, 2},
    // g_r_h1 fwd: 33.2 us (automatic choice before this sweep: 32.9 us)
    {0, 2048, 256, 1536, 2, 64, 4, 1},
    // g_r_h1 dgrad: 34.0 us (automatic choice before this sweep: 34.1 us)
    {1, 512, 256, 6144, 2, 64, 4, 4},
    // g_r_h1_2B fwd: 56.2 us (automatic choice before this sweep: 56.2 us)
    {0, 2048, 512, 1536, 2, 64, 2, 1},
    // g_r_h1_2B dgrad: 55.5 us (automatic choice before this sweep: 57.0 us)
    {1, 512, 512, 6144, 2, 64, 2, 4},
    // g_r_h2 fwd: 23.9 us (automatic choice before this sweep: 24.1 us)
    {0, 512, 256, 3072, 2, 64, 4, 4},
    // g_r_h2 dgrad: 21.5 us (automatic choice before this sweep: 21.5 us)
    {1, 1024, 256, 1536, 2, 64, 2, 2},
    // g_r_h2_2B fwd: 35.4 us (automatic choice before this sweep: 35.0 us)
    {0, 512, 512, 3072, 2, 64, 4, 2},
    // g_r_h2_2B dgrad: 32.0 us (automatic choice before this sweep: 32.2 us)
    {1, 1024, 512, 1536, 2, 64, 4, 1},
    // g_u1 fwd: 49.2 us (automatic choice before this sweep: 49.1 us)
    {0, 2048, 256, 2560, 2, 64, 4, 1},
    // g_u1 dgrad: 49.5 us (automatic choice before this sweep: 49.8 us)
    {1, 512, 256, 10240, 2, 64, 2, 8},
    // g_u1_2B fwd: 86.9 us (automatic choice before this sweep: 87.9 us)
    {0, 2048, 512, 2560, 2, 128, 2, 1},
    // g_u1_2B dgrad: 86.1 us (automatic choice before this sweep: 88.5 us)
    {1, 512, 512, 10240, 2, 64, 2, 4},
    // g_u2 fwd: 48.9 us (automatic choice before this sweep: 49.1 us)
    {0, 1024, 512, 2560, 2, 64, 4, 1},
    // g_u2 dgrad: 48.1 us (automatic choice before this sweep: 48.5 us)
    {1, 512, 512, 5120, 2, 64, 2, 4},
    // g_u2_2B fwd: 87.6 us (automatic choice before this sweep: 87.4 us)
    {0, 1024, 1024, 2560, 2, 64, 2, 1},
    // g_u2_2B dgrad: 84.8 us (automatic choice before this sweep: 84.4 us)
    {1, 512, 1024, 5120, 2, 64, 2, 2},
    // g_o1 fwd: 17.7 us (automatic choice before this sweep: 17.5 us)
    {0, 24, 1024, 3840, 2, 64, 4, 8},
    // g_o1 dgrad: 14.3 us (automatic choice before this sweep: 14.2 us)
    {1, 256, 1024, 512, 2, 64, 1, 2},
    // g_o1_2B fwd: 24.8 us (automatic choice before this sweep: 25.2 us)
    {0, 24, 2048, 3840, 2, 64, 1, 16},
    // g_o1_2B dgrad: 16.6 us (automatic choice before this sweep: 16.2 us)
    {1, 256, 2048, 512, 2, 64, 2, 1},
    // g_h1 dgrad: 16.4 us (automatic choice before this sweep: 16.2 us)
    {1, 24, 1024, 3840, 2, 64, 1, 16},
    // g_h1_2B dgrad: 23.8 us (automatic choice before this sweep: 24.5 us)
    {1, 24, 2048, 3840, 2, 64, 1, 8},
    // d_d1 dgrad: 70.8 us (automatic choice before this sweep: 70.5 us)
    {1, 128, 3072, 2048, 2, 64, 1, 1},
    {1, 128, 3072, 1024, 2, 64, 1, 1},
    {1, 128, 3072, 512, 2, 64, 1, 1},
    // d_d1_2B fwd: 115.1 us (automatic choice before this sweep: 122.0 us)
    {0, 512, 6144, 1152, 2, 128, 1, 1},
    // d_d1_2B dgrad: 119.7 us (automatic choice before this sweep: 118.8 us)
    {1, 128, 6144, 2048, 2, 128, 1, 1},
    {1, 128, 6144, 1024, 2, 128, 1, 1},
    {1, 128, 6144, 512, 2, 128, 1, 1},
    // d_d2 dgrad: 73.6 us (automatic choice before this sweep: 73.4 us)
    {1, 256, 768, 4096, 2, 64, 1, 2},
    {1, 256, 768, 2048, 2, 64, 1, 2},
    {1, 256, 768, 1024, 2, 64, 1, 2},
    // d_d2_2B fwd: 121.0 us (automatic choice before this sweep: 122.7 us)
    {0, 1024, 1536, 2304, 2, 64, 1, 1},
    // d_d2_2B dgrad: 129.5 us (automatic choice before this sweep: 129.8 us)
    {1, 256, 1536, 4096, 2, 64, 2, 1},
    {1, 256, 1536, 2048, 2, 64, 2, 1},
    {1, 256, 1536, 1024, 2, 64, 2, 1},
    // d_d3 dgrad: 239.7 us (automatic choice before this sweep: 243.7 us)
    {1, 512, 384, 24576, 2, 64, 1, 8},
    {1, 512, 384, 12288, 2, 64, 1, 8},
    // d_d3_2B fwd: 430.0 us (automatic choice before this sweep: 465.2 us)
    {0, 2048, 768, 9216, 2, 128, 1, 2},
    // d_d3_2B dgrad: 442.8 us (automatic choice before this sweep: 441.2 us)
    {1, 512, 768, 24576, 2, 64, 1, 4},
    {1, 512, 768, 12288, 2, 64, 1, 4},
    // ---- batch 32 per GPU, 24x128 MCEP (re-measured on the round-4 kernels, profiles/r04_tune_sweep_*.txt)
    // g_d1 fwd: 28.6 us (automatic choice before this sweep: 29.0 us)
    {0, 512, 2048, 640, 2, 128, 2, 1},
    // g_d1 dgrad: 31.7 us (automatic choice before this sweep: 31.8 us)
    {1, 128, 2048, 1024, 2, 64, 4, 1},
    {1, 128, 2048, 1536, 2, 64, 4, 1},
    // g_d1_2B fwd: 48.5 us (automatic choice before this sweep: 47.9 us)
    {0, 512, 4096, 640, 2, 128, 1, 1},
    // g_d1_2B dgrad: 54.1 us (automatic choice before this sweep: 55.1 us)
    {1, 128, 4096, 1024, 2, 128, 2, 1},
    {1, 128, 4096, 1536, 2, 128, 2, 1},
    // g_d2 fwd: 49.6 us (automatic choice before this sweep: 51.0 us)
    {0, 1024, 1024, 1280, 2, 128, 2, 1},
    // g_d2 dgrad: 56.0 us (automatic choice before this sweep: 56.6 us)
    {1, 256, 1024, 2048, 2, 64, 4, 1},
    {1, 256, 1024, 3072, 2, 64, 4, 1},
    // g_d2_2B fwd: 85.4 us (automatic choice before this sweep: 91.6 us)
    {0, 1024, 2048, 1280, 2, 128, 1, 1},
    // g_d2_2B dgrad: 97.7 us (automatic choice before this sweep: 97.3 us)
    {1, 256, 2048, 2048, 2, 64, 2, 2},
    {1, 256, 2048, 3072, 2, 64, 2, 2},
    // g_r_h1 fwd: 100.0 us (automatic choice before this sweep: 101.5 us)
    {0, 2048, 1024, 1536, 2, 128, 1, 1},
    // g_r_h1 dgrad: 100.8 us (automatic choice before this sweep: 101.1 us)
    {1, 512, 1024, 6144, 2, 64, 2, 2},
    // g_r_h1_2B fwd: 191.9 us (automatic choice before this sweep: 207.1 us)
    {0, 2048, 2048, 1536, 2, 128, 1, 1},
    // g_r_h1_2B dgrad: 187.6 us (automatic choice before this sweep: 190.6 us)
    {1, 512, 2048, 6144, 2, 128, 2, 1},
    // g_r_h2 fwd: 56.9 us (automatic choice before this sweep: 57.2 us)
    {0, 512, 1024, 3072, 2, 64, 4, 1},
    // g_r_h2 dgrad: 54.1 us (automatic choice before this sweep: 54.2 us)
    {1, 1024, 1024, 1536, 2, 64, 1, 1},
    // g_r_h2_2B fwd: 103.5 us (automatic choice before this sweep: 102.3 us)
    {0, 512, 2048, 3072, 2, 64, 2, 1},
    // g_r_h2_2B dgrad: 97.3 us (automatic choice before this sweep: 97.4 us)
    {1, 1024, 2048, 1536, 2, 128, 1, 1},
    // g_u1 fwd: 160.3 us (automatic choice before this sweep: 162.1 us)
    {0, 2048, 1024, 2560, 2, 128, 1, 1},
    // g_u1 dgrad: 159.9 us (automatic choice before this sweep: 159.7 us)
    {1, 512, 1024, 10240, 2, 128, 2, 2},
    // g_u1_2B fwd: 313.3 us (automatic choice before this sweep: 332.0 us)
    {0, 2048, 2048, 2560, 2, 128, 1, 1},
    // g_u1_2B dgrad: 303.8 us (automatic choice before this sweep: 308.5 us)
    {1, 512, 2048, 10240, 2, 128, 1, 2},
    // g_u2 fwd: 161.0 us (automatic choice before this sweep: 163.2 us)
    {0, 1024, 2048, 2560, 2, 128, 1, 1},
    // g_u2 dgrad: 158.7 us (automatic choice before this sweep: 159.2 us)
    {1, 512, 2048, 5120, 2, 128, 1, 2},
    // g_u2_2B fwd: 316.7 us (automatic choice before this sweep: 330.5 us)
    {0, 1024, 4096, 2560, 2, 128, 1, 1},
    // g_u2_2B dgrad: 301.9 us (automatic choice before this sweep: 312.9 us)
    {1, 512, 4096, 5120, 2, 128, 1, 1},
    // g_o1 fwd: 39.6 us (automatic choice before this sweep: 40.6 us)
    {0, 24, 4096, 3840, 2, 64, 1, 4},
    // g_o1 dgrad: 24.5 us (automatic choice before this sweep: 24.7 us)
    {1, 256, 4096, 512, 2, 64, 1, 1},
    // g_o1_2B fwd: 67.0 us (automatic choice before this sweep: 67.6 us)
    {0, 24, 8192, 3840, 2, 64, 1, 4},
    // g_o1_2B dgrad: 40.6 us (automatic choice before this sweep: 40.5 us)
    {1, 256, 8192, 512, 2, 128, 1, 1},
    // g_h1 dgrad: 37.2 us (automatic choice before this sweep: 37.9 us)
    {1, 24, 4096, 3840, 2, 64, 1, 4},
    // g_h1_2B dgrad: 64.3 us (automatic choice before this sweep: 65.8 us)
    {1, 24, 8192, 3840, 2, 64, 1, 2},
    // d_d1 dgrad: 230.0 us (automatic choice before this sweep: 232.0 us)
    {1, 128, 12288, 2048, 2, 64, 1, 1},
    {1, 128, 12288, 1024, 2, 64, 1, 1},
    {1, 128, 12288, 512, 2, 64, 1, 1},
    // d_d1_2B fwd: 442.4 us (automatic choice before this sweep: 484.1 us)
    {0, 512, 24576, 1152, 2, 128, 1, 1},
    // d_d1_2B dgrad: 441.2 us (automatic choice before this sweep: 440.5 us)
    {1, 128, 24576, 2048, 2, 128, 1, 1},
    {1, 128, 24576, 1024, 2, 128, 1, 1},
    {1, 128, 24576, 512, 2, 128, 1, 1},
    // d_d2 dgrad: 233.1 us (automatic choice before this sweep: 234.8 us)
    {1, 256, 3072, 4096, 2, 128, 2, 1},
    {1, 256, 3072, 2048, 2, 128, 2, 1},
    {1, 256, 3072, 1024, 2, 128, 2, 1},
    // d_d2_2B fwd: 426.2 us (automatic choice before this sweep: 455.8 us)
    {0, 1024, 6144, 2304, 2, 128, 1, 1},
    // d_d2_2B dgrad: 454.4 us (automatic choice before this sweep: 452.6 us)
    {1, 256, 6144, 4096, 2, 128, 1, 1},
    {1, 256, 6144, 2048, 2, 128, 1, 1},
    {1, 256, 6144, 1024, 2, 128, 1, 1},
    // d_d3 dgrad: 859.3 us (automatic choice before this sweep: 875.3 us)
    {1, 512, 1536, 24576, 2, 64, 1, 4},
    {1, 512, 1536, 12288, 2, 64, 1, 4},
    // d_d3_2B fwd: 1634.6 us (automatic choice before this sweep: 1718.2 us)
    {0, 2048, 3072, 9216, 2, 128, 1, 1},
    // d_d3_2B dgrad: 1675.5 us (automatic choice before this sweep: 1667.0 us)
    {1, 512, 3072, 24576, 2, 128, 1, 4},
    {1, 512, 3072, 12288, 2, 128, 1, 4},
    // ---- batch 64 per GPU, 36x256 MCEP (re-measured on the round-4 kernels, profiles/r04_tune_sweep_*.txt)
    // g_d1 fwd: 89.6 us (automatic choice before this sweep: 98.3 us)
    {0, 512, 8192, 640, 2, 128, 1, 1},
    // g_d1 dgrad: 88.5 us (automatic choice before this sweep: 86.1 us)
    {1, 128, 8192, 1024, 2, 64, 1, 1},
    {1, 128, 8192, 1536, 2, 64, 1, 1},
    // g_d1_2B fwd: 173.9 us (automatic choice before this sweep: 188.8 us)
    {0, 512, 16384, 640, 2, 128, 1, 1},
    // g_d1_2B dgrad: 163.6 us (automatic choice before this sweep: 165.5 us)
    {1, 128, 16384, 1024, 2, 128, 1, 1},
    {1, 128, 16384, 1536, 2, 128, 1, 1},
    // g_d2 fwd: 162.7 us (automatic choice before this sweep: 177.9 us)
    {0, 1024, 4096, 1280, 2, 128, 1, 1},
    // g_d2 dgrad: 163.6 us (automatic choice before this sweep: 167.8 us)
    {1, 256, 4096, 2048, 2, 64, 1, 1},
    {1, 256, 4096, 3072, 2, 64, 1, 1},
    // g_d2_2B fwd: 319.7 us (automatic choice before this sweep: 348.2 us)
    {0, 1024, 8192, 1280, 2, 128, 1, 1},
    // g_d2_2B dgrad: 317.1 us (automatic choice before this sweep: 323.3 us)
    {1, 256, 8192, 2048, 2, 128, 1, 1},
    {1, 256, 8192, 3072, 2, 128, 1, 1},
    // g_r_h1 fwd: 372.7 us (automatic choice before this sweep: 405.7 us)
    {0, 2048, 4096, 1536, 2, 128, 1, 1},
    // g_r_h1 dgrad: 363.1 us (automatic choice before this sweep: 374.8 us)
    {1, 512, 4096, 6144, 2, 128, 1, 1},
    // g_r_h1_2B fwd: 750.2 us (automatic choice before this sweep: 745.6 us)
    {0, 2048, 8192, 1536, 2, 128, 1, 1},
    // g_r_h1_2B dgrad: 715.3 us (automatic choice before this sweep: 737.0 us)
    {1, 512, 8192, 6144, 2, 128, 1, 1},
    // g_r_h2 fwd: 189.1 us (automatic choice before this sweep: 197.9 us)
    {0, 512, 4096, 3072, 2, 128, 1, 1},
    // g_r_h2 dgrad: 189.4 us (automatic choice before this sweep: 192.6 us)
    {1, 1024, 4096, 1536, 2, 128, 1, 1},
    // g_r_h2_2B fwd: 371.2 us (automatic choice before this sweep: 399.4 us)
    {0, 512, 8192, 3072, 2, 128, 1, 1},
    // g_r_h2_2B dgrad: 372.7 us (automatic choice before this sweep: 376.6 us)
    {1, 1024, 8192, 1536, 2, 128, 1, 1},
    // g_u1 fwd: 612.6 us (automatic choice before this sweep: 657.1 us)
    {0, 2048, 4096, 2560, 2, 128, 1, 1},
    // g_u1 dgrad: 594.8 us (automatic choice before this sweep: 609.8 us)
    {1, 512, 4096, 10240, 2, 128, 1, 1},
    // g_u1_2B fwd: 1206.4 us (automatic choice before this sweep: 1221.1 us)
    {0, 2048, 8192, 2560, 2, 128, 1, 1},
    // g_u1_2B dgrad: 1181.9 us (automatic choice before this sweep: 1189.2 us)
    {1, 512, 8192, 10240, 2, 128, 1, 1},
    // g_u2 fwd: 608.9 us (automatic choice before this sweep: 667.2 us)
    {0, 1024, 8192, 2560, 2, 128, 1, 1},
    // g_u2 dgrad: 593.2 us (automatic choice before this sweep: 613.3 us)
    {1, 512, 8192, 5120, 2, 128, 1, 1},
    // g_u2_2B fwd: 1220.5 us (automatic choice before this sweep: 1222.2 us)
    {0, 1024, 16384, 2560, 2, 128, 1, 1},
    // g_u2_2B dgrad: 1188.6 us (automatic choice before this sweep: 1192.0 us)
    {1, 512, 16384, 5120, 2, 128, 1, 1},
    // g_o1 fwd: 124.3 us (automatic choice before this sweep: 131.3 us)
    {0, 36, 16384, 3840, 2, 64, 1, 2},
    // g_o1 dgrad: 124.8 us (automatic choice before this sweep: 124.5 us)
    {1, 256, 16384, 960, 2, 128, 1, 1},
    // g_o1_2B fwd: 240.3 us (automatic choice before this sweep: 243.6 us)
    {0, 36, 32768, 3840, 2, 64, 1, 1},
    // g_o1_2B dgrad: 241.8 us (automatic choice before this sweep: 242.4 us)
    {1, 256, 32768, 960, 2, 128, 1, 1},
    // g_h1 fwd: 126.9 us (automatic choice before this sweep: 130.0 us)
    {0, 256, 16384, 960, 2, 128, 1, 1},
    // g_h1 dgrad: 121.0 us (automatic choice before this sweep: 121.3 us)
    {1, 36, 16384, 3840, 2, 64, 1, 1},
    // g_h1_2B fwd: 245.3 us (automatic choice before this sweep: 247.8 us)
    {0, 256, 32768, 960, 2, 128, 1, 1},
    // g_h1_2B dgrad: 235.2 us (automatic choice before this sweep: 235.4 us)
    {1, 36, 32768, 3840, 2, 64, 1, 2},
    // d_d1 dgrad: 1335.1 us (automatic choice before this sweep: 1353.4 us)
    {1, 128, 73728, 2048, 2, 128, 1, 1},
    {1, 128, 73728, 1024, 2, 128, 1, 1},
    {1, 128, 73728, 512, 2, 128, 1, 1},
    // d_d1_2B fwd: 2671.1 us (automatic choice before this sweep: 2665.6 us)
    {0, 512, 147456, 1152, 2, 128, 1, 1},
    // d_d1_2B dgrad: 2905.1 us (automatic choice before this sweep: 2771.9 us)
    {1, 128, 147456, 2048, 2, 128, 1, 1},
    {1, 128, 147456, 1024, 2, 128, 1, 1},
    {1, 128, 147456, 512, 2, 128, 1, 1},
    // d_d2 dgrad: 1251.0 us (automatic choice before this sweep: 1283.2 us)
    {1, 256, 18432, 4096, 2, 128, 1, 1},
    {1, 256, 18432, 2048, 2, 128, 1, 1},
    {1, 256, 18432, 1024, 2, 128, 1, 1},
    // d_d2_2B fwd: 2588.8 us (automatic choice before this sweep: 2583.9 us)
    {0, 1024, 36864, 2304, 2, 128, 1, 1},
    // d_d2_2B dgrad: 2624.7 us (automatic choice before this sweep: 2603.7 us)
    {1, 256, 36864, 4096, 2, 128, 1, 1},
    {1, 256, 36864, 2048, 2, 128, 1, 1},
    {1, 256, 36864, 1024, 2, 128, 1, 1},
    // d_d3 dgrad: 4916.2 us (automatic choice before this sweep: 4914.4 us)
    {1, 512, 9216, 24576, 2, 128, 1, 4},
    {1, 512, 9216, 12288, 2, 128, 1, 4},
    // d_d3_2B fwd: 9859.5 us (automatic choice before this sweep: 9849.6 us)
    {0, 2048, 18432, 9216, 2, 128, 1, 1},
    // d_d3_2B dgrad: 9712.4 us (automatic choice before this sweep: 9827.8 us)
    {1, 512, 18432, 24576, 2, 128, 1, 2},
    {1, 512, 18432, 12288, 2, 128, 1, 2},
    // ---- batch 1 per GPU, 24x128 MCEP (BASELINE configs[0], the reference's own mini_batch_size; round-4 sweep, profiles/r04_tune_sweep_b1.txt)
    // g_d1 fwd: 12.9 us (automatic choice before this sweep: 16.7 us)
    {0, 512, 64, 640, 2, 64, 2, 8},
    // g_d1 dgrad: 10.7 us (automatic choice before this sweep: 20.2 us)
    {1, 128, 64, 1024, 2, 64, 2, 16},
    {1, 128, 64, 1536, 2, 64, 2, 16},
    // g_d1_2B fwd: 13.5 us (automatic choice before this sweep: 18.0 us)
    {0, 512, 128, 640, 2, 64, 2, 8},
    // g_d1_2B dgrad: 12.5 us (automatic choice before this sweep: 14.4 us)
    {1, 128, 128, 1024, 2, 64, 1, 8},
    {1, 128, 128, 1536, 2, 64, 1, 8},
    // g_d2 fwd: 15.1 us (automatic choice before this sweep: 19.3 us)
    {0, 1024, 32, 1280, 2, 64, 2, 16},
    // g_d2 dgrad: 13.0 us (automatic choice before this sweep: 22.0 us)
    {1, 256, 32, 2048, 2, 64, 1, 16},
    {1, 256, 32, 3072, 2, 64, 1, 16},
    // g_d2_2B fwd: 16.1 us (automatic choice before this sweep: 29.6 us)
    {0, 1024, 64, 1280, 2, 64, 2, 8},
    // g_d2_2B dgrad: 15.6 us (automatic choice before this sweep: 17.6 us)
    {1, 256, 64, 2048, 2, 64, 2, 8},
    {1, 256, 64, 3072, 2, 64, 2, 8},
    // g_r_h1 fwd: 17.9 us (automatic choice before this sweep: 18.7 us)
    {0, 2048, 32, 1536, 2, 64, 2, 4},
    // g_r_h1 dgrad: 16.1 us (automatic choice before this sweep: 15.9 us)
    {1, 512, 32, 6144, 2, 64, 2, 16},
    // g_r_h1_2B fwd: 18.9 us (automatic choice before this sweep: 24.0 us)
    {0, 2048, 64, 1536, 2, 64, 2, 4},
    // g_r_h1_2B dgrad: 19.0 us (automatic choice before this sweep: 21.2 us)
    {1, 512, 64, 6144, 2, 64, 2, 16},
    // g_r_h2 fwd: 15.5 us (automatic choice before this sweep: 17.3 us)
    {0, 512, 32, 3072, 2, 64, 2, 16},
    // g_r_h2 dgrad: 12.0 us (automatic choice before this sweep: 12.4 us)
    {1, 1024, 32, 1536, 2, 64, 1, 8},
    // g_r_h2_2B fwd: 16.5 us (automatic choice before this sweep: 22.5 us)
    {0, 512, 64, 3072, 2, 64, 4, 16},
    // g_r_h2_2B dgrad: 13.9 us (automatic choice before this sweep: 14.7 us)
    {1, 1024, 64, 1536, 2, 64, 1, 8},
    // g_u1 fwd: 23.0 us (automatic choice before this sweep: 25.3 us)
    {0, 2048, 32, 2560, 2, 64, 2, 4},
    // g_u1 dgrad: 21.3 us (automatic choice before this sweep: 20.7 us)
    {1, 512, 32, 10240, 2, 64, 2, 16},
    // g_u1_2B fwd: 24.3 us (automatic choice before this sweep: 32.7 us)
    {0, 2048, 64, 2560, 2, 64, 2, 4},
    // g_u1_2B dgrad: 23.8 us (automatic choice before this sweep: 29.8 us)
    {1, 512, 64, 10240, 2, 64, 2, 16},
    // g_u2 fwd: 18.7 us (automatic choice before this sweep: 21.2 us)
    {0, 1024, 64, 2560, 2, 64, 2, 8},
    // g_u2 dgrad: 16.9 us (automatic choice before this sweep: 16.9 us)
    {1, 512, 64, 5120, 2, 64, 1, 16},
    // g_u2_2B fwd: 23.8 us (automatic choice before this sweep: 31.9 us)
    {0, 1024, 128, 2560, 2, 64, 2, 4},
    // g_u2_2B dgrad: 21.5 us (automatic choice before this sweep: 21.5 us)
    {1, 512, 128, 5120, 2, 64, 2, 8},
    // g_o1 fwd: 12.5 us (automatic choice before this sweep: 18.4 us)
    {0, 24, 128, 3840, 2, 64, 2, 16},
    // g_o1 dgrad: 10.4 us (automatic choice before this sweep: 16.2 us)
    {1, 256, 128, 512, 2, 64, 1, 16},
    // g_o1_2B fwd: 13.1 us (automatic choice before this sweep: 39.9 us)
    {0, 24, 256, 3840, 2, 64, 2, 16},
    // g_o1_2B dgrad: 10.8 us (automatic choice before this sweep: 11.3 us)
    {1, 256, 256, 512, 2, 64, 1, 4},
    // g_h1 dgrad: 11.7 us (automatic choice before this sweep: 15.9 us)
    {1, 24, 128, 3840, 2, 64, 2, 16},
    // g_h1_2B dgrad: 11.9 us (automatic choice before this sweep: 23.6 us)
    {1, 24, 256, 3840, 2, 64, 1, 16},
    // d_d1 dgrad: 21.2 us (automatic choice before this sweep: 58.5 us)
    {1, 128, 384, 2048, 2, 64, 2, 4},
    {1, 128, 384, 1024, 2, 64, 2, 4},
    {1, 128, 384, 512, 2, 64, 2, 4},
    // d_d1_2B fwd: 30.6 us (automatic choice before this sweep: 32.8 us)
    {0, 512, 768, 1152, 2, 64, 4, 1},
    // d_d1_2B dgrad: 32.2 us (automatic choice before this sweep: 64.9 us)
    {1, 128, 768, 2048, 2, 64, 2, 2},
    {1, 128, 768, 1024, 2, 64, 2, 2},
    {1, 128, 768, 512, 2, 64, 2, 2},
    // d_d2 dgrad: 26.1 us (automatic choice before this sweep: 55.4 us)
    {1, 256, 96, 4096, 2, 64, 4, 4},
    {1, 256, 96, 2048, 2, 64, 4, 4},
    {1, 256, 96, 1024, 2, 64, 4, 4},
    // d_d2_2B fwd: 32.0 us (automatic choice before this sweep: 35.1 us)
    {0, 1024, 192, 2304, 2, 64, 1, 4},
    // d_d2_2B dgrad: 33.3 us (automatic choice before this sweep: 66.8 us)
    {1, 256, 192, 4096, 2, 64, 2, 4},
    {1, 256, 192, 2048, 2, 64, 2, 4},
    {1, 256, 192, 1024, 2, 64, 2, 4},
    // d_d3 dgrad: 65.4 us (automatic choice before this sweep: 66.7 us)
    {1, 512, 48, 24576, 2, 64, 4, 16},
    {1, 512, 48, 12288, 2, 64, 4, 16},
    // d_d3_2B fwd: 86.2 us (automatic choice before this sweep: 96.5 us)
    {0, 2048, 96, 9216, 2, 64, 4, 2},
    // d_d3_2B dgrad: 100.3 us (automatic choice before this sweep: 100.5 us)
    {1, 512, 96, 24576, 2, 64, 2, 16},
    {1, 512, 96, 12288, 2, 64, 2, 16},
    // ---- batch 2 per GPU, 24x128 MCEP (round-4 sweep, profiles/r04_tune_sweep_b2.txt)
    // g_d1 fwd: 13.7 us (automatic choice before this sweep: 13.9 us)
    // g_d1 dgrad: 12.3 us (automatic choice before this sweep: 13.1 us)
    // g_d1_2B fwd: 16.2 us (automatic choice before this sweep: 19.1 us)
    {0, 512, 256, 640, 2, 64, 2, 8},
    // g_d1_2B dgrad: 14.8 us (automatic choice before this sweep: 15.6 us)
    {1, 128, 256, 1024, 2, 64, 2, 4},
    {1, 128, 256, 1536, 2, 64, 2, 4},
    // g_d2 fwd: 16.0 us (automatic choice before this sweep: 16.9 us)
    // g_d2 dgrad: 15.6 us (automatic choice before this sweep: 15.0 us)
    // g_d2_2B fwd: 19.9 us (automatic choice before this sweep: 30.6 us)
    {0, 1024, 128, 1280, 2, 64, 2, 4},
    // g_d2_2B dgrad: 18.8 us (automatic choice before this sweep: 34.1 us)
    {1, 256, 128, 2048, 2, 64, 4, 8},
    {1, 256, 128, 3072, 2, 64, 4, 8},
    // g_r_h1 fwd: 19.1 us (automatic choice before this sweep: 20.0 us)
    // g_r_h1 dgrad: 18.3 us (automatic choice before this sweep: 17.7 us)
    // g_r_h1_2B fwd: 24.1 us (automatic choice before this sweep: 26.8 us)
    {0, 2048, 128, 1536, 2, 64, 4, 2},
    // g_r_h1_2B dgrad: 24.9 us (automatic choice before this sweep: 23.6 us)
    {1, 512, 128, 6144, 2, 64, 4, 8},
    // g_r_h2 fwd: 16.1 us (automatic choice before this sweep: 16.7 us)
    // g_r_h2 dgrad: 14.5 us (automatic choice before this sweep: 14.7 us)
    // g_r_h2_2B fwd: 19.5 us (automatic choice before this sweep: 23.9 us)
    {0, 512, 128, 3072, 2, 64, 2, 8},
    // g_r_h2_2B dgrad: 16.8 us (automatic choice before this sweep: 16.9 us)
    {1, 1024, 128, 1536, 2, 64, 2, 4},
    // g_u1 fwd: 24.2 us (automatic choice before this sweep: 25.4 us)
    // g_u1 dgrad: 24.9 us (automatic choice before this sweep: 23.5 us)
    // g_u1_2B fwd: 31.9 us (automatic choice before this sweep: 36.0 us)
    {0, 2048, 128, 2560, 2, 64, 4, 2},
    // g_u1_2B dgrad: 34.0 us (automatic choice before this sweep: 32.1 us)
    {1, 512, 128, 10240, 2, 64, 4, 8},
    // g_u2 fwd: 23.6 us (automatic choice before this sweep: 26.9 us)
    // g_u2 dgrad: 22.0 us (automatic choice before this sweep: 20.9 us)
    // g_u2_2B fwd: 32.1 us (automatic choice before this sweep: 35.7 us)
    {0, 1024, 256, 2560, 2, 64, 4, 2},
    // g_u2_2B dgrad: 32.0 us (automatic choice before this sweep: 30.7 us)
    {1, 512, 256, 5120, 2, 64, 4, 4},
    // g_o1 fwd: 12.8 us (automatic choice before this sweep: 12.4 us)
    // g_o1 dgrad: 10.9 us (automatic choice before this sweep: 10.9 us)
    // g_o1_2B fwd: 13.7 us (automatic choice before this sweep: 40.8 us)
    {0, 24, 512, 3840, 2, 64, 2, 16},
    // g_o1_2B dgrad: 12.4 us (automatic choice before this sweep: 13.0 us)
    {1, 256, 512, 512, 2, 64, 1, 4},
    // g_h1 dgrad: 12.1 us (automatic choice before this sweep: 12.1 us)
    // g_h1_2B dgrad: 13.1 us (automatic choice before this sweep: 24.9 us)
    {1, 24, 512, 3840, 2, 64, 1, 16},
    // d_d1 dgrad: 32.4 us (automatic choice before this sweep: 34.9 us)
    // d_d1_2B fwd: 45.8 us (automatic choice before this sweep: 62.6 us)
    {0, 512, 1536, 1152, 2, 64, 1, 2},
    // d_d1_2B dgrad: 45.9 us (automatic choice before this sweep: 90.6 us)
    {1, 128, 1536, 2048, 2, 64, 4, 1},
    {1, 128, 1536, 1024, 2, 64, 4, 1},
    {1, 128, 1536, 512, 2, 64, 4, 1},
    // d_d2 dgrad: 33.3 us (automatic choice before this sweep: 35.6 us)
    // d_d2_2B fwd: 44.4 us (automatic choice before this sweep: 57.0 us)
    {0, 1024, 384, 2304, 2, 64, 1, 4},
    // d_d2_2B dgrad: 47.3 us (automatic choice before this sweep: 94.0 us)
    {1, 256, 384, 4096, 2, 64, 1, 4},
    {1, 256, 384, 2048, 2, 64, 1, 4},
    {1, 256, 384, 1024, 2, 64, 1, 4},
    // d_d3 dgrad: 100.6 us (automatic choice before this sweep: 108.1 us)
    // d_d3_2B fwd: 127.1 us (automatic choice before this sweep: 181.4 us)
    {0, 2048, 192, 9216, 2, 64, 1, 4},
    // d_d3_2B dgrad: 153.3 us (automatic choice before this sweep: 145.4 us)
    {1, 512, 192, 24576, 2, 64, 1, 16},
    {1, 512, 192, 12288, 2, 64, 1, 16},
    // ---- batch 4 per GPU, 24x128 MCEP (round-4 sweep, profiles/r04_tune_sweep_b4.txt)
    // g_d1 fwd: 16.4 us (automatic choice before this sweep: 18.5 us)
    // g_d1 dgrad: 14.9 us (automatic choice before this sweep: 14.5 us)
    // g_d1_2B fwd: 19.0 us (automatic choice before this sweep: 19.9 us)
    // g_d1_2B dgrad: 17.3 us (automatic choice before this sweep: 16.9 us)
    // g_d2 fwd: 19.5 us (automatic choice before this sweep: 30.2 us)
    // g_d2 dgrad: 18.5 us (automatic choice before this sweep: 33.7 us)
    // g_d2_2B fwd: 24.1 us (automatic choice before this sweep: 27.2 us)
    // g_d2_2B dgrad: 23.2 us (automatic choice before this sweep: 22.9 us)
    // g_r_h1 fwd: 24.0 us (automatic choice before this sweep: 26.5 us)
    // g_r_h1 dgrad: 25.1 us (automatic choice before this sweep: 23.2 us)
    // g_r_h1_2B fwd: 33.8 us (automatic choice before this sweep: 38.3 us)
    // g_r_h1_2B dgrad: 37.1 us (automatic choice before this sweep: 34.6 us)
    // g_r_h2 fwd: 20.1 us (automatic choice before this sweep: 23.7 us)
    // g_r_h2 dgrad: 16.1 us (automatic choice before this sweep: 16.6 us)
    // g_r_h2_2B fwd: 25.3 us (automatic choice before this sweep: 26.7 us)
    // g_r_h2_2B dgrad: 21.8 us (automatic choice before this sweep: 22.5 us)
    // g_u1 fwd: 32.1 us (automatic choice before this sweep: 35.5 us)
    // g_u1 dgrad: 34.4 us (automatic choice before this sweep: 32.3 us)
    // g_u1_2B fwd: 51.0 us (automatic choice before this sweep: 57.3 us)
    // g_u1_2B dgrad: 54.4 us (automatic choice before this sweep: 51.0 us)
    // g_u2 fwd: 32.9 us (automatic choice before this sweep: 35.7 us)
    // g_u2 dgrad: 32.3 us (automatic choice before this sweep: 31.3 us)
    // g_u2_2B fwd: 49.8 us (automatic choice before this sweep: 57.3 us)
    // g_u2_2B dgrad: 50.4 us (automatic choice before this sweep: 49.0 us)
    // g_o1 fwd: 14.1 us (automatic choice before this sweep: 39.4 us)
    // g_o1 dgrad: 12.5 us (automatic choice before this sweep: 12.7 us)
    // g_o1_2B fwd: 18.6 us (automatic choice before this sweep: 19.3 us)
    // g_o1_2B dgrad: 15.0 us (automatic choice before this sweep: 14.7 us)
    // g_h1 dgrad: 12.8 us (automatic choice before this sweep: 23.6 us)
    // g_h1_2B dgrad: 17.5 us (automatic choice before this sweep: 18.2 us)
    // d_d1 dgrad: 45.7 us (automatic choice before this sweep: 96.8 us)
    // d_d1_2B fwd: 70.3 us (automatic choice before this sweep: 91.1 us)
    {0, 512, 3072, 1152, 2, 64, 1, 1},
    // d_d1_2B dgrad: 76.5 us (automatic choice before this sweep: 71.8 us)
    // d_d2 dgrad: 47.5 us (automatic choice before this sweep: 99.2 us)
    // d_d2_2B fwd: 73.1 us (automatic choice before this sweep: 81.0 us)
    {0, 1024, 768, 2304, 2, 64, 1, 2},
    // d_d2_2B dgrad: 76.6 us (automatic choice before this sweep: 74.1 us)
    // d_d3 dgrad: 154.9 us (automatic choice before this sweep: 169.5 us)
    // d_d3_2B fwd: 239.9 us (automatic choice before this sweep: 309.1 us)
    {0, 2048, 384, 9216, 2, 64, 1, 4},
    // d_d3_2B dgrad: 256.3 us (automatic choice before this sweep: 244.4 us)
    // ---- batch 16 per GPU, 24x128 MCEP (round-4 sweep, profiles/r04_tune_sweep_b16.txt)
    // g_d1 fwd: 21.2 us (automatic choice before this sweep: 23.3 us)
    // g_d1 dgrad: 22.2 us (automatic choice before this sweep: 22.5 us)
    // g_d1_2B fwd: 32.5 us (automatic choice before this sweep: 33.7 us)
    // g_d1_2B dgrad: 32.6 us (automatic choice before this sweep: 31.9 us)
    // g_d2 fwd: 32.3 us (automatic choice before this sweep: 35.8 us)
    // g_d2 dgrad: 35.3 us (automatic choice before this sweep: 35.3 us)
    // g_d2_2B fwd: 53.9 us (automatic choice before this sweep: 60.1 us)
    // g_d2_2B dgrad: 56.6 us (automatic choice before this sweep: 57.4 us)
    // g_r_h1 fwd: 59.3 us (automatic choice before this sweep: 66.9 us)
    // g_r_h1 dgrad: 59.1 us (automatic choice before this sweep: 58.8 us)
    // g_r_h1_2B fwd: 104.5 us (automatic choice before this sweep: 122.6 us)
    // g_r_h1_2B dgrad: 101.0 us (automatic choice before this sweep: 102.7 us)
    // g_r_h2 fwd: 37.3 us (automatic choice before this sweep: 41.5 us)
    // g_r_h2 dgrad: 32.2 us (automatic choice before this sweep: 33.4 us)
    // g_r_h2_2B fwd: 56.7 us (automatic choice before this sweep: 67.0 us)
    // g_r_h2_2B dgrad: 54.7 us (automatic choice before this sweep: 55.4 us)
    // g_u1 fwd: 88.5 us (automatic choice before this sweep: 103.9 us)
    // g_u1 dgrad: 89.2 us (automatic choice before this sweep: 90.1 us)
    // g_u1_2B fwd: 164.3 us (automatic choice before this sweep: 198.2 us)
    // g_u1_2B dgrad: 162.1 us (automatic choice before this sweep: 161.7 us)
    // g_u2 fwd: 89.3 us (automatic choice before this sweep: 103.8 us)
    // g_u2 dgrad: 85.7 us (automatic choice before this sweep: 86.8 us)
    // g_u2_2B fwd: 163.9 us (automatic choice before this sweep: 201.7 us)
    // g_u2_2B dgrad: 158.7 us (automatic choice before this sweep: 162.2 us)
    // g_o1 fwd: 27.7 us (automatic choice before this sweep: 28.9 us)
    // g_o1 dgrad: 16.3 us (automatic choice before this sweep: 17.6 us)
    // g_o1_2B fwd: 44.9 us (automatic choice before this sweep: 48.0 us)
    // g_o1_2B dgrad: 24.9 us (automatic choice before this sweep: 27.0 us)
    // g_h1 dgrad: 25.8 us (automatic choice before this sweep: 26.3 us)
    // g_h1_2B dgrad: 39.6 us (automatic choice before this sweep: 40.8 us)
    // d_d1 dgrad: 129.4 us (automatic choice before this sweep: 139.6 us)
    // d_d1_2B fwd: 243.1 us (automatic choice before this sweep: 300.6 us)
    {0, 512, 12288, 1152, 2, 64, 1, 1},
    // d_d1_2B dgrad: 240.7 us (automatic choice before this sweep: 236.8 us)
    // d_d2 dgrad: 134.6 us (automatic choice before this sweep: 150.0 us)
    // d_d2_2B fwd: 232.5 us (automatic choice before this sweep: 294.0 us)
    {0, 1024, 3072, 2304, 2, 64, 1, 1},
    // d_d2_2B dgrad: 242.5 us (automatic choice before this sweep: 242.3 us)
    // d_d3 dgrad: 448.1 us (automatic choice before this sweep: 530.0 us)
    // d_d3_2B fwd: 848.8 us (automatic choice before this sweep: 1068.9 us)
    {0, 2048, 1536, 9216, 2, 128, 1, 4},
    // d_d3_2B dgrad: 862.4 us (automatic choice before this sweep: 880.2 us)
    // ---- batch 64 per GPU, 24x128 MCEP (round-4 sweep, profiles/r04_tune_sweep_b64.txt)
    // g_d1 fwd: 54.5 us (automatic choice before this sweep: 56.7 us)
    // g_d1 dgrad: 55.2 us (automatic choice before this sweep: 55.5 us)
    // g_d1_2B fwd: 95.6 us (automatic choice before this sweep: 110.7 us)
    // g_d1_2B dgrad: 91.9 us (automatic choice before this sweep: 88.3 us)
    // g_d2 fwd: 94.3 us (automatic choice before this sweep: 108.0 us)
    // g_d2 dgrad: 100.1 us (automatic choice before this sweep: 100.5 us)
    // g_d2_2B fwd: 180.2 us (automatic choice before this sweep: 216.8 us)
    // g_d2_2B dgrad: 168.5 us (automatic choice before this sweep: 169.1 us)
    // g_r_h1 fwd: 203.2 us (automatic choice before this sweep: 249.0 us)
    // g_r_h1 dgrad: 187.3 us (automatic choice before this sweep: 192.8 us)
    // g_r_h1_2B fwd: 406.5 us (automatic choice before this sweep: 500.0 us)
    // g_r_h1_2B dgrad: 364.5 us (automatic choice before this sweep: 383.9 us)
    // g_r_h2 fwd: 103.9 us (automatic choice before this sweep: 122.9 us)
    // g_r_h2 dgrad: 97.5 us (automatic choice before this sweep: 97.6 us)
    // g_r_h2_2B fwd: 194.7 us (automatic choice before this sweep: 237.7 us)
    // g_r_h2_2B dgrad: 195.6 us (automatic choice before this sweep: 196.6 us)
    // g_u1 fwd: 327.3 us (automatic choice before this sweep: 414.4 us)
    // g_u1 dgrad: 306.6 us (automatic choice before this sweep: 320.3 us)
    // g_u1_2B fwd: 644.5 us (automatic choice before this sweep: 803.1 us)
    // g_u1_2B dgrad: 595.8 us (automatic choice before this sweep: 633.4 us)
    // g_u2 fwd: 326.4 us (automatic choice before this sweep: 412.0 us)
    // g_u2 dgrad: 310.5 us (automatic choice before this sweep: 327.3 us)
    // g_u2_2B fwd: 645.1 us (automatic choice before this sweep: 796.6 us)
    // g_u2_2B dgrad: 598.6 us (automatic choice before this sweep: 633.6 us)
    // g_o1 fwd: 73.1 us (automatic choice before this sweep: 80.5 us)
    // g_o1 dgrad: 41.4 us (automatic choice before this sweep: 41.6 us)
    // g_o1_2B fwd: 135.3 us (automatic choice before this sweep: 156.9 us)
    {0, 24, 16384, 3840, 2, 64, 4, 1},
    // g_o1_2B dgrad: 75.9 us (automatic choice before this sweep: 75.7 us)
    {1, 256, 16384, 512, 2, 128, 1, 1},
    // g_h1 dgrad: 67.5 us (automatic choice before this sweep: 72.2 us)
    // g_h1_2B dgrad: 130.5 us (automatic choice before this sweep: 143.8 us)
    {1, 24, 16384, 3840, 2, 64, 2, 1},
    // d_d1 dgrad: 486.4 us (automatic choice before this sweep: 546.2 us)
    // d_d1_2B fwd: 910.7 us (automatic choice before this sweep: 1125.8 us)
    {0, 512, 49152, 1152, 2, 128, 1, 1},
    // d_d1_2B dgrad: 891.4 us (automatic choice before this sweep: 934.1 us)
    {1, 128, 49152, 2048, 2, 128, 1, 1},
    {1, 128, 49152, 1024, 2, 128, 1, 1},
    {1, 128, 49152, 512, 2, 128, 1, 1},
    // d_d2 dgrad: 461.4 us (automatic choice before this sweep: 549.3 us)
    // d_d2_2B fwd: 874.1 us (automatic choice before this sweep: 1056.9 us)
    {0, 1024, 12288, 2304, 2, 128, 1, 1},
    // d_d2_2B dgrad: 849.5 us (automatic choice before this sweep: 880.9 us)
    {1, 256, 12288, 4096, 2, 128, 1, 1},
    {1, 256, 12288, 2048, 2, 128, 1, 1},
    {1, 256, 12288, 1024, 2, 128, 1, 1},
    // d_d3 dgrad: 1673.6 us (automatic choice before this sweep: 1839.8 us)
    // d_d3_2B fwd: 3253.0 us (automatic choice before this sweep: 3503.9 us)
    {0, 2048, 6144, 9216, 2, 128, 1, 1},
    // d_d3_2B dgrad: 3236.4 us (automatic choice before this sweep: 3280.7 us)
    {1, 512, 6144, 24576, 2, 128, 1, 2},
    {1, 512, 6144, 12288, 2, 128, 1, 2},
};
