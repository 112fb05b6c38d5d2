// Measured filter-gradient configurations: tools/tune_igemm.py --sweep on MI355X (round 3; profiles/r03_tune_sweep_*.txt), as the
// train step launches them (plain stores; the generator's layers reduce first hop + cycle hop in one launch).  {K tiles,
// Cout, pixel chunks (all applications), groups, plain stores?, tile columns, K groups, split-K}.  Shapes not listed go by the
// cost model in wgrad_config.
static const TunedWgrad kTunedWgrad[] = {
    // ---- batch 8 per GPU, 24x128 MCEP (re-measured on the round-4 kernels, profiles/r04_tune_sweep_*.txt)
    // g_d1_3B: 35.5 us (automatic choice before this sweep: 36.0 us)
    {10, 512, 48, 2, 1, 64, 1, 4},
    // g_d2_3B: 50.9 us (automatic choice before this sweep: 50.7 us)
    {20, 1024, 24, 2, 1, 64, 1, 1},
    // g_r_h1_3B: 91.2 us (automatic choice before this sweep: 91.2 us)
    {24, 2048, 24, 2, 1, 64, 1, 1},
    // g_r_h2_3B: 50.6 us (automatic choice before this sweep: 49.8 us)
    {48, 512, 24, 2, 1, 64, 1, 1},
    // g_u1_3B: 145.4 us (automatic choice before this sweep: 145.0 us)
    {40, 2048, 24, 2, 1, 64, 1, 1},
    // g_u2_3B: 146.4 us (automatic choice before this sweep: 146.9 us)
    {40, 1024, 48, 2, 1, 64, 2, 1},
    // g_o1_3B: 36.6 us (automatic choice before this sweep: 36.0 us)
    {60, 24, 96, 2, 1, 64, 4, 2},
    // g_h1_3B: 23.4 us (automatic choice before this sweep: 23.0 us)
    {8, 256, 96, 2, 1, 64, 4, 4},
    // d_d1_2B: 147.0 us (automatic choice before this sweep: 144.5 us)
    {18, 512, 192, 2, 1, 64, 1, 8},
    // d_d2_2B: 149.3 us (automatic choice before this sweep: 149.3 us)
    {36, 1024, 48, 2, 1, 64, 2, 1},
    // d_d3_2B: 424.5 us (automatic choice before this sweep: 422.3 us)
    {144, 2048, 24, 2, 1, 128, 1, 1},
    // ---- batch 32 per GPU, 24x128 MCEP (re-measured on the round-4 kernels, profiles/r04_tune_sweep_*.txt)
    // g_d1_3B: 88.8 us (automatic choice before this sweep: 88.5 us)
    {10, 512, 192, 2, 1, 64, 1, 8},
    // g_d2_3B: 158.5 us (automatic choice before this sweep: 156.2 us)
    {20, 1024, 96, 2, 1, 64, 2, 2},
    // g_r_h1_3B: 324.3 us (automatic choice before this sweep: 324.7 us)
    {24, 2048, 96, 2, 1, 128, 1, 1},
    // g_r_h2_3B: 173.7 us (automatic choice before this sweep: 173.0 us)
    {48, 512, 96, 2, 1, 64, 1, 1},
    // g_u1_3B: 527.6 us (automatic choice before this sweep: 548.3 us)
    {40, 2048, 96, 2, 1, 128, 1, 1},
    // g_u2_3B: 544.7 us (automatic choice before this sweep: 544.0 us)
    {40, 1024, 192, 2, 1, 64, 2, 1},
    // g_o1_3B: 116.8 us (automatic choice before this sweep: 116.9 us)
    {60, 24, 384, 2, 1, 64, 2, 4},
    // g_h1_3B: 63.2 us (automatic choice before this sweep: 62.9 us)
    {8, 256, 384, 2, 1, 64, 4, 4},
    // d_d1_2B: 519.8 us (automatic choice before this sweep: 516.7 us)
    {18, 512, 768, 2, 1, 64, 1, 8},
    // d_d2_2B: 520.9 us (automatic choice before this sweep: 517.1 us)
    {36, 1024, 192, 2, 1, 64, 1, 2},
    // d_d3_2B: 1663.8 us (automatic choice before this sweep: 1679.9 us)
    {144, 2048, 96, 2, 1, 128, 1, 1},
    // ---- batch 64 per GPU, 36x256 MCEP (re-measured on the round-4 kernels, profiles/r04_tune_sweep_*.txt)
    // g_d1_3B: 295.1 us (automatic choice before this sweep: 296.2 us)
    {10, 512, 768, 2, 1, 64, 1, 16},
    // g_d2_3B: 575.1 us (automatic choice before this sweep: 573.5 us)
    {20, 1024, 384, 2, 1, 128, 2, 4},
    // g_r_h1_3B: 1244.6 us (automatic choice before this sweep: 1240.0 us)
    {24, 2048, 384, 2, 1, 128, 2, 1},
    // g_r_h2_3B: 651.1 us (automatic choice before this sweep: 646.8 us)
    {48, 512, 384, 2, 1, 128, 1, 4},
    // g_u1_3B: 2071.6 us (automatic choice before this sweep: 2062.8 us)
    {40, 2048, 384, 2, 1, 128, 2, 1},
    // g_u2_3B: 2060.7 us (automatic choice before this sweep: 2052.8 us)
    {40, 1024, 768, 2, 1, 128, 1, 4},
    // g_o1_3B: 432.2 us (automatic choice before this sweep: 430.9 us)
    {60, 36, 1536, 2, 1, 64, 1, 8},
    // g_h1_3B: 412.8 us (automatic choice before this sweep: 423.0 us)
    {15, 256, 1536, 2, 1, 128, 1, 8},
    // d_d1_2B: 2943.3 us (automatic choice before this sweep: 2941.7 us)
    {18, 512, 4608, 2, 1, 128, 2, 16},
    // d_d2_2B: 2847.3 us (automatic choice before this sweep: 2837.4 us)
    {36, 1024, 1152, 2, 1, 128, 1, 8},
    // d_d3_2B: 9615.7 us (automatic choice before this sweep: 9643.3 us)
    {144, 2048, 576, 2, 1, 128, 2, 1},
    // ---- batch 1 per GPU, 24x128 MCEP (BASELINE configs[0], the reference's own mini_batch_size; round-4 sweep, profiles/r04_tune_sweep_b1.txt)
    // g_d1_3B: 12.1 us (automatic choice before this sweep: 12.4 us)
    {10, 512, 6, 2, 1, 64, 1, 1},
    // g_d2_3B: 16.9 us (automatic choice before this sweep: 17.5 us)
    {20, 1024, 3, 2, 1, 64, 1, 1},
    // g_r_h1_3B: 29.1 us (automatic choice before this sweep: 36.3 us)
    {24, 2048, 3, 2, 1, 64, 1, 1},
    // g_r_h2_3B: 17.2 us (automatic choice before this sweep: 18.3 us)
    {48, 512, 3, 2, 1, 64, 1, 1},
    // g_u1_3B: 39.5 us (automatic choice before this sweep: 56.1 us)
    {40, 2048, 3, 2, 1, 64, 1, 1},
    // g_u2_3B: 30.0 us (automatic choice before this sweep: 36.0 us)
    {40, 1024, 6, 2, 1, 64, 1, 1},
    // g_o1_3B: 15.0 us (automatic choice before this sweep: 15.4 us)
    {60, 24, 12, 2, 1, 64, 2, 1},
    // g_h1_3B: 11.8 us (automatic choice before this sweep: 15.8 us)
    {8, 256, 12, 2, 1, 64, 2, 8},
    // d_d1_2B: 33.9 us (automatic choice before this sweep: 33.9 us)
    {18, 512, 24, 2, 1, 128, 2, 1},
    // d_d2_2B: 31.0 us (automatic choice before this sweep: 33.9 us)
    {36, 1024, 6, 2, 1, 64, 1, 1},
    // d_d3_2B: 115.5 us (automatic choice before this sweep: 120.1 us)
    {144, 2048, 3, 2, 1, 64, 1, 1},
    // ---- batch 2 per GPU, 24x128 MCEP (round-4 sweep, profiles/r04_tune_sweep_b2.txt)
    // g_d1_3B: 15.6 us (automatic choice before this sweep: 18.2 us)
    {10, 512, 12, 2, 1, 64, 2, 1},
    // g_d2_3B: 20.9 us (automatic choice before this sweep: 21.9 us)
    {20, 1024, 6, 2, 1, 64, 1, 1},
    // g_r_h1_3B: 34.9 us (automatic choice before this sweep: 37.8 us)
    {24, 2048, 6, 2, 1, 64, 1, 1},
    // g_r_h2_3B: 21.4 us (automatic choice before this sweep: 23.1 us)
    {48, 512, 6, 2, 1, 64, 1, 1},
    // g_u1_3B: 50.5 us (automatic choice before this sweep: 55.6 us)
    {40, 2048, 6, 2, 1, 64, 1, 1},
    // g_u2_3B: 47.6 us (automatic choice before this sweep: 52.4 us)
    {40, 1024, 12, 2, 1, 64, 1, 1},
    // g_o1_3B: 18.5 us (automatic choice before this sweep: 23.5 us)
    {60, 24, 24, 2, 1, 64, 2, 2},
    // g_h1_3B: 14.3 us (automatic choice before this sweep: 17.4 us)
    {8, 256, 24, 2, 1, 64, 4, 8},
    // d_d1_2B: 56.3 us (automatic choice before this sweep: 59.6 us)
    {18, 512, 48, 2, 1, 64, 1, 4},
    // d_d2_2B: 48.0 us (automatic choice before this sweep: 47.3 us)
    {36, 1024, 12, 2, 1, 64, 1, 1},
    // d_d3_2B: 161.3 us (automatic choice before this sweep: 167.9 us)
    {144, 2048, 6, 2, 1, 64, 1, 1},
    // ---- batch 4 per GPU, 24x128 MCEP (round-4 sweep, profiles/r04_tune_sweep_b4.txt)
    // g_d1_3B: 22.3 us (automatic choice before this sweep: 24.1 us)
    {10, 512, 24, 2, 1, 64, 4, 1},
    // g_d2_3B: 31.6 us (automatic choice before this sweep: 34.2 us)
    {20, 1024, 12, 2, 1, 64, 1, 1},
    // g_r_h1_3B: 53.8 us (automatic choice before this sweep: 59.6 us)
    {24, 2048, 12, 2, 1, 64, 1, 1},
    // g_r_h2_3B: 31.9 us (automatic choice before this sweep: 33.9 us)
    {48, 512, 12, 2, 1, 64, 1, 1},
    // g_u1_3B: 83.8 us (automatic choice before this sweep: 93.3 us)
    {40, 2048, 12, 2, 1, 64, 1, 1},
    // g_u2_3B: 81.1 us (automatic choice before this sweep: 93.8 us)
    {40, 1024, 24, 2, 1, 64, 1, 1},
    // g_o1_3B: 24.7 us (automatic choice before this sweep: 25.2 us)
    {60, 24, 48, 2, 1, 64, 4, 2},
    // g_h1_3B: 18.1 us (automatic choice before this sweep: 23.1 us)
    {8, 256, 48, 2, 1, 64, 2, 4},
    // d_d1_2B: 90.4 us (automatic choice before this sweep: 89.8 us)
    {18, 512, 96, 2, 1, 64, 1, 8},
    // d_d2_2B: 85.0 us (automatic choice before this sweep: 85.9 us)
    {36, 1024, 24, 2, 1, 64, 1, 1},
    // d_d3_2B: 238.5 us (automatic choice before this sweep: 240.1 us)
    {144, 2048, 12, 2, 1, 128, 1, 1},
    // ---- batch 16 per GPU, 24x128 MCEP (round-4 sweep, profiles/r04_tune_sweep_b16.txt)
    // g_d1_3B: 59.0 us (automatic choice before this sweep: 69.6 us)
    {10, 512, 96, 2, 1, 64, 1, 4},
    // g_d2_3B: 92.8 us (automatic choice before this sweep: 105.2 us)
    {20, 1024, 48, 2, 1, 64, 2, 2},
    // g_r_h1_3B: 175.0 us (automatic choice before this sweep: 203.4 us)
    {24, 2048, 48, 2, 1, 64, 2, 1},
    // g_r_h2_3B: 92.0 us (automatic choice before this sweep: 105.6 us)
    {48, 512, 48, 2, 1, 64, 1, 1},
    // g_u1_3B: 285.4 us (automatic choice before this sweep: 339.3 us)
    {40, 2048, 48, 2, 1, 64, 2, 1},
    // g_u2_3B: 281.5 us (automatic choice before this sweep: 354.2 us)
    {40, 1024, 96, 2, 1, 64, 2, 1},
    // g_o1_3B: 65.4 us (automatic choice before this sweep: 71.8 us)
    {60, 24, 192, 2, 1, 64, 4, 2},
    // g_h1_3B: 39.8 us (automatic choice before this sweep: 37.6 us)
    {8, 256, 192, 2, 1, 64, 4, 4},
    // d_d1_2B: 272.2 us (automatic choice before this sweep: 273.6 us)
    {18, 512, 384, 2, 1, 64, 1, 8},
    // d_d2_2B: 275.6 us (automatic choice before this sweep: 290.2 us)
    {36, 1024, 96, 2, 1, 64, 1, 2},
    // d_d3_2B: 812.2 us (automatic choice before this sweep: 820.9 us)
    {144, 2048, 48, 2, 1, 128, 1, 1},
    // ---- batch 64 per GPU, 24x128 MCEP (round-4 sweep, profiles/r04_tune_sweep_b64.txt)
    // g_d1_3B: 160.4 us (automatic choice before this sweep: 183.7 us)
    {10, 512, 384, 2, 1, 64, 2, 8},
    // g_d2_3B: 295.8 us (automatic choice before this sweep: 376.4 us)
    {20, 1024, 192, 2, 1, 64, 1, 4},
    // g_r_h1_3B: 631.8 us (automatic choice before this sweep: 747.3 us)
    {24, 2048, 192, 2, 1, 128, 2, 1},
    // g_r_h2_3B: 334.5 us (automatic choice before this sweep: 431.4 us)
    {48, 512, 192, 2, 1, 64, 2, 1},
    // g_u1_3B: 1050.1 us (automatic choice before this sweep: 1246.6 us)
    {40, 2048, 192, 2, 1, 128, 2, 1},
    // g_u2_3B: 1066.6 us (automatic choice before this sweep: 1252.6 us)
    {40, 1024, 384, 2, 1, 128, 1, 4},
    // g_o1_3B: 226.1 us (automatic choice before this sweep: 257.7 us)
    {60, 24, 768, 2, 1, 64, 1, 8},
    // g_h1_3B: 124.3 us (automatic choice before this sweep: 131.8 us)
    {8, 256, 768, 2, 1, 64, 1, 16},
    // d_d1_2B: 1036.3 us (automatic choice before this sweep: 1110.6 us)
    {18, 512, 1536, 2, 1, 128, 2, 16},
    // d_d2_2B: 1002.2 us (automatic choice before this sweep: 1042.7 us)
    {36, 1024, 384, 2, 1, 64, 1, 4},
    // d_d3_2B: 3522.6 us (automatic choice before this sweep: 4766.5 us)
    {144, 2048, 192, 2, 1, 128, 2, 1},
};
