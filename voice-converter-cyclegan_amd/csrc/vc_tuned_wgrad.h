// Measured filter-gradient configurations: tools/tune_igemm.py --sweep on MI355X (round 3; profiles/r03_tune_sweep_*.txt), as the
// train step launches them (plain stores; the generator's layers reduce first hop + cycle hop in one launch).  {K tiles,
// Cout, pixel chunks (all applications), groups, plain stores?, tile columns, K groups, split-K}.  Shapes not listed go by the
// cost model in wgrad_config.
static const TunedWgrad kTunedWgrad[] = {
    // ---- batch 8 per GPU, 24x128 MCEP
    // g_d1_3B: 35.9 us (automatic choice before this sweep: 39.2 us)
    {10, 512, 48, 2, 1, 64, 4, 1},
    // g_d2_3B: 53.0 us (automatic choice before this sweep: 57.2 us)
    {20, 1024, 24, 2, 1, 64, 1, 1},
    // g_r_h1_3B: 95.2 us (automatic choice before this sweep: 109.7 us)
    {24, 2048, 24, 2, 1, 64, 1, 1},
    // g_r_h2_3B: 52.0 us (automatic choice before this sweep: 57.7 us)
    {48, 512, 24, 2, 1, 64, 1, 1},
    // g_u1_3B: 153.1 us (automatic choice before this sweep: 184.4 us)
    {40, 2048, 24, 2, 1, 64, 1, 1},
    // g_u2_3B: 153.4 us (automatic choice before this sweep: 189.9 us)
    {40, 1024, 48, 2, 1, 64, 2, 1},
    // g_o1_3B: 37.8 us (automatic choice before this sweep: 40.7 us)
    {60, 24, 96, 2, 1, 64, 4, 2},
    // g_h1_3B: 24.7 us (automatic choice before this sweep: 24.8 us)
    {8, 256, 96, 2, 1, 64, 4, 4},
    // d_d1_2B: 149.1 us (automatic choice before this sweep: 178.7 us)
    {18, 512, 192, 2, 1, 64, 1, 8},
    // d_d2_2B: 152.9 us (automatic choice before this sweep: 172.4 us)
    {36, 1024, 48, 2, 1, 64, 2, 1},
    // d_d3_2B: 419.5 us (automatic choice before this sweep: 423.4 us)
    {144, 2048, 24, 2, 1, 128, 1, 1},
    // ---- batch 32 per GPU, 24x128 MCEP
    // g_d1_3B: 93.4 us (automatic choice before this sweep: 107.7 us)
    {10, 512, 192, 2, 1, 64, 2, 8},
    // g_d2_3B: 162.2 us (automatic choice before this sweep: 208.9 us)
    {20, 1024, 96, 2, 1, 64, 2, 2},
    // g_r_h1_3B: 335.5 us (automatic choice before this sweep: 390.3 us)
    {24, 2048, 96, 2, 1, 128, 2, 1},
    // g_r_h2_3B: 175.3 us (automatic choice before this sweep: 209.6 us)
    {48, 512, 96, 2, 1, 64, 1, 1},
    // g_u1_3B: 552.2 us (automatic choice before this sweep: 644.7 us)
    {40, 2048, 96, 2, 1, 64, 2, 1},
    // g_u2_3B: 552.1 us (automatic choice before this sweep: 690.4 us)
    {40, 1024, 192, 2, 1, 64, 2, 1},
    // g_o1_3B: 121.2 us (automatic choice before this sweep: 133.0 us)
    {60, 24, 384, 2, 1, 64, 4, 2},
    // g_h1_3B: 69.3 us (automatic choice before this sweep: 68.9 us)
    {8, 256, 384, 2, 1, 64, 4, 4},
    // d_d1_2B: 532.8 us (automatic choice before this sweep: 577.0 us)
    {18, 512, 768, 2, 1, 64, 2, 8},
    // d_d2_2B: 526.3 us (automatic choice before this sweep: 597.2 us)
    {36, 1024, 192, 2, 1, 64, 2, 2},
    // d_d3_2B: 1635.7 us (automatic choice before this sweep: 1688.1 us)
    {144, 2048, 96, 2, 1, 128, 1, 1},
    // ---- batch 64 per GPU, 36x256 MCEP
    // g_d1_3B: 305.0 us (automatic choice before this sweep: 386.9 us)
    {10, 512, 768, 2, 1, 64, 2, 8},
    // g_d2_3B: 573.4 us (automatic choice before this sweep: 681.8 us)
    {20, 1024, 384, 2, 1, 64, 1, 4},
    // g_r_h1_3B: 1247.3 us (automatic choice before this sweep: 1360.1 us)
    {24, 2048, 384, 2, 1, 128, 1, 2},
    // g_r_h2_3B: 652.6 us (automatic choice before this sweep: 756.4 us)
    {48, 512, 384, 2, 1, 128, 1, 4},
    // g_u1_3B: 2064.8 us (automatic choice before this sweep: 2209.4 us)
    {40, 2048, 384, 2, 1, 128, 1, 2},
    // g_u2_3B: 2054.8 us (automatic choice before this sweep: 2216.0 us)
    {40, 1024, 768, 2, 1, 128, 1, 4},
    // g_o1_3B: 435.5 us (automatic choice before this sweep: 501.9 us)
    {60, 36, 1536, 2, 1, 64, 2, 8},
    // g_h1_3B: 421.1 us (automatic choice before this sweep: 493.4 us)
    {15, 256, 1536, 2, 1, 128, 1, 16},
    // d_d1_2B: 2929.2 us (automatic choice before this sweep: 2925.8 us)
    {18, 512, 4608, 2, 1, 128, 2, 16},
    // d_d2_2B: 2846.8 us (automatic choice before this sweep: 2915.0 us)
    {36, 1024, 1152, 2, 1, 128, 1, 8},
    // d_d3_2B: 9540.9 us (automatic choice before this sweep: 9729.3 us)
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
