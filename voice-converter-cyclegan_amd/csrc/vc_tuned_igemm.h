// Measured implicit-GEMM configurations (tools/tune_igemm.py on MI355X).  {a_trans, M, Ptot, krows, groups, tile rows, K groups, split-K}
// Shapes not listed go by the cost model in igemm_config.
static const TunedCfg kTunedIgemm[] = {
    {-1, 0, 0, 0, 0, 0, 0, 0},
};
