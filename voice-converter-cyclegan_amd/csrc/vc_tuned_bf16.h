// Measured configurations of conv_bf16_kernel (the 16-bit data path of the bf16 mode): tools/tune_igemm.py --math bf16 --sweep
// on MI355X.  {M, Ptot, K octets, groups, tile rows, LDS ring stages, split-K}.  Shapes not listed go by the tile-count rule
// in conv_bf16_config.
static const TunedBf16 kTunedBf16[] = {
    {0, 0, 0, 0, 128, 2, 1},        // (placeholder: never matches)
};
