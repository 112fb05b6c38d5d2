// Measured filter-gradient configurations: tools/tune_igemm.py --sweep on MI355X.  {K tiles, Cout, pixel chunks (all
// applications), groups, plain stores?, tile columns, K groups, split-K}.  Shapes not listed go by the cost model in wgrad_config.
static const TunedWgrad kTunedWgrad[] = {
    {0, 0, 0, 0, 0, 64, 1, 1},        // (placeholder: never matches)
};
