"""Times the north star's named kernel (vc_gated_conv1d_glu_fwd: G.h1 value + gate conv1d + GLU, module.py:161-163) with HIP
events around everything a call launches, at several batch sizes, fp32 and bf16.  Run on the GPU box:

    python tools/bench_named.py [--batches 256,1024,4096]
    VC_GLU_PACKED=1 python tools/bench_named.py        # the round-2 two-launch form (pack kernel + 8-wave kernel)
    VC_GLU_WAVES8=1 python tools/bench_named.py        # in-kernel filter slab, one 8-wave workgroup per CU
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='256,1024,4096')
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    from voice_converter_cyclegan_amd._lib import lib
    tag = 'packed' if os.environ.get('VC_GLU_PACKED') else 'waves8' if os.environ.get('VC_GLU_WAVES8') else 'default'
    for math, peak in (('fp32', bench.FP32_MFMA_PEAK_TFLOPS), ('bf16', bench.BF16_MFMA_PEAK_TFLOPS)):
        for B in (int(b) for b in args.batches.split(',')):
            r = bench.named_kernel_h1(lib, 'cuda:0', B=B, reps=20, peak=peak, math=math)
            print('%-8s %-5s B=%-5d %8.2f us   HBM %6.1f GB/s = %.3f of 8 TB/s   %7.1f TFLOP/s = %.3f of the %s MFMA peak'
                  % (tag, math, B, r['us'], r['hbm']['achieved_GBs'], r['hbm']['frac'], r['mfma']['achieved_TFLOPs'], r['mfma']['frac'], math), flush=True)


if __name__ == '__main__':
    main()
