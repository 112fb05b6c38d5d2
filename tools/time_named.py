"""time the north-star kernel alone (G.h1 gated conv1d forward, B=256) in every math mode"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
import bench
from voice_converter_cyclegan_amd._lib import lib
for mode, code in (('fp32', 0), ('bf16', 1), ('bf16x3', 2)):
    lib.vc_set_math_mode(code)
    r = bench.named_kernel_h1(lib, 'cuda:0')
    print(mode, '%.1f us' % r['us'], 'hbm frac %.3f' % r['hbm']['frac'])
lib.vc_set_math_mode(0)
