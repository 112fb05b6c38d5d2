"""Regenerates tools/glu_timeline.patch: in-kernel phase stamps for csrc/vc_conv_glu.hip (gated_conv1d_glu_kernel), kept OUT of
the product source.  Apply, build with VC_EXTRA_FLAGS=-DVC_GLU_TIMELINE, run tools/glu_probe.py, `git checkout` the file."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 'voice-converter-cyclegan_amd/csrc/vc_conv_glu.hip'


def sub(s, a, b):
    assert s.count(a) == 1, (s.count(a), a)
    return s.replace(a, b, 1)


def main():
    path = os.path.join(ROOT, P)
    orig = open(path).read()
    s = orig
    s = sub(s, 'template <int CIN, int KW, bool BF16, bool RAW>\n__global__ __launch_bounds__(256, 1) void gated_conv1d_glu_kernel(const WsConvArgs a) {',
            '''#ifdef VC_GLU_TIMELINE
// workgroup (0,0,0), thread 0: [0] entry -> fragments + tile 0 in LDS, [1] -> fragments in registers, [2] sum of trips: start ->
// counted wait reached, [3] sum of trips: wait -> barrier passed, [4] loop end -> kernel end, [5] launches, [6] trips
__device__ unsigned long long g_glu_tl[8];
extern "C" int vc_debug_glu_timeline(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_glu_tl), z, sizeof z); return 0; }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_glu_tl), 8 * sizeof(unsigned long long));
}
#define VC_GTL(x) do { if (tl_on) { x; } } while (0)
#else
#define VC_GTL(x) do { } while (0)
#endif
template <int CIN, int KW, bool BF16, bool RAW>
__global__ __launch_bounds__(256, 1) void gated_conv1d_glu_kernel(const WsConvArgs a) {
#ifdef VC_GLU_TIMELINE
    const bool tl_on = threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    unsigned long long tl0 = __builtin_readcyclecounter(), tl1 = 0, tl2 = 0, tl3 = 0, tla = 0, tlb = 0, ts = 0, tw = 0, ntr = 0;
#endif''')
    s = sub(s, '    __syncthreads();                                        // filter fragments and tile 0 are in LDS\n',
            '    __syncthreads();                                        // filter fragments and tile 0 are in LDS\n    VC_GTL(tl1 = __builtin_readcyclecounter());\n')
    s = sub(s, '    f32x16 accv[2], accg[2];\n', '    VC_GTL(tl2 = __builtin_readcyclecounter());\n    f32x16 accv[2], accg[2];\n')
    s = sub(s, '        f32x16 o;\n        if constexpr (!OVERLAP) {', '        f32x16 o;\n        VC_GTL(ts = __builtin_readcyclecounter());\n        if constexpr (!OVERLAP) {')
    s = sub(s, '                    store_plane(pl, yrs, pp.yb, pp.ok, 0);\n                    wait_counted(xq);\n',
            '                    store_plane(pl, yrs, pp.yb, pp.ok, 0);\n                    VC_GTL(tw = __builtin_readcyclecounter());\n                    wait_counted(xq);\n')
    s = sub(s, '            wait_counted(xq);\n            [&]<int... J>', '            VC_GTL(tw = __builtin_readcyclecounter());\n            wait_counted(xq);\n            [&]<int... J>')
    s = sub(s, '        __syncthreads();                                    // slot P ^ 1 is complete; slot P may be overwritten by the next trip\n',
            '        __syncthreads();                                    // slot P ^ 1 is complete; slot P may be overwritten by the next trip\n        VC_GTL(tla += tw - ts; tlb += __builtin_readcyclecounter() - tw; ++ntr);\n')
    s = sub(s, '    // ---- the last tile\'s GLU + stores', '    VC_GTL(tl3 = __builtin_readcyclecounter());\n    // ---- the last tile\'s GLU + stores')
    # end of kernel: before the launcher declaration
    s = sub(s, '            store_tile(accv[0], accg[0], o, pp);\n        }\n    }\n}\n',
            '''            store_tile(accv[0], accg[0], o, pp);
        }
    }
#ifdef VC_GLU_TIMELINE
    if (tl_on) {
        const unsigned long long te = __builtin_readcyclecounter();
        atomicAdd(&g_glu_tl[0], tl1 - tl0); atomicAdd(&g_glu_tl[1], tl2 - tl1); atomicAdd(&g_glu_tl[2], tla); atomicAdd(&g_glu_tl[3], tlb);
        atomicAdd(&g_glu_tl[4], te - tl3); atomicAdd(&g_glu_tl[5], 1ull); atomicAdd(&g_glu_tl[6], ntr);
    }
#endif
}
''')
    open(path, 'w').write(s)
    try:
        d = subprocess.run(['git', 'diff', '--', P], cwd=ROOT, capture_output=True, text=True, check=True).stdout
    finally:
        open(path, 'w').write(orig)
    hdr = ('Timing probe for csrc/vc_conv_glu.hip (never ship a build with it): -DVC_GLU_TIMELINE makes workgroup (0,0,0) of every launch\n'
           'add its phase durations to a device array (vc_debug_glu_timeline), read by tools/glu_probe.py.  Regenerate with\n'
           'tools/make_glu_timeline_patch.py.\n\n')
    open(os.path.join(ROOT, 'tools', 'glu_timeline.patch'), 'w').write(hdr + d)
    print('tools/glu_timeline.patch: %d lines' % (hdr + d).count('\n'))


if __name__ == '__main__':
    main()
