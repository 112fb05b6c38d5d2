"""Regenerates tools/glu_timeline.patch: in-kernel phase stamps for csrc/vc_conv_glu.hip (gated_conv1d_glu_kernel), kept OUT of
the product source.  Apply (`git apply tools/glu_timeline.patch`), build with VC_EXTRA_FLAGS=-DVC_GLU_TIMELINE, run
tools/glu_probe.py, `git checkout` the file.  Stamps: s_memtime (shader cycles) AND s_memrealtime (100 MHz): the ratio is the clock
the kernel actually ran at."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 'voice-converter-cyclegan_amd/csrc/vc_conv_glu.hip'


def sub(s, a, b, count=1):
    assert s.count(a) == count, (s.count(a), a)
    return s.replace(a, b)


def main():
    path = os.path.join(ROOT, P)
    orig = open(path).read()
    s = orig
    s = sub(s, '// LDS map of the kernel (one allocation, shared by the two bodies)\n',
            '''#ifdef VC_GLU_TIMELINE
// slot = 2 * which + S (which: 0 = workgroup (0,0,0), 1 = the last workgroup of the grid; S: K set, lane 0 of waves 0 / 4); per slot:
// [0] entry -> prologue barrier passed, [1] sum over trips: start -> K loop done, [2] sum: -> hand-over out, own half copied, [3] sum: ->
// barrier passed, [4] loop end -> kernel end, [5] launches, [6] trips, [7] kernel cycles, [8] kernel real-time ticks (100 MHz),
// [9] entry -> prologue loads issued, [10] -> landed (vmcnt 0), [11] -> tile 0 staged, tile 1 requested, [12] -> barrier passed
__device__ unsigned long long g_glu_tl[4][16];
extern "C" int vc_debug_glu_timeline(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[4][16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_glu_tl), z, sizeof z); return 0; }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_glu_tl), sizeof(unsigned long long) * 64);
}
#define VC_GTL(x) do { if (tl_on) { x; } } while (0)
#else
#define VC_GTL(x) do { } while (0)
#endif
// LDS map of the kernel (one allocation, shared by the two bodies)
''')
    s = sub(s, '    typedef GluGeo<CIN, KW, BF16> GG;\n    typedef typename std::conditional<BF16, __bf16, float>::type T;\n',
            '''#ifdef VC_GLU_TIMELINE
    const int tl_which = (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) ? 0 :
                         (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1) ? 1 : -1;
    const bool tl_on = (threadIdx.x & 255) == 0 && tl_which >= 0;
    unsigned long long tl0 = __builtin_readcyclecounter(), tr0 = __builtin_amdgcn_s_memrealtime(), tl1 = 0, tl3 = 0, tp1 = 0, tp2 = 0, tp3 = 0, ta = 0, tb = 0, tc = 0, sa = 0, sb = 0, sc = 0, ntr = 0;
#endif
    typedef GluGeo<CIN, KW, BF16> GG;
    typedef typename std::conditional<BF16, __bf16, float>::type T;
''')
    s = sub(s, '    wait_all2(xr0, xr1);\n', '    VC_GTL(tp1 = __builtin_readcyclecounter());\n    wait_all2(xr0, xr1);\n    VC_GTL(tp2 = __builtin_readcyclecounter());\n')
    s = sub(s, '    // dummy "tile -2" stores:', '    VC_GTL(tp3 = __builtin_readcyclecounter());\n    // dummy "tile -2" stores:')
    s = sub(s, '    __syncthreads();                                        // fragment image (if any) and tile 0 are in LDS\n',
            '    __syncthreads();                                        // fragment image (if any) and tile 0 are in LDS\n    VC_GTL(tl1 = __builtin_readcyclecounter());\n')
    s = sub(s, '            const OutPos pp = out_pos(scur); advance(scur);   // the previous tile\'s output position\n',
            '            VC_GTL(ta = __builtin_readcyclecounter());\n            const OutPos pp = out_pos(scur); advance(scur);   // the previous tile\'s output position\n')
    s = sub(s, '            // hand-over: the half of the partial sums the partner finishes goes to LDS',
            '            VC_GTL(tb = __builtin_readcyclecounter());\n            // hand-over: the half of the partial sums the partner finishes goes to LDS')
    s = sub(s, '            __syncthreads();                                // hand-over written; slot P ^ 1 is complete; slot P may be overwritten by the next trip\n',
            '            VC_GTL(tc = __builtin_readcyclecounter());\n            __syncthreads();                                // hand-over written; slot P ^ 1 is complete; slot P may be overwritten by the next trip\n            VC_GTL(sa += tb - ta; sb += tc - tb; sc += __builtin_readcyclecounter() - tc; ++ntr);\n')
    s = sub(s, '        // ---- the last tile: the partner\'s half lies in the buffer of the last trip\'s parity\n', '        VC_GTL(tl3 = __builtin_readcyclecounter());\n        // ---- the last tile: the partner\'s half lies in the buffer of the last trip\'s parity\n')
    s = sub(s, '        if constexpr (S == 1) __builtin_amdgcn_s_setprio(0);\n    }\n}\n',
            '''        if constexpr (S == 1) __builtin_amdgcn_s_setprio(0);
    }
#ifdef VC_GLU_TIMELINE
    if (tl_on) {
        const unsigned long long te = __builtin_readcyclecounter(), tre = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o_ = g_glu_tl[2 * tl_which + S];
        atomicAdd(&o_[0], tl1 - tl0); atomicAdd(&o_[1], sa); atomicAdd(&o_[2], sb); atomicAdd(&o_[3], sc);
        atomicAdd(&o_[4], te - tl3); atomicAdd(&o_[5], 1ull); atomicAdd(&o_[6], ntr); atomicAdd(&o_[7], te - tl0); atomicAdd(&o_[8], tre - tr0);
        atomicAdd(&o_[9], tp1 - tl0); atomicAdd(&o_[10], tp2 - tp1); atomicAdd(&o_[11], tp3 - tp2); atomicAdd(&o_[12], tl1 - tp3);
    }
#endif
}
''')
    open(path, 'w').write(s)
    try:
        d = subprocess.run(['git', 'diff', '--', P], cwd=ROOT, capture_output=True, text=True, check=True).stdout
    finally:
        open(path, 'w').write(orig)
    hdr = ('Timing probe for csrc/vc_conv_glu.hip (never ship a build with it): -DVC_GLU_TIMELINE makes two workgroups of every launch\n'
           'add their phase durations to a device array (vc_debug_glu_timeline), read by tools/glu_probe.py.  Regenerate with\n'
           'tools/make_glu_timeline_patch.py.\n\n')
    open(os.path.join(ROOT, 'tools', 'glu_timeline.patch'), 'w').write(hdr + d)
    print('tools/glu_timeline.patch: %d lines' % (hdr + d).count('\n'))


if __name__ == '__main__':
    main()
