"""Regenerates tools/igemm_timeline.patch against the current csrc/vc_conv_igemm.hip (run from the repo root after editing the
kernel; tests/test_abi_cpu.py checks that the committed patch still applies).  The probe is kept OUT of the product source."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 'voice-converter-cyclegan_amd/csrc/vc_conv_igemm.hip'
HDR = '''Timing probe for csrc/vc_conv_igemm.hip (never ship a build with it).
Apply with `git apply tools/igemm_timeline.patch`, rebuild with VC_EXTRA_FLAGS=-DVC_TIMELINE, run tools/clk_probe.py, `git checkout` the file.
  -DVC_TIMELINE      workgroup (0,0,0) of every launch adds its phase durations (entry -> tables + first loads issued -> first
                     barrier passed -> K loop done -> end, in s_memtime cycles), the launch count and the K chunks per part to a
                     device array; exported as vc_debug_timeline(out[8], reset), read by tools/clk_probe.py
Round 3 (profiles/r03_launch_timeline.txt) carried load-ablation and clock-probe switches in the same patch; they are in the
history (tools/igemm_ablation.patch at commit 21aeae6).  Regenerate with tools/make_timeline_patch.py.

'''


def sub(s, a, b):
    assert s.count(a) >= 1, a
    return s.replace(a, b, 1)


def main():
    path = os.path.join(ROOT, P)
    orig = open(path).read()
    s = orig
    a = 'template <int BM, bool AT, int KG, int MATH, bool FUSE = false>\n__global__ __launch_bounds__'
    s = sub(s, a, '''#ifdef VC_TIMELINE
// stamps of workgroup (0,0,0): entry -> tables + first loads issued -> first barrier passed (first MFMA) -> K loop done -> end;
// summed over launches, [5] counts launches, [6] sums the K chunks per part
__device__ unsigned long long g_tl[8];
extern "C" int vc_debug_timeline(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl), z, sizeof z); return 0; }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), 8 * sizeof(unsigned long long));
}
#define VC_TL(i) do { if (tl_on) tl[i] = __builtin_readcyclecounter(); } while (0)
#else
#define VC_TL(i) do { } while (0)
#endif
''' + a)
    a = '    const int tid = threadIdx.x, lane = tid & 63;\n    const int kg = __builtin_amdgcn_readfirstlane(tid / GT);'
    s = sub(s, a, '''    const int tid = threadIdx.x, lane = tid & 63;
#ifdef VC_TIMELINE
    const bool tl_on = threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    unsigned long long tl[5] = {};
    VC_TL(0);
#endif
    const int kg = __builtin_amdgcn_readfirstlane(tid / GT);''')
    a = '    fetch_tab(c0 + 2);\n    items(LD0{}, ST0{}, S0{}, S0{}, 0);\n    __syncthreads();\n'
    s = sub(s, a, '    fetch_tab(c0 + 2);\n    VC_TL(1);\n    items(LD0{}, ST0{}, S0{}, S0{}, 0);\n    __syncthreads();\n    VC_TL(2);\n')
    a = '    // ---- epilogue: every K group parks its tile in its own LDS region; all threads then sum the groups, add'
    s = sub(s, a, '    VC_TL(3);\n' + a)
    a = '            if (atomic) atomicAdd(Yg + o, v); else Yg[o] = v;\n        }\n    }\n#undef VC_PIN_S\n}\n'
    s = sub(s, a, '''            if (atomic) atomicAdd(Yg + o, v); else Yg[o] = v;
        }
    }
#undef VC_PIN_S
#ifdef VC_TIMELINE
    if (tl_on) {
        tl[4] = __builtin_readcyclecounter();
        for (int q = 0; q < 4; ++q) atomicAdd(&g_tl[q], tl[q + 1] - tl[q]);
        atomicAdd(&g_tl[5], 1ull); atomicAdd(&g_tl[6], (unsigned long long)cps);
    }
#endif
}
''')
    open(path, 'w').write(s)
    try:
        d = subprocess.run(['git', 'diff', '--', P], cwd=ROOT, capture_output=True, text=True, check=True).stdout
    finally:
        open(path, 'w').write(orig)
    open(os.path.join(ROOT, 'tools', 'igemm_timeline.patch'), 'w').write(HDR + d)
    print('tools/igemm_timeline.patch: %d lines' % (HDR + d).count('\n'))


if __name__ == '__main__':
    main()
