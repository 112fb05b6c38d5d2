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
  -DVC_WGSPAN        (with -DVC_TIMELINE) thread 0 of EVERY workgroup of the launch stores its entry and exit stamps; exported as
                     vc_debug_wgspan(out[4 * 8192 + 4]), read by tools/wg_span.py: how the lifetimes of a launch's workgroups
                     are spread against the kernel's duration (dispatch ramp, imbalance, rounds)
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
#ifdef VC_WGSPAN
__device__ unsigned long long g_span[4 * 8192 + 4];      // workgroup w of the LAST launch: [4 w] entry, [4 w + 1] exit (s_memrealtime, 100 MHz, chip-wide), [4 w + 2] lifetime in shader cycles, [4 w + 3] HW_ID | XCC_ID << 32; tail: grid x, y, z
extern "C" int vc_debug_wgspan(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof g_span); }
#endif
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
#ifdef VC_WGSPAN
    const unsigned wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned long long wg_t0 = __builtin_readcyclecounter(), wg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
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
#ifdef VC_WGSPAN
    __syncthreads();
    if (threadIdx.x == 0 && wg_lin < 8192) {
        g_span[4 * wg_lin] = wg_r0; g_span[4 * wg_lin + 1] = __builtin_amdgcn_s_memrealtime();
        g_span[4 * wg_lin + 2] = __builtin_readcyclecounter() - wg_t0;
        g_span[4 * wg_lin + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
        if (wg_lin == 0) { g_span[4 * 8192] = gridDim.x; g_span[4 * 8192 + 1] = gridDim.y; g_span[4 * 8192 + 2] = gridDim.z; }
    }
#endif
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
