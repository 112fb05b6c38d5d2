"""Instrumentation of conv_bf16_kernel for measurements (profiles/r05_bf16_conv_phases.txt): every workgroup stamps entry / first barrier
passed / K loop done / exit (shader cycles + the 100 MHz clock), summed per configuration in device memory (vc_probe_conv_bf16_phases) and
kept per workgroup for the last launch (vc_probe_conv_bf16_read); VC_PROBE_PRINT=1 prints every launch's shape, configuration and time.
    python tools/conv_probe/apply.py          # edits csrc/vc_conv_bf16.hip IN PLACE (the build id changes: this is not the shipped library)
    gpurun -- 'python tools/conv_probe/probe_step.py 64 serial; python tools/conv_probe/probe_conv.py 128'
    git checkout voice-converter-cyclegan_amd/csrc/vc_conv_bf16.hip
The stamps cost about 7 % of a bf16 step."""
p='voice-converter-cyclegan_amd/csrc/vc_conv_bf16.hip'; s=open(p).read()
def rep(old,new):
    global s
    assert old in s, old[:60]
    s=s.replace(old,new,1)
rep('''template <int BM, int NST, int ROLE, bool SPEC = false> __global__ __launch_bounds__(SPEC ? 512 : 256, (!SPEC && NST <= 2) ? 2 : 1)
void conv_bf16_kernel(const ConvBf16Args p) {
    touch_kernarg(p);''','''__device__ unsigned long long g_cprobe[8 * 4096 + 8];
__device__ unsigned long long g_cphase[32 * 8];
extern "C" int vc_probe_conv_bf16_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cprobe), sizeof(unsigned long long) * (8 * 4096 + 8)); }
extern "C" int vc_probe_conv_bf16_phases(unsigned long long* out, int reset) {
    int r = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cphase), sizeof(unsigned long long) * 32 * 8);
    if (!r && reset) { static unsigned long long z[32 * 8]; r = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cphase), z, sizeof(z)); }
    return r;
}
template <int BM, int NST, int ROLE, bool SPEC = false> __global__ __launch_bounds__(SPEC ? 512 : 256, (!SPEC && NST <= 2) ? 2 : 1)
void conv_bf16_kernel(const ConvBf16Args p) {
    touch_kernarg(p);
    const unsigned long long pt0 = __builtin_readcyclecounter(), pr0 = __builtin_amdgcn_s_memrealtime(); unsigned long long pt1 = 0, pt2 = 0;''')
# stamp after the first barrier of the multiplying path(s)
rep('''            const int stn = st == 0 ? NST - 1 : st - 1;         // (c + NST - 1) % NST: the stage chunk c - 1 has just left
            multiply_chunk(''','''            if (c == 0) pt1 = __builtin_readcyclecounter();
            const int stn = st == 0 ? NST - 1 : st - 1;         // (c + NST - 1) % NST: the stage chunk c - 1 has just left
            multiply_chunk(''')
rep('''            __builtin_amdgcn_sched_barrier(0);
            multiply_chunk(st, 0, 0, c, std::false_type{});''','''            __builtin_amdgcn_sched_barrier(0);
            if (c == 0) pt1 = __builtin_readcyclecounter();
            multiply_chunk(st, 0, 0, c, std::false_type{});''')
rep('''    __syncthreads();                                        // (drains the ring's trailing copies: vmcnt(0))
''','''    __syncthreads();                                        // (drains the ring's trailing copies: vmcnt(0))
    pt2 = __builtin_readcyclecounter();
''')
rep('''            if (atomic) atomicAdd(Yg + o, v); else Yg[o] = v;
        }
    }
}''','''            if (atomic) atomicAdd(Yg + o, v); else Yg[o] = v;
        }
    }
    {
        const int w = blockIdx.x + gridDim.x * blockIdx.z;
        if (tid == 0) {
            const int slot = (((BM == 256 ? 0 : BM == 128 ? 1 : 3) + (NST == 4 ? 1 : 0)) * 2 + ROLE) + (SPEC ? 16 : 0);   // 256/3:0  128/2:1  128/4:2  64/2:3  64/4:4
            unsigned long long* f = g_cphase + 8 * slot;
            const unsigned long long e3 = __builtin_readcyclecounter();
            atomicAdd(f + 0, pt1 - pt0); atomicAdd(f + 1, pt2 - pt1); atomicAdd(f + 2, e3 - pt2); atomicAdd(f + 3, (unsigned long long)cn * (BM / 64) * 2 * 4 * 32); atomicAdd(f + 4, 1ull);
            if (w == 0) atomicAdd(f + 5, 1ull);
        }
        if (tid == 0 && w < 4096) {
            unsigned long long* q = g_cprobe + 8 * w;
            const unsigned long long pt3 = __builtin_readcyclecounter();
            q[0] = pr0; q[1] = __builtin_amdgcn_s_memrealtime(); q[2] = pt1 - pt0; q[3] = pt2 - pt1; q[4] = pt3 - pt2; q[5] = cn; q[6] = BM * 1000 + NST + (SPEC ? 100 : 0);
            if (w == 0) { g_cprobe[8 * 4096] = gridDim.x; g_cprobe[8 * 4096 + 1] = gridDim.z; }
        }
    }
}''')
rep('''    dim3 grid(a.mtiles * a.ntiles, 1, a.groups * a.splitk);
    ProfScope ps(prof_kind, flops, s);''','''    dim3 grid(a.mtiles * a.ntiles, 1, a.groups * a.splitk);
    static const bool pp = getenv("VC_PROBE_PRINT") != nullptr;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (pp) { hipEventCreate(&pe0); hipEventCreate(&pe1); hipEventRecord(pe0, s); }
    struct PP { bool on; hipEvent_t e0, e1; hipStream_t s; const ConvBf16Args& a; double fl; int kind;
        ~PP() { if (!on) return; hipEventRecord(e1, s); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                fprintf(stderr, "PP role %d M %5d P %6d koct %4d g %d ntaps %2d | bm %3d nst %d sk %d tiles %5d | %8.1f us %7.1f GFLOP %6.0f TF acc %d vec4 %d add %d\\n", kind, a.M, a.Ptot, a.koct, a.groups, a.ntaps,
                        a.bm, a.nst, a.splitk, a.mtiles * a.ntiles * a.groups, ms * 1e3, fl / 1e9, fl / (ms * 1e-3) / 1e12, (int)a.accumulate, (int)a.vec4, a.addend ? 1 : 0);
                hipEventDestroy(e0); hipEventDestroy(e1); } } ppg{pp, pe0, pe1, s, a, flops, prof_kind};
    ProfScope ps(prof_kind, flops, s);''')
if '#include <cstdio>' not in s: s=s.replace('#include "vc_internal.h"','#include "vc_internal.h"\n#include <cstdio>',1)
open(p,'w').write(s)
