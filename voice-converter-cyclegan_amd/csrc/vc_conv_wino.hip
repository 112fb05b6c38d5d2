// vc_conv_wino.hip -- Winograd F(4,3) transforms around the residual blocks' convolutions (reference module.py:66-83: conv1d, 3 taps,
// stride 1, SAME padding; 68 % of a generator's multiply-adds).  Four output frames of a row come from six transformed inputs:
//
//     y[4i .. 4i+3] = A^T [ (G w) . (B^T x[4i-1 .. 4i+4]) ]       (Lavin & Gray 2016, interpolation points 0, +-1, +-2, inf)
//
// so the convolution becomes SIX independent channel GEMMs (one per transform position p) over a quarter of the pixels each: 6 multiplies
// per output quad and channel pair instead of 12.  The GEMMs are 1 x 1 convolutions with 6 G groups on the implicit-GEMM kernel
// (vc_conv_igemm.hip) -- exact fp32 products, fp32 accumulation; the transforms are exact in binary floating point up to the rounding of
// their few additions (measured against the fp64 direct convolution at the residual blocks' sizes: 2-3e-6 of the output's maximum, the
// direct fp32 kernel 0.7-1.2e-6; the operator tests hold both to 1e-4).  The input gradient is the adjoint of the same three steps
// (dx = B [ U^T (A dy) ]), on the same transformed filters U = G w; the filter gradient stays on the direct kernel.
//
// Layouts (H = 1, W = 4 nt):  x [g][b][C][W];  V = B^T x [g][p][b][C][nt];  U [g][p][Cin][Cout];  M = U V [g][p][b][Cout][nt].
#include "vc_internal.h"
#include "vc_device_util.h"
#include "vc_wino.h"

namespace {

struct WinoArgs {
    const float* src; float* dst; const float* bias; const float* addend;
    long src_gs, dst_gs, bias_gs, add_gs;      // group strides (floats); for the transformed side: stride of one g (6 planes inside)
    long plane;                                // floats of one transform plane: B * C * nt
    int G, B, C, nt, accumulate;
    int rows;                                  // G * B * C
};

// a thread = one tile of one (g, b, c) row; consecutive threads = consecutive tiles, then consecutive channels
__device__ __forceinline__ bool wino_row(const WinoArgs& p, int& g, long& row_in_g, int& i) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t / p.nt; i = (int)(t - row * p.nt);
    if (row >= p.rows) return false;
    const long per_g = (long)p.B * p.C;
    g = (int)(row / per_g); row_in_g = row - g * per_g;
    return true;
}

// V[p] = B^T d, d = x[4i-1 .. 4i+4] (zero beyond the row)
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoArgs p) {
    int g, i; long r;
    if (!wino_row(p, g, r, i)) return;
    const float* x = p.src + (long)g * p.src_gs + r * (4 * p.nt) + 4 * i;
    const f32x4 c = *reinterpret_cast<const f32x4*>(x);
    wino_store_input(c, i > 0 ? x[-1] : 0.f, i + 1 < p.nt ? x[4] : 0.f, p.dst + (long)g * p.dst_gs + r * p.nt + i, p.plane);
}

// y[4i .. 4i+3] = A^T M + bias
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoArgs p) {
    int g, i; long r;
    if (!wino_row(p, g, r, i)) return;
    const int c = (int)(r % p.C);
    const float b = p.bias ? p.bias[(long)g * p.bias_gs + c] : 0.f;
    *reinterpret_cast<f32x4*>(p.dst + (long)g * p.dst_gs + r * (4 * p.nt) + 4 * i) = wino_load_output(p.src + (long)g * p.src_gs + r * p.nt + i, p.plane, b);
}

// dM[p] = A dy[4i .. 4i+3]
__global__ __launch_bounds__(256) void wino_dy_kernel(const WinoArgs p) {
    int g, i; long r;
    if (!wino_row(p, g, r, i)) return;
    const f32x4 d = *reinterpret_cast<const f32x4*>(p.src + (long)g * p.src_gs + r * (4 * p.nt) + 4 * i);
    wino_store_dy(d, p.dst + (long)g * p.dst_gs + r * p.nt + i, p.plane);
}

// dx[4i .. 4i+3] (+)= B dV (+ addend: the skip path of a residual block head)
__global__ __launch_bounds__(256) void wino_dx_kernel(const WinoArgs p) {
    int g, i; long r;
    if (!wino_row(p, g, r, i)) return;
    f32x4 dx = wino_load_dx(p.src + (long)g * p.src_gs + r * p.nt + i, p.plane, i > 0, i + 1 < p.nt);
    const long o = r * (4 * p.nt) + 4 * i;
    if (p.addend) dx += *reinterpret_cast<const f32x4*>(p.addend + (long)g * p.add_gs + o);
    float* dst = p.dst + (long)g * p.dst_gs + o;
    if (p.accumulate) dx += *reinterpret_cast<const f32x4*>(dst);
    *reinterpret_cast<f32x4*>(dst) = dx;
}

// U[p][cin][cout] = G w[tap][cin][cout]; a thread = 4 consecutive output channels of one input channel
struct WinoFilterJob { const float* w; float* U; long w_gs, U_gs; int n4; };      // n4 = Cin * Cout / 4
#define VC_WINO_JOBS 16
struct WinoFilterJobs { WinoFilterJob j[VC_WINO_JOBS]; int n, G; unsigned blk0[VC_WINO_JOBS + 1]; };
__global__ __launch_bounds__(256) void wino_filter_kernel(const WinoFilterJobs jj) {
    int k = 0;
    while (k + 1 < jj.n && blockIdx.x >= jj.blk0[k + 1]) ++k;
    const WinoFilterJob& j = jj.j[k];
    const int t = (int)(blockIdx.x - jj.blk0[k]) * 256 + threadIdx.x;
    if (t >= j.n4) return;
    const int g = blockIdx.y;
    const long plane = (long)j.n4 * 4;
    const float* w = j.w + (long)g * j.w_gs + (long)t * 4;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(w), g1 = *reinterpret_cast<const f32x4*>(w + plane), g2 = *reinterpret_cast<const f32x4*>(w + 2 * plane);
    float* u = j.U + (long)g * j.U_gs + (long)t * 4;
    const f32x4 s = (g0 + g2) * (-1.f / 6.f), q = g1 * (1.f / 6.f);
    const f32x4 e = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), h = g1 * (1.f / 12.f);
    *reinterpret_cast<f32x4*>(u) = g0 * 0.25f;
    *reinterpret_cast<f32x4*>(u + plane) = s - q;
    *reinterpret_cast<f32x4*>(u + 2 * plane) = s + q;
    *reinterpret_cast<f32x4*>(u + 3 * plane) = e + h;
    *reinterpret_cast<f32x4*>(u + 4 * plane) = e - h;
    *reinterpret_cast<f32x4*>(u + 5 * plane) = g2;
}

int launch_wino(void (*k)(WinoArgs), WinoArgs a, hipStream_t s) {
    a.rows = a.G * a.B * a.C;
    a.plane = (long)a.B * a.C * a.nt;
    const long threads = (long)a.rows * a.nt;
    if (threads <= 0) return 0;
    k<<<dim3((unsigned)((threads + 255) / 256)), 256, 0, s>>>(a);
    VC_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// x [G][B][C][4 nt] -> V [G][6][B][C][nt]
int launch_wino_input(const float* x, long x_gs, float* V, int G, int B, int C, int nt, hipStream_t s) {
    WinoArgs a{}; a.src = x; a.src_gs = x_gs; a.dst = V; a.dst_gs = 6L * B * C * nt; a.G = G; a.B = B; a.C = C; a.nt = nt;
    return launch_wino(wino_input_kernel, a, s);
}
// M [G][6][B][C][nt] -> y [G][B][C][4 nt] (+ bias)
int launch_wino_output(const float* M, const float* bias, long b_gs, float* y, long y_gs, int G, int B, int C, int nt, hipStream_t s) {
    WinoArgs a{}; a.src = M; a.src_gs = 6L * B * C * nt; a.dst = y; a.dst_gs = y_gs; a.bias = bias; a.bias_gs = b_gs; a.G = G; a.B = B; a.C = C; a.nt = nt;
    return launch_wino(wino_output_kernel, a, s);
}
// dy [G][B][C][4 nt] -> dM [G][6][B][C][nt]
int launch_wino_dy(const float* dy, long dy_gs, float* dM, int G, int B, int C, int nt, hipStream_t s) {
    WinoArgs a{}; a.src = dy; a.src_gs = dy_gs; a.dst = dM; a.dst_gs = 6L * B * C * nt; a.G = G; a.B = B; a.C = C; a.nt = nt;
    return launch_wino(wino_dy_kernel, a, s);
}
// dV [G][6][B][C][nt] -> dx [G][B][C][4 nt] = (accumulate ? dx : 0) + addend + B dV
int launch_wino_dx(const float* dV, float* dx, long dx_gs, const float* addend, long add_gs, int accumulate, int G, int B, int C, int nt, hipStream_t s) {
    WinoArgs a{}; a.src = dV; a.src_gs = 6L * B * C * nt; a.dst = dx; a.dst_gs = dx_gs; a.addend = addend; a.add_gs = add_gs; a.accumulate = accumulate;
    a.G = G; a.B = B; a.C = C; a.nt = nt;
    return launch_wino(wino_dx_kernel, a, s);
}
// filters w [G][3][Cin][Cout] -> U [G][6][Cin][Cout], several layers per launch
int launch_wino_filters(const std::vector<WinoFilter>& f, int G, hipStream_t s) {
    for (size_t k0 = 0; k0 < f.size(); k0 += VC_WINO_JOBS) {
        WinoFilterJobs jj{}; jj.G = G; unsigned blk = 0;
        for (size_t k = k0; k < f.size() && k < k0 + VC_WINO_JOBS; ++k) {
            const long n = (long)f[k].Cin * f[k].Cout;
            if (n % 4 != 0 || n / 4 >= (1L << 30)) { vc_set_error("winograd filter transform: Cin * Cout must be a multiple of 4"); return -1; }
            WinoFilterJob& j = jj.j[jj.n]; j.w = f[k].w; j.U = f[k].U; j.w_gs = f[k].w_gs; j.U_gs = 6 * n; j.n4 = (int)(n / 4);
            jj.blk0[jj.n++] = blk; blk += (unsigned)((j.n4 + 255) / 256);
        }
        jj.blk0[jj.n] = blk;
        wino_filter_kernel<<<dim3(blk, (unsigned)G), 256, 0, s>>>(jj);
        VC_HIP(hipGetLastError());
    }
    return 0;
}
