// Winograd F(4,3) tile transforms (Lavin & Gray 2016; points 0, +-1, +-2, inf), shared by the stand-alone transform kernels
// (vc_conv_wino.hip) and the instance-norm kernels that apply them on their way in / out (vc_elementwise.hip).
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
// A transform-domain tensor is six planes [p][b][channel][tile]; `plane` = floats between consecutive p.
#pragma once
#include "vc_internal.h"

// V[p] = B^T d with d = (d0, c[0..3], d5): the six inputs x[4i-1 .. 4i+4] of tile i (zero beyond the row)
__device__ __forceinline__ void wino_store_input(const f32x4 c, float d0, float d5, float* v, long plane) {
    const float d1 = c[0], d2 = c[1], d3 = c[2], d4 = c[3];
    const float t1 = d4 - 4.f * d2, t2 = d3 - 4.f * d1, t3 = d4 - d2, t4 = 2.f * (d3 - d1);
    v[0 * plane] = 4.f * d0 - 5.f * d2 + d4;
    v[1 * plane] = t1 + t2;
    v[2 * plane] = t1 - t2;
    v[3 * plane] = t3 + t4;
    v[4 * plane] = t3 - t4;
    v[5 * plane] = 4.f * d1 - 5.f * d3 + d5;
}
// y[4i .. 4i+3] = A^T M + bias
__device__ __forceinline__ f32x4 wino_load_output(const float* m, long plane, float b) {
    const float m0 = m[0], m1 = m[plane], m2 = m[2 * plane], m3 = m[3 * plane], m4 = m[4 * plane], m5 = m[5 * plane];
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    f32x4 y;
    y[0] = m0 + s12 + s34 + b;
    y[1] = d12 + 2.f * d34 + b;
    y[2] = s12 + 4.f * s34 + b;
    y[3] = d12 + 8.f * d34 + m5 + b;
    return y;
}
// dM[p] = A dy[4i .. 4i+3]   (adjoint of the output transform)
__device__ __forceinline__ void wino_store_dy(const f32x4 d, float* m, long plane) {
    const float a = d[0] + d[2], b = d[1] + d[3], c4 = d[0] + 4.f * d[2], e = 2.f * d[1] + 8.f * d[3];
    m[0 * plane] = d[0];
    m[1 * plane] = a + b;
    m[2 * plane] = a - b;
    m[3 * plane] = c4 + e;
    m[4 * plane] = c4 - e;
    m[5 * plane] = d[3];
}
// dx[4i .. 4i+3] = B dV (adjoint of the input transform).  Tiles overlap by two inputs: position 4i also collects tile i-1's last row
// (dV[5]), position 4i+3 tile i+1's first (4 dV[0]); the gradient of the padding frames is dropped.  v points at tile i of plane 0.
__device__ __forceinline__ f32x4 wino_load_dx(const float* v, long plane, bool has_prev, bool has_next) {
    const float v0 = v[0], v1 = v[plane], v2 = v[2 * plane], v3 = v[3 * plane], v4 = v[4 * plane], v5 = v[5 * plane];
    const float prev5 = has_prev ? v[5 * plane - 1] : 0.f, next0 = has_next ? v[1] : 0.f;
    f32x4 dx;
    dx[0] = 4.f * (v2 - v1) + 2.f * (v4 - v3) + 4.f * v5 + prev5;          // column 1 of B^T: (0, -4, 4, -2, 2, 4)
    dx[1] = -5.f * v0 - 4.f * (v1 + v2) - (v3 + v4);                       // column 2: (-5, -4, -4, -1, -1, 0)
    dx[2] = (v1 - v2) + 2.f * (v3 - v4) - 5.f * v5;                        // column 3: (0, 1, -1, 2, -2, -5)
    dx[3] = v0 + (v1 + v2) + (v3 + v4) + 4.f * next0;                      // column 4: (1, 1, 1, 1, 1, 0)
    return dx;
}
