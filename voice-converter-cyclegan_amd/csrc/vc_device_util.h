// Device-side reductions shared by the non-GEMM kernels.  Wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

// cross-lane add inside a row of 16 lanes on the DPP path (no LDS round trip): CTRL = quad_perm / row_mirror encodings
template <int CTRL> __device__ __forceinline__ float dpp_read(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// after this every lane of a row of 16 holds the row's sum
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_read<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_read<0x4E>(v);       // quad_perm [2,3,0,1]   -> quad sums
    v += dpp_read<0x141>(v);      // row_half_mirror       -> sums of 8
    v += dpp_read<0x140>(v);      // row_mirror            -> sum of the row
    return v;
}

// 64-lane butterfly: every lane gets the total
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum_256(float v, float* red /*[4]*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// 64-lane sum on the DPP path (4 cross-lane VALU adds inside each row of 16, then one readlane per row) -- the
// ds_bpermute route of __shfl_xor costs an LDS round trip per step, which dominates when 72 values are reduced
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = row16_sum(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
