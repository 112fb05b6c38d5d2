// Device-side reductions shared by the non-GEMM kernels.  Wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

// A kernel's by-value argument struct lives in the kernarg segment (device memory the host has just written: its first read
// misses every cache).  Compiled code fetches the fields as it needs them, a 64-byte line at a time, each behind an
// s_waitcnt -- a 320-byte struct costs five DEPENDENT memory round trips before the first table fetch can even be issued
// (round-4 ISA reading of igemm_kernel's prologue: ~12 scalar-load waits in a row; the round-3 probe put 3.8 us between a
// workgroup's entry and its first operand load).  touch_kernarg reads one dword of every line side by side and makes the
// kernel wait for them once: one round trip, after which every field is a scalar-cache hit.
// The address comes from the kernarg segment's base register pair plus the argument's byte offset (every user passes its kernel's
// first -- only -- argument, offset 0; igemm_kernel the element of its argument array it works on), NOT from the C++ object: a
// by-value argument the optimiser materialised in private memory would turn `&a` into a scratch address.
template <class T> __device__ __forceinline__ void touch_kernarg(const T&, unsigned byte_offset = 0) {
    typedef const __attribute__((address_space(4))) int* cint_p;
    cint_p q = (cint_p)__builtin_amdgcn_kernarg_segment_ptr() + (byte_offset >> 2);
    int x = 0;
#pragma unroll
    for (unsigned o = 0; o < sizeof(T) / 4; o += 16) x |= q[o];
    asm volatile("" :: "s"(x));
}

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
