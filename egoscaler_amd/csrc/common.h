// Shared device/host helpers for libegomi.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/egomi.h"

typedef uint16_t bf16_t;   // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define EGOMI_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&h);
}

// SwiGLU backward of one element (HF LlamaMLP, modeling_llama.py:174-176: act = silu(gate) * up): d = d(act) -> og = d(gate), ou = d(up).
// One non-contracted function shared by swiglu_bwd_kernel, the GEMM epilogue EGOMI_EPI_SWIGLU_BWD and its tail-row combine pass, so that the
// three round identically whatever the surrounding loop shape lets -ffp-contract=fast fuse.
__device__ __forceinline__ void swiglu_bwd_elem(float d, float g, float u, float& og, float& ou) {
#pragma clang fp contract(off)
    const float sg = 1.0f / (1.0f + __expf(-g));
    ou = d * g * sg;
    og = d * u * sg * (1.0f + g * (1.0f - sg));
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Cvt<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 8 consecutive elements -> float[8]  (16 B for bf16, 32 B for f32); p must be 16-B aligned
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(r[i] << 16);
        v[2 * i + 1] = __uint_as_float(r[i] & 0xFFFF0000u);
    }
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
    *reinterpret_cast<u32x4*>(p) = r;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024; `red` is >= 16 floats of LDS; all threads get the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

__device__ __forceinline__ float act_apply(float x, int act) {
    if (act == 1) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));   // exact (erf) GELU
    if (act == 2) return fmaxf(x, 0.0f);
    return x;
}

// hipGetLastError() also reports benign leftovers of the caller's own HIP use (e.g. hipErrorNotReady
// from an event query in torch's allocator), so every launch clears it first and records its own.
extern thread_local int egomi_launch_err_;      // defined in api.hip
extern thread_local int egomi_last_hip_error_;
#define EGOMI_LAUNCH(...)                                              \
    do {                                                               \
        (void)hipGetLastError();                                       \
        hipLaunchKernelGGL(__VA_ARGS__);                               \
        const hipError_t e_ = hipGetLastError();                       \
        if (e_ != hipSuccess) { egomi_launch_err_ = 1; egomi_last_hip_error_ = (int)e_; } \
    } while (0)
static inline int egomi_launch_status() {
    const int e = egomi_launch_err_;
    egomi_launch_err_ = 0;
    return e ? EGOMI_E_LAUNCH : EGOMI_OK;
}

#define EGOMI_DISPATCH_DTYPE(dtype, ...)                         \
    do {                                                         \
        if ((dtype) == EGOMI_F32) { typedef float T; __VA_ARGS__; }        \
        else if ((dtype) == EGOMI_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else return EGOMI_E_BADARG;                              \
    } while (0)
