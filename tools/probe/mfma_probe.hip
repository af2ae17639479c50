// Register-only MFMA loops (no memory traffic): sustained rate of the two bf16 MFMA shapes under the chip's power cap.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_probe mfma_probe.hip ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256, 2) void k16(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 3 * i)); }
    f32x4 c[16];
    for (int j = 0; j < 16; ++j) c[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256, 2) void k32(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 3 * i)); }
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 2048;
    for (int rep = 0; rep < 3; ++rep) {
        for (int v = 0; v < 2; ++v) {
            if (v == 0) hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, out, 200); else hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, out, 200);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int l = 0; l < 10; ++l) {
                if (v == 0) hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, out, iters); else hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per launch: grid*4 waves * iters * (v==0: 16 MFMA x 16384 flop ; v==1: 8 MFMA x 32768 flop)
            const double flop = 10.0 * grid * 4.0 * iters * 16.0 * 16384.0;
            printf("%s: %.1f ms, %.1f TFLOP/s\n", v == 0 ? "16x16x32" : "32x32x16", ms, flop / ms / 1e9);
        }
    }
    return 0;
}
