// LDS-read + MFMA loops with no global traffic: what the main loop of a 256x256 bf16 GEMM tile can sustain under the chip's power cap
// for two wave shapes of the same block tile (random operands resident in LDS, fragments re-read every K-step as a GEMM does):
//   k8w: 8 waves (2 x 4), 128 x 64 per wave  -> 12 ds_read_b128 per 32 v_mfma_f32_16x16x32_bf16  (the 8-phase kernel's shape)
//   k4w: 4 waves (2 x 2), 128 x 128 per wave -> 16 ds_read_b128 per 64 MFMAs (one wave per SIMD, accumulators in the AGPR half)
// Question (round 4): does 1/3 fewer LDS bytes per flop raise the sustained rate (DVFS: MI355X_MICROARCH.md "give-back")?
// Build: hipcc -O3 --offload-arch=gfx950 -o lds_mfma_probe lds_mfma_probe.hip ; run: ./lds_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define BK 64

__device__ __forceinline__ const bf16x8* frag_ptr(const unsigned short* tile, int row, int chunk) {
    return reinterpret_cast<const bf16x8*>(tile + row * BK + ((chunk ^ ((row >> 1) & 7)) << 3));
}

template <int WN, int NB>                                   // waves along N, B fragments per wave
__device__ __forceinline__ void body(const unsigned short* src, float* out, int iters, unsigned short* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 256 * BK; i += blockDim.x) smem[i] = src[i];
    __syncthreads();
    const unsigned short* sA = smem;
    const unsigned short* sB = smem + 256 * BK;
    const int wm = (wave / WN) * 128, wn = (wave % WN) * (16 * NB);
    const int c0 = lane >> 4, rl = lane & 15;
    f32x4 acc[NB][8];
    for (int j = 0; j < NB; ++j) for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][8], fb[2][NB];
    int z = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[0][i] = *frag_ptr(sA, wm + 16 * i + rl, c0);
#pragma unroll
    for (int j = 0; j < NB; ++j) fb[0][j] = *frag_ptr(sB, wn + 16 * j + rl, c0);
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+s"(z));                          // the tile is constant: keep the compiler from hoisting the fragment reads
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int nx = ks ^ 1;                           // fragments of the next K-step are read under this step's MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[nx][i] = *frag_ptr(sA + z, wm + 16 * i + rl, c0 + 4 * nx);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[nx][j] = *frag_ptr(sB + z, wn + 16 * j + rl, c0 + 4 * nx);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[j][i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < NB; ++j) for (int i = 0; i < 8; ++i) s += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(512, 2) void k8w(const unsigned short* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    body<4, 4>(src, out, iters, smem);
}
__global__ __launch_bounds__(256, 1) void k4w(const unsigned short* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    body<2, 8>(src, out, iters, smem);
}

int main() {
    const int n = 2 * 256 * BK;
    unsigned short* h = (unsigned short*)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; std::memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    unsigned short* src; float* out;
    hipMalloc(&src, n * 2); hipMalloc(&out, 256 * 512 * 4 * 4);
    hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k8w), hipFuncAttributeMaxDynamicSharedMemorySize, n * 2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k4w), hipFuncAttributeMaxDynamicSharedMemorySize, n * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, grid = 256;                      // one block per CU
    for (int rep = 0; rep < 4; ++rep)
        for (int v = 0; v < 2; ++v) {
            auto launch = [&](int it) {
                if (v == 0) hipLaunchKernelGGL(k8w, dim3(grid), dim3(512), n * 2, 0, src, out, it);
                else hipLaunchKernelGGL(k4w, dim3(grid), dim3(256), n * 2, 0, src, out, it);
            };
            launch(50);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int l = 0; l < 20; ++l) launch(iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 20.0 * grid * (double)iters * 2.0 * 256.0 * 256.0 * BK;
            printf("%s: %.2f ms, %.1f TFLOP/s\n", v == 0 ? "8 waves x 128x64 " : "4 waves x 128x128", ms, flop / ms / 1e9);
        }
    return 0;
}
