// Generic MFMA GEMM (any layout / shape / batch, bf16 or exact-fp32 inputs, fp32 accumulate) with
// fused epilogue.  This is the correctness-first kernel that every dense op of the path can run
// on; gemm_fast.hip holds the tuned bf16 NT kernel the LLaMA projections use.
//
//   C[M,N] = act(alpha * A·B + bias) + residual (+ C if accumulate)
//   A: [M,K] row-major (a_layout 0) or [K,M] (a_layout 1);  B: [N,K] (b_layout 0, nn.Linear weight)
//   or [K,N] (b_layout 1).  Replaces the aten matmul/linear calls of SURVEY.md §8a rows A6-A12.
//
// Tile 128x128xBK, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16 tiles.
// LDS images are always [row][k] (k contiguous) so fragments are one vector read; transposed
// operands are transposed by the staging writes.  bf16: v_mfma_f32_16x16x32_bf16; f32:
// v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 fma chain: MI355X_MICROARCH.md "Matrix cores").
#include "common.h"

#define GT_BM 128
#define GT_BN 128
#define GT_THREADS 256

template <typename T> struct GemmCfg;
template <> struct GemmCfg<bf16_t> { static constexpr int BK = 64, VEC = 8, LDK = 72; };   // 144-B rows: conflict-free b128
template <> struct GemmCfg<float> { static constexpr int BK = 16, VEC = 4, LDK = 17; };

struct GemmArgs {
    const void* A; const void* B; void* C; const void* bias; const void* residual;
    int M, N, K;
    long long lda, ldb, ldc, ldr;
    int batch_inner;
    long long sA0, sA1, sB0, sB1, sC0, sC1;
    float alpha; int accumulate; int act;
};


// Stage one operand tile (ROWS x BK, logical [row][k]) from global into registers.
// TR=false: memory is [row][k] (k contiguous); TR=true: memory is [k][row] (row contiguous).
template <typename T, bool TR, int ROWS>
struct Stager {
    static constexpr int BK = GemmCfg<T>::BK, VEC = GemmCfg<T>::VEC, LDK = GemmCfg<T>::LDK;
    static constexpr int NV = ROWS * BK / VEC / GT_THREADS;     // vectors per thread
    T v[NV][VEC];

    __device__ __forceinline__ void load(const T* base, long long ld, int row0, int k0, int nrows, int K, bool aligned) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * GT_THREADS;
            int r, k;
            if (!TR) { r = idx / (BK / VEC); k = (idx % (BK / VEC)) * VEC; }
            else { k = idx / (ROWS / VEC); r = (idx % (ROWS / VEC)) * VEC; }
            const int gr = row0 + r, gk = k0 + k;
            const T* p = TR ? base + (long long)gk * ld + gr : base + (long long)gr * ld + gk;
            const bool full = TR ? (gk < K && gr + VEC <= nrows) : (gr < nrows && gk + VEC <= K);
            if (full && aligned) {
                const u32x4 x = *reinterpret_cast<const u32x4*>(p);
                *reinterpret_cast<u32x4*>(&v[i][0]) = x;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const bool ok = TR ? (gk < K && gr + j < nrows) : (gr < nrows && gk + j < K);
                    v[i][j] = ok ? p[j] : (T)0;
                }
            }
        }
    }
    __device__ __forceinline__ void store(T* lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * GT_THREADS;
            if (!TR) {
                const int r = idx / (BK / VEC), k = (idx % (BK / VEC)) * VEC;
                if (sizeof(T) == 2) {
                    *reinterpret_cast<u32x4*>(lds + r * LDK + k) = *reinterpret_cast<const u32x4*>(&v[i][0]);
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) lds[r * LDK + k + j] = v[i][j];
                }
            } else {
                const int k = idx / (ROWS / VEC), r = (idx % (ROWS / VEC)) * VEC;
#pragma unroll
                for (int j = 0; j < VEC; ++j) lds[(r + j) * LDK + k] = v[i][j];
            }
        }
    }
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int KSTEP = 32;
    // lane l: A[row l&15][k 8*(l>>4)+j], B[k 8*(l>>4)+j][col l&15]
    static __device__ __forceinline__ f32x4 run(const bf16_t* a_row, const bf16_t* b_row, int lane, f32x4 acc) {
        const int ko = 8 * (lane >> 4);
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_row + ko);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(b_row + ko);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 4;
    static __device__ __forceinline__ f32x4 run(const float* a_row, const float* b_row, int lane, f32x4 acc) {
        const int ko = lane >> 4;
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a_row[ko], b_row[ko], acc, 0, 0, 0);
    }
};

template <typename T, typename TC, bool TA, bool TB>
__global__ __launch_bounds__(GT_THREADS) void gemm_generic_kernel(GemmArgs g, bool a_aligned, bool b_aligned) {
    constexpr int BK = GemmCfg<T>::BK, LDK = GemmCfg<T>::LDK;
    __shared__ __attribute__((aligned(16))) T sA[GT_BM * LDK];
    __shared__ __attribute__((aligned(16))) T sB[GT_BN * LDK];
    const int z = blockIdx.z;
    const int z0 = z / g.batch_inner, z1 = z % g.batch_inner;
    const T* A = reinterpret_cast<const T*>(g.A) + z0 * g.sA0 + z1 * g.sA1;
    const T* B = reinterpret_cast<const T*>(g.B) + z0 * g.sB0 + z1 * g.sB1;
    TC* C = reinterpret_cast<TC*>(g.C) + z0 * g.sC0 + z1 * g.sC1;
    const TC* R = g.residual ? reinterpret_cast<const TC*>(g.residual) + z0 * g.sC0 + z1 * g.sC1 : nullptr;
    const int m0 = blockIdx.y * GT_BM, n0 = blockIdx.x * GT_BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Stager<T, TA, GT_BM> stA;
    Stager<T, !TB ? false : true, GT_BN> stB;   // b_layout 0 = [N,K] (k contiguous) -> TR=false
    const int nt = (g.K + BK - 1) / BK;
    stA.load(A, g.lda, m0, 0, g.M, g.K, a_aligned);
    stB.load(B, g.ldb, n0, 0, g.N, g.K, b_aligned);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        stA.store(sA);
        stB.store(sB);
        __syncthreads();
        if (t + 1 < nt) {
            stA.load(A, g.lda, m0, (t + 1) * BK, g.M, g.K, a_aligned);
            stB.load(B, g.ldb, n0, (t + 1) * BK, g.N, g.K, b_aligned);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += Mma<T>::KSTEP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const T* ar = sA + (wm + i * 16 + (lane & 15)) * LDK + kk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const T* br = sB + (wn + j * 16 + (lane & 15)) * LDK + kk;
                    acc[i][j] = Mma<T>::run(ar, br, lane, acc[i][j]);
                }
            }
        }
    }
    // epilogue: C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + r
    const T* bias = reinterpret_cast<const T*>(g.bias);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn + j * 16 + (lane & 15);
            if (col >= g.N) continue;
            const float bv = bias ? Cvt<T>::ld(bias + col) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + (lane >> 4) * 4 + r;
                if (row >= g.M) continue;
                float v = acc[i][j][r] * g.alpha + bv;
                v = act_apply(v, g.act);
                if (R) v += Cvt<TC>::ld(R + (long long)row * g.ldr + col);
                TC* cp = C + (long long)row * g.ldc + col;
                if (g.accumulate) v += Cvt<TC>::ld(cp);
                Cvt<TC>::st(cp, v);
            }
        }
    }
}

template <typename T, typename TC>
static int launch_generic(const egomi_gemm_desc* d, hipStream_t s) {
    GemmArgs g;
    g.A = d->A; g.B = d->B; g.C = d->C; g.bias = d->bias; g.residual = d->residual;
    g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.ldr = d->ldr;
    g.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
    g.sA0 = d->sA0; g.sA1 = d->sA1; g.sB0 = d->sB0; g.sB1 = d->sB1; g.sC0 = d->sC0; g.sC1 = d->sC1;
    g.alpha = d->alpha; g.accumulate = d->accumulate; g.act = d->act;
    constexpr int VEC = GemmCfg<T>::VEC;
    auto aligned = [&](const void* p, long long ld, long long s0, long long s1) {
        return ((uintptr_t)p % 16 == 0) && (ld % VEC == 0) && (s0 % VEC == 0) && (s1 % VEC == 0);
    };
    const bool aa = aligned(d->A, d->lda, d->sA0, d->sA1), ba = aligned(d->B, d->ldb, d->sB0, d->sB1);
    dim3 grid((d->N + GT_BN - 1) / GT_BN, (d->M + GT_BM - 1) / GT_BM, d->batch > 0 ? d->batch : 1);
    dim3 block(GT_THREADS);
    const bool TA = d->a_layout == 1, TB = d->b_layout == 1;
    if (!TA && !TB) EGOMI_LAUNCH((gemm_generic_kernel<T, TC, false, false>), grid, block, 0, s, g, aa, ba);
    else if (!TA && TB) EGOMI_LAUNCH((gemm_generic_kernel<T, TC, false, true>), grid, block, 0, s, g, aa, ba);
    else if (TA && !TB) EGOMI_LAUNCH((gemm_generic_kernel<T, TC, true, false>), grid, block, 0, s, g, aa, ba);
    else EGOMI_LAUNCH((gemm_generic_kernel<T, TC, true, true>), grid, block, 0, s, g, aa, ba);
    return egomi_launch_status();
}

int egomi_gemm_fast_try(const egomi_gemm_desc* d, hipStream_t s);   // gemm_fast.hip; returns 1 if not applicable
int egomi_gemm_tn_try(const egomi_gemm_desc* d, hipStream_t s);     // gemm_tn.hip (k-major operands); returns 1 if not applicable
extern thread_local hipEvent_t egomi_time_start_, egomi_time_stop_;  // api.hip

extern "C" int egomi_gemm(const egomi_gemm_desc* d, egomi_stream_t stream) {
    if (!d || !d->A || !d->B || !d->C) return EGOMI_E_BADARG;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return EGOMI_E_SHAPE;
    if (d->a_layout < 0 || d->a_layout > 1 || d->b_layout < 0 || d->b_layout > 1 || d->act < 0 || d->act > 2) return EGOMI_E_BADARG;
    const long long min_lda = d->a_layout == 0 ? d->K : d->M, min_ldb = d->b_layout == 0 ? d->K : d->N;
    if (d->lda < min_lda || d->ldb < min_ldb || d->ldc < d->N || (d->residual && d->ldr < d->N)) return EGOMI_E_SHAPE;
    if (d->batch > 65535) return EGOMI_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (!d->force_generic) {
        const int r = egomi_gemm_fast_try(d, s);
        if (r <= 0) return r;
        const int r2 = egomi_gemm_tn_try(d, s);
        if (r2 <= 0) return r2;
    } else {
        egomi_time_start_ = egomi_time_stop_ = nullptr;               // egomi_gemm_time_next: consumed by this call whatever path it takes
    }
    if (d->epilogue != EGOMI_EPI_NONE) return EGOMI_E_UNSUPPORTED;   // fused epilogues exist in the tuned kernels only: never drop one silently
    if (d->ab_dtype == EGOMI_F32 && d->c_dtype == EGOMI_F32) return launch_generic<float, float>(d, s);
    if (d->ab_dtype == EGOMI_BF16 && d->c_dtype == EGOMI_BF16) return launch_generic<bf16_t, bf16_t>(d, s);
    if (d->ab_dtype == EGOMI_BF16 && d->c_dtype == EGOMI_F32) return launch_generic<bf16_t, float>(d, s);
    return EGOMI_E_UNSUPPORTED;
}
