// Kernels only the TRAINABLE point backbone needs (--unfreeze_pc_encoder, model_arch.py:33-36):
// LayerNorm backward, train-mode BatchNorm1d (batch statistics over all B*G*M rows, running-stat
// update, fused ReLU) forward/backward, group max with arg-max + its scatter backward, small-K weight
// gradients, per-sample residual scaling (DropPath).  reference modules: pointbert/dvae.py:189-221
// (Encoder: Conv1d/BatchNorm1d/ReLU/max), pointbert/point_encoder.py:58-76 (Block with DropPath),
// :142 (final LayerNorm).  HBM-bound; fp32 math, I/O in T.
#include "common.h"
#include <math.h>

// ------------------------------------------------------------------------------------------------
// Fixed-order second stage of every column reduction in this file: out[w] (+)= part[0][w] + part[1][w] + ... + part[P-1][w].
// The first stages write one partial row per block into the caller's scratch (no fp32 atomics anywhere: round 3's kernels met in
// dw / db / the batch statistics / dgamma / dbeta / dW with atomicAdd, whose order — and with it the last bits of the train-mode
// BatchNorm OUTPUT — changed from run to run).
// ------------------------------------------------------------------------------------------------
// One 1024-thread block owns 32 columns: row lane rl (0..31) adds the partial rows p = rl, rl + 32, ... in increasing order, the 32 lane sums meet in LDS
// and are added in lane order — a fixed association whatever the launch (the first form walked all P rows in one thread per column: 112 us for P = 512).
// out[w] (+)= sum over p of part[p * stride + off + w], w < W.
__global__ __launch_bounds__(1024) void ordered_partial_sum_strided_kernel(const float* part, int P, long long stride, long long off, long long W, float* out,
                                                                           int accumulate) {
    __shared__ float sm[32][33];
    const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const long long w = (long long)blockIdx.x * 32 + c;
    float s = 0.f;
    if (w < W)
        for (int p = rl; p < P; p += 32) s += part[(long long)p * stride + off + w];
    sm[rl][c] = s;
    __syncthreads();
    if (rl == 0 && w < W) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += sm[r][c];
        out[w] = accumulate ? out[w] + t : t;
    }
}
static int ordered_partial_sum_cols(const float* part, int P, long long stride, long long off, long long W, float* out, int accumulate, hipStream_t s) {
    EGOMI_LAUNCH(ordered_partial_sum_strided_kernel, dim3((unsigned)((W + 31) / 32)), dim3(1024), 0, s, part, P, stride, off, W, out, accumulate);
    return egomi_launch_status();
}
static int ordered_partial_sum(const float* part, int P, long long W, float* out, int accumulate, hipStream_t s) {
    return ordered_partial_sum_cols(part, P, W, 0, W, out, accumulate, s);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward: dx = rstd*(g - mean(g) - xh*mean(g*xh)) (+ dx_add), g = w*dy;
// dw += sum dy*xh, db += sum dy.  One wave per row; every wave keeps its own [2][cols] partial row in LDS (column c of wave v has ONE
// writer: lane c % 64), the block adds its four rows in wave order into partials[block], ordered_partial_sum_strided_kernel adds the blocks.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* dy, const T* x, const T* w, T* dx, const T* dx_add, float* part,
                                                            int rows, int cols, float eps) {
    extern __shared__ float sm[];                     // [4 waves][2][cols]
    for (int c = threadIdx.x; c < 8 * cols; c += 256) sm[c] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* my = sm + (long long)wv * 2 * cols;
    for (long long row = (long long)blockIdx.x * 4 + wv; row < rows; row += (long long)gridDim.x * 4) {
        const T* xr = x + row * cols;
        const T* gr = dy + row * cols;
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += Cvt<T>::ld(xr + c);
        const float mean = wave_sum(s) / cols;
        float q = 0.f;
        for (int c = lane; c < cols; c += 64) { const float d = Cvt<T>::ld(xr + c) - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / cols + eps);
        float sg = 0.f, sgx = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float xh = (Cvt<T>::ld(xr + c) - mean) * rstd, g = Cvt<T>::ld(gr + c) * Cvt<T>::ld(w + c);
            sg += g; sgx += g * xh;
        }
        sg = wave_sum(sg) / cols; sgx = wave_sum(sgx) / cols;
        for (int c = lane; c < cols; c += 64) {
            const float xh = (Cvt<T>::ld(xr + c) - mean) * rstd, gy = Cvt<T>::ld(gr + c), g = gy * Cvt<T>::ld(w + c);
            float v = rstd * (g - sg - xh * sgx);
            if (dx_add) v += Cvt<T>::ld(dx_add + row * cols + c);
            Cvt<T>::st(dx + row * cols + c, v);
            my[c] += gy * xh;
            my[cols + c] += gy;
        }
    }
    __syncthreads();
    if (part)
        for (int c = threadIdx.x; c < 2 * cols; c += 256)
            part[(long long)blockIdx.x * 2 * cols + c] = ((sm[c] + sm[2 * cols + c]) + sm[4 * cols + c]) + sm[6 * cols + c];
}

extern "C" int egomi_layernorm_bwd(const void* dy, const void* x, const void* w, void* dx, const void* dx_add, float* dw, float* db,
                                   int rows, int cols, float eps, float* partials, int64_t partial_floats, int dtype, egomi_stream_t stream) {
    if (!dy || !x || !w || !dx) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0 || cols > 2048) return EGOMI_E_SHAPE;
    const int grid = (rows + 3) / 4 < 512 ? (rows + 3) / 4 : 512;
    const bool red = dw || db;
    if (red && (!partials || partial_floats < (int64_t)grid * 2 * cols)) return EGOMI_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(layernorm_bwd_kernel<T>, dim3(grid), dim3(256), 8 * cols * sizeof(float), s,
                                             (const T*)dy, (const T*)x, (const T*)w, (T*)dx, (const T*)dx_add, red ? partials : nullptr, rows, cols, eps));
    if (dw && db && db == dw + cols) return ordered_partial_sum(partials, grid, 2ll * cols, dw, 1, s);       // back to back: one pass
    // dw / db anywhere: a partial row is [dw cols | db cols], so each half is a strided view of it
    if (dw) { const int rc = ordered_partial_sum_cols(partials, grid, 2ll * cols, 0, cols, dw, 1, s); if (rc != EGOMI_OK) return rc; }
    if (db) { const int rc = ordered_partial_sum_cols(partials, grid, 2ll * cols, cols, cols, db, 1, s); if (rc != EGOMI_OK) return rc; }
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// column statistics: sum[c] += sum_r x[r,c], sumsq[c] += sum_r x[r,c]^2  (optionally of x*mask-free)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colstats_kernel(const T* x, long long R, int C, float* part /* [gridDim.y][2*C] */, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    long long r1 = r0 + rows_per_block; r1 = r1 < R ? r1 : R;
    float a = 0.f, b = 0.f;
    for (long long r = r0; r < r1; ++r) { const float v = Cvt<T>::ld(x + r * C + c); a += v; b += v * v; }
    part[(long long)blockIdx.y * 2 * C + c] = a;
    part[(long long)blockIdx.y * 2 * C + C + c] = b;
}

// y = relu?((x - mean) * rstd * gamma + beta), mean/rstd from the batch sums (biased variance, nn.BatchNorm1d
// training mode); block (0,0) also updates running_mean / running_var (momentum, unbiased variance).
template <typename T>
__global__ __launch_bounds__(256) void bn_train_apply_kernel(const T* x, long long R, int C, const float* sum, const float* sumsq, const T* gamma,
                                                             const T* beta, float eps, int relu, T* y, float* mean_out, float* rstd_out,
                                                             T* running_mean, T* running_var, float momentum) {
    const long long total = R * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float mean = sum[c] / (float)R;
        const float var = fmaxf(sumsq[c] / (float)R - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        float v = (Cvt<T>::ld(x + e) - mean) * rstd * Cvt<T>::ld(gamma + c) + Cvt<T>::ld(beta + c);
        if (relu) v = fmaxf(v, 0.f);
        Cvt<T>::st(y + e, v);
        if (e < C) {                                   // one thread per channel
            mean_out[c] = mean; rstd_out[c] = rstd;
            if (running_mean) {
                const float unb = var * (float)R / (float)(R > 1 ? R - 1 : 1);
                Cvt<T>::st(running_mean + c, (1.f - momentum) * Cvt<T>::ld(running_mean + c) + momentum * mean);
                Cvt<T>::st(running_var + c, (1.f - momentum) * Cvt<T>::ld(running_var + c) + momentum * unb);
            }
        }
    }
}

extern "C" int egomi_bn_train_fwd(const void* x, int64_t R, int C, const void* gamma, const void* beta, float eps, int relu, void* y,
                                  float* stats /* [4*C]: sum, sumsq (scratch), mean, rstd (saved) */, void* running_mean, void* running_var,
                                  float momentum, float* partials, int64_t partial_floats, int dtype, egomi_stream_t stream) {
    if (!x || !gamma || !beta || !y || !stats || !partials) return EGOMI_E_BADARG;
    if (R <= 0 || C <= 0) return EGOMI_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int rpb = 256;
    const int nby = (int)((R + rpb - 1) / rpb);
    if (partial_floats < (int64_t)nby * 2 * C) return EGOMI_E_BADARG;
    dim3 g1((C + 255) / 256, (unsigned)nby);
    const long long total = R * C;
    const int g2 = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    EGOMI_DISPATCH_DTYPE(dtype, {
        EGOMI_LAUNCH(colstats_kernel<T>, g1, dim3(256), 0, s, (const T*)x, (long long)R, C, partials, rpb);
        const int rc = ordered_partial_sum(partials, nby, 2ll * C, stats, 0, s);                 // sum | sumsq, row blocks added in order
        if (rc != EGOMI_OK) return rc;
        EGOMI_LAUNCH(bn_train_apply_kernel<T>, dim3(g2), dim3(256), 0, s, (const T*)x, (long long)R, C, stats, stats + C, (const T*)gamma, (const T*)beta,
                     eps, relu, (T*)y, stats + 2 * C, stats + 3 * C, (T*)running_mean, (T*)running_var, momentum);
    });
    return egomi_launch_status();
}

// backward of y = relu?(bn(x)):  dyr = dy * (y > 0 if relu);  dgamma = sum dyr*xh, dbeta = sum dyr,
// dx = gamma*rstd/R * (R*dyr - dbeta - xh*dgamma)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* dy, const T* x, const T* y, long long R, int C, const float* mean, const float* rstd,
                                                            int relu, float* part /* [gridDim.y][2*C] */, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    long long r1 = r0 + rows_per_block; r1 = r1 < R ? r1 : R;
    const float m = mean[c], rs = rstd[c];
    float a = 0.f, b = 0.f;
    for (long long r = r0; r < r1; ++r) {
        float g = Cvt<T>::ld(dy + r * C + c);
        if (relu && !(Cvt<T>::ld(y + r * C + c) > 0.f)) g = 0.f;
        a += g * (Cvt<T>::ld(x + r * C + c) - m) * rs;
        b += g;
    }
    part[(long long)blockIdx.y * 2 * C + c] = a;
    part[(long long)blockIdx.y * 2 * C + C + c] = b;
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* dy, const T* x, const T* y, long long R, int C, const float* mean, const float* rstd,
                                                           const T* gamma, int relu, const float* dgamma, const float* dbeta, T* dx) {
    const long long total = R * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        float g = Cvt<T>::ld(dy + e);
        if (relu && !(Cvt<T>::ld(y + e) > 0.f)) g = 0.f;
        const float xh = (Cvt<T>::ld(x + e) - mean[c]) * rstd[c];
        Cvt<T>::st(dx + e, Cvt<T>::ld(gamma + c) * rstd[c] / (float)R * ((float)R * g - dbeta[c] - xh * dgamma[c]));
    }
}

extern "C" int egomi_bn_train_bwd(const void* dy, const void* x, const void* y, int64_t R, int C, const float* stats, const void* gamma, int relu,
                                  float* dgamma /* [C] written by this call */, float* dbeta, void* dx, float* partials, int64_t partial_floats, int dtype,
                                  egomi_stream_t stream) {
    if (!dy || !x || !y || !stats || !gamma || !dgamma || !dbeta || !dx || !partials) return EGOMI_E_BADARG;
    if (R <= 0 || C <= 0) return EGOMI_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int rpb = 256;
    const int nby = (int)((R + rpb - 1) / rpb);
    if (partial_floats < (int64_t)nby * 2 * C) return EGOMI_E_BADARG;
    if (hipMemsetAsync(dgamma, 0, (size_t)C * sizeof(float), s) != hipSuccess || hipMemsetAsync(dbeta, 0, (size_t)C * sizeof(float), s) != hipSuccess)
        return EGOMI_E_LAUNCH;
    dim3 g1((C + 255) / 256, (unsigned)nby);
    const long long total = R * C;
    const int g2 = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    EGOMI_DISPATCH_DTYPE(dtype, {
        EGOMI_LAUNCH(bn_bwd_reduce_kernel<T>, g1, dim3(256), 0, s, (const T*)dy, (const T*)x, (const T*)y, (long long)R, C, stats + 2 * C, stats + 3 * C, relu,
                     partials, rpb);
        { const int rc = ordered_partial_sum_cols(partials, nby, 2ll * C, 0, C, dgamma, 1, s); if (rc != EGOMI_OK) return rc; }
        { const int rc = ordered_partial_sum_cols(partials, nby, 2ll * C, C, C, dbeta, 1, s); if (rc != EGOMI_OK) return rc; }
        EGOMI_LAUNCH(bn_bwd_apply_kernel<T>, dim3(g2), dim3(256), 0, s, (const T*)dy, (const T*)x, (const T*)y, (long long)R, C, stats + 2 * C, stats + 3 * C,
                     (const T*)gamma, relu, dgamma, dbeta, (T*)dx);
    });
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// group max with arg-max (first maximum wins, like torch.max) and its backward
//   fwd: x [BG, M, C] -> out [BG, C], idx i32 [BG, C];   bwd: dx[bg, idx, c] (+)= dout[bg, c]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void group_argmax_kernel(const T* x, int M, int C, T* out, int32_t* idx) {
    const long long g = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float mx = -INFINITY; int am = 0;
        for (int m = 0; m < M; ++m) { const float v = Cvt<T>::ld(x + (g * M + m) * C + c); if (v > mx) { mx = v; am = m; } }
        Cvt<T>::st(out + g * C + c, mx);
        idx[g * C + c] = am;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void group_max_bwd_kernel(const T* dout, const int32_t* idx, int M, int C, T* dx, long long ldx, int accumulate) {
    const long long g = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int am = idx[g * C + c];
        const float d = Cvt<T>::ld(dout + g * C + c);
        for (int m = 0; m < M; ++m) {
            T* p = dx + (g * M + m) * ldx + c;
            const float add = (m == am) ? d : 0.f;
            Cvt<T>::st(p, accumulate ? Cvt<T>::ld(p) + add : add);
        }
    }
}
extern "C" int egomi_group_argmax(const void* x, int BG, int M, int C, void* out, int32_t* idx, int dtype, egomi_stream_t stream) {
    if (!x || !out || !idx) return EGOMI_E_BADARG;
    if (BG <= 0 || M <= 0 || C <= 0) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(group_argmax_kernel<T>, dim3(BG), dim3(256), 0, (hipStream_t)stream, (const T*)x, M, C, (T*)out, idx));
    return egomi_launch_status();
}
extern "C" int egomi_group_max_bwd(const void* dout, const int32_t* idx, int BG, int M, int C, void* dx, int64_t ldx, int accumulate, int dtype,
                                   egomi_stream_t stream) {
    if (!dout || !idx || !dx) return EGOMI_E_BADARG;
    if (BG <= 0 || M <= 0 || C <= 0 || ldx < C) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(group_max_bwd_kernel<T>, dim3(BG), dim3(256), 0, (hipStream_t)stream, (const T*)dout, idx, M, C, (T*)dx,
                                             (long long)ldx, accumulate));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// small-K weight gradient: dW[n,k] += sum_r dy[r,n] * x[r,k]   (K <= 8; x may be fp32)
// ------------------------------------------------------------------------------------------------
template <typename TX, typename T>
__global__ __launch_bounds__(256) void smallk_wgrad_kernel(const T* dy, const TX* x, long long R, int N, int K, float* part /* [gridDim.y][N*K] */, int rows_per_block) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    long long r1 = r0 + rows_per_block; r1 = r1 < R ? r1 : R;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long r = r0; r < r1; ++r) {
        const float g = Cvt<T>::ld(dy + r * N + n);
        for (int k = 0; k < K; ++k) acc[k] += g * Cvt<TX>::ld(x + r * K + k);
    }
    for (int k = 0; k < K; ++k) part[(long long)blockIdx.y * N * K + (long long)n * K + k] = acc[k];
}
extern "C" int egomi_smallk_wgrad(const void* dy, const void* x, int x_dtype, int64_t R, int N, int K, float* dW, float* partials, int64_t partial_floats,
                                  int dtype, egomi_stream_t stream) {
    if (!dy || !x || !dW || !partials) return EGOMI_E_BADARG;
    if (R <= 0 || N <= 0 || K <= 0 || K > 8) return EGOMI_E_SHAPE;
    const int rpb = 512;
    const int nby = (int)((R + rpb - 1) / rpb);
    if (partial_floats < (int64_t)nby * N * K) return EGOMI_E_BADARG;
    dim3 g((N + 255) / 256, (unsigned)nby);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == EGOMI_F32 && dtype == EGOMI_F32) EGOMI_LAUNCH((smallk_wgrad_kernel<float, float>), g, dim3(256), 0, s, (const float*)dy, (const float*)x, (long long)R, N, K, partials, rpb);
    else if (x_dtype == EGOMI_F32 && dtype == EGOMI_BF16) EGOMI_LAUNCH((smallk_wgrad_kernel<float, bf16_t>), g, dim3(256), 0, s, (const bf16_t*)dy, (const float*)x, (long long)R, N, K, partials, rpb);
    else if (x_dtype == EGOMI_BF16 && dtype == EGOMI_BF16) EGOMI_LAUNCH((smallk_wgrad_kernel<bf16_t, bf16_t>), g, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (long long)R, N, K, partials, rpb);
    else return EGOMI_E_BADARG;
    return ordered_partial_sum(partials, nby, (long long)N * K, dW, 1, s);
}

// ------------------------------------------------------------------------------------------------
// DropPath residual: out[r,:] = resid[r,:] + scale[r / rows_per_sample] * branch[r,:]   (timm DropPath,
// point_encoder.py:65,74-75).  Also its backward helper: d_branch = scale * dout.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rowscale_add_kernel(const T* resid, const T* branch, const float* scale, long long rows, int cols, int rps, T* out) {
    const long long total = rows * cols;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / cols;
        const float v = scale[r / rps] * Cvt<T>::ld(branch + e) + (resid ? Cvt<T>::ld(resid + e) : 0.f);
        Cvt<T>::st(out + e, v);
    }
}
extern "C" int egomi_rowscale_add(const void* resid, const void* branch, const float* scale, int64_t rows, int cols, int rows_per_sample, void* out,
                                  int dtype, egomi_stream_t stream) {
    if (!branch || !scale || !out) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0 || rows_per_sample <= 0) return EGOMI_E_SHAPE;
    const long long total = rows * cols;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(rowscale_add_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)resid, (const T*)branch, scale,
                                             (long long)rows, cols, rows_per_sample, (T*)out));
    return egomi_launch_status();
}

// group sum: out[bg, c] = sum_m x[(bg*M + m)*ldx + c]   (backward of the expand() of the group-global feature, dvae.py:217)
template <typename T>
__global__ __launch_bounds__(256) void group_sum_kernel(const T* x, int M, int C, long long ldx, T* out) {
    const long long g = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += Cvt<T>::ld(x + (g * M + m) * ldx + c);
        Cvt<T>::st(out + g * C + c, s);
    }
}
extern "C" int egomi_group_sum(const void* x, int BG, int M, int C, int64_t ldx, void* out, int dtype, egomi_stream_t stream) {
    if (!x || !out) return EGOMI_E_BADARG;
    if (BG <= 0 || M <= 0 || C <= 0 || ldx < C) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(group_sum_kernel<T>, dim3(BG), dim3(256), 0, (hipStream_t)stream, (const T*)x, M, C, (long long)ldx, (T*)out));
    return egomi_launch_status();
}
