// HBM-bound row/elementwise kernels of the path: LayerNorm, RMSNorm fwd/bwd, RoPE, SwiGLU, GELU,
// softmax fwd/bwd (unfused attention), embedding gather + point-token splice fwd/bwd, cross-entropy
// fwd+bwd, AdamW, transpose, cast, add, mini-PointNet group max, small-K linear.
// All compute in fp32; I/O in T (float or bf16).  16-byte vector accesses where the shape allows.
#include "common.h"
#include <math.h>

// ------------------------------------------------------------------------------------------------
// LayerNorm forward (+ fused pre-add).  reference: point_encoder.py:60,63,74-75,95-98,142
// one wave per row; x_sum = x + add is written when sum_out != NULL.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* x, const T* add, const T* w, const T* b, T* sum_out, T* y,
                                                            int rows, int cols, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* xr = x + (long long)row * cols;
    const T* ar = add ? add + (long long)row * cols : nullptr;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        float v = Cvt<T>::ld(xr + c);
        if (ar) v += Cvt<T>::ld(ar + c);
        s += v;
    }
    const float mean = wave_sum(s) / cols;
    float q = 0.f;
    for (int c = lane; c < cols; c += 64) {
        float v = Cvt<T>::ld(xr + c);
        if (ar) v += Cvt<T>::ld(ar + c);
        if (sizeof(T) == 2) v = bf2f(f2bf(v));       // the sum is materialised in T by the reference
        const float dlt = v - mean;
        q += dlt * dlt;
    }
    const float rstd = rsqrtf(wave_sum(q) / cols + eps);
    for (int c = lane; c < cols; c += 64) {
        float v = Cvt<T>::ld(xr + c);
        if (ar) v += Cvt<T>::ld(ar + c);
        if (sum_out) Cvt<T>::st(sum_out + (long long)row * cols + c, v);
        if (sizeof(T) == 2) v = bf2f(f2bf(v));
        Cvt<T>::st(y + (long long)row * cols + c, (v - mean) * rstd * Cvt<T>::ld(w + c) + Cvt<T>::ld(b + c));
    }
}

extern "C" int egomi_layernorm_fwd(const void* x, const void* add, const void* w, const void* b, void* sum_out, void* y,
                                   int rows, int cols, float eps, int dtype, egomi_stream_t stream) {
    if (!x || !w || !b || !y) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(layernorm_fwd_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, (const T*)add, (const T*)w, (const T*)b, (T*)sum_out, (T*)y, rows, cols, eps));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// RMSNorm.  reference: HF modeling_llama.py:53-67 (fp32 variance, x_hat cast back, then * weight)
// one 256-thread block per row, 8-element vectors.  cols % 8 == 0.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* x, const T* w, T* y, float* rstd_out, int cols, float eps) {
    __shared__ float red[16];
    const long long row = blockIdx.x;
    const T* xr = x + row * cols;
    float s = 0.f;
    for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
        float v[8];
        load8<T>(xr + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j] * v[j];
    }
    const float rstd = rsqrtf(block_sum(s, red) / cols + eps);
    if (rstd_out && threadIdx.x == 0) rstd_out[row] = rstd;
    for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
        float v[8], ww[8], o[8];
        load8<T>(xr + c, v);
        load8<T>(w + c, ww);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float h = v[j] * rstd;
            if (sizeof(T) == 2) h = bf2f(f2bf(h));      // .to(input_dtype) before the weight multiply
            o[j] = ww[j] * h;
        }
        store8<T>(y + row * cols + c, o);
    }
}

// rmsnorm_fwd_kernel for an input whose last rows are still K-slice slabs of the producing GEMM (EGOMI_EPI_SLABS on a large
// product, include/egomi.h): rows >= row0 are first formed as x = round(sum_s slab[s] (+ residual)) — what the combine pass would
// have stored — written to x, and normalised from registers; rows < row0 take rmsnorm_fwd_kernel's path.  Same arithmetic and
// summation order as the two separate kernels.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_tail_kernel(T* x, const T* w, T* y, float* rstd_out, int cols, float eps, int row0, const float* slabs,
                                                               int sk, long long slab_stride, const T* residual, long long ldr) {
    __shared__ float red[16];
    const long long row = blockIdx.x;
    float xv[MAXV][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (threadIdx.x + i * 256) * 8;
        if (c < cols) {
            if (row >= row0) {                                           // block-uniform
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int s2 = 0; s2 < sk; ++s2) {
                    float t[8];
                    load8<float>(slabs + (long long)s2 * slab_stride + (row - row0) * cols + c, t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += t[j];
                }
                if (residual) {
                    float r[8];
                    load8<T>(residual + row * ldr + c, r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += r[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[i][j] = sizeof(T) == 2 ? bf2f(f2bf(v[j])) : v[j];
                store8<T>(x + row * cols + c, xv[i]);
            } else {
                load8<T>(x + row * cols + c, xv[i]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += xv[i][j] * xv[i][j];
        }
    }
    const float rstd = rsqrtf(block_sum(ss, red) / cols + eps);
    if (rstd_out && threadIdx.x == 0) rstd_out[row] = rstd;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (threadIdx.x + i * 256) * 8;
        if (c < cols) {
            float ww[8], o[8];
            load8<T>(w + c, ww);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float h = xv[i][j] * rstd;
                if (sizeof(T) == 2) h = bf2f(f2bf(h));      // .to(input_dtype) before the weight multiply
                o[j] = ww[j] * h;
            }
            store8<T>(y + row * cols + c, o);
        }
    }
}

// dx = rstd * (g - x_hat * mean(g * x_hat)),  g = w * dy;  dw[c] += sum_rows dy * x_hat.
// one row per block (grid-stride kept for generality); the weight gradient is rmsnorm_dw_kernel's job.
// Tail form (row0 < rows): dy rows >= row0 are still K-slice slabs of the dgrad GEMM that produced dy (EGOMI_EPI_SLABS): they are
// summed and rounded here — and written back to dy, which the weight-gradient pass reads — instead of in a combine pass.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(T* dy, const T* x, const T* w, const float* rstd, T* dx, const T* dx_add, int rows, int cols,
                                                          int row0, const float* slabs, int sk, long long slab_stride) {
    __shared__ float red[16];
    // MAXV = ceil(cols / 2048) chunks of 8 columns per thread (cols <= 8192)
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float rs = rstd[row];
        float dot = 0.f;
        float gw[MAXV][8], xh[MAXV][8], ad[MAXV][8];         // g*w, x*rstd and the residual gradient stay in registers between the passes
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (threadIdx.x + i * 256) * 8;
            if (c < cols) {
                float g[8], xv[8], ww[8];
                if (row >= row0) {                                       // block-uniform
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] = 0.f;
                    for (int s2 = 0; s2 < sk; ++s2) {
                        float t[8];
                        load8<float>(slabs + (long long)s2 * slab_stride + (row - row0) * cols + c, t);
#pragma unroll
                        for (int j = 0; j < 8; ++j) g[j] += t[j];
                    }
                    if (sizeof(T) == 2) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) g[j] = bf2f(f2bf(g[j]));
                    }
                    store8<T>(dy + row * cols + c, g);
                } else {
                    load8<T>(dy + row * cols + c, g);
                }
                load8<T>(x + row * cols + c, xv);
                load8<T>(w + c, ww);
                if (dx_add) load8<T>(dx_add + row * cols + c, ad[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dot += g[j] * ww[j] * xv[j] * rs;
                    gw[i][j] = g[j] * ww[j];
                    xh[i][j] = xv[j] * rs;
                }
            }
        }
        const float mean = block_sum(dot, red) / cols;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (threadIdx.x + i * 256) * 8;
            if (c < cols) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rs * (gw[i][j] - xh[i][j] * mean) + (dx_add ? ad[i][j] : 0.f);
                store8<T>(dx + row * cols + c, o);
            }
        }
    }
}

// dw[c] += sum_r dy[r,c] * (x[r,c] * rstd[r]) as its own pass, DETERMINISTIC (round 4): ONE block owns a strip of 32
// columns over all the rows (4 lanes x 8 columns, 256 rows in flight per iteration), the 256 row-lane partials meet in LDS and are
// summed in a fixed order, and the owner adds the result to dw with a plain read-modify-write.  No atomics: the same bits every run.
// (Round 1's form split the rows over gridDim.y and met in dw with fp32 atomics: ~20 us at 5536 x 4096, order-dependent last bits.)
template <typename T>
__global__ __launch_bounds__(1024) void rmsnorm_dw_kernel(const T* dy, const T* x, const float* rstd, float* dw, int rows, int cols) {
    __shared__ float red[256][33];
    __shared__ float red2[32][33];
    const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int c = blockIdx.x * 32 + cl * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
#pragma unroll 2
        for (long long r = rl; r < rows; r += 256) {
            float g[8], xv[8];
            load8<T>(dy + r * cols + c, g);
            load8<T>(x + r * cols + c, xv);
            const float rs = rstd[r];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += g[j] * (xv[j] * rs);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cl * 8 + j] = acc[j];
    __syncthreads();
    {
        const int cc = threadIdx.x & 31, g = threadIdx.x >> 5;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[g * 8 + k][cc];
        red2[g][cc] = s;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.f;
#pragma unroll 8
        for (int g = 0; g < 32; ++g) s += red2[g][threadIdx.x];
        const int cc = blockIdx.x * 32 + threadIdx.x;
        if (cc < cols) dw[cc] += s;
    }
}

extern "C" int egomi_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int cols, float eps, int dtype,
                                 egomi_stream_t stream) {
    if (!x || !w || !y) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0 || cols % 8) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(rmsnorm_fwd_kernel<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, (const T*)w, (T*)y, rstd, cols, eps));
    return egomi_launch_status();
}

static int rmsnorm_bwd_impl(void* dy, const void* x, const void* w, const float* rstd, void* dx, const void* dx_add, float* dw, int rows, int cols,
                            int row0, const float* slabs, int sk, int dtype, egomi_stream_t stream) {
    if (!dy || !x || !w || !rstd || !dx) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0 || cols % 8) return EGOMI_E_SHAPE;
    if (cols > 8192) return EGOMI_E_UNSUPPORTED;
    const bool tail = row0 < rows;
    const long long stride = (long long)(rows - row0) * cols;
    // one row per block keeps every CU's memory pipe full; the weight gradient, when wanted, is its own pass (above).  With a
    // tail the row kernel runs FIRST: it is the one that materialises dy's last rows, which the weight-gradient pass reads.
    auto dw_pass = [&]() {
        EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(rmsnorm_dw_kernel<T>, dim3((cols + 31) / 32), dim3(1024), 0, (hipStream_t)stream,
                                                       (const T*)dy, (const T*)x, rstd, dw, rows, cols));
        return 0;
    };
    if (dw && !tail) dw_pass();
    const int grid = rows;
#define RNB(V) EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH((rmsnorm_bwd_kernel<T, V>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (T*)dy, (const T*)x, \
                                                        (const T*)w, rstd, (T*)dx, (const T*)dx_add, rows, cols, row0, slabs, sk, stride))
    if (cols <= 2048) RNB(1); else if (cols <= 4096) RNB(2); else RNB(4);
#undef RNB
    if (dw && tail) dw_pass();
    return egomi_launch_status();
}

extern "C" int egomi_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, const void* dx_add,
                                 float* dw, int rows, int cols, int dtype, egomi_stream_t stream) {
    return rmsnorm_bwd_impl(const_cast<void*>(dy), x, w, rstd, dx, dx_add, dw, rows, cols, rows, nullptr, 0, dtype, stream);
}

// the *_tail forms: the last rows of the input are still the K-slice slabs of the GEMM that produced it (include/egomi.h)
extern "C" int egomi_rmsnorm_bwd_tail(void* dy, const void* x, const void* w, const float* rstd, void* dx, const void* dx_add, float* dw, int rows,
                                      int cols, int row0, const float* slabs, int slices, int dtype, egomi_stream_t stream) {
    if (row0 < 0 || row0 > rows || (row0 < rows && (!slabs || slices < 1 || ((uintptr_t)slabs & 15)))) return EGOMI_E_SHAPE;
    return rmsnorm_bwd_impl(dy, x, w, rstd, dx, dx_add, dw, rows, cols, row0, slabs, slices, dtype, stream);
}

extern "C" int egomi_rmsnorm_fwd_tail(void* x, const void* w, void* y, float* rstd, int rows, int cols, float eps, int row0, const float* slabs, int slices,
                                      const void* residual, int64_t ldr, int dtype, egomi_stream_t stream) {
    if (!x || !w || !y) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0 || cols % 8 || row0 < 0 || row0 > rows) return EGOMI_E_SHAPE;
    if (row0 < rows && (!slabs || slices < 1 || ((uintptr_t)slabs & 15) || (residual && (ldr < cols || ldr % 8)))) return EGOMI_E_SHAPE;
    if (cols > 8192) return EGOMI_E_UNSUPPORTED;
    const long long stride = (long long)(rows - row0) * cols;
#define RNF(V) EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH((rmsnorm_fwd_tail_kernel<T, V>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (T*)x, (const T*)w, (T*)y, \
                                                        rstd, cols, eps, row0, slabs, slices, stride, (const T*)residual, (long long)ldr))
    if (cols <= 2048) RNF(1); else if (cols <= 4096) RNF(2); else RNF(4);
#undef RNF
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// RoPE in place on [rows, H, hd] with row stride ld.  reference: HF modeling_llama.py:112-160
// (half-split rotate_half; cos/sin are fp32 tables cast to the activation dtype, products and the
// sum each rounded in that dtype).  inverse=1 applies the transpose rotation (backward).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* x, const float* cos_tab, const float* sin_tab, long long rows, int S, int pos_offset,
                                                   int H, int hd, long long ld, int inverse) {
    const int half = hd >> 1;
    const long long total = rows * H * half;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e % half);
        const int h = (int)((e / half) % H);
        const long long r = e / ((long long)half * H);
        const int pos = pos_offset + (int)(r % S);
        float c = cos_tab[(long long)pos * half + i], s = sin_tab[(long long)pos * half + i];
        if (inverse) s = -s;
        T* p = x + r * ld + (long long)h * hd + i;
        const float a = Cvt<T>::ld(p), b = Cvt<T>::ld(p + half);
        float oa, ob;
        if (sizeof(T) == 2) {
            c = bf2f(f2bf(c)); s = bf2f(f2bf(s));
            oa = bf2f(f2bf(a * c)) + bf2f(f2bf(-b * s));
            ob = bf2f(f2bf(b * c)) + bf2f(f2bf(a * s));
        } else {
            oa = a * c + (-b) * s;
            ob = b * c + a * s;
        }
        Cvt<T>::st(p, oa);
        Cvt<T>::st(p + half, ob);
    }
}

// 8 rotation pairs per thread: 16-B loads of both halves and 32-B loads of the fp32 tables
template <typename T>
__global__ __launch_bounds__(256) void rope_vec8_kernel(T* x, const float* cos_tab, const float* sin_tab, long long rows, int S, int pos_offset,
                                                        int H, int hd, long long ld, int inverse) {
    const int half = hd >> 1, cpv = half >> 3;                      // 8-wide chunks per half
    const long long total = rows * H * cpv;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e % cpv) * 8;
        const int h = (int)((e / cpv) % H);
        const long long r = e / ((long long)cpv * H);
        const int pos = pos_offset + (int)(r % S);
        float c[8], s[8], a[8], b[8], oa[8], ob[8];
        load8<float>(cos_tab + (long long)pos * half + i, c);
        load8<float>(sin_tab + (long long)pos * half + i, s);
        T* p = x + r * ld + (long long)h * hd + i;
        load8<T>(p, a);
        load8<T>(p + half, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float cj = c[j], sj = inverse ? -s[j] : s[j];
            if (sizeof(T) == 2) {
                cj = bf2f(f2bf(cj)); sj = bf2f(f2bf(sj));
                oa[j] = bf2f(f2bf(a[j] * cj)) + bf2f(f2bf(-b[j] * sj));
                ob[j] = bf2f(f2bf(b[j] * cj)) + bf2f(f2bf(a[j] * sj));
            } else {
                oa[j] = a[j] * cj + (-b[j]) * sj;
                ob[j] = b[j] * cj + a[j] * sj;
            }
        }
        store8<T>(p, oa);
        store8<T>(p + half, ob);
    }
}

// rope_vec8_kernel for a q|k|v array [rows, 3*H*hd] whose rows >= row0 are still K-slice slabs of the qkv product (EGOMI_EPI_SLABS
// on a large product): those rows are summed in slice order and rounded first (what the combine pass would have stored); q and k
// get the rotation, v is only materialised.  Rows < row0: the plain in-place rotation of q and k.  Same arithmetic as
// splitk_reduce_kernel followed by rope_vec8_kernel.
template <typename T>
__global__ __launch_bounds__(256) void rope_qkv_tail_kernel(T* x, const float* cos_tab, const float* sin_tab, long long rows, int S, int pos_offset, int H, int hd,
                                                            long long ld, int row0, const float* slabs, int sk, long long slab_stride) {
    const int half = hd >> 1, cpv = half >> 3;
    const long long d = (long long)H * hd, ncols = 3 * d;
    const long long total = rows * 3 * H * cpv;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e % cpv) * 8;
        const int h = (int)((e / cpv) % H);
        const int part = (int)((e / ((long long)cpv * H)) % 3);
        const long long r = e / ((long long)cpv * H * 3);
        const bool tail = r >= row0;
        if (part == 2 && !tail) continue;
        const long long col = part * d + (long long)h * hd + i;
        T* p = x + r * ld + col;
        float a[8], b[8], oa[8], ob[8];
        if (tail) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = 0.f; b[j] = 0.f; }
            for (int s2 = 0; s2 < sk; ++s2) {
                float t[8];
                load8<float>(slabs + (long long)s2 * slab_stride + (r - row0) * ncols + col, t);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += t[j];
                load8<float>(slabs + (long long)s2 * slab_stride + (r - row0) * ncols + col + half, t);
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] += t[j];
            }
            if (sizeof(T) == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { a[j] = bf2f(f2bf(a[j])); b[j] = bf2f(f2bf(b[j])); }
            }
        } else {
            load8<T>(p, a);
            load8<T>(p + half, b);
        }
        if (part < 2) {
            const int pos = pos_offset + (int)(r % S);
            float c[8], sn[8];
            load8<float>(cos_tab + (long long)pos * half + i, c);
            load8<float>(sin_tab + (long long)pos * half + i, sn);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float cj = c[j], sj = sn[j];
                if (sizeof(T) == 2) {
                    cj = bf2f(f2bf(cj)); sj = bf2f(f2bf(sj));
                    oa[j] = bf2f(f2bf(a[j] * cj)) + bf2f(f2bf(-b[j] * sj));
                    ob[j] = bf2f(f2bf(b[j] * cj)) + bf2f(f2bf(a[j] * sj));
                } else {
                    oa[j] = a[j] * cj + (-b[j]) * sj;
                    ob[j] = b[j] * cj + a[j] * sj;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { oa[j] = a[j]; ob[j] = b[j]; }
        }
        store8<T>(p, oa);
        store8<T>(p + half, ob);
    }
}

extern "C" int egomi_rope_qkv_tail(void* qkv, const float* cos_tab, const float* sin_tab, int64_t rows, int S, int pos_offset, int H, int hd, int64_t ld,
                                   int row0, const float* slabs, int slices, int dtype, egomi_stream_t stream) {
    if (!qkv || !cos_tab || !sin_tab) return EGOMI_E_BADARG;
    if (rows <= 0 || S <= 0 || H <= 0 || hd <= 0 || (hd / 2) % 8 || ld % 8 || ld < 3ll * H * hd || pos_offset < 0 || row0 < 0 || row0 > rows) return EGOMI_E_SHAPE;
    if (row0 < rows && (!slabs || slices < 1 || ((uintptr_t)slabs & 15))) return EGOMI_E_SHAPE;
    if ((uintptr_t)qkv & 15) return EGOMI_E_SHAPE;
    const long long tv = rows * 3 * H * (hd / 16);
    const int gv = (int)((tv + 255) / 256 < 16384 ? (tv + 255) / 256 : 16384);
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(rope_qkv_tail_kernel<T>, dim3(gv), dim3(256), 0, (hipStream_t)stream, (T*)qkv, cos_tab, sin_tab, (long long)rows, S,
                                             pos_offset, H, hd, (long long)ld, row0, slabs, slices, (long long)(rows - row0) * 3 * H * hd));
    return egomi_launch_status();
}

extern "C" int egomi_rope(void* x, const float* cos_tab, const float* sin_tab, int64_t rows, int S, int pos_offset, int H, int hd,
                          int64_t ld, int inverse, int dtype, egomi_stream_t stream) {
    if (!x || !cos_tab || !sin_tab) return EGOMI_E_BADARG;
    if (rows <= 0 || S <= 0 || H <= 0 || hd <= 0 || (hd & 1) || ld < (int64_t)H * hd || pos_offset < 0) return EGOMI_E_SHAPE;
    const int esz = dtype == EGOMI_F32 ? 4 : 2;
    if ((hd / 2) % 8 == 0 && ld % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((hd / 2) * esz) % 16 == 0) {
        const long long tv = rows * H * (hd / 16);
        const int gv = (int)((tv + 255) / 256 < 16384 ? (tv + 255) / 256 : 16384);
        EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(rope_vec8_kernel<T>, dim3(gv), dim3(256), 0, (hipStream_t)stream,
                                                 (T*)x, cos_tab, sin_tab, rows, S, pos_offset, H, hd, ld, inverse));
        return egomi_launch_status();
    }
    const long long total = rows * H * (hd / 2);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(rope_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   (T*)x, cos_tab, sin_tab, rows, S, pos_offset, H, hd, ld, inverse));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// SwiGLU.  reference: HF modeling_llama.py:174-176  down(silu(gate(x)) * up(x))
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// il = 1: gate and up are the two halves of every 64-column group of ONE [rows, 2*cols] array ("interleaved-32": hidden
// unit c lives at column 64*(c/32) + c%32, its up value 32 columns further) — the layout the stacked [Wgate;Wup] weight of
// the frozen layers produces, chosen so that one GEMM wave holds gate and up of the same units (fused epilogue, gemm_fast.hip)
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* gate, const T* up, T* out, long long rows, int cols, long long ld_in, long long ld_out, int il) {
    const int cv = cols / 8;
    const long long total = rows * cv;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / cv;
        const int c = (int)(e % cv) * 8;
        const int ci = il ? ((c >> 5) << 6) + (c & 31) : c;
        float g[8], u[8], o[8];
        load8<T>(gate + r * ld_in + ci, g);
        load8<T>(up + r * ld_in + ci, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = g[j] * sigmoidf_(g[j]);
            if (sizeof(T) == 2) a = bf2f(f2bf(a));
            o[j] = a * u[j];
        }
        store8<T>(out + r * ld_out + c, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* dact, const T* gate, const T* up, T* dgate, T* dup, long long rows, int cols,
                                                         long long ld_in, long long ld_act, long long ld_out, int il) {
    const int cv = cols / 8;
    const long long total = rows * cv;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / cv;
        const int c = (int)(e % cv) * 8;
        const int ci = il ? ((c >> 5) << 6) + (c & 31) : c;
        float d[8], g[8], u[8], og[8], ou[8];
        load8<T>(dact + r * ld_act + c, d);
        load8<T>(gate + r * ld_in + ci, g);
        load8<T>(up + r * ld_in + ci, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) swiglu_bwd_elem(d[j], g[j], u[j], og[j], ou[j]);
        store8<T>(dgate + r * ld_out + ci, og);
        store8<T>(dup + r * ld_out + ci, ou);
    }
}

static inline int ew_grid(long long total) { long long g = (total + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

extern "C" int egomi_swiglu_fwd(const void* gate, const void* up, void* out, int64_t rows, int cols, int64_t ld_in, int64_t ld_out,
                                int dtype, egomi_stream_t stream) {
    if (!gate || !up || !out) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0 || cols % 8 || ld_in % 8 || ld_out % 8 || ld_in < cols || ld_out < cols) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(swiglu_fwd_kernel<T>, dim3(ew_grid(rows * (cols / 8))), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)gate, (const T*)up, (T*)out, rows, cols, ld_in, ld_out, 0));
    return egomi_launch_status();
}

extern "C" int egomi_swiglu_bwd(const void* dact, const void* gate, const void* up, void* dgate, void* dup, int64_t rows, int cols,
                                int64_t ld_in, int64_t ld_act, int64_t ld_out, int dtype, egomi_stream_t stream) {
    if (!dact || !gate || !up || !dgate || !dup) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0 || cols % 8 || ld_in % 8 || ld_act % 8 || ld_out % 8) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(swiglu_bwd_kernel<T>, dim3(ew_grid(rows * (cols / 8))), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)dact, (const T*)gate, (const T*)up, (T*)dgate, (T*)dup, rows, cols, ld_in, ld_act, ld_out, 0));
    return egomi_launch_status();
}

// interleaved-32 forms: gu / dgu are [rows, 2*cols] with hidden unit c at column 64*(c/32) + c%32 (gate) and +32 (up)
extern "C" int egomi_swiglu_il_fwd(const void* gu, void* out, int64_t rows, int cols, int64_t ld_gu, int64_t ld_out, int dtype, egomi_stream_t stream) {
    if (!gu || !out) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0 || cols % 32 || ld_gu % 8 || ld_out % 8 || ld_gu < 2 * cols || ld_out < cols) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(swiglu_fwd_kernel<T>, dim3(ew_grid(rows * (cols / 8))), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)gu, (const T*)gu + 32, (T*)out, rows, cols, ld_gu, ld_out, 1));
    return egomi_launch_status();
}
extern "C" int egomi_swiglu_il_bwd(const void* dact, const void* gu, void* dgu, int64_t rows, int cols, int64_t ld_gu, int64_t ld_act, int64_t ld_dgu,
                                   int dtype, egomi_stream_t stream) {
    if (!dact || !gu || !dgu) return EGOMI_E_BADARG;
    if (rows <= 0 || cols <= 0 || cols % 32 || ld_gu % 8 || ld_act % 8 || ld_dgu % 8 || ld_gu < 2 * cols || ld_dgu < 2 * cols) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(swiglu_bwd_kernel<T>, dim3(ew_grid(rows * (cols / 8))), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)dact, (const T*)gu, (const T*)gu + 32, (T*)dgu, (T*)dgu + 32, rows, cols, ld_gu, ld_act, ld_dgu, 1));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// GELU (erf) forward / backward on flat arrays.  reference: nn.GELU() in point_proj, pointllm.py:72-76
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_kernel(const T* x, const T* dy, T* out, long long n) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const float v = Cvt<T>::ld(x + e);
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
        if (dy) {
            const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
            Cvt<T>::st(out + e, Cvt<T>::ld(dy + e) * (cdf + v * pdf));
        } else {
            Cvt<T>::st(out + e, v * cdf);
        }
    }
}
extern "C" int egomi_gelu_fwd(const void* x, void* y, int64_t n, int dtype, egomi_stream_t stream) {
    if (!x || !y) return EGOMI_E_BADARG;
    if (n <= 0) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(gelu_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)nullptr, (T*)y, n));
    return egomi_launch_status();
}
extern "C" int egomi_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, egomi_stream_t stream) {
    if (!x || !dy || !dx) return EGOMI_E_BADARG;
    if (n <= 0) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(gelu_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, n));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Row softmax over fp32 scores (unfused attention path).  reference: point_encoder.py:48-50 and
// HF modeling_llama.py:204-210 (scores + causal/padding mask, softmax in fp32, cast).
// one wave per row, Sk <= 2048.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* sc, const uint8_t* key_mask, long long rows, int heads, int Sq, int Sk,
                                                          long long ld_s, int causal, int q_offset, T* P, long long ld_p) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int qi = (int)(row % Sq);
    const long long z = row / Sq;
    const int b = (int)(z / heads);
    const float* s = sc + row * ld_s;
    const uint8_t* km = key_mask ? key_mask + (long long)b * Sk : nullptr;
    const int lim = causal ? (q_offset + qi) : (Sk - 1);      // keys j <= lim are visible
    constexpr int MAXE = 32;
    float v[MAXE];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int j = lane + i * 64;
        float x = -INFINITY;
        if (j < Sk && j <= lim && (!km || km[j])) x = s[j];
        v[i] = x;
        mx = fmaxf(mx, x);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const float e = (v[i] == -INFINITY) ? 0.f : __expf(v[i] - mx);
        v[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int j = lane + i * 64;
        if (j < Sk) Cvt<T>::st(P + row * ld_p + j, v[i] * inv);
    }
}

// dS = P * (dP - sum_j P*dP)
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* P, long long ld_p, const float* dP, long long ld_dp, T* dS, long long ld_ds,
                                                          long long rows, int Sk) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float dot = 0.f;
    for (int j = lane; j < Sk; j += 64) dot += Cvt<T>::ld(P + row * ld_p + j) * dP[row * ld_dp + j];
    dot = wave_sum(dot);
    for (int j = lane; j < Sk; j += 64) {
        const float p = Cvt<T>::ld(P + row * ld_p + j);
        Cvt<T>::st(dS + row * ld_ds + j, p * (dP[row * ld_dp + j] - dot));
    }
}

extern "C" int egomi_softmax_fwd(const float* scores, int64_t ld_s, const uint8_t* key_mask, int Z, int heads, int Sq, int Sk, int causal,
                                 int q_offset, void* P, int64_t ld_p, int dtype, egomi_stream_t stream) {
    if (!scores || !P) return EGOMI_E_BADARG;
    if (Z <= 0 || heads <= 0 || Sq <= 0 || Sk <= 0 || ld_s < Sk || ld_p < Sk) return EGOMI_E_SHAPE;
    if (Sk > 2048) return EGOMI_E_UNSUPPORTED;
    const long long rows = (long long)Z * Sq;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(softmax_fwd_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                                                   scores, key_mask, rows, heads, Sq, Sk, ld_s, causal, q_offset, (T*)P, ld_p));
    return egomi_launch_status();
}

extern "C" int egomi_softmax_bwd(const void* P, int64_t ld_p, const float* dP, int64_t ld_dp, void* dS, int64_t ld_ds, int64_t rows, int Sk,
                                 int dtype, egomi_stream_t stream) {
    if (!P || !dP || !dS) return EGOMI_E_BADARG;
    if (rows <= 0 || Sk <= 0) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(softmax_bwd_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)P, ld_p, dP, ld_dp, (T*)dS, ld_ds, rows, Sk));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// A10  point-token splice.  reference: model/pointllm.py:131-171 (mm_use_point_start_end = True)
// splice_scan: integer position logic (one block per sample), then one thread walks the batch like the reference's running
// `cur_point_idx` (splice_clouds_kernel).  err: 0 ok, 1 start/end count mismatch (:146), 2 an end token not at start+P+1 (:150),
// 4 the sample's cloud index is past the clouds given (:143 raises IndexError).
// Several segments in one sample, AS THE REFERENCE TREATS THEM: `cur_point_features` is fetched once per sample (:143) and every pass of
// the `for point_start_token_pos` loop rebuilds the row from the ORIGINAL embeddings (:155), so only the LAST segment is spliced — with the
// cloud the sample started at — while cur_point_idx advances once per segment (:156) and shifts the clouds of the following samples.
// start_pos = that last <point_start> (-1: text-only sample, :137-142, which advances the cloud index by one); cloud_idx[b] = index of the
// cloud whose features the sample receives.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splice_scan_kernel(const int64_t* ids, int S, int64_t patch_id, int64_t start_id, int64_t end_id, int P,
                                                          int32_t* start_pos, int32_t* err, int32_t* nseg) {
    const int b = blockIdx.x;
    const int64_t* r = ids + (long long)b * S;
    __shared__ int n_patch, n_start, n_end, last_start, bad_end;
    if (threadIdx.x == 0) { n_patch = 0; n_start = 0; n_end = 0; last_start = -1; bad_end = 0; }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += 256) {
        const int64_t t = r[s];
        if (t == patch_id) atomicAdd(&n_patch, 1);
        if (t == start_id) {
            atomicAdd(&n_start, 1); atomicMax(&last_start, s);
            if (s + P + 1 >= S || r[s + P + 1] != end_id) atomicOr(&bad_end, 1);
        }
        if (t == end_id) atomicAdd(&n_end, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int e = 0, sp = -1;
        if (n_patch > 0) {
            if (n_start != n_end) e = 1;
            else if (bad_end) e = 2;
            else if (n_start >= 1) sp = last_start;
        }
        start_pos[b] = e ? -1 : sp;
        err[b] = e;
        nseg[b] = n_patch > 0 ? n_start : 1;
    }
}
__global__ void splice_clouds_kernel(const int32_t* nseg, int B, int n_clouds, int32_t* start_pos, int32_t* err, int32_t* cloud_idx) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int cur = 0;
    for (int b = 0; b < B; ++b) {
        const bool multimodal = !(nseg[b] == 1 && start_pos[b] < 0 && err[b] == 0);       // text-only samples never index the clouds
        if (multimodal && cur >= n_clouds) { err[b] = 4; start_pos[b] = -1; }              // :143 comes before the token checks of the sample
        cloud_idx[b] = cur < n_clouds ? cur : 0;
        cur += nseg[b];
    }
}

extern "C" int egomi_splice_scan(const int64_t* ids, int B, int S, int64_t patch_id, int64_t start_id, int64_t end_id, int P, int n_clouds,
                                 int32_t* start_pos, int32_t* err, int32_t* cloud_idx, int32_t* scratch, egomi_stream_t stream) {
    if (!ids || !start_pos || !err || !cloud_idx || !scratch) return EGOMI_E_BADARG;
    if (B <= 0 || S <= 0 || P <= 0 || n_clouds < 0) return EGOMI_E_SHAPE;
    EGOMI_LAUNCH(splice_scan_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ids, S, patch_id, start_id, end_id, P, start_pos, err, scratch);
    EGOMI_LAUNCH(splice_clouds_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int32_t*)scratch, B, n_clouds, start_pos, err, cloud_idx);
    return egomi_launch_status();
}

// out[b,s,:] = feats[b, s-start-1, :] if start < s <= start+P else W[ids[b,s], :]   (pointllm.py:107,155)
template <typename T>
__global__ __launch_bounds__(256) void embed_splice_fwd_kernel(const int64_t* ids, const T* W, const T* feats, const int32_t* start_pos,
                                                               const int32_t* cloud_idx, int S, int d, int P, int V, T* out) {
    const long long row = blockIdx.x;
    const int b = (int)(row / S), s = (int)(row % S);
    const int sp = (feats && start_pos) ? start_pos[b] : -1;
    const T* src;
    if (sp >= 0 && s > sp && s <= sp + P) src = feats + ((long long)(cloud_idx ? cloud_idx[b] : b) * P + (s - sp - 1)) * d;
    else {
        long long t = ids[row];
        if (t < 0 || t >= V) t = 0;                         // guarded on the host too
        src = W + t * d;
    }
    for (int c = threadIdx.x * 8; c < d; c += 256 * 8) {
        float v[8];
        load8<T>(src + c, v);
        store8<T>(out + row * d + c, v);
    }
}

// dW[ids] += dout (rows outside the point span); dfeats = dout rows inside the span.  DETERMINISTIC (round 4): a token id that occurs in
// several rows (pad, <tsep>, repeated bins) used to meet in dW with fp32 atomics.  Now the block of the FIRST row that carries an id OWNS
// it: it walks the later rows in increasing order, sums the rows with the same id in registers and adds the result to dW with a plain
// read-modify-write; blocks of later occurrences find an earlier one and leave.  The id list (B*S int64, L2-resident) is scanned 256 rows
// at a time; blockIdx.y splits the d columns into pieces of 1024 so that an id with many rows is summed by several blocks.
__device__ __forceinline__ bool esb_in_span(const int32_t* start_pos, long long row, int S, int P) {
    if (!start_pos) return false;
    const int b = (int)(row / S), s = (int)(row % S);
    const int sp = start_pos[b];
    return sp >= 0 && s > sp && s <= sp + P;
}
template <typename T>
__global__ __launch_bounds__(256) void embed_splice_bwd_kernel(const T* dout, const int64_t* ids, const int32_t* start_pos, const int32_t* cloud_idx,
                                                               int S, int d, int P, int V, long long n_rows, float* dW, T* dfeats) {
    __shared__ unsigned long long hit[4];
    const long long row = blockIdx.x;
    const int b = (int)(row / S), s = (int)(row % S);
    const int sp = start_pos ? start_pos[b] : -1;
    const bool in_span = sp >= 0 && s > sp && s <= sp + P;
    if (in_span) {
        if (!dfeats || blockIdx.y) return;
        T* dst = dfeats + ((long long)(cloud_idx ? cloud_idx[b] : b) * P + (s - sp - 1)) * d;
        for (int c = threadIdx.x * 8; c < d; c += 256 * 8) {
            float v[8];
            load8<T>(dout + row * d + c, v);
            store8<T>(dst + c, v);
        }
        return;
    }
    if (!dW) return;
    const long long t = ids[row];
    if (t < 0 || t >= V) return;
    // an earlier row outside the span with the same id?  then that row's block owns the sum
    int earlier = 0;
    for (long long r = threadIdx.x; r < row; r += 256) earlier |= (ids[r] == t && !esb_in_span(start_pos, r, S, P));
    if (__syncthreads_or(earlier)) return;
    const int c0 = blockIdx.y * 1024 + threadIdx.x * 4;                 // this thread's 4 columns of the piece
    const bool live = c0 < d;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto add_row = [&](long long r) {
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += Cvt<T>::ld(dout + r * d + c0 + j);
        }
    };
    add_row(row);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long long base = row + 1; base < n_rows; base += 256) {
        const long long r = base + threadIdx.x;
        const bool m = r < n_rows && ids[r] == t && !esb_in_span(start_pos, r, S, P);
        const unsigned long long bal = __ballot(m);
        __syncthreads();                                                // the previous chunk's masks have been read by everybody
        if (lane == 0) hit[wv] = bal;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned long long mk = hit[w];
            while (mk) {                                                // increasing row order: the same sum every run
                const int j = __builtin_ctzll(mk);
                mk &= mk - 1;
                add_row(base + 64 * w + j);
            }
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dW[t * d + c0 + j] += acc[j];
    }
}

extern "C" int egomi_embed_splice_fwd(const int64_t* ids, const void* W, const void* feats, const int32_t* start_pos, const int32_t* cloud_idx,
                                      int B, int S, int d, int P, int V, void* out, int dtype, egomi_stream_t stream) {
    if (!ids || !W || !out) return EGOMI_E_BADARG;
    if (B <= 0 || S <= 0 || d <= 0 || d % 8 || V <= 0) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(embed_splice_fwd_kernel<T>, dim3(B * S), dim3(256), 0, (hipStream_t)stream,
                                                   ids, (const T*)W, (const T*)feats, start_pos, cloud_idx, S, d, P, V, (T*)out));
    return egomi_launch_status();
}

extern "C" int egomi_embed_splice_bwd(const void* dout, const int64_t* ids, const int32_t* start_pos, const int32_t* cloud_idx, int B, int S, int d,
                                      int P, int V, float* dW, void* dfeats, int dtype, egomi_stream_t stream) {
    if (!dout || !ids) return EGOMI_E_BADARG;
    if (B <= 0 || S <= 0 || d <= 0 || d % 8 || V <= 0) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(embed_splice_bwd_kernel<T>, dim3(B * S, dW ? (d + 1023) / 1024 : 1), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)dout, ids, start_pos, cloud_idx, S, d, P, V, (long long)B * S, dW, (T*)dfeats));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// A12  cross-entropy over the trajectory span.  reference: train.py:174-181
// F.cross_entropy(logits, targets, ignore_index=pad): mean over non-ignored rows.
// ce_count counts them; ce_fwd_bwd writes -log p[target] of every row into row_loss (0 for ignored rows), a one-block pass adds them to
// loss_sum in a fixed order (round 4: the per-row fp32 atomics made the loss differ in its last bits from run to run) and
// dlogits = (softmax - onehot) * grad_scale / count  (zero rows for ignored targets).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_count_kernel(const int64_t* tg, long long n, int64_t ignore, int32_t* count) {
    int c = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) c += (tg[i] != ignore);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

template <typename T>
__global__ __launch_bounds__(1024) void ce_fwd_bwd_kernel(const T* logits, long long ld, const int64_t* tg, int V, int64_t ignore,
                                                          const int32_t* count, float* row_loss, T* dlogits, long long ldd, float grad_scale) {
    __shared__ float red[16];
    const long long row = blockIdx.x;
    const T* lr = logits + row * ld;
    const int64_t t = tg[row];
    const bool ign = (t == ignore) || t < 0 || t >= V;
    if (ign) {
        if (threadIdx.x == 0) row_loss[row] = 0.f;
        if (dlogits) for (int c = threadIdx.x; c < V; c += 1024) Cvt<T>::st(dlogits + row * ldd + c, 0.f);
        return;
    }
    // in-place use (dlogits == logits): the target logit is read before the first barrier, i.e. before any thread
    // of the block can have reached the gradient stores below
    const float tl = Cvt<T>::ld(lr + t);
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 1024) mx = fmaxf(mx, Cvt<T>::ld(lr + c));
    mx = block_max(mx, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 1024) s += __expf(Cvt<T>::ld(lr + c) - mx);
    s = block_sum(s, red);
    const float lse = mx + __logf(s);
    if (threadIdx.x == 0) row_loss[row] = lse - tl;
    if (dlogits) {
        const float sc = grad_scale / (float)(*count);
        for (int c = threadIdx.x; c < V; c += 1024) {
            float p = __expf(Cvt<T>::ld(lr + c) - lse);
            if (c == t) p -= 1.0f;
            Cvt<T>::st(dlogits + row * ldd + c, p * sc);
        }
    }
}

// loss_sum += sum_i v[i]: one block, strided partials, fixed tree
__global__ __launch_bounds__(1024) void ordered_sum_kernel(const float* v, int n, float* out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) s += v[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out += s;
}

extern "C" int egomi_ce_count(const int64_t* targets, int64_t n, int64_t ignore, int32_t* count, egomi_stream_t stream) {
    if (!targets || !count) return EGOMI_E_BADARG;
    if (n <= 0) return EGOMI_E_SHAPE;
    EGOMI_LAUNCH(ce_count_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, targets, n, ignore, count);
    return egomi_launch_status();
}

extern "C" int egomi_ce_fwd_bwd(const void* logits, int64_t ld, const int64_t* targets, int R, int V, int64_t ignore, const int32_t* count,
                                float* loss_sum, float* row_loss, void* dlogits, int64_t ldd, float grad_scale, int dtype, egomi_stream_t stream) {
    if (!logits || !targets || !count || !loss_sum || !row_loss) return EGOMI_E_BADARG;
    if (R <= 0 || V <= 0 || ld < V || (dlogits && ldd < V)) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(ce_fwd_bwd_kernel<T>, dim3(R), dim3(1024), 0, (hipStream_t)stream,
                                                   (const T*)logits, ld, targets, V, ignore, count, row_loss, (T*)dlogits, ldd, grad_scale));
    EGOMI_LAUNCH(ordered_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)row_loss, R, loss_sum);
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW semantics, reference optimizer: train.py:107-111).  fp32 master/moments;
// optional low-precision model copy written in the same pass.
// ------------------------------------------------------------------------------------------------
template <typename G> __device__ __forceinline__ float adamw_g(const G* g, long long i);
template <> __device__ __forceinline__ float adamw_g<float>(const float* g, long long i) { return g[i]; }
template <> __device__ __forceinline__ float adamw_g<bf16_t>(const bf16_t* g, long long i) { return bf2f(g[i]); }

// one element's update.  Contraction is switched off so that the scalar and the 16-B kernel (and every alignment of the same tensor) round
// identically: with the default -ffp-contract=fast the two loop shapes were fused into FMAs differently (1-ulp differences in master)
__device__ __forceinline__ void adamw_elem(float& p, float& m, float& v, float g, float lr, float b1, float b2, float eps, float wd, float bc1,
                                           float bc2_sqrt, float gscale) {
#pragma clang fp contract(off)
    const float gr = g * gscale;
    float pi = p * (1.0f - lr * wd);
    const float mi = b1 * m + (1.0f - b1) * gr;
    const float vi = b2 * v + ((1.0f - b2) * gr) * gr;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p = pi; m = mi; v = vi;
}

template <typename T, typename G>
__global__ __launch_bounds__(256) void adamw_kernel(float* p, T* p_model, const G* g, float* m, float* v, long long n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float pi = p[i], mi = m[i], vi = v[i];
        adamw_elem(pi, mi, vi, adamw_g<G>(g, i), lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (p_model) Cvt<T>::st(p_model + i, pi);
    }
}

// 16-B form: four elements per lane and access (master, m, v read and written, the gradient read, the model copy written: 30 B per
// element at bf16 — the pass is HBM-bound).  Same arithmetic per element.  G = bf16: the gradient is read straight from the rank-summed
// bf16 wire buffer of dp.GradSync (28 B per element, and no widening pass before it).
template <typename T, typename G>
__global__ __launch_bounds__(256) void adamw_vec4_kernel(float* p, T* p_model, const G* g, float* m, float* v, long long n4, float lr, float b1,
                                                         float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 g4;
        if constexpr (sizeof(G) == 4) g4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        else {
            const u32x2 r = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(g) + i);
            g4[0] = __uint_as_float(r[0] << 16); g4[1] = __uint_as_float(r[0] & 0xFFFF0000u);
            g4[2] = __uint_as_float(r[1] << 16); g4[3] = __uint_as_float(r[1] & 0xFFFF0000u);
        }
        f32x4 p4 = reinterpret_cast<f32x4*>(p)[i], m4 = reinterpret_cast<f32x4*>(m)[i], v4 = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pi = p4[j], mi = m4[j], vi = v4[j];
            adamw_elem(pi, mi, vi, g4[j], lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
            p4[j] = pi; m4[j] = mi; v4[j] = vi;
        }
        reinterpret_cast<f32x4*>(p)[i] = p4; reinterpret_cast<f32x4*>(m)[i] = m4; reinterpret_cast<f32x4*>(v)[i] = v4;
        if (p_model) {
            T o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) Cvt<T>::st(o + j, p4[j]);
            typedef T tx4 __attribute__((ext_vector_type(4)));
            reinterpret_cast<tx4*>(p_model)[i] = tx4{o[0], o[1], o[2], o[3]};            // one 8-B store (bf16)
        }
    }
}

template <typename G>
static int adamw_launch(float* master, void* model_copy, const G* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, int copy_dtype, hipStream_t stream) {
    if (!master || !grad || !m || !v) return EGOMI_E_BADARG;
    if (n <= 0 || step <= 0) return EGOMI_E_SHAPE;
    if (model_copy && copy_dtype != EGOMI_F32 && copy_dtype != EGOMI_BF16) return EGOMI_E_BADARG;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    const long long n4 = n / 4, rest = n - 4 * n4;
    const bool vec = n4 > 0 && (((uintptr_t)master | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)grad & (4 * sizeof(G) - 1)) == 0 &&
                     (!model_copy || (copy_dtype == EGOMI_BF16 && ((uintptr_t)model_copy & 7) == 0));
    if (vec) {
        EGOMI_LAUNCH((adamw_vec4_kernel<bf16_t, G>), dim3(ew_grid(n4)), dim3(256), 0, stream, master, (bf16_t*)model_copy, grad, m, v,
                           n4, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
        if (rest == 0) return egomi_launch_status();
        master += 4 * n4; grad += 4 * n4; m += 4 * n4; v += 4 * n4; n = rest;
        if (model_copy) model_copy = (bf16_t*)model_copy + 4 * n4;
    }
    if (!model_copy || copy_dtype == EGOMI_F32)
        EGOMI_LAUNCH((adamw_kernel<float, G>), dim3(ew_grid(n)), dim3(256), 0, stream, master, (float*)model_copy, grad, m, v,
                           (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    else
        EGOMI_LAUNCH((adamw_kernel<bf16_t, G>), dim3(ew_grid(n)), dim3(256), 0, stream, master, (bf16_t*)model_copy, grad, m, v,
                           (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    return egomi_launch_status();
}

extern "C" int egomi_adamw(float* master, void* model_copy, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int step, float grad_scale, int copy_dtype, egomi_stream_t stream) {
    return adamw_launch<float>(master, model_copy, grad, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, copy_dtype, (hipStream_t)stream);
}
extern "C" int egomi_adamw_g16(float* master, void* model_copy, const void* grad_bf16, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, float grad_scale, int copy_dtype, egomi_stream_t stream) {
    return adamw_launch<bf16_t>(master, model_copy, (const bf16_t*)grad_bf16, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, copy_dtype,
                                (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// transpose [R,C] (ldi) -> [C, ldo] with columns R..ldo-1 zero-filled (32x32 LDS tiles); cast; add
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* in, int R, int C, long long ldi, T* out, long long ldo) {
    __shared__ T tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(long long)r * ldi + c] : (T)0;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < ldo) out[(long long)c * ldo + r] = tile[tx][i];
    }
}

// bf16 fast path: 64x64 tiles, 16-B global loads and stores on both sides, 2-B gathers only inside LDS
__global__ __launch_bounds__(256) void transpose_bf16_vec_kernel(const bf16_t* in, int R, int C, long long ldi, bf16_t* out, long long ldo) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (t >> 3) + 32 * i, cc = (t & 7) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < R && c0 + cc < C) v = *reinterpret_cast<const u32x4*>(in + (long long)(r0 + r) * ldi + c0 + cc);   // C % 8 == 0
        *reinterpret_cast<u32x4*>(&tile[r][cc]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oc = (t >> 3) + 32 * i, rc = (t & 7) * 8;          // output row = input column c0+oc
        if (c0 + oc >= C || r0 + rc >= ldo) continue;
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (uint32_t)tile[rc + 2 * j][oc] | ((uint32_t)tile[rc + 2 * j + 1][oc] << 16);   // rows >= R were zero-filled
        *reinterpret_cast<u32x4*>(out + (long long)(c0 + oc) * ldo + r0 + rc) = v;                                     // ldo % 8 == 0
    }
}

extern "C" int egomi_transpose(const void* in, int R, int C, int64_t ldi, void* out, int64_t ldo, int dtype, egomi_stream_t stream) {
    if (!in || !out) return EGOMI_E_BADARG;
    if (R <= 0 || C <= 0 || ldi < C || ldo < R) return EGOMI_E_SHAPE;
    if (dtype == EGOMI_BF16 && C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0) {
        dim3 g((C + 63) / 64, (unsigned)((ldo + 63) / 64));
        EGOMI_LAUNCH(transpose_bf16_vec_kernel, g, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, R, C, (long long)ldi, (bf16_t*)out, (long long)ldo);
        return egomi_launch_status();
    }
    dim3 grid((C + 31) / 32, (unsigned)((ldo + 31) / 32));
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(transpose_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)in, R, C, ldi, (T*)out, ldo));
    return egomi_launch_status();
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* in, TO* out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) Cvt<TO>::st(out + i, Cvt<TI>::ld(in + i));
}
// 8 elements per thread (16-B / 32-B accesses): the gradient wire casts of dp.GradSync move GBs per step
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast8_kernel(const TI* in, TO* out, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float v[8];
        load8<TI>(in + i * 8, v);
        store8<TO>(out + i * 8, v);
    }
}
extern "C" int egomi_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, egomi_stream_t stream) {
    if (!in || !out) return EGOMI_E_BADARG;
    if (n <= 0) return EGOMI_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype != out_dtype && n % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
        const dim3 g8(ew_grid(n / 8)), b8(256);
        if (in_dtype == EGOMI_F32 && out_dtype == EGOMI_BF16) { EGOMI_LAUNCH((cast8_kernel<float, bf16_t>), g8, b8, 0, s, (const float*)in, (bf16_t*)out, (long long)(n / 8)); return egomi_launch_status(); }
        if (in_dtype == EGOMI_BF16 && out_dtype == EGOMI_F32) { EGOMI_LAUNCH((cast8_kernel<bf16_t, float>), g8, b8, 0, s, (const bf16_t*)in, (float*)out, (long long)(n / 8)); return egomi_launch_status(); }
    }
    const dim3 g(ew_grid(n)), b(256);
    if (in_dtype == EGOMI_F32 && out_dtype == EGOMI_BF16) EGOMI_LAUNCH((cast_kernel<float, bf16_t>), g, b, 0, s, (const float*)in, (bf16_t*)out, (long long)n);
    else if (in_dtype == EGOMI_BF16 && out_dtype == EGOMI_F32) EGOMI_LAUNCH((cast_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)in, (float*)out, (long long)n);
    else if (in_dtype == EGOMI_F32 && out_dtype == EGOMI_F32) EGOMI_LAUNCH((cast_kernel<float, float>), g, b, 0, s, (const float*)in, (float*)out, (long long)n);
    else if (in_dtype == EGOMI_BF16 && out_dtype == EGOMI_BF16) EGOMI_LAUNCH((cast_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)in, (bf16_t*)out, (long long)n);
    else return EGOMI_E_BADARG;
    return egomi_launch_status();
}

// dp.GradSync, direct reduce-scatter: in [W, c] (the chunk every rank sent to this one, rank-major) -> out [c] = sum over
// ranks, accumulated in fp32 in rank order 0..W-1 (the same order on every rank).  c % 8 == 0, 16-B aligned.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void rank_sum_kernel(const TI* in, int W, long long c8, TO* out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < c8; i += (long long)gridDim.x * 256) {
        float acc[8], v[8];
        load8<TI>(in + i * 8, acc);
        for (int w = 1; w < W; ++w) {
            load8<TI>(in + ((long long)w * c8 + i) * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
        store8<TO>(out + i * 8, acc);
    }
}
extern "C" int egomi_rank_sum(const void* in, int in_dtype, int W, int64_t c, void* out, int out_dtype, egomi_stream_t stream) {
    if (!in || !out) return EGOMI_E_BADARG;
    if (W <= 0 || c <= 0 || c % 8) return EGOMI_E_SHAPE;
    if ((((uintptr_t)in | (uintptr_t)out) & 15) != 0) return EGOMI_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(ew_grid(c / 8)), b(256);
    const long long c8 = c / 8;
    if (in_dtype == EGOMI_BF16 && out_dtype == EGOMI_BF16) EGOMI_LAUNCH((rank_sum_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)in, W, c8, (bf16_t*)out);
    else if (in_dtype == EGOMI_BF16 && out_dtype == EGOMI_F32) EGOMI_LAUNCH((rank_sum_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)in, W, c8, (float*)out);
    else if (in_dtype == EGOMI_F32 && out_dtype == EGOMI_F32) EGOMI_LAUNCH((rank_sum_kernel<float, float>), g, b, 0, s, (const float*)in, W, c8, (float*)out);
    else return EGOMI_E_BADARG;
    return egomi_launch_status();
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* a, const T* b, T* out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        Cvt<T>::st(out + i, Cvt<T>::ld(a + i) + Cvt<T>::ld(b + i));
}
extern "C" int egomi_add(const void* a, const void* b, void* out, int64_t n, int dtype, egomi_stream_t stream) {
    if (!a || !b || !out) return EGOMI_E_BADARG;
    if (n <= 0) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(add_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, (T*)out, (long long)n));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// A6 helpers (mini-PointNet, reference: pointbert/dvae.py:207-221)
// group_max: [BG, M, C] -> [BG, C];  concat=1: out [BG*M, 2C] = [max over the group | x]  (:216-217)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void group_max_kernel(const T* x, int M, int C, T* out, int concat) {
    const long long g = blockIdx.x;
    const T* xg = x + g * M * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float mx = -INFINITY;
        for (int m = 0; m < M; ++m) mx = fmaxf(mx, Cvt<T>::ld(xg + (long long)m * C + c));
        if (!concat) {
            Cvt<T>::st(out + g * C + c, mx);
        } else {
            for (int m = 0; m < M; ++m) {
                T* o = out + (g * M + m) * 2 * C;
                Cvt<T>::st(o + c, mx);
                o[C + c] = xg[(long long)m * C + c];
            }
        }
    }
}
extern "C" int egomi_group_max(const void* x, int BG, int M, int C, void* out, int concat, int dtype, egomi_stream_t stream) {
    if (!x || !out) return EGOMI_E_BADARG;
    if (BG <= 0 || M <= 0 || C <= 0) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(group_max_kernel<T>, dim3(BG), dim3(256), 0, (hipStream_t)stream, (const T*)x, M, C, (T*)out, concat));
    return egomi_launch_status();
}

// y[r,n] = act(sum_k x[r,k] * w[n,k] + b[n]) for K <= 8 (pos_embed.0: K=3, point_encoder.py:128;
// first 1x1 conv: K=6, dvae.py:194).  x may be fp32 while y/w/b are T.
template <typename TX, typename T>
__global__ __launch_bounds__(256) void linear_smallk_kernel(const TX* x, const T* w, const T* b, T* y, long long R, int N, int K, int act) {
    const long long total = R * N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / N;
        const int n = (int)(e % N);
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(Cvt<TX>::ld(x + r * K + k), Cvt<T>::ld(w + (long long)n * K + k), acc);
        acc += b ? Cvt<T>::ld(b + n) : 0.f;
        Cvt<T>::st(y + e, act_apply(acc, act));
    }
}

extern "C" int egomi_linear_smallk(const void* x, int x_dtype, const void* w, const void* b, void* y, int64_t R, int N, int K, int act,
                                   int dtype, egomi_stream_t stream) {
    if (!x || !w || !y) return EGOMI_E_BADARG;
    if (R <= 0 || N <= 0 || K <= 0 || K > 8 || act < 0 || act > 2) return EGOMI_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(ew_grid(R * N)), bl(256);
    if (x_dtype == EGOMI_F32 && dtype == EGOMI_F32)
        EGOMI_LAUNCH((linear_smallk_kernel<float, float>), g, bl, 0, s, (const float*)x, (const float*)w, (const float*)b, (float*)y, (long long)R, N, K, act);
    else if (x_dtype == EGOMI_F32 && dtype == EGOMI_BF16)
        EGOMI_LAUNCH((linear_smallk_kernel<float, bf16_t>), g, bl, 0, s, (const float*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, (long long)R, N, K, act);
    else if (x_dtype == EGOMI_BF16 && dtype == EGOMI_BF16)
        EGOMI_LAUNCH((linear_smallk_kernel<bf16_t, bf16_t>), g, bl, 0, s, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, (long long)R, N, K, act);
    else return EGOMI_E_BADARG;
    return egomi_launch_status();
}

// column sums: out[c] += sum_r x[r, c]   (bias gradients of the projector, pointllm.py:67-81 backward).  DETERMINISTIC (round 4): one block
// owns 16 columns over all rows (64 row-lanes: lane rl adds rows rl, rl + 64, ... in increasing order), the 64 lane sums meet in LDS and are added in
// lane order; the row blocks of the earlier form met in `out` with fp32 atomics.  (16-column strips: the bias gradients of the trainable point backbone
// are [4104, 384 .. 1536] — 32-column strips gave 12 .. 48 blocks and 44 us per call.)
template <typename T>
__global__ __launch_bounds__(1024) void colsum_kernel(const T* x, long long R, int C, long long ld, float* out) {
    __shared__ float red[64][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float acc = 0.f;
    if (c < C) {
#pragma unroll 4
        for (long long r = rl; r < R; r += 64) acc += Cvt<T>::ld(x + r * ld + c);
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
        float s = 0.f;
#pragma unroll 8
        for (int g = 0; g < 64; ++g) s += red[g][threadIdx.x];
        out[c] += s;
    }
}
extern "C" int egomi_colsum(const void* x, int64_t R, int C, int64_t ld, float* out, int dtype, egomi_stream_t stream) {
    if (!x || !out) return EGOMI_E_BADARG;
    if (R <= 0 || C <= 0 || ld < C) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(colsum_kernel<T>, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, (const T*)x, (long long)R, C, (long long)ld, out));
    return egomi_launch_status();
}
