// Single-token decode against a static KV cache (A13): kv_append, attn_decode, argmax.
// replaces the cached branch of HF LlamaAttention.forward (modeling_llama.py:243-281 with
// past_key_values.update) + eager attention for one query, and the greedy step of
// GenerationMixin.generate reached from model_arch.py:94-108.  HBM-bound: per step each (batch, head)
// streams its K and V rows once (2 * T * head_dim * 2 B); cache layout [B, H, Smax, hd] keeps every
// stream contiguous.  All launches take their lengths as arguments, so 32 steps can be captured into
// one hipGraph with per-step constants.
#include "common.h"
#include <math.h>

// ------------------------------------------------------------------------------------------------
// kv_append: rows [B*S, (3*)H*hd] of the k / v projections -> cache[b, h, pos0 + s, :]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(const T* k, const T* v, long long ld, T* kc, T* vc, int B, int S, int H, int hd,
                                                        int Smax, int pos0) {
    const int cpr = hd / 8;                                      // 8-element chunks per (row, head)
    const long long total = (long long)B * S * H * cpr;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % cpr);
        const int h = (int)((e / cpr) % H);
        const long long row = e / ((long long)cpr * H);
        const int b = (int)(row / S), s = (int)(row % S);
        const long long src = row * ld + (long long)h * hd + c * 8;
        const long long dst = (((long long)b * H + h) * Smax + pos0 + s) * hd + c * 8;
        float x[8];
        load8<T>(k + src, x); store8<T>(kc + dst, x);
        load8<T>(v + src, x); store8<T>(vc + dst, x);
    }
}

extern "C" int egomi_kv_append(const void* k, const void* v, int64_t ld, void* kcache, void* vcache, int B, int S, int H, int hd, int Smax,
                               int pos0, int dtype, egomi_stream_t stream) {
    if (!k || !v || !kcache || !vcache) return EGOMI_E_BADARG;
    if (B <= 0 || S <= 0 || H <= 0 || hd <= 0 || hd % 8 || ld % 8 || ld < (int64_t)H * hd || pos0 < 0 || pos0 + S > Smax) return EGOMI_E_SHAPE;
    const long long total = (long long)B * S * H * (hd / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(kv_append_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)k, (const T*)v,
                                             (long long)ld, (T*)kcache, (T*)vcache, B, S, H, hd, Smax, pos0));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// attn_decode: one query per (b, h) against keys [0, T).  One 256-thread block per (b, h); a wave
// handles 16 keys per iteration (4 lanes per key, HD/4 dims per lane), online softmax per wave,
// waves combined through LDS.  key_mask [B, T] (1 = visible) may be NULL.
// ------------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* q, long long ld_q, const T* kc, const T* vc, const uint8_t* key_mask,
                                                          long long ld_mask, T* out, long long ld_o, int H, int Smax, int Tlen, float scale) {
    constexpr int DPL = HD / 4;                                    // dims per lane
    __shared__ float sm_m[4], sm_l[4];
    __shared__ float sm_acc[4][HD];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane & 3, kslot = lane >> 2;                  // 16 keys per wave-iteration
    float qv[DPL];
#pragma unroll
    for (int c = 0; c < DPL / 8; ++c) {
        float t[8];
        load8<T>(q + (long long)b * ld_q + (long long)h * HD + part * DPL + c * 8, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[c * 8 + j] = t[j] * scale;
    }
    const T* Kb = kc + ((long long)bh * Smax) * HD + part * DPL;
    const T* Vb = vc + ((long long)bh * Smax) * HD + part * DPL;
    float m = -INFINITY, l = 0.f, acc[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) acc[j] = 0.f;
    for (int k0 = wave * 16; k0 < Tlen; k0 += 64) {
        const int key = k0 + kslot;
        bool ok = key < Tlen;
        if (ok && key_mask) ok = key_mask[(long long)b * ld_mask + key] != 0;
        const int kr = key < Tlen ? key : Tlen - 1;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DPL / 8; ++c) {
            float t[8];
            load8<T>(Kb + (long long)kr * HD + c * 8, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += t[j] * qv[c * 8 + j];
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s = ok ? s : -INFINITY;
        float mx = s;
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float m_new = fmaxf(m, mx);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = m == -INFINITY ? 0.f : __expf(m - m_safe);
        const float p = ok ? __expf(s - m_safe) : 0.f;
        float ps = p;
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) ps += __shfl_xor(ps, o, 64);
        l = l * alpha + ps;
        m = m_new;
#pragma unroll
        for (int c = 0; c < DPL / 8; ++c) {
            float t[8];
            load8<T>(Vb + (long long)kr * HD + c * 8, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[c * 8 + j] = acc[c * 8 + j] * alpha + p * t[j];
        }
    }
    // reduce the 16 key slots of the wave (lanes with equal `part`)
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        float a = acc[j];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
        acc[j] = a;
    }
    if (lane < 4) {
#pragma unroll
        for (int j = 0; j < DPL; ++j) sm_acc[wave][lane * DPL + j] = acc[j];
        if (lane == 0) { sm_m[wave] = m; sm_l[wave] = l; }
    }
    __syncthreads();
    if (threadIdx.x < HD) {
        float mm = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
        const float ms = mm == -INFINITY ? 0.f : mm;
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = sm_m[w] == -INFINITY ? 0.f : __expf(sm_m[w] - ms);
            num += f * sm_acc[w][threadIdx.x];
            den += f * sm_l[w];
        }
        Cvt<T>::st(out + (long long)b * ld_o + (long long)h * HD + threadIdx.x, den > 0.f ? num / den : 0.f);
    }
}

extern "C" int egomi_attn_decode(const void* q, int64_t ld_q, const void* kcache, const void* vcache, const uint8_t* key_mask, int64_t ld_mask,
                                 void* out, int64_t ld_o, int B, int H, int hd, int Smax, int T_len, float scale, int dtype,
                                 egomi_stream_t stream) {
    if (!q || !kcache || !vcache || !out) return EGOMI_E_BADARG;
    if (B <= 0 || H <= 0 || T_len <= 0 || T_len > Smax || ld_q % 8 || ld_q < (int64_t)H * hd || ld_o < (int64_t)H * hd) return EGOMI_E_SHAPE;
    if (key_mask && ld_mask < T_len) return EGOMI_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
#define ADK(TT, HDV)                                                                                                       \
    EGOMI_LAUNCH((attn_decode_kernel<TT, HDV>), dim3(B * H), dim3(256), 0, s, (const TT*)q, (long long)ld_q, (const TT*)kcache, \
                 (const TT*)vcache, key_mask, (long long)ld_mask, (TT*)out, (long long)ld_o, H, Smax, T_len, scale)
    if (dtype == EGOMI_BF16) {
        if (hd == 128) ADK(bf16_t, 128); else if (hd == 64) ADK(bf16_t, 64); else if (hd == 32) ADK(bf16_t, 32); else return EGOMI_E_UNSUPPORTED;
    } else if (dtype == EGOMI_F32) {
        if (hd == 128) ADK(float, 128); else if (hd == 64) ADK(float, 64); else if (hd == 32) ADK(float, 32); else return EGOMI_E_UNSUPPORTED;
    } else return EGOMI_E_BADARG;
#undef ADK
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Consumers of EGOMI_EPI_SLABS products (include/egomi.h): the single-token step's split-K projections leave fp32 K-slice
// slabs; these kernels sum them (slice order, like splitk_reduce_kernel) while doing the next operation of the layer, with
// the rounding sequence of the separate kernels they replace (bit-identical results).
//
// slabs_rmsnorm: x = bf16(sum_s slab[s] + residual)   (the o_proj / down_proj output with its residual, HF modeling_llama.py:
//                 243-281), h = rmsnorm(x) * w         (rmsnorm_fwd_kernel's arithmetic).  One 256-thread block per row.
// ------------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void slabs_rmsnorm_kernel(const float* slabs, int sk, long long slab_stride, int cols, const T* residual, long long ldr,
                                                            const T* w, float eps, T* x_out, long long ldx, T* h_out, long long ldh) {
    __shared__ float red[16];
    const long long row = blockIdx.x;
    float xv[MAXV][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (threadIdx.x + i * 256) * 8;
        if (c < cols) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int s2 = 0; s2 < sk; ++s2) {
                float t[8];
                load8<float>(slabs + (long long)s2 * slab_stride + row * cols + c, t);
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
            for (int j = 0; j < 8; ++j) xv[i][j] = sizeof(T) == 2 ? bf2f(f2bf(v[j])) : v[j];      // what the combine pass would have stored
            store8<T>(x_out + row * ldx + c, xv[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += xv[i][j] * xv[i][j];
        }
    }
    const float rstd = rsqrtf(block_sum(ss, red) / cols + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (threadIdx.x + i * 256) * 8;
        if (c < cols) {
            float ww[8], o[8];
            load8<T>(w + c, ww);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float h = xv[i][j] * rstd;
                if (sizeof(T) == 2) h = bf2f(f2bf(h));
                o[j] = ww[j] * h;
            }
            store8<T>(h_out + row * ldh + c, o);
        }
    }
}

extern "C" int egomi_slabs_rmsnorm(const float* slabs, int slices, int rows, int cols, const void* residual, int64_t ldr, const void* w, float eps,
                                   void* x_out, int64_t ldx, void* h_out, int64_t ldh, int dtype, egomi_stream_t stream) {
    if (!slabs || !w || !x_out || !h_out) return EGOMI_E_BADARG;
    if (slices < 1 || rows <= 0 || cols <= 0 || cols % 8 || ldx < cols || ldh < cols || ldx % 8 || ldh % 8 || (residual && (ldr < cols || ldr % 8))) return EGOMI_E_SHAPE;
    if (cols > 8192) return EGOMI_E_UNSUPPORTED;
    if (((uintptr_t)slabs | (uintptr_t)x_out | (uintptr_t)h_out | (uintptr_t)w | (uintptr_t)residual) & 15) return EGOMI_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const long long stride = (long long)rows * cols;
#define SRN(V) EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH((slabs_rmsnorm_kernel<T, V>), dim3(rows), dim3(256), 0, s, slabs, slices, stride, cols, (const T*)residual, \
                                                        (long long)ldr, (const T*)w, eps, (T*)x_out, (long long)ldx, (T*)h_out, (long long)ldh))
    if (cols <= 2048) SRN(1); else if (cols <= 4096) SRN(2); else SRN(4);
#undef SRN
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// qkv_finish (single-token step): q|k|v = bf16(sum_s slab[s]) [B, 3*H*hd]; RoPE at position `pos` on q and k (rope_vec8_kernel's
// arithmetic, HF apply_rotary_pos_emb), q written to qkv (attn_decode reads it there), k and v written straight into the
// [B,H,Smax,hd] caches at `pos`.  Replaces splitk_reduce + rope + kv_append.  One thread per 8 rotation pairs / 16 v columns.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void qkv_finish_kernel(const float* slabs, int sk, long long slab_stride, T* qkv, long long ld, const float* cos_tab,
                                                         const float* sin_tab, int pos, T* kc, T* vc, int B, int H, int hd, int Smax) {
    const int half = hd >> 1, cpv = half >> 3;
    const long long total = (long long)B * 3 * H * cpv;
    const long long d = (long long)H * hd;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e % cpv) * 8;
        const int h = (int)((e / cpv) % H);
        const int part = (int)((e / ((long long)cpv * H)) % 3);
        const long long b = e / ((long long)cpv * H * 3);
        const long long col = part * d + (long long)h * hd + i;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < sk; ++s2) {
            float t[8];
            load8<float>(slabs + (long long)s2 * slab_stride + b * 3 * d + col, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += t[j];
            load8<float>(slabs + (long long)s2 * slab_stride + b * 3 * d + col + half, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) bb[j] += t[j];
        }
        if (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = bf2f(f2bf(a[j])); bb[j] = bf2f(f2bf(bb[j])); }       // the product as the combine pass would have stored it
        }
        float oa[8], ob[8];
        if (part < 2) {
            float c[8], sn[8];
            load8<float>(cos_tab + (long long)pos * half + i, c);
            load8<float>(sin_tab + (long long)pos * half + i, sn);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float cj = c[j], sj = sn[j];
                if (sizeof(T) == 2) {
                    cj = bf2f(f2bf(cj)); sj = bf2f(f2bf(sj));
                    oa[j] = bf2f(f2bf(a[j] * cj)) + bf2f(f2bf(-bb[j] * sj));
                    ob[j] = bf2f(f2bf(bb[j] * cj)) + bf2f(f2bf(a[j] * sj));
                } else {
                    oa[j] = a[j] * cj + (-bb[j]) * sj;
                    ob[j] = bb[j] * cj + a[j] * sj;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { oa[j] = a[j]; ob[j] = bb[j]; }
        }
        if (part == 0) {
            store8<T>(qkv + b * ld + col, oa);
            store8<T>(qkv + b * ld + col + half, ob);
        } else {
            T* dst = (part == 1 ? kc : vc) + ((b * H + h) * Smax + pos) * hd + i;
            store8<T>(dst, oa);
            store8<T>(dst + half, ob);
        }
    }
}

extern "C" int egomi_qkv_finish(const float* slabs, int slices, void* qkv, int64_t ld, const float* cos_tab, const float* sin_tab, int pos,
                                void* kcache, void* vcache, int B, int H, int hd, int Smax, int dtype, egomi_stream_t stream) {
    if (!slabs || !qkv || !cos_tab || !sin_tab || !kcache || !vcache) return EGOMI_E_BADARG;
    if (slices < 1 || B <= 0 || H <= 0 || hd <= 0 || (hd / 2) % 8 || ld % 8 || ld < 3ll * H * hd || pos < 0 || pos >= Smax) return EGOMI_E_SHAPE;
    if (((uintptr_t)slabs | (uintptr_t)qkv | (uintptr_t)kcache | (uintptr_t)vcache | (uintptr_t)cos_tab | (uintptr_t)sin_tab) & 15) return EGOMI_E_SHAPE;
    const long long total = (long long)B * 3 * H * (hd / 16);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(qkv_finish_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs, slices, (long long)B * 3 * H * hd,
                                             (T*)qkv, (long long)ld, cos_tab, sin_tab, pos, (T*)kcache, (T*)vcache, B, H, hd, Smax));
    return egomi_launch_status();
}

// ------------------------------------------------------------------------------------------------
// greedy step: ids[b] = argmax_v logits[b, v] (lowest index on ties, like torch.argmax on CPU/GPU for
// distinct values); also stored at seq[b * ld_seq + pos] when seq != NULL.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const T* logits, long long ld, int V, int64_t* ids, int64_t* seq, long long ld_seq, int pos) {
    __shared__ unsigned long long red[16];
    const int b = blockIdx.x;
    unsigned long long best = 0ull;
    for (int c = threadIdx.x; c < V; c += 1024) {
        const float v = Cvt<T>::ld(logits + (long long)b * ld + c);
        unsigned u = __float_as_uint(v);
        u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;                // order-preserving
        const unsigned long long key = ((unsigned long long)u << 32) | (0xFFFFFFFFu - (unsigned)c);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(best & 0xFFFFFFFFu), o, 64);
        unsigned hi = __shfl_xor((unsigned)(best >> 32), o, 64);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) best = red[w] > best ? red[w] : best;
        const int64_t id = (int64_t)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFu));
        ids[b] = id;
        if (seq) seq[(long long)b * ld_seq + pos] = id;
    }
}

extern "C" int egomi_argmax_rows(const void* logits, int64_t ld, int B, int V, int64_t* ids, int64_t* seq, int64_t ld_seq, int pos, int dtype,
                                 egomi_stream_t stream) {
    if (!logits || !ids) return EGOMI_E_BADARG;
    if (B <= 0 || V <= 0 || ld < V || (seq && (pos < 0 || pos >= ld_seq))) return EGOMI_E_SHAPE;
    EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH(argmax_rows_kernel<T>, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const T*)logits, (long long)ld, V,
                                             ids, seq, (long long)ld_seq, pos));
    return egomi_launch_status();
}
