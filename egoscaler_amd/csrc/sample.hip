// One decoding step's token choice for a whole batch, on the device (A13): logits -> processed scores -> next token.
//
// replaces what HF GenerationMixin.generate does between two forward passes when the reference calls it with its defaults
// (models/pointllm/model_arch.py:82-108: do_sample=True, top_k=50, top_p=0.95, temperature=1.0, repetition_penalty,
// output_scores=True): RepetitionPenaltyLogitsProcessor -> TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper
// (transformers/generation/logits_process.py:306-366,238-300,542-580,473-540), torch.multinomial on the softmax, the
// eos / pad bookkeeping of unfinished sequences, and the append to input_ids.  `scores` are the PROCESSED scores, as HF
// returns them (fp32, -inf where a warper removed the token).
//
// One 1024-thread workgroup per row; the row lives in registers (V <= 32768) or is re-read from the fp32 scores row (L2).
//   top-k : the k-th largest value by a 32-step radix select on the order-preserving integer image of the floats
//           (count(key >= candidate) per step); everything below it is removed, ties with it are kept (HF: scores < kth).
//   top-p : HF sorts ascending (stable), takes softmax + cumsum and removes the maximal prefix whose cumulative probability
//           is <= 1 - top_p.  With e = exp(x - max): "prefix mass <= thr" is monotone in the (value, index) order, so the same
//           radix walk over the 64-bit key (value image << 32 | index) finds the first kept element without sorting: 32 value
//           steps + ceil(log2 V) index steps, each a block-wide fp32 sum.  Ties at the boundary value are removed lowest index
//           first, exactly what the stable ascending sort does.  The largest element is never removed (min_tokens_to_keep = 1).
//   sample: Gumbel-max — argmax_i (score_i + g_i), g_i = -log(-log u_i), u_i from Philox4x32-10 keyed by (seed; row, i/4,
//           draw counter): an exact draw from softmax(scores) with no scan and no host RNG; seed and counter live in device
//           memory so a captured hipGraph draws fresh numbers at every replay.  do_sample = 0: plain arg-max (lowest index).
// No length is read from the host: `pos` is a launch constant, like every other decode entry point.
#include "common.h"
#include <math.h>

#define SMP_THREADS 1024
#define SMP_NPT 32                       // values per thread held in registers

__device__ __forceinline__ unsigned f2key(float v) {
    unsigned u = __float_as_uint(v);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);           // order-preserving: a < b  <=>  key(a) < key(b)
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float gumbel_from_bits(unsigned r) {
    const float u = ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);          // (0, 1) strictly
    return -logf(-logf(u));
}

__device__ __forceinline__ int block_sum_int(int v, int* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int i = 0; i < SMP_THREADS / 64; ++i) t += red[i];
    return t;
}

struct SampleArgs {
    const void* logits; long long ld; int B, V;
    float* scores; long long ld_scores;
    int64_t* seq; long long ld_seq; int pos, rep_from;
    int64_t* ids; int* done;
    float rep_penalty, temperature; int top_k; float top_p; int do_sample;
    const unsigned long long* rng;        // [0] seed, [1] draw counter base (device memory)
    int draw;                             // added to the counter base: one value per captured step
    long long eos, pad;
};

template <typename T, bool REG>
__global__ __launch_bounds__(SMP_THREADS) void sample_rows_kernel(SampleArgs a) {
    __shared__ float redf[16];
    __shared__ int redi[16];
    __shared__ unsigned long long red64[16];
    const int b = blockIdx.x, tid = threadIdx.x, V = a.V;
    const T* lg = (const T*)a.logits + (long long)b * a.ld;
    float* sc = a.scores + (long long)b * a.ld_scores;
    const int niter = REG ? SMP_NPT : (V + SMP_THREADS - 1) / SMP_THREADS;
    float x[REG ? SMP_NPT : 1];
    const float NEG_INF = -INFINITY;

    // ---- repetition penalty (on the raw logits, tokens seq[b, rep_from : pos]) and temperature.  Element c = i*1024 + tid.
    if (a.rep_penalty != 1.0f && a.seq) {
        for (int c = tid; c < V; c += SMP_THREADS) sc[c] = Cvt<T>::ld(lg + c);
        __syncthreads();
        const int64_t* sq = a.seq + (long long)b * a.ld_seq;
        for (int j = a.rep_from + tid; j < a.pos; j += SMP_THREADS) {
            const long long t = sq[j];
            if (t >= 0 && t < V) {
                const float s = Cvt<T>::ld(lg + t);                            // from the ORIGINAL logit: a token seen twice is penalised once
                sc[t] = s < 0.f ? s * a.rep_penalty : s / a.rep_penalty;
            }
        }
        __syncthreads();
        _Pragma("unroll") for (int i = 0; i < niter; ++i) {
            const int c = i * SMP_THREADS + tid;
            float v = c < V ? sc[c] : NEG_INF;
            if (a.temperature != 1.0f && c < V) v = v / a.temperature;
            if (REG) x[i] = v; else if (c < V) sc[c] = v;
        }
    } else {
        _Pragma("unroll") for (int i = 0; i < niter; ++i) {
            const int c = i * SMP_THREADS + tid;
            float v = c < V ? Cvt<T>::ld(lg + c) : NEG_INF;
            if (a.temperature != 1.0f && c < V) v = v / a.temperature;
            if (REG) x[i] = v; else if (c < V) sc[c] = v;
        }
    }
    if (!REG) __syncthreads();
#define SMP_X(i, c) (REG ? x[i] : ((c) < V ? sc[c] : NEG_INF))

    // ---- top-k: k-th largest key by radix select; remove key < kth
    if (a.top_k > 0 && a.top_k < V) {
        unsigned prefix = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = prefix | (1u << bit);
            int n = 0;
            _Pragma("unroll") for (int i = 0; i < niter; ++i) { const int c = i * SMP_THREADS + tid; n += (c < V && f2key(SMP_X(i, c)) >= cand) ? 1 : 0; }
            if (block_sum_int(n, redi) >= a.top_k) prefix = cand;
        }
        _Pragma("unroll") for (int i = 0; i < niter; ++i) {
            const int c = i * SMP_THREADS + tid;
            if (c < V && f2key(SMP_X(i, c)) < prefix) { if (REG) x[i] = NEG_INF; else sc[c] = NEG_INF; }
        }
        if (!REG) __syncthreads();
    }

    // ---- the largest element in (value, index) order: arg-max with the HIGHEST index on ties = last of the stable ascending sort
    unsigned long long top = 0ull;
    _Pragma("unroll") for (int i = 0; i < niter; ++i) {
        const int c = i * SMP_THREADS + tid;
        if (c < V) { const unsigned long long k = ((unsigned long long)f2key(SMP_X(i, c)) << 32) | (unsigned)c; top = k > top ? k : top; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(top & 0xFFFFFFFFu), o, 64), hi = __shfl_xor((unsigned)(top >> 32), o, 64);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        top = other > top ? other : top;
    }
    __syncthreads();
    if ((tid & 63) == 0) red64[tid >> 6] = top;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) top = red64[w] > top ? red64[w] : top;

    // ---- top-p
    if (a.top_p < 1.0f) {
        // max as a float from its key
        unsigned mk = (unsigned)(top >> 32);
        mk ^= (mk >> 31) ? 0x80000000u : 0xFFFFFFFFu;
        const float m = __uint_as_float(mk);
        float zl = 0.f;
        float e[REG ? SMP_NPT : 1];                                // exp(x - max), 0 for removed / padding lanes (registers when the row is)
        _Pragma("unroll") for (int i = 0; i < niter; ++i) {
            const int c = i * SMP_THREADS + tid;
            const float v = SMP_X(i, c);
            const float ev = (c < V && v != NEG_INF) ? expf(v - m) : 0.f;
            if (REG) e[i] = ev;
            zl += ev;
        }
        const float Z = block_sum(zl, redf);
        const float thr = (float)(1.0 - (double)a.top_p) * Z;
        int idx_bits = 1;
        while ((1 << idx_bits) < V) ++idx_bits;
        unsigned long long prefix = 0ull;
        for (int step = 0; step < 32 + idx_bits; ++step) {
            const int bit = step < 32 ? 63 - step : idx_bits - 1 - (step - 32);
            const unsigned long long cand = prefix | (1ull << bit);
            float s = 0.f;
            _Pragma("unroll") for (int i = 0; i < niter; ++i) {
                const int c = i * SMP_THREADS + tid;
                const float v = SMP_X(i, c);
                const unsigned long long k = ((unsigned long long)f2key(v) << 32) | (unsigned)c;
                s += (c < V && k < cand) ? (REG ? e[i] : (v != NEG_INF ? expf(v - m) : 0.f)) : 0.f;
            }
            if (block_sum(s, redf) <= thr) prefix = cand;
        }
        // everything strictly below `prefix` is the removed prefix of the ascending order
        _Pragma("unroll") for (int i = 0; i < niter; ++i) {
            const int c = i * SMP_THREADS + tid;
            if (c >= V) continue;
            const unsigned long long k = ((unsigned long long)f2key(SMP_X(i, c)) << 32) | (unsigned)c;
            if (k < prefix && k != top) { if (REG) x[i] = NEG_INF; else sc[c] = NEG_INF; }
        }
        if (!REG) __syncthreads();
    }

    // ---- processed scores out (HF output_scores) and the token
    unsigned long long best = 0ull;
    unsigned long long seed = 0, ctr = 0;
    if (a.do_sample && a.rng) { seed = a.rng[0]; ctr = a.rng[1] + (unsigned long long)a.draw; }
    _Pragma("unroll") for (int i = 0; i < niter; ++i) {
        const int c = i * SMP_THREADS + tid;
        if (c >= V) continue;
        float v = SMP_X(i, c);
        if (REG) sc[c] = v;
        if (a.do_sample && v != NEG_INF) {
            unsigned r[4];
            philox4x32((unsigned)(c >> 2), (unsigned)b, (unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
            v += gumbel_from_bits(r[c & 3]);
        }
        // removed tokens (-inf) can never win: their key is the smallest finite-or-infinite image; ties -> lowest index
        const unsigned long long k = ((unsigned long long)f2key(v) << 32) | (0xFFFFFFFFu - (unsigned)c);
        best = k > best ? k : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(best & 0xFFFFFFFFu), o, 64), hi = __shfl_xor((unsigned)(best >> 32), o, 64);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        best = other > best ? other : best;
    }
    __syncthreads();
    if ((tid & 63) == 0) red64[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int w = 0; w < 16; ++w) best = red64[w] > best ? red64[w] : best;
        long long id = (long long)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFu));
        if (a.done) {                                            // HF: finished rows emit pad; a row finishes when it emits eos
            if (a.done[b]) id = a.pad;
            else if (a.eos >= 0 && id == a.eos) a.done[b] = 1;
        }
        a.ids[b] = id;
        if (a.seq) a.seq[(long long)b * a.ld_seq + a.pos] = id;
    }
#undef SMP_X
}

extern "C" int egomi_sample_rows(const void* logits, int64_t ld, int B, int V, float* scores, int64_t ld_scores, int64_t* seq, int64_t ld_seq,
                                 int pos, int rep_from, int64_t* ids, int* done, float repetition_penalty, float temperature, int top_k,
                                 float top_p, int do_sample, const uint64_t* rng, int draw, int64_t eos_id, int64_t pad_id, int dtype,
                                 egomi_stream_t stream) {
    if (!logits || !scores || !ids) return EGOMI_E_BADARG;
    if (B <= 0 || V <= 0 || ld < V || ld_scores < V || (seq && (pos < 0 || pos >= ld_seq || rep_from < 0 || rep_from > pos))) return EGOMI_E_SHAPE;
    if (!(repetition_penalty > 0.f) || !(temperature > 0.f) || top_k < 0 || !(top_p > 0.f) || top_p > 1.f) return EGOMI_E_BADARG;
    if (do_sample && !rng) return EGOMI_E_BADARG;
    if (repetition_penalty != 1.0f && !seq) return EGOMI_E_BADARG;
    SampleArgs a;
    a.logits = logits; a.ld = ld; a.B = B; a.V = V; a.scores = scores; a.ld_scores = ld_scores; a.seq = seq; a.ld_seq = ld_seq; a.pos = pos;
    a.rep_from = rep_from; a.ids = ids; a.done = done; a.rep_penalty = repetition_penalty; a.temperature = temperature; a.top_k = top_k;
    a.top_p = top_p; a.do_sample = do_sample; a.rng = (const unsigned long long*)rng; a.draw = draw; a.eos = eos_id; a.pad = pad_id;
    const bool reg = V <= SMP_THREADS * SMP_NPT;
    if (reg) EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH((sample_rows_kernel<T, true>), dim3(B), dim3(SMP_THREADS), 0, (hipStream_t)stream, a));
    else EGOMI_DISPATCH_DTYPE(dtype, EGOMI_LAUNCH((sample_rows_kernel<T, false>), dim3(B), dim3(SMP_THREADS), 0, (hipStream_t)stream, a));
    return egomi_launch_status();
}
