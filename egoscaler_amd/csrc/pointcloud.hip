// Point-cloud front end: RGB-D un-projection + ordered compaction (A1), pc_norm (A2), farthest
// point sampling (A3), kNN grouping (A4+A5).  SURVEY.md §8a.  All HBM/latency-bound integer+fp32
// work: no MFMA here.  This file is compiled with -ffp-contract=off: the index contracts (FPS and
// kNN indices bit-exact with the reference's fp32 arithmetic) need un-fused mul/add.
#include "common.h"
#include <math.h>

// =================================================================================================
// A1  un-projection.  reference: data/tools/pcm_tools.py:68-96
// two launches, no scan pass and nobody waits on another block:
//   (1) unp_mask_kernel : every pixel read ONCE (7 B: 16-B vector loads, 16 pixels per thread) -> 16 validity bits per
//       thread + one count per 4096-pixel chunk
//   (2) dense form: unp_write_dense_kernel, one block per chunk: sums the sample's chunk counts itself (its own offset; a few
//       KB from L2), then lanes walk consecutive pixels so that ranks, hence output rows, are consecutive per wave;
//       subsample form (first-N-valid strided subsample): unp_pick_kernel, one thread per output row.
// =================================================================================================
#define UNP_ITEMS 16
#define UNP_THREADS 256
#define UNP_CHUNK (UNP_ITEMS * UNP_THREADS)

struct UnpParams {
    const uint8_t* rgb; const float* depth; const int32_t* boxes; int n_boxes;
    int T, H, W; long long L;   // L = T*H*W pixels per sample
    double pp, fx, fy; float d_thres; int use_thres;
    int nchunks;
};

__device__ __forceinline__ bool unp_valid(const UnpParams& p, const uint8_t* rgb, const float* depth, long long i) {
    const uint8_t* c = rgb + i * 3;
    bool ok = (c[0] != 0) & (c[1] != 0) & (c[2] != 0);                // pcm_tools.py:79
    if (p.use_thres) ok = ok && (depth[i] < p.d_thres);               // :87-89 (NaN depth -> false)
    if (p.n_boxes > 0) {                                               // :80-85
        const int hw = p.H * p.W;
        const int r = (int)(i % hw);
        const int v = r / p.W, u = r % p.W;
        for (int k = 0; k < p.n_boxes; ++k) {
            const int32_t* b = p.boxes + 4 * k;
            if (v >= b[0] && v < b[1] && u >= b[2] && u < b[3]) ok = false;
        }
    }
    return ok;
}

__global__ __launch_bounds__(UNP_THREADS) void unp_mask_kernel(UnpParams p, uint16_t* mask, uint16_t* rank, int32_t* chunk_cnt) {
    const int b = blockIdx.y, ch = blockIdx.x;
    const uint8_t* rgb = p.rgb + (long long)b * p.L * 3;
    const float* depth = p.depth + (long long)b * p.L;
    const long long base = (long long)ch * UNP_CHUNK + (long long)threadIdx.x * UNP_ITEMS;
    unsigned m = 0;
    const uint8_t* c = rgb + base * 3;
    if (base + UNP_ITEMS <= p.L && p.n_boxes == 0 && ((((uintptr_t)c) | ((uintptr_t)(depth + base))) & 15) == 0) {
        // fast path: 48 B of colour + 64 B of depth per thread as 16-B vectors, no box test
        const u32x4* c16 = reinterpret_cast<const u32x4*>(c);
        const u32x4 a0 = c16[0], a1 = c16[1], a2 = c16[2];
        const uint32_t w[12] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3]};
        const f32x4* z4 = reinterpret_cast<const f32x4*>(depth + base);
        const f32x4 z0 = z4[0], z1 = z4[1], z2 = z4[2], z3 = z4[3];
        const float z[UNP_ITEMS] = {z0[0], z0[1], z0[2], z0[3], z1[0], z1[1], z1[2], z1[3], z2[0], z2[1], z2[2], z2[3], z3[0], z3[1], z3[2], z3[3]};
#pragma unroll
        for (int k = 0; k < UNP_ITEMS; ++k) {
            const int o = 3 * k;
            const uint32_t c0 = (w[o >> 2] >> (8 * (o & 3))) & 0xFF, c1 = (w[(o + 1) >> 2] >> (8 * ((o + 1) & 3))) & 0xFF,
                           c2 = (w[(o + 2) >> 2] >> (8 * ((o + 2) & 3))) & 0xFF;
            bool ok = (c0 != 0) & (c1 != 0) & (c2 != 0);
            if (p.use_thres) ok = ok && (z[k] < p.d_thres);
            m |= (ok ? 1u : 0u) << k;
        }
    } else {
        for (int k = 0; k < UNP_ITEMS; ++k) {
            const long long i = base + k;
            if (i < p.L && unp_valid(p, rgb, depth, i)) m |= 1u << k;
        }
    }
    const long long mi = (long long)b * p.nchunks * UNP_THREADS + (long long)ch * UNP_THREADS + threadIdx.x;
    mask[mi] = (uint16_t)m;
    // exclusive rank of this thread's first valid pixel inside the chunk (the subsample pass binary-searches these)
    const int cnt = __popc(m);
    int x = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if ((threadIdx.x & 63) >= o) x += y;
    }
    __shared__ int wsum[UNP_THREADS / 64];
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    int woff = 0;
    for (int w2 = 0; w2 < (int)(threadIdx.x >> 6); ++w2) woff += wsum[w2];
    rank[mi] = (uint16_t)(woff + x - cnt);                              // <= 4096 - 16
    if (threadIdx.x == UNP_THREADS - 1) chunk_cnt[b * p.nchunks + ch] = woff + x;
}

// un-projection of one pixel (pcm_tools.py:74-78): fp64 point, fp32 colour
__device__ __forceinline__ void unp_emit(const UnpParams& p, const uint8_t* rgb, const float* depth, long long i, double* op, float* oc) {
    const int hw = p.H * p.W;
    const int r = (int)(i % hw);
    const int v = r / p.W, u = r % p.W;
    const double z = (double)depth[i];
    const double xn = ((double)u - p.pp) / p.fx;              // pcm_tools.py:74
    const double yn = ((double)v - p.pp) / p.fy;              // :75
    op[0] = xn * z;                                           // :77
    op[1] = yn * z;
    op[2] = z;
    const uint8_t* c = rgb + i * 3;
    oc[0] = (float)c[0] / 255.0f;                             // :78 (float32 image / 255.0)
    oc[1] = (float)c[1] / 255.0f;
    oc[2] = (float)c[2] / 255.0f;
}

// (2a) dense form (n_out == 0): every valid pixel, in row-major order.  One block per chunk; lanes walk CONSECUTIVE pixels, so
// the ranks a wave hands out are consecutive rows of the output and its stores land side by side (the 16-pixels-per-thread
// mapping of the mask pass would scatter 24-B pieces: 0.65 TB/s measured).
__global__ __launch_bounds__(UNP_THREADS) void unp_write_dense_kernel(UnpParams p, const uint16_t* mask, const int32_t* chunk_cnt,
                                                                      int32_t* out_count, long long cap, double* out_points, float* out_colors) {
    const int b = blockIdx.y, ch = blockIdx.x;
    const int32_t* cc = chunk_cnt + (long long)b * p.nchunks;
    int pre = 0, tot = 0;
    for (int i = threadIdx.x; i < p.nchunks; i += UNP_THREADS) {
        const int v = cc[i];
        tot += v;
        pre += i < ch ? v : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pre += __shfl_xor(pre, o, 64); tot += __shfl_xor(tot, o, 64); }
    __shared__ int red[2][UNP_THREADS / 64];
    __shared__ int wcnt[UNP_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = pre; red[1][wave] = tot; }
    // each wave owns 1024 consecutive pixels = 64 mask words: lane l holds word l of its wave
    const uint16_t* mw = mask + (long long)b * p.nchunks * UNP_THREADS + (long long)ch * UNP_THREADS + wave * 64;
    const unsigned myword = mw[lane];
    int wc = __popc(myword);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wc += __shfl_xor(wc, o, 64);
    if (lane == 0) wcnt[wave] = wc;
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < UNP_THREADS / 64; ++w) { off += red[0][w]; total += red[1][w]; }
    if (ch == 0 && threadIdx.x == 0) out_count[b] = total;
    if (cc[ch] == 0) return;
    for (int w = 0; w < wave; ++w) off += wcnt[w];
    const uint8_t* rgb = p.rgb + (long long)b * p.L * 3;
    const float* depth = p.depth + (long long)b * p.L;
    const long long pix0 = (long long)ch * UNP_CHUNK + wave * 1024;
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
        // pixel it*64 + lane of the wave: bit (lane & 15) of word it*4 + lane/16
        const unsigned word = __shfl(myword, it * 4 + (lane >> 4), 64);
        const bool ok = (word >> (lane & 15)) & 1u;
        const unsigned long long bal = __ballot(ok);
        if (ok) {
            const int rank = __popcll(bal & ((1ull << lane) - 1ull));
            const long long row = (long long)b * cap + off + rank;
            unp_emit(p, rgb, depth, pix0 + it * 64 + lane, out_points + row * 3, out_colors + row * 3);
        }
        off += __popcll(bal);
    }
}

// (2b) subsample form (n_out > 0): one THREAD per output row j.  The row is valid pixel number j * stride of its sample
// (stride = total / n_out): the block scans the sample's chunk counts once in LDS, each thread binary-searches its chunk,
// walks that chunk's 256 mask words to the word holding its pixel and picks the bit.  O(n_out) work instead of a pass over
// every chunk (44 us -> a few us at 16 x 448^2, B = 8).
#define UNP_PICK_THREADS 1024
__global__ __launch_bounds__(UNP_PICK_THREADS) void unp_pick_kernel(UnpParams p, const uint16_t* mask, const uint16_t* rank, const int32_t* chunk_cnt,
                                                                    int32_t* out_count, int n_out, double* out_points, float* out_colors) {
    extern __shared__ int pfx[];                               // inclusive prefix of the sample's chunk counts
    __shared__ int wtot[UNP_PICK_THREADS / 64];
    __shared__ int carry_s;
    const int b = blockIdx.y;
    const int32_t* cc = chunk_cnt + (long long)b * p.nchunks;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < p.nchunks; base += UNP_PICK_THREADS) {
        const int i = base + threadIdx.x;
        const int v = (i < p.nchunks) ? cc[i] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int y = __shfl_up(x, o, 64);
            if ((threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wtot[w];
        const int carry = carry_s;
        if (i < p.nchunks) pfx[i] = carry + woff + x;
        __syncthreads();
        if (threadIdx.x == UNP_PICK_THREADS - 1) carry_s = carry + woff + x;
        __syncthreads();
    }
    const int total = carry_s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out_count[b] = total < n_out ? -total : total;
    if (total < n_out) return;                                 // too few valid pixels for the subsample (caller checks the sign)
    const int j = blockIdx.x * UNP_PICK_THREADS + threadIdx.x;
    if (j >= n_out) return;
    const int stride = total / n_out;
    const int o = j * stride;                                  // ordinal of the pixel this row takes (< total)
    int lo = 0, hi = p.nchunks - 1;                            // first chunk whose inclusive prefix exceeds o
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pfx[mid] > o) hi = mid; else lo = mid + 1;
    }
    const int rem = o - (lo ? pfx[lo - 1] : 0);                // rank inside chunk lo
    const long long cbase = (long long)b * p.nchunks * UNP_THREADS + (long long)lo * UNP_THREADS;
    const uint16_t* rk = rank + cbase;
    int wl = 0, wh = UNP_THREADS - 1;                          // last thread word whose exclusive rank is <= rem and that holds a pixel
    while (wl < wh) {
        const int mid = (wl + wh + 1) >> 1;
        if ((int)rk[mid] <= rem) wl = mid; else wh = mid - 1;
    }
    // words with no valid pixel share the rank of their successor: the search lands on the LAST of them, which is the holder
    unsigned word = mask[cbase + wl];
    int r = rem - (int)rk[wl];
    int bit = 0;
    for (; bit < 16; ++bit) {
        if ((word >> bit) & 1u) { if (r == 0) break; --r; }
    }
    const long long i = (long long)lo * UNP_CHUNK + (long long)wl * UNP_ITEMS + bit;
    const long long row = (long long)b * n_out + j;
    unp_emit(p, p.rgb + (long long)b * p.L * 3, p.depth + (long long)b * p.L, i, out_points + row * 3, out_colors + row * 3);
}

static inline int unp_nchunks(long long L) { return (int)((L + UNP_CHUNK - 1) / UNP_CHUNK); }

extern "C" size_t egomi_unproject_workspace_bytes(int B, int T, int H, int W) {
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0) return 0;
    const long long L = (long long)T * H * W;
    const long long nch = unp_nchunks(L);
    size_t mask_bytes = (size_t)B * nch * UNP_THREADS * sizeof(uint16_t);
    mask_bytes = (mask_bytes + 255) & ~(size_t)255;
    return 2 * mask_bytes + (size_t)B * nch * sizeof(int32_t);       // validity words | in-chunk ranks | chunk counts
}

extern "C" int egomi_unproject_gather(const uint8_t* rgb, const float* depth, const int32_t* boxes, int n_boxes,
                                      int B, int T, int H, int W, double pp, double fx, double fy, float d_thres,
                                      int n_out, double* out_points, float* out_colors, int32_t* out_count,
                                      void* workspace, size_t workspace_bytes, egomi_stream_t stream) {
    if (!rgb || !depth || !out_points || !out_colors || !out_count || !workspace) return EGOMI_E_BADARG;
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || n_out < 0 || n_boxes < 0 || (n_boxes > 0 && !boxes)) return EGOMI_E_SHAPE;
    if (fx == 0.0 || fy == 0.0) return EGOMI_E_BADARG;
    const long long L = (long long)T * H * W;
    if (L >= (1ll << 31)) return EGOMI_E_UNSUPPORTED;
    if (workspace_bytes < egomi_unproject_workspace_bytes(B, T, H, W)) return EGOMI_E_SHAPE;
    UnpParams p;
    p.rgb = rgb; p.depth = depth; p.boxes = boxes; p.n_boxes = n_boxes;
    p.T = T; p.H = H; p.W = W; p.L = L; p.pp = pp; p.fx = fx; p.fy = fy;
    p.use_thres = !(d_thres != d_thres); p.d_thres = d_thres;
    p.nchunks = unp_nchunks(L);
    size_t mask_bytes = ((size_t)B * p.nchunks * UNP_THREADS * sizeof(uint16_t) + 255) & ~(size_t)255;
    uint16_t* mask = (uint16_t*)workspace;
    uint16_t* rank = (uint16_t*)((char*)workspace + mask_bytes);
    int32_t* cnt = (int32_t*)((char*)workspace + 2 * mask_bytes);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(p.nchunks, B);
    EGOMI_LAUNCH(unp_mask_kernel, grid, dim3(UNP_THREADS), 0, s, p, mask, rank, cnt);
    if (n_out > 0 && (size_t)p.nchunks * sizeof(int) > 60 * 1024) return EGOMI_E_UNSUPPORTED;   // subsample form keeps the chunk prefix in LDS: <= 62.9 M pixels per sample
    if (n_out > 0)
        EGOMI_LAUNCH(unp_pick_kernel, dim3((n_out + UNP_PICK_THREADS - 1) / UNP_PICK_THREADS, B), dim3(UNP_PICK_THREADS), (size_t)p.nchunks * sizeof(int), s,
                     p, mask, rank, cnt, out_count, n_out, out_points, out_colors);
    else
        EGOMI_LAUNCH(unp_write_dense_kernel, grid, dim3(UNP_THREADS), 0, s, p, mask, cnt, out_count, L, out_points, out_colors);
    return egomi_launch_status();
}

// =================================================================================================
// A2  pc_norm.  reference: models/pointllm/pointllm/data/utils.py:146-157.  One block per sample.
// =================================================================================================
__device__ __forceinline__ double block_sum_f64(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__global__ __launch_bounds__(1024) void pc_norm_kernel(const double* points, const float* colors, float* out, int N) {
    const int b = blockIdx.x;
    const double* P = points + (long long)b * N * 3;
    const float* Cc = colors + (long long)b * N * 3;
    float* O = out + (long long)b * N * 6;
    __shared__ double red[16];
    double sx = 0, sy = 0, sz = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) { sx += P[3 * i]; sy += P[3 * i + 1]; sz += P[3 * i + 2]; }
    const double cx = block_sum_f64(sx, red) / (double)N;
    const double cy = block_sum_f64(sy, red) / (double)N;
    const double cz = block_sum_f64(sz, red) / (double)N;
    double r2 = 0.0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double x = P[3 * i] - cx, y = P[3 * i + 1] - cy, z = P[3 * i + 2] - cz;
        r2 = fmax(r2, (x * x + y * y) + z * z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r2 = fmax(r2, __shfl_xor(r2, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = r2;
    __syncthreads();
    double m2 = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m2 = fmax(m2, red[i]);
    const double m = sqrt(m2);                      // max(sqrt(.)) == sqrt(max(.)): sqrt is monotone
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        O[6 * i + 0] = (float)((P[3 * i] - cx) / m);
        O[6 * i + 1] = (float)((P[3 * i + 1] - cy) / m);
        O[6 * i + 2] = (float)((P[3 * i + 2] - cz) / m);
        O[6 * i + 3] = Cc[3 * i];
        O[6 * i + 4] = Cc[3 * i + 1];
        O[6 * i + 5] = Cc[3 * i + 2];
    }
}

extern "C" int egomi_pc_norm(const double* points, const float* colors, float* out, int B, int N, egomi_stream_t stream) {
    if (!points || !colors || !out) return EGOMI_E_BADARG;
    if (B <= 0 || N <= 0) return EGOMI_E_SHAPE;
    EGOMI_LAUNCH(pc_norm_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, points, colors, out, N);
    return egomi_launch_status();
}

// =================================================================================================
// A3  farthest point sampling.  reference: pointbert/misc.py:40-60
// One 1024-thread workgroup per cloud.  Each thread keeps PPT points and their running distances in
// registers; the cloud's xyz is also staged in LDS (12 B/point) so the new centroid is one
// broadcast LDS read.  Per iteration: PPT distance updates, a 64-bit (distance, ~index) arg-max by
// wave shuffles, one barrier (double-buffered per-wave slots).
// =================================================================================================
#define FPS_THREADS 1024

// Wave-wide max of a 64-bit key by DPP row operations (gfx9 scan sequence: row_shr 1,2,4,8 inside the rows of 16, then
// row_bcast15 / row_bcast31 across rows); lanes that receive nothing see the identity 0.  The total ends in lane 63 and is
// returned wave-uniform (v_readlane).  A ds_bpermute butterfly (12 LDS-crossbar round trips for 64 bits) was the longest
// stretch of the FPS iteration's critical path.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long k) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k & 0xFFFFFFFFu), CTRL, ROW_MASK, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, ROW_MASK, 0xF, false);
    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
    return other > k ? other : k;
}
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long k) {      // total of each row of 16 in its last lane
    k = dpp_max_step<0x111, 0xF>(k);
    k = dpp_max_step<0x112, 0xF>(k);
    k = dpp_max_step<0x114, 0xF>(k);
    k = dpp_max_step<0x118, 0xF>(k);
    return k;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long k, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k & 0xFFFFFFFFu), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
    k = row16_max_u64(k);
    k = dpp_max_step<0x142, 0xA>(k);                               // row_bcast15 -> rows 1 and 3
    k = dpp_max_step<0x143, 0xC>(k);                               // row_bcast31 -> rows 2 and 3
    return readlane_u64(k, 63);
}

template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* pts, int N, int C, const int32_t* start, int G,
                                                          int32_t* out_idx, float* out_center) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem);   // [2][16]
    float* lp = reinterpret_cast<float*>(smem + 2 * 16 * sizeof(unsigned long long));   // [N][3]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* P = pts + (long long)b * N * C;
    // points of a thread in pairs: the distance update runs on packed fp32 (v_pk_add/mul_f32; this file is built with
    // -ffp-contract=off, so every element sees misc.py:57's un-fused (dx*dx + dy*dy) + dz*dz), and the thread's arg-max is kept
    // as (distance, index) with a strict compare in increasing index order — the lowest index wins ties, as in the 64-bit key
    constexpr int PP = (PPT + 1) / 2;
    f32x2 px[PP], py[PP], pz[PP], dist[PP];
#pragma unroll
    for (int s = 0; s < 2 * PP; ++s) {
        const int j = s * FPS_THREADS + tid;
        float x = 0.f, y = 0.f, z = 0.f, d0 = -1.f;                        // padding: never selected
        if (s < PPT && j < N) {
            x = P[(long long)j * C]; y = P[(long long)j * C + 1]; z = P[(long long)j * C + 2];
            lp[3 * j] = x; lp[3 * j + 1] = y; lp[3 * j + 2] = z;
            d0 = 1e10f;                                                    // misc.py:51
        }
        px[s >> 1][s & 1] = x; py[s >> 1][s & 1] = y; pz[s >> 1][s & 1] = z; dist[s >> 1][s & 1] = d0;
    }
    __syncthreads();
    int far = start[b];
    for (int i = 0; i < G; ++i) {
        const float cx = lp[3 * far], cy = lp[3 * far + 1], cz = lp[3 * far + 2];
        if (tid == 0) {
            out_idx[(long long)b * G + i] = far;                           // misc.py:55
            float* oc = out_center + ((long long)b * G + i) * 3;
            oc[0] = cx; oc[1] = cy; oc[2] = cz;
        }
        const f32x2 cxv = {cx, cx}, cyv = {cy, cy}, czv = {cz, cz};
        float bd = -1.f;
        unsigned bj = 0u;
#pragma unroll
        for (int s = 0; s < PP; ++s) {
            const f32x2 dx = px[s] - cxv, dy = py[s] - cyv, dz = pz[s] - czv;
            const f32x2 d = (dx * dx + dy * dy) + dz * dz;                 // misc.py:57 (un-fused)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float nd = fminf(dist[s][e], d[e]);                  // misc.py:58; padding stays at -1 and never beats bd
                dist[s][e] = nd;
                const bool c = nd > bd;
                bd = c ? nd : bd;
                bj = c ? (unsigned)((2 * s + e) * FPS_THREADS + tid) : bj;
            }
        }
        unsigned long long best = bd < 0.f ? 0ull : (((unsigned long long)__float_as_uint(bd) << 32) | (0xFFFFFFFFu - bj));
        best = wave_max_u64(best);
        unsigned long long* sl = slots + (i & 1) * 16;
        if ((tid & 63) == 0) sl[tid >> 6] = best;
        __syncthreads();
        // the 16 wave results: one per lane of row 0, reduced inside that row, read back from its last lane
        const unsigned long long k = readlane_u64(row16_max_u64(sl[tid & 15]), 15);
        far = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu));            // misc.py:59, lowest index on ties
    }
}

extern "C" int egomi_fps(const float* pts, int B, int N, int C, const int32_t* start, int G,
                         int32_t* out_idx, float* out_center, egomi_stream_t stream) {
    if (!pts || !start || !out_idx || !out_center) return EGOMI_E_BADARG;
    if (B <= 0 || N <= 0 || C < 3 || G <= 0) return EGOMI_E_SHAPE;
    if (N > 16 * FPS_THREADS) return EGOMI_E_UNSUPPORTED;
    const size_t lds = 2 * 16 * sizeof(unsigned long long) + (size_t)N * 3 * sizeof(float);
    if (lds > 160 * 1024) return EGOMI_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int ppt = (N + FPS_THREADS - 1) / FPS_THREADS;
#define FPS_LAUNCH(P)                                                                                       \
    do {                                                                                                    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        EGOMI_LAUNCH(fps_kernel<P>, dim3(B), dim3(FPS_THREADS), lds, s, pts, N, C, start, G, out_idx, out_center); \
    } while (0)
    if (ppt <= 1) FPS_LAUNCH(1);
    else if (ppt <= 2) FPS_LAUNCH(2);
    else if (ppt <= 4) FPS_LAUNCH(4);
    else if (ppt <= 8) FPS_LAUNCH(8);
    else FPS_LAUNCH(16);
#undef FPS_LAUNCH
    return egomi_launch_status();
}

// =================================================================================================
// A4+A5  kNN grouping.  reference: pointbert/dvae.py:107-140,150-187
// One wave per (cloud, centre).  Each lane holds CAND = 128 candidate distances in registers
// (point j = slot*64 + lane); K rounds of {lane-local arg-min, 64-bit wave arg-min on
// (orderable distance, index), invalidate}.  Rounds come out ordered by (distance, index).
// =================================================================================================
#define KNN_CAND 128
#define KNN_WAVES 4

__device__ __forceinline__ unsigned f32_orderable(float f) {
    unsigned u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

template <typename TO>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_group_kernel(const float* pts, const float* center, int N, int C, int G, int K,
                                                                  int total_groups, int32_t* out_idx, TO* out_nb) {
    const int lane = threadIdx.x & 63;
    const int gidx = blockIdx.x * KNN_WAVES + (threadIdx.x >> 6);        // b*G + g
    if (gidx >= total_groups) return;                                   // whole wave exits together
    const int b = gidx / G;
    const float* P = pts + (long long)b * N * C;
    const float cx = center[(long long)gidx * 3], cy = center[(long long)gidx * 3 + 1], cz = center[(long long)gidx * 3 + 2];
    const float na = (cx * cx + cy * cy) + cz * cz;
    float d[KNN_CAND];
#pragma unroll
    for (int s = 0; s < KNN_CAND; ++s) {
        const int j = s * 64 + lane;
        if (j < N) {
            const float x = P[(long long)j * C], y = P[(long long)j * C + 1], z = P[(long long)j * C + 2];
            const float dot = (cx * x + cy * y) + cz * z;
            const float nb = (x * x + y * y) + z * z;
            float t = -2.0f * dot;                                       // dvae.py:137
            t = t + na;                                                  // dvae.py:138
            t = t + nb;                                                  // dvae.py:139
            d[s] = t;
        } else {
            d[s] = INFINITY;
        }
    }
    // Two-level arg-min: the lane's 128 candidates are kept as 8 groups of 16 with their minima (orderable distance, slot); a
    // round takes the best of the 8 minima, the wave picks the winner by (distance, index), and only the winner's GROUP — a
    // wave-uniform number, so a scalar branch with static register indices — is re-scanned after its candidate is struck out.
    // (The first form re-scanned all 128 candidates of every lane in every round: 32 x 512 VALU instructions per centre.)
    constexpr int KG = 16, KNG = KNN_CAND / KG;
    unsigned gu[KNG];
    int gsl[KNG];
#pragma unroll
    for (int g = 0; g < KNG; ++g) {
        unsigned mu = 0xFFFFFFFFu;
        int ms = g * KG;
#pragma unroll
        for (int s = g * KG; s < (g + 1) * KG; ++s) {
            const unsigned u = f32_orderable(d[s]);
            if (u < mu) { mu = u; ms = s; }
        }
        gu[g] = mu; gsl[g] = ms;
    }
    int mine = 0;
    for (int r = 0; r < K; ++r) {
        unsigned bu = 0xFFFFFFFFu;
        int bs = 0;
#pragma unroll
        for (int g = 0; g < KNG; ++g)
            if (gu[g] < bu) { bu = gu[g]; bs = gsl[g]; }                  // increasing slot order, strict: the lowest index wins ties
        const unsigned long long key = ((unsigned long long)bu << 32) | (unsigned)(bs * 64 + lane);
        const unsigned long long kmin = ~wave_max_u64(~key);              // wave-uniform (distance, index) minimum
        const int widx = (int)(kmin & 0xFFFFFFFFu);
        const int kslot = widx >> 6, kgrp = __builtin_amdgcn_readfirstlane(kslot / KG);
        const bool winner = (widx & 63) == lane;
#pragma unroll
        for (int g = 0; g < KNG; ++g) {
            if (g != kgrp) continue;                                       // wave-uniform
            unsigned mu = 0xFFFFFFFFu;
            int ms = g * KG;
#pragma unroll
            for (int s = g * KG; s < (g + 1) * KG; ++s) {
                if (winner && s == kslot) d[s] = INFINITY;
                const unsigned u = f32_orderable(d[s]);
                if (u < mu) { mu = u; ms = s; }
            }
            gu[g] = mu; gsl[g] = ms;
        }
        if (lane == r) mine = widx;
    }
    if (lane < K) out_idx[(long long)gidx * K + lane] = mine;            // ordered by (distance, index)
    const int tot = K * C;
    for (int e = lane; e < tot; e += 64) {
        const int r = e / C, c = e - r * C;
        const int idx = __shfl(mine, r, 64);
        float v = P[(long long)idx * C + c];
        if (c == 0) v -= cx; else if (c == 1) v -= cy; else if (c == 2) v -= cz;   // dvae.py:182: xyz only
        Cvt<TO>::st(out_nb + (long long)gidx * tot + e, v);
    }
}

extern "C" int egomi_knn_group(const float* pts, const float* center, int B, int N, int C, int G, int K,
                               int32_t* out_idx, void* out_nb, int out_dtype, egomi_stream_t stream) {
    if (!pts || !center || !out_idx || !out_nb) return EGOMI_E_BADARG;
    if (B <= 0 || N <= 0 || C < 3 || G <= 0 || K <= 0 || K > N) return EGOMI_E_SHAPE;
    if (N > KNN_CAND * 64 || K > 64) return EGOMI_E_UNSUPPORTED;
    const int total = B * G;
    const int blocks = (total + KNN_WAVES - 1) / KNN_WAVES;
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == EGOMI_F32)
        EGOMI_LAUNCH(knn_group_kernel<float>, dim3(blocks), dim3(KNN_WAVES * 64), 0, s, pts, center, N, C, G, K, total, out_idx, (float*)out_nb);
    else if (out_dtype == EGOMI_BF16)
        EGOMI_LAUNCH(knn_group_kernel<bf16_t>, dim3(blocks), dim3(KNN_WAVES * 64), 0, s, pts, center, N, C, G, K, total, out_idx, (bf16_t*)out_nb);
    else
        return EGOMI_E_BADARG;
    return egomi_launch_status();
}

// =================================================================================================
// N4  depth map -> dense cloud  (DepthAnything.get_depth after the network, depth.py:46-60)
//   z      = nearest-neighbour resize of the prediction to the frame size.  Pillow's NEAREST walks each axis with a
//            running double (xo = a/2; idx = (int)xo; xo += a, a = n_src/n_dst): the table kernel repeats that
//            sequence (one thread per axis, <= a few thousand additions), because floor((x+0.5)*a) differs from it
//   points = ((u-pp)/fx * z, (v-pp)/fy * z, z) in float64, every pixel, row-major;  colours = u8 / 255.0 in float64
// HBM-bound: 7 B read (+ the small prediction, L2-resident) and 52 B written per pixel.
// =================================================================================================
__global__ void depth_tab_kernel(int h0, int w0, int H, int W, int32_t* tab) {   // tab = [W x-indices | H y-indices]
    if (threadIdx.x > 1 || blockIdx.x) return;
    const int n_src = threadIdx.x ? h0 : w0, n_dst = threadIdx.x ? H : W;
    int32_t* t = tab + (threadIdx.x ? W : 0);
    const double a = (double)n_src / (double)n_dst;
    double xo = a * 0.5;
    for (int x = 0; x < n_dst; ++x) {
        int i = (int)xo;
        t[x] = i < n_src - 1 ? i : n_src - 1;
        xo += a;
    }
}

__global__ __launch_bounds__(256) void depth_cloud_kernel(const float* pred, long long pred_stride, int w0, const uint8_t* rgb, int H, int W,
                                                          double fx, double fy, double pp, const int32_t* tab,
                                                          float* out_z, double* out_points, double* out_colors, long long total) {
    // 256 consecutive pixels per block; the 3-double records (24-B stride per lane) are staged in LDS and leave as 16-B vectors
    __shared__ __attribute__((aligned(16))) double sp[256 * 3], sc[256 * 3];
    const long long i0 = (long long)blockIdx.x * 256;
    const long long i = i0 + threadIdx.x;
    const bool live = i < total;
    if (live) {
        const long long hw = (long long)H * W;
        const int b = (int)(i / hw);
        const int rem = (int)(i - (long long)b * hw);
        const int v = rem / W, u = rem - v * W;
        const float z = pred[(long long)b * pred_stride + (long long)tab[W + v] * w0 + tab[u]];
        out_z[i] = z;
        if (out_points) {
            const double zd = (double)z;
            const double xn = ((double)u - pp) / fx, yn = ((double)v - pp) / fy;     // depth.py:55-56
            sp[threadIdx.x * 3 + 0] = xn * zd; sp[threadIdx.x * 3 + 1] = yn * zd; sp[threadIdx.x * 3 + 2] = zd;   // :57
            const uint8_t* c = rgb + i * 3;
            sc[threadIdx.x * 3 + 0] = (double)c[0] / 255.0; sc[threadIdx.x * 3 + 1] = (double)c[1] / 255.0;       // :58
            sc[threadIdx.x * 3 + 2] = (double)c[2] / 255.0;
        }
    }
    if (!out_points) return;
    __syncthreads();
    const long long left = total - i0;
    const int npix = left < 256 ? (int)left : 256;
    const int nvec = npix * 3 / 2;                                       // i0*3 doubles is 16-B aligned (i0 % 256 == 0)
    typedef __attribute__((ext_vector_type(2))) double f64x2;
    f64x2* gp = reinterpret_cast<f64x2*>(out_points + i0 * 3);
    f64x2* gc = reinterpret_cast<f64x2*>(out_colors + i0 * 3);
    for (int k = threadIdx.x; k < nvec; k += 256) {
        gp[k] = *reinterpret_cast<const f64x2*>(sp + 2 * k);
        gc[k] = *reinterpret_cast<const f64x2*>(sc + 2 * k);
    }
    if ((npix * 3) & 1) {                                                // odd tail double of the last block
        if (threadIdx.x == 0) { out_points[i0 * 3 + npix * 3 - 1] = sp[npix * 3 - 1]; out_colors[i0 * 3 + npix * 3 - 1] = sc[npix * 3 - 1]; }
    }
}

extern "C" int egomi_depth_to_cloud(const float* pred, int B, int h0, int w0, const uint8_t* rgb, int H, int W,
                                    double fx, double fy, double pp, int32_t* tab_ws,
                                    float* out_z, double* out_points, double* out_colors, egomi_stream_t stream) {
    if (!pred || !tab_ws || !out_z) return EGOMI_E_BADARG;
    if ((out_points == nullptr) != (out_colors == nullptr)) return EGOMI_E_BADARG;
    if (out_points && !rgb) return EGOMI_E_BADARG;
    if (B <= 0 || h0 <= 0 || w0 <= 0 || H <= 0 || W <= 0) return EGOMI_E_SHAPE;
    if (out_points && !(fx > 0.0 && fy > 0.0 && pp > 0.0)) return EGOMI_E_BADARG;      // depth.py:53: the reference returns no cloud then
    hipStream_t s = (hipStream_t)stream;
    EGOMI_LAUNCH(depth_tab_kernel, dim3(1), dim3(64), 0, s, h0, w0, H, W, tab_ws);
    const long long total = (long long)B * H * W;
    EGOMI_LAUNCH(depth_cloud_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pred, (long long)h0 * w0, w0, rgb, H, W,
                 fx, fy, pp, tab_ws, out_z, out_points, out_colors, total);
    return egomi_launch_status();
}
