// Fused attention for the LLaMA layers (head_dim 128, bf16, causal + key-padding mask):
// forward with online softmax, and a deterministic two-kernel backward (dK/dV kernel, dQ kernel).
// Replaces HF eager_attention_forward (modeling_llama.py:191-214: scores + mask, fp32 softmax,
// P.V) and its autograd backward; the [S,S] score matrix never touches HBM.
//
// Orientation (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand"):
//   forward / dQ kernel:  X = K.Q^T  (32 keys x 32 queries, v_mfma_f32_32x32x16_bf16)  -> the QUERY
//     sits on the lane, so running max / sum / LSE / delta are lane-local; P (bf16) is then fed
//     back as the B operand of  O^T += V^T.P^T  (resp. dQ^T += K^T.dS^T) with no lane movement,
//     V^T / K^T fragments coming from ds_read_b64_tr_b16 on row-major LDS tiles.
//   dK/dV kernel:  X = Q.K^T (32 queries x 32 keys) -> the KEY sits on the lane; -LSE/scale and
//     -delta enter as initial accumulators; dV^T += dO^T.P, dK^T += Q^T.dS.
// LDS tiles are [rows][128] bf16 (256-B rows) with the dual-use swizzle
//   off(row, ch) = 256*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3)))
// which is conflict-free for both the b128 row reads and the transposed reads (T10 image (b)).
#include "common.h"
#include <math.h>
#include <type_traits>

#define AT_HD 128
// raw v_exp_f32 (1 ulp, results below 2^-126 flush to 0): exp2f() expands to a 6-instruction denormal-safe sequence,
// which made the softmax arithmetic the longest instruction stream of every kernel here
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4;

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
    const bf16_t* dout; float* delta; bf16_t* dq; bf16_t* dk; bf16_t* dv;
    const uint8_t* key_mask;
    int B, H, S;
    long long ld_qkv, ld_o, ld_dqkv;
    float scale;
    int causal;
    const float* rope_cos; const float* rope_sin;
};

__device__ __forceinline__ int sw_off(int row, int ch) {            // byte offset inside a [rows][256 B] tile
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ int rowmap(int reg, int half) {          // C/D row of a 32x32 accumulator register
    return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}
// Per-lane visibility bits of one 32-key sub-tile: bit c (c = rowmap(r, 0)) is set when key  key0 + 4*half + c  may be
// attended by this lane's query (key-padding bits km32 of the sub-tile, causal limit qi).  32-bit arithmetic only.
__device__ __forceinline__ uint32_t visible_bits(uint32_t km32, int half, int key0, int qi, int causal) {
    uint32_t t = km32 >> (4 * half);
    if (causal) {
        const int lim = qi - key0 - 4 * half;                           // c <= lim
        const uint32_t cb = lim < 0 ? 0u : (lim >= 31 ? 0xFFFFFFFFu : ((2u << lim) - 1u));
        t &= cb;
    }
    return t;
}
__device__ __forceinline__ bf16x8 lds_row8(const char* tile, int row, int ch) {
    return *reinterpret_cast<const bf16x8*>(tile + sw_off(row, ch));
}
// transposed fragment: 8 bf16 = column (col0 + lane&31) of rows {r0 + 4*half + 0..3, r0 + 8 + 4*half + 0..3}
__device__ __forceinline__ bf16x8 lds_tr8(const char* tile, int r0, int col0, int lane) {
    const int half = lane >> 5, q = (lane >> 2) & 3, p = lane & 3;
    const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * p;            // element column this lane points at
    const int ch = col >> 3, within = (col & 7) * 2;
    const int ra = r0 + 4 * half + q;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + sw_off(ra, ch) + within));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + sw_off(ra + 8, ch) + within));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
__device__ __forceinline__ bf16x8 pack8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[j];
    return r;
}

// LDS-DMA one ROWS x 128 bf16 tile (rows clamped to row_max) into a swizzled LDS image: the LDS
// destination of a wave instruction is linear (4 rows x 16 chunks), so the swizzle goes on the SOURCE.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
template <int ROWS>
__device__ __forceinline__ void tile_dma(const bf16_t* base, long long ld, int row0, int row_max, char* tile, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < ROWS / 16; ++j) {
        const int rl = (wave * (ROWS / 16) + j) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
        int r = row0 + rl;
        r = r < row_max ? r : row_max;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + (long long)r * ld + ch * 8),
                                         (lds_void_t*)(tile + (wave * (ROWS / 16) + j) * 4 * 256), 16, 0, 0);
    }
}


// ---- head-dim generic forms (forward kernel: HD = 128 for the LLaMA layers, 64 for the PointBERT blocks).
// HD = 64 rows are 128 B, two per 256-B bank row; its swizzle f(row) = ((row>>1)&1)<<2 | (row>>2)&3 keeps both the b128 row
// reads (16 rows x one chunk -> 16 distinct 16-B slots) and the transposed reads (rows r, r+2 in different 64-B groups) conflict-free.
template <int HD> __device__ __forceinline__ int swz(int row) {
    return HD == 128 ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
}
template <int HD> __device__ __forceinline__ int sw_off_t(int row, int ch) { return 2 * HD * row + 16 * (ch ^ swz<HD>(row)); }
template <int HD> __device__ __forceinline__ bf16x8 lds_row8_t(const char* tile, int row, int ch) {
    return *reinterpret_cast<const bf16x8*>(tile + sw_off_t<HD>(row, ch));
}
template <int HD> __device__ __forceinline__ bf16x8 lds_tr8_t(const char* tile, int r0, int col0, int lane) {
    const int half = lane >> 5, q = (lane >> 2) & 3, p = lane & 3;
    const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * p;
    const int ch = col >> 3, within = (col & 7) * 2;
    const int ra = r0 + 4 * half + q;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + sw_off_t<HD>(ra, ch) + within));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + sw_off_t<HD>(ra + 8, ch) + within));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// one wave instruction = 1 KiB of the LDS image = (512 / HD) rows
template <int ROWS, int HD>
__device__ __forceinline__ void tile_dma_t(const bf16_t* base, long long ld, int row0, int row_max, char* tile, int wave, int lane) {
    constexpr int RPI = 512 / HD, CPR = HD / 8, IPW = ROWS / RPI / 4;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int rl = (wave * IPW + j) * RPI + lane / CPR;
        const int ch = (lane % CPR) ^ swz<HD>(rl);
        int r = row0 + rl;
        r = r < row_max ? r : row_max;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + (long long)r * ld + ch * 8),
                                         (lds_void_t*)(tile + (wave * IPW + j) * 1024), 16, 0, 0);
    }
}


// ---- epilogue rows through LDS.  The 32x32 accumulators hold, per lane, one row (query or key = lane & 31) and 4-element
// pieces of its 128 values, so a direct store instruction scatters 8-B pieces over 32 rows of a [*, ld] array (measured:
// 17 us of the 153-us dQ kernel).  Each wave instead drops its 32 x 128 bf16 block into a private LDS region (272-B row
// pitch: the 8-B writes of 32 rows spread over the banks) and stores it back as whole 256-B rows, 16 B per lane.
#define AT_XPITCH 272
#define AT_XBYTES (32 * AT_XPITCH)
__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(v[j] << 16); f[2 * j + 1] = __uint_as_float(v[j] & 0xFFFF0000u); }
}
__device__ __forceinline__ void store_rows_via_lds(char* wbuf, const f32x16 (&acc)[4], const float mul, bf16_t* gbase, const long long ld,
                                                   const int row0, const int nrows, const int lane,
                                                   const float* rcos = nullptr, const float* rsin = nullptr) {
    const int half = lane >> 5, rl = lane & 31;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 w;
            w[0] = (uint32_t)f2bf(acc[dt][4 * g] * mul) | ((uint32_t)f2bf(acc[dt][4 * g + 1] * mul) << 16);
            w[1] = (uint32_t)f2bf(acc[dt][4 * g + 2] * mul) | ((uint32_t)f2bf(acc[dt][4 * g + 3] * mul) << 16);
            *reinterpret_cast<u32x2*>(wbuf + rl * AT_XPITCH + (32 * dt + 8 * g + 4 * half) * 2) = w;
        }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = it * 4 + (lane >> 4), ch = lane & 15;
        u32x4 v = *reinterpret_cast<const u32x4*>(wbuf + r * AT_XPITCH + ch * 16);
        if (rcos && row0 + r < nrows && row0 + r >= 0) {
            // inverse rotation of the pair (d, d+64) on the bf16-rounded values, in egomi_rope's own rounding sequence
            // (elementwise.hip rope_vec8_kernel, inverse = 1): lanes 0..7 of a row hold the first halves, 8..15 the second
            const u32x4 w = *reinterpret_cast<const u32x4*>(wbuf + r * AT_XPITCH + (ch ^ 8) * 16);
            float mine[8], other[8], c[8], sn[8], out[8];
            unpack8(v, mine); unpack8(w, other);
            load8<float>(rcos + (long long)(row0 + r) * (AT_HD / 2) + (ch & 7) * 8, c);
            load8<float>(rsin + (long long)(row0 + r) * (AT_HD / 2) + (ch & 7) * 8, sn);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float cj = bf2f(f2bf(c[j])), sj = bf2f(f2bf(-sn[j]));
                out[j] = ch < 8 ? bf2f(f2bf(mine[j] * cj)) + bf2f(f2bf(-other[j] * sj))          // a*c - b*s
                                : bf2f(f2bf(mine[j] * cj)) + bf2f(f2bf(other[j] * sj));           // b*c + a*s
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (uint32_t)f2bf(out[2 * j]) | ((uint32_t)f2bf(out[2 * j + 1]) << 16);
        }
        if (row0 + r < nrows && row0 + r >= 0) *reinterpret_cast<u32x4*>(gbase + (long long)(row0 + r) * ld + ch * 8) = v;
    }
}

// HD = 64 (PointBERT blocks): the wave's 32 x 64 block through a 144-B-pitch strip, whole 128-B rows out, 16 B per lane; no RoPE
__device__ __forceinline__ void store_rows_via_lds64(char* wbuf, const f32x16 (&acc)[2], const float mul, bf16_t* gbase, const long long ld,
                                                     const int row0, const int nrows, const int lane) {
    const int half = lane >> 5, rl = lane & 31;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 w;
            w[0] = (uint32_t)f2bf(acc[dt][4 * g] * mul) | ((uint32_t)f2bf(acc[dt][4 * g + 1] * mul) << 16);
            w[1] = (uint32_t)f2bf(acc[dt][4 * g + 2] * mul) | ((uint32_t)f2bf(acc[dt][4 * g + 3] * mul) << 16);
            *reinterpret_cast<u32x2*>(wbuf + rl * 144 + (32 * dt + 8 * g + 4 * half) * 2) = w;
        }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), ch = lane & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(wbuf + r * 144 + ch * 16);
        if (row0 + r < nrows) *reinterpret_cast<u32x4*>(gbase + (long long)(row0 + r) * ld + ch * 8) = v;
    }
}
template <int HD>
__device__ __forceinline__ void store_rows_t(char* wbuf, const f32x16 (&acc)[HD / 32], const float mul, bf16_t* gbase, const long long ld,
                                             const int row0, const int nrows, const int lane, const float* rcos = nullptr, const float* rsin = nullptr) {
    if constexpr (HD == 128) store_rows_via_lds(wbuf, acc, mul, gbase, ld, row0, nrows, lane, rcos, rsin);
    else store_rows_via_lds64(wbuf, acc, mul, gbase, ld, row0, nrows, lane);
}

#ifdef ATTN_STAMP
// Timing stamps (debug builds only, -DATTN_STAMP; tools/debug/attn_stamp.py): wave 0 of every block accumulates s_memtime
// deltas per segment of the forward loop and adds them to g_attn_stamp[] at the end.
__device__ unsigned long long g_attn_stamp[16];
extern "C" int egomi_attn_stamp_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamp), sizeof(g_attn_stamp)); }
extern "C" int egomi_attn_stamp_reset() { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp), z, sizeof(z)); }
#define STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_prev = 0, st_t0 = 0; (void)st_t0;
#define STAMP_NOW(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP_START { STAMP_NOW(st_prev) st_t0 = st_prev; }
#define STAMP(i) { unsigned long long st_n; STAMP_NOW(st_n) st_acc[i] += st_n - st_prev; st_prev = st_n; }
#ifndef ATTN_STAMP_WAVE
#define ATTN_STAMP_WAVE 0
#endif
#define STAMP_FLUSH if (threadIdx.x == 64 * ATTN_STAMP_WAVE) { for (int i = 0; i < 8; ++i) atomicAdd(&g_attn_stamp[i], st_acc[i]); atomicAdd(&g_attn_stamp[8], 1ull); }
#else
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#define STAMP_FLUSH
#endif

// Block order of the three kernels (1-D grid of nblk x H x B): rank-major — every (b, h) pair's LONGEST causal block first,
// then every pair's second longest, ... (longest-processing-time dispatch: the blocks still running at the end of the launch
// are the shortest ones).  H*B % 8 == 0 keeps a pair on one XCD for all its blocks (block index mod 8 picks the XCD), so the
// K/V rows (resp. Q/dO rows) its blocks share are fetched into one L2, not eight.  (Pair-major order inside each XCD — a pair's
// blocks back to back, for L2 hits on its K/V tiles — measured slower: fwd 86 vs 79 us, bwd 236 vs 220: the tail decides.)
__device__ __forceinline__ void attn_block_map(const AttnArgs& a, int& rank, int& h, int& b) {
    const int pairs = a.H * a.B;
    rank = blockIdx.x / pairs;
    const int pair = blockIdx.x - rank * pairs;
    b = pair / a.H; h = pair - b * a.H;
    // the divisions run on the VALU: hand the (uniform) results back to the scalar unit, so that every base pointer derived from them is
    // SGPR arithmetic instead of 64-bit VGPR pairs (the kernels here sit at the 256-register line)
    rank = __builtin_amdgcn_readfirstlane(rank); b = __builtin_amdgcn_readfirstlane(b); h = __builtin_amdgcn_readfirstlane(h);
}
// Key-padding mask -> LDS bytes (1 = visible).  The global loads are unconditional (index clamped) and issued together by
// mask_fetch(); mask_commit() writes them to LDS later, after the block's other loads have been issued — a load under a
// per-element condition makes hipcc branch and wait vmcnt(0) per element (cdna_hip_programming.md §5, ".s-level traps" (c)).
#define AT_MASK_IT 4                                                   // x 256 threads = 1024 keys in registers; longer rows loop
__device__ __forceinline__ void mask_fetch(const AttnArgs& a, long long row_base, int nkeys, uint8_t (&mv)[AT_MASK_IT]) {
#pragma unroll
    for (int i = 0; i < AT_MASK_IT; ++i) {
        int j = threadIdx.x + 256 * i;
        j = j < a.S ? j : a.S - 1;
        mv[i] = (a.key_mask && 256 * i < nkeys) ? a.key_mask[row_base + j] : (uint8_t)1;      // block-uniform conditions
    }
}
__device__ __forceinline__ void mask_commit(const AttnArgs& a, long long row_base, int nkeys, const uint8_t (&mv)[AT_MASK_IT], char* sMask) {
#pragma unroll
    for (int i = 0; i < AT_MASK_IT; ++i) {
        const int j = threadIdx.x + 256 * i;
        if (j < nkeys) sMask[j] = j < a.S && mv[i] != 0;
    }
    for (int j = threadIdx.x + 256 * AT_MASK_IT; j < nkeys; j += 256) {
        uint8_t ok = j < a.S;
        if (ok && a.key_mask) ok = a.key_mask[row_base + j] != 0;
        sMask[j] = ok;
    }
}
// =================================================================================================
// forward: grid ceil(S/128) * H * B (attn_block_map), 4 waves x 32 queries, KV tiles of 64 keys, K/V double-buffered
// in LDS by LDS-DMA (one tile in flight across the barrier: counted vmcnt + raw s_barrier)
// =================================================================================================
#define AT_MAXS 4096
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
    constexpr int TB = 64 * 2 * HD;                                    // bytes of one 64-key K or V tile
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][K 16K | V 16K] + key mask bytes
    char* sMask = smem + 2 * 2 * TB;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rank, h, b;
    attn_block_map(a, rank, h, b);
    const int q0 = ((a.S + 127) / 128 - 1 - rank) * 128;             // causal: longest blocks first
    const bool wave_dead = q0 + wave * 32 >= a.S;                     // ragged last block: no query in this wave (it still moves its DMA share)
    const long long row_base = (long long)b * a.S;
    const bf16_t* Q = a.q + row_base * a.ld_qkv + h * HD;
    const bf16_t* K = a.k + row_base * a.ld_qkv + h * HD;
    const bf16_t* V = a.v + row_base * a.ld_qkv + h * HD;
    const int qi = q0 + wave * 32 + (lane & 31);                      // this lane's query
    const int qr = qi < a.S ? qi : a.S - 1;

    int last = q0 + 127 < a.S - 1 ? q0 + 127 : a.S - 1;
    const int ntiles = a.causal ? (last / 64 + 1) : ((a.S + 63) / 64);
    STAMP_DECL
    STAMP_START
    tile_dma_t<64, HD>(K, a.ld_qkv, 0, a.S - 1, smem, wave, lane);
    tile_dma_t<64, HD>(V, a.ld_qkv, 0, a.S - 1, smem + TB, wave, lane);
    bf16x8 qf[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (long long)qr * a.ld_qkv + 16 * ks + 8 * half);
    uint8_t mv[AT_MASK_IT];
    mask_fetch(a, row_base, ntiles * 64, mv);
    // The Q fragments are ordinary loads: consume them HERE (an empty asm that names them), so that the compiler's wait for
    // them — vmcnt(0), it does not count across LDS-DMAs — lands before the loop instead of in front of every tile's first MFMA,
    // where it drained the K/V prefetch of the next tile (cdna_hip_programming.md §5 "three .s-level traps", (b))
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) asm volatile("" :: "v"(qf[ks]));
    mask_commit(a, row_base, ntiles * 64, mv, sMask);                  // visible to the block after the loop's first barrier

    f32x16 o[HD / 32];
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = a.scale * 1.4426950408889634f;

    STAMP(0)                                                           // prologue
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
        char* sK = smem + (t & 1) * (2 * TB);
        char* sV = sK + TB;
        if (t + 1 < ntiles) {
            char* nK = smem + ((t + 1) & 1) * (2 * TB);
            tile_dma_t<64, HD>(K, a.ld_qkv, kv0 + 64, a.S - 1, nK, wave, lane);
            tile_dma_t<64, HD>(V, a.ld_qkv, kv0 + 64, a.S - 1, nK + TB, wave, lane);
            if (HD == 128) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");           // tile t landed; tile t+1 (8 DMAs) stays in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        STAMP(1)                                                       // DMA issue + wait for tile t
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        STAMP(2)                                                       // barrier A
        const unsigned long long kmask = __ballot(sMask[kv0 + lane] != 0);

        // causal: a 32-key sub-tile that starts beyond the wave's last query is masked for every lane -> nothing to do for it
        const int wave_qmax = q0 + wave * 32 + 31;
        const bool live0 = !wave_dead && (!a.causal || kv0 <= wave_qmax);                  // wave-uniform; live1 implies live0
        const bool live1 = !wave_dead && (!a.causal || kv0 + 32 <= wave_qmax);
        if (live0) {
        // Order inside a tile-step: both score chains are ISSUED before anything reads a score (the max of sub-tile 0 runs
        // under the MFMAs of sub-tile 1), and the exponentials of sub-tile 1 come after the P.V MFMAs of sub-tile 0 were
        // issued — MFMAs execute asynchronously, so independent VALU work placed behind them in program order overlaps them.
        f32x16 x[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) x[sub][r] = 0.f;
            if (sub == 1 && !live1) continue;
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks)
                x[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row8_t<HD>(sK, 32 * sub + (lane & 31), 2 * ks + half), qf[ks], x[sub], 0, 0, 0);
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (sub == 1 && !live1) continue;
            // raw scores stay in x[][] (the scale is folded into the exp); masks only where the sub-tile needs them
            const bool interior = ((kmask >> (32 * sub)) & 0xFFFFFFFFull) == 0xFFFFFFFFull &&
                                  (!a.causal || kv0 + 32 * sub + 31 <= q0 + wave * 32);       // wave-uniform
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, x[sub][r]);
            } else {
                asm volatile("");                                          // keeps the two paths apart (otherwise merged into selects)
                const uint32_t vis = visible_bits((uint32_t)(kmask >> (32 * sub)), half, kv0 + 32 * sub, qi, a.causal);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (vis >> rowmap(r, 0)) & 1u ? x[sub][r] : -INFINITY;
                    x[sub][r] = v;
                    mloc = fmaxf(mloc, v);
                }
            }
        }
        STAMP(3)                                                       // QK + masks
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * sc2;            // sc2 > 0: max commutes with the scale
        const float m_new = fmaxf(m_run, mloc);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = m_run == -INFINITY ? 0.f : fast_exp2(m_run - m_safe);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = fast_exp2(fmaf(x[0][r], sc2, -m_safe));   // exp2(-inf) = 0 for masked keys
            x[0][r] = p;
            lsum += p;
        }
        if (!__all(alpha == 1.0f)) {                                   // running max unchanged for the whole wave: O keeps its scale
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (sub == 1) {
                if (!live1) continue;                                  // P is all zero there
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2(fmaf(x[1][r], sc2, -m_safe));
                    x[1][r] = p;
                    lsum += p;
                }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float pv8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv8[j] = x[sub][8 * st + j];
                const bf16x8 pb = pack8(pv8);
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr8_t<HD>(sV, 32 * sub + 16 * st, 32 * dt, lane), pb, o[dt], 0, 0, 0);
            }
        }
        STAMP(4)                                                       // softmax + PV
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        STAMP(5)                                                       // PV issue
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // buffer (t&1) is free for the DMA of tile t+2
        STAMP(6)                                                       // barrier B
    }
    if (HD == 128) {
        if constexpr (HD == 128) {
            // whole 256-B rows through the (now idle) stages; the loop ended on a barrier with no DMA outstanding
            const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
            store_rows_via_lds(smem + wave * AT_XBYTES, o, inv, a.o + row_base * a.ld_o + h * HD, a.ld_o, q0 + wave * 32, a.S, lane);
        }
        if (qi < a.S && half == 0 && a.lse)
            a.lse[((long long)b * a.H + h) * a.S + qi] = (m_run == -INFINITY) ? INFINITY : m_run * 0.6931471805599453f + logf(l_run);
        STAMP(7)                                                       // epilogue
        STAMP_FLUSH
    } else if (qi < a.S) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16_t* orow = a.o + (row_base + qi) * a.ld_o + h * HD;
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = (uint32_t)f2bf(o[dt][4 * g] * inv) | ((uint32_t)f2bf(o[dt][4 * g + 1] * inv) << 16);
                w[1] = (uint32_t)f2bf(o[dt][4 * g + 2] * inv) | ((uint32_t)f2bf(o[dt][4 * g + 3] * inv) << 16);
                *reinterpret_cast<u32x2*>(orow + 32 * dt + 8 * g + 4 * half) = w;
            }
        if (half == 0 && a.lse)                                            // natural-log LSE of the SCALED scores
            a.lse[((long long)b * a.H + h) * a.S + qi] = (m_run == -INFINITY) ? INFINITY : m_run * 0.6931471805599453f + logf(l_run);
    }
}


// =================================================================================================
// forward, second form (round 3; the default — EGOMI_ATTN_FWD=1 selects the kernel above for A/B runs).  Same tiling, same arithmetic
// in the same order (results are bit-identical to attn_fwd_kernel), different instruction stream.  What the .s of the first form showed
// (hipcc, gfx950) and what changes here:
//   * ~150 v_mov per tile-step: the per-sub-tile `continue`s / interior-vs-edge diamonds made the register allocator copy the 32x32
//     accumulators around the joins.  Here a wave walks the block's tile loop in THREE consecutive straight-line loops — interior tiles
//     (no mask at all), edge tiles (key-padding / causal bits applied to both sub-tiles, a dead sub-tile is simply fully masked), dead
//     tiles (DMA share and barriers only) — every wave still executes the same number of barriers.
//   * ~64 VALU per tile-step of 64-bit address arithmetic for the 8 LDS-DMAs: per-lane global pointers are set up once and advanced
//     by one v_lshl_add_u64 each; only the block's last, ragged tile takes the clamped slow path.
//   * `s_waitcnt vmcnt(0)` in front of the first V^T fragment read (the ds_read_b64_tr_b16 builtin carries no alias information), i.e.
//     the K/V prefetch of tile t+1 had to land before P.V of tile t: the transposed reads are inline asm here, issued BEFORE the softmax
//     arithmetic of the sub-tile they belong to (their LDS latency hides under it), waited for with an explicit lgkmcnt.
//   * K fragments: all 16 ds_read_b128 of a tile-step are issued before the first MFMA (the compiler had paired them with lgkmcnt(0)).
// =================================================================================================
// LDS-DMA issued from inline asm: the compiler then knows of no asynchronous LDS writer, so it neither waits vmcnt(0) in front of the
// ds_read_b64_tr_b16 builtin nor copies fragments around; every LDS hand-over in this kernel is explicit anyway (counted vmcnt + s_barrier).
// lds_base: wave-uniform LDS byte address of the 1-KiB piece this instruction fills (lane i writes 16 B at lds_base + 16 i).
// Source = wave-uniform base pointer (SGPR pair) + per-lane 32-bit byte offset: one VGPR per DMA, shared by the K and the V tile.
__device__ __forceinline__ void dma16_asm(const void* gbase, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(gbase), "s"(lds_base) : "memory");     // (m0 is a reserved register: hipcc re-loads it in front of each of its own uses)
}

template <int HD, bool MASKED>
__device__ __forceinline__ void fwd2_tile(const char* sK, const char* sV, const bf16x8 (&qf)[HD / 16],
                                          f32x16 (&o)[HD / 32], float& m_run, float& l_run, const float sc2, const int lane, const int half,
                                          const uint32_t vis0, const uint32_t vis1) {
    constexpr int NK = HD / 16, ND = HD / 32;
    // ---- scores: the 8 K fragments of sub-tile 0 are in flight before the first MFMA, those of sub-tile 1 are issued under sub-tile 0's chain
    // (hipcc pairs each read with its MFMA and waits lgkmcnt(0) in between unless fenced: NK/2 reads stay in flight ahead of the chain)
    constexpr int HK = NK / 2;
    bf16x8 ka[HK], kb[HK];
    f32x16 x0, x1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x0[r] = 0.f; x1[r] = 0.f; }
#pragma unroll
    for (int i = 0; i < HK; ++i) ka[i] = lds_row8_t<HD>(sK, lane & 31, 2 * i + half);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < HK; ++i) {
        kb[i] = lds_row8_t<HD>(sK, lane & 31, 2 * (HK + i) + half);
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[i], qf[i], x0, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < HK; ++i) {
        ka[i] = lds_row8_t<HD>(sK, 32 + (lane & 31), 2 * i + half);
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[i], qf[HK + i], x0, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < HK; ++i) {
        kb[i] = lds_row8_t<HD>(sK, 32 + (lane & 31), 2 * (HK + i) + half);
        x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[i], qf[i], x1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- V^T fragments of sub-tile 0 (keys 0..31 of the tile) are issued under the end of sub-tile 1's chain, consumed after the softmax arithmetic
    bf16x8 vf[2 * ND];
#pragma unroll
    for (int i = 0; i < HK; ++i) {
#pragma unroll
        for (int j = i * (2 * ND) / HK; j < (i + 1) * (2 * ND) / HK; ++j) vf[j] = lds_tr8_t<HD>(sV, 16 * (j / ND), 32 * (j % ND), lane);
        x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[i], qf[HK + i], x1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- online softmax (raw scores stay in x; the scale is folded into the exp)
    if (MASKED) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x0[r] = (vis0 >> rowmap(r, 0)) & 1u ? x0[r] : -INFINITY;
            x1[r] = (vis1 >> rowmap(r, 0)) & 1u ? x1[r] : -INFINITY;
        }
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, x0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, x1[r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * sc2;                  // sc2 > 0: max commutes with the scale
    const float m_new = fmaxf(m_run, mloc);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = m_run == -INFINITY ? 0.f : fast_exp2(m_run - m_safe);
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(x0[r], sc2, -m_safe));           // exp2(-inf) = 0 for masked keys
        x0[r] = p;
        lsum += p;
    }
    // O <- alpha * O, unconditionally (32 packed multiplies; the first form's "skip when alpha == 1 for the whole wave" cost a copy of the
    // 64 accumulator registers at its join)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    // ---- P.V of sub-tile 0, then sub-tile 1 (its fragments are read while sub-tile 0's MFMAs run)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float pv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv8[j] = x0[8 * st + j];
        const bf16x8 pb = pack8(pv8);
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st * ND + dt], pb, o[dt], 0, 0, 0);
    }
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) vf[st * ND + dt] = lds_tr8_t<HD>(sV, 32 + 16 * st, 32 * dt, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(x1[r], sc2, -m_safe));
        x1[r] = p;
        lsum += p;
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float pv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv8[j] = x1[8 * st + j];
        const bf16x8 pb = pack8(pv8);
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st * ND + dt], pb, o[dt], 0, 0, 0);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    l_run = l_run * alpha + lsum;
    m_run = m_new;
}

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(AttnArgs a) {
    constexpr int TB = 64 * 2 * HD;                                    // bytes of one 64-key K or V tile
    constexpr int RPI = 512 / HD, CPR = HD / 8, IPW = 64 / RPI / 4;    // rows per DMA instruction, 16-B chunks per row, DMAs per wave and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][K | V] + key mask bytes
    char* sMask = smem + 2 * 2 * TB;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rank, h, b;
    attn_block_map(a, rank, h, b);
    const int q0 = ((a.S + 127) / 128 - 1 - rank) * 128;             // causal: longest blocks first
    const bool wave_dead = q0 + wave * 32 >= a.S;
    const long long row_base = (long long)b * a.S;
    const bf16_t* Q = a.q + row_base * a.ld_qkv + h * HD;
    const bf16_t* K = a.k + row_base * a.ld_qkv + h * HD;
    const bf16_t* V = a.v + row_base * a.ld_qkv + h * HD;
    const int qi = q0 + wave * 32 + (lane & 31);
    const int qr = qi < a.S ? qi : a.S - 1;
    int last = q0 + 127 < a.S - 1 ? q0 + 127 : a.S - 1;
    const int ntiles = a.causal ? (last / 64 + 1) : ((a.S + 63) / 64);
    const bool ragged = ntiles * 64 > a.S;                            // the last tile has rows beyond S-1: clamped slow path for that one
    // per-lane DMA source offsets (bytes from K resp. V, < 2^32: S <= 4096 rows) of tile 0, rows clamped once here; advanced by 64 rows per tile
    uint32_t koff[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int rl = (wave * IPW + j) * RPI + lane / CPR;
        const int ch = (lane % CPR) ^ swz<HD>(rl);
        const int r = rl < a.S ? rl : a.S - 1;
        koff[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
    }
    const uint32_t tile_stride = (uint32_t)(64 * a.ld_qkv * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_void_t*)smem) + wave * IPW * 1024;      // this wave's first 1-KiB piece of stage 0's K tile
    auto dma_fast = [&](int stage) {
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds0 + stage * (2 * TB));
#pragma unroll
        for (int j = 0; j < IPW; ++j) dma16_asm(K, koff[j], base + j * 1024);
#pragma unroll
        for (int j = 0; j < IPW; ++j) dma16_asm(V, koff[j], base + TB + j * 1024);
    };
    auto dma_slow = [&](int stage, int row0) {                         // rows clamped to S-1 (the block's ragged last tile)
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds0 + stage * (2 * TB));
        uint32_t off[IPW];
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int rl = (wave * IPW + j) * RPI + lane / CPR;
            const int ch = (lane % CPR) ^ swz<HD>(rl);
            int r = row0 + rl;
            r = r < a.S ? r : a.S - 1;
            off[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
        }
#pragma unroll
        for (int j = 0; j < IPW; ++j) dma16_asm(K, off[j], base + j * 1024);
#pragma unroll
        for (int j = 0; j < IPW; ++j) dma16_asm(V, off[j], base + TB + j * 1024);
    };
    dma_fast(0);                                                       // tile 0 (its row clamp is in the offsets)
#pragma unroll
    for (int j = 0; j < IPW; ++j) koff[j] += tile_stride;
    bf16x8 qf[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (long long)qr * a.ld_qkv + 16 * ks + 8 * half);
    uint8_t mv[AT_MASK_IT];
    mask_fetch(a, row_base, ntiles * 64, mv);
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) asm volatile("" :: "v"(qf[ks]));     // consumed before the loop: see attn_fwd_kernel
    mask_commit(a, row_base, ntiles * 64, mv, sMask);

    f32x16 o[HD / 32];
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = a.scale * 1.4426950408889634f;

    // tile classes of this wave along the block's loop: [0, n_int) interior, [n_int, n_live) edge, [n_live, ntiles) dead
    const int wave_q0 = q0 + wave * 32;
    int n_live = wave_dead ? 0 : (a.causal ? (wave_q0 + 31) / 64 + 1 : ntiles);
    n_live = n_live < ntiles ? n_live : ntiles;
    int n_int = a.causal ? (wave_q0 >= 63 ? (wave_q0 - 63) / 64 + 1 : 0) : ntiles;      // tiles with kv0 + 63 <= wave_q0
    n_int = n_int < n_live ? n_int : n_live;
    if (a.S % 64 && n_int == ntiles) n_int = ntiles - 1;               // the ragged last tile has keys >= S: edge path

    auto top_of_tile = [&](int t) {                                    // prefetch tile t+1, wait for tile t, meet the block
        if (t + 1 < ntiles) {
            if (ragged && t + 2 == ntiles) {
                dma_slow((t + 1) & 1, (t + 1) * 64);
            } else {
                dma_fast((t + 1) & 1);
#pragma unroll
                for (int j = 0; j < IPW; ++j) koff[j] += tile_stride;
            }
            if (HD == 128) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto end_of_tile = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // buffer (t&1) is free for the DMA of tile t+2
    };
    int t = 0;
    bool carried = false;                                              // tile t has been opened (top_of_tile, kmask) by the interior loop
    unsigned long long kmask = 0ull;
    for (; t < n_int; ++t) {                                           // interior tiles: no mask anywhere in the body
        top_of_tile(t);
        kmask = __ballot(sMask[t * 64 + lane] != 0);
        if (kmask != ~0ull) { carried = true; break; }                 // a padded key: this wave continues on the edge path
        const char* sK = smem + (t & 1) * (2 * TB);
        fwd2_tile<HD, false>(sK, sK + TB, qf, o, m_run, l_run, sc2, lane, half, 0u, 0u);
        end_of_tile();
    }
    for (; t < n_live; ++t) {                                          // edge tiles: key-padding and causal bits on both sub-tiles
        if (!carried) {
            top_of_tile(t);
            kmask = __ballot(sMask[t * 64 + lane] != 0);
        }
        carried = false;
        const int kv0 = t * 64;
        const char* sK = smem + (t & 1) * (2 * TB);
        const uint32_t v0 = visible_bits((uint32_t)kmask, half, kv0, qi, a.causal), v1 = visible_bits((uint32_t)(kmask >> 32), half, kv0 + 32, qi, a.causal);
        fwd2_tile<HD, true>(sK, sK + TB, qf, o, m_run, l_run, sc2, lane, half, v0, v1);
        end_of_tile();
    }
    for (; t < ntiles; ++t) {                                          // dead tiles of this wave: its DMA share and the barriers
        top_of_tile(t);
        end_of_tile();
    }
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    if constexpr (HD == 128) {
        store_rows_via_lds(smem + wave * AT_XBYTES, o, inv, a.o + row_base * a.ld_o + h * HD, a.ld_o, q0 + wave * 32, a.S, lane);
    } else {
        if (qi < a.S) {
            bf16_t* orow = a.o + (row_base + qi) * a.ld_o + h * HD;
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w;
                    w[0] = (uint32_t)f2bf(o[dt][4 * g] * inv) | ((uint32_t)f2bf(o[dt][4 * g + 1] * inv) << 16);
                    w[1] = (uint32_t)f2bf(o[dt][4 * g + 2] * inv) | ((uint32_t)f2bf(o[dt][4 * g + 3] * inv) << 16);
                    *reinterpret_cast<u32x2*>(orow + 32 * dt + 8 * g + 4 * half) = w;
                }
        }
    }
    if (qi < a.S && half == 0 && a.lse)                                // natural-log LSE of the SCALED scores
        a.lse[((long long)b * a.H + h) * a.S + qi] = (m_run == -INFINITY) ? INFINITY : m_run * 0.6931471805599453f + logf(l_run);
}


// =================================================================================================
// forward, third form (round 4; the default at head_dim 128 — EGOMI_ATTN_FWD=2 / =1 select the forms above for A/B runs).
// The second form's counters said: VALU 44 % / MFMA pipe 19 % of SIMD time, 37 % of a block's life in prologue + epilogue, two barriers
// per 64-key tile, a wave's QK -> softmax -> PV phases strictly one after the other.  What changes:
//   * software pipeline inside the wave: the scores of tile t+1 (8 MFMAs) are issued under the exponentials of tile t, the mask + row max
//     of tile t+1 under the P.V MFMAs of tile t — each phase has work for both pipes of the SIMD;
//   * 32-key tiles in a 4-stage LDS ring (4 x (8 KB K + 8 KB V) = the same 64 KB): tile t+3 is requested at the top of tile t, ONE
//     barrier per tile (the wave's own `vmcnt` for tile t+1 in front of it), nothing waits for a DMA issued less than two tiles ago;
//   * lazy running max (cdna_hip_programming.md T13): the 64 accumulator multiplies and the alpha exponential run only when some query
//     of the wave meets a score more than 2^F3_THR above its reference maximum; l and O stay relative to that reference, LSE is exact
//     (m ln2 + ln l).  P <= 2^F3_THR in bf16: same relative precision, results differ from the earlier forms in the last bits;
//   * query blocks aligned to the END of the sequence (rounded up to 32): the ragged block is the FIRST one (shortest causal range)
//     instead of the last (S = 692: 36 instead of 41 64-key tile-steps per (b, h) pair, and no block runs 11 tiles with half its waves dead);
//   * the row sum is kept per lane (each half-wave sums its own keys) and combined once at the end.
// =================================================================================================
// max of a value with its partner lane (l <-> l ^ 32): v_permlane32_swap needs no lane-index register (a __shfl_xor keeps one alive across the
// whole tile loop and goes through the LDS crossbar)
// max of three in ONE instruction: fmaxf on MFMA outputs makes hipcc canonicalise each operand first (v_max_f32 x, x, x: 16 extra VALU per 32-key
// tile in the forward's row-max chain; MI355X_MICROARCH.md "canonicalising v_max").  The scores are never signalling NaNs, so the bare v_max3 is exact.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float row_max16(const f32x16& x) {
    float m = max3_raw(x[0], x[1], x[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) m = max3_raw(m, x[r], x[r + 1]);
    return max3_raw(m, x[15], x[15]);
}
__device__ __forceinline__ float half_swap_max(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
#define F3_KT 32
#define F3_NST 4
#define F3_TB (F3_KT * 256)                                            // bytes of one 32-key K or V tile (head_dim 128)
#define F3_STAGE (2 * F3_TB)
#define F3_THR 6.0f

// one pipelined step: exponentials of tile t (scores in x, reference max m_run) beside the scores of tile t+1 (-> xn), then P.V of tile t
// beside the mask + row max of tile t+1; the reference max moves only in the rare branch at the end.  HAS_NEXT = false: the wave's last tile.
// masked_next (wave-uniform, runtime): tile t+1 has invisible keys for some lane (diagonal, ragged, padded).  One instantiation serves both
// kinds of tile: with a fully visible and a masked copy of this body inside the persistent kernel's item loop the allocator spilled the Q
// fragments (reloaded — behind vmcnt(0) — in every tile step); the branch sits behind the P.V MFMAs, which run on while the VALU takes it.
template <bool HAS_NEXT>
__device__ __forceinline__ void f3_step(const bool masked_next, const char* sKn, const char* sV, const bf16x8 (&qf)[8], f32x16& x, f32x16 (&o)[4],
                                        float& m_run, float& l_run, const float sc2, const uint32_t vis_next, const int lane, const int half) {
    const float nm = m_run == -INFINITY ? 0.f : -m_run;
    f32x16 xn;
    bf16x8 kf[4];
    if (HAS_NEXT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xn[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) kf[i] = lds_row8(sKn, lane & 31, 2 * i + half);
    }
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (HAS_NEXT) {
            xn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i & 3], qf[i], xn, 0, 0, 0);
            if (i < 4) kf[i] = lds_row8(sKn, lane & 31, 2 * (i + 4) + half);      // second batch of K fragments into the registers just consumed
        }
#pragma unroll
        for (int r = 2 * i; r < 2 * i + 2; ++r) {
            const float p = fast_exp2(fmaf(x[r], sc2, nm));            // exp2(-inf) = 0 for masked keys
            x[r] = p;
            lsum += p;
        }
    }
    l_run += lsum;
    bf16x8 vf[4];                                                      // V^T fragments of the tile's first 16 keys; the second 16 follow into the same registers
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vf[dt] = lds_tr8(sV, 0, 32 * dt, lane);
    float pv8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pv8[j] = x[j];
    const bf16x8 pb0 = pack8(pv8);
#pragma unroll
    for (int j = 0; j < 8; ++j) pv8[j] = x[8 + j];
    const bf16x8 pb1 = pack8(pv8);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt], pb0, o[dt], 0, 0, 0);
        vf[dt] = lds_tr8(sV, 16, 32 * dt, lane);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt], pb1, o[dt], 0, 0, 0);
    if (HAS_NEXT) {
        if (masked_next) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xn[r] = (vis_next >> rowmap(r, 0)) & 1u ? xn[r] : -INFINITY;
        }
        float mloc = row_max16(xn);
        mloc = half_swap_max(mloc) * sc2;                                // sc2 > 0: max commutes with the scale
        if (__any(mloc > m_run + F3_THR)) {                              // rare after the first tiles (always at the first one: m_run = -inf)
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = m_run == -INFINITY ? 0.f : fast_exp2(m_run - m_new);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            l_run *= alpha;
            m_run = m_new;
        }
        x = xn;
    }
}

// Block order of the third form.  group == 0: rank-major as attn_block_map.  group = G > 0 (needs H*B % 8 == 0): blocks that share an XCD
// (equal blockIdx % 8) walk that XCD's (b, h) pairs in GROUPS of G consecutive ranks — the G blocks of a pair are adjacent in the dispatch
// order, so they run at the same time on one XCD and the K/V tiles the first of them brings into that XCD's L2 serve the others; groups
// still go longest ranks first.  Grid = 8 * (pairs/8) * G * ceil(nblk/G); ranks >= nblk (nblk % G != 0) leave at once.
__device__ __forceinline__ bool attn_block_map3(const AttnArgs& a, const int nblk, const int group, int& rank, int& h, int& b) {
    if (group <= 0) { attn_block_map(a, rank, h, b); return true; }
    const int pairs = a.H * a.B, ppx = pairs >> 3;
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int per_phase = ppx * group;
    const int phase = k / per_phase, j = k - phase * per_phase;
    const int pl = j / group, rr = j - pl * group;
    const int pair = pl * 8 + x;
    rank = __builtin_amdgcn_readfirstlane(phase * group + rr);
    b = pair / a.H; h = pair - b * a.H;
    b = __builtin_amdgcn_readfirstlane(b); h = __builtin_amdgcn_readfirstlane(h);
    return rank < nblk;
}

__global__ __launch_bounds__(256, 2) void attn_fwd3_kernel(AttnArgs a, const int group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [4 stages][K 8 KB | V 8 KB] + key mask bytes
    char* sMask = smem + F3_NST * F3_STAGE;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rank, h, b;
    const int S32 = (a.S + 31) & ~31;
    if (!attn_block_map3(a, (S32 + 127) / 128, group, rank, h, b)) return;
    const int q0 = S32 - 128 * (rank + 1);                             // may be negative for the last rank (the shortest block)
    const long long row_base = (long long)b * a.S;
    const bf16_t* Q = a.q + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* K = a.k + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* V = a.v + row_base * a.ld_qkv + h * AT_HD;
    const int wave_q0 = q0 + wave * 32;                                // a multiple of 32
    const int qi = wave_q0 + (lane & 31);
    const int qr = qi < 0 ? 0 : (qi < a.S ? qi : a.S - 1);
    const int last = q0 + 127 < a.S - 1 ? q0 + 127 : a.S - 1;         // >= 0 for every launched block
    const int nkt = (a.S + F3_KT - 1) / F3_KT;
    const int ntiles = a.causal ? last / F3_KT + 1 : nkt;
    // per-lane DMA source offsets (bytes from K resp. V, < 2^32) of this wave's two 1-KiB pieces of tile 0; advanced by 32 rows per request
    uint32_t koff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rl = (wave * 2 + j) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
        const int r = rl < a.S ? rl : a.S - 1;
        koff[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
    }
    const uint32_t tile_stride = (uint32_t)(F3_KT * a.ld_qkv * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_void_t*)smem) + wave * 2 * 1024;
    int next_req = 0;                                                  // tiles are requested in order 0, 1, 2, ...
    auto request = [&]() {                                             // K and V rows of tile next_req -> stage next_req % 4
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds0 + (next_req & (F3_NST - 1)) * F3_STAGE);
        if ((next_req + 1) * F3_KT <= a.S) {
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(K, koff[j], base + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(V, koff[j], base + F3_TB + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) koff[j] += tile_stride;
        } else {                                                       // the sequence's ragged last tile: rows clamped to S - 1
            uint32_t off[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rl = (wave * 2 + j) * 4 + (lane >> 4);
                const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
                int r = next_req * F3_KT + rl;
                r = r < a.S ? r : a.S - 1;
                off[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(K, off[j], base + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(V, off[j], base + F3_TB + j * 1024);
        }
        ++next_req;
    };
    STAMP_DECL
    STAMP_START
    request();                                                         // tile 0
    if (ntiles > 1) request();                                         // tile 1
    STAMP(0)
    bf16x8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (long long)qr * a.ld_qkv + 16 * ks + 8 * half);
    uint8_t mv[AT_MASK_IT];
    mask_fetch(a, row_base, ntiles * F3_KT, mv);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" :: "v"(qf[ks]));            // consumed before the loop: see attn_fwd_kernel (the compiler's wait also lands tiles 0, 1)
    STAMP(1)
    mask_commit(a, row_base, ntiles * F3_KT, mv, sMask);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ntiles > 2) request();                                         // tile 2 stays in flight across the barrier

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = a.scale * 1.4426950408889634f;

    // tile classes of this wave: [0, n_int) fully visible, [n_int, n_live) masked (diagonal, ragged, padded), [n_live, ntiles) not its own
    const int wl = wave_q0 + 31 < a.S - 1 ? wave_q0 + 31 : a.S - 1;   // the wave's last live query (< 0: none)
    const int wf = wave_q0 < 0 ? 0 : wave_q0;
    int n_live = wl < 0 ? 0 : (a.causal ? wl / F3_KT + 1 : ntiles);
    n_live = n_live < ntiles ? n_live : ntiles;
    int n_int = a.causal ? (wf >= F3_KT - 1 ? (wf - (F3_KT - 1)) / F3_KT + 1 : 0) : ntiles;
    n_int = n_int < n_live ? n_int : n_live;
    if (a.S % F3_KT && n_int == nkt) n_int = nkt - 1;                  // the ragged last tile has keys >= S

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                      // tiles 0 and 1 and the mask bytes are in LDS for every wave
    STAMP(2)
    auto top = [&](int t) {                                            // tile t+1 landed for everybody, stage (t+3) % 4 free: request tile t+3
        if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        STAMP(4)
        if (t + 3 < ntiles) request();
        STAMP(6)
    };
    auto kbits = [&](int t) -> uint32_t { return (uint32_t)__ballot(sMask[t * F3_KT + (lane & 31)] != 0); };
    f32x16 x;
    if (n_live > 0) {                                                  // scores and reference max of tile 0
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = 0.f;
        bf16x8 kf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[i] = lds_row8(smem, lane & 31, 2 * i + half);
#pragma unroll
        for (int i = 0; i < 8; ++i) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[i], x, 0, 0, 0);
        const uint32_t v0 = visible_bits(kbits(0), half, 0, qi, a.causal);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = (v0 >> rowmap(r, 0)) & 1u ? x[r] : -INFINITY;
        m_run = half_swap_max(row_max16(x)) * sc2;
    }
    STAMP(3)
    int t = 0;
    for (; t < n_live - 1; ++t) {                                      // tile t+1: fully visible (no mask work) or diagonal / ragged / padded
        top(t);
        const uint32_t km = kbits(t + 1);
        const bool masked = t + 1 >= n_int || km != 0xFFFFFFFFu;
        const uint32_t vn = visible_bits(km, half, (t + 1) * F3_KT, qi, a.causal);
        f3_step<true>(masked, smem + ((t + 1) & 3) * F3_STAGE, smem + (t & 3) * F3_STAGE + F3_TB, qf, x, o, m_run, l_run, sc2, vn, lane, half);
        STAMP(5)
    }
    if (n_live > 0) {                                                  // t == n_live - 1: the wave's last tile
        top(t);
        f3_step<false>(false, smem, smem + (t & 3) * F3_STAGE + F3_TB, qf, x, o, m_run, l_run, sc2, 0u, lane, half);
        STAMP(5)
        ++t;
    }
    for (; t < ntiles; ++t) top(t);                                    // tiles of the block's later waves: DMA share and barriers
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                      // nobody reads a stage any more: the ring becomes the epilogue's strips
    STAMP(6)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    store_rows_via_lds(smem + wave * AT_XBYTES, o, inv, a.o + row_base * a.ld_o + h * AT_HD, a.ld_o, wave_q0, a.S, lane);
    if (qi >= 0 && qi < a.S && half == 0 && a.lse)                     // natural-log LSE of the SCALED scores
        a.lse[((long long)b * a.H + h) * a.S + qi] = (m_run == -INFINITY) ? INFINITY : m_run * 0.6931471805599453f + logf(l_tot);
    STAMP(7)
    STAMP_FLUSH
}


// =================================================================================================
// forward, fourth form (round 4; the default at head_dim 128): attn_fwd3_kernel made PERSISTENT.  What the third form's stamps and counters
// said (profiles/README.md, round 4): 37 % of a block's life is its prologue (first K/V tiles, Q fragments, mask) and 15 % its epilogue, and
// the kernel is NOT bound by traffic — with a (b, h) pair's blocks grouped on one XCD the fetch from beyond L2 falls to the algorithmic 133 MB
// and the launch gets slower (schedule tail).  So the fixed cost per block is what to remove: here 2 x 256 blocks stay resident and walk the
// work items (rank-major: longest causal ranges first; block i takes items i, i + grid, ...) as ONE continuous stream:
//   * the K/V ring never drains: the request that follows an item's last tile is the next item's tile 0 (its tiles 0..2 are in LDS or in
//     flight while the current item still computes);
//   * a wave loads the next item's Q fragments into the registers of the current ones as soon as its last score tile is issued (they land
//     under the last exponentials / P.V, the epilogue and the later waves' tiles);
//   * the next item's key-mask bytes are fetched at the start of the current item and committed to the other of two LDS mask buffers;
//   * O leaves through a wave-private strip that is NOT part of the ring (4 passes of 32 columns, 2.5 KB per wave), so the stores of item n
//     overlap the first tiles of item n+1.
// Arithmetic per item is attn_fwd3_kernel's (f3_step): the same bits.
// =================================================================================================
#define F4_SPITCH 80                                                   // strip row: 32 columns (64 B) + 16 B pad
#define F4_STRIP (32 * F4_SPITCH)

// O^T accumulators -> global rows, 32 columns per pass through the wave's strip
__device__ __forceinline__ void f4_store_rows(char* wbuf, const f32x16 (&acc)[4], const float mul, bf16_t* gbase, const long long ld,
                                              const int row0, const int nrows, const int lane) {
    const int half = lane >> 5, rl = lane & 31;
    const int r = lane >> 1, c0 = (lane & 1) * 2;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 w;
            w[0] = (uint32_t)f2bf(acc[dt][4 * g] * mul) | ((uint32_t)f2bf(acc[dt][4 * g + 1] * mul) << 16);
            w[1] = (uint32_t)f2bf(acc[dt][4 * g + 2] * mul) | ((uint32_t)f2bf(acc[dt][4 * g + 3] * mul) << 16);
            *reinterpret_cast<u32x2*>(wbuf + rl * F4_SPITCH + (8 * g + 4 * half) * 2) = w;
        }
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(wbuf + r * F4_SPITCH + c0 * 16);
        const u32x4 v1 = *reinterpret_cast<const u32x4*>(wbuf + r * F4_SPITCH + (c0 + 1) * 16);
        if (row0 + r < nrows && row0 + r >= 0) {
            bf16_t* dst = gbase + (long long)(row0 + r) * ld + 32 * dt + c0 * 8;
            *reinterpret_cast<u32x4*>(dst) = v0;
            *reinterpret_cast<u32x4*>(dst + 8) = v1;
        }
        __builtin_amdgcn_sched_barrier(0);                             // one pass at a time: scheduled together, the four passes' packed and read-back
    }                                                                  // registers pushed the next item's prefetched Q fragments into scratch
}

// key-mask bytes of one sample for the persistent kernel (S <= 1024): four UNCONDITIONAL byte loads per thread (index clamped), committed later;
// the branch around them is block-uniform.  (mask_fetch's per-element conditions made hipcc branch and wait vmcnt(0) per load once the
// loads sat inside the item loop: cdna_hip_programming.md §5 ".s-level traps" (c).)
__device__ __forceinline__ void f4_mask_fetch(const uint8_t* row, const int S, uint8_t (&mv)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int j = threadIdx.x + 256 * i;
        j = j < S ? j : S - 1;
        mv[i] = row[j];
    }
}
__device__ __forceinline__ void f4_mask_commit(const bool have, const int S, const int S32, const uint8_t (&mv)[4], char* sMask) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = threadIdx.x + 256 * i;
        if (j < S32) sMask[j] = j < S && (!have || mv[i] != 0);
    }
}

__global__ __launch_bounds__(256, 2) void attn_fwd4_kernel(AttnArgs a, const int n_items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [4 stages][K 8 KB | V 8 KB] | 4 strips | 2 key-mask buffers
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S32 = (a.S + 31) & ~31;
    char* strip = smem + F3_NST * F3_STAGE + wave * F4_STRIP;
    char* sMaskBase = smem + F3_NST * F3_STAGE + 4 * F4_STRIP;
    const int pairs = a.H * a.B;
    const int nkt = (a.S + F3_KT - 1) / F3_KT;
    const float sc2 = a.scale * 1.4426950408889634f;
    const uint32_t tile_stride = (uint32_t)(F3_KT * a.ld_qkv * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_void_t*)smem) + wave * 2 * 1024;

    // item -> (rank, b, h): rank-major, every pair's longest block first
    auto decode = [&](int it, int& rank, int& b, int& h) {
        rank = it / pairs;
        const int pair = it - rank * pairs;
        b = pair / a.H; h = pair - b * a.H;
        rank = __builtin_amdgcn_readfirstlane(rank); b = __builtin_amdgcn_readfirstlane(b); h = __builtin_amdgcn_readfirstlane(h);
    };
    auto item_tiles = [&](int rank) -> int {
        const int q0 = S32 - 128 * (rank + 1);
        const int last = q0 + 127 < a.S - 1 ? q0 + 127 : a.S - 1;
        return a.causal ? last / F3_KT + 1 : nkt;
    };

    // ---- request side: one continuous stream of K/V tiles over this block's items
    int rq_it = blockIdx.x, rq_tile = 0, rq_ntiles = 0, g_issued = 0;
    const bf16_t* Kq = a.k; const bf16_t* Vq = a.v;
    uint32_t koff[2];
    auto rq_open = [&]() {                                             // first tile of item rq_it
        int rank, b, h;
        decode(rq_it, rank, b, h);
        rq_ntiles = item_tiles(rank);
        rq_tile = 0;
        const long long base = (long long)b * a.S * a.ld_qkv + h * AT_HD;
        Kq = a.k + base; Vq = a.v + base;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = (wave * 2 + j) * 4 + (lane >> 4);
            const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
            const int r = rl < a.S ? rl : a.S - 1;
            koff[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
        }
    };
    auto request = [&]() {                                             // next tile of the stream -> stage g_issued % 4 (caller: stream not exhausted)
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds0 + (g_issued & (F3_NST - 1)) * F3_STAGE);
        if ((rq_tile + 1) * F3_KT <= a.S) {
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(Kq, koff[j], base + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(Vq, koff[j], base + F3_TB + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) koff[j] += tile_stride;
        } else {                                                       // the sequence's ragged last tile: rows clamped to S - 1
            uint32_t off[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rl = (wave * 2 + j) * 4 + (lane >> 4);
                const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
                int r = rq_tile * F3_KT + rl;
                r = r < a.S ? r : a.S - 1;
                off[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(Kq, off[j], base + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(Vq, off[j], base + F3_TB + j * 1024);
        }
        ++g_issued;
        if (++rq_tile == rq_ntiles) {                                  // the stream moves on to the block's next item
            rq_it += gridDim.x;
            if (rq_it < n_items) rq_open();
        }
    };
    // top of stream tile g: tile g+1 (if requested) has landed for everybody, the stage of tile g-1 is free: keep three tiles requested ahead
    int g = 0;
    auto top = [&]() {
        if (g_issued >= g + 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (rq_it < n_items && g_issued < g + 4) request();
        ++g;
    };

    // ---- first item: the ordinary prologue
    int it = blockIdx.x;
    if (it >= n_items) return;
    rq_open();
    request();
    if (rq_it < n_items) request();
    int rank, b, h;
    decode(it, rank, b, h);
    bf16x8 qf[8];
    {
        const int q0 = S32 - 128 * (rank + 1);
        const int qi = q0 + wave * 32 + (lane & 31);
        const int qr = qi < 0 ? 0 : (qi < a.S ? qi : a.S - 1);
        const bf16_t* Q = a.q + ((long long)b * a.S + qr) * a.ld_qkv + h * AT_HD;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + 16 * ks + 8 * half);
        uint8_t mv[4] = {1, 1, 1, 1};
        if (a.key_mask) f4_mask_fetch(a.key_mask + (long long)b * a.S, a.S, mv);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" :: "v"(qf[ks]));
        f4_mask_commit(a.key_mask != nullptr, a.S, S32, mv, sMaskBase);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (rq_it < n_items) request();                                // a third tile stays in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    int parity = 0;
    const int lane_entry = lane, half_entry = half;
    for (;;) {
        // ---- this item.  The lane index is made opaque per item: everything derived from it (the swizzled LDS read offsets of every
        // instantiation of f3_step below) would otherwise be hoisted out of the item loop and live — spilled — across all of it
        // (cdna_hip_programming.md, persistent-attention pitfalls: recompute lane-constant addresses per block)
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        const int half = lane >> 5;
        (void)half_entry;
        const int q0 = S32 - 128 * (rank + 1);
        const long long row_base = (long long)b * a.S;
        const int wave_q0 = q0 + wave * 32;
        const int qi = wave_q0 + (lane & 31);
        const int ntiles = item_tiles(rank);
        const char* sMask = sMaskBase + parity * S32;
        const int wl = wave_q0 + 31 < a.S - 1 ? wave_q0 + 31 : a.S - 1;
        const int wf = wave_q0 < 0 ? 0 : wave_q0;
        int n_live = wl < 0 ? 0 : (a.causal ? wl / F3_KT + 1 : ntiles);
        n_live = n_live < ntiles ? n_live : ntiles;
        int n_int = a.causal ? (wf >= F3_KT - 1 ? (wf - (F3_KT - 1)) / F3_KT + 1 : 0) : ntiles;
        n_int = n_int < n_live ? n_int : n_live;
        if (a.S % F3_KT && n_int == nkt) n_int = nkt - 1;
        const int g0 = g;                                              // stream index of this item's tile 0
        // ---- the next item of this block: its Q fragments and key-mask bytes are requested when this wave's last score tile has been issued
        // (qf is dead from there on), they land under the last exponentials / P.V and the later waves' tiles
        const int it_n = it + gridDim.x;
        const bool has_next = it_n < n_items;
        int rank_n = 0, b_n = 0, h_n = 0;
        uint8_t mvn[4] = {1, 1, 1, 1};
        auto prefetch_next = [&]() {
            if (has_next) {
                decode(it_n, rank_n, b_n, h_n);
                const int qn = S32 - 128 * (rank_n + 1) + wave * 32 + (lane & 31);
                const int qrn = qn < 0 ? 0 : (qn < a.S ? qn : a.S - 1);
                const bf16_t* Qn = a.q + ((long long)b_n * a.S + qrn) * a.ld_qkv + h_n * AT_HD;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Qn + 16 * ks + 8 * half);
                if (a.key_mask) f4_mask_fetch(a.key_mask + (long long)b_n * a.S, a.S, mvn);
            }
        };
        auto kbits = [&](int t) -> uint32_t { return (uint32_t)__ballot(sMask[t * F3_KT + (lane & 31)] != 0); };

        f32x16 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        f32x16 x;
        if (n_live > 0) {                                              // scores and reference max of tile 0 (in LDS: the previous top() saw to it)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = 0.f;
            const char* sK0 = smem + (g0 & 3) * F3_STAGE;
            bf16x8 kf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) kf[i] = lds_row8(sK0, lane & 31, 2 * i + half);
#pragma unroll
            for (int i = 0; i < 8; ++i) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[i], x, 0, 0, 0);
            const uint32_t v0 = visible_bits(kbits(0), half, 0, qi, a.causal);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = (v0 >> rowmap(r, 0)) & 1u ? x[r] : -INFINITY;
            m_run = half_swap_max(row_max16(x)) * sc2;
        }
        int t = 0;
        for (; t < n_live - 1; ++t) {
            top();
            const uint32_t km = kbits(t + 1);
            const bool masked = t + 1 >= n_int || km != 0xFFFFFFFFu;
            const uint32_t vn = visible_bits(km, half, (t + 1) * F3_KT, qi, a.causal);
            f3_step<true>(masked, smem + ((g0 + t + 1) & 3) * F3_STAGE, smem + ((g0 + t) & 3) * F3_STAGE + F3_TB, qf, x, o, m_run, l_run, sc2, vn, lane, half);
        }
        prefetch_next();                                               // qf is dead: the wave's last score tile has been issued (or it has none)
        if (n_live > 0) {
            top();
            f3_step<false>(false, smem, smem + ((g0 + t) & 3) * F3_STAGE + F3_TB, qf, x, o, m_run, l_run, sc2, 0u, lane, half);
            ++t;
        }
        for (; t < ntiles; ++t) top();                                 // tiles of the block's later waves: DMA share and barriers
        if (has_next) f4_mask_commit(a.key_mask != nullptr, a.S, S32, mvn, sMaskBase + (parity ^ 1) * S32);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // the next item's mask bytes are visible; every wave is done with this item's
        const float l_tot = half_swap_sum(l_run);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        f4_store_rows(strip, o, inv, a.o + row_base * a.ld_o + h * AT_HD, a.ld_o, wave_q0, a.S, lane);
        if (qi >= 0 && qi < a.S && half == 0 && a.lse)
            a.lse[((long long)b * a.H + h) * a.S + qi] = (m_run == -INFINITY) ? INFINITY : m_run * 0.6931471805599453f + logf(l_tot);
        if (!has_next) break;
        it = it_n; rank = rank_n; b = b_n; h = h_n;
        parity ^= 1;
    }
}


// =================================================================================================
// backward 1/2: dQ (and delta).  Same structure as the forward (query on the lane): per 32-key sub-tile
//   X = K.Q^T, dP^T = V.dO^T, dS^T = P^T*(dP^T - delta), dQ^T += K^T.dS^T
// =================================================================================================
template <int OCC, int HD = AT_HD>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int TB = 64 * 2 * HD;                                   // bytes of one 64-row K or V tile (HD = 64: the PointBERT blocks' backward under --unfreeze_pc_encoder)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = OCC == 2 ? 2 : 3;                             // K|V stages: OCC 2 = two blocks per CU (64 KB each), else three stages for the lone wave per SIMD
    char* sMask = smem + NST * 2 * TB;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rank, h, b;
    attn_block_map(a, rank, h, b);
    const int q0 = ((a.S + 127) / 128 - 1 - rank) * 128;             // causal: longest blocks are dispatched first
    const bool wave_dead = q0 + wave * 32 >= a.S;                     // ragged last block: no query in this wave
    const long long row_base = (long long)b * a.S;
    const bf16_t* Q = a.q + row_base * a.ld_qkv + h * HD;
    const bf16_t* K = a.k + row_base * a.ld_qkv + h * HD;
    const bf16_t* V = a.v + row_base * a.ld_qkv + h * HD;
    const bf16_t* DO = a.dout + row_base * a.ld_o + h * HD;
    const int qi = q0 + wave * 32 + (lane & 31);
    const int qr = qi < a.S ? qi : a.S - 1;
    int last = q0 + 127 < a.S - 1 ? q0 + 127 : a.S - 1;
    const int ntiles = a.causal ? (last / 64 + 1) : ((a.S + 63) / 64);
    // NST-1 tiles in flight before the loop; past the last tile the DMA re-loads it into a stage nobody reads any more,
    // which keeps the vmcnt arithmetic constant (8 DMAs per tile and wave)
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) {
        const int ti = i < ntiles ? i : ntiles - 1;
        tile_dma_t<64, HD>(K, a.ld_qkv, ti * 64, a.S - 1, smem + i * (2 * TB), wave, lane);
        tile_dma_t<64, HD>(V, a.ld_qkv, ti * 64, a.S - 1, smem + i * (2 * TB) + TB, wave, lane);
    }
    uint8_t mv[AT_MASK_IT];
    mask_fetch(a, row_base, ntiles * 64, mv);
    bf16x8 qf[HD / 16], dof[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (long long)qr * a.ld_qkv + 16 * ks + 8 * half);
        dof[ks] = *reinterpret_cast<const bf16x8*>(DO + (long long)qr * a.ld_o + 16 * ks + 8 * half);
    }
    const long long st = ((long long)b * a.H + h) * a.S + qr;
    const float lse2 = a.lse[st] * 1.4426950408889634f;
    // delta[b,h,q] = sum_d dO[q,d] * O[q,d]: this lane holds 64 of its query's 128 dO values, the other half-wave the rest.
    // Written out for the dK/dV kernel, which runs after this one on the same stream.
    float dlt = 0.f;
    {
        const bf16_t* Orow = a.o + (row_base + qr) * a.ld_o + h * HD;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(Orow + 16 * ks + 8 * half);
#pragma unroll
            for (int e = 0; e < 8; ++e) dlt = fmaf((float)dof[ks][e], (float)of[e], dlt);
        }
        dlt += __shfl_xor(dlt, 32, 64);
        if (half == 0 && qi < a.S) a.delta[st] = dlt;
    }
    const float sc2 = a.scale * 1.4426950408889634f;
    // ordinary loads are consumed before the first LDS-DMA is outstanding (see attn_fwd_kernel)
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) asm volatile("" :: "v"(qf[ks]), "v"(dof[ks]));
    asm volatile("" :: "v"(lse2), "v"(dlt));
    mask_commit(a, row_base, ntiles * 64, mv, sMask);                  // visible to the block after the loop's first barrier
    f32x16 dq[HD / 32];
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    int stg = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
        char* sK = smem + stg * (2 * TB);
        char* sV = sK + TB;
        {
            const int t2 = t + NST - 1 < ntiles ? t + NST - 1 : ntiles - 1;
            const int s2 = stg >= 1 ? stg - 1 : NST - 1;               // (stg + NST - 1) % NST
            char* nK = smem + s2 * (2 * TB);
            tile_dma_t<64, HD>(K, a.ld_qkv, t2 * 64, a.S - 1, nK, wave, lane);
            tile_dma_t<64, HD>(V, a.ld_qkv, t2 * 64, a.S - 1, nK + TB, wave, lane);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 1) * (HD / 16)) : "memory");   // tile t landed (HD / 16 DMAs per tile and wave); the later ones stay in flight
        }
        stg = stg == NST - 1 ? 0 : stg + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned long long kmask = __ballot(sMask[kv0 + lane] != 0);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (wave_dead || (a.causal && (kv0 + 32 * sub > q0 + wave * 32 + 31))) continue;       // wave-uniform: every P of this sub-tile is 0
            f32x16 x, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) {
                x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row8_t<HD>(sK, 32 * sub + (lane & 31), 2 * ks + half), qf[ks], x, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row8_t<HD>(sV, 32 * sub + (lane & 31), 2 * ks + half), dof[ks], dp, 0, 0, 0);
            }
            float ds[16];
            const bool interior = ((kmask >> (32 * sub)) & 0xFFFFFFFFull) == 0xFFFFFFFFull &&
                                  (!a.causal || kv0 + 32 * sub + 31 <= q0 + wave * 32);       // wave-uniform
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(fmaf(x[r], sc2, -lse2)) * (dp[r] - dlt);
            } else {
                asm volatile("");
                const uint32_t vis = visible_bits((uint32_t)(kmask >> (32 * sub)), half, kv0 + 32 * sub, qi, a.causal);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = (vis >> rowmap(r, 0)) & 1u ? fast_exp2(fmaf(x[r], sc2, -lse2)) : 0.f;
                    ds[r] = p * (dp[r] - dlt);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 db = pack8(&ds[8 * s2]);
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr8_t<HD>(sK, 32 * sub + 16 * s2, 32 * dt, lane), db, dq[dt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the tail's redundant DMAs have landed ...
    __builtin_amdgcn_s_barrier();                                     // ... for every wave: the stages are free for the row exchange
    store_rows_t<HD>(smem + wave * (HD == 128 ? AT_XBYTES : 32 * 144), dq, a.scale, a.dq + row_base * a.ld_dqkv + h * HD, a.ld_dqkv, q0 + wave * 32, a.S, lane,
                     a.rope_cos, a.rope_sin);
}

// =================================================================================================
// backward 1/2, second form (round 3, default; egomi_attn_set_bwd_form(1) / EGOMI_ATTN_BWD=1 select the kernels above and below for A/B runs).
// Same tiling and arithmetic as attn_bwd_dq_kernel<2> — results are bit-identical — with the instruction-stream changes of
// attn_fwd2_kernel: LDS-DMA from inline asm (the compiler put `s_waitcnt vmcnt(0)` in front of the K^T fragment reads of every tile, draining
// the K/V prefetch), per-wave interior / edge / dead tile loops of straight-line code, K / V row fragments read two steps ahead of the
// two interleaved MFMA chains, K^T fragments issued before the exponentials.
// =================================================================================================
template <bool MASKED>
__device__ __forceinline__ void dq2_subtile(const char* sK, const char* sV, const int sub, const bf16x8 (&qf)[8], const bf16x8 (&dof)[8], f32x16 (&dq)[4],
                                            const float sc2, const float lse2, const float dlt, const uint32_t vis, const int lane, const int half) {
    f32x16 x, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = 0.f; dp[r] = 0.f; }
    const int row = 32 * sub + (lane & 31);
    bf16x8 kr[3], vr[3];
    kr[0] = lds_row8(sK, row, half);     vr[0] = lds_row8(sV, row, half);
    kr[1] = lds_row8(sK, row, 2 + half); vr[1] = lds_row8(sV, row, 2 + half);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks + 2 < 8) {
            kr[(ks + 2) % 3] = lds_row8(sK, row, 2 * (ks + 2) + half);
            vr[(ks + 2) % 3] = lds_row8(sV, row, 2 * (ks + 2) + half);
        }
        x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[ks % 3], qf[ks], x, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[ks % 3], dof[ks], dp, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 kt0[4], kt1[4];                                               // K^T fragments of the dQ products: the first half in flight under the exponentials
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) kt0[dt] = lds_tr8(sK, 32 * sub, 32 * dt, lane);
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float pr = fast_exp2(fmaf(x[r], sc2, -lse2));
        if (MASKED) pr = (vis >> rowmap(r, 0)) & 1u ? pr : 0.f;
        ds[r] = pr * (dp[r] - dlt);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) kt1[dt] = lds_tr8(sK, 32 * sub + 16, 32 * dt, lane);
    {
        const bf16x8 db = pack8(&ds[0]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt0[dt], db, dq[dt], 0, 0, 0);
    }
    {
        const bf16x8 db = pack8(&ds[8]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt1[dt], db, dq[dt], 0, 0, 0);
    }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_kernel(AttnArgs a) {
    constexpr int TB = 64 * 256;                                       // bytes of one 64-key K or V tile
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][K | V] + key mask bytes
    char* sMask = smem + 2 * 2 * TB;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rank, h, b;
    attn_block_map(a, rank, h, b);
    // query blocks aligned to the END of the sequence rounded up to 32 (round 4, as attn_fwd3_kernel): the ragged block is the first,
    // shortest one instead of the last (S = 692: 36 instead of 41 tile-steps per (b, h) pair); rows q < 0 are dead lanes / dead waves.
    // Every row still meets the same 64-key tiles in the same order: results are unchanged bit for bit.
    const int q0 = ((a.S + 31) & ~31) - 128 * (rank + 1);
    const long long row_base = (long long)b * a.S;
    const bf16_t* Q = a.q + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* K = a.k + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* V = a.v + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* DO = a.dout + row_base * a.ld_o + h * AT_HD;
    const int qi = q0 + wave * 32 + (lane & 31);
    const int qr = qi < 0 ? 0 : (qi < a.S ? qi : a.S - 1);
    int last = q0 + 127 < a.S - 1 ? q0 + 127 : a.S - 1;
    const int ntiles = a.causal ? (last / 64 + 1) : ((a.S + 63) / 64);
    const bool ragged = ntiles * 64 > a.S;
    uint32_t koff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rl = (wave * 4 + j) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
        const int r = rl < a.S ? rl : a.S - 1;
        koff[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
    }
    const uint32_t tile_stride = (uint32_t)(64 * a.ld_qkv * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_void_t*)smem) + wave * 4 * 1024;
    auto dma_fast = [&](int stage) {
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds0 + stage * (2 * TB));
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16_asm(K, koff[j], base + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16_asm(V, koff[j], base + TB + j * 1024);
    };
    auto dma_slow = [&](int stage, int row0) {
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds0 + stage * (2 * TB));
        uint32_t off[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rl = (wave * 4 + j) * 4 + (lane >> 4);
            const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
            int r = row0 + rl;
            r = r < a.S ? r : a.S - 1;
            off[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16_asm(K, off[j], base + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16_asm(V, off[j], base + TB + j * 1024);
    };
    dma_fast(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) koff[j] += tile_stride;
    uint8_t mv[AT_MASK_IT];
    mask_fetch(a, row_base, ntiles * 64, mv);
    bf16x8 qf[8], dof[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (long long)qr * a.ld_qkv + 16 * ks + 8 * half);
        dof[ks] = *reinterpret_cast<const bf16x8*>(DO + (long long)qr * a.ld_o + 16 * ks + 8 * half);
    }
    const long long st = ((long long)b * a.H + h) * a.S + qr;
    const float lse2 = a.lse[st] * 1.4426950408889634f;
    float dlt = 0.f;                                                   // delta[b,h,q] = sum_d dO[q,d] * O[q,d] (see attn_bwd_dq_kernel)
    {
        const bf16_t* Orow = a.o + (row_base + qr) * a.ld_o + h * AT_HD;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(Orow + 16 * ks + 8 * half);
#pragma unroll
            for (int e = 0; e < 8; ++e) dlt = fmaf((float)dof[ks][e], (float)of[e], dlt);
        }
        dlt += __shfl_xor(dlt, 32, 64);
        if (half == 0 && qi >= 0 && qi < a.S) a.delta[st] = dlt;
    }
    const float sc2 = a.scale * 1.4426950408889634f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" :: "v"(qf[ks]), "v"(dof[ks]));
    asm volatile("" :: "v"(lse2), "v"(dlt));
    mask_commit(a, row_base, ntiles * 64, mv, sMask);
    f32x16 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    const int wave_q0 = q0 + wave * 32;
    const int wl = wave_q0 + 31 < a.S - 1 ? wave_q0 + 31 : a.S - 1;   // the wave's last live query (< 0: none)
    const int wf = wave_q0 < 0 ? 0 : wave_q0;
    int n_live = wl < 0 ? 0 : (a.causal ? wl / 64 + 1 : ntiles);
    n_live = n_live < ntiles ? n_live : ntiles;
    int n_int = a.causal ? (wf >= 63 ? (wf - 63) / 64 + 1 : 0) : ntiles;
    n_int = n_int < n_live ? n_int : n_live;
    if (a.S % 64 && n_int == ntiles) n_int = ntiles - 1;

    auto top_of_tile = [&](int t) {
        if (t + 1 < ntiles) {
            if (ragged && t + 2 == ntiles) {
                dma_slow((t + 1) & 1, (t + 1) * 64);
            } else {
                dma_fast((t + 1) & 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) koff[j] += tile_stride;
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto end_of_tile = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    int t = 0;
    bool carried = false;
    unsigned long long kmask = 0ull;
    for (; t < n_int; ++t) {
        top_of_tile(t);
        kmask = __ballot(sMask[t * 64 + lane] != 0);
        if (kmask != ~0ull) { carried = true; break; }
        const char* sK = smem + (t & 1) * (2 * TB);
        dq2_subtile<false>(sK, sK + TB, 0, qf, dof, dq, sc2, lse2, dlt, 0u, lane, half);
        dq2_subtile<false>(sK, sK + TB, 1, qf, dof, dq, sc2, lse2, dlt, 0u, lane, half);
        end_of_tile();
    }
    for (; t < n_live; ++t) {
        if (!carried) {
            top_of_tile(t);
            kmask = __ballot(sMask[t * 64 + lane] != 0);
        }
        carried = false;
        const int kv0 = t * 64;
        const char* sK = smem + (t & 1) * (2 * TB);
        const uint32_t v0 = visible_bits((uint32_t)kmask, half, kv0, qi, a.causal), v1 = visible_bits((uint32_t)(kmask >> 32), half, kv0 + 32, qi, a.causal);
        dq2_subtile<true>(sK, sK + TB, 0, qf, dof, dq, sc2, lse2, dlt, v0, lane, half);
        dq2_subtile<true>(sK, sK + TB, 1, qf, dof, dq, sc2, lse2, dlt, v1, lane, half);
        end_of_tile();
    }
    for (; t < ntiles; ++t) {
        top_of_tile(t);
        end_of_tile();
    }
    store_rows_via_lds(smem + wave * AT_XBYTES, dq, a.scale, a.dq + row_base * a.ld_dqkv + h * AT_HD, a.ld_dqkv, q0 + wave * 32, a.S, lane,
                       a.rope_cos, a.rope_sin);
}

// =================================================================================================
// backward 1/2, third form (round 4, default; EGOMI_ATTN_BWD=2 / egomi_attn_set_bwd_form(2) select the second form for A/B runs): the arithmetic
// of attn_bwd_dq2_kernel per 32-key tile (dq2_subtile, unchanged: results stay bit-identical) on attn_fwd3_kernel's data path — 32-key
// tiles in a 4-stage LDS ring, tile t+3 requested at the top of tile t, ONE barrier per 32 keys with nothing waiting for a DMA younger than
// two tiles (the second form met two barriers per 64 keys and waited for the DMA it had issued one tile earlier) — and query blocks aligned
// to the end of the sequence.  The intra-wave pipeline of the forward (scores of tile t+1 under the exponentials of tile t) does not fit here:
// x and dP of two tiles next to dQ, Q and dO fragments are 264 registers, and two blocks per CU is worth more.
// =================================================================================================
__global__ __launch_bounds__(256, 2) void attn_bwd_dq3_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [4 stages][K 8 KB | V 8 KB] + key mask bytes
    char* sMask = smem + F3_NST * F3_STAGE;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rank, h, b;
    attn_block_map(a, rank, h, b);
    const int q0 = ((a.S + 31) & ~31) - 128 * (rank + 1);
    const long long row_base = (long long)b * a.S;
    const bf16_t* Q = a.q + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* K = a.k + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* V = a.v + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* DO = a.dout + row_base * a.ld_o + h * AT_HD;
    const int wave_q0 = q0 + wave * 32;
    const int qi = wave_q0 + (lane & 31);
    const int qr = qi < 0 ? 0 : (qi < a.S ? qi : a.S - 1);
    const int last = q0 + 127 < a.S - 1 ? q0 + 127 : a.S - 1;
    const int nkt = (a.S + F3_KT - 1) / F3_KT;
    const int ntiles = a.causal ? last / F3_KT + 1 : nkt;
    uint32_t koff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rl = (wave * 2 + j) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
        const int r = rl < a.S ? rl : a.S - 1;
        koff[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
    }
    const uint32_t tile_stride = (uint32_t)(F3_KT * a.ld_qkv * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_void_t*)smem) + wave * 2 * 1024;
    int next_req = 0;
    auto request = [&]() {
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds0 + (next_req & (F3_NST - 1)) * F3_STAGE);
        if ((next_req + 1) * F3_KT <= a.S) {
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(K, koff[j], base + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(V, koff[j], base + F3_TB + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) koff[j] += tile_stride;
        } else {
            uint32_t off[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rl = (wave * 2 + j) * 4 + (lane >> 4);
                const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
                int r = next_req * F3_KT + rl;
                r = r < a.S ? r : a.S - 1;
                off[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(K, off[j], base + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16_asm(V, off[j], base + F3_TB + j * 1024);
        }
        ++next_req;
    };
    STAMP_DECL
    STAMP_START
    request();
    if (ntiles > 1) request();
    STAMP(0)
    uint8_t mv[AT_MASK_IT];
    mask_fetch(a, row_base, ntiles * F3_KT, mv);
    bf16x8 qf[8], dof[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (long long)qr * a.ld_qkv + 16 * ks + 8 * half);
        dof[ks] = *reinterpret_cast<const bf16x8*>(DO + (long long)qr * a.ld_o + 16 * ks + 8 * half);
    }
    const long long st = ((long long)b * a.H + h) * a.S + qr;
    const float lse2 = a.lse[st] * 1.4426950408889634f;
    float dlt = 0.f;                                                   // delta[b,h,q] = sum_d dO[q,d] * O[q,d] (see attn_bwd_dq_kernel)
    {
        const bf16_t* Orow = a.o + (row_base + qr) * a.ld_o + h * AT_HD;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(Orow + 16 * ks + 8 * half);
#pragma unroll
            for (int e = 0; e < 8; ++e) dlt = fmaf((float)dof[ks][e], (float)of[e], dlt);
        }
        dlt += __shfl_xor(dlt, 32, 64);
        if (half == 0 && qi >= 0 && qi < a.S) a.delta[st] = dlt;
    }
    const float sc2 = a.scale * 1.4426950408889634f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" :: "v"(qf[ks]), "v"(dof[ks]));
    asm volatile("" :: "v"(lse2), "v"(dlt));
    STAMP(1)
    mask_commit(a, row_base, ntiles * F3_KT, mv, sMask);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // tiles 0 and 1 (and the delta store) are done
    if (ntiles > 2) request();                                         // tile 2 stays in flight across the barrier
    f32x16 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    const int wl = wave_q0 + 31 < a.S - 1 ? wave_q0 + 31 : a.S - 1;
    const int wf = wave_q0 < 0 ? 0 : wave_q0;
    int n_live = wl < 0 ? 0 : (a.causal ? wl / F3_KT + 1 : ntiles);
    n_live = n_live < ntiles ? n_live : ntiles;
    int n_int = a.causal ? (wf >= F3_KT - 1 ? (wf - (F3_KT - 1)) / F3_KT + 1 : 0) : ntiles;
    n_int = n_int < n_live ? n_int : n_live;
    if (a.S % F3_KT && n_int == nkt) n_int = nkt - 1;

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                      // tiles 0, 1 and the mask bytes are in LDS for every wave
    STAMP(2)
    // top of tile t: tile t has landed for everybody (tiles t+1, t+2 may still be in flight), stage (t+3) % 4 = tile t-1's is free: request tile t+3
    auto top = [&](int t) {
        if (t > 0) {
            const int ahead = ntiles - 1 - t;                          // tiles after t that have been requested: min(ahead, 2)
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (t + 3 < ntiles) request();
    };
    auto kbits = [&](int t) -> uint32_t { return (uint32_t)__ballot(sMask[t * F3_KT + (lane & 31)] != 0); };
    int t = 0;
    bool carried = false;
    uint32_t km = 0u;
    for (; t < n_int; ++t) {
        top(t);
        STAMP(4)
        km = kbits(t);
        if (km != 0xFFFFFFFFu) { carried = true; break; }
        const char* sK = smem + (t & 3) * F3_STAGE;
        dq2_subtile<false>(sK, sK + F3_TB, 0, qf, dof, dq, sc2, lse2, dlt, 0u, lane, half);
        STAMP(5)
    }
    for (; t < n_live; ++t) {
        if (!carried) { top(t); STAMP(4) km = kbits(t); }
        carried = false;
        const char* sK = smem + (t & 3) * F3_STAGE;
        const uint32_t v0 = visible_bits(km, half, t * F3_KT, qi, a.causal);
        dq2_subtile<true>(sK, sK + F3_TB, 0, qf, dof, dq, sc2, lse2, dlt, v0, lane, half);
        STAMP(5)
    }
    for (; t < ntiles; ++t) top(t);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                      // nobody reads a stage any more: the ring becomes the epilogue's strips
    STAMP(6)
    store_rows_via_lds(smem + wave * AT_XBYTES, dq, a.scale, a.dq + row_base * a.ld_dqkv + h * AT_HD, a.ld_dqkv, wave_q0, a.S, lane,
                       a.rope_cos, a.rope_sin);
    STAMP(7)
    STAMP_FLUSH
}

// =================================================================================================
// backward 2/2: dK, dV.  Key on the lane: a workgroup owns 128 keys (4 waves x 32), keeps their K and
// V fragments and the dK^T / dV^T accumulators in registers (one wave per SIMD, 512-register file) and
// sweeps the queries in tiles of 32 (Q, dO, LSE, delta tiles double-buffered in LDS by LDS-DMA):
//   X = Q.K^T, dP = dO.V^T, P = exp2(sc2*X - lse2), dS = P*(dP - delta), dV^T += dO^T.P, dK^T += Q^T.dS
// =================================================================================================
#define DKV_STAGE (2 * 32 * 256 + 4 * 256)
template <int OCC, int HD = AT_HD>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dkdv_kernel(AttnArgs a) {
    constexpr int QB = 32 * 2 * HD, STAGE = 2 * QB + 4 * 256;          // Q | dO tiles of 32 rows + LSE / delta pieces (HD = 128: DKV_STAGE)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rank, h, b;
    attn_block_map(a, rank, h, b);
    const int kb0 = rank * 128;                                       // causal: key block 0 sweeps the most queries and is dispatched first
    const long long row_base = (long long)b * a.S;
    const bf16_t* Q = a.q + row_base * a.ld_qkv + h * HD;
    const bf16_t* K = a.k + row_base * a.ld_qkv + h * HD;
    const bf16_t* V = a.v + row_base * a.ld_qkv + h * HD;
    const bf16_t* DO = a.dout + row_base * a.ld_o + h * HD;
    const float* LSE = a.lse + ((long long)b * a.H + h) * a.S;
    const float* DEL = a.delta + ((long long)b * a.H + h) * a.S;
    const int kj = kb0 + wave * 32 + (lane & 31);                      // this lane's key
    const int kr = kj < a.S ? kj : a.S - 1;
    bool key_ok = kj < a.S;
    if (key_ok && a.key_mask) key_ok = a.key_mask[row_base + kj] != 0;
    const bool keys_all_ok = __all(key_ok);
    // OCC 2 (two blocks per CU, <= 256 registers): two stages, and the block's 128 V rows live in LDS (32 KB, read per tile)
    // instead of 32 registers per lane; OCC 1: three stages, V fragments in registers
    constexpr int NST = OCC == 2 ? 2 : 3;
    char* sVblk = smem + NST * STAGE;
    bf16x8 kf[HD / 16], vf[OCC == 2 ? 1 : HD / 16];
    if (OCC == 2) tile_dma_t<128, HD>(V, a.ld_qkv, kb0, a.S - 1, sVblk, wave, lane);
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8*>(K + (long long)kr * a.ld_qkv + 16 * ks + 8 * half);
        if (OCC != 2) vf[ks] = *reinterpret_cast<const bf16x8*>(V + (long long)kr * a.ld_qkv + 16 * ks + 8 * half);
    }
    // ordinary loads are consumed before the query-tile DMAs start (see attn_fwd_kernel)
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) asm volatile("" :: "v"(kf[ks]));
    if (OCC != 2) {
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) asm volatile("" :: "v"(vf[ks]));
    }
    f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    const float sc2 = a.scale * 1.4426950408889634f;
    const int nq = (a.S + 31) / 32;
    const int qt0 = a.causal ? (kb0 / 32) : 0;

    auto issue = [&](int qt, char* stage) {
        const int q0 = qt * 32;
        tile_dma_t<32, HD>(Q, a.ld_qkv, q0, a.S - 1, stage, wave, lane);
        tile_dma_t<32, HD>(DO, a.ld_o, q0, a.S - 1, stage + QB, wave, lane);
        int qq = q0 + lane;                                            // 64 floats per DMA; only the first 32 are read
        qq = qq < a.S ? qq : a.S - 1;
        const float* src = (wave & 1) ? (DEL + qq) : (LSE + qq);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(stage + 2 * QB + wave * 256), 4, 0, 0);
    };
    // NST stages, NST-1 query tiles in flight; past the last tile the DMA re-loads it into a stage nobody reads any more,
    // which keeps the vmcnt arithmetic constant (5 DMAs per tile)
    if (qt0 < nq) {
#pragma unroll
        for (int i = 0; i < NST - 1; ++i) issue(qt0 + i < nq ? qt0 + i : nq - 1, smem + i * STAGE);
    }
    int stg = 0;
    for (int qt = qt0; qt < nq; ++qt) {
        const int q0 = qt * 32;
        char* st = smem + stg * STAGE;
        issue(qt + NST - 1 < nq ? qt + NST - 1 : nq - 1, smem + (stg >= 1 ? stg - 1 : NST - 1) * STAGE);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 1) * (HD / 32 + 1)) : "memory");  // tile qt landed; (NST-1) x (HD/64 + HD/64 + 1) DMAs stay in flight
        stg = stg == NST - 1 ? 0 : stg + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* sQ = st;
        const char* sDO = st + QB;
        const float* sL = reinterpret_cast<const float*>(st + 2 * QB);          // LSE of the 32 queries
        const float* sD = sL + 64;                                                      // delta (wave 1's piece)
        f32x16 x, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { x[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row8_t<HD>(sQ, lane & 31, 2 * ks + half), kf[ks], x, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row8_t<HD>(sDO, lane & 31, 2 * ks + half),
                                                         OCC == 2 ? lds_row8_t<HD>(sVblk, wave * 32 + (lane & 31), 2 * ks + half) : vf[OCC == 2 ? 0 : ks], dp, 0, 0, 0);
        }
        float pv[16], ds[16];
        // masks only on edge tiles: every key of the wave visible, every query of the tile in range and (causal) not before any key
        const bool interior = keys_all_ok && q0 + 31 < a.S && (!a.causal || kb0 + wave * 32 + 31 <= q0);      // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + 8 * g + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[4 * g + e] = fast_exp2(fmaf(x[4 * g + e], sc2, -l4[e] * 1.4426950408889634f));
        }
        if (!interior) {
            asm volatile("");                                              // keeps the edge path a real branch
            // query bit c = rowmap(r, 0): visible when q0 + 4*half + c is a real query and (causal) not before this lane's key
            const int lo = a.causal ? kj - q0 - 4 * half : 0;                   // c >= lo
            const int hi = a.S - 1 - q0 - 4 * half;                             // c <= hi
            uint32_t vis = key_ok ? 0xFFFFFFFFu : 0u;
            vis &= lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : ~((1u << lo) - 1u));
            vis &= hi < 0 ? 0u : (hi >= 31 ? 0xFFFFFFFFu : ((2u << hi) - 1u));
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = (vis >> rowmap(r, 0)) & 1u ? pv[r] : 0.f;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + 8 * g + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) ds[4 * g + e] = pv[4 * g + e] * (dp[4 * g + e] - d4[e]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pb = pack8(&pv[8 * s2]);
            const bf16x8 db = pack8(&ds[8 * s2]);
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr8_t<HD>(sDO, 16 * s2, 32 * dt, lane), pb, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr8_t<HD>(sQ, 16 * s2, 32 * dt, lane), db, dk[dt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    store_rows_t<HD>(smem + wave * (HD == 128 ? AT_XBYTES : 32 * 144), dk, a.scale, a.dk + row_base * a.ld_dqkv + h * HD, a.ld_dqkv, kb0 + wave * 32, a.S, lane,
                     a.rope_cos, a.rope_sin);
    store_rows_t<HD>(smem + wave * (HD == 128 ? AT_XBYTES : 32 * 144), dv, 1.0f, a.dv + row_base * a.ld_dqkv + h * HD, a.ld_dqkv, kb0 + wave * 32, a.S, lane);
}

// =================================================================================================
// backward 2/2, second form (round 3, default; see attn_bwd_dq2_kernel).  Same tiling and arithmetic as attn_bwd_dkdv_kernel<2> —
// bit-identical results — with: LDS-DMA from inline asm (Q / dO / LSE / delta tiles and the block's V rows), per-wave dead / diagonal /
// interior / ragged tile loops of straight-line code (a causal wave used to run its dead query tiles through the masked path and add
// zeros; a wave without a real key now only moves its DMA share), Q / dO / V row fragments read two steps ahead of the two MFMA chains,
// the first dO^T / Q^T fragments issued before the exponentials.
// =================================================================================================
__device__ __forceinline__ void dma4_asm(const void* gbase, uint32_t voff, uint32_t lds_base) {       // one dword per lane
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(gbase), "s"(lds_base) : "memory");     // (m0 is a reserved register: hipcc re-loads it in front of each of its own uses)
}
template <bool MASKED>
__device__ __forceinline__ void dkdv2_tile(const char* sQ, const char* sDO, const char* sVrows, const float* sL, const float* sD, const bf16x8 (&kf)[8],
                                           f32x16 (&dk)[4], f32x16 (&dv)[4], const float sc2, const uint32_t vis, const int lane, const int half) {
    f32x16 x, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = 0.f; dp[r] = 0.f; }
    const int row = lane & 31;
    bf16x8 qr[2], dr[2], vr[2];                                          // one step ahead of the two chains (three operands: 24 registers)
    qr[0] = lds_row8(sQ, row, half); dr[0] = lds_row8(sDO, row, half); vr[0] = lds_row8(sVrows, row, half);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
            qr[(ks + 1) & 1] = lds_row8(sQ, row, 2 * (ks + 1) + half);
            dr[(ks + 1) & 1] = lds_row8(sDO, row, 2 * (ks + 1) + half);
            vr[(ks + 1) & 1] = lds_row8(sVrows, row, 2 * (ks + 1) + half);
        }
        x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr[ks & 1], kf[ks], x, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dr[ks & 1], vr[ks & 1], dp, 0, 0, 0);
    }
    bf16x8 dot[4], qt[4];                                                // dO^T fragments of the first 16 queries: in flight under the exponentials
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dot[dt] = lds_tr8(sDO, 0, 32 * dt, lane);
    float pv[16], ds[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + 8 * g + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[4 * g + e] = fast_exp2(fmaf(x[4 * g + e], sc2, -l4[e] * 1.4426950408889634f));
    }
    if (MASKED) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[r] = (vis >> rowmap(r, 0)) & 1u ? pv[r] : 0.f;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) qt[dt] = lds_tr8(sQ, 0, 32 * dt, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + 8 * g + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) ds[4 * g + e] = pv[4 * g + e] * (dp[4 * g + e] - d4[e]);
    }
    {
        const bf16x8 pb = pack8(&pv[0]);
        const bf16x8 db = pack8(&ds[0]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot[dt], pb, dv[dt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dot[dt] = lds_tr8(sDO, 16, 32 * dt, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[dt], db, dk[dt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) qt[dt] = lds_tr8(sQ, 16, 32 * dt, lane);
    }
    {
        const bf16x8 pb = pack8(&pv[8]);
        const bf16x8 db = pack8(&ds[8]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot[dt], pb, dv[dt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[dt], db, dk[dt], 0, 0, 0);
    }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv2_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2] x (Q 8 KB | dO 8 KB | LSE, delta 1 KB) + the block's 128 V rows (32 KB)
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int rank, h, b;
    attn_block_map(a, rank, h, b);
    const int kb0 = rank * 128;
    const long long row_base = (long long)b * a.S;
    const bf16_t* Q = a.q + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* K = a.k + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* V = a.v + row_base * a.ld_qkv + h * AT_HD;
    const bf16_t* DO = a.dout + row_base * a.ld_o + h * AT_HD;
    const float* LSE = a.lse + ((long long)b * a.H + h) * a.S;
    const float* DEL = a.delta + ((long long)b * a.H + h) * a.S;
    const int kj = kb0 + wave * 32 + (lane & 31);                      // this lane's key
    const int kr = kj < a.S ? kj : a.S - 1;
    bool key_ok = kj < a.S;
    if (key_ok && a.key_mask) key_ok = a.key_mask[row_base + kj] != 0;
    const bool keys_all_ok = __all(key_ok);
    const bool wave_dead = kb0 + wave * 32 >= a.S;                     // no real key in this wave
    char* sVblk = smem + 2 * DKV_STAGE;
    const uint32_t lds_smem = (uint32_t)(uintptr_t)((lds_void_t*)smem);
    {   // the block's V rows [kb0, kb0 + 128): 8 DMAs per wave, rows clamped to S-1
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds_smem + 2 * DKV_STAGE + wave * 8 * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rl = (wave * 8 + j) * 4 + (lane >> 4);
            const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
            int r = kb0 + rl;
            r = r < a.S ? r : a.S - 1;
            dma16_asm(V, (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u, base + j * 1024);
        }
    }
    bf16x8 kf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(K + (long long)kr * a.ld_qkv + 16 * ks + 8 * half);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" :: "v"(kf[ks]));
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    const float sc2 = a.scale * 1.4426950408889634f;
    const int nq = (a.S + 31) / 32;
    const int qt0 = a.causal ? (kb0 / 32) : 0;
    const bool ragged = nq * 32 > a.S;

    // per-lane DMA offsets of query tile qt0 (bytes from Q / dO / LSE), advanced by 32 rows per tile; the ragged last tile is clamped
    uint32_t qoff[2], dooff[2], loff;
    auto offsets_for = [&](int q0, uint32_t (&qo)[2], uint32_t (&dofs)[2], uint32_t& lo) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = (wave * 2 + j) * 4 + (lane >> 4);
            const int ch = (lane & 15) ^ (((rl & 3) << 2) | ((rl >> 2) & 3));
            int r = q0 + rl;
            r = r < a.S ? r : a.S - 1;
            qo[j] = (uint32_t)((long long)r * a.ld_qkv + ch * 8) * 2u;
            dofs[j] = (uint32_t)((long long)r * a.ld_o + ch * 8) * 2u;
        }
        // 64 lanes fetch 64 floats, the tile reads the first 32: lanes 32..63 repeat them.  (With q0 + lane the offsets advanced per tile ran
        // 32 floats past the tile, and at S % 32 == 0 — no clamped last tile — the last (b, h) pair's last tile read 128 B past the END of the
        // LSE / delta arrays: a page fault whenever such an array closed a mapped segment; S = 256 was the first such S in the suite.)
        int qq = q0 + (lane & 31);
        qq = qq < a.S ? qq : a.S - 1;
        lo = (uint32_t)qq * 4u;
    };
    offsets_for(qt0 * 32, qoff, dooff, loff);
    const uint32_t q_stride = (uint32_t)(32 * a.ld_qkv * 2), do_stride = (uint32_t)(32 * a.ld_o * 2);
    const float* LD = (wave & 1) ? DEL : LSE;                         // waves 0/2 fetch LSE, waves 1/3 delta (64 floats each; 32 are read)
    auto issue = [&](int stage, const uint32_t (&qo)[2], const uint32_t (&dofs)[2], const uint32_t lo) {
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds_smem + stage * DKV_STAGE + wave * 2 * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16_asm(Q, qo[j], base + j * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16_asm(DO, dofs[j], base + 32 * 256 + j * 1024);
        dma4_asm(LD, lo, __builtin_amdgcn_readfirstlane(lds_smem + stage * DKV_STAGE + 2 * 32 * 256 + wave * 256));
    };
    auto issue_next = [&](int qt_next, int stage) {                    // tile qt_next into `stage`; offsets advance with it
        if (ragged && qt_next == nq - 1) {
            uint32_t qo[2], dofs[2], lo;
            offsets_for(qt_next * 32, qo, dofs, lo);
            issue(stage, qo, dofs, lo);
        } else {
            issue(stage, qoff, dooff, loff);
#pragma unroll
            for (int j = 0; j < 2; ++j) { qoff[j] += q_stride; dooff[j] += do_stride; }
            loff += 128u;
        }
    };
    if (qt0 < nq) issue_next(qt0, 0);
    // V rows (8) + first tile (5) are in flight; the loop waits for "everything but the newest tile"
    int stg = 0;
    auto top_of_tile = [&](int qt) {
        if (qt + 1 < nq) {
            issue_next(qt + 1, stg ^ 1);
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto end_of_tile = [&]() {
        stg ^= 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    const char* sVrows = sVblk + wave * 32 * 256;
    auto run = [&](auto masked_tag, int qt) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const char* st = smem + stg * DKV_STAGE;
        const float* sL = reinterpret_cast<const float*>(st + 2 * 32 * 256);
        uint32_t vis = 0u;
        if (MASKED) {
            const int q0 = qt * 32;
            const int lo = a.causal ? kj - q0 - 4 * half : 0;              // query bit c = rowmap(r, 0) visible when c >= lo ...
            const int hi = a.S - 1 - q0 - 4 * half;                        // ... and c <= hi
            vis = key_ok ? 0xFFFFFFFFu : 0u;
            vis &= lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : ~((1u << lo) - 1u));
            vis &= hi < 0 ? 0u : (hi >= 31 ? 0xFFFFFFFFu : ((2u << hi) - 1u));
        }
        dkdv2_tile<MASKED>(st, st + 32 * 256, sVrows, sL, sL + 64, kf, dk, dv, sc2, vis, lane, half);
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    int qt = qt0;
    const int diag = kb0 / 32 + wave;                                  // causal: the query tile that holds this wave's own keys
    const int dead_end = wave_dead ? nq : (a.causal ? (diag < nq ? diag : nq) : qt0);
    for (; qt < dead_end; ++qt) {                                      // query tiles before this wave's keys (causal) / a wave without keys
        top_of_tile(qt);
        end_of_tile();
    }
    if (a.causal && qt < nq && qt == diag) {                           // the diagonal tile
        top_of_tile(qt);
        run(T_{}, qt);
        end_of_tile();
        ++qt;
    }
    if (keys_all_ok) {
        const int int_end = a.S / 32;                                  // tiles whose 32 queries all exist
        for (; qt < int_end; ++qt) {
            top_of_tile(qt);
            run(F_{}, qt);
            end_of_tile();
        }
    }
    for (; qt < nq; ++qt) {                                            // padded keys in this wave and / or the ragged last tile
        top_of_tile(qt);
        run(T_{}, qt);
        end_of_tile();
    }
    store_rows_via_lds(smem + wave * AT_XBYTES, dk, a.scale, a.dk + row_base * a.ld_dqkv + h * AT_HD, a.ld_dqkv, kb0 + wave * 32, a.S, lane,
                       a.rope_cos, a.rope_sin);
    store_rows_via_lds(smem + wave * AT_XBYTES, dv, 1.0f, a.dv + row_base * a.ld_dqkv + h * AT_HD, a.ld_dqkv, kb0 + wave * 32, a.S, lane);
}

// Both backward kernels run at two blocks per CU (<= 256 VGPRs: dQ 234; dK/dV 256 with its V rows in LDS) — the measured
// lever: one wave per SIMD left every LDS / MFMA latency exposed (dQ 140 -> 97 us, dK/dV 183 -> 123 us per layer).
// EGOMI_ATTN_OCC (A/B switch): bit 0 / bit 1 clear = one block per CU, three stages, for dQ / dKdV
static int attn_occ() {
    static int occ = -1;
    if (occ < 0) { const char* e = getenv("EGOMI_ATTN_OCC"); occ = e ? atoi(e) : 3; }
    return occ;
}
static bool occ_dq2() { return attn_occ() & 1; }

static int g_attn_fwd_form = -1;
static int attn_fwd_form() {
    if (g_attn_fwd_form < 0) { const char* e = getenv("EGOMI_ATTN_FWD"); g_attn_fwd_form = e ? atoi(e) : 4; }
    return g_attn_fwd_form;
}
extern "C" int egomi_attn_set_fwd_form(int form) {
    if (form < 1 || form > 4) return EGOMI_E_BADARG;
    g_attn_fwd_form = form;
    return EGOMI_OK;
}
// A/B switch for the third form's block order (attn_block_map3): EGOMI_ATTN_GROUP / egomi_attn_set_fwd_group
static int g_attn_fwd_group = -1;
static int attn_fwd_group() {
    if (g_attn_fwd_group < 0) { const char* e = getenv("EGOMI_ATTN_GROUP"); g_attn_fwd_group = e ? atoi(e) : 0; }
    return g_attn_fwd_group;
}
extern "C" int egomi_attn_set_fwd_group(int group) {
    if (group < 0 || group > 64) return EGOMI_E_BADARG;
    g_attn_fwd_group = group;
    return EGOMI_OK;
}
static int g_attn_fwd_blocks = 0;
extern "C" int egomi_attn_set_fwd_blocks(int blocks) {                  // 0 = two per CU; > 0 caps the persistent forward's grid (tests)
    if (blocks < 0) return EGOMI_E_BADARG;
    g_attn_fwd_blocks = blocks;
    return EGOMI_OK;
}
static int g_attn_bwd_form = -1;
static int attn_bwd_form() {
    if (g_attn_bwd_form < 0) { const char* e = getenv("EGOMI_ATTN_BWD"); g_attn_bwd_form = e ? atoi(e) : 3; }
    return g_attn_bwd_form;
}
extern "C" int egomi_attn_set_bwd_form(int form) {
    if (form < 1 || form > 3) return EGOMI_E_BADARG;
    g_attn_bwd_form = form;
    return EGOMI_OK;
}

static int attn_check(const egomi_attn_desc* d, bool fwd_only = false) {
    if (!d || !d->q || !d->k || !d->v) return EGOMI_E_BADARG;
    if (d->head_dim != AT_HD && d->head_dim != 64) return EGOMI_E_UNSUPPORTED;
    (void)fwd_only;
    if (d->B <= 0 || d->H <= 0 || d->S <= 0) return EGOMI_E_SHAPE;
    if (d->dtype != EGOMI_BF16) return EGOMI_E_UNSUPPORTED;
    if (d->ld_qkv % 8 || d->ld_qkv < d->head_dim * d->H) return EGOMI_E_SHAPE;
    if (((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) return EGOMI_E_SHAPE;
    return EGOMI_OK;
}

static AttnArgs attn_args(const egomi_attn_desc* d) {
    AttnArgs a;
    a.q = (const bf16_t*)d->q; a.k = (const bf16_t*)d->k; a.v = (const bf16_t*)d->v; a.o = (bf16_t*)d->o; a.lse = d->lse;
    a.dout = (const bf16_t*)d->dout; a.delta = d->delta; a.dq = (bf16_t*)d->dq; a.dk = (bf16_t*)d->dk; a.dv = (bf16_t*)d->dv;
    a.key_mask = d->key_mask; a.B = d->B; a.H = d->H; a.S = d->S;
    a.ld_qkv = d->ld_qkv; a.ld_o = d->ld_o; a.ld_dqkv = d->ld_dqkv; a.scale = d->scale; a.causal = d->causal;
    a.rope_cos = d->rope_cos; a.rope_sin = d->rope_sin;
    return a;
}

extern "C" int egomi_attn_fwd(const egomi_attn_desc* d, egomi_stream_t stream) {
    const int rc = attn_check(d, true);
    if (rc) return rc;
    if (!d->o || d->ld_o % 8 || d->ld_o < d->head_dim * d->H || ((uintptr_t)d->o & 15)) return EGOMI_E_SHAPE;
    if (d->S > AT_MAXS) return EGOMI_E_UNSUPPORTED;
    AttnArgs a = attn_args(d);
    const dim3 grid((unsigned)(((d->S + 127) / 128) * d->H * d->B));
    const size_t lds = 2 * 2 * 64 * 2 * (size_t)d->head_dim + (size_t)((d->S + 63) / 64) * 64;
    int form = attn_fwd_form();                                        // EGOMI_ATTN_FWD=1 / egomi_attn_set_fwd_form(1): the first form (A/B runs, equality tests)
    if ((long long)d->S * d->ld_qkv * 2 >= (1ll << 32)) form = 1;      // the second and third forms address K/V rows with 32-bit byte offsets
    if (d->head_dim == 128 && form == 4 && d->S > 1024) form = 3;       // the persistent form keeps a sample's key-mask bytes in four registers per thread
    if (d->head_dim == 128 && form == 4) {
        const int s32 = (d->S + 31) & ~31, nblk = (s32 + 127) / 128;
        const int n_items = nblk * d->H * d->B;
        static int cus = 0;
        if (!cus) { int dev = 0; hipDeviceProp_t pr; cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
        int grid4 = n_items < 2 * cus ? n_items : 2 * cus;                 // two resident blocks per CU walk the items
        if (g_attn_fwd_blocks > 0 && g_attn_fwd_blocks < grid4) grid4 = g_attn_fwd_blocks;     // tests: several items per block at small shapes
        const size_t lds4 = (size_t)F3_NST * F3_STAGE + 4 * F4_STRIP + 2 * (size_t)s32;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        EGOMI_LAUNCH(attn_fwd4_kernel, dim3((unsigned)grid4), dim3(256), lds4, (hipStream_t)stream, a, n_items);
        return egomi_launch_status();
    }
    if (d->head_dim == 128 && form == 3) {
        const int s32 = (d->S + 31) & ~31, nblk = (s32 + 127) / 128;
        int group = (d->H * d->B) % 8 == 0 ? attn_fwd_group() : 0;
        if (group > nblk) group = nblk;
        const int nb = group > 0 ? ((nblk + group - 1) / group) * group : nblk;
        const dim3 grid3((unsigned)(nb * d->H * d->B));                    // query blocks aligned to the end of the sequence
        const size_t lds3 = (size_t)F3_NST * F3_STAGE + (size_t)s32;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        EGOMI_LAUNCH(attn_fwd3_kernel, grid3, dim3(256), lds3, (hipStream_t)stream, a, group);
        return egomi_launch_status();
    }
    if (d->head_dim == 128) {
        if (form == 1) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            EGOMI_LAUNCH(attn_fwd_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, a);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            EGOMI_LAUNCH(attn_fwd2_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, a);
        }
    } else if (form == 1) {
        EGOMI_LAUNCH(attn_fwd_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, a);
    } else {
        EGOMI_LAUNCH(attn_fwd2_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, a);
    }
    return egomi_launch_status();
}

extern "C" int egomi_attn_bwd(const egomi_attn_desc* d, egomi_stream_t stream) {
    const int rc = attn_check(d);
    if (rc) return rc;
    if (!d->o || !d->lse || !d->dout || !d->delta || !d->dq || !d->dk || !d->dv) return EGOMI_E_BADARG;
    if (d->ld_o % 8 || d->ld_dqkv % 8 || d->ld_o < d->head_dim * d->H || d->ld_dqkv < d->head_dim * d->H) return EGOMI_E_SHAPE;
    if (((uintptr_t)d->dout & 15) || (((uintptr_t)d->dq | (uintptr_t)d->dk | (uintptr_t)d->dv) & 15) || ((uintptr_t)d->o & 15)) return EGOMI_E_SHAPE;
    if (d->S > AT_MAXS) return EGOMI_E_UNSUPPORTED;
    if ((d->rope_cos == nullptr) != (d->rope_sin == nullptr)) return EGOMI_E_BADARG;
    if (d->rope_cos && (((uintptr_t)d->rope_cos | (uintptr_t)d->rope_sin) & 15)) return EGOMI_E_SHAPE;
    AttnArgs a = attn_args(d);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(((d->S + 127) / 128) * d->H * d->B));
    if (d->head_dim == 64) {
        // the PointBERT blocks (point_encoder.py:30-55 under --unfreeze_pc_encoder): the first-form kernels at HD = 64, two blocks per CU; no RoPE
        if (d->rope_cos) return EGOMI_E_UNSUPPORTED;
        const size_t lq = 2 * 2 * 64 * 128 + (size_t)((d->S + 63) / 64) * 64, lk = 2 * (2 * 32 * 128 + 4 * 256) + 128 * 128;
        EGOMI_LAUNCH((attn_bwd_dq_kernel<2, 64>), grid, dim3(256), lq, s, a);
        EGOMI_LAUNCH((attn_bwd_dkdv_kernel<2, 64>), grid, dim3(256), lk, s, a);
        return egomi_launch_status();
    }
    const size_t lds_q = ((occ_dq2() ? 2 : 3) * 2 * 64 * 256) + (size_t)((d->S + 63) / 64) * 64;
    const int occ = attn_occ();
    const bool off32 = (long long)d->S * d->ld_qkv * 2 < (1ll << 32) && (long long)d->S * d->ld_o * 2 < (1ll << 32);     // the second forms address rows with 32-bit byte offsets
    if ((occ & 1) && attn_bwd_form() == 3 && off32) {
        const size_t lds3 = (size_t)F3_NST * F3_STAGE + (size_t)((d->S + 31) & ~31);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        EGOMI_LAUNCH(attn_bwd_dq3_kernel, grid, dim3(256), lds3, s, a);
    } else if ((occ & 1) && attn_bwd_form() == 2 && off32) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        EGOMI_LAUNCH(attn_bwd_dq2_kernel, grid, dim3(256), lds_q, s, a);
    } else if (occ & 1) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        EGOMI_LAUNCH(attn_bwd_dq_kernel<2>, grid, dim3(256), lds_q, s, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        EGOMI_LAUNCH(attn_bwd_dq_kernel<1>, grid, dim3(256), lds_q, s, a);
    }
    const size_t lds_k = (occ & 2) ? 2 * DKV_STAGE + 128 * 256 : 3 * DKV_STAGE;
    if ((occ & 2) && attn_bwd_form() >= 2 && off32) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
        EGOMI_LAUNCH(attn_bwd_dkdv2_kernel, grid, dim3(256), lds_k, s, a);
    } else if (occ & 2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
        EGOMI_LAUNCH(attn_bwd_dkdv_kernel<2>, grid, dim3(256), lds_k, s, a);
    }
    else EGOMI_LAUNCH(attn_bwd_dkdv_kernel<1>, grid, dim3(256), lds_k, s, a);
    return egomi_launch_status();
}
