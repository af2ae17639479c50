// 256x256 8-phase GEMM for operands whose REDUCTION index is the slow one (a_layout / b_layout = 1 of include/egomi.h):
//   weight gradients  dW[N,K] (+)= dY^T . X      A = dY [rows, N] and B = X [rows, K], both "k-major"   (TA = TB = 1)
//   data gradients    dX[M,K]  = dY . W          A = dY [M, N] k-contiguous, B = W [N, K] k-major        (TA = 0, TB = 1)
// replaces the autograd backward of nn.Linear (dW = dY^T X, dX = dY W: the matmuls behind train.py:183 `model_engine.backward(loss)` for
// the q/k/v/o/gate/up/down projections, lm_head and the projector when `--unfreeze_language_model` trains them, model_arch.py:33-51).
//
// Round 1/2 served these products with the K-contiguous kernel of gemm_fast.hip after TRANSPOSING the activations (two passes over
// dY and X per weight: 17 ms of a 247-ms unfrozen step) and keeping a refreshed W^T of every trainable weight (5 ms per step, +13.5 GB).
// Here the k-major operand is staged as it lies in memory — an LDS image [64 k][128 rows] of 256-B rows, filled by LDS-DMA in whole
// 256-B row pieces — and turned on the way out of LDS by ds_read_b64_tr_b16 (cdna_hip_programming.md T10: per 16-lane group a 4-row x
// 16-column block delivered column-major; two of them make the 8 consecutive k of a v_mfma_f32_16x16x32_bf16 operand lane).  Image (b) of
// T10 — off(row, ch) = 256 row + 16 (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) — makes those reads conflict-free (a half's two blocks
// are 8 rows apart in the same columns).  Everything else is the 8-phase schedule of gemm_nt_bf16_8phase_kernel: 2 buffers x 4 half-tiles
// of 16 KB, every operand byte by LDS-DMA that stays in flight across raw s_barriers (counted vmcnt), two wave groups one barrier apart.
// Differences forced by the image: a half-tile is 128 CONSECUTIVE rows (columns) of the tile, so wave row wr owns rows
// {128 h + 64 wr + 0..63 : h = 0, 1} and wave column wc owns columns {128 h + 32 wc + 0..31 : h = 0, 1}; the LDS-DMA is issued from inline
// asm (SGPR base + per-lane 32-bit byte offset), because with a DMA builtin in the kernel hipcc puts `s_waitcnt vmcnt(0)` in front of
// every ds_read_b64_tr_b16 (no alias information), which would drain the prefetch.
// A reduction length that is not a multiple of 64 (M = 5536 rows = 86.5 K-tiles) needs no padded copy: the rows past the end are
// fetched from a zero page (LDS-DMA cannot mask, but every lane's source address is free).
#include "common.h"

#define TN_BK 64
#define TN_HT (128 * 64)                       // elements of one half-tile image (16 KB)

struct TnArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K;                               // C is [M, N]; K = reduction length
    long long lda, ldb, ldc;
    int accumulate;
    int tiles_m, tiles_n;
    int full_tm, full_tiles, tail_s;           // ragged last round (gemm_fast.hip plan_tail): tile rows >= full_tm are cut into tail_s K-slices,
    float* ws;                                 // whose fp32 partial sums go to ws [tail_s][M - 256 full_tm][N] (summed by splitk_reduce_kernel)
};

// 256 B of zeros for the reduction rows past the end of a ragged last K-tile (addressed from device code: no host-side symbol lookup)
__device__ __attribute__((aligned(256))) uint32_t g_tn_zero_page[64];

__device__ __forceinline__ void tn_dma16(const void* gbase, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(gbase), "s"(lds_base) : "memory");     // (m0 is a reserved register: hipcc re-loads it in front of each of its own uses)
}
typedef __attribute__((ext_vector_type(4))) __bf16 tn_bf16x4;
typedef __attribute__((address_space(3))) tn_bf16x4 tn_lds_bf16x4;
typedef __attribute__((address_space(3))) void tn_lds_void;

// One operand's side of the kernel.  T = k-major ([K, rows], row stride ld) or k-contiguous ([rows, K]).
//   half-tile image, T:  [64 k][128 rows] bf16, 256-B rows, chunk swizzle (b)   — read by two ds_read_b64_tr_b16 per fragment
//                  !T:  [128 rows][64 k] bf16, 128-B rows, chunk ^ ((row>>1)&7)  — read by one ds_read_b128 per fragment
template <bool T>
struct TnSide {
    uint32_t voff[2][2];                       // [half][piece] per-lane DMA source offset (bytes from the operand's tile-t base)
    uint32_t rd[T ? 8 : 2];                    // fragment read byte offsets inside a half-tile (T: [block(ii|jj) * 2 + e]; !T: [ks])

    // rows0 = first row (column) of the tile, nrows = valid rows of the operand, ld in elements, sub = wave row / wave column,
    // sub_stride = rows a wave owns per half (64 for A, 32 for B)
    __device__ __forceinline__ void init(int rows0, int nrows, long long ld, int wave, int lane, int sub, int sub_stride) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = wave * 2 + i;
                if (T) {
                    const int r = piece * 4 + (lane >> 4);                        // k-row of the image
                    const int ch = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
                    long long col = rows0 + 128 * h + 8 * ch;
                    // stay inside the OPERAND's own width, not the row stride: a column-sliced view (dgu[:, Fd:], dqkv[:, 2d:]) starts inside a
                    // wider row, so `ld - 8` from its base runs past the row end, and on the last reduction row past the END of the allocation
                    // (the class of the dK/dV over-read of round 3).  Out-of-range outputs are masked on store.
                    const long long lim = (long long)((nrows + 7) & ~7) - 8;
                    col = col < lim ? col : lim;
                    voff[h][i] = (uint32_t)(((long long)r * ld + col) * 2);
                } else {
                    const int lr = piece * 8 + (lane >> 3);                       // row of the image = tile row 128 h + lr
                    const int ch = (lane & 7) ^ ((lr >> 1) & 7);
                    int row = rows0 + 128 * h + lr;
                    row = row < nrows ? row : nrows - 1;
                    voff[h][i] = (uint32_t)(((long long)row * ld + ch * 8) * 2);
                }
            }
        if (T) {
            const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int row = 8 * g + q + 4 * e;                            // + 32 ks: an immediate
                    const int ch = (sub * sub_stride + 16 * blk) / 8 + (p >> 1);
                    rd[blk * 2 + e] = 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (p & 1);
                }
        } else {
            const int sw = ((lane & 15) >> 1) & 7, c0 = lane >> 4;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) rd[ks] = ((sub * sub_stride + (lane & 15)) * 64 + (((c0 + 4 * ks) ^ sw) << 3)) * 2;
        }
    }
    // fragment (16 rows starting at block blk of this wave's rows) x (32 k of k-step ks) of the half-tile at LDS byte address `ht`
    __device__ __forceinline__ bf16x8 frag(const char* ht, int blk, int ks) const {
        if (T) {
            const tn_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tn_lds_bf16x4*)(ht + rd[blk * 2] + ks * (32 * 256)));
            const tn_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tn_lds_bf16x4*)(ht + rd[blk * 2 + 1] + ks * (32 * 256)));
            bf16x8 r;
            r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
            r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
            return r;
        }
        return *reinterpret_cast<const bf16x8*>(ht + rd[ks] + blk * (16 * 128));
    }
};

template <typename TC, bool TA, bool TB>
__global__ __launch_bounds__(512, 2)
void gemm_bf16_8phase_t_kernel(TnArgs g) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 4 * TN_HT];            // [buffer][A-h0, A-h1, B-h0, B-h1]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // XCD-aware grouped tile order (as gemm_nt_bf16_8phase_kernel): 8 M-tiles x consecutive N-tiles run together on one XCD
    // whole tiles first, then the K-slices of the tail rows: the dispatcher hands blocks out in index order, so the short blocks fill the
    // ragged last round
    int tm, tn, kz = 0, ksl = 1;
    if ((int)blockIdx.x < g.full_tiles) {
        const int nwg = g.full_tiles;
        int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int per_group = 8 * g.tiles_n;
        const int grp = bid / per_group, first_tm = grp * 8;
        const int gsz = (g.full_tm - first_tm) < 8 ? (g.full_tm - first_tm) : 8;
        const int in_g = bid - grp * per_group;
        tm = first_tm + in_g % gsz; tn = in_g / gsz;
    } else {
        const int idx = blockIdx.x - g.full_tiles, rows = g.tiles_m - g.full_tm;
        ksl = g.tail_s;
        kz = idx % ksl;
        const int tile = idx / ksl;
        tm = g.full_tm + tile % rows; tn = tile / rows;
    }
    tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
    kz = __builtin_amdgcn_readfirstlane(kz); ksl = __builtin_amdgcn_readfirstlane(ksl);
    const int m0 = tm * 256, n0 = tn * 256;

    TnSide<TA> sa;
    TnSide<TB> sb;
    sa.init(m0, g.M, g.lda, wave, lane, wr, 64);
    sb.init(n0, g.N, g.ldb, wave, lane, wc, 32);

    f32x4 acc[4][8];                                                               // [nh * 2 + jj][mh * 4 + ii]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this block's part of the reduction: everything, or K-slice kz of ksl (whole K-tiles; the ragged end belongs to the last slice)
    int Kl = g.K;
    const bf16_t* gA = g.A;
    const bf16_t* gB = g.B;
    if (ksl > 1) {
        const int nt_all = (g.K + TN_BK - 1) / TN_BK, per = (nt_all + ksl - 1) / ksl;
        const int tb = kz * per, te = tb + per < nt_all ? tb + per : nt_all;
        gA += (long long)tb * (TA ? (long long)TN_BK * g.lda : TN_BK);
        gB += (long long)tb * (TB ? (long long)TN_BK * g.ldb : TN_BK);
        Kl = (te * TN_BK < g.K ? te * TN_BK : g.K) - tb * TN_BK;
    }
    const int nt = (Kl + TN_BK - 1) / TN_BK, t_last = nt - 1;
    const int k_tail = Kl - t_last * TN_BK;                                        // valid k-rows of the last K-tile (64 unless ragged)
    const uint32_t lds0 = (uint32_t)(uintptr_t)((tn_lds_void*)smem);
    const char* sm = reinterpret_cast<const char*>(smem);

    // DMA of one half-tile (slot 0/1 = A-h0/h1, 2/3 = B-h0/h1) of the K-tile whose operand base pointer is `base` into buffer b: two 1-KiB
    // pieces per wave.  Everything but the two per-lane offsets is scalar and set up outside (the first version of this kernel spent
    // 75 SALU instructions per K-tile and wave on it against 19 in gemm_nt_bf16_8phase_kernel and ran at 0.91 PFLOP/s; PMC: tools/debug/tn_pmc_probe.py).
    // `tail` (wave-uniform, rare): the K-tile is the ragged last one — 4-row pieces past the end of the reduction come from the zero page
    // (K % 4 == 0: a piece is never split).  k-contiguous operands (K % 64 == 0 there) never take it.
    const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + wave * 2048);
    const void* zero_page = g_tn_zero_page;
    const uint32_t zoff = (uint32_t)((lane & 15) * 16);
    const bool ragged = k_tail != TN_BK;
    const long long strideA = TA ? (long long)TN_BK * g.lda : TN_BK, strideB = TB ? (long long)TN_BK * g.ldb : TN_BK;      // elements per K-tile
#define TN_PF(b, slot, base, off, tail) { \
        const uint32_t dst_ = wave_lds + ((b) * 4 + (slot)) * (TN_HT * 2); \
        if (!(tail)) { tn_dma16(base, (off)[0], dst_); tn_dma16(base, (off)[1], dst_ + 1024); } \
        else { \
            if ((wave * 2) * 4 >= k_tail) tn_dma16(zero_page, zoff, dst_); else tn_dma16(base, (off)[0], dst_); \
            if ((wave * 2 + 1) * 4 >= k_tail) tn_dma16(zero_page, zoff, dst_ + 1024); else tn_dma16(base, (off)[1], dst_ + 1024); \
        } }

    bf16x8 fa[2][4], fb0[2][2], fb1[2][2];
#define TN_HTADDR(b, slot) (sm + ((b) * 4 + (slot)) * (TN_HT * 2))
#define TN_LDA(b, X) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) fa[ks][ii] = sa.frag(TN_HTADDR(b, X), ii, ks);
#define TN_LDB(dst, b, X) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) dst[ks][jj] = sb.frag(TN_HTADDR(b, 2 + (X)), jj, ks);
#define TN_MMA(mh, fbv, nh) __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) \
            acc[(nh) * 2 + jj][(mh) * 4 + ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbv[ks][jj], fa[ks][ii], acc[(nh) * 2 + jj][(mh) * 4 + ii], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);
#define TN_BAR __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
    // B-h0's reads are retired before the ph1 barrier (its slot is refilled in ph2): the A reads issued behind them number 8 (ds_read_b128)
    // or 16 (tr reads; lgkmcnt holds 15 at most)
#define TN_WAIT_B0 if (TA) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    // one K-tile from buffer b: pA1 = base of A's K-tile t+1, pA2 / pB2 = bases of K-tile t+2 (clamped to the last tile by the caller: DMAs past the
    // end re-load it into buffers nobody reads any more, which keeps the vmcnt arithmetic constant); tl1 / tl2: that tile is the ragged last one
#define TN_TILE_P(b, pA1, pA2, pB2, tl1, tl2) { \
        /* ph1 */ TN_LDB(fb0, b, 0) __builtin_amdgcn_sched_barrier(0); TN_LDA(b, 0) TN_PF((b) ^ 1, 1, pA1, sa.voff[1], TA && (tl1)) TN_WAIT_B0 TN_BAR TN_MMA(0, fb0, 0) TN_BAR \
        /* ph2 */ TN_LDB(fb1, b, 1) TN_PF(b, 2, pB2, sb.voff[0], TB && (tl2)) TN_BAR TN_MMA(0, fb1, 1) TN_BAR \
        /* ph3 */ TN_LDA(b, 1) TN_PF(b, 0, pA2, sa.voff[0], TA && (tl2)) TN_BAR TN_MMA(1, fb1, 1) TN_BAR \
        /* ph4 */ TN_PF(b, 3, pB2, sb.voff[1], TB && (tl2)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); TN_BAR TN_MMA(1, fb0, 0) TN_BAR \
    }
#define TN_TILE_TAIL(b, tt) { \
        const int t1 = (tt) + 1 < t_last ? (tt) + 1 : t_last, t2 = (tt) + 2 < t_last ? (tt) + 2 : t_last; \
        const bf16_t* pA1 = gA + (long long)t1 * strideA; \
        const bf16_t* pA2 = gA + (long long)t2 * strideA; \
        const bf16_t* pB2 = gB + (long long)t2 * strideB; \
        const bool tl1 = ragged && t1 == t_last, tl2 = ragged && t2 == t_last; \
        TN_TILE_P(b, pA1, pA2, pB2, tl1, tl2) \
    }

    // ---- prologue: K-tile 0 complete, three half-tiles of the next one in flight
    {
        const int t1 = 1 < t_last ? 1 : t_last;
        const bf16_t* pA1 = gA + (long long)t1 * strideA;
        const bf16_t* pB1 = gB + (long long)t1 * strideB;
        const bool tl0 = ragged && t_last == 0, tl1 = ragged && t1 == t_last;
        TN_PF(0, 2, gB, sb.voff[0], TB && tl0) TN_PF(0, 0, gA, sa.voff[0], TA && tl0) TN_PF(0, 3, gB, sb.voff[1], TB && tl0) TN_PF(0, 1, gA, sa.voff[1], TA && tl0)
        TN_PF(1, 2, pB1, sb.voff[0], TB && tl1) TN_PF(1, 0, pA1, sa.voff[0], TA && tl1) TN_PF(1, 3, pB1, sb.voff[1], TB && tl1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        TN_BAR
    }
    if (wr == 1) { TN_BAR }                                                        // second wave group runs one barrier behind
    // main loop: K-tiles whose two prefetch targets are whole tiles inside the reduction — running base pointers, no clamp, no tail logic in
    // the instruction stream; the last three or four tiles (where the clamp and the ragged K-tile matter) run the general form
    int t = 0;
    const int nt_main = (nt - 3 > 0 ? nt - 3 : 0) & ~1;
    {
        const bf16_t* pa = gA;
        const bf16_t* pb = gB;
        for (; t < nt_main; t += 2) {
            TN_TILE_P(0, pa + strideA, pa + 2 * strideA, pb + 2 * strideB, false, false)
            TN_TILE_P(1, pa + 2 * strideA, pa + 3 * strideA, pb + 3 * strideB, false, false)
            pa += 2 * strideA; pb += 2 * strideB;
        }
    }
    for (; t + 1 < nt; t += 2) {
        TN_TILE_TAIL(0, t)
        TN_TILE_TAIL(1, t + 1)
    }
    if (t < nt) TN_TILE_TAIL(0, t)
    if (wr == 0) { TN_BAR }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               // the tail's redundant DMAs drain before the block ends
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: acc[nh*2+jj][mh*4+ii][r] = C[m][n + r],  m = m0 + 128 mh + 64 wr + 16 ii + (lane & 15),
    //                                                       n = n0 + 128 nh + 32 wc + 16 jj + 4 (lane >> 4)
    if (ksl > 1) {
        // K-slice of a tail tile: fp32 partial sums into this slice's slab (rows relative to the first tail row, row stride N)
        const int row0 = g.full_tm * 256;
        float* slab = g.ws + (long long)kz * (g.M - row0) * g.N;
        const bool vec4 = (g.N & 3) == 0;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int m = m0 + 128 * mh + 64 * wr + 16 * ii + (lane & 15);
                if (m >= g.M) continue;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int n = n0 + 128 * nh + 32 * wc + 16 * jj + 4 * (lane >> 4);
                        if (n >= g.N) continue;
                        const f32x4 v = acc[nh * 2 + jj][mh * 4 + ii];
                        float* sp = slab + (long long)(m - row0) * g.N + n;
                        if (vec4) *reinterpret_cast<f32x4*>(sp) = v;
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (n + r < g.N) sp[r] = v[r];
                        }
                    }
            }
        return;
    }
    TC* C = reinterpret_cast<TC*>(g.C);
    const bool vec = (g.ldc & 3) == 0;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int m = m0 + 128 * mh + 64 * wr + 16 * ii + (lane & 15);
            if (m >= g.M) continue;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int n = n0 + 128 * nh + 32 * wc + 16 * jj + 4 * (lane >> 4);
                    if (n >= g.N) continue;
                    f32x4 v = acc[nh * 2 + jj][mh * 4 + ii];
                    TC* cp = C + (long long)m * g.ldc + n;
                    if (vec && n + 3 < g.N) {
                        if (sizeof(TC) == 4) {
                            if (g.accumulate) v += *reinterpret_cast<const f32x4*>(cp);
                            *reinterpret_cast<f32x4*>(cp) = v;
                        } else {
                            if (g.accumulate) {
                                const u32x2 cc = *reinterpret_cast<const u32x2*>(cp);
                                v[0] += __uint_as_float(cc[0] << 16); v[1] += __uint_as_float(cc[0] & 0xFFFF0000u);
                                v[2] += __uint_as_float(cc[1] << 16); v[3] += __uint_as_float(cc[1] & 0xFFFF0000u);
                            }
                            u32x2 o;
                            o[0] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                            o[1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                            *reinterpret_cast<u32x2*>(cp) = o;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (n + r >= g.N) continue;
                            float x = v[r];
                            if (g.accumulate) x += Cvt<TC>::ld(cp + r);
                            Cvt<TC>::st(cp + r, x);
                        }
                    }
                }
        }
}

// gemm_fast.hip
int egomi_gemm_tn_rest(const egomi_gemm_desc* d, hipStream_t s);
int egomi_tall_kmajor_try(const egomi_gemm_desc* d, hipStream_t s, bool query);    // data gradients: the 352x256 form where whole rounds pay (0 = taken)
void egomi_plan_tail_rows(int M, int N, int K, long long ws_bytes, int* rows, int* slices);
int egomi_splitk_reduce_rows(void* C, int c_dtype, long long ldc, int rows, int N, const float* ws, int slices, int accumulate, hipStream_t s);

// kernel id reported by egomi_gemm_kernel_id for this kernel
#define TN_KERNEL_ID 3

static bool tn_applicable(const egomi_gemm_desc* d) {
    static int on = -1;                                                            // EGOMI_GEMM_TN=0: k-major operands go back to the generic kernel (A/B runs)
    if (on < 0) { const char* e = getenv("EGOMI_GEMM_TN"); on = e ? atoi(e) : 1; }
    if (!on || d->force_generic) return false;
    if (d->ab_dtype != EGOMI_BF16 || d->batch > 1 || d->b_layout != 1) return false;      // (a_layout, b_layout) = (1, 1) or (0, 1)
    if (d->bias || d->residual || d->act != 0 || d->alpha != 1.0f || d->epilogue != EGOMI_EPI_NONE) return false;
    if (d->c_dtype != EGOMI_BF16 && d->c_dtype != EGOMI_F32) return false;
    if (d->K < 256 || (d->K & 3)) return false;                                    // 4-row DMA pieces must not straddle the end of the reduction
    if (d->a_layout == 0 && (d->K % TN_BK)) return false;                          // a k-contiguous operand has no zero rows to borrow
    if ((d->lda & 7) || (d->ldb & 7) || (((uintptr_t)d->A | (uintptr_t)d->B) & 15)) return false;
    if (d->a_layout == 1 ? d->lda < 8 : d->lda < d->K) return false;
    if (d->ldb < 8) return false;
    // a k-major operand's 16-B DMA chunks are clamped to its own width rounded up to 8 (TnSide::init): that stays inside the operand's row —
    // also for a column slice of a wider array, whose base and row stride are multiples of 8 elements — as long as the row is that long
    if (d->ldb < ((d->N + 7) & ~7) || (d->a_layout == 1 && d->lda < ((d->M + 7) & ~7))) return false;
    const int esz = d->c_dtype == EGOMI_BF16 ? 2 : 4;
    if ((uintptr_t)d->C % (4 * esz)) return false;
    // 32-bit byte offsets inside one K-tile's rows (+ the tile's columns)
    const long long spanA = d->a_layout == 1 ? (long long)TN_BK * d->lda : (long long)d->M * d->lda;
    const long long spanB = (long long)TN_BK * d->ldb;
    if (spanA * 2 >= (1ll << 32) || spanB * 2 >= (1ll << 32)) return false;
    const long long tiles = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256);
    return tiles >= 64;                                                            // fewer tiles: the generic kernel's smaller tiles fill the chip better
}

// ragged last round: the last `rows` tile rows as `slices` K-slices (the plan of the K-contiguous kernel, gemm_fast.hip plan_tail; slabs behind
// the ticket words of the caller's scratch).  rows = 0: every tile whole.
static void tn_tail(const egomi_gemm_desc* d, int* rows, int* slices, float** slabs) {
    *rows = 0; *slices = 1; *slabs = nullptr;
    static int no_tail = -1;
    if (no_tail < 0) { const char* e = getenv("EGOMI_GEMM_NO_TAIL"); no_tail = e ? atoi(e) : 0; }
    if (!d->workspace || no_tail) return;
    char* wsp = (char*)d->workspace;
    long long wsb = d->workspace_bytes;
    if (d->ws_tickets_zeroed) { wsp += 4096; wsb -= 4096; }
    if (wsb <= 0 || (((uintptr_t)wsp) & 15)) return;
    int r = 0, sl = 1;
    egomi_plan_tail_rows(d->M, d->N, d->K, wsb, &r, &sl);
    const int nt = (d->K + TN_BK - 1) / TN_BK;
    if (r <= 0 || sl < 2 || nt / sl < 8) return;                                   // (every slice keeps whole K-tiles to work on)
    *rows = r; *slices = sl; *slabs = (float*)wsp;
}

// 0 and the plan of the launch egomi_gemm would make for this descriptor (*slices = 0: no K-sliced rows), or EGOMI_E_UNSUPPORTED
extern "C" int egomi_gemm_tn_tail_plan(const egomi_gemm_desc* d, int* row0, int* slices) {
    if (!d || !row0 || !slices) return EGOMI_E_BADARG;
    if (!tn_applicable(d)) return EGOMI_E_UNSUPPORTED;
    if (egomi_tall_kmajor_try(d, nullptr, true) == 0) { *row0 = d->M; *slices = 0; return EGOMI_OK; }     // one whole round of 352x256 tiles: no K-sliced rows
    int r, sl; float* w;
    tn_tail(d, &r, &sl, &w);
    *row0 = r ? ((d->M + 255) / 256 - r) * 256 : d->M;
    *slices = r ? sl : 0;
    return EGOMI_OK;
}

extern "C" int egomi_gemm_tn_kernel_id(const egomi_gemm_desc* d) { return (d && tn_applicable(d)) ? TN_KERNEL_ID : 0; }

// returns 0 on success, < 0 on error, 1 when this kernel does not apply
int egomi_gemm_tn_try(const egomi_gemm_desc* d, hipStream_t s) {
    if (!tn_applicable(d)) return 1;
    { const int rc = egomi_tall_kmajor_try(d, s, false); if (rc <= 0) return rc; }
    return egomi_gemm_tn_rest(d, s);
}

// this file's 256x256 kernel (also the second part of a product whose first columns took the 352x256 form: gemm_fast.hip egomi_tall_kmajor_try)
int egomi_gemm_tn_rest(const egomi_gemm_desc* d, hipStream_t s) {
    if (!tn_applicable(d)) return EGOMI_E_UNSUPPORTED;
    TnArgs g;
    g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B; g.C = d->C;
    g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.accumulate = d->accumulate;
    g.tiles_m = (d->M + 255) / 256; g.tiles_n = (d->N + 255) / 256;
    g.full_tm = g.tiles_m; g.full_tiles = g.tiles_m * g.tiles_n; g.tail_s = 1; g.ws = nullptr;
    int rows = 0, slices = 1;
    float* slabs = nullptr;
    tn_tail(d, &rows, &slices, &slabs);
    if (rows > 0) { g.full_tm = g.tiles_m - rows; g.full_tiles = g.full_tm * g.tiles_n; g.tail_s = slices; g.ws = slabs; }
    const dim3 grid(g.full_tiles + rows * g.tiles_n * slices), block(512);
    const bool f32 = d->c_dtype == EGOMI_F32;
    if (d->a_layout == 1) {
        if (f32) EGOMI_LAUNCH((gemm_bf16_8phase_t_kernel<float, true, true>), grid, block, 0, s, g);
        else EGOMI_LAUNCH((gemm_bf16_8phase_t_kernel<bf16_t, true, true>), grid, block, 0, s, g);
    } else {
        if (f32) EGOMI_LAUNCH((gemm_bf16_8phase_t_kernel<float, false, true>), grid, block, 0, s, g);
        else EGOMI_LAUNCH((gemm_bf16_8phase_t_kernel<bf16_t, false, true>), grid, block, 0, s, g);
    }
    if (rows > 0) {
        const long long row0 = (long long)g.full_tm * 256;
        const int esz = f32 ? 4 : 2;
        return egomi_splitk_reduce_rows((char*)d->C + row0 * d->ldc * esz, d->c_dtype, d->ldc, d->M - (int)row0, d->N, g.ws, slices, d->accumulate, s);
    }
    return egomi_launch_status();
}
