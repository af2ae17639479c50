// Tuned bf16 "NT" GEMMs for the LLaMA-sized projections:  C[M,N] = A[M,K] . B[N,K]^T  (+ epilogue)
//   both operands K-contiguous (activation x nn.Linear weight; dgrad uses pre-transposed weights, wgrad uses
//   transposed activations, so every big product of the path has this form).  Two kernels:
//   (1) gemm_nt_bf16_8phase_kernel — 256x256x64 tile, 8 waves, LDS-DMA kept in flight across raw barriers, two wave
//       groups one barrier apart (see its header below).  >= 128 tiles and K >= 2048: every large product of the step.
//       Ragged last round: K-sliced tail rows (plan_tail) or nothing; plain bf16 tiles leave through LDS as row segments.
//   (2) gemm_nt_bf16_kernel<BM,BN> — 128x128 (4 waves, 32 KB LDS, 4 blocks/CU) or 256x128 (8 waves, 48 KB), one wave
//       per 64x64 sub-tile, two barriers per K-step (~1.0 PFLOP/s ceiling).  Small / short-K products and the
//       M <= 512 split-K decode path.
//   Common to both:
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write
//   * LDS image linear [row][64 k] (128-B rows); bank conflicts removed by an XOR swizzle applied to
//     the per-lane SOURCE address and to the fragment read (chunk ^= (row>>1)&7): the 16 rows of a
//     fragment land on 16 distinct 16-B slots of the 256-B bank row (cdna_hip_programming.md rule 21);
//     measured SQ_LDS_BANK_CONFLICT = 0
//   * operands are fed swapped (weights as the MFMA A operand) so each lane's 4 accumulators are 4
//     consecutive output columns
//   * rows beyond M / N are clamped on load and masked on store; K % 64 == 0
//   * XCD-aware block order: each XCD walks a contiguous strip of tiles, grouped 8 M-tiles deep (T1)
#include "common.h"
#include <mutex>
#include <queue>
#include <unordered_map>
#include <vector>

#define FT_BK 64

struct FastArgs {
    const bf16_t* A; const bf16_t* B; void* C; const bf16_t* bias; const void* residual;
    int M, N, K;
    long long lda, ldb, ldc, ldr;
    float alpha; int accumulate; int act;
    int tiles_m, tiles_n;
    int splitk; float* ws;           // splitk > 1: block (tile, blockIdx.y) multiplies its K slice and stores a raw fp32 slab
    int full_tm, full_tiles, tail_s; // 8-phase kernel: M-tile rows >= full_tm are cut into tail_s K-slices (fp32 slabs of those rows only)
    int epi; void* C2; long long ldc2;   // epi 1 (EGOMI_EPI_SWIGLU): C is interleaved-32 gate|up, C2 [M, N/2] receives silu(gate)*up (bf16 only)
                                         // epi 3 (EGOMI_EPI_SWIGLU_BWD): the product is d(act) [M, N]; C2 = gate|up [M, 2N] (read), C = d(gate|up) [M, 2N]
    int* tickets;                    // non-null: the K-slices of a tail tile are summed INSIDE the launch by the slice block that arrives last
                                     // (one ticket word per tail tile, zero on entry, returned to zero); ws then holds register-major slabs
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ void glds16(const bf16_t* g, bf16_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}

// ---- interior fast path of the epilogue: whole sub-tile in bounds, alpha = 1, no bias / activation, 4-aligned leading
//      dimensions.  One pointer per lane, rows advance by a constant step, columns are immediates: a handful of
//      instructions per 4 outputs instead of the general path's per-element checks (which cost ~20 us per 256x256 tile).
template <typename TC, int MT, bool RES, bool ACC>
__device__ __forceinline__ void gemm_epilogue_interior(const FastArgs& g, const f32x4 (&acc)[4][MT], const int mb, const int nb, const int lane) {
    TC* cp = reinterpret_cast<TC*>(g.C) + (long long)(mb + (lane & 15)) * g.ldc + nb + (lane >> 4) * 4;
    const TC* rp = RES ? reinterpret_cast<const TC*>(g.residual) + (long long)(mb + (lane & 15)) * g.ldr + nb + (lane >> 4) * 4 : nullptr;
    const long long cstep = 16 * g.ldc, rstep = 16 * g.ldr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[j][i];
            if (sizeof(TC) == 2) {
                if (RES) {
                    const u32x2 rr = *reinterpret_cast<const u32x2*>(rp + j * 16);
                    v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xFFFF0000u);
                    v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xFFFF0000u);
                }
                if (ACC) {
                    const u32x2 cc = *reinterpret_cast<const u32x2*>(cp + j * 16);
                    v[0] += __uint_as_float(cc[0] << 16); v[1] += __uint_as_float(cc[0] & 0xFFFF0000u);
                    v[2] += __uint_as_float(cc[1] << 16); v[3] += __uint_as_float(cc[1] & 0xFFFF0000u);
                }
                u32x2 o;
                o[0] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                o[1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                *reinterpret_cast<u32x2*>(cp + j * 16) = o;
            } else {
                if (RES) v += *reinterpret_cast<const f32x4*>(rp + j * 16);
                if (ACC) v += *reinterpret_cast<const f32x4*>(cp + j * 16);
                *reinterpret_cast<f32x4*>(cp + j * 16) = v;
            }
        }
        cp += cstep;
        if (RES) rp += rstep;
    }
}

// ---- shared epilogue.  acc[j][i][r] = C[m][n], n = nb+16j+4*(lane>>4)+r, m = mb+16i+(lane&15)
//      (nb, mb: first column / row of the wave's sub-tile)
template <typename TC, int MT>
__device__ __forceinline__ void gemm_epilogue(const FastArgs& g, const f32x4 (&acc)[4][MT], const int mb, const int nb, const int lane,
                                              float* slab = nullptr, const int slab_row0 = 0) {
    if (slab) {                                                     // raw fp32 partial sums, rows relative to slab_row0
        slab -= (long long)slab_row0 * g.N;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nb + j * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = mb + i * 16 + (lane & 15);
                if (m >= g.M) continue;
                if (n + 3 < g.N && (g.N & 3) == 0) {
                    *reinterpret_cast<f32x4*>(slab + (long long)m * g.N + n) = acc[j][i];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < g.N) slab[(long long)m * g.N + n + r] = acc[j][i][r];
                }
            }
        }
        return;
    }
    TC* C = reinterpret_cast<TC*>(g.C);
    const TC* R = reinterpret_cast<const TC*>(g.residual);
    const bool vec_ok = (g.ldc % 4 == 0) && (!R || g.ldr % 4 == 0);
    if (vec_ok && !g.bias && g.act == 0 && g.alpha == 1.0f && mb + 16 * MT <= g.M && nb + 64 <= g.N) {   // wave-uniform
        if (R) { if (g.accumulate) gemm_epilogue_interior<TC, MT, true, true>(g, acc, mb, nb, lane);
                 else              gemm_epilogue_interior<TC, MT, true, false>(g, acc, mb, nb, lane); }
        else   { if (g.accumulate) gemm_epilogue_interior<TC, MT, false, true>(g, acc, mb, nb, lane);
                 else              gemm_epilogue_interior<TC, MT, false, false>(g, acc, mb, nb, lane); }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = nb + j * 16 + (lane >> 4) * 4;
        if (n >= g.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < g.N) bv[r] = bf2f(g.bias[n + r]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = mb + i * 16 + (lane & 15);
            if (m >= g.M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_apply(acc[j][i][r] * g.alpha + bv[r], g.act);
            TC* cp = C + (long long)m * g.ldc + n;
            const TC* rp = R ? R + (long long)m * g.ldr + n : nullptr;
            if (n + 3 < g.N && vec_ok) {
                if (sizeof(TC) == 2) {
                    if (rp) {
                        const u32x2 rr = *reinterpret_cast<const u32x2*>(rp);
                        v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xFFFF0000u);
                        v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xFFFF0000u);
                    }
                    if (g.accumulate) {
                        const u32x2 cc = *reinterpret_cast<const u32x2*>(cp);
                        v[0] += __uint_as_float(cc[0] << 16); v[1] += __uint_as_float(cc[0] & 0xFFFF0000u);
                        v[2] += __uint_as_float(cc[1] << 16); v[3] += __uint_as_float(cc[1] & 0xFFFF0000u);
                    }
                    u32x2 o;
                    o[0] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                    o[1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                    *reinterpret_cast<u32x2*>(cp) = o;
                } else {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    if (rp) o += *reinterpret_cast<const f32x4*>(rp);
                    if (g.accumulate) o += *reinterpret_cast<const f32x4*>(cp);
                    *reinterpret_cast<f32x4*>(cp) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= g.N) continue;
                    float x = v[r];
                    if (rp) x += Cvt<TC>::ld(rp + r);
                    if (g.accumulate) x += Cvt<TC>::ld(cp + r);
                    Cvt<TC>::st(cp + r, x);
                }
            }
        }
    }
}

// NS = LDS stages.  1: load, barrier, multiply, barrier.  2: the DMA of K-step t+1 is issued before the MFMAs of step t and one
// barrier per step remains (the guide's "glds, 2 LDS buffers, BK=64, vmcnt(0) + plain __syncthreads()" row) — the default of
// the 128x128 tile: M = 256 decode projections 27 -> 22 us (N = K = 4096), 46 -> 41 (N = 11008), 41 -> 37 (K = 11008), lm_head
// 98 -> 91; decode step 20.0 -> 19.5 ms.  A 4-stage ring (three K-steps in flight, counted vmcnt, raw barrier, one block per
// CU) was measured SLOWER than two blocks per CU with two stages each (25 / 63 / 52 / 118 us, step 20.9 ms): at M = 256 the
// second resident block is worth more than deeper prefetch.
template <typename TC, int BM, int BN, int NS>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64, NS == 2 ? 2 : ((BM * BN == 256 * 128) ? 4 : 3))
void gemm_nt_bf16_kernel(FastArgs g) {
    constexpr int MT = 4;                                                // 16-row MFMA tiles per wave in M (64 x 64 per wave)
    constexpr int WN = BN / 64, NW = (BM / 64) * WN;
    constexpr int A_PW = BM / 8 / NW, B_PW = BN / 8 / NW;               // 1-KiB DMA pieces (8 rows x 128 B) per wave
    constexpr int STAGE = (BM + BN) * FT_BK;
    constexpr bool PF = NS == 2;
    __shared__ __attribute__((aligned(16))) bf16_t smem[NS * STAGE];
    bf16_t* sA = smem;
    bf16_t* sB = smem + BM * FT_BK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- XCD-aware tile order (bijective for any grid size)
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // grouped order inside the strip: 8 M-tiles x consecutive N-tiles run together, so the tiles an XCD
    // has in flight form a block that re-uses both A rows and B panels from its L2
    const int per_group = 8 * g.tiles_n;
    const int grp = bid / per_group, first_tm = grp * 8;
    const int gsz = (g.tiles_m - first_tm) < 8 ? (g.tiles_m - first_tm) : 8;
    const int in_g = bid - grp * per_group;
    const int tm = first_tm + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane LDS-DMA source pointers
    const bf16_t* srcA[A_PW];
    const bf16_t* srcB[B_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int r = (wave * A_PW + i) * 8 + (lane >> 3);          // row inside the tile
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);              // swizzled 16-B chunk of the row
        int ra = m0 + r; ra = ra < g.M ? ra : g.M - 1;
        srcA[i] = g.A + (long long)ra * g.lda + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
        const int r = (wave * B_PW + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int rb = n0 + r; rb = rb < g.N ? rb : g.N - 1;
        srcB[i] = g.B + (long long)rb * g.ldb + chunk * 8;
    }

    const int wm = (wave / WN) * (16 * MT), wn = (wave % WN) * 64;
    f32x4 acc[4][MT];                                               // [n-tile j][m-tile i]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int offA[MT], offB[4];
#pragma unroll
    for (int i = 0; i < MT; ++i) offA[i] = (wm + i * 16 + (lane & 15)) * FT_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) offB[i] = (wn + i * 16 + (lane & 15)) * FT_BK;
    const int sw = ((lane & 15) >> 1) & 7;                          // (row>>1)&7: wm, wn, 16*i are multiples of 16
    const int c0 = lane >> 4;

    int nt = g.K / FT_BK, t_begin = 0;
    if (g.splitk > 1) {                                             // this block's K slice (whole 64-deep steps)
        const int per = (nt + g.splitk - 1) / g.splitk;
        t_begin = blockIdx.y * per;
        nt = t_begin + per < nt ? t_begin + per : nt;
    }
    if (PF) {
#pragma unroll
        for (int i = 0; i < A_PW; ++i) glds16(srcA[i] + t_begin * FT_BK, sA + (wave * A_PW + i) * 8 * FT_BK);
#pragma unroll
        for (int i = 0; i < B_PW; ++i) glds16(srcB[i] + t_begin * FT_BK, sB + (wave * B_PW + i) * 8 * FT_BK);
    }
    for (int t = t_begin; t < nt; ++t) {
        const int k0 = t * FT_BK;
        const int cur = PF ? ((t - t_begin) & 1) * STAGE : 0;
        const bf16_t* cA = sA + cur;
        const bf16_t* cB = sB + cur;
        if (PF) {
            __syncthreads();                                        // emits vmcnt(0): step t has landed; and every wave is done with the other stage
            if (t + 1 < nt) {
                const int nxt = STAGE - cur;
#pragma unroll
                for (int i = 0; i < A_PW; ++i) glds16(srcA[i] + k0 + FT_BK, sA + nxt + (wave * A_PW + i) * 8 * FT_BK);
#pragma unroll
                for (int i = 0; i < B_PW; ++i) glds16(srcB[i] + k0 + FT_BK, sB + nxt + (wave * B_PW + i) * 8 * FT_BK);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PW; ++i) glds16(srcA[i] + k0, sA + (wave * A_PW + i) * 8 * FT_BK);
#pragma unroll
            for (int i = 0; i < B_PW; ++i) glds16(srcB[i] + k0, sB + (wave * B_PW + i) * 8 * FT_BK);
            __syncthreads();                                        // emits vmcnt(0): the DMA has landed
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[MT], fb[4];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cA + offA[i] + (((c0 + 4 * ks) ^ sw) << 3));
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const bf16x8*>(cB + offB[i] + (((c0 + 4 * ks) ^ sw) << 3));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
        }
        if (NS == 1) __syncthreads();                               // tile consumed before it is overwritten
    }

    gemm_epilogue<TC, MT>(g, acc, m0 + wm, n0 + wn, lane, g.splitk > 1 ? g.ws + (long long)blockIdx.y * g.M * g.N : nullptr, 0);
}

// =================================================================================================
// M <= 16: the projections of a cached decode step at the reference's own evaluation batch size (train.py:223-228 / evaluate.py:116-121 generate
// for bs = 8; HF LlamaAttention / LlamaMLP forward, modeling_llama.py:243-281,174-176).  At eight rows a 128-row tile is 6 % full and its LDS
// round trip is pure overhead (cdna_hip_programming.md §5 "GEMV / M <= 16": operands streamed once per block go straight to VGPRs), so this
// kernel streams the weight rows from HBM into registers and multiplies them on the matrix cores with the activation rows as the OTHER operand:
//   a block = 4 waves on the same 64 weight rows (4 accumulators of v_mfma_f32_16x16x32_bf16: D[n][m] = sum_k W[n][k] x[m][k]), the K range cut
//   over gridDim.y slices x 4 waves in steps of 128 (4 MFMA K-steps); per load instruction the 4 lanes of a row read 64 contiguous bytes of it
//   (the plain 16x16x32 operand map: a lane-contiguous variant, 64 B per LANE over four instructions, scattered every instruction over 64
//   sectors and measured 15-30 % slower);
//   the next step's 20 loads are in flight under the current 16 MFMAs; the 4 waves meet in LDS (fixed order) and the block writes either the
//   finished rows (bias / activation / residual as egomi_gemm defines them) or its fp32 K-slice slab (EGOMI_EPI_SLABS or a planned split,
//   summed by splitk_reduce_kernel / the decode step's slab consumers).
// =================================================================================================
#define GV_BN 64
template <typename TC>
__global__ __launch_bounds__(256) void gemv_m16_kernel(FastArgs g) {
    __shared__ float red[4][GV_BN][17];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * GV_BN;
    const int nl = lane & 15, grp = lane >> 4;
    const int T = g.K / 128;
    const int slot = blockIdx.y * 4 + wave, nslots = g.splitk * 4;
    const int t0 = (int)((long long)T * slot / nslots), t1 = (int)((long long)T * (slot + 1) / nslots);
    const int xm = nl < g.M ? nl : g.M - 1;
    const bf16_t* xp = g.A + (long long)xm * g.lda + 8 * grp;
    const bf16_t* wp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = n0 + 16 * j + nl;
        r = r < g.N ? r : g.N - 1;
        wp[j] = g.B + (long long)r * g.ldb + 8 * grp;
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[2][4], wf[2][4][4];
    auto load = [&](int buf, int t) {
        const long long k0 = (long long)t * 128;
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) xf[buf][p2] = *reinterpret_cast<const bf16x8*>(xp + k0 + 32 * p2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) wf[buf][j][p2] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[j] + k0 + 32 * p2));
    };
    if (t0 < t1) load(0, t0);
    for (int t = t0; t < t1; t += 2) {
        if (t + 1 < t1) load(1, t + 1);
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j][p2], xf[0][p2], acc[j], 0, 0, 0);
        if (t + 1 < t1) {
            if (t + 2 < t1) load(0, t + 2);
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][j][p2], xf[1][p2], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][16 * j + 4 * grp + r][nl] = acc[j][r];     // D: column m = lane & 15, row n = 4 (lane >> 4) + r
    __syncthreads();
    const int m = threadIdx.x >> 4, n4 = (threadIdx.x & 15) * 4;
    if (m >= g.M) return;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ((red[0][n4 + i][m] + red[1][n4 + i][m]) + red[2][n4 + i][m]) + red[3][n4 + i][m];
    const int n = n0 + n4;
    if (g.splitk > 1) {                                                  // this K slice's slab [M, N] fp32 (N % 4 == 0)
        if (n < g.N) *reinterpret_cast<f32x4*>(g.ws + ((long long)blockIdx.y * g.M + m) * g.N + n) = (f32x4){v[0], v[1], v[2], v[3]};
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (n + i >= g.N) continue;
        float xv = act_apply(v[i] * g.alpha + (g.bias ? bf2f(g.bias[n + i]) : 0.f), g.act);
        TC* cp = reinterpret_cast<TC*>(g.C) + (long long)m * g.ldc + n + i;
        if (g.residual) xv += Cvt<TC>::ld(reinterpret_cast<const TC*>(g.residual) + (long long)m * g.ldr + n + i);
        if (g.accumulate) xv += Cvt<TC>::ld(cp);
        Cvt<TC>::st(cp, xv);
    }
}

// =================================================================================================
// M <= 256 projections of the cached decode step (BASELINE.json configs[4], bs = 256: q|k|v, o_proj, gate|up, down_proj, lm_head against the
// KV-cached token; HF LlamaAttention / LlamaMLP forward, modeling_llama.py:243-281,174-176).  One block owns ALL 256 rows x 128 columns, 8 waves
// as 4 (M) x 2 (N) with the 64 x 64 accumulator layout of gemm_nt_bf16_kernel (same epilogue, same split-K slabs and consumers), and walks K in
// 64-deep steps through a THREE-stage LDS ring (48 KB per stage: 32 KB of activations, 16 KB of weights): the DMA of step t+2 is issued right
// after the single barrier of step t, two steps stay in flight across it (counted vmcnt, raw s_barrier).  Why this shape:
// the fill traffic of a BN-wide tile is W (1 + 256 / BN) — the 128 x 128 kernel re-reads the 2-MB activation once per 128 x 128 tile AND keeps only
// one K-step in flight behind a vmcnt(0) barrier (0.67 us per K-step with one block per CU); here a K-step is 32 MFMAs per wave (512 cycles) against
// 48 KB of fill (768 cycles at 64 B/clk), and the weights cross HBM exactly once.
// =================================================================================================
#define M256_STAGE ((256 + 128) * FT_BK)          // elements per stage
template <typename TC>
__global__ __launch_bounds__(512, 2)
void gemm_nt_bf16_m256_kernel(FastArgs g) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[3 * M256_STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int n0 = blockIdx.x * 128;

    // per-lane LDS-DMA source pointers: A 256 rows = 32 pieces of 8 rows (4 per wave), B 128 rows = 16 pieces (2 per wave)
    const bf16_t* srcA[4];
    const bf16_t* srcB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int ra = r < g.M ? r : g.M - 1;
        srcA[i] = g.A + (long long)ra * g.lda + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int rb = n0 + r; rb = rb < g.N ? rb : g.N - 1;
        srcB[i] = g.B + (long long)rb * g.ldb + chunk * 8;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) offA[i] = (wm + i * 16 + (lane & 15)) * FT_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) offB[i] = 256 * FT_BK + (wn + i * 16 + (lane & 15)) * FT_BK;
    const int sw = ((lane & 15) >> 1) & 7, c0 = lane >> 4;

    int nt = g.K / FT_BK, t_begin = 0;
    if (g.splitk > 1) {
        const int per = (nt + g.splitk - 1) / g.splitk;
        t_begin = blockIdx.y * per;
        nt = t_begin + per < nt ? t_begin + per : nt;
    }
    const int t_last = nt - 1;
    auto issue = [&](int t, int stage) {                              // past the slice's last step the DMA re-loads it into a stage nobody reads any more
        const int tt = t < t_last ? t : t_last;
        bf16_t* st = smem + stage * M256_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(srcA[i] + (long long)tt * FT_BK, st + (wave * 4 + i) * 8 * FT_BK);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(srcB[i] + (long long)tt * FT_BK, st + 256 * FT_BK + (wave * 2 + i) * 8 * FT_BK);
    };
    issue(t_begin, 0);
    issue(t_begin + 1, 1);
    int stage = 0;
    for (int t = t_begin; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");              // step t has landed (6 DMAs per step and wave; step t+1 stays in flight)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                 // ... for every wave, and every wave is done with step t-1's stage
        __builtin_amdgcn_sched_barrier(0);
        issue(t + 2, stage >= 1 ? stage - 1 : 2);                     // (stage + 2) % 3 = the stage step t-1 used
        const bf16_t* cS = smem + stage * M256_STAGE;
        stage = stage == 2 ? 0 : stage + 1;
        bf16x8 fa[2][4], fb[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(cS + offA[i] + (((c0 + 4 * ks) ^ sw) << 3));
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[ks][i] = *reinterpret_cast<const bf16x8*>(cS + offB[i] + (((c0 + 4 * ks) ^ sw) << 3));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[j][i], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the tail's redundant DMAs drain before the block's LDS is released
    gemm_epilogue<TC, 4>(g, acc, wm, n0 + wn, lane, g.splitk > 1 ? g.ws + (long long)blockIdx.y * g.M * g.N : nullptr, 0);
}

// =================================================================================================
// 256x256 tile, 8 waves (2 M x 4 N, 128x64 per wave), 8 phases per pair of K-tiles.
// Structure after cdna_hip_programming.md "The 256^2 8-phase template": all operand traffic is LDS-DMA
// that stays in flight across raw s_barriers (counted vmcnt, never 0 in the loop), the two wave groups
// (wr = 0 / 1, one wave of each per SIMD) run one barrier apart so one group's ds_read + DMA issue
// overlaps the other's MFMA cluster.
//   LDS  : 2 buffers x 4 half-tiles x [128 rows][64 k] bf16 = 128 KB.  Half-tiles are cut by CONSUMPTION
//          order, not by wave: A-h{0,1} = rows {0..63, 64..127} of both wave rows, B-h{0,1} = columns
//          {0..31, 32..63} of all four wave columns, so a half-tile is dead as soon as its phase is over.
//   tile t (buffer b):  ph1 reads B-h0, A-h0   MFMA (A0,B0)   DMA A-h1(t+1) -> b^1
//                       ph2 reads B-h1         MFMA (A0,B1)   DMA B-h0(t+2) -> b   (B-h0 reads retired by lgkmcnt(8) in ph1)
//                       ph3 reads A-h1         MFMA (A1,B1)   DMA A-h0(t+2) -> b
//                       ph4 (B0 kept in regs)  MFMA (A1,B0)   DMA B-h1(t+2) -> b ; vmcnt(6): tile t+1 has landed
//   rows beyond M / N clamp on load, mask on store; DMAs past the last K-tile re-load the last tile into
//   buffers nobody reads any more, which keeps the vmcnt arithmetic constant.
// =================================================================================================
#ifdef GEMM_STAMP
// Timing stamps of the per-tile 8-phase kernel (debug builds only, -DGEMM_STAMP; tools/debug/gemm_stamp.py): wave 0 of every
// block adds its s_memtime deltas per segment to g_gemm_stamp[].
__device__ unsigned long long g_gemm_stamp[16];
extern "C" int egomi_gemm_stamp_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamp), sizeof(g_gemm_stamp)); }
extern "C" int egomi_gemm_stamp_reset() { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamp), z, sizeof(z)); }
#define GSTAMP_DECL unsigned long long gs_prev = 0, gs_acc[6] = {0, 0, 0, 0, 0, 0};
#define GSTAMP_NOW(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define GSTAMP_START GSTAMP_NOW(gs_prev)
#define GSTAMP(i) { unsigned long long gs_n; GSTAMP_NOW(gs_n) gs_acc[i] += gs_n - gs_prev; gs_prev = gs_n; }
#define GSTAMP_FLUSH if (threadIdx.x == 0) { for (int i = 0; i < 6; ++i) atomicAdd(&g_gemm_stamp[i], gs_acc[i]); atomicAdd(&g_gemm_stamp[8], 1ull); }
#else
#define GSTAMP_DECL
#define GSTAMP_START
#define GSTAMP(i)
#define GSTAMP_FLUSH
#endif
#define P8_HT (128 * 64)
template <typename TC>
__global__ __launch_bounds__(512, 2)
void gemm_nt_bf16_8phase_kernel(FastArgs g) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 4 * P8_HT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    GSTAMP_DECL
    GSTAMP_START
    // whole tiles first (XCD-aware strips over the first full_tm tile rows), then the K-slices of the tail rows: the
    // dispatcher hands blocks out in index order, so the short blocks fill the ragged last round
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
    const int m0 = tm * 256, n0 = tn * 256;

    // per-lane DMA source offsets (elements; the host checks they fit 31 bits): [half][piece]
    unsigned offA[2][2], offB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (wave * 2 + i) * 8 + (lane >> 3);            // row of the half-tile image
            const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
            int ra = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63); ra = ra < g.M ? ra : g.M - 1;
            int rb = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);  rb = rb < g.N ? rb : g.N - 1;
            offA[h][i] = (unsigned)(ra * g.lda + chunk * 8);
            offB[h][i] = (unsigned)(rb * g.ldb + chunk * 8);
        }

    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (elements) inside a half-tile
    const int sw = ((lane & 15) >> 1) & 7, c0 = lane >> 4;
    int aBase[2], bBase[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aBase[ks] = (wr * 64 + (lane & 15)) * 64 + (((c0 + 4 * ks) ^ sw) << 3);
        bBase[ks] = (wc * 32 + (lane & 15)) * 64 + (((c0 + 4 * ks) ^ sw) << 3);
    }

    int nt = g.K / FT_BK, t_begin = 0;
    if (ksl > 1) {
        const int per = (nt + ksl - 1) / ksl;
        t_begin = kz * per;
        nt = t_begin + per < nt ? t_begin + per : nt;
    }
    const int t_last = nt - 1;

    bf16x8 fa[2][4], fb0[2][2], fb1[2][2];

#define P8_RD(off) (*reinterpret_cast<const bf16x8*>(smem + (off)))
#define P8_LDA(b, X) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) \
        fa[ks][ii] = P8_RD(((b) * 4 + (X)) * P8_HT + aBase[ks] + ii * 16 * 64);
#define P8_LDB(dst, b, X) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) \
        dst[ks][jj] = P8_RD(((b) * 4 + 2 + (X)) * P8_HT + bBase[ks] + jj * 16 * 64);
#define P8_PF(b, slot, base, off) _Pragma("unroll") for (int i = 0; i < 2; ++i) \
        glds16((base) + (off)[i], smem + ((b) * 4 + (slot)) * P8_HT + (wave * 2 + i) * 8 * 64);
#define P8_MMA(mh, fbv, nh) __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) \
            acc[(nh) * 2 + jj][(mh) * 4 + ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbv[ks][jj], fa[ks][ii], acc[(nh) * 2 + jj][(mh) * 4 + ii], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);
#define P8_BAR __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
#define P8_TILE(b, tt) { \
        const int t1 = (tt) + 1 < t_last ? (tt) + 1 : t_last, t2 = (tt) + 2 < t_last ? (tt) + 2 : t_last; \
        const bf16_t* pA1 = g.A + (long long)t1 * FT_BK; \
        const bf16_t* pA2 = g.A + (long long)t2 * FT_BK; \
        const bf16_t* pB2 = g.B + (long long)t2 * FT_BK; \
        /* ph1 */ P8_LDB(fb0, b, 0) __builtin_amdgcn_sched_barrier(0); P8_LDA(b, 0) P8_PF((b) ^ 1, 1, pA1, offA[1]) \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); P8_BAR P8_MMA(0, fb0, 0) P8_BAR \
        /* ph2 */ P8_LDB(fb1, b, 1) P8_PF(b, 2, pB2, offB[0]) P8_BAR P8_MMA(0, fb1, 1) P8_BAR \
        /* ph3 */ P8_LDA(b, 1) P8_PF(b, 0, pA2, offA[0]) P8_BAR P8_MMA(1, fb1, 1) P8_BAR \
        /* ph4 */ P8_PF(b, 3, pB2, offB[1]) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); P8_BAR P8_MMA(1, fb0, 0) P8_BAR \
    }

    // ---- prologue: tile t_begin complete, three half-tiles of the next one in flight
    {
        const int t1 = t_begin + 1 < t_last ? t_begin + 1 : t_last;
        const bf16_t* pA0 = g.A + (long long)t_begin * FT_BK;
        const bf16_t* pB0 = g.B + (long long)t_begin * FT_BK;
        const bf16_t* pA1 = g.A + (long long)t1 * FT_BK;
        const bf16_t* pB1 = g.B + (long long)t1 * FT_BK;
        P8_PF(0, 2, pB0, offB[0]) P8_PF(0, 0, pA0, offA[0]) P8_PF(0, 3, pB0, offB[1]) P8_PF(0, 1, pA0, offA[1])
        P8_PF(1, 2, pB1, offB[0]) P8_PF(1, 0, pA1, offA[0]) P8_PF(1, 3, pB1, offB[1])
        GSTAMP(0)                                                 // set-up + DMA issue
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        P8_BAR
        GSTAMP(1)                                                 // first tile landed
    }
    if (wr == 1) { P8_BAR }                                       // second wave group runs one barrier behind
    int t = t_begin;
    for (; t + 1 < nt; t += 2) {
        P8_TILE(0, t)
        P8_TILE(1, t + 1)
    }
    if (t < nt) P8_TILE(0, t)
    if (wr == 0) { P8_BAR }
    GSTAMP(2)                                                     // main loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the tail's redundant DMAs drain before the block's LDS is released
    __builtin_amdgcn_s_barrier();                                 // ... for every wave (unconditional: the epilogue below re-uses the stages per wave)
    GSTAMP(3)                                                     // drain
    const int row0 = g.full_tm * 256;
    const int mb = m0 + wr * 128, nb = n0 + wc * 64;
    if (ksl > 1 && g.tickets) {
        // ---- K-sliced tail tile, combined in the launch (the protocol of the persistent kernel below; cdna_hip_programming.md §5
        // "Projection GEMM at M = 256" item 2): every slice block drops its partial tile as a register-major fp32 slab (1-KiB wave
        // stores) and draws a ticket; the block that draws the LAST ticket sums the slabs in slice order and runs the epilogue.  Nobody waits for anybody.
        // Slab data moves with sc1 buffer stores / loads (write-through past the XCD's L2, reads that bypass it): no agent-scope
        // release (= whole-L2 write-back) or acquire (= invalidate) in the middle of the launch; the ticket is a relaxed atomic
        // issued after the block's stores have completed (vmcnt(0) + barrier).
        const int tile = (blockIdx.x - g.full_tiles) / ksl;
        char* slab = reinterpret_cast<char*>(g.ws) + (long long)tile * ksl * 262144;
        {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(slab + ((long long)kz * 8 + wave) * 32768, 0, 32768, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), r, ((j * 8 + i) * 64 + lane) * 16, 0, 16);
        }
        int* flag = reinterpret_cast<int*>(smem);                       // the operand stages are idle: every DMA has drained (above)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(g.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int tk = *const_cast<volatile int*>(flag);
        if (tk != ksl - 1) return;                                      // block-uniform: somebody else finishes this tile
        if (threadIdx.x == 0) __hip_atomic_store(g.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // leave the word as we found it
        // two slices: mine + the other (commutative).  More: every slab in slice order, mine read back, so that the sum does not
        // depend on the arrival order
        const bool all = ksl > 2;
        if (all) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int z = 0; z < ksl; ++z) {
            if (z == kz && !all) continue;
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(slab + ((long long)z * 8 + wave) * 32768, 0, 32768, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[j][i] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ((j * 8 + i) * 64 + lane) * 16, 0, 16));
        }
        gemm_epilogue<TC, 8>(g, acc, mb, nb, lane);
        return;
    }
    if (sizeof(TC) == 2 && ksl == 1 && !g.bias && !g.residual && !g.accumulate && g.act == 0 && g.alpha == 1.0f &&
        nb + 64 <= g.N && (g.ldc & 7) == 0 && ((uintptr_t)g.C & 15) == 0) {              // wave-uniform
        // plain bf16 store: the direct form writes 8-B pieces of 16 different rows per instruction (16 line transactions
        // each; ~7 us per 256x256 tile with nothing to overlap it).  Each wave instead drops 64 rows x 64 columns at a time
        // into a private LDS region (144-B pitch) and stores them back as 128-B row segments, 16 B per lane.
        char* wb = reinterpret_cast<char*>(smem) + wave * (64 * 144);
        bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
        if (g.epi == 3) {
            // SwiGLU backward in the epilogue of the down_proj data gradient: the wave's 64 columns of d(act) are two interleaved-32 groups,
            // i.e. 128 consecutive columns of gate|up and of d(gate|up).  d(act) itself never reaches memory.  Four sub-passes of 32 rows:
            // the gate|up rows come in as 256-B row segments (16 B per lane) into a wave-private LDS strip (272-B pitch), every lane combines
            // its 4 units x 2 row blocks in place (rounding sequence of swiglu_bwd_kernel on stored bf16 values: d(act) rounded to bf16
            // first), and the strip leaves as 256-B row segments.  The next sub-pass's rows are in flight meanwhile.
            char* ws3 = reinterpret_cast<char*>(smem) + wave * (32 * 272);
            const bf16_t* GU = reinterpret_cast<const bf16_t*>(g.C2);
            const int lr = lane >> 4, ch = lane & 15;
            u32x4 in[8];
#define P8_GU_FETCH(sp_) _Pragma("unroll") for (int it = 0; it < 8; ++it) { \
                    int m_ = mb + (sp_) * 32 + it * 4 + lr; \
                    m_ = m_ < g.M ? m_ : g.M - 1; \
                    in[it] = *reinterpret_cast<const u32x4*>(GU + (long long)m_ * g.ldc2 + 2 * nb + ch * 8); }
            P8_GU_FETCH(0)
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
#pragma unroll
                for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4*>(ws3 + (it * 4 + lr) * 272 + ch * 16) = in[it];
                if (sp < 3) { P8_GU_FETCH(sp + 1) }
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = acc[j][2 * sp + i2];
                        char* pg = ws3 + (i2 * 16 + (lane & 15)) * 272 + ((j >> 1) * 64 + (j & 1) * 16 + lr * 4) * 2;
                        const u32x2 gg = *reinterpret_cast<const u32x2*>(pg), uu = *reinterpret_cast<const u32x2*>(pg + 64);
                        float og[4], ou[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float gv = __uint_as_float((r & 1) ? (gg[r >> 1] & 0xFFFF0000u) : (gg[r >> 1] << 16));
                            const float uv = __uint_as_float((r & 1) ? (uu[r >> 1] & 0xFFFF0000u) : (uu[r >> 1] << 16));
                            swiglu_bwd_elem(bf2f(f2bf(v[r])), gv, uv, og[r], ou[r]);
                        }
                        u32x2 o;
                        o[0] = (uint32_t)f2bf(og[0]) | ((uint32_t)f2bf(og[1]) << 16);
                        o[1] = (uint32_t)f2bf(og[2]) | ((uint32_t)f2bf(og[3]) << 16);
                        *reinterpret_cast<u32x2*>(pg) = o;
                        o[0] = (uint32_t)f2bf(ou[0]) | ((uint32_t)f2bf(ou[1]) << 16);
                        o[1] = (uint32_t)f2bf(ou[2]) | ((uint32_t)f2bf(ou[3]) << 16);
                        *reinterpret_cast<u32x2*>(pg + 64) = o;
                    }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = mb + sp * 32 + it * 4 + lr;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ws3 + (it * 4 + lr) * 272 + ch * 16);
                    if (m < g.M) *reinterpret_cast<u32x4*>(C + (long long)m * g.ldc + 2 * nb + ch * 8) = v;
                }
            }
#undef P8_GU_FETCH
            GSTAMP(4)
            GSTAMP(5)
            GSTAMP_FLUSH
            return;
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[j][4 * pass + ii];
                    u32x2 o;
                    o[0] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                    o[1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                    *reinterpret_cast<u32x2*>(wb + (ii * 16 + (lane & 15)) * 144 + (j * 16 + (lane >> 4) * 4) * 2) = o;
                }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 8 + (lane >> 3), ch = lane & 7;
                const u32x4 v = *reinterpret_cast<const u32x4*>(wb + r * 144 + ch * 16);
                const int m = mb + pass * 64 + r;
                if (m < g.M) *reinterpret_cast<u32x4*>(C + (long long)m * g.ldc + nb + ch * 8) = v;
            }
            if (g.epi == 1) {
                // SwiGLU in the epilogue (HF LlamaMLP, modeling_llama.py:174-176): the wave's 64 columns are one interleaved-32 group,
                // gate in accumulators j = 0,1 and up of the SAME 32 hidden units in j = 2,3, so silu(gate)*up is lane-local.  Rounding
                // sequence of swiglu_fwd_kernel on the stored bf16 values (bit-identical to the unfused path): g, u rounded to bf16,
                // a = bf16(g*sigmoid(g)), out = bf16(a*u).  Rows leave as 64-B segments through the same strip.
                bf16_t* C2 = reinterpret_cast<bf16_t*>(g.C2);
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 vg = acc[j][4 * pass + ii], vu = acc[j + 2][4 * pass + ii];
                        float o4[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float gg = bf2f(f2bf(vg[r])), uu = bf2f(f2bf(vu[r]));
                            const float a = bf2f(f2bf(gg * (1.0f / (1.0f + __expf(-gg)))));
                            o4[r] = a * uu;
                        }
                        u32x2 o;
                        o[0] = (uint32_t)f2bf(o4[0]) | ((uint32_t)f2bf(o4[1]) << 16);
                        o[1] = (uint32_t)f2bf(o4[2]) | ((uint32_t)f2bf(o4[3]) << 16);
                        *reinterpret_cast<u32x2*>(wb + (ii * 16 + (lane & 15)) * 144 + (j * 16 + (lane >> 4) * 4) * 2) = o;
                    }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = it * 16 + (lane >> 2), ch = lane & 3;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(wb + r * 144 + ch * 16);
                    const int m = mb + pass * 64 + r;
                    if (m < g.M) *reinterpret_cast<u32x4*>(C2 + (long long)m * g.ldc2 + (nb >> 1) + ch * 8) = v;
                }
            }
        }
        GSTAMP(4)                                                 // epilogue issued
#ifdef GEMM_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        GSTAMP(5)                                                 // stores acknowledged
        GSTAMP_FLUSH
        return;
    }
    gemm_epilogue<TC, 8>(g, acc, mb, nb, lane, ksl > 1 ? g.ws + (long long)kz * (g.M - row0) * g.N : nullptr, row0);
#undef P8_TILE
#undef P8_BAR
#undef P8_MMA
#undef P8_PF
#undef P8_LDB
#undef P8_LDA
#undef P8_RD
}


// =================================================================================================
// 352x256 tile: the 8-phase schedule with 11 instead of 8 row blocks per wave (2 M x 4 N waves, 176x64 each) — the form for
// products whose 256x256 tiles leave a ragged last round.  At M = 5536, N = 4096 (o_proj, down_proj and three of the four
// data gradients of a LLaMA layer: 54 % of the step's GEMM flops) 22 x 16 = 352 tiles of 256x256 are 1.375 rounds on 256 CUs,
// which the K-sliced tail runs in 1.56-1.64 tile-times; 16 x 16 = 256 tiles of 352x256 are exactly ONE round of 1.375
// tile-times: no slabs, no combine pass, one prologue / epilogue per CU.  (tall_form() below picks per shape.)
//   LDS  : 2 buffers x (A [352 rows][64 k] + B [256 rows][64 k]) bf16 = 152 KB, rows in tile order, same XOR swizzle.
//          Sub-tiles by CONSUMPTION order: A-c0 / c1 / c2 = row blocks 0-3 / 4-7 / 8-10 of both wave rows, B-h0 / h1 =
//          columns 0-31 / 32-63 of all four wave columns.
//   tile t (buffer b), six phases of [ds_read + DMA issue | barrier | MFMA cluster | barrier], wave groups one barrier apart:
//          ph1 reads A-c0                 MFMA (c0,h0) 16
//          ph2 reads B-h1                 MFMA (c0,h1) 16   DMA B-h0(t+2) -> b
//          ph3 reads A-c1                 MFMA (c1,h1) 16   DMA A-c0(t+2) -> b
//          ph4 (h0 kept in regs)          MFMA (c1,h0) 16   DMA B-h1(t+2) -> b ; vmcnt(6): A-c2(t+1) and everything before it has landed
//          ph5 reads A-c2                 MFMA (c2,h0) 12   DMA A-c1(t+2) -> b
//          ph6 reads B-h0 of tile t+1     MFMA (c2,h1) 12   DMA A-c2(t+1) -> b^1 (h1 kept in regs; h0's registers are free after ph5)
//          A sub-tile is refilled no earlier than two phases after its last read, because the other wave group runs one barrier
//          behind.  No read stage carries more than 8 ds_read_b128 + 2 DMA pieces per wave: a group's read stage runs under the other
//          group's MFMA cluster, and 4 waves x (12 reads + 2 pieces) did not fit under 16 MFMAs (first version: ph1 read B-h0 AND
//          A-c0; tools/debug/tall_ablate.py).
//   DMA  : inline asm, scalar base + one 32-bit per-lane offset per operand: every 8-row group a wave fetches has the wave's
//          own parity, so the swizzled source offset inside a group is the same VGPR for all of them; the group's row offset
//          is scalar.  Needs M % 8 == 0 and N % 8 == 0 (a ragged tile clamps whole 8-row groups), M, N >= 8.
//   88 MFMAs per K-tile and wave on 30 ds_read_b128 (0.34 per MFMA; the 256x256 form: 0.375), 10 (9) DMA instructions.
// =================================================================================================
#define TL_NB 11
#define TL_BM (32 * TL_NB)
#define TL_AB (TL_BM * 128)                       // bytes of the A image of one buffer
#define TL_BUFB (TL_AB + 256 * 128)               // bytes per buffer
__device__ __forceinline__ void tl_dma(const char* gbase, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(gbase), "s"(lds_base) : "memory");
}
typedef __attribute__((ext_vector_type(4))) __bf16 tl_bf16x4;
typedef __attribute__((address_space(3))) tl_bf16x4 tl_lds_bf16x4;
// TBK: B is k-major ([K][N], row stride ldb: a data gradient dX = dY . W against the weight as it lies in memory, gemm_tn.hip's TB operand): its halves are the
// [64 k][128 columns] images of gemm_bf16_8phase_t_kernel (256-B rows, T10 image (b) swizzle, 4-row DMA pieces, fragments by two ds_read_b64_tr_b16), wave column wc
// owns columns {128 X + 32 wc + 0..31 : X = 0, 1} of the tile; everything else — A side, phases, DMA counts per phase, vmcnt arithmetic — is the same kernel.
// Needs N % 256 == 0 (no column clamp) and K % 64 == 0; plain bf16 store only.
template <typename TC, bool TBK>
__global__ __launch_bounds__(512, 2)
void gemm_nt_bf16_tall_kernel(FastArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * TL_BUFB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    int tm, tn;
    {                                                                 // XCD-aware strips, 8 tile rows deep (as the 256x256 kernel)
        const int nwg = gridDim.x;
        int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int gd = g.full_tm;                                     // tile rows per group (launch_tall)
        const int per_group = gd * g.tiles_n;
        const int grp = bid / per_group, first_tm = grp * gd;
        const int gsz = (g.tiles_m - first_tm) < gd ? (g.tiles_m - first_tm) : gd;
        const int in_g = bid - grp * per_group;
        tm = first_tm + in_g % gsz; tn = in_g / gsz;
    }
    const int m0 = tm * TL_BM, n0 = tn * 256;

    // ---- the 8-row groups this wave fetches (index = tile row / 8): A-c0 {w, 22+w}, A-c1 {8+w, 30+w}, A-c2 {16..21, 38, 39 by wave} + {40+w, waves 0-3},
    //      B-h {(w>>2)*8 + 4h + (w&3), +16}: all of parity w & 1
    const int ga2 = wave < 6 ? 16 + wave : 32 + wave;
    const int gA[3][2] = {{wave, 22 + wave}, {8 + wave, 30 + wave}, {ga2, 40 + (wave & 3)}};
    const int gb = (wave >> 2) * 8 + (wave & 3);
    uint32_t sA[3][2], sB[2][2], lA[3][2], lB[2][2];
    const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_void*)smem);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = m0 + gA[c][i] * 8; r = r < g.M - 8 ? r : g.M - 8;
            sA[c][i] = __builtin_amdgcn_readfirstlane((uint32_t)(r * (int)g.lda) * 2u);
            lA[c][i] = __builtin_amdgcn_readfirstlane(lds0 + gA[c][i] * 1024);
        }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (TBK) {                                                  // piece wave*2+i = k-rows 4*(wave*2+i) .. +3 of half h's image
                sB[h][i] = __builtin_amdgcn_readfirstlane((uint32_t)(n0 + 128 * h) * 2u);
                lB[h][i] = __builtin_amdgcn_readfirstlane(lds0 + TL_AB + h * 16384 + (wave * 2 + i) * 1024);
            } else {
                const int gi = gb + 4 * h + 16 * i;
                int r = n0 + gi * 8; r = r < g.N - 8 ? r : g.N - 8;
                sB[h][i] = __builtin_amdgcn_readfirstlane((uint32_t)(r * (int)g.ldb) * 2u);
                lB[h][i] = __builtin_amdgcn_readfirstlane(lds0 + TL_AB + gi * 1024);
            }
        }
    const uint32_t dchunk = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    const uint32_t vA = (uint32_t)((lane >> 3) * (int)g.lda + (int)dchunk * 8) * 2u;
    uint32_t vB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (TBK) {
            const int r = (wave * 2 + i) * 4 + (lane >> 4);             // k-row of the image
            const int ch = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
            vB[i] = (uint32_t)(r * (int)g.ldb + 8 * ch) * 2u;
        } else vB[i] = (uint32_t)((lane >> 3) * (int)g.ldb + (int)dchunk * 8) * 2u;
    }

    f32x4 acc[4][TL_NB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TL_NB; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) inside a buffer
    const int sw = ((lane & 15) >> 1) & 7, c0 = lane >> 4;
    int aRd[2], bRd[TBK ? 4 : 2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aRd[ks] = (wr * (16 * TL_NB) + (lane & 15)) * 128 + (((c0 + 4 * ks) ^ sw) << 4);
        if (!TBK) bRd[ks] = TL_AB + (wc * 64 + (lane & 15)) * 128 + (((c0 + 4 * ks) ^ sw) << 4);
    }
    if (TBK) {                                                          // bRd[jj * 2 + e]: the two transposed 8-byte reads of column block jj (gemm_tn.hip TnSide<true>)
        const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int row = 8 * gq + q + 4 * e;                     // + 32 ks: an immediate
                const int ch = (wc * 32 + 16 * jj) / 8 + (pp >> 1);
                bRd[jj * 2 + e] = TL_AB + 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (pp & 1);
            }
    }
    const int nt = g.K / FT_BK, t_last = nt - 1;
    const char* Ab = reinterpret_cast<const char*>(g.A);
    const char* Bb = reinterpret_cast<const char*>(g.B);
    const long long strideB = TBK ? 128ll * g.ldb : 128ll;             // bytes per K-tile of B

    bf16x8 fa[2][4], fb0[2][2], fb1[2][2];
    // (bo = byte offset of the tile's buffer, bn = of the other one: scalars; the per-lane read offsets aRd / bRd are advanced from buffer to buffer,
    //  because buffer 1's fragments lie beyond the 64-KB reach of a ds_read immediate and a second set of base registers does not fit 256 VGPRs)
#define TL_RD(off) (*reinterpret_cast<const bf16x8*>(smem + (off)))
#define TL_LDA(c, n) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int ii = 0; ii < (n); ++ii) \
        fa[ks][ii] = TL_RD(aRd[ks] + ((c) * 64 + ii * 16) * 128);
#define TL_LDB(dst, X) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) { \
        if (TBK) { \
            const tl_bf16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tl_lds_bf16x4*)(smem + bRd[(TBK ? jj * 2 : 0)] + (X) * 16384 + ks * 8192)); \
            const tl_bf16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tl_lds_bf16x4*)(smem + bRd[(TBK ? jj * 2 + 1 : 0)] + (X) * 16384 + ks * 8192)); \
            bf16x8 r_; r_[0] = lo_[0]; r_[1] = lo_[1]; r_[2] = lo_[2]; r_[3] = lo_[3]; r_[4] = hi_[0]; r_[5] = hi_[1]; r_[6] = hi_[2]; r_[7] = hi_[3]; \
            dst[ks][jj] = r_; \
        } else dst[ks][jj] = TL_RD(bRd[ks] + ((X) * 32 + jj * 16) * 128); }
#define TL_PFA(bo_, c, base) { tl_dma((base) + sA[c][0], vA, lA[c][0] + (bo_)); \
        if ((c) < 2 || wave < 4) tl_dma((base) + sA[c][1], vA, lA[c][1] + (bo_)); }
#define TL_PFB(bo_, h, base) { tl_dma((base) + sB[h][0], vB[0], lB[h][0] + (bo_)); tl_dma((base) + sB[h][1], vB[1], lB[h][1] + (bo_)); }
#define TL_MMA(c, n, fbv, X) __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) _Pragma("unroll") for (int ii = 0; ii < (n); ++ii) \
            acc[(X) * 2 + jj][(c) * 4 + ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbv[ks][jj], fa[ks][ii], acc[(X) * 2 + jj][(c) * 4 + ii], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);
#define TL_BAR __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);

    // (TL_ABL: timing-only ablation builds of tools/debug/tall_ablate.py — 1 no fragment reads, 2 no DMA, 4 one barrier per phase, 8 no MFMA, 16 every DMA reads K-tile 0; results are garbage)
#ifndef TL_ABL
#define TL_ABL 0
#endif
#define TL_LDA_(c, n) if (!(TL_ABL & 1)) { TL_LDA(c, n) }
#define TL_LDB_(dst, X) if (!(TL_ABL & 1)) { TL_LDB(dst, X) }
#define TL_PFA_(bo_, c, base) if (!(TL_ABL & 2)) { TL_PFA(bo_, c, base) }
#define TL_PFB_(bo_, h, base) if (!(TL_ABL & 2)) { TL_PFB(bo_, h, base) }
#define TL_MMA_(c, n, fbv, X) if (!(TL_ABL & 8)) { TL_MMA(c, n, fbv, X) }
#define TL_BAR2 if (!(TL_ABL & 4)) { TL_BAR }

    // ---- prologue: tile 0 complete, all but A-c2 of tile 1 in flight; B-h0 of tile 0 in registers
    {
        const int t1 = 1 < t_last ? 1 : t_last;
        const char* pA1 = Ab + (long long)t1 * 128;
        const char* pB1 = Bb + (long long)t1 * strideB;
        TL_PFB(0, 0, Bb) TL_PFA(0, 0, Ab) TL_PFB(0, 1, Bb) TL_PFA(0, 1, Ab) TL_PFA(0, 2, Ab)
        TL_PFB(TL_BUFB, 0, pB1) TL_PFA(TL_BUFB, 0, pA1) TL_PFB(TL_BUFB, 1, pB1) TL_PFA(TL_BUFB, 1, pA1)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        TL_BAR
        TL_LDB(fb0, 0)
        if (TL_ABL & 1) { TL_LDB(fb1, 1) TL_LDA(0, 4) }
    }
    if (wr == 1) { TL_BAR }                                       // second wave group runs one barrier behind
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
        const uint32_t bo = (t & 1) ? TL_BUFB : 0, bn = TL_BUFB - bo;
        const int t1 = (TL_ABL & 16) ? 0 : (t + 1 < t_last ? t + 1 : t_last), t2 = (TL_ABL & 16) ? 0 : (t + 2 < t_last ? t + 2 : t_last);     // (16: every DMA re-reads K-tile 0 — L2-hot)
        const char* pA1 = Ab + (long long)t1 * 128;
        const char* pA2 = Ab + (long long)t2 * 128;
        const char* pB2 = Bb + (long long)t2 * strideB;
        /* ph1 */ TL_LDA_(0, 4) TL_BAR TL_MMA_(0, 4, fb0, 0) TL_BAR2
        /* ph2 */ TL_LDB_(fb1, 1) TL_PFB_(bo, 0, pB2) TL_BAR TL_MMA_(0, 4, fb1, 1) TL_BAR2
        /* ph3 */ TL_LDA_(1, 4) TL_PFA_(bo, 0, pA2) TL_BAR TL_MMA_(1, 4, fb1, 1) TL_BAR2
        /* ph4 */ TL_PFB_(bo, 1, pB2) if (!(TL_ABL & 2)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        TL_BAR TL_MMA_(1, 4, fb0, 0) TL_BAR2
        /* ph5 */ TL_LDA_(2, 3) TL_PFA_(bo, 1, pA2) TL_BAR TL_MMA_(2, 3, fb0, 0) TL_BAR2
        /* ph6 */
        {
            const int dl = (t & 1) ? -TL_BUFB : TL_BUFB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) aRd[ks] += dl;
#pragma unroll
            for (int q = 0; q < (TBK ? 4 : 2); ++q) bRd[q] += dl;
        }
        TL_LDB_(fb0, 0) TL_PFA_(bn, 2, pA1)
        TL_BAR TL_MMA_(2, 3, fb1, 1) TL_BAR2
    }
    if (wr == 0) { TL_BAR }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the tail's redundant DMAs drain before the stages are re-used below
    __builtin_amdgcn_s_barrier();
#undef TL_BAR
#undef TL_MMA
#undef TL_PFB
#undef TL_PFA
#undef TL_LDB
#undef TL_LDA
#undef TL_RD
    const int mb = m0 + wr * (16 * TL_NB), nb = n0 + wc * 64;
    if (TBK && (sizeof(TC) != 2 || g.bias || g.residual || g.accumulate || g.act != 0 || g.alpha != 1.0f || g.epi != 0 || (g.ldc & 7) || ((uintptr_t)g.C & 15))) return;   // (the launcher admits the plain bf16 store only)
    if (sizeof(TC) == 2 && !g.bias && !g.accumulate && g.act == 0 && g.alpha == 1.0f && nb + 64 <= g.N && (g.ldc & 7) == 0 && ((uintptr_t)g.C & 15) == 0 &&
        (!g.residual || ((g.ldr & 7) == 0 && ((uintptr_t)g.residual & 15) == 0))) {      // wave-uniform
        bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
        if (g.residual) {
            // bf16 tile + residual (o_proj, down_proj): the direct form stores 8-B pieces of 16 rows per instruction; here 64 rows x 64 columns of fp32
            // go through a wave-private strip (272-B pitch) and leave as 128-B row segments, the residual rows arriving the same way.
            // Same arithmetic as gemm_epilogue: fp32 sum, one rounding.
            char* wb = smem + wave * (64 * 272);
            const bf16_t* R = reinterpret_cast<const bf16_t*>(g.residual);
            const int lr = lane >> 3, ch = lane & 7;
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {
                const int nblk = (TL_NB - 4 * pass) < 4 ? (TL_NB - 4 * pass) : 4;
                u32x4 rr[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    if (it * 8 >= nblk * 16) continue;
                    int m = mb + pass * 64 + it * 8 + lr; m = m < g.M ? m : g.M - 1;
                    rr[it] = *reinterpret_cast<const u32x4*>(R + (long long)m * g.ldr + nb + ch * 8);
                }
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    if (ii >= nblk) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<f32x4*>(wb + (ii * 16 + (lane & 15)) * 272 + (j * 16 + (lane >> 4) * 4) * 4) = acc[j][4 * pass + ii];
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    if (it * 8 >= nblk * 16) continue;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(wb + (it * 8 + lr) * 272 + ch * 32);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(wb + (it * 8 + lr) * 272 + ch * 32 + 16);
                    u32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = (q < 2 ? x0[2 * q] : x1[2 * q - 4]) + __uint_as_float(rr[it][q] << 16);
                        const float hi = (q < 2 ? x0[2 * q + 1] : x1[2 * q - 3]) + __uint_as_float(rr[it][q] & 0xFFFF0000u);
                        o[q] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
                    }
                    const int m = mb + pass * 64 + it * 8 + lr;
                    if (m < g.M) *reinterpret_cast<u32x4*>(C + (long long)m * g.ldc + nb + ch * 8) = o;
                }
            }
            return;
        }
        if (g.epi == 3) {
            // SwiGLU backward in the epilogue of the down_proj data gradient (the 256x256 form's epilogue, same arithmetic and rounding sequence):
            // sub-passes of 32 rows = 2 row blocks (the sixth has one), gate|up rows in as 256-B segments into a wave-private strip (272-B pitch),
            // combined in place, out as 256-B segments.  d(act) never reaches memory.
            char* ws3 = smem + wave * (32 * 272);
            const bf16_t* GU = reinterpret_cast<const bf16_t*>(g.C2);
            const int lr = lane >> 4, ch = lane & 15;
            u32x4 in[8];
#define TL_GU_FETCH(sp_) _Pragma("unroll") for (int it = 0; it < 8; ++it) { \
                    if ((sp_) * 32 + it * 4 >= 16 * TL_NB) continue; \
                    int m_ = mb + (sp_) * 32 + it * 4 + lr; \
                    m_ = m_ < g.M ? m_ : g.M - 1; \
                    in[it] = *reinterpret_cast<const u32x4*>(GU + (long long)m_ * g.ldc2 + 2 * nb + ch * 8); }
            TL_GU_FETCH(0)
#pragma unroll
            for (int sp = 0; sp < (TL_NB + 1) / 2; ++sp) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    if (sp * 32 + it * 4 >= 16 * TL_NB) continue;
                    *reinterpret_cast<u32x4*>(ws3 + (it * 4 + lr) * 272 + ch * 16) = in[it];
                }
                if (sp + 1 < (TL_NB + 1) / 2) { TL_GU_FETCH(sp + 1) }
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    if (2 * sp + i2 >= TL_NB) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = acc[j][2 * sp + i2];
                        char* pg = ws3 + (i2 * 16 + (lane & 15)) * 272 + ((j >> 1) * 64 + (j & 1) * 16 + lr * 4) * 2;
                        const u32x2 gg = *reinterpret_cast<const u32x2*>(pg), uu = *reinterpret_cast<const u32x2*>(pg + 64);
                        float og[4], ou[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float gv = __uint_as_float((r & 1) ? (gg[r >> 1] & 0xFFFF0000u) : (gg[r >> 1] << 16));
                            const float uv = __uint_as_float((r & 1) ? (uu[r >> 1] & 0xFFFF0000u) : (uu[r >> 1] << 16));
                            swiglu_bwd_elem(bf2f(f2bf(v[r])), gv, uv, og[r], ou[r]);
                        }
                        u32x2 o;
                        o[0] = (uint32_t)f2bf(og[0]) | ((uint32_t)f2bf(og[1]) << 16);
                        o[1] = (uint32_t)f2bf(og[2]) | ((uint32_t)f2bf(og[3]) << 16);
                        *reinterpret_cast<u32x2*>(pg) = o;
                        o[0] = (uint32_t)f2bf(ou[0]) | ((uint32_t)f2bf(ou[1]) << 16);
                        o[1] = (uint32_t)f2bf(ou[2]) | ((uint32_t)f2bf(ou[3]) << 16);
                        *reinterpret_cast<u32x2*>(pg + 64) = o;
                    }
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    if (sp * 32 + it * 4 >= 16 * TL_NB) continue;
                    const int m = mb + sp * 32 + it * 4 + lr;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ws3 + (it * 4 + lr) * 272 + ch * 16);
                    if (m < g.M) *reinterpret_cast<u32x4*>(C + (long long)m * g.ldc + 2 * nb + ch * 8) = v;
                }
            }
#undef TL_GU_FETCH
            return;
        }
        // plain bf16 store through a wave-private LDS strip (64 rows x 144-B pitch) as 128-B row segments, as the 256x256 kernel
        char* wb = smem + wave * (64 * 144);
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            const int nblk = (TL_NB - 4 * pass) < 4 ? (TL_NB - 4 * pass) : 4;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                if (ii >= nblk) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[j][4 * pass + ii];
                    u32x2 o;
                    o[0] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                    o[1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                    *reinterpret_cast<u32x2*>(wb + (ii * 16 + (lane & 15)) * 144 + (j * 16 + (lane >> 4) * 4) * 2) = o;
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 8 + (lane >> 3), ch = lane & 7;
                if (it * 8 >= nblk * 16) continue;
                const u32x4 v = *reinterpret_cast<const u32x4*>(wb + r * 144 + ch * 16);
                const int m = mb + pass * 64 + r;
                // (k-major B: the wave's 64 columns are two groups of 32, 128 columns apart — strip columns 0-31 / 32-63)
                const int ncol = TBK ? n0 + 128 * (ch >> 2) + 32 * wc + (ch & 3) * 8 : nb + ch * 8;
                if (m < g.M) *reinterpret_cast<u32x4*>(C + (long long)m * g.ldc + ncol) = v;
            }
            if (g.epi == 1) {
                // SwiGLU in the epilogue (the 256x256 form's, same rounding sequence): the wave's 64 columns are one interleaved-32 group, gate in
                // accumulators j = 0,1 and up of the same 32 hidden units in j = 2,3; act rows leave as 64-B segments through the same strip
                bf16_t* C2 = reinterpret_cast<bf16_t*>(g.C2);
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    if (ii >= nblk) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 vg = acc[j][4 * pass + ii], vu = acc[j + 2][4 * pass + ii];
                        float o4[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float gg = bf2f(f2bf(vg[r])), uu = bf2f(f2bf(vu[r]));
                            const float a = bf2f(f2bf(gg * (1.0f / (1.0f + __expf(-gg)))));
                            o4[r] = a * uu;
                        }
                        u32x2 o;
                        o[0] = (uint32_t)f2bf(o4[0]) | ((uint32_t)f2bf(o4[1]) << 16);
                        o[1] = (uint32_t)f2bf(o4[2]) | ((uint32_t)f2bf(o4[3]) << 16);
                        *reinterpret_cast<u32x2*>(wb + (ii * 16 + (lane & 15)) * 144 + (j * 16 + (lane >> 4) * 4) * 2) = o;
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (it * 16 >= nblk * 16) continue;
                    const int r = it * 16 + (lane >> 2), ch = lane & 3;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(wb + r * 144 + ch * 16);
                    const int m = mb + pass * 64 + r;
                    if (m < g.M) *reinterpret_cast<u32x4*>(C2 + (long long)m * g.ldc2 + (nb >> 1) + ch * 8) = v;
                }
            }
        }
        return;
    }
    if (!TBK) gemm_epilogue<TC, TL_NB>(g, acc, mb, nb, lane);
}


// =================================================================================================
// Persistent form of the 8-phase kernel: ONE 512-thread block per CU walks a static list of work items, and the LDS-DMA
// pipeline never drains between them (the two K-tiles the schedule keeps in flight are simply the next item's first two).
//   items of block b (G = grid = number of CUs, T = tiles, nt = K-tiles per tile, even):
//     * full rounds: tile L = j*G + (b&7)*(G/8) + (b>>3) for j < T/G — the 32 blocks that share an XCD (b and b+8 do, as
//       the dispatcher deals blocks round-robin) take 32 consecutive tiles of the grouped order = 8 M-tiles x 4 N-tiles
//     * remainder (R = T mod G tiles, fewer than the chip has CUs): stream-K.  Its R*nt/2 K-tile PAIRS are dealt evenly to
//       the first Gr = min(G, 4R) blocks, contiguous ranges, so a block touches at most two of those tiles and a tile is
//       shared by <= ~4 blocks.  Each sharer drops its fp32 partial tile as a slab (register-major: 1-KiB wave stores) into
//       the caller's workspace, publishes it (agent-scope release) and draws a ticket; whoever draws the LAST ticket of
//       a tile acquires, adds the other slabs to its registers and runs the epilogue.  Nobody ever waits on another block:
//       no spin, no dependence on co-residency, every wave reaches the end of its list (cdna_hip_programming.md §5 "Projection
//       GEMM at M = 256" item 2 is the protocol; tickets are returned to zero by the reducer, so a completed launch leaves
//       the counter words as it found them: zero).
//   item boundary: the wave group that runs one barrier ahead lets the other catch up, every wave drains its accumulators
//   (plain bf16 tiles go through a private 16-row LDS staging strip BESIDE the 128 KB of operand buffers, which are already
//   receiving the next item), clears them and the stagger is re-established.  Extra barriers only add ordering.
// =================================================================================================
#define P8S_MAX_ITEMS 48
struct P8Sched { int G, Gr, full_rounds, R, nt, pp, P; int* tickets; float* slabs; };     // P = R * pp pairs (fits 31 bits: host-checked)
struct P8Item { int m0, n0, kb, ke, rt; };

__device__ __forceinline__ void p8_tile_of(const FastArgs& g, int L, int& tm, int& tn) {
    const int per_group = 8 * g.tiles_n;
    const int grp = L / per_group, first_tm = grp * 8;
    const int gsz = (g.tiles_m - first_tm) < 8 ? (g.tiles_m - first_tm) : 8;
    const int in_g = L - grp * per_group;
    tm = first_tm + in_g % gsz; tn = in_g / gsz;
}
// pair range [lo, hi) of the remainder that block b owns (empty for b >= Gr)
__device__ __forceinline__ void p8_range(const P8Sched& sc, int b, int& lo, int& hi) {
    if (b >= sc.Gr) { lo = hi = 0; return; }
    lo = (int)((long long)b * sc.P / sc.Gr); hi = (int)((long long)(b + 1) * sc.P / sc.Gr);
}
__device__ __forceinline__ int p8_num_items(const P8Sched& sc, int lo, int hi) {
    int n = sc.full_rounds;
    if (hi > lo) { n += 1; if (hi > (lo / sc.pp + 1) * sc.pp) n += 1; }
    return n;
}
__device__ __forceinline__ P8Item p8_item(const FastArgs& g, const P8Sched& sc, int b, int j, int lo, int hi) {
    P8Item it; int L;
    // whole tiles first, the shared (remainder) items last.  Measured the other way round (remainder first, so that the slab
    // traffic runs under the other blocks' whole tiles): 15-25 % SLOWER — blocks leave the remainder at different times and the
    // 32 blocks of an XCD then no longer walk K in lockstep, which is what lets one fetched panel slice serve 4-8 of them from L2
    if (j < sc.full_rounds) { L = j * sc.G + (b & 7) * (sc.G >> 3) + (b >> 3); it.kb = 0; it.ke = sc.nt; it.rt = -1; }
    else {
        const int t1 = lo / sc.pp;
        if (j == sc.full_rounds) {
            const int e = hi < (t1 + 1) * sc.pp ? hi : (t1 + 1) * sc.pp;
            it.rt = t1; it.kb = (lo - t1 * sc.pp) * 2; it.ke = (e - t1 * sc.pp) * 2;
        } else { it.rt = t1 + 1; it.kb = 0; it.ke = (hi - (t1 + 1) * sc.pp) * 2; }
        L = sc.full_rounds * sc.G + it.rt;
        if (it.kb == 0 && it.ke == sc.nt) it.rt = -1;                   // the whole tile after all: no sharing
    }
    int tm, tn; p8_tile_of(g, L, tm, tn);
    it.m0 = tm * 256; it.n0 = tn * 256;
    return it;
}
// block that owns remainder pair x (ranges are floor(b*P/Gr) .. floor((b+1)*P/Gr))
__device__ __forceinline__ int p8_owner(const P8Sched& sc, int x) { return (int)((((long long)x + 1) * sc.Gr - 1) / sc.P); }

#define P8S_STAGE_BYTES (16 * 144)            // per-wave epilogue strip: 16 rows x (128 B + 16 B pad)
template <typename TC>
__global__ __launch_bounds__(512, 2)
void gemm_nt_bf16_p8_kernel(FastArgs g, P8Sched sc) {
    // one LDS array (a second __shared__ object next to an LDS-DMA staging array can make hipcc drain vmcnt before every
    // ds_read): 128 KB operand buffers | 8 private epilogue strips | ticket flag | the block's item table
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 4 * P8_HT + (8 * P8S_STAGE_BYTES + 64 + P8S_MAX_ITEMS * 32) / 2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int b = blockIdx.x;
    // remainder ranges are dealt in XCD order too: the blocks of one XCD (b, b+8, ...) take CONSECUTIVE ranges, i.e. neighbouring
    // tiles of the grouped order that share operand panels in that XCD's L2, and the sharers of a tile sit on one XCD (the
    // reducer reads same-XCD slabs).  v = position of this block in that order; slabs and owners are indexed by v.
    const int v = (sc.Gr & 7) == 0 ? ((b & 7) * (sc.Gr >> 3) + (b >> 3)) : b;
    int lo, hi;
    p8_range(sc, (b < sc.Gr) ? v : sc.Gr, lo, hi);
    const int n_items = p8_num_items(sc, lo, hi);
    if (n_items == 0) return;                                          // uniform for the whole block
    char* const extra = reinterpret_cast<char*>(smem + 2 * 4 * P8_HT);
    char* const stage = extra + wave * P8S_STAGE_BYTES;
    int* const flag = reinterpret_cast<int*>(extra + 8 * P8S_STAGE_BYTES);
    int* const itab = reinterpret_cast<int*>(extra + 8 * P8S_STAGE_BYTES + 64);
    if ((int)threadIdx.x < n_items) {                                  // item table: the divisions happen once, here
        const P8Item it = p8_item(g, sc, b, threadIdx.x, lo, hi);
        int* e = itab + threadIdx.x * 8;
        e[0] = it.m0; e[1] = it.n0; e[2] = it.kb; e[3] = it.ke; e[4] = it.rt;
    }
    __syncthreads();
#define P8S_ITEM(dst, j_) P8Item dst; { const int* e_ = itab + (j_) * 8; dst.m0 = __builtin_amdgcn_readfirstlane(e_[0]); dst.n0 = __builtin_amdgcn_readfirstlane(e_[1]); \
        dst.kb = __builtin_amdgcn_readfirstlane(e_[2]); dst.ke = __builtin_amdgcn_readfirstlane(e_[3]); dst.rt = __builtin_amdgcn_readfirstlane(e_[4]); }

    // ---- per-lane constants of the DMA source addressing
    int lrA[2], lrB[2], chk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (wave * 2 + i) * 8 + (lane >> 3);               // row of the half-tile image
        chk[i] = ((lane & 7) ^ ((lr >> 1) & 7)) * 8;
        lrA[i] = (lr >> 6) * 128 + (lr & 63);
        lrB[i] = (lr >> 5) * 64 + (lr & 31);
    }
    unsigned offA1[2], offA0[2], offB0[2], offB1[2];                   // cursor c1 owns offA1 (A-h1 of K-tile g+1), c2 the other three (g+2)
#define P8S_OFFA(dst, m0_, h) _Pragma("unroll") for (int i = 0; i < 2; ++i) { int ra = (m0_) + lrA[i] + (h) * 64; ra = ra < g.M ? ra : g.M - 1; \
        dst[i] = (unsigned)(ra * g.lda + chk[i]); }
#define P8S_OFFB(dst, n0_, h) _Pragma("unroll") for (int i = 0; i < 2; ++i) { int rb = (n0_) + lrB[i] + (h) * 32; rb = rb < g.N ? rb : g.N - 1; \
        dst[i] = (unsigned)(rb * g.ldb + chk[i]); }

    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int sw = ((lane & 15) >> 1) & 7, c0 = lane >> 4;
    int aBase[2], bBase[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aBase[ks] = (wr * 64 + (lane & 15)) * 64 + (((c0 + 4 * ks) ^ sw) << 3);
        bBase[ks] = (wc * 32 + (lane & 15)) * 64 + (((c0 + 4 * ks) ^ sw) << 3);
    }
    bf16x8 fa[2][4], fb0[2][2], fb1[2][2];

#define P8_RD(off) (*reinterpret_cast<const bf16x8*>(smem + (off)))
#define P8_LDA(b_, X) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) \
        fa[ks][ii] = P8_RD(((b_) * 4 + (X)) * P8_HT + aBase[ks] + ii * 16 * 64);
#define P8_LDB(dst, b_, X) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) \
        dst[ks][jj] = P8_RD(((b_) * 4 + 2 + (X)) * P8_HT + bBase[ks] + jj * 16 * 64);
#define P8_PF(b_, slot, base, off) _Pragma("unroll") for (int i = 0; i < 2; ++i) \
        glds16((base) + (off)[i], smem + ((b_) * 4 + (slot)) * P8_HT + (wave * 2 + i) * 8 * 64);
#define P8_MMA(mh, fbv, nh) __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) \
            acc[(nh) * 2 + jj][(mh) * 4 + ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbv[ks][jj], fa[ks][ii], acc[(nh) * 2 + jj][(mh) * 4 + ii], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);
#define P8_BAR __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);

    // ---- prefetch cursors over the block's stream of K-tiles: (item index, K-tile index inside the tile, end of the item)
    int j1 = 0, k1, ke1, j2 = 0, k2, ke2;
    {
        P8S_ITEM(it0, 0)
        k1 = k2 = it0.kb; ke1 = ke2 = it0.ke;
        P8S_OFFA(offA1, it0.m0, 1) P8S_OFFA(offA0, it0.m0, 0) P8S_OFFB(offB0, it0.n0, 0) P8S_OFFB(offB1, it0.n0, 1)
    }
    // step a cursor to the next K-tile of the stream; past the end of the list it stays on the last K-tile (redundant re-loads
    // into buffers nobody reads any more keep the vmcnt arithmetic constant)
#define P8S_ADV1 { if (k1 + 1 < ke1) ++k1; else if (j1 + 1 < n_items) { ++j1; P8S_ITEM(nx, j1) k1 = nx.kb; ke1 = nx.ke; \
                   P8S_OFFA(offA1, nx.m0, 1) } }
#define P8S_ADV2 { if (k2 + 1 < ke2) ++k2; else if (j2 + 1 < n_items) { ++j2; P8S_ITEM(nx, j2) k2 = nx.kb; ke2 = nx.ke; \
                   P8S_OFFA(offA0, nx.m0, 0) P8S_OFFB(offB0, nx.n0, 0) P8S_OFFB(offB1, nx.n0, 1) } }
#define P8S_TILE(b_) { \
        /* ph1 */ P8_LDB(fb0, b_, 0) __builtin_amdgcn_sched_barrier(0); P8_LDA(b_, 0) P8_PF((b_) ^ 1, 1, g.A + (long long)k1 * FT_BK, offA1) \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); P8_BAR P8_MMA(0, fb0, 0) P8_BAR \
        P8S_ADV1 \
        /* ph2 */ P8_LDB(fb1, b_, 1) P8_PF(b_, 2, g.B + (long long)k2 * FT_BK, offB0) P8_BAR P8_MMA(0, fb1, 1) P8_BAR \
        /* ph3 */ P8_LDA(b_, 1) P8_PF(b_, 0, g.A + (long long)k2 * FT_BK, offA0) P8_BAR P8_MMA(1, fb1, 1) P8_BAR \
        /* ph4 */ P8_PF(b_, 3, g.B + (long long)k2 * FT_BK, offB1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); P8_BAR P8_MMA(1, fb0, 0) P8_BAR \
        P8S_ADV2 \
    }

    // ---- prologue: stream tile 0 complete, three half-tiles of stream tile 1 in flight (its A-h1 follows in tile 0's ph1)
    {
        const bf16_t* pA0 = g.A + (long long)k2 * FT_BK;
        const bf16_t* pB0 = g.B + (long long)k2 * FT_BK;
        P8_PF(0, 2, pB0, offB0) P8_PF(0, 0, pA0, offA0) P8_PF(0, 3, pB0, offB1) P8_PF(0, 1, pA0, offA1)
        P8S_ADV1 P8S_ADV2                                              // both cursors on stream tile 1
        const bf16_t* pA1 = g.A + (long long)k2 * FT_BK;
        const bf16_t* pB1 = g.B + (long long)k2 * FT_BK;
        P8_PF(1, 2, pB1, offB0) P8_PF(1, 0, pA1, offA0) P8_PF(1, 3, pB1, offB1)
        P8S_ADV2                                                       // c2 on stream tile 2; c1 stays on tile 1 (its A-h1 is issued by tile 0's ph1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        P8_BAR
    }

    for (int jc = 0; jc < n_items; ++jc) {
        P8S_ITEM(it, jc)
        if (wr == 1) { P8_BAR }                                        // second wave group runs one barrier behind
        for (int t = it.kb; t < it.ke; t += 2) {
            P8S_TILE(0)
            P8S_TILE(1)
        }
        if (wr == 0) { P8_BAR }                                        // aligned again: every wave is past its last MFMA of this item
        const int mb = it.m0 + wr * 128, nb = it.n0 + wc * 64;

        // ---- shared tile: publish the partial sums, the last arriver reduces
        bool do_epilogue = true;
        if (it.rt >= 0) {
            const int t0 = it.rt * sc.pp;
            const int b_first = p8_owner(sc, t0), b_last = p8_owner(sc, t0 + sc.pp - 1);
            const int which = (it.rt == lo / sc.pp) ? 0 : 1;
            {   // sc1 stores (write-through past this XCD's L2) + vmcnt(0): the slab is visible device-wide before the ticket is
                // drawn, without an agent-scope release (a write-back of the whole L2 in mid-launch)
                __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(sc.slabs) + ((long long)(v * 2 + which) * 8 + wave) * 32768,
                                                                             0, 32768, 0x00020000);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), r, ((j * 8 + i) * 64 + lane) * 16, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (threadIdx.x == 0) {
                const int tk = __hip_atomic_fetch_add(sc.tickets + it.rt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = tk;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int tk = *const_cast<volatile int*>(flag);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // everybody has read the flag before it can be rewritten
            do_epilogue = (tk == b_last - b_first);
            if (do_epilogue) {                                         // block-uniform
                if (threadIdx.x == 0) __hip_atomic_store(sc.tickets + it.rt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // leave the counter as we found it
                // the other slabs are read with sc1 loads (never served from this XCD's L2): no acquire / L2 invalidate, which would
                // also evict the operand panels the neighbouring CUs are streaming
                // two sharers: mine + theirs (commutative, so the arrival order cannot show).  More: sum EVERY slab in block order,
                // my own included (read back from the workspace), so the result does not depend on who arrived last
                const bool all = (b_last - b_first) >= 2;
                if (all) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                for (int ob = b_first; ob <= b_last; ++ob) {                 // ob, b_first, b_last: positions in the v order
                    if (ob == v && !all) continue;
                    int olo, ohi;
                    p8_range(sc, ob, olo, ohi);
                    const int ow = (it.rt == olo / sc.pp) ? 0 : 1;
                    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(sc.slabs) + ((long long)(ob * 2 + ow) * 8 + wave) * 32768,
                                                                                 0, 32768, 0x00020000);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            acc[j][i] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ((j * 8 + i) * 64 + lane) * 16, 0, 16));
                }
            }
        }

        if (do_epilogue) {
            if (sizeof(TC) == 2 && !g.bias && !g.residual && !g.accumulate && g.act == 0 && g.alpha == 1.0f &&
                nb + 64 <= g.N && (g.ldc & 7) == 0 && ((uintptr_t)g.C & 15) == 0) {          // wave-uniform
                // plain bf16 tile: 16 rows x 64 columns at a time through the wave's private strip, out as 128-B row segments
                bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = acc[j][ii];
                        u32x2 o;
                        o[0] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                        o[1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                        *reinterpret_cast<u32x2*>(stage + (lane & 15) * 144 + (j * 16 + (lane >> 4) * 4) * 2) = o;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int itr = 0; itr < 2; ++itr) {
                        const int r = itr * 8 + (lane >> 3), ch = lane & 7;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * 144 + ch * 16);
                        const int m = mb + ii * 16 + r;
                        if (m < g.M) *reinterpret_cast<u32x4*>(C + (long long)m * g.ldc + nb + ch * 8) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // strip is read before the next pass overwrites it
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                gemm_epilogue<TC, 8>(g, acc, mb, nb, lane);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the tail's redundant DMAs drain before the block's LDS is released
#undef P8S_TILE
#undef P8S_ITEM
#undef P8S_ADV2
#undef P8S_ADV1
#undef P8S_OFFB
#undef P8S_OFFA
#undef P8_BAR
#undef P8_MMA
#undef P8_PF
#undef P8_LDB
#undef P8_LDA
#undef P8_RD
}


// split-K combine: C = act(alpha * sum_s slab[s] + bias) + residual (+C); 4 columns per thread (16-B slab loads)
template <typename TC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(FastArgs g) {
    const long long total = (long long)g.M * g.N;
    const int nq = (g.N + 3) >> 2;
    const long long items = (long long)g.M * nq;
    const bool vec = (g.N & 3) == 0;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const int m = (int)(it / nq), n = (int)(it % nq) * 4;
        const long long e = (long long)m * g.N + n;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            for (int s2 = 0; s2 < g.splitk; ++s2) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(g.ws + (long long)s2 * total + e);
                v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
            }
        } else {
            for (int s2 = 0; s2 < g.splitk; ++s2)
                for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += g.ws[(long long)s2 * total + e + r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r >= g.N) continue;
            float x = act_apply(v[r] * g.alpha + (g.bias ? bf2f(g.bias[n + r]) : 0.f), g.act);
            TC* cp = reinterpret_cast<TC*>(g.C) + (long long)m * g.ldc + n + r;
            if (g.residual) x += Cvt<TC>::ld(reinterpret_cast<const TC*>(g.residual) + (long long)m * g.ldr + n + r);
            if (g.accumulate) x += Cvt<TC>::ld(cp);
            Cvt<TC>::st(cp, x);
        }
    }
}

// EGOMI_EPI_SWIGLU_BWD, K-sliced tail rows: d(act) = sum of the slabs (rounded to bf16 as the whole tiles round it), combined with gate|up into
// d(gate|up) in the same pass.  g.M / g.N = tail rows / hidden units; C, C2 already point at the first tail row.  8 units per thread.
__global__ __launch_bounds__(256) void splitk_reduce_swiglu_bwd_kernel(FastArgs g) {
    const long long total = (long long)g.M * g.N;
    const int nq = g.N >> 3;
    const long long items = (long long)g.M * nq;
    const bf16_t* GU = reinterpret_cast<const bf16_t*>(g.C2);
    bf16_t* DGU = reinterpret_cast<bf16_t*>(g.C);
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const int m = (int)(it / nq), c = (int)(it % nq) * 8;
        const long long e = (long long)m * g.N + c;
        float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < g.splitk; ++s2) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(g.ws + (long long)s2 * total + e);
            const f32x4 y = *reinterpret_cast<const f32x4*>(g.ws + (long long)s2 * total + e + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { d[r] += x[r]; d[4 + r] += y[r]; }
        }
        const int ci = ((c >> 5) << 6) + (c & 31);
        float gv[8], uv[8], og[8], ou[8];
        load8<bf16_t>(GU + (long long)m * g.ldc2 + ci, gv);
        load8<bf16_t>(GU + (long long)m * g.ldc2 + ci + 32, uv);
#pragma unroll
        for (int r = 0; r < 8; ++r) swiglu_bwd_elem(bf2f(f2bf(d[r])), gv[r], uv[r], og[r], ou[r]);
        store8<bf16_t>(DGU + (long long)m * g.ldc + ci, og);
        store8<bf16_t>(DGU + (long long)m * g.ldc + ci + 32, ou);
    }
}

static bool fast_applicable(const egomi_gemm_desc* d) {
    if (d->ab_dtype != EGOMI_BF16 || d->a_layout != 0 || d->b_layout != 0) return false;
    if (d->batch > 1) return false;
    if (d->K % FT_BK || d->K < FT_BK) return false;
    if (d->lda % 8 || d->ldb % 8) return false;
    if (((uintptr_t)d->A | (uintptr_t)d->B) & 15) return false;
    const int esz = d->c_dtype == EGOMI_BF16 ? 2 : 4;
    if (((uintptr_t)d->C % (4 * esz)) || (d->residual && ((uintptr_t)d->residual % (4 * esz)))) return false;
    if ((long long)d->M * d->N < 128 * 128) return false;          // tiny products: the generic kernel is fine
    return true;
}

// tile choice: 8 = 256x256 8-phase kernel, 2 = 256x128, 1 = 128x128.  EGOMI_GEMM_TILE=1|2|8 overrides (A/B runs).
static int tile_choice(const egomi_gemm_desc* d) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("EGOMI_GEMM_TILE"); forced = e ? atoi(e) : 0; }
    if (forced == 1 || forced == 2 || forced == 8) return forced;
    // measured (tools/gemm_bench.py, M=5536): 256x128 wins only where N is wide enough to keep every CU at
    // 2 resident blocks to the end (N=11008: 1168 vs 1084 TFLOP/s); at N=4096 its 704 tiles quantise worse
    // than 1408 tiles of 128x128 (952 vs 1010)
    // 256x256 8-phase kernel (tools/gemm_bench.py: 1.06-1.32 PFLOP/s at M=5536, 1.3-1.49 at M=8192 vs ~1.0-1.1): needs
    // enough tiles to occupy the chip at one block per CU and a K long enough to amortise its prologue/epilogue
    const long long t256 = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256);
    if (t256 >= 128 && d->K >= 2048 && (long long)d->M * d->lda < (1ll << 31) && (long long)d->N * d->ldb < (1ll << 31)) return 8;
    // many tiles make up for a shorter K (lm_head wgrad of the step: 32262 x 4096 x 1280 = 2032 tiles, fp32 out: 365 vs 474 us,
    // tools/debug/lmhead_wgrad_probe.py)
    if (t256 >= 512 && d->K >= 1024 && (long long)d->M * d->lda < (1ll << 31) && (long long)d->N * d->ldb < (1ll << 31)) return 8;
    // few tiles but a very long K (the lm_head dgrad of the training step: 1280 x 4096 x 32320 = 80 tiles): every tile row K-sliced
    // (plan_tail with rows = all) fills the chip — tools/debug/lmhead_dgrad_probe.py: 313 us (S = 6) vs 482 us on the 128x128 kernel
    if (t256 >= 32 && t256 < 128 && d->K >= 8192 && d->M > 512 && d->epilogue == EGOMI_EPI_NONE && d->workspace &&
        d->workspace_bytes >= (long long)((d->M + 255) / 256) * 256 * d->N * 4 * 2 + 4096 &&
        (long long)d->M * d->lda < (1ll << 31) && (long long)d->N * d->ldb < (1ll << 31)) return 8;
    return (d->M >= 2048 && d->N >= 8192) ? 2 : 1;
}

extern "C" int egomi_gemm_tn_kernel_id(const egomi_gemm_desc* d);      // gemm_tn.hip: 3 when the k-major 8-phase kernel takes the product
extern "C" int egomi_gemm_kernel_id(const egomi_gemm_desc* d) {
    if (!d) return EGOMI_E_BADARG;
    if (!d->force_generic && !fast_applicable(d)) return egomi_gemm_tn_kernel_id(d);
    if (d->force_generic || !fast_applicable(d)) return 0;
    return tile_choice(d) == 8 ? 2 : 1;
}

static bool slabs_form_ok(const egomi_gemm_desc* d);
static int skinny_splitk(const egomi_gemm_desc* d, int nwg);
static bool gemv_form(const egomi_gemm_desc* d);
static int gemv_splitk(const egomi_gemm_desc* d);
// 128 < M <= 256 with a long K and enough columns: the all-rows ring kernel (gemm_nt_bf16_m256_kernel).  EGOMI_GEMM_M256=0: the 128x128 kernel (A/B runs)
static bool m256_form(const egomi_gemm_desc* d) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("EGOMI_GEMM_M256"); on = e ? atoi(e) : 1; }
    // N >= 8192 (q|k|v, gate|up, lm_head): measured per shape (tools/gemm_bench_decode.py, M = 256): 41 vs 44 us, 57 vs 61, 77 vs 90; the N = 4096 products
    // (o_proj, down_proj: 32 column tiles x 8 K-slices of 8-22 steps) are dominated by per-block fixed cost and stay on the 128x128 kernel (24.7 vs 21.9 us)
    return on && d->M > 128 && d->M <= 256 && d->N >= 8192 && d->K >= 1024;
}
static int m256_splitk(const egomi_gemm_desc* d) {
    const int nwg = (d->N + 127) / 128, nt = d->K / FT_BK;
    if (!d->workspace) return 1;
    int sk = d->split_k > 0 ? d->split_k : (nwg > 170 ? 1 : 256 / nwg);        // one round of <= 256 blocks
    if (sk > nt / 4) sk = nt / 4;                                              // at least 4 K-steps per slice
    const long long per_slab = (long long)d->M * d->N * 4;
    if ((long long)sk * per_slab > d->workspace_bytes) sk = (int)(d->workspace_bytes / per_slab);
    if (sk > 1) { const int per = (nt + sk - 1) / sk; sk = (nt + per - 1) / per; }     // no empty slices
    return sk > 1 ? sk : 1;
}
extern "C" int egomi_gemm_slab_count(const egomi_gemm_desc* d) {
    if (!d) return EGOMI_E_BADARG;
    if (d->force_generic || !fast_applicable(d) || !slabs_form_ok(d) || tile_choice(d) != 1) return 0;
    const int sk = gemv_form(d) ? gemv_splitk(d) : (m256_form(d) ? m256_splitk(d) : skinny_splitk(d, ((d->M + 127) / 128) * ((d->N + 127) / 128)));
    return sk >= 2 ? sk : 0;
}

// skinny products (decode, M <= 512): too few tiles to fill 256 CUs and each block is DMA-latency bound, so the K range is
// split over blockIdx.y into fp32 slabs (caller-provided workspace) and combined.  -> number of K-slices (1 = no split)
static int skinny_splitk(const egomi_gemm_desc* d, int nwg) {
    const int nt = d->K / FT_BK;
    // ... and products of any height whose 128x128 tiles cover well under half the chip while K is long enough to cut
    // (PointBERT fc2 at B = 8: 4104 x 384 x 1536 = 99 tiles x 24 K-steps)
    const bool few_tiles = nwg <= 128 && nt >= 16 && d->epilogue == EGOMI_EPI_NONE;
    if (!(d->workspace && (d->M <= 512 || few_tiles) && nwg < 512)) return 1;
    // measured (tools/gemm_bench_decode.py, M=256): ~256 blocks, and no more than ~32 K-steps per slice
    int sk = d->split_k > 0 ? d->split_k : (nwg >= 256 ? 1 : (256 + nwg - 1) / nwg);
    if (d->split_k <= 0 && sk > 1 && nt / sk > 32) sk *= 2;
    if (sk > nt) sk = nt;
    const long long per_slab = (long long)d->M * d->N * 4;
    if ((long long)sk * per_slab > d->workspace_bytes) sk = (int)(d->workspace_bytes / per_slab);
    if (sk > 1) { const int per = (nt + sk - 1) / sk; sk = (nt + per - 1) / per; }     // no empty slices
    return sk > 1 ? sk : 1;
}

// M <= 16 (gemv_m16_kernel).  EGOMI_GEMM_GEMV=0: the 128x128 kernel (A/B runs)
static bool gemv_form(const egomi_gemm_desc* d) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("EGOMI_GEMM_GEMV"); on = e ? atoi(e) : 1; }
    return on && d->M <= 16 && d->K >= 1024 && d->K % 128 == 0 && d->N >= 64 && (d->epilogue == EGOMI_EPI_NONE || d->epilogue == EGOMI_EPI_SLABS);
}
static int gemv_splitk(const egomi_gemm_desc* d) {
    const int nwg = (d->N + GV_BN - 1) / GV_BN, T = d->K / 128;
    if (!d->workspace || (d->N & 3)) return 1;
    // measured at M = 8 (tools/gemm_bench_decode.py 8, rotating weights): ~700 blocks of 4 waves (2.7 per CU) — q|k|v 4 slices 26.4 us (3 slices 29.8),
    // o_proj 4 slices 14.0, gate|up 2 slices 40.9 (whole K 47.9, 3 slices 44.6), down_proj 4-8 slices 25.2-25.9, lm_head whole K 50.1 us = 5.3 TB/s
    int sk = d->split_k > 0 ? d->split_k : (704 + nwg / 2) / nwg;
    if (sk < 1) sk = 1;
    if (sk > T / 8) sk = T / 8;
    if (d->epilogue == EGOMI_EPI_SLABS && sk < 2 && T >= 16) sk = 2;
    const long long per_slab = (long long)d->M * d->N * 4;
    if ((long long)sk * per_slab > d->workspace_bytes) sk = (int)(d->workspace_bytes / per_slab);
    return sk > 1 ? sk : 1;
}
static int launch_gemv(const egomi_gemm_desc* d, FastArgs& g, hipStream_t s) {
    g.tiles_m = 1; g.tiles_n = (d->N + GV_BN - 1) / GV_BN;
    g.ws = (float*)d->workspace;
    g.splitk = gemv_splitk(d);
    if (d->epilogue == EGOMI_EPI_SLABS && g.splitk < 2) return EGOMI_E_UNSUPPORTED;     // egomi_gemm_slab_count said so
    if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH(gemv_m16_kernel<bf16_t>, dim3(g.tiles_n, g.splitk), dim3(256), 0, s, g);
    else if (d->c_dtype == EGOMI_F32) EGOMI_LAUNCH(gemv_m16_kernel<float>, dim3(g.tiles_n, g.splitk), dim3(256), 0, s, g);
    else return EGOMI_E_UNSUPPORTED;
    if (g.splitk > 1 && d->epilogue != EGOMI_EPI_SLABS) {
        const long long total = (long long)d->M * ((d->N + 3) / 4);
        const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH(splitk_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, g);
        else EGOMI_LAUNCH(splitk_reduce_kernel<float>, dim3(grid), dim3(256), 0, s, g);
    }
    return egomi_launch_status();
}

static int launch_m256(const egomi_gemm_desc* d, FastArgs& g, hipStream_t s) {
    g.tiles_m = 1; g.tiles_n = (d->N + 127) / 128;
    g.ws = (float*)d->workspace;
    g.splitk = m256_splitk(d);
    if (d->epilogue == EGOMI_EPI_SLABS && g.splitk < 2) return EGOMI_E_UNSUPPORTED;     // egomi_gemm_slab_count said so
    if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH(gemm_nt_bf16_m256_kernel<bf16_t>, dim3(g.tiles_n, g.splitk), dim3(512), 0, s, g);
    else if (d->c_dtype == EGOMI_F32) EGOMI_LAUNCH(gemm_nt_bf16_m256_kernel<float>, dim3(g.tiles_n, g.splitk), dim3(512), 0, s, g);
    else return EGOMI_E_UNSUPPORTED;
    if (g.splitk > 1 && d->epilogue != EGOMI_EPI_SLABS) {                // EGOMI_EPI_SLABS: the caller's next kernel sums the slabs
        const long long total = (long long)d->M * ((d->N + 3) / 4);
        const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH(splitk_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, g);
        else EGOMI_LAUNCH(splitk_reduce_kernel<float>, dim3(grid), dim3(256), 0, s, g);
    }
    return egomi_launch_status();
}

template <int BM, int BN>
static int launch_fast(const egomi_gemm_desc* d, FastArgs& g, hipStream_t s) {
    if (BM == 128 && BN == 128 && gemv_form(d)) return launch_gemv(d, g, s);
    if (BM == 128 && BN == 128 && m256_form(d)) return launch_m256(d, g, s);
    g.tiles_m = (d->M + BM - 1) / BM; g.tiles_n = (d->N + BN - 1) / BN;
    const int nwg = g.tiles_m * g.tiles_n;
    constexpr int threads = (BM / 64) * (BN / 64) * 64;
    g.ws = (float*)d->workspace;
    g.splitk = skinny_splitk(d, nwg);
    if (d->epilogue == EGOMI_EPI_SLABS && g.splitk < 2) return EGOMI_E_UNSUPPORTED;     // egomi_gemm_slab_count said so
    static int pf_env = -1;                                            // EGOMI_GEMM_PF=0: one LDS stage (A/B switch)
    if (pf_env < 0) { const char* e = getenv("EGOMI_GEMM_PF"); pf_env = e ? atoi(e) : 1; }
    if (d->c_dtype != EGOMI_BF16 && d->c_dtype != EGOMI_F32) return EGOMI_E_UNSUPPORTED;
    if (BM == 128 && BN == 128 && pf_env) {
        if constexpr (BM == 128 && BN == 128) {
            if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH((gemm_nt_bf16_kernel<bf16_t, 128, 128, 2>), dim3(nwg, g.splitk), dim3(threads), 0, s, g);
            else EGOMI_LAUNCH((gemm_nt_bf16_kernel<float, 128, 128, 2>), dim3(nwg, g.splitk), dim3(threads), 0, s, g);
        }
    } else if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH((gemm_nt_bf16_kernel<bf16_t, BM, BN, 1>), dim3(nwg, g.splitk), dim3(threads), 0, s, g);
    else EGOMI_LAUNCH((gemm_nt_bf16_kernel<float, BM, BN, 1>), dim3(nwg, g.splitk), dim3(threads), 0, s, g);
    if (g.splitk > 1 && d->epilogue != EGOMI_EPI_SLABS) {                // EGOMI_EPI_SLABS: the caller's next kernel sums the slabs
        const long long total = (long long)d->M * ((d->N + 3) / 4);
        const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH(splitk_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, g);
        else EGOMI_LAUNCH(splitk_reduce_kernel<float>, dim3(grid), dim3(256), 0, s, g);
    }
    return egomi_launch_status();
}

// ---- ragged last round.  One 256x256 block per CU means ceil(tiles / 256) rounds; at M = 5536, N = 4096 that is 2
// rounds for 1.375 rounds of work.  The last `rows` tile rows are therefore cut into S K-slices (fp32 slabs of those rows
// only, combined by splitk_reduce_kernel): short blocks that the dispatcher packs into the last round.  (rows, S) come from
// a list-scheduling model in K-tile units, cached per shape.
struct TailPlan { int rows, s; };
static TailPlan plan_tail(int M, int N, int K, long long ws_bytes) {
    static std::mutex mu;
    static std::unordered_map<unsigned long long, TailPlan> cache;
    const unsigned long long key = ((unsigned long long)M << 42) ^ ((unsigned long long)N << 21) ^ (unsigned long long)K ^ (ws_bytes ? 1ull << 63 : 0);
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    const int tm = (M + 255) / 256, tn = (N + 255) / 256, nt = K / FT_BK;
    const double F = 7.0;                                             // prologue + epilogue of one block, in K-tile times (~1.4 us each)
    auto model = [&](int rows, int S) -> double {
        const int full = (tm - rows) * tn, Q = rows * tn * S;
        const double cf = nt + F, cs = (nt + S - 1) / S + F + 1.0;
        std::vector<double> t(256);
        const int R = full / 256, r = full % 256;
        for (int i = 0; i < 256; ++i) t[i] = (i < r ? R + 1 : R) * cf;
        std::priority_queue<double, std::vector<double>, std::greater<double>> pq(t.begin(), t.end());
        double end = (r ? R + 1 : R) * cf;
        for (int q = 0; q < Q; ++q) { double x = pq.top() + cs; pq.pop(); pq.push(x); if (x > end) end = x; }
        if (rows) end += 3.0 + (double)(M - (tm - rows) * 256) * N * 4.0 * (S + 1) / 6.3e6;     // combine pass: launch + slab traffic
        return end;
    };
    TailPlan best = {0, 1};
    double tbest = model(0, 1);
    const double t0 = tbest;
    const int smax = (long long)tm * tn < 128 ? 8 : 4;                // few tiles, long K: finer slices (every row sliced)
    for (int rows = 1; rows <= tm && rows <= 8; ++rows)
        for (int S = 2; S <= smax; ++S) {
            if (nt / S < 8) continue;
            if ((long long)(M - (tm - rows) * 256) * N * 4 * S > ws_bytes) continue;
            const double t = model(rows, S);
            if (t < tbest) { tbest = t; best = {rows, S}; }
        }
    if (tbest > 0.97 * t0) best = {0, 1};                             // not worth a second launch
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = best;
    return best;
}

// the same plan and combine pass for the k-major kernel (gemm_tn.hip)
void egomi_plan_tail_rows(int M, int N, int K, long long ws_bytes, int* rows, int* slices) {
    const TailPlan tp = plan_tail(M, N, (K + FT_BK - 1) / FT_BK * FT_BK, ws_bytes);
    *rows = tp.rows; *slices = tp.s;
}
int egomi_splitk_reduce_rows(void* C, int c_dtype, long long ldc, int rows, int N, const float* ws, int slices, int accumulate, hipStream_t s) {
    FastArgs r = {};
    r.C = C; r.ldc = ldc; r.M = rows; r.N = N; r.ws = const_cast<float*>(ws); r.splitk = slices; r.accumulate = accumulate; r.alpha = 1.0f;
    const long long total = (long long)rows * ((N + 3) / 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (c_dtype == EGOMI_BF16) EGOMI_LAUNCH(splitk_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, r);
    else if (c_dtype == EGOMI_F32) EGOMI_LAUNCH(splitk_reduce_kernel<float>, dim3(grid), dim3(256), 0, s, r);
    else return EGOMI_E_UNSUPPORTED;
    return egomi_launch_status();
}

// persistent launch: schedule in a handful of integers, everything else is derived inside the kernel.  Needs the caller's
// workspace (4 KB of ticket words, zero on entry and left zero, then G*2 slabs of 256 KB) and an even number of K-tiles.
static int p8_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 8) n = v & ~7;
        else n = 256;
    }
    return n;
}
static bool p8_applicable(const egomi_gemm_desc* d, P8Sched& sc) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("EGOMI_GEMM_PERSIST"); off = (e && atoi(e) == 0) ? 1 : 0; }
    if (off || !d->workspace || !d->ws_tickets_zeroed || d->split_k > 0) return false;
    const int nt = d->K / FT_BK;
    if (nt < 4 || (nt & 1)) return false;
    const int tm = (d->M + 255) / 256, tn = (d->N + 255) / 256;
    const long long T = (long long)tm * tn;
    sc.G = p8_cus();
    if ((long long)4096 + (long long)sc.G * 2 * 262144 > d->workspace_bytes) return false;
    sc.full_rounds = (int)(T / sc.G); sc.R = (int)(T % sc.G); sc.nt = nt; sc.pp = nt / 2;
    if (sc.full_rounds + 2 > P8S_MAX_ITEMS || sc.R > 1000) return false;
    // Selection.  Measured at M = 5536 on one box, A/B in one process:
    //   * tools/gemm_bench.py (the same operands launch after launch, i.e. weights resident in the 256-MB Infinity Cache):
    //     persistent +1...5 % over the per-tile kernel + combine launch when the remainder is at most half a round and
    //     K >= 4096; -12 % with a large remainder (N = 11008: 178 of 256 tiles), a tie at K = 2048
    //   * bench.py, the real step (every product reads DIFFERENT weights, cold from HBM): persistent 143.5 ms/step vs 137.1,
    //     its launches averaging 441.6 us vs 417.4 — the static assignment cannot re-balance around slow HBM fetches the way
    //     the dispatcher does when it hands out one tile at a time, and the stream-K remainder walks K out of step (L2 misses).
    // So the library's own rule never picks it; ws_tickets_zeroed = 2 (ops.mm(persistent=True)) or EGOMI_GEMM_PERSIST=2 do.
    static int force = -1;
    if (force < 0) { const char* e = getenv("EGOMI_GEMM_PERSIST"); force = (e && atoi(e) == 2) ? 1 : 0; }
    if (!force && d->ws_tickets_zeroed != 2) return false;
    sc.P = sc.R * sc.pp;
    sc.Gr = sc.R == 0 ? 8 : (sc.G < 4 * sc.R ? sc.G : 4 * sc.R);          // <= ~4 sharers per remainder tile
    if (sc.P > 0 && sc.Gr > sc.P) sc.Gr = sc.P;
    if (sc.Gr >= 8) sc.Gr &= ~7;                                             // multiple of 8: remainder ranges follow the XCD order too
    sc.tickets = (int*)d->workspace;
    sc.slabs = (float*)((char*)d->workspace + 4096);
    return true;
}
static int launch_p8(const egomi_gemm_desc* d, FastArgs& g, const P8Sched& sc, hipStream_t s) {
    g.tiles_m = (d->M + 255) / 256; g.tiles_n = (d->N + 255) / 256;
    g.splitk = 1; g.ws = nullptr; g.full_tm = g.tiles_m; g.full_tiles = g.tiles_m * g.tiles_n; g.tail_s = 1;
    const long long T = (long long)g.tiles_m * g.tiles_n;
    const int grid = T < sc.G ? (sc.Gr > 0 ? sc.Gr : 1) : sc.G;          // fewer tiles than CUs: only the sharers of the remainder have work
    if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH(gemm_nt_bf16_p8_kernel<bf16_t>, dim3(grid), dim3(512), 0, s, g, sc);
    else if (d->c_dtype == EGOMI_F32) EGOMI_LAUNCH(gemm_nt_bf16_p8_kernel<float>, dim3(grid), dim3(512), 0, s, g, sc);
    else return EGOMI_E_UNSUPPORTED;
    return egomi_launch_status();
}

// ---- 352x256 form (gemm_nt_bf16_tall_kernel).  EGOMI_GEMM_TALL / egomi_gemm_set_tall: 0 never, 2 wherever it applies (A/B runs, tests), 1 (default) by the
// round model below, in 256x256 tile-times, fitted to tools/debug/tall_probe.py at M = 5536 (cold weights, one box):
//   256x256 plan: whole rounds + (0.28 + 0.85 x fill) for a ragged last round whose rows are K-sliced (1.6 for 1.375 rounds at N = 4096, 4.39 for 4.125 at N = 12288)
//   352x256 plan: 1.31 per round of its own tiles (1.375 x the flops; 0.34 instead of 0.375 ds_read per MFMA and 12 instead of 16 barriers per 128 MFMAs)
//   N = 4096: K = 4096 183.5 -> 158.8 us, K = 11008 404.9 -> 355.9, K = 12288 452.7 -> 382.1, K = 22016 758.0 -> 712.7; N = 12288 425.9 -> 399.9;
//   N = 11008 / 22016 (0.70 / 0.39 of a round left over, 688 / 1376 tall tiles = 2.69 / 5.375 rounds): 373.8 vs 376.2, 746.9 vs 753.1 — stay on the 256x256 form
static int g_tall_mode = -1;
extern "C" int egomi_gemm_set_tall(int mode) { g_tall_mode = mode < 0 ? -1 : (mode > 2 ? 2 : mode); return EGOMI_OK; }
static bool tall_form(const egomi_gemm_desc* d) {
    if (g_tall_mode < 0) { const char* e = getenv("EGOMI_GEMM_TALL"); g_tall_mode = e ? atoi(e) : 1; if (g_tall_mode < 0 || g_tall_mode > 2) g_tall_mode = 1; }
    const int mode = g_tall_mode;
    if (!mode) return false;
    if (d->epilogue != EGOMI_EPI_NONE && d->epilogue != EGOMI_EPI_SLABS && d->epilogue != EGOMI_EPI_SWIGLU && d->epilogue != EGOMI_EPI_SWIGLU_BWD) return false;
    if ((d->M & 7) || (d->N & 7) || d->M < TL_BM || d->N < 256 || d->K < 2048 || d->split_k > 1) return false;      // (split_k = 1: "no K-sliced rows", the tests' way to compare whole tiles of both forms)
    if ((long long)d->M * d->lda >= (1ll << 31) || (long long)d->N * d->ldb >= (1ll << 31)) return false;
    if (d->c_dtype != EGOMI_BF16 && d->c_dtype != EGOMI_F32) return false;
    if (mode == 2) return true;
    const int ncu = p8_cus();                                            // CUs of this device (256 on MI355X in SPX mode)
    const long long tn = (d->N + 255) / 256;
    const long long t256 = (long long)((d->M + 255) / 256) * tn, t352 = (long long)((d->M + TL_BM - 1) / TL_BM) * tn;
    const int rem = (int)(t256 % ncu);
    const double c256 = (double)(t256 / ncu) + (rem ? (rem * 2 <= ncu && d->workspace ? 0.28 + 0.85 * rem / ncu : 1.0) : 0.0);
    const double c352 = 1.31 * (double)((t352 + ncu - 1) / ncu);
    return c352 < 0.98 * c256;
}
static int launch_tall(const egomi_gemm_desc* d, FastArgs& g, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    g.tiles_m = (d->M + TL_BM - 1) / TL_BM; g.tiles_n = (d->N + 255) / 256;
    g.splitk = 1; g.ws = nullptr; g.tickets = nullptr;
    if (g.epi == EGOMI_EPI_SLABS) g.epi = 0;                          // (a product in this form has no K-sliced rows to leave as slabs)
    static int gdepth = -1;
    if (gdepth < 0) { const char* e = getenv("EGOMI_TALL_GROUP"); gdepth = e ? atoi(e) : 8; if (gdepth < 1) gdepth = 8; }
    g.full_tm = gdepth;
    const int nwg = g.tiles_m * g.tiles_n;
    if (t0) (void)hipEventRecord(t0, s);
    if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH((gemm_nt_bf16_tall_kernel<bf16_t, false>), dim3(nwg), dim3(512), 0, s, g);
    else EGOMI_LAUNCH((gemm_nt_bf16_tall_kernel<float, false>), dim3(nwg), dim3(512), 0, s, g);
    if (t1) (void)hipEventRecord(t1, s);
    return egomi_launch_status();
}

int egomi_gemm_tn_rest(const egomi_gemm_desc* d, hipStream_t s);        // gemm_tn.hip: its 256x256 kernel, without trying this form again
static int split_cols(const egomi_gemm_desc* d);
// data gradient against a k-major weight (a_layout 0, b_layout 1; called by gemm_tn.hip before its own 256x256 kernel): the 352x256 form where the round
// model prefers it.  -> 0 launched, 1 not applicable, < 0 error.  `query`: decide only.
int egomi_tall_kmajor_try(const egomi_gemm_desc* d, hipStream_t s, bool query) {
    static int on = -1;                                                  // EGOMI_GEMM_TALL_KMAJOR=0: gemm_tn.hip's 256x256 kernel for every k-major product (A/B runs)
    if (on < 0) { const char* e = getenv("EGOMI_GEMM_TALL_KMAJOR"); on = e ? atoi(e) : 1; }
    if (!on) return 1;
    if (d->a_layout != 0 || d->b_layout != 1 || d->ab_dtype != EGOMI_BF16 || d->c_dtype != EGOMI_BF16 || d->batch > 1) return 1;
    if (d->bias || d->residual || d->accumulate || d->act != 0 || d->alpha != 1.0f || d->epilogue != EGOMI_EPI_NONE) return 1;
    if ((d->N & 255) || (d->K % FT_BK) || (d->lda & 7) || (d->ldb & 7) || (d->ldc & 7) || (((uintptr_t)d->A | (uintptr_t)d->B | (uintptr_t)d->C) & 15)) return 1;
    if (d->lda < d->K || d->ldb < d->N) return 1;
    egomi_gemm_desc dn = *d;                                             // the shape rules and the round model of the K-contiguous form
    dn.b_layout = 0; dn.ldb = 8;                                         // (ldb only enters tall_form's 31-bit span check, which the k-major B does not need)
    int Na = d->N;                                                       // columns in this form: all of them, or (column split, as split_cols) the first Na
    if (!tall_form(&dn)) {
        Na = split_cols(&dn);
        if (!Na) return 1;
    }
    if (query) return Na == d->N ? 0 : 1;                                // (a split product's second part plans its own K-sliced rows: no plan to report here)
    FastArgs g = {};
    g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B; g.C = d->C;
    g.M = d->M; g.N = Na; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.alpha = 1.0f;
    g.tiles_m = (d->M + TL_BM - 1) / TL_BM; g.tiles_n = Na / 256;
    g.splitk = 1; g.full_tm = 8;
    EGOMI_LAUNCH((gemm_nt_bf16_tall_kernel<bf16_t, true>), dim3(g.tiles_m * g.tiles_n), dim3(512), 0, s, g);
    if (Na == d->N) return egomi_launch_status();
    egomi_gemm_desc d2 = *d;                                             // the rest on gemm_tn.hip's 256x256 tiles (its own tail plan)
    d2.N = d->N - Na;
    d2.B = (const bf16_t*)d->B + Na;
    d2.C = (bf16_t*)d->C + Na;
    return egomi_gemm_tn_rest(&d2, s);
}

// the tail plan launch_8phase will run for this descriptor (d->workspace already points at the slab area)
static TailPlan tail_plan_for(const egomi_gemm_desc* d) {
    const int tiles_m = (d->M + 255) / 256;
    static int no_tail = -1;
    if (no_tail < 0) { const char* e = getenv("EGOMI_GEMM_NO_TAIL"); no_tail = e ? atoi(e) : 0; }
    TailPlan tp = {0, 1};
    if (d->workspace && !no_tail) tp = plan_tail(d->M, d->N, d->K, d->workspace_bytes);
    if (d->split_k > 0 && d->workspace) {                             // explicit override (experiments): split_k = rows * 16 + S
        tp.rows = d->split_k / 16; tp.s = d->split_k % 16;
        if (tp.rows > tiles_m) tp.rows = tiles_m;
        if (tp.s < 2 || tp.rows < 1) tp = {0, 1};
    }
    if (tp.rows) {                                                    // slabs must fit the caller's scratch, slices must be non-empty
        const long long rows_rel = d->M - (long long)(tiles_m - tp.rows) * 256;
        const int nt = d->K / FT_BK;
        if (tp.s > nt) tp.s = nt;
        if (tp.s > 1) { const int per = (nt + tp.s - 1) / tp.s; tp.s = (nt + per - 1) / per; }
        if (tp.s < 2 || rows_rel * d->N * 4 * tp.s > d->workspace_bytes) tp = {0, 1};
    }
    return tp;
}

// EGOMI_EPI_SLABS on a product that takes the 256x256 kernel: rows >= *row0 are left as *slices fp32 slabs [slices][M - row0][N]
// at the start of the slab area (workspace + 4096 when ws_tickets_zeroed); *slices = 0: every row gets the normal epilogue
extern "C" int egomi_gemm_tail_plan(const egomi_gemm_desc* d0, int* row0, int* slices) {
    if (!d0 || !row0 || !slices) return EGOMI_E_BADARG;
    *row0 = d0->M; *slices = 0;
    // the plan is asked BEFORE the caller sets epilogue = EGOMI_EPI_SLABS; the launch that follows carries it, and tile_choice looks at
    // the epilogue (its few-tiles / long-K rule wants a plain one): evaluate the descriptor the launch will see, so that plan and launch
    // agree for every M (M in [1024, 1792] at K >= 8192 used to plan the 256x256 kernel and launch the 128x128 one)
    egomi_gemm_desc dl = *d0;
    dl.epilogue = EGOMI_EPI_SLABS;
    if (dl.force_generic || !fast_applicable(&dl) || tile_choice(&dl) != 8) return EGOMI_E_UNSUPPORTED;
    if (dl.bias || dl.accumulate || dl.act != 0 || dl.alpha != 1.0f || dl.c_dtype != EGOMI_BF16 || (dl.N & 7)) return EGOMI_E_UNSUPPORTED;
    if (dl.ws_tickets_zeroed && dl.workspace) {
        if (dl.workspace_bytes > 4096) { dl.workspace = (char*)dl.workspace + 4096; dl.workspace_bytes -= 4096; }
        else { dl.workspace = nullptr; dl.workspace_bytes = 0; }
    }
    if (tall_form(&dl)) return EGOMI_OK;                               // one round of 352x256 tiles: no K-sliced rows
    const TailPlan tp = tail_plan_for(&dl);
    if (tp.rows) { *row0 = ((dl.M + 255) / 256 - tp.rows) * 256; *slices = tp.s; }
    return EGOMI_OK;
}

static int launch_8phase(const egomi_gemm_desc* d, FastArgs& g, hipStream_t s, int* tickets = nullptr, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr,
                         bool leave_slabs = false) {
    g.tiles_m = (d->M + 255) / 256; g.tiles_n = (d->N + 255) / 256;
    g.splitk = 1; g.ws = (float*)d->workspace;
    const TailPlan tp = tail_plan_for(d);
    g.full_tm = g.tiles_m - tp.rows; g.tail_s = tp.s; g.full_tiles = g.full_tm * g.tiles_n;
    // in-launch combine: needs the caller's ticket words (4 KB ahead of the slabs, include/egomi.h `ws_tickets_zeroed`) and room for
    // whole 256x256 slabs of every tail tile and slice
    g.tickets = nullptr;
    if (tp.rows && tickets && tp.rows * g.tiles_n <= 1024 && (long long)tp.rows * g.tiles_n * tp.s * 262144 <= d->workspace_bytes) g.tickets = tickets;
    const int nwg = g.full_tiles + tp.rows * g.tiles_n * tp.s;
    if (d->c_dtype != EGOMI_BF16 && d->c_dtype != EGOMI_F32) return EGOMI_E_UNSUPPORTED;
    if (t0) (void)hipEventRecord(t0, s);                                 // egomi_gemm_time_next: this kernel alone
    if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH(gemm_nt_bf16_8phase_kernel<bf16_t>, dim3(nwg, 1), dim3(512), 0, s, g);
    else EGOMI_LAUNCH(gemm_nt_bf16_8phase_kernel<float>, dim3(nwg, 1), dim3(512), 0, s, g);
    if (t1) (void)hipEventRecord(t1, s);
    if (tp.rows && g.epi == 1 && g.tickets) g.tickets = nullptr;         // the fused SwiGLU epilogue wants the separate combine + tail pass below
    if (leave_slabs) return egomi_launch_status();                       // EGOMI_EPI_SLABS: the caller's next kernel sums the tail rows' slabs
    if (tp.rows && g.epi == 3) {                                         // K-sliced tail rows of d(act): summed and turned into d(gate|up) in one pass
        FastArgs r = g;
        const long long row0 = (long long)g.full_tm * 256;
        r.M = d->M - (int)row0; r.splitk = tp.s;
        r.C = (char*)g.C + row0 * g.ldc * 2;
        r.C2 = (char*)g.C2 + row0 * g.ldc2 * 2;
        const long long total = (long long)r.M * (d->N / 8);
        const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        EGOMI_LAUNCH(splitk_reduce_swiglu_bwd_kernel, dim3(grid), dim3(256), 0, s, r);
        return egomi_launch_status();
    }
    if (tp.rows && !g.tickets) {
        FastArgs r = g;
        const long long row0 = (long long)g.full_tm * 256;
        const int esz = d->c_dtype == EGOMI_BF16 ? 2 : 4;
        r.M = d->M - (int)row0; r.splitk = tp.s;
        r.C = (char*)g.C + row0 * g.ldc * esz;
        if (g.residual) r.residual = (const char*)g.residual + row0 * g.ldr * esz;
        const long long total = (long long)r.M * ((d->N + 3) / 4);
        const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH(splitk_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, r);
        else EGOMI_LAUNCH(splitk_reduce_kernel<float>, dim3(grid), dim3(256), 0, s, r);
        if (g.epi == 1) {                                              // K-sliced tail rows: gate|up just combined, SwiGLU on those rows only
            const int rc = egomi_swiglu_il_fwd(r.C, (char*)g.C2 + row0 * g.ldc2 * 2, r.M, d->N / 2, g.ldc, g.ldc2, EGOMI_BF16, (egomi_stream_t)s);
            if (rc != EGOMI_OK) return rc;
        }
    }
    return egomi_launch_status();
}

// ---- column split: the 352x256 form on the first Na columns (whole rounds), the 256x256 form on the rest.  Products whose tiles fit neither
// form in whole rounds: gate|up at M = 5536 (N = 22016: 1892 tiles of 256x256 = 7.39 rounds; 16 x 86 tall tiles = 5.375) runs as 16 x 64 tall tiles
// (4 rounds) + 22 x 22 tiles of 256x256 (1.89 rounds), the down_proj data gradient (N = 11008: 3.70 / 2.69 rounds) as 16 x 32 tall (2 rounds) +
// 22 x 11 (0.95 rounds).  Same round model as tall_form(); both parts carry the product's epilogue (plain, SwiGLU, SwiGLU backward); the second
// part plans its own K-sliced tail.  Na is a multiple of 256 columns, so interleaved-32 gate|up groups and 64-column wave strips stay whole.
static double c256_model(long long tiles, bool ws) {
    const int ncu = p8_cus();                                            // CUs of this device (256 on MI355X in SPX mode)
    const int rem = (int)(tiles % ncu);
    return (double)(tiles / ncu) + (rem ? (rem * 2 <= ncu && ws ? 0.28 + 0.85 * rem / ncu : 1.0) : 0.0);
}
static int split_cols(const egomi_gemm_desc* d) {
    if (g_tall_mode < 0) { const char* e = getenv("EGOMI_GEMM_TALL"); g_tall_mode = e ? atoi(e) : 1; if (g_tall_mode < 0 || g_tall_mode > 2) g_tall_mode = 1; }
    static int on = -1;
    if (on < 0) { const char* e = getenv("EGOMI_GEMM_SPLIT"); on = e ? atoi(e) : 1; }
    if (g_tall_mode != 1 || !on) return 0;                             // (mode 2 = "the tall form wherever it applies" is the whole-product A/B arm)
    if (d->epilogue != EGOMI_EPI_NONE && d->epilogue != EGOMI_EPI_SWIGLU && d->epilogue != EGOMI_EPI_SWIGLU_BWD) return 0;
    if ((d->M & 7) || (d->N & 255) || d->M < TL_BM || d->K < 2048 || d->split_k > 0 || d->bias) return 0;
    if ((long long)d->M * d->lda >= (1ll << 31) || (long long)d->N * d->ldb >= (1ll << 31)) return 0;
    if (d->c_dtype != EGOMI_BF16 && d->c_dtype != EGOMI_F32) return 0;
    const int ncu = p8_cus();                                            // CUs of this device (256 on MI355X in SPX mode)
    const long long tn = d->N / 256, tm256 = (d->M + 255) / 256, tm352 = (d->M + TL_BM - 1) / TL_BM;
    const double c0 = c256_model(tm256 * tn, d->workspace != nullptr);
    double best = 0.97 * c0;
    int ja_best = 0;
    for (long long ja = 1; ja < tn; ++ja) {
        if ((tm352 * ja) % ncu) continue;                                // the tall part: whole rounds only
        const double c = 1.31 * (double)(tm352 * ja / ncu) + c256_model(tm256 * (tn - ja), d->workspace != nullptr) + 0.03;
        if (c < best) { best = c; ja_best = (int)ja; }
    }
    return ja_best * 256;
}
static int launch_split(const egomi_gemm_desc* d, const FastArgs& g, int Na, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    const int esz = d->c_dtype == EGOMI_BF16 ? 2 : 4;
    egomi_gemm_desc d1 = *d;
    d1.N = Na;
    FastArgs g1 = g;
    g1.N = Na;
    int rc = launch_tall(&d1, g1, s, t0, nullptr);
    if (rc != EGOMI_OK) return rc;
    egomi_gemm_desc d2 = *d;
    FastArgs g2 = g;
    d2.N = g2.N = d->N - Na;
    d2.B = g2.B = (const bf16_t*)d->B + (long long)Na * d->ldb;
    const long long ccol = d->epilogue == EGOMI_EPI_SWIGLU_BWD ? 2ll * Na : (long long)Na;       // C is d(gate|up) [M, 2N] there
    d2.C = g2.C = (char*)d->C + ccol * esz;
    if (d->residual) d2.residual = g2.residual = (const char*)d->residual + (long long)Na * esz;
    if (d->epilogue == EGOMI_EPI_SWIGLU) d2.C2 = g2.C2 = (char*)d->C2 + (long long)(Na / 2) * 2;   // act [M, N/2]
    if (d->epilogue == EGOMI_EPI_SWIGLU_BWD) d2.C2 = g2.C2 = (char*)d->C2 + 2ll * Na * 2;          // gate|up [M, 2N]
    return launch_8phase(&d2, g2, s, nullptr, nullptr, t1);
}

// returns 0 on success, <0 on error, 1 when the tuned kernel does not apply
// EGOMI_EPI_SLABS (include/egomi.h): plain product only — everything an epilogue could do is the consumer kernel's job
static bool slabs_form_ok(const egomi_gemm_desc* d) {
    return !d->bias && !d->residual && !d->accumulate && d->act == 0 && d->alpha == 1.0f && d->workspace && d->M <= 512 && (d->N & 3) == 0;
}

extern thread_local hipEvent_t egomi_time_start_, egomi_time_stop_;     // api.hip (egomi_gemm_time_next)

int egomi_gemm_fast_try(const egomi_gemm_desc* d0, hipStream_t s) {
    const hipEvent_t t0 = egomi_time_start_, t1 = egomi_time_stop_;       // one-shot request: consumed by this call whatever path it takes
    egomi_time_start_ = egomi_time_stop_ = nullptr;
    if (!fast_applicable(d0)) return 1;
    egomi_gemm_desc dl = *d0;
    const egomi_gemm_desc* d = &dl;
    FastArgs g;
    g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B; g.C = d->C; g.bias = (const bf16_t*)d->bias; g.residual = d->residual;
    g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.ldr = d->ldr;
    g.alpha = d->alpha; g.accumulate = d->accumulate; g.act = d->act; g.tickets = nullptr;
    g.epi = d->epilogue; g.C2 = d->C2; g.ldc2 = d->ldc2;
    const int tc = tile_choice(d);
    if (tc == 8 && (d->epilogue == EGOMI_EPI_NONE || d->epilogue == EGOMI_EPI_SLABS) && tall_form(d)) {
        if (d->epilogue == EGOMI_EPI_SLABS && (d->bias || d->accumulate || d->act != 0 || d->alpha != 1.0f || d->c_dtype != EGOMI_BF16)) return EGOMI_E_UNSUPPORTED;
        return launch_tall(d, g, s, t0, t1);
    }
    if (d->epilogue == EGOMI_EPI_SLABS && tc == 8) {
        // large products: only the K-sliced TAIL rows are left as slabs (egomi_gemm_tail_plan tells which); whole tiles get the
        // normal epilogue, residual included
        if (d->bias || d->accumulate || d->act != 0 || d->alpha != 1.0f || d->c_dtype != EGOMI_BF16 || (d->N & 7) ||
            (long long)d->M * d->lda >= (1ll << 31) || (long long)d->N * d->ldb >= (1ll << 31)) return EGOMI_E_UNSUPPORTED;
        if (dl.ws_tickets_zeroed && dl.workspace) {
            if (dl.workspace_bytes > 4096) { dl.workspace = (char*)dl.workspace + 4096; dl.workspace_bytes -= 4096; }
            else { dl.workspace = nullptr; dl.workspace_bytes = 0; }
        }
        g.epi = 0;
        return launch_8phase(d, g, s, nullptr, t0, t1, true);
    }
    if (d->epilogue == EGOMI_EPI_SLABS) {
        if (!slabs_form_ok(d) || tc != 1) return EGOMI_E_UNSUPPORTED;
        if (dl.ws_tickets_zeroed && dl.workspace) {                     // same scratch convention as below: slabs start behind the ticket words
            if (dl.workspace_bytes > 4096) { dl.workspace = (char*)dl.workspace + 4096; dl.workspace_bytes -= 4096; }
            else return EGOMI_E_UNSUPPORTED;
        }
        g.epi = 0;
        return launch_fast<128, 128>(d, g, s);
    }
    if (d->epilogue != EGOMI_EPI_NONE) {
        // fused epilogues live in the 256x256 per-tile kernel's plain-bf16 store path only: whole interleaved groups per wave
        // (N % 256 == 0), 16-B aligned rows, nothing else in the epilogue
        const bool common = tc == 8 && d->c_dtype == EGOMI_BF16 && d->C2 && !d->bias && !d->residual &&
                            !d->accumulate && d->act == 0 && d->alpha == 1.0f && d->ldc % 8 == 0 && d->ldc2 % 8 == 0 &&
                            (((uintptr_t)d->C | (uintptr_t)d->C2) & 15) == 0 && (long long)d->M * d->lda < (1ll << 31) && (long long)d->N * d->ldb < (1ll << 31);
        const bool ok = common && ((d->epilogue == EGOMI_EPI_SWIGLU && d->N % 256 == 0 && d->ldc2 >= d->N / 2) ||
                                   (d->epilogue == EGOMI_EPI_SWIGLU_BWD && d->N % 64 == 0 && d->ldc >= 2 * d->N && d->ldc2 >= 2 * d->N));
        if (!ok) return EGOMI_E_UNSUPPORTED;
        if (dl.ws_tickets_zeroed && dl.workspace) {
            if (dl.workspace_bytes > 4096) { dl.workspace = (char*)dl.workspace + 4096; dl.workspace_bytes -= 4096; }
            else { dl.workspace = nullptr; dl.workspace_bytes = 0; }
        }
        if (tall_form(d)) return launch_tall(d, g, s, t0, t1);              // the fused epilogues live in both forms
        if (const int Na = split_cols(d)) return launch_split(d, g, Na, s, t0, t1);
        return launch_8phase(d, g, s, nullptr, t0, t1);
    }
    if (tc == 8 && (long long)d->M * d->lda < (1ll << 31) && (long long)d->N * d->ldb < (1ll << 31)) {
        P8Sched sc;
        if (p8_applicable(d, sc)) return launch_p8(d, g, sc, s);
    }
    int* tickets = nullptr;
    if (dl.ws_tickets_zeroed && dl.workspace) {                       // the ticket words are used by the in-launch combines only: every other
        if (dl.workspace_bytes > 4096) { tickets = (int*)dl.workspace; dl.workspace = (char*)dl.workspace + 4096; dl.workspace_bytes -= 4096; }   // scratch starts behind them
        else { dl.workspace = nullptr; dl.workspace_bytes = 0; }
    }
    // In-launch combine of the K-sliced tail tiles (last-arriver tickets) is built and tested but NOT the default.  Measured in the
    // training step on one box, arms alternated: with agent-scope release/acquire fences around the slabs 139.1 ms/step vs 136.6
    // with the separate combine launch; with sc1 slab stores/loads and no fence (the present form) 136.4 vs 136.15 — the fences
    // (whole-L2 write-back / invalidate in mid-launch) were the 2.3 ms, what remains is the last arriver summing S x 256 KB
    // alone at the very end of the launch, where splitk_reduce_kernel spreads the same bytes over every CU.  EGOMI_GEMM_FOLD=1
    // selects it.
    static int fold = -1;
    if (fold < 0) { const char* e = getenv("EGOMI_GEMM_FOLD"); fold = e ? atoi(e) : 0; }
    if (!fold && dl.ws_tickets_zeroed != 2) tickets = nullptr;
    if (tc == 8 && (long long)d->M * d->lda < (1ll << 31) && (long long)d->N * d->ldb < (1ll << 31)) {
        if (const int Na = split_cols(d)) return launch_split(d, g, Na, s, t0, t1);
        return launch_8phase(d, g, s, tickets, t0, t1);
    }
    return tc == 2 ? launch_fast<256, 128>(d, g, s) : launch_fast<128, 128>(d, g, s);
}
