// Tuned bf16 "NT" GEMM for the LLaMA-sized projections:  C[M,N] = A[M,K] . B[N,K]^T  (+ epilogue)
//   * both operands K-contiguous (activation x nn.Linear weight; dgrad uses pre-transposed weights,
//     wgrad uses transposed activations, so every big product of the path has this form)
//   * BM x BN x 64 tile, one wave per 64x64 sub-tile (v_mfma_f32_16x16x32_bf16, fp32 accumulate):
//       256x128 (8 waves, 48 KB LDS, 2 blocks/CU) for large M — rocprofv3 PMC showed the 128x128 tile
//       parked ~49 % of wave time on vmcnt/barrier with zero LDS bank conflicts, i.e. bound by the
//       per-CU LDS-DMA fill rate (~70 GB/s/CU from L2); the wider tile moves 25 % fewer bytes per flop
//       128x128 (4 waves, 32 KB LDS, 4 blocks/CU) otherwise
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write
//   * LDS image linear [row][64 k] (128-B rows); bank conflicts removed by an XOR swizzle applied to
//     the per-lane SOURCE address and to the fragment read (chunk ^= (row>>1)&7): the 16 rows of a
//     fragment land on 16 distinct 16-B slots of the 256-B bank row (cdna_hip_programming.md rule 21);
//     measured SQ_LDS_BANK_CONFLICT = 0
//   * operands are fed swapped (weights as the MFMA A operand) so each lane's 4 accumulators are 4
//     consecutive output columns -> 8-B / 16-B epilogue stores
//   * rows beyond M / N are clamped on load and masked on store; K % 64 == 0
//   * XCD-aware block order: each XCD walks a contiguous strip of tiles, grouped 8 M-tiles deep (T1)
#include "common.h"

#define FT_BK 64

struct FastArgs {
    const bf16_t* A; const bf16_t* B; void* C; const bf16_t* bias; const void* residual;
    int M, N, K;
    long long lda, ldb, ldc, ldr;
    float alpha; int accumulate; int act;
    int tiles_m, tiles_n;
    int splitk; float* ws;           // splitk > 1: block (tile, blockIdx.y) multiplies its K slice and stores a raw fp32 slab
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ void glds16(const bf16_t* g, bf16_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}

template <typename TC, int BM, int BN, int DB, int MT = 4>
__global__ __launch_bounds__((BM / (16 * MT)) * (BN / 64) * 64, DB ? 2 : ((BM * BN == 256 * 128) ? 4 : 3))
void gemm_nt_bf16_kernel(FastArgs g) {
    constexpr int WN = BN / 64, NW = (BM / (16 * MT)) * WN;
    constexpr int A_PW = BM / 8 / NW, B_PW = BN / 8 / NW;               // 1-KiB DMA pieces (8 rows x 128 B) per wave
    constexpr int STAGE = (BM + BN) * FT_BK;
    __shared__ __attribute__((aligned(16))) bf16_t smem[(DB ? 2 : 1) * STAGE];   // one array (cdna guide: second-__shared__ trap)
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
    if (DB) {
        // two LDS stages: the DMA of tile t+1 is in flight while tile t is multiplied (counted vmcnt + raw
        // s_barrier, cdna_hip_programming.md "Pipelining across barriers")
#pragma unroll
        for (int i = 0; i < A_PW; ++i) glds16(srcA[i] + t_begin * FT_BK, sA + (t_begin & 1) * STAGE + (wave * A_PW + i) * 8 * FT_BK);
#pragma unroll
        for (int i = 0; i < B_PW; ++i) glds16(srcB[i] + t_begin * FT_BK, sB + (t_begin & 1) * STAGE + (wave * B_PW + i) * 8 * FT_BK);
    }
    for (int t = t_begin; t < nt; ++t) {
        const int k0 = t * FT_BK;
        const bf16_t* cA = sA;
        const bf16_t* cB = sB;
        if (DB) {
            cA = sA + (t & 1) * STAGE;
            cB = sB + (t & 1) * STAGE;
            if (t + 1 < nt) {
                bf16_t* nA = sA + ((t + 1) & 1) * STAGE;
                bf16_t* nB = sB + ((t + 1) & 1) * STAGE;
#pragma unroll
                for (int i = 0; i < A_PW; ++i) glds16(srcA[i] + k0 + FT_BK, nA + (wave * A_PW + i) * 8 * FT_BK);
#pragma unroll
                for (int i = 0; i < B_PW; ++i) glds16(srcB[i] + k0 + FT_BK, nB + (wave * B_PW + i) * 8 * FT_BK);
                if (A_PW + B_PW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
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
        if (DB) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                           // stage (t&1) may be refilled by the DMA of tile t+2
        } else {
            __syncthreads();                                        // tile consumed before it is overwritten
        }
    }

    // ---- epilogue.  acc[j][i][r] = C[m][n], n = n0+wn+16j+4*(lane>>4)+r, m = m0+wm+16i+(lane&15)
    if (g.splitk > 1) {
        float* slab = g.ws + (long long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn + j * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + wm + i * 16 + (lane & 15);
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
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn + j * 16 + (lane >> 4) * 4;
        if (n >= g.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < g.N) bv[r] = bf2f(g.bias[n + r]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm + i * 16 + (lane & 15);
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

// tile choice: 2 = 256x128, 1 = 128x128.  EGOMI_GEMM_TILE=1|2 overrides (A/B experiments).
static int tile_choice(const egomi_gemm_desc* d) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("EGOMI_GEMM_TILE"); forced = e ? atoi(e) : 0; }
    if (forced >= 1 && forced <= 5) return forced;
    // measured (tools/gemm_bench.py, M=5536): 256x128 wins only where N is wide enough to keep every CU at
    // 2 resident blocks to the end (N=11008: 1168 vs 1084 TFLOP/s); at N=4096 its 704 tiles quantise worse
    // than 1408 tiles of 128x128 (952 vs 1010)
    return (d->M >= 2048 && d->N >= 8192) ? 2 : 1;
}

extern "C" int egomi_gemm_kernel_id(const egomi_gemm_desc* d) {
    if (!d) return EGOMI_E_BADARG;
    return (!d->force_generic && fast_applicable(d)) ? 1 : 0;
}

template <int BM, int BN, int DB, int MT = 4>
static int launch_fast(const egomi_gemm_desc* d, FastArgs& g, hipStream_t s) {
    g.tiles_m = (d->M + BM - 1) / BM; g.tiles_n = (d->N + BN - 1) / BN;
    const int nwg = g.tiles_m * g.tiles_n;
    constexpr int threads = (BM / (16 * MT)) * (BN / 64) * 64;
    // skinny products (decode, M <= 512): too few tiles to fill 256 CUs and each block is DMA-latency bound,
    // so the K range is split over blockIdx.y into fp32 slabs (caller-provided workspace) and combined
    g.splitk = 1; g.ws = (float*)d->workspace;
    const int nt = d->K / FT_BK;
    if (d->workspace && d->M <= 512 && nwg < 512) {
        // measured (tools/gemm_bench_decode.py, M=256): ~256 blocks, and no more than ~32 K-steps per slice
        int sk = d->split_k > 0 ? d->split_k : (nwg >= 256 ? 1 : (256 + nwg - 1) / nwg);
        if (d->split_k <= 0 && sk > 1 && nt / sk > 32) sk *= 2;
        if (sk > nt) sk = nt;
        const long long per_slab = (long long)d->M * d->N * 4;
        if ((long long)sk * per_slab > d->workspace_bytes) sk = (int)(d->workspace_bytes / per_slab);
        if (sk > 1) { const int per = (nt + sk - 1) / sk; sk = (nt + per - 1) / per; }     // no empty slices
        if (sk > 1) g.splitk = sk;
    }
    if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH((gemm_nt_bf16_kernel<bf16_t, BM, BN, DB, MT>), dim3(nwg, g.splitk), dim3(threads), 0, s, g);
    else if (d->c_dtype == EGOMI_F32) EGOMI_LAUNCH((gemm_nt_bf16_kernel<float, BM, BN, DB, MT>), dim3(nwg, g.splitk), dim3(threads), 0, s, g);
    else return EGOMI_E_UNSUPPORTED;
    if (g.splitk > 1) {
        const long long total = (long long)d->M * ((d->N + 3) / 4);
        const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (d->c_dtype == EGOMI_BF16) EGOMI_LAUNCH(splitk_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, g);
        else EGOMI_LAUNCH(splitk_reduce_kernel<float>, dim3(grid), dim3(256), 0, s, g);
    }
    return egomi_launch_status();
}

// returns 0 on success, <0 on error, 1 when the tuned kernel does not apply
int egomi_gemm_fast_try(const egomi_gemm_desc* d, hipStream_t s) {
    if (!fast_applicable(d)) return 1;
    FastArgs g;
    g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B; g.C = d->C; g.bias = (const bf16_t*)d->bias; g.residual = d->residual;
    g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.ldr = d->ldr;
    g.alpha = d->alpha; g.accumulate = d->accumulate; g.act = d->act;
    const int tc = tile_choice(d);
    if (tc == 3) return launch_fast<128, 128, 1>(d, g, s);          // experiment: double-buffered 128x128
    if (tc == 4) return launch_fast<256, 256, 1, 8>(d, g, s);       // experiment: 256x256, 8 waves of 128x64, double-buffered (128 KB LDS)
    if (tc == 5) return launch_fast<256, 64, 0>(d, g, s);           // skinny M (decode): all 256 rows x 64 columns per block, split-K
    return tc == 2 ? launch_fast<256, 128, 0>(d, g, s) : launch_fast<128, 128, 0>(d, g, s);
}
