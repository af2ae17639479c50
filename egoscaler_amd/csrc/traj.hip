// A14 on the device: trajectory <-> token ids for whole batches, and displacement metrics.
// replaces (integer contracts, bit-exact):
//   models/pointllm/utils/utils.py:13-16  discretize_action   np.digitize(v, linspace(-1,1,nb)) - 1
//   models/pointllm/utils/utils.py:18-21  token_to_action     linspace[idx]
//   models/pointllm/utils/utils.py:47-104 str_to_float (rt2, 6-DoF): split on <tsep>, first run of six
//       <p*> tokens per segment, unmatched segments repeat the previous step
//   models/pointllm/dataset.py:16-19,150-194 sequence layout  <ts> (p*6 <tsep>)*T <te> eos pad...
//   models/utils/metrics.py:7-55          ADE / FDE (documented [T,D] form)
// The bin edges are computed once on the host in float64 exactly as numpy.linspace does and passed in.
#include "common.h"
#include <math.h>

// ids[b, :] = <ts> (p p p p p p <tsep>) x steps[b] <te> <eos> pad...   ; mask = 1 on non-pad
__global__ __launch_bounds__(256) void traj_tokenize_kernel(const float* traj, const int32_t* steps, int Tmax, const double* bins, int nb,
                                                            int64_t p0, int64_t ts, int64_t tsep, int64_t te, int64_t eos, int64_t pad,
                                                            int L, int64_t* ids, uint8_t* mask, int32_t* err) {
    const int b = blockIdx.x;
    int T = steps ? steps[b] : Tmax;
    T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
    const int need = 1 + 7 * T + 2;
    if (need > L) { if (threadIdx.x == 0) err[b] = 1; T = (L - 3) / 7; }
    else if (threadIdx.x == 0) err[b] = 0;
    const int real = 1 + 7 * T + 2;
    for (int j = threadIdx.x; j < L; j += 256) {
        int64_t t;
        if (j == 0) t = ts;
        else if (j < 1 + 7 * T) {
            const int s = (j - 1) / 7, c = (j - 1) % 7;
            if (c == 6) t = tsep;
            else {
                const double v = (double)traj[((long long)b * Tmax + s) * 6 + c];
                // np.digitize(v, bins) = number of edges <= v  (bins increasing, right=False); NaN -> nb
                int lo = 0, hi = nb;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (bins[mid] <= v) lo = mid + 1; else hi = mid; }
                int bin = (v != v) ? nb : lo;
                bin -= 1;                                                   // utils.py:15
                bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);          // keep the id inside <p0..p{nb-1}>
                t = p0 + bin;
            }
        } else if (j == 1 + 7 * T) t = te;
        else if (j == 2 + 7 * T) t = eos;
        else t = pad;
        ids[(long long)b * L + j] = t;
        mask[(long long)b * L + j] = j < real;
    }
}

// one thread per sample: scan ids, emit values[b, step, 6] (bin centres linspace[idx]) and n_steps[b]
__global__ __launch_bounds__(64) void traj_detokenize_kernel(const int64_t* ids, int B, int L, const double* bins, int nb, int64_t p0, int64_t tsep,
                                                             int64_t eos, int Tmax, float* out, int32_t* n_steps) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int64_t* r = ids + (long long)b * L;
    int n = 0, end = L;
    for (int j = 0; j < L; ++j) if (r[j] == eos) { end = j; break; }       // train.py:241-242: cut at the first eos
    int seg0 = 0;
    bool have_last = false;
    float last[6];
    while (seg0 <= end && n < Tmax) {
        int seg1 = seg0;
        while (seg1 < end && r[seg1] != tsep) ++seg1;                       // segment [seg0, seg1)
        // first run of six consecutive <p*> ids inside the segment (the regex of utils.py:52-57)
        int run = 0, hit = -1;
        for (int j = seg0; j < seg1; ++j) {
            if (r[j] >= p0 && r[j] < p0 + nb) { if (++run == 6) { hit = j - 5; break; } } else run = 0;
        }
        if (hit >= 0) {
            for (int c = 0; c < 6; ++c) last[c] = (float)bins[(int)(r[hit + c] - p0)];
            have_last = true;
        }
        if (hit >= 0 || have_last) {                                        // utils.py:88-90 copy-forward
            for (int c = 0; c < 6; ++c) out[((long long)b * Tmax + n) * 6 + c] = last[c];
            ++n;
        }
        if (seg1 >= end) break;
        seg0 = seg1 + 1;
    }
    n_steps[b] = n;
}

// per sample: gen padded with its last step / cut to len_gt (metrics.py:40-52), then
// ade = mean_t ||gt_t - gen_t||_2, fde = ||gt_last - gen_last||_2 over all D dims (float64 like numpy)
__global__ __launch_bounds__(64) void traj_metrics_kernel(const float* gen, const int32_t* n_gen, const float* gt, const int32_t* n_gt, int B, int Tmax,
                                                          int D, double* ade, double* fde) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int ng = n_gen ? n_gen[b] : Tmax, nt = n_gt ? n_gt[b] : Tmax;
    if (ng <= 0 || nt <= 0) { ade[b] = NAN; fde[b] = NAN; return; }
    double acc = 0.0, lastd = 0.0;
    for (int t = 0; t < nt; ++t) {
        const int tg = t < ng ? t : ng - 1;
        double s = 0.0;
        for (int c = 0; c < D; ++c) {
            const double df = (double)gt[((long long)b * Tmax + t) * D + c] - (double)gen[((long long)b * Tmax + tg) * D + c];
            s += df * df;
        }
        lastd = sqrt(s);
        acc += lastd;
    }
    ade[b] = acc / nt;
    fde[b] = lastd;
}

extern "C" int egomi_traj_tokenize(const float* traj, const int32_t* steps, int B, int Tmax, const double* bins, int num_bins, int64_t p0,
                                   int64_t ts, int64_t tsep, int64_t te, int64_t eos, int64_t pad, int L, int64_t* ids, uint8_t* mask,
                                   int32_t* err, egomi_stream_t stream) {
    if (!traj || !bins || !ids || !mask || !err) return EGOMI_E_BADARG;
    if (B <= 0 || Tmax <= 0 || num_bins <= 1 || L < 3) return EGOMI_E_SHAPE;
    EGOMI_LAUNCH(traj_tokenize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, traj, steps, Tmax, bins, num_bins, p0, ts, tsep, te, eos, pad, L, ids, mask, err);
    return egomi_launch_status();
}

extern "C" int egomi_traj_detokenize(const int64_t* ids, int B, int L, const double* bins, int num_bins, int64_t p0, int64_t tsep, int64_t eos,
                                     int Tmax, float* out, int32_t* n_steps, egomi_stream_t stream) {
    if (!ids || !bins || !out || !n_steps) return EGOMI_E_BADARG;
    if (B <= 0 || L <= 0 || Tmax <= 0 || num_bins <= 1) return EGOMI_E_SHAPE;
    EGOMI_LAUNCH(traj_detokenize_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, ids, B, L, bins, num_bins, p0, tsep, eos, Tmax, out, n_steps);
    return egomi_launch_status();
}

extern "C" int egomi_traj_metrics(const float* gen, const int32_t* n_gen, const float* gt, const int32_t* n_gt, int B, int Tmax, int D, double* ade,
                                  double* fde, egomi_stream_t stream) {
    if (!gen || !gt || !ade || !fde) return EGOMI_E_BADARG;
    if (B <= 0 || Tmax <= 0 || D <= 0) return EGOMI_E_SHAPE;
    EGOMI_LAUNCH(traj_metrics_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, gen, n_gen, gt, n_gt, B, Tmax, D, ade, fde);
    return egomi_launch_status();
}
