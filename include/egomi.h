/* egomi.h — C-ABI of libegomi.so: the MI355X (gfx950) kernels under EgoScaler's trajectory-generator
 * hot path.
 *
 * The reference has NO native/FFI boundary on this path: its boundary is the
 * Python nn.Module API of egoscaler/models/pointllm/{builder.py,model_arch.py}, which
 * egoscaler_amd/ mirrors.  This header is the boundary UNDER that API: every entry point names the
 * reference function (file:line, relative to /root/reference/egoscaler/) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked host; caller owns all memory
 *   - no entry point allocates, synchronises or throws; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream)
 *   - return 0 (EGOMI_OK) or a negative EGOMI_E_* code; shapes are validated on the host before
 *     any launch, so a bad call never reaches the GPU
 *   - dtype arguments: EGOMI_F32 / EGOMI_BF16 (raw uint16 storage)
 *   - thread-safe for distinct streams
 */
#ifndef EGOMI_H
#define EGOMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGOMI_OK 0
#define EGOMI_E_BADARG (-1)
#define EGOMI_E_SHAPE (-2)
#define EGOMI_E_LAUNCH (-3)
#define EGOMI_E_UNSUPPORTED (-4)

#define EGOMI_F32 0
#define EGOMI_BF16 1

typedef void* egomi_stream_t;

int egomi_version(void);
const char* egomi_strerror(int code);

/* ------------------------------------------------------------------------------------------------
 * A1  RGB-D un-projection + ordered compaction (+ optional strided subsample)
 *     replaces  data/tools/pcm_tools.py:68-96  get_points_colors
 *     (frame loop / rgbd concat: vis/interactive.py:22-32, data/train/7_get_object_trajectory.py:244-253)
 *
 * rgb   u8  [B,T,H,W,3]      depth f32 [B,T,H,W]
 * boxes i32 [n_boxes,4] = (ymin,ymax,xmin,xmax) pixels masked out in every frame (may be NULL)
 * x = (u - pp)/fx * z,  y = (v - pp)/fy * z  in float64 exactly as numpy evaluates it;
 * colour = float32(c)/255.0f; valid = all(rgb != 0) & outside boxes & (z < d_thres).
 * d_thres = NaN disables the depth test (d_thres=None in the reference).
 * Output order is the index contract: frame-major, then row-major pixel order, valid pixels only.
 *   n_out == 0 : write every valid pixel  -> out_points f64 [B,cap,3], out_colors f32 [B,cap,3],
 *                cap = T*H*W rows reserved per sample; out_count[b] = n_valid
 *   n_out  > 0 : first-N strided subsample (stride = floor(n_valid / n_out), rows j*stride),
 *                cap = n_out; a sample with n_valid < n_out sets out_count[b] = -n_valid and its
 *                rows are left untouched (the host mirror raises ValueError)
 */
size_t egomi_unproject_workspace_bytes(int B, int T, int H, int W);
int egomi_unproject_gather(const uint8_t* rgb, const float* depth, const int32_t* boxes, int n_boxes,
                           int B, int T, int H, int W, double pp, double fx, double fy, float d_thres,
                           int n_out, double* out_points, float* out_colors, int32_t* out_count,
                           void* workspace, size_t workspace_bytes, egomi_stream_t stream);

/* A2  pc_norm: centre xyz on the centroid, divide by the largest radius, in float64; colours pass
 *     through.   replaces  models/pointllm/pointllm/data/utils.py:146-157
 * points f64 [B,N,3], colors f32 [B,N,3] -> out f32 [B,N,6] */
int egomi_pc_norm(const double* points, const float* colors, float* out, int B, int N, egomi_stream_t stream);

/* A3  farthest point sampling.   replaces  models/pointllm/pointllm/model/pointbert/misc.py:40-60
 * pts f32 [B,N,C] (xyz = first 3 channels, C>=3), start i32 [B] (the reference draws it from the
 * global torch RNG, misc.py:52) -> out_idx i32 [B,G], out_center f32 [B,G,3].
 * fp32, d = (dx*dx+dy*dy)+dz*dz, running min from 1e10, arg-max with the LOWEST index on ties:
 * indices are bit-exact with the reference.  N <= 16384. */
int egomi_fps(const float* pts, int B, int N, int C, const int32_t* start, int G,
              int32_t* out_idx, float* out_center, egomi_stream_t stream);

/* A4+A5  kNN grouping.   replaces  pointbert/dvae.py:107-140 (square_distance, knn_point) and
 *        dvae.py:150-187 (Group.forward gather, centre subtraction on xyz only)
 * pts f32 [B,N,C], center f32 [B,G,3] -> out_idx i32 [B,G,K] ordered by (distance, index)
 * ascending, out_nb [B,G,K,C] (dtype out_dtype).  dist = ((-2*dot)+|c|^2)+|p|^2 in fp32 with
 * dot = (cx*px+cy*py)+cz*pz; no [G,N] matrix touches HBM.  N <= 8192, K <= 64. */
int egomi_knn_group(const float* pts, const float* center, int B, int N, int C, int G, int K,
                    int32_t* out_idx, void* out_nb, int out_dtype, egomi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EGOMI_H */
