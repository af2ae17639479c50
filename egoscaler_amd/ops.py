"""Thin torch<->C-ABI marshalling: tensors in, tensors out, raw device pointers underneath.
PyTorch is used for device memory and streams only; every op here runs a kernel of libegomi.so.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import c_p, c_i, c_d, c_f, c_sz, c_i64, call

F32, BF16 = 0, 1


def dt(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t}")


def P(t):
    if t is None:
        return c_p(None)
    if not t.is_cuda:
        raise _lib.EgomiError("egoscaler_amd ops need CUDA/HIP tensors (no CPU fallback)")
    return c_p(t.data_ptr())


def S():
    return c_p(torch.cuda.current_stream().cuda_stream)


def _c(t, dtype=None):
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------ A1/A2
def unproject_gather(rgb, depth, pp, fx, fy, d_thres=None, boxes=None, n_out=0):
    """rgb u8 [B,T,H,W,3], depth f32 [B,T,H,W] -> (points f64 [B,cap,3], colors f32 [B,cap,3],
    count i32 [B]).  See include/egomi.h (A1)."""
    rgb, depth = _c(rgb, torch.uint8), _c(depth, torch.float32)
    B, T, H, W, _ = rgb.shape
    cap = n_out if n_out > 0 else T * H * W
    dev = rgb.device
    pts = torch.empty(B, cap, 3, dtype=torch.float64, device=dev)
    col = torch.empty(B, cap, 3, dtype=torch.float32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    ws_bytes = _lib.lib().egomi_unproject_workspace_bytes(B, T, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    bx, nb = None, 0
    if boxes is not None and len(boxes):
        bx = torch.tensor([[b["ymin"], b["ymax"], b["xmin"], b["xmax"]] for b in boxes], dtype=torch.int32, device=dev)
        nb = bx.shape[0]
    call("egomi_unproject_gather", P(rgb), P(depth), P(bx), c_i(nb), c_i(B), c_i(T), c_i(H), c_i(W),
         c_d(pp), c_d(fx), c_d(fy), c_f(float("nan") if d_thres is None else d_thres), c_i(n_out),
         P(pts), P(col), P(cnt), P(ws), c_sz(ws_bytes), S())
    return pts, col, cnt


def depth_to_cloud(pred, rgb, final_width, final_height, focal_len_x=0, focal_len_y=0, principal_point=0):
    """N4 (depth.py:35-62): pred f32 [B,h0,w0] or [h0,w0], rgb u8 [B,H,W,3] or [H,W,3] ->
    (z f32 [.., H, W], points f64 [.., H*W, 3] | None, colors f64 [.., H*W, 3] | None)."""
    single = pred.dim() == 2
    pred = _c(pred if not single else pred[None], torch.float32)
    B, h0, w0 = pred.shape
    H, W = int(final_height), int(final_width)
    dev = pred.device
    want = focal_len_x > 0 and focal_len_y > 0 and principal_point > 0            # depth.py:53
    z = torch.empty(B, H, W, dtype=torch.float32, device=dev)
    tab = torch.empty(W + H, dtype=torch.int32, device=dev)
    pts = col = None
    if want:
        rgb = _c(rgb if rgb.dim() == 4 else rgb[None], torch.uint8)
        if tuple(rgb.shape) != (B, H, W, 3):
            raise ValueError(f"depth_to_cloud: rgb must be [{B},{H},{W},3], got {tuple(rgb.shape)}")
        pts = torch.empty(B, H * W, 3, dtype=torch.float64, device=dev)
        col = torch.empty(B, H * W, 3, dtype=torch.float64, device=dev)
    call("egomi_depth_to_cloud", P(pred), c_i(B), c_i(h0), c_i(w0), P(rgb) if want else None, c_i(H), c_i(W),
         c_d(float(focal_len_x)), c_d(float(focal_len_y)), c_d(float(principal_point)), P(tab), P(z), P(pts), P(col), S())
    if single:
        return z[0], (pts[0] if want else None), (col[0] if want else None)
    return z, pts, col


def pc_norm(points, colors):
    points, colors = _c(points, torch.float64), _c(colors, torch.float32)
    B, N, _ = points.shape
    out = torch.empty(B, N, 6, dtype=torch.float32, device=points.device)
    call("egomi_pc_norm", P(points), P(colors), P(out), c_i(B), c_i(N), S())
    return out


# ------------------------------------------------------------------------------------------ A3-A5
def fps(pts, start, num_group):
    pts = _c(pts, torch.float32)
    B, N, C = pts.shape
    if torch.is_tensor(start) and start.is_cuda and start.dtype == torch.int32 and start.numel() == B:
        start = start.contiguous()          # resident start vector: validated by its producer (no host sync here)
    else:
        host = torch.as_tensor(start).reshape(-1).to(torch.int64).cpu()
        if host.numel() != B or int(host.min()) < 0 or int(host.max()) >= N:
            raise ValueError("fps: start must hold one index in [0,N) per cloud")
        start = host.to(torch.int32).to(pts.device)
    idx = torch.empty(B, num_group, dtype=torch.int32, device=pts.device)
    cen = torch.empty(B, num_group, 3, dtype=torch.float32, device=pts.device)
    call("egomi_fps", P(pts), c_i(B), c_i(N), c_i(C), P(start), c_i(num_group), P(idx), P(cen), S())
    return idx, cen


def knn_group(pts, center, k, out_dtype=torch.float32):
    pts, center = _c(pts, torch.float32), _c(center, torch.float32)
    B, N, C = pts.shape
    G = center.shape[1]
    idx = torch.empty(B, G, k, dtype=torch.int32, device=pts.device)
    nb = torch.empty(B, G, k, C, dtype=out_dtype, device=pts.device)
    call("egomi_knn_group", P(pts), P(center), c_i(B), c_i(N), c_i(C), c_i(G), c_i(k), P(idx), P(nb), c_i(dt(out_dtype)), S())
    return idx, nb


# ------------------------------------------------------------------------------------------ GEMM
class GemmDesc(ctypes.Structure):
    _fields_ = [("A", c_p), ("B", c_p), ("C", c_p), ("bias", c_p), ("residual", c_p),
                ("M", c_i), ("N", c_i), ("K", c_i),
                ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64), ("ldr", c_i64),
                ("a_layout", c_i), ("b_layout", c_i), ("ab_dtype", c_i), ("c_dtype", c_i),
                ("batch", c_i), ("batch_inner", c_i),
                ("sA0", c_i64), ("sA1", c_i64), ("sB0", c_i64), ("sB1", c_i64), ("sC0", c_i64), ("sC1", c_i64),
                ("alpha", c_f), ("accumulate", c_i), ("act", c_i), ("force_generic", c_i),
                ("workspace", c_p), ("workspace_bytes", c_i64), ("split_k", c_i), ("ws_tickets_zeroed", c_i),
                ("epilogue", c_i), ("C2", c_p), ("ldc2", c_i64)]


ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2


class GemmProfiler:
    """HIP-event timing of GEMM launches on the stream they are launched on (bench.py roofline)."""

    def __init__(self, min_flops=1e9, tuned_only=True, kernel_ids=None):
        """kernel_ids: egomi_gemm_kernel_id() values to record (None: 1 and 2 when tuned_only, else everything)."""
        self.min_flops, self.recs, self.enabled, self.tuned_only = min_flops, [], True, tuned_only
        self.kernel_ids = kernel_ids if kernel_ids is not None else ((1, 2) if tuned_only else None)

    def summary(self):
        """launches / flops / ms: the recorded brackets.  For launches of the 256x256 kernel (kernel id 2) `ms` is that kernel alone
        (events recorded by the library around it: egomi_gemm_time_next) and `call_ms` the whole egomi_gemm call, i.e. including
        the slab-combine pass of K-sliced tail rows; for other kernels both are the call."""
        torch.cuda.synchronize()
        n, flops, ms, call_ms = 0, 0.0, 0.0, 0.0
        L = _lib.lib()
        for e0, e1, f, k0, k1 in self.recs:
            n += 1
            flops += f
            c = e0.elapsed_time(e1)
            call_ms += c
            if k0 is not None:
                t = ctypes.c_float()
                ms += t.value if L.egomi_event_elapsed_ms(k0, k1, ctypes.byref(t)) == 0 else c     # never recorded (opt-in persistent form): the call
            else:
                ms += c
        return {"launches": n, "flops": flops, "ms": ms, "call_ms": call_ms}

    def __del__(self):
        try:
            L = _lib.lib()
            for _, _, _, k0, k1 in self.recs:
                if k0 is not None:
                    L.egomi_event_destroy(k0)
                    L.egomi_event_destroy(k1)
        except Exception:
            pass


PROFILER = None


_TAIL_WS = {}


def _tail_workspace(device, nbytes=(4096 + 256 * 2 * 262144)):
    """One fp32 scratch buffer per device, shared by every large product (launches are stream-ordered): 4 KB of ticket words
    (zero here, returned to zero by every launch: include/egomi.h `ws_tickets_zeroed`) + the fp32 slabs of shared tiles."""
    ws = _TAIL_WS.get(device)
    if ws is None:
        ws = _TAIL_WS[device] = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
    return ws


def gemm_raw(A, B, C, M, N, K, lda, ldb, ldc, a_layout=0, b_layout=0, bias=None, residual=None, ldr=0,
             act=0, alpha=1.0, accumulate=False, batch=1, batch_inner=1, strides=(0, 0, 0, 0, 0, 0), force_generic=False,
             workspace=None, split_k=0, persistent=None, swiglu_out=None, slabs=False, count_only=False, defer_tail=False, swiglu_bwd_gu=None):
    """C = act(alpha*A.B + bias) + residual (+C).  A/B/C are tensors whose data_ptr() is the first
    element of the (first) operand; all strides in elements.  See include/egomi.h."""
    if A.dtype != B.dtype:
        raise TypeError("gemm: A and B dtypes differ")
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    if bias is not None and bias.dtype != A.dtype:
        raise TypeError("gemm: bias dtype must equal the operand dtype")
    if residual is not None and residual.dtype != C.dtype:
        raise TypeError("gemm: residual dtype must equal the output dtype")
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.a_layout, d.b_layout = a_layout, b_layout
    d.ab_dtype, d.c_dtype = dt(A.dtype), dt(C.dtype)
    d.batch, d.batch_inner = batch, batch_inner
    d.sA0, d.sA1, d.sB0, d.sB1, d.sC0, d.sC1 = strides
    d.alpha, d.accumulate, d.act, d.force_generic = alpha, int(accumulate), act, int(force_generic)
    if swiglu_out is not None:                            # EGOMI_EPI_SWIGLU: C = interleaved-32 gate|up, swiglu_out [M, N/2] = silu(gate)*up
        d.epilogue, d.C2, d.ldc2 = 1, swiglu_out.data_ptr(), _ld(swiglu_out)
    if swiglu_bwd_gu is not None:                         # EGOMI_EPI_SWIGLU_BWD: the product is d(act); C = d(gate|up) [M, 2N], C2 = gate|up [M, 2N] (interleaved-32)
        if swiglu_out is not None or swiglu_bwd_gu.dtype != C.dtype:
            raise TypeError("gemm: swiglu_bwd_gu excludes swiglu_out and must have the output dtype")
        d.epilogue, d.C2, d.ldc2 = 3, swiglu_bwd_gu.data_ptr(), _ld(swiglu_bwd_gu)
    if workspace is None and M >= 1024 and batch <= 1:
        workspace = _tail_workspace(A.device)            # ticket words + fp32 slabs for the shared tiles of the 256x256 kernel
        if persistent is False:                          # A/B runs, tests: the non-persistent kernel + combine launch; its slabs must
            workspace = workspace[1024:]                 # stay clear of the ticket words the persistent launches rely on
        else:
            d.ws_tickets_zeroed = 2 if persistent else 1  # True: wherever the persistent form can run; None: the library's measured rule
    if workspace is not None:
        d.workspace, d.workspace_bytes, d.split_k = workspace.data_ptr(), workspace.numel() * workspace.element_size(), split_k
    for t in (A, B, C):
        if not t.is_cuda:
            raise _lib.EgomiError("gemm needs device tensors")
    if slabs or count_only:                               # EGOMI_EPI_SLABS: K-slice slabs stay unsummed in `workspace`; -> their number
        d.epilogue = 2
        n = _lib.lib().egomi_gemm_slab_count(ctypes.byref(d))
        if count_only:
            return n
        if n < 2:
            raise _lib.EgomiError("gemm: the library would not split this product (egomi_gemm_slab_count == 0)")
        call("egomi_gemm", ctypes.byref(d), S())
        return n
    tail = None
    if defer_tail and d.workspace and bias is None and act == 0 and alpha == 1.0 and not accumulate and swiglu_out is None and \
            C.dtype == torch.bfloat16:
        # EGOMI_EPI_SLABS on a large product: the K-sliced tail rows stay as fp32 slabs for the caller's next kernel (ops.rmsnorm /
        # ops.rmsnorm_bwd with tail=...); nothing changes when the library's plan has no such rows for this shape
        row0, slices = c_i(0), c_i(0)
        if _lib.lib().egomi_gemm_tail_plan(ctypes.byref(d), ctypes.byref(row0), ctypes.byref(slices)) == 0 and slices.value >= 2:
            d.epilogue = 2
            if _lib.lib().egomi_gemm_kernel_id(ctypes.byref(d)) == 2:    # the launch must take the kernel the plan was made for
                tail = (row0.value, slices.value, d.workspace + (4096 if d.ws_tickets_zeroed else 0))
            else:
                d.epilogue = 0
    prof = PROFILER
    flops = 2.0 * M * N * K * max(1, batch)
    kid = _lib.lib().egomi_gemm_kernel_id(ctypes.byref(d)) if (prof is not None and prof.enabled and flops >= prof.min_flops) else None
    if kid is not None and (prof.kernel_ids is None or kid in prof.kernel_ids):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0 = k1 = None
        if kid == 2:                                        # the 256x256 kernel: the library brackets the kernel itself
            L = _lib.lib()
            k0, k1 = c_p(), c_p()
            _lib.check(L.egomi_event_create(ctypes.byref(k0)), "egomi_event_create")
            _lib.check(L.egomi_event_create(ctypes.byref(k1)), "egomi_event_create")
            _lib.check(L.egomi_gemm_time_next(k0, k1), "egomi_gemm_time_next")
        e0.record()
        call("egomi_gemm", ctypes.byref(d), S())
        e1.record()
        prof.recs.append((e0, e1, flops, k0, k1))
    else:
        call("egomi_gemm", ctypes.byref(d), S())
    return (C, tail) if defer_tail else C


def _ld(t):
    if t.dim() != 2 or (t.shape[1] != 1 and t.stride(1) != 1):
        raise ValueError("expected a 2-D view with unit inner stride")
    return max(t.stride(0), t.shape[1]) if t.shape[0] > 1 else t.shape[1]


def mm(a, b, out=None, a_layout=0, b_layout=0, out_dtype=None, **kw):
    """2-D product on views with unit inner stride.  a: [M,K] (layout 0) or [K,M] (1);
    b: [N,K] (layout 0, nn.Linear weight) or [K,N] (1)."""
    M, K = (a.shape if a_layout == 0 else (a.shape[1], a.shape[0]))
    N, Kb = (b.shape if b_layout == 0 else (b.shape[1], b.shape[0]))
    if K != Kb:
        raise ValueError(f"mm: inner dims differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or a.dtype, device=a.device)
    res = kw.get("residual")
    return gemm_raw(a, b, out, M, N, K, _ld(a), _ld(b), _ld(out), a_layout, b_layout,
                    ldr=_ld(res) if res is not None else 0, **kw)


def mm_slabs(a, b, out_like, workspace, count_only=False):
    """a [M,K] . b[N,K]^T with M <= 512 as UNSUMMED fp32 K-slice slabs [slices, M, N] at the start of `workspace` (include/egomi.h,
    EGOMI_EPI_SLABS); returns the number of slices (count_only: without launching; 0 = the library would not split it).
    `out_like` [M,N] only lends its dtype / alignment to the descriptor, it is not written."""
    M, K = a.shape
    N = b.shape[0]
    return gemm_raw(a, b, out_like, M, N, K, _ld(a), _ld(b), _ld(out_like), workspace=workspace, slabs=not count_only, count_only=count_only)


def slabs_rmsnorm(workspace, slices, residual, w, eps, x_out, h_out):
    """x_out = round(sum of the slabs + residual), h_out = rmsnorm(x_out) * w  (egomi_slabs_rmsnorm)."""
    rows, cols = x_out.shape
    call("egomi_slabs_rmsnorm", P(workspace), c_i(slices), c_i(rows), c_i(cols), P(residual), c_i64(_ld(residual) if residual is not None else 0), P(w),
         c_f(eps), P(x_out), c_i64(_ld(x_out)), P(h_out), c_i64(_ld(h_out)), c_i(dt(x_out.dtype)), S())


def qkv_finish(workspace, slices, qkv, cos, sin, pos, kc, vc, B, H, hd, Smax):
    """q|k|v = round(sum of the slabs); RoPE(pos) on q and k; q -> qkv, k / v -> the caches at `pos`  (egomi_qkv_finish)."""
    call("egomi_qkv_finish", P(workspace), c_i(slices), P(qkv), c_i64(_ld(qkv)), P(cos), P(sin), c_i(pos), P(kc), P(vc), c_i(B), c_i(H), c_i(hd),
         c_i(Smax), c_i(dt(qkv.dtype)), S())


# ------------------------------------------------------------------------------------------ rows
def layernorm(x, w, b, eps=1e-5, add=None, sum_out=None, out=None):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    call("egomi_layernorm_fwd", P(x), P(add), P(w), P(b), P(sum_out), P(out), c_i(rows), c_i(cols), c_f(eps), c_i(dt(x.dtype)), S())
    return out


def rmsnorm(x, w, eps, rstd=None, out=None, tail=None, tail_residual=None):
    """tail = (row0, slices, slab address) from mm(..., defer_tail=True): rows >= row0 of x are formed here from the product's
    K-slice slabs (+ tail_residual's rows), written to x and normalised in one pass (egomi_rmsnorm_fwd_tail)."""
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    if tail is not None:
        call("egomi_rmsnorm_fwd_tail", P(x), P(w), P(out), P(rstd), c_i(rows), c_i(cols), c_f(eps), c_i(tail[0]), c_p(tail[2]), c_i(tail[1]),
             P(tail_residual), c_i64(_ld(tail_residual) if tail_residual is not None else 0), c_i(dt(x.dtype)), S())
        return out
    call("egomi_rmsnorm_fwd", P(x), P(w), P(out), P(rstd), c_i(rows), c_i(cols), c_f(eps), c_i(dt(x.dtype)), S())
    return out


def rmsnorm_bwd(dy, x, w, rstd, dx_add=None, dw=None, out=None, tail=None):
    """tail: as in rmsnorm() — dy's rows >= row0 are still the slabs of the dgrad product that made dy (egomi_rmsnorm_bwd_tail)."""
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    if tail is not None:
        call("egomi_rmsnorm_bwd_tail", P(dy), P(x), P(w), P(rstd), P(out), P(dx_add), P(dw), c_i(rows), c_i(cols), c_i(tail[0]), c_p(tail[2]), c_i(tail[1]),
             c_i(dt(x.dtype)), S())
        return out
    call("egomi_rmsnorm_bwd", P(dy), P(x), P(w), P(rstd), P(out), P(dx_add), P(dw), c_i(rows), c_i(cols), c_i(dt(x.dtype)), S())
    return out


def rope_tables(seq, head_dim, theta):
    """fp32 cos/sin [S, hd/2], computed exactly as HF LlamaRotaryEmbedding does (modeling_llama.py:93-127)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = torch.arange(seq, dtype=torch.float32)[:, None] * inv[None, :]
    return freqs.cos().contiguous(), freqs.sin().contiguous()


def rope_(x, cos, sin, rows, seq, pos_offset, H, hd, ld, inverse=False):
    call("egomi_rope", P(x), P(cos), P(sin), c_i64(rows), c_i(seq), c_i(pos_offset), c_i(H), c_i(hd), c_i64(ld), c_i(int(inverse)), c_i(dt(x.dtype)), S())
    return x


def rope_qkv_tail_(qkv, cos, sin, rows, seq, pos_offset, H, hd, ld, tail):
    """rope_ on the q and k heads of a stacked q|k|v array (H = heads of ONE of them) whose rows >= tail[0] are still the K-slice
    slabs of the product (mm(..., defer_tail=True)): summed, rounded, rotated / materialised in this pass (egomi_rope_qkv_tail)."""
    call("egomi_rope_qkv_tail", P(qkv), P(cos), P(sin), c_i64(rows), c_i(seq), c_i(pos_offset), c_i(H), c_i(hd), c_i64(ld), c_i(tail[0]), c_p(tail[2]),
         c_i(tail[1]), c_i(dt(qkv.dtype)), S())
    return qkv


def swiglu(gate, up, out):
    rows, cols = gate.shape
    call("egomi_swiglu_fwd", P(gate), P(up), P(out), c_i64(rows), c_i(cols), c_i64(_ld(gate)), c_i64(_ld(out)), c_i(dt(gate.dtype)), S())
    return out


def swiglu_il(gu, out):
    """Interleaved-32 gate|up array gu [rows, 2*cols] -> out [rows, cols] = silu(gate)*up."""
    rows, cols = out.shape
    call("egomi_swiglu_il_fwd", P(gu), P(out), c_i64(rows), c_i(cols), c_i64(_ld(gu)), c_i64(_ld(out)), c_i(dt(gu.dtype)), S())
    return out


def swiglu_il_bwd(dact, gu, dgu):
    rows, cols = dact.shape
    call("egomi_swiglu_il_bwd", P(dact), P(gu), P(dgu), c_i64(rows), c_i(cols), c_i64(_ld(gu)), c_i64(_ld(dact)), c_i64(_ld(dgu)), c_i(dt(gu.dtype)), S())
    return dgu


def gemm_kernel_id(M, N, K, dtype=torch.bfloat16, out_dtype=torch.bfloat16):
    """Which kernel egomi_gemm would pick for a plain contiguous NT product of this shape (2 = 256x256 8-phase)."""
    d = GemmDesc()
    d.A = d.B = d.C = 256                                 # aligned placeholders: the selection looks at shapes and alignment only
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, K, K, N
    d.ab_dtype, d.c_dtype, d.batch = dt(dtype), dt(out_dtype), 1
    return _lib.lib().egomi_gemm_kernel_id(ctypes.byref(d))


def mm_kernel_id(a, b, out, a_layout=0, b_layout=0, accumulate=False):
    """Which kernel egomi_gemm would run for mm(a, b, out=out, a_layout=..., b_layout=...): 0 generic, 1 / 2 the K-contiguous tuned kernels,
    3 the k-major 8-phase kernel of csrc/gemm_tn.hip (weight gradients, data gradients against an un-transposed weight)."""
    M, K = (a.shape if a_layout == 0 else (a.shape[1], a.shape[0]))
    N = b.shape[0] if b_layout == 0 else b.shape[1]
    d = GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, _ld(a), _ld(b), _ld(out)
    d.a_layout, d.b_layout, d.ab_dtype, d.c_dtype, d.batch = a_layout, b_layout, dt(a.dtype), dt(out.dtype), 1
    d.alpha, d.accumulate = 1.0, int(accumulate)
    return _lib.lib().egomi_gemm_kernel_id(ctypes.byref(d))


def swiglu_bwd(dact, gate, up, dgate, dup):
    rows, cols = gate.shape
    call("egomi_swiglu_bwd", P(dact), P(gate), P(up), P(dgate), P(dup), c_i64(rows), c_i(cols), c_i64(_ld(gate)), c_i64(_ld(dact)),
         c_i64(_ld(dgate)), c_i(dt(gate.dtype)), S())


def gelu(x, out=None):
    out = torch.empty_like(x) if out is None else out
    call("egomi_gelu_fwd", P(x), P(out), c_i64(x.numel()), c_i(dt(x.dtype)), S())
    return out


def gelu_bwd(dy, x, out=None):
    out = torch.empty_like(x) if out is None else out
    call("egomi_gelu_bwd", P(dy), P(x), P(out), c_i64(x.numel()), c_i(dt(x.dtype)), S())
    return out


def softmax(scores, Z, heads, Sq, Sk, out, causal=False, q_offset=0, key_mask=None):
    call("egomi_softmax_fwd", P(scores), c_i64(Sk), P(key_mask), c_i(Z), c_i(heads), c_i(Sq), c_i(Sk), c_i(int(causal)), c_i(q_offset),
         P(out), c_i64(Sk), c_i(dt(out.dtype)), S())
    return out


def softmax_bwd(Pm, dP, dS, rows, Sk):
    call("egomi_softmax_bwd", P(Pm), c_i64(Sk), P(dP), c_i64(Sk), P(dS), c_i64(Sk), c_i64(rows), c_i(Sk), c_i(dt(Pm.dtype)), S())
    return dS


def splice_scan(ids, tok, Pn, n_clouds=None):
    """-> (start_pos, err, cloud_idx), int32 [B] each (include/egomi.h: the reference's position checks and its running cloud index)."""
    B, Sl = ids.shape
    sp, err, cloud, scratch = (torch.empty(B, dtype=torch.int32, device=ids.device) for _ in range(4))
    call("egomi_splice_scan", P(ids), c_i(B), c_i(Sl), c_i64(tok.point_patch), c_i64(tok.point_start), c_i64(tok.point_end), c_i(Pn),
         c_i(B if n_clouds is None else n_clouds), P(sp), P(err), P(cloud), P(scratch), S())
    return sp, err, cloud


def embed_splice(ids, W, feats, start_pos, Pn, out=None, cloud_idx=None):
    B, Sl = ids.shape
    V, d = W.shape
    out = torch.empty(B, Sl, d, dtype=W.dtype, device=W.device) if out is None else out
    call("egomi_embed_splice_fwd", P(ids), P(W), P(feats), P(start_pos), P(cloud_idx), c_i(B), c_i(Sl), c_i(d), c_i(Pn), c_i(V), P(out), c_i(dt(W.dtype)), S())
    return out


def embed_splice_bwd(dout, ids, start_pos, Pn, V, dW=None, dfeats=None, cloud_idx=None):
    B, Sl, d = dout.shape
    call("egomi_embed_splice_bwd", P(dout), P(ids), P(start_pos), P(cloud_idx), c_i(B), c_i(Sl), c_i(d), c_i(Pn), c_i(V), P(dW), P(dfeats),
         c_i(dt(dout.dtype)), S())


def cross_entropy(logits, targets, ignore_index, dlogits=None, grad_scale=1.0):
    """Returns (loss_sum fp32 [1], count i32 [1]); mean loss = loss_sum / count."""
    R, V = logits.shape
    cnt = torch.zeros(1, dtype=torch.int32, device=logits.device)
    ls = torch.zeros(1, dtype=torch.float32, device=logits.device)
    rows = torch.empty(R, dtype=torch.float32, device=logits.device)
    call("egomi_ce_count", P(targets), c_i64(R), c_i64(ignore_index), P(cnt), S())
    call("egomi_ce_fwd_bwd", P(logits), c_i64(_ld(logits)), P(targets), c_i(R), c_i(V), c_i64(ignore_index), P(cnt), P(ls), P(rows),
         P(dlogits), c_i64(_ld(dlogits) if dlogits is not None else 0), c_f(grad_scale), c_i(dt(logits.dtype)), S())
    return ls, cnt


def adamw(master, model_copy, grad, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    if grad.dtype not in (torch.float32, torch.bfloat16) or not grad.is_contiguous() or grad.numel() != master.numel():
        raise _lib.EgomiError("adamw: the gradient must be a contiguous fp32 or bf16 tensor of the parameter's size")
    call("egomi_adamw" if grad.dtype == torch.float32 else "egomi_adamw_g16", P(master), P(model_copy), P(grad), P(m), P(v), c_i64(master.numel()), c_f(lr), c_f(beta1), c_f(beta2), c_f(eps),
         c_f(wd), c_i(step), c_f(grad_scale), c_i(dt(model_copy.dtype) if model_copy is not None else F32), S())


def transpose(x, ldo=None, out=None):
    R, C = x.shape
    ldo = R if ldo is None else ldo
    out = torch.empty(C, ldo, dtype=x.dtype, device=x.device) if out is None else out
    call("egomi_transpose", P(x), c_i(R), c_i(C), c_i64(_ld(x)), P(out), c_i64(ldo), c_i(dt(x.dtype)), S())
    return out


def cast(x, dtype, out=None):
    out = torch.empty(x.shape, dtype=dtype, device=x.device) if out is None else out
    call("egomi_cast", P(x), c_i(dt(x.dtype)), P(out), c_i(dt(dtype)), c_i64(x.numel()), S())
    return out


def rank_sum(chunks, out):
    """chunks [W, c] (bf16 / fp32, contiguous) -> out [c] (bf16 / fp32): fp32-accumulated sum over the leading (rank) axis."""
    W, c = chunks.shape
    call("egomi_rank_sum", P(chunks), c_i(dt(chunks.dtype)), c_i(W), c_i64(c), P(out), c_i(dt(out.dtype)), S())
    return out


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    call("egomi_add", P(a), P(b), P(out), c_i64(a.numel()), c_i(dt(a.dtype)), S())
    return out


def colsum_(x, out):
    """out (fp32 [C]) += column sums of x [R,C]."""
    R, C = x.shape
    call("egomi_colsum", P(x), c_i64(R), c_i(C), c_i64(_ld(x)), P(out), c_i(dt(x.dtype)), S())
    return out


def group_max(x, BG, M, C, concat=False, out=None):
    if out is None:
        out = torch.empty((BG * M, 2 * C) if concat else (BG, C), dtype=x.dtype, device=x.device)
    call("egomi_group_max", P(x), c_i(BG), c_i(M), c_i(C), P(out), c_i(int(concat)), c_i(dt(x.dtype)), S())
    return out


def linear_smallk(x, w, b, act=0, out=None):
    R, K = x.numel() // x.shape[-1], x.shape[-1]
    N = w.shape[0]
    out = torch.empty(R, N, dtype=w.dtype, device=w.device) if out is None else out
    call("egomi_linear_smallk", P(x), c_i(dt(x.dtype)), P(w), P(b), P(out), c_i64(R), c_i(N), c_i(K), c_i(act), c_i(dt(w.dtype)), S())
    return out


# ------------------------------------------------------------------------------------------ fused attention
class AttnDesc(ctypes.Structure):
    _fields_ = [("q", c_p), ("k", c_p), ("v", c_p), ("o", c_p), ("lse", c_p),
                ("dout", c_p), ("delta", c_p), ("dq", c_p), ("dk", c_p), ("dv", c_p), ("key_mask", c_p),
                ("B", c_i), ("H", c_i), ("S", c_i), ("head_dim", c_i),
                ("ld_qkv", c_i64), ("ld_o", c_i64), ("ld_dqkv", c_i64),
                ("scale", c_f), ("causal", c_i), ("dtype", c_i), ("rope_cos", c_p), ("rope_sin", c_p)]


def _attn_desc(qkv, B, Sq, H, hd, scale, causal, key_mask):
    d = AttnDesc()
    dm = H * hd
    d.q, d.k, d.v = qkv[:, :dm].data_ptr(), qkv[:, dm:2 * dm].data_ptr(), qkv[:, 2 * dm:].data_ptr()
    d.key_mask = key_mask.data_ptr() if key_mask is not None else None
    d.B, d.H, d.S, d.head_dim = B, H, Sq, hd
    d.ld_qkv = qkv.stride(0)
    d.scale, d.causal, d.dtype = scale, int(causal), dt(qkv.dtype)
    return d


def attn_fwd(qkv, B, Sq, H, hd, scale, out, lse, causal=True, key_mask=None):
    """qkv [B*S, 3*H*hd] (q|k|v column blocks) -> out [B*S, H*hd], lse fp32 [B,H,S]."""
    d = _attn_desc(qkv, B, Sq, H, hd, scale, causal, key_mask)
    d.o, d.lse, d.ld_o = out.data_ptr(), lse.data_ptr() if lse is not None else None, out.stride(0)
    call("egomi_attn_fwd", ctypes.byref(d), S())
    return out


def attn_bwd(qkv, out, lse, dout, dqkv, delta, B, Sq, H, hd, scale, causal=True, key_mask=None, rope=None):
    """dq|dk|dv written into the column blocks of dqkv [B*S, 3*H*hd]; delta fp32 [B,H,S] is scratch.
    rope=(cos, sin) fp32 [>=S, hd/2]: dq and dk come out rotated back (== rope_(dqkv, inverse=True) afterwards)."""
    d = _attn_desc(qkv, B, Sq, H, hd, scale, causal, key_mask)
    if rope is not None:
        cos, sin = rope
        if cos.dtype != torch.float32 or sin.dtype != torch.float32 or cos.shape[0] < Sq or cos.shape[1] != hd // 2 \
                or not cos.is_contiguous() or not sin.is_contiguous():
            raise ValueError("attn_bwd: rope tables must be contiguous fp32 [>=S, head_dim/2]")
        d.rope_cos, d.rope_sin = cos.data_ptr(), sin.data_ptr()
    dm = H * hd
    d.o, d.lse, d.ld_o = out.data_ptr(), lse.data_ptr(), out.stride(0)
    if dout.stride(0) != out.stride(0):
        raise ValueError("attn_bwd: dout and out must share their row stride")
    d.dout, d.delta = dout.data_ptr(), delta.data_ptr()
    d.dq, d.dk, d.dv = dqkv[:, :dm].data_ptr(), dqkv[:, dm:2 * dm].data_ptr(), dqkv[:, 2 * dm:].data_ptr()
    d.ld_dqkv = dqkv.stride(0)
    call("egomi_attn_bwd", ctypes.byref(d), S())
    return dqkv
