"""ctypes binding of include/hidvae.h (the C-ABI drop-in boundary).

Every wrapper passes raw device pointers + sizes + the current HIP stream, checks the int return code and
raises RuntimeError(hidvae_last_error()).  There is NO fallback: if libhidvae_hip.so is missing or a tensor
is not a contiguous fp32 CUDA(HIP) tensor, the call fails loudly."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhidvae_hip.so")
_lib = None

MODE_GUMBEL, MODE_STE, MODE_ROTATION = 1, 2, 3
DIST_L2, DIST_COSINE = 0, 1
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_NONE, EPI_SILU, EPI_RELU, EPI_GELU, EPI_SIGMOID = 0, 1, 2, 3, 4
EPI_DSILU, EPI_DRELU, EPI_DGELU, EPI_DSIGMOID = 16, 17, 18, 19
MAX_LEVELS = 8
EMBED_DIM = 32

_vp, _i, _i64, _f, _u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32

_SIGNATURES = {
    "hidvae_gemm_f32": [_i, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i, _vp, _i64, _vp, _i64, _f, _vp, _u32, _u32, _i, _vp, _i, _vp],
    "hidvae_linear_bwd": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _f, _vp, _i, _vp, _i, _vp],
    "hidvae_layernorm_bwd_partial": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "hidvae_layernorm_param_final_many": [_vp, _i, _vp],
    "hidvae_predictor_fwd": [_vp, _i64, _i64, _vp, _i, _vp, _vp],
    "hidvae_predictor_bwd": [_vp, _i64, _i64, _vp, _i, _vp, _i64, _vp],
    "hidvae_gate_fwd": [_vp, _i64, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "hidvae_gate_bwd": [_vp, _i64, _vp, _i64, _i64, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "hidvae_colsum": [_vp, _i64, _i64, _i64, _vp, _i, _vp, _vp],
    "hidvae_codebook_prepare": [_vp, _vp, _i, _i64, _vp, _vp, _i, _vp],
    "hidvae_rq_forward": [_vp, _i64, _i, _vp, _vp, _i, _i64, _i, _i, _f, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "hidvae_bottleneck_fwd": [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i64, _i, _f, _vp, _vp, _vp, _i64, _vp, _vp,
                              _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "hidvae_rq_backward": [_vp, _vp, _i64, _i, _vp, _vp, _i, _i64, _i, _f, _vp, _vp, _i64, _vp, _vp, _i64, _f, _vp, _i64, _vp, _vp, _i, _vp],
    "hidvae_rq_backward_slices": [_vp, _vp, _i64, _i, _vp, _vp, _i, _i64, _i, _f, _vp, _vp, _vp, _i, _vp, _vp, _i64, _f, _vp, _i64, _vp, _vp, _i, _vp],
    "hidvae_uniq_loss": [_vp, _vp, _i64, _i, _f, _f, _vp, _vp, _i, _vp],
    "hidvae_total_loss": [_vp, _vp, _i64, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "hidvae_total_loss_bwd": [_vp, _i64, _i, _f, _f, _f, _vp, _vp, _vp, _i, _vp],
    "hidvae_codebook_grad": [_vp, _vp, _i64, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "hidvae_l2norm_fwd_pair": [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _f, _vp],
    "hidvae_l2norm_bwd_pair": [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _f, _vp],
    "hidvae_recon_fwd_bwd": [_vp, _vp, _i64, _i64, _i, _f, _vp, _i64, _vp, _vp, _vp, _vp],
    "hidvae_l2norm_fwd": [_vp, _i64, _i64, _i64, _f, _vp, _i64, _vp, _vp],
    "hidvae_l2norm32_fwd": [_vp, _i64, _i64, _f, _vp, _i64, _vp, _vp],
    "hidvae_l2norm_bwd": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _f, _vp, _i64, _i, _vp],
    "hidvae_id_stats": [_vp, _i64, _vp, _i64, _i, _vp, _vp, _vp, _i, _vp],
    "hidvae_act_bwd": [_vp, _vp, _i64, _i, _vp, _f, _vp, _vp],
    "hidvae_binary": [_i, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp],
    "hidvae_sum_prefix_slices": [_vp, _vp, _i, _i64, _i64, _vp, _vp],
    "hidvae_layernorm_fwd": [_vp, _i64, _i64, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp, _f, _vp, _vp, _u32, _u32, _vp],
    "hidvae_rng_advance": [_vp, _vp],
    "hidvae_dropout_mask": [_vp, _i64, _vp, _u32, _u32, _vp],
    "hidvae_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _vp, _f, _vp, _vp],
    "hidvae_layernorm_param_grad": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _vp, _f, _vp, _vp, _i, _vp, _vp],
    "hidvae_layernorm_bwd_all": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _vp, _f, _vp, _vp, _vp, _i, _vp, _vp],
    "hidvae_batchnorm_fwd": [_vp, _i64, _i64, _i64, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, _u32, _u32, _vp, _vp],
    "hidvae_batchnorm_bwd": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i, _vp, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "hidvae_infonce_rows": [_vp, _i64, _f, _f, _vp, _vp, _vp],
    "hidvae_infonce_dlogits": [_vp, _i64, _f, _f, _vp, _vp],
    "hidvae_infonce_lse_chunk": [_vp, _i64, _i64, _i64, _i64, _f, _vp, _vp, _vp, _i, _vp],
    "hidvae_infonce_lse_finish": [_vp, _vp, _vp, _i64, _f, _f, _vp, _vp, _vp, _vp],
    "hidvae_infonce_dlogits_chunk": [_vp, _i64, _i64, _i64, _i64, _f, _f, _vp, _vp, _vp],
    "hidvae_mixup_plan": [_vp, _i64, _i, _i64, _vp, _f, _vp, _vp, _vp, _vp, _vp],
    "hidvae_tag_loss_fwd": [_vp, _i64, _i64, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "hidvae_tag_loss_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp],
    "hidvae_kmeans_iter": [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "hidvae_gumbel_rows_fwd": [_vp, _vp, _vp, _vp, _i64, _i64, _f, _vp, _i, _i, _vp],
    "hidvae_gumbel_loss": [_vp, _vp, _i64, _f, _vp, _i, _vp],
    "hidvae_gumbel_gemb": [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _i, _vp],
    "hidvae_gumbel_rows_bwd": [_vp, _vp, _i64, _i64, _f, _vp, _vp],
    "hidvae_gumbel_finish": [_vp, _vp, _vp, _vp, _vp, _i64, _f, _i64, _vp, _vp, _vp, _i64, _i, _vp],
    "hidvae_cat_recon_rows": [_vp, _i64, _vp, _i64, _i64, _i64, _i, _vp, _i64, _vp, _vp, _vp],
    "hidvae_loss_fwd": [_vp, _vp, _i64, _i64, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "hidvae_loss_bwd": [_vp, _vp, _vp, _i64, _i64, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _i, _f, _vp],
    "hidvae_padded_to_jagged": [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp],
    "hidvae_jagged_to_padded": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp],
    "hidvae_gather_rows": [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp],
    "hidvae_codebook_prepare_adamw": [_vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _i64, _i64, _f, _vp, _i, _vp],
    "hidvae_adamw_prepare": [_vp, _vp, _vp, _i, _f, _f, _f, _i64, _i64, _f, _vp, _vp],
    "hidvae_adamw_step": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _f, _f, _f, _f, _vp],
    "hidvae_query_workspace": [_i, _vp, _i, _vp],
    "hidvae_sqdiff_rows": [_vp, _i64, _vp, _i64, _i64, _i64, _f, _vp, _vp],
    "hidvae_sqdiff_rows_bwd": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _f, _f, _vp, _vp, _vp],
    "hidvae_gumbel_noise": [_vp, _i64, _f, _vp, _vp],
    "hidvae_gumbel_softmax_rows": [_vp, _vp, _i64, _i64, _f, _vp, _vp],
    "hidvae_softmax_argmax_rows": [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp],
    "hidvae_timestamp": [_vp, _vp],
    "hidvae_linear_bwd_group": [_vp, _i, _vp],
}
WS_GEMM, WS_LINEAR_BWD, WS_COLSUM, WS_CODEBOOK_GRAD, WS_LAYERNORM_PARAM_GRAD, WS_LAYERNORM_BWD_ALL = 1, 2, 3, 4, 5, 6
WS_BATCHNORM_FWD, WS_BATCHNORM_BWD, WS_ID_CENSUS, WS_KMEANS, WS_TAG_LOSS, WS_LINEAR_BWD_ZEROED, WS_RQ_FORWARD = 7, 8, 9, 10, 11, 12, 13


class LaunchStamps:
    """In-step launch timeline: while active, EVERY C-ABI launch is bracketed by two hidvae_timestamp launches on its
    own stream, also inside a HIP-graph capture (HIP events cannot be recorded there on ROCm).  After the step has run / the graph
    has been replayed, rows() gives (entry point, microseconds between the two stamps).  A bracket spans stamp kernel + dependent-
    launch gap + the launch itself + another gap; `empty_us` (a bracket around nothing, taken on every begin) measures the first
    two, so rows() reports bracket - empty = the launch's duration plus ONE dependent-launch gap, i.e. what the launch costs the
    step.  Measurement aid only: the stamps add two tiny launches per launch and are never part of a timed region."""
    TICK_US = 0.01  # wall_clock64 runs at 100 MHz on gfx950

    def __init__(self, device, capacity=4096):
        self.buf = torch.zeros((2 * capacity,), dtype=torch.int64, device=device)
        self.names, self.dims = [], []  # dims: (M, N, K, flops) of GEMM-class launches, else None
        self.streams = []               # the stream handle every launch went to (tools/step_gantt.py draws one lane per stream)

    def _slot(self, name, dims=None, stream=None):
        if 2 * len(self.names) + 2 > self.buf.numel():
            raise RuntimeError("LaunchStamps: capacity exceeded")
        self.names.append(name)
        self.dims.append(dims)
        self.streams.append(int(getattr(stream, "value", stream) or 0))
        return self.buf.data_ptr() + 16 * (len(self.names) - 1)

    def rows(self):
        t = self.buf[: 2 * len(self.names)].cpu().reshape(-1, 2)
        d = ((t[:, 1] - t[:, 0]).double() * self.TICK_US).tolist()
        empty = [v for n, v in zip(self.names, d) if n == "(empty)"]
        base = sorted(empty)[len(empty) // 2] if empty else 0.0
        return [(n, v - base) for n, v in zip(self.names, d) if n != "(empty)"], base

    def spans(self):
        """[(entry, stream handle, start us, end us)] on the device's wall clock, relative to the first stamp (absolute positions:
        which launches of which stream ran beside each other, where a stream sat idle)"""
        t = self.buf[: 2 * len(self.names)].cpu().reshape(-1, 2).double() * self.TICK_US
        t0 = float(t[:, 0].min()) if len(self.names) else 0.0
        return [(n, s, float(a) - t0, float(b) - t0) for n, s, (a, b) in zip(self.names, self.streams, t.tolist()) if n != "(empty)"]

    def rows_with_dims(self):
        rows, base = self.rows()
        dims = [dm for n, dm in zip(self.names, self.dims) if n != "(empty)"]
        return [(n, v, dm) for (n, v), dm in zip(rows, dims)], base


class PhaseMarks:
    """A handful of single device timestamps at phase boundaries of the step (each one hidvae_timestamp launch on the stream that is
    current where phase_mark() is called), also inside a graph capture.  Unlike LaunchStamps -- two extra launches around EVERY launch,
    which stretches the replay and re-times the streams against each other -- ~30 marks leave the step's structure as it is, so
    rows() shows where the un-instrumented step really is at each boundary (tools/step_phases.py)."""

    def __init__(self, device, capacity=512):
        self.buf = torch.zeros((capacity,), dtype=torch.int64, device=device)
        self.labels = []

    def rows(self):
        t = self.buf[: len(self.labels)].cpu().double() * LaunchStamps.TICK_US
        t0 = float(t.min()) if len(self.labels) else 0.0
        return [(lab, st, float(v) - t0) for (lab, st), v in zip(self.labels, t.tolist())]


_PHASES = None


def phases_begin(device, capacity=512):
    global _PHASES
    _PHASES = PhaseMarks(device, capacity)
    return _PHASES


def phases_end():
    global _PHASES
    p, _PHASES = _PHASES, None
    return p


def phase_mark(label):
    """no-op unless phases_begin() is active: one device timestamp on the current stream, remembered under `label`"""
    p = _PHASES
    if p is None:
        return
    i = len(p.labels)
    if i >= p.buf.numel():
        raise RuntimeError("PhaseMarks: capacity exceeded")
    st = torch.cuda.current_stream().cuda_stream
    p.labels.append((label, int(st)))
    lib()._L.hidvae_timestamp(ctypes.c_void_p(p.buf.data_ptr() + 8 * i), ctypes.c_void_p(st))


_STAMPS = None
_NOT_LAUNCHES = ("hidvae_last_error", "hidvae_version", "hidvae_query_workspace", "hidvae_timestamp")


def stamps_begin(device, capacity=4096):
    global _STAMPS
    _STAMPS = LaunchStamps(device, capacity)
    return _STAMPS


def stamps_end():
    global _STAMPS
    st, _STAMPS = _STAMPS, None
    return st


def stamp_empty_bracket():
    """a bracket around nothing on the current stream (calibration of LaunchStamps.rows)"""
    st = _STAMPS
    if st is not None:
        p = st._slot("(empty)")
        lib()._L.hidvae_timestamp(ctypes.c_void_p(p), _stream())
        lib()._L.hidvae_timestamp(ctypes.c_void_p(p + 8), _stream())


def _gemm_dims(name, a):
    """(M, N, K, flops) of a GEMM-class launch from its C arguments (for the in-step roofline rows of bench.py)"""
    if name == "hidvae_gemm_f32":
        return (int(a[1]), int(a[2]), int(a[3]), 2.0 * a[1] * a[2] * a[3])
    if name == "hidvae_linear_bwd":  # dW = g^T x (always) + dX = g W (when dX != NULL)
        B, n_out, n_in = int(a[6]), int(a[7]), int(a[8])
        return (B, n_out, n_in, (4.0 if a[12] is not None else 2.0) * B * n_out * n_in)
    if name == "hidvae_bottleneck_fwd":  # enc[-2:] + L levels of K codes + dec[:2] on B items (see include/hidvae.h for the argument order)
        B, K2, N2, L, K, Nd0, Nd1 = int(a[1]), int(a[2]), int(a[3]), int(a[12]), int(a[13]), int(a[22]), int(a[23])
        D = EMBED_DIM
        return (B, N2, K2, 2.0 * B * (K2 * N2 + N2 * D + L * K * D + D * Nd0 + Nd0 * Nd1))
    return None


class _LibProxy:
    """the loaded library; entry points that launch are wrapped so LaunchStamps can bracket them (every one takes the stream last)"""

    def __init__(self, L):
        self._L = L

    def __getattr__(self, name):
        fn = getattr(self._L, name)
        if name in _NOT_LAUNCHES or not name.startswith("hidvae_"):
            return fn
        L = self._L

        def call(*a):
            st = _STAMPS
            if st is None:
                return fn(*a)
            p = st._slot(name, _gemm_dims(name, a), a[-1])
            L.hidvae_timestamp(ctypes.c_void_p(p), a[-1])
            rc = fn(*a)
            L.hidvae_timestamp(ctypes.c_void_p(p + 8), a[-1])
            return rc

        setattr(self, name, call)
        return call


def lib():
    """Load the HIP extension; raise (never fall back) if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"HIP extension missing: {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback on the product path.")
        L = ctypes.CDLL(LIB_PATH)
        L.hidvae_last_error.restype = ctypes.c_char_p
        L.hidvae_version.restype = ctypes.c_char_p
        for name, sig in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = sig
            fn.restype = ctypes.c_int
        _lib = _LibProxy(L)
    return _lib


def exported_symbols():
    return ["hidvae_version", "hidvae_last_error"] + list(_SIGNATURES)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().hidvae_last_error().decode()}")


def _p(t):
    """device pointer of a tensor (None -> NULL); refuses anything the kernels cannot take."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("hidvae HIP kernels need device tensors (got a CPU tensor); there is no CPU fallback")
    return ctypes.c_void_p(t.data_ptr())


def _f32(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a float32 device tensor, got {t.dtype} on {t.device}")
    return t


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _row_stride(t, name):
    """Row stride in elements of a 2-D tensor whose last dim is contiguous (views over wider rows are fine)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"{name}: expected a 2-D tensor with a contiguous last dim, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _vec_stride(t):
    """element stride of a per-item vector (0 for an expanded scalar, or for a 0-d tensor)."""
    if t is None or t.dim() == 0:
        return 0
    return t.stride(0)


def workspace_bytes(op, *dims):
    """hidvae_query_workspace: bytes the `workspace` argument of entry point `op` needs at these dimensions (host-only call)"""
    arr = (ctypes.c_int64 * len(dims))(*[int(d) for d in dims])
    out = ctypes.c_int64(0)
    _check(lib().hidvae_query_workspace(int(op), arr, len(dims), ctypes.byref(out)), "hidvae_query_workspace")
    return int(out.value)


def _ws(op, device, *dims):
    """fp32 workspace tensor of the size the library asks for (None when it needs none)"""
    n = workspace_bytes(op, *dims)
    return torch.empty((n // 4,), device=device, dtype=torch.float32) if n else None


# hidvae_linear_bwd's balanced kernel keeps arrival counters at the head of its workspace: zero on entry, zero again on return, and
# never shared by launches that may run at the same time.  Launches on one stream are serialised, so ONE persistent zero-filled
# buffer per (device, stream) serves them all -- keyed by the ACTUAL stream every launch goes to (round 2 mapped every unregistered
# stream to one shared lane: two Linear backwards on different user streams would have shared counters, silently).  Buffers are
# allocated lazily and never freed (a captured graph may hold their address).
_WS_LANE_BUFFERS = {}
_WS_RETIRED = []  # outgrown buffers stay alive: a captured graph may still hold their address


_SIDE_STREAMS = set()  # handles of the streams that run launches BESIDE the caller's stream (the tag heads' level streams)


def register_ws_lane(stream):
    """Announce a side stream.  Every stream gets a hidvae_linear_bwd workspace of its own on first use anyway; what the announcement
    changes is the launch shape: Linear backwards issued on an announced stream ask for workgroups that leave room on the CUs for the
    other streams' launches (co_resident of hidvae_linear_bwd)."""
    _SIDE_STREAMS.add(int(stream.cuda_stream))
    return int(stream.cuda_stream)


_WS_LAST_CALLER = {}  # device index -> key of the lane most recently used, outside a capture, by a stream that was never announced


def _lane_ws(device, nbytes):
    """the hidvae_linear_bwd workspace of the current stream (its leading arrival counters are zero between launches and must never be
    shared by launches that run at the same time): one persistent zero-filled buffer per (device, stream).  A graph capture runs on a
    stream of its own that no eager step ever saw; it stands in for the caller's stream -- the replay will run there, serialised with
    the caller's other launches -- so it takes the buffer of the unannounced stream that ran the eager warm-up steps.  (Round 3 first
    gave a capturing stream a fresh torch.zeros per CALL: one 17 MB fill kernel per balanced launch and replay, +4-5 % on the untagged
    steps.)"""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    handle = int(torch.cuda.current_stream(device).cuda_stream)
    key = (dev, handle)
    buf = _WS_LANE_BUFFERS.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if buf is None and capturing and handle not in _SIDE_STREAMS:
        stand_in = _WS_LANE_BUFFERS.get(_WS_LAST_CALLER.get(dev))
        if stand_in is not None and stand_in.numel() * 4 >= nbytes:
            return stand_in
    if buf is None or buf.numel() * 4 < nbytes:
        if capturing:
            # first use inside a capture with nothing to stand in: a buffer of the capture's own (its zero-fill replays with the graph)
            return torch.zeros((nbytes // 4,), device=device, dtype=torch.float32)
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = torch.zeros((nbytes // 4,), device=device, dtype=torch.float32)
        _WS_LANE_BUFFERS[key] = buf
    if not capturing and handle not in _SIDE_STREAMS:
        _WS_LAST_CALLER[dev] = key
    return buf


def reset_lane_workspaces():
    """zero every persistent hidvae_linear_bwd workspace again (called when a launch reported an error: a launch that did not run to
    its end may have left arrival counters behind, and every later launch on that stream would silently add wrong partial tiles)"""
    for buf in list(_WS_LANE_BUFFERS.values()) + _WS_RETIRED:
        buf.zero_()


def lane_counters_clean():
    """True when every persistent workspace's arrival counters are zero (debug / tests; synchronises)"""
    n = workspace_bytes(WS_LINEAR_BWD_ZEROED, 1024, 768, 512, 0) // 4
    return all(not bool(buf[:n].any()) for buf in _WS_LANE_BUFFERS.values())


def census_scratch(B, device):
    """A fresh, zero-filled id-census table for batches of B items (hidvae_id_stats / hidvae_bottleneck_fwd).  The CALLER owns it:
    its address is baked into any HIP graph that captured a launch using it, so it must live as long as that graph does."""
    return torch.zeros((workspace_bytes(WS_ID_CENSUS, B) // 8,), device=device, dtype=torch.int64)


def _host_ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class DropSpec:
    """Dropout evaluated INSIDE the launch that produces the activation (include/hidvae.h, hidvae_rng_advance): stands where a keep-mask
    tensor would.  state: device int64[2] {seed, step}; site: the dropout layer's number within the forward pass; p: drop probability."""
    __slots__ = ("state", "site", "p")

    def __init__(self, state, site, p):
        self.state, self.site, self.p = state, int(site), float(p)

    @property
    def threshold(self):
        return min(0xFFFFFFFF, int(round(self.p * 4294967296.0)))


def _mask_args(mask):
    """(keep_mask pointer, rng_state pointer, site, threshold) for an entry point that takes either a keep-mask tensor or a DropSpec"""
    if isinstance(mask, DropSpec):
        return None, _p(mask.state), mask.site, mask.threshold
    return _p(mask), None, 0, 0


def rng_advance(state):
    _check(lib().hidvae_rng_advance(_p(state), _stream()), "hidvae_rng_advance")


def dropout_mask(spec, shape):
    """the 0/1 keep-mask a launch given `spec` applies to an activation of this shape (tests; element index = row-major position)"""
    out = torch.empty(tuple(shape), device=spec.state.device, dtype=torch.float32)
    _check(lib().hidvae_dropout_mask(_p(out), out.numel(), _p(spec.state), spec.site, spec.threshold, _stream()), "hidvae_dropout_mask")
    return out


def gemm(layout, A, B, out=None, bias=None, epilogue=EPI_NONE, aux=None, split_k=1, accumulate=False, mask=None,
         mask_scale=1.0):
    """C = epilogue(op(A) op(B) + bias).  NT: A[M,K] B[N,K]; NN: A[M,K] B[K,N]; TN: A[K,M] B[K,N]."""
    _f32(A, "A"), _f32(B, "B")
    lda, ldb = _row_stride(A, "A"), _row_stride(B, "B")
    if layout == GEMM_NT:
        M, K = A.shape
        N, K2 = B.shape
    elif layout == GEMM_NN:
        M, K = A.shape
        K2, N = B.shape
    else:
        K, M = A.shape
        K2, N = B.shape
    if K != K2:
        raise RuntimeError(f"gemm: inner dimensions differ ({K} vs {K2})")  # the reference's shape assert (encoder.py:35)
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    ws = _ws(WS_GEMM, A.device, M, N, K, split_k)  # slabs of the LDS-tiled (large-batch) path / deep-K weight gradients
    ldaux = _row_stride(aux, "aux") if aux is not None else 0
    mp, rs, site, thr = _mask_args(mask)
    ldmask = _row_stride(mask, "mask") if mp is not None else 0
    _check(lib().hidvae_gemm_f32(layout, M, N, K, _p(A), lda, _p(B), ldb, _p(bias), _p(out), _row_stride(out, "C"),
                                 epilogue, _p(aux), ldaux, mp, ldmask, float(mask_scale), rs, site, thr, split_k, _p(ws),
                                 int(accumulate), _stream()), "hidvae_gemm_f32")
    return out


def linear_bwd(g, x, w, need_dx=True, epilogue=EPI_NONE, aux=None, dW=None, accumulate=False, bias=False, db=None, accumulate_db=False,
               dx_scale=1.0):
    """backward of y = x W^T in one launch -> (dW [n_out,n_in], dX [B,n_in] or None); dX = epilogue(g W) with a D* code + aux.
    dx_scale multiplies the EPI_DRELU result: the backward through ReLU -> Dropout(keep_scale) read off that layer's saved output."""
    _f32(g, "g"), _f32(x, "x")
    B, n_out = g.shape
    n_in = x.shape[1]
    if x.shape[0] != B or (need_dx and tuple(w.shape) != (n_out, n_in)):
        raise RuntimeError(f"linear_bwd: shapes g {tuple(g.shape)} x {tuple(x.shape)} W {tuple(w.shape)}")
    if dW is None:
        dW = torch.empty((n_out, n_in), device=g.device, dtype=torch.float32)
        accumulate = False
    elif tuple(dW.shape) != (n_out, n_in) or not dW.is_contiguous():
        raise RuntimeError(f"linear_bwd: dW slot has shape {tuple(dW.shape)}, expected {(n_out, n_in)} contiguous")
    dX = torch.empty((B, n_in), device=g.device, dtype=torch.float32) if need_dx else None
    if bias and db is None:
        db = torch.empty((n_out,), device=g.device, dtype=torch.float32)
        accumulate_db = False
    if workspace_bytes(WS_LINEAR_BWD_ZEROED, B, n_out, n_in, int(bool(bias))):
        ws = _lane_ws(g.device, workspace_bytes(WS_LINEAR_BWD, B, n_out, n_in, int(bool(bias))))
    else:
        ws = _ws(WS_LINEAR_BWD, g.device, B, n_out, n_in, int(bool(bias)))
    rc = lib().hidvae_linear_bwd(_p(g), _row_stride(g, "g"), _p(x), _row_stride(x, "x"), _p(w if need_dx else None),
                                 _row_stride(w, "W") if need_dx else 0, B, n_out, n_in, _p(dW), n_in, int(bool(accumulate)), _p(dX), n_in,
                                 int(epilogue), _p(aux), _row_stride(aux, "aux") if aux is not None else 0, float(dx_scale),
                                 _p(db if bias else None), int(bool(accumulate_db)), _p(ws),
                                 int(int(torch.cuda.current_stream(g.device).cuda_stream) in _SIDE_STREAMS), _stream())
    if rc != 0:
        msg = lib().hidvae_last_error().decode()
        reset_lane_workspaces()  # the zero-on-entry contract of the arrival counters may no longer hold
        raise RuntimeError(f"hidvae_linear_bwd failed ({rc}): {msg}")
    if bias:
        return dW, dX, db
    return dW, dX


def colsum(X, out=None, accumulate=False):
    _f32(X, "X")
    M, N = X.shape
    if out is None:
        out = torch.empty((N,), device=X.device, dtype=torch.float32)
    ws = _ws(WS_COLSUM, X.device, M, N)
    _check(lib().hidvae_colsum(_p(X), M, N, _row_stride(X, "X"), _p(out), int(accumulate), _p(ws), _stream()), "hidvae_colsum")
    return out


def check_embed_dim(D):
    if not (D == EMBED_DIM or (4 <= D <= 64 and D % 4 == 0)):
        raise NotImplementedError(f"embed_dim={D}: the quantiser kernels take 32 (fused) or another multiple of 4 up to 64 (width-independent form)")


def codebook_prepare(tables, normalize_flags):
    """tables: list of L raw [K,D] tensors -> (cb_eff [L,K,D], cc [L,K]).  D = 32: the fused kernels; other multiples of 4 up to 64: the
    width-independent kernels (csrc/rq_generic.hip)."""
    L = len(tables)
    K, D = tables[0].shape
    check_embed_dim(D)
    for t in tables:
        _f32(t, "codebook")
        if tuple(t.shape) != (K, D) or not t.is_contiguous():
            raise RuntimeError("codebooks must be contiguous and share one shape")
    cb = torch.empty((L, K, D), device=tables[0].device, dtype=torch.float32)
    cc = torch.empty((L, K), device=tables[0].device, dtype=torch.float32)
    flags = (ctypes.c_int32 * L)(*[int(bool(f)) for f in normalize_flags])
    carried = take_pending_adamw(tables[0].device)
    if carried is not None:  # the optimizer's per-step scalars ride along in this launch (one workgroup more)
        desc, step_dev, b1, b2, eta_min, T_max, step_size, gamma = carried
        _check(lib().hidvae_codebook_prepare_adamw(_host_ptr_array(tables), flags, L, K, _p(cb), _p(cc), _p(step_dev), _p(desc["lr"]),
                                                   _p(desc["wd"]), int(desc["n"]), float(b1), float(b2), float(eta_min), int(T_max),
                                                   int(step_size), float(gamma), _p(desc["hyper"]), int(D), _stream()),
               "hidvae_codebook_prepare_adamw")
        return cb, cc
    _check(lib().hidvae_codebook_prepare(_host_ptr_array(tables), flags, L, K, _p(cb), _p(cc), int(D), _stream()), "hidvae_codebook_prepare")
    return cb, cc


# The optimizer's start-of-step launch (adamw_prepare) may be deferred: whoever launches codebook_prepare next on the same device
# carries it, and optimizer.step() launches it itself if nobody did.  `owner` lets the optimizer tell whether it is still pending.
_PENDING_ADAMW = {}


def defer_adamw_prepare(device, owner, args):
    _PENDING_ADAMW[torch.device(device).index or 0] = (owner, args)


def take_pending_adamw(device, owner=None):
    key = torch.device(device).index or 0
    ent = _PENDING_ADAMW.get(key)
    if ent is None or (owner is not None and ent[0] is not owner):
        return None
    del _PENDING_ADAMW[key]
    ent[0]._prepared = True
    return ent[1]


def rq_forward(y, cb_eff, cc, normalize_input, mode, training, beta, want_res=False, want_z=True, distance=DIST_L2):
    _f32(y, "y")
    L, K, D = cb_eff.shape
    if y.dim() != 2 or y.shape[1] != D or not y.is_contiguous():
        raise RuntimeError(f"rq_forward: expected contiguous [B,{D}] input, got {tuple(y.shape)}")  # quantize.py:101
    B = y.shape[0]
    dev = y.device
    z = torch.empty((B, D), device=dev, dtype=torch.float32) if (want_z or normalize_input) else None
    ids = torch.empty((B, L), device=dev, dtype=torch.int64)
    emb_cat = torch.empty((B, L * D), device=dev, dtype=torch.float32)
    emb_sum = torch.empty((B, D), device=dev, dtype=torch.float32)
    res = torch.empty((B, L * D), device=dev, dtype=torch.float32) if want_res else None
    qloss = torch.empty((B,), device=dev, dtype=torch.float32)
    ws = _ws(WS_RQ_FORWARD, dev, B, L, K) if (D == EMBED_DIM and distance == DIST_L2) else None
    _check(lib().hidvae_rq_forward(_p(y), B, int(bool(normalize_input)), _p(cb_eff), _p(cc), L, K, int(mode), int(bool(training)),
                                   float(beta), _p(z), _p(ids), _p(emb_cat), L * D, _p(emb_sum), _p(res), _p(qloss),
                                   _p(ws), int(D), int(distance), _stream()), "hidvae_rq_forward")
    return z if z is not None else y, ids, emb_cat, emb_sum, res, qloss


def rq_ids(y, cb_eff, cc, normalize_input=False):
    """Semantic ids only (eval-mode search, o = e): hidvae_rq_forward with every output but `ids` NULL -- the corpus tokenisation of
    HSemanticIdTokenizer.precompute_corpus_ids (reference h_semids.py:109-195).  At corpus sizes the launch takes the ids-only form of
    the prefilter kernel (no output rows, no winner fetch after the last level); the ids are those of rq_forward bit for bit."""
    _f32(y, "y")
    L, K, D = cb_eff.shape
    if y.dim() != 2 or y.shape[1] != D or not y.is_contiguous():
        raise RuntimeError(f"rq_ids: expected contiguous [B,{D}] input, got {tuple(y.shape)}")
    B = y.shape[0]
    ids = torch.empty((B, L), device=y.device, dtype=torch.int64)
    ws = _ws(WS_RQ_FORWARD, y.device, B, L, K) if D == EMBED_DIM else None
    _check(lib().hidvae_rq_forward(_p(y), B, int(bool(normalize_input)), _p(cb_eff), _p(cc), L, K, MODE_STE, 0, 0.0, None, _p(ids),
                                   None, L * D, None, None, None, _p(ws), int(D), DIST_L2, _stream()), "hidvae_rq_forward")
    return ids


def bottleneck_eligible(B, K2, N2, Nd0, Nd1, L, K):
    """the fused middle-of-the-step launch: small batches, widths the kernel keeps in LDS, codebooks resident beside them"""
    Kp = (K + 255) // 256 * 256  # (the launch's 8 waves each scan a share of the codes, 32 at a time)
    lds = L * (33 * Kp + 64) * 4 + (2 * 16 * 64 + 2 * 64) * 16
    return (B <= 4096 and all(v % 16 == 0 and v >= 16 for v in (K2, N2, Nd0, Nd1)) and max(K2, N2, Nd0) <= 256 and Kp <= 1024
            and lds <= 160 * 1024 - 1024)


def census_eligible(L, K):
    """can the fused middle launch carry the id census (10 bits per level in a 40-bit slot field)?"""
    return L <= 4 and K <= 1024


def bottleneck_fwd(h1, W2, W3, cb_eff, cc, normalize_input, mode, beta, Wd0, Wd1, id_stats=False, scratch=None):
    """-> dict of every tensor the launch writes (see include/hidvae.h); id_stats: also embs_norm [B,L] and p_unique, through the
    caller-owned census table `scratch` (census_scratch(B, device))"""
    _f32(h1, "h1")
    B, K2 = h1.shape
    N2, Nd0, Nd1 = W2.shape[0], Wd0.shape[0], Wd1.shape[0]
    L, K, _ = cb_eff.shape
    dev = h1.device
    f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    o = dict(pre2=f(B, N2), h2=f(B, N2), y=f(B, EMBED_DIM), z=f(B, EMBED_DIM), ids=torch.empty((B, L), device=dev, dtype=torch.int64),
             emb_cat=f(B, L * EMBED_DIM), emb_sum=f(B, EMBED_DIM), qloss=f(B), pre_d0=f(B, Nd0), d0=f(B, Nd0), pre_d1=f(B, Nd1), d1=f(B, Nd1))
    if not (h1.is_contiguous() and W2.is_contiguous() and W3.is_contiguous() and Wd0.is_contiguous() and Wd1.is_contiguous()):
        raise RuntimeError("bottleneck_fwd: operands must be contiguous")
    if tuple(W2.shape) != (N2, K2) or tuple(W3.shape) != (EMBED_DIM, N2) or tuple(Wd0.shape) != (Nd0, EMBED_DIM) or tuple(Wd1.shape) != (Nd1, Nd0):
        raise RuntimeError("bottleneck_fwd: layer shapes do not chain")
    if id_stats:
        if not census_eligible(L, K):
            raise RuntimeError(f"bottleneck_fwd: the fused id census needs L <= 4 and K <= 1024 (got {L}, {K})")
        if scratch is None or scratch.numel() * 8 != workspace_bytes(WS_ID_CENSUS, B) or scratch.dtype != torch.int64:
            raise RuntimeError("bottleneck_fwd: id_stats needs the caller's census table for this batch size (_C.census_scratch(B, device))")
        o["embs_norm"] = f(B, L)
        o["p_unique"] = torch.empty((), device=dev, dtype=torch.float32)
    _check(lib().hidvae_bottleneck_fwd(_p(h1), B, K2, N2, _p(W2), _p(W3), _p(o["pre2"]), _p(o["h2"]), _p(o["y"]), int(bool(normalize_input)),
                                       _p(cb_eff), _p(cc), L, K, int(mode), float(beta), _p(o["z"]), _p(o["ids"]), _p(o["emb_cat"]),
                                       L * EMBED_DIM, _p(o["emb_sum"]), _p(o["qloss"]), Nd0, Nd1, _p(Wd0), _p(Wd1), _p(o["pre_d0"]),
                                       _p(o["d0"]), _p(o["pre_d1"]), _p(o["d1"]), _p(o.get("embs_norm")), _p(o.get("p_unique")),
                                       _p(scratch if id_stats else None), _stream()), "hidvae_bottleneck_fwd")
    return o


def rq_backward(y, z, cb_eff, cc, normalize_input, mode, beta, ids, g_cat, g_sum, g_z_in, gq, gq_items):
    """g_cat: the gradient of emb_cat [B, L*D], or a LIST of its prefix slices still to be summed (contiguous [B, w_s] tensors: what the
    tag heads' early backward leaves behind) -- added up inside the launch at embed_dim 32, by hidvae_sum_prefix_slices first otherwise"""
    B = z.shape[0]
    L, K, D = cb_eff.shape
    g_y = torch.empty_like(z)
    dE = torch.empty((B, L * D), device=z.device, dtype=torch.float32)
    if isinstance(g_cat, (list, tuple)):
        live = [t for t in g_cat if t is not None]
        if not live:
            g_cat = None
        elif D != EMBED_DIM or len(live) > 2 * MAX_LEVELS or any(not t.is_contiguous() or t.shape[1] % D for t in live):
            g_cat = sum_prefix_slices(live, B, L * D)
        else:
            ptrs = (ctypes.c_void_p * len(live))(*[t.data_ptr() for t in live])
            widths = (ctypes.c_int32 * len(live))(*[t.shape[1] for t in live])
            _check(lib().hidvae_rq_backward_slices(_p(y), _p(z), B, int(bool(normalize_input)), _p(cb_eff), _p(cc), L, K, int(mode), float(beta),
                                                   _p(ids), ptrs, widths, len(live), _p(g_sum), _p(g_z_in),
                                                   int(g_z_in.shape[0]) if g_z_in is not None else 0, float(gq), _p(gq_items),
                                                   _vec_stride(gq_items), _p(g_y), _p(dE), int(D), _stream()), "hidvae_rq_backward_slices")
            return g_y, dE
    ldg = _row_stride(g_cat, "g_cat") if g_cat is not None else 0
    _check(lib().hidvae_rq_backward(_p(y), _p(z), B, int(bool(normalize_input)), _p(cb_eff), _p(cc), L, K, int(mode), float(beta),
                                    _p(ids), _p(g_cat), ldg, _p(g_sum), _p(g_z_in),
                                    int(g_z_in.shape[0]) if g_z_in is not None else 0, float(gq), _p(gq_items), _vec_stride(gq_items), _p(g_y), _p(dE),
                                    int(D), _stream()), "hidvae_rq_backward")
    return g_y, dE


def codebook_grad(ids, dE_rows, tables, cb_eff, normalize_flags, grads=None, accumulate=False):
    B, L = ids.shape
    K, D = tables[0].shape
    if grads is None:
        grads = [torch.empty_like(t) for t in tables]
    flags = (ctypes.c_int32 * L)(*[int(bool(f)) for f in normalize_flags])
    ws = _ws(WS_CODEBOOK_GRAD, ids.device, B, L, K) if D == EMBED_DIM else None
    _check(lib().hidvae_codebook_grad(_p(ids), _p(dE_rows), B, L, K, _host_ptr_array(tables), _p(cb_eff), flags,
                                      _host_ptr_array(grads), int(accumulate), _p(ws), int(D), _stream()), "hidvae_codebook_grad")
    return grads


def _n_cat(n_cat, N):
    n_cat = int(n_cat)
    if not 0 <= n_cat < N:
        raise RuntimeError(f"n_cat_features={n_cat}: must leave at least one of the {N} columns non-categorical")
    return n_cat


def recon_fwd_bwd(y, x, gscale=1.0, gscale_items=None, want_xhat=False, want_grad=False, n_cat=0):
    """decoder tail; n_cat > 0: the last n_cat columns enter as BCE-with-logits (reference loss.py:15-33, h_rqvae.py:610-613)"""
    _f32(y, "y"), _f32(x, "x")
    if y.shape != x.shape or not y.is_contiguous() or not x.is_contiguous():
        raise RuntimeError(f"recon: shapes differ or not contiguous ({tuple(y.shape)} vs {tuple(x.shape)})")
    B, N = y.shape
    x_hat = torch.empty_like(y) if want_xhat else None
    recon = torch.empty((B,), device=y.device, dtype=torch.float32)
    g_y = torch.empty_like(y) if want_grad else None
    _check(lib().hidvae_recon_fwd_bwd(_p(y), _p(x), B, N, _n_cat(n_cat, N), float(gscale), _p(gscale_items), _vec_stride(gscale_items), _p(x_hat), _p(recon),
                                      _p(g_y), _stream()),
           "hidvae_recon_fwd_bwd")
    return recon, x_hat, g_y


def cat_recon_rows(x_hat, x, n_cat, g=None):
    """CategoricalReconstructionLoss.forward on given x_hat (reference loss.py:15-33): -> out [M], or with g [M] -> g_xhat [M,N]"""
    _f32(x_hat, "x_hat"), _f32(x, "x")
    if x_hat.shape != x.shape or x_hat.dim() != 2:
        raise RuntimeError(f"cat_recon_rows: shapes differ ({tuple(x_hat.shape)} vs {tuple(x.shape)})")
    M, N = x_hat.shape
    if not 0 <= int(n_cat) <= N:
        raise RuntimeError(f"cat_recon_rows: n_cat={n_cat} of {N} columns")
    out = torch.empty((M,), device=x.device, dtype=torch.float32) if g is None else None
    g_xhat = torch.empty((M, N), device=x.device, dtype=torch.float32) if g is not None else None
    _check(lib().hidvae_cat_recon_rows(_p(x_hat), _row_stride(x_hat, "x_hat"), _p(x), _row_stride(x, "x"), M, N, int(n_cat), _p(g), _vec_stride(g),
                                       _p(out), _p(g_xhat), _stream()), "hidvae_cat_recon_rows")
    return out if g is None else g_xhat


def l2norm_fwd(x, eps=1e-12):
    _f32(x, "x")
    M, N = x.shape
    out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    norms = torch.empty((M,), device=x.device, dtype=torch.float32)
    ldx = _row_stride(x, "x")
    if N == EMBED_DIM and ldx % 4 == 0 and x.data_ptr() % 16 == 0:
        _check(lib().hidvae_l2norm32_fwd(_p(x), M, ldx, float(eps), _p(out), N, _p(norms), _stream()), "hidvae_l2norm32_fwd")
    else:
        _check(lib().hidvae_l2norm_fwd(_p(x), M, N, ldx, float(eps), _p(out), N, _p(norms), _stream()), "hidvae_l2norm_fwd")
    return out, norms


def l2norm_fwd_pair(x0, x1, eps=1e-12):
    """two row normalisations of the same number of rows in one launch -> (out0, norms0, out1, norms1)"""
    _f32(x0, "x0"), _f32(x1, "x1")
    M = x0.shape[0]
    if x1.shape[0] != M or x0.dim() != 2 or x1.dim() != 2:
        raise RuntimeError(f"l2norm_fwd_pair: row counts differ ({tuple(x0.shape)} vs {tuple(x1.shape)})")
    f = lambda *shape: torch.empty(shape, device=x0.device, dtype=torch.float32)
    o0, n0, o1, n1 = f(M, x0.shape[1]), f(M), f(M, x1.shape[1]), f(M)
    _check(lib().hidvae_l2norm_fwd_pair(_p(x0), _row_stride(x0, "x0"), x0.shape[1], _p(o0), _p(n0), _p(x1), _row_stride(x1, "x1"), x1.shape[1],
                                        _p(o1), _p(n1), M, float(eps), _stream()), "hidvae_l2norm_fwd_pair")
    return o0, n0, o1, n1


def l2norm_bwd_pair(g0, out0, norms0, g1, out1, norms1, eps=1e-12):
    M = out0.shape[0]
    gx0, gx1 = torch.empty_like(out0), torch.empty_like(out1)
    _check(lib().hidvae_l2norm_bwd_pair(_p(g0), _row_stride(g0, "g0"), _p(out0), _p(norms0), out0.shape[1], _p(gx0), _p(g1), _row_stride(g1, "g1"),
                                        _p(out1), _p(norms1), out1.shape[1], _p(gx1), M, float(eps), _stream()), "hidvae_l2norm_bwd_pair")
    return gx0, gx1


def l2norm_bwd(g, out, norms, eps=1e-12, gx=None, accumulate=False):
    M, N = out.shape
    if gx is None:
        gx = torch.empty((M, N), device=out.device, dtype=torch.float32)
    _check(lib().hidvae_l2norm_bwd(_p(g), _row_stride(g, "g"), _p(out), _row_stride(out, "out"), _p(norms), M, N, float(eps), _p(gx),
                                   _row_stride(gx, "gx"), int(accumulate), _stream()), "hidvae_l2norm_bwd")
    return gx


def id_stats(emb_cat, ids, want_norms=True, scratch=None):
    """scratch: the caller's census table for this batch size (census_scratch(B, device)); None allocates a fresh one for this call"""
    B, L = ids.shape
    dev = ids.device
    if scratch is None:
        scratch = census_scratch(B, dev)
    elif scratch.numel() * 8 != workspace_bytes(WS_ID_CENSUS, B) or scratch.dtype != torch.int64:
        raise RuntimeError("id_stats: the census table does not match this batch size")
    norms = torch.empty((B, L), device=dev, dtype=torch.float32) if want_norms else None
    pu = torch.empty((), device=dev, dtype=torch.float32)
    ld = _row_stride(emb_cat, "emb_cat") if emb_cat is not None else 0
    D = emb_cat.shape[1] // L if emb_cat is not None else EMBED_DIM
    _check(lib().hidvae_id_stats(_p(emb_cat if want_norms else None), ld, _p(ids), B, L, _p(norms), _p(pu), _p(scratch), int(D), _stream()),
           "hidvae_id_stats")
    return norms, pu


def adamw_prepare(desc, step_dev, beta1, beta2, eta_min, T_max, step_size=0, gamma=1.0):
    _check(lib().hidvae_adamw_prepare(_p(step_dev), _p(desc["lr"]), _p(desc["wd"]), int(desc["n"]), float(beta1), float(beta2),
                                      float(eta_min), int(T_max), int(step_size), float(gamma), _p(desc["hyper"]), _stream()),
           "hidvae_adamw_prepare")


def adamw_step(desc, beta1, beta2, eps, grad_scale, lo=0, hi=None, g_host=None):
    """desc: device tables built by optim.HidvaeAdamW (p/m/v pointer tables, numel, hyper) + the host gradient table.
    lo / hi: only the tensors [lo, hi) of the tables (g_host then holds just their gradient pointers): the tables are plain arrays, so a
    sub-range is the same call on offset pointers."""
    n = int(desc["n"])
    hi = n if hi is None else int(hi)
    lo = int(lo)
    if not 0 <= lo < hi <= n:
        raise RuntimeError(f"adamw_step: tensor range [{lo}, {hi}) of {n}")
    off = lambda t, k: ctypes.c_void_p(t.data_ptr() + k * t.element_size())
    _check(lib().hidvae_adamw_step(off(desc["p"], lo), desc["g_host"] if g_host is None else g_host, off(desc["m"], lo), off(desc["v"], lo),
                                   off(desc["numel"], lo), off(desc["hyper"], 3 * lo), hi - lo, int(desc["max_numel"]), float(beta1),
                                   float(beta2), float(eps), float(grad_scale), _stream()), "hidvae_adamw_step")


def uniq_loss(ids, z, weight, margin, want_grad=False):
    B, L = ids.shape
    loss = torch.empty((), device=ids.device, dtype=torch.float32)
    D = z.shape[1]
    g_rows = torch.empty((L, D), device=ids.device, dtype=torch.float32) if want_grad else None
    _check(lib().hidvae_uniq_loss(_p(ids), _p(z), B, L, float(weight), float(margin), _p(loss), _p(g_rows), int(D), _stream()), "hidvae_uniq_loss")
    return loss, g_rows


def total_loss(recon, qloss, aligns, preds, accs, tag_div, ids, z, uniq_weight, uniq_margin, w_a, w_p, w_u, want_grad):
    """-> (loss, uniq, g_rows, tagstats).  aligns/preds/accs: lists of per-level 0-d device tensors ([] if untagged)."""
    dev = recon.device
    loss = torch.empty((), device=dev, dtype=torch.float32)
    uniq = torch.empty((), device=dev, dtype=torch.float32)
    L = ids.shape[1] if ids is not None else 0
    n_tag = len(aligns)
    D = z.shape[1] if z is not None else EMBED_DIM
    g_rows = torch.empty((L, D), device=dev, dtype=torch.float32) if (want_grad and ids is not None) else None
    tagstats = torch.empty((3 + 3 * n_tag,), device=dev, dtype=torch.float32) if n_tag else None
    arr = lambda ts: _host_ptr_array(ts) if ts else None
    _check(lib().hidvae_total_loss(_p(recon), _p(qloss), recon.shape[0], arr(aligns), arr(preds), arr(accs), n_tag, float(tag_div),
                                   _p(ids), _p(z), L, float(uniq_weight), float(uniq_margin), float(w_a), float(w_p), float(w_u),
                                   _p(loss), _p(uniq), _p(g_rows), _p(tagstats), None, int(D), _stream()), "hidvae_total_loss")
    return loss, uniq, g_rows, tagstats


def total_loss_bwd(g_loss, B, L, w_a, w_p, w_u, g_rows, want_gz, embed_dim=EMBED_DIM):
    scal = torch.empty((3,), device=g_loss.device, dtype=torch.float32)
    D = g_rows.shape[1] if g_rows is not None else int(embed_dim)
    g_z = torch.empty((B, D), device=g_loss.device, dtype=torch.float32) if want_gz else None
    _check(lib().hidvae_total_loss_bwd(_p(g_loss), B, L, float(w_a), float(w_p), float(w_u), _p(g_rows), _p(scal), _p(g_z), int(D), _stream()),
           "hidvae_total_loss_bwd")
    return scal, g_z


def loss_fwd(y, x, qloss, aligns, preds, accs, tag_div, ids, z, uniq_weight, uniq_margin, w_a, w_p, w_u, want_grad, n_cat=0):
    """decoder tail + total loss -> (loss, recon, uniq, g_rows, tagstats, summary [6] = the training log row)"""
    _f32(y, "y"), _f32(x, "x")
    if y.shape != x.shape or not y.is_contiguous() or not x.is_contiguous():
        raise RuntimeError(f"loss: shapes differ or not contiguous ({tuple(y.shape)} vs {tuple(x.shape)})")
    dev = y.device
    B, N = y.shape
    recon = torch.empty((B,), device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    uniq = torch.empty((), device=dev, dtype=torch.float32)
    L = ids.shape[1] if ids is not None else 0
    n_tag = len(aligns)
    D = z.shape[1] if z is not None else EMBED_DIM
    g_rows = torch.empty((L, D), device=dev, dtype=torch.float32) if (want_grad and ids is not None) else None
    tagstats = torch.empty((3 + 3 * n_tag,), device=dev, dtype=torch.float32) if n_tag else None
    summary = torch.empty((6,), device=dev, dtype=torch.float32)
    arr = lambda ts: _host_ptr_array(ts) if ts else None
    _check(lib().hidvae_loss_fwd(_p(y), _p(x), B, N, _n_cat(n_cat, N), _p(qloss), arr(aligns), arr(preds), arr(accs), n_tag, float(tag_div), _p(ids), _p(z), L,
                                 float(uniq_weight), float(uniq_margin), float(w_a), float(w_p), float(w_u), _p(recon), _p(loss), _p(uniq),
                                 _p(g_rows), _p(tagstats), _p(summary), int(D), _stream()), "hidvae_loss_fwd")
    return loss, recon, uniq, g_rows, tagstats, summary


def loss_bwd(g_loss, y, x, L, w_a, w_p, w_u, g_rows, want_gz, embed_dim=EMBED_DIM, n_cat=0, expect_g=0.0):
    B, N = y.shape
    scal = torch.empty((3,), device=y.device, dtype=torch.float32)
    g_y = torch.empty_like(y)
    D = g_rows.shape[1] if g_rows is not None else int(embed_dim)
    g_z = torch.empty((B, D), device=y.device, dtype=torch.float32) if want_gz else None
    _check(lib().hidvae_loss_bwd(_p(g_loss), _p(y), _p(x), B, N, _n_cat(n_cat, N), L, float(w_a), float(w_p), float(w_u), _p(g_rows), _p(g_y), _p(scal), _p(g_z),
                                 int(D), float(expect_g), _stream()), "hidvae_loss_bwd")
    return g_y, scal, g_z


# ------------------------------------------------------------------------------------------------ tag path
def act_bwd(g, ref, act, mask=None, mask_scale=1.0):
    out = torch.empty_like(g)
    _check(lib().hidvae_act_bwd(_p(g), _p(ref), g.numel(), int(act), _p(mask), float(mask_scale), _p(out), _stream()), "hidvae_act_bwd")
    return out


def binary(op, a, b):
    """op 0: a*b, op 1: a+b, op 2: a-b (row-strided views welcome)."""
    M, N = a.shape
    out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    _check(lib().hidvae_binary(int(op), _p(a), _row_stride(a, "a"), _p(b), _row_stride(b, "b"), M, N, _p(out), N, _stream()), "hidvae_binary")
    return out


def mul(a, b):
    return binary(0, a, b)


def sum_prefix_slices(srcs, M, N):
    """srcs: list of contiguous [M, w_i] tensors (or None) -> dst [M, N] with each added into the first w_i columns."""
    live = [t for t in srcs if t is not None]
    dst = torch.empty((M, N), device=live[0].device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * len(live))(*[t.data_ptr() for t in live])
    widths = (ctypes.c_int32 * len(live))(*[t.shape[1] for t in live])
    _check(lib().hidvae_sum_prefix_slices(ptrs, widths, len(live), M, N, _p(dst), _stream()), "hidvae_sum_prefix_slices")
    return dst


def layernorm_fwd(x, gamma, beta, eps, relu, mask, mask_scale, residual):
    M, N = x.shape
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    mean = torch.empty((M,), device=x.device, dtype=torch.float32)
    rstd = torch.empty((M,), device=x.device, dtype=torch.float32)
    mp, rs, site, thr = _mask_args(mask)
    _check(lib().hidvae_layernorm_fwd(_p(x), M, N, _p(gamma), _p(beta), float(eps), _p(y), _p(mean), _p(rstd), int(relu), mp,
                                      float(mask_scale), _p(residual), rs, site, thr, _stream()), "hidvae_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(gy, x, gamma, beta, mean, rstd, relu, mask, mask_scale):
    M, N = x.shape
    gx = torch.empty((M, N), device=x.device, dtype=torch.float32)
    _check(lib().hidvae_layernorm_bwd(_p(gy), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), M, N, int(relu), _p(mask),
                                      float(mask_scale), _p(gx), _stream()), "hidvae_layernorm_bwd")
    return gx


def layernorm_param_grad(gy, x, gamma, beta, mean, rstd, relu, mask, mask_scale):
    M, N = x.shape
    gg = torch.empty((N,), device=x.device, dtype=torch.float32)
    gb = torch.empty((N,), device=x.device, dtype=torch.float32)
    ws = _ws(WS_LAYERNORM_PARAM_GRAD, x.device, M, N)
    _check(lib().hidvae_layernorm_param_grad(_p(gy), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), M, N, int(relu), _p(mask),
                                             float(mask_scale), _p(gg), _p(gb), 0, _p(ws), _stream()), "hidvae_layernorm_param_grad")
    return gg, gb


def layernorm_bwd_all(gy, x, gamma, beta, mean, rstd, relu, mask, mask_scale, need_gx=True, gg=None, gb=None, accumulate=False):
    """input + affine gradients from one pass -> (gx or None, ggamma, gbeta); gg/gb: optional destination slots"""
    M, N = x.shape
    gx = torch.empty((M, N), device=x.device, dtype=torch.float32) if need_gx else None
    if gg is None or gb is None:
        gg = torch.empty((N,), device=x.device, dtype=torch.float32)
        gb = torch.empty((N,), device=x.device, dtype=torch.float32)
        accumulate = False
    ws = _ws(WS_LAYERNORM_BWD_ALL, x.device, M, N)
    _check(lib().hidvae_layernorm_bwd_all(_p(gy), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), M, N, int(relu), _p(mask),
                                          float(mask_scale), _p(gx), _p(gg), _p(gb), int(bool(accumulate)), _p(ws), _stream()),
           "hidvae_layernorm_bwd_all")
    return gx, gg, gb


def layernorm_bwd_partial(gy, x, gamma, beta, mean, rstd, relu, y_out, keep_scale, in_relu_scale=0.0, need_gx=True, gy2=None, want_sum=False):
    """LayerNorm backward up to its seam: -> (gx or None, partials).  The affine gradients follow from `partials` in
    layernorm_param_final_many (one launch for many LayerNorms).  y_out: the forward output (ReLU -> Dropout gate, no mask needed);
    in_relu_scale != 0: gx is returned already taken through the ReLU -> Dropout that produced the LayerNorm's input.
    gy2: a second gradient of the same output, added on the way in; want_sum: -> (gx, partials, gy + gy2)."""
    M, N = x.shape
    if gy2 is not None and (gy2.shape != gy.shape or not gy2.is_contiguous()):
        raise RuntimeError(f"layernorm_bwd_partial: gy2 {tuple(gy2.shape)} vs gy {tuple(gy.shape)}")
    gx = torch.empty((M, N), device=x.device, dtype=torch.float32) if need_gx else None
    gsum = torch.empty((M, N), device=x.device, dtype=torch.float32) if want_sum else None
    part = _ws(WS_LAYERNORM_BWD_ALL, x.device, M, N)
    _check(lib().hidvae_layernorm_bwd_partial(_p(gy), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), M, N, int(relu), _p(y_out),
                                              float(keep_scale), float(in_relu_scale), _p(gy2), _p(gsum), _p(gx), _p(part), _stream()),
           "hidvae_layernorm_bwd_partial")
    return (gx, part, gsum) if want_sum else (gx, part)


class LnFinal(ctypes.Structure):  # hidvae_ln_final
    _fields_ = [("partials", _vp), ("M", _i64), ("N", _i64), ("ggamma", _vp), ("gbeta", _vp), ("accumulate", _i)]


def layernorm_param_final_many(problems):
    """problems: list of (partials, M, N, ggamma, gbeta, accumulate): the affine gradients of all of them from ONE launch"""
    arr = (LnFinal * len(problems))()
    for q, (part, M, N, gg, gb, acc) in zip(arr, problems):
        q.partials, q.M, q.N, q.ggamma, q.gbeta, q.accumulate = part.data_ptr(), int(M), int(N), gg.data_ptr(), gb.data_ptr(), int(bool(acc))
    _check(lib().hidvae_layernorm_param_final_many(ctypes.cast(arr, _vp), len(problems), _stream()), "hidvae_layernorm_param_final_many")


class PredUnit(ctypes.Structure):  # hidvae_pred_unit
    _fields_ = [("W", _vp), ("bias", _vp), ("gamma", _vp), ("beta", _vp), ("N", _i), ("K", _i),
                ("act1", _i), ("act2", _i), ("residual", _i), ("carry", _i),
                ("drop_site1", ctypes.c_uint), ("drop_threshold1", ctypes.c_uint), ("drop_site2", ctypes.c_uint), ("drop_threshold2", ctypes.c_uint),
                ("drop_scale1", ctypes.c_float), ("drop_scale2", ctypes.c_float), ("eps", ctypes.c_float),
                ("lin", _vp), ("y", _vp), ("mean", _vp), ("rstd", _vp), ("g_lin", _vp), ("partials", _vp)]


PRED_WMAX, PRED_MAX_UNITS = 256, 10


def predictor_fwd(h, units):
    """TagPredictor behind its gate in one row-local launch.  units: list of dicts with lin (nn.Linear), norm (nn.LayerNorm or None),
    act1 / act2 (bool), drop1 / drop2 ((DropSpec, scale) or None), residual, carry.  -> per unit dict(lin=..., y=..., mean=..., rstd=...)
    with y / mean / rstd None for units without a LayerNorm."""
    _f32(h, "h")
    B = h.shape[0]
    dev = h.device
    arr = (PredUnit * len(units))()
    outs, state = [], None
    for q, u in zip(arr, units):
        lin, norm = u["lin"], u.get("norm")
        N, K = lin.out_features, lin.in_features
        o = dict(lin=torch.empty((B, N), device=dev, dtype=torch.float32), y=None, mean=None, rstd=None)
        q.W, q.bias = lin.weight.data_ptr(), (lin.bias.data_ptr() if lin.bias is not None else None)
        q.N, q.K = N, K
        q.act1, q.act2, q.residual, q.carry = int(bool(u.get("act1"))), int(bool(u.get("act2"))), int(bool(u.get("residual"))), int(bool(u.get("carry")))
        q.eps = 1e-5
        if norm is not None:
            o["y"] = torch.empty((B, N), device=dev, dtype=torch.float32)
            o["mean"], o["rstd"] = torch.empty((B,), device=dev, dtype=torch.float32), torch.empty((B,), device=dev, dtype=torch.float32)
            q.gamma, q.beta, q.eps = norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps)
            q.y, q.mean, q.rstd = o["y"].data_ptr(), o["mean"].data_ptr(), o["rstd"].data_ptr()
        q.lin = o["lin"].data_ptr()
        for k, (fs, ft, fc) in (("drop1", ("drop_site1", "drop_threshold1", "drop_scale1")), ("drop2", ("drop_site2", "drop_threshold2", "drop_scale2"))):
            d = u.get(k)
            if d is not None and d[0] is not None:
                spec, scale = d
                if not isinstance(spec, DropSpec):
                    raise RuntimeError("predictor_fwd: dropout is decided inside the launch (a DropSpec), not from a keep-mask tensor")
                setattr(q, fs, spec.site), setattr(q, ft, spec.threshold), setattr(q, fc, float(scale))
                state = spec.state
            else:
                setattr(q, fc, 1.0)
        outs.append(o)
    _check(lib().hidvae_predictor_fwd(_p(h), _row_stride(h, "h"), B, ctypes.cast(arr, _vp), len(units), _p(state), _stream()), "hidvae_predictor_fwd")
    return outs


def predictor_bwd(g_out, units, outs, E):
    """the input-gradient chain of predictor_fwd's units in one launch.  units / outs: as given to / returned by predictor_fwd.
    -> (g_h [B, E], per unit dict(g_lin=[B, N], partials=[ceil(B/4), 2, N] or None))"""
    _f32(g_out, "g_out")
    B = g_out.shape[0]
    dev = g_out.device
    arr = (PredUnit * len(units))()
    res = []
    for q, u, o in zip(arr, units, outs):
        lin, norm = u["lin"], u.get("norm")
        N, K = lin.out_features, lin.in_features
        r = dict(g_lin=torch.empty((B, N), device=dev, dtype=torch.float32), partials=None)
        q.W, q.N, q.K = lin.weight.data_ptr(), N, K
        q.act1, q.act2, q.residual, q.carry = int(bool(u.get("act1"))), int(bool(u.get("act2"))), int(bool(u.get("residual"))), int(bool(u.get("carry")))
        q.drop_scale1 = float(u["drop1"][1]) if u.get("drop1") is not None else 1.0
        q.drop_scale2 = float(u["drop2"][1]) if u.get("drop2") is not None else 1.0
        q.lin, q.g_lin = o["lin"].data_ptr(), r["g_lin"].data_ptr()
        if norm is not None:
            r["partials"] = torch.empty(((B + 3) // 4, 2, N), device=dev, dtype=torch.float32)
            q.gamma, q.y, q.mean, q.rstd, q.partials = norm.weight.data_ptr(), o["y"].data_ptr(), o["mean"].data_ptr(), o["rstd"].data_ptr(), r["partials"].data_ptr()
        res.append(r)
    g_h = torch.empty((B, E), device=dev, dtype=torch.float32)
    _check(lib().hidvae_predictor_bwd(_p(g_out), _row_stride(g_out, "g_out"), B, ctypes.cast(arr, _vp), len(units), _p(g_h), E, _stream()),
           "hidvae_predictor_bwd")
    return g_h, res


def gate_fwd(x, W0, b0, W2, b2, W4, b4, normalize, eps=1e-12):
    """TagPredictor's attention gate in one launch -> (h, saved = (a1, pre2, a2, a3, nrm))"""
    _f32(x, "x")
    B, E = x.shape
    dev = x.device
    f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    a1, pre2, a2, a3, h = f(B, E // 4), f(B, E // 2), f(B, E // 2), f(B, E), f(B, E)
    nrm = f(B) if normalize else None
    for t in (W0, b0, W2, b2, W4, b4):
        if not t.is_contiguous():
            raise RuntimeError("gate_fwd: parameters must be contiguous")
    if tuple(W0.shape) != (E // 4, E) or tuple(W2.shape) != (E // 2, E // 4) or tuple(W4.shape) != (E, E // 2):
        raise RuntimeError(f"gate_fwd: weight shapes {tuple(W0.shape)} {tuple(W2.shape)} {tuple(W4.shape)} do not fit E={E}")
    _check(lib().hidvae_gate_fwd(_p(x), _row_stride(x, "x"), B, E, _p(W0), _p(b0), _p(W2), _p(b2), _p(W4), _p(b4), int(bool(normalize)),
                                 float(eps), _p(a1), _p(pre2), _p(a2), _p(a3), _p(h), _p(nrm), _stream()), "hidvae_gate_fwd")
    return h, (a1, pre2, a2, a3, nrm)


def gate_bwd(gh, x, W0, W2, W4, normalize, saved, eps=1e-12):
    """-> (gx through both uses of x, g3, g2, g1: the gradients at the three pre-activations)"""
    a1, pre2, _a2, a3, nrm = saved
    B, E = x.shape
    dev = x.device
    f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    gx, g3, g2, g1 = f(B, E), f(B, E), f(B, E // 2), f(B, E // 4)
    _check(lib().hidvae_gate_bwd(_p(gh), _row_stride(gh, "gh"), _p(x), _row_stride(x, "x"), B, E, _p(W0), _p(W2), _p(W4), int(bool(normalize)),
                                 float(eps), _p(a1), _p(pre2), _p(a3), _p(nrm), _p(gx), _p(g3), _p(g2), _p(g1), _stream()), "hidvae_gate_bwd")
    return gx, g3, g2, g1


def batchnorm_fwd(x, gamma, beta, eps, momentum, training, running_mean, running_var, relu, mask, mask_scale, num_batches=None):
    M, N = x.shape
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    sm = torch.empty((N,), device=x.device, dtype=torch.float32) if training else None
    sr = torch.empty((N,), device=x.device, dtype=torch.float32) if training else None
    ws = _ws(WS_BATCHNORM_FWD, x.device, M, N) if training else None
    mp, rs, site, thr = _mask_args(mask)
    _check(lib().hidvae_batchnorm_fwd(_p(x), _row_stride(x, "x"), M, N, _p(gamma), _p(beta), float(eps), float(momentum), int(training),
                                      _p(running_mean), _p(running_var), _p(num_batches), _p(y), _p(sm), _p(sr), int(relu), mp,
                                      float(mask_scale), rs, site, thr, _p(ws), _stream()), "hidvae_batchnorm_fwd")
    return y, sm, sr


def batchnorm_bwd(gy, x, gamma, beta, save_mean, save_rstd, relu, mask, mask_scale, need_gx=True, y_out=None):
    """y_out (instead of mask): the forward output, off which the ReLU -> Dropout gate is read"""
    M, N = x.shape
    gx = torch.empty((M, N), device=x.device, dtype=torch.float32) if need_gx else None
    gg = torch.empty((N,), device=x.device, dtype=torch.float32)
    gb = torch.empty((N,), device=x.device, dtype=torch.float32)
    ws = _ws(WS_BATCHNORM_BWD, x.device, M, N)
    _check(lib().hidvae_batchnorm_bwd(_p(gy), _p(x), _row_stride(x, "x"), _p(gamma), _p(beta), _p(save_mean), _p(save_rstd), M, N,
                                      int(relu), _p(mask if y_out is None else None), float(mask_scale), _p(y_out), _p(gx), _p(gg), _p(gb), 0,
                                      _p(ws), _stream()), "hidvae_batchnorm_bwd")
    return gx, gg, gb


def infonce_rows(S, tau, scale):
    B = S.shape[0]
    row_loss = torch.empty((B,), device=S.device, dtype=torch.float32)
    loss = torch.empty((), device=S.device, dtype=torch.float32)
    _check(lib().hidvae_infonce_rows(_p(S), B, float(tau), float(scale), _p(row_loss), _p(loss), _stream()), "hidvae_infonce_rows")
    return loss


def infonce_dlogits(P, tau, scale, g):
    _check(lib().hidvae_infonce_dlogits(_p(P), P.shape[0], float(tau), float(scale), _p(g), _stream()), "hidvae_infonce_dlogits")
    return P


def infonce_lse_chunk(Sc, col0, tau, m, l, diag, first):
    B, C = Sc.shape
    _check(lib().hidvae_infonce_lse_chunk(_p(Sc), B, C, _row_stride(Sc, "Sc"), int(col0), float(tau), _p(m), _p(l), _p(diag), int(bool(first)),
                                          _stream()), "hidvae_infonce_lse_chunk")


def infonce_lse_finish(m, l, diag, tau, scale):
    B = m.shape[0]
    row_loss = torch.empty((B,), device=m.device, dtype=torch.float32)
    lse = torch.empty((B,), device=m.device, dtype=torch.float32)
    loss = torch.empty((), device=m.device, dtype=torch.float32)
    _check(lib().hidvae_infonce_lse_finish(_p(m), _p(l), _p(diag), B, float(tau), float(scale), _p(row_loss), _p(lse), _p(loss), _stream()),
           "hidvae_infonce_lse_finish")
    return loss, lse


def infonce_dlogits_chunk(Sc, col0, tau, scale, lse, g):
    B, C = Sc.shape
    _check(lib().hidvae_infonce_dlogits_chunk(_p(Sc), B, C, _row_stride(Sc, "Sc"), int(col0), float(tau), float(scale), _p(lse), _p(g), _stream()),
           "hidvae_infonce_dlogits_chunk")
    return Sc


MIXUP_PLAN_MAX_B = 16384


def mixup_plan(targets, uniforms, alpha, rng_state=None):
    """targets [B, L] int64, uniforms [L, B+64] (or None with rng_state: the kernel draws them from the counter-based generator)
    -> partner [L,B], inverse [L,B] (int64), lam [L]: one launch for all levels"""
    B, L = targets.shape
    if targets.dtype != torch.int64 or targets.stride(1) != 1:
        raise RuntimeError("mixup_plan: expected int64 targets [B,L] with contiguous rows")
    if uniforms is not None:
        if tuple(uniforms.shape) != (L, B + 64) or not uniforms.is_contiguous():
            raise RuntimeError("mixup_plan: expected float32 uniforms [L,B+64]")
        _f32(uniforms, "uniforms")
    elif rng_state is None:
        raise RuntimeError("mixup_plan: uniforms or rng_state")
    dev = targets.device
    partner = torch.empty((L, B), device=dev, dtype=torch.int64)
    inverse = torch.empty((L, B), device=dev, dtype=torch.int64)
    lam = torch.empty((L,), device=dev, dtype=torch.float32)
    _check(lib().hidvae_mixup_plan(_p(targets), B, L, targets.stride(0) if B > 1 else L, _p(uniforms), float(alpha), _p(partner), _p(inverse),
                                   _p(lam), _p(rng_state if uniforms is None else None), _stream()), "hidvae_mixup_plan")
    return partner, inverse, lam


def tag_loss_fwd(logits, target, partner, lam, focal, gamma, alpha, smooth, ce_ls, want_grad):
    B, C = logits.shape
    dev = logits.device
    f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    loss, acc, nv = f(), f(), f()
    row_loss, row_hit, zbuf = f(B), f(B), f(B, C)
    dmix = f(B, C) if want_grad else None
    dkl = f(B, C) if (want_grad and not focal) else None
    _check(lib().hidvae_tag_loss_fwd(_p(logits), B, C, _p(target), _p(partner), _p(lam), int(bool(focal)), float(gamma), float(alpha),
                                     float(smooth), float(ce_ls), _p(loss), _p(acc), _p(nv), _p(row_loss), _p(row_hit), _p(zbuf),
                                     _p(dmix), _p(dkl), _stream()), "hidvae_tag_loss_fwd")
    return loss, acc, nv, dmix, dkl


def tag_loss_bwd(dmix, dkl, target, inverse, lam, g, n_valid):
    B, C = dmix.shape
    out = torch.empty((B, C), device=dmix.device, dtype=torch.float32)
    _check(lib().hidvae_tag_loss_bwd(_p(dmix), _p(dkl), _p(target), _p(inverse), _p(lam), B, C, _p(g), _p(n_valid), _p(out), _stream()),
           "hidvae_tag_loss_bwd")
    return out


def kmeans_iter(x, centroids, assign, reseed_idx, new_centroids, shift_scratch, shift):
    _check(lib().hidvae_kmeans_iter(_p(x), x.shape[0], _p(centroids), centroids.shape[0], _p(assign), _p(reseed_idx), _p(new_centroids),
                                    _p(shift_scratch), _p(shift), int(x.shape[1]), _stream()), "hidvae_kmeans_iter")


# ------------------------------------------------------------------------------------------------ gumbel branch
def gumbel_rows_fwd(S, x, cc, U, temperature, cosine=False):
    B, K = S.shape
    ids = torch.empty((B,), device=S.device, dtype=torch.int64)
    _check(lib().hidvae_gumbel_rows_fwd(_p(S), _p(x), _p(cc), _p(U), B, K, float(temperature), _p(ids), int(x.shape[1]), int(bool(cosine)),
                                        _stream()), "hidvae_gumbel_rows_fwd")
    return ids


def gumbel_loss(x, emb, beta):
    loss = torch.empty((x.shape[0],), device=x.device, dtype=torch.float32)
    _check(lib().hidvae_gumbel_loss(_p(x), _p(emb), x.shape[0], float(beta), _p(loss), int(x.shape[1]), _stream()), "hidvae_gumbel_loss")
    return loss


def gumbel_gemb(g_out, g_l, x, emb):
    out = torch.empty_like(x)
    _check(lib().hidvae_gumbel_gemb(_p(g_out), _p(g_l), _vec_stride(g_l), _p(x), _p(emb), x.shape[0], _p(out), int(x.shape[1]), _stream()),
           "hidvae_gumbel_gemb")
    return out


def gumbel_rows_bwd(P, gP, temperature):
    B, K = P.shape
    g_xx = torch.empty((B,), device=P.device, dtype=torch.float32)
    _check(lib().hidvae_gumbel_rows_bwd(_p(P), _p(gP), B, K, float(temperature), _p(g_xx), _stream()), "hidvae_gumbel_rows_bwd")
    return g_xx


def gumbel_finish(g_x, x, emb, g_xx, g_l, beta, g_cb, cb, gS_colsum):
    _check(lib().hidvae_gumbel_finish(_p(g_x), _p(x), _p(emb), _p(g_xx), _p(g_l), _vec_stride(g_l), float(beta), x.shape[0], _p(g_cb), _p(cb),
                                      _p(gS_colsum), cb.shape[0], int(x.shape[1]), _stream()), "hidvae_gumbel_finish")


# ------------------------------------------------------------------------------------------------ grouped launches
class LinearBwdProblem(ctypes.Structure):  # hidvae_linear_bwd_problem
    _fields_ = [("g", _vp), ("ldg", _i64), ("x", _vp), ("ldx", _i64), ("W", _vp), ("ldw", _i64), ("B", _i64), ("n_out", _i64),
                ("n_in", _i64), ("dW", _vp), ("lddw", _i64), ("accumulate_dw", _i), ("dX", _vp), ("lddx", _i64), ("dx_epilogue", _i),
                ("aux", _vp), ("ldaux", _i64), ("db", _vp), ("accumulate_db", _i), ("workspace", _vp)]


def _dp(t):
    return t.data_ptr() if t is not None else None


def linear_bwd_group(problems):
    """problems: list of dicts with the arguments of linear_bwd() (g, x, w, need_dx, epilogue, aux, dW, accumulate, bias, db,
    accumulate_db); ONE launch for every dW, dX and db of the group.  -> list of (dW, dX or None, db or None)"""
    arr = (LinearBwdProblem * len(problems))()
    outs, keep = [], []
    for q, pr in zip(arr, problems):
        g, x, w = pr["g"], pr["x"], pr["w"]
        _f32(g, "g"), _f32(x, "x")
        B, n_out = g.shape
        n_in = x.shape[1]
        need_dx = bool(pr.get("need_dx", True))
        if x.shape[0] != B or (need_dx and tuple(w.shape) != (n_out, n_in)):
            raise RuntimeError(f"linear_bwd_group: shapes g {tuple(g.shape)} x {tuple(x.shape)} W {tuple(w.shape)}")
        dW, acc = pr.get("dW"), bool(pr.get("accumulate", False))
        if dW is None:
            dW, acc = torch.empty((n_out, n_in), device=g.device, dtype=torch.float32), False
        dX = torch.empty((B, n_in), device=g.device, dtype=torch.float32) if need_dx else None
        db, accb = (pr.get("db"), bool(pr.get("accumulate_db", False))) if pr.get("bias") else (None, False)
        if pr.get("bias") and db is None:
            db, accb = torch.empty((n_out,), device=g.device, dtype=torch.float32), False
        aux = pr.get("aux")
        ws = _ws(WS_LINEAR_BWD, g.device, B, n_out, n_in, int(db is not None))
        q.g, q.ldg, q.x, q.ldx = g.data_ptr(), _row_stride(g, "g"), x.data_ptr(), _row_stride(x, "x")
        q.W, q.ldw = (w.data_ptr(), _row_stride(w, "W")) if need_dx else (None, 0)
        q.B, q.n_out, q.n_in = B, n_out, n_in
        q.dW, q.lddw, q.accumulate_dw = dW.data_ptr(), n_in, int(acc)
        q.dX, q.lddx, q.dx_epilogue = _dp(dX), n_in, int(pr.get("epilogue", EPI_NONE))
        q.aux, q.ldaux = _dp(aux), (_row_stride(aux, "aux") if aux is not None else 0)
        q.db, q.accumulate_db, q.workspace = _dp(db), int(accb), _dp(ws)
        outs.append((dW, dX, db))
        keep.append(ws)
    _check(lib().hidvae_linear_bwd_group(ctypes.cast(arr, _vp), len(problems), _stream()), "hidvae_linear_bwd_group")
    return outs


# ------------------------------------------------------------------------------------------------ stand-alone loss / sampling modules
def sqdiff_rows(a, b, extra=0.0):
    """s = sum_j (a-b)^2 over the last dim; out[m] = s + extra*s (2-D views with a contiguous last dim)"""
    _f32(a, "a"), _f32(b, "b")
    if a.shape != b.shape:
        raise RuntimeError(f"sqdiff_rows: shapes differ ({tuple(a.shape)} vs {tuple(b.shape)})")
    M, N = a.shape
    out = torch.empty((M,), device=a.device, dtype=torch.float32)
    _check(lib().hidvae_sqdiff_rows(_p(a), _row_stride(a, "a"), _p(b), _row_stride(b, "b"), M, N, float(extra), _p(out), _stream()),
           "hidvae_sqdiff_rows")
    return out


def sqdiff_rows_bwd(g, a, b, scale_a, scale_b, want_a=True, want_b=True):
    M, N = a.shape
    ga = torch.empty((M, N), device=a.device, dtype=torch.float32) if want_a else None
    gb = torch.empty((M, N), device=a.device, dtype=torch.float32) if want_b else None
    _check(lib().hidvae_sqdiff_rows_bwd(_p(g), _vec_stride(g), _p(a), _row_stride(a, "a"), _p(b), _row_stride(b, "b"), M, N, float(scale_a),
                                        float(scale_b), _p(ga), _p(gb), _stream()), "hidvae_sqdiff_rows_bwd")
    return ga, gb


def softmax_argmax_rows(logits, pred, conf, col):
    """logits [B, C]; pred int64 [B, L], conf float32 [B, L]: column `col` of both receives the row's first arg max and its softmax
    probability (HRqVae.predict_tags, one launch per level, no [B, C] softmax intermediate)"""
    _f32(logits, "logits")
    B, C = logits.shape
    if pred.dtype != torch.int64 or conf.dtype != torch.float32 or pred.shape != conf.shape or pred.shape[0] != B or not pred.is_contiguous() \
            or not conf.is_contiguous():
        raise RuntimeError("softmax_argmax_rows: expected contiguous pred int64 [B, L] and conf float32 [B, L]")
    L = pred.shape[1]
    _check(lib().hidvae_softmax_argmax_rows(_p(logits), B, C, _row_stride(logits, "logits"), ctypes.c_void_p(pred.data_ptr() + 8 * col), L,
                                            ctypes.c_void_p(conf.data_ptr() + 4 * col), L, _stream()), "hidvae_softmax_argmax_rows")


def gumbel_noise(U, eps=1e-20):
    _f32(U, "U")
    U = U.contiguous()
    out = torch.empty_like(U)
    _check(lib().hidvae_gumbel_noise(_p(U), U.numel(), float(eps), _p(out), _stream()), "hidvae_gumbel_noise")
    return out


def gumbel_softmax_rows(logits, U, temperature):
    _f32(logits, "logits"), _f32(U, "U")
    if logits.shape != U.shape or not logits.is_contiguous() or not U.is_contiguous():
        raise RuntimeError("gumbel_softmax_rows: logits and U must be contiguous and of one shape")
    K = logits.shape[-1]
    out = torch.empty_like(logits)
    _check(lib().hidvae_gumbel_softmax_rows(_p(logits), _p(U), logits.numel() // K, K, float(temperature), _p(out), _stream()),
           "hidvae_gumbel_softmax_rows")
    return out


# ------------------------------------------------------------------------------------------------ jagged copy (stage-2 op)
def padded_to_jagged(x, offsets, total):
    """x [B,N,D] (last dim contiguous, any dtype), offsets [B+1] int64 device -> values [total, D]"""
    B, N, D = x.shape
    if not x.is_cuda or x.stride(2) != 1:
        raise RuntimeError("padded_to_jagged: expected a device tensor [B,N,D] with a contiguous last dim")
    es = x.element_size()
    values = torch.empty((total, D), device=x.device, dtype=x.dtype)
    _check(lib().hidvae_padded_to_jagged(_p(x), x.stride(0) * es, x.stride(1) * es, _p(offsets), ctypes.c_void_p(values.data_ptr()), B, N,
                                         D * es, _stream()), "hidvae_padded_to_jagged")
    return values


def jagged_to_padded(values, offsets, B, N):
    D = values.shape[1]
    es = values.element_size()
    x = torch.empty((B, N, D), device=values.device, dtype=values.dtype)
    _check(lib().hidvae_jagged_to_padded(ctypes.c_void_p(values.data_ptr()), _p(offsets), _p(x), N * D * es, D * es, B, N, D * es, _stream()),
           "hidvae_jagged_to_padded")
    return x


GATHER_MAX = 4


def gather_rows(idx, tables, outs):
    """outs[t][r] = tables[t][idx[r]] for every table, ONE launch (hidvae_gather_rows): a batch gathered from the resident item
    tables straight into its destination (the step's input buffers).  idx: int64 device vector; tables / outs: contiguous device
    tensors of matching dtype and row shape, outs[t] with len(idx) rows; element sizes in whole dwords per row."""
    n = len(tables)
    if not (1 <= n <= GATHER_MAX and len(outs) == n):
        raise RuntimeError(f"gather_rows: {n} tables / {len(outs)} destinations (1 .. {GATHER_MAX}, equally many)")
    if idx.dtype != torch.int64 or idx.dim() != 1 or not idx.is_contiguous() or not idx.is_cuda:
        raise RuntimeError("gather_rows: idx must be a contiguous int64 device vector")
    rows = idx.numel()
    rb, sr = [], []
    for t, o in zip(tables, outs):
        if not (t.is_cuda and o.is_cuda and t.is_contiguous() and o.is_contiguous() and t.dtype == o.dtype and t.shape[1:] == o.shape[1:]
                and o.shape[0] == rows and t.dim() >= 1):
            raise RuntimeError(f"gather_rows: table {tuple(t.shape)} {t.dtype} -> destination {tuple(o.shape)} {o.dtype} for {rows} rows")
        b = (t[0].numel() if t.dim() > 1 else 1) * t.element_size()
        if b % 4:
            raise RuntimeError(f"gather_rows: rows of {b} bytes (whole dwords only)")
        rb.append(b)
        sr.append(t.shape[0])
    vp = ctypes.c_void_p * n
    i64 = ctypes.c_int64 * n
    _check(lib().hidvae_gather_rows(_p(idx), rows, n, vp(*[t.data_ptr() for t in tables]), vp(*[o.data_ptr() for o in outs]), i64(*rb), i64(*sr),
                                    _stream()), "hidvae_gather_rows")
    return outs
