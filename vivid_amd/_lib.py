"""ctypes binding of libvivid_hip.so (the C ABI in include/vivid_hip.h).

The structures below mirror the header field for field.  There is no CPU
fallback: if the shared library is missing, :func:`lib` raises with the build
command, and every op raises :class:`VividHipError` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIVID_HIP_LIB: load another build of the same ABI instead (diagnostic builds: `make stamp`, A/B of compile-time knobs)
LIB_PATH = os.environ.get("VIVID_HIP_LIB") or os.path.join(_HERE, "libvivid_hip.so")

c_float_p = C.POINTER(C.c_float)
ABI_VERSION = 5          # VH_ABI_VERSION of include/vivid_hip.h these structures mirror


class VividHipError(RuntimeError):
    pass


class PrepWeightArgs(C.Structure):
    _fields_ = [("w", C.c_void_p), ("cout", C.c_int), ("cin", C.c_int), ("taps", C.c_int),
                ("cin_pad", C.c_int), ("k_pad", C.c_int), ("gain_ptr", C.c_void_p),
                ("gain_value", C.c_float), ("wt", C.c_void_p), ("dst_col0", C.c_int), ("dst_cols", C.c_int),
                ("split", C.c_int), ("k_off", C.c_int), ("k_stride", C.c_int)]


class QkvEpilogue(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("heads", C.c_int), ("nj", C.c_int),
                ("rows_per_b", C.c_int), ("koff", C.c_int), ("kl", C.c_int), ("qscale", C.c_float)]


class S8Sink(C.Structure):
    """vh_s8_sink"""
    _fields_ = [("ptr", C.c_void_p), ("c_total", C.c_int), ("c_off", C.c_int), ("scale", C.c_float), ("silu", C.c_int)]


class ConvArgs(C.Structure):
    _fields_ = [("src0", C.c_void_p), ("src1", C.c_void_p), ("c0", C.c_int), ("c1", C.c_int),
                ("scale0", C.c_float), ("scale1", C.c_float),
                ("rows", C.c_int), ("h", C.c_int), ("w", C.c_int), ("up", C.c_int), ("taps", C.c_int),
                ("pro", C.c_int), ("wt", C.c_void_p), ("cin_pad", C.c_int), ("k_pad", C.c_int),
                ("zeros", C.c_void_p), ("zeros_bytes", C.c_size_t), ("cout", C.c_int),
                ("scratch", C.c_void_p), ("scratch_floats", C.c_size_t), ("out", C.c_void_p), ("out_s8", C.c_void_p), ("out_s8_c", C.c_int),
                ("prec", C.c_int), ("kernel", C.c_int), ("epi", C.c_int),
                ("cvec", C.c_void_p), ("cvec_ld", C.c_int), ("res", C.c_void_p), ("res_up", C.c_int), ("res_scale", C.c_void_p),
                ("ta", C.c_float), ("tb", C.c_float), ("clip", C.c_float), ("qkv", C.c_void_p), ("stagger", C.c_int),
                ("korder", C.c_int), ("tile", C.c_int), ("sink", S8Sink * 2),
                ("tail_f32", C.c_int), ("src2", C.c_void_p), ("c2", C.c_int), ("scale2", C.c_float), ("src_f32", C.c_int)]


class PixnormArgs(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int), ("h", C.c_int),
                ("w", C.c_int), ("c", C.c_int), ("pool", C.c_int), ("norm", C.c_int), ("out_s8", C.c_void_p), ("scale_out", C.c_void_p)]


class SplitArgs(C.Structure):
    _fields_ = [("src0", C.c_void_p), ("src1", C.c_void_p), ("c0", C.c_int), ("c1", C.c_int),
                ("scale0", C.c_float), ("scale1", C.c_float), ("pro", C.c_int), ("npix", C.c_longlong),
                ("c_pad", C.c_int), ("out", C.c_void_p), ("out_raw", C.c_void_p), ("out_c_total", C.c_int), ("out_c_off", C.c_int)]


class QkvSplitArgs(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("rows", C.c_int), ("s", C.c_int), ("heads", C.c_int),
                ("d", C.c_int), ("nj", C.c_int), ("rows_per_b", C.c_int), ("koff", C.c_int),
                ("kl", C.c_int), ("qscale", C.c_float), ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p)]


class AttentionArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("b", C.c_int),
                ("heads", C.c_int), ("s", C.c_int), ("kl", C.c_int), ("d", C.c_int),
                ("n_zero_keys", C.c_float), ("out", C.c_void_p), ("out_s8", C.c_int), ("logit_bound", C.c_float)]


class EmbedArgs(C.Structure):
    _fields_ = [("sigma", C.c_void_p), ("sigma_stride", C.c_int), ("time_scale", C.c_float),
                ("geometry", C.c_void_p), ("label_dim", C.c_int), ("geometry_scale", C.c_float),
                ("freqs", C.c_void_p), ("phases", C.c_void_p), ("cnoise", C.c_int),
                ("w_noise", C.c_void_p), ("w_noise_kpad", C.c_int),
                ("w_label", C.c_void_p), ("w_label_kpad", C.c_int),
                ("label_balance", C.c_float), ("rows", C.c_int), ("cemb", C.c_int), ("raw", C.c_int),
                ("emb", C.c_void_p)]


class LinearArgs(C.Structure):
    _fields_ = [("emb", C.c_void_p), ("rows", C.c_int), ("cemb", C.c_int), ("wt", C.c_void_p),
                ("k_pad", C.c_int), ("cols", C.c_int), ("bias", C.c_float), ("out", C.c_void_p)]


class Segment(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("kind", C.c_int), ("c", C.c_int), ("c_src", C.c_int),
                ("row_mul", C.c_int), ("scale_cin", C.c_int)]


class AssembleArgs(C.Structure):
    _fields_ = [("seg", Segment * 4), ("nseg", C.c_int), ("sigma", C.c_void_p), ("sigma_data", C.c_float),
                ("rows", C.c_int), ("h", C.c_int), ("w", C.c_int), ("c_pad", C.c_int), ("out", C.c_void_p)]


class PrecondOutArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("row_mul", C.c_int), ("f", C.c_void_p), ("fc", C.c_int),
                ("sigma", C.c_void_p), ("sigma_data", C.c_float), ("rows", C.c_int), ("c", C.c_int),
                ("h", C.c_int), ("w", C.c_int), ("out", C.c_void_p)]


class WarpArgs(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("src_c", C.c_int), ("depth_ch", C.c_int), ("geometry", C.c_void_p),
                ("mean", C.c_float * 20), ("std", C.c_float * 20), ("freqs", C.c_void_p), ("phases", C.c_void_p),
                ("rows", C.c_int), ("s", C.c_int), ("grid_feat", C.c_void_p), ("warp_feat", C.c_void_p),
                ("nonzero_flag", C.c_void_p), ("uv_out", C.c_void_p)]


class NonzeroArgs(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("rows", C.c_int), ("c_used", C.c_int), ("c_total", C.c_int), ("hw", C.c_int),
                ("flag", C.c_void_p)]


class ResampleArgs(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int), ("h", C.c_int), ("w", C.c_int), ("c", C.c_int),
                ("up", C.c_int), ("ntaps", C.c_int), ("taps", C.c_float * 8)]


class MomentsArgs(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("n", C.c_int), ("fa", C.c_int), ("fb", C.c_int),
                ("outer", C.c_void_p), ("sum_a", C.c_void_p)]


class PsnrArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("images", C.c_int), ("elems", C.c_size_t), ("dtype", C.c_int),
                ("acc", C.c_void_p), ("per_image", C.c_void_p)]


class CodecArgs(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("n", C.c_size_t), ("decode", C.c_int)]


class AddDepthArgs(C.Structure):
    _fields_ = [("src", C.c_void_p), ("c", C.c_int), ("depth", C.c_void_p), ("rows", C.c_int), ("h", C.c_int),
                ("w", C.c_int), ("inv_norm", C.c_int), ("out", C.c_void_p)]


class ResizeArgs(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("planes", C.c_int), ("hin", C.c_int), ("win", C.c_int),
                ("hout", C.c_int), ("wout", C.c_int), ("antialias", C.c_int), ("mode", C.c_int), ("align_corners", C.c_int),
                ("ch_scale", C.c_void_p), ("ch_bias", C.c_void_p), ("channels", C.c_int)]


class LayoutArgs(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int), ("c", C.c_int), ("hw", C.c_int), ("to_nchw", C.c_int)]


class AxpyArgs(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("s", C.c_float), ("out", C.c_void_p), ("n", C.c_size_t)]


class SamplerStepArgs(C.Structure):
    _fields_ = [("x_hat", C.c_void_p), ("x_probe", C.c_void_p), ("d_cond", C.c_void_p), ("d_ref", C.c_void_p),
                ("guidance", C.c_float), ("d_cur", C.c_void_p), ("t_hat", C.c_float), ("t_next", C.c_float),
                ("rows", C.c_int), ("row_mul", C.c_int), ("row_elems", C.c_size_t), ("x_next", C.c_void_p)]


class NetConfigC(C.Structure):
    """vh_net_config"""
    _fields_ = [("img_resolution", C.c_int), ("img_channels", C.c_int), ("source_label_dim", C.c_int), ("target_label_dim", C.c_int),
                ("model_channels", C.c_int), ("channel_mult", C.c_int * 8), ("num_levels", C.c_int), ("num_blocks", C.c_int),
                ("attn_resolutions", C.c_int * 8), ("num_attn_resolutions", C.c_int), ("extra_attn", C.c_int),
                ("channel_mult_noise", C.c_int), ("channel_mult_emb", C.c_int),
                ("label_balance", C.c_double), ("concat_balance", C.c_double), ("res_balance", C.c_double), ("attn_balance", C.c_double),
                ("clip_act", C.c_double), ("sigma_data", C.c_double), ("logvar_channels", C.c_int),
                ("super_res", C.c_int), ("no_time_enc", C.c_int), ("depth_input", C.c_int), ("warp_depth_coor", C.c_int), ("uncond", C.c_int),
                ("dual_source", C.c_int), ("geom_mean", C.c_float * 20), ("geom_std", C.c_float * 20), ("noisy_sr", C.c_double),
                ("resample_ntaps", C.c_int), ("resample_filter", C.c_float * 8), ("fp32", C.c_int)]


RANDN_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)        # vh_randn_fn(user, device_dst, n, stream)


class SamplerConfigC(C.Structure):
    """vh_sampler_config"""
    _fields_ = [("num_steps", C.c_int), ("sigma_min", C.c_double), ("sigma_max", C.c_double), ("rho", C.c_double), ("guidance", C.c_double),
                ("S_churn", C.c_double), ("S_min", C.c_double), ("S_max", C.c_double), ("S_noise", C.c_double), ("t_steps", C.POINTER(C.c_float)),
                ("randn", RANDN_FN), ("randn_user", C.c_void_p), ("guidance_overlap", C.c_int)]


# every symbol include/vivid_hip.h declares: name -> (args struct or None)
OPS = {
    "vh_prep_weight": PrepWeightArgs, "vh_conv": ConvArgs, "vh_pixnorm": PixnormArgs, "vh_split": SplitArgs,
    "vh_qkv_split": QkvSplitArgs, "vh_attention": AttentionArgs, "vh_embed": EmbedArgs,
    "vh_qkv_split_x3": QkvSplitArgs, "vh_attention_x3": AttentionArgs,
    "vh_linear": LinearArgs, "vh_assemble": AssembleArgs, "vh_precond_out": PrecondOutArgs,
    "vh_warp_features": WarpArgs, "vh_sampler_step": SamplerStepArgs, "vh_codec": CodecArgs, "vh_add_depth": AddDepthArgs,
    "vh_resize_bilinear": ResizeArgs, "vh_resize": ResizeArgs,
    "vh_nonzero_flag": NonzeroArgs, "vh_resample": ResampleArgs, "vh_moments": MomentsArgs, "vh_psnr_sum": PsnrArgs,
    "vh_layout": LayoutArgs, "vh_axpy": AxpyArgs,
}
TAGS = ["conv3x3", "conv1x1", "attention", "pixnorm", "qkv_split", "embed", "assemble", "sampler", "prep", "warp", "split"]
CONTROL = ["vh_abi_version", "vh_diag_flags", "vh_last_error", "vh_ctx_create", "vh_ctx_destroy", "vh_ctx_set_stream", "vh_set_knob", "vh_conv_takes_patch",
           "vh_profile_enable", "vh_profile_read", "vh_profile_read_list",
           "vh_plan_begin", "vh_plan_end", "vh_plan_abort", "vh_plan_capture_graph", "vh_plan_run", "vh_plan_num_ops", "vh_plan_destroy"]

NET = ["vh_net_create", "vh_net_destroy", "vh_net_num_params", "vh_net_param_info", "vh_net_bind_param", "vh_net_prepared_bytes",
       "vh_net_prepare", "vh_net_workspace_bytes", "vh_net_record", "vh_net_run",
       "vh_net_workspace_bytes_mode", "vh_net_record_mode", "vh_net_encode", "vh_net_run_bound",
       "vh_net_num_features", "vh_net_feature_shape", "vh_net_features", "vh_net_run_inject", "vh_net_logvar",
       "vh_edm_sampler_workspace_bytes", "vh_edm_sampler"]

_lib = None


def lib():
    """Load libvivid_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VividHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C vivid_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.vh_abi_version.restype = C.c_int
    if L.vh_abi_version() != ABI_VERSION:
        raise VividHipError(f"{LIB_PATH} was built from another version of include/vivid_hip.h (library ABI {L.vh_abi_version()}, "
                            f"bindings {ABI_VERSION}): rebuild it with `make -C vivid_amd/csrc`")
    L.vh_diag_flags.restype = C.c_int
    if L.vh_diag_flags() and not os.environ.get("VIVID_HIP_LIB") and os.path.basename(LIB_PATH) == "libvivid_hip.so":
        raise VividHipError(f"{LIB_PATH} is a DIAGNOSTIC build (vh_diag_flags = {L.vh_diag_flags()}: clock stamps or timing ablations "
                            f"that compute wrong results): rebuild the product library with `make -C vivid_amd/csrc clean all`")
    L.vh_last_error.restype = C.c_char_p
    L.vh_ctx_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.vh_ctx_destroy.argtypes = [C.c_void_p]
    L.vh_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.vh_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.vh_profile_read.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.vh_set_knob.argtypes = [C.c_char_p, C.c_int]
    L.vh_plan_begin.argtypes = [C.c_void_p]
    L.vh_plan_abort.argtypes = [C.c_void_p]
    L.vh_plan_end.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.vh_plan_run.argtypes = [C.c_void_p, C.c_void_p]
    L.vh_plan_capture_graph.argtypes = [C.c_void_p, C.c_void_p]
    L.vh_plan_num_ops.argtypes = [C.c_void_p]
    L.vh_plan_destroy.argtypes = [C.c_void_p]
    for name, st in OPS.items():
        fn = getattr(L, name)
        fn.argtypes = [C.c_void_p, C.POINTER(st)]
        fn.restype = C.c_int
    for name in CONTROL:
        getattr(L, name)
    L.vh_conv_takes_patch.argtypes = [C.POINTER(ConvArgs)]
    L.vh_conv_takes_patch.restype = C.c_int
    L.vh_net_create.argtypes = [C.c_void_p, C.POINTER(NetConfigC), C.POINTER(C.c_void_p)]
    L.vh_net_destroy.argtypes = [C.c_void_p]
    L.vh_net_num_params.argtypes = [C.c_void_p]
    L.vh_net_param_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.vh_net_bind_param.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    L.vh_net_prepared_bytes.argtypes = [C.c_void_p]
    L.vh_net_prepared_bytes.restype = C.c_size_t
    L.vh_net_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.vh_net_workspace_bytes.argtypes = [C.c_void_p, C.c_int]
    L.vh_net_workspace_bytes.restype = C.c_size_t
    L.vh_net_record.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.vh_net_run.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
    L.vh_net_workspace_bytes_mode.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.vh_net_workspace_bytes_mode.restype = C.c_size_t
    L.vh_net_record_mode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.vh_net_encode.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 3
    L.vh_net_run_bound.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
    L.vh_net_num_features.argtypes = [C.c_void_p]
    L.vh_net_feature_shape.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.vh_net_features.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.vh_net_run_inject.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.POINTER(C.c_void_p), C.c_void_p]
    L.vh_net_logvar.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.vh_edm_sampler_workspace_bytes.argtypes = [C.c_void_p, C.c_int]
    L.vh_edm_sampler_workspace_bytes.restype = C.c_size_t
    L.vh_edm_sampler.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SamplerConfigC), C.c_int] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]
    _lib = L
    return L


def set_knob(name: str, value: int):
    """Process-wide scheduling knob of the library (vh_set_knob): for A/B measurements, never needed for results."""
    check(lib().vh_set_knob(name.encode(), int(value)), "vh_set_knob")


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().vh_last_error()
        raise VividHipError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


class Context:
    """A vh_ctx bound to a HIP stream."""

    def __init__(self, stream_handle: int = 0):
        self._L = lib()
        h = C.c_void_p()
        check(self._L.vh_ctx_create(C.c_void_p(stream_handle), C.byref(h)), "vh_ctx_create")
        self.handle = h

    def set_stream(self, stream_handle: int):
        check(self._L.vh_ctx_set_stream(self.handle, C.c_void_p(stream_handle)), "vh_ctx_set_stream")

    def call(self, name: str, args):
        check(getattr(self._L, name)(self.handle, C.byref(args)), name)

    def profile_enable(self, on: bool):
        check(self._L.vh_profile_enable(self.handle, 1 if on else 0), "vh_profile_enable")

    def profile_read(self):
        """{family: dict(ms, flops, bytes, launches)} accumulated since profiling was enabled / last read."""
        n = len(TAGS)
        ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        ln = (C.c_longlong * n)()
        check(self._L.vh_profile_read(self.handle, n, ms, fl, by, ln), "vh_profile_read")
        return {TAGS[i]: dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=ln[i]) for i in range(n)}

    def profile_read_list(self, max_n: int = 1 << 16):
        """[(family, ms, flops, bytes)] per launch, in launch order; clears the records."""
        tags = (C.c_int * max_n)()
        ms, fl, by = (C.c_double * max_n)(), (C.c_double * max_n)(), (C.c_double * max_n)()
        n = C.c_int()
        self._L.vh_profile_read_list.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        check(self._L.vh_profile_read_list(self.handle, max_n, tags, ms, fl, by, C.byref(n)), "vh_profile_read_list")
        return [(TAGS[tags[i]], ms[i], fl[i], by[i]) for i in range(n.value)]

    def plan_begin(self):
        check(self._L.vh_plan_begin(self.handle), "vh_plan_begin")

    def plan_abort(self):
        check(self._L.vh_plan_abort(self.handle), "vh_plan_abort")

    def plan_end(self) -> "Plan":
        p = C.c_void_p()
        check(self._L.vh_plan_end(self.handle, C.byref(p)), "vh_plan_end")
        return Plan(self, p)

    def close(self):
        if getattr(self, "handle", None):
            self._L.vh_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    def __init__(self, ctx: Context, handle):
        self.ctx, self.handle = ctx, handle

    @property
    def num_ops(self) -> int:
        return self.ctx._L.vh_plan_num_ops(self.handle)

    graph_captured = False

    def capture_graph(self):
        check(self.ctx._L.vh_plan_capture_graph(self.ctx.handle, self.handle), "vh_plan_capture_graph")
        self.graph_captured = True

    def run(self):
        check(self.ctx._L.vh_plan_run(self.ctx.handle, self.handle), "vh_plan_run")

    def __del__(self):
        try:
            if self.handle:
                self.ctx._L.vh_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
