"""ctypes binding of libhibayes_gpu.so (include/hibayes_gpu.h).

The library is the product: if it is missing or no HIP device is usable, every compute entry
point raises — there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIBAYES_GPU_LIB") or os.path.join(_HERE, "libhibayes_gpu.so")  # (override: A/B builds of the library)
HB_MAX_FOLD = 8
_lib = None


class HibayesError(RuntimeError):
    """Raised with the library's hb_last_error() text (the reference's exception texts for
    argument validation, src/Bayes.cpp:92-117)."""

    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)
INTERRUPT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
LOG_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)


class BayesArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32),
        ("y", C.c_void_p),
        ("X_f64", C.c_void_p), ("ld_f64", C.c_int64),
        ("X_i8", C.c_void_p), ("ld_i8", C.c_int64),
        ("model", C.c_char_p),
        ("Pi", C.c_void_p), ("n_pi", C.c_int32),
        ("Kival", C.c_void_p), ("Ki", C.c_void_p),
        ("C", C.c_void_p), ("nc", C.c_int32),
        ("R", C.c_void_p), ("nr", C.c_int32),
        ("fold", C.c_void_p), ("n_fold", C.c_int32),
        ("niter", C.c_int32), ("nburn", C.c_int32), ("thin", C.c_int32),
        ("epsl_y_J", C.c_void_p), ("epsl_Gi", C.c_void_p), ("epsl_index", C.c_void_p),
        ("has_dfvr", C.c_int32), ("has_s2vr", C.c_int32), ("has_vg", C.c_int32), ("has_dfvg", C.c_int32),
        ("has_s2vg", C.c_int32), ("has_ve", C.c_int32), ("has_dfve", C.c_int32), ("has_s2ve", C.c_int32),
        ("dfvr", C.c_double), ("s2vr", C.c_double), ("vg", C.c_double), ("dfvg", C.c_double),
        ("s2vg", C.c_double), ("ve", C.c_double), ("dfve", C.c_double), ("s2ve", C.c_double),
        ("windindx", C.c_void_p),
        ("outfreq", C.c_int32), ("threads", C.c_int32), ("verbose", C.c_int32),
        ("seed", C.c_uint64),
        ("device", C.c_int32), ("panel", C.c_int32), ("precise", C.c_int32), ("store_alpha", C.c_int32),
        ("rank", C.c_int32), ("world", C.c_int32),
        ("m_global", C.c_int64), ("m_offset", C.c_int64),
        ("allreduce", ALLREDUCE_FN), ("allreduce_user", C.c_void_p),
        ("exchange_buf", C.c_void_p),
        ("interrupt", INTERRUPT_FN), ("interrupt_user", C.c_void_p),
        ("log", LOG_FN), ("log_user", C.c_void_p),
        ("ctx", C.c_void_p),
        ("g_init", C.c_void_p),
        ("comm", C.c_void_p),
        ("sync_blocks", C.c_int32),
        ("genotype_bits", C.c_int32),
        ("shard_rows", C.c_int32), ("n_global", C.c_int64), ("row_offset", C.c_int64),
        ("warm", C.c_void_p),
    ]


class WarmState(C.Structure):
    """hb_warm_state (ABI 6): the scalars a continued chain starts from; see include/hibayes_gpu.h."""
    _fields_ = [("mu", C.c_double), ("vare", C.c_double), ("varg", C.c_double), ("lambda2", C.c_double),
                ("pi", C.c_double * HB_MAX_FOLD), ("vargL", C.c_void_p)]

    @classmethod
    def make(cls, mu, vare, varg=0.0, pi=(), lambda2=0.0, vargL=None):
        import numpy as np
        w = cls()
        w.mu, w.vare, w.varg, w.lambda2 = float(mu), float(vare), float(varg), float(lambda2)
        for j, p in enumerate(pi):
            w.pi[j] = float(p)
        w._keep = None
        if vargL is not None:
            w._keep = np.ascontiguousarray(vargL, dtype=np.float64)
            w.vargL = w._keep.ctypes.data
        return w

    def as_dict(self, n_pi):
        return {"mu": self.mu, "vare": self.vare, "varg": self.varg, "lambda2": self.lambda2, "pi": [self.pi[j] for j in range(n_pi)]}

    @classmethod
    def from_info(cls, info, n_pi, vargL=None):
        """the state hb_run_state() reported (RunInfo) as the start of another run"""
        return cls.make(info.mu, info.vare, info.varg, [info.pi[j] for j in range(n_pi)], info.lambda2, vargL)


class BayesOut(C.Structure):
    _fields_ = [
        ("Vg", C.c_double), ("Ve", C.c_double), ("h2", C.c_double), ("mu", C.c_double),
        ("n_records", C.c_int32), ("nzct", C.c_int32), ("nw", C.c_int32), ("n_levels", C.c_int32),
        ("beta", C.c_void_p), ("alpha", C.c_void_p), ("pi", C.c_void_p), ("Vr", C.c_void_p),
        ("r_est", C.c_void_p), ("r_term_nlevels", C.c_void_p),
        ("g", C.c_void_p), ("e", C.c_void_p), ("pip", C.c_void_p), ("gwas", C.c_void_p),
        ("s_Vg", C.c_void_p), ("s_Ve", C.c_void_p), ("s_h2", C.c_void_p), ("s_mu", C.c_void_p),
        ("s_beta", C.c_void_p), ("s_alpha", C.c_void_p), ("s_pi", C.c_void_p), ("s_Vr", C.c_void_p),
        ("s_r", C.c_void_p),
        ("alpha_sd", C.c_void_p),
        ("setup_seconds", C.c_double), ("loop_seconds", C.c_double),
        ("iters_done", C.c_int32),
        ("mean_events", C.c_double),
        ("sweeps_replayed", C.c_int32), ("resident_bits", C.c_int32),
        ("last", WarmState), ("g_last", C.c_void_p), ("vargL_last", C.c_void_p),
    ]


class SBayesArgs(C.Structure):
    _fields_ = [
        ("m", C.c_int32), ("sumstat", C.c_void_p), ("ld_sumstat", C.c_int64), ("ldm", C.c_void_p), ("ld_ldm", C.c_int64),
        ("model", C.c_char_p), ("Pi", C.c_void_p), ("n_pi", C.c_int32),
        ("niter", C.c_int32), ("nburn", C.c_int32), ("thin", C.c_int32),
        ("fold", C.c_void_p), ("n_fold", C.c_int32), ("windindx", C.c_void_p),
        ("has_vg", C.c_int32), ("has_dfvg", C.c_int32), ("has_s2vg", C.c_int32), ("has_ve", C.c_int32), ("has_dfve", C.c_int32), ("has_s2ve", C.c_int32),
        ("vg", C.c_double), ("dfvg", C.c_double), ("s2vg", C.c_double), ("ve", C.c_double), ("dfve", C.c_double), ("s2ve", C.c_double),
        ("outfreq", C.c_int32), ("threads", C.c_int32), ("verbose", C.c_int32),
        ("seed", C.c_uint64), ("device", C.c_int32), ("store_alpha", C.c_int32),
        ("interrupt", INTERRUPT_FN), ("interrupt_user", C.c_void_p), ("log", LOG_FN), ("log_user", C.c_void_p),
    ]


class SBayesOut(C.Structure):
    _fields_ = [
        ("Vg", C.c_double), ("Ve", C.c_double), ("h2", C.c_double),
        ("n_records", C.c_int32), ("nzct", C.c_int32), ("nw", C.c_int32), ("n", C.c_int32), ("count_y", C.c_int32),
        ("alpha", C.c_void_p), ("pi", C.c_void_p), ("pip", C.c_void_p), ("gwas", C.c_void_p),
        ("s_Vg", C.c_void_p), ("s_Ve", C.c_void_p), ("s_h2", C.c_void_p), ("s_alpha", C.c_void_p), ("s_pi", C.c_void_p),
        ("r_hat", C.c_void_p), ("g_last", C.c_void_p),
        ("setup_seconds", C.c_double), ("loop_seconds", C.c_double), ("iters_done", C.c_int32), ("mean_events", C.c_double),
    ]


class RunInfo(C.Structure):
    _fields_ = [
        ("iter", C.c_int32), ("records", C.c_int32), ("nnz", C.c_double),
        ("vara", C.c_double), ("vare", C.c_double), ("varg", C.c_double), ("mu", C.c_double),
        ("pi", C.c_double * HB_MAX_FOLD), ("mean_events", C.c_double), ("mean_misses", C.c_double), ("mean_redo", C.c_double),
        ("loop_seconds", C.c_double), ("setup_seconds", C.c_double), ("gram_seconds", C.c_double),
        ("sweeps_replayed", C.c_int32), ("resident_bits", C.c_int32),
        ("lambda2", C.c_double),
    ]


class CtxParams(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("panel", C.c_int32),
        ("precise", C.c_int32), ("m_offset", C.c_int64), ("seed", C.c_uint64),
    ]


class SweepIn(C.Structure):
    _fields_ = [
        ("model_index", C.c_int32), ("n_fold", C.c_int32), ("iter", C.c_int64),
        ("vare", C.c_double), ("varg", C.c_double), ("s2varg_df", C.c_double), ("dfvara", C.c_double),
        ("logpi", C.c_double * HB_MAX_FOLD), ("fold", C.c_double * HB_MAX_FOLD),
        ("vara_fold", C.c_double * HB_MAX_FOLD),
        ("lambda_", C.c_double), ("lambda2", C.c_double),
        ("count_pip", C.c_int32), ("store", C.c_int32),
    ]


class SweepOut(C.Structure):
    _fields_ = [
        ("sum_g2", C.c_double), ("class_count", C.c_double * HB_MAX_FOLD), ("sum_vargL", C.c_double),
        ("sum_r", C.c_double), ("sum_r2", C.c_double), ("var_u", C.c_double), ("n_events", C.c_double),
        ("n_cache_miss", C.c_double), ("n_redo", C.c_double),
    ]


class SweepTiming(C.Structure):
    _fields_ = [
        ("total_ms", C.c_double), ("dot_ms", C.c_double), ("dot_launches", C.c_int32),
        ("chain_ms", C.c_double), ("update_ms", C.c_double), ("other_ms", C.c_double),
    ]


class LaunchStats(C.Structure):
    _fields_ = [
        ("launches", C.c_int32), ("launches_all", C.c_int32), ("blocks", C.c_int32), ("cols_per_launch", C.c_int32),
        ("avg_ms", C.c_double), ("min_ms", C.c_double), ("max_ms", C.c_double), ("sum_ms", C.c_double), ("span_ms", C.c_double),
    ]


# every symbol include/hibayes_gpu.h declares
SYMBOLS = [
    "hb_abi_version", "hb_version", "hb_last_error", "hb_device_count", "hb_exchange_count", "hb_bayes_run", "hb_sbayes_run",
    "hb_ctx_create", "hb_ctx_destroy", "hb_ctx_panel", "hb_ctx_ld", "hb_ctx_upload_genotype_i8",
    "hb_ctx_upload_genotype_f64", "hb_ctx_upload_bed", "hb_ctx_generate_genotype", "hb_ctx_download_genotype",
    "hb_ctx_marker_stats", "hb_ctx_build_gram", "hb_ctx_download_gram", "hb_ctx_set_residual",
    "hb_ctx_get_residual", "hb_ctx_set_effects", "hb_ctx_get_effects", "hb_ctx_dot", "hb_ctx_residual_sums",
    "hb_ctx_residual_shift", "hb_ctx_set_covariates", "hb_ctx_cov_dot", "hb_ctx_cov_axpy", "hb_ctx_set_levels",
    "hb_ctx_level_sums", "hb_ctx_level_axpy", "hb_ctx_blocks_setup", "hb_ctx_blocks_step", "hb_ctx_blocks_state", "hb_ctx_sweep", "hb_ctx_sweep_range", "hb_ctx_sweep_end", "hb_ctx_get_counters", "hb_ctx_set_windows",
    "hb_ctx_get_windows", "hb_ctx_last_timing", "hb_ctx_set_profiling", "hb_ctx_matvec", "hb_ctx_set_pipeline", "hb_ctx_time_matvec", "hb_ctx_matvec_stamps", "hb_ctx_set_layout", "hb_ctx_get_layout",
    "hb_ctx_download_gram_band", "hb_ctx_set_adaptive", "hb_ctx_get_pipeline", "hb_ctx_get_events", "hb_ctx_pipeline_note", "hb_ctx_matmul",
    "hb_comm_unique_id", "hb_comm_init", "hb_comm_world", "hb_comm_rank", "hb_comm_selftest", "hb_comm_destroy",
    "hb_ctx_debug_inject_abort", "hb_ctx_set_matvec_kernel", "hb_ctx_time_stream_read",
    "hb_run_create", "hb_run_step", "hb_run_state", "hb_run_ctx", "hb_run_finish", "hb_run_destroy",
]


def lib():
    """Load the shared library (once). Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). hibayes_amd has no CPU fallback." % LIB_PATH)
    # ONE HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64 / librccl (sonames without a version, so
    # they do not dedupe against /opt/rocm's) and load them RTLD_GLOBAL: whichever is loaded first wins the symbol lookup of
    # everything loaded later. If torch arrived between two first calls into a lazily bound library, its hip* calls would be
    # split over two runtimes (a stream of one handed to the other: "unhandled cuda error" from RCCL). So where torch is
    # importable it is loaded first — this library, the RCCL it dlopen()s and torch's tensors (TorchComm's exchange buffer)
    # then all sit on the same runtime — and every symbol is bound at load time, not at first call.
    # (HIBAYES_NO_TORCH=1 skips it for a single-GPU process that will never import torch: seconds of start-up. The order cannot be
    # repaired later — a torch imported after this library brings the second runtime — hence the default.)
    if not os.environ.get("HIBAYES_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH, mode=os.RTLD_NOW)
    L.hb_version.restype = C.c_char_p
    L.hb_last_error.restype = C.c_char_p
    L.hb_exchange_count.restype = C.c_size_t
    L.hb_exchange_count.argtypes = [C.c_int32]
    L.hb_bayes_run.argtypes = [C.POINTER(BayesArgs), C.POINTER(BayesOut)]
    L.hb_sbayes_run.argtypes = [C.POINTER(SBayesArgs), C.POINTER(SBayesOut)]
    L.hb_ctx_create.argtypes = [C.POINTER(CtxParams), C.POINTER(C.c_void_p)]
    L.hb_ctx_destroy.argtypes = [C.c_void_p]
    L.hb_ctx_destroy.restype = None
    L.hb_ctx_panel.argtypes = [C.c_void_p]
    L.hb_ctx_ld.argtypes = [C.c_void_p]
    L.hb_ctx_ld.restype = C.c_int64
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.hb_ctx_upload_genotype_i8.argtypes = [vp, vp, i64, i32, i32]
    L.hb_ctx_upload_genotype_f64.argtypes = [vp, vp, i64, i32, i32]
    L.hb_ctx_upload_bed.argtypes = [vp, vp, i64, i32, vp, i32, i32]
    L.hb_ctx_generate_genotype.argtypes = [vp, C.c_uint64, i32]
    L.hb_ctx_download_genotype.argtypes = [vp, vp, i64, i32, i32]
    L.hb_ctx_marker_stats.argtypes = [vp, vp, vp, C.POINTER(dbl), C.POINTER(i32)]
    L.hb_ctx_build_gram.argtypes = [vp, C.POINTER(dbl)]
    L.hb_ctx_download_gram.argtypes = [vp, i32, vp]
    L.hb_ctx_download_gram_band.argtypes = [vp, i32, i32, vp]
    L.hb_ctx_get_pipeline.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.hb_ctx_get_events.argtypes = [vp, vp, vp, vp]
    L.hb_ctx_matmul.argtypes = [vp, vp, i64, i32, vp, i64]
    L.hb_comm_unique_id.argtypes = [vp]
    L.hb_comm_init.argtypes = [C.POINTER(vp), vp, i32, i32, i32]
    L.hb_comm_world.argtypes = [vp]
    L.hb_comm_rank.argtypes = [vp]
    L.hb_comm_selftest.argtypes = [vp]
    L.hb_comm_destroy.argtypes = [vp]
    L.hb_comm_destroy.restype = None
    L.hb_ctx_pipeline_note.argtypes = [vp]
    L.hb_ctx_pipeline_note.restype = C.c_char_p
    L.hb_ctx_set_residual.argtypes = [vp, vp, vp]
    L.hb_ctx_get_residual.argtypes = [vp, vp, vp]
    L.hb_ctx_set_effects.argtypes = [vp, vp, vp, vp]
    L.hb_ctx_get_effects.argtypes = [vp, vp, vp, vp]
    L.hb_ctx_dot.argtypes = [vp, i32, i32, vp]
    L.hb_ctx_residual_sums.argtypes = [vp, C.POINTER(dbl), C.POINTER(dbl)]
    L.hb_ctx_residual_shift.argtypes = [vp, dbl]
    L.hb_ctx_set_covariates.argtypes = [vp, vp, i32]
    L.hb_ctx_cov_dot.argtypes = [vp, i32, C.POINTER(dbl)]
    L.hb_ctx_cov_axpy.argtypes = [vp, i32, dbl]
    L.hb_ctx_set_levels.argtypes = [vp, vp, i32, vp]
    L.hb_ctx_level_sums.argtypes = [vp, i32, vp]
    L.hb_ctx_level_axpy.argtypes = [vp, i32, vp]
    L.hb_ctx_blocks_setup.argtypes = [vp, vp, vp, vp]
    L.hb_ctx_blocks_step.argtypes = [vp, dbl, vp, vp, vp, dbl, dbl]
    L.hb_ctx_blocks_state.argtypes = [vp, vp, vp, vp, vp]
    L.hb_ctx_sweep.argtypes = [vp, C.POINTER(SweepIn), C.POINTER(SweepOut)]
    L.hb_ctx_sweep_range.argtypes = [vp, C.POINTER(SweepIn), i32, i32]
    L.hb_ctx_sweep_end.argtypes = [vp, C.POINTER(SweepOut)]
    L.hb_ctx_get_counters.argtypes = [vp, vp, vp, vp]
    L.hb_ctx_set_windows.argtypes = [vp, vp, i32]
    L.hb_ctx_get_windows.argtypes = [vp, vp]
    L.hb_ctx_last_timing.argtypes = [vp, C.POINTER(SweepTiming)]
    L.hb_ctx_set_profiling.argtypes = [vp, i32]
    L.hb_ctx_matvec.argtypes = [vp, vp, vp]
    L.hb_ctx_set_pipeline.argtypes = [vp, i32, i32, i32]
    L.hb_ctx_set_adaptive.argtypes = [vp, i32]
    L.hb_ctx_time_matvec.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(i32), C.POINTER(i32)]
    L.hb_ctx_time_stream_read.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(i64)]
    L.hb_ctx_set_layout.argtypes = [vp, i32, i32]
    L.hb_ctx_get_layout.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.hb_ctx_matvec_stamps.argtypes = [vp, C.POINTER(LaunchStats)]
    L.hb_ctx_debug_inject_abort.argtypes = [vp, i32, i32]
    L.hb_ctx_set_matvec_kernel.argtypes = [vp, i32]
    L.hb_run_create.argtypes = [C.POINTER(BayesArgs), C.POINTER(vp)]
    L.hb_run_step.argtypes = [vp, i32, C.POINTER(i32)]
    L.hb_run_state.argtypes = [vp, C.POINTER(RunInfo)]
    L.hb_run_ctx.argtypes = [vp]
    L.hb_run_ctx.restype = vp
    L.hb_run_finish.argtypes = [vp, C.POINTER(BayesOut)]
    L.hb_run_destroy.argtypes = [vp]
    L.hb_run_destroy.restype = None
    _lib = L
    return L


def check(rc):
    if rc:
        raise HibayesError(rc, lib().hb_last_error().decode("utf-8", "replace"))
