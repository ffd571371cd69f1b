"""ctypes binding of libgse_hip.so (the C ABI declared in include/gse.h).

There is no fallback: if the library is missing, or a call fails, an exception is raised.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libgse_hip.so")

GSE_NX, GSE_NU, GSE_NY, GSE_NSIGMA = 5, 2, 2, 11
GSE_MAX_ND, GSE_MAX_B, GSE_TILE = 4, 32, 1024
GSE_RESAMPLE_TILES_VALID = 1
GSE_RESAMPLE_SEPARATE_SCAN = 2
GSE_RESAMPLE_PREFIX_FROM_SLOT, GSE_RESAMPLE_SLOT1 = 4, 8
GSE_OP_GATHER, GSE_OP_PREDICT, GSE_OP_UPDATE = 1, 2, 4
GSE_OP_SCAN_IN, GSE_OP_SCAN_OUT, GSE_OP_SCAN_IN_SLOT1, GSE_OP_SCAN_OUT_SLOT1 = 8, 16, 32, 64
GSE_OP_U_DEVICE, GSE_OP_Z_DEVICE = 128, 256
GSE_OP_PREFIX_GLOBAL = 512
GSE_OP_WEIGHT_SCALE = 1024
GSE_OP_CLUSTER = 2048
GSE_OP_SAMPLES = 4096
GSE_OP_UNIFORM_WEIGHTS = 8192
GSE_SLOT_QUARTERS = 2
GSE_CLUSTER_MAX_TILES = 256        # policy: filters whose B x tiles stay within this run four workgroups per tile (GSE_OP_CLUSTER)
GSE_RESAMPLE_WEIGHT_SCALE = 16
GSE_IO_U_DEVICE, GSE_IO_Z_DEVICE = 1, 2
GSE_IO_SUMS_SLOT1 = 4
GSE_CYCLE_LANES_MAX_TILES = 256     # gse_gsukf_cycle_lanes: banks of up to this many tiles of 1024 Gaussians
GSE_CYCLE_LANES_MAX_WORKGROUPS = 512   # ... and the class's policy: while the launch is one generation of two workgroups per CU
GSE_SCAN_MAX_TILES = 2048      # policy: up to 2^21 particles every workgroup of the cycle kernel rebuilds the tile prefix in LDS
                               # (one launch per cycle); beyond, a one-workgroup scan of the tile sums runs first and the
                               # cycle kernel stages windows of ITS prefix (GSE_OP_PREFIX_GLOBAL: two launches, the second
                               # still the whole cycle -- round 2 fell back to cumsum + staged search there)
GSE_GSUKF_CYCLE_MAX = 64
GSE_LOOP_ARGS_BYTES = 4096
GSE_SMALL_RESAMPLE_MAX = 4096       # banks up to here find their ancestors in one launch (small_resample_kernel)
GSE_GSUKF_MOMENTS = 46             # doubles per row of gse_gsukf_moments
GSE_GSUKF_LANES_MIN = 65         # from this many Gaussians on: one lane per Gaussian, predict + update in one launch,
                                 # the row copy of resample() deferred into it (gse_gsukf.hip: lane_per_gaussian)

c_f = ctypes.c_float
c_d = ctypes.c_double
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_u32 = ctypes.c_uint32
c_u64 = ctypes.c_uint64
c_vp = ctypes.c_void_p


class Mixture(ctypes.Structure):
    """struct gse_mixture (include/gse.h)."""
    _fields_ = [("nd", c_i32), ("nx", c_i32),
                ("weights", c_f * GSE_MAX_ND),
                ("consts", c_f * GSE_MAX_ND),
                ("means", (c_f * GSE_NX) * GSE_MAX_ND),
                ("chol", ((c_f * GSE_NX) * GSE_NX) * GSE_MAX_ND),
                ("inv_cov", ((c_d * GSE_NX) * GSE_NX) * GSE_MAX_ND)]


GSE_N_MAX_NX = 8


class MixtureN(ctypes.Structure):
    """struct gse_mixture_n (include/gse.h): a mixture of up to 8 dimensions, as the dimension-generic kernels take it."""
    _fields_ = [("nd", c_i32), ("n", c_i32), ("cumw", c_f * GSE_MAX_ND), ("cw", c_d * GSE_MAX_ND),
                ("mean", (c_f * GSE_N_MAX_NX) * GSE_MAX_ND), ("chol", ((c_f * GSE_N_MAX_NX) * GSE_N_MAX_NX) * GSE_MAX_ND),
                ("inv", ((c_d * GSE_N_MAX_NX) * GSE_N_MAX_NX) * GSE_MAX_ND)]


class Rng(ctypes.Structure):
    """struct gse_rng (include/gse.h)."""
    _fields_ = [("seed", c_u64), ("step", c_u64), ("element_offset", c_u64),
                ("filter_offset", c_u32), ("reserved", c_u32)]


class LoopSpec(ctypes.Structure):
    """struct gse_loop_spec (include/gse.h)."""
    _fields_ = [("B", c_i32), ("N", c_i32), ("ticks", c_i32), ("reserved", c_i32), ("dt", c_d)] + \
               [(name, c_vp) for name in ("w_sigma", "state_pdf", "rng", "measurement_pdf", "plant_state_noise",
                                          "plant_meas_noise", "plant_rng", "law", "glue", "schedule", "r", "mu", "P", "w", "X", "u",
                                          "z", "y_pred", "known", "us", "xs", "ys", "ys_meas", "moments", "status", "chol_status",
                                          "args_buffer")]


class LoopTail(ctypes.Structure):
    """struct gse_loop_tail (include/gse.h)."""
    _fields_ = [(name, c_vp) for name in ("law", "glue", "y_meas", "u", "y_pred", "known", "status", "X")] + [("dt", c_d)] + \
               [(name, c_vp) for name in ("state_noise", "meas_noise", "rng", "state_draw", "meas_draw", "y", "y_meas_out", "z",
                                          "log_x", "log_u")]


class PfLoopSpec(ctypes.Structure):
    """struct gse_pf_loop_spec (include/gse.h)."""
    _fields_ = [("B", c_i32), ("ticks", c_i32), ("N", c_i64), ("ld", c_i64), ("dt", c_d), ("flags", c_u32), ("samples_ok", c_u32),
                ("x", c_vp * 2), ("w", c_vp * 2), ("workspace", c_vp), ("state_pdf", c_vp), ("measurement_pdf", c_vp),
                ("rng", Rng), ("law", c_vp), ("glue", c_vp), ("X", c_vp), ("u", c_vp), ("z", c_vp), ("y_pred", c_vp),
                ("known", c_vp), ("plant_state_noise", c_vp), ("plant_meas_noise", c_vp), ("plant_rng", Rng),
                ("schedule", c_vp), ("r", c_vp), ("us", c_vp), ("xs", c_vp), ("ys", c_vp), ("ys_meas", c_vp), ("moments", c_vp),
                ("status", c_vp),
                ("x_current", c_i32), ("w_current", c_i32), ("pend_gather", c_i32), ("pend_predict", c_i32),
                ("gather_slot", c_i32), ("scan_slot", c_i32), ("r_pending", c_d * GSE_MAX_B),
                ("island", c_vp), ("w_uniform", c_i32), ("reserved2", c_i32)]


GSE_PF_LOOP_WEIGHT_SCALE = 1
GSE_PF_LOOP_CLUSTER = 2


class GsukfCyclesSpec(ctypes.Structure):
    """struct gse_gsukf_cycles_spec (include/gse.h)."""
    _fields_ = [("B", c_i32), ("regime", c_i32), ("N", c_i64), ("dt", c_d), ("mu", c_vp * 2), ("P", c_vp * 2), ("w", c_vp * 2),
                ("idx", c_vp), ("workspace", c_vp), ("status", c_vp), ("w_sigma", c_vp), ("state_pdf", c_vp),
                ("measurement_pdf", c_vp), ("rng", Rng), ("r", c_vp),
                ("bank_current", c_i32), ("w_current", c_i32), ("pend_resample", c_i32), ("pend_gather", c_i32),
                ("sums_valid", c_i32), ("sums_slot", c_i32), ("r_pending", c_d * GSE_MAX_B)]


GSE_GSUKF_REGIME_CYCLE, GSE_GSUKF_REGIME_LANES, GSE_GSUKF_REGIME_TWO = 0, 1, 2


class MergeSpec(ctypes.Structure):
    """struct gse_merge_spec (include/gse.h)."""
    _fields_ = [("P", c_i32), ("rank", c_i32), ("flags", c_i32), ("slot", c_i32), ("r", c_d), ("island", c_vp), ("wscale", c_vp),
                ("common_pivot", c_d * GSE_NX), ("rows", c_vp), ("moments26", c_vp), ("merged26", c_vp), ("estimate6", c_vp),
                ("status", c_vp)]


GSE_MERGE_STATUS_NO_WEIGHT, GSE_MERGE_STATUS_OVERFLOW = 1, 2
GSE_MERGE_PENDING, GSE_MERGE_COV_SIZE = 1, 2
GSE_WSTATUS_SUM_NOT_FINITE = 1

P_MIX = ctypes.POINTER(Mixture)
P_RNG = ctypes.POINTER(Rng)
P_D = ctypes.POINTER(c_d)
P_F = ctypes.POINTER(c_f)

# name -> (restype, argtypes); device pointers are passed as integers (tensor.data_ptr()).
PROTOTYPES = {
    "gse_version": (ctypes.c_int, []),
    "gse_last_error": (ctypes.c_char_p, []),
    "gse_workspace_bytes": (c_i64, [ctypes.c_int, c_i64]),
    "gse_workspace_weight_scale_offset": (c_i64, [ctypes.c_int, c_i64]),
    "gse_workspace_status_offset": (c_i64, [ctypes.c_int, c_i64]),
    "gse_pf_slot_to_tiles": (ctypes.c_int, [ctypes.c_int, c_i64, ctypes.c_int, c_vp, c_vp]),
    "gse_stream_synchronize": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "gse_pf_predict": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, P_D, c_d, P_MIX, P_RNG, c_vp]),
    "gse_pf_predict_injected": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, P_D, c_d, c_vp, c_vp]),
    "gse_pf_update": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_int, c_i64, P_D, P_D, P_MIX, c_vp, c_vp]),
    "gse_pf_resample": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_int, c_i64, P_D, c_u32, c_vp, c_vp, c_vp]),
    "gse_pf_gather_index": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_int, c_i64, c_vp]),
    "gse_pf_resample_begin": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_u32, c_vp, c_vp, c_vp]),
    "gse_pf_step": (ctypes.c_int, [c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_int, c_i64, P_D, c_d, P_D, P_D,
                                   P_MIX, P_RNG, c_vp, P_MIX, c_vp, c_vp, c_vp]),
    "gse_pf_moments": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_int, c_i64, c_vp, c_vp, c_vp]),
    "gse_pf_moments_resampled": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_int, c_i64, P_D, ctypes.c_int, c_vp, c_vp, c_vp]),
    "gse_pf_moments_resampled_tail": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_int, c_i64, P_D, ctypes.c_int, c_vp, c_vp,
                                                     ctypes.POINTER(LoopTail), c_vp]),
    "gse_model_create": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, c_u64, ctypes.c_char_p, ctypes.c_char_p,
                                        ctypes.c_char_p, ctypes.POINTER(c_vp)]),
    "gse_model_destroy": (ctypes.c_int, [c_vp]),
    "gse_generic_model_create": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_u64, ctypes.c_char_p,
                                                ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(c_vp)]),
    "gse_model_launch": (ctypes.c_int, [c_vp, ctypes.c_char_p, c_u32, c_u32, c_u32, c_u32, c_u32, c_u32, c_u32,
                                        ctypes.POINTER(c_vp), c_vp]),
    "gse_model_prepare": (ctypes.c_int, [c_vp, ctypes.c_char_p]),
    "gse_model_compile_check": (c_i64, [c_vp, ctypes.c_char_p]),
    "gse_model_kernels": (c_i64, [c_vp]),
    "gse_pf_step_model": (ctypes.c_int, [c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_int, c_i64, P_D, c_d, P_D, P_D,
                                         P_MIX, P_RNG, c_vp, P_MIX, c_vp, c_vp, c_vp]),
    "gse_model_f_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, P_D, c_d, c_vp]),
    "gse_model_g_u_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, P_D, c_vp]),
    "gse_gsukf_sigma_points": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_vp]),
    "gse_gsukf_predict": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_d, P_F, P_MIX, P_RNG, c_vp, c_vp, c_vp]),
    "gse_gsukf_update": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, P_D, P_F, P_MIX, c_vp, c_vp, c_vp]),
    "gse_gsukf_predict_update": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_d, P_D, P_F, P_MIX, P_RNG, c_vp, P_MIX,
                                                c_vp, c_vp]),
    "gse_gsukf_cycle": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_d, P_D, P_D, P_F, P_MIX, P_RNG,
                                       c_vp, P_MIX, c_vp, c_vp]),
    "gse_gsukf_resample": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_u32, c_vp, c_vp]),
    "gse_gsukf_moments": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_vp, c_vp]),
    "gse_gsukf_predict_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_d, P_F, P_MIX, P_RNG, c_vp, c_vp, c_u32, c_vp]),
    "gse_gsukf_update_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_vp, P_F, P_MIX, c_vp, c_vp, c_u32, c_vp]),
    "gse_gsukf_predict_update_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_d, c_vp, P_F, P_MIX, P_RNG, c_vp,
                                                    P_MIX, c_vp, c_u32, c_vp]),
    "gse_gsukf_cycle_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_d, c_vp, P_D, P_F, P_MIX, P_RNG,
                                           c_vp, P_MIX, c_vp, c_u32, c_vp]),
    "gse_gsukf_resample_begin": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_vp, c_vp]),
    "gse_gsukf_gather_rows": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp]),
    "gse_gsukf_predict_update_from": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_d, c_vp, P_F,
                                                     P_MIX, P_RNG, c_vp, P_MIX, c_vp, c_u32, c_vp]),
    "gse_gsukf_predict_update_sums": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_d, c_vp, P_F,
                                                     P_MIX, P_RNG, c_vp, P_MIX, c_vp, c_vp, c_u32, c_vp]),
    "gse_gsukf_resample_begin_sums": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_vp, c_vp]),
    "gse_gsukf_resample_begin_slot": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_vp, ctypes.c_int, c_vp]),
    "gse_gsukf_cycle_lanes": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_d, c_vp, P_D, P_F,
                                             P_MIX, P_RNG, c_vp, P_MIX, c_vp, c_vp, c_u32, c_vp]),
    "gse_gsukf_gather_rows_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp]),
    "gse_gsukf_predict_update_from_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_d,
                                                           c_vp, P_F, P_MIX, P_RNG, c_vp, P_MIX, c_vp, c_u32, c_vp]),
    "gse_gsukf_cycles": (ctypes.c_int, [ctypes.POINTER(GsukfCyclesSpec), ctypes.c_int, c_vp, c_vp, c_vp]),
    "gse_gsukf_moments26": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_vp, c_vp]),
    "gse_gsukf_sigma_points_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_vp]),
    "gse_gsukf_predict_model": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_d, P_F, P_MIX, P_RNG, c_vp, c_vp, c_vp]),
    "gse_gsukf_update_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, P_D, P_F, P_MIX, c_vp, c_vp, c_vp]),
    "gse_gsukf_predict_update_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_d, P_D, P_F, P_MIX, P_RNG,
                                                      c_vp, P_MIX, c_vp, c_vp]),
    "gse_gsukf_cycle_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_d, P_D, P_D, P_F, P_MIX, P_RNG,
                                             c_vp, P_MIX, c_vp, c_vp]),
    "gse_gsukf_resample_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, P_D, c_u32, c_vp,
                                                c_vp]),
    "gse_gsukf_moments_model": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_vp, c_vp]),
    "gse_model_f": (ctypes.c_int, [c_vp, c_vp, c_i64, P_D, c_d, c_vp]),
    "gse_model_g": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "gse_model_g_u": (ctypes.c_int, [c_vp, c_vp, c_i64, P_D, c_vp]),
    "gse_model_tag": (c_u64, []),
    "gse_mixture_pdf_f32": (ctypes.c_int, [c_vp, c_i64, P_MIX, c_vp, c_vp]),
    "gse_mixture_pdf_f64": (ctypes.c_int, [c_vp, c_i64, P_MIX, c_vp, c_vp]),
    "gse_mixture_draw": (ctypes.c_int, [c_vp, c_i64, c_i64, P_MIX, P_RNG, c_vp]),
    "gse_mixture_draw_n": (ctypes.c_int, [c_vp, c_i64, ctypes.POINTER(MixtureN), P_RNG, c_vp]),
    "gse_mixture_pdf_n": (ctypes.c_int, [c_vp, c_i64, ctypes.POINTER(MixtureN), c_vp, c_vp]),
    "gse_cumsum_normalised": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "gse_systematic_index": (ctypes.c_int, [c_vp, c_vp, c_i64, c_d, c_vp]),
    "gse_rng_raw": (ctypes.c_int, [c_vp, c_i64, P_RNG, c_u32, c_vp]),
    "gse_mpc_step": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "gse_loop_control": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "gse_pf_tile_sums": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_vp, c_vp]),
    "gse_tile_prefix": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "gse_pf_shard_cumsum": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "gse_pf_shard_gather": (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_d, c_vp, c_vp, c_vp]),
    "gse_plant_step": (ctypes.c_int, [c_vp, c_vp, c_d, P_MIX, P_MIX, P_RNG, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "gse_gsukf_closed_loop": (ctypes.c_int, [ctypes.POINTER(LoopSpec), c_vp]),
    "gse_pf_closed_loop": (ctypes.c_int, [ctypes.POINTER(PfLoopSpec), c_vp]),
    "gse_pf_cycles": (ctypes.c_int, [ctypes.POINTER(PfLoopSpec), ctypes.c_int, c_vp, c_vp, c_vp]),
    "gse_moments_estimates": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "gse_merge_row": (ctypes.c_int, [c_vp, c_vp, c_vp, P_D, c_vp, c_vp]),
    "gse_merge_rows": (ctypes.c_int, [c_vp, ctypes.c_int, P_D, c_vp, c_vp, c_vp]),
    "gse_pf_merge_begin": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, ctypes.POINTER(MergeSpec), c_vp, c_vp]),
    "gse_merge_end": (ctypes.c_int, [ctypes.POINTER(MergeSpec), c_vp]),
    "gse_island_absorb": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "gse_plant_step_injected": (ctypes.c_int, [c_vp, c_vp, c_d, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
}

_lib = None


class GseError(RuntimeError):
    pass


def load():
    """Load libgse_hip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gpu_se_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python gpu_se_amd/build.py`).  There is no CPU fallback." % LIB_PATH)
    # torch first: it brings its own HIP runtime (torch/lib/libamdhip64.so), and the buffers and streams handed to the
    # kernels live in that runtime.  Loading libgse_hip.so before torch would bind it to the system copy of the
    # runtime instead -- a second runtime in the process, which sees no device ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().gse_last_error()
        raise GseError("libgse_hip: error %d: %s" % (rc, msg.decode() if msg else "?"))


def call(name, *args):
    check(getattr(load(), name)(*args))


def doubles(values, n):
    """Host array-like -> ctypes double[n]."""
    vals = [float(v) for v in values]
    if len(vals) != n:
        raise ValueError("expected %d values, got %d" % (n, len(vals)))
    return (c_d * n)(*vals)
