"""ctypes binding of libttk.so (the C ABI in include/ttk.h).

The shared library is built in-tree by `__graft_entry__.build()` (hipcc,
--offload-arch=gfx950).  There is no fallback: if the library is missing every
GPU operation raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTK_LIB_PATH") or os.path.join(_HERE, "libttk.so")   # (override: diagnostic builds)
SRC_PATH = os.path.join(_HERE, "csrc", "ttk.hip")
SRC_PATHS = [SRC_PATH, os.path.join(_HERE, "csrc", "ttk_dense.hip")]
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

_c_dp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double

# name -> (restype, argtypes); must list every symbol include/ttk.h declares
SIGNATURES = {
    "ttk_version": (_int, []),
    "ttk_env_step_left_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _c_dp]),
    "ttk_env_step_right_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp]),
    "ttk_predict_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _c_dp]),
    "ttk_kr_rows_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _c_dp]),
    "ttk_loss_f64": (_int, [_int, _dbl, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _c_dp]),
    "ttk_batch_mean_f64": (_int, [_c_dp, _i64, _int, _i64, _c_dp, _c_dp]),
    "ttk_hess_left_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _c_dp]),
    "ttk_gram_ws_bytes": (_i64, [_i64, _int, _int, _int, _int, _int]),
    "ttk_gram_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _c_dp, _c_dp, _i64, _c_dp]),
    "ttk_gram_cancellable_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _c_dp, _c_dp, _i64,
                                        _c_dp, ctypes.c_uint, _c_dp]),
    "ttk_gram_cancel": (_int, [_c_dp, ctypes.c_uint, _c_dp]),
    "ttk_gram_scale_f64": (_int, [_c_dp, _int, _int, _int, _int, _c_dp, _c_dp]),
    "ttk_gram_expand_f64": (_int, [_c_dp, _int, _int, _int, _int, _c_dp, _dbl, _c_dp, _c_dp]),
    "ttk_gram_timing": (_int, [_int]),
    "ttk_gram_timing_read": (_int, [_c_dp, _int]),
    "ttk_rhs_ws_bytes": (_i64, [_i64, _int, _int, _int, _int]),
    "ttk_rhs_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_rhs_static_max_bytes": (_i64, []),
    "ttk_rhs_static_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _c_dp]),
    "ttk_matvec_fused_ws_bytes": (_i64, [_i64, _int, _int, _int, _int]),
    "ttk_matvec_fused_f64": (_int, [_c_dp] * 6 + [_i64, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_cg_fused_iter_f64": (_int, [_c_dp] * 8 + [_int, _i64, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_jvp_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp]),
    "ttk_solve_rhs_f64": (_int, [_c_dp, _c_dp, _c_dp, _dbl, _c_dp, _i64, _c_dp]),
    "ttk_chol_ws_bytes": (_i64, [_int]),
    "ttk_chol_prepare_f64": (_int, [_c_dp, _int, _int, _int, _int, _c_dp, _dbl, _c_dp, _c_dp, _c_dp, _c_dp]),
    "ttk_chol_solve_f64": (_int, [_c_dp, _int, _c_dp, _c_dp, _c_dp]),
    "ttk_chol_solve_gram_f64": (_int, [_c_dp, _int, _int, _int, _int, _c_dp, _dbl, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]),
    "ttk_chol_mode": (_int, [_int]),
    "ttk_qr_lds_bytes": (_i64, [_int, _int]),
    "ttk_qr_f64": (_int, [_c_dp, _int, _int, _c_dp, _c_dp, _c_dp]),
    "ttk_qr_right_f64": (_int, [_c_dp, _int, _int, _c_dp, _c_dp, _c_dp]),
    "ttk_dot_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _c_dp]),
    "ttk_axpy_f64": (_int, [_c_dp, _dbl, _c_dp, _c_dp, _i64, _c_dp]),
    "ttk_gl_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _c_dp]),
    "ttk_cg_update_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp]),
    "ttk_cg_update_ws_bytes": (_i64, [_i64]),
    "ttk_cg_update_ws_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _i64, _c_dp]),
    "ttk_minres_update_f64": (_int, [_c_dp] * 8 + [_i64, _c_dp]),
    "ttk_minres_update_ws_bytes": (_i64, [_i64]),
    "ttk_minres_update_ws_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _i64, _c_dp]),
    "ttk_feature_map_f64": (_int, [_c_dp, _i64, _int, _int, _int, _c_dp, _c_dp]),
    "ttk_patchify": (_int, [_c_dp, _i64, _int, _int, _int, _int, _int, _int, _c_dp, _c_dp]),
    "ttk_conv_feature_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _c_dp]),
    "ttk_conv_feature_f32": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _c_dp]),
    "ttk_gemm_rows_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _int, _c_dp]),
    "ttk_env_right_dense_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp]),
    "ttk_lcontract_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp]),
    "ttk_lcontract_sum_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _c_dp]),
    "ttk_lcontract_strided_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _i64, _i64, _int, _c_dp]),
    "ttk_patch_contract_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _i64, _i64, _i64, _int, _c_dp]),
    "ttk_patch_contract_f32": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _i64, _i64, _i64, _int, _c_dp]),
    "ttk_patch_contract_multi_ok": (_int, [_int, _int, _int, _int]),
    "ttk_patch_contract_multi_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp]),
    "ttk_patch_contract_multi_f32": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp]),
    "ttk_gemm_tn_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _c_dp]),
    "ttk_rhs_feature_ws_bytes": (_i64, [_i64, _int, _int, _int]),
    "ttk_rhs_feature_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_filter_ok": (_int, [_i64, _int, _int, _int, _int]),
    "ttk_filter_ws_bytes": (_i64, [_i64, _int, _int, _int, _int]),
    "ttk_filter_matvec_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_filter_jv_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp]),
    "ttk_filter_rhs_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_bond_ok": (_int, [_i64, _int, _int, _int, _int, _int, _int]),
    "ttk_bond_z_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _c_dp]),
    "ttk_bond_lhl_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _c_dp]),
    "ttk_bond_jv_ws_bytes": (_i64, [_i64, _int, _int]),
    "ttk_bond_jv_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_bond_gl_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _c_dp]),
    "ttk_bond_rhs_ws_bytes": (_i64, [_i64, _int, _int, _int]),
    "ttk_bond_rhs_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_gemm_rows_mm32": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _int, _int, _c_dp]),
    "ttk_env_right_dense_mm32": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp]),
    "ttk_env_right_dense_ok": (_int, [_i64, _int, _int, _int, _int]),
    "ttk_rhs_feature_mm32": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_dense_scale_f64": (_int, [_c_dp, _int, _c_dp, _c_dp]),
    "ttk_chol_prepare_dense_f64": (_int, [_c_dp, _int, _c_dp, _dbl, _c_dp, _c_dp, _c_dp, _c_dp]),
    "ttk_lanczos_update_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _int, _i64, _dbl, _c_dp, _c_dp]),
    "ttk_lanczos_fused_block_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp, _int, _int, _dbl, _c_dp,
                                           _i64, _int, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_lanczos_finish_ws_bytes": (_i64, [_int]),
    "ttk_lanczos_finish_f64": (_int, [_c_dp, _i64, _c_dp, _c_dp, _c_dp, _int, _int, _c_dp, _c_dp, _i64, _c_dp, _i64, _c_dp]),
    "ttk_rows_dot_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _c_dp]),
    "ttk_rows_matvec_ws_bytes": (_i64, [_i64, _int, _int]),
    "ttk_rows_matvec_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_rows_rhs_f64": (_int, [_c_dp, _c_dp, _c_dp, _i64, _int, _int, _c_dp, _i64, _c_dp]),
    "ttk_lhl_f64": (_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _int, _int, _c_dp]),
}

_lib = None


class TTKError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile csrc/ttk.hip for gfx950 into libttk.so (cross-compiles without a GPU)."""
    import subprocess
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(
            [os.path.getmtime(s) for s in SRC_PATHS] + [os.path.getmtime(os.path.join(INCLUDE_DIR, "ttk.h"))]):
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", INCLUDE_DIR,
           "-o", LIB_PATH] + SRC_PATHS
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TTKError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no non-HIP fallback for the core-update path")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_ERR = {-1: "invalid argument", -2: "workspace too small", -3: "kernel launch failed"}


def check(rc, what):
    if rc != 0:
        raise TTKError(f"{what} failed: {_ERR.get(rc, rc)}")
