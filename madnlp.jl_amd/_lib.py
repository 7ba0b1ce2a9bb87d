"""ctypes binding of libmadnlp_hip.so (the C ABI in include/madnlp_hip.h).

The library is built in-tree by `build()` (hipcc, gfx950 only).  There is no CPU
fallback: if the shared object is missing or cannot be loaded, importing the
bindings raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libmadnlp_hip.so")
_LIB_OVERRIDE = os.environ.get("MNK_LIBPATH")   # A/B runs of a diagnostic build of the same ABI (tools/ab_*.sh); never a fallback
SOURCES = ["gemm_f64.hip", "dag.hip", "factor.hip", "solve.hip", "ls.hip", "sparse_kkt.hip", "dense_kkt.hip", "bk.hip", "schur.hip", "ipm_vec.hip", "opf_eval.hip"]
HEADERS = ["common.h", "ls.h", "kkt_vec.h", "gemm_tile.h", "leaf64.h", "gemm_macro.h", os.path.join("..", "..", "include", "madnlp_hip.h")]

MNK_HOST, MNK_DEVICE = 0, 1
MNK_BUNCHKAUFMAN, MNK_LU, MNK_QR, MNK_CHOLESKY, MNK_LDL, MNK_EVD = 1, 2, 3, 4, 5, 6
MNK_SC_JT, MNK_SC_HESS, MNK_SC_AUG, MNK_SC_DIAGBUF = 0, 1, 2, 3


def _stale() -> bool:
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


# Per-file backend flags.  `-amdgpu-mfma-vgpr-form` (MFMA accumulators in VGPRs, no AGPR copies) is worth +3 % on the
# 168-VGPR tile kernels of gemm_f64.hip.  It must NOT be applied to factor.hip: the persistent panel kernel needs more
# than 256 registers per lane, and with the flag the accumulators are pinned to the 256 architectural VGPRs and the rest
# is shuffled through AGPRs -- hipcc 7.2 then miscompiles it (wrong pivots / memory faults that appear and disappear with
# unrelated, never-executed code; bisected in round 2).
FILE_FLAGS = {"gemm_f64.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "dag.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
              "bk.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 (one object per source, in parallel) and link
    madnlp.jl_amd/lib/libmadnlp_hip.so."""
    if not force and not _stale():
        return LIBPATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = base + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", LIBPATH]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return LIBPATH


_lib = None

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/madnlp_hip.h
SIGNATURES = {
    "mnk_version": (C.c_int, []),
    "mnk_last_error_string": (C.c_char_p, []),
    "mnk_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "mnk_ctx_destroy": (C.c_int, [_vp]),
    "mnk_ctx_synchronize": (C.c_int, [_vp]),
    "mnk_ctx_stream": (_vp, [_vp]),
    "mnk_sc_create": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, C.c_int64, _vp, _vp,
                                C.c_int, C.POINTER(_vp)]),
    "mnk_sc_destroy": (C.c_int, [_vp]),
    "mnk_sc_sizes": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "mnk_sc_get_structure": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mnk_sc_get_map": (C.c_int, [_vp, C.c_int, _vp]),
    "mnk_sc_get_ptrs": (C.c_int, [_vp] + [_vp] * 8),
    "mnk_sc_compress_jacobian": (C.c_int, [_vp, _vp, C.c_int]),
    "mnk_sc_compress_hessian": (C.c_int, [_vp, _vp, C.c_int]),
    "mnk_sc_build": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "mnk_sc_get_values": (C.c_int, [_vp, C.c_int, _vp, C.c_int]),
    "mnk_sc_spmv": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, _vp, C.c_double, _vp]),
    "mnk_dc_create": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, C.c_int,
                                C.POINTER(_vp)]),
    "mnk_dc_destroy": (C.c_int, [_vp]),
    "mnk_dc_set_hess": (C.c_int, [_vp, _vp, C.c_int64, C.c_int]),
    "mnk_dc_set_jac": (C.c_int, [_vp, _vp, C.c_int64, C.c_int]),
    "mnk_dc_build": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "mnk_dc_order": (C.c_int64, [_vp]),
    "mnk_dc_get_aug": (C.c_int, [_vp, _vp, C.c_int]),
    "mnk_ls_create": (C.c_int, [_vp, C.c_int64, C.c_int, C.POINTER(_vp)]),
    "mnk_ls_destroy": (C.c_int, [_vp]),
    "mnk_ls_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "mnk_ls_factorize_sc": (C.c_int, [_vp, _vp, C.POINTER(C.c_int)]),
    "mnk_ls_factorize_dc": (C.c_int, [_vp, _vp, C.POINTER(C.c_int)]),
    "mnk_ls_factorize_dense": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.POINTER(C.c_int)]),
    "mnk_ls_factorize_csc": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.POINTER(C.c_int)]),
    "mnk_ls_factorize_sc_async": (C.c_int, [_vp, _vp]),
    "mnk_ls_factorize_dc_async": (C.c_int, [_vp, _vp]),
    "mnk_ls_inertia": (C.c_int, [_vp, _i64p, _i64p, _i64p]),
    "mnk_ls_solve": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int]),
    "mnk_ls_check_solve": (C.c_int, [_vp]),
    "mnk_ls_get_factor": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "mnk_sc_set_aug_diagonal": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_int]),
    "mnk_sc_regularize_diagonal": (C.c_int, [_vp, C.c_double, C.c_double]),
    "mnk_sc_save_diagonals": (C.c_int, [_vp]),
    "mnk_sc_restore_diagonals": (C.c_int, [_vp]),
    "mnk_sc_get_diagonals": (C.c_int, [_vp] + [_vp] * 7),
    "mnk_dc_set_aug_diagonal": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_int]),
    "mnk_dc_regularize_diagonal": (C.c_int, [_vp, C.c_double, C.c_double]),
    "mnk_dc_get_diagonals": (C.c_int, [_vp] + [_vp] * 7),
    "mnk_ipm_create": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, C.c_int64, _vp, C.c_int, C.POINTER(_vp)]),
    "mnk_ipm_destroy": (C.c_int, [_vp]),
    "mnk_ipm_batch_begin": (C.c_int, [_vp]),
    "mnk_ipm_batch_end": (C.c_int, [_vp]),
    "mnk_ipm_get_varphi": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_inf_du": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_inf_compl": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_min_complementarity": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_double)]),
    "mnk_ipm_get_average_complementarity": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_double)]),
    "mnk_ipm_get_varphi_d": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_alpha_max": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_alpha_z": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_rel_search_norm": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_double)]),
    "mnk_ipm_get_sd_sc": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_norms": (C.c_int, [_vp, _vp, C.c_int64, C.POINTER(C.c_double)]),
    "mnk_ipm_set_perturbation_sets": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int]),
    "mnk_ipm_set_aug_rhs": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "mnk_ipm_dual_inf_perturbation": (C.c_int, [_vp, _vp, C.c_double, C.c_double]),
    "mnk_ipm_adjust_boundary": (C.c_int, [_vp, _vp, _vp, _vp, C.c_double]),
    "mnk_ipm_reset_bound_dual": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double]),
    "mnk_ipm_get_obj_val_R": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_theta_R": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.POINTER(C.c_double)]),
    "mnk_ipm_get_inf_pr_R": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.POINTER(C.c_double)]),
    "mnk_ipm_get_inf_du_R": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_inf_compl_R": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_alpha_max_R": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_alpha_z_R": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_varphi_R": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_F": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_get_varphi_d_R": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "mnk_ipm_populate_RR_nn": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_double, C.c_double]),
    "mnk_ipm_initialize_robust_restorer": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mnk_ipm_set_f_RR": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_double]),
    "mnk_ipm_set_aug_rhs_RR": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp, _vp]),
    "mnk_ipm_finish_aug_solve_RR": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_double, C.c_double]),
    "mnk_ipm_reset_bound_dual_1": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_double, C.c_double]),
    "mnk_ipm_set_initial_bounds": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_double]),
    "mnk_ipm_set_initial_rhs": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp]),
    "mnk_ipm_set_aug_rhs_ifr": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp]),
    "mnk_ipm_set_g_ifr": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double]),
    "mnk_ipm_initialize_variables": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_double, C.c_double]),
    "mnk_ipm_get_dot": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.POINTER(C.c_double)]),
    "mnk_ipm_get_sum": (C.c_int, [_vp, _vp, C.c_int64, C.POINTER(C.c_double)]),
    "mnk_ipm_get_norm2": (C.c_int, [_vp, _vp, C.c_int64, C.POINTER(C.c_double)]),
    "mnk_ipm_vec_copy": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "mnk_ipm_vec_fill": (C.c_int, [_vp, _vp, C.c_int64, C.c_double]),
    "mnk_ipm_vec_axpby": (C.c_int, [_vp, _vp, C.c_double, _vp, C.c_double, _vp, C.c_int64]),
    "mnk_ipm_vec_scatter_axpy": (C.c_int, [_vp, _vp, _vp, C.c_double, _vp, C.c_int64]),
    "mnk_ipm_vec_gather": (C.c_int, [_vp, _vp, C.c_double, _vp, _vp, C.c_int64]),
    "mnk_ipm_bound_dual_axpy": (C.c_int, [_vp, _vp, _vp, C.c_double, _vp, _vp]),
    "mnk_ipm_bound_dual_fill": (C.c_int, [_vp, _vp, _vp, C.c_double]),
    "mnk_ipm_gemv": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, C.c_double, _vp, C.c_int64, _vp, C.c_double, _vp]),
    "mnk_opf_create": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "mnk_opf_destroy": (C.c_int, [_vp]),
    "mnk_opf_sizes": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "mnk_opf_obj_terms": (C.c_int, [_vp, _vp, _vp]),
    "mnk_opf_grad": (C.c_int, [_vp, _vp, _vp]),
    "mnk_opf_cons": (C.c_int, [_vp, _vp, _vp]),
    "mnk_opf_jac_coord": (C.c_int, [_vp, _vp, _vp]),
    "mnk_opf_hess_coord": (C.c_int, [_vp, _vp, _vp, C.c_double, _vp]),
    "mnk_sc_set_aug_RR": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double]),
    "mnk_dc_set_aug_RR": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double]),
    "mnk_ls_bk_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp, _vp]),
    "mnk_ls_get_stat": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
    "mnk_sc_set_bounds": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int]),
    "mnk_sc_set_barrier_terms": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "mnk_sc_solve_kkt": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "mnk_sc_mul": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_int]),
    "mnk_dc_set_bounds": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int]),
    "mnk_dc_set_barrier_terms": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "mnk_dc_solve_kkt": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "mnk_dc_mul": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_int]),
    "mnk_schur_create": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(_vp)]),
    "mnk_schur_destroy": (C.c_int, [_vp]),
    "mnk_schur_set_block": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int]),
    "mnk_schur_build_local": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int64]),
    "mnk_schur_factorize_s": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.POINTER(C.c_int)]),
    "mnk_schur_inertia_s": (C.c_int, [_vp, _i64p, _i64p, _i64p]),
    "mnk_schur_scenario_inertia": (C.c_int, [_vp, C.c_int64, _i64p, _i64p, _i64p]),
    "mnk_schur_forward": (C.c_int, [_vp, _vp, _vp]),
    "mnk_schur_solve_s": (C.c_int, [_vp, _vp]),
    "mnk_schur_backward": (C.c_int, [_vp, _vp, _vp]),
    "mnk_schur_s_buffer": (C.c_void_p, [_vp]),
    "mnk_schur_set_structure": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, C.c_int64, _vp, _vp,
                                          C.c_int64, _vp, C.c_int64, _vp, C.c_int, C.c_int64, _vp, C.c_int]),
    "mnk_schur_assemble": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int]),
    "mnk_schur_s0_buffer": (_vp, [_vp]),
    "mnk_schur_get_block": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    "mnk_schur_solve": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "mnk_ls_debug_solve_trace": (C.c_int, [_vp, _vp, C.c_int64]),
    "mnk_ls_debug_dag_state": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int64, _vp]),
    "mnk_debug_grid_at_launch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mnk_debug_shader_clock": (C.c_int, [_vp, _vp]),
    "mnk_debug_dag_tasks": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "mnk_debug_dag_merged_tasks": (C.c_int, [C.c_int] * 8 + [_vp, C.c_int]),
    "mnk_factorize_batch_begin": (C.c_int, []),
    "mnk_factorize_batch_end": (C.c_int, []),
    "mnk_solve_batch_begin": (C.c_int, []),
    "mnk_solve_batch_end": (C.c_int, []),
    "mnk_release_idle_streams": (C.c_int, [C.c_int]),
    "mnk_sc_step_batch": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "mnk_ls_inertia_batch": (C.c_int, [C.c_int, _vp, _i64p, _i64p, _i64p]),
    "mnk_ls_solve_batch": (C.c_int, [C.c_int, _vp, _vp, C.c_int]),
    "mnk_debug_update": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, _vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int,
                                   C.POINTER(C.c_double)]),
    "mnk_gemm_nt": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, _vp, C.c_int64, _vp,
                              C.c_int64, _vp, C.c_int64]),
}


def lib():
    """Load libmadnlp_hip.so (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise ImportError(
            f"{LIBPATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the KKT hot path.")
    # PyTorch-ROCm bundles its own HIP runtime; if it is going to be used in this process it must be
    # loaded first so that both share ONE runtime (two runtimes -> "no ROCm-capable device").
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    l = C.CDLL(_LIB_OVERRIDE if _LIB_OVERRIDE and os.path.exists(_LIB_OVERRIDE) else LIBPATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(l, name)  # AttributeError if the symbol is not exported
        f.restype = res
        f.argtypes = args
    _lib = l
    return l


class HipError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().mnk_last_error_string()
        raise HipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
