"""lis_amd -- MI355X-native SpMV + Krylov hot path behind the Lis C API.

The product is the C-ABI shared library ``lis_amd/lib/liblis_amd.so`` (hand-written HIP for gfx950 +
the Lis C API in C).  This Python package is plumbing for tests and bench.py only: it loads the
library with ctypes and moves numpy arrays in and out of HBM.  There is no CPU fallback: if the
library is missing, or no GPU is present when a kernel is called, the call fails loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _capi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "liblis_amd.so")
CSRC = os.path.join(HERE, "csrc")

_vp, _ci, _cd, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
_pvp = C.POINTER(C.c_void_p)

_LISHIP = {
    "liship_device_count": (_ci, [C.POINTER(_ci)]),
    "liship_set_device": (_ci, [_ci]),
    "liship_get_device": (_ci, [C.POINTER(_ci)]),
    "liship_device_name": (_ci, [C.c_char_p, _ci]),
    "liship_malloc": (_ci, [_pvp, _sz]),
    "liship_free": (_ci, [_vp]),
    "liship_memset": (_ci, [_vp, _ci, _sz, _vp]),
    "liship_memcpy_h2d": (_ci, [_vp, _vp, _sz, _vp]),
    "liship_memcpy_d2h": (_ci, [_vp, _vp, _sz, _vp]),
    "liship_stream_yardstick": (_ci, [_ci, _sz, _vp, _vp, _ci, _vp]),
    "liship_memcpy_d2d": (_ci, [_vp, _vp, _sz, _vp]),
    "liship_stream_create": (_ci, [_pvp]),
    "liship_stream_destroy": (_ci, [_vp]),
    "liship_stream_synchronize": (_ci, [_vp]),
    "liship_device_synchronize": (_ci, []),
    "liship_graph_capture_begin": (_ci, [_vp]),
    "liship_graph_capture_end": (_ci, [_vp, _pvp]),
    "liship_graph_launch": (_ci, [_vp, _vp]),
    "liship_graph_destroy": (_ci, [_vp]),
    "liship_malloc_host": (_ci, [_pvp, _sz]),
    "liship_free_host": (_ci, [_vp]),
    "liship_event_create": (_ci, [_pvp]),
    "liship_event_destroy": (_ci, [_vp]),
    "liship_event_record": (_ci, [_vp, _vp]),
    "liship_event_synchronize": (_ci, [_vp]),
    "liship_csr_row_facts": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_csr_to_ell": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_csr_to_ell_rows": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_csr_dia_offsets": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_csr_to_dia": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_dia_row_counts": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_dia_to_rows": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_csr_bsr_count": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_csr_to_bsr": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_csr_to_jad": (_ci, [_ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_stream_wait_event": (_ci, [_vp, _vp]),
    "liship_timer_create": (_ci, [_pvp]),
    "liship_timer_destroy": (_ci, [_vp]),
    "liship_timer_start": (_ci, [_vp, _vp]),
    "liship_timer_stop": (_ci, [_vp, _vp]),
    "liship_timer_elapsed_ms": (_ci, [_vp, C.POINTER(C.c_float)]),
    "liship_error_string": (C.c_char_p, [_ci]),
    "liship_csr_plan_create": (_ci, [_pvp, _ci, _vp, _vp]),
    "liship_csr_plan_destroy": (_ci, [_vp]),
    "liship_csr_plan_set_first_term_initialises": (_ci, [_vp, _ci]),
    "liship_csr_plan_info": (_ci, [_vp, C.POINTER(_ci), C.POINTER(C.c_longlong), C.POINTER(_ci)]),
    "liship_spmv_csr_f64": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_spmv_csr_dot_f64": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp]),
    "liship_csr_plan_encode_indices": (_ci, [_vp, _vp, _vp, _vp]),
    "liship_csr_plan_coded": (_ci, [_vp]),
    "liship_csr_plan_encode_row_patterns": (_ci, [_vp, _vp, _vp]),
    "liship_csr_plan_row_patterns": (_ci, [_vp]),
    "liship_csr_plan_pattern_records": (_ci, [_vp]),
    "liship_csr_plan_team_records": (_ci, [_vp]),
    "liship_csr_plan_team_form": (_ci, [_vp]),
    "liship_csr_plan_fused_dots": (_ci, [_vp]),
    "liship_csr_plan_fused_slots": (C.c_longlong, [_vp]),
    "liship_csr_plan_wide_dominant": (_ci, [_vp]),
    "liship_spmv_csr_set_team": (_ci, [_ci]),
    "liship_spmv_csr_set_wide_union": (_ci, [_ci]),
    "liship_spmv_csr_set_dom_march": (_ci, [_ci]),
    "liship_csr_plan_box_planes": (_ci, [_vp]),
    "liship_csr_plan_marching": (_ci, [_vp]),
    "liship_csr_plan_encode_block_rows": (_ci, [_vp, _ci, _vp, _vp]),
    "liship_csr_plan_block_rows": (_ci, [_vp]),
    "liship_spmv_csr_set_block_rows": (_ci, [_ci]),
    "liship_spmv_bsr_rows_f64": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp]),
    "liship_spmv_bsr_set_team": (_ci, [_ci]),
    "liship_csr_plan_encode_row_values": (_ci, [_vp, _vp, _vp, _vp]),
    "liship_csr_plan_value_records": (_ci, [_vp]),
    "liship_csr_plan_dominant_pattern": (_ci, [_vp]),
    "liship_spmv_csr_set_row_values": (_ci, [_ci]),
    "liship_spmv_csr_set_row_patterns": (_ci, [_ci]),
    "liship_csr_plan_localize_columns": (_ci, [_vp, _vp, _vp, _vp]),
    "liship_csr_plan_localized": (C.c_longlong, [_vp]),
    "liship_spmv_csr_set_local_columns": (_ci, [_ci]),
    "liship_spmv_csr_set_local_register_positions": (_ci, [_ci]),
    "liship_spmv_csr_set_xcd_strips": (_ci, [_ci]),
    "liship_csr_plan_scan_band": (_ci, [_vp, _vp, _vp, _vp]),
    "liship_set_sync_timeout": (_ci, [C.c_double]),
    "liship_csr_plan_strip_rows": (_ci, [_vp]),
    "liship_csr_plan_box27": (_ci, [_vp]),
    "liship_csr_plan_block2_march": (_ci, [_vp]),
    "liship_csr_plan_local_runs": (_ci, [_vp]),
    "liship_spmv_csr_set_local_runs": (_ci, [_ci]),
    "liship_spmv_csr_set_local_pairs": (_ci, [_ci]),
    "liship_spmv_csr_set_local_short_rows": (_ci, [_ci]),
    "liship_csr_plan_reorder": (_ci, [_vp, _vp, _vp, _vp, _ci, _vp]),
    "liship_csr_plan_reordered": (C.c_longlong, [_vp]),
    "liship_csr_plan_reorder_with": (_ci, [_vp, _vp, _vp, _vp, _ci, _vp, _vp]),
    "liship_csr_plan_reorder_permutation": (_ci, [_vp, _vp]),
    "liship_spmv_csr_set_reorder": (_ci, [_ci]),
    "liship_csr_plan_reordered_form": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_csr_plan_set_ghost_columns": (_ci, [_vp, _ci]),
    "liship_csr_plan_reordered_inner_rows": (_ci, [_vp]),
    "liship_csr_plan_lists_failed": (_ci, [_vp]),
    "liship_permute_rows_of_list": (_ci, [_ci, _vp, _ci, _vp, _vp, _vp]),
    "liship_permute_gather_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_permute_scatter_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_spmv_csr_set_long_row_tree": (_ci, [_ci]),
    "liship_spmv_csr_set_uniform_rows": (_ci, [_ci]),
    "liship_spmv_csr_set_row_block_dots": (_ci, [_ci]),
    "liship_spmv_csr_switches": (_ci, []),
    "liship_spmv_csr_set_index_codes": (_ci, [_ci]),
    "liship_spmv_csr_rows_f64": (_ci, [_vp, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_spmv_csr_rows_dot_f64": (_ci, [_vp, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _ci, _vp, _vp]),
    "liship_spmv_csr_dot_finish_f64": (_ci, [_ci, _ci, _vp, _vp, _vp]),
    "liship_spmv_csr_set_variant": (_ci, [_ci]),
    "liship_spmv_ell_f64": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp]),
    "liship_spmv_dia_f64": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]),
    "liship_spmv_ell_rows_f64": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp]),
    "liship_spmv_dia_rows_f64": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _vp]),
    "liship_spmv_ell_dot_f64": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp]),
    "liship_spmv_dia_dot_f64": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp]),
    "liship_spmv_jad_f64": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_spmv_bsr_f64": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_ell_encode_indices": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp]),
    "liship_spmv_ell_coded_f64": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp]),
    "liship_spmv_bsr_dot_f64": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp]),
    "liship_spmv_bsr_nnz_f64": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_axpy_f64": (_ci, [_ci, _cd, _vp, _vp, _vp]),
    "liship_xpay_f64": (_ci, [_ci, _vp, _cd, _vp, _vp]),
    "liship_axpyz_f64": (_ci, [_ci, _cd, _vp, _vp, _vp, _vp]),
    "liship_scale_f64": (_ci, [_ci, _cd, _vp, _vp]),
    "liship_scale_to_f64": (_ci, [_ci, _cd, _vp, _vp, _vp]),
    "liship_pmul_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_pdiv_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_set_all_f64": (_ci, [_ci, _cd, _vp, _vp]),
    "liship_abs_f64": (_ci, [_ci, _vp, _vp]),
    "liship_reciprocal_f64": (_ci, [_ci, _vp, _vp]),
    "liship_shift_f64": (_ci, [_ci, _cd, _vp, _vp]),
    "liship_rsqrt_abs_f64": (_ci, [_ci, _vp, _vp]),
    "liship_csr_scale_f64": (_ci, [_ci, _vp, _vp, _vp, _vp, _ci, _vp]),
    "liship_axpy2_f64": (_ci, [_ci, _cd, _vp, _cd, _vp, _vp, _vp]),
    "liship_axpy_xpay_f64": (_ci, [_ci, _cd, _vp, _vp, _cd, _vp, _vp]),
    "liship_cg_update_f64": (_ci, [_ci, _cd, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_cg_update_jacobi_f64": (_ci, [_ci, _cd, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_pmul_xpay_f64": (_ci, [_ci, _vp, _vp, _cd, _vp, _vp]),
    "liship_bicgstab_end_dev_f64": (_ci, [_ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_cg_direction_dev_f64": (_ci, [_ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_cg_residual_jacobi_dev_f64": (_ci, [_ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_cg_direction_uniform_dev_f64": (_ci, [_ci, _vp, _vp, _vp, _cd, _vp, _vp, _vp]),
    "liship_cg_residual_jacobi_uniform_dev_f64": (_ci, [_ci, _vp, _vp, _cd, _vp, _vp, _vp, _vp]),
    "liship_count_ne_f64": (_ci, [_ci, _vp, _cd, _vp, _vp, _vp]),
    "liship_axpy_sumsq_f64": (_ci, [_ci, _cd, _vp, _vp, _vp, _vp, _vp]),
    "liship_axpy_sumsq_dot_f64": (_ci, [_ci, _cd, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_mgs_step_f64": (_ci, [_ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_scale_inv_norm_f64": (_ci, [_ci, _vp, _vp, _vp]),
    "liship_lincomb_f64": (_ci, [_ci, _ci, _vp, _vp, _ci, _vp, _vp]),
    "liship_reduce_work_bytes": (_sz, []),
    "liship_set_reference_reductions": (_ci, [_ci]),
    "liship_get_reference_reductions": (_ci, []),
    "liship_dot_f64": (_ci, [_ci, _vp, _vp, _vp, _vp, _vp]),
    "liship_nrm2_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_sumsq_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_nrm1_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_sum_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_dot2_f64": (_ci, [_ci, _vp, _vp, _vp, _vp, _vp]),
    "liship_csr_diagonal_f64": (_ci, [_ci, _vp, _vp, _vp, _vp, _vp]),
    "liship_ell_diagonal_f64": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp]),
    "liship_dia_diagonal_f64": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp]),
    "liship_spmv_formats_set_plane": (_ci, [_ci]),
    "liship_ell_scan_band": (_ci, [_ci, _ci, _vp, _vp, _vp]),
    "liship_spmv_csr_transposed_chunked_f64": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_bsr_to_rows": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_csr_transpose_f64": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liship_gather_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_scatter_add_f64": (_ci, [_ci, _vp, _vp, _vp, _vp]),
    "liship_poisson3d_nnz": (C.c_longlong, [_ci, _ci, _ci, _ci, _ci]),
    "liship_poisson3d_csr": (_ci, [_ci, _ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]),
    "liship_poisson3d_rhs": (_ci, [_ci, _ci, _ci, _ci, _ci, _vp, _vp]),
}


def build(verbose=False):
    """Compile every HIP kernel and the C host layer into lis_amd/lib/liblis_amd.so (gfx950)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", CSRC, "-j8"], check=True, stdout=out)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build did not produce " + LIB_PATH)


class HipError(RuntimeError):
    pass


_lib = None


def load():
    """Load liblis_amd.so (no silent fallback: raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    mode = os.RTLD_NOW | os.RTLD_LOCAL
    if not os.environ.get("LIS_AMD_NO_DEEPBIND"):          # sanitizer runs need it off
        mode |= getattr(os, "RTLD_DEEPBIND", 0)
    lib = _capi.LisLib(LIB_PATH, mode=mode)
    lib.liship_missing = []
    for name, (res, args) in _LISHIP.items():
        try:
            fn = getattr(lib.dll, name)
        except AttributeError:
            lib.liship_missing.append(name)
            continue
        fn.restype, fn.argtypes = res, args
        setattr(lib, name, fn)
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().liship_error_string(code)
        raise HipError(f"liship error {code}: {msg.decode() if msg else '?'}")


class DeviceArray:
    """A typed 1-D buffer in HBM owned through liship_malloc/liship_free."""

    def __init__(self, count, dtype):
        self.dtype = np.dtype(dtype)
        self.count = int(count)
        self.nbytes = self.count * self.dtype.itemsize
        p = C.c_void_p()
        check(load().liship_malloc(C.byref(p), max(self.nbytes, 16)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        d = cls(arr.size, arr.dtype)
        d.upload(arr)
        return d

    @classmethod
    def zeros(cls, count, dtype):
        d = cls(count, dtype)
        check(load().liship_memset(d.ptr, 0, d.nbytes, None))
        return d

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.size == self.count
        check(load().liship_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes, None))
        check(load().liship_device_synchronize())

    def to_host(self, count=None):
        count = self.count if count is None else count
        out = np.empty(count, self.dtype)
        check(load().liship_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None))
        check(load().liship_device_synchronize())
        return out

    def free(self):
        if self.ptr:
            load().liship_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def gpu_available():
    try:
        n = C.c_int(0)
        return load().liship_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False
