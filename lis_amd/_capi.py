"""ctypes view of the Lis C API (include/lis.h of this repo == the reference's include/lis.h:489-1045).

Plumbing only: the same binding drives either this repo's ``liblis_amd.so`` or -- in tests -- the
reference compiled as ``oracle/_ref/liblis_ref.so``, because both export the same symbols and the
same struct layouts.  No torch, no compute here.
"""
import ctypes as C
import os

LIS_INT = C.c_int
LIS_SCALAR = C.c_double
LIS_REAL = C.c_double
P_INT = C.POINTER(C.c_int)
P_DBL = C.POINTER(C.c_double)

# constants (lis.h:175-260, 1052-1063 of the reference)
LIS_MATRIX_CSR, LIS_MATRIX_CSC, LIS_MATRIX_MSR, LIS_MATRIX_DIA, LIS_MATRIX_ELL = 1, 2, 3, 4, 5
LIS_MATRIX_JAD, LIS_MATRIX_BSR = 6, 7
LIS_INS_VALUE, LIS_ADD_VALUE = 0, 1
LIS_SUCCESS, LIS_ERR_ILL_ARG, LIS_BREAKDOWN, LIS_ERR_OUT_OF_MEMORY = 0, 1, 2, 3
LIS_MAXITER, LIS_ERR_NOT_IMPLEMENTED = 4, 5
LIS_COMM_WORLD = 1
FORMAT_ID = {"csr": 1, "csc": 2, "dia": 4, "ell": 5, "jad": 6, "bsr": 7}


class _Header(C.Structure):
    """Common prefix of LIS_VECTOR_STRUCT / LIS_MATRIX_STRUCT (lis.h:513-530, 621-638)."""
    _fields_ = [
        ("label", LIS_INT), ("status", LIS_INT), ("precision", LIS_INT), ("gn", LIS_INT),
        ("n", LIS_INT), ("np", LIS_INT), ("pad", LIS_INT), ("origin", LIS_INT),
        ("is_copy", LIS_INT), ("is_destroy", LIS_INT), ("is_scaled", LIS_INT),
        ("my_rank", LIS_INT), ("nprocs", LIS_INT), ("comm", LIS_INT),
        ("is_", LIS_INT), ("ie", LIS_INT), ("ranges", P_INT),
    ]


class Vector(C.Structure):
    _fields_ = _Header._fields_ + [
        ("value", P_DBL), ("value_lo", P_DBL), ("work", P_DBL), ("intvalue", LIS_INT),
    ]


class CommTable(C.Structure):
    _fields_ = [
        ("comm", LIS_INT), ("pad", LIS_INT), ("neibpetot", LIS_INT), ("imnnz", LIS_INT),
        ("exnnz", LIS_INT), ("wssize", LIS_INT), ("wrsize", LIS_INT),
        ("neibpe", P_INT), ("import_ptr", P_INT), ("import_index", P_INT),
        ("export_ptr", P_INT), ("export_index", P_INT), ("ws", P_DBL), ("wr", P_DBL),
    ]


class MatrixCore(C.Structure):
    """LIS_MATRIX_CORE_STRUCT (lis.h:569-589): the L / U part of a split matrix."""
    _fields_ = [(k, LIS_INT) for k in ("nnz", "ndz", "bnr", "bnc", "nr", "nc", "bnnz", "nnd", "maxnzr")] + \
               [(k, P_INT) for k in ("ptr", "row", "col", "index", "bptr", "bindex")] + [("value", P_DBL), ("work", P_DBL)]


class MatrixDiag(C.Structure):
    """LIS_MATRIX_DIAG_STRUCT (lis.h:591-619)."""
    _fields_ = [(k, LIS_INT) for k in ("label", "status", "precision", "gn", "n", "np", "pad", "origin", "is_copy", "is_destroy",
                                       "is_scaled", "my_rank", "nprocs", "comm", "is_", "ie")] + \
               [("ranges", P_INT), ("value", P_DBL), ("work", P_DBL), ("bn", LIS_INT), ("nr", LIS_INT), ("bns", P_INT), ("ptr", P_INT),
                ("v_value", C.c_void_p)]


class Matrix(C.Structure):
    _fields_ = _Header._fields_ + [
        ("matrix_type", LIS_INT), ("nnz", LIS_INT), ("ndz", LIS_INT), ("bnr", LIS_INT),
        ("bnc", LIS_INT), ("nr", LIS_INT), ("nc", LIS_INT), ("bnnz", LIS_INT),
        ("nnd", LIS_INT), ("maxnzr", LIS_INT),
        ("ptr", P_INT), ("row", P_INT), ("col", P_INT), ("index", P_INT),
        ("bptr", P_INT), ("bindex", P_INT), ("value", P_DBL), ("work", P_DBL),
        ("L", C.POINTER(MatrixCore)), ("U", C.POINTER(MatrixCore)), ("D", C.POINTER(MatrixDiag)), ("WD", C.c_void_p),
        ("is_block", LIS_INT), ("pad_comm", LIS_INT), ("is_pmat", LIS_INT),
        ("is_sorted", LIS_INT), ("is_splited", LIS_INT), ("is_save", LIS_INT),
        ("is_comm", LIS_INT), ("is_fallocated", LIS_INT), ("use_wd", LIS_INT),
        ("conv_bnr", LIS_INT), ("conv_bnc", LIS_INT),
        ("conv_row", P_INT), ("conv_col", P_INT), ("options", LIS_INT * 10),
        ("w_annz", LIS_INT), ("w_nnz", P_INT), ("w_row", P_INT),
        ("w_index", C.c_void_p), ("w_value", C.c_void_p), ("v_value", C.c_void_p),
        ("l2g_map", P_INT), ("commtable", C.POINTER(CommTable)),
    ]


LIS_OPTIONS_LEN, LIS_PARAMS_LEN = 27, 15


class Solver(C.Structure):
    _fields_ = [
        ("A", C.POINTER(Matrix)), ("Ah", C.POINTER(Matrix)),
        ("b", C.POINTER(Vector)), ("x", C.POINTER(Vector)),
        ("xx", C.POINTER(Vector)), ("d", C.POINTER(Vector)),
        ("WD", C.c_void_p), ("precon", C.c_void_p), ("work", C.c_void_p),
        ("rhistory", P_DBL), ("worklen", LIS_INT),
        ("options", LIS_INT * LIS_OPTIONS_LEN), ("params", LIS_SCALAR * LIS_PARAMS_LEN),
        ("retcode", LIS_INT), ("iter", LIS_INT), ("iter2", LIS_INT), ("resid", LIS_REAL),
        ("time", C.c_double), ("itime", C.c_double), ("ptime", C.c_double),
        ("p_c_time", C.c_double), ("p_i_time", C.c_double),
        ("precision", LIS_INT), ("bnrm", LIS_REAL), ("tol", LIS_REAL), ("tol_switch", LIS_REAL),
        ("setup", LIS_INT),
    ]


PV = C.POINTER(Vector)
PM = C.POINTER(Matrix)
PS = C.POINTER(Solver)

_PROTOS = {
    # utilities (lis.h:1030-1045)
    "lis_initialize": (LIS_INT, [C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_char_p))]),
    "lis_finalize": (LIS_INT, []),
    "lis_wtime": (C.c_double, []),
    # vectors (lis.h:824-859)
    "lis_vector_create": (LIS_INT, [LIS_INT, C.POINTER(PV)]),
    "lis_vector_set_size": (LIS_INT, [PV, LIS_INT, LIS_INT]),
    "lis_vector_destroy": (LIS_INT, [PV]),
    "lis_vector_duplicate": (LIS_INT, [C.c_void_p, C.POINTER(PV)]),
    "lis_vector_get_size": (LIS_INT, [PV, P_INT, P_INT]),
    "lis_vector_get_range": (LIS_INT, [PV, P_INT, P_INT]),
    "lis_vector_get_value": (LIS_INT, [PV, LIS_INT, P_DBL]),
    "lis_vector_get_values": (LIS_INT, [PV, LIS_INT, LIS_INT, P_DBL]),
    "lis_vector_set_value": (LIS_INT, [LIS_INT, LIS_INT, LIS_SCALAR, PV]),
    "lis_vector_set_values": (LIS_INT, [LIS_INT, LIS_INT, P_INT, P_DBL, PV]),
    "lis_vector_set_values2": (LIS_INT, [LIS_INT, LIS_INT, LIS_INT, P_DBL, PV]),
    "lis_vector_scatter": (LIS_INT, [P_DBL, PV]),
    "lis_vector_gather": (LIS_INT, [PV, P_DBL]),
    "lis_vector_is_null": (LIS_INT, [PV]),
    "lis_vector_swap": (LIS_INT, [PV, PV]),
    "lis_vector_copy": (LIS_INT, [PV, PV]),
    "lis_vector_axpy": (LIS_INT, [LIS_SCALAR, PV, PV]),
    "lis_vector_xpay": (LIS_INT, [PV, LIS_SCALAR, PV]),
    "lis_vector_axpyz": (LIS_INT, [LIS_SCALAR, PV, PV, PV]),
    "lis_vector_scale": (LIS_INT, [LIS_SCALAR, PV]),
    "lis_vector_pmul": (LIS_INT, [PV, PV, PV]),
    "lis_vector_pdiv": (LIS_INT, [PV, PV, PV]),
    "lis_vector_set_all": (LIS_INT, [LIS_SCALAR, PV]),
    "lis_vector_abs": (LIS_INT, [PV]),
    "lis_vector_reciprocal": (LIS_INT, [PV]),
    "lis_vector_conjugate": (LIS_INT, [PV]),
    "lis_vector_shift": (LIS_INT, [LIS_SCALAR, PV]),
    "lis_vector_dot": (LIS_INT, [PV, PV, P_DBL]),
    "lis_vector_nhdot": (LIS_INT, [PV, PV, P_DBL]),
    "lis_vector_nrm1": (LIS_INT, [PV, P_DBL]),
    "lis_vector_nrm2": (LIS_INT, [PV, P_DBL]),
    "lis_vector_nrmi": (LIS_INT, [PV, P_DBL]),
    "lis_vector_sum": (LIS_INT, [PV, P_DBL]),
    # matrices (lis.h:865-914)
    "lis_matrix_create": (LIS_INT, [LIS_INT, C.POINTER(PM)]),
    "lis_matrix_destroy": (LIS_INT, [PM]),
    "lis_matrix_assemble": (LIS_INT, [PM]),
    "lis_matrix_is_assembled": (LIS_INT, [PM]),
    "lis_matrix_duplicate": (LIS_INT, [PM, C.POINTER(PM)]),
    "lis_matrix_set_size": (LIS_INT, [PM, LIS_INT, LIS_INT]),
    "lis_matrix_get_size": (LIS_INT, [PM, P_INT, P_INT]),
    "lis_matrix_get_range": (LIS_INT, [PM, P_INT, P_INT]),
    "lis_matrix_get_nnz": (LIS_INT, [PM, P_INT]),
    "lis_matrix_set_type": (LIS_INT, [PM, LIS_INT]),
    "lis_matrix_get_type": (LIS_INT, [PM, P_INT]),
    "lis_matrix_set_value": (LIS_INT, [LIS_INT, LIS_INT, LIS_INT, LIS_SCALAR, PM]),
    "lis_matrix_get_diagonal": (LIS_INT, [PM, PV]),
    "lis_matrix_convert": (LIS_INT, [PM, PM]),
    "lis_matrix_split": (LIS_INT, [PM]),
    "lis_matrix_merge": (LIS_INT, [PM]),
    "lis_matrix_copy": (LIS_INT, [PM, PM]),
    "lis_matrix_set_blocksize": (LIS_INT, [PM, LIS_INT, LIS_INT, P_INT, P_INT]),
    "lis_matrix_malloc_csr": (LIS_INT, [LIS_INT, LIS_INT, C.POINTER(P_INT), C.POINTER(P_INT), C.POINTER(P_DBL)]),
    "lis_matrix_set_csr": (LIS_INT, [LIS_INT, P_INT, P_INT, P_DBL, PM]),
    "lis_matrix_malloc_csc": (LIS_INT, [LIS_INT, LIS_INT, C.POINTER(P_INT), C.POINTER(P_INT), C.POINTER(P_DBL)]),
    "lis_matrix_set_csc": (LIS_INT, [LIS_INT, P_INT, P_INT, P_DBL, PM]),
    "lis_matrix_malloc_bsr": (LIS_INT, [LIS_INT, LIS_INT, LIS_INT, LIS_INT, C.POINTER(P_INT), C.POINTER(P_INT), C.POINTER(P_DBL)]),
    "lis_matrix_set_bsr": (LIS_INT, [LIS_INT, LIS_INT, LIS_INT, P_INT, P_INT, P_DBL, PM]),
    "lis_matrix_malloc_ell": (LIS_INT, [LIS_INT, LIS_INT, C.POINTER(P_INT), C.POINTER(P_DBL)]),
    "lis_matrix_set_ell": (LIS_INT, [LIS_INT, P_INT, P_DBL, PM]),
    "lis_matrix_malloc_jad": (LIS_INT, [LIS_INT, LIS_INT, LIS_INT, C.POINTER(P_INT), C.POINTER(P_INT), C.POINTER(P_INT), C.POINTER(P_DBL)]),
    "lis_matrix_set_jad": (LIS_INT, [LIS_INT, LIS_INT, P_INT, P_INT, P_INT, P_DBL, PM]),
    "lis_matrix_malloc_dia": (LIS_INT, [LIS_INT, LIS_INT, C.POINTER(P_INT), C.POINTER(P_DBL)]),
    "lis_matrix_set_dia": (LIS_INT, [LIS_INT, P_INT, P_DBL, PM]),
    # matvec (lis.h:920)
    "lis_matvec": (LIS_INT, [PM, PV, PV]),
    "lis_matvech": (LIS_INT, [PM, PV, PV]),
    "lis_matvec_optimize": (LIS_INT, [PM, C.POINTER(LIS_INT)]),
    "lis_matrix_scale": (LIS_INT, [PM, PV, PV, LIS_INT]),
    "lis_matrix_set_values": (LIS_INT, [LIS_INT, LIS_INT, P_DBL, PM]),
    "lis_matrix_malloc": (LIS_INT, [PM, LIS_INT, C.POINTER(LIS_INT)]),
    # solvers (lis.h:961-984)
    "lis_solver_create": (LIS_INT, [C.POINTER(PS)]),
    "lis_solver_destroy": (LIS_INT, [PS]),
    "lis_solver_get_iter": (LIS_INT, [PS, P_INT]),
    "lis_solver_get_iterex": (LIS_INT, [PS, P_INT, P_INT, P_INT]),
    "lis_solver_get_time": (LIS_INT, [PS, P_DBL]),
    "lis_solver_get_timeex": (LIS_INT, [PS, P_DBL, P_DBL, P_DBL, P_DBL, P_DBL]),
    "lis_solver_get_residualnorm": (LIS_INT, [PS, P_DBL]),
    "lis_solver_get_solver": (LIS_INT, [PS, P_INT]),
    "lis_solver_get_precon": (LIS_INT, [PS, P_INT]),
    "lis_solver_get_status": (LIS_INT, [PS, P_INT]),
    "lis_solver_get_rhistory": (LIS_INT, [PS, PV]),
    "lis_solver_set_option": (LIS_INT, [C.c_char_p, PS]),
    "lis_solver_set_optionC": (LIS_INT, [PS]),
    "lis_solve": (LIS_INT, [PM, PV, PV, PS]),
    "lis_output_vector": (LIS_INT, [PV, LIS_INT, C.c_char_p]),
    "lis_input": (LIS_INT, [PM, PV, PV, C.c_char_p]),
    "lis_input_matrix": (LIS_INT, [PM, C.c_char_p]),
    "lis_input_vector": (LIS_INT, [PV, C.c_char_p]),
    "lis_output": (LIS_INT, [PM, PV, PV, LIS_INT, C.c_char_p]),
    "lis_output_matrix": (LIS_INT, [PM, LIS_INT, C.c_char_p]),
    "lis_solver_output_rhistory": (LIS_INT, [PS, C.c_char_p]),
    "lis_solver_get_solvername": (LIS_INT, [LIS_INT, C.c_char_p]),
    "lis_solver_get_preconname": (LIS_INT, [LIS_INT, C.c_char_p]),
    # memory (lis.h:1037-1042)
    "lis_malloc": (C.c_void_p, [C.c_size_t, C.c_char_p]),
    "lis_free": (None, [C.c_void_p]),
}


class LisLib:
    """A loaded library exporting the Lis C API."""

    def __init__(self, path, mode=None):
        if mode is None:
            mode = getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2)
        self.path = path
        self.dll = C.CDLL(path, mode=mode)
        self.missing = []
        for name, (res, args) in _PROTOS.items():
            try:
                fn = getattr(self.dll, name)
            except AttributeError:
                self.missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def initialize(self, args=()):
        argv_list = [b"lis"] + [a.encode() if isinstance(a, str) else a for a in args]
        argc = C.c_int(len(argv_list))
        arr = (C.c_char_p * (len(argv_list) + 1))(*argv_list, None)
        self._argv_keep = arr
        argv = C.cast(arr, C.POINTER(C.c_char_p))
        pargv = C.pointer(argv)
        self._pargv_keep = pargv
        return self.lis_initialize(C.byref(argc), pargv)
