/*
 * lis.h -- the Lis C API as served by liblis_amd.so (MI355X-native SpMV + Krylov hot path).
 *
 * Drop-in contract: every type, constant and function below has the name, argument meaning, return
 * convention and -- for the structs -- the byte layout of anishida/lis 2.1.11 (reference
 * include/lis.h; the line each item mirrors is cited as "ref:NNN").  A program written against the
 * reference header compiles against this one and links against liblis_amd.so instead of liblis.
 * Only the slice on the hot path and the rows next to it are provided (six storage
 * formats, A x and A^T x, the Krylov solvers, none/Jacobi preconditioning, scaling, Matrix Market / Harwell-Boeing
 * files); anything else is absent, and what exists in the enumerations but is not served (MSR/BSC/VBR/COO/DNS
 * storage, Gauss-Seidel / SOR, the ILU-family preconditioners) returns LIS_ERR_NOT_IMPLEMENTED at run time
 * exactly where the reference would dispatch to it.
 *
 * Default build of the reference is assumed: LIS_INT = int (ref:461), LIS_SCALAR = LIS_REAL = double
 * (ref:446-447), no MPI (LIS_Comm = LIS_INT, ref:485), no quad precision.
 *
 * Where the data lives is an extension, see lis_amd.h.
 */
#ifndef __LIS_H__
#define __LIS_H__

#include <stdio.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIS_VERSION "2.1.11"          /* ref:36  API level served */

typedef int          LIS_INT;          /* ref:461 */
typedef unsigned int LIS_UNSIGNED_INT; /* ref:462 */
typedef double       LIS_SCALAR;       /* ref:446 */
typedef double       LIS_REAL;         /* ref:447 */
typedef LIS_INT      LIS_Comm;         /* ref:485 */
#define LIS_COMM_WORLD ((LIS_Comm)0x1) /* ref:486 */

#define _max(a,b) ((a) >= (b) ? (a) : (b))   /* ref:52-53: drivers use them */
#define _min(a,b) ((a) <= (b) ? (a) : (b))

/* ---- return codes, ref:1050-1063 ------------------------------------------------------------ */
#define LIS_TRUE  1
#define LIS_FALSE 0
#define LIS_FAILS (-1)
#define LIS_SUCCESS 0
#define LIS_ILL_OPTION 1
#define LIS_ERR_ILL_ARG 1
#define LIS_BREAKDOWN 2
#define LIS_OUT_OF_MEMORY 3
#define LIS_ERR_OUT_OF_MEMORY 3
#define LIS_MAXITER 4
#define LIS_ERR_NOT_IMPLEMENTED 5
#define LIS_ERR_FILE_IO 6

/* ---- solver option slots, ref:67-132 -------------------------------------------------------- */
#define LIS_OPTIONS_LEN 27
#define LIS_OPTIONS_SOLVER 0
#define LIS_OPTIONS_PRECON 1
#define LIS_OPTIONS_MAXITER 2
#define LIS_OPTIONS_OUTPUT 3
#define LIS_OPTIONS_RESTART 4
#define LIS_OPTIONS_ELL 5
#define LIS_OPTIONS_SCALE 6
#define LIS_OPTIONS_FILL 7
#define LIS_OPTIONS_M 8
#define LIS_OPTIONS_PSOLVER 9
#define LIS_OPTIONS_PMAXITER 10
#define LIS_OPTIONS_PRESTART 11
#define LIS_OPTIONS_PELL 12
#define LIS_OPTIONS_PPRECON 13
#define LIS_OPTIONS_ISLEVEL 14
#define LIS_OPTIONS_INITGUESS_ZEROS 15
#define LIS_OPTIONS_ADDS 16
#define LIS_OPTIONS_ADDS_ITER 17
#define LIS_OPTIONS_PRECISION 18
#define LIS_OPTIONS_USE_AT 19
#define LIS_OPTIONS_SWITCH_MAXITER 20
#define LIS_OPTIONS_SAAMG_UNSYM 21
#define LIS_OPTIONS_STORAGE 22
#define LIS_OPTIONS_STORAGE_BLOCK 23
#define LIS_OPTIONS_CONV_COND 24
#define LIS_OPTIONS_INIT_SHADOW_RESID 25
#define LIS_OPTIONS_IDRS_RESTART 26

#define LIS_PARAMS_LEN 15
#define LIS_PARAMS_RESID        (LIS_OPTIONS_LEN+0)
#define LIS_PARAMS_OMEGA        (LIS_OPTIONS_LEN+1)
#define LIS_PARAMS_RELAX        (LIS_OPTIONS_LEN+2)
#define LIS_PARAMS_DROP         (LIS_OPTIONS_LEN+3)
#define LIS_PARAMS_ALPHA        (LIS_OPTIONS_LEN+4)
#define LIS_PARAMS_TAU          (LIS_OPTIONS_LEN+5)
#define LIS_PARAMS_SIGMA        (LIS_OPTIONS_LEN+6)
#define LIS_PARAMS_GAMMA        (LIS_OPTIONS_LEN+7)
#define LIS_PARAMS_SSOR_OMEGA   (LIS_OPTIONS_LEN+8)
#define LIS_PARAMS_PRESID       (LIS_OPTIONS_LEN+9)
#define LIS_PARAMS_POMEGA       (LIS_OPTIONS_LEN+10)
#define LIS_PARAMS_SWITCH_RESID (LIS_OPTIONS_LEN+11)
#define LIS_PARAMS_RATE         (LIS_OPTIONS_LEN+12)
#define LIS_PARAMS_RESID_WEIGHT (LIS_OPTIONS_LEN+13)
#define LIS_PARAMS_SAAMG_THETA  (LIS_OPTIONS_LEN+14)

#define LIS_PRINT_NONE 0              /* ref:142-145 */
#define LIS_PRINT_MEM 1
#define LIS_PRINT_OUT 2
#define LIS_PRINT_ALL 3

#define LIS_SCALE_NONE 0              /* ref:152-154 */
#define LIS_SCALE_JACOBI 1
#define LIS_SCALE_SYMM_DIAG 2

#define LIS_CONV_COND_DEFAULT 0       /* ref:156-159 */
#define LIS_CONV_COND_NRM2_R 0
#define LIS_CONV_COND_NRM2_B 1
#define LIS_CONV_COND_NRM1_B 2

/* solver ids, ref:161-187 (served: CG, BICGSTAB, GMRES) */
#define LIS_SOLVER_LEN 25
#define LIS_SOLVER_CG 1
#define LIS_SOLVER_BICG 2
#define LIS_SOLVER_CGS 3
#define LIS_SOLVER_BICGSTAB 4
#define LIS_SOLVER_BICGSTABL 5
#define LIS_SOLVER_GPBICG 6
#define LIS_SOLVER_TFQMR 7
#define LIS_SOLVER_ORTHOMIN 8
#define LIS_SOLVER_GMRES 9
#define LIS_SOLVER_JACOBI 10
#define LIS_SOLVER_GS 11
#define LIS_SOLVER_SOR 12
#define LIS_SOLVER_BICGSAFE 13
#define LIS_SOLVER_CR 14
#define LIS_SOLVER_BICR 15
#define LIS_SOLVER_CRS 16
#define LIS_SOLVER_BICRSTAB 17
#define LIS_SOLVER_GPBICR 18
#define LIS_SOLVER_BICRSAFE 19
#define LIS_SOLVER_FGMRES 20
#define LIS_SOLVER_IDRS 21
#define LIS_SOLVER_IDR1 22
#define LIS_SOLVER_MINRES 23
#define LIS_SOLVER_COCG 24
#define LIS_SOLVER_COCR 25

#define LIS_FMT_AUTO 0                /* ref:55-64; served: PLAIN/MM/LIS(ascii) vectors, MM/MMB matrices */
#define LIS_FMT_PLAIN 1
#define LIS_FMT_MM 2
#define LIS_FMT_LIS 3
#define LIS_FMT_LIS_ASCII 3
#define LIS_FMT_LIS_BINARY 4
#define LIS_FMT_FREE 5
#define LIS_FMT_ITBL 6
#define LIS_FMT_HB 7
#define LIS_FMT_MMB 8
#define LIS_BINARY_BIG 0              /* ref:66-67 */
#define LIS_BINARY_LITTLE 1

#define LIS_INS_VALUE 0               /* ref:207-209 */
#define LIS_ADD_VALUE 1
#define LIS_SUB_VALUE 2

#define LIS_ORIGIN_0 0                /* ref:215-216 */
#define LIS_ORIGIN_1 1

#define LIS_RESID 0                   /* ref:218-219 */
#define LIS_RANDOM 1

#define LIS_PRECISION_DEFAULT 0       /* ref:221-224 */
#define LIS_PRECISION_DOUBLE 0
#define LIS_PRECISION_QUAD 1
#define LIS_PRECISION_SWITCH 2

#define LIS_LABEL_VECTOR 0            /* ref:226-227 */
#define LIS_LABEL_MATRIX 1

#define LIS_VECTOR_NULL (-1)          /* ref:231-233 */
#define LIS_VECTOR_ASSEMBLING 0
#define LIS_VECTOR_ASSEMBLED 1

/* preconditioner ids, ref:238-251 (served: NONE, JACOBI) */
#define LIS_PRECONNAME_MAX 10
#define LIS_PRECON_TYPE_LEN 12
#define LIS_PRECON_TYPE_NONE 0
#define LIS_PRECON_TYPE_JACOBI 1
#define LIS_PRECON_TYPE_ILU 2
#define LIS_PRECON_TYPE_SSOR 3
#define LIS_PRECON_TYPE_HYBRID 4
#define LIS_PRECON_TYPE_IS 5
#define LIS_PRECON_TYPE_SAI 6
#define LIS_PRECON_TYPE_SAAMG 7
#define LIS_PRECON_TYPE_ILUC 8
#define LIS_PRECON_TYPE_ILUT 9
#define LIS_PRECON_TYPE_BJACOBI 10
#define LIS_PRECON_TYPE_ADDS 11

/* storage formats, ref:253-283 (served: CSR, CSC, DIA, ELL, JAD, BSR) */
#define LIS_MATRIX_ASSEMBLING 0
#define LIS_MATRIX_CSR 1
#define LIS_MATRIX_CSC 2
#define LIS_MATRIX_MSR 3
#define LIS_MATRIX_DIA 4
#define LIS_MATRIX_CDS 4
#define LIS_MATRIX_ELL 5
#define LIS_MATRIX_JAD 6
#define LIS_MATRIX_BSR 7
#define LIS_MATRIX_BSC 8
#define LIS_MATRIX_VBR 9
#define LIS_MATRIX_COO 10
#define LIS_MATRIX_DENSE 11
#define LIS_MATRIX_DNS 11
#define LIS_MATRIX_RCO 255
#define LIS_MATRIX_DECIDING_SIZE (-(LIS_MATRIX_RCO+1))
#define LIS_MATRIX_NULL          (-(LIS_MATRIX_RCO+2))
#define LIS_MATRIX_DEFAULT LIS_MATRIX_CSR
#define LIS_MATRIX_POINT   LIS_MATRIX_CSR
#define LIS_MATRIX_BLOCK   LIS_MATRIX_BSR

/* ---- objects.  Field order and types are the ABI: drivers read A->n, A->nnz, A->ptr, v->value ... --- */

/* ref:489-509 halo tables of a row-block partition */
struct LIS_COMMTABLE_STRUCT
{
	LIS_Comm comm;
	LIS_INT pad;
	LIS_INT neibpetot;       /* number of neighbour ranks */
	LIS_INT imnnz;           /* ghost entries received */
	LIS_INT exnnz;           /* owned entries sent */
	LIS_INT wssize;
	LIS_INT wrsize;
	LIS_INT *neibpe;         /* neighbour ranks, ascending */
	LIS_INT *import_ptr;     /* [neibpetot+1] */
	LIS_INT *import_index;   /* local ghost slots n..np-1 */
	LIS_INT *export_ptr;     /* [neibpetot+1] */
	LIS_INT *export_index;   /* local owned rows to pack */
	LIS_SCALAR *ws;
	LIS_SCALAR *wr;
};
typedef struct LIS_COMMTABLE_STRUCT *LIS_COMMTABLE;

/* ref:513-537 */
struct LIS_VECTOR_STRUCT
{
	LIS_INT label;
	LIS_INT status;
	LIS_INT precision;
	LIS_INT gn;
	LIS_INT n;
	LIS_INT np;
	LIS_INT pad;
	LIS_INT origin;
	LIS_INT is_copy;
	LIS_INT is_destroy;
	LIS_INT is_scaled;
	LIS_INT my_rank;
	LIS_INT nprocs;
	LIS_Comm comm;
	LIS_INT is;
	LIS_INT ie;
	LIS_INT *ranges;
	LIS_SCALAR *value;
	LIS_SCALAR *value_lo;
	LIS_SCALAR *work;
	LIS_INT intvalue;
};
typedef struct LIS_VECTOR_STRUCT *LIS_VECTOR;

#define LIS_MATRIX_OPTION_LEN 10

/* ref:569-589: the strictly lower / upper part of a split matrix (lis_matrix_split), in the layout of the matrix's own format */
struct LIS_MATRIX_CORE_STRUCT
{
	LIS_INT nnz, ndz, bnr, bnc, nr, nc, bnnz, nnd, maxnzr;
	LIS_INT *ptr, *row, *col, *index, *bptr, *bindex;
	LIS_SCALAR *value, *work;
};
typedef struct LIS_MATRIX_CORE_STRUCT *LIS_MATRIX_CORE;

/* ref:591-619 */
struct LIS_MATRIX_DIAG_STRUCT
{
	LIS_INT label, status, precision, gn, n, np, pad, origin, is_copy, is_destroy, is_scaled;
	LIS_INT my_rank, nprocs;
	LIS_Comm comm;
	LIS_INT is, ie;
	LIS_INT *ranges;
	LIS_SCALAR *value;
	LIS_SCALAR *work;
	LIS_INT bn, nr;
	LIS_INT *bns, *ptr;
	LIS_SCALAR **v_value;
};
typedef struct LIS_MATRIX_DIAG_STRUCT *LIS_MATRIX_DIAG;

/* ref:621-690 */
struct LIS_MATRIX_STRUCT
{
	LIS_INT label;
	LIS_INT status;
	LIS_INT precision;
	LIS_INT gn;
	LIS_INT n;
	LIS_INT np;
	LIS_INT pad;
	LIS_INT origin;
	LIS_INT is_copy;
	LIS_INT is_destroy;
	LIS_INT is_scaled;
	LIS_INT my_rank;
	LIS_INT nprocs;
	LIS_Comm comm;
	LIS_INT is;
	LIS_INT ie;
	LIS_INT *ranges;

	LIS_INT matrix_type;
	LIS_INT nnz;      /* CSR,CSC,JAD */
	LIS_INT ndz;
	LIS_INT bnr;      /* BSR */
	LIS_INT bnc;      /* BSR */
	LIS_INT nr;       /* BSR */
	LIS_INT nc;       /* BSR */
	LIS_INT bnnz;     /* BSR */
	LIS_INT nnd;      /* DIA */
	LIS_INT maxnzr;   /* ELL,JAD */
	LIS_INT *ptr;     /* CSR,CSC,JAD */
	LIS_INT *row;     /* JAD */
	LIS_INT *col;
	LIS_INT *index;   /* CSR,CSC,DIA,ELL,JAD */
	LIS_INT *bptr;    /* BSR */
	LIS_INT *bindex;  /* BSR */
	LIS_SCALAR *value;
	LIS_SCALAR *work;

	LIS_MATRIX_CORE L;
	LIS_MATRIX_CORE U;
	LIS_MATRIX_DIAG D;
	LIS_MATRIX_DIAG WD;

	LIS_INT is_block;
	LIS_INT pad_comm;
	LIS_INT is_pmat;
	LIS_INT is_sorted;
	LIS_INT is_splited;
	LIS_INT is_save;
	LIS_INT is_comm;
	LIS_INT is_fallocated;
	LIS_INT use_wd;
	LIS_INT conv_bnr;
	LIS_INT conv_bnc;
	LIS_INT *conv_row;
	LIS_INT *conv_col;
	LIS_INT options[LIS_MATRIX_OPTION_LEN];

	LIS_INT w_annz;
	LIS_INT *w_nnz;
	LIS_INT *w_row;
	LIS_INT **w_index;
	LIS_SCALAR **w_value;
	LIS_SCALAR ***v_value;

	LIS_INT *l2g_map;
	LIS_COMMTABLE commtable;
};
typedef struct LIS_MATRIX_STRUCT *LIS_MATRIX;

struct LIS_MATRIX_ILU_STRUCT;                      /* ref:693-704, opaque here */
typedef struct LIS_MATRIX_ILU_STRUCT *LIS_MATRIX_ILU;
struct LIS_SOLVER_STRUCT;

/* ref:706-728 */
struct LIS_PRECON_STRUCT
{
	LIS_INT precon_type;
	LIS_MATRIX A;
	LIS_MATRIX Ah;
	LIS_MATRIX_ILU L;
	LIS_MATRIX_ILU U;
	LIS_MATRIX_DIAG WD;
	LIS_VECTOR D;            /* Jacobi: 1/diag(A) */
	LIS_VECTOR Pb;
	LIS_VECTOR temp;
	LIS_REAL theta;
	LIS_VECTOR *work;
	struct LIS_SOLVER_STRUCT *solver;
	LIS_INT worklen;
	LIS_INT level_num;
	LIS_INT wsize;
	LIS_INT solver_comm;
	LIS_INT my_rank;
	LIS_INT nprocs;
	LIS_INT is_copy;
	LIS_COMMTABLE commtable;
};
typedef struct LIS_PRECON_STRUCT *LIS_PRECON;

/* ref:731-758 */
struct LIS_SOLVER_STRUCT
{
	LIS_MATRIX A,Ah;
	LIS_VECTOR b,x,xx,d;
	LIS_MATRIX_DIAG WD;
	LIS_PRECON precon;
	LIS_VECTOR *work;
	LIS_REAL *rhistory;
	LIS_INT worklen;
	LIS_INT options[LIS_OPTIONS_LEN];
	LIS_SCALAR params[LIS_PARAMS_LEN];
	LIS_INT retcode;
	LIS_INT iter;
	LIS_INT iter2;
	LIS_REAL resid;
	double time;
	double itime;
	double ptime;
	double p_c_time;
	double p_i_time;
	LIS_INT precision;
	LIS_REAL bnrm;
	LIS_REAL tol;
	LIS_REAL tol_switch;
	LIS_INT setup;
};
typedef struct LIS_SOLVER_STRUCT *LIS_SOLVER;

/* kernel-level SpMV entry points, ref include/lis_matvec.h:76,91-181: raw HOST arrays in, raw host
 * array out; the matrix is the (cached) device copy of A */
typedef void (*LIS_MATVEC_XXX)(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);
typedef LIS_INT (*LIS_MATVEC_FUNC)(LIS_MATRIX A, LIS_VECTOR X, LIS_VECTOR Y);
/* the dispatch pointers of the reference (include/lis_matvec.h:84-85, src/matvec/lis_matvec.c:50-51: only the hybrid preconditioner ever swaps them) */
extern LIS_MATVEC_FUNC LIS_MATVEC;
extern LIS_MATVEC_FUNC LIS_MATVECH;
/* the reference's storage-format auto-tuner (src/matvec/lis_matvec.c:354-461, undeclared there too): converts A to every format the library serves, times
 * 1e7 / nnz + 1 products of each (device time, one synchronize per format) and returns the fastest in *matrix_type_maxperf; prints the reference's report lines */
LIS_INT lis_matvec_optimize(LIS_MATRIX A, LIS_INT *matrix_type_maxperf);
void lis_matvec_csr(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);   /* src/matvec/lis_matvec_csr.c:53 */
void lis_matvec_csc(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);   /* src/matvec/lis_matvec_csc.c:53 */
void lis_matvec_ell(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);   /* src/matvec/lis_matvec_ell.c:50 */
void lis_matvec_dia(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);   /* src/matvec/lis_matvec_dia.c:50 */
void lis_matvec_jad(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);   /* src/matvec/lis_matvec_jad.c:51 */
void lis_matvec_bsr(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);   /* src/matvec/lis_matvec_bsr.c:57 */
/* y = A^T x (A^H; real build).  The reference scatters; here A^T is kept as a second CSR whose rows list the
 * contributions in the reference's scatter order, so the result has the reference's bits (1 thread). */
void lis_matvech_csr(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);  /* src/matvec/lis_matvec_csr.c:113 */
void lis_matvech_csc(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);  /* src/matvec/lis_matvec_csc.c:148 */
void lis_matvech_ell(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);  /* src/matvec/lis_matvec_ell.c:133 */
void lis_matvech_dia(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);  /* src/matvec/lis_matvec_dia.c:177 */
void lis_matvech_jad(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);  /* src/matvec/lis_matvec_jad.c:484 */
void lis_matvech_bsr(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]);  /* src/matvec/lis_matvec_bsr.c:860 */

/* ---- vectors, ref:824-859 (src/vector/lis_vector.c, lis_vector_ops.c, lis_vector_opv.c) ---------- */
LIS_INT lis_vector_create(LIS_Comm comm, LIS_VECTOR *vec);
LIS_INT lis_vector_set_size(LIS_VECTOR vec, LIS_INT local_n, LIS_INT global_n);
LIS_INT lis_vector_destroy(LIS_VECTOR vec);
LIS_INT lis_vector_duplicate(void *vin, LIS_VECTOR *vout);          /* vin: a vector OR a matrix */
LIS_INT lis_vector_get_size(LIS_VECTOR v, LIS_INT *local_n, LIS_INT *global_n);
LIS_INT lis_vector_get_range(LIS_VECTOR v, LIS_INT *is, LIS_INT *ie);
LIS_INT lis_vector_get_value(LIS_VECTOR v, LIS_INT i, LIS_SCALAR *value);
LIS_INT lis_vector_get_values(LIS_VECTOR v, LIS_INT start, LIS_INT count, LIS_SCALAR value[]);
LIS_INT lis_vector_set_value(LIS_INT flag, LIS_INT i, LIS_SCALAR value, LIS_VECTOR v);
LIS_INT lis_vector_set_values(LIS_INT flag, LIS_INT count, LIS_INT index[], LIS_SCALAR value[], LIS_VECTOR v);
LIS_INT lis_vector_set_values2(LIS_INT flag, LIS_INT start, LIS_INT count, LIS_SCALAR value[], LIS_VECTOR v);
LIS_INT lis_vector_print(LIS_VECTOR x);
LIS_INT lis_vector_scatter(LIS_SCALAR value[], LIS_VECTOR v);
LIS_INT lis_vector_gather(LIS_VECTOR v, LIS_SCALAR value[]);
LIS_INT lis_vector_is_null(LIS_VECTOR v);
LIS_INT lis_vector_swap(LIS_VECTOR vsrc, LIS_VECTOR vdst);
LIS_INT lis_vector_copy(LIS_VECTOR vsrc, LIS_VECTOR vdst);
LIS_INT lis_vector_axpy(LIS_SCALAR alpha, LIS_VECTOR vx, LIS_VECTOR vy);
LIS_INT lis_vector_xpay(LIS_VECTOR vx, LIS_SCALAR alpha, LIS_VECTOR vy);
LIS_INT lis_vector_axpyz(LIS_SCALAR alpha, LIS_VECTOR vx, LIS_VECTOR vy, LIS_VECTOR vz);
LIS_INT lis_vector_scale(LIS_SCALAR alpha, LIS_VECTOR vx);
LIS_INT lis_vector_pmul(LIS_VECTOR vx, LIS_VECTOR vy, LIS_VECTOR vz);
LIS_INT lis_vector_pdiv(LIS_VECTOR vx, LIS_VECTOR vy, LIS_VECTOR vz);
LIS_INT lis_vector_set_all(LIS_SCALAR alpha, LIS_VECTOR vx);
LIS_INT lis_vector_abs(LIS_VECTOR vx);
LIS_INT lis_vector_reciprocal(LIS_VECTOR vx);
LIS_INT lis_vector_conjugate(LIS_VECTOR vx);
LIS_INT lis_vector_shift(LIS_SCALAR sigma, LIS_VECTOR vx);
LIS_INT lis_vector_dot(LIS_VECTOR vx, LIS_VECTOR vy, LIS_SCALAR *value);
LIS_INT lis_vector_nhdot(LIS_VECTOR vx, LIS_VECTOR vy, LIS_SCALAR *value);
LIS_INT lis_vector_nrm1(LIS_VECTOR vx, LIS_REAL *value);
LIS_INT lis_vector_nrm2(LIS_VECTOR vx, LIS_REAL *value);
LIS_INT lis_vector_nrmi(LIS_VECTOR vx, LIS_REAL *value);
LIS_INT lis_vector_sum(LIS_VECTOR vx, LIS_SCALAR *value);

/* ---- matrices, ref:865-914 (src/matrix/lis_matrix.c, lis_matrix_ops.c, lis_matrix_<fmt>.c) -------- */
LIS_INT lis_matrix_create(LIS_Comm comm, LIS_MATRIX *Amat);
LIS_INT lis_matrix_destroy(LIS_MATRIX Amat);
LIS_INT lis_matrix_assemble(LIS_MATRIX A);
LIS_INT lis_matrix_is_assembled(LIS_MATRIX A);
LIS_INT lis_matrix_duplicate(LIS_MATRIX Ain, LIS_MATRIX *Aout);
LIS_INT lis_matrix_set_size(LIS_MATRIX A, LIS_INT local_n, LIS_INT global_n);
LIS_INT lis_matrix_get_size(LIS_MATRIX A, LIS_INT *local_n, LIS_INT *global_n);
LIS_INT lis_matrix_get_range(LIS_MATRIX A, LIS_INT *is, LIS_INT *ie);
LIS_INT lis_matrix_get_nnz(LIS_MATRIX A, LIS_INT *nnz);
LIS_INT lis_matrix_set_type(LIS_MATRIX A, LIS_INT matrix_type);
LIS_INT lis_matrix_get_type(LIS_MATRIX A, LIS_INT *matrix_type);
LIS_INT lis_matrix_set_value(LIS_INT flag, LIS_INT i, LIS_INT j, LIS_SCALAR value, LIS_MATRIX A);
LIS_INT lis_matrix_set_values(LIS_INT flag, LIS_INT n, LIS_SCALAR value[], LIS_MATRIX A);   /* ref:878, lis_matrix.c:808 (dense block) */
LIS_INT lis_matrix_malloc(LIS_MATRIX A, LIS_INT nnz_row, LIS_INT nnz[]);                  /* ref:880, lis_matrix.c:592 (row capacity hint) */
LIS_INT lis_matrix_get_diagonal(LIS_MATRIX A, LIS_VECTOR d);
LIS_INT lis_matrix_scale(LIS_MATRIX A, LIS_VECTOR B, LIS_VECTOR D, LIS_INT action);   /* ref:882, src/matrix/lis_matrix_ops.c:579 (-scale) */
LIS_INT lis_matrix_convert(LIS_MATRIX Ain, LIS_MATRIX Aout);
/* A = L + D + U in A->L / A->D / A->U (is_splited): lis_matvec then adds D x first, the L entries, the U entries.  The
 * reference keeps these two in its internal header (src/matrix lis_matrix.h; lis_matrix_ops.c:860, :1052) */
LIS_INT lis_matrix_split(LIS_MATRIX A);
LIS_INT lis_matrix_merge(LIS_MATRIX A);
LIS_INT lis_matrix_copy(LIS_MATRIX Ain, LIS_MATRIX Aout);
LIS_INT lis_matrix_set_blocksize(LIS_MATRIX A, LIS_INT bnr, LIS_INT bnc, LIS_INT row[], LIS_INT col[]);
LIS_INT lis_matrix_unset(LIS_MATRIX A);
LIS_INT lis_matrix_set_destroyflag(LIS_MATRIX A, LIS_INT flag);
/* lis_matrix_malloc_<fmt>: DIFFERENT FROM THE REFERENCE IN ONE RESPECT.  The arrays live on pages of the library's own (not malloc memory): release them with
 * lis_free() or lis_matrix_destroy() -- NEVER free() --, and while the adopting matrix has a copy in HBM they are read-only: a store into them is caught (one page
 * fault) and the next product uses the new values, but read(2) / fread / MPI_Recv INTO them fails with EFAULT until lis_amd_matrix_host_modified(A) or
 * lis_matrix_unset(A) opens them.  LIS_AMD_PLAIN_MALLOC=1 / lis_amd_set_matrix_pages(0) (lis_amd.h) gives plain malloc memory instead. */
LIS_INT lis_matrix_malloc_csr(LIS_INT n, LIS_INT nnz, LIS_INT **ptr, LIS_INT **index, LIS_SCALAR **value);
LIS_INT lis_matrix_set_csr(LIS_INT nnz, LIS_INT *ptr, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A);
LIS_INT lis_matrix_malloc_csc(LIS_INT n, LIS_INT nnz, LIS_INT **ptr, LIS_INT **index, LIS_SCALAR **value);
LIS_INT lis_matrix_set_csc(LIS_INT nnz, LIS_INT *ptr, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A);
LIS_INT lis_matrix_malloc_bsr(LIS_INT n, LIS_INT bnr, LIS_INT bnc, LIS_INT bnnz, LIS_INT **bptr, LIS_INT **bindex, LIS_SCALAR **value);
LIS_INT lis_matrix_set_bsr(LIS_INT bnr, LIS_INT bnc, LIS_INT bnnz, LIS_INT *bptr, LIS_INT *bindex, LIS_SCALAR *value, LIS_MATRIX A);
LIS_INT lis_matrix_malloc_ell(LIS_INT n, LIS_INT maxnzr, LIS_INT **index, LIS_SCALAR **value);
LIS_INT lis_matrix_set_ell(LIS_INT maxnzr, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A);
LIS_INT lis_matrix_malloc_jad(LIS_INT n, LIS_INT nnz, LIS_INT maxnzr, LIS_INT **perm, LIS_INT **ptr, LIS_INT **index, LIS_SCALAR **value);
LIS_INT lis_matrix_set_jad(LIS_INT nnz, LIS_INT maxnzr, LIS_INT *perm, LIS_INT *ptr, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A);
LIS_INT lis_matrix_malloc_dia(LIS_INT n, LIS_INT nnd, LIS_INT **index, LIS_SCALAR **value);
LIS_INT lis_matrix_set_dia(LIS_INT nnd, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A);

/* ---- matrix-vector product, ref:920 (src/matvec/lis_matvec.c:55) ------------------------------- */
LIS_INT lis_matvec(LIS_MATRIX A, LIS_VECTOR x, LIS_VECTOR y);
LIS_INT lis_matvech(LIS_MATRIX A, LIS_VECTOR x, LIS_VECTOR y);        /* ref:921, src/matvec/lis_matvec.c:191 (BiCG's A^H p~) */

/* ---- linear solvers, ref:961-984 (src/solver/lis_solver.c, _cg.c, _bicgstab.c, _gmres.c) ---------- */
LIS_INT lis_solver_create(LIS_SOLVER *solver);
LIS_INT lis_solver_destroy(LIS_SOLVER solver);
LIS_INT lis_solver_get_iter(LIS_SOLVER solver, LIS_INT *iter);
LIS_INT lis_solver_get_iterex(LIS_SOLVER solver, LIS_INT *iter, LIS_INT *iter_double, LIS_INT *iter_quad);
LIS_INT lis_solver_get_time(LIS_SOLVER solver, double *time);
LIS_INT lis_solver_get_timeex(LIS_SOLVER solver, double *time, double *itime, double *ptime, double *p_c_time, double *p_i_time);
LIS_INT lis_solver_get_residualnorm(LIS_SOLVER solver, LIS_REAL *residual);
LIS_INT lis_solver_get_solver(LIS_SOLVER solver, LIS_INT *nsol);
LIS_INT lis_solver_get_precon(LIS_SOLVER solver, LIS_INT *precon_type);
LIS_INT lis_solver_get_status(LIS_SOLVER solver, LIS_INT *status);
LIS_INT lis_solver_get_rhistory(LIS_SOLVER solver, LIS_VECTOR v);
LIS_INT lis_solver_set_option(char *text, LIS_SOLVER solver);
LIS_INT lis_solver_set_optionC(LIS_SOLVER solver);
LIS_INT lis_solver_set_matrix(LIS_MATRIX A, LIS_SOLVER solver);
LIS_INT lis_solve(LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x, LIS_SOLVER solver);
LIS_INT lis_solve_kernel(LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x, LIS_SOLVER solver, LIS_PRECON precon);
LIS_INT lis_precon_create(LIS_SOLVER solver, LIS_PRECON *precon);       /* src/precon/lis_precon.c:119 */
LIS_INT lis_precon_destroy(LIS_PRECON precon);
LIS_INT lis_solver_get_solvername(LIS_INT solver, char *solvername);
LIS_INT lis_solver_get_preconname(LIS_INT precon_type, char *preconname);
LIS_INT lis_solver_output_rhistory(LIS_SOLVER solver, char *filename);  /* ref:1022, src/system/lis_output.c:586 */
/* ---- file I/O, ref:1019-1024 (src/system/lis_input.c, lis_input_mm.c, lis_output.c, lis_output_mm.c) ----
 * Matrix Market coordinate real general|symmetric, with Lis's size-line extension "nr nc nnz isb isx [isbin]"
 * (right-hand side / initial guess appended, optional binary records).  The reader fixes the in-row entry
 * order (file order, mirrored entry first) and with it the bits of every SpMV on the matrix.
 * `array` (dense) files are read straight into CSR (non-zero values, columns ascending), which is what the
 * reference's DNS -> CSR conversion yields.  Anything that is not Matrix Market is read as Harwell-Boeing
 * (real unsymmetric assembled only, right-hand sides skipped: lis_input_hb.c). */
LIS_INT lis_input(LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x, char *filename);      /* lis_input.c:67 */
LIS_INT lis_input_matrix(LIS_MATRIX A, char *filename);                          /* lis_input.c:174 */
LIS_INT lis_input_vector(LIS_VECTOR v, char *filename);                          /* lis_input.c:188 (MM, Lis ascii, plain) */
LIS_INT lis_output(LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x, LIS_INT mode, char *path); /* lis_output.c:63 (MM, MMB) */
LIS_INT lis_output_matrix(LIS_MATRIX A, LIS_INT mode, char *path);               /* lis_output.c:101 */
LIS_INT lis_output_vector(LIS_VECTOR v, LIS_INT format, char *filename);         /* lis_output.c:146 (PLAIN, MM, LIS) */

/* ---- utilities, ref:1030-1045 (src/system) ----------------------------------------------------- */
LIS_INT lis_initialize(int *argc, char **argv[]);
LIS_INT lis_finalize(void);
double  lis_wtime(void);
void    CHKERR(LIS_INT err);
void   *lis_malloc(size_t size, char *tag);
void   *lis_calloc(size_t size, char *tag);
void   *lis_realloc(void *p, size_t size);
void    lis_free(void *p);
void    lis_free2(LIS_INT n, ...);
LIS_INT lis_is_malloc(void *p);
LIS_INT lis_printf(LIS_Comm comm, const char *mess, ...);
void    lis_sort_id(LIS_INT is, LIS_INT ie, LIS_INT *i1, LIS_SCALAR *d1);  /* src/system/lis_sort.c:90; test/spmvtest3.c:194 */

/* the reference's drivers bracket main() with these (ref:286-292); trace output is a _DEBUG build feature */
#define LIS_DEBUG_FUNC_IN
#define LIS_DEBUG_FUNC_OUT

/* static row split every reference kernel and the rank partition use, ref:1067-1078 */
#define LIS_GET_ISIE(id,nprocs,n,is,ie) \
		if( (id) < (n)%(nprocs) ) \
		{ \
			(ie) = (n)/(nprocs)+1; \
			(is) = (ie)*(id); \
		} \
		else \
		{ \
			(ie) = (n)/(nprocs); \
			(is) = (ie)*(id) + (n)%(nprocs); \
		} \
		(ie) = (ie)+(is);

#ifdef __cplusplus
}
#endif
#endif
