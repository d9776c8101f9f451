/*
 * lis_solver.c -- lis_solve() and the three Krylov loops of the hot path, iterating entirely in HBM.
 *
 * Orchestration follows the reference (src/solver/lis_solver.c: lis_solve :367, lis_solve_kernel :441,
 * initial residual :957, defaults :242-284, option table :175-197); the recurrences are the reference's
 *   CG        src/solver/lis_solver_cg.c:129-235
 *   BiCGSTAB  src/solver/lis_solver_bicgstab.c:137-315
 *   GMRES(m)  src/solver/lis_solver_gmres.c:135-342
 * with every vector operation in the same order and with the same element-wise arithmetic, executed by
 * the HIP kernels; only the association order inside dot / nrm2 differs (fixed tree instead of the
 * reference's thread-count dependent left-to-right sum).  Work vectors are raw HBM buffers:
 * b and x cross PCIe once on entry, x once on exit.  Scalars (alpha, beta, rho, Givens rotations, the
 * Hessenberg matrix) stay on the host exactly as in the reference.
 */
#include <ctype.h>
#include <stdio.h>
#include "lis_internal.h"

static const char *solver_keys[] = {"cg", "bicg", "cgs", "bicgstab", "bicgstabl", "gpbicg", "tfqmr", "orthomin", "gmres",
	"jacobi", "gs", "sor", "bicgsafe", "cr", "bicr", "crs", "bicrstab", "gpbicr", "bicrsafe", "fgmres", "idrs", "idr1",
	"minres", "cocg", "cocr"};
static const char *solver_names[] = {"", "CG", "BiCG", "CGS", "BiCGSTAB", "BiCGSTAB(l)", "GPBiCG", "TFQMR", "Orthomin", "GMRES",
	"Jacobi", "Gauss-Seidel", "SOR", "BiCGSafe", "CR", "BiCR", "CRS", "BiCRSTAB", "GPBiCR", "BiCRSafe", "FGMRES", "IDR(s)",
	"IDR(1)", "MINRES", "COCG", "COCR"};
static const char *precon_keys[] = {"none", "jacobi", "ilu", "ssor", "hybrid", "is", "sainv", "saamg", "iluc", "ilut", "bjacobi"};
static const char *precon_names[] = {"none", "Jacobi", "ILU", "SSOR", "Hybrid", "I+S", "SAINV", "SAAMG", "Crout ILU", "ILUT", "Block Jacobi"};
static const char *storage_keys[] = {"csr", "csc", "msr", "dia", "ell", "jad", "bsr", "bsc", "vbr", "coo", "dns"};
static const char *storage_names[] = {"CSR", "CSC", "MSR", "DIA", "ELL", "JAD", "BSR", "BSC", "VBR", "COO", "DNS"};
static const char *print_keys[] = {"none", "mem", "out", "all"};
static const char *scale_keys[] = {"none", "jacobi", "symm_diag"};
static const char *bool_keys[] = {"false", "true"};
static const char *precision_keys[] = {"double", "quad", "switch"};
static const char *conv_keys[] = {"nrm2_r", "nrm2_b", "nrm1_b"};
static const char *retcode_names[] = {"LIS_SUCCESS", "LIS_ILL_OPTION", "LIS_BREAKDOWN", "LIS_OUT_OF_MEMORY", "LIS_MAXITER", "LIS_NOT_IMPLEMENTED", "LIS_ERR_FILE_IO"};
#define COUNT(a) ((int)(sizeof(a) / sizeof((a)[0])))

/* ------------------------------------------------------------------ solver object */
static void solver_defaults(LIS_SOLVER s)
{	/* ref lis_solver.c:221-290 */
	memset(s, 0, sizeof(*s));
	LIS_INT *o = s->options; LIS_SCALAR *p = s->params;
	o[LIS_OPTIONS_SOLVER] = LIS_SOLVER_BICG;   o[LIS_OPTIONS_PRECON] = LIS_PRECON_TYPE_NONE;
	o[LIS_OPTIONS_OUTPUT] = LIS_FALSE;         o[LIS_OPTIONS_MAXITER] = 1000;
	o[LIS_OPTIONS_RESTART] = 40;               o[LIS_OPTIONS_ELL] = 2;
	o[LIS_OPTIONS_SCALE] = LIS_SCALE_NONE;     o[LIS_OPTIONS_FILL] = 0;
	o[LIS_OPTIONS_M] = 3;                      o[LIS_OPTIONS_PSOLVER] = LIS_SOLVER_SOR;
	o[LIS_OPTIONS_PMAXITER] = 25;              o[LIS_OPTIONS_PRESTART] = 40;
	o[LIS_OPTIONS_PELL] = 2;                   o[LIS_OPTIONS_PPRECON] = LIS_PRECON_TYPE_NONE;
	o[LIS_OPTIONS_ISLEVEL] = 1;                o[LIS_OPTIONS_INITGUESS_ZEROS] = LIS_TRUE;
	o[LIS_OPTIONS_ADDS] = LIS_FALSE;           o[LIS_OPTIONS_ADDS_ITER] = 1;
	o[LIS_OPTIONS_PRECISION] = LIS_PRECISION_DOUBLE; o[LIS_OPTIONS_USE_AT] = LIS_FALSE;
	o[LIS_OPTIONS_SWITCH_MAXITER] = -1;        o[LIS_OPTIONS_SAAMG_UNSYM] = LIS_FALSE;
	o[LIS_OPTIONS_STORAGE] = 0;                o[LIS_OPTIONS_STORAGE_BLOCK] = 2;
	o[LIS_OPTIONS_CONV_COND] = 0;              o[LIS_OPTIONS_INIT_SHADOW_RESID] = LIS_RESID;
	o[LIS_OPTIONS_IDRS_RESTART] = 2;
	p[LIS_PARAMS_RESID - LIS_OPTIONS_LEN] = 1.0e-12;       p[LIS_PARAMS_RESID_WEIGHT - LIS_OPTIONS_LEN] = 1.0;
	p[LIS_PARAMS_OMEGA - LIS_OPTIONS_LEN] = 1.9;           p[LIS_PARAMS_SSOR_OMEGA - LIS_OPTIONS_LEN] = 1.0;
	p[LIS_PARAMS_RELAX - LIS_OPTIONS_LEN] = 1.0;           p[LIS_PARAMS_DROP - LIS_OPTIONS_LEN] = 0.05;
	p[LIS_PARAMS_ALPHA - LIS_OPTIONS_LEN] = 1.0;           p[LIS_PARAMS_TAU - LIS_OPTIONS_LEN] = 0.05;
	p[LIS_PARAMS_SIGMA - LIS_OPTIONS_LEN] = 2.0;           p[LIS_PARAMS_GAMMA - LIS_OPTIONS_LEN] = 1.0;
	p[LIS_PARAMS_PRESID - LIS_OPTIONS_LEN] = 1.0e-3;       p[LIS_PARAMS_POMEGA - LIS_OPTIONS_LEN] = 1.5;
	p[LIS_PARAMS_SWITCH_RESID - LIS_OPTIONS_LEN] = 1.0e-12; p[LIS_PARAMS_RATE - LIS_OPTIONS_LEN] = 5.0;
	p[LIS_PARAMS_SAAMG_THETA - LIS_OPTIONS_LEN] = 0.05;
	s->precision = LIS_PRECISION_DOUBLE;
}

LIS_INT lis_solver_create(LIS_SOLVER *solver)
{
	*solver = (LIS_SOLVER)malloc(sizeof(struct LIS_SOLVER_STRUCT));
	if (!*solver) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)sizeof(struct LIS_SOLVER_STRUCT));
	solver_defaults(*solver);
	lisi_register(*solver, LISI_KIND_SOLVER);
	return LIS_SUCCESS;
}

LIS_INT lis_solver_destroy(LIS_SOLVER solver)
{
	if (solver && lisi_is_registered(solver)) {
		if (solver->d) lis_vector_destroy(solver->d);
		free(solver->rhistory);
		lisi_unregister(solver);
		free(solver);
	}
	return LIS_SUCCESS;
}

LIS_INT lis_solver_set_matrix(LIS_MATRIX A, LIS_SOLVER solver) { solver->A = A; return LIS_SUCCESS; }
LIS_INT lis_solver_get_iter(LIS_SOLVER s, LIS_INT *iter) { *iter = s->iter; return LIS_SUCCESS; }
LIS_INT lis_solver_get_iterex(LIS_SOLVER s, LIS_INT *iter, LIS_INT *d, LIS_INT *q)
{ *iter = s->iter; *d = s->iter2; *q = s->iter - s->iter2; return LIS_SUCCESS; }
LIS_INT lis_solver_get_time(LIS_SOLVER s, double *t) { *t = s->time; return LIS_SUCCESS; }
LIS_INT lis_solver_get_timeex(LIS_SOLVER s, double *time, double *itime, double *ptime, double *pc, double *pi)
{
	*time = s->time;
	if (itime) *itime = s->itime;
	if (ptime) *ptime = s->ptime;
	if (pc) *pc = s->p_c_time;
	if (pi) *pi = s->p_i_time;
	return LIS_SUCCESS;
}
LIS_INT lis_solver_get_residualnorm(LIS_SOLVER s, LIS_REAL *r) { *r = s->resid; return LIS_SUCCESS; }
LIS_INT lis_solver_get_solver(LIS_SOLVER s, LIS_INT *n) { *n = s->options[LIS_OPTIONS_SOLVER]; return LIS_SUCCESS; }
LIS_INT lis_solver_get_precon(LIS_SOLVER s, LIS_INT *p) { *p = s->options[LIS_OPTIONS_PRECON]; return LIS_SUCCESS; }
LIS_INT lis_solver_get_status(LIS_SOLVER s, LIS_INT *st) { *st = s->retcode; return LIS_SUCCESS; }

LIS_INT lis_solver_get_rhistory(LIS_SOLVER s, LIS_VECTOR v)
{	/* ref lis_solver.c:1715-1740 */
	LIS_INT count = s->iter + 1;
	if (s->retcode != LIS_SUCCESS) count--;
	if (count > v->n) count = v->n;
	LISCHK(lisd_vec_host_write(v, 1));
	for (LIS_INT i = 0; i < count; i++) v->value[i] = s->rhistory[i];
	lis_amd_vector_host_modified(v);
	return LIS_SUCCESS;
}

LIS_INT lis_solver_output_rhistory(LIS_SOLVER s, char *filename)
{	/* ref src/system/lis_output.c:586-640: one %e per line, entries 0..iter */
	if (!s->rhistory) return LISI_ERR(LIS_ERR_ILL_ARG, "residual history is empty\n");
	if (lisg.rank != 0) return LIS_SUCCESS;
	FILE *f = fopen(filename, "w");
	if (!f) return LISI_ERR(LIS_ERR_FILE_IO, "cannot open file %s\n", filename);
	LIS_INT count = s->iter + 1;
	if (s->retcode != LIS_SUCCESS) count--;
	for (LIS_INT i = 0; i < count; i++) fprintf(f, "%e\n", s->rhistory[i]);
	fclose(f);
	return LIS_SUCCESS;
}

LIS_INT lis_solver_get_solvername(LIS_INT solver, char *name)
{
	if (solver < 1 || solver > LIS_SOLVER_LEN) return LIS_FAILS;
	strcpy(name, solver_names[solver]);
	return LIS_SUCCESS;
}
LIS_INT lis_solver_get_preconname(LIS_INT precon_type, char *name)
{
	if (precon_type < 0 || precon_type > LIS_PRECON_TYPE_LEN - 2) return LIS_FAILS;
	strcpy(name, precon_names[precon_type]);
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ options */
static LIS_INT pick(const char *val, const char **keys, int nkeys, int base, LIS_INT *slot, const char *what)
{
	if (val[0] >= '0' && val[0] <= '9') { *slot = atoi(val); return LIS_SUCCESS; }
	for (int i = 0; i < nkeys; i++) if (strcmp(val, keys[i]) == 0) { *slot = i + base; return LIS_SUCCESS; }
	return LISI_ERR(LIS_ERR_ILL_ARG, "Parameter %s is not correct\n", what);
}

static LIS_INT set_one(const char *name, const char *val, LIS_SOLVER s)
{
	static const struct { const char *name; int slot; } table[] = {   /* ref lis_solver.c:175-197 */
		{"-maxiter", LIS_OPTIONS_MAXITER}, {"-tol", LIS_PARAMS_RESID}, {"-print", LIS_OPTIONS_OUTPUT}, {"-scale", LIS_OPTIONS_SCALE},
		{"-ssor_omega", LIS_PARAMS_SSOR_OMEGA}, {"-ilu_fill", LIS_OPTIONS_FILL}, {"-ilu_relax", LIS_PARAMS_RELAX},
		{"-is_alpha", LIS_PARAMS_ALPHA}, {"-is_level", LIS_OPTIONS_ISLEVEL}, {"-is_m", LIS_OPTIONS_M},
		{"-hybrid_maxiter", LIS_OPTIONS_PMAXITER}, {"-hybrid_ell", LIS_OPTIONS_PELL}, {"-hybrid_restart", LIS_OPTIONS_PRESTART},
		{"-hybrid_tol", LIS_PARAMS_PRESID}, {"-hybrid_omega", LIS_PARAMS_POMEGA}, {"-hybrid_i", LIS_OPTIONS_PSOLVER},
		{"-sainv_drop", LIS_PARAMS_DROP}, {"-ric2s_tau", LIS_PARAMS_TAU}, {"-ric2s_sigma", LIS_PARAMS_SIGMA}, {"-ric2s_gamma", LIS_PARAMS_GAMMA},
		{"-restart", LIS_OPTIONS_RESTART}, {"-ell", LIS_OPTIONS_ELL}, {"-omega", LIS_PARAMS_OMEGA}, {"-i", LIS_OPTIONS_SOLVER},
		{"-p", LIS_OPTIONS_PRECON}, {"-hybrid_p", LIS_OPTIONS_PPRECON}, {"-initx_zeros", LIS_OPTIONS_INITGUESS_ZEROS},
		{"-adds", LIS_OPTIONS_ADDS}, {"-adds_iter", LIS_OPTIONS_ADDS_ITER}, {"-f", LIS_OPTIONS_PRECISION}, {"-use_at", LIS_OPTIONS_USE_AT},
		{"-switch_tol", LIS_PARAMS_SWITCH_RESID}, {"-switch_maxiter", LIS_OPTIONS_SWITCH_MAXITER}, {"-saamg_unsym", LIS_OPTIONS_SAAMG_UNSYM},
		{"-iluc_drop", LIS_PARAMS_DROP}, {"-iluc_gamma", LIS_PARAMS_GAMMA}, {"-iluc_rate", LIS_PARAMS_RATE},
		{"-storage", LIS_OPTIONS_STORAGE}, {"-storage_block", LIS_OPTIONS_STORAGE_BLOCK}, {"-conv_cond", LIS_OPTIONS_CONV_COND},
		{"-tol_w", LIS_PARAMS_RESID_WEIGHT}, {"-saamg_theta", LIS_PARAMS_SAAMG_THETA}, {"-irestart", LIS_OPTIONS_IDRS_RESTART},
	};
	for (int i = 0; i < COUNT(table); i++) {
		if (strcmp(name, table[i].name) != 0) continue;
		const int slot = table[i].slot;
		LIS_INT *o = s->options;
		switch (slot) {
		case LIS_OPTIONS_SOLVER:  return pick(val, solver_keys, COUNT(solver_keys), 1, &o[slot], "LIS_OPTIONS_SOLVER");
		case LIS_OPTIONS_PSOLVER: return pick(val, solver_keys, COUNT(solver_keys), 1, &o[slot], "LIS_OPTIONS_PSOLVER");
		case LIS_OPTIONS_PRECON:  return pick(val, precon_keys, COUNT(precon_keys), 0, &o[slot], "LIS_OPTIONS_PRECON");
		case LIS_OPTIONS_PPRECON: return pick(val, precon_keys, COUNT(precon_keys), 0, &o[slot], "LIS_OPTIONS_PPRECON");
		case LIS_OPTIONS_SCALE:   return pick(val, scale_keys, COUNT(scale_keys), 0, &o[slot], "LIS_OPTIONS_SCALE");
		case LIS_OPTIONS_OUTPUT:  return pick(val, print_keys, COUNT(print_keys), 0, &o[slot], "LIS_OPTIONS_OUTPUT");
		case LIS_OPTIONS_STORAGE: return pick(val, storage_keys, COUNT(storage_keys), 1, &o[slot], "LIS_OPTIONS_STORAGE");
		case LIS_OPTIONS_CONV_COND: return pick(val, conv_keys, COUNT(conv_keys), 0, &o[slot], "LIS_OPTIONS_CONV_COND");
		case LIS_OPTIONS_PRECISION: return pick(val, precision_keys, COUNT(precision_keys), 0, &o[slot], "LIS_OPTIONS_PRECISION");
		case LIS_OPTIONS_INITGUESS_ZEROS: case LIS_OPTIONS_ADDS: case LIS_OPTIONS_USE_AT: case LIS_OPTIONS_SAAMG_UNSYM:
			return pick(val, bool_keys, COUNT(bool_keys), 0, &o[slot], "true/false");
		default:
			if (slot < LIS_OPTIONS_LEN) o[slot] = atoi(val);
			else s->params[slot - LIS_OPTIONS_LEN] = strtod(val, NULL);
			return LIS_SUCCESS;
		}
	}
	return LIS_SUCCESS;      /* unknown options are ignored, as in the reference */
}

static LIS_INT set_from_tokens(char **tok, int n, LIS_SOLVER s)
{
	for (int i = 0; i + 1 < n; i++) {
		if (tok[i][0] != '-' || (tok[i][1] >= '0' && tok[i][1] <= '9')) continue;
		char name[64], val[256];
		size_t k;
		for (k = 0; tok[i][k] && k + 1 < sizeof(name); k++) name[k] = (char)tolower((unsigned char)tok[i][k]);
		name[k] = 0;
		for (k = 0; tok[i + 1][k] && k + 1 < sizeof(val); k++) val[k] = (char)tolower((unsigned char)tok[i + 1][k]);
		val[k] = 0;
		LIS_INT err = set_one(name, val, s);
		if (err) { s->retcode = err; return err; }
	}
	return LIS_SUCCESS;
}

LIS_INT lis_solver_set_option(char *text, LIS_SOLVER solver)
{
	char **tok;
	int n = lisi_tokenize(text, &tok);
	LIS_INT err = set_from_tokens(tok, n, solver);
	lisi_tokens_free(tok, n);
	return err;
}

LIS_INT lis_solver_set_optionC(LIS_SOLVER solver)
{	/* options given on the command line captured by lis_initialize (ref lis_solver.c:1095) */
	return set_from_tokens(lisi_cmd_argv, lisi_cmd_argc, solver);
}

/* ------------------------------------------------------------------ preconditioner (none, Jacobi) */
LIS_INT lis_precon_create(LIS_SOLVER solver, LIS_PRECON *precon)
{
	const LIS_INT type = solver->options[LIS_OPTIONS_PRECON];
	*precon = NULL;
	if (type != LIS_PRECON_TYPE_NONE && type != LIS_PRECON_TYPE_JACOBI)
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "preconditioner %D is not served by liblis_amd (none, jacobi)\n", type);
	LIS_PRECON p = (LIS_PRECON)calloc(1, sizeof(struct LIS_PRECON_STRUCT));
	if (!p) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)sizeof(struct LIS_PRECON_STRUCT));
	p->precon_type = type;
	lisi_register(p, LISI_KIND_PRECON);
	if (type == LIS_PRECON_TYPE_JACOBI) {      /* D = 1 / diag(A): ref lis_precon_jacobi.c:61-85 */
		LIS_INT err = lis_vector_duplicate(solver->A, &p->D);
		if (!err) err = lis_matrix_get_diagonal(solver->A, p->D);
		if (!err) err = lis_vector_reciprocal(p->D);
		if (err) { lis_precon_destroy(p); return err; }
	}
	*precon = p;
	return LIS_SUCCESS;
}

LIS_INT lis_precon_destroy(LIS_PRECON precon)
{
	if (precon && lisi_is_registered(precon)) {
		if (precon->D) lis_vector_destroy(precon->D);
		lisi_unregister(precon);
		free(precon);
	}
	return LIS_SUCCESS;
}

#include "lis_krylov.h"

/* residual norm from a sum of squares the fused kernels already produced (lis_solver.c:1792); the
 * 1-norm criterion (:1804) has no fused form and takes its own pass */
static LIS_INT resid_from_sumsq(ctx_t *c, const double *r, double sumsq, double *nrm)
{
	if (c->s->options[LIS_OPTIONS_CONV_COND] == LIS_CONV_COND_NRM1_B) return lisd_nrm1(c->n, r, nrm);
	*nrm = sqrt(sumsq) * c->bnrm;
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ device-driven loops (include/liship.h)
 * The loops above wait for the host after every reduction (2 round trips per CG iteration, 4 per BiCGSTAB
 * iteration): short next to a 512^3 product, a tenth of a 256^3 iteration, most of a small one, and in a
 * multi-rank job each of them also stalls the neighbours.  Here the scalars stay in HBM: the host enqueues
 * LISD_BATCH whole iterations, every kernel of which is a no-op once the convergence flag is up, then reads the
 * state block back once.  Same kernels, same per-element expressions, same IEEE scalar operations in the same
 * order -- iteration counts, x and the residual history are bit-identical to the host-scalar loops
 * (tests/test_device_loops_gpu.py runs both), which stay selectable with LIS_AMD_HOST_SCALARS=1.
 * Not used for -conv_cond nrm1_b (needs a second reduction of another kind) or with the callback communicator
 * (its fold is a host function). */
#define LISD_BATCH 16
static int device_scalars_ok(const ctx_t *c)
{
	if (lisg.host_scalars) return 0;
	if (c->s->options[LIS_OPTIONS_CONV_COND] == LIS_CONV_COND_NRM1_B) return 0;
	if (lisg.nprocs > 1 && lisg.comm_kind != 1) return 0;
	return 1;
}

typedef struct {
	double *st, *rh;           /* HBM: state block, residual history (NULL unless asked for) */
	size_t bytes;              /* of the pooled allocation holding both */
	double host[LISHIP_KS_LEN];
	LIS_INT printed;           /* history entries already passed to note() */
} dev_loop;

static LIS_INT dev_loop_begin(ctx_t *c, dev_loop *L, const double *init)
{
	const size_t hist = c->output ? (size_t)c->maxiter + 2 : 0;
	L->bytes = sizeof(double) * (LISHIP_KS_LEN + hist);
	L->printed = 0;
	LISCHK(lisd_pool_get(L->bytes, (void **)&L->st));
	L->rh = hist ? L->st + LISHIP_KS_LEN : NULL;
	HIPCHK(liship_memset(L->st, 0, L->bytes, lisg.stream));
	memcpy(L->host, init, sizeof(L->host));
	HIPCHK(liship_memcpy_h2d(L->st, L->host, sizeof(L->host), lisg.stream));
	HIPCHK(liship_stream_synchronize(lisg.stream));
	return LIS_SUCCESS;
}

/* The scalar step that consumes a reduction's sums st[slot..slot+count).  Single-rank job: announced before the
 * reduction is launched, it runs inside the reduction's last kernel (dev_step then only catches the case where it
 * found none to ride in).  RCCL job: the sums are gathered from all ranks first, then the step runs on its own. */
static LIS_INT dev_announce(dev_loop *L, int step)
{
	if (lisg.comm_kind != 1) HIPCHK(liship_krylov_chain(step, L->st, L->rh));
	return LIS_SUCCESS;
}
static LIS_INT dev_step(dev_loop *L, int step, int slot, int count)
{
	if (lisg.comm_kind != 1) { HIPCHK(liship_krylov_chain_flush(lisg.stream)); return LIS_SUCCESS; }
	LISCHK(lisc_gather_device(L->st + slot, count));
	HIPCHK(liship_krylov_step(step, L->st, L->rh, lisg.gather_out, lisg.nprocs, lisg.stream));
	return LIS_SUCCESS;
}

/* one read-back per batch: the state block, and the new history entries when the caller wants them */
static LIS_INT dev_loop_sync(ctx_t *c, dev_loop *L)
{
	HIPCHK(liship_memcpy_d2h(lisg.host_out, L->st, sizeof(L->host), lisg.stream));
	HIPCHK(liship_stream_synchronize(lisg.stream));
	memcpy(L->host, lisg.host_out, sizeof(L->host));
	if (L->rh) {
		LIS_INT upto = (LIS_INT)L->host[LISHIP_KS_NHIST];
		if (upto > c->maxiter + 1) upto = c->maxiter + 1;
		if (upto > L->printed) {
			const size_t cnt = (size_t)(upto - L->printed);
			double *tmp = (double *)malloc(sizeof(double) * cnt);
			if (!tmp) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)cnt);
			int rc = liship_memcpy_d2h(tmp, L->rh + L->printed + 1, sizeof(double) * cnt, lisg.stream);
			if (!rc) rc = liship_stream_synchronize(lisg.stream);
			if (rc) { free(tmp); HIPCHK(rc); }
			for (size_t i = 0; i < cnt; i++) note(c, L->printed + 1 + (LIS_INT)i, tmp[i]);
			free(tmp);
			L->printed = upto;
		}
	}
	return LIS_SUCCESS;
}

/* The batches of a device-driven loop.  `enqueue` puts `batch` whole iterations on the stream (the first one is
 * iteration queued + 1).  With LIS_AMD_GRAPHS=1 a single-rank solve replays a hipGraph of one batch from the second full
 * batch on (every argument of every kernel is the same from batch to batch -- the scalars live in the state block -- and
 * a batch behind a raised convergence flag is as harmless replayed as enqueued).  Opt-in because it does not pay here:
 * 32^3 and 64^3 run the same 44 000 CG iterations/s (5 dependent kernels in 22.5 us), so the floor is the GPU's
 * dispatch of a dependent kernel, not the host's launch rate, and the replay is 0-5 % SLOWER from 32^3 to 320^3
 * (profiles/r02_graph_sweep.txt). */
typedef LIS_INT (*dev_batch_fn)(ctx_t *c, dev_loop *L, void *vars, LIS_INT queued, LIS_INT batch);
static LIS_INT dev_loop_run(ctx_t *c, dev_loop *L, dev_batch_fn enqueue, void *vars)
{
	LIS_INT err = 0;
	void *gexec = NULL;
	int graphs = lisg.graphs && lisg.nprocs == 1;
	for (LIS_INT queued = 0; queued < c->maxiter && L->host[LISHIP_KS_DONE] == 0.0; ) {
		const LIS_INT batch = (c->maxiter - queued < LISD_BATCH) ? c->maxiter - queued : LISD_BATCH;
		int replayed = 0;
		if (graphs && queued > 0 && batch == LISD_BATCH) {
			if (!gexec) {
				lisd_capture_mark(1);
				int rc = liship_graph_capture_begin(lisg.stream);
				if (rc == 0) {
					const LIS_INT e = enqueue(c, L, vars, queued, batch);
					rc = liship_graph_capture_end(lisg.stream, e ? NULL : &gexec);
					if (e || rc) { (void)liship_krylov_chain(0, NULL, NULL); gexec = NULL; }
				}
				lisd_capture_mark(0);
				if (!gexec) graphs = 0;                    /* not capturable here: plain launches from now on */
			}
			if (gexec) { KTRY(liship_graph_launch(gexec, lisg.stream)); replayed = 1; lisg.last_graph_replays++; }
		}
		if (!replayed) TRY(enqueue(c, L, vars, queued, batch));
		queued += batch;
		TRY(dev_loop_sync(c, L));
	}
done:
	(void)liship_graph_destroy(gexec);
	return err;
}

static LIS_INT dev_loop_finish(ctx_t *c, dev_loop *L, LIS_INT err)
{
	LIS_SOLVER s = c->s;
	(void)liship_krylov_guard(NULL);
	(void)liship_krylov_chain(0, NULL, NULL);      /* an error may have left a step announced */
	if (!err) {
		s->resid = L->host[LISHIP_KS_NRM2];
		if (L->host[LISHIP_KS_STATUS] == 1.0) { s->retcode = LIS_SUCCESS; s->iter = (LIS_INT)L->host[LISHIP_KS_ITER]; }
		else if (L->host[LISHIP_KS_STATUS] == 2.0) { s->retcode = LIS_BREAKDOWN; s->iter = (LIS_INT)L->host[LISHIP_KS_ITER]; err = LIS_BREAKDOWN; }
		else { s->retcode = LIS_MAXITER; s->iter = c->maxiter + 1; err = LIS_MAXITER; }
	}
	if (L->st) { (void)liship_stream_synchronize(lisg.stream); lisd_pool_put(L->st, L->bytes); L->st = NULL; }
	return err;
}

typedef struct { double *q, *r, *p; } cg_vars;
static LIS_INT cg_batch(ctx_t *c, dev_loop *L, void *vars, LIS_INT queued, LIS_INT batch)
{
	LIS_INT err = 0;
	const int n = c->n;
	const cg_vars *v = (const cg_vars *)vars;
	double *q = v->q, *r = v->r, *p = v->p, *st = L->st;
	for (LIS_INT k = 0; k < batch; k++) {
		/* x += alpha p of the PREVIOUS iteration rides in this pass, which reads p anyway (none before the first) */
		if (c->duniform) KTRY(liship_cg_direction_uniform_dev_f64(n, queued + k ? st + LISHIP_KS_ALPHA : NULL, st + LISHIP_KS_BETA, r, c->dconst, p, c->x, lisg.stream));
		else KTRY(liship_cg_direction_dev_f64(n, queued + k ? st + LISHIP_KS_ALPHA : NULL, st + LISHIP_KS_BETA, r, c->dinv, p, c->x, lisg.stream));
		TRY(dev_announce(L, LISHIP_STEP_CG_ALPHA));
		TRY(lisd_spmv_dot_launch_to(c->A, p, q, p, 0, st + LISHIP_KS_DOT0));
		TRY(dev_step(L, LISHIP_STEP_CG_ALPHA, LISHIP_KS_DOT0, 1));
		TRY(dev_announce(L, c->dinv ? LISHIP_STEP_CG_RESID_PRE : LISHIP_STEP_CG_RESID));
		if (c->duniform) KTRY(liship_cg_residual_jacobi_uniform_dev_f64(n, st + LISHIP_KS_NALPHA, q, c->dconst, r, st + LISHIP_KS_SUM0, lisg.reduce_work, lisg.stream));
		else if (c->dinv) KTRY(liship_cg_residual_jacobi_dev_f64(n, st + LISHIP_KS_NALPHA, q, c->dinv, r, st + LISHIP_KS_SUM0, lisg.reduce_work, lisg.stream));
		else         KTRY(liship_axpy_sumsq_dev_f64(n, st + LISHIP_KS_NALPHA, q, r, st + LISHIP_KS_SUM0, lisg.reduce_work, lisg.stream));
		TRY(dev_step(L, c->dinv ? LISHIP_STEP_CG_RESID_PRE : LISHIP_STEP_CG_RESID, LISHIP_KS_SUM0, c->dinv ? 2 : 1));
	}
done:
	return err;
}

static LIS_INT run_cg_device(ctx_t *c)
{
	LIS_INT err = 0;
	const int n = c->n;
	dev_loop L = {0};
	TRY(work_alloc(c, 3));
	double *q = c->work[0], *r = c->work[1], *p = c->work[2];
	double rho, init[LISHIP_KS_LEN] = {0};
	int st0 = initial_residual(c, r);
	if (st0) { work_free(c); return st0 < 0 ? -st0 : 0; }
	if (c->dinv) { KTRY(liship_pmul_f64(n, r, c->dinv, q, lisg.stream)); TRY(lisd_dot(n, r, q, &rho)); }
	else TRY(lisd_dot(n, r, r, &rho));
	init[LISHIP_KS_RHO] = rho; init[LISHIP_KS_RHO_OLD] = 1.0; init[LISHIP_KS_BETA] = rho / 1.0;
	init[LISHIP_KS_BNRM] = c->bnrm; init[LISHIP_KS_TOL] = c->tol; init[LISHIP_KS_NOT_HALF] = 1.0;
	TRY(dev_loop_begin(c, &L, init));
	double *st = L.st;
	KTRY(liship_krylov_guard(st + LISHIP_KS_DONE));
	cg_vars vars = { q, r, p };
	TRY(dev_loop_run(c, &L, cg_batch, &vars));
	/* the x update of the last iteration that formed an alpha: owed unless the loop ended on <p,q> = 0 (the reference
	 * leaves before its axpy then, lis_solver_cg.c:193-197) -- the kernels queued behind a raised flag touched neither p nor alpha */
	if (c->maxiter > 0 && L.host[LISHIP_KS_STATUS] != 2.0) {
		KTRY(liship_krylov_guard(NULL));
		KTRY(liship_axpy_dev_f64(n, st + LISHIP_KS_ALPHA, p, c->x, lisg.stream));
	}
done:
	err = dev_loop_finish(c, &L, err);
	work_free(c);
	return err;
}

typedef struct { double *rtld, *r, *t, *p, *v, *phat, *shat; int pre; } bicgstab_vars;
static LIS_INT bicgstab_batch(ctx_t *c, dev_loop *L, void *vars, LIS_INT queued, LIS_INT batch)
{
	LIS_INT err = 0;
	const int n = c->n;
	const bicgstab_vars *w = (const bicgstab_vars *)vars;
	double *rtld = w->rtld, *r = w->r, *t = w->t, *p = w->p, *v = w->v, *phat = w->phat, *shat = w->shat, *st = L->st;
	double *sv = r;                                    /* s aliases r: lis_solver_bicgstab.c:160-161 */
	const int pre = w->pre;
	for (LIS_INT k = 0; k < batch; k++) {
		KTRY(liship_krylov_guard(st + LISHIP_KS_DONE));
		if (queued + k == 0) TRY(d_copy(c, r, p));
		else KTRY(liship_axpy_xpay_dev_f64(n, st + LISHIP_KS_NOMEGA, v, r, st + LISHIP_KS_BETA, p, lisg.stream));
		if (pre) KTRY(liship_pmul_f64(n, p, c->dinv, phat, lisg.stream));
		TRY(dev_announce(L, LISHIP_STEP_BICGSTAB_ALPHA));
		TRY(lisd_spmv_dot_launch_to(c->A, phat, v, rtld, 0, st + LISHIP_KS_DOT0));
		TRY(dev_step(L, LISHIP_STEP_BICGSTAB_ALPHA, LISHIP_KS_DOT0, 1));
		TRY(dev_announce(L, LISHIP_STEP_BICGSTAB_HALF));
		KTRY(liship_axpy_sumsq_dev_f64(n, st + LISHIP_KS_NALPHA, v, r, st + LISHIP_KS_SUM0, lisg.reduce_work, lisg.stream));
		TRY(dev_step(L, LISHIP_STEP_BICGSTAB_HALF, LISHIP_KS_SUM0, 1));
		/* converged at the half step: x += alpha*phat (:240-258), and only then -- NOT_HALF is 0 from that
		 * step until the next one */
		KTRY(liship_krylov_guard(st + LISHIP_KS_NOT_HALF));
		KTRY(liship_axpy_dev_f64(n, st + LISHIP_KS_ALPHA, phat, c->x, lisg.stream));
		KTRY(liship_krylov_guard(st + LISHIP_KS_DONE));
		if (pre) KTRY(liship_pmul_f64(n, sv, c->dinv, shat, lisg.stream));
		TRY(dev_announce(L, LISHIP_STEP_BICGSTAB_OMEGA));
		TRY(lisd_spmv_dot_launch_to(c->A, shat, t, sv, 1, st + LISHIP_KS_DOT0));
		TRY(dev_step(L, LISHIP_STEP_BICGSTAB_OMEGA, LISHIP_KS_DOT0, 2));
		if (pre) KTRY(liship_axpy2_dev_f64(n, st + LISHIP_KS_ALPHA, phat, st + LISHIP_KS_OMEGA, shat, c->x, lisg.stream));
		TRY(dev_announce(L, LISHIP_STEP_BICGSTAB_RESID));
		if (pre) KTRY(liship_axpy_sumsq_dot_dev_f64(n, st + LISHIP_KS_NOMEGA, t, r, rtld, st + LISHIP_KS_SUM0, lisg.reduce_work, lisg.stream));
		else     /* shat is s itself: the iterate and the residual in one pass (s read once) */
			KTRY(liship_bicgstab_end_dev_f64(n, st + LISHIP_KS_ALPHA, st + LISHIP_KS_OMEGA, st + LISHIP_KS_NOMEGA, phat, t, rtld, c->x, r,
			                                 st + LISHIP_KS_SUM0, lisg.reduce_work, lisg.stream));
		TRY(dev_step(L, LISHIP_STEP_BICGSTAB_RESID, LISHIP_KS_SUM0, 2));
	}
done:
	return err;
}

static LIS_INT run_bicgstab_device(ctx_t *c)
{
	LIS_INT err = 0;
	const int n = c->n;
	const int pre = c->dinv != NULL;
	dev_loop L = {0};
	TRY(work_alloc(c, pre ? 7 : 5));
	double *rtld = c->work[0], *r = c->work[1], *t = c->work[2], *p = c->work[3], *v = c->work[4];
	double *phat = pre ? c->work[5] : p, *shat = pre ? c->work[6] : r;
	double rho, init[LISHIP_KS_LEN] = {0};
	int st0 = initial_residual(c, r);
	if (st0) { work_free(c); return st0 < 0 ? -st0 : 0; }
	TRY(d_copy(c, r, rtld));
	TRY(lisd_dot(n, rtld, r, &rho));
	init[LISHIP_KS_RHO] = rho; init[LISHIP_KS_RHO_OLD] = 1.0;
	init[LISHIP_KS_ALPHA] = 1.0; init[LISHIP_KS_NALPHA] = -1.0; init[LISHIP_KS_OMEGA] = 1.0; init[LISHIP_KS_NOMEGA] = -1.0;
	init[LISHIP_KS_BNRM] = c->bnrm; init[LISHIP_KS_TOL] = c->tol; init[LISHIP_KS_NOT_HALF] = 1.0;
	TRY(dev_loop_begin(c, &L, init));
	bicgstab_vars vars = { rtld, r, t, p, v, phat, shat, pre };
	TRY(dev_loop_run(c, &L, bicgstab_batch, &vars));
done:
	err = dev_loop_finish(c, &L, err);
	work_free(c);
	return err;
}

/* Preconditioned CG, lis_solver_cg.c:176-215, the reference's operation order per element with the
 * passes over HBM fused:
 *   p = M^-1 r + beta p              one pass   (the solve is a copy or r.*dinv, folded into the xpay)
 *   q = A p ; <p,q>                  the product, the dot in its epilogue
 *   x += alpha p ; r -= alpha q ; ||r|| ; rho' = <r, M^-1 r>     one pass
 * rho' is the next iteration's rho (the reference computes it at :180 from the same r).
 * LIS_AMD_NO_FUSION=1 runs the one-kernel-per-reference-call loop instead. */
static LIS_INT run_cg_unfused(ctx_t *c);
static LIS_INT run_cg_device(ctx_t *c);
static LIS_INT run_cg(ctx_t *c)
{
	if (lisg.no_fusion) return run_cg_unfused(c);
	if (device_scalars_ok(c)) return run_cg_device(c);
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter;
	const int n = c->n;
	TRY(work_alloc(c, 3));
	double *q = c->work[0], *r = c->work[1], *p = c->work[2];
	double alpha, beta, rho, rho_old = 1.0, dot_pq, nrm2 = 0.0, sums[2];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	/* rho of the first iteration; z lives in q for this one pass */
	if (c->dinv) { KTRY(liship_pmul_f64(n, r, c->dinv, q, lisg.stream)); TRY(lisd_dot(n, r, q, &rho)); }
	else TRY(lisd_dot(n, r, r, &rho));
	for (iter = 1; iter <= c->maxiter; iter++) {
		beta = rho / rho_old;
		if (c->dinv) KTRY(liship_pmul_xpay_f64(n, r, c->dinv, beta, p, lisg.stream));
		else         KTRY(liship_xpay_f64(n, r, beta, p, lisg.stream));
		TRY(lisd_spmv_dot_launch(c->A, p, q, p, 0));
		TRY(lisd_fetch(1, &dot_pq));
		if (dot_pq == 0.0) { s->retcode = LIS_BREAKDOWN; s->iter = iter; s->resid = nrm2; err = LIS_BREAKDOWN; goto done; }
		alpha = rho / dot_pq;
		if (c->dinv) {
			KTRY(liship_cg_update_jacobi_f64(n, alpha, p, q, c->dinv, c->x, r, lisg.reduce_out, lisg.reduce_work, lisg.stream));
			TRY(lisd_fetch(2, sums));
		} else {
			KTRY(liship_cg_update_f64(n, alpha, p, q, c->x, r, lisg.reduce_out, lisg.reduce_work, lisg.stream));
			TRY(lisd_fetch(1, sums));
			sums[1] = sums[0];
		}
		TRY(resid_from_sumsq(c, r, sums[0], &nrm2));
		note(c, iter, nrm2);
		if (c->tol >= nrm2) { s->retcode = LIS_SUCCESS; s->iter = iter; s->resid = nrm2; goto done; }
		rho_old = rho;
		rho = sums[1];
	}
	s->retcode = LIS_MAXITER; s->iter = iter; s->resid = nrm2; err = LIS_MAXITER;
done:
	work_free(c);
	return err;
}

static LIS_INT run_cg_unfused(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter;
	const int n = c->n;
	TRY(work_alloc(c, 4));
	double *z = c->work[0], *q = c->work[1], *r = c->work[2], *p = c->work[3];
	double alpha, beta, rho, rho_old = 1.0, dot_pq, nrm2 = 0.0;
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	KTRY(liship_set_all_f64(n, 0.0, p, lisg.stream));
	for (iter = 1; iter <= c->maxiter; iter++) {
		TRY(d_psolve(c, r, z));
		TRY(lisd_dot(n, r, z, &rho));
		beta = rho / rho_old;
		KTRY(liship_xpay_f64(n, z, beta, p, lisg.stream));
		TRY(d_matvec(c, p, q));
		TRY(lisd_dot(n, p, q, &dot_pq));
		if (dot_pq == 0.0) { s->retcode = LIS_BREAKDOWN; s->iter = iter; s->resid = nrm2; err = LIS_BREAKDOWN; goto done; }
		alpha = rho / dot_pq;
		KTRY(liship_axpy_f64(n, alpha, p, c->x, lisg.stream));
		KTRY(liship_axpy_f64(n, -alpha, q, r, lisg.stream));
		TRY(d_resid(c, r, &nrm2));
		note(c, iter, nrm2);
		if (c->tol >= nrm2) { s->retcode = LIS_SUCCESS; s->iter = iter; s->resid = nrm2; goto done; }
		rho_old = rho;
	}
	s->retcode = LIS_MAXITER; s->iter = iter; s->resid = nrm2; err = LIS_MAXITER;
done:
	work_free(c);
	return err;
}

/* BiCGSTAB, lis_solver_bicgstab.c:186-290, same treatment:
 *   p = r + beta (p - omega v)                      one pass
 *   v = A M^-1 p ; <rtld,v>                         product + epilogue (M^-1 p aliases p without a preconditioner)
 *   s = r - alpha v ; ||s||                         one pass
 *   t = A M^-1 s ; <t,s>, <t,t>                     product + epilogue
 *   x += alpha phat + omega shat                    one pass
 *   r = s - omega t ; ||r|| ; rho' = <rtld,r>       one pass (rho' is :190 of the next iteration) */
static LIS_INT run_bicgstab_unfused(ctx_t *c);
static LIS_INT run_bicgstab_device(ctx_t *c);
static LIS_INT run_bicgstab(ctx_t *c)
{
	if (lisg.no_fusion) return run_bicgstab_unfused(c);
	if (device_scalars_ok(c)) return run_bicgstab_device(c);
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter;
	const int n = c->n;
	const int pre = c->dinv != NULL;
	TRY(work_alloc(c, pre ? 7 : 5));
	double *rtld = c->work[0], *r = c->work[1], *t = c->work[2], *p = c->work[3], *v = c->work[4];
	double *phat = pre ? c->work[5] : p, *shat = pre ? c->work[6] : r;
	double *sv = r;                                    /* s aliases r: lis_solver_bicgstab.c:160-161 */
	double alpha = 1.0, omega = 1.0, rho_old = 1.0, rho, beta, nrm2 = 0.0, d1, d2[2], sums[2];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	TRY(d_copy(c, r, rtld));                           /* shadow residual = r0 (lis_solver.c:1862) */
	TRY(lisd_dot(n, rtld, r, &rho));
	for (iter = 1; iter <= c->maxiter; iter++) {
		if (rho == 0.0) { s->retcode = LIS_BREAKDOWN; s->iter = iter; s->resid = nrm2; err = LIS_BREAKDOWN; goto done; }
		if (iter == 1) TRY(d_copy(c, r, p));
		else {
			beta = (rho / rho_old) * (alpha / omega);
			KTRY(liship_axpy_xpay_f64(n, -omega, v, r, beta, p, lisg.stream));
		}
		if (pre) KTRY(liship_pmul_f64(n, p, c->dinv, phat, lisg.stream));
		TRY(lisd_spmv_dot_launch(c->A, phat, v, rtld, 0));
		TRY(lisd_fetch(1, &d1));
		alpha = rho / d1;
		KTRY(liship_axpy_sumsq_f64(n, -alpha, v, r, lisg.reduce_out, lisg.reduce_work, lisg.stream));
		TRY(lisd_fetch(1, sums));
		TRY(resid_from_sumsq(c, sv, sums[0], &nrm2));
		if (nrm2 <= c->tol) {
			note(c, iter, nrm2);
			KTRY(liship_axpy_f64(n, alpha, phat, c->x, lisg.stream));
			s->retcode = LIS_SUCCESS; s->iter = iter; s->resid = nrm2; goto done;
		}
		if (pre) KTRY(liship_pmul_f64(n, sv, c->dinv, shat, lisg.stream));
		TRY(lisd_spmv_dot_launch(c->A, shat, t, sv, 1));
		TRY(lisd_fetch(2, d2));                         /* <t,s>, <t,t> (:267-268) */
		omega = d2[0] / d2[1];
		KTRY(liship_axpy2_f64(n, alpha, phat, omega, shat, c->x, lisg.stream));
		KTRY(liship_axpy_sumsq_dot_f64(n, -omega, t, r, rtld, lisg.reduce_out, lisg.reduce_work, lisg.stream));
		TRY(lisd_fetch(2, sums));
		TRY(resid_from_sumsq(c, r, sums[0], &nrm2));
		note(c, iter, nrm2);
		if (c->tol >= nrm2) { s->retcode = LIS_SUCCESS; s->iter = iter; s->resid = nrm2; goto done; }
		if (omega == 0.0) { s->retcode = LIS_BREAKDOWN; s->iter = iter; s->resid = nrm2; err = LIS_BREAKDOWN; goto done; }
		rho_old = rho;
		rho = sums[1];
	}
	s->retcode = LIS_MAXITER; s->iter = iter; s->resid = nrm2; err = LIS_MAXITER;
done:
	work_free(c);
	return err;
}

static LIS_INT run_bicgstab_unfused(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter;
	const int n = c->n;
	TRY(work_alloc(c, 7));
	double *rtld = c->work[0], *r = c->work[1], *t = c->work[2], *p = c->work[3], *v = c->work[4],
	       *phat = c->work[5], *shat = c->work[6];
	double *sv = r;                                    /* s aliases r: lis_solver_bicgstab.c:160-161 */
	double alpha = 1.0, omega = 1.0, rho_old = 1.0, rho, beta, nrm2 = 0.0, d1, d2[2];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	TRY(d_copy(c, r, rtld));                           /* shadow residual = r0 (lis_solver.c:1862) */
	for (iter = 1; iter <= c->maxiter; iter++) {
		TRY(lisd_dot(n, rtld, r, &rho));
		if (rho == 0.0) { s->retcode = LIS_BREAKDOWN; s->iter = iter; s->resid = nrm2; err = LIS_BREAKDOWN; goto done; }
		if (iter == 1) TRY(d_copy(c, r, p));
		else {
			beta = (rho / rho_old) * (alpha / omega);
			KTRY(liship_axpy_f64(n, -omega, v, p, lisg.stream));
			KTRY(liship_xpay_f64(n, r, beta, p, lisg.stream));
		}
		TRY(d_psolve(c, p, phat));
		TRY(d_matvec(c, phat, v));
		TRY(lisd_dot(n, rtld, v, &d1));
		alpha = rho / d1;
		KTRY(liship_axpy_f64(n, -alpha, v, r, lisg.stream));
		TRY(d_resid(c, sv, &nrm2));
		if (nrm2 <= c->tol) {
			note(c, iter, nrm2);
			KTRY(liship_axpy_f64(n, alpha, phat, c->x, lisg.stream));
			s->retcode = LIS_SUCCESS; s->iter = iter; s->resid = nrm2; goto done;
		}
		TRY(d_psolve(c, sv, shat));
		TRY(d_matvec(c, shat, t));
		TRY(lisd_dot2(n, t, sv, d2));                   /* <t,s> and <t,t> in one pass (:267-268) */
		omega = d2[0] / d2[1];
		KTRY(liship_axpy_f64(n, alpha, phat, c->x, lisg.stream));
		KTRY(liship_axpy_f64(n, omega, shat, c->x, lisg.stream));
		KTRY(liship_axpy_f64(n, -omega, t, r, lisg.stream));
		TRY(d_resid(c, r, &nrm2));
		note(c, iter, nrm2);
		if (c->tol >= nrm2) { s->retcode = LIS_SUCCESS; s->iter = iter; s->resid = nrm2; goto done; }
		if (omega == 0.0) { s->retcode = LIS_BREAKDOWN; s->iter = iter; s->resid = nrm2; err = LIS_BREAKDOWN; goto done; }
		rho_old = rho;
	}
	s->retcode = LIS_MAXITER; s->iter = iter; s->resid = nrm2; err = LIS_MAXITER;
done:
	work_free(c);
	return err;
}

/* BiCG, lis_solver_bicg.c:135-268 -- Lis's default solver (lis_solver.c:242).  The dual recurrence needs A^T:
 * lisd_spmv_t (lis_matvech.c) serves it from a transposed CSR in the reference's scatter order.
 *   p = M^-1 r + beta p ; p~ = M^-1 r~ + beta p~          one pass each (solve folded into the xpay)
 *   q = A p ; <p~,q>                                       product + epilogue
 *   q~ = A^T p~                                            product
 *   x += alpha p ; r -= alpha q ; ||r||                    one pass
 *   r~ -= alpha q~ ; rho' = <r~, M^-1 r>                   one pass (rho' is :187 of the next iteration) */
static LIS_INT run_bicg_device(ctx_t *c);
static LIS_INT run_bicg(ctx_t *c)
{
	if (!lisg.no_fusion && device_scalars_ok(c)) return run_bicg_device(c);
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter;
	const int n = c->n;
	TRY(work_alloc(c, 7));
	double *r = c->work[0], *rtld = c->work[1], *q = c->work[2], *qtld = c->work[3], *p = c->work[4], *ptld = c->work[5];
	double *z = c->work[6];                            /* M^-1 r, only kept for the Jacobi dot */
	double alpha, beta, rho, rho_old = 1.0, d1, nrm2 = 0.0, sums[2];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	TRY(d_copy(c, r, rtld));                           /* shadow residual = r0 (lis_solver.c:1862) */
	if (c->dinv) { KTRY(liship_pmul_f64(n, r, c->dinv, z, lisg.stream)); TRY(lisd_dot(n, rtld, z, &rho)); }
	else TRY(lisd_dot(n, rtld, r, &rho));
	for (iter = 1; iter <= c->maxiter; iter++) {
		if (rho == 0.0) { s->retcode = LIS_BREAKDOWN; s->iter = iter; s->resid = nrm2; err = LIS_BREAKDOWN; goto done; }
		beta = rho / rho_old;
		if (c->dinv) {
			KTRY(liship_pmul_xpay_f64(n, r, c->dinv, beta, p, lisg.stream));
			KTRY(liship_pmul_xpay_f64(n, rtld, c->dinv, beta, ptld, lisg.stream));
		} else {
			KTRY(liship_xpay_f64(n, r, beta, p, lisg.stream));
			KTRY(liship_xpay_f64(n, rtld, beta, ptld, lisg.stream));
		}
		TRY(lisd_spmv_dot_launch(c->A, p, q, ptld, 0));
		TRY(lisd_fetch(1, &d1));
		TRY(lisd_spmv_t(c->A, ptld, qtld));
		if (d1 == 0.0) { s->retcode = LIS_BREAKDOWN; s->iter = iter; s->resid = nrm2; err = LIS_BREAKDOWN; goto done; }
		alpha = rho / d1;
		KTRY(liship_cg_update_f64(n, alpha, p, q, c->x, r, lisg.reduce_out, lisg.reduce_work, lisg.stream));
		TRY(lisd_fetch(1, sums));
		TRY(resid_from_sumsq(c, r, sums[0], &nrm2));
		note(c, iter, nrm2);
		if (c->tol >= nrm2) { s->retcode = LIS_SUCCESS; s->iter = iter; s->resid = nrm2; goto done; }
		if (c->dinv) KTRY(liship_pmul_f64(n, r, c->dinv, z, lisg.stream));
		KTRY(liship_axpy_sumsq_dot_f64(n, -alpha, qtld, rtld, c->dinv ? z : r, lisg.reduce_out, lisg.reduce_work, lisg.stream));
		TRY(lisd_fetch(2, sums));
		rho_old = rho;
		rho = sums[1];
	}
	s->retcode = LIS_MAXITER; s->iter = iter; s->resid = nrm2; err = LIS_MAXITER;
done:
	work_free(c);
	return err;
}

typedef struct { double *r, *rtld, *q, *qtld, *p, *ptld, *z; } bicg_vars;
static LIS_INT bicg_batch(ctx_t *c, dev_loop *L, void *vars, LIS_INT queued, LIS_INT batch)
{
	LIS_INT err = 0;
	const int n = c->n;
	const bicg_vars *w = (const bicg_vars *)vars;
	double *r = w->r, *rtld = w->rtld, *q = w->q, *qtld = w->qtld, *p = w->p, *ptld = w->ptld, *z = w->z, *st = L->st;
	(void)queued;
	for (LIS_INT k = 0; k < batch; k++) {
		if (c->dinv) {
			KTRY(liship_pmul_xpay_dev_f64(n, r, c->dinv, st + LISHIP_KS_BETA, p, lisg.stream));
			KTRY(liship_pmul_xpay_dev_f64(n, rtld, c->dinv, st + LISHIP_KS_BETA, ptld, lisg.stream));
		} else {
			KTRY(liship_xpay_dev_f64(n, r, st + LISHIP_KS_BETA, p, lisg.stream));
			KTRY(liship_xpay_dev_f64(n, rtld, st + LISHIP_KS_BETA, ptld, lisg.stream));
		}
		TRY(dev_announce(L, LISHIP_STEP_BICG_ALPHA));
		TRY(lisd_spmv_dot_launch_to(c->A, p, q, ptld, 0, st + LISHIP_KS_DOT0));
		TRY(dev_step(L, LISHIP_STEP_BICG_ALPHA, LISHIP_KS_DOT0, 1));
		TRY(lisd_spmv_t(c->A, ptld, qtld));           /* recomputed from scratch each time: harmless behind a raised flag */
		TRY(dev_announce(L, LISHIP_STEP_BICG_RESID));
		KTRY(liship_cg_update_dev_f64(n, st + LISHIP_KS_ALPHA, p, q, NULL, c->x, r, st + LISHIP_KS_SUM0, lisg.reduce_work, lisg.stream));
		TRY(dev_step(L, LISHIP_STEP_BICG_RESID, LISHIP_KS_SUM0, 1));
		if (c->dinv) KTRY(liship_pmul_f64(n, r, c->dinv, z, lisg.stream));
		TRY(dev_announce(L, LISHIP_STEP_BICG_RHO));
		KTRY(liship_axpy_sumsq_dot_dev_f64(n, st + LISHIP_KS_NALPHA, qtld, rtld, c->dinv ? z : r, st + LISHIP_KS_SUM0, lisg.reduce_work, lisg.stream));
		TRY(dev_step(L, LISHIP_STEP_BICG_RHO, LISHIP_KS_SUM0, 2));
	}
done:
	return err;
}

static LIS_INT run_bicg_device(ctx_t *c)
{
	LIS_INT err = 0;
	const int n = c->n;
	dev_loop L = {0};
	TRY(work_alloc(c, 7));
	double *r = c->work[0], *rtld = c->work[1], *q = c->work[2], *qtld = c->work[3], *p = c->work[4], *ptld = c->work[5];
	double *z = c->work[6];
	double rho, init[LISHIP_KS_LEN] = {0};
	int st0 = initial_residual(c, r);
	if (st0) { work_free(c); return st0 < 0 ? -st0 : 0; }
	TRY(d_copy(c, r, rtld));
	if (c->dinv) { KTRY(liship_pmul_f64(n, r, c->dinv, z, lisg.stream)); TRY(lisd_dot(n, rtld, z, &rho)); }
	else TRY(lisd_dot(n, rtld, r, &rho));
	init[LISHIP_KS_RHO] = rho; init[LISHIP_KS_RHO_OLD] = 1.0; init[LISHIP_KS_BETA] = rho / 1.0;
	init[LISHIP_KS_BNRM] = c->bnrm; init[LISHIP_KS_TOL] = c->tol; init[LISHIP_KS_NOT_HALF] = 1.0;
	TRY(dev_loop_begin(c, &L, init));
	KTRY(liship_krylov_guard(L.st + LISHIP_KS_DONE));
	bicg_vars vars = { r, rtld, q, qtld, p, ptld, z };
	TRY(dev_loop_run(c, &L, bicg_batch, &vars));
done:
	err = dev_loop_finish(c, &L, err);
	work_free(c);
	return err;
}

/* RCCL job: replace this rank's sums in HBM by the sums over all ranks, without leaving the stream */
static LIS_INT globalize(double *slot, int count)
{
	if (lisg.comm_kind != 1) return LIS_SUCCESS;
	LISCHK(lisc_gather_device(slot, count));
	HIPCHK(liship_rank_fold_f64(count, lisg.gather_out, lisg.nprocs, slot, lisg.stream));
	return LIS_SUCCESS;
}

static LIS_INT run_gmres(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0;
	const int n = c->n, m = s->options[LIS_OPTIONS_RESTART], ld = m + 1;
	const int CS = (m + 1) * ld, SN = (m + 2) * ld;
	double *h = (double *)calloc((size_t)(ld + 1) * (size_t)(ld + 2), sizeof(double));   /* Hessenberg + rotations, host */
	double *g = (double *)calloc((size_t)ld + 1, sizeof(double));                      /* the reference's vector s */
	double nrm2 = 0.0, rnorm, t;
	double *hdev = NULL;
	int iter = 0;
	if (!h || !g) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", ld); goto done; }
	TRY(work_alloc(c, m + 3));
	double *r = c->work[0], *z = c->work[1], **v = &c->work[2];
	/* device-chained Gram-Schmidt: in an RCCL job each coefficient is made global on the device as well
	 * (all-gather + rank-order fold); the callback communicator folds on the host, so it takes the other branch */
	const int chained = !lisg.no_fusion && (lisg.nprocs == 1 || lisg.comm_kind == 1);
	if (chained) KTRY(lisd_malloc((void **)&hdev, sizeof(double) * (size_t)(m + 4)));
	int st = initial_residual(c, v[0]);                /* :193 leaves the unpreconditioned residual in v0 */
	if (st) { err = st < 0 ? -st : 0; goto done; }
	while (iter < c->maxiter) {
		TRY(lisd_nrm2(n, v[0], &rnorm));
		KTRY(liship_scale_f64(n, 1.0 / rnorm, v[0], lisg.stream));
		for (int j = 0; j <= m; j++) g[j] = 0.0;
		g[0] = rnorm;
		int i = 0, ii = 0, i1 = 0;
		do {
			iter++; i++;
			ii = i - 1; i1 = i;
			double *hc = h + (size_t)ii * ld;
			/* M^-1 v: without a preconditioner the reference copies (lis_precon.c:365-384); the product reads v itself */
			double *zin = z;
			if (c->dinv || lisg.no_fusion) TRY(d_psolve(c, v[ii], z)); else zin = v[ii];
			if (!chained) TRY(d_matvec(c, zin, v[i1]));
			if (chained) {
				/* modified Gram-Schmidt with the coefficients kept in HBM: step k reads h[k-1] from the previous
				 * step's reduction, so the whole column costs ONE host synchronisation instead of i+1; the first
				 * coefficient <A z, v0> is formed in the product's own pass */
				TRY(lisd_spmv_dot_launch_to(c->A, zin, v[i1], v[0], 0, hdev));
				TRY(globalize(hdev, 1));
				for (int k = 1; k < i; k++) {
					KTRY(liship_mgs_step_f64(n, hdev + k - 1, v[k - 1], v[i1], v[k], hdev + k, lisg.reduce_work, lisg.stream));
					TRY(globalize(hdev + k, 1));
				}
				KTRY(liship_mgs_step_f64(n, hdev + i - 1, v[i - 1], v[i1], NULL, hdev + i, lisg.reduce_work, lisg.stream));
				TRY(globalize(hdev + i, 1));
				KTRY(liship_scale_inv_norm_f64(n, hdev + i, v[i1], lisg.stream));
				if (i + 1 <= 256) {                        /* through the page-locked landing zone */
					KTRY(liship_memcpy_d2h(lisg.host_out, hdev, sizeof(double) * (size_t)(i + 1), lisg.stream));
					KTRY(liship_stream_synchronize(lisg.stream));
					memcpy(hc, lisg.host_out, sizeof(double) * (size_t)(i + 1));
				} else {
					KTRY(liship_memcpy_d2h(hc, hdev, sizeof(double) * (size_t)(i + 1), lisg.stream));
					KTRY(liship_stream_synchronize(lisg.stream));
				}
				hc[i1] = sqrt(hc[i1]);
			} else {
				for (int k = 0; k < i; k++) {                  /* modified Gram-Schmidt */
					TRY(lisd_dot(n, v[i1], v[k], &t));
					hc[k] = t;
					KTRY(liship_axpy_f64(n, -t, v[k], v[i1], lisg.stream));
				}
				TRY(lisd_nrm2(n, v[i1], &t));
				hc[i1] = t;
				KTRY(liship_scale_f64(n, 1.0 / t, v[i1], lisg.stream));
			}
			for (int k = 1; k <= ii; k++) {                /* apply the previous rotations */
				const int jj = k - 1;
				const double tt = hc[jj];
				double aa = h[jj + CS] * tt;  aa += h[jj + SN] * hc[k];
				double bb = -h[jj + SN] * tt; bb += h[jj + CS] * hc[k];
				hc[jj] = aa; hc[k] = bb;
			}
			double aa = hc[ii], bb = hc[i1];
			const double a2 = aa * aa, b2 = bb * bb;
			double rr = sqrt(a2 + b2);
			if (rr == 0.0) rr = 1.0e-17;
			h[ii + CS] = aa / rr;
			h[ii + SN] = bb / rr;
			g[i1] = -h[ii + SN] * g[ii];
			g[ii] =  h[ii + CS] * g[ii];
			aa  = h[ii + CS] * hc[ii];
			aa += h[ii + SN] * hc[i1];
			hc[ii] = aa;
			nrm2 = fabs(g[i1]) * c->bnrm;
			note(c, iter, nrm2);
			if (c->tol >= nrm2) break;
		} while (i < m && iter < c->maxiter);

		g[ii] = g[ii] / h[ii + (size_t)ii * ld];           /* back substitution */
		for (int k = 1; k <= ii; k++) {
			const int jj = ii - k;
			double tt = g[jj];
			for (int j = jj + 1; j <= ii; j++) tt -= h[jj + (size_t)j * ld] * g[j];
			g[jj] = tt / h[jj + (size_t)jj * ld];
		}
		if (!lisg.no_fusion) KTRY(liship_lincomb_f64(n, ii + 1, (const double *const *)v, g, 0, z, lisg.stream));   /* z = sum y_j v_j, one pass */
		else {
			KTRY(liship_scale_to_f64(n, g[0], v[0], z, lisg.stream));     /* z = y0 v0  (:290-296) */
			for (int j = 1; j <= ii; j++) KTRY(liship_axpy_f64(n, g[j], v[j], z, lisg.stream));
		}
		if (c->dinv || lisg.no_fusion) {
			TRY(d_psolve(c, z, r));
			KTRY(liship_axpy_f64(n, 1.0, r, c->x, lisg.stream));
		} else KTRY(liship_axpy_f64(n, 1.0, z, c->x, lisg.stream));      /* the copy of psolve_none left out: same addend */
		if (c->tol >= nrm2) { s->retcode = LIS_SUCCESS; s->iter = iter; s->resid = nrm2; goto done; }
		for (int j = 1; j <= i; j++) {
			const int jj = i1 - j + 1;
			g[jj - 1] = -h[jj - 1 + SN] * g[jj];
			g[jj]     =  h[jj - 1 + CS] * g[jj];
		}
		if (!lisg.no_fusion) {                               /* v0 += (g0-1) v0 + sum g_j v_j  (:323-329), one pass */
			g[0] = g[0] - 1.0;
			KTRY(liship_lincomb_f64(n, i1 + 1, (const double *const *)v, g, 1, v[0], lisg.stream));
		} else for (int j = 0; j <= i1; j++) {
			double tt = g[j];
			if (j == 0) tt = tt - 1.0;
			KTRY(liship_axpy_f64(n, tt, v[j], v[0], lisg.stream));
		}
	}
	s->retcode = LIS_MAXITER; s->iter = iter + 1; s->resid = nrm2; err = LIS_MAXITER;
done:
	work_free(c);
	(void)liship_free(hdev);
	free(h); free(g);
	return err;
}

/* ------------------------------------------------------------------ lis_solve */
LIS_INT lis_solve(LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x, LIS_SOLVER solver)
{
	LIS_PRECON precon;
	solver->A = A;
	if (solver->options[LIS_OPTIONS_PRECON] < 0 || solver->options[LIS_OPTIONS_PRECON] > LIS_PRECONNAME_MAX)
		return LISI_ERR(LIS_ERR_ILL_ARG, "Parameter LIS_OPTIONS_PRECON is %D (Set between 0 to %D)\n", solver->options[LIS_OPTIONS_PRECON], LIS_PRECONNAME_MAX);
	LIS_INT err = lis_precon_create(solver, &precon);
	if (err) { solver->retcode = err; return err; }
	err = lis_solve_kernel(A, b, x, solver, precon);
	lis_precon_destroy(precon);
	if (err) { solver->retcode = err; return err; }
	return LIS_SUCCESS;
}

/* the transposed copy of the matrix a solve runs on: A^T in the caller's numbering <-> (P A P^T)^T in the plan's (lis_internal.h: rt_*) */
static void swap_transposed(lisd_mat *d)
{
	int ti; int *tp; double *tv; liship_csr_plan_t tq;
	ti = d->t_ready; d->t_ready = d->rt_ready; d->rt_ready = ti;
	ti = d->t_nnz; d->t_nnz = d->rt_nnz; d->rt_nnz = ti;
	tp = d->t_ptr; d->t_ptr = d->rt_ptr; d->rt_ptr = tp;
	tp = d->t_index; d->t_index = d->rt_index; d->rt_index = tp;
	tv = d->t_value; d->t_value = d->rt_value; d->rt_value = tv;
	tq = d->t_plan; d->t_plan = d->rt_plan; d->rt_plan = tq;
}

LIS_INT lis_solve_kernel(LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x, LIS_SOLVER solver, LIS_PRECON precon)
{
	const LIS_INT nsolver = solver->options[LIS_OPTIONS_SOLVER], maxiter = solver->options[LIS_OPTIONS_MAXITER];
	const LIS_INT output = solver->options[LIS_OPTIONS_OUTPUT], storage = solver->options[LIS_OPTIONS_STORAGE];
	const LIS_INT conv = solver->options[LIS_OPTIONS_CONV_COND];
	const double tol = solver->params[LIS_PARAMS_RESID - LIS_OPTIONS_LEN];
	LIS_MATRIX Awork = A;
	LIS_INT err = 0;
	ctx_t c;
	memset(&c, 0, sizeof(c));
	int renumbered = 0;                        /* the solve runs in the numbering of a reordered plan (below) */
	const int *renum = NULL;
	double *renum_b = NULL, *renum_d = NULL;
	liship_csr_plan_t held_plan = NULL;
	int *held_ptr = NULL, *held_index = NULL;
	double *held_value = NULL;

	if (lisp_lazy()) lisp_check_handler();          /* a SIGSEGV handler the program installed since would take the protected pages' faults away */
	/* parameter checks, ref :482-537 */
	if (nsolver < 1 || nsolver > LIS_SOLVER_LEN) return LISI_ERR(LIS_ERR_ILL_ARG, "Parameter LIS_OPTIONS_SOLVER is %D (Set between 1 to %D)\n", nsolver, LIS_SOLVER_LEN);
	switch (nsolver) {
	case LIS_SOLVER_CG: case LIS_SOLVER_BICG: case LIS_SOLVER_BICGSTAB: case LIS_SOLVER_GMRES:
	case LIS_SOLVER_CGS: case LIS_SOLVER_CR: case LIS_SOLVER_GPBICG: case LIS_SOLVER_TFQMR: case LIS_SOLVER_BICGSAFE:
	case LIS_SOLVER_ORTHOMIN: case LIS_SOLVER_BICR: case LIS_SOLVER_CRS: case LIS_SOLVER_BICRSTAB: case LIS_SOLVER_GPBICR:
	case LIS_SOLVER_BICRSAFE: case LIS_SOLVER_FGMRES: case LIS_SOLVER_MINRES: case LIS_SOLVER_COCG: case LIS_SOLVER_COCR:
	case LIS_SOLVER_IDRS: case LIS_SOLVER_BICGSTABL: case LIS_SOLVER_IDR1: case LIS_SOLVER_JACOBI:
		break;
	default:      /* Gauss-Seidel / SOR: sparse triangular solves, a different kernel family */
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "solver %s is not served by liblis_amd\n", solver_names[nsolver]);
	}
	if (maxiter < 0) return LISI_ERR(LIS_ERR_ILL_ARG, "Parameter LIS_OPTIONS_MAXITER(=%D) is less than 0\n", maxiter);
	if (conv > 0 && (nsolver == LIS_SOLVER_GMRES || nsolver == LIS_SOLVER_TFQMR || nsolver == LIS_SOLVER_FGMRES || nsolver == LIS_SOLVER_MINRES || nsolver == LIS_SOLVER_JACOBI)) return LISI_ERR(LIS_ERR_ILL_ARG, "Option conv_cond is not implemented for solver %s\n", solver_names[nsolver]);
	if (solver->options[LIS_OPTIONS_PRECISION] != LIS_PRECISION_DOUBLE) return LISI_ERR(LIS_ERR_ILL_ARG, "Quad precision is not enabled\n");
	LIS_INT scale = solver->options[LIS_OPTIONS_SCALE];
	if ((nsolver == LIS_SOLVER_GMRES || nsolver == LIS_SOLVER_ORTHOMIN || nsolver == LIS_SOLVER_FGMRES) && solver->options[LIS_OPTIONS_RESTART] < 0)
		return LISI_ERR(LIS_ERR_ILL_ARG, "Parameter LIS_OPTIONS_RESTART(=%D) is less than 0\n", solver->options[LIS_OPTIONS_RESTART]);
	if (A->n != b->n || A->n != x->n) return LISI_ERR(LIS_ERR_ILL_ARG, "sizes of A, b and x do not match\n");
	solver->A = A; solver->b = b;
	solver->precision = LIS_PRECISION_DOUBLE;
	free(solver->rhistory);
	solver->rhistory = (LIS_REAL *)calloc((size_t)(maxiter + 2), sizeof(LIS_REAL));   /* zeroed: an iteration that breaks down before its residual is formed leaves 0, not heap contents */
	if (!solver->rhistory) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", maxiter + 2);
	solver->rhistory[0] = 1.0;
	solver->ptime = 0.0;

	/* -scale: A and b are scaled in place and stay scaled, CG turns jacobi into symm_diag (ref :686-724).  The
	 * Jacobi preconditioner was built from the UNSCALED matrix by lis_solve before this point, as in the reference. */
	if (scale && storage == LIS_MATRIX_BSR && scale == LIS_SCALE_JACOBI) {
		/* block-diagonal scaling: A becomes BSR, is split and multiplied by the inverse diagonal blocks, b likewise; the
		 * iterations then run on the split product (ref :659-690) */
		{	/* feasibility BEFORE A and b are touched: the solvers that multiply by A^T need the transposed split product, which lisd_mat_ready_t serves
			 * for square blocks only (lis_matvech.c) -- a refusal after the retype / split / scaling would hand the caller back a mutated A and b */
			const int needs_t = nsolver == LIS_SOLVER_BICG || nsolver == LIS_SOLVER_BICR || nsolver == LIS_SOLVER_CRS || nsolver == LIS_SOLVER_BICRSTAB ||
			                    nsolver == LIS_SOLVER_GPBICR || nsolver == LIS_SOLVER_BICRSAFE;
			const LIS_INT blk = solver->options[LIS_OPTIONS_STORAGE_BLOCK];
			const int square = A->matrix_type == LIS_MATRIX_BSR ? A->bnr == A->bnc : (blk > 0 || A->conv_bnr == A->conv_bnc);      /* (-storage_block b: b x b blocks) */
			if (needs_t && !square) {
				solver->retcode = LIS_ERR_NOT_IMPLEMENTED;
				return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "solver %s with -scale jacobi -storage bsr needs A^T x of a split matrix: served for square blocks only (A and b are untouched)\n", solver_names[nsolver]);
			}
		}
		if (A->matrix_type != LIS_MATRIX_BSR) err = lisi_matrix_retype(A, LIS_MATRIX_BSR, solver->options[LIS_OPTIONS_STORAGE_BLOCK]);
		if (!err) err = lisi_matrix_bscale_bsr(A, b);
		if (err) { solver->retcode = err; return err; }
		scale = 0;
	}
	if (scale) {
		if (!solver->d) { err = lis_vector_duplicate(A, &solver->d); if (err) { solver->retcode = err; return err; } }
		if (scale == LIS_SCALE_JACOBI && nsolver == LIS_SOLVER_CG) scale = LIS_SCALE_SYMM_DIAG;
		if (!A->is_scaled) err = lis_matrix_scale(A, b, solver->d, scale);
		else if (!b->is_scaled) {
			err = lisd_vec_host_write(b, 1);
			if (!err) err = lisd_vec_to_host(solver->d);
			if (!err) { for (LIS_INT i = 0; i < A->n; i++) b->value[i] = b->value[i] * solver->d->value[i]; lis_amd_vector_host_modified(b); }
		}
		if (err) { solver->retcode = err; return err; }
	}

	double t_itime = lis_wtime();
	/* -storage: the caller's matrix itself is converted and stays converted (ref lis_matrix_convert_self,
	 * lis_matrix_ops.c:326-372) */
	if (storage && storage != A->matrix_type) {
		err = lisi_matrix_retype(A, storage, storage == LIS_MATRIX_BSR ? solver->options[LIS_OPTIONS_STORAGE_BLOCK] : 0);
		if (err) { solver->retcode = err; return err; }
	}

	if (output) {
		lis_printf(LIS_COMM_WORLD, "initial vector x      : %s\n", solver->options[LIS_OPTIONS_INITGUESS_ZEROS] ? "all components set to 0" : "user defined");
		lis_printf(LIS_COMM_WORLD, "precision             : double\n");
		lis_printf(LIS_COMM_WORLD, "linear solver         : %s\n", solver_names[nsolver]);
		lis_printf(LIS_COMM_WORLD, "preconditioner        : %s\n", precon_names[precon->precon_type]);
		if (conv == LIS_CONV_COND_NRM2_R) lis_printf(LIS_COMM_WORLD, "convergence condition : ||b-Ax||_2 <= %6.1e * ||b-Ax_0||_2\n", tol);
		if (Awork->matrix_type == LIS_MATRIX_BSR) lis_printf(LIS_COMM_WORLD, "matrix storage format : %s(%D x %D)\n", storage_names[Awork->matrix_type - 1], Awork->bnr, Awork->bnc);
		else lis_printf(LIS_COMM_WORLD, "matrix storage format : %s\n", storage_names[Awork->matrix_type - 1]);
	}

	c.s = solver; c.A = Awork; c.n = A->n;
	c.output = output; c.maxiter = maxiter;
	if ((err = lisd_mat_ready(Awork))) goto out;
	{
		size_t len = (size_t)(Awork->np + Awork->pad) + 16;
		if (Awork->matrix_type == LIS_MATRIX_BSR) {
			const size_t a = (size_t)Awork->nc * Awork->bnc + 16, bb = (size_t)Awork->nr * Awork->bnr + 16;
			if (a > len) len = a;
			if (bb > len) len = bb;
		}
		c.len = len;
	}
	/* A plan that renumbered the matrix (liship.h: liship_csr_plan_reorder -- the caller's numbering has no locality): the WHOLE solve runs in the plan's numbering.
	 * b, x0 and 1/diag are gathered once, the iterations see P A P^T as the matrix (its plan with its fused reductions, its arrays), x is scattered back at the end:
	 * no product pays for a permutation.  The same recurrences on renumbered vectors; their sums fold in another order, so not in the reference-order mode. */
	{
		lisd_mat *dm = MDEV(Awork);
		const int needs_t = nsolver == LIS_SOLVER_BICG || nsolver == LIS_SOLVER_BICR || nsolver == LIS_SOLVER_CRS || nsolver == LIS_SOLVER_BICRSTAB ||
		                    nsolver == LIS_SOLVER_GPBICR || nsolver == LIS_SOLVER_BICRSAFE;
		liship_csr_plan_t in = NULL;
		const int *rp = NULL, *ri = NULL;
		const double *rv = NULL;
		/* A^T x: the HBM copy is transposed in HBM (lis_matvech.c) -- of an unsplit CSR matrix, that is P A P^T's while it is swapped in; it gets a set of fields of its own */
		/* (several ranks: each rank renumbers its own rows and owned columns, the ghost columns and the halo slots keep their place -- lisc_halo_renumbered --; the
		 * transposed copy of the swapped-in arrays has the np local columns as its rows like any rank's A^T, and the reverse halo adds the neighbours' sums at the
		 * renumbered export rows: lisd_spmv_t / lisc_reduce_device read the same swapped tables) */
		const int t_ok = !needs_t || (Awork->matrix_type == LIS_MATRIX_CSR && !Awork->is_splited);
		const int multi = lisg.nprocs > 1 && Awork->commtable;
		/* the renumbered form is built LAZILY: by the first solve that finds the plan has served lisg.reorder_after products (lis_device.c) */
		if (!scale && !Awork->is_scaled && !lisg.ref_reductions && !lisg.no_reorder && t_ok) { if ((err = lisd_mat_lazy_reorder(Awork))) goto out; }
		if (!scale && !Awork->is_scaled && !lisg.ref_reductions && !lisg.no_reorder && t_ok && dm->type == LIS_MATRIX_CSR && !dm->split_jad &&
		    dm->plan && (Awork->np == Awork->n || (multi && Awork->matrix_type == LIS_MATRIX_CSR)) && dm->n == Awork->n && liship_csr_plan_reordered_form(dm->plan, &in, &rp, &ri, &rv, &renum) == 0) {
			if (multi && (err = lisc_halo_renumbered(Awork, renum, liship_csr_plan_reordered_inner_rows(dm->plan)))) goto out;
			held_plan = dm->plan; held_ptr = dm->ptr; held_index = dm->index; held_value = dm->value;
			dm->plan = in; dm->ptr = (int *)rp; dm->index = (int *)ri; dm->value = (double *)rv;
			dm->solve_holds = 1;
			renumbered = 1;
			swap_transposed(dm);
		}
	}
	/* b: once over PCIe (or already resident); x: a private HBM iterate "xx" (ref :545-592), zero or copy of x */
	{
		double *db, *dx0;
		if ((err = lisd_vec_in(b, &db))) goto out;
		c.b = db;
		if ((err = lisd_pool_get(c.len * sizeof(double), (void **)&c.x))) goto out;
		int rc = liship_memset(c.x, 0, c.len * sizeof(double), lisg.stream);
		if (rc) { err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); goto out; }
		if (renumbered) {
			if ((err = lisd_pool_get(c.len * sizeof(double), (void **)&renum_b))) goto out;
			if ((rc = liship_memset(renum_b, 0, c.len * sizeof(double), lisg.stream)) || (rc = liship_permute_gather_f64(A->n, renum, db, renum_b, lisg.stream))) { err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); goto out; }
			c.b = renum_b;
		}
		if (!solver->options[LIS_OPTIONS_INITGUESS_ZEROS]) {
			if ((err = lisd_vec_in(x, &dx0))) goto out;
			rc = renumbered ? liship_permute_gather_f64(A->n, renum, dx0, c.x, lisg.stream) : liship_memcpy_d2d(c.x, dx0, sizeof(double) * (size_t)A->n, lisg.stream);
			if (rc) { err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); goto out; }
		}
	}
	lisg.last_uniform_jacobi = 0;
	lisg.last_graph_replays = 0;
	lisg.last_renumbered = renumbered;
	if (precon && precon->precon_type == LIS_PRECON_TYPE_JACOBI) {
		if ((err = lisd_vec_in(precon->D, &c.dinv))) goto out;
		if (renumbered) {
			int rc;
			if ((err = lisd_pool_get(c.len * sizeof(double), (void **)&renum_d))) goto out;
			if ((rc = liship_memset(renum_d, 0, c.len * sizeof(double), lisg.stream)) || (rc = liship_permute_gather_f64(A->n, renum, c.dinv, renum_d, lisg.stream))) { err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); goto out; }
			c.dinv = renum_d;
		}
		if (nsolver == LIS_SOLVER_CG && !lisg.no_uniform_jacobi) {        /* (every rank: the count below is a collective) */
			/* a constant diagonal (constant-coefficient stencils): z = r.*dinv is r*dinv[0] in every bit, and the fused CG passes
			 * need not read the array -- one counting pass per solve decides (all ranks: the count is folded like any sum) */
			double d0 = 0.0, differ = 1.0;
			int rc = A->n > 0 ? liship_memcpy_d2h(&d0, c.dinv, sizeof(double), lisg.stream) : 0;
			if (!rc) rc = liship_stream_synchronize(lisg.stream);
			if (!rc) rc = liship_count_ne_f64(A->n, c.dinv, d0, lisg.reduce_out, lisg.reduce_work, lisg.stream);
			if (rc) { err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); goto out; }
			if ((err = lisd_fetch(1, &differ))) goto out;
			if (differ == 0.0) { c.duniform = 1; c.dconst = d0; }
			lisg.last_uniform_jacobi = c.duniform;
		}
	}
	solver->x = NULL; solver->xx = x; solver->precon = precon;

	switch (nsolver) {
	case LIS_SOLVER_CG:       err = run_cg(&c); break;
	case LIS_SOLVER_BICG:     err = run_bicg(&c); break;
	case LIS_SOLVER_BICGSTAB: err = run_bicgstab(&c); break;
	case LIS_SOLVER_CGS:      err = lisk_cgs(&c); break;
	case LIS_SOLVER_CR:       err = lisk_cr(&c); break;
	case LIS_SOLVER_GPBICG:   err = lisk_gpbicg(&c); break;
	case LIS_SOLVER_TFQMR:    err = lisk_tfqmr(&c); break;
	case LIS_SOLVER_BICGSAFE: err = lisk_bicgsafe(&c); break;
	case LIS_SOLVER_ORTHOMIN: err = lisk_orthomin(&c); break;
	case LIS_SOLVER_BICR:     err = lisk_bicr(&c); break;
	case LIS_SOLVER_CRS:      err = lisk_crs(&c); break;
	case LIS_SOLVER_BICRSTAB: err = lisk_bicrstab(&c); break;
	case LIS_SOLVER_GPBICR:   err = lisk_gpbicr(&c); break;
	case LIS_SOLVER_BICRSAFE: err = lisk_bicrsafe(&c); break;
	case LIS_SOLVER_FGMRES:   err = lisk_fgmres(&c); break;
	case LIS_SOLVER_MINRES:   err = lisk_minres(&c); break;
	case LIS_SOLVER_IDRS: case LIS_SOLVER_IDR1: err = lisk_idrs(&c); break;
	case LIS_SOLVER_BICGSTABL: err = lisk_bicgstabl(&c); break;
	case LIS_SOLVER_JACOBI:   err = lisk_jacobi(&c); break;
	case LIS_SOLVER_COCG:     err = run_cg(&c); break;          /* real build: lis_cocg is lis_cg's arithmetic (lis_solver_cg.c:632-739) */
	case LIS_SOLVER_COCR:     err = lisk_cr(&c); break;         /* and lis_cocr is lis_cr's (:1155-1274) */
	default:                  err = run_gmres(&c); break;
	}
	const LIS_INT solver_code = err;
	if (err == LIS_MAXITER || err == LIS_BREAKDOWN) err = 0;    /* reported through retcode only (ref :874,952) */
	else if (err) goto out;
	solver->retcode = solver_code;

	/* xx -> x (ref :890) */
	{
		double *dx;
		if ((err = lisd_vec_out(x, &dx))) goto out;
		int rc;
		if (scale == LIS_SCALE_SYMM_DIAG) {             /* x = xx .* d  (ref :876-885) */
			double *dd;
			if ((err = lisd_vec_in(solver->d, &dd))) goto out;
			rc = liship_pmul_f64(A->n, c.x, dd, dx, lisg.stream);
		} else if (renumbered) rc = liship_permute_scatter_f64(A->n, renum, c.x, dx, lisg.stream);
		else rc = liship_memcpy_d2d(dx, c.x, sizeof(double) * (size_t)A->n, lisg.stream);
		if (rc) { err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); goto out; }
		if ((err = lisd_vec_done(x))) goto out;
	}
	{
		int rc = liship_stream_synchronize(lisg.stream);
		if (rc) { err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); goto out; }
	}
	t_itime = lis_wtime() - t_itime;
	solver->itime = t_itime; solver->p_i_time = 0.0; solver->p_c_time = 0.0; solver->ptime = 0.0;
	solver->time = t_itime;
	solver->iter2 = solver->iter;
	if (output) {
		if (solver_code) lis_printf(LIS_COMM_WORLD, "linear solver status  : %s(code=%D)\n\n", retcode_names[solver_code], solver_code);
		else lis_printf(LIS_COMM_WORLD, "linear solver status  : normal end\n\n");
	}
out:
	if (renumbered) {           /* the caller's matrix again, whatever happened */
		lisd_mat *dm = MDEV(Awork);
		dm->plan = held_plan; dm->ptr = held_ptr; dm->index = held_index; dm->value = held_value;
		dm->solve_holds = 0;
		swap_transposed(dm);
		lisc_halo_restore(Awork);
	}
	if (renum_b) lisd_pool_put(renum_b, c.len * sizeof(double));
	if (renum_d) lisd_pool_put(renum_d, c.len * sizeof(double));
	if (c.x) lisd_pool_put(c.x, c.len * sizeof(double));
	solver->precon = NULL;
	if (err) solver->retcode = err;
	return err;
}
