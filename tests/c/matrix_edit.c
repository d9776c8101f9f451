/*
 * matrix_edit.c -- a plain Lis program (no lis_amd_* call) that edits A->value[] between two solves, as programs written against the reference do: the
 * reference adopts the caller's arrays (src/matrix/lis_matrix_csr.c:98-103) and reads them live on every product (src/matvec/lis_matvec_csr.c:97-109).
 * Built twice by tests/test_matrix_edit_gpu.py -- against liblis_amd.so and against the reference library (oracle/_ref) -- and the outputs are compared.
 *
 *   matrix_edit <lis|malloc> N      arrays from lis_matrix_malloc_csr (lis_matrix_csr.c:170), or from the program's own malloc
 *
 * 1. the N^3 7-point Poisson matrix (test/test3.c:114-127), y = A w, CG + Jacobi on A x = A 1
 * 2. every diagonal entry 6 -> 6 + (i mod 5) written straight into A->value[], the same three steps again
 * 3. one off-diagonal entry changed through A->value, y = A w again (a single write, far from the first)
 * Printed: hexfloat sums and samples of y (bit-exact quantities), iteration counts, the solutions' 2-norms.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lis.h"

static void report(const char *tag, LIS_MATRIX A, LIS_VECTOR w, LIS_VECTOR y, LIS_VECTOR b, LIS_VECTOR x, LIS_VECTOR ones, int solve)
{
	LIS_INT n, gn, iter = -1;
	LIS_REAL resid = -1.0, xn = 0.0;
	lis_matrix_get_size(A, &n, &gn);
	lis_matvec(A, w, y);
	double s = 0.0;
	for (LIS_INT i = 0; i < n; i++) s += y->value[i];           /* left to right: the same rounding wherever y came from */
	printf("%s y_sum %a y0 %a ymid %a ylast %a\n", tag, s, y->value[0], y->value[n / 2], y->value[n - 1]);
	if (solve) {
		LIS_SOLVER solver;
		lis_matvec(A, ones, b);
		lis_solver_create(&solver);
		lis_solver_set_option("-i cg -p jacobi -tol 1e-12 -maxiter 5000 -print none", solver);
		lis_vector_set_all(0.0, x);
		lis_solve(A, b, x, solver);
		lis_solver_get_iter(solver, &iter);
		lis_solver_get_residualnorm(solver, &resid);
		lis_vector_nrm2(x, &xn);
		printf("%s iter %d resid_ok %d xnorm %.10e\n", tag, (int)iter, resid <= 1e-12, xn);
		lis_solver_destroy(solver);
	}
	fflush(stdout);
}

int main(int argc, char *argv[])
{
	lis_initialize(&argc, &argv);
	if (argc < 3) { printf("usage: matrix_edit <lis|malloc> N\n"); return 2; }
	const int own = strcmp(argv[1], "malloc") == 0;
	const LIS_INT N = atoi(argv[2]), n = N * N * N;
	LIS_INT *ptr, *index;
	LIS_SCALAR *value;
	if (own) {
		ptr = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(n + 1));
		index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(7 * n));
		value = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(7 * n));
	} else if (lis_matrix_malloc_csr(n, 7 * n, &ptr, &index, &value)) return 3;
	LIS_INT k = 0;
	ptr[0] = 0;
	for (LIS_INT i = 0; i < N; i++) for (LIS_INT j = 0; j < N; j++) for (LIS_INT l = 0; l < N; l++) {
		const LIS_INT r = (i * N + j) * N + l;
		if (i > 0)     { index[k] = r - N * N; value[k++] = -1.0; }
		if (j > 0)     { index[k] = r - N;     value[k++] = -1.0; }
		if (l > 0)     { index[k] = r - 1;     value[k++] = -1.0; }
		index[k] = r; value[k++] = 6.0;
		if (l < N - 1) { index[k] = r + 1;     value[k++] = -1.0; }
		if (j < N - 1) { index[k] = r + N;     value[k++] = -1.0; }
		if (i < N - 1) { index[k] = r + N * N; value[k++] = -1.0; }
		ptr[r + 1] = k;
	}
	LIS_MATRIX A;
	lis_matrix_create(LIS_COMM_WORLD, &A);
	lis_matrix_set_size(A, n, 0);
	if (lis_matrix_set_csr(k, ptr, index, value, A) || lis_matrix_assemble(A)) return 4;
	LIS_VECTOR w, y, b, x, ones;
	lis_vector_duplicate(A, &w); lis_vector_duplicate(A, &y); lis_vector_duplicate(A, &b); lis_vector_duplicate(A, &x); lis_vector_duplicate(A, &ones);
	lis_vector_set_all(1.0, ones);
	for (LIS_INT i = 0; i < n; i++) lis_vector_set_value(LIS_INS_VALUE, i, 0.25 + (double)((i * 37) % 101) / 64.0, w);

	report("first", A, w, y, b, x, ones, 1);
	for (LIS_INT r = 0; r < n; r++)                              /* the program edits the matrix in place, through the struct's own field */
		for (LIS_INT q = A->ptr[r]; q < A->ptr[r + 1]; q++)
			if (A->index[q] == r) A->value[q] = 6.0 + (double)(r % 5);
	report("second", A, w, y, b, x, ones, 1);
	A->value[A->ptr[n - 2]] = -0.5;                              /* one more write, after the arrays were uploaded again */
	report("third", A, w, y, b, x, ones, 0);

	lis_vector_destroy(w); lis_vector_destroy(y); lis_vector_destroy(b); lis_vector_destroy(x); lis_vector_destroy(ones);
	lis_matrix_destroy(A);                                       /* frees the arrays either way (is_destroy, lis_matrix.c:387-396) */
	lis_finalize();
	return 0;
}
