/*
 * pages_threads.c -- what an OpenMP Lis program does around its library calls: several threads read (or write) x->value at once.
 * Test driver of lis_amd/csrc/host/lis_pages.c (built by tests/test_host_cpu.py / tests/test_pages_gpu.py with gcc -fopenmp against
 * include/ and liblis_amd.so).  Every mode prints "ok ..." and exits 0, or prints what it saw and exits 1.
 *
 *   cpu-readers  T R   no GPU: a host buffer plays the HBM copy (lis_amd_vector_page_test_source, copied home in two halves, the second held until another
 *                      thread waits for the copy);
 *                      T threads read disjoint slices of v->value at once, R rounds over the same vector (same thread -> same slice:
 *                      the second and third round fault at the addresses of the first)
 *   cpu-writers  T     the same, the threads WRITE their slices
 *   cpu-fwrite         a protected v->value handed to stdio and to the kernel: a small fwrite (copied by stdio in user space: faults, served),
 *                      write(2) of the protected pages (EFAULT) -- and the documented ways round it (lis_amd_vector_sync_host, eager coherence)
 *   gpu-solve    T N   CG + Jacobi on the N^3 Poisson matrix (test/test3.c:114-127), then T threads read x->value at once: every entry
 *                      must be the solution's, under page protection and -- the same bits -- under eager coherence
 */
#include <errno.h>
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "lis.h"
#include "lis_amd.h"

static int fail(const char *what, long a, long b) { printf("FAIL %s: %ld %ld\n", what, a, b); return 1; }

static int cpu_threads(int T, int rounds, int writers)
{
	const LIS_INT n = 1 << 20;
	LIS_VECTOR v;
	lis_vector_create(LIS_COMM_WORLD, &v);
	lis_vector_set_size(v, n, 0);
	double *src = (double *)malloc(sizeof(double) * (size_t)n);
	long waits0 = lis_amd_page_fault_waits();
	LIS_INT r0, w0, r1, w1;
	lis_amd_page_faults(&r0, &w0);
	for (int round = 0; round < rounds; round++) {
		for (LIS_INT i = 0; i < n; i++) src[i] = i + 0.5 + round;
		lis_amd_vector_host_modified(v);
		for (LIS_INT i = 0; i < n; i++) v->value[i] = -7.0;          /* what a thread must never see once "a kernel wrote v" */
		/* the copy comes home in two halves; with several threads the second half is held until another thread waits for the copy (deterministic), else 150 ms apart */
		if (lis_amd_vector_page_test_source(v, src, T > 1 ? -1 : 150) != 0) return fail("no page protection here", 0, 0);
		if (lis_amd_vector_page_state(v) != 2) return fail("state before", lis_amd_vector_page_state(v), 2);
		long bad = 0;
#pragma omp parallel num_threads(T) reduction(+ : bad)
		{
			const int t = omp_get_thread_num();
			const LIS_INT is = (LIS_INT)((long long)n * t / T), ie = (LIS_INT)((long long)n * (t + 1) / T);
			if (writers) {
				for (LIS_INT i = ie - 1; i >= is; i--) { if (v->value[i] != src[i]) bad++; v->value[i] = 2.0 * i; }
			} else {
				for (LIS_INT i = ie - 1; i >= is; i--) if (v->value[i] != src[i]) bad++;      /* (downwards: the late half of the copy first) */
			}
		}
		if (bad) return fail("threads saw stale entries", bad, round);
		const int want_state = writers ? 0 : 1;
		if (lis_amd_vector_page_state(v) != want_state) return fail("state after", lis_amd_vector_page_state(v), want_state);
		if (writers) for (LIS_INT i = 0; i < n; i++) if (v->value[i] != 2.0 * i) return fail("written value lost", i, round);
	}
	lis_amd_vector_page_test_source(v, NULL, 0);
	lis_amd_page_faults(&r1, &w1);
	if (r1 - r0 != rounds) return fail("one read fault per round brings the vector home", r1 - r0, rounds);
	if (writers && w1 - w0 < rounds) return fail("write faults", w1 - w0, rounds);
	printf("ok threads=%d rounds=%d writers=%d read_faults=%d write_faults=%d waits=%ld\n", T, rounds, writers, (int)(r1 - r0), (int)(w1 - w0),
	       (long)lis_amd_page_fault_waits() - waits0);
	lis_vector_destroy(v);
	free(src);
	return 0;
}

static int cpu_fwrite(void)
{
	FILE *f = tmpfile();          /* (a real file: /dev/null never reads the buffer) */
	if (!f) return fail("open", errno, 0);
	/* small: stdio copies into its own buffer in user space -- the copy faults, the handler serves it */
	LIS_VECTOR s;
	lis_vector_create(LIS_COMM_WORLD, &s);
	lis_vector_set_size(s, 100, 0);
	double small_src[100];
	for (int i = 0; i < 100; i++) small_src[i] = i;
	lis_amd_vector_page_test_source(s, small_src, 0);
	size_t got = fwrite(s->value, sizeof(double), 100, f);
	if (got != 100 || s->value[99] != 99.0) return fail("small fwrite", (long)got, 100);
	/* large: what stdio does with a big buffer depends on the stream's state (glibc tops up its own buffer first -- a user-space copy that
	 * faults and is served -- or hands the caller's pointer to write(2) when it has no buffer yet).  write(2) itself is the hard case: the
	 * kernel does not take the fault on the program's behalf, it returns EFAULT */
	const LIS_INT n = 1 << 20;
	LIS_VECTOR v;
	lis_vector_create(LIS_COMM_WORLD, &v);
	lis_vector_set_size(v, n, 0);
	double *src = (double *)malloc(sizeof(double) * (size_t)n);
	for (LIS_INT i = 0; i < n; i++) src[i] = i;
	lis_amd_vector_page_test_source(v, src, 0);
	fflush(f);
	errno = 0;
	const long wrote = (long)write(fileno(f), v->value, sizeof(double) * (size_t)n);
	const int e = errno;
	if (wrote >= 0) return fail("write(2) of protected pages was expected to fail with EFAULT", wrote, n);
	if (e != EFAULT) return fail("errno", e, EFAULT);
	if (lis_amd_vector_page_state(v) != 2) return fail("the failed write must not have changed the pages", lis_amd_vector_page_state(v), 2);
	/* the documented way: make the host array current first */
	if (lis_amd_vector_sync_host(v) != 0) return fail("sync_host", 0, 0);
	got = fwrite(v->value, sizeof(double), (size_t)n, f);
	if (got != (size_t)n) return fail("fwrite after lis_amd_vector_sync_host", (long)got, n);
	for (LIS_INT i = 0; i < n; i += 4097) if (v->value[i] != (double)i) return fail("contents", i, 0);
	/* ... or eager coherence: the pages keep full access */
	lis_amd_set_coherence(0);
	lis_amd_vector_device_modified(v);
	if (lis_amd_vector_page_state(v) == 2) return fail("eager coherence never takes the access away", lis_amd_vector_page_state(v), 2);
	got = fwrite(v->value, sizeof(double), (size_t)n, f);
	if (got != (size_t)n) return fail("fwrite under eager coherence", (long)got, n);
	lis_amd_set_coherence(1);
	fclose(f);
	printf("ok fwrite small=served write2=EFAULT sync_host=full eager=full\n");
	return 0;
}

static int gpu_solve(int T, int N)
{
	const LIS_INT n = (LIS_INT)N * N * N;
	unsigned long long sums[2] = {0, 0};
	for (int eager = 0; eager < 2; eager++) {
		lis_amd_set_coherence(eager ? 0 : 1);
		LIS_MATRIX A;
		LIS_VECTOR b, x;
		LIS_SOLVER solver;
		lis_matrix_create(LIS_COMM_WORLD, &A);
		lis_matrix_set_size(A, n, 0);
		for (LIS_INT i = 0; i < N; i++) for (LIS_INT j = 0; j < N; j++) for (LIS_INT k = 0; k < N; k++) {
			const LIS_INT ii = (i * N + j) * N + k;
			if (i > 0) lis_matrix_set_value(LIS_INS_VALUE, ii, ii - N * N, -1.0, A);
			if (i < N - 1) lis_matrix_set_value(LIS_INS_VALUE, ii, ii + N * N, -1.0, A);
			if (j > 0) lis_matrix_set_value(LIS_INS_VALUE, ii, ii - N, -1.0, A);
			if (j < N - 1) lis_matrix_set_value(LIS_INS_VALUE, ii, ii + N, -1.0, A);
			if (k > 0) lis_matrix_set_value(LIS_INS_VALUE, ii, ii - 1, -1.0, A);
			if (k < N - 1) lis_matrix_set_value(LIS_INS_VALUE, ii, ii + 1, -1.0, A);
			lis_matrix_set_value(LIS_INS_VALUE, ii, ii, 6.0, A);
		}
		lis_matrix_set_type(A, LIS_MATRIX_CSR);
		lis_matrix_assemble(A);
		lis_vector_duplicate(A, &b);
		lis_vector_duplicate(A, &x);
		LIS_VECTOR u;
		lis_vector_duplicate(A, &u);
		lis_vector_set_all(1.0, u);
		lis_matvec(A, u, b);
		lis_solver_create(&solver);
		lis_solver_set_option("-i cg -p jacobi -tol 1e-12 -maxiter 2000", solver);
		for (int round = 0; round < 3; round++) {             /* three solves: the same threads fault at the same addresses again */
			if (lis_solve(A, b, x, solver) != 0) return fail("lis_solve", 0, 0);
			if (!eager && lis_amd_vector_page_state(x) != 2) return fail("x should be in HBM only after the solve", lis_amd_vector_page_state(x), 2);
			long bad = 0;
			unsigned long long h = 0;
#pragma omp parallel num_threads(T) reduction(+ : bad) reduction(^ : h)
			{
				const int t = omp_get_thread_num();
				const LIS_INT is = (LIS_INT)((long long)n * t / T), ie = (LIS_INT)((long long)n * (t + 1) / T);
				for (LIS_INT i = ie - 1; i >= is; i--) {
					const double xi = x->value[i];
					if (!(fabs(xi - 1.0) < 1e-8)) bad++;      /* the solution is 1 everywhere; a stale page holds the previous contents or 0 */
					unsigned long long bits;
					memcpy(&bits, &xi, 8);
					h ^= bits * (unsigned long long)(2 * i + 1);
				}
			}
			if (bad) return fail("threads saw entries that are not the solution's", bad, round);
			if (round == 0) sums[eager] = h; else if (h != sums[eager]) return fail("solution bits changed between rounds", round, 0);
			/* the program scribbles over x between solves (x0 is ignored: -initx_zeros true): host writes, several threads at once */
#pragma omp parallel for num_threads(T)
			for (LIS_INT i = 0; i < n; i++) x->value[i] = -3.0;
		}
		lis_solver_destroy(solver);
		lis_vector_destroy(u); lis_vector_destroy(x); lis_vector_destroy(b);
		lis_matrix_destroy(A);
	}
	lis_amd_set_coherence(1);
	if (sums[0] != sums[1]) return fail("page protection and eager coherence disagree on the solution's bits", 0, 0);
	printf("ok gpu-solve threads=%d N=%d waits=%d\n", T, N, (int)lis_amd_page_fault_waits());
	return 0;
}

int main(int argc, char **argv)
{
	lis_initialize(&argc, &argv);
	int rc = 2;
	if (argc >= 4 && !strcmp(argv[1], "cpu-readers")) rc = cpu_threads(atoi(argv[2]), atoi(argv[3]), 0);
	else if (argc >= 3 && !strcmp(argv[1], "cpu-writers")) rc = cpu_threads(atoi(argv[2]), 2, 1);
	else if (argc >= 2 && !strcmp(argv[1], "cpu-fwrite")) rc = cpu_fwrite();
	else if (argc >= 4 && !strcmp(argv[1], "gpu-solve")) rc = gpu_solve(atoi(argv[2]), atoi(argv[3]));
	else printf("usage: pages_threads cpu-readers T R | cpu-writers T | cpu-fwrite | gpu-solve T N\n");
	fflush(stdout);
	if (rc == 0) lis_finalize();
	return rc;
}
