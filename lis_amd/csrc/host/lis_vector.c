/*
 * lis_vector.c -- LIS_VECTOR lifecycle and the BLAS-1 entry points of the Lis API on HBM copies.
 *
 * Lifecycle and argument checks follow the reference (src/vector/lis_vector.c: create :116, set_size
 * :156, duplicate :370, destroy :353, set_value :561, get_values :814); the arithmetic of every
 * lis_vector_* operation (src/vector/lis_vector_opv.c, lis_vector_ops.c) runs in the HIP kernels of
 * kernels/vector_ops.hip -- there is no host loop to fall back to.
 */
#include <stdio.h>
#include "lis_internal.h"

static LIS_INT vec_alloc(LIS_VECTOR *out)
{
	lisi_vector *v = (lisi_vector *)calloc(1, sizeof(lisi_vector));
	if (!v) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)sizeof(lisi_vector));
	v->pub.label = LIS_LABEL_VECTOR;
	v->pub.status = LIS_VECTOR_NULL;
	v->pub.is_destroy = LIS_TRUE;
	v->dev.host_valid = 1;
	lisi_register(v, LISI_KIND_VECTOR);
	*out = &v->pub;
	return LIS_SUCCESS;
}

static LIS_INT vec_check(LIS_VECTOR v)
{
	if (!lisi_is_registered(v)) return LISI_ERR(LIS_ERR_ILL_ARG, "vector v is undefined\n");
	return LIS_SUCCESS;
}

LIS_INT lis_vector_create(LIS_Comm comm, LIS_VECTOR *vec)
{
	*vec = NULL;
	LISCHK(vec_alloc(vec));
	(*vec)->comm = comm;
	(*vec)->nprocs = lisg.nprocs ? lisg.nprocs : 1;
	(*vec)->my_rank = lisg.rank;
	return LIS_SUCCESS;
}

LIS_INT lis_vector_set_size(LIS_VECTOR vec, LIS_INT local_n, LIS_INT global_n)
{
	if (global_n > 0 && local_n > global_n)
		return LISI_ERR(LIS_ERR_ILL_ARG, "local n(=%D) is larger than global n(=%D)\n", local_n, global_n);
	if (local_n < 0 || global_n < 0)
		return LISI_ERR(LIS_ERR_ILL_ARG, "local n(=%D) or global n(=%D) are less than 0\n", local_n, global_n);
	LIS_INT *ranges, is, ie, nprocs, my_rank;
	LISCHK(lisc_ranges_create(vec->comm, &local_n, &global_n, &ranges, &is, &ie, &nprocs, &my_rank));
	free(vec->ranges);                           /* (a second set_size: the first one's storage goes, its pages leave the fault registry) */
	vec->ranges = ranges;
	if (vec->value) {
		lisd_vec_free(vec);
		if (VDEV(vec)->region) lisp_free(vec); else free(vec->value);
		vec->value = NULL;
	}
	vec->value = lisp_alloc(vec, (size_t)local_n);          /* pages of its own, zero-filled: their protection follows the HBM copy (lis_pages.c) */
	if (!vec->value) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", local_n);
	VDEV(vec)->hlen = (size_t)local_n;
	vec->is_copy = LIS_TRUE;
	vec->status = LIS_VECTOR_ASSEMBLED;
	vec->n = local_n; vec->gn = global_n; vec->np = local_n;
	vec->my_rank = my_rank; vec->nprocs = nprocs;
	vec->is = is; vec->ie = ie;
	return LIS_SUCCESS;
}

LIS_INT lis_vector_duplicate(void *vin, LIS_VECTOR *vout)
{
	LIS_VECTOR src = (LIS_VECTOR)vin;              /* vectors and matrices share this header (ref lis.h:513-530) */
	if (src->label != LIS_LABEL_VECTOR && src->label != LIS_LABEL_MATRIX)
		return LISI_ERR(LIS_ERR_ILL_ARG, "First argument is not LIS_VECTOR or LIS_MATRIX\n");
	*vout = NULL;
	LISCHK(vec_alloc(vout));
	LIS_VECTOR v = *vout;
	const size_t len = (size_t)(src->np + src->pad);
	v->value = lisp_alloc(v, len);
	if (!v->value) { lis_vector_destroy(v); *vout = NULL; return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)len); }
	VDEV(v)->hlen = len;
	if (src->ranges) {
		v->ranges = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(src->nprocs + 1));
		memcpy(v->ranges, src->ranges, sizeof(LIS_INT) * (size_t)(src->nprocs + 1));
	}
	v->is_copy = LIS_TRUE;
	v->status = LIS_VECTOR_ASSEMBLED;
	v->precision = LIS_PRECISION_DEFAULT;
	v->n = src->n; v->gn = src->gn; v->np = src->np; v->pad = src->pad;
	v->comm = src->comm; v->my_rank = src->my_rank; v->nprocs = src->nprocs;
	v->is = src->is; v->ie = src->ie; v->origin = src->origin;
	v->is_destroy = src->is_destroy;
	return LIS_SUCCESS;
}

LIS_INT lis_vector_destroy(LIS_VECTOR vec)
{
	if (vec && lisi_is_registered(vec)) {
		lisd_vec_free(vec);
		if (VDEV(vec)->region) lisp_free(vec);
		else if (vec->value && vec->is_destroy) free(vec->value);
		free(vec->work);
		free(vec->ranges);
		lisi_unregister(vec);
		free(vec);
	}
	return LIS_SUCCESS;
}

LIS_INT lis_vector_is_null(LIS_VECTOR v) { return v->status == LIS_VECTOR_NULL ? LIS_TRUE : LIS_FALSE; }

LIS_INT lis_vector_get_size(LIS_VECTOR v, LIS_INT *local_n, LIS_INT *global_n)
{
	LISCHK(vec_check(v));
	*local_n = v->n; *global_n = v->gn;
	return LIS_SUCCESS;
}

LIS_INT lis_vector_get_range(LIS_VECTOR v, LIS_INT *is, LIS_INT *ie)
{
	LISCHK(vec_check(v));
	*is = v->is; *ie = v->ie;
	return LIS_SUCCESS;
}

/* host-side element access: make value[] current first; a writer also makes the pages writable and marks the HBM copy stale */
static LIS_INT host_begin(LIS_VECTOR v) { return VDEV(v)->host_valid ? LIS_SUCCESS : lisd_vec_to_host(v); }
static LIS_INT host_begin_write(LIS_VECTOR v) { return lisd_vec_host_write(v, 1); }
static void host_wrote(LIS_VECTOR v) { VDEV(v)->host_valid = 1; VDEV(v)->dev_valid = 0; }

static LIS_INT index_error(const char *what, LIS_INT i, LIS_VECTOR v)
{
	const LIS_INT o = v->origin ? 1 : 0;
	return LISI_ERR(LIS_ERR_ILL_ARG, "%s(=%D) is less than %D or not less than %D\n", what, i + o, v->is + o, v->ie + o);
}

LIS_INT lis_vector_set_value(LIS_INT flag, LIS_INT i, LIS_SCALAR value, LIS_VECTOR v)
{
	if (v->origin) i--;
	if (i < v->is || i >= v->ie) return index_error("i", i, v);
	LISCHK(host_begin_write(v));
	if (flag == LIS_INS_VALUE) v->value[i - v->is] = value; else v->value[i - v->is] += value;
	host_wrote(v);
	return LIS_SUCCESS;
}

LIS_INT lis_vector_set_values(LIS_INT flag, LIS_INT count, LIS_INT index[], LIS_SCALAR value[], LIS_VECTOR v)
{
	LISCHK(host_begin_write(v));
	for (LIS_INT k = 0; k < count; k++) {
		LIS_INT i = index[k] - (v->origin ? 1 : 0);
		if (i < v->is || i >= v->ie) return index_error("index[k]", i, v);
		if (flag == LIS_INS_VALUE) v->value[i - v->is] = value[k]; else v->value[i - v->is] += value[k];
	}
	host_wrote(v);
	return LIS_SUCCESS;
}

LIS_INT lis_vector_set_values2(LIS_INT flag, LIS_INT start, LIS_INT count, LIS_SCALAR value[], LIS_VECTOR v)
{
	LISCHK(host_begin_write(v));
	if (v->origin) start--;
	for (LIS_INT k = 0; k < count; k++) {
		LIS_INT i = start + k;
		if (i < v->is || i >= v->ie) return index_error("start+i", i, v);
		if (flag == LIS_INS_VALUE) v->value[i - v->is] = value[k]; else v->value[i - v->is] += value[k];
	}
	host_wrote(v);
	return LIS_SUCCESS;
}

LIS_INT lis_vector_get_value(LIS_VECTOR v, LIS_INT i, LIS_SCALAR *value)
{
	LISCHK(vec_check(v));
	if (v->origin) i--;
	if (i < v->is || i >= v->ie) return index_error("i", i, v);
	LISCHK(host_begin(v));
	*value = v->value[i - v->is];
	return LIS_SUCCESS;
}

LIS_INT lis_vector_get_values(LIS_VECTOR v, LIS_INT start, LIS_INT count, LIS_SCALAR value[])
{
	LISCHK(vec_check(v));
	if (v->origin) start--;
	if (start < v->is || start >= v->ie) return index_error("start", start, v);
	if (start - v->is + count > v->n)
		return LISI_ERR(LIS_ERR_ILL_ARG, "start(=%D) + count(=%D) exceeds the range of vector v(=%D).\n", start, count, v->ie);
	LISCHK(host_begin(v));
	memcpy(value, v->value + (start - v->is), sizeof(LIS_SCALAR) * (size_t)count);
	return LIS_SUCCESS;
}

LIS_INT lis_vector_scatter(LIS_SCALAR value[], LIS_VECTOR v)
{	/* ref lis_vector.c:943: every rank holds the full array and keeps its slice */
	LISCHK(vec_check(v));
	LISCHK(lisd_vec_host_write(v, v->np + v->pad > v->n));      /* (ghost / padding entries beyond n keep what they hold) */
	memcpy(v->value, value + v->is, sizeof(LIS_SCALAR) * (size_t)v->n);
	host_wrote(v);
	return LIS_SUCCESS;
}

LIS_INT lis_vector_gather(LIS_VECTOR v, LIS_SCALAR value[])
{	/* ref lis_vector.c:1011: the full vector on every rank */
	LISCHK(vec_check(v));
	LISCHK(host_begin(v));
	if (v->nprocs <= 1) { memcpy(value, v->value, sizeof(LIS_SCALAR) * (size_t)v->n); return LIS_SUCCESS; }
	/* ranks own unequal slices: gather fixed-size blocks of the largest slice, then compact */
	LIS_INT maxn = 0;
	for (LIS_INT r = 0; r < v->nprocs; r++) if (v->ranges[r + 1] - v->ranges[r] > maxn) maxn = v->ranges[r + 1] - v->ranges[r];
	double *send = (double *)calloc((size_t)maxn, sizeof(double)), *recv = (double *)malloc(sizeof(double) * (size_t)maxn * (size_t)v->nprocs);
	memcpy(send, v->value, sizeof(double) * (size_t)v->n);
	LIS_INT err = lisc_allgather_host(send, recv, sizeof(double) * (size_t)maxn);
	if (!err) for (LIS_INT r = 0; r < v->nprocs; r++)
		memcpy(value + v->ranges[r], recv + (size_t)r * maxn, sizeof(double) * (size_t)(v->ranges[r + 1] - v->ranges[r]));
	free(send); free(recv);
	return err;
}

LIS_INT lis_vector_print(LIS_VECTOR x)
{
	LISCHK(vec_check(x));
	LISCHK(host_begin(x));
	for (LIS_INT i = 0; i < x->n; i++) printf("%6d  %e\n", i + x->is + (x->origin ? 1 : 0), x->value[i]);
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ BLAS-1 on the device */
static LIS_INT same_len(LIS_VECTOR a, LIS_VECTOR b)
{
	if (a->n != b->n) return LISI_ERR(LIS_ERR_ILL_ARG, "length of vector x and y is not equal\n");
	return LIS_SUCCESS;
}

LIS_INT lis_vector_copy(LIS_VECTOR vsrc, LIS_VECTOR vdst)
{
	LISCHK(same_len(vsrc, vdst));
	double *s, *d;
	LISCHK(lisd_vec_in(vsrc, &s)); LISCHK(lisd_vec_out(vdst, &d));
	HIPCHK(liship_memcpy_d2d(d, s, sizeof(double) * (size_t)vsrc->n, lisg.stream));
	return lisd_vec_done(vdst);
}

LIS_INT lis_vector_swap(LIS_VECTOR vsrc, LIS_VECTOR vdst)
{
	LISCHK(same_len(vsrc, vdst));
	LISCHK(lisd_vec_host_write(vsrc, 1)); LISCHK(lisd_vec_host_write(vdst, 1));
	for (LIS_INT i = 0; i < vsrc->n; i++) { double t = vsrc->value[i]; vsrc->value[i] = vdst->value[i]; vdst->value[i] = t; }
	host_wrote(vsrc); host_wrote(vdst);
	return LIS_SUCCESS;
}

LIS_INT lis_vector_axpy(LIS_SCALAR alpha, LIS_VECTOR vx, LIS_VECTOR vy)
{
	LISCHK(same_len(vx, vy));
	double *x, *y;
	LISCHK(lisd_vec_in(vx, &x)); LISCHK(lisd_vec_in(vy, &y));
	HIPCHK(liship_axpy_f64(vx->n, alpha, x, y, lisg.stream));
	return lisd_vec_done(vy);
}

LIS_INT lis_vector_xpay(LIS_VECTOR vx, LIS_SCALAR alpha, LIS_VECTOR vy)
{
	LISCHK(same_len(vx, vy));
	double *x, *y;
	LISCHK(lisd_vec_in(vx, &x)); LISCHK(lisd_vec_in(vy, &y));
	HIPCHK(liship_xpay_f64(vx->n, x, alpha, y, lisg.stream));
	return lisd_vec_done(vy);
}

LIS_INT lis_vector_axpyz(LIS_SCALAR alpha, LIS_VECTOR vx, LIS_VECTOR vy, LIS_VECTOR vz)
{
	if (vx->n != vy->n || vx->n != vz->n) return LISI_ERR(LIS_ERR_ILL_ARG, "length of vector x and y and z is not equal\n");
	double *x, *y, *z;
	LISCHK(lisd_vec_in(vx, &x)); LISCHK(lisd_vec_in(vy, &y)); LISCHK(lisd_vec_out(vz, &z));
	HIPCHK(liship_axpyz_f64(vx->n, alpha, x, y, z, lisg.stream));
	return lisd_vec_done(vz);
}

LIS_INT lis_vector_scale(LIS_SCALAR alpha, LIS_VECTOR vx)
{
	double *x;
	LISCHK(lisd_vec_in(vx, &x));
	HIPCHK(liship_scale_f64(vx->n, alpha, x, lisg.stream));
	return lisd_vec_done(vx);
}

LIS_INT lis_vector_pmul(LIS_VECTOR vx, LIS_VECTOR vy, LIS_VECTOR vz)
{
	if (vx->n != vy->n || vx->n != vz->n) return LISI_ERR(LIS_ERR_ILL_ARG, "length of vector x and y and z is not equal\n");
	double *x, *y, *z;
	LISCHK(lisd_vec_in(vx, &x)); LISCHK(lisd_vec_in(vy, &y)); LISCHK(lisd_vec_out(vz, &z));
	HIPCHK(liship_pmul_f64(vx->n, x, y, z, lisg.stream));
	return lisd_vec_done(vz);
}

LIS_INT lis_vector_pdiv(LIS_VECTOR vx, LIS_VECTOR vy, LIS_VECTOR vz)
{
	if (vx->n != vy->n || vx->n != vz->n) return LISI_ERR(LIS_ERR_ILL_ARG, "length of vector x and y and z is not equal\n");
	double *x, *y, *z;
	LISCHK(lisd_vec_in(vx, &x)); LISCHK(lisd_vec_in(vy, &y)); LISCHK(lisd_vec_out(vz, &z));
	HIPCHK(liship_pdiv_f64(vx->n, x, y, z, lisg.stream));
	return lisd_vec_done(vz);
}

LIS_INT lis_vector_set_all(LIS_SCALAR alpha, LIS_VECTOR vx)
{
	double *x;
	LISCHK(lisd_vec_out(vx, &x));
	HIPCHK(liship_set_all_f64(vx->n, alpha, x, lisg.stream));
	return lisd_vec_done(vx);
}

#define UNARY(name, call)                                            \
LIS_INT name(LIS_VECTOR vx)                                          \
{                                                                    \
	double *x;                                                       \
	LISCHK(lisd_vec_in(vx, &x));                                     \
	HIPCHK(call);                                                    \
	return lisd_vec_done(vx);                                        \
}
UNARY(lis_vector_abs, liship_abs_f64(vx->n, x, lisg.stream))
UNARY(lis_vector_reciprocal, liship_reciprocal_f64(vx->n, x, lisg.stream))

LIS_INT lis_vector_conjugate(LIS_VECTOR vx) { (void)vx; return LIS_SUCCESS; }   /* real scalars: ref lis_vector_opv.c:490-492 */

LIS_INT lis_vector_shift(LIS_SCALAR sigma, LIS_VECTOR vx)
{
	double *x;
	LISCHK(lisd_vec_in(vx, &x));
	HIPCHK(liship_shift_f64(vx->n, sigma, x, lisg.stream));
	return lisd_vec_done(vx);
}

LIS_INT lis_vector_dot(LIS_VECTOR vx, LIS_VECTOR vy, LIS_SCALAR *value)
{
	LISCHK(same_len(vx, vy));
	double *x, *y;
	LISCHK(lisd_vec_in(vx, &x)); LISCHK(lisd_vec_in(vy, &y));
	return lisd_dot(vx->n, x, y, value);
}
LIS_INT lis_vector_nhdot(LIS_VECTOR vx, LIS_VECTOR vy, LIS_SCALAR *value) { return lis_vector_dot(vx, vy, value); }

LIS_INT lis_vector_nrm2(LIS_VECTOR vx, LIS_REAL *value)
{
	double *x;
	LISCHK(lisd_vec_in(vx, &x));
	return lisd_nrm2(vx->n, x, value);
}

LIS_INT lis_vector_nrm1(LIS_VECTOR vx, LIS_REAL *value)
{
	double *x;
	LISCHK(lisd_vec_in(vx, &x));
	return lisd_nrm1(vx->n, x, value);
}

LIS_INT lis_vector_nrmi(LIS_VECTOR vx, LIS_REAL *value)
{	/* max |x_i| (ref lis_vector_ops.c:344): off the Krylov path, taken on the host copy */
	LISCHK(lisd_vec_to_host(vx));
	double m = 0.0;
	for (LIS_INT i = 0; i < vx->n; i++) if (fabs(vx->value[i]) > m) m = fabs(vx->value[i]);
	if (vx->nprocs > 1) {
		double *all = (double *)malloc(sizeof(double) * (size_t)vx->nprocs);
		LIS_INT err = lisc_allgather_host(&m, all, sizeof(double));
		if (!err) for (LIS_INT r = 0; r < vx->nprocs; r++) if (all[r] > m) m = all[r];
		free(all);
		if (err) return err;
	}
	*value = m;
	return LIS_SUCCESS;
}

LIS_INT lis_vector_sum(LIS_VECTOR vx, LIS_SCALAR *value)
{
	double *x;
	LISCHK(lisd_vec_in(vx, &x));
	HIPCHK(liship_sum_f64(vx->n, x, lisg.reduce_out, lisg.reduce_work, lisg.stream));
	return lisd_fetch(1, value);
}
