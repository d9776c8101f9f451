/* gen_queen_class.c -- a deterministic stand-in for SuiteSparse Queen_4147 (BASELINE config 4), written as a SYMMETRIC
 * coordinate Matrix Market file so that it reaches the library the way the real file would: through lis_input
 * (reference reader: src/system/lis_input_mm.c:699-1069, symmetric expansion :986-1036).
 *
 * Test infrastructure only (compiled with gcc by tests/queen_class.py on whichever box runs the test); not part of the product.
 *
 *   gen_queen_class G BAND out.mtx
 *
 * The matrix: 3 unknowns per node of a G x G x G grid, 27-node connectivity, about one node pair in eight uncoupled
 * (rows of 3 .. 81 entries, ~71 on average), symmetric, strictly diagonally dominant.  G = 111 gives 4 102 893 rows and ~2.9e8
 * non-zeros -- Queen_4147 has 4 147 110 rows and 3.2e8.  The NODE NUMBERING is scrambled: inside every run of BAND nodes of the
 * natural (lexicographic) order the numbers are permuted at random, as a mesh generator's numbering is local but not regular --
 * (column - row) takes thousands of values, no two rows share an offset pattern, neighbouring rows belong to nodes that are not
 * neighbours in space.  The file lists, row by row, the entries with column <= row in the order the node's neighbours are
 * visited -- NOT sorted by column -- so the in-row order of the expanded matrix is whatever the reader's expansion makes of it.
 * Values: off-diagonal -k/16 (k = 1..16 from a hash of the entry), diagonal = sum |off-diagonal| of the whole row + 1 + (hash % 8),
 * all exact in binary and printed exactly.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t mix64(uint64_t z)
{
	z += 0x9e3779b97f4a7c15ULL;
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}

static char *put_uint(char *p, uint32_t v)
{
	char tmp[12];
	int k = 0;
	do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
	while (k) *p++ = tmp[--k];
	return p;
}

static const char *FRAC16[16] = {"", ".0625", ".125", ".1875", ".25", ".3125", ".375", ".4375", ".5", ".5625", ".625", ".6875", ".75", ".8125", ".875", ".9375"};

/* v in sixteenths, printed exactly */
static char *put_sixteenths(char *p, int neg, uint32_t s16)
{
	if (neg) *p++ = '-';
	p = put_uint(p, s16 >> 4);
	const char *f = FRAC16[s16 & 15];
	while (*f) *p++ = *f++;
	return p;
}

int main(int argc, char **argv)
{
	if (argc != 4) { fprintf(stderr, "usage: %s G BAND out.mtx\n", argv[0]); return 2; }
	const int G = atoi(argv[1]);
	const int band = atoi(argv[2]);
	if (G < 2 || G > 800 || band < 1) return 2;
	const int64_t nodes = (int64_t)G * G * G;
	if (3 * nodes >= 0x7fffffffLL) return 2;
	uint32_t *perm = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nodes);    /* natural index -> new number */
	uint32_t *inv = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nodes);     /* new number -> natural index */
	if (!perm || !inv) return 3;
	for (int64_t lo = 0; lo < nodes; lo += band) {                             /* Fisher-Yates inside each band */
		const int64_t hi = lo + band < nodes ? lo + band : nodes;
		for (int64_t i = lo; i < hi; i++) perm[i] = (uint32_t)i;
		uint64_t s = mix64(0x5eedULL + (uint64_t)lo);
		for (int64_t i = hi - 1; i > lo; i--) {
			s = mix64(s);
			const int64_t j = lo + (int64_t)(s % (uint64_t)(i - lo + 1));
			const uint32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
		}
	}
	for (int64_t p = 0; p < nodes; p++) inv[perm[p]] = (uint32_t)p;

	FILE *f = fopen(argv[3], "wb");
	if (!f) return 4;
	const size_t cap = (size_t)32 << 20;
	char *buf = (char *)malloc(cap + 8192);
	if (!buf) return 3;
	char *w = buf;
	int64_t stored = 0;
	for (int pass = 0; pass < 2; pass++) {
		if (pass == 1) {
			w += sprintf(w, "%%%%MatrixMarket matrix coordinate real symmetric\n%% queen-class stand-in: G=%d band=%d (tests/golden/gen_queen_class.c)\n%lld %lld %lld\n",
			             G, band, (long long)(3 * nodes), (long long)(3 * nodes), (long long)stored);
		}
		for (int64_t q = 0; q < nodes; q++) {
			const int64_t p = inv[q];
			const int a = (int)(p / ((int64_t)G * G)), b = (int)(p / G % G), c = (int)(p % G);
			uint32_t nb[27];
			int nnb = 0;
			for (int da = -1; da <= 1; da++) for (int db = -1; db <= 1; db++) for (int dc = -1; dc <= 1; dc++) {
				const int aa = a + da, bb = b + db, cc = c + dc;
				if (aa < 0 || aa >= G || bb < 0 || bb >= G || cc < 0 || cc >= G) continue;
				const uint32_t q2 = perm[((int64_t)aa * G + bb) * G + cc];
				if (q2 != (uint32_t)q) {                                    /* one node pair in eight is not coupled */
					const uint64_t lo = q2 < (uint32_t)q ? q2 : (uint32_t)q, hi = q2 < (uint32_t)q ? (uint32_t)q : q2;
					if ((mix64((hi << 32) | lo) & 7) == 0) continue;
				}
				nb[nnb++] = q2;
			}
			for (int d = 0; d < 3; d++) {
				const uint32_t row = 3 * (uint32_t)q + (uint32_t)d;
				if (pass == 0) {
					for (int k = 0; k < nnb; k++) for (int e = 0; e < 3; e++) stored += (3 * nb[k] + (uint32_t)e <= row);
					continue;
				}
				uint32_t sum16 = 0;
				for (int k = 0; k < nnb; k++) for (int e = 0; e < 3; e++) {
					const uint32_t col = 3 * nb[k] + (uint32_t)e;
					if (col == row) continue;
					const uint64_t lo = col < row ? col : row, hi = col < row ? row : col;
					sum16 += 1 + (uint32_t)(mix64((hi << 32) | lo | 0x8000000000000000ULL) & 15);
				}
				for (int k = 0; k < nnb; k++) for (int e = 0; e < 3; e++) {
					const uint32_t col = 3 * nb[k] + (uint32_t)e;
					if (col > row) continue;
					w = put_uint(w, row + 1); *w++ = ' ';
					w = put_uint(w, col + 1); *w++ = ' ';
					if (col == row) w = put_sixteenths(w, 0, sum16 + 16 * (1 + (uint32_t)(mix64(row) & 7)));
					else w = put_sixteenths(w, 1, 1 + (uint32_t)(mix64(((uint64_t)row << 32) | col | 0x8000000000000000ULL) & 15));
					*w++ = '\n';
				}
				if ((size_t)(w - buf) > cap) { if (fwrite(buf, 1, (size_t)(w - buf), f) != (size_t)(w - buf)) return 5; w = buf; }
			}
		}
	}
	if (w > buf && fwrite(buf, 1, (size_t)(w - buf), f) != (size_t)(w - buf)) return 5;
	if (fclose(f)) return 5;
	printf("%lld %lld\n", (long long)(3 * nodes), (long long)stored);
	return 0;
}
