/*
 * test_custom_algo.c -- an algorithm class registered by the caller (algo.c:106-120), scored through its own
 * overlap_probability / match_probability functions on the host path of libpandaseq_amd (pamd_compat.cpp, "the host path").
 * The class is a twin of simple_bayesian written from the published formulas (algo_simple_bayes.c:33-75) on the tables the
 * library hands out (pamd_tables_get), so every record it produces must equal, bit for bit, what the library's own
 * simple_bayesian -- the kernels -- produces for the same pair.
 * usage: test_custom_algo <pairs.bin> [primers]    prints "same <n> pairs, <ok> assembled" or the first difference
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pandaseq_amd.h"
#include "pandaseq_amd_compat.h"

static double g_tab[PAMD_TABLE_DOUBLES];
#define QUAL_NN (g_tab[0])
#define MATCH_SB(a, b) (g_tab[1 + 47 + 47 + (a) * 47 + (b)])
#define MISMATCH_SB(a, b) (g_tab[1 + 47 + 47 + 47 * 47 + (a) * 47 + (b)])
#define CLAMPQ(q) ((q) > 46 ? 46 : ((q) < 0 ? 0 : (q)))

struct twin {
	double pmatch, pmismatch;
};

static double twin_overlap(void *data, const panda_qual *forward, size_t forward_length, const panda_qual *reverse, size_t reverse_length, size_t overlap) {
	const struct twin *t = data;
	size_t unknown = 0, matches = 0, mismatches = 0;
	for (size_t i = 0; i < overlap; i++) {
		const long fi = (long) forward_length + (long) i - (long) overlap, ri = (long) reverse_length - (long) i - 1;
		if (fi < 0 || ri < 0 || (size_t) fi >= forward_length || (size_t) ri >= reverse_length)
			continue;
		const unsigned char f = (unsigned char) forward[fi].nt, r = (unsigned char) reverse[ri].nt;
		if (f == 15 || r == 15)
			unknown++;
		else if ((f & r) != 0)
			matches++;
		else
			mismatches++;
	}
	if (overlap >= forward_length && overlap >= reverse_length)
		return QUAL_NN * unknown + matches * t->pmatch + mismatches * t->pmismatch;
	return QUAL_NN * (forward_length + reverse_length - 2 * overlap + unknown) + matches * t->pmatch + mismatches * t->pmismatch;
}

static double twin_match(void *data, bool match, char a, char b) {
	(void) data;
	return match ? MATCH_SB(CLAMPQ(a), CLAMPQ(b)) : MISMATCH_SB(CLAMPQ(a), CLAMPQ(b));
}

static PandaAlgorithm twin_create(const char *args) {
	(void) args;
	return NULL;	/* (only built through panda_algorithm_new below) */
}

static struct panda_algorithm_class twin_class_storage;	/* prob_unpaired is const: filled through a byte copy in main */

static int same_record(const panda_result_seq *x, const panda_result_seq *y, size_t i) {
	if ((x == NULL) != (y == NULL)) {
		printf("pair %zu: one assembled, the other did not\n", i);
		return 0;
	}
	if (x == NULL)
		return 1;
	if (memcmp(&x->quality, &y->quality, 8) || memcmp(&x->estimated_overlap_probability, &y->estimated_overlap_probability, 8) || x->overlap != y->overlap || x->sequence_length != y->sequence_length || x->degenerates != y->degenerates ||
	    x->overlap_mismatches != y->overlap_mismatches || x->overlaps_examined != y->overlaps_examined || x->forward_offset != y->forward_offset || x->reverse_offset != y->reverse_offset) {
		printf("pair %zu: scalars differ (overlap %zu / %zu, length %zu / %zu, quality %.17g / %.17g)\n", i, x->overlap, y->overlap, x->sequence_length, y->sequence_length, x->quality, y->quality);
		return 0;
	}
	for (size_t j = 0; j < x->sequence_length; j++)
		if (x->sequence[j].nt != y->sequence[j].nt || memcmp(&x->sequence[j].p, &y->sequence[j].p, 8)) {
			printf("pair %zu: base %zu differs\n", i, j);
			return 0;
		}
	return 1;
}

/* what the pair-by-pair loop saw, for the array form of the call to be held against */
struct digest {
	int assembled;
	size_t overlap, length;
	double quality;
	uint64_t sum;
};
static struct digest *g_digest;
static size_t g_many_bad;

static uint64_t row_sum(const panda_result_seq *r) {
	uint64_t s = 0;
	for (size_t j = 0; j < r->sequence_length; j++) {
		uint64_t b;
		memcpy(&b, &r->sequence[j].p, 8);
		s = s * 1099511628211ull + (uint64_t) (unsigned char) r->sequence[j].nt + b;
	}
	return s;
}

static void many_visit(size_t index, const panda_result_seq *r, void *user_data) {
	(void) user_data;
	const struct digest *d = &g_digest[index];
	if ((r != NULL) != d->assembled || (r != NULL && (r->overlap != d->overlap || r->sequence_length != d->length || memcmp(&r->quality, &d->quality, 8) || row_sum(r) != d->sum)))
		g_many_bad++;
}

int main(int argc, char **argv) {
	if (argc < 2)
		return 2;
	FILE *fp = fopen(argv[1], "rb");
	if (fp == NULL)
		return 2;
	uint32_t hdr[2];
	if (fread(hdr, 4, 2, fp) != 2)
		return 2;
	const uint32_t n = hdr[0], stride = hdr[1];
	const size_t rec = 4 + (size_t) stride * 4;
	unsigned char *blob = malloc((size_t) n * rec);
	if (blob == NULL || fread(blob, rec, n, fp) != n)
		return 2;
	fclose(fp);
	if (pamd_tables_get(g_tab, PAMD_TABLE_DOUBLES) != 0)
		return 3;

	struct panda_algorithm_class proto = { sizeof(struct twin), "sb_twin", twin_create, NULL, twin_overlap, twin_match, QUAL_NN };
	memcpy(&twin_class_storage, &proto, sizeof proto);
	panda_algorithm_register(&twin_class_storage);
	PandaAlgorithm twin = panda_algorithm_new(&twin_class_storage);
	if (twin == NULL || !panda_algorithm_is_a(twin, &twin_class_storage))
		return 4;
	struct twin *td = panda_algorithm_data(twin);
	const double q = 0.36;	/* algo_simple_bayes.c:121-135: the default error estimation */
	td->pmatch = log(0.25 * (1 - 2 * q + q * q));
	td->pmismatch = log((3 * q - 2 * q * q) / 18.0);

	/* a class with no functions of its own and no device form is refused by the setter; the assembler keeps what it had */
	static struct panda_algorithm_class hollow = { 0, "hollow", twin_create, NULL, NULL, NULL, -1.0 };
	PandaAlgorithm nothing = panda_algorithm_new(&hollow);

	PandaAssembler dev = panda_assembler_new(NULL, NULL, NULL, NULL), host = panda_assembler_new(NULL, NULL, NULL, NULL);
	if (dev == NULL || host == NULL)
		return 5;
	if (nothing != NULL) {
		panda_assembler_set_algorithm(dev, nothing);
		if (!panda_algorithm_is_a(panda_assembler_get_algorithm(dev), &panda_algorithm_simple_bayes_class)) {
			printf("a class without functions replaced the assembler's algorithm\n");
			return 1;
		}
	}
	panda_assembler_set_algorithm(host, twin);
	if (!panda_algorithm_is_a(panda_assembler_get_algorithm(host), &twin_class_storage)) {
		printf("the caller's class was refused\n");
		return 1;
	}
	if (argc > 2) {	/* `-p` / `-q`: config 2's primers, the reverse one complemented as the command line does */
		panda_nt fpr[32], rpr[32];
		const char *fs = "CCTACGGGAGGCAGCAG", *rs = "ATTACCGCGGCTGCTGG";
		for (size_t k = 0; k < strlen(fs); k++)	/* A 1, C 2, G 4, T 8 (pandaseq-nt.h:39-51) */
			fpr[k] = (panda_nt) (fs[k] == 'A' ? 1 : fs[k] == 'C' ? 2 : fs[k] == 'G' ? 4 : 8);
		for (size_t k = 0; k < strlen(rs); k++)
			rpr[k] = (panda_nt) (rs[k] == 'A' ? 8 : rs[k] == 'C' ? 4 : rs[k] == 'G' ? 2 : 1);
		panda_assembler_set_forward_primer(dev, fpr, strlen(fs));
		panda_assembler_set_forward_primer(host, fpr, strlen(fs));
		panda_assembler_set_reverse_primer(dev, rpr, strlen(rs));
		panda_assembler_set_reverse_primer(host, rpr, strlen(rs));
	}
	size_t ok = 0;
	g_digest = calloc(n, sizeof *g_digest);
	panda_seq_identifier *ids = calloc(n, sizeof *ids);
	const panda_qual **fs = calloc(n, sizeof *fs), **rs = calloc(n, sizeof *rs);
	size_t *fls = calloc(n, sizeof *fls), *rls = calloc(n, sizeof *rls);
	for (uint32_t i = 0; i < n; i++) {
		const unsigned char *r = blob + (size_t) i * rec;
		uint16_t fl, rl;
		memcpy(&fl, r, 2);
		memcpy(&rl, r + 2, 2);
		panda_seq_identifier id;
		memset(&id, 0, sizeof id);
		id.x = (int) i;
		const panda_qual *f = (const panda_qual *) (r + 4), *rv = (const panda_qual *) (r + 4 + (size_t) stride * 2);
		ids[i] = id;
		fs[i] = f;
		rs[i] = rv;
		fls[i] = fl;
		rls[i] = rl;
		const panda_result_seq *x = panda_assembler_assemble(dev, &id, f, fl, rv, rl);
		if (x != NULL) {
			g_digest[i].assembled = 1;
			g_digest[i].overlap = x->overlap;
			g_digest[i].length = x->sequence_length;
			g_digest[i].quality = x->quality;
			g_digest[i].sum = row_sum(x);
		}
		/* (the record of `dev` is valid until its next call: compare before it is reused) */
		const panda_result_seq *y = panda_assembler_assemble(host, &id, f, fl, rv, rl);
		if (!same_record(x, y, i))
			return 1;
		ok += x != NULL;
	}
	if (panda_assembler_get_count(dev) != panda_assembler_get_count(host) || panda_assembler_get_ok_count(dev) != panda_assembler_get_ok_count(host) ||
	    panda_assembler_get_failed_alignment_count(dev) != panda_assembler_get_failed_alignment_count(host) || panda_assembler_get_low_quality_count(dev) != panda_assembler_get_low_quality_count(host) ||
	    panda_assembler_get_bad_read_count(dev) != panda_assembler_get_bad_read_count(host) || panda_assembler_get_slow_count(dev) != panda_assembler_get_slow_count(host) ||
	    panda_assembler_get_no_forward_primer_count(dev) != panda_assembler_get_no_forward_primer_count(host) || panda_assembler_get_no_reverse_primer_count(dev) != panda_assembler_get_no_reverse_primer_count(host)) {
		printf("counters differ\n");
		return 1;
	}
	/* the array form of the call (pamd_compat_assemble_many): the same records and counters as pair by pair */
	PandaAssembler many = panda_assembler_new(NULL, NULL, NULL, NULL);
	if (many == NULL)
		return 5;
	panda_assembler_copy_configuration(many, dev);
	const size_t got = pamd_compat_assemble_many(many, ids, fs, fls, rs, rls, n, many_visit, NULL);
	if (got != ok || g_many_bad != 0 || panda_assembler_get_count(many) != panda_assembler_get_count(dev) || panda_assembler_get_ok_count(many) != panda_assembler_get_ok_count(dev) ||
	    panda_assembler_get_failed_alignment_count(many) != panda_assembler_get_failed_alignment_count(dev) || panda_assembler_get_no_forward_primer_count(many) != panda_assembler_get_no_forward_primer_count(dev)) {
		printf("pamd_compat_assemble_many: %zu assembled against %zu, %zu records differ\n", got, ok, g_many_bad);
		return 1;
	}
	panda_assembler_unref(many);
	printf("same %u pairs, %zu assembled\n", n, ok);
	panda_assembler_unref(dev);
	panda_assembler_unref(host);
	panda_algorithm_unref(twin);
	return 0;
}
