/*
 * test_per_pair.c -- the per-pair entry point of the reference's API, panda_assembler_assemble() (pandaseq-assembler.h:137-163,
 * assembler.c:368-383: what panda_diff calls, diff.c:26-99), in plain C.
 *
 * usage: test_per_pair <pairs.bin> <out.bin> <algo> <path> [option value]...
 *   pairs.bin: u32 n, u32 stride, then n * (u16 flen, u16 rlen, stride*2 bytes fwd, stride*2 bytes rev)
 *   path:  host   panda_assembler_assemble() on an assembler that has NO device (pamd_compat_host_only_assembler_new: the
 *                 library's per-pair host path and nothing else; runs without a GPU)
 *          pair   panda_assembler_assemble() on an ordinary assembler (the per-pair path the library ships: host, or the
 *                 device with PAMD_COMPAT_PER_PAIR=device in the environment)
 *          many   pamd_compat_assemble_many() over all pairs at once (the device)
 *   options: -p <forward primer> -q <reverse primer> (one hex digit per base: the panda_nt code) -a (primers after assembly)
 *            -t <threshold> -o <minimum overlap> -O <maximum overlap> -T <primer penalty> -m <module spec, e.g. min_phred:4>
 *            -k <num_kmers> (first option if given: the assembler is made by panda_assembler_new_kmer, assembler_support.c:36-100)
 *            -r <passes> (host / pair: that many untimed-output passes over the pairs first, every base read; prints pairs/s to stderr)
 *   out.bin: per pair: u32 returned?, then (if returned) u32 index, overlap, seq_len, mism, degen, fo, ro, examined, f64 quality,
 *            f64 est, seq_len nt bytes, seq_len f64 -- test_compat.c's record; then 10 i64: count, ok, noalgn, lowq, badr, slow,
 *            nofp, norp, longest overlap, rejected by the first module
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "pandaseq_amd_compat.h"

PandaAssembler pamd_compat_host_only_assembler_new(void);	/* test hook of libpandaseq_amd (pamd_compat.cpp) */
PandaAssembler pamd_compat_host_only_assembler_new_kmer(size_t num_kmers);

static FILE *out;

static void dump(uint32_t index, const panda_result_seq *res) {
	uint32_t got = res != NULL;
	fwrite(&got, 4, 1, out);
	if (!got)
		return;
	uint32_t h[8] = { index, (uint32_t) res->overlap, (uint32_t) res->sequence_length, (uint32_t) res->overlap_mismatches, (uint32_t) res->degenerates, (uint32_t) res->forward_offset, (uint32_t) res->reverse_offset, (uint32_t) res->overlaps_examined };
	fwrite(h, 4, 8, out);
	fwrite(&res->quality, 8, 1, out);
	fwrite(&res->estimated_overlap_probability, 8, 1, out);
	for (size_t j = 0; j < res->sequence_length; j++)
		fwrite(&res->sequence[j].nt, 1, 1, out);
	for (size_t j = 0; j < res->sequence_length; j++)
		fwrite(&res->sequence[j].p, 8, 1, out);
}

static void visit(size_t index, const panda_result_seq *res, void *data) {
	(void) data;
	dump((uint32_t) index, res);
}

static bool first_rejected(PandaAssembler a, PandaModule m, size_t rejected, void *data) {
	(void) a; (void) m;
	*(int64_t *) data = (int64_t) rejected;
	return false;
}

static size_t parse_primer(const char *hex, panda_nt *dst) {
	size_t n = 0;
	for (; hex[n] != '\0'; n++) {
		const char c = hex[n];
		dst[n] = (panda_nt) (c >= 'a' ? c - 'a' + 10 : (c >= 'A' ? c - 'A' + 10 : c - '0'));
	}
	return n;
}

int main(int argc, char **argv) {
	if (argc < 5)
		return 2;
	FILE *in = fopen(argv[1], "rb");
	uint32_t n, stride;
	if (!in || fread(&n, 4, 1, in) != 1 || fread(&stride, 4, 1, in) != 1)
		return 3;
	const size_t rec = 4 + (size_t) stride * 4;
	unsigned char *blob = malloc((size_t) n * rec);
	if (fread(blob, rec, n, in) != n)
		return 3;
	fclose(in);
	const char *path = argv[4];
	size_t num_kmers = PANDA_DEFAULT_NUM_KMERS;
	int first_option = 5;
	if (argc >= 7 && strcmp(argv[5], "-k") == 0) {
		num_kmers = (size_t) atol(argv[6]);
		first_option = 7;
	}
	PandaAssembler a = strcmp(path, "host") == 0 ? pamd_compat_host_only_assembler_new_kmer(num_kmers) : panda_assembler_new_kmer(NULL, NULL, NULL, NULL, num_kmers);
	if (a == NULL)
		return 4;
	if (panda_assembler_get_num_kmer(a) != num_kmers)
		return 6;
	PandaAlgorithm algo = NULL;
	if (strcmp(argv[3], "pear") == 0)
		algo = panda_algorithm_pear_new();
	else if (strcmp(argv[3], "rdp_mle") == 0)
		algo = panda_algorithm_rdp_mle_new();
	else if (strcmp(argv[3], "uparse") == 0)
		algo = panda_algorithm_uparse_new();
	else if (strcmp(argv[3], "flash") == 0)
		algo = panda_algorithm_flash_new();
	else if (strcmp(argv[3], "ea_util") == 0)
		algo = panda_algorithm_ea_util_new();
	else if (strcmp(argv[3], "stitch") == 0)
		algo = panda_algorithm_stitch_new();
	else
		algo = panda_algorithm_simple_bayes_new();
	panda_assembler_set_algorithm(a, algo);
	panda_algorithm_unref(algo);
	int passes = 0;
	for (int i = first_option; i < argc; i++) {
		const char *o = argv[i];
		if (strcmp(o, "-a") == 0) {
			panda_assembler_set_primers_after(a, true);
			continue;
		}
		if (i + 1 >= argc)
			return 2;
		const char *v = argv[++i];
		panda_nt primer[512];
		if (strcmp(o, "-p") == 0)
			panda_assembler_set_forward_primer(a, primer, parse_primer(v, primer));
		else if (strcmp(o, "-q") == 0)
			panda_assembler_set_reverse_primer(a, primer, parse_primer(v, primer));
		else if (strcmp(o, "-t") == 0)
			panda_assembler_set_threshold(a, atof(v));
		else if (strcmp(o, "-o") == 0)
			panda_assembler_set_minimum_overlap(a, atoi(v));
		else if (strcmp(o, "-O") == 0)
			panda_assembler_set_maximum_overlap(a, atoi(v));
		else if (strcmp(o, "-r") == 0)
			passes = atoi(v);
		else if (strcmp(o, "-T") == 0)
			panda_assembler_set_primer_penalty(a, atof(v));
		else if (strcmp(o, "-m") == 0) {
			PandaModule m = panda_module_load(NULL, v);
			if (m == NULL)
				return 5;
			panda_assembler_add_module(a, m);
			panda_module_unref(m);
		} else
			return 2;
	}
	out = fopen(argv[2], "wb");
	if (strcmp(path, "many") == 0) {
		panda_seq_identifier *ids = calloc(n, sizeof *ids);
		const panda_qual **f = malloc(n * sizeof *f), **r = malloc(n * sizeof *r);
		size_t *fl = malloc(n * sizeof *fl), *rl = malloc(n * sizeof *rl);
		for (uint32_t i = 0; i < n; i++) {
			unsigned char *p = blob + (size_t) i * rec;
			uint16_t a16, b16;
			memcpy(&a16, p, 2);
			memcpy(&b16, p + 2, 2);
			fl[i] = a16;
			rl[i] = b16;
			f[i] = (const panda_qual *) (p + 4);
			r[i] = (const panda_qual *) (p + 4 + (size_t) stride * 2);
			ids[i].x = (int) i;
		}
		pamd_compat_assemble_many(a, ids, f, fl, r, rl, n, visit, NULL);
	} else {
		if (passes > 0) {	/* the rate of the per-pair path, every merged base read (no output) */
			struct timespec t0, t1;
			double sum = 0;
			clock_gettime(CLOCK_MONOTONIC, &t0);
			for (int pass = 0; pass < passes; pass++)
				for (uint32_t i = 0; i < n; i++) {
					unsigned char *p = blob + (size_t) i * rec;
					uint16_t fl, rl;
					memcpy(&fl, p, 2);
					memcpy(&rl, p + 2, 2);
					panda_seq_identifier id;
					memset(&id, 0, sizeof id);
					id.x = (int) i;
					const panda_result_seq *res = panda_assembler_assemble(a, &id, (const panda_qual *) (p + 4), fl, (const panda_qual *) (p + 4 + (size_t) stride * 2), rl);
					if (res != NULL)
						for (size_t j = 0; j < res->sequence_length; j++)
							sum += res->sequence[j].p + res->sequence[j].nt;
				}
			clock_gettime(CLOCK_MONOTONIC, &t1);
			const double sec = (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
			fprintf(stderr, "%.0f pairs/s (%d passes over %u pairs, checksum %g)\n", (double) passes * n / sec, passes, n, sum);
			/* (the counters below then count passes + 1 passes) */
		}
		for (uint32_t i = 0; i < n; i++) {
			unsigned char *p = blob + (size_t) i * rec;
			uint16_t fl, rl;
			memcpy(&fl, p, 2);
			memcpy(&rl, p + 2, 2);
			panda_seq_identifier id;
			memset(&id, 0, sizeof id);
			id.x = (int) i;
			dump(i, panda_assembler_assemble(a, &id, (const panda_qual *) (p + 4), fl, (const panda_qual *) (p + 4 + (size_t) stride * 2), rl));
		}
	}
	int64_t c[10] = { panda_assembler_get_count(a), panda_assembler_get_ok_count(a), panda_assembler_get_failed_alignment_count(a), panda_assembler_get_low_quality_count(a), panda_assembler_get_bad_read_count(a), panda_assembler_get_slow_count(a), panda_assembler_get_no_forward_primer_count(a), panda_assembler_get_no_reverse_primer_count(a), (int64_t) panda_assembler_get_longest_overlap(a), 0 };
	panda_assembler_foreach_module(a, first_rejected, &c[9]);
	fwrite(c, 8, 10, out);
	fclose(out);
	panda_assembler_unref(a);
	free(blob);
	return 0;
}
