/*
 * test_compat.c -- a plain-C client of the reference's per-assembler API as re-implemented by
 * libpandaseq_amd.so (include/pandaseq_amd_compat.h).  It is written the way a PANDAseq embedder
 * writes its loop (pool.c:76: while ((result = panda_assembler_next(a)) != NULL) output(result)).
 *
 * usage: test_compat <pairs.bin> <out.bin> <algo> <mode>
 *   pairs.bin: u32 n, u32 stride, then n * (u16 flen, u16 rlen, stride*2 bytes fwd, stride*2 bytes rev)
 *   mode 0: plain;  mode 1: fused min_overlapbits:15 + host check module + precheck module
 *   mode 2: <plugin.so> <module name> <args, or "(null)" for no ":args" at all>: a check plugin BINARY compiled from the reference's own
 *           plugin source (oracle/Makefile `ref`), opened the way module.c:227-298 does (dlopen, its
 *           <name>_LTX_opener) and attached as a host callback with panda_module_new
 *   mode 3: <name:args>: the same filter fused on the device (panda_module_load)
 *   mode 4: a host check module FIRST, then min_phred:4 and min_overlapbits:15 (which therefore run on the host too:
 *           the kernel only takes a leading run of them, module.c:128-134 blames the first module that fails)
 *   counters[8] = sequences rejected by the first module (panda_assembler_foreach_module)
 *
 * With -DUSE_REFERENCE_HEADERS the client is compiled against the reference's own pandaseq.h instead of
 * include/pandaseq_amd_compat.h: the structs this program reads are then laid out by the reference.
 *   out.bin: per returned result: u32 index, u32 overlap, u32 seq_len, u32 mism, u32 degen, u32 fo, u32 ro, u32 examined,
 *            f64 quality, f64 est, seq_len nt bytes, seq_len f64;  then u32 0xFFFFFFFF and 10 i64 counters
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef USE_REFERENCE_HEADERS
#include <pandaseq.h>
void pamd_compat_set_batch_pairs(size_t pairs);
#else
#include "pandaseq_amd_compat.h"
#endif
#include <dlfcn.h>
#include <stdarg.h>

/* Plugins report bad arguments through the log proxy, which is not on the path and not part of
 * libpandaseq_amd (in a real integration the reference's proxy.c provides these).  The plugin binaries
 * resolve them from this executable (-rdynamic); they are only reached on malformed arguments. */
void panda_log_proxy_write_str(PandaLogProxy proxy, const char *str) {
	(void) proxy;
	fprintf(stderr, "plugin: %s", str);
}

void panda_log_proxy_write_f(PandaLogProxy proxy, const char *format, ...) {
	va_list va;
	(void) proxy;
	va_start(va, format);
	vfprintf(stderr, format, va);
	va_end(va);
}

typedef bool (*plugin_opener) (PandaLogProxy logger, const char *args, PandaPreCheck *precheck, PandaCheck *check, void **user_data, PandaDestroy *destroy);

static bool first_rejected(PandaAssembler a, PandaModule m, size_t rejected, void *data) {
	(void) a; (void) m;
	*(int64_t *) data = (int64_t) rejected;
	return false;
}

struct source {
	uint32_t n, stride, pos;
	unsigned char *blob;
	size_t rec;
};

static bool next_pair(panda_seq_identifier *id, const panda_qual **forward, size_t *forward_length, const panda_qual **reverse, size_t *reverse_length, void *user_data) {
	struct source *s = user_data;
	if (s->pos >= s->n)
		return false;
	unsigned char *rec = s->blob + (size_t) s->pos * s->rec;
	uint16_t fl, rl;
	memcpy(&fl, rec, 2);
	memcpy(&rl, rec + 2, 2);
	memset(id, 0, sizeof *id);
	strcpy(id->instrument, "SIM");
	id->x = (int) s->pos;
	*forward = (const panda_qual *) (rec + 4);
	*forward_length = fl;
	*reverse = (const panda_qual *) (rec + 4 + (size_t) s->stride * 2);
	*reverse_length = rl;
	s->pos++;
	return true;
}

static long noalgn_seen = 0;
static void on_noalgn(PandaAssembler a, const panda_seq_identifier *id, const panda_qual *f, size_t fl, const panda_qual *r, size_t rl, void *data) {
	(void) a; (void) id; (void) f; (void) fl; (void) r; (void) rl; (void) data;
	noalgn_seen++;
}

static bool check_few_mismatches(PandaLogProxy logger, const panda_result_seq *seq, void *data) {
	(void) logger; (void) data;
	return seq->overlap_mismatches <= 3;
}

static bool precheck_not_multiple_of_11(PandaLogProxy logger, const panda_seq_identifier *id, const panda_qual *f, size_t fl, const panda_qual *r, size_t rl, void *data) {
	(void) logger; (void) f; (void) fl; (void) r; (void) rl; (void) data;
	return id->x % 11 != 0;
}

static void dump(FILE *out, const panda_result_seq *res) {
	uint32_t h[8] = { (uint32_t) res->name.x, (uint32_t) res->overlap, (uint32_t) res->sequence_length, (uint32_t) res->overlap_mismatches, (uint32_t) res->degenerates, (uint32_t) res->forward_offset, (uint32_t) res->reverse_offset, (uint32_t) res->overlaps_examined };
	fwrite(h, 4, 8, out);
	fwrite(&res->quality, 8, 1, out);
	fwrite(&res->estimated_overlap_probability, 8, 1, out);
	for (size_t j = 0; j < res->sequence_length; j++)
		fwrite(&res->sequence[j].nt, 1, 1, out);
	for (size_t j = 0; j < res->sequence_length; j++)
		fwrite(&res->sequence[j].p, 8, 1, out);
}

int main(int argc, char **argv) {
	if (argc < 5)
		return 2;
	FILE *in = fopen(argv[1], "rb");
	struct source src;
	if (!in || fread(&src.n, 4, 1, in) != 1 || fread(&src.stride, 4, 1, in) != 1)
		return 3;
	src.pos = 0;
	src.rec = 4 + (size_t) src.stride * 4;
	src.blob = malloc((size_t) src.n * src.rec);
	if (fread(src.blob, src.rec, src.n, in) != src.n)
		return 3;
	fclose(in);
	const int mode = atoi(argv[4]);

	pamd_compat_set_batch_pairs(1000);	/* several batches, the last one partial */
	PandaAssembler a = panda_assembler_new(next_pair, &src, NULL, NULL);
	if (a == NULL)
		return 4;
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
	panda_assembler_set_fail_alignment(a, on_noalgn, NULL, NULL);
	panda_assembler_set_name(a, "gpu0");
	if (mode == 1) {
		PandaModule m1 = panda_module_load(NULL, "min_overlapbits:15");
		PandaModule m2 = panda_module_new("few_mismatches", check_few_mismatches, NULL, NULL, NULL);
		PandaModule m3 = panda_module_new("not11", NULL, precheck_not_multiple_of_11, NULL, NULL);
		if (!m1 || !m2 || !m3)
			return 5;
		panda_assembler_add_module(a, m1);
		panda_assembler_add_module(a, m2);
		panda_assembler_add_module(a, m3);
		panda_module_unref(m1);
		panda_module_unref(m2);
		panda_module_unref(m3);
	}
	if (mode == 4) {
		PandaModule m1 = panda_module_new("few_mismatches", check_few_mismatches, NULL, NULL, NULL);
		PandaModule m2 = panda_module_load(NULL, "min_phred:4");
		PandaModule m3 = panda_module_load(NULL, "min_overlapbits:15");
		if (!m1 || !m2 || !m3)
			return 5;
		panda_assembler_add_module(a, m1);
		panda_assembler_add_module(a, m2);
		panda_assembler_add_module(a, m3);
		panda_module_unref(m1);
		panda_module_unref(m2);
		panda_module_unref(m3);
	}
	if (mode == 2) {
		if (argc < 8)
			return 2;
		void *h = dlopen(argv[5], RTLD_NOW | RTLD_LOCAL);
		if (h == NULL) {
			fprintf(stderr, "%s\n", dlerror());
			return 7;
		}
		char sym[128];
		snprintf(sym, sizeof sym, "%s_LTX_opener", argv[6]);
		plugin_opener opener = (plugin_opener) dlsym(h, sym);
		snprintf(sym, sizeof sym, "%s_LTX_api", argv[6]);
		int *api = dlsym(h, sym);
		PandaCheck check = NULL;
		PandaPreCheck precheck = NULL;
		void *user_data = NULL;
		PandaDestroy destroy = NULL;
		if (opener == NULL || api == NULL || *api != 3 /* PANDA_API, pandaseq-common.h */ || !opener(NULL, strcmp(argv[7], "(null)") == 0 ? NULL : argv[7], &precheck, &check, &user_data, &destroy))
			return 7;
		PandaModule m = panda_module_new(argv[6], check, precheck, user_data, destroy);
		if (m == NULL)
			return 5;
		panda_assembler_add_module(a, m);
		panda_module_unref(m);
	} else if (mode == 3) {
		if (argc < 6)
			return 2;
		PandaModule m = panda_module_load(NULL, argv[5]);
		if (m == NULL)
			return 5;
		panda_assembler_add_module(a, m);
		panda_module_unref(m);
	}
	/* getters mirror what was set, with the reference's conversions */
	if (panda_assembler_get_minimum_overlap(a) != 2 || panda_assembler_get_maximum_overlap(a) != 0 || panda_assembler_get_num_kmer(a) != 2)
		return 6;
	if (strcmp(panda_assembler_get_name(a), "gpu0") != 0)
		return 6;

	FILE *out = fopen(argv[2], "wb");
	const panda_result_seq *res;
	while ((res = panda_assembler_next(a)) != NULL)
		dump(out, res);
	uint32_t end = 0xFFFFFFFFu;
	fwrite(&end, 4, 1, out);
	int64_t c[10] = { panda_assembler_get_count(a), panda_assembler_get_ok_count(a), panda_assembler_get_failed_alignment_count(a), panda_assembler_get_low_quality_count(a), panda_assembler_get_bad_read_count(a), panda_assembler_get_slow_count(a), (int64_t) panda_assembler_get_longest_overlap(a), noalgn_seen, 0, 0 };
	if (mode >= 2)
		panda_assembler_foreach_module(a, first_rejected, &c[8]);
	fwrite(c, 8, 10, out);

	/* the single-pair entry point on the first pairs of the same input, through a second assembler
	 * that inherits the configuration (what pool.c:152 does per worker thread) */
	PandaAssembler b = panda_assembler_new(NULL, NULL, NULL, NULL);
	panda_assembler_copy_configuration(b, a);
	uint32_t m = src.n < 40 ? src.n : 40;
	fwrite(&m, 4, 1, out);
	for (uint32_t i = 0; i < m; i++) {
		unsigned char *rec = src.blob + (size_t) i * src.rec;
		uint16_t fl, rl;
		memcpy(&fl, rec, 2);
		memcpy(&rl, rec + 2, 2);
		panda_seq_identifier id;
		memset(&id, 0, sizeof id);
		id.x = (int) i;
		res = panda_assembler_assemble(b, &id, (const panda_qual *) (rec + 4), fl, (const panda_qual *) (rec + 4 + (size_t) src.stride * 2), rl);
		uint32_t got = res != NULL;
		fwrite(&got, 4, 1, out);
		if (res)
			dump(out, res);
	}
	fclose(out);
	panda_assembler_unref(b);
	panda_assembler_unref(a);
	free(src.blob);
	return 0;
}
