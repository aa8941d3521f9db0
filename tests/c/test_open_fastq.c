/*
 * test_open_fastq.c -- the reference's file-level entry point, panda_assembler_open() (pandaseq-assembler.h:96),
 * then the pool.c:76 loop over panda_assembler_next(), writing what panda_output_fasta() (output.c:85-103) writes.
 * Plain C against include/pandaseq_amd_compat.h, or with -DUSE_REFERENCE_HEADERS against the reference's own
 * pandaseq.h.  tests/test_pipe_gpu.py compares the output file with the oracle chain.
 *
 * usage: test_open_fastq <forward.fastq[.gz]> <reverse.fastq[.gz]> <out.fasta> <algorithm name>
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef USE_REFERENCE_HEADERS
#include <pandaseq.h>
int pamd_compat_last_fastq_stop(void);
#else
#include "pandaseq_amd_compat.h"
#endif

int main(int argc, char **argv) {
	if (argc < 5)
		return 1;
	PandaAssembler a = panda_assembler_open(argv[1], argv[2], NULL, 33, PANDA_TAG_OPTIONAL);
	if (a == NULL) {
		fprintf(stderr, "cannot open\n");
		return 2;
	}
	PandaAlgorithm algo = NULL;
	if (strcmp(argv[4], "pear") == 0)
		algo = panda_algorithm_pear_new();
	else if (strcmp(argv[4], "rdp_mle") == 0)
		algo = panda_algorithm_rdp_mle_new();
	else if (strcmp(argv[4], "uparse") == 0)
		algo = panda_algorithm_uparse_new();
	else
		algo = panda_algorithm_simple_bayes_new();
	panda_assembler_set_algorithm(a, algo);
	panda_algorithm_unref(algo);
	FILE *out = fopen(argv[3], "wb");
	const panda_result_seq *res;
	static const char letters[] = "NACMGRSVTWYHKDBN";
	while ((res = panda_assembler_next(a)) != NULL) {	/* pool.c:76 */
		if (res->sequence_length == 0)
			continue;
		fprintf(out, ">%s:%s:%s:%d:%d:%d:%d:%s;%f\n", res->name.instrument, res->name.run, res->name.flowcell, res->name.lane, res->name.tile, res->name.x, res->name.y, res->name.tag, exp(res->quality));
		for (size_t i = 0; i < res->sequence_length; i++) {
			const int nt = res->sequence[i].nt;
			fputc(nt < 0 || nt > 15 ? 'N' : letters[nt], out);
		}
		fputc('\n', out);
	}
	fclose(out);
	printf("STAT\tREADS\t%ld\nSTAT\tOK\t%ld\nSTAT\tNOALGN\t%ld\nSTAT\tLOWQ\t%ld\nSTAT\tBADR\t%ld\nSTOP\t%d\n", (long) panda_assembler_get_count(a), (long) panda_assembler_get_ok_count(a),
	       (long) panda_assembler_get_failed_alignment_count(a), (long) panda_assembler_get_low_quality_count(a), (long) panda_assembler_get_bad_read_count(a), pamd_compat_last_fastq_stop());
	panda_assembler_unref(a);
	return 0;
}
