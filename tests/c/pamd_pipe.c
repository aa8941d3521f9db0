/*
 * pamd_pipe.c -- two FASTQ files in, one FASTA / FASTQ file out, through the C ABI only: pamd_pipe_run()
 * (include/pandaseq_amd_io.h), the loop a PANDAseq front end runs around the three device stages -- reader threads, pinned
 * blocks, asynchronous copies on streams of their own, writer threads.  It stands where main.c's
 * reader -> pool.c:76 -> output.c chain stands, minus option parsing and decompression.  A test client
 * (tests/test_pipe_gpu.py compares its output file byte for byte with the oracle chain; tools/pipe_file_rate.py times it),
 * not a command-line tool.
 *
 * usage: pamd_pipe <forward.fastq> <reverse.fastq> <out> <algo 0..6> <fasta|fastq> <pairs per chunk> [primers_after fwd_primer rev_primer]
 * PAMD_PIPE_DEVICES=0,0,1 in the environment: pamd_pipe_run_multi() over those devices (an assembler each, configured alike, as
 * pool.c:142-150 makes its workers') instead of pamd_pipe_run() on device 0.  Either way the run's counter block is printed as the
 * reference's STAT block (pamd_stat_block: pool.c:86-104) and, for tests, as 916 numbers.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pandaseq_amd.h"
#include "pandaseq_amd_io.h"

#define CHECK_PAMD(x) do { int r_ = (x); if (r_ != PAMD_OK) { fprintf(stderr, "%s: %s\n", #x, pamd_last_error()); return 3; } } while (0)

/* panda_nt_from_ascii / _complement (nt.c:152-160) for plain ACGTN primers */
static size_t primer_codes(const char *s, int complement, uint8_t *out) {
	size_t n = 0;
	for (; *s; s++) {
		uint8_t v = *s == 'A' ? 1 : *s == 'C' ? 2 : *s == 'G' ? 4 : *s == 'T' ? 8 : 15;
		if (complement && v != 15)
			v = (uint8_t) (((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3));
		out[n++] = v;
	}
	return n;
}

int main(int argc, char **argv) {
	if (argc < 7) {
		fprintf(stderr, "usage: %s fwd.fastq rev.fastq out algo fasta|fastq pairs_per_chunk [primers_after fprimer rprimer]\n", argv[0]);
		return 1;
	}
	int devices[64], n_devices = 0;
	pamd_assembler *as[64];
	if (getenv("PAMD_PIPE_DEVICES")) {
		char *list = strdup(getenv("PAMD_PIPE_DEVICES"));
		for (char *t = strtok(list, ","); t != NULL && n_devices < 64; t = strtok(NULL, ","))
			devices[n_devices++] = atoi(t);
		free(list);
	}
	const int multi = n_devices > 0;
	if (!multi)
		devices[n_devices++] = 0;
	for (int d = 0; d < n_devices; d++) {
		pamd_assembler *a = as[d] = pamd_assembler_create(devices[d]);
		if (!a) {
			fprintf(stderr, "%s\n", pamd_last_error());
			return 3;
		}
		CHECK_PAMD(pamd_set_algorithm(a, atoi(argv[4]), NAN));
		if (argc >= 10) {
			uint8_t codes[64];
			CHECK_PAMD(pamd_set_primers_after(a, atoi(argv[7])));
			CHECK_PAMD(pamd_set_forward_primer(a, codes, primer_codes(argv[8], 0, codes)));
			CHECK_PAMD(pamd_set_reverse_primer(a, codes, primer_codes(argv[9], 1, codes)));	/* complemented, args_assembler.c:222 */
		}
	}
	pamd_pipe_options o;
	pamd_pipe_stats s;
	memset(&o, 0, sizeof o);
	o.forward_path = argv[1];
	o.reverse_path = argv[2];
	o.out_path = argv[3];
	o.format = strcmp(argv[5], "fastq") == 0 ? PAMD_OUT_FASTQ : PAMD_OUT_FASTA;
	o.qualmin = 33;
	o.policy = PAMD_TAG_OPTIONAL;
	o.pairs_per_chunk = (size_t) atol(argv[6]);
	if (getenv("PAMD_PIPE_READERS"))
		o.reader_threads = atoi(getenv("PAMD_PIPE_READERS"));
	if (getenv("PAMD_PIPE_WRITERS"))
		o.writer_threads = atoi(getenv("PAMD_PIPE_WRITERS"));
	if (multi)
		CHECK_PAMD(pamd_pipe_run_multi(as, devices, n_devices, &o, &s));
	else
		CHECK_PAMD(pamd_pipe_run(as[0], 0, &o, &s));
	/* records the reader drops never reach the assembler in the reference (fastq.c:192) */
	printf("STAT\tREADS\t%lld\nSTAT\tNOALGN\t%lld\nSTAT\tLOWQ\t%lld\nSTAT\tBADR\t%lld\nSTAT\tOK\t%lld\nSTAT\tNOFP\t%lld\nSTAT\tNORP\t%lld\nSTOP\t%d\t%llu\n",
	       (long long) s.counters[PAMD_CTR_COUNT] - (long long) s.dropped, (long long) s.counters[PAMD_CTR_NOALGN], (long long) s.counters[PAMD_CTR_LOWQ],
	       (long long) s.counters[PAMD_CTR_BADR] - (long long) s.dropped, (long long) s.counters[PAMD_CTR_OK], (long long) s.counters[PAMD_CTR_NOFP], (long long) s.counters[PAMD_CTR_NORP],
	       (int) s.stop_code, (unsigned long long) s.stop_record);
	printf("PIPE\tpairs\t%llu\tchunks\t%llu\tin_bytes\t%llu\tout_bytes\t%llu\tsetup_s\t%.6f\tpump_s\t%.6f\n", (unsigned long long) s.pairs, (unsigned long long) s.chunks,
	       (unsigned long long) s.in_bytes, (unsigned long long) s.out_bytes, s.seconds_setup, s.seconds_pump);
	{	/* the run's STAT block as a reference worker prints it; then the whole counter block */
		static char block[32768];
		int64_t ctr[PAMD_NUM_COUNTERS];
		memcpy(ctr, s.counters, sizeof ctr);
		ctr[PAMD_CTR_COUNT] -= (int64_t) s.dropped;
		ctr[PAMD_CTR_BADR] -= (int64_t) s.dropped;
		if (pamd_stat_block(ctr, argc >= 10, argc >= 10, NULL, 0, "pamd", block, sizeof block) < 0) {
			fprintf(stderr, "%s\n", pamd_last_error());
			return 3;
		}
		fputs(block, stdout);
		printf("COUNTERS");
		for (int i = 0; i < PAMD_NUM_COUNTERS; i++)
			printf(" %lld", (long long) s.counters[i]);
		printf("\n");
	}
	for (int d = 0; d < n_devices; d++)
		pamd_assembler_destroy(as[d]);
	return 0;
}
