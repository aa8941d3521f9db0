/*
 * panda_oracle_io.h -- CPU ORACLE for the two data formats either side of the hot path
 * (SURVEY.md 8f rows 1 and 2): FASTQ ingest and FASTA / FASTQ result formatting.
 *
 * TEST INFRASTRUCTURE ONLY, like panda_oracle.h: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (pandaseq_amd/csrc) never does.
 *
 * A plain-C restatement, written from reading the reference, of
 *   - the line reader               linebuf.c:57-89
 *   - the ASCII <-> nt tables       nt.c:25-114,152-173
 *   - the sequence-header parser    seqid.c:128-285 (+ equality :96-101, printing :109-116)
 *   - the paired FASTQ reader       fastq.c:45-198
 *   - FASTA / FASTQ formatting      output.c:85-126, writer.c:234-238
 *
 * PINNING STATUS: the header parser is checked against the reference's OWN golden vectors
 * (check_parser.c:29-36: six headers with their expected fields, direction and format; committed
 * as data in tests/golden/seqid_check_parser.json).  Everything else here is PARITY UNPINNED
 * against a live reference: fastq.c / linebuf.c / nt.c / output.c include the autoconf-generated
 * config.h, which this image cannot produce, and the reference ships no fixtures for them.
 *
 * Where the reference's behaviour is undefined (buffer overruns), this restatement picks a
 * defined behaviour and says so at the spot; the product follows the same choice.
 */
#ifndef PANDA_ORACLE_IO_H
#define PANDA_ORACLE_IO_H

#include <stddef.h>
#include <stdint.h>
#include "panda_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PO_TAG_LEN 50		/* PANDA_TAG_LEN, pandaseq-common.h:45 */
#define PO_LINE_MAX (10 * PO_MAX_LEN)	/* struct panda_linebuf.data, linebuf.c:25 */

/* PandaTagging, pandaseq-common.h:165-178 */
enum { PO_TAG_PRESENT = 0, PO_TAG_ABSENT = 1, PO_TAG_OPTIONAL = 2 };
/* PandaIdFmt, pandaseq-common.h:188-195 */
enum { PO_IDFMT_UNKNOWN = 0, PO_IDFMT_SRA, PO_IDFMT_CASAVA_1_4, PO_IDFMT_CASAVA_1_7, PO_IDFMT_EBI_SRA, PO_IDFMT_CASAVA_CONVERTED };

/* panda_seq_identifier, pandaseq-common.h:235-244 */
typedef struct {
	char instrument[100];
	char run[100];
	char flowcell[100];
	int lane, tile, x, y;
	char tag[PO_TAG_LEN];
} po_seqid;

/* why the paired reader stopped (the PandaCode it logs, pandaseq-common.h:126-155) */
enum {
	PO_FQ_END = 0,		/* ran out of whole records */
	PO_FQ_BADID_FWD = 1,	/* PANDA_CODE_ID_PARSE_FAILURE on the forward header (fastq.c:126-129) */
	PO_FQ_BADID_REV = 2,	/* ... on the reverse header (fastq.c:133-136) */
	PO_FQ_NOTPAIRED = 3,	/* PANDA_CODE_NOT_PAIRED (fastq.c:137-140) */
	PO_FQ_BADNT = 4,	/* PANDA_CODE_BAD_NT (fastq.c:63-66) */
	PO_FQ_READLEN = 5,	/* PANDA_CODE_READ_TOO_LONG (fastq.c:75-77) */
	PO_FQ_BADSEQ = 6,	/* PANDA_CODE_PARSE_FAILURE (fastq.c:78-81) */
	PO_FQ_NOQUAL = 7,	/* PANDA_CODE_NO_QUALITY_INFO (fastq.c:96-99) */
	PO_FQ_EOF = 8		/* PANDA_CODE_PREMATURE_EOF: a record that lacks some of its lines */
};

typedef struct {
	uint64_t n_pairs;	/* pairs handed on (records with an empty forward read are dropped, fastq.c:192) */
	uint64_t n_records;	/* records consumed, dropped ones included */
	uint64_t n_dropped;
	int32_t stop_code;	/* PO_FQ_* */
	int32_t stop_in_reverse;	/* the failing read_seq() was the reverse one */
	uint64_t stop_record;	/* record index the reader stopped at (== n_records) */
	int32_t seen_under_64;	/* fastq.c:88-90,141-144 */
	int32_t non_empty;	/* fastq.c:105 */
} po_fastq_status;

uint8_t po_nt_from_ascii(char c);		/* nt.c:152-155 */
uint8_t po_nt_from_ascii_complement(char c);	/* nt.c:157-160 */
char po_nt_to_ascii(uint8_t nt);		/* nt.c:167-173 */

void po_seqid_clear(po_seqid *id);
/* panda_seqid_parse_fail (seqid.c:142-285): returns the mate direction, 0 = not parsed */
int po_seqid_parse(po_seqid *id, const char *input, int policy, int *format, size_t *end_offset);
int po_seqid_equal(const po_seqid *a, const po_seqid *b);	/* seqid.c:96-101 */
int po_idfmt_has_direction(int format);				/* seqid.c:41-44 */
/* "%s:%s:%s:%d:%d:%d:%d:%s" (seqid.c:109-116); returns the length written, buf needs >= 420 bytes */
int po_seqid_print(const po_seqid *id, char *buf);

/* The paired reader run to its end over two in-memory FASTQ texts (fastq.c:108-198 called until it
 * returns false).  Rows are po_qual at the given strides (multiples of 8 bases, >= PO_MAX_LEN rounded
 * up); bases past a read's length are zeroed.  ids may be NULL.  Stops after max_pairs pairs. */
void po_fastq_read(const char *fwd_text, size_t fwd_bytes, const char *rev_text, size_t rev_bytes, int qualmin, int policy, size_t max_pairs, po_qual *fwd, size_t fstride, po_qual *rev, size_t rstride, uint16_t *flen, uint16_t *rlen, po_seqid *ids, po_fastq_status *st);

/* panda_output_fasta / panda_output_fastq (output.c:85-126) for one result; returns bytes written
 * (0 for an empty sequence).  buf needs room for 430 + 2 * len + 8 bytes. */
size_t po_output_fasta(const po_seqid *name, double quality, const uint8_t *seq_nt, size_t len, char *buf);
size_t po_output_fastq(const po_tables *t, const po_seqid *name, double quality, const uint8_t *seq_nt, const double *seq_p, size_t len, char *buf);

#ifdef __cplusplus
}
#endif
#endif
