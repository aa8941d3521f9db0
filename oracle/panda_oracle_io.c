/*
 * panda_oracle_io.c -- CPU oracle for FASTQ ingest and FASTA / FASTQ formatting.
 * TEST INFRASTRUCTURE ONLY; see panda_oracle_io.h for scope and pinning status.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "panda_oracle_io.h"

/* ---- ASCII <-> nt (nt.c:25-114,152-173) ---------------------------------------------------- */
/* The reference indexes two 32-entry tables with (c & 0x1F).  The forward table is the IUPAC
 * meaning of each letter (U reads as T, X as N, every other letter as 0 = invalid); the reverse
 * table is the same set complemented base by base (A<->T, C<->G). */
static uint8_t iupac_set(int letter_index) {	/* 1 = 'A' ... 26 = 'Z' */
	static const char *const sets[27] = {
		"", "A", "CGT", "C", "AGT", "", "", "G", "ACT", "", "", "GT", "", "AC", "ACGT",
		"", "", "", "AG", "CG", "T", "T", "ACG", "AT", "ACGT", "CT", ""
	};
	uint8_t v = 0;
	if (letter_index < 1 || letter_index > 26)
		return 0;
	for (const char *s = sets[letter_index]; *s; s++)
		v |= *s == 'A' ? 1 : (*s == 'C' ? 2 : (*s == 'G' ? 4 : 8));
	return v;
}

uint8_t po_nt_from_ascii(char c) {
	return iupac_set((int) c & 0x1F);
}

uint8_t po_nt_from_ascii_complement(char c) {
	const uint8_t v = iupac_set((int) c & 0x1F);
	return (uint8_t) (((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3));
}

char po_nt_to_ascii(uint8_t nt) {
	/* panda_nt is a (signed) char: anything outside 0..15 prints as N (nt.c:169-171) */
	static const char letters[17] = "NACMGRSVTWYHKDBN";
	if ((int8_t) nt < 0 || (int8_t) nt > 15)
		return 'N';
	return letters[nt];
}

/* ---- sequence headers (seqid.c) ------------------------------------------------------------- */
void po_seqid_clear(po_seqid *id) {
	memset(id, 0, sizeof(*id));
}

int po_idfmt_has_direction(int format) {
	return format != PO_IDFMT_EBI_SRA && format != PO_IDFMT_SRA;
}

int po_seqid_equal(const po_seqid *a, const po_seqid *b) {
	return a->lane == b->lane && a->tile == b->tile && a->x == b->x && a->y == b->y
		&& strncmp(a->instrument, b->instrument, sizeof(a->instrument)) == 0
		&& strncmp(a->run, b->run, sizeof(a->run)) == 0
		&& strncmp(a->flowcell, b->flowcell, sizeof(a->flowcell)) == 0
		&& strncmp(a->tag, b->tag, PO_TAG_LEN) == 0;
}

int po_seqid_print(const po_seqid *id, char *buf) {
	return sprintf(buf, "%s:%s:%s:%d:%d:%d:%d:%s", id->instrument, id->run, id->flowcell, id->lane, id->tile, id->x, id->y, id->tag);
}

typedef struct {
	const char *s;
	size_t i;
} cursor;

static int peek(const cursor *c) {
	return (unsigned char) c->s[c->i];
}

/* the characters that end a chunk (PARSE_CHUNK_MAYBE, seqid.c:136) */
static int ends_chunk(int ch) {
	return ch == '\0' || ch == ':' || ch == '#' || ch == '/' || ch == ' ';
}

/* PARSE_PUSH (seqid.c:141): step over one separator, fail at the end of the string */
static int step(cursor *c) {
	if (peek(c) == '\0')
		return 0;
	c->i++;
	return 1;
}

/* PARSE_INT (seqid.c:138): fails at the end of the string and on any non-digit; an empty chunk is 0.
 * The reference's `int` arithmetic overflows for more than 9 digits (undefined); we wrap at 32 bits. */
static int chunk_int(cursor *c, int *out) {
	uint32_t v = 0;
	if (peek(c) == '\0')
		return 0;
	while (!ends_chunk(peek(c))) {
		const int ch = peek(c);
		if (ch < '0' || ch > '9')
			return 0;
		v = v * 10u + (uint32_t) (ch - '0');
		c->i++;
	}
	*out = (int) v;
	return 1;
}

/* PARSE_SRA_INT (seqid.c:139): ends at '.', ' ' or the end of the string, no end-of-string failure */
static int sra_int(cursor *c, int *out) {
	uint32_t v = 0;
	for (;;) {
		const int ch = peek(c);
		if (ch == '\0' || ch == '.' || ch == ' ')
			break;
		if (ch < '0' || ch > '9')
			return 0;
		v = v * 10u + (uint32_t) (ch - '0');
		c->i++;
	}
	*out = (int) v;
	return 1;
}

/* PARSE_STR (seqid.c:140).  DEFINED CHOICE: the reference lets a chunk of 100 or 101 characters run
 * over the end of its 100-byte field (and fails from 102); we fail from 100, the first length whose
 * terminator no longer fits the field. */
static int chunk_str(cursor *c, char *dst, size_t cap) {
	size_t n = 0;
	if (peek(c) == '\0')
		return 0;
	while (!ends_chunk(peek(c))) {
		if (n + 1 >= cap)
			return 0;
		dst[n++] = (char) peek(c);
		c->i++;
	}
	dst[n] = '\0';
	return 1;
}

/* the tag chunk (seqid.c:222-228,268-274): may be empty, fails at its 51st character (a 50-character
 * tag's terminator lands in the structure's padding in the reference; po_seqid.tag is compared over
 * PO_TAG_LEN characters exactly as panda_seqid_equal does) */
static int chunk_tag(cursor *c, po_seqid *id, char *spill) {
	size_t n = 0;
	id->tag[0] = '\0';
	while (!ends_chunk(peek(c))) {
		if (n >= PO_TAG_LEN)
			return 0;
		id->tag[n++] = (char) peek(c);
		c->i++;
		if (n < PO_TAG_LEN)
			id->tag[n] = '\0';
		else
			*spill = '\0';
	}
	return 1;
}

static int policy_ok(const po_seqid *id, int policy) {
	if (policy == PO_TAG_OPTIONAL)
		return 1;
	return policy == (id->tag[0] == '\0' ? PO_TAG_ABSENT : PO_TAG_PRESENT);
}

#define NEED(x) do { if (!(x)) goto fail; } while (0)

int po_seqid_parse(po_seqid *id, const char *input, int policy, int *format, size_t *end_offset) {
	cursor c = { input, 0 };
	int value = 0, fmt = PO_IDFMT_UNKNOWN, result = 0;
	char spill;

	if (strlen(input) > 3 && (input[0] == 'E' || input[0] == 'S') && input[1] == 'R' && input[2] == 'R') {
		/* Short Read Archive (seqid.c:157-172): <E|S>RR<number>.<number> and nothing else */
		fmt = input[0] == 'S' ? PO_IDFMT_SRA : PO_IDFMT_EBI_SRA;
		c.i = 3;
		po_seqid_clear(id);
		NEED(sra_int(&c, &value));
		NEED(step(&c));
		sprintf(id->instrument, "%cRR%d", (int) input[0], value);
		NEED(sra_int(&c, &value));
		NEED(step(&c));
		id->lane = value;
		NEED(step(&c));
		result = 1;
	} else if (strchr(input, '/') != NULL) {
		/* CASAVA before 1.7 (seqid.c:173-235): six colons before any '#' means a 1.7 name converted back */
		size_t colons = 0;
		for (const char *p = input; *p != '\0' && *p != '#'; p++)
			colons += *p == ':';
		if (colons == 6) {
			fmt = PO_IDFMT_CASAVA_CONVERTED;
			NEED(chunk_str(&c, id->instrument, sizeof(id->instrument)));
			NEED(step(&c));
			NEED(chunk_str(&c, id->run, sizeof(id->run)));
			NEED(step(&c));
			NEED(chunk_str(&c, id->flowcell, sizeof(id->flowcell)));
			NEED(step(&c));
		} else {
			fmt = PO_IDFMT_CASAVA_1_4;
			id->run[0] = '\0';
			id->flowcell[0] = '\0';
			NEED(chunk_str(&c, id->instrument, sizeof(id->instrument)));
			NEED(step(&c));
		}
		NEED(chunk_int(&c, &value));
		NEED(step(&c));
		id->lane = value;
		NEED(chunk_int(&c, &value));
		NEED(step(&c));
		id->tile = value;
		NEED(chunk_int(&c, &value));
		NEED(step(&c));
		id->x = value;
		NEED(chunk_int(&c, &value));
		NEED(step(&c));
		id->y = value;
		id->tag[0] = '\0';
		if (input[c.i - 1] == '#') {
			NEED(chunk_tag(&c, id, &spill));
			NEED(step(&c));
		}
		NEED(policy_ok(id, policy));
		NEED(chunk_int(&c, &value));
		result = value;
	} else {
		/* CASAVA 1.7+ (seqid.c:236-284): instrument:run:flowcell:lane:tile:x:y mate:filtered:control:tag */
		int mate;
		fmt = PO_IDFMT_CASAVA_1_7;
		NEED(chunk_str(&c, id->instrument, sizeof(id->instrument)));
		NEED(step(&c));
		NEED(chunk_str(&c, id->run, sizeof(id->run)));
		NEED(step(&c));
		NEED(chunk_str(&c, id->flowcell, sizeof(id->flowcell)));
		NEED(step(&c));
		NEED(chunk_int(&c, &value));
		NEED(step(&c));
		id->lane = value;
		NEED(chunk_int(&c, &value));
		NEED(step(&c));
		id->tile = value;
		NEED(chunk_int(&c, &value));
		NEED(step(&c));
		id->x = value;
		NEED(chunk_int(&c, &value));
		NEED(step(&c));
		id->y = value;
		NEED(chunk_int(&c, &mate));
		NEED(step(&c));
		/* "filtered": any chunk, but not the end of the string */
		NEED(peek(&c) != '\0');
		while (!ends_chunk(peek(&c)))
			c.i++;
		NEED(step(&c));
		/* control bits */
		NEED(chunk_int(&c, &value));
		NEED(step(&c));
		NEED(chunk_tag(&c, id, &spill));
		NEED(policy_ok(id, policy));
		result = mate;
	}
fail:
	if (format)
		*format = fmt;
	if (end_offset)
		*end_offset = c.i;
	return result;
}

#undef NEED

/* ---- the line reader (linebuf.c:57-89) over an in-memory text -------------------------------- */
typedef struct {
	const char *p;
	size_t n, pos;
	char line[PO_LINE_MAX + 1];
} lines;

/* The next line as a C string (so an embedded NUL cuts it short, as it does for every consumer of
 * panda_linebuf_next), or NULL: no bytes left, no '\n' within the next PO_LINE_MAX bytes, or -- DEFINED
 * CHOICE -- a last line that lacks its '\n' (the reference reads one byte past its buffered data
 * there, linebuf.c:71-77). */
static const char *next_line(lines *L) {
	const size_t left = L->n - L->pos;
	const size_t window = left < PO_LINE_MAX ? left : PO_LINE_MAX;
	if (left == 0)
		return NULL;
	const char *nl = memchr(L->p + L->pos, '\n', window);
	if (nl == NULL) {
		L->pos = L->n;
		return NULL;
	}
	size_t len = (size_t) (nl - (L->p + L->pos));
	memcpy(L->line, L->p + L->pos, len);
	L->pos += len + 1;
	if (len > 0 && L->line[len - 1] == '\r')	/* DOS line ends, linebuf.c:81-84 */
		len--;
	L->line[len] = '\0';
	return L->line;
}

/* read_seq (fastq.c:45-107): returns 0 or the PO_FQ_* code it fails with */
static int read_one(lines *L, int complement, int qualmin, po_qual *row, size_t *length, int *under_64) {
	const char *in = next_line(L);
	size_t pos = 0, qpos = 0;
	if (in == NULL)
		return PO_FQ_EOF;
	for (; *in != '\0' && pos < PO_MAX_LEN; in++) {
		const uint8_t nt = complement ? po_nt_from_ascii_complement(*in) : po_nt_from_ascii(*in);
		row[pos++].nt = nt;
		if (nt == 0)
			return PO_FQ_BADNT;
	}
	in = next_line(L);
	if (in == NULL)
		return PO_FQ_EOF;
	if (*in != '+')
		return po_nt_from_ascii(*in) != 0 ? PO_FQ_READLEN : PO_FQ_BADSEQ;
	in = next_line(L);
	if (in == NULL)
		return PO_FQ_EOF;
	for (; *in != '\0'; in++, qpos++) {
		const int v = (int) (signed char) *in;
		if (v < 64)
			*under_64 = 1;
		/* TOINDEX (fastq.c:44), its clamp quirk included: above qualmin + 46 the result is 46 - qualmin.
		 * Beyond PO_MAX_LEN qualities the reference writes past its buffer and then fails on the
		 * length test below; we skip the writes. */
		if (qpos < PO_MAX_LEN)
			row[qpos].qual = (int8_t) (v < qualmin ? 0 : ((v > qualmin + PO_PHREDMAX ? PO_PHREDMAX : v) - qualmin));
	}
	if (qpos != pos)
		return PO_FQ_NOQUAL;
	*length = pos;
	return 0;
}

void po_fastq_read(const char *fwd_text, size_t fwd_bytes, const char *rev_text, size_t rev_bytes, int qualmin, int policy, size_t max_pairs, po_qual *fwd, size_t fstride, po_qual *rev, size_t rstride, uint16_t *flen, uint16_t *rlen, po_seqid *ids, po_fastq_status *st) {
	static __thread lines F, R;
	po_qual frow[PO_MAX_LEN], rrow[PO_MAX_LEN];
	F.p = fwd_text;
	F.n = fwd_bytes;
	F.pos = 0;
	R.p = rev_text;
	R.n = rev_bytes;
	R.pos = 0;
	memset(st, 0, sizeof(*st));
	while (st->n_pairs < max_pairs) {
		po_seqid id, rid;
		int format = PO_IDFMT_UNKNOWN, fdir, rdir, code;
		size_t fl = 0, rl = 0;
		const char *line = next_line(&F);
		if (line == NULL)
			break;
		/* DEFINED CHOICE: an empty header line (the reference parses the byte after its terminator) fails */
		po_seqid_clear(&id);
		fdir = line[0] == '\0' ? 0 : po_seqid_parse(&id, line + 1, policy, &format, NULL);
		if (fdir == 0) {
			st->stop_code = PO_FQ_BADID_FWD;
			break;
		}
		line = next_line(&R);
		if (line == NULL)
			break;
		po_seqid_clear(&rid);
		rdir = line[0] == '\0' ? 0 : po_seqid_parse(&rid, line + 1, policy, NULL, NULL);
		if (rdir == 0) {
			st->stop_code = PO_FQ_BADID_REV;
			break;
		}
		if (!po_seqid_equal(&id, &rid) || (po_idfmt_has_direction(format) && rdir == fdir)) {
			st->stop_code = PO_FQ_NOTPAIRED;
			break;
		}
		if (format == PO_IDFMT_CASAVA_1_7)
			st->seen_under_64 = 1;
		memset(frow, 0, sizeof(frow));
		memset(rrow, 0, sizeof(rrow));
		code = read_one(&F, 0, qualmin, frow, &fl, &st->seen_under_64);
		if (code != 0) {
			st->stop_code = code;
			break;
		}
		st->non_empty = 1;
		code = read_one(&R, 1, qualmin, rrow, &rl, &st->seen_under_64);
		if (code != 0) {
			st->stop_code = code;
			st->stop_in_reverse = 1;
			break;
		}
		st->n_records++;
		if (fl == 0) {	/* `while (*forward_length == 0)`, fastq.c:192 */
			st->n_dropped++;
			continue;
		}
		const size_t k = st->n_pairs++;
		memset(fwd + k * fstride, 0, fstride * sizeof(po_qual));
		memset(rev + k * rstride, 0, rstride * sizeof(po_qual));
		memcpy(fwd + k * fstride, frow, fl * sizeof(po_qual));
		memcpy(rev + k * rstride, rrow, rl * sizeof(po_qual));
		flen[k] = (uint16_t) fl;
		rlen[k] = (uint16_t) rl;
		if (ids)
			ids[k] = id;
	}
	st->stop_record = st->n_records;
}

/* ---- result formatting (output.c:85-126) ----------------------------------------------------- */
static size_t put_head(char *buf, char lead, const po_seqid *name, double quality) {
	size_t n = 0;
	buf[n++] = lead;
	n += (size_t) po_seqid_print(name, buf + n);
	n += (size_t) sprintf(buf + n, ";%f", exp(quality));
	buf[n++] = '\n';
	return n;
}

size_t po_output_fasta(const po_seqid *name, double quality, const uint8_t *seq_nt, size_t len, char *buf) {
	size_t n;
	if (len == 0)
		return 0;
	n = put_head(buf, '>', name, quality);
	for (size_t i = 0; i < len; i++)
		buf[n++] = po_nt_to_ascii(seq_nt[i]);
	buf[n++] = '\n';
	return n;
}

size_t po_output_fastq(const po_tables *t, const po_seqid *name, double quality, const uint8_t *seq_nt, const double *seq_p, size_t len, char *buf) {
	size_t n;
	if (len == 0)
		return 0;
	n = put_head(buf, '@', name, quality);
	for (size_t i = 0; i < len; i++)
		buf[n++] = po_nt_to_ascii(seq_nt[i]);
	buf[n++] = '\n';
	buf[n++] = '+';
	buf[n++] = '\n';
	for (size_t i = 0; i < len; i++)
		buf[n++] = (char) (33 + po_result_phred(t, seq_p[i]));
	buf[n++] = '\n';
	return n;
}
