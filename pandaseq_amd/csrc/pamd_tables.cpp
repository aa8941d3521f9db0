// pamd_tables.cpp -- PHRED -> log-probability tables of the hot path, built on the host.
//
// What the reference does (mktable.c:23-155 + tablebuilder.c:73-183): for every PHRED score Q in
// [0,46] form p = 10^(-Q/10) (prob.h:21), evaluate a closed-form probability, take its log, and
// PRINT it with "%g" into table.c -- so the values the assembler computes with are 6-significant-
// digit decimals, not full-precision logs.  A full-precision table is off by up to ~5e-5 per entry,
// which breaks both the 1e-6 per-base tolerance and (through qual_nn * ~500) argmax decisions.
// We therefore run the same formulas and push every value through the same text round trip.
//
// This file is product code: it never touches oracle/.  tests/test_tables.py compares it with the
// table.c emitted by the reference's own generator (oracle/_ref, built from /root/reference).
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "pamd_internal.h"

namespace {

double through_text(double v) {
	char text[64];
	std::snprintf(text, sizeof text, "%g", v);	// tablebuilder.c:86,124,147
	return std::strtod(text, nullptr);
}

double err_prob(int phred) {	// PROBABILITY(), prob.h:21
	return std::pow(10.0, (-(double) phred) / 10.0);
}

// Closed forms, one per table; x and y are the error probabilities of the two bases.
double pr_match(double x, double y) {	// mktable.c:23-31
	return (1 - x) * (1 - y) + x * y / 3;
}

double pr_mismatch(double x, double y) {	// mktable.c:33-41 (and :84-92, same expression)
	return (1 - x) * y / 3 + (1 - y) * x / 3 + 2 * x * y / 9;
}

double pr_match_pear(double x, double y) {	// mktable.c:43-51
	return (1 - (1 - y) * x / 3 - (1 - x) * y / 3 - 2 * (1 - x) * (1 - y) / 9);
}

double pr_mismatch_pear(double x, double y) {	// mktable.c:53-61
	return (1 - x) * y / 3 + (1 - y) * x / 3 + x * y / 2;
}

double pr_mismatch_rdp_assembled(double x, double y) {	// mktable.c:94-104
	double lo = (x <= y) ? x : y;
	double v = 1 - (lo - x * y / 3.0) / (x + y - 4.0 / 3.0 * x * y);
	return (v == 0) ? DBL_MIN : v;
}

double pr_match_uparse(double x, double y) {	// mktable.c:106-117
	const double v = 1 - x * y / (1 - x - y + 4 * x * y / 3);
	return (v <= 0) ? DBL_MIN : v;
}

double pr_mismatch_uparse(double x, double y) {	// mktable.c:119-130
	const double v = 1 - (x + y / 3) / (x + y - 4 * x * y / 3);
	return (v <= 0) ? DBL_MIN : v;
}

}	// namespace

void pamd_build_tables(PamdTables *t) {
	t->qual_nn = through_text(std::log(0.25));	// mktable.c:141 -> the literal -1.38629
	for (int a = 0; a < PAMD_NQ; a++) {
		const double x = err_prob(a);
		t->score[a] = through_text(x == 1 ? -2.0 : std::log(1.0 - x));	// mktable.c:63-73
		t->score_err[a] = through_text(std::log(x));	// mktable.c:75-82
		for (int b = 0; b < PAMD_NQ; b++) {
			const double y = err_prob(b);
			const int at = a * PAMD_NQ + b;
			t->match_sb[at] = through_text(std::log(pr_match(x, y)));
			t->mismatch_sb[at] = through_text(std::log(pr_mismatch(x, y)));
			t->match_pear[at] = through_text(std::log(pr_match_pear(x, y)));
			t->mismatch_pear[at] = through_text(std::log(pr_mismatch_pear(x, y)));
			t->mismatch_rdp[at] = through_text(std::log(pr_mismatch(x, y)));
			t->mismatch_rdp_asm[at] = through_text(std::log(pr_mismatch_rdp_assembled(x, y)));
			t->match_uparse[at] = through_text(std::log(pr_match_uparse(x, y)));
			t->mismatch_uparse[at] = through_text(std::log(pr_mismatch_uparse(x, y)));
		}
	}
}
