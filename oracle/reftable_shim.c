/*
 * reftable_shim.c -- exposes the one #define of the reference-generated table.h
 * (qual_nn_simple_bayesian, tablebuilder.c:120-125) as a symbol, so tests can read it through
 * ctypes next to the arrays of the reference-generated table.c.  Test infrastructure only.
 */
#include "table.h"

const double ref_qual_nn_simple_bayesian = qual_nn_simple_bayesian;
