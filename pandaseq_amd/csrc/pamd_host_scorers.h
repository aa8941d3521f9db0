// pamd_host_scorers.h -- the seven built-in classes' overlap_probability / match_probability on the host (pamd_host_scorers.cpp)
#ifndef PAMD_HOST_SCORERS_H
#define PAMD_HOST_SCORERS_H
#include "../../include/pandaseq_amd_compat.h"

#define PAMD_NUM_ALGOS 7
struct PamdHostScorer {
	PandaComputeOverlap overlap_probability;
	PandaComputeMatch match_probability;
};
extern const PamdHostScorer pamd_host_scorers[PAMD_NUM_ALGOS];	// indexed by PAMD_ALGO_*
#endif
