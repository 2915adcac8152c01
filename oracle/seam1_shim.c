/*
 * seam1_shim.c -- TEST INFRASTRUCTURE: the binding INTEGRATION.md section B shows, compiled for real.
 *
 * oracle/Makefile links this file with the reference's own objects (its main(), loader, report, scoring; the
 * reference's extend_alignment symbol is weakened with objcopy so that this definition wins) and with libramx.so.
 * The result, oracle/_ref/RAMExtend_seam1, is "the reference program with only its extension loop replaced":
 * tests/test_gpu_cli.py requires its output to be byte-identical to the unmodified reference's.
 */
#include <stdio.h>
#include <stdint.h>
#include "common.h"          /* reference headers (compiled where they lie under /root/reference) */
#include "sequence.h"
#include "score_system.h"
#define RAMX_USE_REFERENCE_STRUCTS
#include "ramx.h"

extern int VERBOSE, WHEN_TO_STOP, l;   /* reference globals, ram_extend.c:40,52,61 */

int extend_alignment(int direction, struct coreAlignment *coreAlign, int ****score, struct sequenceLibrary *seqLib,
                     char *master, int BANDWIDTH, int CAPPENALTY, int MINIMPROVEMENT, int L, int N,
                     struct scoringSystem *scoreParams, FILE *pathStringFile)
{
  ramx_set_runtime(VERBOSE, WHEN_TO_STOP, l);
  return ramx_extend_alignment(direction, coreAlign, score, seqLib, master, BANDWIDTH, CAPPENALTY, MINIMPROVEMENT, L, N,
                               scoreParams, pathStringFile);
}
