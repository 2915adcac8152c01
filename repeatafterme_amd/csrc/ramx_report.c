/*
 * ramx_report.c -- the core-edge table printed before extension (reference report.c:161-502,
 * printCoreEdges).  Pure formatting; byte-identical output is the contract (wrappers and the
 * golden stdout fixtures depend on it), so the column-width rules and their quirks are kept:
 *   - BED-range column is (2*w+1) wide where w = digits of the largest coordinate, but whenever
 *     2*w+1 < 9 the reference sets w itself to 9, i.e. the column becomes 19 wide (report.c:228-229);
 *   - "Seq" column is max(4, digits(number of cores + 1)) wide;
 *   - flank previews are 10 bp, '*' marks a flank cut short by the window.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ramx_internal.h"

static char code_to_char(int z)
{
  static const char t[8] = { 'A', 'C', 'G', 'T', 'a', 'c', 'g', 't' };
  return (z >= 0 && z < 8) ? t[z] : 'N';      /* sequence.c:1090-1110 */
}

static int code_compl(int c)                   /* sequence.c:1141-1160 */
{
  if (c >= 0 && c <= 3) return 3 - c;
  if (c >= 4 && c <= 7) return 4 + (7 - c);
  return RAMX_SYM_N;
}

static void append_base(char *dst, const struct sequenceLibrary *lib, uint64_t at, int complement)
{
  size_t n = strlen(dst);
  int code = ramx_lib_code(lib, at);      /* one byte per base, or the packed twin */
  dst[n] = code_to_char(complement ? code_compl(code) : code);
  dst[n + 1] = 0;
}

static const char *flag_name(enum CoreBoundFlag f)
{
  switch (f)
  {
    case L_BOUNDARY: return "L_BOUNDARY";
    case SEQ_BOUNDARY: return "SEQ_BOUNDARY";
    case CORE_BOUNDARY: return "CORE_BOUNDARY";
    case EXT_BOUNDARY: return "EXT_BOUNDARY";
  }
  return "";
}

static int is_blank(const struct coreAlignment *c)
{
  return c->seqIdx == 0 && c->leftSeqPos == 0 && c->rightSeqPos == 0;
}

struct edges_ctx
{
  const struct coreAlignment **arr;
  const struct sequenceLibrary *seqLib;
  int maxIDLen, maxPosLen, maxSArrayLen, maxIdxLen, debug;
};

/* rows [lo_i, hi_i) of the core table (report.c:161-502) */
static void edges_rows(int lo_i, int hi_i, FILE **outs, void *user)
{
  const struct edges_ctx *t = (const struct edges_ctx *)user;
  const struct sequenceLibrary *seqLib = t->seqLib;
  const int maxIDLen = t->maxIDLen, maxPosLen = t->maxPosLen, maxSArrayLen = t->maxSArrayLen, maxIdxLen = t->maxIdxLen, debug = t->debug;
  FILE *out = outs[0];
  for (int n = lo_i; n < hi_i; n++)
  {
    const struct coreAlignment *c = t->arr[n];
    const char *ident = seqLib->identifiers[c->seqIdx];
    uint64_t off = (seqLib->offsets != NULL && seqLib->offsets[c->seqIdx] > 0) ? seqLib->offsets[c->seqIdx] : 0;
    uint64_t lo = c->seqIdx > 0 ? seqLib->boundaries[c->seqIdx - 1] : 0;
    uint64_t hi = seqLib->boundaries[c->seqIdx] - 1;
    int coreWidth = abs((int)(c->rightSeqPos - c->leftSeqPos)) + 1;
    char idBuff[64], range[64], rangeCol[128], sarr[64], sarrCol[128];
    char leftExt[32], coreSeq[64], rightExt[32];
    int i, j, disp;

    /* identifier: truncated / padded to maxIDLen, "..." when longer than 50 */
    strncpy(idBuff, ident, (size_t)maxIDLen);
    idBuff[maxIDLen] = 0;
    for (i = (int)strlen(idBuff); i < maxIDLen; i++) { idBuff[i] = ' '; idBuff[i + 1] = 0; }
    if (strlen(ident) > 50)
      for (i = 0; i < 3; i++) idBuff[maxIDLen - i - 1] = '.';

    /* core coordinates back in BED form, right-justified */
    if (c->orient)
      sprintf(range, "%ld-%ld", (long)(off + c->rightSeqPos - lo), (long)(off + c->leftSeqPos - lo + 1));
    else
      sprintf(range, "%ld-%ld", (long)(off + c->leftSeqPos - lo), (long)(off + c->rightSeqPos - lo + 1));
    rangeCol[0] = 0;
    for (i = 0; i < (maxPosLen * 2 + 1) - (int)strlen(range); i++) strcat(rangeCol, " ");
    strcat(rangeCol, range);
    sprintf(sarr, "%ld-%ld", (long)c->leftSeqPos, (long)c->rightSeqPos);
    sarrCol[0] = 0;
    for (i = 0; i < (maxSArrayLen * 2 + 1) - (int)strlen(sarr); i++) strcat(sarrCol, " ");
    strcat(sarrCol, sarr);

    leftExt[0] = coreSeq[0] = rightExt[0] = 0;
    if (c->orient)
    {
      /* reverse strand: left is the higher coordinate; everything is shown complemented */
      disp = 10;
      if (c->leftSeqPos + 1 + disp > hi) disp = (int)(hi - c->leftSeqPos - 1);
      if (disp < 10) strcat(leftExt, "*");
      for (j = disp; j > 0; j--) append_base(leftExt, seqLib, c->leftSeqPos + j, 1);
      if (coreWidth < 20)
      {
        for (j = 0; j < ((24 - coreWidth) / 2); j++) strcat(coreSeq, " ");
        for (j = 0; j > -coreWidth; j--) append_base(coreSeq, seqLib, c->leftSeqPos + j, 1);
      }
      else
      {
        for (j = 0; j > -10; j--) append_base(coreSeq, seqLib, c->leftSeqPos + j, 1);
        strcat(coreSeq, "....");
        for (j = 9; j >= 0; j--) append_base(coreSeq, seqLib, c->rightSeqPos + j, 1);
      }
      disp = 10;
      if (c->rightSeqPos < (uint64_t)disp) disp = (int)c->rightSeqPos;
      else if (c->rightSeqPos - disp < lo) disp = (int)(c->rightSeqPos - lo);
      for (j = 0; j > -disp; j--) append_base(rightExt, seqLib, c->rightSeqPos + j - 1, 1);
      if (disp < 10) strcat(rightExt, "*");
      fprintf(out, "%-*d %s %s -      %d/%d ", maxIdxLen, n, idBuff, rangeCol, c->leftExtendable, c->rightExtendable);
    }
    else
    {
      disp = 10;
      if (c->leftSeqPos < (uint64_t)disp) disp = (int)c->leftSeqPos;
      else if (c->leftSeqPos - disp < lo) disp = (int)(c->leftSeqPos - lo);
      if (disp < 10) strcat(leftExt, "*");
      for (j = -disp; j <= -1; j++) append_base(leftExt, seqLib, c->leftSeqPos + j, 0);
      if (coreWidth < 20)
      {
        for (j = 0; j < ((24 - coreWidth) / 2); j++) strcat(coreSeq, " ");
        for (j = 0; j < coreWidth; j++) append_base(coreSeq, seqLib, c->leftSeqPos + j, 0);
      }
      else
      {
        for (j = 0; j < 10; j++) append_base(coreSeq, seqLib, c->leftSeqPos + j, 0);
        strcat(coreSeq, "....");
        for (j = -9; j <= 0; j++) append_base(coreSeq, seqLib, c->rightSeqPos + j, 0);
      }
      disp = 10;
      if (c->rightSeqPos + disp > hi) disp = (int)(hi - c->rightSeqPos);
      for (j = 1; j <= disp; j++) append_base(rightExt, seqLib, c->rightSeqPos + j, 0);
      if (disp < 10) strcat(rightExt, "*");
      fprintf(out, "%-*d %s %s +      %d/%d ", maxIdxLen, n, idBuff, rangeCol, c->leftExtendable, c->rightExtendable);
    }
    if (c->leftExtendable) fprintf(out, "%11s", leftExt);
    else fprintf(out, "           ");
    fprintf(out, " [%-24s] ", coreSeq);
    if (c->rightExtendable) fprintf(out, "%-11s", rightExt);
    else fprintf(out, "           ");
    if (debug == 1)
      fprintf(out, " %s %ld-%ld %ld-%ld %s/%s", sarrCol, (long)lo, (long)hi, (long)c->lowerSeqBound, (long)c->upperSeqBound,
             flag_name(c->lowerSeqBoundFlag), flag_name(c->upperSeqBoundFlag));
    fprintf(out, "\n");
  }
}

void ramx_print_core_edges(struct coreAlignment *coreAlign, struct sequenceLibrary *seqLib, char omitBlanks, char debug)
{
  struct coreAlignment *c;
  uint64_t maxPos = 0;
  int maxIDLen = 0, maxIdx = 1;
  char num[64];

  for (c = coreAlign; c != NULL; c = c->next)
  {
    if (omitBlanks == 1 && is_blank(c)) break;
    uint64_t off = (seqLib->offsets != NULL && seqLib->offsets[c->seqIdx] > 0) ? seqLib->offsets[c->seqIdx] : 0;
    uint64_t lo = c->seqIdx > 0 ? seqLib->boundaries[c->seqIdx - 1] : 0;
    int idl = (int)strlen(seqLib->identifiers[c->seqIdx]);
    if (idl > maxIDLen) maxIDLen = idl;
    if (off + c->leftSeqPos - lo + 1 > maxPos) maxPos = off + c->leftSeqPos - lo + 1;
    if (off + c->rightSeqPos - lo + 1 > maxPos) maxPos = off + c->rightSeqPos - lo + 1;
    maxIdx++;
  }
  if (maxIDLen > 50) maxIDLen = 50;
  if (maxIDLen < 5) maxIDLen = 5;
  snprintf(num, sizeof(num), "%ld", (long)maxPos);
  int maxPosLen = (int)strlen(num);
  snprintf(num, sizeof(num), "%ld", (long)seqLib->length);
  if (maxPosLen * 2 + 1 < 9) maxPosLen = 9;
  const int maxSArrayLen = (int)strlen(num);
  int maxIdxLen = (int)floor(log10(maxIdx)) + 1;
  if (maxIdxLen < 4) maxIdxLen = 4;

  if (debug == 1)
  {
    printf("%-*s %-*s %-*s %-*s %-*s  Left-Flank           Core           Right-Flank  %-*s Hard-Bounds Soft-Bounds BoundFlags(Lower/Upper)\n",
           maxIdxLen, "Seq", maxIDLen, "Ident", maxPosLen * 2 + 1, "BED-range", 6, "Orient", 4, "L/R?",
           maxSArrayLen, "seq[]-core");
    printf("------------------------------------------------------------------------------------------------------------------------------------------------------------\n");
  }
  else
  {
    printf("%-*s %-*s %-*s %-*s %-*s  Left-Flank           Core           Right-Flank\n",
           maxIdxLen, "Seq", maxIDLen, "Ident", maxPosLen * 2 + 1, "BED-range", 6, "Orient", 4, "L/R?");
    printf("------------------------------------------------------------------------------------\n");
  }

  /* the rows: formatted chunk by chunk on the host's cores, written in order */
  {
    int cnt = 0;
    for (c = coreAlign; c != NULL; c = c->next) { if (omitBlanks == 1 && is_blank(c)) break; cnt++; }
    const struct coreAlignment **arr = (const struct coreAlignment **)malloc(sizeof(*arr) * (size_t)(cnt ? cnt : 1));
    cnt = 0;
    for (c = coreAlign; c != NULL; c = c->next) { if (omitBlanks == 1 && is_blank(c)) break; arr[cnt++] = c; }
    struct edges_ctx ctx;
    ctx.arr = arr; ctx.seqLib = seqLib; ctx.maxIDLen = maxIDLen; ctx.maxPosLen = maxPosLen; ctx.maxSArrayLen = maxSArrayLen;
    ctx.maxIdxLen = maxIdxLen; ctx.debug = debug;
    FILE *outs[1] = { stdout };
    ramx_parallel_chunks(cnt, 1, outs, edges_rows, &ctx);
    free(arr);
  }
}
