/*
 * ramx_score.c -- scoring systems (reference score_system.c / score_system.h:23-37).
 *
 * Same interface and the same numbers as the reference: four RepeatMasker-derived, non-symmetric
 * nucleotide matrices indexed [consensus][sequence] with per-matrix affine gap penalties, and the
 * RepeatScout match/mismatch/linear-gap scheme.  Table-driven instead of assignment lists; cells
 * the reference leaves uninitialised are zero here.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ramx_internal.h"

#define MSIZE 100

struct named_matrix
{
  const char *name;
  int gapopen, gapextn;
  int v[4][4];            /* [consensus A,C,G,T][sequence A,C,G,T] */
};

/* values: reference score_system.c:207-376 (rows there are written in A,G,C,T order) */
static const struct named_matrix k_matrices[] = {
  { "14p43g", -33, -7, { {   9, -18, -10, -21 }, { -18,  11, -18,  -7 }, {  -7, -18,  11, -18 }, { -21, -10, -18,   9 } } },
  { "18p43g", -30, -6, { {   9, -15,  -8, -18 }, { -16,  10, -16,  -5 }, {  -5, -16,  10, -16 }, { -18,  -8, -15,   9 } } },
  { "20p43g", -28, -5, { {   9, -15,  -8, -17 }, { -15,  10, -15,  -4 }, {  -4, -15,  10, -15 }, { -17,  -8, -15,   9 } } },
  { "25p43g", -25, -5, { {   8, -13,  -6, -15 }, { -13,   9, -13,  -2 }, {  -2, -13,   9, -13 }, { -15,  -6, -13,   8 } } },
};

static struct scoringSystem *new_system(void)
{
  struct scoringSystem *s = (struct scoringSystem *)calloc(1, sizeof(*s));
  int **m = (int **)malloc(MSIZE * sizeof(int *));
  if (!s || !m)
  {
    printf("Could not allocate space for dna scoring matrix!\n");
    exit(1);
  }
  for (int i = 0; i < MSIZE; i++)
  {
    m[i] = (int *)calloc(MSIZE, sizeof(int));
    if (!m[i])
    {
      printf("Could not allocate space for dna scoring matrix row!\n");
      exit(1);
    }
  }
  s->matrix = m;
  s->msize = MSIZE;
  s->alphabet = (char *)"ACGTN";
  return s;
}

/* A,C,G,T block + the N row/column */
static void fill_core(int **m, const int v[4][4], int nscore)
{
  for (int a = 0; a < 4; a++)
  {
    for (int b = 0; b < 4; b++) m[a][b] = v[a][b];
    m[a][RAMX_SYM_N] = nscore;
    m[RAMX_SYM_N][a] = nscore;
  }
  m[RAMX_SYM_N][RAMX_SYM_N] = nscore;
}

/* soft-masked codes a,c,g,t = 4..7 (score_system.c:152-163 and 384-395) */
static void fill_softmasked(int **m, int against_upper)
{
  for (int i = 4; i <= 7; i++)
  {
    for (int j = 4; j <= 7; j++) { m[i][j] = -1; m[j][i] = -1; }
    for (int j = 0; j <= 3; j++) { m[i][j] = against_upper; m[j][i] = against_upper; }
    m[i][RAMX_SYM_N] = against_upper;
    m[RAMX_SYM_N][i] = against_upper;
  }
}

/* score_system.c:37-89: smallest lambda with sum_ij p_i p_j exp(lambda s_ij) >= 1, doubling then bisection */
double ramx_calculate_lambda(struct scoringSystem *s)
{
  double lambda = 0.5, lo = 0, hi = 0;
  for (;;)
  {
    double sum = 0, check = 0;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++)
      {
        sum += s->m_bg_freqs[i] * s->m_bg_freqs[j] * exp(lambda * s->matrix[i][j]);
        check += s->m_bg_freqs[i] * s->m_bg_freqs[j];
      }
    if (check > 1.001 || check < 0.999) return -1.0;
    if (sum >= 1.0) break;
    lo = lambda;
    lambda *= 2.0;
  }
  hi = lambda;
  while (hi - lo > 0.00001)
  {
    double sum = 0;
    lambda = (lo + hi) / 2.0;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++)
        sum += s->m_bg_freqs[i] * s->m_bg_freqs[j] * exp(lambda * s->matrix[i][j]);
    if (sum >= 1.0) hi = lambda;
    else lo = lambda;
  }
  return lambda;
}

struct scoringSystem *ramx_get_matrix(const char *matrixName)
{
  for (size_t k = 0; k < sizeof(k_matrices) / sizeof(k_matrices[0]); k++)
  {
    if (strcmp(matrixName, k_matrices[k].name) == 0)
    {
      struct scoringSystem *s = new_system();
      /* the reference leaves `name` unset for 18p43g (score_system.c:250-253); we always set it */
      s->name = (char *)k_matrices[k].name;
      s->gapopen = k_matrices[k].gapopen;
      s->gapextn = k_matrices[k].gapextn;
      fill_core(s->matrix, k_matrices[k].v, -1);
      fill_softmasked(s->matrix, -1);
      s->m_bg_freqs[0] = 0.285; s->m_bg_freqs[1] = 0.215; s->m_bg_freqs[2] = 0.215; s->m_bg_freqs[3] = 0.285;
      s->m_lambda = ramx_calculate_lambda(s);
      return s;
    }
  }
  /* score_system.c:377-380 */
  printf("Custom matrices not supported ( yet ).  %s is not an internally coded matrix!\n", matrixName);
  exit(1);
}

struct scoringSystem *ramx_get_matrix_using_gap_penalties(const char *matrixName, int gapopen, int gapextn)
{
  struct scoringSystem *s = ramx_get_matrix(matrixName);
  s->gapopen = gapopen;
  s->gapextn = gapextn;
  return s;
}

struct scoringSystem *ramx_get_repeatscout_matrix(int match, int mismatch, int gap)
{
  struct scoringSystem *s = new_system();
  int v[4][4];
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) v[a][b] = (a == b) ? match : mismatch;
  s->name = (char *)"repeatscout";
  s->gapopen = 0;        /* linear gap model: score_system.c:118-119 */
  s->gapextn = gap;
  fill_core(s->matrix, (const int (*)[4])v, mismatch);
  fill_softmasked(s->matrix, mismatch);
  for (int i = 0; i < 4; i++) s->m_bg_freqs[i] = 0.25;
  s->m_lambda = ramx_calculate_lambda(s);
  return s;
}

void ramx_free_scoring_system(struct scoringSystem *s)
{
  if (!s) return;
  for (int i = 0; i < s->msize; i++) free(s->matrix[i]);
  free(s->matrix);
  free(s);
}
