/*
 * ramx_oracle.c -- CPU restatement of the reference's extension loop, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see ramx_oracle.h).  It deliberately keeps the
 * reference's *structure* (five row evaluations per column, unsigned index
 * wrap, int32 column sums) so that it can be read side by side with the
 * reference; it is not tuned.  State is one flat int32 array instead of the
 * reference's int**** (bnw_extend.c:87-135); everything else follows the
 * cited lines.
 */
#include <stdlib.h>
#include <string.h>
#include "ramx_oracle.h"

#define IMPOSSIBLE (-1000000000)   /* bnw_extend.c:760,802-804 */
#define OOBSENTINEL (-987654321)   /* bnw_extend.c:767 */
#define SYM_N 99                    /* sequence.h:15 */

/* sequence.c:1141-1160 */
static int oracle_compl(int c)
{
  switch (c)
  {
    case 0: return 3;
    case 1: return 2;
    case 2: return 1;
    case 3: return 0;
    case 4: return 7;
    case 5: return 6;
    case 6: return 5;
    case 7: return 4;
    default: return SYM_N;
  }
}

#define SC(ff, n, j, s) score[((((size_t)(ff) * n_align + (n)) * B + (j)) << 1) + (s)]

/* bnw_extend.c:750-1048 */
int ramx_oracle_nw_row(int direction, int row_idx, int n, int n_align, int cons_base,
                       int64_t left_pos, int64_t right_pos, int orient,
                       int32_t *score, uint64_t lower_seq_bound, uint64_t upper_seq_bound,
                       const int8_t *sequence, int *max_score_sequence_idx,
                       const int32_t *matrix, int gapopen, int gapextn, int bandwidth)
{
  const int B = 2 * bandwidth + 1;
  const int cur = row_idx % 2;         /* :770 */
  const int prev = !cur;               /* :771 */
  int best_row_score = IMPOSSIBLE;
  uint64_t start_seq_pos;

  /* :778-788 first base outside the core on the extending side */
  if (direction)
    start_seq_pos = orient ? (uint64_t)right_pos - 1 : (uint64_t)right_pos + 1;
  else
    start_seq_pos = orient ? (uint64_t)left_pos + 1 : (uint64_t)left_pos - 1;

  for (int offset = -bandwidth; offset <= bandwidth; offset++)   /* :800 */
  {
    int ins_score = IMPOSSIBLE, del_score = IMPOSSIBLE, sub_score = IMPOSSIBLE;
    int seq_offset, out_of_bounds = 0, t;
    uint64_t seq_idx;
    const int j = offset + bandwidth;

    /* :824-837 */
    if (direction == orient)
      seq_offset = -offset - row_idx;
    else
      seq_offset = offset + row_idx;
    seq_idx = start_seq_pos + (uint64_t)(int64_t)seq_offset;

    /* :843-868 (the int operands are converted to uint64_t exactly as C does there) */
    if (start_seq_pos < (uint64_t)(int64_t)bandwidth && seq_offset < 0 &&
        (uint64_t)(int64_t)abs(seq_offset) > start_seq_pos)
      out_of_bounds = 1;
    else if (seq_idx > upper_seq_bound)
      out_of_bounds = 1;
    else if (seq_idx < lower_seq_bound)
      out_of_bounds = 1;

    if (!out_of_bounds)
    {
      int subvalue;
      if (offset < bandwidth)           /* :892-905 deletion */
      {
        t = SC(prev, n, j + 1, 0) + gapopen + gapextn;
        if (t > del_score) del_score = t;
        t = SC(prev, n, j + 1, 1) + gapextn;
        if (t > del_score) del_score = t;
      }
      /* :928-956 substitution */
      if (orient)
        subvalue = matrix[cons_base * RAMX_ORACLE_MSIZE + oracle_compl(sequence[seq_idx])];
      else
        subvalue = matrix[cons_base * RAMX_ORACLE_MSIZE + sequence[seq_idx]];
      t = SC(prev, n, j, 0) + subvalue;
      if (t > sub_score) sub_score = t;
      t = SC(prev, n, j, 1) + subvalue;
      if (t > sub_score) sub_score = t;
      if (offset > -bandwidth)          /* :972-985 insertion (current row, previous cell) */
      {
        t = SC(cur, n, j - 1, 0) + gapopen + gapextn;
        if (t > ins_score) ins_score = t;
        t = SC(cur, n, j - 1, 1) + gapextn;
        if (t > ins_score) ins_score = t;
      }
    }
    else
    {
      /* :990-1002 */
      if (offset < 0 && row_idx < bandwidth)
        sub_score = gapopen + ((row_idx + 1) * gapextn);
      else
        sub_score = OOBSENTINEL;
      ins_score = sub_score;
      del_score = sub_score;
    }
    /* :1005-1024 */
    SC(cur, n, j, 0) = sub_score;
    int gap_score = (ins_score > del_score) ? ins_score : del_score;
    SC(cur, n, j, 1) = gap_score;
    int cell_score = (gap_score > sub_score) ? gap_score : sub_score;
    if (cell_score > best_row_score)
    {
      *max_score_sequence_idx = row_idx + offset;
      best_row_score = cell_score;
    }
  }
  return best_row_score;
}

/* ram_extend.c:859-1258 */
int ramx_oracle_extend(int direction, ramx_oracle_cores *c, const int8_t *sequence,
                       int8_t *master, const ramx_oracle_params *p, ramx_oracle_trace *tr)
{
  const int N = c->n, W = p->bandwidth, L = p->L;
  const int B = 2 * W + 1;
  const int n_align = N > 0 ? N : 1;
  int32_t *score = (int32_t *)malloc((size_t)2 * n_align * B * 2 * sizeof(int32_t));
  int *high = (int *)calloc(n_align, sizeof(int));       /* overall_sequence_high_score      :900 */
  int *pos = (int *)calloc(n_align, sizeof(int));        /* overall_sequence_high_score_pos  :901 */
  int *thigh = (int *)calloc(n_align, sizeof(int));      /* trimmed_sequence_high_score      :902 */
  int *tpos = (int *)calloc(n_align, sizeof(int));       /* trimmed_sequence_high_score_pos  :903 */
  int curr_extension_score = 0, max_extension_score = 0, max_row = -1;
  int row_idx, n, best_idx = 0;

  /* :909-960 boundary row lives in flip-flop slot 1 */
  for (n = 0; n < N; n++)
    for (int o = -W; o <= W; o++)
    {
      int v = (o == 0) ? 0 : (abs(o) * p->gapextn + p->gapopen);
      SC(1, n, o + W, 0) = v;
      SC(1, n, o + W, 1) = v;
    }

  for (row_idx = 0; row_idx < L; row_idx++)              /* :970 */
  {
    int besta = 0;
    curr_extension_score = 0;
    for (int a = 0; a < 4; a++)                          /* :975 */
    {
      int score_given_cons = 0;                          /* int, as in the reference (:874) */
      for (n = 0; n < N; n++)
      {
        if ((direction && c->right_ext[n]) || (!direction && c->left_ext[n]))
        {
          int best = ramx_oracle_nw_row(direction, row_idx, n, n_align, a, c->left_pos[n], c->right_pos[n],
                                        c->orient[n], score, (uint64_t)c->lower[n], (uint64_t)c->upper[n],
                                        sequence, &best_idx, p->matrix, p->gapopen, p->gapextn, W);
          if (best < 0) best = 0;                        /* :1042 */
          if (best >= high[n] + p->cappenalty)           /* :1052-1062 */
            score_given_cons += best;
          else
            score_given_cons += high[n] + p->cappenalty;
        }
      }
      if (tr && tr->col_sums) tr->col_sums[(size_t)row_idx * 4 + a] = score_given_cons;
      if (score_given_cons > curr_extension_score)       /* :1081-1085 */
      {
        curr_extension_score = score_given_cons;
        besta = a;
      }
    }
    if (direction)                                       /* :1092-1095 */
      master[L + p->l + row_idx] = (int8_t)besta;
    else
      master[L - row_idx - 1] = (int8_t)besta;
    if (tr && tr->col_base) tr->col_base[row_idx] = (int8_t)besta;
    if (tr && tr->col_score) tr->col_score[row_idx] = curr_extension_score;

    for (n = 0; n < N; n++)                              /* :1105-1168 fifth pass with the winner */
    {
      if (tr && tr->row_best) { tr->row_best[(size_t)row_idx * N + n] = -1; tr->row_best_idx[(size_t)row_idx * N + n] = 0; }
      if ((direction && c->right_ext[n]) || (!direction && c->left_ext[n]))
      {
        int best = ramx_oracle_nw_row(direction, row_idx, n, n_align, besta, c->left_pos[n], c->right_pos[n],
                                      c->orient[n], score, (uint64_t)c->lower[n], (uint64_t)c->upper[n],
                                      sequence, &best_idx, p->matrix, p->gapopen, p->gapextn, W);
        if (tr && tr->row_best) { tr->row_best[(size_t)row_idx * N + n] = best; tr->row_best_idx[(size_t)row_idx * N + n] = best_idx; }
        if (best > high[n])                              /* :1140-1150 */
        {
          high[n] = best;
          pos[n] = best_idx;
        }
      }
    }
    /* :1194-1208 fit-preferred rule */
    if (curr_extension_score >= max_extension_score + (abs(max_row - row_idx) * p->minimprovement))
    {
      max_row = row_idx;
      max_extension_score = curr_extension_score;
      for (n = 0; n < N; n++)
      {
        thigh[n] = high[n];
        tpos[n] = pos[n];
      }
    }
    if (abs(row_idx - max_row) >= p->when_to_stop)        /* :1216-1223 */
      break;
  }
  if (tr && tr->rows_executed) *tr->rows_executed = (row_idx < L) ? row_idx + 1 : L;
  if (tr && tr->limit_warning) *tr->limit_warning = (row_idx == L - 1);   /* :1225-1231 */

  for (n = 0; n < N; n++)                                 /* :1234-1247 */
  {
    if (thigh[n] > 0 && tpos[n] >= 0)
    {
      if (direction)
        c->right_len[n] = tpos[n] + 1;
      else
        c->left_len[n] = tpos[n] + 1;
      c->score[n] += thigh[n];
    }
  }
  free(score); free(high); free(pos); free(thigh); free(tpos);
  return max_row + 1;                                     /* :1257 */
}

/* ------------------------------------------------------------------ scoring systems */

static void fill_rows(int32_t *m, const int v[4][4], int nscore)
{
  /* v[cons][base] for A,C,G,T (codes 0..3); N column / row = nscore */
  for (int a = 0; a < 4; a++)
  {
    for (int b = 0; b < 4; b++) m[a * RAMX_ORACLE_MSIZE + b] = v[a][b];
    m[a * RAMX_ORACLE_MSIZE + SYM_N] = nscore;
    m[SYM_N * RAMX_ORACLE_MSIZE + a] = nscore;
  }
  m[SYM_N * RAMX_ORACLE_MSIZE + SYM_N] = nscore;
}

/* score_system.c:384-395 (matrices) / :152-163 (repeatscout): soft-masked codes 4..7 */
static void fill_lower(int32_t *m, int lower_vs_upper)
{
  for (int i = 4; i <= 7; i++)
  {
    for (int j = 4; j <= 7; j++) { m[i * RAMX_ORACLE_MSIZE + j] = -1; m[j * RAMX_ORACLE_MSIZE + i] = -1; }
    for (int j = 0; j <= 3; j++) { m[i * RAMX_ORACLE_MSIZE + j] = lower_vs_upper; m[j * RAMX_ORACLE_MSIZE + i] = lower_vs_upper; }
    m[i * RAMX_ORACLE_MSIZE + SYM_N] = lower_vs_upper;
    m[SYM_N * RAMX_ORACLE_MSIZE + i] = lower_vs_upper;
  }
}

int ramx_oracle_get_matrix(const char *name, int32_t *m, int *gapopen, int *gapextn)
{
  /* values: score_system.c:207-376, index order [cons][seq] with A,C,G,T = 0..3 */
  static const int m14[4][4] = { {9,-18,-10,-21}, {-18,11,-18,-7}, {-7,-18,11,-18}, {-21,-10,-18,9} };
  static const int m18[4][4] = { {9,-15,-8,-18},  {-16,10,-16,-5}, {-5,-16,10,-16}, {-18,-8,-15,9} };
  static const int m20[4][4] = { {9,-15,-8,-17},  {-15,10,-15,-4}, {-4,-15,10,-15}, {-17,-8,-15,9} };
  static const int m25[4][4] = { {8,-13,-6,-15},  {-13,9,-13,-2},  {-2,-13,9,-13},  {-15,-6,-13,8} };
  memset(m, 0, sizeof(int32_t) * RAMX_ORACLE_MSIZE * RAMX_ORACLE_MSIZE);
  if (!strcmp(name, "14p43g")) { fill_rows(m, m14, -1); *gapopen = -33; *gapextn = -7; }
  else if (!strcmp(name, "18p43g")) { fill_rows(m, m18, -1); *gapopen = -30; *gapextn = -6; }
  else if (!strcmp(name, "20p43g")) { fill_rows(m, m20, -1); *gapopen = -28; *gapextn = -5; }
  else if (!strcmp(name, "25p43g")) { fill_rows(m, m25, -1); *gapopen = -25; *gapextn = -5; }
  else return -1;
  fill_lower(m, -1);
  return 0;
}

void ramx_oracle_get_repeatscout_matrix(int match, int mismatch, int gap, int32_t *m, int *gapopen, int *gapextn)
{
  int v[4][4];
  memset(m, 0, sizeof(int32_t) * RAMX_ORACLE_MSIZE * RAMX_ORACLE_MSIZE);
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) v[a][b] = (a == b) ? match : mismatch;
  fill_rows(m, (const int (*)[4])v, mismatch);
  fill_lower(m, mismatch);
  *gapopen = 0;       /* score_system.c:118 */
  *gapextn = gap;     /* score_system.c:119 */
}
