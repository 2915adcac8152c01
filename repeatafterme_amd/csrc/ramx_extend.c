/*
 * ramx_extend.c -- seam 1 of include/ramx.h: the extend_alignment()-compatible entry.
 *
 * Host side of the extension loop (reference ram_extend.c:859-1258): walks the caller's core
 * list, resolves every extendable core into a flank descriptor (start, step, complement, in-bounds
 * interval), hands the flanks to the device layer (ramx_device.hip), and writes the results back
 * into master[] and the cores exactly where the reference does.  No DP arithmetic happens here.
 */
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>

#include "ramx_internal.h"

/* the reference's hot-path globals (ram_extend.c:40,52,61) */
static int g_verbose = 0;
static int g_when_to_stop = 100;
static int g_l = 1;

void ramx_set_runtime(int verbose, int when_to_stop, int l)
{
  g_verbose = verbose;
  g_when_to_stop = when_to_stop;
  g_l = l;
}
int ramx_runtime_verbose(void) { return g_verbose; }
int ramx_runtime_when_to_stop(void) { return g_when_to_stop; }
int ramx_runtime_l(void) { return g_l; }

static ramx_dev *g_dev = NULL;
/* The device keeps the library between calls (the reference's main() runs both directions on one seqLib).  The
 * cache key is (pointer, length, content fingerprint): a caller that rewrites the buffer in place, or whose new
 * library lands at a recycled address with the same length, must never be served the stale device copy.  Libraries
 * above 64 MiB are fingerprinted in 16 MiB chunks on the host's cores (round 4: ~6 ms for 1 GB on 16 threads against 18 ms
 * for sending it again, which is what every call did before); batch mode keeps RAMX_FP_MAX for its concatenated libraries.
 * ramx_invalidate_library() drops the cache explicitly. */
#define RAMX_FP_MAX (64ull << 20)
#define RAMX_FP_CHUNK (16ull << 20)
static const int8_t *g_lib_ptr = NULL;
static uint64_t g_lib_len = 0, g_lib_fp = 0;
static int g_lib_trusted = 0;      /* ramx_preload_library: the caller vouches for the buffer until ramx_invalidate_library() */
/* batch mode: the (pointer, length, fingerprint) of every family whose concatenation the device currently holds */
static const int8_t **g_bl_ptr = NULL;
static uint64_t *g_bl_len = NULL, *g_bl_fp = NULL;
static int g_bl_n = 0;

static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

/* 64-bit content fingerprint, four independent multiply-rotate lanes over 32-byte stripes (memory-bound, ~10 GB/s) */
static uint64_t fingerprint(const int8_t *p, uint64_t n)
{
  const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full;
  uint64_t a = P1 ^ n, b = P2, c = ~P1, d = ~P2, w[4];
  uint64_t i = 0;
  for (; i + 32 <= n; i += 32)
  {
    memcpy(w, p + i, 32);
    a = rotl64(a + w[0] * P2, 31) * P1; b = rotl64(b + w[1] * P2, 31) * P1;
    c = rotl64(c + w[2] * P2, 31) * P1; d = rotl64(d + w[3] * P2, 31) * P1;
  }
  uint64_t h = rotl64(a, 1) + rotl64(b, 7) + rotl64(c, 12) + rotl64(d, 18);
  for (; i < n; i++) h = rotl64(h ^ ((uint64_t)(uint8_t)p[i] * P1), 11) * P2;
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P1; h ^= h >> 32;
  return h ? h : 1;
}

/* the same over a large buffer: chunk fingerprints by worker threads, folded in chunk order (ramx_par.c) */
struct fp_ctx { const int8_t *p; uint64_t n; uint64_t *out; };
static void fp_range(int lo, int hi, void *user)
{
  struct fp_ctx *c = (struct fp_ctx *)user;
  for (int i = lo; i < hi; i++)
  {
    const uint64_t at = (uint64_t)i * RAMX_FP_CHUNK, len = c->n - at < RAMX_FP_CHUNK ? c->n - at : RAMX_FP_CHUNK;
    c->out[i] = fingerprint(c->p + at, len);
  }
}
static uint64_t fingerprint_large(const int8_t *p, uint64_t n)
{
  if (n <= RAMX_FP_MAX) return fingerprint(p, n);
  const uint64_t nchunk = (n + RAMX_FP_CHUNK - 1) / RAMX_FP_CHUNK;
  if (nchunk > 0x7fffffffull) return 0;
  uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nchunk);
  if (h == NULL) return 0;                     /* 0 = not fingerprinted: the caller uploads */
  struct fp_ctx c = { p, n, h };
  ramx_parallel_for((int)nchunk, 1, fp_range, &c);
  const uint64_t r = fingerprint((const int8_t *)h, sizeof(uint64_t) * nchunk) ^ rotl64(n, 17);
  free(h);
  return r ? r : 1;
}

/* the fingerprint of a buffer the device copy is believed to match, taken BESIDE the direction it is needed for (seam 1 joins
 * it before anything is written back; a mismatch repeats the direction on the new content) */
struct fp_bg { const int8_t *p; uint64_t n; uint64_t fp; };
static void *fp_bg_worker(void *arg)
{
  struct fp_bg *j = (struct fp_bg *)arg;
  j->fp = fingerprint_large(j->p, j->n);
  return NULL;
}
#define RAMX_FP_ASYNC_MIN (4ull << 20)      /* below this the fingerprint costs less than a thread */

/* the library the device holds came from a packed twin (ramx_preload_library_packed): the seqLib it belongs to */
static const struct sequenceLibrary *g_packed_owner = NULL;

void ramx_invalidate_library(void)
{
  g_lib_ptr = NULL; g_lib_len = 0; g_lib_fp = 0; g_lib_trusted = 0; g_bl_n = 0; g_packed_owner = NULL;
}

/* the loader calls this when it frees a library: a later library that malloc places at the same address must not be taken for the
 * one the device holds (the packed path is keyed on the owner's pointer alone; the one-byte path also checks length and content) */
void ramx_forget_library_owner(const struct sequenceLibrary *lib)
{
  if (lib != NULL && g_packed_owner == lib) ramx_invalidate_library();
}

int ramx_preload_library_packed(const struct sequenceLibrary *seqLib, const struct ramx_packed_library *pl)
{
  if (!seqLib || !pl) { ramx_set_error("ramx_preload_library_packed: bad argument"); return RAMX_ERR_ARG; }
  ramx_dev *d = ramx_default_device();
  if (!d) return RAMX_ERR_NO_DEVICE;
  ramx_invalidate_library();
  const int rc = ramx_dev_load_library_packed(d, pl);
  if (rc != RAMX_OK) return rc;
  g_packed_owner = seqLib;
  return RAMX_OK;
}

int ramx_preload_library(const int8_t *sequence, uint64_t seq_len)
{
  if (!sequence && seq_len) { ramx_set_error("ramx_preload_library: bad argument"); return RAMX_ERR_ARG; }
  ramx_dev *d = ramx_default_device();
  if (!d) return RAMX_ERR_NO_DEVICE;
  ramx_invalidate_library();
  const int rc = ramx_dev_load_library(d, sequence, seq_len);
  if (rc != RAMX_OK) return rc;
  g_lib_ptr = sequence; g_lib_len = seq_len; g_lib_fp = 0; g_lib_trusted = 1;
  return RAMX_OK;
}

ramx_dev *ramx_default_device(void)
{
  if (!g_dev)
  {
    int ord = 0;
    const char *e = getenv("RAMX_DEVICE");
    if (e) ord = atoi(e);
    if (ramx_dev_create(ord, &g_dev) != RAMX_OK) g_dev = NULL;
  }
  return g_dev;
}

/* ---- -outmat: the reference's per-row trace (ram_extend.c:1122-1132, bnw_extend.c:1027-1044) -------------------------
 * written from what the device reports for every executed row: per band cell which state holds the cell's score, per
 * flank the row's best score and index.  The characters themselves come from the caller's library. */
struct trace_ctx
{
  FILE *out;
  int direction, W, nx;
  const int32_t *core_index;      /* flank -> position of its core in the list (the reference's n) */
  const ramx_flank *flanks;
  const ramx_flat_cores *cores;
  const int8_t *sequence;               /* one byte per base, or NULL: */
  const ramx_packed_library *packed;    /* the packed twin */
};
static FILE *g_trace_file = NULL;

/* ---- -vvvv: the reference's per-row lines (ram_extend.c:992-1090 candidate passes, 1134-1214 after the winner's pass), written
 * from what the device reports for every launch: the four candidate rows of the NEXT row (best cell, its band cell, gap state of
 * the first / last cell) and the winner row's best cell and end-cell gap states.  All sums in `int`, as the reference's. */
struct verbose_ctx
{
  int direction, W, cap, minimp, nx;
  const int32_t *core_index;       /* flank -> position of its core in the list (the reference's n) */
  int *high;                       /* overall_sequence_high_score per flank */
  int32_t *cand_prev;              /* [nx][16]: the candidate rows of the row about to be reported */
  int max_ext, max_row;
  /* -vvvvv (VERBOSE >= 12): every cell of those candidate rows (ram_extend.c:1013-1024) and the sequence around the edge of
   * every core (:1066, report.c printExtensionRegion) */
  int level;
  int32_t *band_prev;              /* [nx][4][2W+1][2] */
  const ramx_flat_cores *cores;
  const int8_t *sequence;          /* one byte per base, or NULL: */
  const ramx_packed_library *packed;
  uint64_t seq_len;
  const int32_t *seq_idx;          /* per core: its record of the library (NULL: the caller came through ramx_extend_flat) */
  const uint64_t *boundaries;
};
/* set by ramx_extend_alignment for the duration of a call: which record a core belongs to, and the records' ends */
static const int32_t *g_edge_seq_idx = NULL;
static const uint64_t *g_edge_boundaries = NULL;

static char trace_num_to_char(int8_t z);
static int edge_code(const struct verbose_ctx *t, int64_t p, int complement)
{
  int c = RAMX_SYM_N;
  if (p >= 0 && (uint64_t)p < t->seq_len)
  {
    if (t->sequence) c = t->sequence[p];
    else if (t->packed) { char b = RAMX_SYM_N; (void)ramx_packed_decode(t->packed, (uint64_t)p, 1, &b); c = (int8_t)b; }
  }
  if (complement && c >= 0 && c < 8) c = (c & 4) | (3 - (c & 3));       /* A<->T, C<->G, case kept (sequence.c:1141-1160) */
  return c;
}

/* The line report.c:20-150 prints for a core at VERBOSE >= 12: ten bases on the core side of the last aligned position, that
 * base, and the bases the extension is about to walk into -- '*' on a record's end, blanks beyond it; minus-strand cores read
 * the other way round and complemented; right extensions put the core first, left extensions last.  (One quirk is part of the
 * format: a left extension of a minus-strand core shows nine bases ahead, not ten.) */
static void print_edge_region(const struct verbose_ctx *t, int n, int row)
{
  const int sidx = t->seq_idx ? t->seq_idx[n] : 0;
  const int64_t lo = (t->boundaries && sidx > 0) ? (int64_t)t->boundaries[sidx - 1] : 0;
  const int64_t up = t->boundaries ? (int64_t)t->boundaries[sidx] : (int64_t)t->seq_len;
  const int minus = t->cores->orient[n] ? 1 : 0;
  /* the walk goes up the library for (right, +) and (left, -), down for the other two */
  const int up_walk = (t->direction != 0) != (minus != 0);
  const int64_t edge = t->direction ? t->cores->right_pos[n] : t->cores->left_pos[n];
  const int64_t last = up_walk ? edge + row + 1 : edge - row - 1;
  char core[16], ext[16];
  int nc = 0, ne = 0;
  /* core side: behind the walk; written in reading order of the printed strand */
  for (int k = 10; k >= 1; k--)
  {
    /* position k steps behind `last` */
    const int64_t p = up_walk ? last - k : last + k;
    char ch;
    if (up_walk) ch = (p == lo) ? '*' : (p < lo) ? ' ' : trace_num_to_char((int8_t)edge_code(t, p, minus));
    else ch = (p == up) ? '*' : (p > up) ? ' ' : trace_num_to_char((int8_t)edge_code(t, p, minus));
    core[nc++] = ch;
  }
  const int ahead = (!t->direction && minus) ? 9 : 10;
  for (int k = 1; k <= ahead; k++)
  {
    const int64_t p = up_walk ? last + k : last - k;
    char ch;
    if (up_walk) ch = (p == up) ? '*' : (p > up) ? ' ' : trace_num_to_char((int8_t)edge_code(t, p, minus));
    else ch = (p == lo) ? '*' : (p < lo) ? ' ' : trace_num_to_char((int8_t)edge_code(t, p, minus));
    ext[ne++] = ch;
  }
  core[nc] = '\0'; ext[ne] = '\0';
  const char mid = trace_num_to_char((int8_t)edge_code(t, last, minus));
  printf("C_EDGE: sIdx=%d, orient=%d, pos=%ld: ", sidx, minus, (long)(last - lo + 1));
  if (t->direction) printf("%s] {%c} %s\n", core, mid, ext);
  else
  {
    /* left extensions: what lies ahead first (furthest base first), the core side behind the bracket (nearest base first) */
    char r1[16], r2[16];
    for (int k = 0; k < ne; k++) r1[k] = ext[ne - 1 - k];
    for (int k = 0; k < nc; k++) r2[k] = core[nc - 1 - k];
    r1[ne] = '\0'; r2[nc] = '\0';
    printf("%s {%c} [%s\n", r1, mid, r2);
  }
}

static void verbose_row(int32_t row, int32_t besta, int32_t n_flanks, const int32_t *best_score, const int32_t *best_idx,
                        const int32_t *gfl, const int32_t *cand, const int32_t *band, void *user)
{
  struct verbose_ctx *t = (struct verbose_ctx *)user;
  static const char base[4] = { 'A', 'C', 'G', 'T' };
  (void)besta; (void)best_idx;
  const int nx = n_flanks < t->nx ? n_flanks : t->nx;
  if (row >= 0)
  {
    int curr = 0, chosen = 0;
    for (int a = 0; a < 4; a++)
    {
      int sum = 0;
      for (int i = 0; i < nx; i++)
      {
        const int32_t *c = t->cand_prev + (size_t)i * 16;
        printf(t->direction ? "RIGHT ROW %d with '%c': n = %d\n" : "LEFT ROW %d with '%c': n = %d\n", row, base[a], t->core_index[i]);
        if (t->level >= 12 && t->band_prev)
        {
          const int Bw = 2 * t->W + 1;
          const int32_t *bp = t->band_prev + ((size_t)i * 4 + a) * Bw * 2;
          printf("    Gap: ");
          for (int j = 0; j < Bw; j++) printf(" %d", bp[2 * j + 1]);
          printf("\n    Sub: ");
          for (int j = 0; j < Bw; j++) printf(" %d", bp[2 * j]);
          printf("\n");
        }
        int b = c[a];
        const int col = row + c[4 + a] - t->W, hi = t->high[i];
        if (b < 0) printf("  best score = %d @ column %d -- max(0,best_score) = 0!, prev best score = %d\n", b, col, hi);
        else printf("  best score = %d @ column %d, prev best score = %d\n", b, col, hi);
        if ((t->direction && c[8 + a] < -279000) || (!t->direction && c[12 + a] < -279000)) printf(" **OUT_OF_SEQ**");
        if (b < 0) b = 0;
        if (b >= hi + t->cap) sum += b;
        else { printf(" **CAPPED** contributing = %d", hi + t->cap); sum += hi + t->cap; }
        printf("\n");
        if (t->level >= 12 && t->cores) print_edge_region(t, t->core_index[i], row);
      }
      printf("  Total Score for '%c' = %d\n", base[a], sum);
      if (sum > curr) { curr = sum; chosen = a; }
    }
    printf("ROW %d complete, '%c' chosen as the consensus. curr_ext_score = %d\n", row, base[chosen], curr);
    int num_out = 0, num_high = 0;
    for (int i = 0; i < nx; i++)
    {
      if ((!t->direction && gfl[2 * i] < -279000) || (t->direction && gfl[2 * i + 1] < -279000)) num_out++;
      if (best_score[i] > t->high[i]) { t->high[i] = best_score[i]; num_high++; }
    }
    float since = 0.0f;
    if (abs(row - t->max_row) > 0) since = ((float)(curr - t->max_ext) / abs(row - t->max_row));
    printf("Alignment Extension: curr_extension_score = %d, max_extension_score = %d @ row %d\n                     "
           "score_per_position_overall = %0.1f, score_per_position_since_max = %0.1f\n"
           "                     total_edges = %d, num_extending = %d, num_out_of_seq = %d\n",
           curr, t->max_ext, t->max_row, ((float)curr / (row + 1)), since, nx, num_high, num_out);
    if (curr >= t->max_ext + (abs(t->max_row - row) * t->minimp))
    {
      printf("                     **This row is now the new max**\n");
      t->max_row = row;
      t->max_ext = curr;
    }
    else
      printf("                     extensions since last max score %d\n", abs(row - t->max_row));
  }
  memcpy(t->cand_prev, cand, sizeof(int32_t) * 16 * (size_t)nx);
  if (t->band_prev && band) memcpy(t->band_prev, band, sizeof(int32_t) * 8 * (size_t)(2 * t->W + 1) * (size_t)nx);
}

static char trace_num_to_char(int8_t z)      /* sequence.c:1091-1111 */
{
  static const char t[8] = { 'A', 'C', 'G', 'T', 'a', 'c', 'g', 't' };
  return (z >= 0 && z < 8) ? t[z] : 'N';
}

static int8_t trace_code(const struct trace_ctx *t, uint64_t p)
{
  if (t->sequence) return t->sequence[p];
  char c = RAMX_SYM_N;
  if (t->packed) (void)ramx_packed_decode(t->packed, p, 1, &c);
  return (int8_t)c;
}

static void trace_row(int32_t row, int32_t besta, const int8_t *codes, const int32_t *best_score, const int32_t *best_idx, void *user)
{
  const struct trace_ctx *t = (const struct trace_ctx *)user;
  const int W = t->W, B = 2 * W + 1;
  char *path = (char *)malloc((size_t)B + 1);
  for (int i = 0; i < t->nx; i++)
  {
    const ramx_flank *f = &t->flanks[i];
    const int n = t->core_index[i];
    for (int j = 0; j < B; j++)
    {
      /* bnw_extend.c:1029-1031: the base as stored (not complemented), 'X' outside the flank */
      const int64_t tt = (int64_t)(j - W) + row;
      const int64_t p = f->start + (int64_t)f->step * tt;
      char base = 'X';
      if (p >= 0 && p >= t->cores->lower[n] && p <= t->cores->upper[n]) base = trace_num_to_char(trace_code(t, (uint64_t)p));
      const int c = codes[(size_t)i * B + j];
      path[j] = c == 0 ? base : (c == 1 ? '-' : (char)tolower((unsigned char)base));
    }
    path[2 * W] = '\0';                              /* ram_extend.c:1123: the last cell is cut off */
    fprintf(t->out, "dir=%s:n=%d:row=%d: %s best:score=%d:offset=%d:cons=%c\n", t->direction ? "right" : "left", n, row, path,
            best_score[i], best_idx[i] - row + W, trace_num_to_char((int8_t)besta));
  }
  free(path);
}

static double wall_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int64_t clamp64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/*
 * Resolve one core into a flank for `direction` (reference bnw_extend.c:778-788, 824-868).
 * With p(t) = start + step * t, t = offset + row, the reference's unsigned index logic is
 * equivalent to "out of bounds iff p < lowerSeqBound || p > upperSeqBound || p < 0"
 * (SURVEY.md App. D rule 1; tests/test_scoring_and_flanks.py re-checks it against the oracle).
 */
static void resolve_flank(int direction, int64_t left_pos, int64_t right_pos, int64_t lower, int64_t upper,
                          int orient, int W, int L, ramx_flank *f)
{
  int64_t start, lo = lower < 0 ? 0 : lower, tlo, thi;
  if (direction)
    start = orient ? right_pos - 1 : right_pos + 1;
  else
    start = orient ? left_pos + 1 : left_pos - 1;
  f->start = start;
  f->step = ((direction != 0) == (orient != 0)) ? -1 : 1;
  f->compl_ = orient ? 1 : 0;
  memset(f->pad_, 0, sizeof(f->pad_));
  if (f->step > 0) { tlo = lo - start; thi = upper - start; }
  else { tlo = start - upper; thi = start - lo; }
  /* only t in [-W, L+W] is ever looked at; clamp so the device can use int32 */
  f->t_lo = (int32_t)clamp64(tlo, -(int64_t)W - 2, (int64_t)L + W + 2);
  f->t_hi = (int32_t)clamp64(thi, -(int64_t)W - 2, (int64_t)L + W + 2);
}

int ramx_resolve_flanks(int direction, const ramx_flat_cores *c, int bandwidth, int L, ramx_flank *flanks,
                        int32_t *core_index)
{
  int nx = 0;
  for (int n = 0; n < c->n; n++)
  {
    /* ram_extend.c:984-985: only cores extendable in this direction take part (but keep index n) */
    if ((direction && c->right_ext[n]) || (!direction && c->left_ext[n]))
    {
      resolve_flank(direction, c->left_pos[n], c->right_pos[n], c->lower[n], c->upper[n], c->orient[n], bandwidth, L,
                    &flanks[nx]);
      core_index[nx++] = n;
    }
  }
  return nx;
}

/* packed != NULL: the device already holds the library (ramx_preload_library_packed); `sequence` is not looked at */
static int extend_flat_impl(int direction, ramx_flat_cores *c, const int8_t *sequence, uint64_t seq_len,
                            int8_t *master, const ramx_params *p, ramx_run_info *info, const ramx_packed_library *packed)
{
  ramx_run_info local;
  if (!info) info = &local;
  memset(info, 0, sizeof(*info));
  if (!c || !p || !master || (c->n > 0 && !sequence && !packed)) { ramx_set_error("ramx_extend_flat: bad argument"); return RAMX_ERR_ARG; }
  ramx_dev *d = ramx_default_device();
  if (!d) return RAMX_ERR_NO_DEVICE;
  const double t0 = wall_ms();
  const int timing = getenv("RAMX_TIMING") != NULL;
  double tph = t0;
#define SEAM1_PHASE(name) do { if (timing) { const double t_ = wall_ms(); fprintf(stderr, "RAMX_TIMING     seam1 %-22s %8.3f ms\n", name, t_ - tph); tph = t_; } } while (0)
  const int N = c->n, W = p->bandwidth, L = p->L;
  int rc;
  int *map = (int *)malloc(sizeof(int) * (N > 0 ? N : 1));
  ramx_flank *fl = (ramx_flank *)malloc(sizeof(ramx_flank) * (N > 0 ? N : 1));
  const int nx = ramx_resolve_flanks(direction, c, W, L, fl, map);
  SEAM1_PHASE("resolve flanks");
  /* the library is shared by both directions: upload once per (pointer, length, content); nothing to upload when no
   * core is extendable in this direction */
  /* Round 4: when pointer and length are those of the device copy, the content check (6-8 ms for 1 GB even on 16 threads) runs in
   * a helper thread BESIDE the direction, which starts on the device copy at once; the thread is joined before anything is
   * written back, and a mismatch (the caller rewrote the buffer in place) uploads the buffer and repeats the direction.  Not
   * with the row traces (-outmat, -vvvv: they print while the direction runs).  RAMX_SYNC_FINGERPRINT=1: check first, as before. */
  int fp_async = 0;
  pthread_t fp_tid;
  struct fp_bg bg;
  memset(&bg, 0, sizeof(bg));
  if (!packed && nx > 0 && !(g_lib_trusted && sequence == g_lib_ptr && seq_len == g_lib_len) && sequence == g_lib_ptr &&
      seq_len == g_lib_len && g_lib_fp != 0 && seq_len >= RAMX_FP_ASYNC_MIN && g_trace_file == NULL && g_verbose < 10 &&
      getenv("RAMX_SYNC_FINGERPRINT") == NULL)
  {
    bg.p = sequence; bg.n = seq_len;
    if (pthread_create(&fp_tid, NULL, fp_bg_worker, &bg) == 0) fp_async = 1;
  }
#define FP_JOIN() do { if (fp_async) { pthread_join(fp_tid, NULL); fp_async = 0; } } while (0)
  if (!fp_async && !packed && nx > 0 && !(g_lib_trusted && sequence == g_lib_ptr && seq_len == g_lib_len))
  {
    const uint64_t fp = fingerprint_large(sequence, seq_len);
    if (sequence != g_lib_ptr || seq_len != g_lib_len || fp == 0 || fp != g_lib_fp)
    {
      ramx_invalidate_library();
      if ((rc = ramx_dev_load_library(d, sequence, seq_len)) != RAMX_OK) { free(map); free(fl); return rc; }
      g_lib_ptr = sequence;
      g_lib_len = seq_len;
      g_lib_fp = fp;
    }
  }
  SEAM1_PHASE("library (hash/upload)");
  int8_t *cons;
  int32_t *th, *tp;
run_again:
  cons = (int8_t *)malloc((size_t)L + 16);
  th = NULL; tp = NULL;
  struct trace_ctx tctx;
  if (g_trace_file != NULL)
  {
    tctx.out = g_trace_file; tctx.direction = direction; tctx.W = W; tctx.nx = nx; tctx.core_index = map; tctx.flanks = fl;
    tctx.cores = c; tctx.sequence = sequence; tctx.packed = packed;
    ramx_dev_set_row_trace(d, trace_row, &tctx);
  }
  /* -vvvv and above: the per-row lines of the reference, one column launch at a time */
  struct verbose_ctx vctx;
  memset(&vctx, 0, sizeof(vctx));
  const int verbose_rows = g_verbose >= 10;
  if (verbose_rows)
  {
    vctx.direction = direction; vctx.W = W; vctx.cap = p->cappenalty; vctx.minimp = p->minimprovement; vctx.nx = nx; vctx.core_index = map;
    vctx.high = (int *)calloc((size_t)(nx > 0 ? nx : 1), sizeof(int));
    vctx.cand_prev = (int32_t *)calloc((size_t)(nx > 0 ? nx : 1) * 16, sizeof(int32_t));
    vctx.max_ext = 0; vctx.max_row = -1;
    vctx.level = g_verbose;
    if (g_verbose >= 12)
    {
      vctx.band_prev = (int32_t *)calloc((size_t)(nx > 0 ? nx : 1) * 8 * (size_t)(2 * W + 1), sizeof(int32_t));
      vctx.cores = c; vctx.sequence = sequence; vctx.packed = packed; vctx.seq_len = packed ? packed->length : seq_len;
      vctx.seq_idx = g_edge_seq_idx; vctx.boundaries = g_edge_boundaries;
      ramx_dev_set_verbose_band(d, 1);
    }
    ramx_dev_set_row_verbose(d, verbose_row, &vctx);
  }
  if (g_trace_file == NULL && !verbose_rows && nx == 0 && getenv("RAMX_NO_HOST_EMPTY") == NULL)
  {
    /* No core is extendable in this direction: every candidate sum of every row is 0, so the vote (ram_extend.c:1064-1086:
     * strict > from 0) gives 'A' with a score of 0 in every row and the fit-preferred rule (:1194-1223) runs on scalars.
     * Nothing to launch (the reference walks its L rows over an empty list the same way). */
    long long max_ext = 0;
    int max_row = -1, rows_done = 0, stopped = 0;
    for (int r = 0; r < L; r++)
    {
      long long dist = (long long)max_row - r;
      if (dist < 0) dist = -dist;
      if (0 >= max_ext + dist * (long long)p->minimprovement) { max_row = r; max_ext = 0; }
      int d2 = r - max_row;
      if (d2 < 0) d2 = -d2;
      stopped = d2 >= p->when_to_stop;
      cons[r] = 0;
      rows_done = r + 1;
      if (stopped) break;
    }
    info->ret = max_row + 1;
    info->rows_executed = rows_done;
    info->limit_warning = (stopped && rows_done - 1 == L - 1) ? 1 : 0;
    info->persistent = 1; info->launches = 0; info->lanes_per_flank = 1; info->n_extendable = 0;     /* (no column launches) */
    info->prep_ms = wall_ms() - t0;
    th = (int32_t *)malloc(sizeof(int32_t));
    tp = (int32_t *)malloc(sizeof(int32_t));
    rc = RAMX_OK;
  }
  else if (g_trace_file == NULL && !verbose_rows && nx > 0 && nx <= ramx_dev_family_route_max(d, p) && L > 0 && W >= 1 && getenv("RAMX_NO_FAMILY_ROUTE") == NULL)
  {
    /* a family that fits one workgroup needs no device-wide barrier: run it as a batch of one (block-local vote) */
    const int npad = (nx + 63) & ~63;
    ramx_flank *pf = (ramx_flank *)malloc(sizeof(ramx_flank) * (size_t)npad);
    memcpy(pf, fl, sizeof(ramx_flank) * (size_t)nx);
    for (int i = nx; i < npad; i++) { memset(&pf[i], 0, sizeof(ramx_flank)); pf[i].t_lo = 1; pf[i].t_hi = 0; pf[i].step = 1; }
    th = (int32_t *)malloc(sizeof(int32_t) * (size_t)npad);
    tp = (int32_t *)malloc(sizeof(int32_t) * (size_t)npad);
    const int32_t first = 0, count = nx;
    info->prep_ms = wall_ms() - t0;
    rc = ramx_dev_run_families(d, pf, npad, &first, &count, 1, p, info, cons, th, tp);
    free(pf);
    if (rc != RAMX_OK) { FP_JOIN(); free(cons); free(th); free(tp); free(map); free(fl); return rc; }
  }
  else
  {
    rc = ramx_dev_begin_direction(d, fl, nx, p);
    info->prep_ms = wall_ms() - t0;
    SEAM1_PHASE("begin (alloc, pack)");
    if (rc == RAMX_OK) rc = ramx_dev_run_direction(d, info);
    SEAM1_PHASE("run direction");
    if (g_trace_file != NULL) ramx_dev_set_row_trace(d, NULL, NULL);
    if (verbose_rows) { ramx_dev_set_row_verbose(d, NULL, NULL); ramx_dev_set_verbose_band(d, 0); free(vctx.high); free(vctx.cand_prev); free(vctx.band_prev); vctx.high = NULL; vctx.cand_prev = NULL; vctx.band_prev = NULL; }
    if (rc != RAMX_OK) { FP_JOIN(); free(cons); free(map); free(fl); return rc; }
    th = (int32_t *)malloc(sizeof(int32_t) * (nx > 0 ? nx : 1));
    tp = (int32_t *)malloc(sizeof(int32_t) * (nx > 0 ? nx : 1));
    rc = ramx_dev_download(d, cons, L + 16, th, tp);
  }
  if (fp_async)
  {
    FP_JOIN();
    if (bg.fp == 0 || bg.fp != g_lib_fp)
    {
      /* the buffer no longer holds what the device copy was made from: nothing has been written back yet -- upload it and run
       * the direction again */
      free(cons); free(th); free(tp);
      ramx_invalidate_library();
      if ((rc = ramx_dev_load_library(d, sequence, seq_len)) != RAMX_OK) { free(map); free(fl); return rc; }
      g_lib_ptr = sequence; g_lib_len = seq_len; g_lib_fp = bg.fp;
      memset(info, 0, sizeof(*info));
      goto run_again;
    }
    SEAM1_PHASE("fingerprint joined");
  }
  if (rc == RAMX_OK)
  {
    /* ram_extend.c:1092-1095 */
    for (int r = 0; r < info->rows_executed; r++)
    {
      if (direction) master[(size_t)L + p->l + r] = cons[r];
      else master[(size_t)L - r - 1] = cons[r];
    }
    /* ram_extend.c:1234-1247 */
    for (int i = 0; i < nx; i++)
    {
      if (th[i] > 0 && tp[i] >= 0)
      {
        const int n = map[i];
        if (direction) c->right_len[n] = tp[i] + 1;
        else c->left_len[n] = tp[i] + 1;
        c->score[n] += th[i];
      }
    }
  }
  SEAM1_PHASE("download + write-back");
#undef SEAM1_PHASE
#undef FP_JOIN
  free(cons); free(th); free(tp); free(map); free(fl);
  return rc == RAMX_OK ? info->ret : rc;
}

int ramx_extend_flat(int direction, ramx_flat_cores *c, const int8_t *sequence, uint64_t seq_len,
                     int8_t *master, const ramx_params *p, ramx_run_info *info)
{
  return extend_flat_impl(direction, c, sequence, seq_len, master, p, info, NULL);
}

int ramx_extend_alignment(int direction, struct coreAlignment *coreAlign, int ****score,
                          struct sequenceLibrary *seqLib, char *master, int BANDWIDTH,
                          int CAPPENALTY, int MINIMPROVEMENT, int L, int N,
                          struct scoringSystem *scoreParams, FILE *pathStringFile)
{
  (void)score;
  g_trace_file = pathStringFile;          /* -outmat: ramx_extend_flat runs the direction row by row and writes the trace */
  if (g_verbose >= 3)   /* ram_extend.c:886-892 */
  {
    if (direction) printf("extend_alignment(right): Called with %d edges\n", N);
    else printf("extend_alignment(left): Called with %d edges\n", N);
  }
  if (g_verbose > 12)
    fprintf(stderr, "RAMExtend(ramx): the per-cell lines compute_nw_row prints above -vvvvv (bnw_extend.c, VERBOSE > 12) are not "
                    "produced by the device path; the band dumps of -vvvvv and the per-row lines of -vvvv are\n");
  if (g_verbose >= 12)
  {
    /* ram_extend.c:949-959: the boundary row of every core, as initialised at :909-946 */
    for (int k = 0; k < N; k++)
    {
      printf("SW Matrix Boundary Conditions ( n = %d ):\n", k);
      for (int st = 0; st < 2; st++)
      {
        printf(st == 0 ? "  GAP: " : "\n  SUB: ");
        for (int o = -BANDWIDTH; o <= BANDWIDTH; o++)
          printf(" %d", o == 0 ? 0 : (o < 0 ? -o : o) * scoreParams->gapextn + scoreParams->gapopen);
      }
      printf("\n");
    }
  }

  const int n = N > 0 ? N : 0;
  int64_t *i64 = (int64_t *)malloc(sizeof(int64_t) * 4 * (n + 1));
  int8_t *i8 = (int8_t *)malloc(3 * (n + 1));
  int32_t *i32 = (int32_t *)calloc(3 * (n + 1), sizeof(int32_t));
  int32_t *sidx = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));      /* -vvvvv: the record of every core (print_edge_region) */
  ramx_flat_cores fc;
  fc.left_pos = i64; fc.right_pos = i64 + n; fc.lower = i64 + 2 * n; fc.upper = i64 + 3 * n;
  fc.orient = i8; fc.left_ext = i8 + n; fc.right_ext = i8 + 2 * n;
  fc.left_len = i32; fc.right_len = i32 + n; fc.score = i32 + 2 * n;
  int k = 0;
  struct coreAlignment *cc;
  /* the reference walks the whole list for the DP but only the first N nodes for the write-back
   * (ram_extend.c:980 vs :1235); score[] is sized for N, so N nodes is all that is meaningful */
  for (cc = coreAlign; cc != NULL && k < n; cc = cc->next, k++)
  {
    ((int64_t *)fc.left_pos)[k] = (int64_t)cc->leftSeqPos;
    ((int64_t *)fc.right_pos)[k] = (int64_t)cc->rightSeqPos;
    ((int64_t *)fc.lower)[k] = (int64_t)cc->lowerSeqBound;
    ((int64_t *)fc.upper)[k] = (int64_t)cc->upperSeqBound;
    ((int8_t *)fc.orient)[k] = cc->orient ? 1 : 0;
    ((int8_t *)fc.left_ext)[k] = cc->leftExtendable ? 1 : 0;
    ((int8_t *)fc.right_ext)[k] = cc->rightExtendable ? 1 : 0;
    fc.left_len[k] = cc->leftExtensionLen;
    fc.right_len[k] = cc->rightExtensionLen;
    fc.score[k] = cc->score;
    sidx[k] = cc->seqIdx;
  }
  fc.n = k;
  g_edge_seq_idx = sidx; g_edge_boundaries = seqLib->boundaries;

  int32_t *mflat = (int32_t *)malloc(sizeof(int32_t) * 100 * 100);
  /* only [0..3][0..7,99] is defined in the reference's matrix (score_system.c:187-395) */
  memset(mflat, 0, sizeof(int32_t) * 100 * 100);
  for (int a = 0; a < 4; a++)
  {
    for (int b = 0; b < 8; b++) mflat[a * 100 + b] = scoreParams->matrix[a][b];
    mflat[a * 100 + RAMX_SYM_N] = scoreParams->matrix[a][RAMX_SYM_N];
  }
  ramx_params p;
  p.bandwidth = BANDWIDTH; p.cappenalty = CAPPENALTY; p.minimprovement = MINIMPROVEMENT; p.L = L;
  p.when_to_stop = g_when_to_stop; p.l = g_l; p.gapopen = scoreParams->gapopen; p.gapextn = scoreParams->gapextn;
  p.matrix = mflat;
  ramx_run_info info;
  /* a library that was loaded packed (ramx_load_sequence_subset_packed: ->sequence is NULL) runs from its packed twin, which
   * goes to the device now unless ramx_preload_library_packed has put it there already */
  const ramx_packed_library *packed = NULL;
  if (seqLib->sequence == NULL && fc.n > 0)
  {
    packed = ramx_packed_of(seqLib);
    if (!packed) { fprintf(stderr, "RAMExtend(ramx): the sequence library holds no bases (sequence == NULL and no packed twin)\n"); exit(1); }
    if (g_packed_owner != seqLib && ramx_preload_library_packed(seqLib, packed) != RAMX_OK)
    {
      fprintf(stderr, "RAMExtend(ramx): device extension failed: %s\n", ramx_last_error());
      exit(1);
    }
  }
  int ret = extend_flat_impl(direction, &fc, (const int8_t *)seqLib->sequence, seqLib->length, (int8_t *)master, &p, &info, packed);
  if (ret < 0)
  {
    /* the reference has no error return on this path: print + exit(1) like its other failures */
    fprintf(stderr, "RAMExtend(ramx): device extension failed: %s\n", ramx_last_error());
    exit(1);
  }
  if (info.overflow32)
    fprintf(stderr, "RAMExtend(ramx): note: a column sum exceeded the int32 range; the reference's int accumulator "
                    "(ram_extend.c:874-878) would have wrapped here, the device keeps int64\n");
  if (info.ret >= 0 && info.rows_executed > 0 && g_verbose >= 3 && info.rows_executed <= L)
  {
    /* ram_extend.c:1216-1223: printed only when the loop broke */
    const int last = info.rows_executed - 1;
    if (abs(last - (info.ret - 1)) >= g_when_to_stop)
      printf("Ending...due to row_idx=%d - max_extension_score_row_idx=%d <= -WHEN_TO_STOP=%d\n", last,
             info.ret - 1, g_when_to_stop);
  }
  if (info.limit_warning)   /* ram_extend.c:1225-1231 */
  {
    if (direction) printf("WARNING: Extended sequence right to the limit ( L=%d ).\n", L);
    else printf("WARNING: Extended sequence left to the limit ( L=%d ).\n", L);
  }
  k = 0;
  for (cc = coreAlign; cc != NULL && k < fc.n; cc = cc->next, k++)
  {
    cc->leftExtensionLen = fc.left_len[k];
    cc->rightExtensionLen = fc.right_len[k];
    cc->score = fc.score[k];
  }
  g_edge_seq_idx = NULL; g_edge_boundaries = NULL;
  free(i64); free(i8); free(i32); free(sidx); free(mflat);
  g_trace_file = NULL;
  return ret;
}


/* batch mode's host passes over every family's library (fingerprints; the copy into the concatenated library): by family, threaded */
struct batch_lib_ctx { const ramx_family *fam; uint64_t *fps; const uint64_t *at_of; int8_t *lib; int do_fp; };
static void batch_lib_range(int lo, int hi, void *user)
{
  const struct batch_lib_ctx *c = (const struct batch_lib_ctx *)user;
  for (int f = lo; f < hi; f++)
  {
    if (c->fps) c->fps[f] = c->do_fp ? fingerprint(c->fam[f].sequence, c->fam[f].seq_len) : 0;
    if (c->lib && c->fam[f].seq_len) memcpy(c->lib + c->at_of[f], c->fam[f].sequence, c->fam[f].seq_len);
  }
}

/*
 * Batch mode: many families, one launch (include/ramx.h).  Libraries are concatenated into one device buffer,
 * every family's flanks are padded to a multiple of 64 with empty flanks, results are written back per family
 * exactly as ramx_extend_flat does (ram_extend.c:1092-1095, 1234-1247).
 */
int ramx_extend_batch(int direction, ramx_family *fam, int32_t F, const ramx_params *p, ramx_run_info *infos)
{
  if (F < 0 || (F && !fam) || !p) { ramx_set_error("ramx_extend_batch: bad argument"); return RAMX_ERR_ARG; }
  ramx_run_info *own = NULL;
  if (!infos) infos = own = (ramx_run_info *)calloc((size_t)(F ? F : 1), sizeof(ramx_run_info));
  ramx_dev *d = ramx_default_device();
  if (!d) { free(own); return RAMX_ERR_NO_DEVICE; }
  const int W = p->bandwidth, L = p->L;
  int rc = RAMX_OK;
  const int timing = getenv("RAMX_TIMING") != NULL;
  double tph = timing ? wall_ms() : 0;
#define BATCH_PHASE(name) do { if (timing) { const double t_ = wall_ms(); fprintf(stderr, "RAMX_TIMING     batch %-26s %8.3f ms\n", name, t_ - tph); tph = t_; } } while (0)
  /* which families can the batch kernel take? */
  /* every band width and gap sign has a family kernel (register-resident for W = 14/20/40 with non-positive
   * penalties, streaming otherwise); only families above one workgroup (512 flanks) go one by one */
  int batchable = W >= 1 && L >= 0;
  uint64_t total_len = 0;
  size_t total_pad = 0;
  int *take = (int *)calloc((size_t)(F ? F : 1), sizeof(int));
  for (int f = 0; f < F; f++)
  {
    int nx = 0;
    for (int n = 0; n < fam[f].cores.n; n++)
      if ((direction && fam[f].cores.right_ext[n]) || (!direction && fam[f].cores.left_ext[n])) nx++;
    take[f] = batchable && nx <= 512;
    if (take[f]) { total_len += fam[f].seq_len; total_pad += (size_t)((nx + 63) / 64) * 64; }
  }
  /* The concatenated library holds EVERY family at a fixed offset (prefix sums of seq_len), so it is the same for both
   * directions: like seam 1, it is built and uploaded again only when a family's (pointer, length) changes. */
  uint64_t *at_of = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(F ? F : 1));
  total_len = 0;
  for (int f = 0; f < F; f++) { at_of[f] = total_len; total_len += fam[f].seq_len; }
  int lib_cached = (g_lib_ptr == (const int8_t *)&g_bl_n) && g_bl_n == F && g_lib_len == total_len && total_len <= RAMX_FP_MAX;
  uint64_t *fps = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(F ? F : 1));
  {
    struct batch_lib_ctx bc = { fam, fps, at_of, NULL, total_len <= RAMX_FP_MAX };
    ramx_parallel_for(F, 8, batch_lib_range, &bc);       /* ~10 GB/s per thread: 100 MB of families cost 10 ms on one */
  }
  for (int f = 0; f < F && lib_cached; f++)
    if (g_bl_ptr[f] != fam[f].sequence || g_bl_len[f] != fam[f].seq_len || g_bl_fp[f] != fps[f]) lib_cached = 0;
  BATCH_PHASE("library fingerprints");
  int8_t *lib = lib_cached ? NULL : (int8_t *)malloc(total_len ? total_len : 1);
  ramx_flank *fl = (ramx_flank *)malloc(sizeof(ramx_flank) * (total_pad ? total_pad : 1));
  int32_t *map = (int32_t *)malloc(sizeof(int32_t) * (total_pad ? total_pad : 1));
  int32_t *first = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F ? F : 1));
  int32_t *count = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F ? F : 1));
  int32_t *fidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F ? F : 1));
  size_t fpos = 0;
  int nb = 0;
  if (lib)
  {
    struct batch_lib_ctx bc = { fam, NULL, at_of, lib, 0 };
    ramx_parallel_for(F, 8, batch_lib_range, &bc);
  }
  BATCH_PHASE("library copy");
  for (int f = 0; f < F; f++)
  {
    if (!take[f]) continue;
    const uint64_t at = at_of[f];
    ramx_flank *tmp = (ramx_flank *)malloc(sizeof(ramx_flank) * (size_t)(fam[f].cores.n > 0 ? fam[f].cores.n : 1));
    int32_t *tmap = (int32_t *)malloc(sizeof(int32_t) * (size_t)(fam[f].cores.n > 0 ? fam[f].cores.n : 1));
    const int nx = ramx_resolve_flanks(direction, &fam[f].cores, W, L, tmp, tmap);
    first[nb] = (int32_t)fpos; count[nb] = nx; fidx[nb] = f;
    for (int i = 0; i < nx; i++) { fl[fpos] = tmp[i]; fl[fpos].start += (int64_t)at; map[fpos] = tmap[i]; fpos++; }
    while (fpos & 63) { memset(&fl[fpos], 0, sizeof(ramx_flank)); fl[fpos].t_lo = 1; fl[fpos].t_hi = 0; fl[fpos].step = 1; map[fpos] = -1; fpos++; }
    free(tmp); free(tmap);
    nb++;
  }
  BATCH_PHASE("resolve flanks");
  if (nb > 0)
  {
    ramx_run_info *binfo = (ramx_run_info *)calloc((size_t)nb, sizeof(ramx_run_info));
    int8_t *cons = (int8_t *)malloc((size_t)nb * (size_t)(L > 0 ? L : 1));
    int32_t *th = (int32_t *)malloc(sizeof(int32_t) * (fpos ? fpos : 1));
    int32_t *tp = (int32_t *)malloc(sizeof(int32_t) * (fpos ? fpos : 1));
    if (!lib_cached)
    {
      g_lib_ptr = NULL; g_lib_len = 0; g_lib_fp = 0; g_lib_trusted = 0; g_packed_owner = NULL; /* whatever library the device held is replaced */
      rc = ramx_dev_load_library(d, lib, total_len);
      if (rc == RAMX_OK)
      {
        g_bl_ptr = (const int8_t **)realloc((void *)g_bl_ptr, sizeof(*g_bl_ptr) * (size_t)(F ? F : 1));
        g_bl_len = (uint64_t *)realloc(g_bl_len, sizeof(*g_bl_len) * (size_t)(F ? F : 1));
        g_bl_fp = (uint64_t *)realloc(g_bl_fp, sizeof(*g_bl_fp) * (size_t)(F ? F : 1));
        for (int f = 0; f < F; f++) { g_bl_ptr[f] = fam[f].sequence; g_bl_len[f] = fam[f].seq_len; g_bl_fp[f] = fps[f]; }
        g_bl_n = F;
        g_lib_ptr = (const int8_t *)&g_bl_n;         /* sentinel: the device holds the batch library described by g_bl_* */
        g_lib_len = total_len;
      }
    }
    BATCH_PHASE("library upload");
    if (rc == RAMX_OK) rc = ramx_dev_run_families(d, fl, (int32_t)fpos, first, count, nb, p, binfo, cons, th, tp);
    BATCH_PHASE("run families (device)");
    if (rc == RAMX_OK)
    {
      for (int b = 0; b < nb; b++)
      {
        const int f = fidx[b];
        ramx_flat_cores *c = &fam[f].cores;
        infos[f] = binfo[b];
        for (int r = 0; r < binfo[b].rows_executed; r++)
        {
          if (direction) fam[f].master[(size_t)L + p->l + r] = cons[(size_t)b * L + r];
          else fam[f].master[(size_t)L - r - 1] = cons[(size_t)b * L + r];
        }
        for (int i = 0; i < count[b]; i++)
        {
          const size_t g = (size_t)first[b] + i;
          if (th[g] > 0 && tp[g] >= 0)
          {
            const int n = map[g];
            if (direction) c->right_len[n] = tp[g] + 1;
            else c->left_len[n] = tp[g] + 1;
            c->score[n] += th[g];
          }
        }
      }
    }
    free(binfo); free(cons); free(th); free(tp);
    BATCH_PHASE("write-back");
  }
  /* families the batch kernel cannot take (other band widths, positive penalties, > 512 flanks): one by one */
  for (int f = 0; f < F && rc == RAMX_OK; f++)
  {
    if (take[f]) continue;
    int r1 = ramx_extend_flat(direction, &fam[f].cores, fam[f].sequence, fam[f].seq_len, fam[f].master, p, &infos[f]);
    if (r1 < 0) rc = r1;
  }
  free(lib); free(at_of); free(fl); free(map); free(first); free(count); free(fidx); free(take); free(own); free(fps);
  return rc;
}

/* bnw_extend.c:87-155 -- kept for link compatibility: the device path never reads it.  The block has the reference's
 * index shape score[2][num_align][2*bandwidth+1][2], so code written against the reference that dereferences it (the
 * VERBOSE >= 12 dump of ram_extend.c:949-959, unit tests) reads zeros instead of faulting -- but every [n] shares ONE
 * row of cells (2 * num_align * (2W+1) separate mallocs are exactly what this library exists to avoid). */
int ****ramx_allocate_score(int num_align, int bandwidth)
{
  if (num_align < 1 || bandwidth < 0) return NULL;
  const int B = 2 * bandwidth + 1;
  int ****score = (int ****)calloc(2, sizeof(int ***));
  int **row = (int **)calloc((size_t)B, sizeof(int *));
  int *cells = (int *)calloc((size_t)B * 2, sizeof(int));
  if (!score || !row || !cells) { free(score); free(row); free(cells); return NULL; }
  for (int j = 0; j < B; j++) row[j] = cells + 2 * j;
  for (int ff = 0; ff < 2; ff++)
  {
    score[ff] = (int ***)calloc((size_t)num_align, sizeof(int **));
    if (!score[ff]) { free(score[0]); free(score); free(row); free(cells); return NULL; }
    for (int n = 0; n < num_align; n++) score[ff][n] = row;
  }
  return score;
}

void ramx_free_score(int num_align, int bandwidth, int ****score)
{
  (void)bandwidth;
  if (!score) return;
  if (num_align >= 1 && score[0] && score[0][0])
  {
    free(score[0][0][0]);     /* the shared cells */
    free(score[0][0]);        /* the shared row */
  }
  free(score[0]); free(score[1]);
  free(score);
}
