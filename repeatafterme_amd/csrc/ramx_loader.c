/*
 * ramx_loader.c -- input surface: BED-6 ranges + UCSC .2bit -> sequenceLibrary + core list.
 *
 * Replaces reference sequence.c:493-976 (loadSequenceSubsetMinimal, readBEDRanges,
 * bedNameStartRevEndCmp) and the parts of the vendored kent library it drives
 * (kentsrc/twoBitNew.c: header/index :232-287, record header :369-397, fragment read :487-640;
 * kentsrc/linefile.c:690-706 line chopping).  Own implementation: one pass to size every window,
 * one allocation, then the reads -- no per-range realloc.
 *
 * Bit-compatible behaviours that are kept on purpose:
 *   - ranges are re-sorted by (name, start, -end) with a stable sort before use;
 *   - a neighbouring core on the same sequence within max_flanking_bp clips the window; because the
 *     reference compares the strand *pointers* (sequence.c:635,662,693,721) the "same strand" test is
 *     never true, so the window stops at the midpoint unless the neighbour's own flag is 0;
 *   - flanks exist only on extendable sides; each range becomes its own library entry;
 *   - bases are upper-cased, A,C,G,T -> 0..3, everything else (N blocks) -> 99.
 */
#include <ctype.h>
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "ramx_internal.h"

/* ------------------------------------------------------------------ errors like kent's errAbort */
static void die255(const char *fmt, const char *a, long b, long c)
{
  fflush(stdout);
  fprintf(stderr, fmt, a, b, c);
  fprintf(stderr, "\n");
  exit(255);   /* kentsrc/errAbort.c: exit(-1) */
}

/* ------------------------------------------------------------------ .2bit reader */
#define TWOBIT_SIG 0x1A412743u
#define TWOBIT_SIG_SWAPPED 0x4327411Au

struct tb_index { char *name; uint64_t offset; };

struct tb_file
{
  FILE *f;
  const char *path;
  int swapped, version;
  uint32_t seq_count;
  struct tb_index *index;       /* sorted by name for bsearch */
  /* header of the most recently used record (ranges arrive sorted by name) */
  const struct tb_index *cur;
  uint32_t size, n_count;
  uint32_t *n_start, *n_size;
  uint64_t dna_offset;
};

static uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }
static uint64_t bswap64(uint64_t v) { return ((uint64_t)bswap32((uint32_t)v) << 32) | bswap32((uint32_t)(v >> 32)); }

static uint32_t rd32(struct tb_file *t)
{
  uint32_t v;
  if (fread(&v, 4, 1, t->f) != 1) die255("%s is truncated", t->path, 0, 0);
  return t->swapped ? bswap32(v) : v;
}
static uint64_t rd64(struct tb_file *t)
{
  uint64_t v;
  if (fread(&v, 8, 1, t->f) != 1) die255("%s is truncated", t->path, 0, 0);
  return t->swapped ? bswap64(v) : v;
}

static int idx_cmp(const void *a, const void *b)
{
  return strcmp(((const struct tb_index *)a)->name, ((const struct tb_index *)b)->name);
}

static struct tb_file *tb_open(const char *path)
{
  struct tb_file *t = (struct tb_file *)calloc(1, sizeof(*t));
  t->path = path;
  t->f = fopen(path, "rb");
  if (!t->f)
  {
    fflush(stdout);
    fprintf(stderr, "mustOpen: Can't open %s to read: %s\n", path, "No such file or directory");
    exit(255);
  }
  uint32_t sig = 0;
  if (fread(&sig, 4, 1, t->f) != 1) sig = 0;
  if (sig == TWOBIT_SIG_SWAPPED) t->swapped = 1;
  else if (sig != TWOBIT_SIG) die255("%s doesn't have a valid twoBitSig", path, 0, 0);
  t->version = (int)rd32(t);
  if (t->version != 0 && t->version != 1)
  {
    fflush(stdout);
    fprintf(stderr, "Can only handle version 0 or version 1 of this file. This is version %d\n", t->version);
    exit(255);
  }
  t->seq_count = rd32(t);
  (void)rd32(t);   /* reserved */
  t->index = (struct tb_index *)calloc(t->seq_count ? t->seq_count : 1, sizeof(struct tb_index));
  for (uint32_t i = 0; i < t->seq_count; i++)
  {
    int len = fgetc(t->f);
    if (len == EOF) die255("%s is truncated", path, 0, 0);
    char *nm = (char *)malloc((size_t)len + 1);
    if (len && fread(nm, 1, (size_t)len, t->f) != (size_t)len) die255("%s is truncated", path, 0, 0);
    nm[len] = 0;
    t->index[i].name = nm;
    t->index[i].offset = (t->version == 1) ? rd64(t) : rd32(t);
  }
  qsort(t->index, t->seq_count, sizeof(struct tb_index), idx_cmp);
  return t;
}

static void tb_close(struct tb_file *t)
{
  for (uint32_t i = 0; i < t->seq_count; i++) free(t->index[i].name);
  free(t->index); free(t->n_start); free(t->n_size);
  fclose(t->f);
  free(t);
}

/* position on a record and cache its header (size, N blocks, where the packed DNA starts) */
static void tb_select(struct tb_file *t, const char *name)
{
  if (t->cur && strcmp(t->cur->name, name) == 0) return;
  struct tb_index key;
  key.name = (char *)name;
  const struct tb_index *ix = (const struct tb_index *)bsearch(&key, t->index, t->seq_count, sizeof(key), idx_cmp);
  if (!ix)
  {
    fflush(stdout);
    fprintf(stderr, "%s is not in %s\n", name, t->path);
    exit(255);
  }
  fseeko(t->f, (off_t)ix->offset, SEEK_SET);
  t->size = rd32(t);
  t->n_count = rd32(t);
  free(t->n_start); free(t->n_size);
  t->n_start = (uint32_t *)malloc(sizeof(uint32_t) * (t->n_count ? t->n_count : 1));
  t->n_size = (uint32_t *)malloc(sizeof(uint32_t) * (t->n_count ? t->n_count : 1));
  for (uint32_t i = 0; i < t->n_count; i++) t->n_start[i] = rd32(t);
  for (uint32_t i = 0; i < t->n_count; i++) t->n_size[i] = rd32(t);
  uint32_t mask_count = rd32(t);
  fseeko(t->f, (off_t)mask_count * 8 + 4, SEEK_CUR);   /* soft-mask blocks + reserved word: case is discarded */
  t->dna_offset = (uint64_t)ftello(t->f);
  t->cur = ix;
}

struct nblocks { uint32_t count; uint32_t *start, *size; struct nblocks *next_alloc; };

struct window
{
  int flank_start, flank_end, lower_flank_len;
  enum CoreBoundFlag lower_flag, upper_flag;
  uint64_t dna_offset;              /* file offset of the packed DNA of the window's record */
  const struct nblocks *nb;         /* N blocks of that record (shared by the windows of the record) */
};

/* decode [start,end) of a record straight into reference base codes.  Thread-safe: pread on the descriptor, a
 * 256-entry table turns one packed byte into four codes (2bit: T=0 C=1 A=2 G=3, kentsrc/dnautil.h:23-27). */
static uint32_t g_quad[256];
static void quad_init(void)
{
  static const unsigned char val_to_code[4] = { 3, 1, 0, 2 };
  for (int b = 0; b < 256; b++)
  {
    unsigned char q[4];
    for (int k = 0; k < 4; k++) q[k] = val_to_code[(b >> (6 - 2 * k)) & 3];
    memcpy(&g_quad[b], q, 4);
  }
}

static int tb_decode(int fd, const char *path, uint64_t dna_offset, const struct nblocks *nb, int start, int end, char *out,
                     unsigned char **scratch, size_t *scratch_cap)
{
  const int p0 = start >> 2, p1 = (end + 3) >> 2;
  const size_t nbytes = (size_t)(p1 - p0);
  if (nbytes > *scratch_cap)
  {
    free(*scratch);
    *scratch_cap = nbytes + (nbytes >> 2) + 64;
    *scratch = (unsigned char *)malloc(*scratch_cap);
  }
  unsigned char *packed = *scratch;
  size_t got = 0;
  while (got < nbytes)
  {
    ssize_t k = pread(fd, packed + got, nbytes - got, (off_t)(dna_offset + (uint64_t)p0 + got));
    if (k <= 0) return -1;
    got += (size_t)k;
  }
  (void)path;
  int i = start;
  /* head: up to the next multiple of four */
  for (; i < end && (i & 3); i++)
  {
    const unsigned char b = packed[(i >> 2) - p0];
    out[i - start] = (char)((g_quad[b] >> (8 * (i & 3))) & 0xff);
  }
  /* body: four bases per table lookup */
  for (; i + 4 <= end; i += 4)
  {
    const uint32_t q = g_quad[packed[(i >> 2) - p0]];
    memcpy(out + (i - start), &q, 4);
  }
  for (; i < end; i++)
  {
    const unsigned char b = packed[(i >> 2) - p0];
    out[i - start] = (char)((g_quad[b] >> (8 * (i & 3))) & 0xff);
  }
  for (uint32_t k = 0; k < nb->count; k++)
  {
    long s0 = nb->start[k], e0 = s0 + nb->size[k];
    if (s0 >= end) break;
    if (s0 < start) s0 = start;
    if (e0 > end) e0 = end;
    if (s0 < e0) memset(out + (s0 - start), RAMX_SYM_N, (size_t)(e0 - s0));
  }
  return 0;
}

struct decode_job
{
  int fd;
  const char *path;
  const struct window *win;
  const uint64_t *at;               /* start of window i in lib->sequence */
  char *sequence;
  int lo, hi;                       /* windows [lo, hi) */
  int failed;
};

static void *decode_worker(void *arg)
{
  struct decode_job *j = (struct decode_job *)arg;
  unsigned char *scratch = NULL;
  size_t cap = 0;
  for (int i = j->lo; i < j->hi && !j->failed; i++)
    if (tb_decode(j->fd, j->path, j->win[i].dna_offset, j->win[i].nb, j->win[i].flank_start, j->win[i].flank_end,
                  j->sequence + j->at[i], &scratch, &cap) != 0)
      j->failed = 1;
  free(scratch);
  return NULL;
}

/* ------------------------------------------------------------------ BED-6 ranges */
struct range
{
  char *name;
  int start, end;
  char *left_flag, *right_flag, *strand;   /* BED name / score / strand fields (kept as text) */
  int order;
};

static char *dupstr(const char *s)
{
  size_t n = strlen(s);
  char *r = (char *)malloc(n + 1);
  memcpy(r, s, n + 1);
  return r;
}

/* sequence.c:942-976 + kentsrc/linefile.c:690-706, common.c chopByChar */
static struct range *read_ranges(const char *bedFile, int *count)
{
  FILE *f = fopen(bedFile, "r");
  if (!f)
  {
    fflush(stdout);
    fprintf(stderr, "Couldn't open %s , %s\n", bedFile, "No such file or directory");
    exit(255);
  }
  size_t cap = 1024, n = 0;
  struct range *r = (struct range *)malloc(cap * sizeof(*r));
  char *line = NULL;
  size_t lcap = 0;
  ssize_t len;
  while ((len = getline(&line, &lcap, f)) >= 0)
  {
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
    if (line[0] == '#' || line[0] == 0) continue;
    char *fields[6];
    int nf = 0;
    char *p = line;
    while (nf < 6)
    {
      fields[nf++] = p;
      char *tab = strchr(p, '\t');
      if (!tab) break;
      *tab = 0;
      p = tab + 1;
    }
    if (nf < 6 || !(fields[5][0] == '+' || fields[5][0] == '-'))
    {
      /* the reference reads fields[5] unconditionally (undefined for short lines) */
      printf("Error: ranges file does not appear to be in the correct format!\n");
      exit(1);
    }
    if (n == cap) { cap *= 2; r = (struct range *)realloc(r, cap * sizeof(*r)); }
    r[n].name = dupstr(fields[0]);
    r[n].start = (int)strtol(fields[1], NULL, 0);
    r[n].end = (int)strtol(fields[2], NULL, 0);
    r[n].left_flag = dupstr(fields[3]);
    r[n].right_flag = dupstr(fields[4]);
    r[n].strand = dupstr(fields[5]);
    r[n].order = (int)n;
    n++;
  }
  free(line);
  fclose(f);
  *count = (int)n;
  return r;
}

/* sequence.c:493-503 */
static int range_cmp(const struct range *a, const struct range *b)
{
  int diff = strcmp(a->name, b->name);
  if (diff == 0) diff = a->start - b->start;
  if (diff == 0) diff = b->end - a->end;
  return diff;
}

/* stable merge sort (the reference goes through glibc qsort, which merges and is stable here) */
static void sort_ranges(struct range **v, struct range **tmp, int n)
{
  if (n < 2) return;
  int h = n / 2;
  sort_ranges(v, tmp, h);
  sort_ranges(v + h, tmp, n - h);
  int i = 0, j = h, k = 0;
  while (i < h && j < n) tmp[k++] = (range_cmp(v[j], v[i]) < 0) ? v[j++] : v[i++];
  while (i < h) tmp[k++] = v[i++];
  while (j < n) tmp[k++] = v[j++];
  memcpy(v, tmp, sizeof(*v) * (size_t)n);
}


/* flank clipping for one range: sequence.c:546-743 */
static void plan_window(const struct range *s, const struct range *prev, const struct range *next,
                        int seq_size, int max_flanking_bp, struct window *w)
{
  const int minus = strcmp(s->strand, "-") == 0;
  const int left_ext = atoi(s->left_flag) == 1, right_ext = atoi(s->right_flag) == 1;
  int prev_dist = 0, next_dist = 0;
  const struct range *pc = NULL, *nc = NULL;
  if (prev && strcmp(s->name, prev->name) == 0)
  {
    if (s->start < prev->end)
      printf("WARNING: core sequences overlap  %s:%d-%d and previous %s:%d-%d\n", s->name, s->start, s->end,
             prev->name, prev->start, prev->end);
    else
      prev_dist = s->start - prev->end;
    pc = prev;
  }
  if (next && strcmp(s->name, next->name) == 0)
  {
    if (next->start < s->end)
      printf("WARNING: core sequences overlap  %s:%d-%d and next %s:%d-%d\n", s->name, s->start, s->end,
             next->name, next->start, next->end);
    else
      next_dist = next->start - s->end;
    nc = next;
  }
  w->flank_start = s->start;
  w->flank_end = s->end;
  w->lower_flank_len = 0;
  w->lower_flag = L_BOUNDARY;
  w->upper_flag = L_BOUNDARY;
  /* which of this core's flags governs the high-coordinate side and which the low one */
  const int up_ext = minus ? left_ext : right_ext;
  const int lo_ext = minus ? right_ext : left_ext;
  if (up_ext)
  {
    if (nc && next_dist <= max_flanking_bp)
    {
      /* neighbour's flag on the facing side; the reference's strand test is a pointer compare -> false */
      const int n_flag = minus ? atoi(nc->left_flag) : atoi(nc->right_flag);
      const int d = (n_flag == 0) ? next_dist : next_dist / 2;
      w->flank_end = s->end + d;
      w->upper_flag = CORE_BOUNDARY;
    }
    else if (s->end + max_flanking_bp < seq_size)
    {
      w->flank_end = s->end + max_flanking_bp;
      w->upper_flag = L_BOUNDARY;
    }
    else
    {
      w->flank_end = seq_size;
      w->upper_flag = SEQ_BOUNDARY;
    }
  }
  if (lo_ext)
  {
    if (pc && prev_dist <= max_flanking_bp)
    {
      const int p_flag = minus ? atoi(pc->right_flag) : atoi(pc->left_flag);
      const int d = (p_flag == 0) ? prev_dist : prev_dist / 2;
      w->flank_start = s->start - d;
      w->lower_flank_len = d;
      w->lower_flag = CORE_BOUNDARY;
    }
    else if (max_flanking_bp < s->start)
    {
      w->flank_start = s->start - max_flanking_bp;
      w->lower_flank_len = max_flanking_bp;
      w->lower_flag = L_BOUNDARY;
    }
    else
    {
      w->flank_start = 0;
      w->lower_flank_len = s->start;
      w->lower_flag = SEQ_BOUNDARY;
    }
  }
}

static double ld_now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e3 + 1e-6 * (double)ts.tv_nsec;
}
#define LD_PHASE(name) do { if (timing) { const double t_ = ld_now(); fprintf(stderr, "RAMX_TIMING   loader: %-14s %9.3f ms\n", name, t_ - t_last); t_last = t_; } } while (0)

struct sequenceLibrary *ramx_load_sequence_subset_minimal(const char *twoBitName, const char *rangeBEDName,
                                                          struct coreAlignment **core_align, int *num_cores,
                                                          int max_flanking_bp)
{
  const int timing = getenv("RAMX_TIMING") != NULL;
  double t_last = timing ? ld_now() : 0;
  int n = 0;
  struct range *ranges = read_ranges(rangeBEDName, &n);
  LD_PHASE("read ranges");
  struct tb_file *tb = tb_open(twoBitName);
  LD_PHASE("2bit index");
  struct range **order = (struct range **)malloc(sizeof(*order) * (size_t)(n ? n : 1));
  struct range **tmp = (struct range **)malloc(sizeof(*tmp) * (size_t)(n ? n : 1));
  for (int i = 0; i < n; i++) order[i] = &ranges[i];
  sort_ranges(order, tmp, n);
  free(tmp);
  LD_PHASE("sort");

  struct window *win = (struct window *)malloc(sizeof(*win) * (size_t)(n ? n : 1));
  uint64_t *at_of = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
  struct nblocks *nb_list = NULL;
  const struct tb_index *nb_for = NULL;
  uint64_t total = 0;
  for (int i = 0; i < n; i++)
  {
    const struct range *s = order[i];
    tb_select(tb, s->name);
    if (nb_for != tb->cur)          /* the N blocks of this record, kept for the decoding threads */
    {
      struct nblocks *nb = (struct nblocks *)malloc(sizeof(*nb));
      nb->count = tb->n_count;
      nb->start = (uint32_t *)malloc(sizeof(uint32_t) * (tb->n_count ? tb->n_count : 1));
      nb->size = (uint32_t *)malloc(sizeof(uint32_t) * (tb->n_count ? tb->n_count : 1));
      memcpy(nb->start, tb->n_start, sizeof(uint32_t) * tb->n_count);
      memcpy(nb->size, tb->n_size, sizeof(uint32_t) * tb->n_count);
      nb->next_alloc = nb_list;
      nb_list = nb;
      nb_for = tb->cur;
    }
    win[i].dna_offset = tb->dna_offset;
    win[i].nb = nb_list;
    at_of[i] = total;
    plan_window(s, i ? order[i - 1] : NULL, i + 1 < n ? order[i + 1] : NULL, (int)tb->size, max_flanking_bp, &win[i]);
    if ((uint32_t)win[i].flank_end > tb->size)
      die255("twoBitReadSeqFrag in %s end (%ld) >= seqSize (%ld)", s->name, win[i].flank_end, tb->size);
    if (win[i].flank_end - win[i].flank_start < 1)
      die255("twoBitReadSeqFrag in %s start (%ld) >= end (%ld)", s->name, win[i].flank_start, win[i].flank_end);
    total += (uint64_t)(win[i].flank_end - win[i].flank_start);
  }

  LD_PHASE("plan windows");
  struct sequenceLibrary *lib = (struct sequenceLibrary *)calloc(1, sizeof(*lib));
  lib->sequence = (char *)malloc(total + 1);
  lib->identifiers = (char **)calloc((size_t)n + 1, sizeof(char *));
  lib->boundaries = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
  lib->offsets = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
  struct coreAlignment *cores = (struct coreAlignment *)calloc((size_t)(n ? n : 1), sizeof(*cores));
  /* decode every window: independent reads (pread) and writes, split over the host cores by bases */
  {
    quad_init();
    int nthreads = 1;
    const char *env = getenv("RAMX_LOADER_THREADS");
    if (env) nthreads = atoi(env);
    else if (total >= (8u << 20))
    {
      long c = sysconf(_SC_NPROCESSORS_ONLN);
      nthreads = c > 16 ? 16 : (int)c;
    }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > n) nthreads = n > 0 ? n : 1;
    struct decode_job *jobs = (struct decode_job *)calloc((size_t)nthreads, sizeof(*jobs));
    pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof(*tid));
    int lo = 0;
    for (int t = 0; t < nthreads; t++)
    {
      /* windows up to the t+1-th share of the bases */
      const uint64_t goal = total / (uint64_t)nthreads * (uint64_t)(t + 1);
      int hi = lo;
      while (hi < n && (t == nthreads - 1 || at_of[hi] < goal)) hi++;
      jobs[t].fd = fileno(tb->f); jobs[t].path = tb->path; jobs[t].win = win; jobs[t].at = at_of;
      jobs[t].sequence = lib->sequence; jobs[t].lo = lo; jobs[t].hi = hi; jobs[t].failed = 0;
      lo = hi;
    }
    for (int t = 1; t < nthreads; t++)
      if (pthread_create(&tid[t], NULL, decode_worker, &jobs[t]) != 0) { decode_worker(&jobs[t]); tid[t] = 0; }
    decode_worker(&jobs[0]);
    int failed = jobs[0].failed;
    for (int t = 1; t < nthreads; t++)
    {
      if (tid[t]) pthread_join(tid[t], NULL);
      failed |= jobs[t].failed;
    }
    free(jobs); free(tid);
    if (failed) die255("%s is truncated", tb->path, 0, 0);
  }
  LD_PHASE("decode");
  uint64_t at = 0;
  for (int i = 0; i < n; i++)
  {
    const struct range *s = order[i];
    const uint64_t size = (uint64_t)(win[i].flank_end - win[i].flank_start);
    lib->identifiers[i] = dupstr(s->name);
    lib->boundaries[i] = at + size;                 /* cumulative end, 0-terminated list */
    lib->offsets[i] = (uint64_t)win[i].flank_start;
    struct coreAlignment *c = &cores[i];
    c->next = (i + 1 < n) ? &cores[i + 1] : NULL;
    c->seqIdx = i;
    c->lowerSeqBound = at;                          /* sequence.c:847-850 */
    c->upperSeqBound = at + size - 1;
    c->lowerSeqBoundFlag = win[i].lower_flag;
    c->upperSeqBoundFlag = win[i].upper_flag;
    c->leftExtendable = (atoi(s->left_flag) == 1) ? 1 : 0;
    c->rightExtendable = (atoi(s->right_flag) == 1) ? 1 : 0;
    if (strcmp(s->strand, "-") == 0)                /* sequence.c:886-897 */
    {
      c->orient = 1;
      c->rightSeqPos = at + (uint64_t)win[i].lower_flank_len;
      c->leftSeqPos = c->rightSeqPos + (uint64_t)(s->end - s->start) - 1;
    }
    else
    {
      c->orient = 0;
      c->leftSeqPos = at + (uint64_t)win[i].lower_flank_len;
      c->rightSeqPos = c->leftSeqPos + (uint64_t)(s->end - s->start) - 1;
    }
    at += size;
  }
  lib->length = total;
  lib->count = n;
  *core_align = n ? cores : NULL;
  if (!n) free(cores);
  *num_cores = n;

  for (int i = 0; i < n; i++) { free(ranges[i].name); free(ranges[i].left_flag); free(ranges[i].right_flag); free(ranges[i].strand); }
  LD_PHASE("cores");
  free(ranges); free(order); free(win); free(at_of);
  while (nb_list) { struct nblocks *nx = nb_list->next_alloc; free(nb_list->start); free(nb_list->size); free(nb_list); nb_list = nx; }
  tb_close(tb);
  return lib;
}

void ramx_free_library(struct sequenceLibrary *lib, struct coreAlignment *cores)
{
  if (lib)
  {
    for (int i = 0; i < lib->count; i++) free(lib->identifiers[i]);
    free(lib->identifiers); free(lib->boundaries); free(lib->offsets); free(lib->sequence);
    free(lib);
  }
  free(cores);   /* one block, see above */
}

/*
 * Overlap avoidance between the right and the left pass: reference ram_extend.c:445-499.
 * Same visiting order and the same printed lines, but cores are bucketed by identifier first so the
 * cost is O(sum of bucket sizes squared) instead of O(N^2) strcmp's.
 */
struct ov_ent { const char *ident; int idx; };
static int ov_cmp(const void *a, const void *b)
{
  const struct ov_ent *x = (const struct ov_ent *)a, *y = (const struct ov_ent *)b;
  int d = strcmp(x->ident, y->ident);
  return d ? d : (x->idx - y->idx);
}

void ramx_overlap_avoidance(struct coreAlignment *coreAlign, struct sequenceLibrary *seqLib)
{
  int n = 0;
  struct coreAlignment *c;
  for (c = coreAlign; c; c = c->next) n++;
  if (!n) return;
  struct coreAlignment **node = (struct coreAlignment **)malloc(sizeof(*node) * (size_t)n);
  struct ov_ent *ent = (struct ov_ent *)malloc(sizeof(*ent) * (size_t)n);
  int *bucket_lo = (int *)malloc(sizeof(int) * (size_t)n), *bucket_hi = (int *)malloc(sizeof(int) * (size_t)n);
  int k = 0;
  for (c = coreAlign; c; c = c->next, k++)
  {
    node[k] = c;
    ent[k].ident = seqLib->identifiers[c->seqIdx];
    ent[k].idx = k;
  }
  qsort(ent, (size_t)n, sizeof(*ent), ov_cmp);
  for (int i = 0; i < n;)
  {
    int j = i;
    while (j < n && strcmp(ent[j].ident, ent[i].ident) == 0) j++;
    for (int q = i; q < j; q++) { bucket_lo[ent[q].idx] = i; bucket_hi[ent[q].idx] = j; }
    i = j;
  }
  for (int si = 0; si < n; si++)          /* outer loop in list order, as the reference */
  {
    struct coreAlignment *s = node[si];
    const int s_idx = s->seqIdx;
    uint64_t s_lower = s_idx > 0 ? seqLib->boundaries[s_idx - 1] : 0;
    uint64_t extended_pos = s->orient ? (s->rightSeqPos - (uint64_t)s->rightExtensionLen)
                                      : (s->rightSeqPos + (uint64_t)s->rightExtensionLen);
    uint64_t g = seqLib->offsets[s_idx] + (extended_pos - s_lower + 1);
    for (int q = bucket_lo[si]; q < bucket_hi[si]; q++)   /* same identifier, ascending list order */
    {
      struct coreAlignment *r = node[ent[q].idx];
      const int r_idx = r->seqIdx;
      uint64_t r_lower = r_idx > 0 ? seqLib->boundaries[r_idx - 1] : 0;
      if (g > seqLib->offsets[r_idx])
      {
        uint64_t p = r_lower + (g - seqLib->offsets[r_idx]);
        if (r->orient)
        {
          if (p >= r->leftSeqPos && p <= r->upperSeqBound)
          {
            printf("OVERLAP AVOIDANCE: seqid %d extended to %ld, limits seqid %d with existing upper_bound = %ld because it's pos_in_r=%ld\n",
                   s_idx, (long)g, r_idx, (long)r->upperSeqBound, (long)p);
            r->upperSeqBound = p;
            r->upperSeqBoundFlag = EXT_BOUNDARY;
          }
        }
        else
        {
          if (p >= r->lowerSeqBound && p <= r->leftSeqPos)
          {
            printf("OVERLAP AVOIDANCE: seqid %d extended to %ld, limits seqid %d with existing lower_bound = %ld because it's pos_in_r=%ld\n",
                   s_idx, (long)g, r_idx, (long)r->lowerSeqBound, (long)p);
            r->lowerSeqBound = p;
            r->lowerSeqBoundFlag = EXT_BOUNDARY;
          }
        }
      }
    }
  }
  free(node); free(ent); free(bucket_lo); free(bucket_hi);
}
