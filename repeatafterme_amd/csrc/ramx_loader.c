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
#include <sys/mman.h>
#include <sys/stat.h>
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
  const unsigned char *map;     /* the whole file, mapped read-only */
  uint64_t map_len;
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
  {
    struct stat st;
    if (fstat(fileno(t->f), &st) != 0 || st.st_size <= 0) die255("%s is truncated", path, 0, 0);
    t->map_len = (uint64_t)st.st_size;
    void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(t->f), 0);
    if (m == MAP_FAILED) die255("cannot map %s", path, 0, 0);
    t->map = (const unsigned char *)m;
  }
  return t;
}

static void tb_close(struct tb_file *t)
{
  for (uint32_t i = 0; i < t->seq_count; i++) free(t->index[i].name);
  free(t->index);
  if (t->map) munmap((void *)t->map, (size_t)t->map_len);
  fclose(t->f);
  free(t);
}

struct nblocks { uint32_t count; uint32_t *start, *size; struct nblocks *next_alloc; };

struct window
{
  int flank_start, flank_end, lower_flank_len;
  enum CoreBoundFlag lower_flag, upper_flag;
  uint64_t dna_offset;              /* file offset of the packed DNA of the window's record */
  const struct nblocks *nb;         /* N blocks of that record (shared by the windows of the record) */
};

/* first N block of a record that may reach beyond `pos` (blocks are sorted by start and do not overlap, kentsrc/twoBitNew.c:597-613:
 * a binary search instead of a scan from block 0 for every window -- scaffold-rich assemblies have thousands per record) */
static uint32_t first_nblock(const struct nblocks *nb, long pos)
{
  uint32_t lo = 0, hi = nb->count;
  while (lo < hi)
  {
    const uint32_t mid = lo + (hi - lo) / 2;
    if ((long)nb->start[mid] + (long)nb->size[mid] <= pos) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* decode [start,end) of a record straight into reference base codes.  Thread-safe: reads the mapped file, a
 * 256-entry table turns one packed byte into four codes (2bit: T=0 C=1 A=2 G=3, kentsrc/dnautil.h:23-27). */
static uint32_t g_quad[256];
static void quad_init(void)
{
  static int done = 0;
  if (__atomic_load_n(&done, __ATOMIC_ACQUIRE)) return;
  static const unsigned char val_to_code[4] = { 3, 1, 0, 2 };
  for (int b = 0; b < 256; b++)
  {
    unsigned char q[4];
    for (int k = 0; k < 4; k++) q[k] = val_to_code[(b >> (6 - 2 * k)) & 3];
    memcpy(&g_quad[b], q, 4);
  }
  __atomic_store_n(&done, 1, __ATOMIC_RELEASE);
}

static int tb_decode(const unsigned char *map, uint64_t map_len, uint64_t dna_offset, const struct nblocks *nb, int start, int end, char *out)
{
  const int p0 = start >> 2, p1 = (end + 3) >> 2;
  const size_t nbytes = (size_t)(p1 - p0);
  if (dna_offset + (uint64_t)p0 > map_len || nbytes > map_len - (dna_offset + (uint64_t)p0)) return -1;
  const unsigned char *packed = map + dna_offset + (uint64_t)p0;
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
  for (uint32_t k = first_nblock(nb, start); k < nb->count; k++)
  {
    long s0 = nb->start[k], e0 = s0 + nb->size[k];
    if (s0 >= end) break;
    if (s0 < start) s0 = start;
    if (e0 > end) e0 = end;
    if (s0 < e0) memset(out + (s0 - start), RAMX_SYM_N, (size_t)(e0 - s0));
  }
  return 0;
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

/* ---- record headers, read by worker threads from the mapped file -------------------------------------------------------- */
struct rec_hdr
{
  uint64_t offset;                  /* of the record in the file (from the index) */
  uint32_t size;
  struct nblocks nb;
  uint64_t dna_offset;
  int failed;
};
struct hdr_job { const unsigned char *map; uint64_t map_len; int swapped; struct rec_hdr *rec; int lo, hi; };

/* n bytes at `off` of the mapped file (the mapping replaces a pread per record: 100,000 records are 300,000 system calls) */
static int map_read(const unsigned char *map, uint64_t map_len, void *buf, size_t n, uint64_t off)
{
  if (off > map_len || n > map_len - off) return -1;
  memcpy(buf, map + off, n);
  return 0;
}

/* kentsrc/twoBitNew.c:369-397: dnaSize, nBlockCount, nStarts[], nSizes[], maskBlockCount, maskStarts[], maskSizes[], reserved, DNA */
static void *hdr_worker(void *arg)
{
  struct hdr_job *j = (struct hdr_job *)arg;
  for (int i = j->lo; i < j->hi; i++)
  {
    struct rec_hdr *r = &j->rec[i];
    uint32_t h[2];
    if (map_read(j->map, j->map_len, h, 8, r->offset) != 0) { r->failed = 1; continue; }
    r->size = j->swapped ? bswap32(h[0]) : h[0];
    r->nb.count = j->swapped ? bswap32(h[1]) : h[1];
    const size_t nbn = r->nb.count;
    /* a count the file cannot hold (corrupt or truncated header) is refused before anything is allocated for it */
    if (r->offset + 8 > j->map_len || (uint64_t)8 * nbn > j->map_len - (r->offset + 8)) { r->nb.count = 0; r->failed = 1; continue; }
    r->nb.start = (uint32_t *)malloc(sizeof(uint32_t) * (nbn ? nbn : 1));
    r->nb.size = (uint32_t *)malloc(sizeof(uint32_t) * (nbn ? nbn : 1));
    if (r->nb.start == NULL || r->nb.size == NULL) { r->nb.count = 0; r->failed = 2; continue; }
    if (nbn && (map_read(j->map, j->map_len, r->nb.start, 4 * nbn, r->offset + 8) != 0 ||
                map_read(j->map, j->map_len, r->nb.size, 4 * nbn, r->offset + 8 + 4 * nbn) != 0)) { r->failed = 1; continue; }
    if (j->swapped) for (size_t k = 0; k < nbn; k++) { r->nb.start[k] = bswap32(r->nb.start[k]); r->nb.size[k] = bswap32(r->nb.size[k]); }
    uint32_t mc;
    if (map_read(j->map, j->map_len, &mc, 4, r->offset + 8 + 8 * nbn) != 0) { r->failed = 1; continue; }
    if (j->swapped) mc = bswap32(mc);
    r->dna_offset = r->offset + 8 + 8 * nbn + 4 + 8 * (uint64_t)mc + 4;   /* soft-mask blocks + reserved word: case is discarded */
  }
  return NULL;
}

static int loader_threads(uint64_t work_items, uint64_t per_thread_min)
{
  int nthreads = 1;
  const char *env = getenv("RAMX_LOADER_THREADS");
  if (env) nthreads = atoi(env);
  else if (work_items >= per_thread_min)
  {
    long c = sysconf(_SC_NPROCESSORS_ONLN);
    nthreads = c > 16 ? 16 : (int)c;
  }
  if (nthreads < 1) nthreads = 1;
  return nthreads;
}

/* ---- the windows' packed payload and / or their 1-byte decoding, by worker threads ------------------------------------- */
struct fill_job
{
  const unsigned char *map; uint64_t map_len;
  const struct window *win;
  const uint64_t *at;               /* start of window i in library coordinates (= in lib->sequence) */
  const uint64_t *byte_at;          /* start of window i's packed bytes in `packed` */
  char *sequence;                   /* NULL: no 1-byte decoding */
  unsigned char *packed;            /* NULL: packed payload not kept */
  int lo, hi;                       /* windows [lo, hi) */
  int failed;
};

static void *fill_worker(void *arg)
{
  struct fill_job *j = (struct fill_job *)arg;
  for (int i = j->lo; i < j->hi && !j->failed; i++)
  {
    const struct window *w = &j->win[i];
    if (j->packed)
    {
      /* the window's bytes of the record's packed DNA, as they are in the file (kentsrc/twoBitNew.c:531-594) */
      const int p0 = w->flank_start >> 2, p1 = (w->flank_end + 3) >> 2;
      if (map_read(j->map, j->map_len, j->packed + j->byte_at[i], (size_t)(p1 - p0), w->dna_offset + (uint64_t)p0) != 0) { j->failed = 1; break; }
    }
    if (j->sequence && tb_decode(j->map, j->map_len, w->dna_offset, w->nb, w->flank_start, w->flank_end, j->sequence + j->at[i]) != 0)
      j->failed = 1;
  }
  return NULL;
}

/* registry: libraries of this loader that carry a packed twin (ramx_packed_of) */
#define RAMX_MAX_PACKED 64
static struct { const struct sequenceLibrary *lib; ramx_packed_library *pl; } g_packed[RAMX_MAX_PACKED];
static pthread_mutex_t g_packed_mu = PTHREAD_MUTEX_INITIALIZER;
static unsigned g_packed_gen = 1;       /* bumped whenever a twin is registered or released: invalidates ramx_lib_code's caches */

const ramx_packed_library *ramx_packed_of(const struct sequenceLibrary *lib)
{
  const ramx_packed_library *r = NULL;
  pthread_mutex_lock(&g_packed_mu);
  for (int i = 0; i < RAMX_MAX_PACKED; i++) if (g_packed[i].lib == lib && lib != NULL) { r = g_packed[i].pl; break; }
  pthread_mutex_unlock(&g_packed_mu);
  return r;
}

static void packed_free(ramx_packed_library *pl)
{
  if (!pl) return;
  free((void *)pl->win_start); free((void *)pl->win_byte); free((void *)pl->win_phase); free((void *)pl->bytes);
  free((void *)pl->n_start); free((void *)pl->n_len);
  free(pl);
}

/* mode bit 0: 1-byte lib->sequence (the reference's data model); bit 1: packed twin, registered for ramx_packed_of */
static struct sequenceLibrary *load_subset(const char *twoBitName, const char *rangeBEDName, struct coreAlignment **core_align,
                                           int *num_cores, int max_flanking_bp, int mode)
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

  /* the records the ranges name, in range order (sorted by name: equal names are neighbours); a name that is not in the
   * file ends the run at the first range that uses it, as the reference's twoBitSeqSize does */
  int *rec_of = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
  struct rec_hdr *rec = (struct rec_hdr *)calloc((size_t)(n ? n : 1), sizeof(*rec));
  int nrec = 0;
  for (int i = 0; i < n; i++)
  {
    if (i > 0 && strcmp(order[i]->name, order[i - 1]->name) == 0) { rec_of[i] = nrec - 1; continue; }
    struct tb_index key;
    key.name = order[i]->name;
    const struct tb_index *ix = (const struct tb_index *)bsearch(&key, tb->index, tb->seq_count, sizeof(key), idx_cmp);
    if (!ix)
    {
      fflush(stdout);
      fprintf(stderr, "%s is not in %s\n", order[i]->name, tb->path);
      exit(255);
    }
    rec[nrec].offset = ix->offset;
    rec_of[i] = nrec++;
  }
  {
    int nthreads = loader_threads((uint64_t)nrec, 2048);
    if (nthreads > nrec) nthreads = nrec > 0 ? nrec : 1;
    struct hdr_job *jobs = (struct hdr_job *)calloc((size_t)nthreads, sizeof(*jobs));
    pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof(*tid));
    for (int t = 0; t < nthreads; t++)
    {
      jobs[t].map = tb->map; jobs[t].map_len = tb->map_len; jobs[t].swapped = tb->swapped; jobs[t].rec = rec;
      jobs[t].lo = (int)((long long)nrec * t / nthreads); jobs[t].hi = (int)((long long)nrec * (t + 1) / nthreads);
    }
    for (int t = 1; t < nthreads; t++)
      if (pthread_create(&tid[t], NULL, hdr_worker, &jobs[t]) != 0) { hdr_worker(&jobs[t]); tid[t] = 0; }
    hdr_worker(&jobs[0]);
    for (int t = 1; t < nthreads; t++) if (tid[t]) pthread_join(tid[t], NULL);
    free(jobs); free(tid);
    for (int k = 0; k < nrec; k++)
    {
      if (rec[k].failed == 2) die255("out of memory reading the N blocks of %s", tb->path, 0, 0);
      if (rec[k].failed) die255("%s is truncated", tb->path, 0, 0);
    }
  }
  LD_PHASE("record headers");

  struct window *win = (struct window *)malloc(sizeof(*win) * (size_t)(n ? n : 1));
  uint64_t *at_of = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
  uint64_t *byte_of = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
  uint64_t total = 0, total_bytes = 0, n_runs = 0;
  for (int i = 0; i < n; i++)
  {
    const struct range *s = order[i];
    const struct rec_hdr *r = &rec[rec_of[i]];
    win[i].dna_offset = r->dna_offset;
    win[i].nb = &r->nb;
    at_of[i] = total;
    byte_of[i] = total_bytes;
    plan_window(s, i ? order[i - 1] : NULL, i + 1 < n ? order[i + 1] : NULL, (int)r->size, max_flanking_bp, &win[i]);
    if ((uint32_t)win[i].flank_end > r->size)
      die255("twoBitReadSeqFrag in %s end (%ld) >= seqSize (%ld)", s->name, win[i].flank_end, r->size);
    if (win[i].flank_end - win[i].flank_start < 1)
      die255("twoBitReadSeqFrag in %s start (%ld) >= end (%ld)", s->name, win[i].flank_start, win[i].flank_end);
    total += (uint64_t)(win[i].flank_end - win[i].flank_start);
    total_bytes += (uint64_t)(((win[i].flank_end + 3) >> 2) - (win[i].flank_start >> 2));
    if (mode & 2)                                    /* N runs that touch the window (twoBitNew.c:597-613); only the packed twin keeps them */
      for (uint32_t k = first_nblock(&r->nb, win[i].flank_start); k < r->nb.count; k++)
      {
        const long s0 = r->nb.start[k], e0 = s0 + r->nb.size[k];
        if (s0 >= win[i].flank_end) break;
        if (e0 > win[i].flank_start) n_runs++;
      }
  }
  at_of[n] = total;
  byte_of[n] = total_bytes;
  LD_PHASE("plan windows");

  struct sequenceLibrary *lib = (struct sequenceLibrary *)calloc(1, sizeof(*lib));
  lib->sequence = (mode & 1) ? (char *)malloc(total + 1) : NULL;
  lib->identifiers = (char **)calloc((size_t)n + 1, sizeof(char *));
  lib->boundaries = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
  lib->offsets = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
  struct coreAlignment *cores = (struct coreAlignment *)calloc((size_t)(n ? n : 1), sizeof(*cores));
  ramx_packed_library *pl = NULL;
  unsigned char *packed = NULL;
  if (mode & 2)
  {
    pl = (ramx_packed_library *)calloc(1, sizeof(*pl));
    packed = (unsigned char *)malloc(total_bytes + 16);
    uint8_t *phase = (uint8_t *)malloc((size_t)n + 1);
    uint64_t *ns = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n_runs + 1));
    uint32_t *nl = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n_runs + 1));
    uint64_t q = 0;
    for (int i = 0; i < n; i++)
    {
      phase[i] = (uint8_t)(win[i].flank_start & 3);
      const struct nblocks *nb = win[i].nb;
      for (uint32_t k = first_nblock(nb, win[i].flank_start); k < nb->count; k++)      /* clipped to the window, in library coordinates (sorted: windows are) */
      {
        long s0 = nb->start[k], e0 = s0 + nb->size[k];
        if (s0 >= win[i].flank_end) break;
        if (s0 < win[i].flank_start) s0 = win[i].flank_start;
        if (e0 > win[i].flank_end) e0 = win[i].flank_end;
        if (s0 < e0) { ns[q] = at_of[i] + (uint64_t)(s0 - win[i].flank_start); nl[q] = (uint32_t)(e0 - s0); q++; }
      }
    }
    pl->length = total; pl->n_windows = n; pl->win_start = at_of; pl->win_byte = byte_of; pl->win_phase = phase;
    pl->bytes = packed; pl->n_bytes = total_bytes; pl->n_start = ns; pl->n_len = nl; pl->n_blocks = (int32_t)q;
  }
  /* every window: independent reads of the mapped file and writes, split over the host cores by bases */
  {
    quad_init();
    int nthreads = loader_threads(total, 8u << 20);
    if (nthreads > n) nthreads = n > 0 ? n : 1;
    struct fill_job *jobs = (struct fill_job *)calloc((size_t)nthreads, sizeof(*jobs));
    pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof(*tid));
    int lo = 0;
    for (int t = 0; t < nthreads; t++)
    {
      /* windows up to the t+1-th share of the bases */
      const uint64_t goal = total / (uint64_t)nthreads * (uint64_t)(t + 1);
      int hi = lo;
      while (hi < n && (t == nthreads - 1 || at_of[hi] < goal)) hi++;
      jobs[t].map = tb->map; jobs[t].map_len = tb->map_len; jobs[t].win = win; jobs[t].at = at_of; jobs[t].byte_at = byte_of;
      jobs[t].sequence = lib->sequence; jobs[t].packed = packed; jobs[t].lo = lo; jobs[t].hi = hi; jobs[t].failed = 0;
      lo = hi;
    }
    for (int t = 1; t < nthreads; t++)
      if (pthread_create(&tid[t], NULL, fill_worker, &jobs[t]) != 0) { fill_worker(&jobs[t]); tid[t] = 0; }
    fill_worker(&jobs[0]);
    int failed = jobs[0].failed;
    for (int t = 1; t < nthreads; t++)
    {
      if (tid[t]) pthread_join(tid[t], NULL);
      failed |= jobs[t].failed;
    }
    free(jobs); free(tid);
    if (failed) die255("%s is truncated", tb->path, 0, 0);
  }
  LD_PHASE((mode & 1) ? "decode" : "read packed");
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
  free(ranges); free(order); free(win); free(rec_of);
  for (int k = 0; k < nrec; k++) { free(rec[k].nb.start); free(rec[k].nb.size); }
  free(rec);
  if (pl)
  {
    int put = 0;
    pthread_mutex_lock(&g_packed_mu);
    for (int i = 0; i < RAMX_MAX_PACKED && !put; i++) if (g_packed[i].lib == NULL) { g_packed[i].lib = lib; g_packed[i].pl = pl; put = 1; }
    __atomic_add_fetch(&g_packed_gen, 1, __ATOMIC_RELEASE);
    pthread_mutex_unlock(&g_packed_mu);
    if (!put) { fprintf(stderr, "RAMExtend(ramx): too many packed libraries alive (%d)\n", RAMX_MAX_PACKED); exit(1); }
  }
  else { free(at_of); free(byte_of); }
  tb_close(tb);
  return lib;
}

struct sequenceLibrary *ramx_load_sequence_subset_minimal(const char *twoBitName, const char *rangeBEDName,
                                                          struct coreAlignment **core_align, int *num_cores,
                                                          int max_flanking_bp)
{
  return load_subset(twoBitName, rangeBEDName, core_align, num_cores, max_flanking_bp, 1);
}

/* The same windows, never expanded to one byte per base on the host: lib->sequence is NULL, *packed holds
 * the windows' bytes of the .2bit payload (kentsrc/twoBitNew.c:531-594: four bases per byte, first base in the most
 * significant bits, T C A G = 0 1 2 3) and the runs of N (:597-613) in library coordinates. */
struct sequenceLibrary *ramx_load_sequence_subset_packed(const char *twoBitName, const char *rangeBEDName,
                                                         struct coreAlignment **core_align, int *num_cores,
                                                         int max_flanking_bp, const ramx_packed_library **packed)
{
  struct sequenceLibrary *lib = load_subset(twoBitName, rangeBEDName, core_align, num_cores, max_flanking_bp, 2);
  if (packed) *packed = ramx_packed_of(lib);
  return lib;
}

/* bases [from, from + count) of a packed library as reference base codes (A C G T = 0..3, N = 99) */
int ramx_packed_decode(const ramx_packed_library *pl, uint64_t from, uint64_t count, char *out)
{
  static const char val_to_code[4] = { 3, 1, 0, 2 };     /* T C A G (kentsrc/dnautil.h:23-27) -> sequence.h:7-15 */
  if (!pl || (!out && count) || from > pl->length || count > pl->length - from) return RAMX_ERR_ARG;
  if (count == 0) return RAMX_OK;
  quad_init();
  /* window of `from`: last win_start <= from */
  int lo = 0, hi = pl->n_windows;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pl->win_start[mid] <= from) lo = mid; else hi = mid; }
  int w = lo;
  uint64_t p = from, done = 0;
  while (done < count)
  {
    while (w + 1 < pl->n_windows && pl->win_start[w + 1] <= p) w++;
    const uint64_t wend = pl->win_start[w + 1];
    const unsigned char *b = pl->bytes + pl->win_byte[w];
    uint64_t q = p - pl->win_start[w] + pl->win_phase[w];
    const uint64_t stop = (from + count < wend) ? from + count : wend;
    for (; p < stop && (q & 3); p++, q++, done++) out[done] = val_to_code[(b[q >> 2] >> (6 - 2 * (q & 3))) & 3];
    for (; p + 4 <= stop; p += 4, q += 4, done += 4) memcpy(out + done, &g_quad[b[q >> 2]], 4);     /* four bases per lookup */
    for (; p < stop; p++, q++, done++) out[done] = val_to_code[(b[q >> 2] >> (6 - 2 * (q & 3))) & 3];
  }
  /* runs of N that touch [from, from + count) */
  if (pl->n_blocks > 0)
  {
    int a = 0, z = pl->n_blocks;                     /* first run that ends behind `from` */
    while (a < z) { const int mid = (a + z) >> 1; if (pl->n_start[mid] + pl->n_len[mid] <= from) a = mid + 1; else z = mid; }
    for (int k = a; k < pl->n_blocks && pl->n_start[k] < from + count; k++)
    {
      uint64_t s0 = pl->n_start[k], e0 = s0 + pl->n_len[k];
      if (s0 < from) s0 = from;
      if (e0 > from + count) e0 = from + count;
      if (s0 < e0) memset(out + (s0 - from), RAMX_SYM_N, (size_t)(e0 - s0));
    }
  }
  return RAMX_OK;
}

/* one base of a library of either kind (report and output writers) */
int ramx_lib_code(const struct sequenceLibrary *lib, uint64_t at)
{
  if (lib->sequence) return lib->sequence[at];
  static __thread const struct sequenceLibrary *c_lib = NULL;
  static __thread const ramx_packed_library *c_pl = NULL;
  static __thread unsigned c_gen = 0;
  static __thread int c_w = 0;
  const unsigned gen = __atomic_load_n(&g_packed_gen, __ATOMIC_ACQUIRE);
  if (c_lib != lib || c_gen != gen) { c_pl = ramx_packed_of(lib); c_lib = lib; c_gen = gen; c_w = 0; }
  const ramx_packed_library *pl = c_pl;
  if (!pl || at >= pl->length) return RAMX_SYM_N;
  int w = c_w;
  if (w >= pl->n_windows || at < pl->win_start[w] || at >= pl->win_start[w + 1])
  {
    int lo = 0, hi = pl->n_windows;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pl->win_start[mid] <= at) lo = mid; else hi = mid; }
    c_w = w = lo;
  }
  static const char val_to_code[4] = { 3, 1, 0, 2 };
  const uint64_t q = at - pl->win_start[w] + pl->win_phase[w];
  int code = val_to_code[(pl->bytes[pl->win_byte[w] + (q >> 2)] >> (6 - 2 * (q & 3))) & 3];
  if (pl->n_blocks > 0)
  {
    int a = -1, z = pl->n_blocks;                      /* last run that starts at or before `at` */
    while (z - a > 1) { const int mid = (a + z) >> 1; if (pl->n_start[mid] <= at) a = mid; else z = mid; }
    if (a >= 0 && at < pl->n_start[a] + pl->n_len[a]) code = RAMX_SYM_N;
  }
  return code;
}

void ramx_free_library(struct sequenceLibrary *lib, struct coreAlignment *cores)
{
  if (lib)
  {
    ramx_packed_library *pl = NULL;
    ramx_forget_library_owner(lib);
    pthread_mutex_lock(&g_packed_mu);
    for (int i = 0; i < RAMX_MAX_PACKED; i++) if (g_packed[i].lib == lib) { pl = g_packed[i].pl; g_packed[i].lib = NULL; g_packed[i].pl = NULL; }
    __atomic_add_fetch(&g_packed_gen, 1, __ATOMIC_RELEASE);
    pthread_mutex_unlock(&g_packed_mu);
    packed_free(pl);
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
