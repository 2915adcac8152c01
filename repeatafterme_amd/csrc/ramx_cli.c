/*
 * ramx_cli.c -- the RAMExtend command line (reference ram_extend.c:123-826 main/usage/
 * print_parameters, cmd_line_opts.c).  Same argv semantics (prefix matching, per-matrix
 * defaults), same stdout / -cons / -outtsv / -outfa bytes; the two extend_alignment calls go to
 * the device path (ramx_extend_alignment).
 */
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "ramx_internal.h"

static const char *k_version = "0.0.7";          /* interface level of the reference this mirrors */
static const char *k_build = "ramx-gfx950";
static const char *k_build_date = "20261003";

/* ------------------------------------------------------------------ argv scanning
 * cmd_line_opts.c:9-107: the FIRST argv entry that begins with '-' and of which `text` is a
 * prefix wins; its value is the following argv entry. */
static int opt_find(int argc, char **argv, const char *text)
{
  const size_t n = strlen(text);
  for (int i = 0; i < argc; i++)
    if (argv[i][0] == '-' && strncmp(text, argv[i], n) == 0) return i;
  return -1;
}
static const char *opt_value(int argc, char **argv, int i) { return (i + 1 < argc) ? argv[i + 1] : ""; }
static int opt_int(int argc, char **argv, const char *text, int *res)
{
  int i = opt_find(argc, argv, text);
  if (i < 0) return 0;
  *res = atoi(opt_value(argc, argv, i));
  return 1;
}
static int opt_bool(int argc, char **argv, const char *text) { return opt_find(argc, argv, text) >= 0; }
static int opt_string(int argc, char **argv, const char *text, const char **res)
{
  int i = opt_find(argc, argv, text);
  if (i < 0) return 0;
  *res = (i + 1 < argc) ? argv[i + 1] : NULL;
  return 1;
}

static char code_to_char(int z)
{
  static const char t[8] = { 'A', 'C', 'G', 'T', 'a', 'c', 'g', 't' };
  return (z >= 0 && z < 8) ? t[z] : 'N';
}
static int code_compl(int c)
{
  if (c >= 0 && c <= 3) return 3 - c;
  if (c >= 4 && c <= 7) return 4 + (7 - c);
  return RAMX_SYM_N;
}

static void usage(void)
{
  printf("RAMExtend Version %s - build %s date %s\n", k_version, k_build, k_build_date);
  fputs(
    "\n"
    "  Perform a multiple sequence alignment (MSA) extension given\n"
    "  an existing core MSA and the flanking sequences.  The core\n"
    "  may be a set of word matches, a previously development MSA,\n"
    "  or the the result of any process that defines a core set of\n"
    "  sequence relationships.  The only requirement is that the sequences are\n"
    "  aligned to nearly the same point along one or both of the core\n"
    "  edges.  The extension process is a form of anchored alignment\n"
    "  where the details of the core alignment are considered fixed.\n"
    "\n"
    "Usage: \n"
    "  RAMExtend -twobit <seq.2bit> -ranges <ranges.tsv> [opts]\n"
    "\n"
    "--------------------------------------------------------------------------------------------\n"
    "     -cons <seq.fa>        # Save the left/right consensus sequences to a FASTA file.\n"
    "     -outtsv <final.tsv>   # Save the final sequence ranges to a TSV file.\n"
    "     -outfa <seq.fa>       # Save all the final sequences ( original range + extension )\n"
    "                           #   to a file.\n"
    "     -addflanking <num>    # Include an additional<num> bp of sequence to each sequence\n"
    "                           #   when using the -outfa option.\n"
    "     -L <num>              # Size of region to extend left or right (10000). \n"
    "     -minimprovement <num> # Amount that a the alignment score needs to improve each step\n"
    "                           #   to be considered progress (original: 3, nucleotide matrix: 27).\n"
    "     -maxoccurrences <num> # Cap on the number of sequences to align (10,000). \n"
    "     -cappenalty <num>     # Cap on penalty for exiting alignment of a sequence \n"
    "                           #   (original: -20, nucleotide matrix:-90).\n"
    "     -stopafter <num>      # Stop the alignment after this number of no-progress columns (100).\n"
    "     -minlength <num>      # Minimum required length for a sequence to be reported (50).\n"
    "     -outmat <file>        # Dump the dp matrix paths to a file for debugging.\n"
    "     -version              # Print out one-line version information\n"
    "     -v[v[v[v]]]           # How verbose do you want it to be?  -vvvv is super-verbose.\n"
    "\n"
    "   New Scoring System Options:\n"
    "     -matrix 14p43g |      # There are several internally-coded matrices available.\n"
    "             18p43g |      #   The DNA substitution matrices are borrowed from RepeatMasker\n"
    "             20p43g |      #   and are tuned for a 43% GC background.  The four encoded\n"
    "             25p43g        #   matrices are are divergence tuned, ranging from 14% to 25%.\n"
    "                           #   For example the '14p43g' matrix is tuned for 14% divergence\n"
    "                           #   43% GC background.  Each matrix has it's own default gap_open,\n"
    "                           #   gap_extension parameters.  These may be overrided by using the\n"
    "                           #   -gapopen and -gapextn options.\n"
    "     -gapopen <num>        # Affine gap open penalty to override per-matrix default\n"
    "     -gapextn <num>        # Affine gap extension penalty to override per-matrix default\n"
    "\n"
    "     -bandwidth <num>      # The maximum number of unbalanced gaps allowed (14)\n"
    "                           #   Half the bandwidth of the banded Smith-Waterman\n"
    "                           #   algorithm.  The default allows for at most 14bp\n"
    "                           #   of unbalanced insertions/deletions in any aligned\n"
    "                           #   sequence.  The full bandwidth is 2*bandwidth+1.\n"
    " or\n"
    "   Original RepeatScout Scoring System:\n"
    "     -matrix repeatscout   # The original RepeatScout scoring system is enabled if this\n"
    "                           #   option is set.  This uses a simple match/mismatch penalty\n"
    "                           #   and a linear gap model.  The original RepeatScout values\n"
    "                           #   for these parameters may be overriden with the following\n"
    "                           #   additional options:\n"
    "     -match <num>          # If '-matrix repeatscout' used, apply this reward for match\n"
    "                           #   (default +1)  \n"
    "     -mismatch <num>       # If '-matrix repeatscout' used, apply this penalty for a mismatch\n"
    "                           #   (default: -1) \n"
    "     -gap <num>            # If '-matrix repeatscout' used, apply this penalty for a gap\n"
    "                           #   (default: -5)\n"
    "\n"
    "Ranges:\n\n"
    "   Ranges are supplied in the form of a modified BED-6 format:\n\n"
    "      field-1:chrom     : sequence identifier\n"
    "      field-2:chromStart: lower aligned position ( 0 based )\n"
    "      field-3:chromEnd  : upper aligned position ( 0 based, half open )\n"
    "      field-4:left-ext  : left extendable flag ( 0 = no, 1 = yes ) [BED-6 'name' field]\n"
    "      field-5:right-ext : right extendable flag ( 0 = no, 1 = yes ) [BED-6 'score' field]\n"
    "      field-6:strand    : strand ( '+' = forward, '-' = reverse )\n"
    "\n"
    "   The fields are tab separated. Coordinates are zero-based half\n"
    "   open.\n"
    "\n"
    "   Left/Right extension flags are used to turn off extension for sequences that\n"
    "   do not reach the edge of the core alignment (e.g. fragments aligning in the\n"
    "   center of the core MSA, or one or the other edge only).  These sequences are\n"
    "   less likely to include related flanking sequence and weigh down the extension\n"
    "   score uneccesarily.  For these sequences it is desirable to flag one edge or\n"
    "   the other as 'unextendable'.  The 'left'/'right' designations refer to the\n"
    "   edges of the core MSA alignment. The 'left' flag may refer to the start or\n"
    "   the end coordinate depending on the state of the orientation flag\n"
    "   ('+' left=start, '-' left=end etc.).\n", stdout);
  exit(1);
}

/* one FASTA record body, 80-column wrap whose phase is keyed on masterstart for BOTH records
 * (reference ram_extend.c:518-540 / :561-584) */
static void write_wrapped(FILE *fp, const char *master, uint64_t from, uint64_t to, uint64_t masterstart)
{
  uint64_t x;
  for (x = from; x < to; x++)
  {
    fputc(code_to_char(master[x]), fp);
    if ((x - masterstart) % 80 == 79) fputc('\n', fp);
  }
  if ((x - masterstart) % 80 > 0) fputc('\n', fp);
}


/* RAMX_TIMING=1: phase timings on stderr (stdout stays byte-compatible) */
static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static int g_timing = 0;
static double g_t_last = 0, g_t_main = 0;
static void phase_done(const char *what)
{
  if (!g_timing) return;
  fflush(stdout);
  const double t = now_s();
  fprintf(stderr, "RAMX_TIMING %-18s %10.3f ms   (at %8.3f ms after main)\n", what, (t - g_t_last) * 1e3, (t - g_t_main) * 1e3);
  g_t_last = t;
}
static void timing_at_exit(void) { if (g_timing) { const double t = now_s(); fprintf(stderr, "RAMX_TIMING %-18s %10.3f ms   (at %8.3f ms after main)\n", "first atexit handler", (t - g_t_last) * 1e3, (t - g_t_main) * 1e3); } }

/* The first HIP call of a process costs 0.1-0.3 s (runtime start-up, device context).  It does not depend on the
 * input, so it runs in a helper thread while the loader reads the .2bit; joined before the first extension. */
static pthread_t g_warm_thread;
static int g_warm_started = 0;
/* the helper thread starts the HIP runtime while the loader works and then, as soon as the main thread hands it the
 * library, uploads it while the main thread prints the core table (1 GB of flanks: ~60 ms that would otherwise sit in
 * front of the right extension) */
static pthread_mutex_t g_warm_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_warm_cv = PTHREAD_COND_INITIALIZER;
static const struct sequenceLibrary *g_warm_lib = NULL;
static int g_warm_quit = 0;
static void *warm_device(void *unused)
{
  (void)unused;
  (void)ramx_default_device();          /* failure is reported by the main thread's own call later */
  /* (Loading the code objects here as well -- ramx_dev_warm -- was measured and dropped: the runtime's start-up already takes as
   * long as the loader does, the main thread then WAITS for this thread (20-100 ms), and the first launches were no faster.) */
  pthread_mutex_lock(&g_warm_mu);
  while (g_warm_lib == NULL && !g_warm_quit) pthread_cond_wait(&g_warm_cv, &g_warm_mu);
  const struct sequenceLibrary *lib = g_warm_lib;
  pthread_mutex_unlock(&g_warm_mu);
  if (lib != NULL)                      /* a failure: seam 1 uploads (and reports) itself */
  {
    const ramx_packed_library *pl = ramx_packed_of(lib);
    if (pl) (void)ramx_preload_library_packed(lib, pl);
    else if (lib->sequence) (void)ramx_preload_library((const int8_t *)lib->sequence, lib->length);
  }
  return NULL;
}
static void warm_offer_library(const struct sequenceLibrary *lib)
{
  pthread_mutex_lock(&g_warm_mu);
  g_warm_lib = lib;
  pthread_cond_signal(&g_warm_cv);
  pthread_mutex_unlock(&g_warm_mu);
}
static void warm_join(void)
{
  if (g_warm_started)
  {
    g_warm_started = 0;
    pthread_mutex_lock(&g_warm_mu);
    g_warm_quit = 1;
    pthread_cond_signal(&g_warm_cv);
    pthread_mutex_unlock(&g_warm_mu);
    pthread_join(g_warm_thread, NULL);
  }
}
static void warm_start(void)
{
  if (getenv("RAMX_NO_WARM_THREAD") == NULL && pthread_create(&g_warm_thread, NULL, warm_device, NULL) == 0)
  {
    g_warm_started = 1;
    atexit(warm_join);                  /* every exit path (loader errors exit(255)) lets the runtime finish starting first */
  }
}

/* everything the command line decides */
struct cli_opts
{
  const char *ranges_file, *outtsv, *outfa, *outmat, *cons_file, *seq_file, *matrix_name, *batch_file;
  int flanking, L, bandwidth, maxn, when_to_stop, num_threads, verbose;
  int gap_ext, gap_open, match, mismatch, cappenalty, minimprovement, is_rs;
  struct scoringSystem *sp;
};

/* banner + parameters, ram_extend.c:397-400, 793-826 */
static void print_header(const struct cli_opts *o, const char *ranges_file, int N, const struct sequenceLibrary *lib)
{
  printf("\nRAMExtend Version %s - build %s date %s\n", k_version, k_build, k_build_date);
  printf("--------------------------------------------------------------\n");
  printf("Parameters:\n");
  printf("  VERBOSE %d\n", o->verbose);
  printf("  SEQUENCE_FILE %s\n", o->seq_file);
  printf("  RANGES_FILE %s\n", ranges_file);
  if (o->num_threads)
    printf("  Multi-Threaded-Masking (EXPERIMENTAL): num_threads = %d\n", o->num_threads);
  printf("  L %d\n", o->L);
  printf("  BANDWIDTH (bandwidth) %d\n", o->bandwidth);
  printf("  MAXN %d\n", o->maxn);
  if (o->is_rs)
  {
    printf("  SCORING SYSTEM: Original RepeatScout method\n");
    printf("     - GAP = %d\n", o->sp->gapextn);
    printf("     - MATCH = %d\n", o->match);
    printf("     - MISMATCH = %d\n", o->mismatch);
  }
  else
  {
    printf("  SCORING SYSTEM: Internally coded matrix '%s'\n", o->matrix_name);
    printf("     - GAP_OPEN = %d\n", o->sp->gapopen);
    printf("     - GAP_EXT = %d\n", o->sp->gapextn);
  }
  printf("     - CAPPENALTY %d\n", o->cappenalty);
  printf("     - MINIMPROVEMENT %d\n", o->minimprovement);
  printf("  WHEN_TO_STOP %d\n", o->when_to_stop);
  printf("--------------------------------------------------------------\n");
  printf("Read in %d ranges, and %ld bp of sequence\n\n", N, (long)lib->length);
}

/* rows [lo_i, hi_i) of the report (stdout), of -outtsv and of -outfa: reference ram_extend.c:606-779 */
struct results_ctx
{
  struct coreAlignment **arr;
  struct sequenceLibrary *lib;
  int flanking;
};

static void results_rows(int lo_i, int hi_i, FILE **outs, void *user)
{
  const struct results_ctx *t = (const struct results_ctx *)user;
  struct sequenceLibrary *lib = t->lib;
  const int flanking = t->flanking;
  FILE *out = outs[0], *fp = outs[1], *fp_fa = outs[2];
  for (long x = lo_i; x < hi_i; x++)
  {
    struct coreAlignment *s = t->arr[x];

      const int si = s->seqIdx;
      const char *ident = lib->identifiers[si];
      const uint64_t lo = si > 0 ? lib->boundaries[si - 1] : 0;
      const uint64_t hi = lib->boundaries[si];
      const uint64_t off = (lib->offsets != NULL && lib->offsets[si] > 0) ? lib->offsets[si] : 0;
      uint64_t ext_start, ext_end, core_start, core_end;
      char orient = '+';
      if (s->orient)
      {
        orient = '-';
        ext_end = (s->leftSeqPos + (uint64_t)s->leftExtensionLen) - lo + 1;
        ext_start = (s->rightSeqPos - (uint64_t)s->rightExtensionLen) - lo + 1;
        core_start = s->rightSeqPos - lo + 1 + off;
        core_end = s->leftSeqPos - lo + 1 + off;
      }
      else
      {
        ext_start = s->leftSeqPos - (uint64_t)s->leftExtensionLen - lo + 1;
        ext_end = (s->rightSeqPos + (uint64_t)s->rightExtensionLen) - lo + 1;
        core_start = s->leftSeqPos - lo + 1 + off;
        core_end = s->rightSeqPos - lo + 1 + off;
      }
      const int ext_len = (int)(ext_end - ext_start + 1);
      char lbuf[32], rbuf[32];
      if (s->leftExtendable) snprintf(lbuf, sizeof(lbuf), "%d", s->leftExtensionLen); else strcpy(lbuf, "*");
      if (s->rightExtendable) snprintf(rbuf, sizeof(rbuf), "%d", s->rightExtensionLen); else strcpy(rbuf, "*");

      fprintf(out, "%s\t%ld\t%ld\t%c\tn=%ld,anchor_range=%ld-%ld,extended_left=%s,extended_right=%s,len=%d,score=%d",
             ident, (long)(off + ext_start), (long)(off + ext_end), orient, x, (long)core_start, (long)core_end,
             lbuf, rbuf, ext_len, s->score);
      /* limit annotations, ram_extend.c:669-693 */
      if (orient == '+')
      {
        if (s->leftExtendable && ((s->leftSeqPos - (uint64_t)s->leftExtensionLen) - s->lowerSeqBound) < 20)
        {
          if (s->lowerSeqBoundFlag == SEQ_BOUNDARY) fprintf(out, ",leftSeqLimit");
          else if (s->lowerSeqBoundFlag == L_BOUNDARY) fprintf(out, ",leftExtLimit");
          else if (s->lowerSeqBoundFlag == CORE_BOUNDARY) fprintf(out, ",leftCoreLimit");
        }
        if (s->rightExtendable && (s->upperSeqBound - (s->rightSeqPos + (uint64_t)s->rightExtensionLen)) < 20)
        {
          if (s->upperSeqBoundFlag == SEQ_BOUNDARY) fprintf(out, ",rightSeqLimit");
          else if (s->upperSeqBoundFlag == L_BOUNDARY) fprintf(out, ",rightExtLimit");
          else if (s->upperSeqBoundFlag == CORE_BOUNDARY) fprintf(out, ",rightCoreLimit");
        }
      }
      else
      {
        if (s->leftExtendable && (s->upperSeqBound - (s->leftSeqPos + (uint64_t)s->leftExtensionLen)) < 20) fprintf(out, ",leftCoreLimit");
        if (s->rightExtendable && ((s->rightSeqPos - (uint64_t)s->rightExtensionLen) - s->lowerSeqBound) < 20) fprintf(out, ",rightCoreLimit");
      }
      fprintf(out, "\n");

      if (fp != NULL)   /* NB: anchor_range here is window-relative, without the genomic offset (ram_extend.c:696-712) */
        fprintf(fp, "%s\t%ld\t%ld\t%c\tn=%ld,anchor_range=%ld-%ld,extended_left=%s,extended_right=%s,len=%d,score=%d\n",
                ident, (long)(off + ext_start), (long)(off + ext_end), orient, x, (long)(s->leftSeqPos - lo + 1),
                (long)(s->rightSeqPos - lo + 1), lbuf, rbuf, ext_len, s->score);

      if (fp_fa != NULL)
      {
        if (s->leftExtensionLen < 0 || s->rightExtensionLen < 0) { fprintf(stderr, "Error: Negative extension length detected\n"); exit(1); }
        uint64_t from = s->leftSeqPos - (uint64_t)s->leftExtensionLen, to = s->rightSeqPos + (uint64_t)s->rightExtensionLen;
        if (s->orient) { to = s->leftSeqPos + (uint64_t)s->leftExtensionLen; from = s->rightSeqPos - (uint64_t)s->rightExtensionLen; }
        if (flanking > 0)
        {
          if (lo == 0 && (uint64_t)flanking > from) from = 1;   /* sic: ram_extend.c:729-730 */
          else from -= (uint64_t)flanking;
          if (to + (uint64_t)flanking > hi) to = hi;
          else to += (uint64_t)flanking;
          fprintf(fp_fa, ">%s:%ld-%ld_%c  n=%ld,anchor_range=%ld-%ld,extended_left=%d,extended_right=%d,len=%d,flanking=%d,score=%d\n",
                  ident, (long)(off + from - lo + 1), (long)(off + to - lo + 1), orient, x, (long)(s->leftSeqPos - lo + 1),
                  (long)(s->rightSeqPos - lo + 1), s->leftExtensionLen, s->rightExtensionLen, ext_len, flanking, s->score);
        }
        else
          fprintf(fp_fa, ">%s:%ld-%ld_%c  n=%ld,anchor_range=%ld-%ld,extended_left=%d,extended_right=%d,len=%d,score=%d\n",
                  ident, (long)(off + from - lo + 1), (long)(off + to - lo + 1), orient, x, (long)(s->leftSeqPos - lo + 1),
                  (long)(s->rightSeqPos - lo + 1), s->leftExtensionLen, s->rightExtensionLen, ext_len, s->score);
        /* sequence line: one buffer + one fwrite instead of a fprintf per base */
        size_t cnt = 0;
        if (to >= from)
        {
          const size_t len = (size_t)(to - from + 1);
          char *buf = (char *)malloc(len + 1);
          const char *codes = lib->sequence ? lib->sequence + from : NULL;
          char *tmpc = NULL;
          if (!codes)                   /* packed library: decode just this stretch */
          {
            tmpc = (char *)malloc(len + 1);
            const ramx_packed_library *pl = ramx_packed_of(lib);
            if (!pl || ramx_packed_decode(pl, from, len, tmpc) != RAMX_OK) { fprintf(stderr, "RAMExtend(ramx): cannot read %zu bases at %ld of the packed library\n", len, (long)from); exit(1); }
            codes = tmpc;
          }
          if (orient == '-')
            for (size_t j = len; j-- > 0;) buf[cnt++] = code_to_char(code_compl(codes[j]));
          else
            for (size_t j = 0; j < len; j++) buf[cnt++] = code_to_char(codes[j]);
          fwrite(buf, 1, cnt, fp_fa);
          free(buf); free(tmpc);
        }
        else if (orient == '-')
        {
          /* the reference's do/while emits one base even for an inverted interval */
          fputc(code_to_char(code_compl(ramx_lib_code(lib, to))), fp_fa);
          cnt = 1;
        }
        fputc('\n', fp_fa);
        if (cnt == 0) { fprintf(stderr, "Error: No sequence emitted for %s:%ld-%ld_%c\n", ident, (long)from, (long)to, orient); exit(1); }
      }
      }
}

/* consensus records, report table, -cons / -outtsv / -outfa: reference ram_extend.c:515-781 */
static void write_results(const struct cli_opts *o, struct coreAlignment *cores, struct sequenceLibrary *lib, const char *master,
                          int rightbp, int leftbp, const char *cons_file, const char *outtsv, const char *outfa)
{
  const int L = o->L, l = 1, flanking = o->flanking;
  const long masterend = (long)L + l + rightbp;
  const long masterstart = (long)L - leftbp;
  if (rightbp > 0 || leftbp > 0)
  {
    FILE *fp = NULL, *fp_fa = NULL;
    if (leftbp > 0) { printf(">left-extension\n"); write_wrapped(stdout, master, (uint64_t)masterstart, (uint64_t)L, (uint64_t)masterstart); }
    if (rightbp > 0) { printf(">right-extension\n"); write_wrapped(stdout, master, (uint64_t)L + l, (uint64_t)masterend, (uint64_t)masterstart); }
    if (cons_file != NULL)
    {
      if ((fp = fopen(cons_file, "w")) == NULL) { fprintf(stderr, "Could not open input file %s\n", cons_file); exit(1); }
      if (leftbp > 0) { fprintf(fp, ">left-extension %d bp\n", leftbp); write_wrapped(fp, master, (uint64_t)masterstart, (uint64_t)L, (uint64_t)masterstart); }
      if (rightbp > 0) { fprintf(fp, ">right-extension %d bp\n", rightbp); write_wrapped(fp, master, (uint64_t)L + l, (uint64_t)masterend, (uint64_t)masterstart); }
      fclose(fp);   /* the reference only closes it inside the right-extension branch (ram_extend.c:583) */
      fp = NULL;
    }
    if (outtsv != NULL && (fp = fopen(outtsv, "w")) == NULL) { fprintf(stderr, "Could not create the TSV output file %s\n", outtsv); exit(1); }
    if (outfa != NULL && (fp_fa = fopen(outfa, "w")) == NULL) { fprintf(stderr, "Could not create the FASTA output file %s\n", outfa); exit(1); }

    printf("\n\nExtended Sequences Report:\n");
    printf("  *** Extended sequences are in 1-based, fully closed coordinates ***\n");
    printf("SEQID  SEQSTART SEQEND ORIENT  EXTENSION_DETAILS\n");
    {
      int cnt = 0;
      for (struct coreAlignment *s = cores; s != NULL; s = s->next) cnt++;
      struct coreAlignment **arr = (struct coreAlignment **)malloc(sizeof(*arr) * (size_t)(cnt ? cnt : 1));
      cnt = 0;
      for (struct coreAlignment *s = cores; s != NULL; s = s->next) arr[cnt++] = s;
      struct results_ctx ctx;
      ctx.arr = arr; ctx.lib = lib; ctx.flanking = flanking;
      FILE *outs[3] = { stdout, fp, fp_fa };
      ramx_parallel_chunks(cnt, 3, outs, results_rows, &ctx);   /* formatted chunk by chunk on the host's cores, written in order */
      free(arr);
    }
    if (fp != NULL) fclose(fp);
    if (fp_fa != NULL) fclose(fp_fa);
  }

}


/*
 * -batch <list>: many families in one process and ONE launch per direction (no counterpart in the reference, whose
 * wrapper util/extend-stk.pl:242-371 starts one RAMExtend per family).  Every non-empty, non-# line of <list> is
 *     ranges.tsv <TAB> log <TAB> cons.fa <TAB> out.tsv <TAB> out.fa          ("-" = not wanted)
 * All other options (-twobit, -L, -bandwidth, -matrix ...) are shared.  For every family the log file receives
 * exactly what a stand-alone run prints on stdout, and the three output files are what -cons/-outtsv/-outfa give.
 */
struct batch_item
{
  char *ranges, *log, *cons, *tsv, *fa;
  struct coreAlignment *cores;
  struct sequenceLibrary *lib;
  int N, rightbp, leftbp;
  char *master;
};

static void to_log(const char *path, int truncate)
{
  fflush(stdout);
  int fd = open(path, O_WRONLY | O_CREAT | (truncate ? O_TRUNC : O_APPEND), 0644);
  if (fd < 0) { fprintf(stderr, "Could not open log file %s\n", path); exit(1); }
  dup2(fd, 1);
  close(fd);
}

static char *field_or_null(char *s) { return (s == NULL || s[0] == 0 || strcmp(s, "-") == 0) ? NULL : s; }

/* flat view of a core list for ramx_extend_batch (arrays owned by the caller's arena) */
static void flatten_cores(struct coreAlignment *cores, int N, ramx_flat_cores *fc)
{
  int64_t *i64 = (int64_t *)malloc(sizeof(int64_t) * 4 * (size_t)(N + 1));
  int8_t *i8 = (int8_t *)malloc(3 * (size_t)(N + 1));
  int32_t *i32 = (int32_t *)calloc(3 * (size_t)(N + 1), sizeof(int32_t));
  fc->n = N;
  fc->left_pos = i64; fc->right_pos = i64 + N; fc->lower = i64 + 2 * N; fc->upper = i64 + 3 * N;
  fc->orient = i8; fc->left_ext = i8 + N; fc->right_ext = i8 + 2 * N;
  fc->left_len = i32; fc->right_len = i32 + N; fc->score = i32 + 2 * N;
  int k = 0;
  for (struct coreAlignment *c = cores; c != NULL && k < N; c = c->next, k++)
  {
    i64[k] = (int64_t)c->leftSeqPos; i64[N + k] = (int64_t)c->rightSeqPos;
    i64[2 * N + k] = (int64_t)c->lowerSeqBound; i64[3 * N + k] = (int64_t)c->upperSeqBound;
    i8[k] = c->orient ? 1 : 0; i8[N + k] = c->leftExtendable ? 1 : 0; i8[2 * N + k] = c->rightExtendable ? 1 : 0;
    i32[k] = c->leftExtensionLen; i32[N + k] = c->rightExtensionLen; i32[2 * N + k] = c->score;
  }
}
static void unflatten_results(struct coreAlignment *cores, const ramx_flat_cores *fc)
{
  int k = 0;
  for (struct coreAlignment *c = cores; c != NULL && k < fc->n; c = c->next, k++)
  {
    c->leftExtensionLen = fc->left_len[k]; c->rightExtensionLen = fc->right_len[k]; c->score = fc->score[k];
  }
}
static void free_flat(ramx_flat_cores *fc) { free((void *)fc->left_pos); free((void *)fc->orient); free(fc->left_len); }

/* the lines extend_alignment prints around the loop (ram_extend.c:886-892, 1216-1231) */
static void print_loop_lines(int direction, int N, int verbose, int when_to_stop, int L, const ramx_run_info *info, int before)
{
  if (before)
  {
    if (verbose >= 3) printf(direction ? "extend_alignment(right): Called with %d edges\n" : "extend_alignment(left): Called with %d edges\n", N);
    return;
  }
  if (info->rows_executed > 0 && verbose >= 3)
  {
    const int last = info->rows_executed - 1;
    if (abs(last - (info->ret - 1)) >= when_to_stop)
      printf("Ending...due to row_idx=%d - max_extension_score_row_idx=%d <= -WHEN_TO_STOP=%d\n", last, info->ret - 1, when_to_stop);
  }
  if (info->limit_warning)
    printf(direction ? "WARNING: Extended sequence right to the limit ( L=%d ).\n" : "WARNING: Extended sequence left to the limit ( L=%d ).\n", L);
}

static int run_batch(struct cli_opts *o, time_t t_start)
{
  const int L = o->L, l = 1;
  if (o->outmat != NULL) { fprintf(stderr, "RAMExtend(ramx): -outmat traces one family; it cannot be combined with -batch\n"); exit(1); }
  FILE *lf = fopen(o->batch_file, "r");
  if (!lf) { fprintf(stderr, "Could not open batch list %s\n", o->batch_file); exit(1); }
  size_t cap = 64, F = 0;
  struct batch_item *it = (struct batch_item *)calloc(cap, sizeof(*it));
  char *line = NULL;
  size_t lcap = 0;
  ssize_t len;
  while ((len = getline(&line, &lcap, lf)) >= 0)
  {
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
    if (line[0] == '#' || line[0] == 0) continue;
    char *f[5] = { NULL, NULL, NULL, NULL, NULL };
    char *p = line;
    for (int k = 0; k < 5 && p; k++) { f[k] = p; char *t = strchr(p, '\t'); if (t) { *t = 0; p = t + 1; } else p = NULL; }
    if (!f[0] || !f[1]) { fprintf(stderr, "batch list: every line needs at least <ranges><TAB><log>\n"); exit(1); }
    if (F == cap) { cap *= 2; it = (struct batch_item *)realloc(it, cap * sizeof(*it)); memset(it + F, 0, (cap - F) * sizeof(*it)); }
    it[F].ranges = strdup(f[0]); it[F].log = strdup(f[1]);
    it[F].cons = field_or_null(f[2]) ? strdup(f[2]) : NULL;
    it[F].tsv = field_or_null(f[3]) ? strdup(f[3]) : NULL;
    it[F].fa = field_or_null(f[4]) ? strdup(f[4]) : NULL;
    F++;
  }
  free(line);
  fclose(lf);
  fflush(stdout);
  const int saved = dup(1);

  /* phase 1: load every family; its log gets the banner, the parameters and the core table */
  ramx_family *fam = (ramx_family *)calloc(F ? F : 1, sizeof(*fam));
  for (size_t i = 0; i < F; i++)
  {
    to_log(it[i].log, 1);
    it[i].lib = ramx_load_sequence_subset_minimal(o->seq_file, it[i].ranges, &it[i].cores, &it[i].N, L + o->bandwidth);
    print_header(o, it[i].ranges, it[i].N, it[i].lib);
    ramx_print_core_edges(it[i].cores, it[i].lib, 0, o->verbose ? 1 : 0);
    it[i].master = (char *)calloc((size_t)(2 * (long)L + l + 1), 1);
    it[i].master[L] = RAMX_SYM_N;
    fam[i].sequence = (const int8_t *)it[i].lib->sequence;
    fam[i].seq_len = it[i].lib->length;
    fam[i].master = (int8_t *)it[i].master;
  }
  fflush(stdout);
  dup2(saved, 1);

  int32_t *mflat = (int32_t *)calloc(100 * 100, sizeof(int32_t));
  for (int a = 0; a < 4; a++)
  {
    for (int b = 0; b < 8; b++) mflat[a * 100 + b] = o->sp->matrix[a][b];
    mflat[a * 100 + RAMX_SYM_N] = o->sp->matrix[a][RAMX_SYM_N];
  }
  ramx_params p;
  p.bandwidth = o->bandwidth; p.cappenalty = o->cappenalty; p.minimprovement = o->minimprovement; p.L = L;
  p.when_to_stop = o->when_to_stop; p.l = l; p.gapopen = o->sp->gapopen; p.gapextn = o->sp->gapextn; p.matrix = mflat;
  ramx_run_info *ir = (ramx_run_info *)calloc(F ? F : 1, sizeof(*ir)), *il = (ramx_run_info *)calloc(F ? F : 1, sizeof(*il));

  warm_join();
  /* phase 2: right extension of all families in one launch; phase 3: per family, overlap avoidance */
  for (size_t i = 0; i < F; i++) flatten_cores(it[i].cores, it[i].N, &fam[i].cores);
  if (ramx_extend_batch(1, fam, (int32_t)F, &p, ir) < 0) { fprintf(stderr, "RAMExtend(ramx): batch extension failed: %s\n", ramx_last_error()); exit(1); }
  for (size_t i = 0; i < F; i++)
  {
    unflatten_results(it[i].cores, &fam[i].cores);
    free_flat(&fam[i].cores);
    it[i].rightbp = ir[i].ret;
    to_log(it[i].log, 0);
    print_loop_lines(1, it[i].N, o->verbose, o->when_to_stop, L, &ir[i], 1);
    print_loop_lines(1, it[i].N, o->verbose, o->when_to_stop, L, &ir[i], 0);
    printf("Extended right: %d bp\n", it[i].rightbp);
    ramx_overlap_avoidance(it[i].cores, it[i].lib);
  }
  fflush(stdout);
  dup2(saved, 1);

  /* phase 4: left extension in one launch; phase 5: per family, results */
  for (size_t i = 0; i < F; i++) flatten_cores(it[i].cores, it[i].N, &fam[i].cores);
  if (ramx_extend_batch(0, fam, (int32_t)F, &p, il) < 0) { fprintf(stderr, "RAMExtend(ramx): batch extension failed: %s\n", ramx_last_error()); exit(1); }
  for (size_t i = 0; i < F; i++)
  {
    unflatten_results(it[i].cores, &fam[i].cores);
    free_flat(&fam[i].cores);
    it[i].leftbp = il[i].ret;
    to_log(it[i].log, 0);
    print_loop_lines(0, it[i].N, o->verbose, o->when_to_stop, L, &il[i], 1);
    print_loop_lines(0, it[i].N, o->verbose, o->when_to_stop, L, &il[i], 0);
    printf("Extended left : %d bp\n", it[i].leftbp);
    write_results(o, it[i].cores, it[i].lib, it[i].master, it[i].rightbp, it[i].leftbp, it[i].cons, it[i].tsv, it[i].fa);
    const double duration = difftime(time(0), t_start);
    printf("Program duration is %.1f sec = %.1f min = %.1f hr\n", duration, duration / 60.0, duration / 3600.0);
  }
  fflush(stdout);
  dup2(saved, 1);
  close(saved);
  printf("RAMExtend(ramx) batch: %zu families done\n", F);
  for (size_t i = 0; i < F; i++)
  {
    ramx_free_library(it[i].lib, it[i].cores);
    free(it[i].master); free(it[i].ranges); free(it[i].log); free(it[i].cons); free(it[i].tsv); free(it[i].fa);
  }
  free(it); free(fam); free(ir); free(il); free(mflat);
  ramx_free_scoring_system(o->sp);
  return 0;
}

int ramx_cli_main(int argc, char **argv)
{
  const time_t t_start = time(0);
  struct cli_opts o;
  memset(&o, 0, sizeof(o));
  const int l = 1;
  g_t_main = now_s();

  if (opt_bool(argc, argv, "-version"))
  {
    printf("RAMExtend Version %s - build %s date %s\n", k_version, k_build, k_build_date);
    exit(0);
  }
  opt_string(argc, argv, "-batch", &o.batch_file);
  if (!opt_string(argc, argv, "-ranges", &o.ranges_file) && !o.batch_file) usage();

  opt_int(argc, argv, "-addflanking", &o.flanking);
  opt_string(argc, argv, "-outtsv", &o.outtsv);
  opt_string(argc, argv, "-outfa", &o.outfa);
  opt_string(argc, argv, "-outmat", &o.outmat);
  opt_string(argc, argv, "-cons", &o.cons_file);
  if (!opt_int(argc, argv, "-L", &o.L)) o.L = 10000;
  if (!opt_int(argc, argv, "-bandwidth", &o.bandwidth)) o.bandwidth = 14;
  if (!opt_int(argc, argv, "-maxoccurrences", &o.maxn)) o.maxn = 10000;
  if (!opt_int(argc, argv, "-stopafter", &o.when_to_stop)) o.when_to_stop = 100;
  if (!opt_int(argc, argv, "-threads", &o.num_threads)) o.num_threads = 0;
  o.verbose = opt_bool(argc, argv, "-vvvvvv") ? 20 : opt_bool(argc, argv, "-vvvvv") ? 12 :
              opt_bool(argc, argv, "-vvvv") ? 10 : opt_bool(argc, argv, "-vvv") ? 3 :
              opt_bool(argc, argv, "-vv") ? 2 : opt_bool(argc, argv, "-v") ? 1 : 0;

  if (!opt_string(argc, argv, "-matrix", &o.matrix_name)) o.matrix_name = "20p43g";
  if (o.matrix_name == NULL) o.matrix_name = "";
  o.is_rs = strcmp(o.matrix_name, "repeatscout") == 0;
  if (o.is_rs)
  {
    if (opt_int(argc, argv, "-match", &o.match) && opt_int(argc, argv, "-mismatch", &o.mismatch) &&
        opt_int(argc, argv, "-gap", &o.gap_ext))
      o.sp = ramx_get_repeatscout_matrix(o.match, o.mismatch, o.gap_ext);
    else
      o.sp = ramx_get_repeatscout_matrix(1, -1, -5);
  }
  else
  {
    if (opt_int(argc, argv, "-gapopen", &o.gap_open) && opt_int(argc, argv, "-gapext", &o.gap_ext))
      o.sp = ramx_get_matrix_using_gap_penalties(o.matrix_name, o.gap_open, o.gap_ext);
    else
      o.sp = ramx_get_matrix(o.matrix_name);
  }
  /* per-matrix defaults, ram_extend.c:301-344 */
  {
    int def_min, def_cap;
    if (!strcmp(o.matrix_name, "14p43g") || !strcmp(o.matrix_name, "18p43g") || !strcmp(o.matrix_name, "20p43g")) { def_min = 27; def_cap = -90; }
    else if (!strcmp(o.matrix_name, "25p43g")) { def_min = 24; def_cap = -90; }
    else if (o.is_rs) { def_min = 3; def_cap = -20; }
    else { printf("Matrix name not found!\n"); exit(1); }
    if (!opt_int(argc, argv, "-minimprovement", &o.minimprovement)) o.minimprovement = def_min;
    if (!opt_int(argc, argv, "-cappenalty", &o.cappenalty)) o.cappenalty = def_cap;
  }
  if (!opt_string(argc, argv, "-twobit", &o.seq_file))
  {
    if (opt_string(argc, argv, "-sequence", &o.seq_file))
    {
      printf("-sequence is deprecated!....may return someday\n");
      exit(1);
    }
    usage();
  }
  warm_start();
  if (o.batch_file) return run_batch(&o, t_start);

  g_timing = getenv("RAMX_TIMING") != NULL;
  g_t_last = now_s();
  if (g_timing) atexit(timing_at_exit);
  phase_done("options");
  const int L = o.L;
  char *master = (char *)malloc((size_t)(2 * (long)L + l + 1));
  if (!master) { fprintf(stderr, "Could not allocate space for master array\n"); exit(1); }
  memset(master, 0, (size_t)(2 * (long)L + l + 1));

  struct coreAlignment *cores = NULL;
  int N = 0;
  /* the windows stay packed (4 bases per byte, as in the .2bit file) on the host and go to the device packed: the report and
   * -outfa decode the few bases they print.  RAMX_CLI_BYTES=1: the one-byte-per-base library of the reference (A/B) */
  struct sequenceLibrary *lib = getenv("RAMX_CLI_BYTES") != NULL
                                    ? ramx_load_sequence_subset_minimal(o.seq_file, o.ranges_file, &cores, &N, L + o.bandwidth)
                                    : ramx_load_sequence_subset_packed(o.seq_file, o.ranges_file, &cores, &N, L + o.bandwidth, NULL);
  FILE *fp_mat = NULL;
  if (o.outmat != NULL)       /* ram_extend.c:384-390 */
  {
    if ((fp_mat = fopen(o.outmat, "w")) == NULL)
    {
      fprintf(stderr, "Could not create the output matrix file %s\n", o.outmat);
      exit(1);
    }
  }
  phase_done("load");
  if (g_warm_started && lib != NULL && N > 0) warm_offer_library(lib);
  print_header(&o, o.ranges_file, N, lib);
  ramx_print_core_edges(cores, lib, 0, o.verbose ? 1 : 0);
  phase_done("core table");
  master[L] = RAMX_SYM_N;   /* the l = 1 spacer, never printed (ram_extend.c:415-416) */

  ramx_set_runtime(o.verbose, o.when_to_stop, l);
  fflush(stdout);
  warm_join();
  phase_done("device ready");
  int rightbp = ramx_extend_alignment(1, cores, NULL, lib, master, o.bandwidth, o.cappenalty, o.minimprovement, L, N, o.sp, fp_mat);
  printf("Extended right: %d bp\n", rightbp);
  phase_done("extend right");
  ramx_overlap_avoidance(cores, lib);
  phase_done("overlap avoidance");
  int leftbp = ramx_extend_alignment(0, cores, NULL, lib, master, o.bandwidth, o.cappenalty, o.minimprovement, L, N, o.sp, fp_mat);
  printf("Extended left : %d bp\n", leftbp);
  phase_done("extend left");
  write_results(&o, cores, lib, master, rightbp, leftbp, o.cons_file, o.outtsv, o.outfa);
  if (fp_mat != NULL) fclose(fp_mat);     /* ram_extend.c:778-779 */
  phase_done("report + outputs");

  const double duration = difftime(time(0), t_start);
  printf("Program duration is %.1f sec = %.1f min = %.1f hr\n", duration, duration / 60.0, duration / 3600.0);
  ramx_free_scoring_system(o.sp);
  ramx_free_library(lib, cores);
  free(master);
  phase_done("free");
  return 0;
}
