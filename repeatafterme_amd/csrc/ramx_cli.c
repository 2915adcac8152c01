/*
 * ramx_cli.c -- the RAMExtend command line (reference ram_extend.c:123-826 main/usage/
 * print_parameters, cmd_line_opts.c).  Same argv semantics (prefix matching, per-matrix
 * defaults), same stdout / -cons / -outtsv / -outfa bytes; the two extend_alignment calls go to
 * the device path (ramx_extend_alignment).
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>

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

int ramx_cli_main(int argc, char **argv)
{
  const time_t t_start = time(0);
  const char *ranges_file = NULL, *outtsv = NULL, *outfa = NULL, *outmat = NULL, *cons_file = NULL;
  const char *seq_file = NULL, *matrix_name = NULL;
  int flanking = 0, L, bandwidth, maxn, when_to_stop, num_threads = 0, verbose;
  int gap_ext = 0, gap_open = 0, match = 0, mismatch = 0, cappenalty = 0, minimprovement = 0;
  const int l = 1;
  struct scoringSystem *sp;

  if (opt_bool(argc, argv, "-version"))
  {
    printf("RAMExtend Version %s - build %s date %s\n", k_version, k_build, k_build_date);
    exit(0);
  }
  if (!opt_string(argc, argv, "-ranges", &ranges_file)) usage();

  opt_int(argc, argv, "-addflanking", &flanking);
  opt_string(argc, argv, "-outtsv", &outtsv);
  opt_string(argc, argv, "-outfa", &outfa);
  opt_string(argc, argv, "-outmat", &outmat);
  opt_string(argc, argv, "-cons", &cons_file);
  if (!opt_int(argc, argv, "-L", &L)) L = 10000;
  if (!opt_int(argc, argv, "-bandwidth", &bandwidth)) bandwidth = 14;
  if (!opt_int(argc, argv, "-maxoccurrences", &maxn)) maxn = 10000;
  if (!opt_int(argc, argv, "-stopafter", &when_to_stop)) when_to_stop = 100;
  if (!opt_int(argc, argv, "-threads", &num_threads)) num_threads = 0;
  verbose = opt_bool(argc, argv, "-vvvvvv") ? 20 : opt_bool(argc, argv, "-vvvvv") ? 12 :
            opt_bool(argc, argv, "-vvvv") ? 10 : opt_bool(argc, argv, "-vvv") ? 3 :
            opt_bool(argc, argv, "-vv") ? 2 : opt_bool(argc, argv, "-v") ? 1 : 0;

  if (!opt_string(argc, argv, "-matrix", &matrix_name)) matrix_name = "20p43g";
  if (matrix_name == NULL) matrix_name = "";
  const int is_rs = strcmp(matrix_name, "repeatscout") == 0;
  if (is_rs)
  {
    if (opt_int(argc, argv, "-match", &match) && opt_int(argc, argv, "-mismatch", &mismatch) &&
        opt_int(argc, argv, "-gap", &gap_ext))
      sp = ramx_get_repeatscout_matrix(match, mismatch, gap_ext);
    else
      sp = ramx_get_repeatscout_matrix(1, -1, -5);
  }
  else
  {
    if (opt_int(argc, argv, "-gapopen", &gap_open) && opt_int(argc, argv, "-gapext", &gap_ext))
      sp = ramx_get_matrix_using_gap_penalties(matrix_name, gap_open, gap_ext);
    else
      sp = ramx_get_matrix(matrix_name);
  }
  /* per-matrix defaults, ram_extend.c:301-344 */
  {
    int def_min, def_cap;
    if (!strcmp(matrix_name, "14p43g") || !strcmp(matrix_name, "18p43g") || !strcmp(matrix_name, "20p43g")) { def_min = 27; def_cap = -90; }
    else if (!strcmp(matrix_name, "25p43g")) { def_min = 24; def_cap = -90; }
    else if (is_rs) { def_min = 3; def_cap = -20; }
    else { printf("Matrix name not found!\n"); exit(1); }
    if (!opt_int(argc, argv, "-minimprovement", &minimprovement)) minimprovement = def_min;
    if (!opt_int(argc, argv, "-cappenalty", &cappenalty)) cappenalty = def_cap;
  }

  char *master = (char *)malloc((size_t)(2 * (long)L + l + 1));
  if (!master) { fprintf(stderr, "Could not allocate space for master array\n"); exit(1); }
  memset(master, 0, (size_t)(2 * (long)L + l + 1));

  struct coreAlignment *cores = NULL;
  struct sequenceLibrary *lib = NULL;
  int N = 0;
  if (opt_string(argc, argv, "-twobit", &seq_file))
    lib = ramx_load_sequence_subset_minimal(seq_file, ranges_file, &cores, &N, L + bandwidth);
  else if (opt_string(argc, argv, "-sequence", &seq_file))
  {
    printf("-sequence is deprecated!....may return someday\n");
    exit(1);
  }
  else
    usage();

  if (outmat != NULL)
  {
    fprintf(stderr, "RAMExtend(ramx): -outmat (per-cell DP path dump) is not available on the device path\n");
    exit(1);
  }

  /* banner + parameters, ram_extend.c:397-400, 793-826 */
  printf("\nRAMExtend Version %s - build %s date %s\n", k_version, k_build, k_build_date);
  printf("--------------------------------------------------------------\n");
  printf("Parameters:\n");
  printf("  VERBOSE %d\n", verbose);
  printf("  SEQUENCE_FILE %s\n", seq_file);
  printf("  RANGES_FILE %s\n", ranges_file);
  if (num_threads)
    printf("  Multi-Threaded-Masking (EXPERIMENTAL): num_threads = %d\n", num_threads);
  printf("  L %d\n", L);
  printf("  BANDWIDTH (bandwidth) %d\n", bandwidth);
  printf("  MAXN %d\n", maxn);
  if (is_rs)
  {
    printf("  SCORING SYSTEM: Original RepeatScout method\n");
    printf("     - GAP = %d\n", sp->gapextn);
    printf("     - MATCH = %d\n", match);
    printf("     - MISMATCH = %d\n", mismatch);
  }
  else
  {
    printf("  SCORING SYSTEM: Internally coded matrix '%s'\n", matrix_name);
    printf("     - GAP_OPEN = %d\n", sp->gapopen);
    printf("     - GAP_EXT = %d\n", sp->gapextn);
  }
  printf("     - CAPPENALTY %d\n", cappenalty);
  printf("     - MINIMPROVEMENT %d\n", minimprovement);
  printf("  WHEN_TO_STOP %d\n", when_to_stop);
  printf("--------------------------------------------------------------\n");
  printf("Read in %d ranges, and %ld bp of sequence\n\n", N, (long)lib->length);

  ramx_print_core_edges(cores, lib, 0, verbose ? 1 : 0);
  master[L] = RAMX_SYM_N;   /* the l = 1 spacer, never printed (ram_extend.c:415-416) */

  ramx_set_runtime(verbose, when_to_stop, l);
  fflush(stdout);
  int rightbp = ramx_extend_alignment(1, cores, NULL, lib, master, bandwidth, cappenalty, minimprovement, L, N, sp, NULL);
  printf("Extended right: %d bp\n", rightbp);
  ramx_overlap_avoidance(cores, lib);
  const long masterend = (long)L + l + rightbp;
  int leftbp = ramx_extend_alignment(0, cores, NULL, lib, master, bandwidth, cappenalty, minimprovement, L, N, sp, NULL);
  printf("Extended left : %d bp\n", leftbp);
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
    long x = 0;
    for (struct coreAlignment *s = cores; s != NULL; s = s->next, x++)
    {
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

      printf("%s\t%ld\t%ld\t%c\tn=%ld,anchor_range=%ld-%ld,extended_left=%s,extended_right=%s,len=%d,score=%d",
             ident, (long)(off + ext_start), (long)(off + ext_end), orient, x, (long)core_start, (long)core_end,
             lbuf, rbuf, ext_len, s->score);
      /* limit annotations, ram_extend.c:669-693 */
      if (orient == '+')
      {
        if (s->leftExtendable && ((s->leftSeqPos - (uint64_t)s->leftExtensionLen) - s->lowerSeqBound) < 20)
        {
          if (s->lowerSeqBoundFlag == SEQ_BOUNDARY) printf(",leftSeqLimit");
          else if (s->lowerSeqBoundFlag == L_BOUNDARY) printf(",leftExtLimit");
          else if (s->lowerSeqBoundFlag == CORE_BOUNDARY) printf(",leftCoreLimit");
        }
        if (s->rightExtendable && (s->upperSeqBound - (s->rightSeqPos + (uint64_t)s->rightExtensionLen)) < 20)
        {
          if (s->upperSeqBoundFlag == SEQ_BOUNDARY) printf(",rightSeqLimit");
          else if (s->upperSeqBoundFlag == L_BOUNDARY) printf(",rightExtLimit");
          else if (s->upperSeqBoundFlag == CORE_BOUNDARY) printf(",rightCoreLimit");
        }
      }
      else
      {
        if (s->leftExtendable && (s->upperSeqBound - (s->leftSeqPos + (uint64_t)s->leftExtensionLen)) < 20) printf(",leftCoreLimit");
        if (s->rightExtendable && ((s->rightSeqPos - (uint64_t)s->rightExtensionLen) - s->lowerSeqBound) < 20) printf(",rightCoreLimit");
      }
      printf("\n");

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
          if (orient == '-')
            for (uint64_t j = to;; j--) { buf[cnt++] = code_to_char(code_compl(lib->sequence[j])); if (j == from) break; }
          else
            for (uint64_t j = from; j <= to; j++) buf[cnt++] = code_to_char(lib->sequence[j]);
          fwrite(buf, 1, cnt, fp_fa);
          free(buf);
        }
        else if (orient == '-')
        {
          /* the reference's do/while emits one base even for an inverted interval */
          fputc(code_to_char(code_compl(lib->sequence[to])), fp_fa);
          cnt = 1;
        }
        fputc('\n', fp_fa);
        if (cnt == 0) { fprintf(stderr, "Error: No sequence emitted for %s:%ld-%ld_%c\n", ident, (long)from, (long)to, orient); exit(1); }
      }
    }
    if (fp != NULL) fclose(fp);
    if (fp_fa != NULL) fclose(fp_fa);
  }

  const double duration = difftime(time(0), t_start);
  printf("Program duration is %.1f sec = %.1f min = %.1f hr\n", duration, duration / 60.0, duration / 3600.0);
  ramx_free_scoring_system(sp);
  ramx_free_library(lib, cores);
  free(master);
  return 0;
}
