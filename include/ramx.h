/*
 * ramx.h -- C-ABI of libramx.so: the MI355X-native RAMExtend extension loop.
 *
 * Plain C, plain pointers and sizes; no torch / HIP types cross this boundary.
 * Two concentric seams:
 *
 *   seam 1  extend_alignment()-compatible entry      (replaces reference ram_extend.c:859-1258,
 *           ramx_extend_alignment / ramx_extend_flat   declared ram_extend.h:9-13)
 *   seam 2  thin device API  ramx_dev_*               (init / upload / run_direction / download /
 *                                                       destroy; what seam 1 is built from)
 *
 * plus the surfaces either side of the path that the RAMExtend CLI needs:
 *   ramx_get_matrix* / ramx_free_scoring_system       (replaces score_system.c:23-30,91-180,182-400)
 *   ramx_load_sequence_subset_minimal                 (replaces sequence.c:505-923 + 942-976)
 *   ramx_print_core_edges                             (replaces report.c:161-502)
 *   ramx_allocate_score / ramx_free_score             (replaces bnw_extend.c:87-155; the device path
 *                                                       keeps its own state, these are kept only so a
 *                                                       caller written against the reference still links)
 *
 * There is NO CPU fallback behind any of these: every entry that computes runs the HIP kernels and
 * fails loudly (message on stderr + non-zero status / exit(1) where the reference would exit) if
 * no gfx950 device is usable.
 */
#ifndef RAMX_H
#define RAMX_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Data model -- layout-identical to the reference structs so a reference caller can pass its
 * own objects (common.h:80-98, sequence.h:36-46, score_system.h:7-17).  When compiling inside
 * the reference tree define RAMX_USE_REFERENCE_STRUCTS and include its headers first.
 * ------------------------------------------------------------------------------------------ */
#ifndef RAMX_USE_REFERENCE_STRUCTS
enum CoreBoundFlag { L_BOUNDARY = 0, SEQ_BOUNDARY = 1, CORE_BOUNDARY = 2, EXT_BOUNDARY = 3 };

struct coreAlignment
{
  struct coreAlignment *next;
  int seqIdx;
  uint64_t leftSeqPos;
  uint64_t rightSeqPos;
  char leftExtendable;
  char rightExtendable;
  uint64_t lowerSeqBound;
  uint64_t upperSeqBound;
  enum CoreBoundFlag lowerSeqBoundFlag;
  enum CoreBoundFlag upperSeqBoundFlag;
  int leftExtensionLen;
  int rightExtensionLen;
  int score;
  char orient;
};

struct sequenceLibrary
{
  char *sequence;
  char **identifiers;
  uint64_t *boundaries;
  uint64_t *offsets;
  uint64_t length;
  int count;
  int markov_chain_order;
  uint32_t **markov_chain_prob_tables;
};

struct scoringSystem
{
  char *name;
  int **matrix;
  int msize;
  char *alphabet;
  double m_lambda;
  double m_bg_freqs[4];
  int gapopen;
  int gapextn;
};
#endif

/* base codes, reference sequence.h:7-15 */
#define RAMX_SYM_N 99

/* ------------------------------------------------------------------------------------------
 * Seam 1: the extension loop
 * ------------------------------------------------------------------------------------------ */

/* The three globals the reference reads inside the hot path (ram_extend.c:40 `l`, :52 `VERBOSE`,
 * :61 `WHEN_TO_STOP`).  Call before ramx_extend_alignment; defaults are 0 / 100 / 1. */
void ramx_set_runtime(int verbose, int when_to_stop, int l);

/* Drop-in for reference extend_alignment (ram_extend.h:9-13, ram_extend.c:859-1258): same
 * arguments, same return value (max_extension_score_row_idx + 1), same side effects on
 * master[], coreAlign[*].{left,right}ExtensionLen and .score, same stdout lines at VERBOSE<10.
 * `score` (the reference's int**** DP state) is ignored: state lives in HBM.  With pathStringFile
 * (-outmat) the direction runs one column launch at a time and the reference's trace lines
 * (ram_extend.c:1122-1132) are written from the device's per-cell path codes. */
int ramx_extend_alignment(int direction, struct coreAlignment *coreAlign, int ****score,
                          struct sequenceLibrary *seqLib, char *master, int BANDWIDTH,
                          int CAPPENALTY, int MINIMPROVEMENT, int L, int N,
                          struct scoringSystem *scoreParams, FILE *pathStringFile);

/* Same loop on flat arrays (what ramx_extend_alignment flattens to, and what the Python mirror
 * binds).  matrix is int32[100*100], row-major [consensus][sequence base].  Returns the
 * reference's return value, or a negative ramx error code. */
typedef struct ramx_flat_cores
{
  int32_t n;
  const int64_t *left_pos, *right_pos, *lower, *upper;
  const int8_t *orient, *left_ext, *right_ext;
  int32_t *left_len, *right_len, *score;       /* in/out */
} ramx_flat_cores;

typedef struct ramx_params
{
  int32_t bandwidth, cappenalty, minimprovement, L, when_to_stop, l, gapopen, gapextn;
  const int32_t *matrix;
} ramx_params;

typedef struct ramx_run_info
{
  int32_t ret;              /* max_extension_score_row_idx + 1 */
  int32_t rows_executed;    /* row_idx iterations executed (the metric's "columns") */
  int32_t limit_warning;    /* 1 iff the reference would print "WARNING: Extended ... to the limit" */
  int32_t overflow32;       /* 1 iff a column sum left the int32 range (reference would have wrapped) */
  int32_t n_extendable;     /* flanks on the device in this direction */
  int32_t launches;         /* column-kernel launches issued (>= rows_executed: the host runs ahead) */
  double  loop_ms;          /* HIP-event time of the column loop on the library's stream */
  double  kernel_ms_avg;    /* mean duration of sampled single column-kernel launches (HIP events) */
  int32_t kernel_samples;
  double  prep_ms;          /* host flatten + H2D + pack kernel (wall clock) */
  int32_t persistent;       /* 1: the whole loop ran as ONE persistent launch (rows resident on chip) */
  int32_t lanes_per_flank;  /* 1: one lane per flank; 2..16: the cell-parallel kernels split a band row over that many lanes */
  int32_t respeculated_rows; /* device-wide cell-parallel kernel and packed-row kernel: rows that workgroup 0 computed a second time
                               because its own guess of the vote was wrong (it runs ahead of the device-wide vote); 0 on every other route */
  int32_t packed_rows;      /* columns that ran in the packed-row persistent kernel (two cells per register, int16 relative to a
                               per-flank base: csrc/ramx_kernels_packed.h); 0 on every other route */
  int32_t lean_rows;        /* ... of which the first wave ran as LEAN rows (no candidate rows, no best-cell index) */
} ramx_run_info;

int ramx_extend_flat(int direction, ramx_flat_cores *cores, const int8_t *sequence, uint64_t seq_len,
                     int8_t *master, const ramx_params *p, ramx_run_info *info);

/* Seam 1 keeps the library on the device between calls, keyed on (pointer, length, 64-bit content fingerprint), so
 * the second direction does not upload it again (libraries above 64 MiB are fingerprinted in chunks by worker threads; when
 * pointer and length match the device copy the content check runs beside the direction and is joined before the write-back).
 * The reference has no such state (ram_extend.c reads seqLib->sequence on every call): a caller who
 * wants to be explicit can drop the device copy with this call. */
void ramx_invalidate_library(void);
/* Upload a library ahead of the first extension call (e.g. from a helper thread while the caller still prints its
 * report tables): the next ramx_extend_flat / ramx_extend_alignment calls on the same (pointer, length) use the device
 * copy without looking at the buffer again -- the caller vouches that it does not change until
 * ramx_invalidate_library() or a call with another buffer. */
int ramx_preload_library(const int8_t *sequence, uint64_t seq_len);
/* The packed twin of a library loaded with ramx_load_sequence_subset_packed goes to the device; ramx_extend_alignment calls
 * on that seqLib (whose ->sequence is NULL) then run on it, until ramx_invalidate_library() or another preload. */
struct ramx_packed_library;
int ramx_preload_library_packed(const struct sequenceLibrary *seqLib, const struct ramx_packed_library *pl);

/* ------------------------------------------------------------------------------------------
 * Seam 2: thin device API (one ramx_dev per GPU / per process rank)
 * ------------------------------------------------------------------------------------------ */
typedef struct ramx_dev ramx_dev;

#define RAMX_OK            0
#define RAMX_ERR_NO_DEVICE (-101)
#define RAMX_ERR_HIP       (-102)
#define RAMX_ERR_ARG       (-103)
#define RAMX_ERR_STATE     (-104)
#define RAMX_ERR_COMM      (-105)
#define RAMX_ERR_UNSUPPORTED (-106)

const char *ramx_last_error(void);
int ramx_device_count(void);                                   /* gfx950 devices visible; <0 on error */
int ramx_dev_create(int device_ordinal, ramx_dev **out);       /* init */
void ramx_dev_destroy(ramx_dev *d);                            /* destroy */

/* upload: the 1-byte-per-base library (seqLib->sequence) goes to HBM once and is shared by both
 * directions. */
int ramx_dev_load_library(ramx_dev *d, const int8_t *sequence, uint64_t length);
/* upload of a packed library (ramx_packed_library, below): the pack kernel of every direction then builds the flank windows
 * straight from the 2-bit payload; the one-byte-per-base form never exists on the host or the device */
struct ramx_packed_library;
int ramx_dev_load_library_packed(ramx_dev *d, const struct ramx_packed_library *pl);

/* One flank (an extendable core seen from one direction), already resolved by the host:
 * the base aligned to band cell (row r, offset o) is library[start + step*(o + r)], complemented
 * if compl != 0, and is inside the flank iff t_lo <= o + r <= t_hi  (SURVEY.md App. D rule 1). */
typedef struct ramx_flank
{
  int64_t start;
  int32_t t_lo, t_hi;
  int8_t  step;       /* +1 / -1 */
  int8_t  compl_;     /* reverse-strand core: complement the base */
  int8_t  pad_[6];
} ramx_flank;

/* Host-side resolution of the cores that are extendable in `direction` into flank descriptors
 * (reference bnw_extend.c:778-788,824-868).  flanks / core_index must hold cores->n entries;
 * core_index[i] is the position in the core list of flank i.  Returns the number of flanks. */
int ramx_resolve_flanks(int direction, const ramx_flat_cores *cores, int bandwidth, int L,
                        ramx_flank *flanks, int32_t *core_index);

/* upload (per direction): flank descriptors -> HBM, pack kernel builds the transposed 4-bit
 * windows, DP state / vote / control buffers are (re)initialised. */
int ramx_dev_begin_direction(ramx_dev *d, const ramx_flank *flanks, int32_t n_flanks, const ramx_params *p);

/* run_direction: the whole column loop, asynchronous on the device's stream with the stop rule
 * evaluated on the device; returns when the loop has stopped. */
int ramx_dev_run_direction(ramx_dev *d, ramx_run_info *info);

/* download: consensus bases of the executed columns (cons[0..rows_executed)), and the
 * trimmed per-flank high score / position (trimmed_sequence_high_score[_pos], ram_extend.c:902-903). */
int ramx_dev_download(ramx_dev *d, int8_t *cons, int32_t cons_cap, int32_t *trim_high, int32_t *trim_pos);

/* debug / test hook: current DP row state of one flank as [2W+1][2] int32 + high,pos.  The device stores each
 * cell transformed: (m, e) = (max(sub,gap), max(sub+gapopen,gap)+gapextn) -- see csrc/ramx_kernels_common.h. */
int ramx_dev_peek_state(ramx_dev *d, int32_t flank, int32_t *cells, int32_t *high, int32_t *pos);

/* debug / test hook: with RAMX_CP_PEEK=1 in the environment the cell-parallel family kernel keeps the final DP row of
 * every flank of the last ramx_dev_run_families call; cells receives [2W+1][2] int32 in the (m, e) encoding of
 * ramx_dev_peek_state.  `flank` indexes the (padded) flank array of that call; d == NULL means the process-wide
 * session that seam 1 (ramx_extend_flat / ramx_extend_batch) runs on. */
int ramx_dev_peek_family_state(ramx_dev *d, int32_t flank, int32_t *cells);

/* -outmat support (reference ram_extend.c:1122-1132, bnw_extend.c:1027-1044): with a trace callback set, a direction runs
 * one column launch at a time and hands every executed row to the callback: the winning base, and per flank (in flank
 * order) the path code of each band cell (0: the substitution holds the cell's score, 1: the deletion, 2: the insertion),
 * the row's best score and its sequence index (row + offset).  d == NULL: the process-wide session of seam 1.  A
 * debugging aid: slow by construction.  cb == NULL switches it off. */
typedef void (*ramx_row_trace_cb)(int32_t row, int32_t besta, const int8_t *codes /* [n_flanks][2W+1] */,
                                  const int32_t *best_score, const int32_t *best_idx, void *user);
int ramx_dev_set_row_trace(ramx_dev *d, ramx_row_trace_cb cb, void *user);

/* Batch mode (no counterpart in the reference, whose wrapper util/extend-stk.pl:242-371 starts one
 * RAMExtend process per family): many families in ONE launch, one workgroup per family, every family with its own
 * consensus / vote / stop rule.  Flanks are family-major; every family starts at a multiple of 64 in the flank
 * array (pad with empty flanks: t_lo = 1, t_hi = 0) and has at most 512 flanks (else RAMX_ERR_UNSUPPORTED: run such
 * a family through seam 1).  Band widths 14, 20 and 40 with non-positive gap penalties keep the rows in registers
 * (infos[].persistent == 1); every other band width / gap sign streams them through L2 (persistent == 2).
 * cons is [n_families][L]; trim_* are per (padded) flank; infos per family.  The library must be loaded first. */
int ramx_dev_run_families(ramx_dev *d, const ramx_flank *flanks, int32_t n_padded, const int32_t *fam_first,
                          const int32_t *fam_count, int32_t n_families, const ramx_params *p,
                          ramx_run_info *infos, int8_t *cons, int32_t *trim_high, int32_t *trim_pos);

/* The same on the reference-like flat data model: family f = cores[f] on its own library sequence[f]; master[f] and
 * the cores' extension lengths / scores are updated exactly as ramx_extend_flat does for one family.  Families with
 * more than 512 extendable cores are run one by one.  Returns 0 or a negative error code; infos[f].ret is the per-family
 * return value of extend_alignment. */
typedef struct ramx_family
{
  ramx_flat_cores cores;
  const int8_t *sequence;
  uint64_t seq_len;
  int8_t *master;
} ramx_family;
int ramx_extend_batch(int direction, ramx_family *families, int32_t n_families, const ramx_params *p, ramx_run_info *infos);

/* multi-GPU: flanks are sharded over ranks; each column's 4 candidate sums are all-reduced
 * (4 x int64, RCCL over xGMI).  unique_id is the 128-byte ncclUniqueId made by rank 0
 * (ramx_comm_unique_id) and handed to the other ranks by the launcher (e.g. torch.distributed). */
int ramx_comm_unique_id(uint8_t id[128]);
int ramx_dev_comm_init(ramx_dev *d, const uint8_t id[128], int rank, int nranks);
/* ranks of the RCCL communicator as RCCL itself counts them (ncclCommCount); 1 without a communicator */
int ramx_dev_comm_size(ramx_dev *d);
/* Cross-device persistent path (optional, faster than one RCCL call per column): every rank exports the IPC handle
 * of its mailbox (64 bytes), the launcher all-gathers the handles, every rank imports them; a two-phase self-test
 * (phase 0: write tokens into all boxes -- synchronise the ranks -- phase 1: returns 1 if every rank's token arrived)
 * lets the launcher enable the path only if it works on ALL ranks.  With the path enabled the whole direction runs
 * as one persistent launch per rank and the per-column vote travels over xGMI from inside the kernels; if any rank
 * gives up (bounded spins) all ranks repeat the direction with per-column launches + RCCL. */
int ramx_dev_peer_export(ramx_dev *d, uint8_t handle[64]);
int ramx_dev_peer_import(ramx_dev *d, const uint8_t *handles /* [nranks][64] */, int rank, int nranks);
int ramx_dev_peer_selftest(ramx_dev *d, int phase, unsigned long long token);
int ramx_dev_peer_enable(ramx_dev *d, int on);

/* Host-memory variant of the same mailboxes (second choice when the device-memory boxes cannot be mapped into the
 * other processes or fail their self-test): every rank's box lives in one POSIX shared-memory segment that each
 * process registers with HIP; the kernels' system-scope stores and polls then cross PCIe.  All ranks of the node
 * call attach with the same name, run the self-test of ramx_dev_peer_selftest and ramx_dev_peer_enable as above;
 * afterwards rank 0 may unlink the name. */
int ramx_dev_hostbox_attach(ramx_dev *d, const char *shm_name, int rank, int nranks);
int ramx_hostbox_unlink(const char *shm_name);

/* test hook: replaces RCCL by a caller-supplied all-reduce so the sharded control flow (fold kernel, reduced
 * vote consumed by the next column, replicated stop rule) can be exercised where RCCL cannot run, e.g. two
 * ranks sharing the single GPU of a test box.  cb must sum 4 int64 in place across ranks and block until
 * done.  Slow by construction (one stream synchronisation per column); never used by bench.py. */
typedef void (*ramx_allreduce_cb)(long long *vals4, void *user);
int ramx_dev_set_allreduce_cb(ramx_dev *d, ramx_allreduce_cb cb, void *user);

/* ------------------------------------------------------------------------------------------
 * Scoring systems (reference score_system.h:23-37)
 * ------------------------------------------------------------------------------------------ */
struct scoringSystem *ramx_get_matrix(const char *matrixName);                       /* getMatrix */
struct scoringSystem *ramx_get_matrix_using_gap_penalties(const char *matrixName,
                                                          int gapopen, int gapextn); /* getMatrixUsingGapPenalties */
struct scoringSystem *ramx_get_repeatscout_matrix(int match, int mismatch, int gap); /* getRepeatScoutMatrix */
void ramx_free_scoring_system(struct scoringSystem *s);                              /* freeScoringSystem */
double ramx_calculate_lambda(struct scoringSystem *s);                               /* calculateLambda */

/* ------------------------------------------------------------------------------------------
 * Input surface (reference sequence.h:58-61): BED-6 ranges + 2bit -> library + cores
 * ------------------------------------------------------------------------------------------ */
struct sequenceLibrary *ramx_load_sequence_subset_minimal(const char *twoBitName, const char *rangeBEDName,
                                                          struct coreAlignment **core_align,
                                                          int *num_cores, int max_flanking_bp);
void ramx_free_library(struct sequenceLibrary *lib, struct coreAlignment *cores);

/* The same windows kept as the .2bit file stores them (replaces the expansion to one byte per base of
 * kentsrc/twoBitNew.c:531-613 + sequence.c:759,812-822): four bases per byte, first base in the most significant bits,
 * T C A G = 0 1 2 3; runs of N as (start, length) in library coordinates -- the coordinates of seqLib->sequence, i.e. what
 * coreAlignment.leftSeqPos / lowerSeqBound ... count in.  A quarter of the bytes to read, keep, upload and free. */
typedef struct ramx_packed_library
{
  uint64_t length;             /* bases */
  int32_t n_windows;
  const uint64_t *win_start;   /* [n_windows + 1] first base of window i; win_start[n_windows] = length */
  const uint64_t *win_byte;    /* [n_windows + 1] offset of window i's first packed byte in bytes[] */
  const uint8_t *win_phase;    /* [n_windows] position (0..3) of the window's first base inside that byte */
  const uint8_t *bytes;        /* the windows' bytes of the records' packed DNA, window after window */
  uint64_t n_bytes;
  const uint64_t *n_start;     /* runs of N, clipped to the windows, sorted */
  const uint32_t *n_len;
  int32_t n_blocks;
} ramx_packed_library;

/* loadSequenceSubsetMinimal without the expansion: lib->sequence is NULL, *packed (owned by the library, released by
 * ramx_free_library) holds the bases.  Everything else -- identifiers, boundaries, offsets, cores -- as above. */
struct sequenceLibrary *ramx_load_sequence_subset_packed(const char *twoBitName, const char *rangeBEDName,
                                                         struct coreAlignment **core_align, int *num_cores,
                                                         int max_flanking_bp, const ramx_packed_library **packed);
/* bases [from, from + count) as the reference's codes (A C G T = 0..3, N = 99): what seqLib->sequence would hold there */
int ramx_packed_decode(const ramx_packed_library *pl, uint64_t from, uint64_t count, char *out);

/* overlap avoidance between the two directions (reference ram_extend.c:445-499), prints the same lines */
void ramx_overlap_avoidance(struct coreAlignment *coreAlign, struct sequenceLibrary *seqLib);

/* report.c:161-502 */
void ramx_print_core_edges(struct coreAlignment *coreAlign, struct sequenceLibrary *seqLib,
                           char omitBlanks, char debug);

/* bnw_extend.c:87-155 -- link compatibility only (see top of file) */
int ****ramx_allocate_score(int num_align, int bandwidth);
void ramx_free_score(int num_align, int bandwidth, int ****score);

/* the CLI, callable as a function (RAMExtend's main(); reference ram_extend.c:217-790) */
int ramx_cli_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif
