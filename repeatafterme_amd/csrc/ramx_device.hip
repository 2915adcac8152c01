// ramx_device.hip -- seam 2 of include/ramx.h: the thin device API and the gfx950 kernels.
//
// Replaces, on the device, the reference's per-column work:
//   compute_nw_row            bnw_extend.c:750-1048   (banded affine row, 2 states per cell)
//   candidate vote + cap      ram_extend.c:973-1086
//   recompute-with-winner     ram_extend.c:1097-1168  (folded away: see "column step" below)
//   fit-preferred stop rule   ram_extend.c:1169-1223
//   boundary row              ram_extend.c:909-960
//
// Layout in HBM (Np = flanks padded to a multiple of 64, W = half band, B = 2W+1, Q = W+1):
//   bases  uint32 [KW][Np]        4-bit base classes, 8 per word, pre-oriented per flank so that
//                                 nibble (t' & 7) of word (t' >> 3) is the base aligned to band
//                                 cell (row r, offset o) with t' = o + r + W.  Transposed: the 64
//                                 lanes of a wave read 256 contiguous bytes.
//   state  int4   [Np/64][Q][64]  one DP row per flank: slot q holds cells 2q and 2q+1 as
//                                 (sub,gap,sub,gap); the last slot holds cell B-1 and (high,pos)
//                                 (overall_sequence_high_score[_pos], ram_extend.c:900-901).
//                                 Exactly 16*B+16 bytes per flank, read once and written once per
//                                 column: the algorithmic traffic of SURVEY.md section 8(d).
//   trim   int2   [Np]            trimmed_sequence_high_score[_pos] (ram_extend.c:902-903)
//   sums   int64  [3][32][4]      sharded per-candidate column sums (vote), rotated by column
//   ctl    RamxCtl[2]             stop-rule state, flip-flopped by column
//
// Column step K(r), one launch per consensus column, ONE LANE PER FLANK:
//   every wave:  fold the 32x4 vote shards of row r -> besta(r), new-max / stop decision
//   every lane:  stream its previous row S(r-1) (coalesced 16 B loads, 8 deep in flight),
//                compute row r against besta(r) -> S(r) (stored), track best cell -> high/pos,
//                and, skewed by one cell, the four candidate rows r+1 from S(r) (never stored) ->
//                capped contributions -> wave shuffle reduce -> LDS block reduce -> 4 int64 atomics
//                into one of 32 shards.
// The serial dependency through the insertion term (cells -W..+W in order) stays a plain serial
// loop inside the lane, so the values are the reference's bit for bit; parallelism comes from the
// flanks (N >= 64 k fills the chip).  No MFMA: integer max-plus recurrences, HBM-bound.

#include <rccl/rccl.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>


#include "ramx_kernels_common.h"
#include "ramx_kernels_stream.h"
#include "ramx_kernels_resident.h"
#include "ramx_cp_api.h"
#include "ramx_pk_api.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512];

extern "C" void ramx_set_error(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *ramx_last_error(void) { return g_err; }

#define HIPCHK(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      ramx_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #call); \
      return RAMX_ERR_HIP;                                                                        \
    }                                                                                             \
  } while (0)

static double now_ms(void);
// RAMX_TIMING: host-side marks inside a direction (ms since the mark before)
static void rt_mark(const char *what)
{
  static double last = 0;
  static int on = -1;
  if (on < 0) on = getenv("RAMX_TIMING") != NULL;
  if (!on) return;
  const double t = now_ms();
  if (what) fprintf(stderr, "RAMX_TIMING       run: %-44s %8.3f ms\n", what, t - last);
  last = t;
}
static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// ------------------------------------------------------------------------------------------
// host side of seam 2
// ------------------------------------------------------------------------------------------
#define RAMX_NGROUP (RAMX_CP_NCLASS + 4)   // batch mode: cell-parallel classes, then the four lane-per-flank workgroup shapes

struct ramx_dev
{
  int ordinal;
  hipStream_t stream;
  signed char *d_lib; unsigned long long lib_len; unsigned long long lib_cap;
  // packed library (ramx_dev_load_library_packed): the .2bit payload of the windows + window / N-run tables; lib_packed = 1
  int lib_packed;
  unsigned char *d_pk_bytes, *d_pk_phase; unsigned long long *d_pk_wstart, *d_pk_wbyte, *d_pk_nstart; unsigned *d_pk_nlen;
  size_t cap_pk_bytes, cap_pk_phase, cap_pk_wstart, cap_pk_wbyte, cap_pk_nstart, cap_pk_nlen, cap_flank_win;
  int pk_windows, pk_nblocks; int *d_flank_win;
  // per direction
  ramx_flank *d_flanks; unsigned *d_bases; int2 *d_bounds; int4 *d_state[2]; int2 *d_trim;
  long long *d_sums; RamxCtl *d_ctl; signed char *d_cons;
  PShard *d_vote; unsigned *d_err;   // persistent kernel: fused vote / barrier words
  int last_persistent;
  size_t cap_flanks, cap_bases, cap_state, cap_cons, cap_bounds, cap_trim, cap_fam, cap_famctl;
  void *d_fam; RamxCtl *d_famctl;    // batch mode: family descriptors / per-family control blocks (kept between calls)
  RamxCtl *h_ctl;   // pinned, [2 checkpoints][2 slots]
  void *h_stage; size_t cap_stage;   // pinned staging buffer of ramx_dev_download (a pageable 800 KB copy took 8 ms)
  hipEvent_t ev_chk[2], ev_begin, ev_end, ev_s0[MAX_SAMPLES], ev_s1[MAX_SAMPLES];
  int Nx, Np, KW;
  ramx_params p; int tab[RAMX_NCLASS][4];
  int ready;
  RamxCtl final_ctl;
  // multi-GPU
  ncclComm_t comm; int rank, nranks;
  ramx_allreduce_cb cb; void *cb_user;
  // cross-device persistent path: this rank's box (fine-grained), every rank's box as mapped here, device copy of that table
  PeerBox *xbox; PeerBox *peer[RAMX_MAX_RANKS]; PeerBox **d_peer; int peer_ready;
  void *peer_ipc[RAMX_MAX_RANKS];   // what hipIpcOpenMemHandle returned for the other ranks' boxes (closed on re-import / destroy)
  // host-memory variant of the boxes (POSIX shared memory registered with HIP): xbox/peer point into it
  void *hostbox_map; size_t hostbox_bytes; PeerBox *hostbox_host; int hostbox_registered;
  PeerBox *hostbox_mirror;   // device-memory copy of my host box, kept current by block 0 (the other blocks poll it)
  PeerBox *devbox;     // this rank's fine-grained device-memory box (exported over hipIpc)
  hipStream_t cls_stream[RAMX_NGROUP]; hipEvent_t cls_ready, cls_done[RAMX_NGROUP]; int cls_init;   // batch mode: one stream per workgroup shape
  int force_chain;   // RAMX_FORCE_CHAIN=1: always run the full candidate recurrence (test hook)
  ramx_row_trace_cb trace_cb; void *trace_user;   // -outmat: per-row trace (forces the per-column launches)
  ramx_row_verbose_cb verbose_cb; void *verbose_user;   // -vvvv: per-row candidate trace (per-column launches, full candidate recurrence)
  signed char *d_dbg_codes; int2 *d_dbg_best; size_t cap_dbg_codes, cap_dbg_best;
  int *d_dbg_cand; int2 *d_dbg_gap; size_t cap_dbg_cand, cap_dbg_gap;
  int *d_dbg_band; size_t cap_dbg_band; int verbose_band;      // -vvvvv: every cell of the candidate rows
  CpDevDesc *d_devdesc; size_t cap_devdesc;       // device-wide cell-parallel launches: one descriptor per workgroup
  PShard *d_vote_sets; size_t cap_vote_sets; unsigned *d_err_sets; size_t cap_err_sets;   // batch mode: per-set vote / error words
  int packed_kw;       // words of every window packed so far (begin_direction packs the first piece, see pack_rest)
  hipStream_t pack_stream; hipEvent_t pack_done; int pack_busy;      // the following pieces are packed beside the running one
  int pk_r0;           // begin_direction: first row from which no flank has a low out-of-bounds cell (the packed-row kernel starts there); -1: none
  int last_packed_r0;  // last direction: first row of the packed-row kernel, -1 if it did not run
  int cp_flanks_ok;    // begin_direction: every flank is empty or has t_lo <= 0 (what the cell-parallel kernels take)
  int2 *d_cpstate; size_t cap_cpstate; int cpstate_W, cpstate_n;   // RAMX_CP_PEEK=1: final rows of the cell-parallel kernel (tests)
};

extern "C" int ramx_device_count(void)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { ramx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return RAMX_ERR_NO_DEVICE; }
  return n;
}

extern "C" int ramx_dev_create(int ordinal, ramx_dev **out)
{
  int n = ramx_device_count();
  if (n <= 0) { ramx_set_error("no HIP device visible (libramx has no CPU path)"); return RAMX_ERR_NO_DEVICE; }
  if (ordinal < 0 || ordinal >= n) { ramx_set_error("device ordinal %d out of range (%d devices)", ordinal, n); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(ordinal));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, ordinal));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
  {
    ramx_set_error("device %d is %s; libramx carries gfx950 code objects only", ordinal, prop.gcnArchName);
    return RAMX_ERR_NO_DEVICE;
  }
  ramx_dev *d = (ramx_dev *)calloc(1, sizeof(ramx_dev));
  d->ordinal = ordinal;
  // any failure below releases what has been created so far (ramx_dev_destroy copes with a partly built session)
#define CRCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    ramx_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #call); ramx_dev_destroy(d); return RAMX_ERR_HIP; } } while (0)
  CRCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
  CRCHK(hipStreamCreateWithFlags(&d->pack_stream, hipStreamNonBlocking));
  CRCHK(hipEventCreateWithFlags(&d->pack_done, hipEventDisableTiming));
  CRCHK(hipHostMalloc((void **)&d->h_ctl, 4 * sizeof(RamxCtl), hipHostMallocDefault));
  // staging buffer of ramx_dev_download: allocated here (pinning takes milliseconds; the executable creates the session in its
  // helper thread), 2 MB = 262,144 flanks, grown on demand
  CRCHK(hipHostMalloc(&d->h_stage, (size_t)2 << 20, hipHostMallocDefault));
  d->cap_stage = (size_t)2 << 20;
  CRCHK(hipMalloc((void **)&d->d_sums, (3 * NSHARD * 4 + 8) * sizeof(long long)));
  CRCHK(hipMalloc((void **)&d->d_ctl, 2 * sizeof(RamxCtl)));
  CRCHK(hipMalloc((void **)&d->d_vote, PRK_NSETS * NSHARD * sizeof(PShard)));   // four rotating vote sets (both persistent kernels)
  CRCHK(hipMalloc((void **)&d->d_err, 64));

  for (int i = 0; i < 2; i++) CRCHK(hipEventCreate(&d->ev_chk[i]));
  CRCHK(hipEventCreate(&d->ev_begin));
  CRCHK(hipEventCreate(&d->ev_end));
  for (int i = 0; i < MAX_SAMPLES; i++) { CRCHK(hipEventCreate(&d->ev_s0[i])); CRCHK(hipEventCreate(&d->ev_s1[i])); }
#undef CRCHK
  const char *eb = getenv("RAMX_FORCE_CHAIN");
  d->force_chain = (eb && atoi(eb) != 0) ? 1 : 0;
  *out = d;
  return RAMX_OK;
}

extern "C" void ramx_dev_destroy(ramx_dev *d)
{
  if (!d) return;
  (void)hipSetDevice(d->ordinal);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->pack_stream) { (void)hipStreamSynchronize(d->pack_stream); (void)hipStreamDestroy(d->pack_stream); }
  if (d->pack_done) (void)hipEventDestroy(d->pack_done);
  if (d->comm) (void)ncclCommDestroy(d->comm);
  // mappings of the other ranks' mailboxes (hipIpcOpenMemHandle): closed before anything of this device is released
  for (int q = 0; q < RAMX_MAX_RANKS; q++)
    if (d->peer_ipc[q]) { (void)hipIpcCloseMemHandle(d->peer_ipc[q]); d->peer_ipc[q] = NULL; }
  (void)hipFree(d->d_pk_bytes); (void)hipFree(d->d_pk_phase); (void)hipFree(d->d_pk_wstart); (void)hipFree(d->d_pk_wbyte);
  (void)hipFree(d->d_pk_nstart); (void)hipFree(d->d_pk_nlen); (void)hipFree(d->d_flank_win);
  (void)hipFree(d->d_lib); (void)hipFree(d->d_flanks); (void)hipFree(d->d_bases); (void)hipFree(d->d_bounds);
  (void)hipFree(d->d_state[0]); if (d->d_state[1] != d->d_state[0]) (void)hipFree(d->d_state[1]); (void)hipFree(d->d_trim); (void)hipFree(d->d_sums);
  (void)hipFree(d->d_ctl); (void)hipFree(d->d_cons); (void)hipFree(d->d_vote); (void)hipFree(d->d_err);
  if (d->h_ctl) (void)hipHostFree(d->h_ctl);
  if (d->h_stage) (void)hipHostFree(d->h_stage);
  if (d->hostbox_map)
  {
    if (d->hostbox_registered) (void)hipHostUnregister(d->hostbox_map);
    munmap(d->hostbox_map, d->hostbox_bytes);
  }
  if (d->cls_init)
  {
    for (int c = 0; c < RAMX_NGROUP; c++) { (void)hipStreamDestroy(d->cls_stream[c]); (void)hipEventDestroy(d->cls_done[c]); }
    (void)hipEventDestroy(d->cls_ready);
  }
  if (d->devbox) (void)hipFree(d->devbox);
  (void)hipFree(d->d_fam); (void)hipFree(d->d_famctl); (void)hipFree(d->d_cpstate); (void)hipFree(d->d_dbg_codes); (void)hipFree(d->d_dbg_best); (void)hipFree(d->d_dbg_cand); (void)hipFree(d->d_dbg_gap); (void)hipFree(d->d_dbg_band); (void)hipFree(d->d_devdesc); (void)hipFree(d->d_vote_sets); (void)hipFree(d->d_err_sets);
  if (d->hostbox_mirror) (void)hipFree(d->hostbox_mirror);
  if (d->d_peer) (void)hipFree(d->d_peer);
  for (int i = 0; i < 2; i++) if (d->ev_chk[i]) (void)hipEventDestroy(d->ev_chk[i]);
  if (d->ev_begin) (void)hipEventDestroy(d->ev_begin);
  if (d->ev_end) (void)hipEventDestroy(d->ev_end);
  for (int i = 0; i < MAX_SAMPLES; i++) { if (d->ev_s0[i]) (void)hipEventDestroy(d->ev_s0[i]); if (d->ev_s1[i]) (void)hipEventDestroy(d->ev_s1[i]); }
  if (d->stream) (void)hipStreamDestroy(d->stream);
  free(d);
}

extern "C" int ramx_dev_load_library(ramx_dev *d, const int8_t *sequence, uint64_t length)
{
  if (!d || (!sequence && length)) { ramx_set_error("ramx_dev_load_library: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  if (length > d->lib_cap)
  {
    if (d->d_lib) HIPCHK(hipFree(d->d_lib));
    d->d_lib = NULL;
    HIPCHK(hipMalloc((void **)&d->d_lib, length ? length : 1));
    d->lib_cap = length;
  }
  if (length) HIPCHK(hipMemcpyAsync(d->d_lib, sequence, length, hipMemcpyHostToDevice, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  d->lib_len = length;
  d->lib_packed = 0;
  return RAMX_OK;
}

template <typename T>
static int ensure(T **p, size_t *cap, size_t need_bytes);

extern "C" int ramx_dev_load_library_packed(ramx_dev *d, const struct ramx_packed_library *pl)
{
  if (!d || !pl || pl->n_windows < 0 || (pl->n_windows && (!pl->win_start || !pl->win_byte || !pl->win_phase || !pl->bytes)) ||
      pl->n_blocks < 0 || (pl->n_blocks && (!pl->n_start || !pl->n_len)))
  { ramx_set_error("ramx_dev_load_library_packed: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  const size_t nw = (size_t)pl->n_windows, nb = (size_t)pl->n_blocks;
  int rc;
  if ((rc = ensure(&d->d_pk_bytes, &d->cap_pk_bytes, (size_t)pl->n_bytes + 16))) return rc;
  if ((rc = ensure(&d->d_pk_wstart, &d->cap_pk_wstart, (nw + 1) * sizeof(unsigned long long)))) return rc;
  if ((rc = ensure(&d->d_pk_wbyte, &d->cap_pk_wbyte, (nw + 1) * sizeof(unsigned long long)))) return rc;
  if ((rc = ensure(&d->d_pk_phase, &d->cap_pk_phase, nw + 1))) return rc;
  if ((rc = ensure(&d->d_pk_nstart, &d->cap_pk_nstart, (nb + 1) * sizeof(unsigned long long)))) return rc;
  if ((rc = ensure(&d->d_pk_nlen, &d->cap_pk_nlen, (nb + 1) * sizeof(unsigned)))) return rc;
  if (pl->n_bytes) HIPCHK(hipMemcpyAsync(d->d_pk_bytes, pl->bytes, (size_t)pl->n_bytes, hipMemcpyHostToDevice, d->stream));
  if (nw)
  {
    HIPCHK(hipMemcpyAsync(d->d_pk_wstart, pl->win_start, (nw + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(d->d_pk_wbyte, pl->win_byte, (nw + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(d->d_pk_phase, pl->win_phase, nw, hipMemcpyHostToDevice, d->stream));
  }
  if (nb)
  {
    HIPCHK(hipMemcpyAsync(d->d_pk_nstart, pl->n_start, nb * sizeof(unsigned long long), hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(d->d_pk_nlen, pl->n_len, nb * sizeof(unsigned), hipMemcpyHostToDevice, d->stream));
  }
  HIPCHK(hipStreamSynchronize(d->stream));
  d->lib_len = pl->length; d->pk_windows = pl->n_windows; d->pk_nblocks = pl->n_blocks;
  d->lib_packed = 1;
  return RAMX_OK;
}

// flank descriptors (already on the device) -> transposed, pre-oriented 4-bit windows + bounds, from either kind of library:
// words [k_lo, k_hi) of every flank's window, on stream `st`
static int launch_pack(ramx_dev *d, int Nx, int Np, int W, int k_lo, int k_hi, hipStream_t st)
{
  if (k_hi <= k_lo) return RAMX_OK;
  if (!d->lib_packed)
  {
    dim3 grid((Np + 255) / 256, k_hi - k_lo);
    hipLaunchKernelGGL(ramx_pack_kernel, grid, dim3(256), 0, st, d->d_lib, (unsigned long long)d->lib_len,
                       d->d_flanks, Nx, Np, W, k_lo, d->d_bases, d->d_bounds);
    HIPCHK(hipGetLastError());
    return RAMX_OK;
  }
  PkLib L;
  L.bytes = d->d_pk_bytes; L.win_start = d->d_pk_wstart; L.win_byte = d->d_pk_wbyte; L.win_phase = d->d_pk_phase;
  L.n_start = d->d_pk_nstart; L.n_len = d->d_pk_nlen; L.length = d->lib_len; L.n_windows = d->pk_windows; L.n_blocks = d->pk_nblocks;
  if (k_lo == 0)
  {
    int rc;
    if ((rc = ensure(&d->d_flank_win, &d->cap_flank_win, (size_t)Np * sizeof(int)))) return rc;
    if (Nx > 0)
    {
      hipLaunchKernelGGL(ramx_flank_window_kernel, dim3((Nx + 255) / 256), dim3(256), 0, st, L, d->d_flanks, Nx, d->d_flank_win);
      HIPCHK(hipGetLastError());
    }
  }
  hipLaunchKernelGGL(ramx_pack2_kernel, dim3((Np + 255) / 256, (k_hi - k_lo + RAMX_PK_WORDS - 1) / RAMX_PK_WORDS), dim3(256), 0, st, L, d->d_flanks,
                     d->d_flank_win, Nx, Np, W, k_lo, k_hi, d->d_bases, d->d_bounds);
  HIPCHK(hipGetLastError());
  return RAMX_OK;
}

// A direction's windows are L + 2W columns long, and most runs stop after a fraction of them (the reference's default: 100 columns
// behind the end of the alignment): begin_direction packs the first RAMX_PK_SEGMENT columns, the packed-row route packs the
// following pieces on a second stream while the piece before them runs (prk_run), and every other route packs the rest here.
static int pack_rest(ramx_dev *d)
{
  if (d->packed_kw >= d->KW) return RAMX_OK;
  if (d->pack_busy) { HIPCHK(hipStreamWaitEvent(d->stream, d->pack_done, 0)); d->pack_busy = 0; }
  const int rc = launch_pack(d, d->Nx, d->Np, d->p.bandwidth, d->packed_kw, d->KW, d->stream);
  d->packed_kw = d->KW;
  return rc;
}
static int pk_segment_columns(int L, int when_to_stop)
{
  const char *e = getenv("RAMX_PK_SEGMENT");
  // a direction whose stop rule cannot fire before its last column (-stopafter >= L: every column is wanted) reads its whole
  // window anyway: one piece
  const int v = e ? atoi(e) : (when_to_stop >= L ? 0 : 2048);
  return v < 0 ? 0 : v;          // 0: the whole direction in one piece (everything packed at once)
}

template <typename T>
static int ensure(T **p, size_t *cap, size_t need_bytes)
{
  if (need_bytes <= *cap && *p) return RAMX_OK;
  if (*p) HIPCHK(hipFree(*p));
  *p = NULL;
  HIPCHK(hipMalloc((void **)p, need_bytes ? need_bytes : 16));
  *cap = need_bytes;
  return RAMX_OK;
}

extern "C" int ramx_dev_begin_direction(ramx_dev *d, const ramx_flank *flanks, int32_t n_flanks, const ramx_params *p)
{
  if (!d || !p || n_flanks < 0 || (n_flanks && !flanks)) { ramx_set_error("ramx_dev_begin_direction: bad argument"); return RAMX_ERR_ARG; }
  if (p->bandwidth < 0 || p->L < 0) { ramx_set_error("bandwidth and L must be >= 0"); return RAMX_ERR_ARG; }
  if (!p->matrix) { ramx_set_error("scoring matrix missing"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  rt_mark(NULL);
  const int W = p->bandwidth, Q = W + 1;
  const int Nx = n_flanks;
  const int Np = ((Nx + 63) / 64) * 64 > 0 ? ((Nx + 63) / 64) * 64 : 64;
  // t'' = o + r + W + 8 runs over [7, L + 2W + 9]
  const int KW = (p->L + 2 * W + 2) / 8 + 12;  // + pad word in front, + lookahead words read by the kernels
  d->Nx = Nx; d->Np = Np; d->KW = KW; d->p = *p;
  // class table: tab[class][cand] = matrix[cand][code(class)], reference index order [cons][seq]
  for (int c = 0; c < RAMX_NCLASS; c++)
  {
    const int code = (c == 8) ? RAMX_SYM_N : c;
    for (int k = 0; k < 4; k++) d->tab[c][k] = p->matrix[k * 100 + code];
  }
  int rc;
  if ((rc = ensure(&d->d_flanks, &d->cap_flanks, (size_t)Np * sizeof(ramx_flank)))) return rc;
  if ((rc = ensure(&d->d_bases, &d->cap_bases, (size_t)KW * Np * sizeof(unsigned)))) return rc;
  if ((rc = ensure(&d->d_bounds, &d->cap_bounds, (size_t)Np * sizeof(int2)))) return rc;   // kept between calls: hipFree/hipMalloc
  if ((rc = ensure(&d->d_trim, &d->cap_trim, (size_t)Np * sizeof(int2)))) return rc;       // per call cost ~0.5 ms of device sync
  const size_t state_bytes = (size_t)Np * Q * sizeof(int4);
  if (state_bytes > d->cap_state || !d->d_state[0])
  {
    // The row is updated IN PLACE: a lane only ever touches its own 16-byte column of the tile, reads slot q+16
    // before it writes slot q, and launches are serialised, so one buffer serves as S(r-1) and S(r).  Halving the
    // footprint (65.6 MB at N = 100,000) keeps the whole row set inside the Infinity Cache.
    const bool pingpong = getenv("RAMX_PINGPONG") != NULL;     // A/B switch kept for profiling
    if (d->d_state[0]) HIPCHK(hipFree(d->d_state[0]));
    if (d->d_state[1] && d->d_state[1] != d->d_state[0]) HIPCHK(hipFree(d->d_state[1]));
    d->d_state[0] = d->d_state[1] = NULL;
    HIPCHK(hipMalloc((void **)&d->d_state[0], state_bytes));
    if (pingpong) HIPCHK(hipMalloc((void **)&d->d_state[1], state_bytes));
    else d->d_state[1] = d->d_state[0];
    d->cap_state = state_bytes;
  }
  if ((rc = ensure(&d->d_cons, &d->cap_cons, (size_t)p->L + 16))) return rc;
  rt_mark("begin: buffers");
  if (Nx) HIPCHK(hipMemcpyAsync(d->d_flanks, flanks, (size_t)Nx * sizeof(ramx_flank), hipMemcpyHostToDevice, d->stream));
  rt_mark("begin: flanks to the device (enqueued)");
  {
    if (d->pack_busy) { HIPCHK(hipStreamWaitEvent(d->stream, d->pack_done, 0)); d->pack_busy = 0; }    // (a piece of the direction before still in flight)
    const int seg = pk_segment_columns(p->L, p->when_to_stop);
    const int first = seg > 0 ? ((seg + 8) >> 3) + 28 : KW;      // words the first piece's columns read (+ the widest band's window and look-ahead)
    d->packed_kw = first < KW ? first : KW;
    if ((rc = launch_pack(d, Nx, Np, W, 0, d->packed_kw, d->stream)) != RAMX_OK) return rc;
  }
  HIPCHK(hipMemsetAsync(d->d_sums, 0, 3 * NSHARD * 4 * sizeof(long long), d->stream));
  HIPCHK(hipMemsetAsync(d->d_cons, 0, (size_t)p->L + 16, d->stream));
  rt_mark("begin: pack of the first piece enqueued");
  HIPCHK(hipStreamSynchronize(d->stream));
  rt_mark("begin: synchronised");
  // the cell-parallel kernels take flanks that are empty or start at or before the first base behind the core edge
  d->cp_flanks_ok = 1;
  for (int i = 0; i < Nx; i++)
    if (flanks[i].t_lo > 0 && flanks[i].t_lo <= flanks[i].t_hi) { d->cp_flanks_ok = 0; break; }
  // the packed-row kernel (ramx_kernels_packed.h) takes rows in which no flank has an out-of-bounds cell at the LOW end of the
  // band: cell 0 of row r is flank position t = r - W, so a non-empty flank is clear from row t_lo + W on
  d->pk_r0 = d->cp_flanks_ok ? 0 : -1;
  if (d->cp_flanks_ok)
    for (int i = 0; i < Nx; i++)
      if (flanks[i].t_lo <= flanks[i].t_hi && flanks[i].t_lo + W > d->pk_r0) d->pk_r0 = flanks[i].t_lo + W;
  d->ready = 1;
  return RAMX_OK;
}

template <bool INIT>
static void launch_column(ramx_dev *d, const KArgs &a)
{
  const int tiles = d->Np / 64;
  const dim3 grid((tiles + 3) / 4), block(256);
  // the chain-free candidate evaluation is exact iff neither gap penalty is positive (see the kernel header)
  if (a.go <= 0 && a.ge <= 0 && !d->force_chain)
    hipLaunchKernelGGL((ramx_column_kernel<INIT, false, 256>), grid, block, 0, d->stream, a);
  else
    hipLaunchKernelGGL((ramx_column_kernel<INIT, true, 256>), grid, block, 0, d->stream, a);
}

// -outmat trace: the DBG instantiation also writes the per-cell path codes and the row's best cell of every flank;
// -vvvv (a.dbg_cand != NULL): the full candidate recurrence, whose rows' best cells and end-cell gap states are written too
static void launch_column_trace(ramx_dev *d, const KArgs &a, bool init = false)
{
  const int tiles = d->Np / 64;
  const dim3 grid((tiles + 3) / 4), block(256);
  if (init) { hipLaunchKernelGGL((ramx_column_kernel<true, true, 256, true>), grid, block, 0, d->stream, a); return; }
  if (a.dbg_cand != NULL) { hipLaunchKernelGGL((ramx_column_kernel<false, true, 256, true>), grid, block, 0, d->stream, a); return; }
  if (a.go <= 0 && a.ge <= 0 && !d->force_chain)
    hipLaunchKernelGGL((ramx_column_kernel<false, false, 256, true>), grid, block, 0, d->stream, a);
  else
    hipLaunchKernelGGL((ramx_column_kernel<false, true, 256, true>), grid, block, 0, d->stream, a);
}

// does the vote of a column cross ranks?  (a communicator of ONE rank -- RAMX_COMM_SINGLE=1 at ramx_dev_comm_init -- counts with
// RAMX_FORCE_COLLECTIVE=1: the RCCL calls of the per-column route then run on a single GPU, tests/test_gpu_sharded.py)
static bool dev_is_multi(const ramx_dev *d)
{
  return (d->comm != NULL && (d->nranks > 1 || getenv("RAMX_FORCE_COLLECTIVE") != NULL)) || d->cb != NULL;
}

// ---- host-side collective on 4 x int64 in device memory (RCCL, or the test hook) ---------------
static int host_allreduce_shards(ramx_dev *d, long long *dptr)   // dptr: NSHARD x 4 int64, summed over ranks in place
{
  if (d->cb)
  {
    long long h[NSHARD * 4], h4[4] = { 0, 0, 0, 0 };
    HIPCHK(hipMemcpyAsync(h, dptr, sizeof(h), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    for (int i = 0; i < NSHARD * 4; i++) h4[i & 3] += h[i];
    d->cb(h4, d->cb_user);
    memset(h, 0, sizeof(h));
    memcpy(h, h4, sizeof(h4));
    HIPCHK(hipMemcpyAsync(dptr, h, sizeof(h), hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    return RAMX_OK;
  }
  ncclResult_t nr = ncclAllReduce(dptr, dptr, NSHARD * 4, ncclInt64, ncclSum, d->comm, d->stream);
  if (nr != ncclSuccess) { ramx_set_error("ncclAllReduce: %s", ncclGetErrorString(nr)); return RAMX_ERR_COMM; }
  return RAMX_OK;
}

// one small integer agreed over all ranks (max): used to agree on a fallback
static int host_allreduce_flag(ramx_dev *d, int *flag)
{
  if (d->cb)
  {
    long long h4[4] = { *flag ? 1 : 0, 0, 0, 0 };
    d->cb(h4, d->cb_user);
    *flag = h4[0] != 0;
    return RAMX_OK;
  }
  long long *tmp = d->d_sums + 3 * NSHARD * 4;   // spare 4 words behind the three vote slots
  long long h = *flag ? 1 : 0;
  HIPCHK(hipMemcpyAsync(tmp, &h, sizeof(h), hipMemcpyHostToDevice, d->stream));
  ncclResult_t nr = ncclAllReduce(tmp, tmp, 1, ncclInt64, ncclMax, d->comm, d->stream);
  if (nr != ncclSuccess) { ramx_set_error("ncclAllReduce: %s", ncclGetErrorString(nr)); return RAMX_ERR_COMM; }
  HIPCHK(hipMemcpyAsync(&h, tmp, sizeof(h), hipMemcpyDeviceToHost, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  *flag = h != 0;
  return RAMX_OK;
}

// ---- peer boxes ------------------------------------------------------------------------------------
extern "C" int ramx_dev_peer_export(ramx_dev *d, uint8_t handle[64])
{
  if (!d || !handle) { ramx_set_error("ramx_dev_peer_export: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  if (!d->devbox)
  {
    hipError_t e = hipExtMallocWithFlags((void **)&d->devbox, sizeof(PeerBox), hipDeviceMallocFinegrained);
    if (e != hipSuccess) { d->devbox = NULL; ramx_set_error("fine-grained allocation for the peer box failed: %s", hipGetErrorString(e)); return RAMX_ERR_HIP; }
    HIPCHK(hipMemset(d->devbox, 0, sizeof(PeerBox)));
  }
  hipIpcMemHandle_t h;
  HIPCHK(hipIpcGetMemHandle(&h, d->devbox));
  static_assert(sizeof(hipIpcMemHandle_t) <= 64, "ipc handle size");
  memset(handle, 0, 64);
  memcpy(handle, &h, sizeof(h));
  return RAMX_OK;
}

__global__ void ramx_peer_token_kernel(PeerBox *const *peers, int rank, int nranks, unsigned long long token)
{
  if (threadIdx.x < nranks)
    __hip_atomic_store(&peers[threadIdx.x]->token[rank], token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int ramx_dev_peer_import(ramx_dev *d, const uint8_t *handles, int rank, int nranks)
{
  if (!d || !handles || rank < 0 || rank >= nranks || nranks > RAMX_MAX_RANKS || !d->devbox)
  { ramx_set_error("ramx_dev_peer_import: bad argument (export first; at most %d ranks)", RAMX_MAX_RANKS); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  d->peer_ready = 0;
  d->rank = rank; d->nranks = nranks;
  d->xbox = d->devbox; d->hostbox_host = NULL;
  for (int q = 0; q < RAMX_MAX_RANKS; q++)
    if (d->peer_ipc[q]) { (void)hipIpcCloseMemHandle(d->peer_ipc[q]); d->peer_ipc[q] = NULL; }
  for (int q = 0; q < nranks; q++)
  {
    if (q == rank) { d->peer[q] = d->xbox; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + 64 * q, sizeof(h));
    void *ptr = NULL;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { ramx_set_error("hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e)); return RAMX_ERR_HIP; }
    d->peer[q] = (PeerBox *)ptr;
    d->peer_ipc[q] = ptr;
  }
  if (!d->d_peer) HIPCHK(hipMalloc((void **)&d->d_peer, RAMX_MAX_RANKS * sizeof(PeerBox *)));
  HIPCHK(hipMemcpy(d->d_peer, d->peer, nranks * sizeof(PeerBox *), hipMemcpyHostToDevice));
  return RAMX_OK;
}

/* Host-memory mailboxes: the same PeerBox protocol with every rank's box in ONE POSIX shared-memory segment that each
 * process registers with HIP (hipHostRegisterMapped).  Stores and polls cross PCIe instead of xGMI; no IPC handles,
 * no peer access between devices needed.  Used when the device-memory boxes cannot be mapped or fail their
 * self-test.  Every rank calls attach with the same name (ranks of one node); after a barrier rank 0 may unlink. */
extern "C" int ramx_dev_hostbox_attach(ramx_dev *d, const char *shm_name, int rank, int nranks)
{
  if (!d || !shm_name || rank < 0 || rank >= nranks || nranks > RAMX_MAX_RANKS)
  { ramx_set_error("ramx_dev_hostbox_attach: bad argument (at most %d ranks)", RAMX_MAX_RANKS); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  d->peer_ready = 0;
  const size_t bytes = (((size_t)nranks * sizeof(PeerBox)) + 4095) & ~(size_t)4095;
  // rank 0 creates the segment exclusively (a leftover or pre-created segment of that name is an error, never silently
  // adopted) and sizes it; the other ranks open it -- the caller synchronises the ranks between rank 0's call and theirs
  int fd = rank == 0 ? shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(shm_name, O_RDWR, 0600);
  if (fd < 0) { ramx_set_error("shm_open(%s): %s", shm_name, strerror(errno)); return RAMX_ERR_COMM; }
  if (rank == 0 && ftruncate(fd, (off_t)bytes) != 0) { ramx_set_error("ftruncate(%s): %s", shm_name, strerror(errno)); close(fd); shm_unlink(shm_name); return RAMX_ERR_COMM; }
  void *map = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (map == MAP_FAILED) { ramx_set_error("mmap(%s): %s", shm_name, strerror(errno)); return RAMX_ERR_COMM; }
  hipError_t e = hipHostRegister(map, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (e != hipSuccess) { munmap(map, bytes); ramx_set_error("hipHostRegister(shared boxes): %s", hipGetErrorString(e)); return RAMX_ERR_HIP; }
  void *dptr = NULL;
  e = hipHostGetDevicePointer(&dptr, map, 0);
  if (e != hipSuccess) { (void)hipHostUnregister(map); munmap(map, bytes); ramx_set_error("hipHostGetDevicePointer: %s", hipGetErrorString(e)); return RAMX_ERR_HIP; }
  if (d->hostbox_map) { if (d->hostbox_registered) (void)hipHostUnregister(d->hostbox_map); munmap(d->hostbox_map, d->hostbox_bytes); }
  d->hostbox_map = map; d->hostbox_bytes = bytes; d->hostbox_registered = 1;
  d->hostbox_host = (PeerBox *)map + rank;
  memset(d->hostbox_host, 0, sizeof(PeerBox));          // my own box; the others clear theirs
  d->rank = rank; d->nranks = nranks;
  // the device-memory box (if any) is no longer the exchange target
  d->xbox = (PeerBox *)dptr + rank;
  for (int q = 0; q < nranks; q++) d->peer[q] = (PeerBox *)dptr + q;
  if (!d->d_peer) HIPCHK(hipMalloc((void **)&d->d_peer, RAMX_MAX_RANKS * sizeof(PeerBox *)));
  HIPCHK(hipMemcpy(d->d_peer, d->peer, nranks * sizeof(PeerBox *), hipMemcpyHostToDevice));
  return RAMX_OK;
}

extern "C" int ramx_hostbox_unlink(const char *shm_name)
{
  if (!shm_name) return RAMX_ERR_ARG;
  return shm_unlink(shm_name) == 0 ? RAMX_OK : RAMX_ERR_COMM;
}

/* self-test of the mapped boxes: write a token into every rank's box, then (after the caller has synchronised the
 * ranks) check that every rank's token arrived here.  phase 0 = write, phase 1 = check (returns 1 if all there). */
extern "C" int ramx_dev_peer_selftest(ramx_dev *d, int phase, unsigned long long token)
{
  if (!d || !d->d_peer) { ramx_set_error("ramx_dev_peer_selftest: import first"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  if (phase == 0)
  {
    hipLaunchKernelGGL(ramx_peer_token_kernel, dim3(1), dim3(64), 0, d->stream, (PeerBox *const *)d->d_peer, d->rank, d->nranks, token);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->stream));
    return RAMX_OK;
  }
  PeerBox h;
  for (int tries = 0; tries < 200; tries++)
  {
    if (d->hostbox_host) memcpy(&h, (const void *)d->hostbox_host, sizeof(h));
    else HIPCHK(hipMemcpy(&h, d->xbox, sizeof(h), hipMemcpyDeviceToHost));
    int ok = 1;
    for (int q = 0; q < d->nranks; q++) ok &= (h.token[q] == token);
    if (ok) return 1;
    struct timespec ts = { 0, 1000000 };
    nanosleep(&ts, NULL);
  }
  return 0;
}

extern "C" int ramx_dev_peer_enable(ramx_dev *d, int on)
{
  if (!d) return RAMX_ERR_ARG;
  d->peer_ready = (on && d->d_peer && d->xbox) ? 1 : 0;
  return RAMX_OK;
}

// The fast band packs (score << 4 | cell) keys: exact while every reachable |score| < 2^27 (a row-r cell is at most
// (r + W + 2) steps of at most max(|matrix|, |go| + |ge|) away from 0), and reads matrix entries as int8.  Scoring
// systems outside these bounds take the general (masked-path) band for every wave.
static int fast_pack_ok(const int (&tab)[RAMX_NCLASS][4], int go, int ge, int L, int W)
{
  long long mx = (long long)(go < 0 ? -go : go) + (long long)(ge < 0 ? -ge : ge);
  for (int c = 0; c < RAMX_NCLASS; c++)
    for (int k = 0; k < 4; k++)
    {
      const long long v = tab[c][k] < 0 ? -(long long)tab[c][k] : (long long)tab[c][k];
      if (v > mx) mx = v;
    }
  for (int c = 0; c < RAMX_NCLASS; c++)
    for (int k = 0; k < 4; k++)
      if (tab[c][k] < -128 || tab[c][k] > 127) return 0;       // the fast tables hold the candidates' scores as int8
  return ((long long)L + 2LL * W + 4) * mx < (1LL << 27) ? 1 : 0;
}

// P of the LEAN test of the register-resident bands (ramx_kernels_resident.h, prk_band_fast): max(0, largest matrix entry);
// -1 (never LEAN) with RAMX_NO_LEAN (A/B and tests) or when a gap penalty is positive
static int lean_p_of(const int (&tab)[RAMX_NCLASS][4], int go, int ge)
{
  if (getenv("RAMX_NO_LEAN") != NULL || go > 0 || ge > 0) return -1;
  int mx = 0;
  for (int c = 0; c < RAMX_NCLASS; c++)
    for (int k = 0; k < 4; k++) if (tab[c][k] > mx) mx = tab[c][k];
  return mx;
}

// ---- persistent path --------------------------------------------------------------------------
// Block shape of the persistent launch: at most ONE barrier participant per CU.
//   <= 4 tiles per CU (N <= 65,536): 256-thread blocks, one wave per SIMD, up to 256 blocks;
//   otherwise 512-thread blocks (two waves per SIMD), up to 256 blocks = 131,072 flanks.
template <int W, int BLOCK>
static int prk_capacity_blocks(int *out)
{
  static int cached = -1;              // (one device per process; the query is a runtime call per direction otherwise)
  if (cached >= 0) { *out = cached; return RAMX_OK; }
  int per_cu = 0, dev = 0, cus = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ramx_persistent_kernel<W, BLOCK>, BLOCK, 0));
  if (per_cu > 1) per_cu = 1;          // one block per CU by design; never trust the API for more
  *out = per_cu * cus;
  cached = *out;
  return RAMX_OK;
}

static void test_delay_rank(const ramx_dev *d)
{
  // test hook: RAMX_TEST_DELAY_RANK="rank:milliseconds" holds that rank back just before its single launch, so that its
  // peers' kernels wait for its first words
  const char *e = getenv("RAMX_TEST_DELAY_RANK");
  int r = -1, ms = 0;
  if (e && sscanf(e, "%d:%d", &r, &ms) == 2 && r == d->rank && ms > 0) usleep((useconds_t)ms * 1000);
}

template <int W, int BLOCK>
static int prk_launch(ramx_dev *d, PArgs &pa, int blocks)
{
  // A PLAIN launch.  hipLaunchCooperativeKernel buys only the launch-time comparison of the grid with the occupancy query
  // (MI355X_MICROARCH.md, "coop-launch") -- prk_plan has made that comparison already (at most one workgroup per CU, never
  // more workgroups than CUs) -- and costs 15-19 us of host time per launch; the kernel has its own bounded barrier and
  // never calls grid.sync().  It also made every process that was profiled with rocprofv3 die with SIGSEGV inside exit():
  // the runtime's cooperative queue is torn down after the tool has finalised (tools/rocprof_exit_probe.sh isolates it:
  // streaming launches exit cleanly, one cooperative launch does not, with or without ramx_dev_destroy).
  // RAMX_COOP_LAUNCH=1 restores the cooperative launch.
  if (pa.nranks > 1) test_delay_rank(d);
  if (getenv("RAMX_COOP_LAUNCH") != NULL)
  {
    void *args[] = { (void *)&pa };
    HIPCHK(hipLaunchCooperativeKernel((const void *)ramx_persistent_kernel<W, BLOCK>, dim3(blocks), dim3(BLOCK), args, 0, d->stream));
    return RAMX_OK;
  }
  hipLaunchKernelGGL((ramx_persistent_kernel<W, BLOCK>), dim3(blocks), dim3(BLOCK), 0, d->stream, pa);
  HIPCHK(hipGetLastError());
  return RAMX_OK;
}

// block shape that keeps the whole flank set resident, or 0
template <int W>
static int prk_plan(int tiles, int *block, int *blocks)
{
  int cap = 0, rc;
  *block = 0; *blocks = 0;
  if ((rc = prk_capacity_blocks<W, 256>(&cap)) != RAMX_OK) return rc;
  if ((tiles + 3) / 4 <= cap)
  {
    *block = 256; *blocks = (tiles + 3) / 4;
    return RAMX_OK;
  }
  if constexpr (W <= 40)     // wider bands have no two-waves-per-SIMD shape: the row needs more than 256 registers
  {
    if ((rc = prk_capacity_blocks<W, 512>(&cap)) != RAMX_OK) return rc;
    if ((tiles + 7) / 8 <= cap) { *block = 512; *blocks = (tiles + 7) / 8; }
  }
  return RAMX_OK;
}

// which band widths have a register-resident instantiation
static bool prk_has_width(int W) { return W == 14 || W == 20 || W == 40 || W == 80; }
// ... and which have a one-workgroup-per-family instantiation (ramx_family_kernel: two waves per SIMD, 256 registers)
static bool fam_has_width(int W) { return W == 14 || W == 20 || W == 40; }

// this rank's own answer to "can the lane-per-flank persistent kernel run this direction", and its launch shape
// A few words from device memory to the host at the end of a launch (control blocks, the error word, the consensus): written by
// a kernel into the session's pinned buffer.  hipMemcpy would do, but the first device-to-host copy of a process costs
// milliseconds of DMA set-up on this stack (6-10 ms of a 33 ms direction at N = 100,000), a store over the bus none.
// At most 64 KB (the tail of the staging buffer; the trim records use its head); the kernel moves whole 8-byte words (a source
// that is not a multiple of 8 bytes is over-read by up to 7 bytes, inside its allocation's granule).
static int d2h_small(ramx_dev *d, void *dst, const void *src, size_t bytes)
{
  const size_t off = d->cap_stage - ((size_t)64 << 10);
  if (bytes == 0) return RAMX_OK;
  if (bytes > ((size_t)64 << 10) || d->h_stage == NULL || d->cap_stage < ((size_t)128 << 10) || getenv("RAMX_DOWNLOAD_DMA") != NULL)
  {
    HIPCHK(hipStreamSynchronize(d->stream));
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return RAMX_OK;
  }
  int2 *stage = (int2 *)((char *)d->h_stage + off);
  const int n = (int)((bytes + 7) / 8);
  hipLaunchKernelGGL(ramx_to_host_kernel, dim3((n + 255) / 256), dim3(256), 0, d->stream, (const int2 *)src, stage, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(d->stream));
  memcpy(dst, stage, bytes);
  return RAMX_OK;
}

#define RAMX_CP_MIN_K_OVER_PACKED 4      // the cell-parallel kernel goes first with at least this many lanes per flank
static int prk_local_can(ramx_dev *d, const KArgs &a, int L, bool multi, bool *can_out, int *block, int *blocks)
{
  const int W = a.W;
  const int tiles = d->Np / 64;
  int rc;
  *block = 0; *blocks = 0;
  bool can = !(getenv("RAMX_NO_PERSISTENT") != NULL || d->force_chain || a.go > 0 || a.ge > 0 || a.go + a.ge < -32768 ||
               !prk_has_width(W) || L <= 0);
  if (multi && (!d->peer_ready || d->nranks < 2 || L >= 65536 || getenv("RAMX_NO_PEER") != NULL)) can = false;
  if (can)
  {
    rc = (W == 14) ? prk_plan<14>(tiles, block, blocks) : (W == 20) ? prk_plan<20>(tiles, block, blocks) :
         (W == 40) ? prk_plan<40>(tiles, block, blocks) : prk_plan<80>(tiles, block, blocks);
    if (rc != RAMX_OK) return rc;
    if (*block == 0) can = false;          // not co-resident: keep the streaming kernel
  }
  *can_out = can;
  return RAMX_OK;
}

// Will the packed-row kernel (ramx_kernels_packed.h) take this direction (from row d->pk_r0 on)?  Single GPU, a scoring system whose
// rows fit int16 relative to a per-flank base, the flank set resident, and -- where flanks have low out-of-bounds cells in the
// first rows -- the int32 kernel for those rows.
static int pk_route(ramx_dev *d, const KArgs &a, int L, bool multi, bool int32_can, bool *pk_out, int *spread, int *rebase, int *pk_block, int *pk_blocks)
{
  *pk_out = false; *pk_block = 0; *pk_blocks = 0;
  const int pk_r0 = (getenv("RAMX_NO_PERSISTENT") != NULL || d->force_chain || L <= 0) ? -1 : d->pk_r0;
  // (multi-rank: the packed kernel speaks the same mailbox protocol as the int32 one, ramx_kernels_vote.h; which of the two a rank
  // runs, from which row on and in how many pieces is that rank's own business)
  bool pk = pk_r0 >= 0 && pk_r0 < L && ramx_pk_plan(a.W, a.go, a.ge, a.tab, spread, rebase) != 0;
  if (multi && (!d->peer_ready || d->nranks < 2 || L >= 65536 || getenv("RAMX_NO_PEER") != NULL || getenv("RAMX_NO_PK_MULTI") != NULL)) pk = false;
  if (pk)
  {
    if (ramx_pk_shape(a.W, d->Np / 64, pk_block, pk_blocks) != RAMX_OK) { ramx_set_error("packed-row kernel: occupancy query failed"); return RAMX_ERR_HIP; }
    if (*pk_block == 0 || (pk_r0 > 0 && !int32_can)) pk = false;
  }
  *pk_out = pk;
  return RAMX_OK;
}

static int prk_run(ramx_dev *d, const KArgs &a, int L, bool *used)
{
  *used = false;
  const int W = a.W;
  const bool multi = dev_is_multi(d);
  int block = 0, blocks = 0, rc;
  bool can = false;
  if ((rc = prk_local_can(d, a, L, multi, &can, &block, &blocks)) != RAMX_OK) return rc;
  if (multi)
  {
    // every rank must take the same path: agree (max over ranks of "cannot")
    int cannot = can ? 0 : 1;
    if ((rc = host_allreduce_flag(d, &cannot)) != RAMX_OK) return rc;
    can = !cannot;
  }
  const bool int32_can = can;
  const int pk_r0 = (getenv("RAMX_NO_PERSISTENT") != NULL || d->force_chain || L <= 0) ? -1 : d->pk_r0;
  if (!can && (multi || pk_r0 < 0)) return RAMX_OK;
  *used = true;        // agreed (multi-rank: by all ranks): from here on the caller handles a local failure collectively
  PArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.S = d->d_state[0]; pa.bases = d->d_bases; pa.bounds = d->d_bounds; pa.trim = d->d_trim;
  pa.sums0 = d->d_sums; pa.vote = d->d_vote; pa.ctl_out = d->d_ctl; pa.cons_out = d->d_cons; pa.err = d->d_err;
  pa.Np = d->Np; pa.Nx = d->Nx; pa.r0 = 0; pa.L = L; pa.go = a.go; pa.ge = a.ge; pa.cap = a.cap; pa.minimp = a.minimp;
  pa.when_to_stop = a.when_to_stop; pa.nblocks = blocks;
  pa.nranks = 1; pa.rank = 0; pa.peers = NULL; pa.box = NULL;
  if (multi)
  {
    pa.nranks = d->nranks; pa.rank = d->rank; pa.peers = (PeerBox *const *)d->d_peer; pa.box = d->xbox;
    pa.mirror = NULL;
    // Rank-local steps first, WITHOUT leaving on an error: the collective below is executed by every rank in any case
    // (a rank that skipped it would pair its next collective with this one on the other ranks).
    auto local_prep = [&]() -> int
    {
      if (d->hostbox_host)
      {
        if (!d->hostbox_mirror) HIPCHK(hipMalloc((void **)&d->hostbox_mirror, sizeof(PeerBox)));
        HIPCHK(hipMemsetAsync(d->hostbox_mirror, 0, sizeof(PeerBox), d->stream));
        pa.mirror = d->hostbox_mirror;
      }
      // my box is cleared BEFORE the collective below, which no remote launch can get past without my taking part:
      // nobody writes a word of this run into it too early, and nothing of the last run survives
      if (d->hostbox_host)
      {
        HIPCHK(hipStreamSynchronize(d->stream));
        memset((void *)d->hostbox_host, 0, sizeof(PeerBox));
        __sync_synchronize();
      }
      else
      {
        // complete, not just enqueued: the agreement below may be a host-only collective, and a peer that gets past it
        // launches and writes into this box at once
        HIPCHK(hipMemsetAsync(d->xbox, 0, sizeof(PeerBox), d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
      }
      return RAMX_OK;
    };
    const int lrc = local_prep();
    if ((rc = host_allreduce_shards(d, d->d_sums)) != RAMX_OK) return rc;     // vote of row 0 (from K(-1)) over ranks
    if (lrc != RAMX_OK) return lrc;
    // test hook: RAMX_TEST_FAIL_PRK_RANK=<rank> makes that rank fail where a launch error would
    const char *tf = getenv("RAMX_TEST_FAIL_PRK_RANK");
    if (tf && atoi(tf) == d->rank) { ramx_set_error("test hook: forced failure of the persistent launch on rank %d", d->rank); return RAMX_ERR_HIP; }
  }
  memcpy(pa.tab, a.tab, sizeof(pa.tab));
  pa.pack_ok = getenv("RAMX_NO_FASTPACK") ? 0 : fast_pack_ok(pa.tab, a.go, a.ge, L, W);
  if (pa.pack_ok && getenv("RAMX_NO_MASKHI") == NULL) pa.pack_ok = 2;      // 2: the far-end-masked fast band may be used too
  pa.lean_p = lean_p_of(pa.tab, a.go, a.ge);
  // a leader costs its wave ~150 instructions in prk_leader_rows, the full band 640 more than LEAN; RAMX_LEADER_MAX=0 switches
  // the leader path off (A/B and tests), larger values exercise it on waves with many leaders
  { const char *lm = getenv("RAMX_LEADER_MAX"); pa.leader_max = lm ? atoi(lm) : 3; if (pa.leader_max < 0) pa.leader_max = 0; if (pa.leader_max > 64) pa.leader_max = 64; }
  // ---- the packed-row kernel (ramx_kernels_packed.h) takes the direction from row r0 = d->pk_r0 on when the scoring system's
  // rows fit int16 relative to a per-flank base; the rows before (flanks whose cores are shorter than the band have
  // out-of-bounds cells at the LOW end of the band there) stay on this kernel, which hands over the sums of row r0 --------
  PKArgs ka;
  memset(&ka, 0, sizeof(ka));
  int pk_block = 0, pk_blocks = 0;
  bool pk = false;
  if ((rc = pk_route(d, a, L, multi, int32_can, &pk, &ka.spread, &ka.rebase, &pk_block, &pk_blocks)) != RAMX_OK) return rc;
  if (!pk && !int32_can) { *used = false; return RAMX_OK; }
  if (!pk && (rc = pack_rest(d)) != RAMX_OK) return rc;        // the int32 kernel reads the whole window
  HIPCHK(hipMemsetAsync(d->d_vote, 0, PRK_NSETS * NSHARD * sizeof(PShard), d->stream));
  HIPCHK(hipMemsetAsync(d->d_err, 0, 64, d->stream));
  long long *sums_next = d->d_sums + (size_t)NSHARD * 4;        // slot 1: zeroed by K(-1)
  const bool head = !pk || pk_r0 > 0;                            // this kernel runs (all of the direction, or its first rows)
  if (pk && pk_r0 > 0) { pa.L = pk_r0; pa.sums_next = sums_next; }
#ifdef RAMX_PRK_TIMING
  const size_t nw = pk ? (size_t)pk_blocks * (pk_block / 64) : (size_t)blocks * (block / 64);
  unsigned long long *dbgbuf = NULL;
  HIPCHK(hipMalloc((void **)&dbgbuf, nw * 8 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(dbgbuf, 0, nw * 8 * sizeof(unsigned long long)));
  if (pk) ka.dbg = dbgbuf; else pa.dbg = dbgbuf;                 // timing build: with a packed launch, its phases are the ones printed
#endif
  rc = RAMX_OK;
  const bool tmarks = getenv("RAMX_TIMING") != NULL;
  const double tm0 = now_ms();
  rt_mark("persistent route, memsets");
  if (head)
  {
    if (block == 256)
      rc = (W == 14) ? prk_launch<14, 256>(d, pa, blocks) : (W == 20) ? prk_launch<20, 256>(d, pa, blocks) :
           (W == 40) ? prk_launch<40, 256>(d, pa, blocks) : prk_launch<80, 256>(d, pa, blocks);
    else
      rc = (W == 14) ? prk_launch<14, 512>(d, pa, blocks) : (W == 20) ? prk_launch<20, 512>(d, pa, blocks) : prk_launch<40, 512>(d, pa, blocks);
  }
  rt_mark("first rows launched (int32 kernel)");
  if (rc == RAMX_OK && pk)
  {
    ka.S = d->d_state[0]; ka.bases = d->d_bases; ka.bounds = d->d_bounds; ka.trim = d->d_trim;
    ka.vote = d->d_vote; ka.cons_out = d->d_cons; ka.err = d->d_err;
    ka.peers = pa.peers; ka.box = pa.box; ka.mirror = pa.mirror; ka.rank = pa.rank; ka.nranks = pa.nranks;
    ka.xblock = 0;
    if (ka.nranks > 1 && getenv("RAMX_NO_PK_EXCHANGER") == NULL)
    {
      // a CU to spare: the device's exchange duties get a workgroup of their own (ramx_kernels_packed.h)
      int cap = 0;
      if (ramx_pk_capacity(W, pk_block, &cap) == RAMX_OK && pk_blocks + 1 <= cap) ka.xblock = pk_blocks;
    }
    ka.Np = d->Np; ka.Nx = d->Nx; ka.L = L; ka.go = a.go; ka.ge = a.ge; ka.cap = a.cap; ka.minimp = a.minimp;
    ka.when_to_stop = a.when_to_stop; ka.nblocks = pk_blocks;
    memcpy(ka.tab, a.tab, sizeof(ka.tab));
    ka.lean_p = pa.lean_p; ka.leader_max = pa.leader_max;
    ka.spec_on = getenv("RAMX_NO_PK_SPEC") == NULL ? 1 : 0;
    { const char *we = getenv("RAMX_TEST_PK_WRONG_EVERY"); ka.test_wrong_every = we ? atoi(we) : 0; }
    // test hook: a span smaller than the scoring system's makes the kernel's own checks (rows at entry, every 64th row) refuse
    { const char *ts = getenv("RAMX_TEST_PK_SPREAD"); if (ts) ka.spread = atoi(ts); }
    ka.spread_rows = ka.spread;
    { const char *ts = getenv("RAMX_TEST_PK_SPREAD_ROWS"); if (ts) ka.spread_rows = atoi(ts); }
    d->last_packed_r0 = pk_r0;
    // The direction runs in pieces of RAMX_PK_SEGMENT columns (one launch each: rows written back, sums of the next row handed
    // over as plain words, control block passed on), so that only the base words a run really reads are ever packed: the
    // next piece's words are packed on a second stream while this piece runs.
    const int seg = pk_segment_columns(L, a.when_to_stop);
    const int NWw = (2 * W + 1 + 8) / 8 + 2;
    long long *sbuf[2] = { d->d_sums + (size_t)NSHARD * 4, d->d_sums + (size_t)2 * NSHARD * 4 };     // slots 1 and 2
    const long long *sums_in = pk_r0 > 0 ? sbuf[0] : d->d_sums;
    int nextbuf = pk_r0 > 0 ? 1 : 0;
    int ctl_i = 0;                           // the control block of the launch before sits in d_ctl[ctl_i] (pk_r0 == 0: not read)
    auto words_for = [&](int r_end) { int wn = ((r_end + 7) >> 3) + NWw + 2; return wn < d->KW ? wn : d->KW; };
    if (ka.nranks > 1 && !head) test_delay_rank(d);
    for (int r = pk_r0, s_i = 0; r < L && rc == RAMX_OK; s_i++)
    {
      const int r1 = seg > 0 ? std::min(L, (r / seg + 1) * seg) : L;
      // this piece's words: normally packed already (begin_direction, or beside the piece before)
      if (d->pack_busy) { HIPCHK(hipStreamWaitEvent(d->stream, d->pack_done, 0)); d->pack_busy = 0; }
      if (d->packed_kw < words_for(r1))
      {
        if ((rc = launch_pack(d, d->Nx, d->Np, W, d->packed_kw, words_for(r1), d->stream)) != RAMX_OK) break;
        d->packed_kw = words_for(r1);
      }
      HIPCHK(hipMemsetAsync(d->d_vote, 0, PRK_NSETS * NSHARD * sizeof(PShard), d->stream));
      ka.r0 = r; ka.Lseg = r1; ka.sums_in = sums_in;
      ka.sums_next = r1 < L ? sbuf[nextbuf] : NULL;
      if (r1 < L) HIPCHK(hipMemsetAsync(sbuf[nextbuf], 0, (size_t)NSHARD * 4 * sizeof(long long), d->stream));
      ka.ctl_in = d->d_ctl + ctl_i; ka.ctl_out = d->d_ctl + (ctl_i ^ 1);
      if (r == 0) ka.ctl_out = d->d_ctl;     // (first launch of the direction: d_ctl[1] still holds K(-1)'s block, nothing is read)
      const double tm1 = now_ms();
      rc = ramx_pk_launch(d->stream, W, pk_block, pk_blocks, ka);
      if (rc != RAMX_OK) { ramx_set_error("packed-row kernel: launch failed (W %d, %d workgroups of %d threads)", W, pk_blocks, pk_block); break; }
      if (tmarks) fprintf(stderr, "RAMX_TIMING       run: piece %d (rows %d..%d, %d x %d threads) launched %.3f ms after the first launch (launch call %.3f ms)\n", s_i, r, r1, pk_blocks, pk_block, now_ms() - tm0, now_ms() - tm1);
      const int out_i = (r == 0) ? 0 : (ctl_i ^ 1);
      if (r1 >= L) break;
      // the next piece's words, beside this launch
      const int r2 = std::min(L, (r1 / seg + 1) * seg);
      if (d->packed_kw < words_for(r2))
      {
        if ((rc = launch_pack(d, d->Nx, d->Np, W, d->packed_kw, words_for(r2), d->pack_stream)) != RAMX_OK) break;
        HIPCHK(hipEventRecord(d->pack_done, d->pack_stream));
        d->pack_busy = 1;
        d->packed_kw = words_for(r2);
      }
      // did this piece end the direction?  (one synchronisation per piece: 2,048 columns)
      RamxCtl hc;
      unsigned errw = 0;
      HIPCHK(hipStreamSynchronize(d->stream));
      if (tmarks) fprintf(stderr, "RAMX_TIMING       run: piece %d complete %.3f ms after the first launch\n", s_i, now_ms() - tm0);
      {
        unsigned long long ew = 0;
        if ((rc = d2h_small(d, &hc, d->d_ctl + out_i, sizeof(hc))) != RAMX_OK) break;
        if ((rc = d2h_small(d, &ew, d->d_err, sizeof(ew))) != RAMX_OK) break;
        errw = (unsigned)ew;
      }
      rt_mark("piece: control block and error word read");
      if (hc.stopped || hc.pad != 0 || errw != 0 || hc.rows_done < r1) break;
      sums_in = sbuf[nextbuf]; nextbuf ^= 1; ctl_i = out_i; r = r1;
    }
    // (the caller reads both control blocks and takes the one with more rows)
    if (pk) { block = pk_block; blocks = pk_blocks; }
  }
  else d->last_packed_r0 = -1;
#ifdef RAMX_PRK_TIMING
  if (rc == RAMX_OK)
  {
    // debug build: phase breakdown per column, in ns (wall_clock64 ticks are 10 ns)
    HIPCHK(hipStreamSynchronize(d->stream));
    unsigned long long *h = (unsigned long long *)malloc(nw * 8 * sizeof(unsigned long long));
    HIPCHK(hipMemcpy(h, dbgbuf, nw * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    static const char *nm[6] = { "wait vote", "block barrier 1", "band", pk ? "early arrival" : "reduce+barrier 2", pk ? "late arrival" : "issue atomics", "loop top" };
    const int wpb = block / 64;
    fprintf(stderr, "PRK_TIMING %s blocks %d x %d threads, L %d, first row %d (ns per column)\n", pk ? "packed rows" : "int32 rows", blocks, block, L, pk ? pk_r0 : 0);
    for (int k = 0; k < 6; k++)
    {
      double s0 = 0, sO = 0, mx = 0, mn = 1e30;
      for (size_t i = 0; i < nw; i++)
      {
        const double v = 10.0 * (double)h[i * 8 + k] / L;
        if ((int)(i % wpb) == 0) s0 += v; else sO += v;
        if (v > mx) mx = v;
        if (v < mn) mn = v;
      }
      fprintf(stderr, "PRK_TIMING %-18s wave0 avg %8.1f  other waves avg %8.1f  min %8.1f  max %8.1f\n", nm[k], s0 / blocks,
              wpb > 1 ? sO / (double)(nw - blocks) : 0.0, mn, mx);
    }
    fprintf(stderr, "PRK_TIMING band by wave index:");
    for (int wv = 0; wv < wpb; wv++)
    {
      double sw = 0;
      for (int b = 0; b < blocks - 1; b++) sw += 10.0 * (double)h[((size_t)b * wpb + wv) * 8 + 2] / L;
      fprintf(stderr, " w%d %.0f", wv, sw / (blocks > 1 ? blocks - 1 : 1));
    }
    fprintf(stderr, "\n");
    {
      // the workgroup everybody waits for: the one whose vote wave waits least for the vote
      int gb = 0;
      for (int b = 1; b < blocks; b++) if (h[(size_t)b * wpb * 8 + 0] < h[(size_t)gb * wpb * 8 + 0]) gb = b;
      fprintf(stderr, "PRK_TIMING workgroup %d (shortest vote wait), per wave: wait vote | barrier | band | early | late | top\n", gb);
      for (int wv = 0; wv < wpb; wv++)
      {
        const unsigned long long *q = h + ((size_t)gb * wpb + wv) * 8;
        fprintf(stderr, "PRK_TIMING   w%d %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f   rows full %llu lean %llu\n", wv, 10.0 * q[0] / L, 10.0 * q[1] / L, 10.0 * q[2] / L,
                10.0 * q[3] / L, 10.0 * q[4] / L, 10.0 * q[5] / L, q[6], q[7]);
      }
    }
    {
      // second half of the run: waves that were kept from the LEAN band, and by how many lanes
      const double half = L - L / 2;
      size_t never = 0, always = 0, wg_any = 0;
      double lanes = 0, cols = 0;
      int hist[6] = { 0, 0, 0, 0, 0, 0 };       // waves by average number of such lanes in their non-lean columns: <=1, <=2, <=4, <=8, <=16, more
      for (int b = 0; b < blocks; b++)
      {
        bool any = false;
        for (int wv = 0; wv < wpb; wv++)
        {
          const size_t i = (size_t)b * wpb + wv;
          const double c = (double)h[i * 8 + 6], l = (double)h[i * 8 + 7];
          if (c == 0) { never++; continue; }
          any = any || c > 0.5 * half;
          if (c > 0.9 * half) always++;
          lanes += l; cols += c;
          const double avg = l / c;
          hist[avg <= 1 ? 0 : avg <= 2 ? 1 : avg <= 4 ? 2 : avg <= 8 ? 3 : avg <= 16 ? 4 : 5]++;
        }
        if (any) wg_any++;
      }
      fprintf(stderr, "PRK_LEANSTAT second half (%.0f columns): %zu of %zu waves always LEAN, %zu non-LEAN in > 90 %% of the columns; workgroups with a wave "
              "non-LEAN in > 50 %%: %zu of %d; lanes per non-LEAN wave-column %.2f; waves by that average <=1: %d <=2: %d <=4: %d <=8: %d <=16: %d more: %d\n",
              half, never, nw, always, wg_any, blocks, cols > 0 ? lanes / cols : 0.0, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
    }
    free(h);
    (void)hipFree(dbgbuf);
  }
#endif
  return rc;
}

// test hooks of the vote-wave mode of the cell-parallel kernel (ramx_kernels_cp.h): RAMX_TEST_CP_WRONG_EVERY=n computes every
// n-th row on a deliberately wrong guess, RAMX_TEST_CP_VOTE_DELAY=units holds the vote wave back so that the band waves have
// finished their row on the guess when the decision arrives
static void cp_test_hooks(CPArgs &ca)
{
  const char *we = getenv("RAMX_TEST_CP_WRONG_EVERY"), *vd = getenv("RAMX_TEST_CP_VOTE_DELAY");
  ca.test_wrong_every = we ? atoi(we) : 0;
  ca.test_vote_delay = vd ? atoi(vd) : 0;
  ca.lean_p = lean_p_of(ca.tab, ca.go, ca.ge);
}

// ---- batch mode -------------------------------------------------------------------------------
template <int W, int BLOCK>
static int fam_launch(ramx_dev *d, const FArgs &fa, int F, hipStream_t st)
{
  (void)d;
  hipLaunchKernelGGL((ramx_family_kernel<W, BLOCK>), dim3(F), dim3(BLOCK), 0, st, fa);
  HIPCHK(hipGetLastError());
  return RAMX_OK;
}

// One pass over a batch.  allow_dev: families above one cell-parallel workgroup may take the device-wide mode (several
// workgroups voting through ticket words with a bounded spin).  timed_out[f] != 0 on return: that family's device-wide vote
// gave up (e.g. its workgroups were not all resident because a co-tenant held CUs) -- its outputs are invalid and the caller
// repeats it with allow_dev = false; with timed_out == NULL such a family makes the whole call fail.
static int run_families_pass(ramx_dev *d, const ramx_flank *flanks, int32_t n_padded, const int32_t *fam_first,
                             const int32_t *fam_count, int32_t n_families, const ramx_params *p,
                             ramx_run_info *infos, int8_t *cons, int32_t *trim_high, int32_t *trim_pos,
                             bool allow_dev, unsigned char *timed_out)
{
  if (!d || !p || !p->matrix || n_families < 0 || n_padded < 0 || (n_padded & 63) || (n_families && (!fam_first || !fam_count || !flanks)))
  { ramx_set_error("ramx_dev_run_families: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  const int W = p->bandwidth, L = p->L;
  if (W < 1 || L < 0) { ramx_set_error("ramx_dev_run_families: bad bandwidth / L"); return RAMX_ERR_ARG; }
  // register-resident family kernel where it applies, the streaming family kernel for everything else
  const bool resident = fam_has_width(W) && p->gapopen <= 0 && p->gapextn <= 0 && p->gapopen + p->gapextn >= -32768 &&
                        !d->force_chain && getenv("RAMX_NO_PERSISTENT") == NULL;
  int maxn = 0;
  for (int f = 0; f < n_families; f++)
  {
    if (fam_count[f] < 0 || fam_first[f] < 0 || (fam_first[f] & 63) || (long long)fam_first[f] + fam_count[f] > n_padded) { ramx_set_error("ramx_dev_run_families: bad family layout"); return RAMX_ERR_ARG; }
    if (fam_count[f] > maxn) maxn = fam_count[f];
  }
  if (maxn > 512) { ramx_set_error("batch mode: a family has more than 512 flanks"); return RAMX_ERR_UNSUPPORTED; }
  if (n_families == 0) return RAMX_OK;
  const int Np = n_padded > 0 ? n_padded : 64;
  const int KW = (L + 2 * W + 2) / 8 + 12;
  int rc;
  if ((rc = ensure(&d->d_flanks, &d->cap_flanks, (size_t)Np * sizeof(ramx_flank)))) return rc;
  if ((rc = ensure(&d->d_bases, &d->cap_bases, (size_t)KW * Np * sizeof(unsigned)))) return rc;
  if ((rc = ensure(&d->d_bounds, &d->cap_bounds, (size_t)Np * sizeof(int2)))) return rc;   // kept between calls: hipFree/hipMalloc
  if ((rc = ensure(&d->d_trim, &d->cap_trim, (size_t)Np * sizeof(int2)))) return rc;       // per call cost ~0.5 ms of device sync
  if ((rc = ensure(&d->d_cons, &d->cap_cons, (size_t)n_families * (L > 0 ? L : 1) + 16))) return rc;
  // Route of every family.  Groups 0 .. RAMX_CP_NCLASS-1: the cell-parallel kernel (K lanes per flank, ramx_cp.hip) for
  // families it can take -- supported band width and scoring system, small enough for one workgroup at K lanes per
  // flank, every flank either empty or starting at or before the first base behind the core edge (t_lo <= 0; the
  // masked band's carry argument needs the out-of-bounds prefix to end at or before the band centre).  Groups
  // RAMX_CP_NCLASS ..: one lane per flank, by workgroup shape (64, 128, 256, 512 threads = 1, 2, 4, 8 tiles).
  int tab9[RAMX_NCLASS][4];
  for (int c = 0; c < RAMX_NCLASS; c++)
  {
    const int code = (c == 8) ? RAMX_SYM_N : c;
    for (int k = 0; k < 4; k++) tab9[c][k] = p->matrix[k * 100 + code];
  }
  const int cp_max = d->force_chain ? 0 : ramx_cp_max_family(W, p->gapopen, p->gapextn, tab9, L);
  FamDesc *hfd = (FamDesc *)malloc(sizeof(FamDesc) * n_families);
  int *grp = (int *)malloc(sizeof(int) * n_families);
  int cls_first[RAMX_NGROUP + 1], cls_count[RAMX_NGROUP], cp_k[RAMX_CP_NCLASS], cp_threads[RAMX_CP_NCLASS];
  memset(cls_count, 0, sizeof(cls_count));
  memset(cp_k, 0, sizeof(cp_k)); memset(cp_threads, 0, sizeof(cp_threads));
  auto cls_of = [](int nx) { return nx <= 64 ? 0 : nx <= 128 ? 1 : nx <= 256 ? 2 : 3; };
  // Families above one cell-parallel workgroup (group -1): several workgroups per family, the family's vote through its
  // own ticket words -- the device-wide mode of the same kernel, all such families in ONE launch of at most one
  // workgroup per CU (every workgroup must be resident).
  int n_cp = 0, n_dev = 0, dev_k = 0, dev_threads = 0, dev_vw = 0, cus = 0;
  std::vector<CpDevDesc> hdev;
  std::vector<int> dev_round_first, dev_round_blocks;   // rounds of at most `cus` workgroups, launched one after the other
  {
    int devo = 0;
    if (hipGetDevice(&devo) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devo) != hipSuccess) cus = 0;
  }
  const bool dev_route = allow_dev && cp_max > 0 && cus > 0 && getenv("RAMX_NO_CP_DEVICE") == NULL && getenv("RAMX_NO_PERSISTENT") == NULL;
  // small workgroups (four band waves, lowest latency) when all the multi-workgroup families then fit the CUs together
  int dev_wide = 0;
  if (dev_route)
  {
    long long need = 0;
    for (int f = 0; f < n_families; f++)
    {
      if (fam_count[f] <= cp_max) continue;
      int k = 0, th = 0, nb = 0;
      int vwf = 0;
      ramx_cp_device_plan(W, fam_count[f], cus, 0, &k, &th, &nb, &vwf);
      need += k > 0 ? nb : 0;
    }
    dev_wide = need > cus;
  }
  for (int f = 0; f < n_families; f++)
  {
    int g = RAMX_CP_NCLASS + cls_of(fam_count[f]);
    if (fam_count[f] > 0 && cp_max > 0)
    {
      bool ok = true;
      for (int i = 0; i < fam_count[f] && ok; i++)
      {
        const ramx_flank &x = flanks[fam_first[f] + i];
        ok = (x.t_lo <= 0) || (x.t_lo > x.t_hi);
      }
      int k = 0, th = 0;
      const int c = (ok && fam_count[f] <= cp_max) ? ramx_cp_class(W, fam_count[f], &k, &th) : -1;
      if (c >= 0) { g = c; cp_k[c] = k; cp_threads[c] = th; n_cp++; }
      else if (ok && dev_route)
      {
        int nb = 0;
        int vwf = 0;
        ramx_cp_device_plan(W, fam_count[f], cus, dev_wide, &k, &th, &nb, &vwf);
        if (k > 0 && nb <= cus && (dev_k == 0 || (k == dev_k && th == dev_threads && vwf == dev_vw)))
        {
          dev_k = k; dev_threads = th; dev_vw = vwf;
          if (dev_round_first.empty() || dev_round_blocks.back() + nb > cus)
          {
            dev_round_first.push_back((int)hdev.size());
            dev_round_blocks.push_back(0);
          }
          dev_round_blocks.back() += nb;
          for (int b = 0; b < nb; b++)
          {
            CpDevDesc x;
            memset(&x, 0, sizeof(x));
            x.first = fam_first[f]; x.nx = fam_count[f]; x.b = b; x.nb = nb; x.id = f; x.vs = n_dev;
            hdev.push_back(x);
          }
          n_dev++;
          g = -1;
        }
      }
    }
    grp[f] = g;
    if (g >= 0) cls_count[g]++;
  }
  cls_first[0] = 0;
  for (int c = 0; c < RAMX_NGROUP; c++) cls_first[c + 1] = cls_first[c] + cls_count[c];
  {
    int fill[RAMX_NGROUP];
    for (int c = 0; c < RAMX_NGROUP; c++) fill[c] = cls_first[c];
    for (int f = 0; f < n_families; f++)
    {
      if (grp[f] < 0) continue;
      FamDesc &x = hfd[fill[grp[f]]++];
      x.tile0 = fam_first[f] / 64; x.ntiles = (fam_count[f] + 63) / 64; x.nx = fam_count[f]; x.id = f;
    }
  }
  // from here on every failure leaves through `done` (host buffers are released there)
  RamxCtl *hctl = NULL;
  int2 *tmp = NULL;
  float ms = 0;
  unsigned long long *cp_dbg = NULL;
#define FAMCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    ramx_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #call); rc = RAMX_ERR_HIP; goto done; } } while (0)
  FamDesc *dfd; RamxCtl *dctl;
  FArgs fa;
  const bool legacy_any = n_cp + n_dev < n_families;
  if ((rc = ensure(&d->d_fam, &d->cap_fam, sizeof(FamDesc) * (size_t)n_families))) goto done;
  if ((rc = ensure(&d->d_famctl, &d->cap_famctl, sizeof(RamxCtl) * (size_t)n_families))) goto done;
  dfd = (FamDesc *)d->d_fam; dctl = d->d_famctl;
  FAMCHK(hipMemcpyAsync(dfd, hfd, sizeof(FamDesc) * n_families, hipMemcpyHostToDevice, d->stream));
  if (n_padded) FAMCHK(hipMemcpyAsync(d->d_flanks, flanks, (size_t)n_padded * sizeof(ramx_flank), hipMemcpyHostToDevice, d->stream));
  if ((rc = launch_pack(d, n_padded, Np, W, 0, KW, d->stream)) != RAMX_OK) goto done;
  d->packed_kw = KW;
  FAMCHK(hipMemsetAsync(d->d_cons, 0, (size_t)n_families * (L > 0 ? L : 1) + 16, d->stream));   // columns a family never ran read as 0
  memset(&fa, 0, sizeof(fa));
  fa.bases = d->d_bases; fa.bounds = d->d_bounds; fa.fam = dfd; fa.trim = d->d_trim; fa.ctl_out = dctl; fa.cons_out = d->d_cons;
  fa.Np = Np; fa.L = L; fa.go = p->gapopen; fa.ge = p->gapextn; fa.cap = p->cappenalty; fa.minimp = p->minimprovement;
  fa.when_to_stop = p->when_to_stop;
  memcpy(fa.tab, tab9, sizeof(fa.tab));
  fa.pack_ok = getenv("RAMX_NO_FASTPACK") ? 0 : fast_pack_ok(fa.tab, fa.go, fa.ge, L, W);
  if (fa.pack_ok && getenv("RAMX_NO_MASKHI") == NULL) fa.pack_ok = 2;
  fa.lean_p = lean_p_of(fa.tab, fa.go, fa.ge);
#ifdef RAMX_PRK_TIMING
  FAMCHK(hipMalloc((void **)&fa.dbg, 8 * 8 * sizeof(unsigned long long)));
  FAMCHK(hipMemset(fa.dbg, 0, 8 * 8 * sizeof(unsigned long long)));
#endif
  FAMCHK(hipEventRecord(d->ev_begin, d->stream));
  // the groups run side by side: one stream per group, forked from / joined into the library's stream
  if (!d->cls_init)
  {
    for (int c = 0; c < RAMX_NGROUP; c++)
    {
      FAMCHK(hipStreamCreateWithFlags(&d->cls_stream[c], hipStreamNonBlocking));
      FAMCHK(hipEventCreateWithFlags(&d->cls_done[c], hipEventDisableTiming));
    }
    FAMCHK(hipEventCreateWithFlags(&d->cls_ready, hipEventDisableTiming));
    d->cls_init = 1;
  }
  if (n_dev > 0)
  {
    // everything the device-wide launch needs is uploaded and cleared on the library's stream, BEFORE the fork: the launch
    // itself is then the first thing the idle device sees, and its workgroups are resident before any other group's
    // (vote / error words: one set per multi-workgroup family, not per family of the batch)
    if ((rc = ensure(&d->d_devdesc, &d->cap_devdesc, sizeof(CpDevDesc) * hdev.size()))) goto done;
    if ((rc = ensure(&d->d_vote_sets, &d->cap_vote_sets, sizeof(PShard) * RAMX_CP_NSETS * NSHARD * (size_t)n_dev))) goto done;
    if ((rc = ensure(&d->d_err_sets, &d->cap_err_sets, 64 * (size_t)n_dev))) goto done;
    FAMCHK(hipMemcpyAsync(d->d_devdesc, hdev.data(), sizeof(CpDevDesc) * hdev.size(), hipMemcpyHostToDevice, d->stream));
    FAMCHK(hipMemsetAsync(d->d_vote_sets, 0, sizeof(PShard) * RAMX_CP_NSETS * NSHARD * (size_t)n_dev, d->stream));
    FAMCHK(hipMemsetAsync(d->d_err_sets, 0, 64 * (size_t)n_dev, d->stream));
  }
  FAMCHK(hipEventRecord(d->cls_ready, d->stream));
  // every group stream waits (the register-resident branch below merges shapes AFTER this point: a stream that is
  // launched on must never have skipped the wait -- uploads and the pack kernel run on the library's stream)
  for (int c = 0; c < RAMX_NGROUP; c++) FAMCHK(hipStreamWaitEvent(d->cls_stream[c], d->cls_ready, 0));
  // ---- families above one workgroup: device-wide mode, one launch, enqueued first ----------------
  if (n_dev > 0)
  {
    CPArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.bases = d->d_bases; ca.bounds = d->d_bounds; ca.trim = d->d_trim; ca.ctl_out = dctl; ca.cons_out = d->d_cons;
    ca.Np = Np; ca.KW = KW; ca.L = L; ca.go = p->gapopen; ca.ge = p->gapextn; ca.cap = p->cappenalty; ca.minimp = p->minimprovement;
    ca.when_to_stop = p->when_to_stop;
    memcpy(ca.tab, tab9, sizeof(ca.tab));
    ca.nranks = 1; ca.rank = 0;
    {
      hipStream_t st = d->cls_stream[RAMX_NGROUP - 1];       // shares a stream with the last lane-per-flank shape, enqueued before it
      ca.dev = d->d_devdesc; ca.vote = d->d_vote_sets; ca.err = d->d_err_sets; ca.vote_wave = dev_vw;
      // test hooks: RAMX_TEST_CP_DROP_TICKET=row withholds one workgroup's words for that row, in every multi-workgroup
      // family or only in family RAMX_TEST_CP_DROP_FAMILY (index in the batch)
      { const char *td = getenv("RAMX_TEST_CP_DROP_TICKET"), *tf = getenv("RAMX_TEST_CP_DROP_FAMILY");
        ca.test_drop_row = td ? atoi(td) : 0; ca.test_drop_id = tf ? atoi(tf) : -1; }
      cp_test_hooks(ca);
#ifdef RAMX_CP_TIMING
      if (n_cp == 0)
      {
        FAMCHK(hipMalloc((void **)&ca.dbg, (16 * 16 + 8) * sizeof(unsigned long long)));
        FAMCHK(hipMemset(ca.dbg, 0, (16 * 16 + 8) * sizeof(unsigned long long)));
        cp_dbg = ca.dbg;
      }
#endif
      for (size_t ro = 0; ro < dev_round_first.size(); ro++)
      {
        ca.dev = d->d_devdesc + dev_round_first[ro];
        rc = ramx_cp_launch_device(st, W, dev_k, dev_threads, dev_round_blocks[ro], ca);
        if (rc != RAMX_OK) { ramx_set_error("cell-parallel multi-family device launch failed (W %d, %d workgroups)", W, dev_round_blocks[ro]); goto done; }
      }
    }
  }
  // ---- cell-parallel groups ------------------------------------------------------------------
  if (n_cp > 0)
  {
    CPArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.bases = d->d_bases; ca.bounds = d->d_bounds; ca.trim = d->d_trim; ca.ctl_out = dctl; ca.cons_out = d->d_cons;
    ca.state_out = NULL;
    ca.Np = Np; ca.KW = KW; ca.L = L; ca.go = p->gapopen; ca.ge = p->gapextn; ca.cap = p->cappenalty; ca.minimp = p->minimprovement;
    ca.when_to_stop = p->when_to_stop;
    memcpy(ca.tab, tab9, sizeof(ca.tab));
    ca.lean_p = lean_p_of(ca.tab, ca.go, ca.ge);
    if (getenv("RAMX_CP_PEEK") != NULL)
    {
      // test hook: keep the final rows of every flank, [flank][2W+1] (m, e), for ramx_dev_peek_family_state
      const size_t need = (size_t)Np * (2 * W + 1) * sizeof(int2);
      if ((rc = ensure(&d->d_cpstate, &d->cap_cpstate, need))) goto done;
      FAMCHK(hipMemsetAsync(d->d_cpstate, 0, need, d->stream));
      FAMCHK(hipEventRecord(d->cls_ready, d->stream));
      for (int c = 0; c < RAMX_CP_NCLASS; c++) FAMCHK(hipStreamWaitEvent(d->cls_stream[c], d->cls_ready, 0));
      ca.state_out = (int2 *)d->d_cpstate;
      d->cpstate_W = W; d->cpstate_n = Np;
    }
#ifdef RAMX_CP_TIMING
    FAMCHK(hipMalloc((void **)&ca.dbg, (16 * 16 + 8) * sizeof(unsigned long long)));
    FAMCHK(hipMemset(ca.dbg, 0, (16 * 16 + 8) * sizeof(unsigned long long)));
    cp_dbg = ca.dbg;
#endif
    for (int c = 0; c < RAMX_CP_NCLASS; c++)
    {
      if (cls_count[c] == 0) continue;
      ca.fam = dfd + cls_first[c];
      rc = ramx_cp_launch_families(d->cls_stream[c], W, cp_k[c], cp_threads[c], cls_count[c], ca);
      if (rc != RAMX_OK) { ramx_set_error("cell-parallel family kernel launch failed (W %d, %d lanes per flank)", W, cp_k[c]); goto done; }
    }
  }
  // ---- one lane per flank ---------------------------------------------------------------------
  if (legacy_any && !resident)
  {
    // rows of every family in the (in-place) row buffer: 16 B x (W + 1) slots per flank
    const size_t state_bytes = (size_t)Np * (W + 1) * sizeof(int4);
    if (state_bytes > d->cap_state || !d->d_state[0])
    {
      if (d->d_state[0]) FAMCHK(hipFree(d->d_state[0]));
      if (d->d_state[1] && d->d_state[1] != d->d_state[0]) FAMCHK(hipFree(d->d_state[1]));
      d->d_state[0] = d->d_state[1] = NULL; d->cap_state = 0;
      FAMCHK(hipMalloc((void **)&d->d_state[0], state_bytes));
      d->d_state[1] = d->d_state[0];
      d->cap_state = state_bytes;
    }
    FSArgs fs;
    memset(&fs, 0, sizeof(fs));
    fs.k.bases = d->d_bases; fs.k.bounds = d->d_bounds; fs.k.trim = d->d_trim; fs.k.S_in = d->d_state[0]; fs.k.S_out = d->d_state[0];
    fs.k.Np = Np; fs.k.Nx = Np; fs.k.W = W; fs.k.go = p->gapopen; fs.k.ge = p->gapextn; fs.k.cap = p->cappenalty;
    fs.k.minimp = p->minimprovement; fs.k.when_to_stop = p->when_to_stop;
    memcpy(fs.k.tab, fa.tab, sizeof(fs.k.tab));
    fs.fam = dfd; fs.ctl_out = dctl; fs.cons_out = d->d_cons; fs.L = L;
    const bool chain = p->gapopen > 0 || p->gapextn > 0 || d->force_chain;
#define RAMX_FS_LAUNCH(CH, BL, C) do { const int g_ = RAMX_CP_NCLASS + (C); if (cls_count[g_] > 0) { fs.fam = dfd + cls_first[g_]; \
      hipLaunchKernelGGL((ramx_family_stream_kernel<CH, BL>), dim3(cls_count[g_]), dim3(BL), 0, d->cls_stream[g_], fs); } } while (0)
    if (chain) { RAMX_FS_LAUNCH(true, 64, 0); RAMX_FS_LAUNCH(true, 128, 1); RAMX_FS_LAUNCH(true, 256, 2); RAMX_FS_LAUNCH(true, 512, 3); }
    else { RAMX_FS_LAUNCH(false, 64, 0); RAMX_FS_LAUNCH(false, 128, 1); RAMX_FS_LAUNCH(false, 256, 2); RAMX_FS_LAUNCH(false, 512, 3); }
#undef RAMX_FS_LAUNCH
    FAMCHK(hipGetLastError());
  }
  else if (legacy_any)
  {
    // the register-resident kernel gains nothing from smaller workgroups (measured: 11.5 vs 10.3 ms; rotating the live
    // waves over the SIMDs by arrival order on the CU did not help either): 256 threads up to 256 flanks (the three
    // smaller shapes are adjacent in the descriptor array), 512 above
    const int g0 = RAMX_CP_NCLASS;
    const int n256 = cls_count[g0] + cls_count[g0 + 1] + cls_count[g0 + 2], n512 = cls_count[g0 + 3];
    if (n256 > 0)
    {
      fa.fam = dfd + cls_first[g0];
      hipStream_t st = d->cls_stream[g0 + 2];
      rc = (W == 14) ? fam_launch<14, 256>(d, fa, n256, st) : (W == 20) ? fam_launch<20, 256>(d, fa, n256, st) : fam_launch<40, 256>(d, fa, n256, st);
      if (rc != RAMX_OK) goto done;
      cls_count[g0 + 2] = n256; cls_count[g0] = cls_count[g0 + 1] = 0;     // for the join below
    }
    if (n512 > 0)
    {
      fa.fam = dfd + cls_first[g0 + 3];
      hipStream_t st = d->cls_stream[g0 + 3];
      rc = (W == 14) ? fam_launch<14, 512>(d, fa, n512, st) : (W == 20) ? fam_launch<20, 512>(d, fa, n512, st) : fam_launch<40, 512>(d, fa, n512, st);
      if (rc != RAMX_OK) goto done;
    }
  }
  for (int c = 0; c < RAMX_NGROUP; c++)
    if (cls_count[c] > 0 || (c == RAMX_NGROUP - 1 && n_dev > 0))
    { FAMCHK(hipEventRecord(d->cls_done[c], d->cls_stream[c])); FAMCHK(hipStreamWaitEvent(d->stream, d->cls_done[c], 0)); }
  FAMCHK(hipEventRecord(d->ev_end, d->stream));
  FAMCHK(hipStreamSynchronize(d->stream));
  FAMCHK(hipEventElapsedTime(&ms, d->ev_begin, d->ev_end));
#ifdef RAMX_PRK_TIMING
  if (resident && legacy_any)
  {
    unsigned long long h[64];
    FAMCHK(hipMemcpy(h, fa.dbg, sizeof(h), hipMemcpyDeviceToHost));
    static const char *nm[6] = { "vote + stop rule", "winner table+barrier", "band + contrib", "wave reduce", "end barrier", "loop top" };
    const double cols = h[6] ? (double)h[6] : 1.0;
    fprintf(stderr, "FAM_TIMING family 0 (block 0), %.0f columns, ns per column per wave:\n", cols);
    for (int k = 0; k < 6; k++)
      fprintf(stderr, "FAM_TIMING %-22s w0 %7.1f  w1 %7.1f  w2 %7.1f  w3 %7.1f\n", nm[k], 10.0 * h[k] / cols, 10.0 * h[8 + k] / cols,
              10.0 * h[16 + k] / cols, 10.0 * h[24 + k] / cols);
  }
  if (fa.dbg) (void)hipFree(fa.dbg);
#endif
#ifdef RAMX_CP_TIMING
  if (cp_dbg)
  {
    unsigned long long h[16 * 16 + 8];
    FAMCHK(hipMemcpy(h, cp_dbg, sizeof(h), hipMemcpyDeviceToHost));
    static const char *nm[12] = { "vote read + stop rule", "(after speculative band)", "row update", "reductions", "records/slide/window",
                                  "sum+atomics+barrier", "-", "loop top", "wait: drain", "wait: samples + polling", "wait: fold",
                                  "wait: LDS + barrier" };
    const double cols = h[16 * 16] ? (double)h[16 * 16] : 1.0;
    fprintf(stderr, "CP_TIMING block 0, %.0f columns, shader clocks per column (waves 0, 1, last two):\n", cols);
    int nw = 0;
    for (int w = 0; w < 16; w++) if (h[w * 16 + 0]) nw = w + 1;
    for (int k = 0; k < 12; k++)
      if (k == 6) fprintf(stderr, "CP_TIMING mispredicted columns: %.0f of %.0f\n", (double)h[6], cols);
      else fprintf(stderr, "CP_TIMING %-24s w0 %7.1f  w1 %7.1f  w%d %7.1f  w%d %7.1f\n", nm[k], h[k] / cols, h[16 + k] / cols,
                          nw > 1 ? nw - 2 : 0, h[(nw > 1 ? nw - 2 : 0) * 16 + k] / cols, nw > 0 ? nw - 1 : 0, h[(nw > 0 ? nw - 1 : 0) * 16 + k] / cols);
  }
#endif
  hctl = (RamxCtl *)malloc(sizeof(RamxCtl) * n_families);
  FAMCHK(hipMemcpy(hctl, dctl, sizeof(RamxCtl) * n_families, hipMemcpyDeviceToHost));
  if (cons && L > 0) FAMCHK(hipMemcpy(cons, d->d_cons, (size_t)n_families * L, hipMemcpyDeviceToHost));
  if ((trim_high || trim_pos) && n_padded)
  {
    tmp = (int2 *)malloc((size_t)n_padded * sizeof(int2));
    FAMCHK(hipMemcpy(tmp, d->d_trim, (size_t)n_padded * sizeof(int2), hipMemcpyDeviceToHost));
    for (int i = 0; i < n_padded; i++) { if (trim_high) trim_high[i] = tmp[i].x; if (trim_pos) trim_pos[i] = tmp[i].y; }
  }
  for (int f = 0; f < n_families; f++)
  {
    const bool gave_up = grp[f] < 0 && hctl[f].pad != 0;
    if (timed_out) timed_out[f] = gave_up ? 1 : 0;
    if (gave_up && !timed_out) { ramx_set_error("batch mode: the vote of a multi-workgroup family timed out (bounded spin gave up)"); rc = RAMX_ERR_STATE; goto done; }
    if (!infos) continue;
    memset(&infos[f], 0, sizeof(ramx_run_info));
    infos[f].ret = hctl[f].max_row + 1;
    infos[f].rows_executed = hctl[f].rows_done;
    infos[f].limit_warning = (hctl[f].stopped && hctl[f].rows_done - 1 == L - 1) ? 1 : 0;
    infos[f].overflow32 = hctl[f].overflow;
    infos[f].n_extendable = fam_count[f];
    infos[f].launches = 1;
    infos[f].loop_ms = ms;
    /* 1: rows resident in registers (3: of K lanes per flank, the cell-parallel kernel); 2: streaming family kernel */
    infos[f].persistent = grp[f] < RAMX_CP_NCLASS ? 1 : (resident ? 1 : 2);
    infos[f].lanes_per_flank = grp[f] < 0 ? dev_k : (grp[f] < RAMX_CP_NCLASS ? cp_k[grp[f]] : 1);
    infos[f].respeculated_rows = grp[f] < 0 ? hctl[f].besta : 0;
  }
  rc = RAMX_OK;
done:
#undef FAMCHK
  if (cp_dbg) (void)hipFree(cp_dbg);
  free(hctl); free(tmp); free(hfd); free(grp);
  d->ready = 0;       // the single-family buffers were reused: begin_direction must be called again before run_direction
  return rc;
}

extern "C" int ramx_dev_run_families(ramx_dev *d, const ramx_flank *flanks, int32_t n_padded, const int32_t *fam_first,
                                     const int32_t *fam_count, int32_t n_families, const ramx_params *p,
                                     ramx_run_info *infos, int8_t *cons, int32_t *trim_high, int32_t *trim_pos)
{
  std::vector<unsigned char> to((size_t)(n_families > 0 ? n_families : 1), 0);
  int rc = run_families_pass(d, flanks, n_padded, fam_first, fam_count, n_families, p, infos, cons, trim_high, trim_pos, true, to.data());
  if (rc != RAMX_OK) return rc;
  // Batch mode degrades, it does not fail: a multi-workgroup family whose device-wide vote gave up is repeated on the
  // one-lane-per-flank route (one workgroup per family, block-local vote: nothing to wait for), the other families'
  // results are kept.
  std::vector<int> redo;
  for (int f = 0; f < n_families; f++) if (to[f]) redo.push_back(f);
  if (redo.empty()) return RAMX_OK;
  fprintf(stderr, "ramx: batch mode: the device-wide vote of %zu multi-workgroup famil%s timed out (bounded spin); repeating "
                  "%s on the lane-per-flank route\n", redo.size(), redo.size() == 1 ? "y" : "ies", redo.size() == 1 ? "it" : "them");
  const int L = p->L;
  std::vector<ramx_flank> fl2;
  std::vector<int32_t> first2, count2;
  for (int f : redo)
  {
    first2.push_back((int32_t)fl2.size());
    count2.push_back(fam_count[f]);
    const int padded = ((fam_count[f] + 63) / 64) * 64;
    for (int i = 0; i < padded; i++)
    {
      ramx_flank x;
      if (i < fam_count[f]) x = flanks[fam_first[f] + i];
      else { memset(&x, 0, sizeof(x)); x.t_lo = 1; x.t_hi = 0; x.step = 1; }      // empty padding flank
      fl2.push_back(x);
    }
  }
  const int n2 = (int)redo.size(), np2 = (int)fl2.size();
  std::vector<ramx_run_info> inf2((size_t)n2);
  std::vector<int8_t> cons2((size_t)n2 * (L > 0 ? L : 1));
  std::vector<int32_t> th2((size_t)np2 + 1), tp2((size_t)np2 + 1);
  rc = run_families_pass(d, fl2.data(), np2, first2.data(), count2.data(), n2, p, inf2.data(), cons2.data(), th2.data(), tp2.data(), false, NULL);
  if (rc != RAMX_OK) return rc;
  for (int k = 0; k < n2; k++)
  {
    const int f = redo[k];
    if (infos) infos[f] = inf2[k];
    if (cons && L > 0) memcpy(cons + (size_t)f * L, cons2.data() + (size_t)k * L, (size_t)L);
    for (int i = 0; i < fam_count[f]; i++)
    {
      if (trim_high) trim_high[fam_first[f] + i] = th2[first2[k] + i];
      if (trim_pos) trim_pos[fam_first[f] + i] = tp2[first2[k] + i];
    }
  }
  return RAMX_OK;
}

extern "C" int ramx_dev_run_direction(ramx_dev *d, ramx_run_info *info)
{
  if (!d || !d->ready) { ramx_set_error("ramx_dev_run_direction: begin_direction has not been called"); return RAMX_ERR_STATE; }
  HIPCHK(hipSetDevice(d->ordinal));
  const ramx_params &p = d->p;
  const int L = p.L;
  KArgs a;
  memset(&a, 0, sizeof(a));
  a.bases = d->d_bases; a.bounds = d->d_bounds; a.trim = d->d_trim; a.cons_out = d->d_cons;
  a.Np = d->Np; a.Nx = d->Nx; a.W = p.bandwidth; a.go = p.gapopen; a.ge = p.gapextn; a.cap = p.cappenalty;
  a.minimp = p.minimprovement; a.when_to_stop = p.when_to_stop;
  memcpy(a.tab, d->tab, sizeof(a.tab));
  const bool multi = dev_is_multi(d);

  auto slot = [&](int r) { return d->d_sums + (size_t)(((r % 3) + 3) % 3) * NSHARD * 4; };
  int launches = 0, nsamp = 0, pending = -1, chk = 0, lanes = 1, prk_local_rc = RAMX_OK;
  bool persistent = false;
  // ---- cell-parallel route (single GPU): K lanes per flank, the whole direction in one cooperative launch of at most one
  // workgroup per CU; boundary row and every column inside the kernel ---------------------------------------------------
  const bool tracing = d->trace_cb != NULL || d->verbose_cb != NULL;
  const bool verbose_trace = d->verbose_cb != NULL;
  const int Bt = 2 * p.bandwidth + 1;
  // hands the DBG buffers of the launch that has just completed to the callbacks
  auto deliver_trace = [&](int r, int besta) -> int
  {
    std::vector<int8_t> codes;
    std::vector<int2> best((size_t)d->Nx + 1), gaps((size_t)d->Nx + 1);
    std::vector<int32_t> bs((size_t)d->Nx + 1), bi((size_t)d->Nx + 1), gfl(2 * (size_t)d->Nx + 2), cand(16 * (size_t)d->Nx + 16);
    if (d->Nx && r >= 0) HIPCHK(hipMemcpy(best.data(), d->d_dbg_best, (size_t)d->Nx * sizeof(int2), hipMemcpyDeviceToHost));
    for (int i = 0; i < d->Nx; i++) { bs[i] = best[i].x; bi[i] = best[i].y; }
    if (d->trace_cb && r >= 0)
    {
      codes.resize((size_t)d->Nx * Bt + 1);
      if (d->Nx) HIPCHK(hipMemcpy(codes.data(), d->d_dbg_codes, (size_t)d->Nx * Bt, hipMemcpyDeviceToHost));
      d->trace_cb(r, besta, codes.data(), bs.data(), bi.data(), d->trace_user);
    }
    if (d->verbose_cb)
    {
      if (d->Nx)
      {
        HIPCHK(hipMemcpy(cand.data(), d->d_dbg_cand, (size_t)d->Nx * 16 * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(gaps.data(), d->d_dbg_gap, (size_t)d->Nx * sizeof(int2), hipMemcpyDeviceToHost));
      }
      for (int i = 0; i < d->Nx; i++) { gfl[2 * i] = gaps[i].x; gfl[2 * i + 1] = gaps[i].y; }
      std::vector<int32_t> band;
      if (d->verbose_band && d->Nx)
      {
        band.resize((size_t)d->Nx * 8 * Bt);
        HIPCHK(hipMemcpy(band.data(), d->d_dbg_band, band.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      }
      d->verbose_cb(r, besta, d->Nx, bs.data(), bi.data(), gfl.data(), cand.data(), band.empty() ? NULL : band.data(), d->verbose_user);
    }
    return RAMX_OK;
  };
  auto trace_buffers = [&](KArgs &ka) -> int
  {
    int trc;
    if ((trc = ensure(&d->d_dbg_best, &d->cap_dbg_best, (size_t)d->Np * sizeof(int2))) != RAMX_OK) return trc;
    ka.dbg_best = d->d_dbg_best;
    ka.dbg_codes = NULL; ka.dbg_cand = NULL; ka.dbg_gap = NULL; ka.dbg_band = NULL;
    if (d->trace_cb)
    {
      if ((trc = ensure(&d->d_dbg_codes, &d->cap_dbg_codes, (size_t)d->Np * Bt)) != RAMX_OK) return trc;
      ka.dbg_codes = d->d_dbg_codes;
    }
    if (verbose_trace)
    {
      if ((trc = ensure(&d->d_dbg_cand, &d->cap_dbg_cand, (size_t)d->Np * 16 * sizeof(int))) != RAMX_OK) return trc;
      if ((trc = ensure(&d->d_dbg_gap, &d->cap_dbg_gap, (size_t)d->Np * sizeof(int2))) != RAMX_OK) return trc;
      ka.dbg_cand = d->d_dbg_cand; ka.dbg_gap = d->d_dbg_gap;
      if (d->verbose_band)
      {
        if ((trc = ensure(&d->d_dbg_band, &d->cap_dbg_band, (size_t)d->Np * 8 * Bt * sizeof(int))) != RAMX_OK) return trc;
        ka.dbg_band = d->d_dbg_band;
      }
    }
    return RAMX_OK;
  };
  // Multi-rank: the vote crosses the devices through the mailboxes (ramx_dev_peer_* set-up), exactly as in the
  // lane-per-flank persistent kernel; every rank must take this route or none (one agreement before, one after).
  // every route but the packed-row one reads base words of the whole window: pack what begin_direction left (pack_rest)
  rt_mark(NULL);
  {
    bool lazy = false;
    if (!tracing && d->Nx > 0)
    {
      int blk = 0, blks = 0, sp = 0, rb = 0, pb = 0, pbs = 0;
      bool can32 = false;
      int lrc = prk_local_can(d, a, L, multi, &can32, &blk, &blks);
      if (lrc != RAMX_OK) return lrc;
      if ((lrc = pk_route(d, a, L, multi, can32, &lazy, &sp, &rb, &pb, &pbs)) != RAMX_OK) return lrc;
      // (the device-wide cell-parallel kernel goes first where it applies: it takes the whole window)
      // (multi-rank: the route is agreed below; whichever kernel then needs the whole window packs the rest first)
      if (lazy && !multi && d->cp_flanks_ok && getenv("RAMX_NO_CP_DEVICE") == NULL && ramx_cp_max_family(a.W, a.go, a.ge, d->tab, L) > 0)
      {
        int dev = 0, cus = 0, k = 0, th = 0, nb = 0, vwf = 0;
        HIPCHK(hipGetDevice(&dev));
        HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const char *mx = getenv("RAMX_CP_DEVICE_MAXN");
        if (mx == NULL || d->Nx <= atoi(mx)) ramx_cp_device_plan(a.W, d->Nx, cus, 0, &k, &th, &nb, &vwf);
        if (k > 0 && k < RAMX_CP_MIN_K_OVER_PACKED && getenv("RAMX_CP_K") == NULL) k = 0;      // (the packed-row kernel is the faster one there, see below)
        if (k > 0) lazy = false;
      }
    }
    if (!lazy) { const int prc = pack_rest(d); if (prc != RAMX_OK) return prc; }
  }
  rt_mark("route (occupancy answers)");
  const bool cp_multi_ok = !multi || (d->peer_ready && d->nranks >= 2 && L < 65536 && getenv("RAMX_NO_PEER") == NULL);
  if (!tracing && cp_multi_ok && !d->force_chain && L > 0 && getenv("RAMX_NO_PERSISTENT") == NULL && getenv("RAMX_NO_CP_DEVICE") == NULL &&
      (multi || d->Nx > 0))
  {
    int dev = 0, cus = 0, k = 0, th = 0, nb = 0, vwf = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const char *mx = getenv("RAMX_CP_DEVICE_MAXN");
    if (d->cp_flanks_ok && ramx_cp_max_family(a.W, a.go, a.ge, d->tab, L) > 0 && (mx == NULL || d->Nx <= atoi(mx)))
      ramx_cp_device_plan(a.W, d->Nx > 0 ? d->Nx : 1, cus, 0, &k, &th, &nb, &vwf);
    if (k > 0 && k < RAMX_CP_MIN_K_OVER_PACKED && getenv("RAMX_CP_K") == NULL)
    {
      // Two lanes per flank (33,000-65,536 flanks at W = 40) against the packed-row kernel with one wave per SIMD, whole bench launch
      // at 50,000 / 65,000 flanks: 5.44 / 5.55 against 4.76 / 4.66 us per column (aligned phase alike, the capped tail is the packed
      // kernel's; profiles/r04_route_ab.log) -- where the packed rows can run they take those sets.
      int blk = 0, blks = 0, sp = 0, rb = 0, pb = 0, pbs = 0;
      bool can32 = false, pkq = false;
      int lrc = prk_local_can(d, a, L, multi, &can32, &blk, &blks);
      if (lrc != RAMX_OK) return lrc;
      if ((lrc = pk_route(d, a, L, multi, can32, &pkq, &sp, &rb, &pb, &pbs)) != RAMX_OK) return lrc;
      if (pkq) k = 0;
    }
    if (multi)
    {
      // my mailbox is cleared BEFORE the agreement, which no remote launch can get past without my taking part
      if (d->hostbox_host)
      {
        HIPCHK(hipStreamSynchronize(d->stream));
        memset((void *)d->hostbox_host, 0, sizeof(PeerBox));
        __sync_synchronize();
      }
      else
      {
        // complete, not just enqueued: the agreement below may be a host-only collective, and a peer that gets past it
        // launches and writes into this box at once
        HIPCHK(hipMemsetAsync(d->xbox, 0, sizeof(PeerBox), d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
      }
      int cannot = k > 0 ? 0 : 1;
      int arc = host_allreduce_flag(d, &cannot);
      if (arc != RAMX_OK) return arc;
      if (cannot) k = 0;
    }
    if (k > 0)
    {
      // from here on a multi-rank run never leaves before the second agreement (a rank that did would leave the others'
      // kernels spinning to their limit and then waiting in a collective it never joins)
      int lrc = RAMX_OK;
      bool accepted = false;      // the launch call went through
      auto launch = [&]() -> int
      {
        { const int prc = pack_rest(d); if (prc != RAMX_OK) return prc; }      // this kernel reads the whole window
        CPArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.bases = d->d_bases; ca.bounds = d->d_bounds; ca.trim = d->d_trim; ca.ctl_out = d->d_ctl; ca.cons_out = d->d_cons;
        ca.Np = d->Np; ca.KW = d->KW; ca.L = L; ca.go = a.go; ca.ge = a.ge; ca.cap = a.cap; ca.minimp = a.minimp; ca.when_to_stop = a.when_to_stop;
        memcpy(ca.tab, d->tab, sizeof(ca.tab));
        {
          std::vector<CpDevDesc> hd((size_t)nb);
          for (int i = 0; i < nb; i++) { memset(&hd[i], 0, sizeof(CpDevDesc)); hd[i].first = 0; hd[i].nx = d->Nx; hd[i].b = i; hd[i].nb = nb; hd[i].id = 0; }
          int drc = ensure(&d->d_devdesc, &d->cap_devdesc, sizeof(CpDevDesc) * (size_t)nb);
          if (drc != RAMX_OK) return drc;
          HIPCHK(hipMemcpyAsync(d->d_devdesc, hd.data(), sizeof(CpDevDesc) * (size_t)nb, hipMemcpyHostToDevice, d->stream));
          HIPCHK(hipStreamSynchronize(d->stream));       // hd goes out of scope
        }
        ca.dev = d->d_devdesc; ca.vote = d->d_vote; ca.err = d->d_err; ca.S = d->d_state[0]; ca.vote_wave = vwf;
        { const char *td = getenv("RAMX_TEST_CP_DROP_TICKET"); ca.test_drop_row = td ? atoi(td) : 0; ca.test_drop_id = -1; }
        cp_test_hooks(ca);
        ca.nranks = 1; ca.rank = 0;
        if (multi)
        {
          ca.nranks = d->nranks; ca.rank = d->rank; ca.peers = (PeerBox *const *)d->d_peer; ca.box = d->xbox; ca.mirror = NULL;
          if (d->hostbox_host)
          {
            if (!d->hostbox_mirror) HIPCHK(hipMalloc((void **)&d->hostbox_mirror, sizeof(PeerBox)));
            HIPCHK(hipMemsetAsync(d->hostbox_mirror, 0, sizeof(PeerBox), d->stream));
            ca.mirror = d->hostbox_mirror;
          }
          const char *tf = getenv("RAMX_TEST_FAIL_PRK_RANK");      // test hook: this rank fails where a launch error would
          if (tf && atoi(tf) == d->rank) { ramx_set_error("test hook: forced failure of the cell-parallel launch on rank %d", d->rank); return RAMX_ERR_HIP; }
        }
        HIPCHK(hipMemsetAsync(d->d_vote, 0, RAMX_CP_NSETS * NSHARD * sizeof(PShard), d->stream));
        HIPCHK(hipMemsetAsync(d->d_err, 0, 64, d->stream));
        HIPCHK(hipMemsetAsync(d->d_ctl, 0, 2 * sizeof(RamxCtl), d->stream));
#ifdef RAMX_CP_TIMING
        HIPCHK(hipMalloc((void **)&ca.dbg, (16 * 16 + 8) * sizeof(unsigned long long)));
        HIPCHK(hipMemset(ca.dbg, 0, (16 * 16 + 8) * sizeof(unsigned long long)));
#endif
        HIPCHK(hipEventRecord(d->ev_begin, d->stream));
        if (multi) test_delay_rank(d);
        int crc = ramx_cp_launch_device(d->stream, a.W, k, th, nb, ca);
        if (crc != RAMX_OK)
        {
#ifdef RAMX_CP_TIMING
          (void)hipFree(ca.dbg);
#endif
          ramx_set_error("cell-parallel device launch failed (W %d, %d lanes per flank, %d workgroups)", a.W, k, nb);
          return crc;
        }
        accepted = true;          // from here on an error is the kernel's (an execution error: not something another route can serve)
        HIPCHK(hipEventRecord(d->ev_end, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
#ifdef RAMX_CP_TIMING
        if (vwf)
        {
          unsigned long long h[16 * 16 + 8];
          HIPCHK(hipMemcpy(h, ca.dbg, sizeof(h), hipMemcpyDeviceToHost));
          const double cols = h[16 * 16] ? (double)h[16 * 16] : 1.0;
          fprintf(stderr, "SP_TIMING workgroup 0, %d workgroups x %d threads, K %d, %.0f rows; shader clocks (100 MHz) per row\n", nb, th, k, cols);
          fprintf(stderr, "SP_TIMING vote wave: tickets wait+fold %.1f | winner+forward %.1f | stop rule, decision %.1f | loop %.1f | waiting for band sums %.1f | forwarded at once %.0f %%\n",
                  h[0] / cols, h[1] / cols, h[2] / cols, h[3] / cols, h[4] / cols, 100.0 * h[5] / cols);
          for (int wv = 1; wv < th / 64; wv++)
            fprintf(stderr, "SP_TIMING band wave %d: idle %.1f | planning %.1f | rows %.1f\n", wv, h[wv * 16 + 0] / cols, h[wv * 16 + 1] / cols, h[wv * 16 + 2] / cols);
        }
        (void)hipFree(ca.dbg);
#endif
        return RAMX_OK;
      };
      lrc = launch();
      RamxCtl c0;
      memset(&c0, 0, sizeof(c0));
      int bad = lrc != RAMX_OK;
      if (!bad && hipMemcpy(&c0, d->d_ctl, sizeof(c0), hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
      bad = bad || c0.pad != 0;
      if (multi)
      {
        if (lrc != RAMX_OK) (void)hipStreamSynchronize(d->stream);
        int frc = host_allreduce_flag(d, &bad);
        if (frc != RAMX_OK) return frc;
        if (bad)
        {
          fprintf(stderr, "ramx: cross-device cell-parallel launch gave up or failed on some rank; repeating the direction with per-column "
                          "launches and leaving the mailbox path off for the rest of this process\n");
          d->peer_ready = 0;        // agreed by all ranks (the flag above is reduced)
        }
      }
      else if (lrc != RAMX_OK && accepted)
        return lrc;               // the kernel ran and failed (a fault is sticky: the context is gone): the error as it was reported
      else if (lrc != RAMX_OK)
      {
        // single GPU: the other routes can serve the same flank set -- an occupancy query that says no, a launch configuration
        // this one instantiation refuses -- so a refusal here is not the direction's failure
        (void)hipStreamSynchronize(d->stream);
        (void)hipGetLastError();
        fprintf(stderr, "ramx: cell-parallel device launch not possible (%s); continuing on the lane-per-flank route\n", ramx_last_error());
      }
      else if (bad)
        fprintf(stderr, "ramx: device-wide vote of the cell-parallel launch timed out (bounded spin); repeating the direction with "
                        "per-column launches\n");
      if (!bad) { persistent = true; lanes = k; launches = 1; }
    }
  }
  const bool cp_done = persistent;
  rt_mark("cell-parallel plan / launch");
  if (!cp_done)
  {
  HIPCHK(hipMemsetAsync(d->d_sums, 0, 3 * NSHARD * 4 * sizeof(long long), d->stream));
  HIPCHK(hipEventRecord(d->ev_begin, d->stream));
  // K(-1): boundary row + candidates of row 0
  a.r = -1; a.S_in = d->d_state[0]; a.S_out = d->d_state[1]; a.ctl_in = d->d_ctl; a.ctl_out = d->d_ctl + 1;
  a.sums_in = slot(0); a.sums_out = slot(0); a.sums_zero = slot(1); a.nshards_in = NSHARD;
  if (verbose_trace)
  {
    int trc = trace_buffers(a);
    if (trc != RAMX_OK) return trc;
    launch_column_trace(d, a, true);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->stream));
    if ((trc = deliver_trace(-1, 0)) != RAMX_OK) return trc;
  }
  else launch_column<true>(d, a);
  HIPCHK(hipGetLastError());
  rt_mark("boundary row launched");
  {
    // the in-place row buffer holds S(-1) after K(-1); d_ctl[1] holds the initial control block, the persistent
    // kernel writes its final one to d_ctl[0]
    if (d->d_state[0] != d->d_state[1]) { ramx_set_error("persistent path needs the in-place row buffer"); }
    else if (!tracing)
    {
      prk_local_rc = prk_run(d, a, L, &persistent);
      // Single GPU: a failure is simply returned.  Multi-rank: once the ranks have agreed on the persistent path
      // (prk_run sets `persistent` after that agreement) a rank-local failure -- an allocation, a memset, the launch
      // itself -- must NOT leave before the agreement below, or the other ranks' kernels spin to their limit and then
      // wait in a collective this rank never joins.
      if (prk_local_rc != RAMX_OK && !(multi && persistent)) return prk_local_rc;
    }
  }
  if (persistent && multi)
  {
    // agree over all ranks whether the cross-device launch went through; if any rank gave up (bounded spin) or failed
    // locally, every rank repeats the direction with the per-column launches and the host collective
    int bad = prk_local_rc != RAMX_OK;
    RamxCtl c0[2];
    unsigned errw = 0;
    memset(c0, 0, sizeof(c0));
    if (!bad)
    {
      // (both control blocks: a direction that ran in pieces leaves its last one in either; the error word: a refused entry check)
      if (hipStreamSynchronize(d->stream) != hipSuccess || hipMemcpy(c0, d->d_ctl, sizeof(c0), hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(&errw, d->d_err, sizeof(errw), hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
    }
    else (void)hipStreamSynchronize(d->stream);
    bad = bad || c0[0].pad != 0 || c0[1].pad != 0 || errw != 0;
    int frc = host_allreduce_flag(d, &bad);
    if (frc != RAMX_OK) return frc;
    if (bad)
    {
      fprintf(stderr, "ramx: cross-device persistent launch gave up on some rank; repeating the direction with per-column launches "
                      "and leaving the mailbox path off for the rest of this process\n");
      persistent = false;
      d->peer_ready = 0;          // agreed by all ranks (the flag above is reduced): nobody tries the mailboxes again
      HIPCHK(hipMemsetAsync(d->d_sums, 0, 3 * NSHARD * 4 * sizeof(long long), d->stream));
      a.r = -1; a.S_in = d->d_state[0]; a.S_out = d->d_state[1]; a.ctl_in = d->d_ctl; a.ctl_out = d->d_ctl + 1;
      a.sums_in = slot(0); a.sums_out = slot(0); a.sums_zero = slot(1); a.nshards_in = NSHARD;
      launch_column<true>(d, a);
      HIPCHK(hipMemsetAsync(d->d_ctl, 0, sizeof(RamxCtl), d->stream));
    }
  }
  else if (persistent)
  {
    // single GPU: a barrier that timed out (bounded spin, e.g. a co-tenant holding CUs) is not fatal -- repeat the
    // direction with the per-column launches, exactly as the multi-rank branch above does
    HIPCHK(hipStreamSynchronize(d->stream));
    RamxCtl c0[2];
    unsigned errw = 0;
    {
      unsigned long long ew = 0;
      int drc;
      if ((drc = d2h_small(d, c0, d->d_ctl, sizeof(c0))) != RAMX_OK) return drc;
      if ((drc = d2h_small(d, &ew, d->d_err, sizeof(ew))) != RAMX_OK) return drc;
      errw = (unsigned)ew;
    }
    rt_mark("after the launch: control blocks, error word");
    if (c0[0].pad != 0 || c0[1].pad != 0 || errw != 0)
    {
      if (errw == 2 || errw == 3)
        fprintf(stderr, "ramx: the packed-row kernel refused %s (a cell outside the span computed for this scoring system); repeating "
                        "the direction with per-column launches\n", errw == 3 ? "the rows it was handed" : "a row it computed");
      else
      fprintf(stderr, "ramx: device-wide barrier of the persistent launch timed out (bounded spin); repeating the direction with "
                      "per-column launches\n");
      persistent = false;
      HIPCHK(hipMemsetAsync(d->d_sums, 0, 3 * NSHARD * 4 * sizeof(long long), d->stream));
      a.r = -1; a.S_in = d->d_state[0]; a.S_out = d->d_state[1]; a.ctl_in = d->d_ctl; a.ctl_out = d->d_ctl + 1;
      a.sums_in = slot(0); a.sums_out = slot(0); a.sums_zero = slot(1); a.nshards_in = NSHARD;
      launch_column<true>(d, a);
      HIPCHK(hipMemsetAsync(d->d_ctl, 0, sizeof(RamxCtl), d->stream));
    }
  }
  }
  d->last_persistent = persistent ? 1 : 0;
  if (!persistent) { const int prc = pack_rest(d); if (prc != RAMX_OK) return prc; }      // the column launches read the whole window
  const int CHUNK = 64;
  const int stride = L > MAX_SAMPLES * 4 ? L / MAX_SAMPLES : 4;
  bool stopped = false;
  memset(d->h_ctl, 0, 4 * sizeof(RamxCtl));
  for (int r = 0; r < L && !stopped && !persistent; r++)
  {
    if (multi)
    {
      // the vote shards of row r (32 x 4 x int64 = 1 KB) are summed across ranks in place; the column kernel then
      // folds them exactly as in the single-GPU case.  One collective per column, no extra kernel.
      int hrc = host_allreduce_shards(d, slot(r));
      if (hrc != RAMX_OK) return hrc;
    }
    a.sums_in = slot(r); a.nshards_in = NSHARD;
    a.r = r; a.S_in = d->d_state[(r + 1) & 1]; a.S_out = d->d_state[r & 1];
    a.ctl_in = d->d_ctl + ((r + 1) & 1); a.ctl_out = d->d_ctl + (r & 1);
    a.sums_out = slot(r + 1); a.sums_zero = slot(r + 2);
    if (tracing)
    {
      // one row at a time: launch, wait, hand the row's trace to the caller (debugging aid, no attempt at speed)
      int trc = trace_buffers(a);
      if (trc != RAMX_OK) return trc;
      launch_column_trace(d, a);
      launches++;
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(d->stream));
      RamxCtl c;
      HIPCHK(hipMemcpy(&c, a.ctl_out, sizeof(c), hipMemcpyDeviceToHost));
      if (c.rows_done == r + 1)      // the launch ran (a stopped predecessor makes the kernel return at once)
        if ((trc = deliver_trace(r, c.besta)) != RAMX_OK) return trc;
      if (c.stopped) stopped = true;
      continue;
    }
    const bool sample = (r % stride) == (stride / 2) && nsamp < MAX_SAMPLES;
    if (sample) HIPCHK(hipEventRecord(d->ev_s0[nsamp], d->stream));
    launch_column<false>(d, a);
    if (sample) { HIPCHK(hipEventRecord(d->ev_s1[nsamp], d->stream)); nsamp++; }
    launches++;
    if (((r + 1) % CHUNK) == 0 || r == L - 1)
    {
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(d->h_ctl + 2 * chk, d->d_ctl, 2 * sizeof(RamxCtl), hipMemcpyDeviceToHost, d->stream));
      HIPCHK(hipEventRecord(d->ev_chk[chk], d->stream));
      if (pending >= 0)
      {
        HIPCHK(hipEventSynchronize(d->ev_chk[pending]));
        const RamxCtl *h = d->h_ctl + 2 * pending;
        if (h[0].stopped || h[1].stopped) stopped = true;
      }
      pending = chk;
      chk ^= 1;
    }
  }
  if (!cp_done) HIPCHK(hipEventRecord(d->ev_end, d->stream));
  rt_mark("loop enqueued / pieces run");
  HIPCHK(hipStreamSynchronize(d->stream));
  RamxCtl h[2];
  { const int drc = d2h_small(d, h, d->d_ctl, sizeof(h)); if (drc != RAMX_OK) return drc; }
  rt_mark("final synchronisation, control blocks");
  const RamxCtl &f = (L == 0) ? h[1] : ((h[0].rows_done > h[1].rows_done) ? h[0] : h[1]);
  d->final_ctl = f;
  if (persistent)
  {
    launches = 1;
    if (f.pad != 0) { ramx_set_error("persistent kernel: device-wide barrier timed out (bounded spin gave up)"); return RAMX_ERR_STATE; }
  }
  if (info)
  {
    info->ret = f.max_row + 1;
    info->rows_executed = f.rows_done;
    info->limit_warning = (f.stopped && f.rows_done - 1 == L - 1) ? 1 : 0;   // ram_extend.c:1225-1231
    info->overflow32 = f.overflow;
    info->n_extendable = d->Nx;
    info->launches = launches;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, d->ev_begin, d->ev_end));
    info->loop_ms = ms;
    double acc = 0; int cnt = 0;
    for (int i = 0; i < nsamp; i++)
    {
      float t = 0;
      if (hipEventElapsedTime(&t, d->ev_s0[i], d->ev_s1[i]) == hipSuccess) { acc += t; cnt++; }
    }
    info->kernel_ms_avg = cnt ? acc / cnt : 0.0;
    info->kernel_samples = cnt;
    info->persistent = persistent ? 1 : 0;
    info->lanes_per_flank = lanes;
    info->respeculated_rows = cp_done ? f.besta : 0;
    info->packed_rows = (persistent && !cp_done && d->last_packed_r0 >= 0 && f.rows_done > d->last_packed_r0) ? f.rows_done - d->last_packed_r0 : 0;
    info->lean_rows = (persistent && !cp_done && d->last_packed_r0 >= 0) ? (f.besta & 0xffff) : 0;
    if (persistent && !cp_done && d->last_packed_r0 >= 0) info->respeculated_rows = (f.besta >> 16) & 0xffff;
  }
  return RAMX_OK;
}

extern "C" int ramx_dev_download(ramx_dev *d, int8_t *cons, int32_t cons_cap, int32_t *trim_high, int32_t *trim_pos)
{
  if (!d || !d->ready) { ramx_set_error("ramx_dev_download: nothing to download"); return RAMX_ERR_STATE; }
  HIPCHK(hipSetDevice(d->ordinal));
  const int rows = d->final_ctl.rows_done;
  const bool timing = getenv("RAMX_TIMING") != NULL;
  const double td0 = now_ms();
  if (cons)
  {
    if (cons_cap < rows) { ramx_set_error("cons buffer too small (%d < %d)", cons_cap, rows); return RAMX_ERR_ARG; }
    if (rows) { const int drc = d2h_small(d, cons, d->d_cons, (size_t)rows); if (drc != RAMX_OK) return drc; }
  }
  if ((trim_high || trim_pos) && d->Nx)
  {
    // through a pinned buffer kept with the session: the copy into pageable memory pins the caller's pages on the fly (8 ms for
    // 800 KB at N = 100,000)
    const size_t need = (size_t)d->Nx * sizeof(int2);
    if (need > d->cap_stage)
    {
      if (d->h_stage) HIPCHK(hipHostFree(d->h_stage));
      d->h_stage = NULL; d->cap_stage = 0;
      HIPCHK(hipHostMalloc(&d->h_stage, need + need / 2, hipHostMallocDefault));
      d->cap_stage = need + need / 2;
    }
    int2 *tmp = (int2 *)d->h_stage;
    const double td1 = now_ms();
    if (getenv("RAMX_DOWNLOAD_DMA"))
      HIPCHK(hipMemcpyAsync(tmp, d->d_trim, need, hipMemcpyDeviceToHost, d->stream));
    else
    {
      hipLaunchKernelGGL(ramx_to_host_kernel, dim3((d->Nx + 255) / 256), dim3(256), 0, d->stream, (const int2 *)d->d_trim, tmp, d->Nx);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(d->stream));
    if (timing) fprintf(stderr, "RAMX_TIMING       download: consensus %.3f ms, trim copy %.3f ms\n", td1 - td0, now_ms() - td1);
    for (int i = 0; i < d->Nx; i++)
    {
      if (trim_high) trim_high[i] = tmp[i].x;
      if (trim_pos) trim_pos[i] = tmp[i].y;
    }
  }
  return RAMX_OK;
}

extern "C" int ramx_dev_peek_state(ramx_dev *d, int32_t flank, int32_t *cells, int32_t *high, int32_t *pos)
{
  if (!d || !d->ready || flank < 0 || flank >= d->Nx) { ramx_set_error("ramx_dev_peek_state: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  const int W = d->p.bandwidth, Q = W + 1, B = 2 * W + 1;
  const int rows = d->final_ctl.rows_done;
  const int4 *S = d->d_state[(rows - 1) & 1];   // K(r) wrote state[r & 1]; rows == 0 -> boundary row in slot 1
  const int tile = flank / 64, lane = flank % 64;
  for (int q = 0; q < Q; q++)
  {
    int4 v;
    HIPCHK(hipMemcpy(&v, S + ((size_t)tile * Q + q) * 64 + lane, sizeof(v), hipMemcpyDeviceToHost));
    cells[4 * q] = v.x; cells[4 * q + 1] = v.y;
    if (2 * q + 1 < B) { cells[4 * q + 2] = v.z; cells[4 * q + 3] = v.w; }
    else { if (high) *high = v.z; if (pos) *pos = v.w; }
  }
  return RAMX_OK;
}

extern "C" int ramx_dev_peek_family_state(ramx_dev *d, int32_t flank, int32_t *cells)
{
  if (!d) d = ramx_default_device();      // seam 1 runs on the process-wide session
  if (!d || !d->d_cpstate || flank < 0 || flank >= d->cpstate_n || !cells)
  { ramx_set_error("ramx_dev_peek_family_state: nothing kept (set RAMX_CP_PEEK=1 before the batch call)"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  const int B = 2 * d->cpstate_W + 1;
  HIPCHK(hipMemcpy(cells, d->d_cpstate + (size_t)flank * B, (size_t)B * sizeof(int2), hipMemcpyDeviceToHost));
  return RAMX_OK;
}

// Largest family (extendable cores in one direction) that seam 1 should hand to the one-workgroup-per-family route; larger
// sets go through ramx_dev_run_direction, whose device-wide cell-parallel kernel beats the lane-per-flank family kernel
// (W = 40, 500 flanks: 2.6 vs 7.3 us per column).  512 where the cell-parallel kernel does not apply.
extern "C" int ramx_dev_family_route_max(ramx_dev *d, const ramx_params *p)
{
  if (!d || !p || !p->matrix) return 512;
  int tab9[RAMX_NCLASS][4];
  for (int c = 0; c < RAMX_NCLASS; c++)
  {
    const int code = (c == 8) ? RAMX_SYM_N : c;
    for (int k = 0; k < 4; k++) tab9[c][k] = p->matrix[k * 100 + code];
  }
  if (d->force_chain || getenv("RAMX_NO_CP_DEVICE") != NULL || getenv("RAMX_NO_PERSISTENT") != NULL) return 512;
  const int m = ramx_cp_max_family(p->bandwidth, p->gapopen, p->gapextn, tab9, p->L);
  if (m <= 0) return 512;
  const int s1 = ramx_cp_single_family_max(p->bandwidth);
  return s1 < m ? s1 : m;
}

extern "C" int ramx_dev_set_row_trace(ramx_dev *d, ramx_row_trace_cb cb, void *user)
{
  if (!d) d = ramx_default_device();
  if (!d) return RAMX_ERR_NO_DEVICE;
  d->trace_cb = cb;
  d->trace_user = user;
  return RAMX_OK;
}

extern "C" int ramx_dev_set_row_verbose(ramx_dev *d, ramx_row_verbose_cb cb, void *user)
{
  if (!d) d = ramx_default_device();
  if (!d) return RAMX_ERR_NO_DEVICE;
  d->verbose_cb = cb;
  d->verbose_user = user;
  return RAMX_OK;
}

extern "C" int ramx_dev_set_verbose_band(ramx_dev *d, int on)
{
  if (!d) d = ramx_default_device();
  if (!d) return RAMX_ERR_NO_DEVICE;
  d->verbose_band = on ? 1 : 0;
  return RAMX_OK;
}

extern "C" int ramx_dev_set_allreduce_cb(ramx_dev *d, ramx_allreduce_cb cb, void *user)
{
  if (!d) { ramx_set_error("ramx_dev_set_allreduce_cb: bad argument"); return RAMX_ERR_ARG; }
  d->cb = cb;
  d->cb_user = user;
  return RAMX_OK;
}

extern "C" int ramx_comm_unique_id(uint8_t id[128])
{
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) { ramx_set_error("ncclGetUniqueId: %s", ncclGetErrorString(r)); return RAMX_ERR_COMM; }
  memcpy(id, &u, 128);
  return RAMX_OK;
}

extern "C" int ramx_dev_comm_init(ramx_dev *d, const uint8_t id[128], int rank, int nranks)
{
  if (!d || rank < 0 || rank >= nranks) { ramx_set_error("ramx_dev_comm_init: bad argument"); return RAMX_ERR_ARG; }
  HIPCHK(hipSetDevice(d->ordinal));
  d->rank = rank; d->nranks = nranks;
  if (nranks == 1 && getenv("RAMX_COMM_SINGLE") == NULL) return RAMX_OK;       // one rank: nothing to exchange (RAMX_COMM_SINGLE=1: make the communicator all the same)
  ncclUniqueId u;
  memcpy(&u, id, 128);
  ncclResult_t r = ncclCommInitRank(&d->comm, nranks, u, rank);
  if (r != ncclSuccess) { ramx_set_error("ncclCommInitRank: %s", ncclGetErrorString(r)); d->comm = NULL; return RAMX_ERR_COMM; }
  return RAMX_OK;
}

extern "C" int ramx_dev_comm_size(ramx_dev *d)
{
  if (!d) { ramx_set_error("ramx_dev_comm_size: bad argument"); return RAMX_ERR_ARG; }
  if (!d->comm) return 1;
  int n = 0;
  ncclResult_t r = ncclCommCount(d->comm, &n);
  if (r != ncclSuccess) { ramx_set_error("ncclCommCount: %s", ncclGetErrorString(r)); return RAMX_ERR_COMM; }
  return n;
}
