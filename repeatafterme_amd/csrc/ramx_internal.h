/* ramx_internal.h -- shared between the host C files and the HIP device layer (not installed). */
#ifndef RAMX_INTERNAL_H
#define RAMX_INTERNAL_H

#include "ramx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RAMX_NEG_IMPOSSIBLE (-1000000000)   /* reference bnw_extend.c:760,802-804 */
#define RAMX_OOB_SENTINEL   (-987654321)    /* reference bnw_extend.c:767 */
#define RAMX_NCLASS 9                        /* base classes on the device: A C G T a c g t N */

void ramx_set_error(const char *fmt, ...);
int ramx_runtime_verbose(void);
int ramx_runtime_when_to_stop(void);
int ramx_runtime_l(void);

/* seam 1's routing: families up to this many extendable cores run as a batch of one (csrc/ramx_device.hip) */
int ramx_dev_family_route_max(ramx_dev *d, const ramx_params *p);

/* the packed twin of a library made by ramx_load_sequence_subset_packed (NULL for any other library) */
const ramx_packed_library *ramx_packed_of(const struct sequenceLibrary *lib);
/* a library is about to be freed: the device copy it may own is forgotten (csrc/ramx_extend.c) */
void ramx_forget_library_owner(const struct sequenceLibrary *lib);
/* one base of a library of either kind (1 byte per base, or packed) */
int ramx_lib_code(const struct sequenceLibrary *lib, uint64_t at);

/* host threads for a loop over `items` (at most 16, at most one per min_items_per_thread; RAMX_HOST_THREADS overrides) */
int ramx_host_threads(long items, long min_items_per_thread);
/* fn formats items [lo, hi) into outs[0..n_outs) (memory streams; an entry is NULL where the real stream is); the chunks'
 * bytes reach real_outs[] in chunk order (csrc/ramx_par.c) */
#define RAMX_PAR_MAX_OUTS 4
typedef void (*ramx_chunk_fn)(int lo, int hi, FILE **outs, void *user);
void ramx_parallel_chunks(int n, int n_outs, FILE **real_outs, ramx_chunk_fn fn, void *user);
/* fn(lo, hi, user) over [0, n) on the host's cores, at most one thread per min_items items */
typedef void (*ramx_range_fn)(int lo, int hi, void *user);
void ramx_parallel_for(int n, long min_items, ramx_range_fn fn, void *user);

/* -vvvv support (reference ram_extend.c:992-1090, 1134-1214): with this callback set a direction runs one column launch at a
 * time (full candidate recurrence) and hands over, after the boundary launch (row = -1) and after every executed row: per
 * flank the winner row's best score / index / gap state of its first and last cell (row >= 0), and the four candidate rows
 * row + 1: cand[i][16] = best[4], best cell[4], gap of the first cell[4], gap of the last cell[4].  d == NULL: seam 1's
 * session.  A debugging aid: slow by construction. */
typedef void (*ramx_row_verbose_cb)(int32_t row, int32_t besta, int32_t n_flanks, const int32_t *best_score, const int32_t *best_idx,
                                    const int32_t *gap_first_last /* [n][2] */, const int32_t *cand /* [n][16] */,
                                    const int32_t *band /* -vvvvv: [n][4][2W+1][2] = (sub, gap) of every candidate cell; else NULL */, void *user);
int ramx_dev_set_row_verbose(ramx_dev *d, ramx_row_verbose_cb cb, void *user);
/* -vvvvv (ram_extend.c:1013-1024): the callback also gets every cell of the four candidate rows */
int ramx_dev_set_verbose_band(ramx_dev *d, int on);

/* process-wide device session used by seam 1 (created on first use) */
ramx_dev *ramx_default_device(void);

#ifdef __cplusplus
}
#endif
#endif
