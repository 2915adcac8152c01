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
/* one base of a library of either kind (1 byte per base, or packed) */
int ramx_lib_code(const struct sequenceLibrary *lib, uint64_t at);

/* host threads for a loop over `items` (at most 16, at most one per min_items_per_thread; RAMX_HOST_THREADS overrides) */
int ramx_host_threads(long items, long min_items_per_thread);
/* fn formats items [lo, hi) into outs[0..n_outs) (memory streams; an entry is NULL where the real stream is); the chunks'
 * bytes reach real_outs[] in chunk order (csrc/ramx_par.c) */
#define RAMX_PAR_MAX_OUTS 4
typedef void (*ramx_chunk_fn)(int lo, int hi, FILE **outs, void *user);
void ramx_parallel_chunks(int n, int n_outs, FILE **real_outs, ramx_chunk_fn fn, void *user);

/* process-wide device session used by seam 1 (created on first use) */
ramx_dev *ramx_default_device(void);

#ifdef __cplusplus
}
#endif
#endif
