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

/* process-wide device session used by seam 1 (created on first use) */
ramx_dev *ramx_default_device(void);

#ifdef __cplusplus
}
#endif
#endif
