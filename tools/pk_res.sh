#!/bin/bash
# resources of ONE instantiation of the packed-row kernel (fast compile): tools/pk_res.sh [W] [BLOCK] [extra hipcc flags]
cd "$(dirname "$0")/../repeatafterme_amd/csrc"
W=${1:-40}; BL=${2:-512}; shift; shift
cat > /tmp/pk_one_$$.hip <<EOT
#define RAMX_SECONDARY_TU 1
#include "ramx_kernels_packed.h"
template __global__ void ramx_packed_kernel<$W, $BL>(const PKArgs a);
EOT
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -I. "$@" -Rpass-analysis=kernel-resource-usage -save-temps=obj -c /tmp/pk_one_$$.hip -o /tmp/pk_one_$$.o 2>&1 |
  grep -E "error|VGPRs:|Spill|ScratchSize|LDS Size|Occupancy|SGPRs:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | tr '\n' ';'; echo
cp /tmp/pk_one_$$-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/pk_one.s 2>/dev/null
rm -f /tmp/pk_one_$$*
