#!/bin/bash
# Register / scratch / LDS use of every kernel of a HIP translation unit (compile only, gfx950): tools/kernel_resources.sh ramx_cp.hip [extra flags]
cd "$(dirname "$0")/../repeatafterme_amd/csrc"
src=${1:-ramx_cp.hip}; shift
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -I. "$@" -Rpass-analysis=kernel-resource-usage -c $src -o /tmp/kres_$$.o 2>&1 |
  sed -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {sc=$NF} /VGPRs Spill:/ {vs=$NF} /SGPRs Spill:/ {ss=$NF} /LDS Size/ {printf "%-52s VGPRs %3s  scratch %4s B  VGPR spills %3s  SGPR spills %3s  LDS %6s\n", name, v, sc, vs, ss, $NF}' | sort
rm -f /tmp/kres_$$.o
