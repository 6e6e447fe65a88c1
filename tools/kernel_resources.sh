#!/bin/bash
# Per-kernel register / LDS / occupancy report of one .hip source (hipcc remarks), e.g.
#   tools/kernel_resources.sh l3ac_amd/csrc/kernels/gemm_split.hip
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I"$REPO/include" -I"$REPO/l3ac_amd/csrc" \
    $L3AC_EXTRA_HIPCC_FLAGS -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | grep "remark:" |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {o=$NF} /VGPRs Spill/ {sp=$NF} /LDS Size/ {print name, "vgpr="v, "agpr="a, "scratch="sc, "spill="sp, "occ="o, "lds="$NF}' |
  c++filt | sed 's/(anonymous namespace):://g; s/(.*) vgpr/ vgpr/' | sort -u
