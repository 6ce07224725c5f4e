#!/bin/bash
# Register / scratch / LDS use of every kernel (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).
# The packed solve kernels must stay at ScratchSize 0: their register allocation sits at the edge (256 VGPR + ~238 AGPR).
# The lean ones (btrapz_lean.hip) must stay at Occupancy 2 (256 registers, no AGPRs, 20 KB of LDS).
cd "$(dirname "$0")/../spectral_amd/csrc" || exit 1
for f in btrapz_kernels.hip btrapz_lean.hip btrapz_lean_warm.hip corridor_kernels.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden \
    -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 |
    grep -E "Function Name|TotalSGPRs|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | paste - - - - - - - |
    sed -E 's/remark: [^ ]* //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[^ ]*\.hip:[0-9]+:[0-9]+://g; s/ +/ /g'
done
