#!/bin/bash
# Compact per-kernel resource table for one HIP source (default: the resident Euclidean kernels):
# name-suffix VGPR AGPR spill scratch occupancy
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -c ${1:-vq_amd/csrc/kernels_resident.hip} -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|  VGPRs:|AGPRs:|VGPRs Spill|ScratchSize|Occupancy" | sed 's/^.*remark: *//; s/\[-Rpass.*//; s/Function Name: //; s/ *$//' | paste -d' ' - - - - - - \
 | sed 's/_ZN2vq8k_engineI//; s/EEvNS_10EngineArgsEPKj//; s/VGPRs: /v=/; s/AGPRs: /a=/; s/ScratchSize \[bytes\/lane\]: /scr=/; s/Occupancy \[waves\/SIMD\]: /occ=/; s/VGPRs Spill: /spill=/; s/  */ /g'
