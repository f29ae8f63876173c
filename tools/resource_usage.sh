#!/bin/bash
# Per-kernel VGPR / AGPR / scratch / occupancy of a HIP source, from hipcc's resource-usage remarks.
# usage: tools/resource_usage.sh annonet_amd/csrc/kernels_mfma.hip [grep-pattern]
src="$1"; pat="${2:-.}"
mkdir -p /tmp/ru
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$(dirname "$src")" \
    -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/ru/out.o 2> /tmp/ru/ru.txt
grep -E "error" -A5 /tmp/ru/ru.txt | head -30
grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" /tmp/ru/ru.txt | sed 's/.*remark: //' | paste - - - - - \
    | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/ \+/ /g; s/Function Name: _ZN3anh12_GLOBAL__N_1[0-9]*//' | grep -E "$pat" | cut -c1-200
