#!/bin/bash
# usage: tools/kernel_resources.sh csrc-file [grep-pattern]: VGPRs / scratch / occupancy per kernel (cross-compiled, no GPU)
f=$1; pat=${2:-.}
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -c $f -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "error|Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/\[-Rpass[^]]*\]//g; s/^.*remark: *//' | paste - - - - \
 | sed 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/' | grep -E "$pat"
