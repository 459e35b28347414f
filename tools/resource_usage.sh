#!/bin/bash
# Print VGPR / scratch / occupancy per kernel for one HIP source (compiler view, gfx950).
f=$1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/ {name=$3} /VGPRs:/ {v=$2} /AGPRs:/ {a=$2} /ScratchSize/ {s=$3} /Occupancy/ {o=$3} /LDS Size/ {printf "%-110s vgpr=%s agpr=%s scratch=%s occ=%s lds=%s\n", substr(name,1,110), v, a, s, o, $4}' \
 | grep -v rocprim
