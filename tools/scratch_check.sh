#!/bin/bash
# Kernels whose threads use scratch (private) memory, from the compiler's own resource report: tools/scratch_check.sh
# A by-value struct indexed with a per-lane value, or a spilled array, lives there -- 164 bytes per thread of it were the whole
# time of the track-query kernels in rounds 3-5 (DESIGN lesson 58).  Expected output: the three rare-path resolvers only.
cd "$(dirname "$0")/../refactored_orb_slam2_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt \
    -fno-gpu-rdc --cuda-device-only -c -o /dev/null -Rpass-analysis=kernel-resource-usage $f 2>&1 | \
    sed -n 's/.*remark: *\(Function Name\|ScratchSize \[bytes\/lane\]\|VGPRs Spill\): \([^ ]*\).*/\1 \2/p' | \
    awk -v f=$f '$1=="Function"{k=$3} $1=="ScratchSize"{ if ($3+0 > 0) print f": "substr(k,1,60)" scratch "$3" B/lane" } $1=="VGPRs"{ if ($3+0 > 0) print f": "substr(k,1,60)" VGPR spills "$3 }'
done
