#!/bin/bash
# A/B builds of liborbfe.so for kernel experiments: tools/ab_build.sh <name> "<extra -D flags>" [file.hip | file.cpp ...]
# Recompiles the named .hip files (default: extract_kernels.hip) with the extra flags and links them with the objects of the
# regular build into refactored_orb_slam2_amd/csrc/_ab/liborbfe_<name>.so (git-ignored; travels to the GPU box).
# NOTE: runs `make` first -- called on a `git stash`ed tree it rebuilds liborbfe.so from the stashed sources; run `make` again after
# `git stash pop` (an A/B of "old vs new" was once old vs old that way).
# Select it with ORBFE_AB_LIB=<name> in tools/stage_times.py / bench.py --ab-lib (tools only; the product loads liborbfe.so).
set -e
cd "$(dirname "$0")/../refactored_orb_slam2_amd/csrc"
name=$1; extra=$2; shift 2 || true
files=${@:-extract_kernels.hip}
make -s
mkdir -p _ab/obj_$name
objs=""
for o in _obj/*.o; do
  b=$(basename $o .o)
  if echo " $files " | grep -q " $b.hip "; then
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $extra \
      --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -c $b.hip -o _ab/obj_$name/$b.o
    objs="$objs _ab/obj_$name/$b.o"
  elif echo " $files " | grep -q " $b.cpp "; then   # a host file that shares the macro (e.g. the plan side of a layout switch)
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $extra -c $b.cpp -o _ab/obj_$name/$b.o
    objs="$objs _ab/obj_$name/$b.o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o _ab/liborbfe_$name.so $objs -lz -ldl
echo "built _ab/liborbfe_$name.so"
