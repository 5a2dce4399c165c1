#!/bin/bash
# A/B builds of the library: bash profiles/microbench/build_variant.sh NAME "<extra hipcc flags>" [file.hip ...]
# recompiles the named sources (default: conv.hip) with the extra flags and links abl/lib_NAME.so with the in-tree objects of
# everything else.  abl/ is git-ignored but travels to the GPU box (profiles/microbench/ab_lib.sh, conv_layers.py: ABL_LIB).
set -e
cd "$(dirname "$0")/../../centernet-uda_amd/csrc"
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-conv.hip}
mkdir -p ../../abl/obj_$NAME
OBJS=""
for f in *.hip; do
  o=build/${f%.hip}.o
  for g in $FILES; do
    if [ "$f" = "$g" ]; then
      o=../../abl/obj_$NAME/${f%.hip}.o
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $FLAGS -c $f -o $o &
    fi
  done
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../abl/lib_$NAME.so $OBJS
echo built abl/lib_$NAME.so
