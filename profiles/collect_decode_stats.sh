#!/bin/bash
# rocprofv3 --kernel-trace --stats directly on a decode-only run (140 calls per map shape)
#   gpurun -- 'bash profiles/collect_decode_stats.sh r3'  ->  gpurun_out/<tag>_decode_kernel_stats.csv
TAG=${1:-rX}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_decode
rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/profiles/decode_only.py > $O.log 2>&1
cat $O.log | grep "us per call"
cp $(ls -t $O/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_decode_kernel_stats.csv
head -8 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_decode_kernel_stats.csv | cut -c1-200
