#!/bin/bash
# rocprofv3 --kernel-trace --stats on the inference wrapper alone (13 calls on a batch of 16 x 512 x 512)
#   gpurun -- 'bash profiles/collect_infer_stats.sh r3'  ->  gpurun_out/<tag>_infer_kernel_stats.csv
TAG=${1:-rX}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_infer
rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/profiles/infer_only.py > $O.log 2>&1
grep "inference:" $O.log
cp $(ls -t $O/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_infer_kernel_stats.csv
head -40 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_infer_kernel_stats.csv | cut -c1-160
