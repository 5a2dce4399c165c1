#!/bin/bash
# per-kernel times of decode_detection with 1 / 2 / 4 row bands per plane (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for BANDS in 1 2 4; do
  O=$R/gpurun_out/prof_decode_b$BANDS
  rm -rf $O
  CNUDA_DECODE_BANDS=$BANDS rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/profiles/decode_only.py 6 128 > $O.log 2>&1
  echo "bands $BANDS: $(grep 'us per call' $O.log)"
  head -4 $(ls -t $O/*/*kernel_stats.csv | head -1) | cut -c1-160
done
