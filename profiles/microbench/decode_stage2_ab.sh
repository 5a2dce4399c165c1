#!/bin/bash
# per-kernel times of decode_detection (C = 6, 128 x 128) with 256 / 512 / 1024 threads in stage 2, bands 1 and 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for BANDS in 1 2; do for T in 256 512 1024; do
  O=$R/gpurun_out/prof_decode_t${T}_b$BANDS
  rm -rf $O
  CNUDA_DECODE_BANDS=$BANDS CNUDA_DECODE_STAGE2_THREADS=$T rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/profiles/decode_only.py 6 128 > $O.log 2>&1
  echo "bands $BANDS stage-2 threads $T: $(grep 'us per call' $O.log)"
  head -3 $(ls -t $O/*/*kernel_stats.csv | head -1) | tail -2 | cut -d, -f1,2,4,6 | cut -c1-200
done; done
python3 $R/profiles/decode_only.py
