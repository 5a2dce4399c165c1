#!/bin/bash
# Round-4 evidence beside profiles/collect_all.sh (one gpurun call): counters of the convolution tiles and of the DCN
# forward kernels, the opt-in one-kernel DCN backward (counters, layer times, whole-step A/B), the inference wrapper's
# kernel trace, where the remaining non-library kernels of a step come from, and the LDS microbenchmarks.
#   gpurun --timeout 2400 -- 'bash profiles/collect_r4_extras.sh r4'
TAG=${1:-rX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/profiles/collect_pmc_conv.sh > $O/${TAG}_pmc_conv.txt 2>&1; cp $O/pmc_conv.md $O/${TAG}_pmc_conv.md
bash $R/profiles/collect_pmc_dcnw.sh > $O/${TAG}_pmc_dcnw.txt 2>&1; cp $O/pmc_dcnw.md $O/${TAG}_pmc_dcnw.md
( export CNUDA_DCNQ=1; bash $R/profiles/collect_pmc_dcn.sh > $O/${TAG}_pmc_dcnq.txt 2>&1 ); cp $O/pmc_dcn.md $O/${TAG}_pmc_dcnq_raw.md; cp $O/pmc_dcn.json $O/${TAG}_pmc_dcnq.json
{ for off in small zero; do for q in 0 1; do echo "== CNUDA_DCNQ=$q offsets=$off"; CNUDA_DCNQ=$q python3 $R/profiles/dcn_layer.py --time --iters 3 --offsets $off 2>/dev/null | grep -E "^B=|dcn|shortk"; done; done; } > $O/${TAG}_dcnq_layer_times.txt
bash $R/profiles/microbench/ab_dcnq.sh > $O/${TAG}_dcnq_step_ab.txt 2>&1
CNUDA_DCNQ=1 python3 $R/profiles/microbench/dcnq_scaling.py 2>/dev/null | grep "B=" > $O/${TAG}_dcnq_scaling.txt
bash $R/profiles/collect_infer_stats.sh $TAG > $O/${TAG}_infer_stats.txt 2>&1
python3 $R/profiles/aten_sources.py 2>/dev/null | grep -v Warn > $O/${TAG}_non_library_kernels.txt
hipcc -O3 --offload-arch=gfx950 $R/profiles/microbench/lds_atomic_bench.hip -o /tmp/lab 2>/dev/null && /tmp/lab > $O/${TAG}_lds_atomic_bench.txt
hipcc -O3 --offload-arch=gfx950 $R/profiles/microbench/occ.hip -o /tmp/occ 2>/dev/null && /tmp/occ > $O/${TAG}_lds_occupancy.txt
