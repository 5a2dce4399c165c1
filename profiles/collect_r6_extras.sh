#!/bin/bash
# Round-6 evidence beside profiles/collect_all.sh (one gpurun call): counters of the convolution tiles, the LDS-window DCN
# forward and the decode kernels; the inference wrapper's kernel trace; per-layer tables (convolutions, DCN layers on the maps of
# a 512 x 512 and of a 640 x 640 input, small maps); the non-library kernels of a step; the sigma = 2 px bench line.
#   gpurun --timeout 3000 -- 'bash profiles/collect_r6_extras.sh r6'
TAG=${1:-rX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/profiles/collect_pmc_conv.sh > $O/${TAG}_pmc_conv.txt 2>&1; cp $O/pmc_conv.md $O/${TAG}_pmc_conv.md
bash $R/profiles/collect_pmc_dcnw.sh > $O/${TAG}_pmc_dcnw.txt 2>&1; cp $O/pmc_dcnw.md $O/${TAG}_pmc_dcnw.md
bash $R/profiles/collect_pmc_decode.sh $TAG > $O/${TAG}_pmc_decode.txt 2>&1
bash $R/profiles/collect_infer_stats.sh $TAG > $O/${TAG}_infer_stats.txt 2>&1
python3 $R/profiles/aten_sources.py 2>/dev/null | grep -v Warn > $O/${TAG}_non_library_kernels.txt
python3 $R/profiles/conv_layers.py --iters 3 > $O/${TAG}_conv_layers.txt 2>&1
DCN_LAYER_SHAPES=small_maps python3 $R/profiles/dcn_layer.py --time --iters 3 > $O/${TAG}_dcn_small_map_layers.txt 2>&1
DCN_LAYER_SHAPES=maps640 python3 $R/profiles/dcn_layer.py --time --iters 3 > $O/${TAG}_dcn_maps640_layers.txt 2>&1
python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --dcn-offset-std 2 > $O/${TAG}_bench_sigma2_line.json 2>/dev/null
ls -la $O/${TAG}_* | tail -n 30
