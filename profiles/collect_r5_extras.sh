#!/bin/bash
# Round-5 evidence beside profiles/collect_all.sh (one gpurun call): counters of the convolution tiles, the DCN kernels and
# the LDS-window forward; the inference wrapper's kernel trace; per-layer tables (incl. the 16- / 32-channel early layers);
# the data-gradient walk against its window margin and the offset scale; one-process A/Bs (epilogue statistics, margin).
#   gpurun --timeout 3000 -- 'bash profiles/collect_all.sh r5; bash profiles/collect_r5_extras.sh r5'
TAG=${1:-rX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/profiles/collect_pmc_conv.sh > $O/${TAG}_pmc_conv.txt 2>&1; cp $O/pmc_conv.md $O/${TAG}_pmc_conv.md
bash $R/profiles/collect_pmc_dcnw.sh > $O/${TAG}_pmc_dcnw.txt 2>&1; cp $O/pmc_dcnw.md $O/${TAG}_pmc_dcnw.md
bash $R/profiles/collect_infer_stats.sh $TAG > $O/${TAG}_infer_stats.txt 2>&1
python3 $R/profiles/aten_sources.py 2>/dev/null | grep -v Warn > $O/${TAG}_non_library_kernels.txt
python3 $R/profiles/conv_layers.py --iters 3 > $O/${TAG}_conv_layers.txt 2>&1
{ for off in small sigma1 sigma1.4 sigma2; do for mg in 2 3 4; do
    echo "== offsets=$off margin=$mg"; timeout 120 python3 $R/profiles/dcn_layer.py --offsets $off --margin $mg --iters 3 --time 2>&1 | grep -E "B=|dcn_bwd_data|dcnw_fwd"
  done; done; } > $O/${TAG}_dcn_margin_sweep.txt
python3 $R/profiles/microbench/ab_bn_stats.py 2>/dev/null > $O/${TAG}_ab_bn_epilogue_stats.txt
python3 $R/profiles/microbench/ab_margin.py 2>/dev/null > $O/${TAG}_ab_scatter_margin.txt
python3 $R/profiles/microbench/dcn_offset_stats.py 3 2>/dev/null > $O/${TAG}_dcn_offset_regime.txt
python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --dcn-offset-std 2 > $O/${TAG}_bench_sigma2_line.json 2>/dev/null
DCN_LAYER_SHAPES=small_maps python3 $R/profiles/dcn_layer.py --time --iters 3 > $O/${TAG}_dcn_small_map_layers.txt 2>&1
