# Everything the round's profile set is made of, in one gpurun call:
#   gpurun --timeout 2400 -- 'bash profiles/collect_all.sh r2'
# -> gpurun_out/<tag>_bench_line.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_traffic.json, <tag>_pmc_mfma.{json,md}, <tag>_decode_kernel_stats.csv, <tag>_pmc_dcn*,
#    <tag>_bench_configs.jsonl   (copy them to profiles/ and run profiles/make_summary.py <tag>)
TAG=${1:-rX}
R=$GRAFT_REPO_ROOT
# the counter passes first, and their result into the box's own profiles/: bench.py's `roofline.traffic` reads the newest
# profiles/r*_pmc_traffic.json and says whether it was collected from the code the run executes
bash $R/profiles/collect_pmc_traffic.sh > $R/gpurun_out/${TAG}_pmc_traffic.txt 2>&1
cp $R/gpurun_out/pmc_traffic.json $R/gpurun_out/${TAG}_pmc_traffic.json
cp $R/gpurun_out/pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
head -12 $R/gpurun_out/${TAG}_pmc_traffic.txt
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_line.json 2> $R/gpurun_out/${TAG}_bench_line.err
tail -c 300 $R/gpurun_out/${TAG}_bench_line.json
bash $R/profiles/collect_kernel_stats.sh > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
cp $(ls -t $R/gpurun_out/prof_tmp/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_bench_kernel_stats.csv
cp $R/gpurun_out/launches_per_step.json $R/gpurun_out/${TAG}_launches_per_step.json
head -12 $R/gpurun_out/${TAG}_kernel_stats.txt; tail -1 $R/gpurun_out/${TAG}_kernel_stats.txt
bash $R/profiles/collect_pmc_mfma.sh > $R/gpurun_out/${TAG}_pmc_mfma.txt 2>&1
cp $R/gpurun_out/pmc_mfma.json $R/gpurun_out/${TAG}_pmc_mfma.json; cp $R/gpurun_out/pmc_mfma.md $R/gpurun_out/${TAG}_pmc_mfma.md
head -12 $R/gpurun_out/${TAG}_pmc_mfma.txt
bash $R/profiles/collect_config_lines.sh $TAG
bash $R/profiles/collect_decode_stats.sh $TAG > $R/gpurun_out/${TAG}_decode_stats.txt 2>&1
head -6 $R/gpurun_out/${TAG}_decode_stats.txt
bash $R/profiles/collect_pmc_dcn.sh > $R/gpurun_out/${TAG}_pmc_dcn.txt 2>&1
cp $R/gpurun_out/pmc_dcn.json $R/gpurun_out/${TAG}_pmc_dcn.json; cp $R/gpurun_out/pmc_dcn.md $R/gpurun_out/${TAG}_pmc_dcn_raw.md
python3 $R/profiles/dcn_layer.py --time --iters 3 > $R/gpurun_out/${TAG}_dcn_layer_times.txt 2>&1
python3 $R/profiles/dcn_layer.py --time --iters 3 --offsets zero >> $R/gpurun_out/${TAG}_dcn_layer_times.txt 2>&1
