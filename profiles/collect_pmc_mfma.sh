#!/bin/bash
# MFMA-pipe utilisation per kernel of one bench step, from PMC counters (own run: --pmc with --kernel-trace only).
#   util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)      (cycles at the clock the chip actually held)
# usage (GPU box): bash profiles/collect_pmc_mfma.sh  -> gpurun_out/pmc_mfma.json, gpurun_out/pmc_mfma.md
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
rm -rf $O
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 0 > $O.log 2>&1
python3 - $O <<'PY'
import csv, glob, collections, json, sys, os
root = sys.argv[1]
tr = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(glob.glob(root + '/*/*kernel_trace.csv')[0]))}
cc = collections.defaultdict(dict); names = {}
for r in csv.DictReader(open(glob.glob(root + '/*/*counter_collection.csv')[0])):
    cc[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    names[r['Dispatch_Id']] = r['Kernel_Name'].replace('cnuda::(anonymous namespace)::', '').replace('cnuda::', '').split('(')[0].replace('void ', '').replace(', false>', '>')
agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for d, v in cc.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in v or 'GRBM_GUI_ACTIVE' not in v: continue
    a = agg[names[d]]; a[0] += v['SQ_VALU_MFMA_BUSY_CYCLES']; a[1] += v['GRBM_GUI_ACTIVE']; a[2] += tr.get(d, 0); a[3] += 1
rows = sorted(((n, a) for n, a in agg.items() if a[0] > 0), key=lambda x: -x[1][2])
out = {n: {'launches': a[3], 'ms': a[2] / 1e6, 'mfma_busy_cycles': a[0], 'gui_active_cycles': a[1],
           'mfma_util': a[0] / (1024.0 * a[1] / 8.0), 'clock_ghz': (a[1] / 8.0) / a[2] if a[2] else None} for n, a in rows}
go = os.path.dirname(root)
json.dump(out, open(go + '/pmc_mfma.json', 'w'), indent=1)
md = ['| kernel | launches (2 steps) | ms | MFMA pipe busy | active cycles per XCD ÷ kernel time (GHz, indicative) |', '|---|---|---|---|---|']
for n, v in out.items():
    md.append('| `%s` | %d | %.2f | %.1f %% | %.2f |' % (n[:80], v['launches'], v['ms'], 100 * v['mfma_util'], v['clock_ghz']))
open(go + '/pmc_mfma.md', 'w').write('\n'.join(md) + '\n')
print('\n'.join(md[:14]))
PY
