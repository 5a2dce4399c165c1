#!/bin/bash
# Counter evidence for the decode kernels (round 3): instruction mix and issue / wait cycles of the two stages, C = 80 on
# 128 x 128 maps (1,280 planes: five rounds of one plane per CU).  Separate rocprofv3 passes (--pmc with --kernel-trace
# only).  usage (GPU box): bash profiles/collect_pmc_decode.sh <tag> -> gpurun_out/<tag>_pmc_decode.{json,md}
TAG=${1:-rX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_decode
rm -rf $O; mkdir -p $O
pass() {   # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/profiles/decode_only.py 80 128 > $O/$n.log 2>&1 || echo "pass $n failed" >> $O/failed.txt
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA
pass sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
TAG_=$TAG python3 - <<'PY'
import collections, csv, glob, json, os
R = os.environ['GRAFT_REPO_ROOT']; root = R + '/gpurun_out/pmc_decode'; tag = os.environ.get('TAG_', 'rX')
per = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for p in sorted(glob.glob(root + '/*/')):
    tr = glob.glob(p + '*/*kernel_trace.csv') + glob.glob(p + '*kernel_trace.csv')
    cc = glob.glob(p + '*/*counter_collection.csv') + glob.glob(p + '*counter_collection.csv')
    if not tr or not cc:
        continue
    t = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(tr[0]))}
    seen = set()
    for r in csv.DictReader(open(cc[0])):
        k = 'plane_topk_kernel (stage 1)' if 'plane_topk' in r['Kernel_Name'] else ('merge_decode_kernel (stage 2)' if 'merge_decode' in r['Kernel_Name'] else None)
        if not k:
            continue
        per[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if os.path.basename(p.rstrip('/')) == 'sq1' and (k, r['Dispatch_Id']) not in seen:
            seen.add((k, r['Dispatch_Id'])); dur[k].append(t.get(r['Dispatch_Id'], 0))
res = {}
for k, c in per.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    m['mean_us_under_counters'] = sum(dur[k]) / max(1, len(dur[k])) / 1e3
    res[k] = m
json.dump(res, open('%s/gpurun_out/%s_pmc_decode.json' % (R, tag), 'w'), indent=1)
md = ['| kernel | us (under counters) | waves | VALU / SALU / LDS / VMEM-rd / VMEM-wr instructions per wave | VALU instructions per wave-kilocycle | '
      'share of wave cycles: issuing VALU / LDS / scalar, waiting for anything | LDS bank-conflict cycles / LDS active cycles | '
      'CU-level: VALU issue cycles per SIMD / kernel cycles |', '|' + '---|' * 8]
for k, m in res.items():
    g = lambda n: m.get(n, float('nan'))
    wc, waves = g('SQ_WAVE_CYCLES'), g('SQ_WAVES')
    act = g('GRBM_GUI_ACTIVE') / 8.0                      # per-XCD active cycles
    md.append('| `%s` | %.1f | %.0f | %.0f / %.0f / %.0f / %.0f / %.0f | %.0f | %.2f / %.2f / %.2f, %.2f | %.2f | %.2f |' % (
        k, m['mean_us_under_counters'], waves, g('SQ_INSTS_VALU') / waves, g('SQ_INSTS_SALU') / waves, g('SQ_INSTS_LDS') / waves,
        g('SQ_INSTS_VMEM_RD') / waves, g('SQ_INSTS_VMEM_WR') / waves, 1e3 * g('SQ_INSTS_VALU') / wc,
        g('SQ_ACTIVE_INST_VALU') / wc, g('SQ_ACTIVE_INST_LDS') / wc, g('SQ_ACTIVE_INST_SCA') / wc, g('SQ_WAIT_ANY') / wc,
        g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1.0),
        4.0 * g('SQ_INSTS_VALU') / (256 * 4) / act))
open('%s/gpurun_out/%s_pmc_decode_raw.md' % (R, tag), 'w').write('\n'.join(md) + '\n')
print('\n'.join(md))
PY
