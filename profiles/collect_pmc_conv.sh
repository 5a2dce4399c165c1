#!/bin/bash
# Counter evidence for the convolution kernels (round 4): matrix-pipe busy share, wave wait shares, LDS conflicts per
# kernel over profiles/conv_layers.py.  Separate rocprofv3 passes (--pmc with --kernel-trace only).
# usage (GPU box): bash profiles/collect_pmc_conv.sh [extra env assignments, e.g. CNUDA_HCONV=0]  -> gpurun_out/pmc_conv*.md
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT
TAG=${PMC_TAG:-conv}
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
pass() {   # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/profiles/conv_layers.py --iters 2 > $O/$n.log 2>&1 || echo "pass $n failed" >> $O/failed.txt
}
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
pass sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq4 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_' + os.environ.get('PMC_TAG', 'conv')
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for p in ('sq2', 'sq3', 'sq4'):
    fs = glob.glob(root + '/%s/*/*counter_collection.csv' % p)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        n = r['Kernel_Name'].replace('cnuda::(anonymous namespace)::', '').replace('cnuda::', '').replace('void ', '').split('(')[0]
        a = agg[n][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
    ks = glob.glob(root + '/%s/*/*kernel_trace.csv' % p)
    if ks and p == 'sq2':
        for r in csv.DictReader(open(ks[0])):
            n = r['Kernel_Name'].replace('cnuda::(anonymous namespace)::', '').replace('cnuda::', '').replace('void ', '').split('(')[0]
            dur[n][0] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3; dur[n][1] += 1
def m(n, c):
    a = agg[n].get(c)
    return a[0] / a[1] if a and a[1] else float('nan')
rows = []
for n in agg:
    if not any(t in n for t in ('hconv', 'igemm', 'smallc_fwd', 'smallc_wgrad', 'dgrad_s2_c16_kernel', 'hwgrad')):
        continue
    wc = m(n, 'SQ_WAVE_CYCLES') or float('nan')
    rows.append((dur[n][0], n, dur[n][1], dur[n][0] / max(dur[n][1], 1),
                 m(n, 'SQ_VALU_MFMA_BUSY_CYCLES') / (m(n, 'GRBM_GUI_ACTIVE') / 8 * 1024),
                 m(n, 'SQ_WAIT_ANY') / wc, m(n, 'SQ_WAIT_INST_ANY') / wc, m(n, 'SQ_WAIT_INST_LDS') / wc,
                 m(n, 'SQ_ACTIVE_INST_VALU') / wc, m(n, 'SQ_ACTIVE_INST_LDS') / wc,
                 m(n, 'SQ_LDS_BANK_CONFLICT') / (m(n, 'SQ_LDS_IDX_ACTIVE') or float('nan')),
                 m(n, 'SQ_LDS_IDX_ACTIVE') / (m(n, 'GRBM_GUI_ACTIVE') / 8 * 256),
                 m(n, 'SQ_INSTS_VALU') / max(m(n, 'SQ_INSTS_VALU_MFMA_MOPS_F32'), 1e-9)))
out = ['| kernel | launches | mean us | MFMA busy | wait any | wait inst | wait inst LDS | active VALU | active LDS | LDS conflict / active | LDS active / CU-cycle | VALU insts / MFMA mops |', '|' + '---|' * 12]
for r in sorted(rows, reverse=True):
    out.append('| `%s` | %d | %.1f | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f | %.2f |' % r[1:])
open(root + '.md', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
