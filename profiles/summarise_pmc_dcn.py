"""Joins the passes of collect_pmc_dcn.sh per kernel (and per layer shape: the two shapes are told apart by the
dispatch order) into gpurun_out/pmc_dcn.{json,md}."""
import collections
import csv
import glob
import json
import os
import sys

root, out_dir = sys.argv[1], sys.argv[2]
KEEP = ('dcn_', 'dcnq', 'dcnw', 'igemm_fwd_kernel<64, DcnFwd', 'igemm_fwd_kernel<128, DcnFwd', 'shortk', 'DcnColW')


def short(n):
    n = n.replace('cnuda::(anonymous namespace)::', '').replace('cnuda::', '').replace('void ', '')
    return n.split('(')[0].replace(', false>', '>').replace(',false>', '>')


per = collections.defaultdict(lambda: collections.defaultdict(list))    # kernel -> counter -> [values per dispatch]
dur = collections.defaultdict(list)
for p in sorted(glob.glob(root + '/*/')):
    tr = glob.glob(p + '*/*kernel_trace.csv') + glob.glob(p + '*kernel_trace.csv')
    cc = glob.glob(p + '*/*counter_collection.csv') + glob.glob(p + '*counter_collection.csv')
    if not tr or not cc:
        continue
    t = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(tr[0]))}
    seen = set()
    for r in csv.DictReader(open(cc[0])):
        k = short(r['Kernel_Name'])
        if not any(s in k for s in KEEP):
            continue
        per[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if (k, r['Dispatch_Id']) not in seen and os.path.basename(p.rstrip('/')) == 'sq1':
            seen.add((k, r['Dispatch_Id']))
            dur[k].append(t.get(r['Dispatch_Id'], 0))
res = {}
for k, c in per.items():
    # the LAST third of the dispatches of a kernel = the measured iteration of the LAST shape is not separable in
    # general; report the mean over all dispatches of the run (both shapes, warm-up included: same kernels, same inputs)
    m = {n: sum(v) / len(v) for n, v in c.items()}
    m['launches_in_run'] = len(next(iter(c.values())))
    m['mean_us'] = sum(dur[k]) / max(1, len(dur[k])) / 1e3
    res[k] = m
json.dump(res, open(os.path.join(out_dir, 'pmc_dcn.json'), 'w'), indent=1)


def g(m, n):
    return m.get(n, float('nan'))


md = ['| kernel | mean µs | wave-cycles waiting (any / LDS issue) | VALU / LDS / VMEM-read / VMEM-write instructions per wave-kilocycle | '
      'LDS bank-conflict cycles ÷ LDS active cycles | TA busy ÷ (256 TAs × active cycles) | TA stalled by TC (addr / data) ÷ TA busy | '
      'L1 accesses per TA wavefront | L1→L2 read requests ÷ L1 accesses | HBM fetch + write MB | MFMA busy ÷ (1024 SIMDs × active cycles) |', '|' + '---|' * 11]
for k, m in sorted(res.items(), key=lambda kv: -kv[1].get('mean_us', 0)):
    wc = g(m, 'SQ_WAVE_CYCLES')
    act = g(m, 'GRBM_GUI_ACTIVE') / 8.0
    md.append('| `%s` | %.0f | %.2f / %.2f | %.1f / %.1f / %.1f / %.1f | %.2f | %.2f | %.2f / %.2f | %.1f | %.2f | %.0f + %.0f | %.2f |' % (
        k[:70], m['mean_us'], g(m, 'SQ_WAIT_ANY') / wc, g(m, 'SQ_WAIT_INST_LDS') / wc,
        1e3 * g(m, 'SQ_INSTS_VALU') / wc, 1e3 * g(m, 'SQ_INSTS_LDS') / wc, 1e3 * g(m, 'SQ_INSTS_VMEM_RD') / wc,
        1e3 * g(m, 'SQ_INSTS_VMEM_WR') / wc,
        g(m, 'SQ_LDS_BANK_CONFLICT') / max(g(m, 'SQ_LDS_IDX_ACTIVE'), 1.0),
        g(m, 'TA_TA_BUSY_sum') / (256.0 * act),
        g(m, 'TA_ADDR_STALLED_BY_TC_CYCLES_sum') / max(g(m, 'TA_TA_BUSY_sum'), 1.0),
        g(m, 'TA_DATA_STALLED_BY_TC_CYCLES_sum') / max(g(m, 'TA_TA_BUSY_sum'), 1.0),
        g(m, 'TCP_TOTAL_CACHE_ACCESSES_sum') / max(g(m, 'TA_TOTAL_WAVEFRONTS_sum'), 1.0),
        g(m, 'TCP_TCC_READ_REQ_sum') / max(g(m, 'TCP_TOTAL_CACHE_ACCESSES_sum'), 1.0),
        g(m, 'FETCH_SIZE') / 1024.0, g(m, 'WRITE_SIZE') / 1024.0,
        g(m, 'SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * act)))
open(os.path.join(out_dir, 'pmc_dcn.md'), 'w').write('\n'.join(md) + '\n')
print('\n'.join(md))
if os.path.exists(root + '/failed.txt'):
    print(open(root + '/failed.txt').read())
