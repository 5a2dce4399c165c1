#!/usr/bin/env python3
"""Writes profiles/<tag>_final_summary.md from the committed artefacts of a round's final measurement
(`profiles/collect_all.sh <tag>`, copied from gpurun_out/ into profiles/):
<tag>_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of collect_kernel_stats.sh, 10 steps),
<tag>_bench_line.json (python bench.py), <tag>_pmc_traffic.json, <tag>_pmc_mfma.md, <tag>_bench_configs.jsonl.

    python profiles/make_summary.py r2          # round 1's files carry the prefix r1_final_ / r1_"""
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r3'
ROUND = TAG.lstrip('r')


def path(stem, *alts):
    for s in (stem,) + alts:
        p = os.path.join(HERE, s)
        if os.path.exists(p):
            return p
    return None


rows = list(csv.DictReader(open(path('%s_bench_kernel_stats.csv' % TAG, '%s_final_bench_kernel_stats.csv' % TAG))))
line = json.load(open(path('%s_bench_line.json' % TAG, '%s_final_bench_line.json' % TAG)))
STEPS = 10
tot = sum(float(r['TotalDurationNs']) for r in rows)
short = lambda n: n.replace('cnuda::(anonymous namespace)::', '').replace('cnuda::', '').replace('void ', '').split('(')[0]
rf = line['roofline']
out = []
out.append('# %s final -- state at the end of round %s\n' % (TAG, ROUND))
out.append('Command (MI355X box): `profiles/collect_kernel_stats.sh` = `rocprofv3 --kernel-trace --stats --output-format csv -- '
           'python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --profile-steps 0`\n'
           '(%d steps in the trace; DLA-34 + 16 DCNv2, 512x512, 16 source + 16 target images, fp32 (matrix mode 0), entropy '
           'minimisation; `--no-extras` leaves the decode / inference / split-mode legs out of the trace). Kernel names carry the '
           'template default `, false` = f32-MFMA variant.\n' % STEPS)
out.append('Total kernel time %.1f ms = **%.1f ms/step** under the profiler; un-profiled wall time %.2f ms/step = %.1f source '
           'images/s (`%s_bench_line.json`). The step is GPU-bound (kernel time = wall time).\n'
           % (tot / 1e6, tot / 1e6 / STEPS, line['ms_per_step'], line['value'], TAG))
if TAG == 'r1':
    out.append('First correct path (`r1a_*`): 388.8 ms/step, 41.3 img/s.\n')
else:
    pl = path('r%d_bench_line.json' % (int(ROUND) - 1), 'r%d_final_bench_line.json' % (int(ROUND) - 1))
    if pl:
        pj = json.load(open(pl))
        out.append('End of round %d (`%s`): %.1f ms/step, %.1f img/s.\n' % (int(ROUND) - 1, os.path.basename(pl), pj['ms_per_step'], pj['value']))
out.append('| kernel | launches/step | ms/step | avg us | % |\n|---|---|---|---|---|')
for r in rows[:48]:
    out.append('| `%s` | %d | %.2f | %.1f | %.1f |' % (short(r['Name'])[:92], int(r['Calls']) // STEPS,
                                                     float(r['TotalDurationNs']) / 1e6 / STEPS, float(r['AverageNs']) / 1e3,
                                                     100 * float(r['TotalDurationNs']) / tot))
out.append('')
out.append('Launches per step: %d.\n' % (sum(int(r['Calls']) for r in rows) // STEPS))
if rf.get('bound') == 'hbm':      # (since round 6 `roofline` is the top kernel over all kernels, against its own bound)
    out.append('Top kernel by time `%s` (`roofline`): avg launch %.4f ms (in-library hipEvents; rocprofv3 above), %.1f MB algorithmic '
               'per launch => %.0f GB/s = %.1f %% of 8 TB/s; HBM traffic %.0f MB per launch from the PMC passes.\n'
               % (rf['kernel'], rf['avg_launch_ms'], rf.get('algorithmic_mb_per_launch', 0.0), rf['achieved'], 100 * rf['frac'],
                  (rf['traffic'] or 0) / 1e6))
    if rf.get('hbm_bytes_per_step'):
        out.append('HBM bytes of one step, all kernels (PMC): %.1f GB = %.1f ms at 6.3 TB/s; DCN family %.1f ms per step.\n'
                   % (rf['hbm_bytes_per_step'] / 1e9, rf['hbm_ms_at_6p3TBps'], rf['dcn_family_ms']))
rf = rf.get('top_mfma_kernel') or rf
out.append('%s `%s`: avg launch %.4f ms (in-library hipEvents; rocprofv3 above), %.2f algorithmic GFLOP per launch '
           '=> %.1f TFLOP/s = %.1f %% of the %.1f TFLOP/s fp32 MFMA peak; HBM traffic %.0f MB per launch from the PMC passes '
           '(`%s_pmc_traffic.json`, `profiles/collect_pmc_traffic.sh`: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs, KB per '
           'launch, FETCH uncorrected).\n' % ('Largest MFMA kernel' if line['roofline'].get('bound') == 'hbm' else 'Dominant kernel',
                                              rf['kernel'], rf['avg_launch_ms'], rf['algorithmic_gflop_per_launch'], rf['achieved'],
                                              100 * rf['frac'], rf['peak'], (rf['traffic'] or 0) / 1e6, TAG))
rf = line['roofline']
hb = rf.get('hbm_kernels') or {}
if hb:
    out.append('HBM-streaming kernels of the DCN backward (algorithmic bytes / time): ' +
               ', '.join('`%s` %.2f ms/step at %.0f GB/s' % (k, v['ms_per_step'], v['gb_per_s']) for k, v in hb.items()) + '.\n')
d = line.get('decode_latency')
if d:
    parts = []
    for k, v in d.items():
        if isinstance(v, dict) and 'us' in v:
            parts.append('%s %.1f us' % (k, v['us']) + (' (CPU oracle %.0f us)' % v['cpu_port_us'] if v.get('cpu_port_us') else ''))
    out.append('Decode (B=16, K=150): ' + ', '.join(parts) + '.\n')
i = line.get('inference')
if i:
    out.append('Inference wrapper (`export.CenterNet`, BatchNorm folded): %.0f img/s (%.2f ms per batch of 16).\n'
               % (i['images_per_s'], i['ms_per_batch']))
s = line.get('matrix_mode_split')
if s:
    out.append('Split-operand matrix mode (opt-in, DESIGN.md 4a), same step: %.1f ms = %.1f img/s.\n' % (s['ms_per_step'], s['value']))
c = line.get('cpu_baseline')
if c:
    out.append('CPU baseline (`kind: %s`): %.4f img/s on %d host threads (%s).\n' % (c['kind'], c['value'], c['cores'], c['sample'][:160]))
cfgs = path('%s_bench_configs.jsonl' % TAG)
if cfgs:
    out.append('Other BASELINE.json configs (`profiles/collect_config_lines.sh`, one `bench.py --config i` line each):\n')
    out.append('| workload | ms/step | img/s |\n|---|---|---|')
    for ln in open(cfgs):
        ln = ln.strip()
        if not ln.startswith('{'):
            continue
        j = json.loads(ln)
        out.append('| %s | %.2f | %.1f |' % (j['config']['workload'][:120], j['ms_per_step'], j['value']))
    out.append('')
dk = path('%s_decode_kernel_stats.csv' % TAG)
if dk:
    out.append('Decode-only kernel stats (`profiles/collect_decode_stats.sh`: rocprofv3 --kernel-trace --stats on '
               '`profiles/decode_only.py`, 140 calls per map shape: C=6 / C=80 at 128x128 and 160x160 mixed):\n')
    out.append('| kernel | calls | avg us | min us | max us |\n|---|---|---|---|---|')
    for r in csv.DictReader(open(dk)):
        out.append('| `%s` | %s | %.1f | %.1f | %.1f |' % (short(r['Name'])[:60], r['Calls'], float(r['AverageNs']) / 1e3,
                                                         float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
    out.append('')
dp = path('%s_pmc_dcn_raw.md' % TAG)
if dp:
    out.append('Counters for the DCN kernels (`profiles/collect_pmc_dcn.sh`; commentary: `%s_pmc_dcn.md`):\n' % TAG)
    out.append(open(dp).read())
m = path('%s_pmc_mfma.md' % TAG)
if m:
    out.append('MFMA-pipe utilisation per kernel from PMC (`profiles/collect_pmc_mfma.sh`: `--pmc SQ_VALU_MFMA_BUSY_CYCLES '
               'GRBM_GUI_ACTIVE` over two steps; busy cycles / (1024 SIMDs x active cycles per XCD), i.e. against the clock the '
               'chip actually held):\n')
    out.append(open(m).read())
open(os.path.join(HERE, '%s_final_summary.md' % TAG), 'w').write('\n'.join(out))
print('\n'.join(out[:8]))
