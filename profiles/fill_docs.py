#!/usr/bin/env python3
"""Rewrites the measured-state blocks of DESIGN.md (between <!-- R<n>_STATE --> markers) and README.md
(<!-- R<n>_README_TABLE -->) from the committed artefacts profiles/<tag>_* (and the previous round's, for the
comparison column): nothing in those blocks is typed by hand.

    python profiles/fill_docs.py r3"""
import csv
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r3'
N = int(TAG.lstrip('r'))
PREV = 'r%d' % (N - 1)
STEPS = 10
prev = json.load(open(os.path.join(HERE, '%s_bench_line.json' % PREV)))
prev_rows = list(csv.DictReader(open(os.path.join(HERE, '%s_bench_kernel_stats.csv' % PREV))))
prev_cfgs = {json.loads(l)['config']['workload'].split(':')[0]: json.loads(l) for l in open(os.path.join(HERE, '%s_bench_configs.jsonl' % PREV)) if l.startswith('{')}

line = json.load(open(os.path.join(HERE, '%s_bench_line.json' % TAG)))
rows = list(csv.DictReader(open(os.path.join(HERE, '%s_bench_kernel_stats.csv' % TAG))))
cfgs = [json.loads(l) for l in open(os.path.join(HERE, '%s_bench_configs.jsonl' % TAG)) if l.startswith('{')]
mfma = json.load(open(os.path.join(HERE, '%s_pmc_mfma.json' % TAG)))
rf = line['roofline']


def bucket(name):
    n = name
    if 'DcnColW' in n or 'DcnWLoader' in n: return 'DCN wgrad'
    if 'DcnFwdLoader' in n or 'DcnCols' in n or 'dcn_sample' in n or 'dcnw_fwd' in n: return 'DCN forward'
    if 'dcnq_kernel' in n: return 'DCN backward, one kernel (opt-in)'
    if 'add_kernel' in n: return 'gradient fan-in sums no epilogue takes (cnuda_add)'
    if 'dcn_bwd_data' in n or 'dcn_prep' in n: return 'DCN coord-grad + col2im (one launch)'
    if 'dcn_coord' in n: return 'DCN coord-grad'
    if 'dcn_col2im' in n: return 'DCN col2im'
    if 'igemm_wgrad' in n or 'smallc_wgrad' in n or 'slab_reduce' in n or 'hwgrad' in n: return 'conv wgrad'
    if 'Dgrad' in n: return 'conv dgrad'
    if 'igemm_fwd' in n or 'smallc_fwd' in n or 'hconv_kernel' in n: return 'conv fwd (incl. DCN column-gradient GEMMs, stride-1 smallc dgrad)'
    if n.startswith('bn_') or 'bn_' in n.split('<')[0]: return 'BatchNorm'
    if 'at::native' in n: return 'ATen (loss scalars, arena flush)' if N >= 4 else 'ATen (autograd fan-in sums, loss scalars)'
    return 'rest'


short = lambda n: n.replace('cnuda::(anonymous namespace)::', '').replace('cnuda::', '').replace('void ', '')
tot = sum(float(r['TotalDurationNs']) for r in rows)
bk = {}
for r in rows:
    b = bucket(short(r['Name']))
    bk[b] = bk.get(b, 0.0) + float(r['TotalDurationNs']) / 1e6 / STEPS
launches = sum(int(r['Calls']) for r in rows) // STEPS
launch_note = ''
lp = os.path.join(ROOT, 'profiles', '%s_launches_per_step.json' % TAG)
if os.path.exists(lp):       # the step's own launches: differential of two traces (collect_kernel_stats.sh)
    lj = json.load(open(lp))
    launch_note = (' (the trace\'s total / steps; it includes %d one-time launches of the process -- arena and optimizer-state '
                   'copies -- and the step\'s own count, from the difference of a 10- and a 5-step trace, is %d)'
                   % (round(lj['one_time_launches']), round(lj['launches_per_step'])))
# since round 6 `roofline` is the top kernel over ALL kernels (its own bound); the largest MFMA kernel sits beside it
tm = rf.get('top_mfma_kernel') or rf
busy = None
dom = tm['kernel'].split(' (')[0].replace(' ', '')
for k, v in mfma.items():          # the PMC row of the line's dominant kernel (profiler names carry template defaults)
    if isinstance(v, dict) and 'mfma_util' in v and k.replace(' ', '').startswith(dom.rstrip('>')):
        busy = v['mfma_util']
fam = rf.get('forward_ws128_family')
d, inf, cb, sp = line['decode_latency'], line['inference'], line['cpu_baseline'], line['matrix_mode_split']

prev_launches = sum(int(r['Calls']) for r in prev_rows) // STEPS
state = []
state.append('Measured state at the end of round %d (MI355X, `profiles/%s_*`, written by `profiles/fill_docs.py`): '
             '**%.1f ms/step = %.1f source img/s** (sd %.2f ms over the timed steps; round %d: %.1f ms, %.1f img/s; different '
             'boxes differ by ±1 %%); kernel time under rocprofv3 %.1f ms/step in %d launches (round %d: %d)%s: the step is '
             'GPU-bound. `step_mfma_fraction` = %.3f of the fp32 MFMA peak on SURVEY\'s nominal 6.26 TFLOP, '
             '`step_mfma_fraction_executed` = %.3f on the %.2f TFLOP the GEMM launches execute.'
             % (N, TAG, line['ms_per_step'], line['value'], line.get('ms_per_step_sd', 0.0), N - 1, prev['ms_per_step'],
                prev['value'], tot / 1e6 / STEPS, launches, N - 1, prev_launches, launch_note, line['step_mfma_fraction'],
                line.get('step_mfma_fraction_executed') or 0.0, rf.get('executed_tflop_per_step') or 0.0))
if rf.get('bound') == 'hbm':
    state.append('Top kernel by time (`roofline`, over all kernels since round 6) `%s`: %.1f ms over %d launches, %.0f GB/s of algorithmic '
                 'traffic = `roofline.frac` %.3f of 8 TB/s, %.0f MB of HBM traffic per launch (PMC) against %.0f MB algorithmic.'
                 % (rf['kernel'], rf['kernel_ms_per_step'], rf['launches_per_step'], rf['achieved'], rf['frac'],
                    (rf['traffic'] or 0) / 1e6, rf.get('algorithmic_mb_per_launch', 0.0)))
state.append('%s `%s`: %.1f ms over %d launches, %.1f TFLOP/s = %.3f of the fp32 MFMA peak%s, %.0f MB of HBM traffic per '
             'launch (PMC).' % ('Largest MFMA kernel (`roofline.top_mfma_kernel`)' if rf.get('bound') == 'hbm' else 'Dominant kernel',
                                tm['kernel'], tm['kernel_ms_per_step'], tm['launches_per_step'], tm['achieved'], tm['frac'],
                                (', MFMA pipe busy %.0f %% (PMC)' % (100 * busy)) if busy else '', (tm['traffic'] or 0) / 1e6))
if rf.get('dcn_family_ms') is not None:
    state.append('DCN family (every kernel of the deformable convolutions, `roofline.dcn_family_ms`): %.1f ms per step. HBM bytes of one '
                 'step (PMC, all kernels; `roofline.hbm_bytes_per_step`): %s.'
                 % (rf['dcn_family_ms'], ('%.1f GB = %.1f ms at 6.3 TB/s' % (rf['hbm_bytes_per_step'] / 1e9, rf['hbm_ms_at_6p3TBps']))
                    if rf.get('hbm_bytes_per_step') else 'not collected'))
if fam:
    state.append('The dominant kernel of rounds 1-4, `igemm_fwd_ws_kernel<128, ConvFwdBufLoader>`, now exists as two instances (with / '
                 'without the BatchNorm-statistics tail, §13): together %.1f ms per step at %.1f TFLOP/s = %.3f of the peak.'
                 % (fam['ms_per_step'], fam['tflops'], fam['frac']))
state.append('Per-step kernel time by bucket: ' + ', '.join('%s %.1f ms' % (k, v) for k, v in sorted(bk.items(), key=lambda kv: -kv[1])) + '.')
state.append('Other BASELINE configs at full size on one GPU (`bench.py --config i`, `profiles/%s_bench_configs.jsonl`; the default '
             'run times them too, `other_configs`): ' % TAG +
             '; '.join('%s %.1f ms = %.1f img/s' % (c['config']['workload'].split(':')[0] + ' (' + c['config']['workload'].split('uda=')[1].split(',')[0] + ', %dx%d)' % tuple(c['config']['input'][1:]),
                                                    c['ms_per_step'], c['value']) for c in cfgs) + '.')
state.append('Decode (B=16, K=150): 128×128 C=6 %.0f µs, C=80 %.0f µs; 160×160 C=6 %.0f µs, C=80 %.0f µs. Inference wrapper '
             '(`export.CenterNet`, BatchNorm folded, eval forward + decode, fp32): %.0f img/s (%.2f ms per batch of 16). '
             '%sCPU baseline (`kind: %s`, %d threads, '
             'median of %d 512×512 steps): %.3f img/s.'
             % (d['C6']['us'], d['C80']['us'], d['C6_160']['us'], d['C80_160']['us'], inf['images_per_s'],
                inf['ms_per_batch'], ('Split-operand matrix mode (opt-in, §4a): %.1f ms/step = %.1f img/s. ' % (sp['ms_per_step'], sp['value'])) if sp else '',
                cb['kind'], cb['cores'],
                cb.get('s_per_step_512', {}).get('repeats', 1) if isinstance(cb.get('s_per_step_512'), dict) else 1, cb['value']))
state_txt = '\n'.join(state)

pd, pinf, pcb, psp, prf = prev['decode_latency'], prev['inference'], prev['cpu_baseline'], prev['matrix_mode_split'], prev['roofline']
tab = ['| | round %d | round %d |' % (N - 1, N), '|---|---|---|',
       '| UDA step, DLA-34 + DCNv2 512², 16 + 16 images (the headline, `bench.py`) | %.1f ms = %.1f img/s | **%.1f ms = %.1f img/s** |'
       % (prev['ms_per_step'], prev['value'], line['ms_per_step'], line['value']),
       '| largest MFMA kernel by time (fp32 MFMA, peak 157.3 TFLOP/s; `roofline` until round 5, `roofline.top_mfma_kernel` since) | `%s`: %.1f TF = %.3f | `%s`: %.1f TF = %.3f |'
       % ((prf.get('top_mfma_kernel') or prf)['kernel'].replace('igemm_', ''), (prf.get('top_mfma_kernel') or prf)['achieved'], (prf.get('top_mfma_kernel') or prf)['frac'],
          tm['kernel'].replace('igemm_', ''), tm['achieved'], tm['frac'])]
if rf.get('bound') == 'hbm':
    ph = (prf.get('hbm_kernels') or {}).get(rf['kernel'])
    tab.append('| `roofline` since round 6: the top kernel over all kernels, `%s` (HBM-bound: algorithmic bytes / time against 8 TB/s) | %s | %.1f ms/step, %.0f GB/s = %.3f |'
               % (rf['kernel'], ('%.1f ms/step, %.0f GB/s = %.3f' % (ph['ms_per_step'], ph['gb_per_s'], ph['frac_of_8TBps'])) if ph else '—',
                  rf['kernel_ms_per_step'], rf['achieved'], rf['frac']))
if fam:
    tab.append('| the forward kernel `igemm_fwd_ws_kernel<128, ConvFwdBuf*>` (rounds 1-4\'s dominant one; two instances since round 5) | %.1f TF = %.3f | %.1f TF = %.3f |'
               % (prf['achieved'], prf['frac'], fam['tflops'], fam['frac']))
tab += [
       '| kernel launches per step (whole trace / steps%s) | %d | %d%s |'
       % (('; the step\'s own, by trace difference' if launch_note else ''), prev_launches, launches,
          (' (%d)' % round(lj['launches_per_step'])) if launch_note else '')]
for c in cfgs:
    name = c['config']['workload'].split(':')[0]
    pc = prev_cfgs.get(name)
    tab.append('| %s (%s, %d²) | %s | %.1f ms = %.1f img/s |' % (name, c['config']['workload'].split('uda=')[1].split(',')[0], c['config']['input'][1],
                                                                ('%.1f ms = %.1f img/s' % (pc['ms_per_step'], pc['value'])) if pc else '—', c['ms_per_step'], c['value']))
tab += ['| decode B=16, K=150, 128² maps: C=6 / C=80 | %.0f / %.0f µs | %.0f / %.0f µs |' % (pd['C6']['us'], pd['C80']['us'], d['C6']['us'], d['C80']['us']),
        '| decode 160² maps (cfg5 shape): C=6 / C=80 | %.0f / %.0f µs | %.0f / %.0f µs |' % (pd['C6_160']['us'], pd['C80_160']['us'], d['C6_160']['us'], d['C80_160']['us']),
        '| inference wrapper (eval forward + decode, batch 16) | %.0f img/s | %.0f img/s |' % (pinf['images_per_s'], inf['images_per_s']),
        '| split-operand matrix mode (opt-in; since round 6 only with `--matrix-mode-split`) | %s | %s |'
        % (('%.1f ms' % psp['ms_per_step']) if psp else '—', ('%.1f ms' % sp['ms_per_step']) if sp else 'not in the default run'),
        '| CPU baseline (oracle port, %d host threads; 1 / seconds per 512² step) | %.3f img/s | %.3f img/s |' % (cb['cores'], pcb['value'], cb['value'])]
tab_txt = '\n'.join(tab)


def fill(path, marker, body):
    s = open(path).read()
    pat = re.compile(r'<!-- %s -->.*?(<!-- /%s -->\n?|(?=\n\n))' % (marker, marker), re.S)
    block = '<!-- %s -->\n%s\n<!-- /%s -->\n' % (marker, body, marker)
    assert pat.search(s), marker
    open(path, 'w').write(pat.sub(lambda m: block, s, count=1))


fill(os.path.join(ROOT, 'DESIGN.md'), 'R%d_STATE' % N, state_txt)
fill(os.path.join(ROOT, 'README.md'), 'R%d_README_TABLE' % N, tab_txt)
print(state_txt)
print(tab_txt)
