"""SQ counters of one kernel from the three rocprofv3 --pmc passes of tools/measure_r3.sh (gpurun_out/pmc{1,2,3}): per-wave averages, share of
wave cycles waiting / issuing, matrix-pipe busy fraction."""
import collections
import csv
import glob
import sys
pat = sys.argv[1] if len(sys.argv) > 1 else 'conv_wino8p'
tot, nl = {}, 1
for d in ('pmc1', 'pmc2', 'pmc3'):
    files = glob.glob('gpurun_out/%s/**/*counter_collection.csv' % d, recursive=True)
    n = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                if d != 'pmc1' and r['Counter_Name'] in ('SQ_WAVES', 'SQ_WAVE_CYCLES') and r['Counter_Name'] in tot and n[r['Counter_Name']] == 0 and d == 'pmc2':
                    pass
                tot.setdefault((d, r['Counter_Name']), 0.0)
                tot[(d, r['Counter_Name'])] += float(r['Counter_Value'])
                n[r['Counter_Name']] += 1
    nl = max(list(n.values()) + [1])
    for k in list(tot):
        if k[0] == d:
            tot[k] /= nl
v = {}
for (d, k), x in tot.items():
    v.setdefault(k, x)          # first pass that holds a counter wins (SQ_WAVES / SQ_WAVE_CYCLES are in two passes)
w = v.get('SQ_WAVES', 1.0)
print('kernel pattern: %s   launches averaged per pass: %d   waves: %.0f' % (pat, nl, w))
for k in sorted(v):
    print('%-28s %12.4g   per wave %10.1f' % (k, v[k], v[k] / w))
wc = v.get('SQ_WAVE_CYCLES')
if wc and 'SQ_WAIT_ANY' in v:
    print('wait_any %.1f%%  wait_inst %.1f%%  active %.1f%% of wave cycles' % (100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc, 100 * v['SQ_ACTIVE_INST_ANY'] / wc))
if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
    print('MFMA busy / (GUI_ACTIVE/8 * 1024 SIMDs) = %.1f%%' % (100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024)))
if 'SQ_LDS_BANK_CONFLICT' in v:
    print('LDS bank-conflict cycles / LDS active cycles = %.1f%%' % (100 * v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']))
