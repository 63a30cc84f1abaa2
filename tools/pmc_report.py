import csv, collections, sys
pat = sys.argv[1] if len(sys.argv) > 1 else 'conv_fwd'
tot = {}
for d in ('pmc1', 'pmc2'):
    rows = list(csv.DictReader(open('gpurun_out/%s/a_counter_collection.csv' % d)))
    n = collections.Counter()
    for r in rows:
        if pat in r['Kernel_Name']:
            tot[r['Counter_Name']] = tot.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            n[r['Counter_Name']] += 1
    nl = max(n.values()) if n else 1
for k in tot: tot[k] /= nl
w = tot['SQ_WAVES']
print('launches averaged:', nl, ' waves:', w)
for k, v in sorted(tot.items()):
    print('%-28s %12.4g   per wave %10.1f' % (k, v, v / w))
wc = tot['SQ_WAVE_CYCLES']
print('wait_any %.1f%%  wait_inst %.1f%%  active %.1f%% of wave cycles' % (100*tot['SQ_WAIT_ANY']/wc, 100*tot['SQ_WAIT_INST_ANY']/wc, 100*tot['SQ_ACTIVE_INST_ANY']/wc))
print('MFMA busy / (GUI_ACTIVE/8 * 1024 SIMDs) = %.1f%%' % (100 * tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (tot['GRBM_GUI_ACTIVE'] / 8 * 1024)))
