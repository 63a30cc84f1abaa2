"""ms per step per kernel from a rocprofv3 kernel_stats.csv: python tools/show_stats.py file.csv nsteps [top]"""
import csv, sys
f, n = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows) / n / 1e6
print('%s: total %.2f ms/step over %d kernels' % (f.split('/')[-1], tot, len(rows)))
for r in rows[:top]:
    print('  %7.3f ms/step  %5.1f calls/step  %8.1f us avg  %s' % (float(r['TotalDurationNs']) / n / 1e6, float(r['Calls']) / n, float(r['AverageNs']) / 1e3, r['Name'][:110]))
