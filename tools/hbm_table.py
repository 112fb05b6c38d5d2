"""Per-kernel time and HBM-side traffic per train step from a tools/profile_round.sh output set.
usage: python tools/hbm_table.py <pmc_hbm_traffic.txt> <kernel_stats_serial.csv> <profiled steps, 13 for profile_round.sh> > table.txt"""
import csv
import sys

tr = {}
for line in open(sys.argv[1]):
    p = [x.strip() for x in line.rsplit(',', 3)]
    if len(p) == 4 and p[1].replace('.', '').isdigit():
        tr[p[0]] = (float(p[1]), float(p[2]), float(p[3]))          # launches/step, fetch MiB/step (x2 corrected), write MiB/step
tm = {r['Name'].split('(')[0].replace('void ', ''): float(r['TotalDurationNs']) for r in csv.DictReader(open(sys.argv[2]))}
steps = int(sys.argv[3])
rows = sorted(((tm[k] / steps / 1e6, k) + v for k, v in tr.items() if k in tm), reverse=True)
print('time: rocprofv3 --kernel-trace --stats of bench.py --serial (%d steps); traffic: separate --pmc FETCH_SIZE / WRITE_SIZE passes' % steps)
print('(FETCH x2 per MI355X_MICROARCH.md; Infinity-Cache hits are counted -> upper bound on HBM bytes)\n')
print('%-44s %9s %10s %10s %10s %9s' % ('kernel', 'launches', 'ms/step', 'fetch MB', 'write MB', 'TB/s'))
tot_ms = tot_gb = 0.0
for ms, k, lps, fe, wr in rows:
    if ms < 0.02:
        continue
    gb = (fe + wr) * 1.048576 / 1e3
    print('%-44s %9.0f %10.3f %10.1f %10.1f %9.2f' % (k[:44], lps, ms, fe, wr, gb / ms))
    tot_ms += ms; tot_gb += gb
print('\nsum over the listed kernels: %.2f ms/step serialised, %.1f GB/step -> %.2f TB/s average' % (tot_ms, tot_gb, tot_gb / tot_ms))
