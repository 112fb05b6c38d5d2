"""Timeline summary of a rocprofv3 --kernel-trace CSV of bench.py: for the last complete train steps, wall time per step,
the union of kernel-busy time (any kernel running), the sum of kernel durations (concurrency = sum / union), idle gaps."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''), r.get('Queue_Id', '?'))
            for r in rows)
# a step starts with the first copyBuffer burst after a loss_finalize... simpler: split at every loss_kernel (one per step)
loss = [i for i, e in enumerate(ev) if e[2].startswith('loss_kernel')]
if len(loss) < 4:
    print('not enough steps'); sys.exit(0)
steps = []
for a, b in zip(loss[-4:-1], loss[-3:]):
    steps.append(ev[a:b])
for k, st in enumerate(steps):
    t0, t1 = st[0][0], st[-1][0]
    wall = (t1 - t0) / 1e3
    # union of busy intervals
    iv = sorted((s, e) for s, e, _, _ in st)
    busy = 0; cs, ce = iv[0]
    gaps = []
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs; gaps.append((s - ce, ce)); cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    tot = sum(e - s for s, e, _, _ in st)
    print('step %d: %d kernels, wall %.1f us, busy (union) %.1f us = %.1f %%, sum of durations %.1f us, mean concurrency %.2f'
          % (k, len(st), wall, busy / 1e3, 100 * busy / 1e3 / wall, tot / 1e3, tot / busy))
    gaps.sort(reverse=True)
    print('   idle gaps: %d, total %.1f us; largest: %s' % (len(gaps), sum(g for g, _ in gaps) / 1e3, ['%.1f' % (g / 1e3) for g, _ in gaps[:8]]))
    by = defaultdict(float)
    for s, e, n, q in st:
        by[n.split('<')[0]] += (e - s) / 1e3
    print('   by kernel family (us): ' + ', '.join('%s %.0f' % kv for kv in sorted(by.items(), key=lambda kv: -kv[1])[:10]))
    d = sorted((e - s) / 1e3 for s, e, _, _ in st)
    print('   kernel durations (us): min %.1f, p10 %.1f, median %.1f, p90 %.1f, max %.1f; kernels under 5 us: %d (%.0f us in total)'
          % (d[0], d[len(d) // 10], d[len(d) // 2], d[9 * len(d) // 10], d[-1], sum(1 for x in d if x < 5), sum(x for x in d if x < 5)))

# ---- where the step is NOT multiplying: intervals in which no MFMA kernel (implicit GEMM, filter gradient, gated conv1d + GLU, bf16
# conv) runs, and what runs instead -- the part of the step a faster convolution cannot shorten
MFMA = ('igemm_kernel', 'wgrad_kernel', 'gated_conv1d_glu_kernel', 'conv_bf16_kernel', 'wsconv1d_kernel')
for k, st in enumerate(steps[-1:]):
    t0 = st[0][0]
    mf = sorted((s, e) for s, e, n, _ in st if n.startswith(MFMA))
    merged = []
    for s, e in mf:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    mu = sum(e - s for s, e in merged) / 1e3
    wall = (st[-1][0] - t0) / 1e3
    print('last step: wall %.1f us, some MFMA kernel running %.1f us (%.1f %%), none %.1f us' % (wall, mu, 100 * mu / wall, wall - mu))
    holes = [(merged[i][1], merged[i + 1][0]) for i in range(len(merged) - 1) if merged[i + 1][0] > merged[i][1]]
    holes = [(st[0][0], merged[0][0])] + holes
    holes.sort(key=lambda h: h[0] - h[1])
    print('   largest MFMA-free intervals (start us after the step began, length us, kernels overlapping it):')
    for a, b in holes[:14]:
        inside = sorted({n.split('<')[0] for s, e, n, _ in st if s < b and e > a and not n.startswith(MFMA)})
        print('     +%7.1f  %6.1f us  %s' % ((a - t0) / 1e3, (b - a) / 1e3, ', '.join(inside)[:150]))
    # how long does each MFMA kernel run alone vs beside another MFMA kernel?
    pts = sorted([(s, 1) for s, e in mf] + [(e, -1) for s, e in mf])
    lvl, last, hist = 0, pts[0][0], defaultdict(float)
    for t, d in pts:
        hist[lvl] += (t - last) / 1e3; last = t; lvl += d
    print('   MFMA kernels in flight: ' + ', '.join('%d: %.0f us' % (l, v) for l, v in sorted(hist.items()) if l > 0))
