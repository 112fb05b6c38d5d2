"""Per-kernel MFMA-pipe utilisation from one rocprofv3 --pmc pass over bench.py (counters SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE in the same pass).  busy fraction = MFMA_BUSY / 1024 SIMDs / (GUI_ACTIVE / 8 XCDs): SQ_VALU_MFMA_BUSY_CYCLES
sums busy cycles over all SIMDs, GRBM_GUI_ACTIVE sums active cycles over the 8 XCDs (MI355X_MICROARCH.md, counter notes).
usage: python tools/summarize_mfma.py <counter_collection.csv>"""
import collections
import csv
import sys

busy, act, n = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
        busy[k] += float(r['Counter_Value']); n[k] += 1
    elif r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        act[k] += float(r['Counter_Value'])
print('kernel, launches, MFMA busy cycles per launch, GUI_ACTIVE per launch, MFMA pipe busy fraction of kernel time')
tb = ta = 0.0
for k in sorted(busy, key=lambda k: -busy[k]):
    if busy[k] == 0:
        continue
    f = busy[k] / 1024.0 / (act[k] / 8.0)
    print('%s, %d, %.3e, %.3e, %.3f' % (k, n[k], busy[k] / n[k], act[k] / n[k], f))
    tb += busy[k]; ta += act[k]
print('all MFMA kernels, -, -, -, %.3f' % (tb / 1024.0 / (ta / 8.0)))
