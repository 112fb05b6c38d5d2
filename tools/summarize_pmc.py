"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into per-kernel HBM traffic per step.
usage: python tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps incl. warm-up> [<out.json> <key>]
FETCH_SIZE is doubled: on gfx950 it reports 1/2 of the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM section);
Infinity-Cache hits are included, so this is an upper bound on HBM bytes."""
import collections
import csv
import sys


def load(fn, ctr):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if r['Counter_Name'] == ctr:
            d[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return d


f, w, steps = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE'), int(sys.argv[3])
print('kernel, launches/step, fetch MB/step (x2 corrected), write MB/step')
tot = 0.0
for k in sorted(f, key=lambda k: -sum(f[k])):
    fe, wr = 2 * sum(f[k]) / steps / 1024, sum(w.get(k, [0])) / steps / 1024
    tot += fe + wr
    print('%s, %.1f, %.1f, %.1f' % (k, len(f[k]) / steps, fe, wr))
print('TOTAL MB/step, %.1f' % tot)
# the two roofline entries of bench.py: `roofline` = convolutions + input gradients (igemm_kernel; in bf16 conv_bf16_kernel<.., 0>
# plus the few launches below its tile size that stay on igemm_kernel), `roofline_wgrad` = filter gradients (wgrad_kernel; in bf16
# conv_bf16_kernel<.., 1> as a GEMM, + wgrad_kernel for o1)
def role(k):
    if k.startswith('conv_bf16_kernel'):
        args = k[k.index('<') + 1:].rstrip('>').replace(' ', '').split(',')      # <tile rows, ring stages, ROLE[, SPEC]>
        return 'wgrad' if len(args) > 2 and args[2] == '1' else 'igemm'
    return 'igemm' if k.startswith('igemm_kernel') else 'wgrad' if (k.startswith('wgrad_kernel') or 'wino_wgrad_kernel' in k) else None


summary = {}
for tag in ('igemm', 'wgrad'):
    ks = [k for k in f if role(k) == tag]
    if not ks:
        continue
    n = sum(len(f[k]) for k in ks)
    b = 2 * sum(sum(f[k]) for k in ks) * 1024 + sum(sum(w.get(k, [0])) for k in ks) * 1024
    summary[tag] = {'launches_per_step': n / steps, 'bytes_per_launch': b / n, 'bytes_per_step': b / steps}
    print('%s: %.1f launches/step, %.2f MB per launch, %.2f GB per step' % (tag, n / steps, b / n / 1e6, b / steps / 1e9))
if len(sys.argv) > 4:       # machine-readable copy: key -> the summary above (bench.py reads profiles/r04_pmc_traffic.json)
    import json
    import os
    out, key = sys.argv[4], sys.argv[5]
    d = json.load(open(out)) if os.path.exists(out) else {}
    d[key] = summary
    json.dump(d, open(out, 'w'), indent=1)
