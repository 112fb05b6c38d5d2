#!/bin/bash
# PMC counters of ONE layer's launches under rocprofv3 (run on the GPU box):  tools/pmc_layer.sh <tag> <layer substring> [env assignments...]
# e.g. tools/pmc_layer.sh d3_skip d_d3_2B ; tools/pmc_layer.sh d3_noskip d_d3_2B VC_NO_ROWSKIP=1
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; LAYER=$2; shift 2
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  env "$@" timeout 200 rocprofv3 --pmc $pass --output-format csv -d $O/$n -o p -- python $R/tools/tune_igemm.py --only $LAYER --reps 3 > $O/$n.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob('$O/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
        if 'igemm' not in k and 'wgrad' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('GRBM_GUI_ACTIVE', 'SQ_WAIT_ANY', 'SQ_INSTS_VALU', 'TCC_HIT_sum', 'FETCH_SIZE'): cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('    %-28s %.4e' % (c, v))
    if d.get('GRBM_GUI_ACTIVE'): print('    MFMA busy frac %.3f' % (d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (d['GRBM_GUI_ACTIVE'] / 8.0)))
    if d.get('TCC_HIT_sum'): print('    L2 hit rate %.3f' % (d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum'])))
PY
rm -rf $O/*/
