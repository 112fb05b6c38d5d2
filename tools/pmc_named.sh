#!/bin/bash
# PMC counters of the named kernel (vc_gated_conv1d_glu_fwd[_packed]) at one batch size:  tools/pmc_named.sh <tag> <batch> [maths]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; B=${2:-4096}; M=${3:-fp32,bf16}
O=$R/gpurun_out/pmc_named_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $pass --output-format csv -d $O/$n -o p -- python $R/tools/bench_named.py --batches $B --maths $M --flags ${VC_GLU_FLAGS:-0} > $O/$n.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$O/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
        if 'gated' not in k and 'roof' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('    %-28s %.4e per launch' % (c, v / max(cnt[k][c], 1)))
    if d.get('GRBM_GUI_ACTIVE'): print('    MFMA busy frac %.3f' % ((d['SQ_VALU_MFMA_BUSY_CYCLES'] / cnt[k]['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024.0 / (d['GRBM_GUI_ACTIVE'] / cnt[k]['GRBM_GUI_ACTIVE'] / 8.0)))
PY
rm -rf $O/*/
