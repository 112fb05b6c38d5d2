#!/bin/bash
# copies the summaries of a tools/round_end.sh run (gpurun_out/<tag>, <tag>_fp32, <tag>_bf16) into profiles/ under this round's names
TAG=${1:-r5z}; R=${2:-r05}; P=profiles; Z=gpurun_out/$TAG
cp $Z/bench.json $P/${R}_bench_b8.json
for n in b1 b32 b256 36x256_b64 bf16_b64; do cp $Z/bench_$n.json $P/${R}_bench_$n.json; done
cp $Z/bench_dp.json $P/${R}_bench_dp_selftest.json; cp $Z/launch_selftest.json $P/${R}_launch_selftest.json
grep "bench.py" $Z/bench_gpus2.err | grep -v "^\[W\|FAILED" > $P/${R}_bench_gpus2_on_one_gpu.txt
echo "(python bench.py --gpus 2 --steps 3 on a ONE-GPU box: the process re-launches itself through torch.distributed.run, both ranks stop at device selection, exit code 1 from the launcher)" >> $P/${R}_bench_gpus2_on_one_gpu.txt
for m in fp32:b8 bf16:bf16_b64; do a=${m%%:*}; b=${m##*:}
  cp gpurun_out/${TAG}_$a/kernel_stats_serial.csv $P/${R}_kernel_stats_${b}_serial.csv; cp gpurun_out/${TAG}_$a/kernel_stats_streams.csv $P/${R}_kernel_stats_${b}_streams.csv
  cp gpurun_out/${TAG}_$a/pmc_hbm_traffic.txt $P/${R}_pmc_hbm_traffic_$b.txt; cp gpurun_out/${TAG}_$a/pmc_mfma_busy.txt $P/${R}_pmc_mfma_busy_$b.txt
  python tools/hbm_table.py gpurun_out/${TAG}_$a/pmc_hbm_traffic.txt gpurun_out/${TAG}_$a/kernel_stats_serial.csv 13 > $P/${R}_hbm_gbs_per_kernel_$b.txt; done
python - <<PY
import json
a=json.load(open('gpurun_out/${TAG}_fp32/pmc_traffic.json')); a.update(json.load(open('gpurun_out/${TAG}_bf16/pmc_traffic.json')))
json.dump(a, open('profiles/${R}_pmc_traffic.json','w'), indent=1)
PY
echo "copied $Z -> $P/${R}_*"
