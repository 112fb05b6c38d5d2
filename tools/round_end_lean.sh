#!/bin/bash
# The round's last run when GPU minutes are short: GPU tests, the default bench line, and the bf16 B = 64 step's kernel statistics / PMC passes
# (the fp32 kernels have not changed since the full tools/round_end.sh run).  bash tools/round_end_lean.sh [tag]
TAG=${1:-lean}; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/${TAG}; O=$R/gpurun_out/${TAG}_bf16; mkdir -p $D $O
(timeout 900 python -m pytest tests -m gpu -q > $D/t_all.log 2>&1; echo rc=$? >> $D/t_all.log)
(timeout 500 python bench.py > $D/bench.json 2> $D/bench.err; echo rc=$? >> $D/bench.err)
(timeout 120 python bench.py --math bf16 --batch 64 --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $D/bench_bf16_b64.json 2>> $D/bench.err)
EXTRA="--math bf16 --batch 64"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra --serial $EXTRA > $O/serial.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/streams -o streams -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra $EXTRA > $O/streams.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --serial $EXTRA > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --serial $EXTRA > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o mfma -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --serial $EXTRA > $O/mfma.log 2>&1
F=$(find $O/fetch -name '*counter_collection.csv' | head -1); W=$(find $O/write -name '*counter_collection.csv' | head -1); M=$(find $O/mfma -name '*counter_collection.csv' | head -1)
python $R/tools/summarize_pmc.py $F $W 4 $O/pmc_traffic.json "$(python $R/bench.py --print-key $EXTRA)" > $O/pmc_hbm_traffic.txt 2>&1
python $R/tools/summarize_mfma.py $M > $O/pmc_mfma_busy.txt 2>&1
cp $(find $O/serial -name '*kernel_stats.csv' | head -1) $O/kernel_stats_serial.csv 2>/dev/null
cp $(find $O/streams -name '*kernel_stats.csv' | head -1) $O/kernel_stats_streams.csv 2>/dev/null
rm -rf $O/serial $O/streams $O/fetch $O/write $O/mfma
cd $R; tail -3 $D/t_all.log; cut -c1-300 $D/bench.json; cut -c1-200 $D/bench_bf16_b64.json; tail -4 $O/pmc_hbm_traffic.txt; tail -2 $O/pmc_mfma_busy.txt
