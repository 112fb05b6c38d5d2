#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun): default bench line, rocprofv3 kernel stats (serial and
# multi-stream), separate FETCH_SIZE / WRITE_SIZE PMC passes and an MFMA-busy pass.  Outputs land in gpurun_out/<tag>/ and
# are copied to profiles/ by hand (tools/profile_round.sh [tag] ["extra bench flags"]).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-round}            # output directory under gpurun_out/
EXTRA=${2:-}               # extra bench.py flags, e.g. "--math bf16 --batch 64" (BASELINE configs[2])
O=$R/gpurun_out/$TAG
NOCPU=""; [ -n "$EXTRA" ] && NOCPU="--no-cpu-baseline"
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py $EXTRA $NOCPU 2>$O/bench.err | tail -1 > $O/final_bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --serial $EXTRA > $O/serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/streams -o streams -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline $EXTRA > $O/streams.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --serial $EXTRA > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --serial $EXTRA > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o mfma -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --serial $EXTRA > $O/mfma.log 2>&1
F=$(find $O/fetch -name '*counter_collection.csv' | head -1); W=$(find $O/write -name '*counter_collection.csv' | head -1); M=$(find $O/mfma -name '*counter_collection.csv' | head -1)
python $R/tools/summarize_pmc.py $F $W 4 $O/pmc_traffic.json "$(python $R/bench.py --print-key $EXTRA)" > $O/pmc_hbm_traffic.txt 2>&1
python $R/tools/summarize_mfma.py $M > $O/pmc_mfma_busy.txt 2>&1
cp $(find $O/serial -name '*kernel_stats.csv' | head -1) $O/kernel_stats_serial.csv 2>/dev/null
cp $(find $O/streams -name '*kernel_stats.csv' | head -1) $O/kernel_stats_streams.csv 2>/dev/null
# keep only the small summaries
rm -rf $O/serial $O/streams $O/fetch $O/write $O/mfma
find $O -type f | head -40
cut -c1-700 $O/final_bench.json
tail -4 $O/pmc_hbm_traffic.txt; tail -3 $O/pmc_mfma_busy.txt
