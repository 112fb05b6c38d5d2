#!/bin/bash
# kernel-trace timeline of a few overlapped train steps: per-step busy time, concurrency and gaps (tools/analyze_trace.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-trace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline ${2:-} > $O/kt.log 2>&1
F=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python $R/tools/analyze_trace.py $F > $O/trace_summary.txt 2>&1
rm -rf $O/kt
cat $O/trace_summary.txt | head -60
