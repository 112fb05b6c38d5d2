#!/bin/bash
# Everything a round ends with, in one gpurun call: GPU tests, the default bench line, rocprofv3 / PMC summaries of the fp32 B = 8 and
# the bf16 B = 64 step and of the named kernel, the other reported configurations, the data-parallel self-test.
TAG=${1:-round_end}; export TMPDIR=/tmp; D=gpurun_out/${TAG}; mkdir -p $D
(timeout 1200 python -m pytest tests -m gpu -q > $D/t_all.log 2>&1; echo rc=$? >> $D/t_all.log)
(timeout 600 python bench.py > $D/bench.json 2> $D/bench.err; echo rc=$? >> $D/bench.err)
bash tools/profile_round.sh ${TAG}_fp32 > $D/prof_fp32.log 2>&1
bash tools/profile_round.sh ${TAG}_bf16 "--math bf16 --batch 64" > $D/prof_bf16.log 2>&1
bash tools/pmc_named.sh ${TAG} 4096 > $D/pmc_named_4096.txt 2>&1
bash tools/pmc_named.sh ${TAG}256 256 > $D/pmc_named_256.txt 2>&1
cd $GRAFT_REPO_ROOT
(timeout 200 python bench.py --batch 1 --no-cpu-baseline --no-extra > $D/bench_b1.json 2>> $D/bench.err)
(timeout 200 python bench.py --batch 32 --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $D/bench_b32.json 2>> $D/bench.err)
(timeout 300 python bench.py --features 36 --frames 256 --batch 64 --no-cpu-baseline --no-extra --steps 5 --warmup 2 > $D/bench_36x256_b64.json 2>> $D/bench.err)
(timeout 200 python bench.py --dp-selftest --no-cpu-baseline --no-extra > $D/bench_dp.json 2>> $D/bench.err)
(timeout 200 python tools/bench_named.py > $D/named.log 2>&1)
tail -3 $D/t_all.log; cut -c1-400 $D/bench.json; cat $D/named.log; tail -5 $D/prof_bf16.log
