#!/bin/bash
# Everything a round ends with, in one gpurun call: GPU tests, the default bench line, rocprofv3 / PMC summaries of the fp32 B = 8 and
# the bf16 B = 64 step and of the named kernel, the other reported configurations, the data-parallel self-test, the launch self-test.
TAG=${1:-round_end}; export TMPDIR=/tmp; D=gpurun_out/${TAG}; mkdir -p $D
(timeout 1200 python -m pytest tests -m gpu -q > $D/t_all.log 2>&1; echo rc=$? >> $D/t_all.log)
(timeout 700 python bench.py > $D/bench.json 2> $D/bench.err; echo rc=$? >> $D/bench.err)
bash tools/profile_round.sh ${TAG}_fp32 > $D/prof_fp32.log 2>&1
bash tools/profile_round.sh ${TAG}_bf16 "--math bf16 --batch 64" > $D/prof_bf16.log 2>&1
bash tools/pmc_named.sh ${TAG} 4096 bf16 > $D/pmc_named_4096.txt 2>&1
bash tools/pmc_named.sh ${TAG}256 256 fp32,bf16 > $D/pmc_named_256.txt 2>&1
cd $GRAFT_REPO_ROOT
(timeout 200 python bench.py --batch 1 --no-cpu-baseline --no-extra > $D/bench_b1.json 2>> $D/bench.err)
(timeout 200 python bench.py --batch 32 --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $D/bench_b32.json 2>> $D/bench.err)
(timeout 300 python bench.py --global-batch 256 --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $D/bench_b256.json 2>> $D/bench.err)
(timeout 300 python bench.py --features 36 --frames 256 --batch 64 --no-cpu-baseline --no-extra --steps 5 --warmup 2 > $D/bench_36x256_b64.json 2>> $D/bench.err)
(timeout 200 python bench.py --math bf16 --batch 64 --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $D/bench_bf16_b64.json 2>> $D/bench.err)
(timeout 200 python bench.py --dp-selftest --no-cpu-baseline --no-extra > $D/bench_dp.json 2>> $D/bench.err)
(timeout 200 python bench.py --gpus 2 --steps 3 > $D/bench_gpus2.json 2> $D/bench_gpus2.err; echo rc=$? >> $D/bench_gpus2.err)
(timeout 200 python bench.py --gpus 1 --launch-selftest > $D/launch_selftest.json 2>> $D/bench.err)
(timeout 300 python tools/bench_named.py > $D/named.log 2>&1)
(timeout 200 python tools/bench_named.py --batches 256,4096 --maths bf16 --data zero > $D/named_zero.log 2>&1)
for b in 1 8; do for g in "" "--graph"; do timeout 200 python bench.py --batch $b $g --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b $g', round(d['ms_per_step'],3), d['config']['launch'])"; done; done > $D/graph_control.txt 2>&1
tail -3 $D/t_all.log; cut -c1-400 $D/bench.json; cat $D/named.log | grep -v amdgpu; tail -5 $D/prof_bf16.log; tail -4 $D/bench_gpus2.err
