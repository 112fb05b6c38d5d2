#!/bin/bash
# A/B: per-layer update events on / off, packed fp32 on / off in the elementwise translation unit; ms per step, interleaved
out=gpurun_out/ab_events.txt; : > $out
q="--no-cpu-baseline --no-roofline --no-extra"
one() { python bench.py $q "$@" 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"; }
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for rep in 1 2 3; do
  for ev in 1 0; do
    echo "packed build, VC_LAYER_EVENTS=$ev: B=1 $(VC_LAYER_EVENTS=$ev one --batch 1 --steps 100 --warmup 40) B=8 $(VC_LAYER_EVENTS=$ev one --batch 8 --steps 40 --warmup 10) bf16 B=64 $(VC_LAYER_EVENTS=$ev one --batch 64 --math bf16 --steps 20 --warmup 5)" >> $out
  done
done
export VC_EXTRA_FLAGS="-Xclang -target-feature -Xclang -packed-fp32-ops" VC_EXTRA_FLAGS_ONLY="vc_elementwise.hip"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for rep in 1 2 3; do
  for ev in 1 0; do
    echo "no packed fp32 in vc_elementwise, VC_LAYER_EVENTS=$ev: B=1 $(VC_LAYER_EVENTS=$ev one --batch 1 --steps 100 --warmup 40) B=8 $(VC_LAYER_EVENTS=$ev one --batch 8 --steps 40 --warmup 10) bf16 B=64 $(VC_LAYER_EVENTS=$ev one --batch 64 --math bf16 --steps 20 --warmup 5)" >> $out
  done
done
unset VC_EXTRA_FLAGS_ONLY
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for rep in 1 2; do
    echo "no packed fp32 anywhere, VC_LAYER_EVENTS=1: B=1 $(one --batch 1 --steps 100 --warmup 40) B=8 $(one --batch 8 --steps 40 --warmup 10) bf16 B=64 $(one --batch 64 --math bf16 --steps 20 --warmup 5)" >> $out
done
cat $out
