#!/bin/bash
# Which build / launch variants show the bf16 step's wrong 16-lane groups (profiles/r06_wrong_lane_group.txt)?  tools/octet_probe2.py counts
# steps whose pixel-octet buffers differ from the control step, 80 trials a line.  Run on a GPU box: bash tools/hazard_matrix.sh
PK="-Xclang -target-feature -Xclang +packed-fp32-ops"      # build.py's FLAGS carry -packed-fp32-ops; the later flag wins
out=gpurun_out/hazard_matrix.txt; : > $out
run() { echo "== $1" >> $out; shift; env "$@" python tools/octet_probe2.py 2>&1 | grep "bad runs found" >> $out; }
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run "shipped build (no packed fp32), per-layer updates on" LAYER_UPDATES=1
run "shipped build, per-layer updates off" LAYER_UPDATES=0
export VC_EXTRA_FLAGS="$PK" VC_EXTRA_FLAGS_ONLY="vc_elementwise.hip"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run "packed fp32 in vc_elementwise only, per-layer updates on" LAYER_UPDATES=1
run "packed fp32 in vc_elementwise only, per-layer updates off" LAYER_UPDATES=0
unset VC_EXTRA_FLAGS_ONLY
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run "packed fp32 everywhere (the round-5 build), per-layer updates on" LAYER_UPDATES=1
cat $out
