#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_libs.sh "<bench flags>" [repetitions]; expects csrc/libvcgan_hip_old.so beside the product .so
export TMPDIR=/tmp; L=voice-converter-cyclegan_amd/csrc; cp $L/libvcgan_hip.so /tmp/new.so; cp $L/libvcgan_hip_old.so /tmp/old.so
run() { python bench.py --no-cpu-baseline --no-extra $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', '$1', round(d['ms_per_step'],3), round(d['roofline']['frac'],3), round(d['roofline_wgrad']['frac'],3))"; }
for rep in $(seq 1 ${2:-2}); do for v in new old; do cp /tmp/$v.so $L/libvcgan_hip.so; run "$1" $v; done; done
cp /tmp/new.so $L/libvcgan_hip.so
