#!/bin/bash
# On the GPU box: build libvcgan_hip.so from the probe-patched kernel source (tools/_probe/vc_conv_glu.hip = the product source with
# tools/glu_timeline.patch applied, generated on the build host), run tools/glu_probe.py, restore the product library.
set -e
cd $GRAFT_REPO_ROOT
cp voice-converter-cyclegan_amd/csrc/vc_conv_glu.hip /tmp/vc_conv_glu.hip.orig
cp voice-converter-cyclegan_amd/csrc/libvcgan_hip.so /tmp/libvcgan_hip.so.orig
cp tools/_probe/vc_conv_glu.hip voice-converter-cyclegan_amd/csrc/vc_conv_glu.hip
cd voice-converter-cyclegan_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -munsafe-fp-atomics -fPIC -DVC_GLU_TIMELINE -c vc_conv_glu.hip -o /tmp/vc_conv_glu_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvcgan_hip.so $(ls build/*.o | grep -v vc_conv_glu.o) /tmp/vc_conv_glu_probe.o -ldl
cd $GRAFT_REPO_ROOT
cp /tmp/vc_conv_glu.hip.orig voice-converter-cyclegan_amd/csrc/vc_conv_glu.hip
touch voice-converter-cyclegan_amd/csrc/libvcgan_hip.so
VC_ALLOW_STALE_LIB=1 python tools/glu_probe.py || true
cp /tmp/libvcgan_hip.so.orig voice-converter-cyclegan_amd/csrc/libvcgan_hip.so
touch voice-converter-cyclegan_amd/csrc/libvcgan_hip.so
