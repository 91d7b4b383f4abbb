#!/bin/bash
# A/B of conv_pw builds on one box: tools/bench_pw_libs.sh libA.so libB.so ... (names under image-generation-models_amd/lib; two interleaved rounds)
L=$PWD/image-generation-models_amd/lib
for rnd in 1 2; do for lib in "$@"; do echo "== $lib"; MI_DDPM_LIB=$L/$lib SHORT=1 python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids | cut -c1-150; done; done
