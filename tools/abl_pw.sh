#!/bin/bash
# ablations of conv_pw (needs lib/libmi_ddpm_abl.so built with -DMI_PW_ABL_BUILD): MI_PW_ABL bits 1 no fragment DMA, 2 no activation
# DMA, 4 no stores, 8 one workgroup per CU
export MI_DDPM_LIB=$PWD/image-generation-models_amd/lib/libmi_ddpm_abl.so
for a in ${ABLS:-0 1 2 4 7 8}; do echo "== MI_PW_ABL=$a"; MI_PW_ABL=$a python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids | cut -c1-75; done
