#!/bin/bash
TAG=${1:-r03f}
mkdir -p gpurun_out
V="SAMD_MS_VAR=1"
timeout 900 python tools/ms_ab.py --cn minsum --out gpurun_out/ms_cost_$TAG.json base:$V \
  s24:$V,SAMD_MS_CN_SLOPE=24 s28:$V,SAMD_MS_CN_SLOPE=28 s36:$V,SAMD_MS_CN_SLOPE=36 \
  vs:$V,SAMD_MS_VN_SINGLE=1 s28vs:$V,SAMD_MS_CN_SLOPE=28,SAMD_MS_VN_SINGLE=1 s36vs:$V,SAMD_MS_CN_SLOPE=36,SAMD_MS_VN_SINGLE=1 \
  s28vs_o300:$V,SAMD_MS_CN_SLOPE=28,SAMD_MS_VN_SINGLE=1,SAMD_MS_CN_OVH=300 s28vs_v150:$V,SAMD_MS_CN_SLOPE=28,SAMD_MS_VN_SINGLE=1,SAMD_MS_VN_OVH=150 \
  s28vs_v300:$V,SAMD_MS_CN_SLOPE=28,SAMD_MS_VN_SINGLE=1,SAMD_MS_VN_OVH=300 base2:$V 2>&1 | tail -11 | tee gpurun_out/ms_cost_$TAG.txt
