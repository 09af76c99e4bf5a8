#!/bin/bash
# round 5, trip 15: cir_to_ofdm with the paths in passes (fewer registers, more resident waves) against the r05k kernel
TAG=${1:-r05o}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/c2o_ab.py full: no_stores:SAMD_C2O_ABL=2 one_pass_only:SAMD_C2O_ABL=4 no_sincos:SAMD_C2O_ABL=8 no_gather:SAMD_C2O_ABL=16 nothing_but_stores:SAMD_C2O_ABL=28 nothing:SAMD_C2O_ABL=30 > $OUT/c2o_ab.txt 2>&1; cat $OUT/c2o_ab.txt
timeout 600 python -m pytest tests/test_gpu_ofdm.py -q -x -k "cir or channel or c4" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
