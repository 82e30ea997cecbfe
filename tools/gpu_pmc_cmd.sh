#!/bin/bash
# one occupancy / matrix-pipe counter pass over an arbitrary python tool:  tools/gpu_pmc_cmd.sh <tag> tools/xyz.py
set -u
TAG=$1; shift
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o o -- python3 $R/$1 > $R/gpurun_out/pmc_$TAG.log 2>&1; rc=$?
echo "[pmc] rc=$rc"; cd $R
O=$(find gpurun_out/pmc_$TAG -name "*counter_collection.csv" | head -1)
python tools/pmc_occ.py $O | head -12
rm -rf gpurun_out/pmc_$TAG
