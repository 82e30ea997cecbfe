#!/bin/bash
# Per-kernel durations with every kernel running alone (single stream: OCRL_OVERLAP=0, OCRL_DW_SIDE=0): what each kernel costs without
# the contention of the overlapped streams.   usage: tools/gpu_serial_stats.sh <tag>
set -u
TAG=${1:-ser}; R=$PWD
mkdir -p gpurun_out
export OCRL_OVERLAP=0 OCRL_DW_SIDE=0
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exploratory > $R/gpurun_out/${TAG}_prof.log 2>&1; rc=$?; echo "[serial] profile rc=$rc"
if [ "$rc" != 0 ]; then tail -5 $R/gpurun_out/${TAG}_prof.log; exit $rc; fi
cd $R
DB=$(find gpurun_out/${TAG}_prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/${TAG}_kernel_stats.csv 4 > gpurun_out/${TAG}_stats.txt
rm -rf gpurun_out/${TAG}_prof
tail -2 gpurun_out/${TAG}_prof.log | cut -c1-300
head -60 gpurun_out/${TAG}_stats.txt
