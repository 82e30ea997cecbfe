#!/bin/bash
# kernel-trace summary of the Slot-Attention (use_bcdec) bench line
set -u
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/sap -o sap -- python3 $R/bench.py --workload slotattn --steps 4 --warmup 2 --no-cpu-baseline --no-exploratory > $R/gpurun_out/sap.log 2>&1; rc=$?
echo "rc=$rc"; cd $R
DB=$(find gpurun_out/sap -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/slotattn_kernel_stats.csv 4 > gpurun_out/slotattn_stats.txt
rm -rf gpurun_out/sap
head -24 gpurun_out/slotattn_stats.txt | cut -c1-150
