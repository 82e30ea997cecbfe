#!/bin/bash
# kernel-trace summary of the IODINE bench line
set -u
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/iop -o iop -- python3 $R/bench.py --workload iodine --obs-size 64 --num-slots 7 --steps 4 --warmup 2 --no-cpu-baseline --no-exploratory > $R/gpurun_out/iop.log 2>&1; rc=$?
echo "rc=$rc"; cd $R
DB=$(find gpurun_out/iop -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/iodine_kernel_stats.csv 4 > gpurun_out/iodine_stats.txt
rm -rf gpurun_out/iop
head -40 gpurun_out/iodine_stats.txt | cut -c1-150
