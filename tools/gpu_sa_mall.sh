#!/bin/bash
# per-launch durations of the slot-attention kernels at several batch sizes (does a batch whose inputs fit the 256 MB memory-side cache stream faster?)
set -u
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for b in 16 32 48 64 128; do
  export B=$b
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/sam_$b -o sam -- python3 $R/tools/bench_slot_attention.py > $R/gpurun_out/sam_$b.log 2>&1; rc=$?
  if [ "$rc" != 0 ]; then echo "B=$b rc=$rc"; tail -3 $R/gpurun_out/sam_$b.log; exit $rc; fi
  DB=$(find $R/gpurun_out/sam_$b -name "*.db" | head -1)
  echo "== B=$b: $(grep fwd $R/gpurun_out/sam_$b.log)"
  python $R/tools/rocpd_stats.py $DB /dev/null 1 | grep -E "sa_stream|sa_slot"
  rm -rf $R/gpurun_out/sam_$b
done
