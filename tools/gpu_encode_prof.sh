#!/bin/bash
# kernel timeline of the small-batch encode() path (tools/bench_encode.py)
set -u
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/encp -o encp -- python3 $R/tools/bench_encode.py > $R/gpurun_out/encp.log 2>&1; rc=$?
cat $R/gpurun_out/encp.log | grep encode
if [ "$rc" != 0 ]; then tail -3 $R/gpurun_out/encp.log; exit $rc; fi
cd $R
DB=$(find gpurun_out/encp -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB "" 1000000 > gpurun_out/encp_timeline.txt
rm -rf gpurun_out/encp
python - <<'PY'
lines = open("gpurun_out/encp_timeline.txt").read().splitlines()
idx = [i for i, l in enumerate(lines) if "nchw_to_nhwc8" in l]
# the 10th call = S=64, B=1 steady state
a, b = idx[10], idx[11]
print("\n".join(l[:150] for l in lines[a:b]))
PY
