#!/bin/bash
# stochastic PC sampling of one micro-benchmark:  tools/pcsample.sh <tag> <python tool> [args...]
# leaves gpurun_out/<tag>_pcs/ (csv files) and a listing of what was produced
TAG=$1; shift
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 \
    --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pcs -o ${TAG} -- python3 "$@" > $R/gpurun_out/${TAG}_pcs.log 2>&1
echo "[pcs] rc=$?"
cd $R
tail -5 gpurun_out/${TAG}_pcs.log
find gpurun_out/${TAG}_pcs -type f | head -20
for f in $(find gpurun_out/${TAG}_pcs -name "*pc_sampling*" | head -3); do echo "== $f"; head -5 $f; wc -l $f; done
