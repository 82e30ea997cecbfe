#!/bin/bash
# rocprofv3 counter passes over the headline bench (separate --pmc passes; kernel-trace only): HBM traffic per launch and occupancy.
#   usage: tools/gpu_pmc.sh <round tag> <build id>      -> gpurun_out/<tag>_pmc_step_traffic.json, <tag>_pmc_slot_attention.json, <tag>_pmc_occupancy.txt
set -u
TAG=${1:-r03}; BUILD=${2:-unknown}
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "o SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -o $name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exploratory > $R/gpurun_out/pmc_$name.log 2>&1
  rc=$?; echo "[pmc] pass $name rc=$rc"
  if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then exit $rc; fi
done
cd $R
F=$(find gpurun_out/pmc_f -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/pmc_w -name "*counter_collection.csv" | head -1); O=$(find gpurun_out/pmc_o -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W gpurun_out/traffic_tmp.json > gpurun_out/${TAG}_pmc_traffic.txt
python tools/pmc_occ.py $O > gpurun_out/${TAG}_pmc_occupancy.txt
python tools/pmc_merge.py gpurun_out/traffic_tmp.json $O $BUILD 128 128 $TAG
cp profiles/${TAG}_pmc_step_traffic.json profiles/${TAG}_pmc_slot_attention.json gpurun_out/
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_o gpurun_out/traffic_tmp.json
head -12 gpurun_out/${TAG}_pmc_traffic.txt; head -14 gpurun_out/${TAG}_pmc_occupancy.txt
