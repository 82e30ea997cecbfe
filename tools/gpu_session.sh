#!/bin/bash
# One GPU-box session: the gpu test suite, a bench line and a kernel-trace profile of the same command.  A step that times out or is
# killed ends the session (no further GPU step is started after it).
#   usage: tools/gpu_session.sh <tag> [tests|notests] [pytest args...]
set -u
TAG=${1:-x}; WHAT=${2:-tests}; shift 2 || true
R=$PWD
mkdir -p gpurun_out
guard() { rc=$1; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "[session] step killed (rc=$rc): stopping"; exit "$rc"; fi; }
if [ "$WHAT" = tests ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/${TAG}_tests.log 2>&1; rc=$?; echo "[session] tests rc=$rc"; tail -4 gpurun_out/${TAG}_tests.log; guard $rc
fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; rc=$?; echo "[session] bench rc=$rc"; guard $rc
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("[session] bench:", d["value"], "img/s", d["ms_per_step"], "ms/step (median", d.get("ms_per_step_median"), ") step_mfma_frac", d.get("step_mfma_frac"), "conv frac", d["roofline"]["frac"])
    sa = d.get("slot_attention", {})
    if sa: print("[session] slot attention fwd", sa["fwd"], "bwd", sa["bwd"], "in-step fwd", sa["in_step"]["fwd"]["avg_ms"], "bwd", sa["in_step"]["bwd"]["avg_ms"])
except Exception as e:
    print("[session] bench line unreadable:", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exploratory > $R/gpurun_out/${TAG}_prof.log 2>&1; rc=$?; echo "[session] profile rc=$rc"; guard $rc
cd $R
DB=$(find gpurun_out/${TAG}_prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/${TAG}_kernel_stats.csv 6 > gpurun_out/${TAG}_stats.txt
python tools/rocpd_timeline.py $DB "" 1000000 > gpurun_out/${TAG}_timeline_all.txt
# keep one steady-state step of the timeline (the last) and drop the database
python - <<PY
lines = open("gpurun_out/${TAG}_timeline_all.txt").read().splitlines()
idx = [i for i, l in enumerate(lines) if "obs_u8_to_f32" in l]
if len(idx) >= 2:
    open("gpurun_out/${TAG}_timeline_step.txt", "w").write("\n".join(lines[idx[-2]:idx[-1]]) + "\n")
    print("[session] one step:", idx[-1] - idx[-2], "launches")
PY
rm -rf gpurun_out/${TAG}_prof gpurun_out/${TAG}_timeline_all.txt
head -30 gpurun_out/${TAG}_stats.txt
