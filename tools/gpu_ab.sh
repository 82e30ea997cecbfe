#!/bin/bash
# A/B of environment switches on the headline bench: tools/gpu_ab.sh <tag> "VAR=a" "VAR=b" ...   (one bench run per setting; stops after a killed run)
TAG=$1; shift
mkdir -p gpurun_out
for kv in "$@"; do
  name=$(echo "$kv" | tr ' =' '__')
  env $kv timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exploratory > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err; rc=$?
  if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "[ab] $kv killed rc=$rc"; exit $rc; fi
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_${name}.json").read().strip().splitlines()[-1])
    print("[ab] $kv:", d["value"], "img/s", d["ms_per_step"], "ms (median", d.get("ms_per_step_median"), ") frac", d.get("step_mfma_frac"), "conv", d["roofline"]["frac"])
except Exception as e:
    print("[ab] $kv: unreadable", e, open("gpurun_out/${TAG}_${name}.err").read()[-400:])
PY
done
