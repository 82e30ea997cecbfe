"""Writes the two PMC profile files bench.py reads from this round's rocprofv3 passes over `bench.py --steps 3 --warmup 1 --no-cpu-baseline`:
  profiles/<tag>_pmc_step_traffic.json    per-kernel HBM bytes per launch (tools/pmc_traffic.py output + build / workload keys)
  profiles/<tag>_pmc_slot_attention.json  the slot-attention kernels: HBM bytes per launch + waves per SIMD + matrix-pipe busy fraction
usage: python tools/pmc_merge.py traffic.json o_counter_collection.csv <build> <batch> <obs_size> [round tag, default r03]"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
traffic = json.load(open(sys.argv[1]))
build, batch, obs = sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
TAG = sys.argv[6] if len(sys.argv) > 6 else "r03"
traffic.update(batch=batch, obs_size=obs, build=build,
               command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (two passes)")
json.dump(traffic, open(os.path.join(ROOT, "profiles", TAG + "_pmc_step_traffic.json"), "w"), indent=1)

acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[2])):
    acc[re.sub(r"\s+", " ", r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, v in traffic["kernels"].items():
    if "sa_s" not in k:
        continue
    d = acc[k]
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0
    key = re.sub(r"\(.*", "", k)
    out[key] = dict(hbm_bytes_per_launch=v["hbm_bytes_per_launch"], launches=v["launches"],
                    waves_per_simd=round(4 * d["SQ_WAVE_CYCLES"] / (cyc * 1024), 2), mfma_busy=round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 2))
json.dump(dict(note="rocprofv3 PMC passes over bench.py (B=%d, %dx%d, 6 slots): HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB (tools/pmc_traffic.py); "
                    "waves per SIMD and matrix-pipe busy fraction from SQ_WAVE_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE (tools/pmc_occ.py)" % (batch, obs, obs),
               build=build, batch=batch, obs_size=obs, kernels=out), open(os.path.join(ROOT, "profiles", TAG + "_pmc_slot_attention.json"), "w"), indent=1)
print("wrote", len(traffic["kernels"]), "kernels;", len(out), "slot-attention kernels")
