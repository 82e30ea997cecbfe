"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv out.json [substr ...]

hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB and FETCH_SIZE reports half of a wide coalesced
read stream on gfx950 (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[re.sub(r"\s+", " ", r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    f, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    w, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    keep = sys.argv[4:]
    out = {}
    for k in sorted(f, key=lambda k: -f[k] * nf[k]):
        if keep and not any(s in k for s in keep):
            continue
        out[k] = {"launches": nf[k], "FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w.get(k, 0.0), "hbm_bytes_per_launch": (2 * f[k] + w.get(k, 0.0)) * 1024}
    json.dump({"note": __doc__.strip().splitlines()[-2].strip() + " " + __doc__.strip().splitlines()[-1].strip(), "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out.items())[:25]:
        print(f"{v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch x{v['launches']:4d}  {k[:90]}")


main()
