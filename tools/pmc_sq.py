"""Sum rocprofv3 --pmc counters per kernel from a counter_collection.csv (development aid).
usage: python tools/pmc_sq.py gpurun_out/pmc_x/x_counter_collection.csv [kernel substring ...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
subs = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in rows:
    k = r.get("Kernel_Name") or r.get("Kernel Name")
    if subs and not any(s in k for s in subs): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r.get("Dispatch_Id"))
    if key not in seen: seen.add(key); cnt[k] += 1
for k, d in acc.items():
    print(k[:70], "launches", cnt[k])
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    for n, v in sorted(d.items()):
        print(f"   {n:28s} {v:16.0f}" + (f"  {v / wc:6.3f} of WAVE_CYCLES" if wc and n != "SQ_WAVE_CYCLES" else ""))
