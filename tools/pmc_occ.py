"""Per-kernel wave occupancy / matrix-pipe view from one rocprofv3 --pmc pass (development aid).
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv ...
usage: python tools/pmc_occ.py x_counter_collection.csv
avg waves/SIMD = 4*SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024);  mfma busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024)"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r.get("Kernel_Name") or r.get("Kernel Name")][r["Counter_Name"]] += float(r["Counter_Value"])
rows = []
for k, d in acc.items():
    cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0: continue
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    rows.append((cyc, k, 4 * wc / (cyc * 1024), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), d.get("SQ_WAIT_INST_ANY", 0) / max(wc, 1), d.get("SQ_WAIT_ANY", 0) / max(wc, 1),
                 d.get("SQ_ACTIVE_INST_ANY", 0) / max(wc, 1)))
tot = sum(r[0] for r in rows)
print(f"{'kernel':62s} {'%cyc':>6s} {'waves/SIMD':>10s} {'mfma':>6s} {'waitinst':>8s} {'waitany':>8s} {'active':>7s}")
for cyc, k, occ, mf, wi, wa, ac in sorted(rows, reverse=True)[:40]:
    print(f"{k[:62]:62s} {100 * cyc / tot:6.2f} {occ:10.2f} {mf:6.2f} {wi:8.2f} {wa:8.2f} {ac:7.2f}")
