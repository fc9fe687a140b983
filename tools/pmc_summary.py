"""Sums rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_summary.py <dir with run*/ subdirectories>."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for run in sorted(glob.glob(os.path.join(root, "run*"))):  # run1, run_sq, ...
    if not os.path.isdir(run):
        continue
    files = glob.glob(os.path.join(run, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?")[:60]
                acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
                ndisp[name].add(row.get("Dispatch_Id"))
    for name, c in acc.items():
        nd = max(1, len(ndisp[name]))
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        line = f"{os.path.basename(run)} {name} dispatches {nd}"
        for k in sorted(c):
            line += f" | {k} {c[k] / nd:.4g}"
        if wc:
            line += (f" || per wave-cycle: WAIT_ANY {c.get('SQ_WAIT_ANY', 0) / wc:.2f} WAIT_INST_ANY {c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}"
                     f" ACTIVE {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}")
        if c.get("SQ_BUSY_CYCLES"):
            line += f" MFMA_BUSY/SQ_BUSY {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / c['SQ_BUSY_CYCLES']:.3f}"
        print(line)
