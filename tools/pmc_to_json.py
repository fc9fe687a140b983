"""
The PMC passes of tools/pmc_kernel.sh (gpurun_out/pmc_<tag>/run_*) as the JSON record bench.py reads for roofline.traffic:
  python tools/pmc_to_json.py gpurun_out/pmc_r05_bt2 k_bt2_apply 6000 64 > profiles/r05_bt2_pmc_fetch_write.json
Per-launch means over the dispatches of the kernel; FETCH_SIZE / WRITE_SIZE are reported in KiB, and FETCH_SIZE counts
128-byte requests as 64 bytes on gfx950 (MI355X_MICROARCH.md; calibrated in profiles/r01_bt2_pmc_fetch_write.json), so
read bytes = 2 x FETCH_SIZE x 1024.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, kernel, n, batch = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
acc = defaultdict(float)
disp = defaultdict(set)
for run in sorted(glob.glob(os.path.join(root, "run*"))):
    if not os.path.isdir(run):
        continue
    for f in glob.glob(os.path.join(run, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel not in row.get("Kernel_Name", ""):
                    continue
                c = row["Counter_Name"]
                acc[c] += float(row["Counter_Value"])
                disp[c].add((os.path.basename(run), row.get("Dispatch_Id")))
mean = {c: acc[c] / max(1, len(disp[c])) for c in acc}
# the Z window of a matrix is read and written once per sweep group; the diamonds' fragments are read once per column chunk
# of the XCD's L2 at best -- counted once, as in rounds 1 - 3
ngroups = (n - 2 + 63) // 64
ndia = sum((n - 1 - 64 * g + 63) // 64 for g in range(ngroups))
z_bytes = 8 * n * sum(min(n, 64 * g + 1 + 64 * ((n - 1 - 64 * g + 63) // 64) + 63) - (64 * g + 1) for g in range(ngroups))
frag_bytes = ndia * 4 * 40 * 64 * 8
alg = batch * (2 * z_bytes + frag_bytes)
out = {
    "command": "bash tools/r06_final.sh b  ->  bash tools/pmc_kernel.sh %s <tag>  (rocprofv3 --pmc, counters only, separate passes: "
               "SQ counters | FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; --kernel-trace --kernel-include-regex; "
               "python3 bench.py --no-cpu-baseline --steps 1 --warmup 0), summed by tools/pmc_to_json.py" % kernel,
    "kernel": kernel, "n": n, "batch": batch,
    "launches_sampled": max([len(v) for v in disp.values()] + [0]),
    "counters_per_launch_mean": {c: mean[c] for c in sorted(mean)},
    "units_and_corrections": "FETCH_SIZE / WRITE_SIZE in KiB; read bytes = 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B), "
                             "WRITE_SIZE exact; Infinity-Cache hits included (memory-side request counters of the L2)",
}
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    rd = 2.0 * mean["FETCH_SIZE"] * 1024.0
    wr = mean["WRITE_SIZE"] * 1024.0
    out.update({"hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr,
                "hbm_bytes_per_launch_corrected": rd + wr, "algorithmic_bytes_per_launch": alg,
                "traffic_over_algorithmic": (rd + wr) / alg if alg else None})
if mean.get("TCC_HIT_sum") is not None and mean.get("TCC_MISS_sum") is not None:
    out["l2_hit_rate"] = mean["TCC_HIT_sum"] / max(1.0, mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
print(json.dumps(out, indent=1))
