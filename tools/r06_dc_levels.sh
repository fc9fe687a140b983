#!/bin/bash
# round 6: duration of the D&C merge GEMM of every level (one bench step under rocprofv3 --kernel-trace)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_dc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o r --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# kernels between the first k_dc_leaves and k_dc_unscale of the LAST solve
names = [r["Kernel_Name"] for r in rows]
last_leaves = max(i for i, n in enumerate(names) if "k_dc_leaves" in n)
end = min(i for i, n in enumerate(names) if i > last_leaves and "k_dc_unscale" in n)
lev = -1
acc = {}
for r in rows[last_leaves:end]:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if "k_dc_zero_offdiag" in n:
        lev += 1
    key = "gemm" if "k_gemm" in n else n.split("(")[0].split("::")[-1][:24]
    acc.setdefault(lev, {}).setdefault(key, 0.0)
    acc[lev][key] += d
tot_g = tot_o = 0.0
for l in sorted(acc):
    if l < 0: continue
    g = acc[l].get("gemm", 0.0)
    o = sum(v for k, v in acc[l].items() if k != "gemm")
    tot_g += g; tot_o += o
    print(f"level {l}: merge GEMM {g:8.2f} ms, other kernels {o:7.2f} ms  " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(acc[l].items()) if k != "gemm"))
print(f"all levels: GEMM {tot_g:.1f} ms, other {tot_o:.1f} ms  (kernel durations of one solve, overlap with the second stream not subtracted)")
PY
