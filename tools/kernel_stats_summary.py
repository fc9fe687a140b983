"""rocprofv3 kernel_stats CSV of `solves` single-structure solves -> the kernels with the largest share, per solve.
    python tools/kernel_stats_summary.py <kernel_stats.csv> <N> [solves = 21] > profiles/..."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = sys.argv[2]
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 21
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats -- python3 tools/single_solves.py {n} {solves - 1}  ({solves} solves of ONE N = {n} structure; "
      "bash tools/r06_final.sh g)")
print(f"# kernel time per solve {tot / solves / 1e6:.3f} ms; the 16 kernels with the largest share:")
for r in rows[:16]:
    name = r["Name"].replace("(anonymous namespace)::", "")[:70]
    print(f"{name:72s} calls/solve {int(r['Calls']) / solves:6.1f}  avg {float(r['AverageNs']) / 1e3:9.1f} us  "
          f"per solve {float(r['TotalDurationNs']) / solves / 1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
