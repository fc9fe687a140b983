"""Idle time of the GPU inside the benchmark step, from a rocprofv3 --kernel-trace CSV:
    python tools/trace_gaps.py <..._kernel_trace.csv> [marker kernel substring = k_bt2_apply]
The marker kernel runs once per step; the window goes from its 2nd to its 4th start (= two full steps of the timed
region of `bench.py --steps 3 --warmup 1`).  Kernel intervals of all streams are merged; what is left uncovered is the
gap sum (time in which NO kernel was executing)."""
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_bt2_apply"
iv, marks = [], []
with open(path, newline="") as f:
    rd = csv.DictReader(f)
    for r in rd:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        iv.append((s, e))
        if marker in r["Kernel_Name"]:
            marks.append(s)
marks.sort()
if len(marks) < 4:
    sys.exit(f"only {len(marks)} launches of {marker} in the trace")
w0, w1 = marks[1], marks[3]
iv = sorted((max(s, w0), min(e, w1)) for s, e in iv if e > w0 and s < w1)
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e in iv:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
busy += cur_e - cur_s
total = w1 - w0
gap = total - busy
gaps.sort(reverse=True)
print(f"window: {len(iv)} kernel launches in {total / 1e6:.1f} ms (two steps, {total / 2e6:.1f} ms per step)")
print(f"GPU executing at least one kernel: {busy / 1e6:.1f} ms; gap sum {gap / 1e6:.2f} ms = {100.0 * gap / total:.2f} % of the window")
print(f"gaps: {len(gaps)}; the ten longest (us): " + ", ".join(f"{g / 1e3:.1f}" for g in gaps[:10]))
for lim in (5e3, 20e3, 100e3):
    part = sum(g for g in gaps if g >= lim)
    print(f"  gaps >= {lim / 1e3:.0f} us: {sum(1 for g in gaps if g >= lim)} of them, {part / 1e6:.2f} ms")
