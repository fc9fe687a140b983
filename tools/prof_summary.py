"""Per-kernel summary of a rocprofv3 rocpd database (kernel name, calls, total ms, mean us)."""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(grid_x), max(grid_x) "
                   "from kernels group by name order by 3 desc limit 30").fetchall()
for r in rows:
    print(f"{r[2] / div:9.1f} ms {int(r[1] / div):7d} calls {r[3]:9.1f} us  grid_x {r[4]}..{r[5]}  {r[0][:100]}")
if len(sys.argv) > 3:
    # durations of one kernel in launch order
    rows = cur.execute(f"select start, end, grid_x, grid_y, grid_z from kernels where name like '%{sys.argv[3]}%' order by start").fetchall()
    d = np.array([(r[1] - r[0]) / 1e3 for r in rows])
    print(sys.argv[3], "n", len(d), "pct 1/10/50/90/99", np.percentile(d, [1, 10, 50, 90, 99]).round(1))
    for r in rows[:: max(1, len(rows) // 40)]:
        print(f"   grid {r[2]}x{r[3]}x{r[4]}  {(r[1] - r[0]) / 1e3:9.1f} us")
