"""Per-kernel totals of a rocprofv3 --kernel-trace run (the .db it writes): python tools/kernel_times.py <results.db> [n]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(r[0][:90].ljust(90), "%6d calls %10.3f ms total %9.1f us avg %5.1f %%" % (r[1], r[2] / 1e6, r[3] / 1e3, 100 * r[2] / tot))
