"""Top kernels of a rocprofv3 kernel-trace database by total time (all queues; optionally only launches after the first `skip` seconds).
    python tools/prof_top.py <db> [n] [skip_fraction]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 25; frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
t0, t1 = db.execute("select min(start), max(end) from kernels").fetchone()
cut = t0 + (t1 - t0) * frac
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, max(end-start)/1e3 from kernels where start>? group by name order by 3 desc", (cut,)).fetchall()
tot = sum(r[2] for r in rows)
print(f"kernels started in the last {100 * (1 - frac):.0f} % of the trace: {tot / 1e3:.2f} ms of kernel time")
for name, cnt, t, a, mx in rows[:n]:
    print(f"{t / 1e3:8.2f} ms {100 * t / tot:5.1f} %  {cnt:6d} x {a:8.1f} us (max {mx:7.1f})  {name.replace('(anonymous namespace)::', '').replace('void ', '')[:90]}")
