"""Per-queue kernel breakdown of a rocprofv3 results DB: python tools/prof_streams.py <db> <steps> [queue_id]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]); q = int(sys.argv[3]) if len(sys.argv) > 3 else 1
c = db.cursor()
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels where queue_id=? group by name order by 3 desc", (q,)).fetchall()
tot = sum(r[2] for r in rows)
print(f"queue {q}: {tot/1e3/steps:.2f} ms/step kernel time")
for n, cnt, t, a in rows[:45]:
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')[:64]
    print(f"{t/1e3/steps:8.2f} ms/step {cnt/steps:7.1f}/step {a:8.1f} us  {n}")
