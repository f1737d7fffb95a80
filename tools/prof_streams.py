"""Per-queue kernel breakdown of a rocprofv3 results DB.

    python tools/prof_streams.py <db> <steps> [queue_id]              every kernel of the process, divided by <steps>
    python tools/prof_streams.py <db> --steady <n> [queue_id]         only the LAST n optimizer steps of the process

--steady trims by timestamp: a step ends with its `adam_flat_kernel` launch, so the window from the end of the (n+1)-th last
launch of that kernel to the end of the last one holds exactly n complete steps -- model construction (ATen casts / copies of
the frozen weights), warm-up allocations and the first-call initialisations are outside it."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
steady = sys.argv[2] == "--steady"
if steady:
    n = int(sys.argv[3]); q = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    ends = [r[0] for r in c.execute("select end from kernels where name like '%adam_flat_kernel%' order by end").fetchall()]
    assert len(ends) > n, f"only {len(ends)} optimizer steps in the trace"
    t0, t1 = ends[-n - 1], ends[-1]
    where, args, steps = "queue_id=? and start>? and start<=?", (q, t0, t1), float(n)
    print(f"steady-state window: the last {n} optimizer steps, {(t1 - t0) / 1e6 / n:.2f} ms per step wall (under the profiler)")
else:
    steps = float(sys.argv[2]); q = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    where, args = "queue_id=?", (q,)
rows = c.execute(f"select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels where {where} group by name order by 3 desc", args).fetchall()
tot = sum(r[2] for r in rows)
print(f"queue {q}: {tot/1e3/steps:.2f} ms/step kernel time")
for n_, cnt, t, a in rows[:60]:
    n_ = n_.replace('(anonymous namespace)::', '').replace('void ', '')[:64]
    print(f"{t/1e3/steps:8.2f} ms/step {cnt/steps:7.1f}/step {a:8.1f} us  {n_}")
