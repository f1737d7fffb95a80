import sqlite3, sys
db=sqlite3.connect(sys.argv[1]); steps=float(sys.argv[2]) if len(sys.argv)>2 else 5
cur=db.cursor()
rows=cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
tot=sum(r[2] for r in rows)
print(f"total kernel ms/step: {tot/1e3/steps:.1f}")
for n,c,t,a,p in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 26]:
    n=n.replace('(anonymous namespace)::','').replace('void ','')[:70]
    print(f"{t/1e3/steps:8.2f} ms/step {c/steps:7.1f} calls/step {a:9.1f} us  {p:5.1f}%  {n}")
