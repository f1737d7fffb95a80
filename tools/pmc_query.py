import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm"
cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
rows = cur.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
agg = collections.defaultdict(list)
for r in rows:
    name = r[ix.get("kernel_name", ix.get("name", 0))]
    if pat in str(name):
        agg[(r[ix["counter_name"]])].append(r[ix["value"]])
for k, v in agg.items():
    v2 = v[len(v)//2:]  # skip warm-up launches
    print(f"{k:32s} n={len(v):3d} mean(last half)={sum(v2)/len(v2):.4g}")
