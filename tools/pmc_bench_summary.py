"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_bench.sh per kernel family.
usage: python tools/pmc_bench_summary.py <dir> <instrumented steps>     -> prints a table, writes <dir>/traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request and is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import collections, json, sqlite3, sys
root, steps = sys.argv[1], float(sys.argv[2])
res = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(f"{root}/{c}/p_results.db")
    cur = db.cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    ix = {k: i for i, k in enumerate(cols)}
    for r in cur.execute("select * from counters_collection").fetchall():
        name = str(r[ix.get("kernel_name", ix.get("name", 0))])
        if r[ix["counter_name"]] != c:
            continue
        fam = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        fam = fam.replace("fblgemm::", "")
        fam = fam.split("<")[0] if not fam.startswith("gemm") else fam  # keep the GEMM tile/epilogue variant
        res[fam][c] += float(r[ix["value"]])
        if c == "FETCH_SIZE":
            res[fam]["launches"] += 1
out = {}
for fam, v in sorted(res.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] * 2 + kv[1]["WRITE_SIZE"])):
    rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
    n = max(v["launches"], 1)
    out[fam] = {"launches_per_step": v["launches"] / steps, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
                "bytes_per_launch": (rd + wr) / n, "GB_per_step": (rd + wr) / steps / 1e9}
for fam, v in list(out.items())[:14]:
    print(f"{v['GB_per_step']:8.2f} GB/step {v['launches_per_step']:7.1f} launches/step {v['bytes_per_launch']/1e6:9.1f} MB/launch  {fam[:60]}")
print(f"total {sum(v['GB_per_step'] for v in out.values()):.1f} GB/step")
json.dump(out, open(f"{root}/traffic.json", "w"), indent=1)
