"""Per-kernel SQ busy figures of the training step from the passes of tools/pmc_step.sh.
usage: python tools/pmc_step_summary.py <dir> <instrumented steps>   -> markdown table on stdout, <dir>/pmc_step.json

Units (MI355X_MICROARCH.md, rocprofv3 section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over the
chip; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
  MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)         kernel cycles = GRBM_GUI_ACTIVE / 8
  VALU busy  = 4 x SQ_ACTIVE_INST_VALU / (1024 x kernel cycles)
  LDS busy   = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles)
  parked     = SQ_WAIT_ANY / SQ_WAVE_CYCLES  (waves waiting at s_waitcnt / barriers)"""
import collections, json, os, sqlite3, sys

root, steps = sys.argv[1], float(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n_launch = collections.Counter()
for sub in sorted(os.listdir(root)):
    p = os.path.join(root, sub)
    dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(p) for f in fs if f.endswith("_results.db")] if os.path.isdir(p) else []
    for dbp in dbs:
        cur = sqlite3.connect(dbp).cursor()
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        ix = {k: i for i, k in enumerate(cols)}
        first = None
        for r in cur.execute("select * from counters_collection").fetchall():
            name = str(r[ix.get("kernel_name", ix.get("name", 0))])
            fam = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("fblgemm::", "").split("(")[0]
            if "at::native" in fam or "rocclr" in fam:
                continue
            c = r[ix["counter_name"]]
            agg[fam][c] += float(r[ix["value"]])
            if first is None:
                first = c
            if c == first and sub == "s1":
                n_launch[fam] += 1
rows = []
for fam, v in agg.items():
    cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0:
        continue
    rows.append(dict(kernel=fam, launches_per_step=n_launch[fam] / steps, kcycles_per_step=cyc / steps / 1e3,
                     mfma_busy=v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc),
                     valu_busy=4 * v.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024 * cyc),
                     lds_busy=v.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256 * cyc),
                     lds_conflict=v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
                     parked=v.get("SQ_WAIT_ANY", 0.0) / max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0),
                     issuing=v.get("SQ_ACTIVE_INST_ANY", 0.0) / max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0),
                     insts_mfma_per_step=v.get("SQ_INSTS_MFMA", 0.0) / steps, insts_valu_per_step=v.get("SQ_INSTS_VALU", 0.0) / steps))
rows.sort(key=lambda r: -r["kcycles_per_step"])
tot = sum(r["kcycles_per_step"] for r in rows)
print("| kernel | launches/step | kernel kcycles/step (share) | MFMA busy | VALU busy | LDS busy | LDS bank-conflict share | waves parked | issuing |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows[:24]:
    print(f"| `{r['kernel'][:70]}` | {r['launches_per_step']:.0f} | {r['kcycles_per_step']:.0f} ({100 * r['kcycles_per_step'] / tot:.1f} %) | "
          f"{100 * r['mfma_busy']:.1f} % | {100 * r['valu_busy']:.1f} % | {100 * r['lds_busy']:.1f} % | {100 * r['lds_conflict']:.1f} % | "
          f"{100 * r['parked']:.0f} % | {100 * r['issuing']:.0f} % |")
mf = sum(r["mfma_busy"] * r["kcycles_per_step"] for r in rows) / max(tot, 1e-9)
print(f"\nwhole step (all queues, cycle-weighted): MFMA pipe {100 * mf:.1f} % busy over {tot / 1e3:.1f} M kernel cycles per step")
json.dump(rows, open(os.path.join(root, "pmc_step.json"), "w"), indent=1)
