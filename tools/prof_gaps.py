"""Idle time of the main queue in a rocprofv3 kernel-trace DB, per optimizer step: the bubble between steps (end of adam_flat_kernel -> next
kernel start) and the sum of all other inter-kernel gaps.    python tools/prof_gaps.py <db> [n_last_steps] [queue]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 4; q = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = db.execute("select start, end, name from kernels where queue_id=? order by start", (q,)).fetchall()
ad = [i for i, r in enumerate(rows) if "adam_flat_kernel" in r[2]]
for a0, a1 in zip(ad[-n - 1:-1], ad[-n:]):
    seg = rows[a0:a1 + 1]
    bubble = (seg[1][0] - seg[0][1]) / 1e3
    gaps = sorted(((seg[i + 1][0] - max(r[1] for r in seg[:i + 1])) / 1e3, seg[i][2][:40], seg[i + 1][2][:40]) for i in range(1, len(seg) - 1))
    tot = sum(max(g[0], 0) for g in gaps)
    big = max(range(1, len(seg) - 1), key=lambda i: seg[i + 1][0] - max(r[1] for r in seg[:i + 1]))
    print("   around the largest gap: " + " | ".join(f"{r[2].split('(')[0][-34:]}@{(r[0] - seg[0][1]) / 1e3:.0f}us+{(r[1] - r[0]) / 1e3:.0f}" for r in seg[max(big - 3, 0):big + 4]))
    print(f"step: {len(seg)} kernels, wall {(seg[-1][1] - seg[0][1]) / 1e6:.2f} ms, bubble after adam {bubble:.1f} us, other gaps {tot / 1e3:.2f} ms "
          f"(mean {tot / max(len(gaps), 1):.2f} us); largest: " + "; ".join(f"{g[0]:.0f}us after {g[1]}" for g in gaps[-4:]))
