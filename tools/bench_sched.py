import os, sys, subprocess
for sc in (0, 1, 2, 3, 4):
    env = dict(os.environ, FBL_GEMM_SCHED=str(sc))
    out = subprocess.run([sys.executable, "tools/bench_gemm.py", "--iters", "20"], env=env, capture_output=True, text=True).stdout
    keep = [l for l in out.splitlines() if " bf16 " in l]
    print(f"--- SCHED {sc}"); print("\n".join(keep))
