"""per-step diagnostics of the bench loop: wall time, device mallocs/frees, Python GC pauses"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from frozenbilm_amd.model import DebertaV2Config, DebertaV2ForMaskedLM
from frozenbilm_amd.optim import FusedAdam
gc_log = []
def cb(phase, info):
    if phase == "start": cb.t = time.time()
    else: gc_log.append((info["generation"], (time.time() - cb.t) * 1e3, info["collected"]))
gc.callbacks.append(cb)
dev = torch.device("cuda", 0)
cfg = DebertaV2Config()
model = DebertaV2ForMaskedLM(cfg, max_feats=10, features_dim=1024, ds_factor_attn=8, ds_factor_ff=8, dropout=0.1).to(dev)
model.train(); opt = FusedAdam(model, lr=3e-5, betas=(0.9, 0.95))
batch = B.synth_batch(32, 10, 1024, 256, cfg.vocab_size, seed=1, device=dev)
def step():
    opt.zero_grad(set_to_none=False); loss = model(**batch).loss; loss.backward(); opt.step(clip_max_norm=0.1)
rows = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    st0 = torch.cuda.memory_stats(); n0 = len(gc_log)
    t = time.time(); step(); torch.cuda.synchronize(); dt = (time.time() - t) * 1e3
    if os.environ.get("DIAG_GC"):
        gc.collect()
    st1 = torch.cuda.memory_stats()
    g = [(a, round(b, 1)) for a, b, c in gc_log[n0:] if b > 2.0]
    rows.append((i, round(dt, 1), st1["num_device_alloc"] - st0["num_device_alloc"], st1["num_device_free"] - st0["num_device_free"],
                 round(st1["reserved_bytes.all.current"] / 1e9, 1), g))
for r in rows: print(r)
if os.environ.get("DIAG_CYCLES"):
    from frozenbilm_amd.engine import Run
    gc.collect()
    runs = [o for o in gc.get_objects() if isinstance(o, Run)]
    print("live Run objects:", len(runs))
    for r in runs[:2]:
        refs = gc.get_referrers(r)
        print("  referrers:", [type(x).__name__ + (":" + ",".join(list(x.keys())[:6]) if isinstance(x, dict) else "") for x in refs][:8])
