"""Soak of `model.training_graphs` (GPU box): thousands of replayed MLM training steps at the BASELINE config-2 shape, with

  (a) a finiteness test of the whole flat gradient buffer after EVERY backward replay -- taken on the device (a count per
      step in a device array, read back every `--sync-every` steps), so the replays run back to back as in a training loop;
  (b) every `--eager-every`-th step re-run eagerly from the same state (same parameters, same mask-stream position) and
      compared with the replay bit for bit (loss and all 30.1 M gradient words);
  (c) optionally (`--b1-every N`) eight replays of a B=1 graph every N steps -- the sequence of bench.py's `graphed_step`
      leg, in which round 5 saw about one non-finite replay in 500.

The step it replays: reference main.py:59-90 (forward, loss, backward, clip, Adam).  Prints one JSON summary line at the end
(also written to gpurun_out/soak_graphs.json).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--replays", type=int, default=5000)
ap.add_argument("--layers", type=int, default=24)
ap.add_argument("--sync-every", type=int, default=50)
ap.add_argument("--eager-every", type=int, default=250)
ap.add_argument("--b1-every", type=int, default=0)
ap.add_argument("--out", default="gpurun_out/soak_graphs.json")
args = ap.parse_args()

sys.argv = ["bench.py"]
import bench  # noqa: E402
from frozenbilm_amd import lib as L  # noqa: E402
from frozenbilm_amd.model import DebertaV2Config, DebertaV2ForMaskedLM  # noqa: E402
from frozenbilm_amd.optim import FusedAdam  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load()
cfg = DebertaV2Config(num_hidden_layers=args.layers)
torch.manual_seed(0)
model = DebertaV2ForMaskedLM(cfg, max_feats=10, features_dim=1024, ds_factor_attn=8, ds_factor_ff=8, dropout=0.1).to(dev).train()
eng = model.engine()
opt = FusedAdam(model, lr=3e-5, betas=(0.9, 0.95))
B, T, F, Lt = 32, 10, 1024, 256
batch = bench.synth_batch(B, T, F, Lt, cfg.vocab_size, seed=1, device=dev)
small = bench.synth_batch(1, T, F, Lt, cfg.vocab_size, seed=77, device=dev)


def one_step(b, update=True):
    opt.zero_grad(set_to_none=False)
    out = model(**b)
    out.loss.backward()
    return out.loss


for _ in range(3):  # eager warm-up (lazy initialisations)
    one_step(batch)
    opt.step(clip_max_norm=0.1)
torch.cuda.synchronize()

model.training_graphs = True
n = args.replays
nf_count = torch.zeros(n, dtype=torch.int32, device=dev)  # non-finite gradient words of replay i
loss_nf = torch.zeros(n, dtype=torch.int32, device=dev)
ever_bad = torch.zeros_like(eng.flat_grad, dtype=torch.bool)
findings, eager_checks, eager_mismatch, b1_bad = [], 0, [], 0
good_flat = eng.flat.clone()
t0 = time.time()
last_sync = 0
for i in range(n):
    check_eager = args.eager_every and i % args.eager_every == args.eager_every - 1
    if check_eager:
        seed_before = model.step_seed
    loss = one_step(batch)
    g = model.engine().flat_grad
    bad = ~torch.isfinite(g)
    nf_count[i] = bad.sum()
    loss_nf[i] = (~torch.isfinite(loss.detach())).to(torch.int32)
    ever_bad |= bad
    if check_eager:  # the same step eagerly: same parameters (no update yet), same mask-stream position
        g_rep, l_rep = g.clone(), loss.detach().clone()
        model.training_graphs = False
        model.step_seed = seed_before
        l_eag = one_step(batch).detach()
        g_eag = model.engine().flat_grad
        same = bool(torch.equal(g_rep, g_eag)) and bool(torch.equal(l_rep, l_eag))
        eager_checks += 1
        if not same:
            d = (g_rep - g_eag).abs()
            eager_mismatch.append({"replay": i, "max_abs_diff": float(d.max()), "words": int((d != 0).sum()),
                                   "loss_replay": float(l_rep), "loss_eager": float(l_eag)})
            print(f"[replay {i}] replay != eager: {eager_mismatch[-1]}", flush=True)
        model.training_graphs = True
        g.copy_(g_rep)
    opt.step(clip_max_norm=0.1)
    if args.b1_every and i % args.b1_every == args.b1_every - 1:
        for _ in range(8):
            l1 = one_step(small)
            ok = bool(torch.isfinite(model.engine().flat_grad).all()) and bool(torch.isfinite(l1))
            b1_bad += not ok
            if ok:
                opt.step(clip_max_norm=0.1)
    if (i + 1) % args.sync_every == 0 or i == n - 1:
        torch.cuda.synchronize()
        c = nf_count[last_sync:i + 1].cpu()
        ln = loss_nf[last_sync:i + 1].cpu()
        hit = [(last_sync + k, int(c[k]), int(ln[k])) for k in range(c.numel()) if int(c[k]) or int(ln[k])]
        if hit:
            e = model.engine()
            names = []
            for nm in e.order:
                o, k = e.offsets[nm], e.named[nm].numel()
                if bool(ever_bad[o:o + k].any()):
                    names.append(nm)
            findings.append({"first_replay": hit[0][0], "replays": [h[0] for h in hit][:8], "nonfinite_words": hit[0][1],
                             "loss_nonfinite": hit[0][2], "tensors_hit": len(names), "names": names[:8]})
            print(f"[replays {last_sync}..{i}] NON-FINITE: {findings[-1]}", flush=True)
            e.flat.copy_(good_flat)  # parameters and moments back to the last good state, carry on
            e.params_version += 1
            if opt._m is not None:
                opt._m.zero_()
                opt._v.zero_()
            ever_bad.zero_()
        else:
            good_flat.copy_(model.engine().flat)
        last_sync = i + 1
        if (i + 1) % 500 == 0:
            print(f"[{i + 1} replays] {time.time() - t0:.0f} s, findings {len(findings)}, eager checks {eager_checks} "
                  f"(mismatches {len(eager_mismatch)}), B=1 bad {b1_bad}", flush=True)
dt = time.time() - t0
res = {"replays": n, "layers": args.layers, "batch": B, "b1_every": args.b1_every, "seconds": round(dt, 1),
       "ms_per_replayed_step": round(1e3 * dt / n, 2), "nonfinite_findings": findings, "eager_checks": eager_checks,
       "eager_mismatches": eager_mismatch, "b1_nonfinite": b1_bad, "captures": model.__dict__.get("_train_graph_captures", 0)}
print(json.dumps(res), flush=True)
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
with open(args.out, "w") as f:
    json.dump(res, f, indent=1)
