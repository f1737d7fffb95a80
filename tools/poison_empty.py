"""Does any result of the step depend on memory nobody wrote?  (GPU box; `python tools/poison_empty.py [--no-poison]`)

Every floating-point `torch.empty` / `torch.empty_like` on the GPU is filled with NaN before the caller sees it (integers are
left alone: an index buffer full of junk is a fault, not a finding), then the phases `bench.py` goes through after its timed
region are replayed in the same order on the same model -- padded step, `.logits` read, eval forward, B=1 step, graphed step,
packed-rows step, padded step again, `main.train_one_epoch` -- and after EVERY optimizer step the loss, every trainable
gradient and every trainable parameter must be finite.  A NaN names the phase and the parameters it reached.  Why: a default
`bench.py` run ended with a silent exit status 1 in ~1 run of 15 -- `main.train_one_epoch` stopping on a non-finite loss with its
message swallowed by the bench's stdout redirection.  The sequence of launches is deterministic, the content of recycled allocator
blocks is not (frees deferred by `record_stream`, what the previous process left in VRAM).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--no-poison", action="store_true")
ap.add_argument("--layers", type=int, default=24)
ap.add_argument("--phases", default="padded,logits,eval,b1,graphed,packed,padded2,loop")
ap.add_argument("--pmc-first", action="store_true",
                help="before the phases: the two `rocprofv3 --pmc` child runs bench.py makes for roofline.traffic (another process "
                     "collecting counters on this GPU while this one holds its context), as in a default bench run")
ap.add_argument("--poison-vram", action="store_true",
                help="before every phase: release the allocator's cache, fill 90 %% of the free device memory with the bit pattern "
                     "0x7FC07FC0 (NaN as fp32 and as both bf16 halves), release it again -- segments the allocator (or a graph's "
                     "private pool) then gets from the driver hold NaN wherever nobody writes: reads of unwritten memory AND reads "
                     "past the end of a tensor show up (what a process finds in VRAM that an earlier process left there)")
ap.add_argument("--cycles", type=int, default=1, help="repeat the phase list (soak: --no-poison --cycles 30 --sparse-checks)")
ap.add_argument("--sparse-checks", action="store_true",
                help="check at the END of a phase only: its steps run back to back, as in bench.py (a check synchronises)")
args = ap.parse_args()

FLOATS = (torch.float32, torch.bfloat16, torch.float16, torch.float64)
_empty, _empty_like = torch.empty, torch.empty_like
N_POISONED = [0]


def _poison(t):
    if t.is_cuda and t.numel() and t.dtype in FLOATS:
        t.fill_(float("nan"))
        N_POISONED[0] += 1
    return t


if not args.no_poison:
    torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))

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
bad_total = 0
flat0 = eng.flat.clone()


def restore():
    """a finding must not leak into the next phase: parameters and Adam moments back to the start"""
    e = model.engine()
    e.flat.copy_(flat0)
    e.params_version += 1
    if opt._m is not None:
        opt._m.zero_()
        opt._v.zero_()
    print("    (parameters and optimizer state restored)", flush=True)


def poison_vram(frac=0.9):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    left, chunks = int(free * frac), []
    while left > (1 << 28):
        sz = min(8 << 30, left)
        t = _empty(sz // 4, dtype=torch.int32, device=dev)
        t.fill_(0x7FC07FC0)
        chunks.append(t)
        left -= sz
    torch.cuda.synchronize()
    n = len(chunks)
    del chunks, t
    torch.cuda.empty_cache()
    print(f"    (device memory poisoned: {n} chunks, {free * frac / 2**30:.0f} GiB)", flush=True)


def bad_names(flat):
    names = []
    for n in eng.order:
        o, k = eng.offsets[n], eng.named[n].numel()
        if not torch.isfinite(flat[o:o + k]).all():
            names.append(n)
    return names


def check(tag, loss=None):
    global bad_total
    torch.cuda.synchronize()
    e = model.engine()
    msgs = []
    if loss is not None and not torch.isfinite(loss).all():
        msgs.append(f"loss {loss.item()}")
    if not torch.isfinite(e.flat_grad).all():
        b = bad_names(e.flat_grad)
        msgs.append(f"{len(b)} gradients non-finite: {b[:6]}{' ...' if len(b) > 6 else ''}")
    if not torch.isfinite(e.flat).all():
        b = bad_names(e.flat)
        msgs.append(f"{len(b)} PARAMETERS non-finite: {b[:6]}{' ...' if len(b) > 6 else ''}")
    print(f"[{tag}] " + ("ok" if not msgs else "NON-FINITE: " + "; ".join(msgs)) + (f"  loss {loss.item():.5f}" if loss is not None else ""),
          flush=True)
    bad_total += bool(msgs)
    return not msgs


def step(b, tag, full_logits=False):
    opt.zero_grad(set_to_none=False)
    out = model(**b)
    loss = out.loss
    if full_logits:
        lg = out.logits
        if not torch.isfinite(lg).all():
            print(f"[{tag}] logits non-finite", flush=True)
    loss.backward()
    if args.sparse_checks and not tag.endswith("last"):
        opt.step(clip_max_norm=0.1)
        return True
    ok_g = check(tag + " after backward", loss)
    opt.step(clip_max_norm=0.1)
    ok = check(tag + " after update") and ok_g
    if not ok:
        restore()
    return ok


phases = args.phases.split(",")
for i in range(12):  # bench.py's pre-warm
    step(batch, f"prewarm {i}" + (" last" if i == 11 else ""))
if args.pmc_first:
    import types

    t_p = time.time()
    live = bench.measure_traffic_live(types.SimpleNamespace(batch=B, text_len=Lt, layers=args.layers))
    print(f"[pmc children] {'ok' if live else 'no result'} in {time.time() - t_p:.0f} s", flush=True)
t0 = time.time()
for cyc, ph in ((c, p) for c in range(args.cycles) for p in phases):
    if args.cycles > 1 and ph == phases[0]:
        print(f"--- cycle {cyc} (+{time.time() - t0:.0f} s, findings so far: {bad_total})", flush=True)
    if args.poison_vram:
        poison_vram()
    if ph in ("padded", "padded2"):
        for i in range(3):
            step(batch, f"{ph} {i}" + (" last" if i == 2 else ""))
    elif ph == "logits":
        for i in range(2):
            step(batch, f"logits {i}" + (" last" if i == 1 else ""), full_logits=True)
    elif ph == "eval":
        model.eval()
        with torch.no_grad(), model.weights_frozen():
            for i in range(2):
                out = model(**batch)
                torch.cuda.synchronize()
                print(f"[eval {i}] loss {out.loss.item():.5f} finite={bool(torch.isfinite(out.loss))}", flush=True)
        model.train()
    elif ph == "b1":
        for i in range(3):
            step(small, f"b1 {i}" + (" last" if i == 2 else ""))
    elif ph == "graphed":
        model.training_graphs = True
        for i in range(4):
            step(batch, f"graphed {i}" + (" last" if i == 3 else ""))
        model.training_graphs = False
        model.__dict__.pop("_train_graphs", None)
    elif ph == "graphed_b1":
        # bench.py's `graphed_step` leg exactly: B=32 graph (capture + replays), B=1 graph (capture + replays), B=32 replays again
        model.__dict__["_train_graph_captures"] = 0
        model.training_graphs = True
        dense = (cyc % 2 == 1) or not args.sparse_checks
        for part, b, n in (("B32", batch, 8), ("B1", small, 8), ("B32 again", batch, 4)):
            for i in range(n):
                last = i == n - 1
                if dense:
                    opt.zero_grad(set_to_none=False)
                    out = model(**b)
                    out.loss.backward()
                    okg = check(f"graphed_b1 c{cyc} {part} {i} after backward", out.loss)
                    opt.step(clip_max_norm=0.1)
                    if not (check(f"graphed_b1 c{cyc} {part} {i} after update") and okg):
                        restore()
                else:
                    step(b, f"graphed_b1 c{cyc} {part} {i}" + (" last" if last else ""))
        model.training_graphs = False
        model.__dict__.pop("_train_graphs", None)
    elif ph == "packed":
        model.packed_rows = True
        try:
            for i in range(3):
                step(batch, f"packed {i}" + (" last" if i == 2 else ""))
            model.eval()
            with torch.no_grad(), model.weights_frozen():
                out = model(**batch)
                torch.cuda.synchronize()
                print(f"[packed eval] loss {out.loss.item():.5f}", flush=True)
            model.train()
        finally:
            model.packed_rows = False
    elif ph == "loop":
        for name, kw in (("reference_order", dict(delayed=False)), ("delayed", dict(delayed=True)),
                         ("graphed loop", dict(delayed=False, graphs=True)), ("packed loop", dict(delayed=True, packed=True))):
            try:
                r = bench.measure_train_loop(model, cfg, opt, B, T, F, Lt, 4, **kw)
                print(f"[loop {name}] {r['value']:.1f} samples/s", flush=True)
            except SystemExit as e:
                print(f"[loop {name}] train_one_epoch stopped the run: SystemExit({e.code})", flush=True)
                bad_total += 1
            if not check(f"loop {name}"):
                restore()
print(f"poisoned allocations: {N_POISONED[0]}; phases with a finding: {bad_total}; {time.time() - t0:.1f} s", flush=True)
