"""Benchmark of the FrozenBiLM masked-LM fwd+bwd hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one optimizer step of `main.train_one_epoch`'s body on a synthetic WebVid-shape batch already resident in
HBM: forward (train mode, dropout live as in the reference) + backward + [RCCL all-reduce of the 30.1 M trainable grads]
+ global-norm clip + Adam.  Workload = BASELINE configs[1]: DeBERTa-v2-XLarge + adapters (ds 8/8), B=32 per GPU,
T=10x1024 CLIP features, L=256 text tokens (S=266), seeded random weights (no checkpoints offline), bf16 MFMA compute.
Weak scaling: per-GPU batch fixed, value = total samples/s over all ranks.

Head: like `main.train_one_epoch` (main.py:67) the timed step reads only `.loss`; the model computes the prediction head
on the labelled rows and fills the full [B,S,128100] logits on first access (SURVEY.md section 7 "full logits only when
asked").  `--full-logits` reads `.logits` in every step (the reference's eager behaviour); the default run also times
a few such steps and reports them as `with_full_logits`.  FLOP figures always use the reference-faithful op list.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline     -- the dominant kernel (gemm8_kernel, the 8-phase 256x256 bf16 MFMA GEMM: ~45 % of the step's kernel time):
                  algorithmic FLOPs of its launches in one step / their summed HIP-event durations (measured live on the
                  launch stream in an instrumented replay of the same step), against the 2.5 PFLOP/s dense bf16 peak.
                  `family` inside it gives the same figures over EVERY launch of the GEMM family (small / split-K / dW
                  kernels included -- the round-1 definition), `executed_gemm_tflops_per_step` discloses what ran.
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's PyTorch path) timed on this box's host cores on a
                  bounded sample (B=2 sequences of the same shape, 1 warm-up + timed passes,
                  fwd+bwd, threads = physical cores), rank 0, N=1 only.
  eval_forward -- the eval-mode forward alone (north_star's ">= 40 % of peak on the fused forward"), timed in the same run.
  host         -- host cost of enqueueing one step, measured where the GPU cannot hide it (same launch sequence at B=1).

At --gpus 1 without a launcher the measurement runs in a child process (supervise_single_rank): if that process dies before
it printed its line (a signal, not a Python exception) it is started once more and the line says `"attempts": 2`; `--no-retry`
measures in this process (what the tools that put rocprofv3 in front of this script do).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (guides/MI355X_MICROARCH.md)


def algorithmic_flops_per_sample(S=266, H=1536, I=6144, A=192, P=512, V=128100, T=10, F=1024):
    """SURVEY.md section 8d, reference-faithful op list (2*M*N*K per GEMM)."""
    layer = 24 * S * H * H + 8 * S * H * A + 4 * S * S * H + 4 * S * P * H
    fwd = 26 * layer + 6 * S * H * H + 2 * T * F * H + 2 * S * H * H + 2 * S * H * V
    layer_bwd = 24 * S * H * H + 2 * (8 * S * H * A + 4 * S * S * H + 4 * S * P * H)
    bwd = 25 * layer_bwd + 6 * S * H * H + 2 * (2 * T * F * H) + 2 * S * H * H + 2 * S * H * V
    return fwd, bwd


def executed_flops_per_sample(S=266, H=1536, I=6144, A=192, P=512, V=128100, T=10, F=1024, rows_labelled=0.0, layers=24):
    """What the loss-only step really runs (per sample): no dead in-encoder pass of the last layer (layers + 1 layer
    executions instead of layers + 2), the vocabulary GEMM (forward and backward) on the labelled rows only."""
    layer = 24 * S * H * H + 8 * S * H * A + 4 * S * S * H + 4 * S * P * H
    fwd = (layers + 1) * layer + 6 * S * H * H + 2 * T * F * H + 2 * S * H * H + 2 * rows_labelled * H * V
    layer_bwd = 24 * S * H * H + 2 * (8 * S * H * A + 4 * S * S * H + 4 * S * P * H)
    bwd = (layers + 1) * layer_bwd + 6 * S * H * H + 2 * (2 * T * F * H) + 2 * S * H * H + 2 * rows_labelled * H * V
    return fwd, bwd


def synth_batch(B, T, F, L, V, seed, device):
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(B, T, F, generator=g).half().float()
    vlen = torch.randint(1, T + 1, (B,), generator=g)
    vlen[0] = T
    tlen = torch.randint(32, L + 1, (B,), generator=g)
    tlen[-1] = L
    ids = torch.randint(5, V, (B, L), generator=g)
    amask = (torch.arange(L)[None] < tlen[:, None]).long()
    ids = ids * amask
    vmask = (torch.arange(T)[None] < vlen[:, None]).long()
    sel = (torch.rand(B, L, generator=g) < 0.15) & amask.bool()
    sel[:, 1] = True
    labels = torch.where(sel, ids, torch.full_like(ids, -100))
    return {k: v.to(device) for k, v in dict(video=video, video_mask=vmask, input_ids=ids, attention_mask=amask,
                                               labels=labels).items()}


PREWARM_STEPS = 12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--text-len", type=int, default=256)
    ap.add_argument("--layers", type=int, default=24, help="debug only; anything but 24 is reported as a reduced config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eval-forward", action="store_true", help="time the eval forward only (reported under its own metric)")
    ap.add_argument("--full-logits", action="store_true", help="read .logits in every timed step (reference-eager head)")
    ap.add_argument("--no-extras", action="store_true", help="skip the with_full_logits / eval_forward / host side measurements")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not re-measure roofline.traffic (two short rocprofv3 --pmc child runs of this script); the newest "
                         "committed profiles/*_traffic.json is reported instead")
    ap.add_argument("--prewarm", type=int, default=PREWARM_STEPS, help=argparse.SUPPRESS)
    ap.add_argument("--no-retry", action="store_true",
                    help="measure in this process (default at --gpus 1: in a child process that is started once more if it dies "
                         "without printing its line)")
    ap.add_argument("--packed-rows", action="store_true",
                    help="time the step with model.packed_rows = True (NOT the headline: reported under its own metric name; "
                         "for profiling the packed step)")
    ap.add_argument("--training-graphs", action="store_true",
                    help="time the step with model.training_graphs = True (reported in config; for profiling the replayed step)")
    ap.add_argument("--no-inference-graphs", action="store_true",
                    help="downstream workloads: leave the loops' opt-in args.inference_graphs off (eager launches)")
    ap.add_argument("--dp-overlap", default=None, choices=["attention_windows", "backward", "after"],
                    help="where in backward the gradient collectives are launched (parallel.GradReducer.overlap; default: "
                         "attention_windows) -- tools/scale_sweep.sh sweeps it")
    ap.add_argument("--force-reducer", action="store_true", help="attach the GradReducer at N=1 too (bucket bookkeeping without collectives)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST HOOK, never used by the driver: every rank on cuda:0, collectives over gloo -- exercises the N > 1 "
                         "code path on a one-GPU box; the line says rccl_ranks 0")
    ap.add_argument("--workload", default="mlm", choices=["mlm", "videoqa", "mc"],
                    help="mlm = BASELINE configs[1] (the headline); videoqa = configs[3] (zero-shot open-ended eval loop, "
                         "n_ans=1000); mc = configs[4] (4-way multiple choice, B=8, S=512)")
    args = ap.parse_args()
    if _wants_supervisor(args):
        sys.exit(supervise_single_rank())
    import faulthandler

    faulthandler.enable()  # a SIGSEGV / SIGABRT of the measuring process leaves a Python traceback on stderr instead of nothing
    import atexit

    atexit.register(_report_exit_without_line)
    if args.workload != "mlm":
        return run_downstream(args)
    if args.training_graphs:  # (a replayed step issues no launches the roofline instrumentation could bracket)
        args.no_roofline = args.no_extras = True
    if args.packed_rows:
        args.no_extras = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        rc = spawn_ranks(args.gpus)
        _phase("done")  # (this process only launched the ranks: rank 0 printed the line)
        sys.exit(rc)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # --share-gpu (test hook, never set by the driver): every rank uses cuda:0 and the collectives go through gloo, so the
    # N > 1 code path (rank spawning, reducer, max-over-ranks timing) can be exercised on a one-GPU box.
    share = bool(args.share_gpu)
    local = 0 if share else local
    _phase("set_device")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo" if share else "nccl", init_method="env://", world_size=world, rank=rank)

    from frozenbilm_amd import lib as L
    from frozenbilm_amd.model import DebertaV2Config, DebertaV2ForMaskedLM
    from frozenbilm_amd.optim import FusedAdam
    from frozenbilm_amd.parallel import GradReducer

    _phase("load libfbl.so")
    L.load()
    _phase("build model (host)")
    cfg = DebertaV2Config(num_hidden_layers=args.layers)
    torch.manual_seed(0)
    t_build = time.time()
    model = DebertaV2ForMaskedLM(cfg, max_feats=10, features_dim=1024, ds_factor_attn=8, ds_factor_ff=8, dropout=0.1)
    _phase("model.to(device)")
    model.to(dev)
    model.train(not args.eval_forward)
    model.training_graphs = bool(args.training_graphs)
    model.packed_rows = bool(args.packed_rows)
    _phase("engine (operand copies)")
    eng = model.engine()
    opt = FusedAdam(model, lr=3e-5, betas=(0.9, 0.95))
    # --force-reducer exercises the bucket bookkeeping on a single GPU (the collectives are skipped at world 1)
    red = GradReducer.attach(model, overlap=args.dp_overlap) if (world > 1 or args.force_reducer) else None
    if world > 1 and not share and red.rccl_ranks != world:
        # a multi-GPU line whose collectives did not run over RCCL on all N ranks is not a scaling measurement: refuse to print one
        raise RuntimeError(f"--gpus {world}: the gradient exchange would run over {red.rccl_ranks} RCCL ranks "
                           f"(backend {dist.get_backend()}), expected {world}")
    B, T, F, Lt = args.batch, 10, 1024, args.text_len
    batch = synth_batch(B, T, F, Lt, cfg.vocab_size, seed=1 + rank, device=dev)
    t_build = time.time() - t_build

    def step(full_logits=args.full_logits):
        if args.eval_forward:
            with torch.no_grad():
                return model(**batch).loss
        opt.zero_grad(set_to_none=False)
        out = model(**batch)
        loss = out.loss
        if full_logits:
            out.logits  # materialises the [B,S,V] tensor (filled on first access)
        loss.backward()
        opt.step(clip_max_norm=0.1)
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clock / power-state ramp: a box that has just been handed over can sit in a low-power state for the first second of
    # load (one round-1 run measured 2x slower end to end for that reason), so a fixed untimed pre-warm precedes the W
    # warm-up steps the contract asks for
    _phase("pre-warm steps")
    for _ in range(args.prewarm):
        step()
    _phase("warm-up + timed steps")
    for _ in range(args.warmup):
        loss = step()
    sync()
    t0 = time.time()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_marks = []
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
        host_marks.append(time.time())
    t_host = time.time() - t0  # host time of the loop (it blocks once per step, on the label-row count at the start of forward)
    sync()
    dt = time.time() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    loss_value = float(loss.item())
    params_finite = bool(torch.isfinite(eng.flat).all().item())  # after prewarm + warm-up + the timed steps
    step_ms_gpu = [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(args.steps)]  # diagnostic: per-step GPU timeline
    step_ms_host = [round((b - a) * 1e3, 2) for a, b in zip([t0] + host_marks[:-1], host_marks)]
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    S = T + Lt
    fwd_f, bwd_f = algorithmic_flops_per_sample(S=S)
    step_flops = (fwd_f if args.eval_forward else fwd_f + bwd_f) * B
    whole_step_tflops = step_flops / (ms_per_step * 1e-3) / 1e12

    # box calibration (context for the reader, not a result): the pool's boxes differ -- the same tree measured 46.5-47.7 ms per
    # step on most of them and 57.6 on one -- so the line carries what THIS box gives a fixed, well-known launch: the 8-phase
    # GEMM at 4096^3 (1270-1370 TFLOP/s on a healthy box), right after the timed region.
    box = None
    _phase("box calibration")
    if rank == 0:
        ca = (torch.rand(4096, 4096, device=dev) * 2 - 1).to(torch.bfloat16)
        cb = (torch.rand(4096, 4096, device=dev) * 2 - 1).to(torch.bfloat16)
        co = torch.empty(4096, 4096, dtype=torch.bfloat16, device=dev)
        for _ in range(5):
            L.gemm(ca, cb, out_bf16=co)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            L.gemm(ca, cb, out_bf16=co)
        e1.record()
        torch.cuda.synchronize()
        box = {"gemm_4096_tflops": 2.0 * 4096 ** 3 * 30 / (e0.elapsed_time(e1) * 1e-3) / 1e12,
               "note": "fbl_gemm_bf16_nt at 4096^3 on this box right after the timed region (1270-1370 on most boxes of the pool)"}
        del ca, cb, co

    roofline = None
    _phase("roofline replay / traffic passes")
    if not args.no_roofline:
        # every rank replays the instrumented steps (they contain the gradient collectives); rank 0 reports its own
        roofline = measure_gemm_roofline(L, step)
        if rank == 0 and world == 1 and not args.no_traffic and not args.eval_forward:
            live = measure_traffic_live(args)
            if live is not None:
                roofline.update(live)
        roofline["whole_step_algorithmic_tflops"] = whole_step_tflops
        roofline["whole_step_frac_of_peak"] = whole_step_tflops / PEAK_BF16_TFLOPS
        sync()

    extras = {}
    _phase("extras")
    full_cfg = args.layers == 24 and B == 32 and Lt == 256
    if not args.no_extras and not args.eval_forward:
        def timed(fn, n):
            for _ in range(2):
                fn()
            sync()
            t = time.time()
            for _ in range(n):
                fn()
            sync()
            d = time.time() - t
            if world > 1:
                tm = torch.tensor([d], device=dev, dtype=torch.float64)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                d = tm.item()
            return d / n

        n_x = max(3, min(args.steps, 6))
        batch0 = dict(batch)
        snap = {"flat": eng.flat.clone(), "m": None if opt._m is None else opt._m.clone(),
                "v": None if opt._v is None else opt._v.clone(), "step": opt._step}

        class Leg:
            """One auxiliary measurement.  The headline was measured above: a leg that fails (an exception; the product's training
            loop stopping the run on a non-finite loss) is recorded under `failed_legs` in the line and costs nothing else -- the
            model goes back to the state the next leg expects.  After every leg the trainable parameters must be finite; if they
            are not, the leg is named under `nonfinite_parameters_after` and parameters + Adam moments return to their values at
            the end of the timed region.  (Single process only: a rank that skipped a leg would leave the others in its
            collectives.)"""

            def __init__(self, name):
                self.name = name

            def __enter__(self):
                _phase("extras: " + self.name)
                return self

            def __exit__(self, et, ev, tb):
                failed = et is not None and issubclass(et, (Exception, SystemExit))
                if failed and world > 1:
                    return False
                if failed:
                    extras.setdefault("failed_legs", {})[self.name] = f"{et.__name__}: {ev}"[:600]
                    print(f"[bench] extras leg {self.name!r} failed: {et.__name__}: {ev}", file=sys.stderr, flush=True)
                    model.training_graphs = False
                    model.__dict__.pop("_train_graphs", None)
                    model.packed_rows = False
                    model.train()
                    batch.clear()
                    batch.update(batch0)
                torch.cuda.synchronize()
                e = model.engine()
                if e.flat.numel() == snap["flat"].numel() and not bool(torch.isfinite(e.flat).all()):
                    bad = [n for n in e.order if not bool(torch.isfinite(e.flat[e.offsets[n]: e.offsets[n] + e.named[n].numel()]).all())]
                    extras.setdefault("nonfinite_parameters_after", []).append({"leg": self.name, "tensors": len(bad), "first": bad[:6]})
                    print(f"[bench] {len(bad)} trainable tensors are non-finite after extras leg {self.name!r} (first: {bad[:6]}): "
                          "parameters and Adam moments restored", file=sys.stderr, flush=True)
                    e.flat.copy_(snap["flat"])
                    e.params_version += 1
                    if snap["m"] is not None:
                        opt._m.copy_(snap["m"])
                        opt._v.copy_(snap["v"])
                    opt._step = snap["step"]
                return failed

        def fwd_only():
            with torch.no_grad():
                return model(**batch).loss

        rows_lab = float((batch["labels"] != -100).sum().item()) / B
        ex_f, ex_b = executed_flops_per_sample(S=S, rows_labelled=rows_lab, layers=args.layers)
        if not args.full_logits:
            with Leg("with_full_logits"):
                d = timed(lambda: step(True), n_x)
                extras["with_full_logits"] = {"value": world * B / d, "unit": "samples/s", "ms_per_step": d * 1e3, "steps": n_x,
                                              "note": "same step, .logits read every step: the [B,S,128100] fp32 tensor is produced"}
        with Leg("eval_forward"):
            model.eval()
            with model.weights_frozen():  # as main.evaluate runs it: nothing writes to the parameters between these forwards
                d = timed(fwd_only, n_x)
            model.train()
            # primary figure: FLOPs this forward EXECUTED over its time; the reference-op-list figure is kept as secondary
            extras["eval_forward"] = {"value": world * B / d, "unit": "samples/s", "ms_per_step": d * 1e3, "steps": n_x,
                                      "executed_tflops": ex_f * B / d / 1e12, "executed_frac_of_peak": ex_f * B / d / 1e12 / PEAK_BF16_TFLOPS,
                                      "oplist_tflops": fwd_f * B / d / 1e12, "oplist_frac_of_peak": fwd_f * B / d / 1e12 / PEAK_BF16_TFLOPS,
                                      "note": "eval-mode forward with labels (loss on the labelled rows; logits filled on access). "
                                              "executed_* counts only what this forward ran (vocabulary GEMM on the labelled rows, 25 "
                                              "layer executions); oplist_* divides the reference's op list (SURVEY 8d: full-vocabulary "
                                              "head on every row, dead layer-23 pass) by the same time"}

        def fwd_logits():
            with torch.no_grad():
                out = model(**batch)
                return out.loss + out.logits[0, 0, 0]  # touching .logits produces the [B,S,V] fp32 tensor (model/deberta.py:1474-1479)

        with Leg("eval_forward_full_logits"):
            model.eval()
            with model.weights_frozen():
                d = timed(fwd_logits, n_x)
            model.train()
            # executed here = the op list minus the dead layer-23 pass: the full-vocabulary GEMM runs on every row
            ex_full, _ = executed_flops_per_sample(S=S, rows_labelled=float(S), layers=args.layers)
            extras["eval_forward"]["with_full_logits"] = {
                "value": world * B / d, "unit": "samples/s", "ms_per_step": d * 1e3, "steps": n_x,
                "executed_tflops": ex_full * B / d / 1e12, "executed_frac_of_peak": ex_full * B / d / 1e12 / PEAK_BF16_TFLOPS,
                "note": "the same forward with .logits read: the [B,S,128100] fp32 tensor is produced (the reference-eager behaviour)"}
        extras["executed_tflops_per_step"] = (ex_f + ex_b) * B / 1e12
        # single-GPU characterisations (host cost, launch graphs, packed rows): not repeated on every rank of a multi-GPU run
        if world == 1:
            # host cost of one step where the GPU cannot hide it: the same launch sequence on a B=1 batch
            small = synth_batch(1, T, F, Lt, cfg.vocab_size, seed=77, device=dev)
            keep = dict(batch)

            def host_issue_ms(n=4):
                # what the HOST spends issuing one full-size step: the queue is empty when the step starts and nobody waits for
                # the GPU afterwards (the label count at the start of forward finds its tiny kernel done at once)
                tot = 0.0
                for _ in range(n):
                    sync()
                    t = time.time()
                    step(False)
                    tot += time.time() - t
                sync()
                return tot / n * 1e3

            with Leg("host"):
                batch.clear(); batch.update(small)
                d = timed(lambda: step(False), n_x)
                batch.clear(); batch.update(keep)
                extras["host"] = {"enqueue_ms_per_step": d * 1e3, "issue_ms_per_step": host_issue_ms(),
                                  "note": "enqueue_ms_per_step: wall time per step of the same launch sequence at B=1 (GPU work per launch "
                                          "negligible: ~1450 dependent launches cost that much on the GPU side too); issue_ms_per_step: "
                                          "host time to issue one full-size step into an empty queue, nobody waiting for the GPU"}
            # the same step with model.training_graphs (forward and backward replayed as two hipGraphs; clip + Adam eager): GPU
            # time per step, and the host cost where the GPU cannot hide it (B=1, as `host` above)
            with Leg("graphed_step"):
                model.training_graphs = True
                d = timed(lambda: step(False), n_x)
                batch.clear(); batch.update(small)
                dh = timed(lambda: step(False), n_x)
                batch.clear(); batch.update(keep)
                gi = host_issue_ms()
                model.training_graphs = False
                model.__dict__.pop("_train_graphs", None)  # (each captured shape holds one step's activations)
                extras["graphed_step"] = {"value": world * B / d, "unit": "samples/s", "ms_per_step": d * 1e3, "steps": n_x,
                                          "enqueue_ms_per_step": dh * 1e3, "issue_ms_per_step": gi,
                                          "note": "model.training_graphs = True: forward and backward of the step replayed as two "
                                                  "hipGraphs (frozenbilm_amd/train_graph.py); enqueue_ms_per_step / issue_ms_per_step "
                                                  "as under `host`"}
            # the same step with model.packed_rows: the batch is ragged (text 32..256 tokens, 1..10 frames) and every GEMM,
            # LayerNorm and adapter of the headline step also processes the padding rows behind each sample's last token, as
            # the reference does.  Packed, those rows do not exist.  A separately named object: its fraction
            # counts EXECUTED FLOPs only (per sample: the op list of executed_flops_per_sample at that sample's own length)
            # and earns nothing against the padded op list; the headline `value` stays reference-shaped.
            with Leg("packed_rows"):
                model.packed_rows = True
                try:
                    d = timed(lambda: step(False), n_x)
                    with torch.no_grad():
                        model.eval()
                        pk = model(**batch)._run.pk
                        with model.weights_frozen():
                            d_f = timed(fwd_only, n_x)
                        model.train()
                finally:
                    model.packed_rows = False
                if pk is not None:
                    plen = (pk.row0[1:] - pk.row0[:-1]).tolist()
                    ex_p = [executed_flops_per_sample(S=s, rows_labelled=rows_lab, layers=args.layers) for s in plen]
                    exf_p, exb_p = sum(e[0] for e in ex_p), sum(e[1] for e in ex_p)
                    extras["packed_rows"] = {
                        "value": world * B / d, "unit": "samples/s", "ms_per_step": d * 1e3, "steps": n_x,
                        "rows": int(pk.n), "grid_rows": B * S,
                        "executed_tflops_per_step": (exf_p + exb_p) / 1e12,
                        "executed_frac_of_peak": (exf_p + exb_p) / d / 1e12 / PEAK_BF16_TFLOPS,
                        "eval_forward": {"value": world * B / d_f, "unit": "samples/s", "ms_per_step": d_f * 1e3,
                                         "executed_frac_of_peak": exf_p / d_f / 1e12 / PEAK_BF16_TFLOPS},
                        "note": "model.packed_rows = True (opt-in extension, frozenbilm_amd.engine.Packing): the same batch, same "
                                "loss and gradients (tests/test_gpu_model.py::test_packed_rows_*), without the padding rows behind "
                                "each sample's last token; NOT the headline: the reference computes those rows and the headline "
                                "counts them"}
        if world == 1 and full_cfg:
            # the loop a user runs (main.train_one_epoch: host-side masking, copies, loss logging) with NO opt-in set
            # (`reference_order`: what the two-line swap of INTEGRATION.md gives -- the backward is enqueued before the host reads
            # the loss, the non-finite check still precedes the update: loops.LossLog) and with the opt-ins
            n_l = max(4, min(args.steps, 8))
            extras["train_one_epoch"] = {"note": "frozenbilm_amd.main.train_one_epoch over synthetic batches that start on the "
                                                 "host (CPU mask_tokens, H2D copies, loss logging): the loop, not the step body; "
                                                 "reference_order = the default loop, no opt-in (loss read after the backward "
                                                 "is enqueued, before optimizer.step: same stop-before-update behaviour as main.py:73-84)"}
            for key, kw in (("reference_order", dict(delayed=False)), ("delayed_loss_check", dict(delayed=True)),
                            ("reference_order_graphed", dict(delayed=False, graphs=True)),
                            ("delayed_loss_check_packed_rows", dict(delayed=True, packed=True))):
                with Leg("train_one_epoch." + key):
                    extras["train_one_epoch"][key] = measure_train_loop(model, cfg, opt, B, T, F, Lt, n_l, **kw)

    cpu_baseline = cpu_baseline_cfg1 = None
    _phase("cpu baselines")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = measure_cpu_baseline(model, cfg, T, F, Lt, fwd_only=args.eval_forward)
        cpu_baseline_cfg1 = measure_cpu_baseline_cfg1()

    if not args.no_extras and not args.eval_forward and world == 1 and full_cfg:
        # BASELINE configs[3] / configs[4] through the product's evaluate loops on the same model (last: the answer table is
        # swapped in and the engine rebuilt), with the loops' opt-in inference graphs and with eager launches
        for wl, key in (("videoqa", "videoqa_eval"), ("mc", "mc_eval")):
            with Leg(key):
                on = measure_downstream(model, cfg, wl, n_x, 1, args.layers, graphs=True)
                off = measure_downstream(model, cfg, wl, n_x, 1, args.layers, graphs=False)
                on["eager_launches"] = {"value": off["value"], "ms_per_step": off["ms_per_step"]}
                pkd = measure_downstream(model, cfg, wl, n_x, 1, args.layers, graphs=False, packed=True)
                on["packed_rows"] = {"value": pkd["value"], "ms_per_step": pkd["ms_per_step"],
                                     "note": "model.packed_rows = True (eager launches): the same loop without the padding rows "
                                             "behind each sample's last token"}
                extras[key] = on

    if rank == 0:
        full = args.layers == 24 and B == 32 and Lt == 256
        out = {
            "metric": ("video-text samples/sec (MLM fwd+bwd) DeBERTa-XL+adapters" if not args.eval_forward
                       else "video-text samples/sec (MLM eval forward) DeBERTa-XL+adapters") + ("" if full else " [REDUCED CONFIG]")
                      + (" [PACKED ROWS: not the reference-shaped headline]" if args.packed_rows else ""),
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"DeBERTa-v2-XLarge({args.layers}L)+adapters ds8/8, B={B}/GPU, T=10x1024, L={Lt} (S={S}), "
                                   "MLM fwd+bwd+allreduce+clip+Adam, dropout 0.1 live, seeded random weights",
                       "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}",
                       "head": ("full fp32 logits [B,S,128100] read every step; " if args.full_logits else
                                "loss-only step as in main.py:67: full fp32 logits [B,S,128100] filled on access (not read here); ")
                               + "CE + head backward on labelled rows",
                       "dead_layer23_encoder_pass": "skipped (output unused, SURVEY fact 6); FLOPs still counted",
                       **({"model.training_graphs": True} if args.training_graphs else {}),
                       **({"model.packed_rows": True} if args.packed_rows else {})},
            "prewarm_steps": PREWARM_STEPS, "step_ms_gpu": step_ms_gpu, "step_ms_host": step_ms_host, "loadavg": os.getloadavg()[0],
            "loss": loss_value, "trainable_parameters_finite_after_timed_steps": params_finite,
            "host_loop_ms_per_step": t_host / args.steps * 1e3,
            # data parallel: ranks RCCL's collectives ran over (0 = no reducer / not the nccl backend) and where in backward
            # they are launched (parallel.GradReducer.overlap; --dp-overlap)
            "rccl_ranks": red.rccl_ranks if red is not None else 0, "dp_overlap": red.overlap if red is not None else None,
            "algorithmic_tflops_per_step": step_flops / 1e12,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "cpu_baseline_cfg1": cpu_baseline_cfg1, "model_build_s": t_build,
            "box_calibration": box,
        }
        out.update(extras)
        if os.environ.get(CHILD_MARK) == "2":  # second attempt of supervise_single_rank: say so, and how the first one ended
            out["attempts"] = 2
            out["first_attempt_exit_status"] = int(os.environ.get("FBL_BENCH_FIRST_RC", "0"))
            out["first_attempt_last_phase"] = os.environ.get("FBL_BENCH_FIRST_PHASE", "unknown")
        print(json.dumps(out), flush=True)
    _phase("done")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure_downstream(model, cfg, workload, steps, warmup, layers=24, graphs=True, packed=False):
    """BASELINE configs[3] / configs[4] on one GPU: the product's videoqa.evaluate / mc.evaluate loops (reference
    signatures) over synthetic batches resident in HBM; a step = one batch through the loop body (forward, [MASK]-row
    head, softmax, top-k / candidate arg-max).  `model` is put into eval mode and gets the workload's answer table
    (set_answer_embeddings: n_ans = 1000 / 2).  graphs: `args.inference_graphs` of the loops (one hipGraph per batch shape);
    packed: `model.packed_rows` (the ragged batch without its padding rows; takes precedence over the graphs)."""
    import types

    from frozenbilm_amd import mc as P_mc
    from frozenbilm_amd import videoqa as P_vqa

    dev = model.device
    vqa = workload == "videoqa"
    n_ans = 1000 if vqa else 2
    model.eval()
    g = torch.Generator().manual_seed(5)
    a2tok = torch.randint(5, cfg.vocab_size, (n_ans, 5), generator=g)
    a2tok = a2tok * (torch.arange(5)[None] < torch.randint(1, 6, (n_ans, 1), generator=g))
    model.set_answer_embeddings(a2tok.to(dev))
    model.inference_graphs = False
    model.packed_rows = bool(packed)
    MASK, B, T, F = 128000, (32 if vqa else 8), 10, 1024
    Lt = 256 if vqa else 502
    C = 1 if vqa else 4

    class Tok:  # the loops only need the ids; texts are pre-tokenised id lists
        mask_token_id, pad_token_id, sep_token_id = MASK, 0, 2

        def __call__(self, text, **kw):
            ids = torch.stack(text)
            return {"input_ids": ids, "attention_mask": (ids != 0).long()}

    def texts(seed):
        gg = torch.Generator().manual_seed(seed)
        tlen = torch.randint(Lt // 8, Lt + 1, (B,), generator=gg)
        tlen[-1] = Lt
        ids = torch.randint(5, 127000, (B, Lt), generator=gg) * (torch.arange(Lt)[None] < tlen[:, None])
        ids[torch.arange(B), torch.stack([torch.randint(1, int(t), (1,), generator=gg) for t in tlen]).view(-1)] = MASK
        return list(ids)

    video = torch.randn(B, T, F, generator=g).half().float()
    vlen = torch.randint(1, T + 1, (B,), generator=g)
    batch = dict(video=video.to(dev), video_len=vlen, qid=list(range(B)), type=[0] * B,
                 answer_id=torch.randint(0, n_ans if vqa else C, (B,), generator=g),
                 text=texts(11) if vqa else [texts(11 + c) for c in range(C)])
    largs = types.SimpleNamespace(max_feats=T, use_video=True, suffix="", use_context=True, max_tokens=Lt, print_freq=10 ** 9,
                                  inference_graphs=bool(graphs))

    class Loader(list):
        dataset = list(range(B))

    tok = Tok()

    def step():
        import contextlib
        import io

        with contextlib.redirect_stdout(io.StringIO()):
            if vqa:
                return P_vqa.evaluate(model, tok, Loader([batch]), dev, "msrvtt", largs, thresholds=[1, 10])
            return P_mc.evaluate(model, tok, Loader([batch]), dev, "how2qa", largs)

    for _ in range(warmup + 3):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.time() - t0
    S = T + Lt
    fwd_f, _ = algorithmic_flops_per_sample(S=S, V=n_ans)
    tf = fwd_f * B * C * steps / dt / 1e12
    n_graphs = len(model.__dict__.get("_graph_cache", {}))
    model.inference_graphs = False
    model.packed_rows = False
    model.__dict__.pop("_graph_cache", None)
    return {"metric": ("zero-shot open-ended VideoQA eval samples/sec (videoqa.evaluate, n_ans=1000)" if vqa else
                       "multiple-choice VideoQA eval questions/sec (mc.evaluate, 4 candidates, S=512)"),
            "value": B * steps / dt, "unit": "samples/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{3 if vqa else 4}]: DeBERTa-v2-XLarge({layers}L)+adapters, B={B}, T=10x1024, L={Lt} "
                                   f"(S={S}), n_ans={n_ans}, {C} candidate(s) per question in ONE forward of {B * C} samples, eval loop "
                                   "incl. host-side result bookkeeping, head on the [MASK] rows only",
                       "args.inference_graphs": bool(graphs), "inference_graphs_captured": n_graphs},
            "candidate_forwards_per_s": B * C * steps / dt, "algorithmic_tflops": tf, "frac_of_peak": tf / PEAK_BF16_TFLOPS}


def run_downstream(args):
    """`--workload videoqa | mc`: the downstream line on its own (own metric names: not the headline).  The loops' opt-in
    `args.inference_graphs` is on unless --no-inference-graphs; the line says which."""
    from frozenbilm_amd.model import DebertaV2Config, DebertaV2ForMaskedLM

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = DebertaV2Config(num_hidden_layers=args.layers)
    torch.manual_seed(0)
    model = DebertaV2ForMaskedLM(cfg, max_feats=10, features_dim=1024, ds_factor_attn=8, ds_factor_ff=8, dropout=0.1,
                                 n_ans=1000 if args.workload == "videoqa" else 2)
    model.to(dev).eval()
    print(json.dumps(measure_downstream(model, cfg, args.workload, args.steps, args.warmup, args.layers,
                                        graphs=not args.no_inference_graphs)), flush=True)
    _phase("done")


def measure_train_loop(model, cfg, opt, B, T, F, Lt, steps, delayed, graphs=False, packed=False):
    """The product's `main.train_one_epoch` (reference signature, main.py:24-96) over `steps` synthetic batches: host-side
    tokenisation stand-in + `mask_tokens` on the CPU generator + host-to-device copies + forward + loss logging + backward +
    clip + Adam -- the headline times the step body on resident inputs, this is the loop a user runs.  delayed =
    `args.delayed_loss_check` (the loss of step i is read when step i+1 calls instead of before its own backward); graphs =
    `model.training_graphs` (forward and backward replayed as hipGraphs: the host needs ~2 ms to issue the backward once it
    has read the loss, instead of ~12)."""
    import contextlib
    import io
    import types

    from frozenbilm_amd import main as P_main

    class Tok:
        mask_token, _pad_token, pad_token_id, mask_token_id, cls_token_id, sep_token_id = "[MASK]", "[PAD]", 0, 128000, 1, 2

        def __len__(self):
            return cfg.vocab_size

        def convert_tokens_to_ids(self, t):
            return {"[MASK]": self.mask_token_id, "[PAD]": 0}[t]

        def get_special_tokens_mask(self, row, already_has_special_tokens=True):
            return [1 if v in (0, 1, 2, 128000) else 0 for v in row]

        def __call__(self, text, **kw):
            ids = torch.stack(text)
            return {"input_ids": ids, "attention_mask": (ids != 0).long()}

    g = torch.Generator().manual_seed(21)
    batches = []
    n_warm = 5 if graphs else 2  # (graphs: the labelled-row capacities of the timed batches should have been captured)
    for _ in range(steps + n_warm):
        tlen = torch.randint(Lt // 8, Lt + 1, (B,), generator=g)
        tlen[-1] = Lt
        ids = torch.randint(5, 127000, (B, Lt), generator=g) * (torch.arange(Lt)[None] < tlen[:, None])
        vlen = torch.randint(1, T + 1, (B,), generator=g)
        vlen[0] = T
        batches.append(dict(video=torch.randn(B, T, F, generator=g).half().float(), video_len=vlen, text=list(ids)))

    class Loader(list):
        dataset = list(range(B * (steps + n_warm)))

    largs = types.SimpleNamespace(max_tokens=Lt, mlm_prob=0.15, print_freq=10 ** 9, epochs=1, lr=3e-5, schedule="",
                                  fraction_warmup_steps=0.1, delayed_loss_check=delayed, packed_rows=bool(packed))
    model.train()
    model.training_graphs = bool(graphs)
    said = io.StringIO()  # the loop's own prints (MetricLogger lines; the message in front of its sys.exit(1) on a non-finite loss)
    try:
        with contextlib.redirect_stdout(said):
            P_main.train_one_epoch(model, Tok(), Loader(batches[:n_warm]), opt, model.device, 0, largs, 0.1)  # warm-up
            torch.cuda.synchronize()
            t0 = time.time()
            P_main.train_one_epoch(model, Tok(), Loader(batches[n_warm:]), opt, model.device, 0, largs, 0.1)
            torch.cuda.synchronize()
        dt = time.time() - t0
    except SystemExit as e:  # main.py:75-78 behaviour of the loop; here: a failed measurement WITH its message, not a silent exit
        tail = " | ".join(said.getvalue().strip().splitlines()[-3:])
        raise RuntimeError(f"main.train_one_epoch stopped the run with exit status {e.code}: {tail}") from None
    finally:
        model.training_graphs = False
        model.packed_rows = False
        model.__dict__.pop("_train_graphs", None)
    return {"value": B * steps / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "args.delayed_loss_check": bool(delayed), "model.training_graphs": bool(graphs), "args.packed_rows": bool(packed)}


CHILD_MARK = "FBL_BENCH_MEASURING"  # set in the environment of the process that measures (never read by the library)
PHASE_FILE = "FBL_BENCH_PHASE_FILE"  # where the measuring process leaves the name of the phase it is in (for the supervisor)
_T_START = time.time()
_PHASE = ["start"]


def _phase(name: str):
    """breadcrumb: which part of the run the measuring process is in (read by the supervisor if the process dies without a line)"""
    _PHASE[0] = name
    path = os.environ.get(PHASE_FILE)
    if path:
        try:
            with open(path, "w") as f:
                f.write(f"{name} (+{time.time() - _T_START:.1f} s)")
        except OSError:
            pass


def _report_exit_without_line():
    """atexit of the measuring process: an exit that came through the interpreter (sys.exit, an exception) says where it was;
    an exit from inside a native library (exit(), _exit()) never gets here -- the supervisor then only has the phase file"""
    if _PHASE[0] != "done":
        print(f"[bench] the interpreter is exiting before the line was printed; phase: {_PHASE[0]} (+{time.time() - _T_START:.1f} s)",
              file=sys.stderr, flush=True)


def _wants_supervisor(args) -> bool:
    """The plain single-GPU invocation (the driver's `python bench.py --gpus 1 ...`) measures in a child process.  Not when this
    process already is that child, a rank of a launcher, or runs under rocprofv3 (the profile must be the measuring process's own)."""
    if args.gpus != 1 or "WORLD_SIZE" in os.environ or os.environ.get(CHILD_MARK) or args.no_retry:
        return False
    return not any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ)


def _die_with_parent_fn():
    """A function for Popen's preexec_fn (runs in the child between fork and exec): prctl(PR_SET_PDEATHSIG, SIGKILL) -- no measuring
    process outlives its supervisor on the GPU.  libc is resolved HERE, in the parent: the child only makes the call."""
    import ctypes
    import signal

    try:
        prctl = ctypes.CDLL("libc.so.6", use_errno=True).prctl
    except Exception:  # noqa: BLE001  (not Linux / no libc by that name: the signal forwarding still covers SIGTERM / SIGINT)
        return None
    kill = int(signal.SIGKILL)
    return lambda: prctl(1, kill, 0, 0, 0) and None


def supervise_single_rank(cmd=None) -> int:
    """Run the measurement in a child process and pass its output through.  A child that ends unsuccessfully WITHOUT having printed
    its JSON line is run once more; the line of the second attempt carries `"attempts": 2` and how / where the first one ended
    (`_phase` breadcrumbs).  History: three default runs of round 5 ended without a line and without a word; the
    breadcrumbs of this supervisor located the exit (`main.train_one_epoch` stopping on a non-finite loss inside an auxiliary leg,
    after the replayed-step leg had left non-finite parameters) -- the legs are guarded now and the suspected cause is removed, the
    supervisor stays as the last line of defence for the one number the driver reads.  Termination requests are passed on and are
    never answered with a second attempt; the child dies with this process."""
    import signal
    import subprocess
    import tempfile
    import threading

    cmd = cmd or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    first_rc = None
    stopped = []
    die_with_me = _die_with_parent_fn()
    fd, phase_path = tempfile.mkstemp(prefix="fbl_bench_phase_", dir="/tmp")
    os.close(fd)
    try:
        for attempt in (1, 2):
            env = dict(os.environ, **{CHILD_MARK: str(attempt), PHASE_FILE: phase_path})
            if first_rc is not None:
                env["FBL_BENCH_FIRST_RC"] = str(first_rc)
                env["FBL_BENCH_FIRST_PHASE"] = first_phase
            p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, preexec_fn=die_with_me)
            # whoever stops this process stops the measurement: SIGTERM / SIGINT are passed on, and the child asked the kernel for a
            # SIGKILL of its own should this process disappear without a chance to do so (a time-out's SIGKILL)
            def pass_on(s_, f_, p=p):
                stopped.append(s_)
                p.send_signal(s_)

            old = {sg: signal.signal(sg, pass_on) for sg in (signal.SIGTERM, signal.SIGINT)} \
                if threading.current_thread() is threading.main_thread() else {}
            try:
                got_line = False
                for line in p.stdout:
                    got_line |= line.startswith("{") and '"metric"' in line
                    sys.stdout.write(line)
                    sys.stdout.flush()
                rc = p.wait()
            finally:
                for sg, h in old.items():
                    signal.signal(sg, h)
            if got_line or rc == 0 or attempt == 2 or stopped:  # (stopped: somebody asked THIS process to end -- no second attempt)
                return rc if rc >= 0 else 128 - rc
            first_rc = rc
            try:
                first_phase = open(phase_path).read().strip() or "unknown"
            except OSError:
                first_phase = "unknown"
            print(f"[bench] the measuring process ended with status {rc} before printing its line (last phase: {first_phase}); "
                  "running it once more", file=sys.stderr, flush=True)
    finally:
        try:
            os.unlink(phase_path)
        except OSError:
            pass
    return 1


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same
    environment torch.distributed.run would set, rendezvous on 127.0.0.1) and pass rank 0's JSON line through."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def measure_gemm_roofline(L, step_fn):
    """Instrumented replay: HIP events (torch's current stream == the launch stream) around every launch of the GEMM family
    of one step (fbl_gemm_bf16_nt, fbl_dense_adapter_down_fwd, fbl_adapter_down_fwd, fbl_adapter_up_resid_fwd,
    fbl_gemm_bf16_tn_acc, fbl_adapter_bwd_dw), on whichever stream the engine issues it."""
    import frozenbilm_amd.lib as lib

    recs = []
    orig = lib.gemm

    def timed(A, B, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        orig(A, B, **kw)
        e.record()
        A2 = A[0] if A.dim() == 3 else A
        B2 = B[0] if B.dim() == 3 else B
        M = kw.get("M") or A2.shape[0]
        N = kw.get("N") or B2.shape[0]
        K = kw.get("K") or A2.shape[1]  # (k-blocked A operands pass the true contraction length explicitly)
        nb = A.shape[0] if A.dim() == 3 else 1
        big = nb == 1 and kw.get("splitk", 1) <= 1 and _takes_gemm8(M, N, K)
        recs.append((s, e, 2.0 * M * N * K * nb, (M, N, K, nb), big))

    # the other entry points of the GEMM family: merged dense + adapter-down, stand-alone adapter-down, dW (A^T.B)
    orig_dad, orig_ad, orig_tn, orig_dw = lib.dense_adapter_down_fwd, lib.adapter_down_fwd, lib.gemm_tn_acc, lib.adapter_bwd_dw
    orig_up = lib.adapter_up_resid_fwd

    def bracket(fn, shape_of, big_ok=False):
        def wrapped(*a, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn(*a, **kw)
            e.record()
            M, N, K = shape_of(*a, **kw)
            recs.append((s, e, 2.0 * M * N * K, (M, N, K, 1), big_ok and _takes_gemm8(M, N, K)))
        return wrapped

    lib.gemm = timed
    lib.dense_adapter_down_fwd = bracket(orig_dad, lambda x, wm, *a, **kw: (x.shape[0], wm.shape[0], x.shape[1]), big_ok=True)
    lib.adapter_down_fwd = bracket(orig_ad, lambda x, wd, b, z, A=None, **kw: (x.shape[0], A or wd.shape[0], x.shape[1]))
    lib.adapter_up_resid_fwd = bracket(orig_up, lambda z, wu, bu, x, t, A=None, **kw: (x.shape[0], x.shape[1], A or wu.shape[1]))
    lib.gemm_tn_acc = bracket(orig_tn, lambda A_, B_, o, ws, M=None, N=None, K=None, **kw:
                              (M or A_.shape[1], N or B_.shape[1], K or min(A_.shape[0], B_.shape[0])))

    def dw_shape(groups, A):  # grouped adapter gradients: two [H x A] products over N rows per segment, as one "batch"
        dy = groups[0][0][0][0]
        return (dy.shape[1], A, dy.shape[0] * 2 * sum(len(g[0]) for g in groups))

    lib.adapter_bwd_dw = bracket(orig_dw, dw_shape)
    try:
        step_fn()
        recs.clear()
        step_fn()
        torch.cuda.synchronize()
    finally:
        lib.gemm, lib.dense_adapter_down_fwd, lib.adapter_down_fwd, lib.gemm_tn_acc = orig, orig_dad, orig_ad, orig_tn
        lib.adapter_bwd_dw, lib.adapter_up_resid_fwd = orig_dw, orig_up

    def summary(rs):
        ms = sum(s.elapsed_time(e) for s, e, *_ in rs)
        fl = sum(r[2] for r in rs)
        return ms, fl, (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)

    dom = [r for r in recs if r[4]]
    dom_ms, dom_fl, dom_ach = summary(dom)
    tot_ms, tot_fl, ach = summary(recs)
    by_shape = {}
    for s, e, f, shp, big in recs:
        d = by_shape.setdefault(shp, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += s.elapsed_time(e)
        d[2] += f
    top = sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:8]
    # HBM-side bytes of the same kernels over the same step come from separate rocprofv3 --pmc passes (FETCH_SIZE x2 for
    # gfx950 + WRITE_SIZE; tools/pmc_bench.sh -> profiles/*_traffic.json): PMC passes cannot run inside this process, so
    # the newest COMMITTED summary is read back ("source" says so).
    traffic, traffic_src, fam_gb = None, None, None
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    cands = sorted(f for f in os.listdir(pdir) if f.endswith("_traffic.json")) if os.path.isdir(pdir) else []
    if cands and recs:
        with open(os.path.join(pdir, cands[-1])) as f:
            tj = json.load(f)
        fam_gb = sum(v["GB_per_step"] for k, v in tj.items() if k.startswith("gemm"))
        d8 = [v for k, v in tj.items() if k.startswith("gemm8_kernel")]
        if d8:
            traffic = sum(v["GB_per_step"] for v in d8) * 1e9 / sum(v["launches_per_step"] for v in d8)
            traffic_src = ("committed: profiles/%s -- %.1f GB/step over %d gemm8_kernel launches (rocprofv3 --pmc FETCH_SIZE, "
                           "WRITE_SIZE passes over this command at the commit that wrote the file), bytes per launch; not "
                           "re-measured by this run" % (cands[-1], sum(v["GB_per_step"] for v in d8),
                                                        round(sum(v["launches_per_step"] for v in d8))))
    return {"bound": "mfma", "kernel": "gemm8_kernel (8-phase 256x256 / 224x256 bf16 MFMA GEMM, all epilogues; a split launch "
                                       "includes its 64x128-tile remainder)",
            "achieved": dom_ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": dom_ach / PEAK_BF16_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": len(dom),
            "avg_launch_us": dom_ms * 1e3 / max(len(dom), 1), "kernel_ms_per_step": dom_ms,
            "algorithmic_tflops_per_step": dom_fl / 1e12,
            "family": {"what": "every launch of the GEMM family (gemm8_kernel, gemm_bf16_nt_kernel incl. split-K / batched "
                               "position-table products, the grouped adapter-gradient kernel adapter_dw_kernel)", "achieved": ach,
                       "frac": ach / PEAK_BF16_TFLOPS, "launches_per_step": len(recs), "gemm_ms_per_step": tot_ms,
                       "hbm_gb_per_step_committed": fam_gb},
            "gemm_ms_per_step": tot_ms, "executed_gemm_tflops_per_step": tot_fl / 1e12,
            "top_shapes_MNKb_count_ms_tflops": [[list(k), v[0], round(v[1], 3), round(v[2] / (v[1] * 1e-3) / 1e12, 1)]
                                                for k, v in top]}


def measure_traffic_live(args):
    """roofline.traffic measured by THIS run: two child runs of this script under `rocprofv3 --pmc FETCH_SIZE` and `--pmc
    WRITE_SIZE` (separate passes, kernel-trace only -- never combined with other trace domains; MI355X_MICROARCH.md, HBM section:
    FETCH_SIZE in KiB, doubled on gfx950), a few steps each, summed over the gemm8_kernel launches.  Returns None -- the caller keeps
    the committed figure -- when rocprofv3 is missing, a pass fails or takes longer than its time-out."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    here = os.path.abspath(__file__)
    steps = 3  # instrumented steps per pass: 2 pre-warm + 1 timed
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    launches = 0
    whole = 0.0
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            env = dict(os.environ, TMPDIR="/tmp", **{CHILD_MARK: "pmc"})
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(td, c)
                cmd = [exe, "--pmc", c, "--kernel-trace", "-d", out, "-o", "p", "--", sys.executable, here, "--steps", "1", "--warmup",
                       "0", "--prewarm", "2", "--batch", str(args.batch), "--text-len", str(args.text_len), "--layers", str(args.layers),
                       "--no-cpu-baseline", "--no-roofline", "--no-extras", "--no-traffic"]
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
                dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("_results.db")]
                if r.returncode != 0 or not dbs:
                    return None
                cur = sqlite3.connect(dbs[0]).cursor()
                cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
                ix = {k: i for i, k in enumerate(cols)}
                n = 0
                for row in cur.execute("select * from counters_collection").fetchall():
                    if row[ix["counter_name"]] != c:
                        continue
                    byts = float(row[ix["value"]]) * 1024 * (2 if c == "FETCH_SIZE" else 1)
                    name = str(row[ix.get("kernel_name", ix.get("name", 0))])
                    if "at::native" not in name and "rocclr" not in name:  # the library's own kernels: the child's model
                        whole += byts                                      # construction (ATen fills / copies) stays out
                    if "gemm8_kernel" in name:
                        tot[c] += byts
                        n += 1
                launches = n if c == "FETCH_SIZE" else launches
    except Exception:  # noqa: BLE001  (time-out, missing tables, ...): keep the committed figure
        return None
    if launches == 0:
        return None
    return {"traffic": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / launches,
            "traffic_source": "measured by this run: two child runs of this script under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                              "(kernel-trace only), %d steps each; FETCH_SIZE x2 for gfx950; bytes per gemm8_kernel launch over %d "
                              "launches; all kernels of the library %.1f GB per step" % (steps, launches, whole / steps / 1e9),
            "traffic_gemm8_gb_per_step": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / steps / 1e9,
            "traffic_library_kernels_gb_per_step": whole / steps / 1e9}


def _takes_gemm8(M, N, K):
    """launches the dispatcher of csrc/gemm.hip gives to the 8-phase kernel (the library's own host-side query)"""
    import frozenbilm_amd.lib as lib

    return lib.gemm_plan(M, N, K) == 8


def measure_cpu_baseline(model, cfg, T, F, Lt, fwd_only):
    """CPU oracle (fp32 torch port of the reference path) on the host cores, SURVEY.md section 8d protocol: B=2 sequences
    of the benchmark shape, one warm-up pass, then timed passes (2, or 1 when a pass takes longer than 20 s so that the
    default run stays within minutes), torch threads = physical cores."""
    from oracle import deberta_oracle as O

    try:
        import psutil

        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(int(phys))
    ocfg = O.OracleConfig(num_hidden_layers=cfg.num_hidden_layers)
    P = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items() if "position_ids" not in k}
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k) and not fwd_only)
    Bc = 2
    cb = synth_batch(Bc, T, F, Lt, cfg.vocab_size, seed=99, device="cpu")

    def one_pass():
        for v in P.values():
            v.grad = None
        t = time.time()
        with torch.set_grad_enabled(not fwd_only):
            out = O.forward(P, ocfg, **cb)
            if not fwd_only:
                out["loss"].backward()
        return time.time() - t

    warm = one_pass()
    n_timed = 2 if warm < 20.0 else 1
    times = [one_pass() for _ in range(n_timed)]
    dt = sum(times) / len(times)
    torch.set_num_threads(prev_threads)
    return {"value": Bc / dt, "unit": "samples/s", "cores": int(phys), "kind": "port", "loadavg": os.getloadavg()[0],
            "logical_cpus": os.cpu_count(),
            "sample": f"B={Bc} sequences (T=10, L={Lt}, S={T + Lt}) {'forward' if fwd_only else 'fwd+bwd'} through the fp32 CPU "
                      f"oracle, eval-mode math (no dropout): 1 warm-up pass ({warm:.1f} s) + {n_timed} timed "
                      f"({', '.join('%.1f' % x for x in times)} s), torch threads={int(phys)} (physical cores)"}


def measure_cpu_baseline_cfg1():
    """BASELINE configs[0] -- "BERT-base no-adapter, 4 synthetic videos (T=10x768 CLIP feats, L_text=64), MLM forward on CPU
    reference path" -- timed on THIS box's host cores through the oracle restatement of model/bert.py:792-872 (oracle/bert_oracle.py,
    pinned by golden G8 at these dimensions): SURVEY 8d inputs (B=4, video ~ N(0,1) [4,10,768], ids ~ U[1000,30522) [4,64], mask all
    ones, seeded N(0,0.02) weights), forward only, 2 warm-up + 5 timed iterations, torch threads = physical cores."""
    from oracle import bert_oracle as BO

    try:
        import psutil

        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    prev = torch.get_num_threads()
    cfg = BO.BertOracleConfig()  # BertConfig() defaults: 12 layers, H=768, 12 heads, I=3072, vocab 30522; features_dim 768
    P = BO.synth_params(cfg, seed=0, std=0.02)
    g = torch.Generator().manual_seed(1)
    Bc, T, Lt = 4, 10, 64
    video = torch.randn(Bc, T, cfg.features_dim, generator=g)
    ids = torch.randint(1000, cfg.vocab_size, (Bc, Lt), generator=g)
    am, vm = torch.ones(Bc, Lt, dtype=torch.long), torch.ones(Bc, T, dtype=torch.long)
    runs = {}
    try:
        # a 0.1 s forward does not scale to 128 cores (and the box is shared: see loadavg): timed with all physical cores AND
        # with 16 threads, the faster one is reported -- `cores` says which
        for nthr in sorted({int(phys), min(int(phys), 16)}, reverse=True):
            torch.set_num_threads(nthr)
            times = []
            with torch.no_grad():
                for i in range(7):
                    t = time.time()
                    out = BO.forward(cfg, P, ids, am, video, vm)
                    assert out["logits"].shape == (Bc, T + Lt, cfg.vocab_size)
                    if i >= 2:
                        times.append(time.time() - t)
            runs[nthr] = sum(times) / len(times)
    finally:
        torch.set_num_threads(prev)
    phys, dt = min(runs.items(), key=lambda kv: kv[1])
    return {"value": Bc / dt, "unit": "samples/s", "cores": int(phys), "kind": "port", "loadavg": os.getloadavg()[0],
            "logical_cpus": os.cpu_count(), "ms_per_forward": dt * 1e3,
            "sample": f"BASELINE configs[0]: BERT-base (12L, H=768, vocab 30522) + linear_video, B={Bc}, T=10x768, L={Lt} (S={T + Lt}), MLM "
                      f"forward through the fp32 CPU oracle (oracle/bert_oracle.py): 2 warm-up + {len(times)} timed iterations per thread "
                      f"count, ms per forward by torch threads: {({k: round(v * 1e3, 1) for k, v in runs.items()})}, reported: "
                      f"{int(phys)} threads; survey container (8 vCPU): 27.8 samples/s"}


if __name__ == "__main__":
    main()
