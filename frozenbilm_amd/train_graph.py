"""The MLM training step (model forward + backward, main.py:59-84) as two replayable launch graphs (opt-in).

One training step of the engine is ~1450 kernel launches; issued one by one they cost the host ~19 ms (bench.py `host`).  At
one GPU that hides behind ~47 ms of GPU work; with eight processes per node, or once the kernels get faster, it does not.
`model.training_graphs = True` makes `model(**batch)` -- training mode, `labels` given, nothing else requested -- capture the
launch sequence of the forward and of the backward once per batch shape (hipGraph through `torch.cuda.CUDAGraph`, both graphs in
ONE private memory pool: the backward reads the activations the forward left in it) and replay them afterwards:

    out = model(**batch)        # copies the batch into the graph's static inputs, replays the forward graph
    out.loss.backward()         # replays the backward graph: gradients land in p.grad (views of the flat buffer)
    optimizer.step(...)         # stays eager: two launches whose arguments (lr, Adam step) change every step

What had to move for this:
  * dropout seeds are per-site constants + the step's position in the mask stream; eager launches add the two on the host,
    captured launches keep the constants as (frozen) kernel arguments and read the position from a device word that is
    rewritten before every replay (include/fbl.h "Dropout seeds"): the same sums, hence the same masks;
  * the list of labelled rows has a fixed capacity per graph (next multiple of 256): the host still counts the labels at the
    start of the step -- on an input, as the eager path does -- and pads the list with an unlabelled row, whose loss term and
    gradient are exactly zero (fbl_ce_fwd / fbl_ce_bwd_rows ignore labels < 0);
  * the gradient exchange of data-parallel runs leaves after the backward replay (`GradReducer.finish`: the "after"
    placement), not from inside it.
Replayed and eager steps run the same kernels on the same inputs with the same seeds: they agree bit for bit
(tests/test_gpu_model.py::test_graphed_training_step_equals_the_eager_step).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import lib as L
from .engine import Run

F32 = torch.float32
ROW_BUCKET = 256  # capacity granularity of the labelled-row list
MAX_GRAPHS = 4    # captured (batch shape, row capacity) configurations kept per model (each owns one step's activations)
MAX_CAPTURES = 12  # captures per model before the feature switches itself off with a warning (a shape-static loop needs 1-3)


class _Replay(torch.autograd.Function):
    """autograd node of a graphed step: backward replays the backward graph"""

    @staticmethod
    def forward(ctx, step, loss_t, *params):
        ctx.step = step
        ctx.set_materialize_grads(False)
        return loss_t.detach().clone()

    @staticmethod
    def backward(ctx, gloss):
        step = ctx.step
        if gloss is not None:
            step.run_backward(gloss)
        return (None, None) + tuple(None for _ in step.eng.order)


class GraphedStep:
    def __init__(self, model, eng, feed: Dict[str, torch.Tensor], r_cap: int):
        self.model, self.eng = model, eng
        self.r_cap = r_cap
        self.static = {k: v.clone() for k, v in feed.items()}
        dev = eng.dev
        self.rows = torch.zeros(r_cap, dtype=torch.int64, device=dev)
        self.gloss = torch.ones((), dtype=F32, device=dev)
        self.seed_word = torch.zeros(1, dtype=torch.int64, device=dev)
        self.pool = torch.cuda.graph_pool_handle()
        self.g_fwd, self.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.run: Optional[Run] = None
        self.loss_t = None
        self.pending_backward = False

    # ------------------------------------------------------------------ the two launch sequences
    def _forward_launches(self):
        eng, m, st = self.eng, self.model, self.static
        cfg = eng.cfg
        input_ids = st["input_ids"]
        B, Lt = input_ids.shape
        video = st.get("video")
        T = video.shape[1] if video is not None else 0
        run = Run(B=B, S=T + Lt, T=T, Lt=Lt, train=True, save=True, seed_base=0, p_hid=cfg.hidden_dropout_prob,
                  p_att=cfg.attention_probs_dropout_prob, p_ad=m.adapter_dropout)
        run.seed_word = self.seed_word
        am = st["attention_mask"]
        if T:
            vm = st.get("video_mask")
            if vm is None:
                vm = torch.ones(video.shape[:2], device=eng.dev, dtype=am.dtype)
            mask = torch.cat([vm.to(am.dtype), am], 1)
            labels = torch.cat([torch.full((B, T), -100, dtype=torch.long, device=eng.dev), st["labels"]], 1)
        else:
            mask, labels = am, st["labels"]
        run.mask = mask.to(torch.int32).contiguous().view(-1)
        run.labels = labels.contiguous().view(-1)
        run.rows = self.rows
        eng._refresh_if_stale(True)
        with L.seed_word(self.seed_word):
            _, loss_t = eng._forward(run, input_ids, video, False, False)
        return run, loss_t

    def capture(self):
        eng = self.eng
        red, eng.reducer = eng.reducer, None  # the gradient exchange stays outside the graphs (run_backward)
        # The cyclic garbage collector must not run while a stream is capturing: an object it frees may own a HIP event (the
        # events of an earlier pass's side-stream joins, a dropped engine's) and destroying an event during a capture aborts
        # the process -- seen twice in round 6 as "Fatal Python error: Aborted ... Garbage-collecting" inside this function,
        # with whatever allocation happened to cross the collector's threshold.  torch.cuda.graph collects on entry; nothing
        # may collect until the capture has ended.
        import gc

        gc_was_on = gc.isenabled()
        try:
            gc.collect()
            gc.disable()
            with torch.cuda.graph(self.g_fwd, pool=self.pool):
                self.run, self.loss_t = self._forward_launches()
            eng.attach_grads()
            with torch.cuda.graph(self.g_bwd, pool=self.pool):
                eng.backward(self.run, self.gloss, attach=False)
        finally:
            if gc_was_on:
                gc.enable()
            eng.reducer = red
        return self

    # ------------------------------------------------------------------ per step
    def run_forward(self, feed, rows_padded, seed_word_value: int):
        for k, v in feed.items():
            self.static[k].copy_(v, non_blocking=True)
        self.rows.copy_(rows_padded, non_blocking=True)
        self.seed_word.fill_(seed_word_value)
        self.g_fwd.replay()
        self.run.logits_pending = True  # (the [N, V] logits of THIS replay are filled on access, by an eager launch)
        self.pending_backward = True
        return _Replay.apply(self, self.loss_t, *[self.eng.named[n] for n in self.eng.order])

    def run_backward(self, gloss):
        eng = self.eng
        self.gloss.copy_(gloss.detach().to(F32).reshape(()))
        named, G = eng.named, eng.G
        if not all(named[n].grad is not None and named[n].grad.data_ptr() == G[n].data_ptr() for n in eng.order):
            eng.attach_grads()  # (skipped while every p.grad still is its view of the flat buffer -- zero_grad(set_to_none=True)
            #                      or a foreign / dropped .grad on ANY parameter brings the full pass back; ~300 pointer compares)
        self.g_bwd.replay()
        self.pending_backward = False
        if eng.reducer is not None:
            eng.reducer.finish()


def restore_reducer_overlap(eng):
    """The CALLER switched `model.training_graphs` off again (on every rank, like any other change of the training loop): the
    reducer goes back to the overlap placement it had before the graphs forced "after" mode -- otherwise the data-parallel
    overlap stays lost for the rest of the run.  Not used when the feature switched ITSELF off on this rank (capture budget)."""
    red = getattr(eng, "reducer", None)
    prev = None if red is None else red.__dict__.pop("_overlap_before_graphs", None)
    if prev is not None:
        red.overlap = prev


def graphed_forward(model, eng, input_ids, attention_mask, video, video_mask, labels):
    """`model.forward` of a training step through the captured graphs (captured on first use of a shape).  Returns the loss
    (connected to the trainable parameters for `.backward()`) and the Run (for `logits` on demand), or None when this call
    cannot be served from a graph (the caller takes the eager path)."""
    if labels is None or input_ids is None or not torch.is_grad_enabled():
        return None
    # Data parallel: a rank served by a graph exchanges ONE [0, n) collective after the replay (run_backward); a rank that
    # falls back to the eager step (no labelled row in its shard, a failed capture) must issue the same pattern, or the ranks'
    # collectives no longer match in count and size.  While training_graphs is on the reducer therefore runs in "after" mode.
    if eng.reducer is not None and eng.reducer.overlap != "after":
        eng.reducer.__dict__.setdefault("_overlap_before_graphs", eng.reducer.overlap)  # restored by restore_reducer_overlap()
        eng.reducer.overlap = "after"
    dev = eng.dev
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    feed = {"input_ids": input_ids.contiguous(), "attention_mask": attention_mask.contiguous(), "labels": labels.contiguous()}
    use_video = bool(eng.F) and video is not None
    B, Lt = input_ids.shape
    T = video.shape[1] if use_video else 0
    if T + Lt > eng.cfg.max_position_embeddings:
        return None  # (the eager path raises the reference's error)
    if use_video:
        feed["video"] = video.contiguous()
        if video_mask is not None:
            feed["video_mask"] = video_mask.contiguous()
    # labelled rows: a function of the input, counted before anything is queued (the eager path does the same)
    full = labels if not T else torch.cat([torch.full((B, T), -100, dtype=torch.long, device=dev), labels], 1)
    flat = full.reshape(-1)
    rows = torch.nonzero(flat != -100).view(-1)
    R = rows.numel()
    N = flat.numel()
    if R == 0:
        return None
    r_cap = min(N, -(-R // ROW_BUCKET) * ROW_BUCKET)
    if r_cap > R:  # pad with the first unlabelled row (found on the device: no second synchronisation; R < N: one exists)
        free = torch.argmax((flat == -100).to(torch.int8)).view(1)
        rows = torch.cat([rows, free.expand(r_cap - R)])
    key = (id(eng), B, Lt, T, "video_mask" in feed, r_cap, tuple(sorted((k, str(v.dtype)) for k, v in feed.items())))
    cache = model.__dict__.setdefault("_train_graphs", {})
    step = cache.pop(key, None)
    seed0 = model.step_seed
    if step is None:
        for k in [k for k in cache if k[0] != id(eng)]:
            del cache[k]
        while len(cache) >= MAX_GRAPHS:
            del cache[next(iter(cache))]
        # a loop whose batch shapes keep changing (text padded to the longest sample, label counts crossing a row bucket) would
        # capture -- one eager warm-up step plus two captures, ~10 GB of pooled activations each -- and evict continuously
        # (counted: DISTINCT configurations of the current engine -- a rebuilt engine (load_state_dict, .to(), new answer
        #  embeddings) or a cache the caller cleared starts over, and re-capturing a configuration seen before does not count)
        seen = model.__dict__.setdefault("_train_graph_keys", (id(eng), set()))
        if seen[0] != id(eng):
            seen = model.__dict__["_train_graph_keys"] = (id(eng), set())
        seen[1].add(key[1:])
        n_cap = len(seen[1])
        model.__dict__["_train_graph_captures"] = n_cap
        if n_cap > MAX_CAPTURES:
            import warnings

            warnings.warn(f"model.training_graphs: {n_cap} distinct (batch shape, label capacity) configurations seen -- the loop "
                          "is not shape-static (pad the text to a fixed length / bucket it); staying on the eager path")
            model.training_graphs = False
            # (the reducer STAYS in "after" mode: the other ranks may still be served by their graphs -- row capacities differ per
            #  rank, so the budgets run out at different steps -- and every rank must keep issuing the same one collective)
            model.__dict__["_train_graphs_auto_off"] = True
            cache.clear()
            return None
        # one eager step of this shape first: lazy initialisations (kernel attributes, workspaces, allocator warm-up) happen
        # outside the capture.  It leaves no trace: the caller's accumulated gradients, the STATE of every p.grad (None stays
        # None: the next attach_grads then zero-fills as it would have; a view of the flat buffer gets its old contents back)
        # and the mask-stream position are restored.
        saved = eng.flat_grad.clone()
        grads_before = [eng.named[n].grad for n in eng.order]
        red, eng.reducer = eng.reducer, None  # (no collectives: ranks may capture at different steps -- their row capacities differ)
        try:
            res = eng.run(feed["input_ids"], feed["attention_mask"], feed.get("video"), feed.get("video_mask"), feed["labels"],
                          False, False)
            res["loss"].backward()
        finally:
            eng.reducer = red
        eng.flat_grad.copy_(saved)
        for n, g in zip(eng.order, grads_before):
            eng.named[n].grad = g
        del res, saved, grads_before
        torch.cuda.synchronize(dev)
        model.step_seed = seed0
        step = GraphedStep(model, eng, feed, r_cap)
        try:
            step.capture()
        except Exception as e:  # noqa: BLE001
            import warnings

            torch.cuda.synchronize(dev)
            warnings.warn(f"training graph capture failed ({type(e).__name__}: {e}); staying on the eager path")
            model.training_graphs = False
            return None
    cache[key] = step
    if step.pending_backward:  # the graph owns ONE set of activations
        raise RuntimeError("graphed training step: forward called again before backward() of the previous output; "
                           "set model.training_graphs = False for loops that keep several forwards alive")
    model.step_seed = seed0 + 1
    loss = step.run_forward(feed, rows, eng.seed_word_value())
    return loss, step.run
