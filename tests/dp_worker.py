"""One rank of the two-rank data-parallel check of the HIP engine (launched by tests/test_gpu_model.py).

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment; argv[1] = output file (rank 0 writes it),
    argv[2] = GradReducer.overlap mode (optional), argv[3] = "tiny" (default), "tiny+packed" (model.packed_rows) or "xl4": the true xlarge dimensions
    (H = 1536, 24 heads, I = 6144, 192-wide adapters) with 4 layers, adapter gradients leaving in groups of 4 so that the
    stage buckets become final out of order under a real collective.

Backend: nccl (= RCCL) with one GPU per rank when the box has at least WORLD_SIZE GPUs, otherwise gloo with every rank on
cuda:0 -- the same engine / GradReducer code path either way (bucketed async all-reduce of the flat trainable gradient
buffer launched from inside the backward pipeline, 1/world scaling in finish()).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    config = sys.argv[3] if len(sys.argv) > 3 else "tiny"
    xl = config == "xl4"
    multi = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if multi else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if multi else "gloo", init_method="env://", world_size=world, rank=rank)

    from frozenbilm_amd.parallel import GradReducer
    from oracle import deberta_oracle as O
    from tests.golden.make_goldens import _tiny_cfg, synth_batch
    from tests.test_gpu_model import build

    if xl:
        cfg = O.OracleConfig()
        cfg.num_hidden_layers, cfg.vocab_size = 4, 4096
    else:
        cfg = _tiny_cfg()
    # eval mode: dropout off, gradients on; xl4: 4 adapters per gradient launch (read when the engine is built)
    m = build(cfg, O.synth_params(cfg, seed=41, std=0.02 if xl else 0.05, ln_jitter=0.1), engine_options={"dw_group": 4} if xl else None)
    m.to(dev)
    if config == "tiny+packed":  # model.packed_rows: every rank packs its own ragged shard (different row counts per rank)
        m.packed_rows = True
    if config == "tiny+graphs_mixed":
        # model.training_graphs on both ranks, but rank 1's capture budget is already used up: it switches the feature off by
        # itself and serves every step eagerly while rank 0 replays its graphs -- the two must keep issuing the SAME collectives
        # (one [0, n) all-reduce per step: the reducer stays in "after" mode on the rank that fell back)
        import frozenbilm_amd.train_graph as TG

        m.train()  # (dropout live: this configuration only compares the ranks with each other)
        m.training_graphs = True
        if rank == 1:
            TG.MAX_CAPTURES = 0
    # small buckets: several collectives in flight during backward; argv[2] = where they are launched (GradReducer.overlap)
    red = GradReducer.attach(m, min_bucket_elems=1 << 10, overlap=sys.argv[2] if len(sys.argv) > 2 else None)
    per = 2
    batch = synth_batch(cfg, B=per * world, L=96 if xl else 60, seed=9)
    mine = {k: v[rank * per:(rank + 1) * per].to(dev) for k, v in batch.items()}
    losses = []
    for _ in range(2):  # two steps: the reducer's cursor / bucket bookkeeping must reset between them
        m.zero_grad(set_to_none=False)
        out = m(**mine)
        out.loss.backward()  # the engine's backward launches the bucket collectives and joins them (GradReducer.finish)
        losses.append(out.loss.item())
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.requires_grad}
    n_coll = len(red.last_launched)
    pattern_same = True
    if config == "tiny+graphs_mixed":
        # both ranks: one collective over the whole flat buffer per step, whichever way the step was served
        mine_pat = torch.tensor([len(red.last_launched), red.last_launched[0][0], red.last_launched[-1][1], red.n_collectives],
                                dtype=torch.int64, device=dev)
        both = [torch.zeros_like(mine_pat) for _ in range(world)]
        dist.all_gather(both, mine_pat)
        pattern_same = all(torch.equal(both[0], b_) for b_ in both) and int(mine_pat[0]) == 1
        served = "graph" if m.__dict__.get("_train_graphs") else "eager"
        print(f"[dp_worker] rank {rank}: served by {served}, collectives {both}", flush=True)
        assert (served == "graph") == (rank == 0), served
    # a third step with the loops' loss bookkeeping: the logged loss rides in front of the first gradient bucket (no collective
    # of its own), and the host reads the rank-averaged value between backward and the update (loops.LossLog.begin / check)
    from frozenbilm_amd.loops import LossLog

    class _Meter:
        loss_log = None
        vals = None

        def log(self, **kw):
            self.vals = kw

    meter = _Meter()
    log = LossLog(meter, "mlm_loss", reducer=red)
    m.zero_grad(set_to_none=False)
    out = m(**mine)
    c0 = red.n_collectives
    log.begin(out.loss)
    out.loss.backward()
    log.check()
    extra = red.n_collectives - c0 - len(red.last_launched)
    mean_loss = torch.tensor([out.loss.item()], device=dev)
    dist.all_reduce(mean_loss)
    loss_rides = (meter.vals is not None and abs(meter.vals["loss"] - mean_loss.item() / world) < 1e-6
                  and abs(meter.vals["mlm_loss"] - meter.vals["loss"]) < 1e-12)
    spans = sorted(red.last_launched)
    covers = spans[0][0] == 0 and spans[-1][1] == red.flat_grad.numel() and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    print(f"[dp_worker] rank {rank}/{world}: backend {dist.get_backend()}, RCCL saw {red.rccl_ranks} ranks, overlap "
          f"{red.overlap}, {n_coll} collectives per step, launch order {red.last_launched[:6]}...", flush=True)
    # every rank must hold the same reduced gradients
    flat = torch.cat([g.reshape(-1) for g in grads.values()]).to(dev)
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(ref, flat))
    ok = torch.tensor([1.0 if same else 0.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        torch.save({"grads": grads, "losses": losses, "backend": dist.get_backend(), "collectives": n_coll,
                    "rccl_ranks": red.rccl_ranks, "overlap": red.overlap, "covers": covers, "launch_order": list(red.last_launched),
                    "ranks_agree": bool(ok.item() == 1.0), "world": world, "loss_rides": bool(loss_rides),
                    "extra_collectives_for_the_loss": int(extra), "pattern_same": bool(pattern_same)}, sys.argv[1])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
