"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (parallel.GradReducer) and the dist helpers."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frozenbilm_amd.parallel import GradReducer
        from frozenbilm_amd.util import dist as D

        n = 5000
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(n, generator=g)
        mine = flat.clone()
        ends = {"head": 16, "layer2": 1800, "layer1": 3600, "conv": 3616, "layer0": 4900, "relln": 4916, "emb": n}
        red = GradReducer(flat, ends, min_bucket_elems=64, overlap="backward")
        for key in ("head", "layer2", "layer1", "conv", "layer0", "relln", "emb"):
            red.ready(key)
        red.finish()
        others = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        expect = sum(others) / world
        ok = torch.allclose(flat, expect, atol=1e-6)
        # buckets: contiguous, ordered, covering; tiny ones coalesced into the next
        spans = red.last_launched
        ok &= spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        ok &= all(e - s >= 64 for s, e in spans) and len(spans) >= 3
        # second step reuses the reducer; stages become final OUT of order (the repeated last layer's adapters leave with a
        # later group launch than the layers behind it): runs of adjacent ready stages leave, nothing twice, everything once
        flat.copy_(mine)
        for key in ("head", "layer1", "conv", "layer2", "relln", "layer0", "emb"):
            red.ready(key)
        red.finish()
        ok &= torch.allclose(flat, expect, atol=1e-6)
        sp = sorted(red.last_launched)
        ok &= sp[0][0] == 0 and sp[-1][1] == n and all(a[1] == b[0] for a, b in zip(sp, sp[1:]))
        # "attention_windows": ready stages are held until the engine opens a window; "after": one collective at the end
        for mode in ("attention_windows", "after"):
            flat.copy_(mine)
            rw = GradReducer(flat, ends, min_bucket_elems=64, overlap=mode)
            rw.ready("head"); rw.ready("layer2")
            ok &= not rw.launched
            rw.window()
            ok &= (rw.launched == [(0, 1800)]) if mode == "attention_windows" else (not rw.launched)
            for key in ("layer1", "conv", "layer0", "relln", "emb"):
                rw.ready(key)
            rw.finish()
            ok &= torch.allclose(flat, expect, atol=1e-6)
            ok &= sorted(rw.last_launched) == ([(0, 1800), (1800, n)] if mode == "attention_windows" else [(0, n)])
        # several backward passes feeding one step (mc.py): nothing leaves before the context closes
        flat.copy_(mine)
        with red.accumulate():
            for _ in range(3):
                for key in ("head", "layer2", "layer1", "conv", "layer0", "relln", "emb"):
                    red.ready(key)
                red.finish()
                ok &= torch.equal(flat, mine) and not red.pending
        ok &= torch.allclose(flat, expect, atol=1e-6) and red.last_launched == [(0, n)]
        # the step's logged loss rides in front of the first bucket (util/dist.py:89-113 is a collective of its own
        # in the reference): same number of collectives as without it, averaged value on every rank, gradients unchanged
        from frozenbilm_amd.parallel import SCALAR_SLOT

        full = torch.zeros(SCALAR_SLOT + n)
        fl = full[SCALAR_SLOT:]
        for mode in ("backward", "attention_windows", "after"):
            fl.copy_(mine)
            rs = GradReducer(fl, ends, min_bucket_elems=64, overlap=mode, flat_full=full)
            plain = GradReducer(mine.clone(), ends, min_bucket_elems=64, overlap=mode)
            for r_ in (rs, plain):
                if r_ is rs:
                    r_.stage_scalars(torch.tensor([3.0 + rank]))
                    ok &= r_.take_scalars() is None  # nothing exchanged yet
                for i_, key in enumerate(("head", "layer2", "layer1", "conv", "layer0", "relln", "emb")):
                    r_.ready(key)
                    if i_ == 1:
                        r_.window()
                r_.finish()
            ok &= rs.n_collectives == plain.n_collectives == len(rs.last_launched)  # the loss added none
            got = rs.take_scalars()
            ok &= got is not None and abs(got[0] - 3.5) < 1e-6 and rs.take_scalars() is None
            ok &= torch.allclose(fl, expect, atol=1e-6)
            # a second step without a staged loss: plain exchange, the slot is left alone
            fl.copy_(mine)
            for key in ("head", "layer2", "layer1", "conv", "layer0", "relln", "emb"):
                rs.ready(key)
            rs.finish()
            ok &= torch.allclose(fl, expect, atol=1e-6) and rs.take_scalars() is None
        # DP equivalence: all-reduced per-rank gradients == single-process gradients of the mean of the
        # per-rank mean losses, on the oracle model with the flat layout / bucket order of the engine
        from oracle import deberta_oracle as O
        from oracle.model_wrapper import OracleModel
        from frozenbilm_amd.model.config import DebertaV2Config
        from frozenbilm_amd.model.deberta import flat_order
        from tests.golden.make_goldens import _tiny_cfg, synth_batch

        cfg = _tiny_cfg(num_hidden_layers=2)
        P = O.synth_params(cfg, seed=3, std=0.05, ln_jitter=0.1)
        batch = synth_batch(cfg, B=4, L=12, seed=5)

        def grads(sl):
            m = OracleModel(_tiny_cfg(num_hidden_layers=2), P)
            out = m(**{k: v[sl] for k, v in batch.items()})
            out["loss"].backward()
            return {k: p.grad for k, p in m.named_ref_parameters().items() if p.requires_grad}, out["loss"].item()

        c = DebertaV2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=2,
                            num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                            max_position_embeddings=cfg.max_position_embeddings, position_buckets=cfg.position_buckets)
        mine_g, _ = grads(slice(2 * rank, 2 * rank + 2))
        order = flat_order(c, list(mine_g))
        offs, tot = {}, 0
        for k in order:
            offs[k] = tot
            tot += (mine_g[k].numel() + 7) // 8 * 8
        fg = torch.zeros(tot)
        for k in order:
            fg[offs[k]: offs[k] + mine_g[k].numel()] = mine_g[k].flatten()
        r2 = GradReducer(fg, {"emb": tot}, min_bucket_elems=1)
        r2.ready("emb"); r2.finish()
        g0, _ = grads(slice(0, 2)); g1, _ = grads(slice(2, 4))
        for k in order:
            want = 0.5 * (g0[k] + g1[k])
            got = fg[offs[k]: offs[k] + want.numel()].view_as(want)
            ok &= torch.allclose(got, want, rtol=1e-5, atol=1e-7)
        rd = D.reduce_dict({"b": torch.tensor(float(rank)), "a": torch.tensor(10.0 + rank)})
        ok &= abs(rd["a"].item() - 10.5) < 1e-6 and abs(rd["b"].item() - 0.5) < 1e-6
        gathered = D.all_gather({"rank": rank, "payload": list(range(rank + 3))})
        ok &= [x["rank"] for x in gathered] == list(range(world)) and len(gathered[1]["payload"]) == 4
        ok &= D.get_world_size() == world and D.get_rank() == rank and D.is_main_process() == (rank == 0)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_and_dist_helpers_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)], res


def test_single_process_is_identity():
    from frozenbilm_amd.parallel import GradReducer
    from frozenbilm_amd.util import dist as D

    flat = torch.arange(10.0)
    red = GradReducer(flat, {"a": 4, "b": 10}, min_bucket_elems=1, overlap="backward")
    red.ready("a"); red.ready("b"); red.finish()
    assert torch.equal(flat, torch.arange(10.0)) and red.last_launched == [(0, 4), (4, 10)]
    d = {"x": torch.tensor(1.0)}
    assert D.reduce_dict(d) is d and D.all_gather(3) == [3]
