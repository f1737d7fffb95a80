"""CPU: host-side logic of the product (integer tables, masking, schedules, parameter bookkeeping) against the golden
vectors from the reference -- bit exact for the integer work."""
import math
import types

import numpy as np
import pytest
import torch

from frozenbilm_amd.model import DebertaV2Config, DebertaV2ForMaskedLM, build_model
from frozenbilm_amd.model.deberta import MaskedLMOutput, flat_order, param_shapes
from frozenbilm_amd.model.relpos import bucket_of_delta, rel_index_vector
from frozenbilm_amd.util import misc
from oracle import deberta_oracle as O


@pytest.mark.parametrize("S", [1, 2, 74, 129, 266, 512])
def test_relpos_bit_exact(golden, S):
    g = golden("G2_relpos")
    col, row = g[f"col_{S}"].numpy(), g[f"row_{S}"].numpy()
    d = np.arange(S)
    assert (bucket_of_delta(d, 256, 512) == col).all()
    assert (bucket_of_delta(-d, 256, 512) == row).all()
    v = rel_index_vector(S, 256, 512, 256)
    assert v.dtype == np.int16 and v.shape == (2 * S - 1,)
    assert (v[S - 1:] == np.clip(col + 256, 0, 511)).all()
    assert (v[:S][::-1] == np.clip(row + 256, 0, 511)).all()
    assert (v == O.rel_index_by_delta(S, O.OracleConfig())).all()


class Tok:
    mask_token = "[MASK]"
    _pad_token = "[PAD]"
    pad_token_id = 0

    def __len__(self):
        return 1000

    def get_special_tokens_mask(self, val, already_has_special_tokens=True):
        return [1 if v in (1, 2) else 0 for v in val]

    def convert_tokens_to_ids(self, t):
        return 4


def test_mask_tokens_get_mask_lr_bit_exact(golden):
    g = golden("G7_misc")
    assert torch.equal(misc.get_mask(g["video_len"], 10), g["get_mask"])
    assert misc.get_mask(g["video_len"], 10).dtype == torch.int64
    for seed in (0, 1):
        ids = g[f"ids_{seed}"].clone()
        torch.manual_seed(seed)
        inp, lab = misc.mask_tokens(ids, Tok(), 0.15)
        assert inp is ids  # in place, like the reference
        assert torch.equal(inp, g[f"inputs_{seed}"]) and torch.equal(lab, g[f"labels_{seed}"])
    lrs = []
    for sched in ("", "linear_with_warmup"):
        a = types.SimpleNamespace(lr=3e-4, schedule=sched, fraction_warmup_steps=0.1)
        for step in (0, 1, 9, 10, 11, 50, 99, 100):
            o = types.SimpleNamespace(param_groups=[{"lr": 0.0}])
            misc.adjust_learning_rate(o, step, 100, a)
            lrs.append(o.param_groups[0]["lr"])
    assert np.array_equal(np.array(lrs), g["lrs"].numpy())
    t = Tok()
    t.mask_token = None
    with pytest.raises(ValueError):
        misc.mask_tokens(torch.ones(1, 3, dtype=torch.long), t, 0.15)


def test_parameter_names_freeze_policy_and_flat_order():
    cfg = DebertaV2Config()
    sh = param_shapes(cfg, 1024, 8, 8, 0)
    osh = O.param_shapes(O.OracleConfig())
    assert dict(sh) == dict(osh)  # reference state_dict layout
    trainable = [n for n in sh if O.is_trainable(n)]
    assert sum(math.prod(sh[n]) for n in trainable) == 30128640
    assert len(trainable) == 298
    order = flat_order(cfg, trainable)
    assert sorted(order) == sorted(trainable) and len(set(order)) == len(order)
    # backward-completion order: head LN, layer 23 ... layer 1, conv, layer 0, rel LN, embeddings
    first = lambda pre: min(i for i, n in enumerate(order) if n.startswith(pre))
    assert first("lm_predictions.") < first("deberta.encoder.layer.23.") < first("deberta.encoder.layer.1.")
    assert first("deberta.encoder.layer.1.") < first("deberta.encoder.conv.") < first("deberta.encoder.layer.0.")
    assert first("deberta.encoder.layer.0.") < first("deberta.encoder.LayerNorm") < first("deberta.embeddings.")


def test_tiny_model_module_tree_and_errors():
    c = DebertaV2Config(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    m = DebertaV2ForMaskedLM(c, features_dim=32)
    names = dict(m.named_parameters())
    assert "deberta.encoder.layer.1.attention.output.adapter.down.weight" in names
    assert "deberta.embeddings.position_ids" in m.state_dict()
    for n, p in names.items():
        assert p.requires_grad == O.is_trainable(n), n
    assert names["deberta.embeddings.word_embeddings.weight"][0].abs().sum() == 0  # padding row zeroed
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(input_ids=torch.ones(1, 3, dtype=torch.long))
    with pytest.raises(ValueError):
        m()
    with pytest.raises(ValueError):
        m(input_ids=torch.ones(1, 3, dtype=torch.long), inputs_embeds=torch.zeros(1, 3, 128))
    with pytest.raises(NotImplementedError):
        DebertaV2ForMaskedLM(DebertaV2Config(hidden_size=1024, num_attention_heads=8), features_dim=32)  # head_dim 128
    with pytest.raises(NotImplementedError):
        DebertaV2ForMaskedLM(c, features_dim=32, freeze_lm=False)
    args = types.SimpleNamespace(model_name="deberta-v2-xlarge", features_dim=32, max_feats=10, ds_factor_attn=8,
                                 ds_factor_ff=0, dropout=0.1, use_video=True)
    m2 = build_model(args, config=c)
    assert not any("output.adapter" in n and "attention" not in n for n, _ in m2.named_parameters())


def test_masked_lm_output_access():
    o = MaskedLMOutput(loss=1, logits=2, hidden_states=None, attentions=None)
    assert o.loss == 1 and o["logits"] == 2 and o[0] == 1 and o[1] == 2
    with pytest.raises(AttributeError):
        o.nope


@pytest.mark.slow
def test_oracle_xlarge_golden(golden):
    """The oracle at true DeBERTa-v2-XLarge dims against the reference's own output (G6)."""
    from tests.golden.make_goldens import synth_batch

    g = golden("G6_xlarge")
    cfg = O.OracleConfig()
    P = O.synth_params(cfg, seed=0)
    batch = synth_batch(cfg, B=2, L=256, seed=66)
    with torch.no_grad():
        out = O.forward(P, cfg, **batch)
    lg = out["logits"]
    assert (lg[:, ::19, ::997] - g["logits_slice"]).abs().max().item() < 1e-3
    assert abs(out["loss"].item() - g["loss"].item()) < 1e-4
    assert abs(lg.double().sum().item() - g["logits_sum"].item()) < 1e-3 * g["logits_abs_sum"].item() * 1e-3 + 5.0
    assert torch.equal(lg.argmax(-1), g["argmax"])
    assert torch.equal(lg.topk(5, -1).indices[:, ::7], g["top5"])


def test_adapter_gradient_grouping_policy(monkeypatch):
    """Engine._dw_flush (host logic, no GPU): parked adapter-gradient products leave in launches of `dw_group` distinct
    adapters, an adapter executed twice (the last layer, enhanced mask decoder) is split over two launches, and the gradient
    reducer hears about a stage as soon as every product parked DURING that stage has been launched -- the repeated last
    layer's stage one launch later than the layers behind it, which do not wait for it."""
    from frozenbilm_amd import engine as E

    launches = []
    monkeypatch.setattr(E.L, "adapter_bwd_dw", lambda groups, A: launches.append([g[1] for g in groups]))
    nL = 8
    names = lambda li: [f"layer.{li}.a1", f"layer.{li}.a2"]
    G = {n + sfx: n for li in range(nL) for n in names(li) for sfx in (".up.weight", ".down.weight", ".down.bias")}
    eng = types.SimpleNamespace(dw_group=6, G=G, _bucket_key=lambda n: "layer" + n.split(".")[1], _dyz_pool={}, _dyz_shape=None)
    run = types.SimpleNamespace(dw_pending=[], dw_ready_keys=[], dw_count=0)
    ready = []
    red = types.SimpleNamespace(ready=ready.append)

    def park(li):
        for n in names(li):
            run.dw_pending.append((192, n, (None, None, None, None), run.dw_count, None))
            run.dw_count += 1

    def stage_done(key):
        run.dw_ready_keys.append(key)
        E.Engine._dw_flush(eng, run, red)

    stage_done("head")
    assert ready == ["head"] and not launches  # nothing parked yet: final at once
    park(nL - 1)
    park(nL - 1)  # second execution of the last layer
    stage_done(f"layer{nL - 1}")
    for li in range(nL - 2, -1, -1):
        park(li)
        stage_done(f"layer{li}")
        if li == 5:  # six distinct adapters pending: the first launch leaves, the repeated pair stays behind ...
            assert launches == [names(7) + names(6) + names(5)]
            assert ready == ["head", "layer6", "layer5"]  # ... and with it layer 7's bucket only
    E.Engine._dw_flush(eng, run, red, force=True)
    assert not run.dw_pending and not run.dw_ready_keys
    assert all(len(set(l)) == len(l) <= 6 for l in launches)  # an adapter at most once per launch
    flat = [n for l in launches for n in l]
    assert sorted(flat) == sorted(names(nL - 1) * 2 + [n for li in range(nL - 1) for n in names(li)])
    assert launches[1][:2] == names(7)  # the deferred pair leads the next launch
    assert sorted(ready) == sorted(["head"] + [f"layer{li}" for li in range(nL)]) and len(ready) == nL + 1
    assert ready.index("layer7") > ready.index("layer5")  # final only with the second launch


def test_packed_row_layout_of_a_ragged_batch():
    """engine.Packing (model.packed_rows): every sample keeps the rows of its positions 0 .. last used position (valid token,
    label, requested logit row; at least the video slots); the maps between the padded grid and the packed rows are
    consistent; a batch without droppable rows is not packed."""
    from frozenbilm_amd.engine import Engine

    eng = types.SimpleNamespace(dev=torch.device("cpu"))
    B, T, Lt = 4, 3, 9
    S = T + Lt
    mask = torch.zeros(B, S, dtype=torch.int32)
    mask[0, :2] = 1; mask[0, T:T + 4] = 1           # 2 frames, 4 tokens: last valid position T+3
    mask[1, :T] = 1; mask[1, T:] = 1                # full length
    mask[2, :1] = 1; mask[2, T:T + 1] = 1           # 1 frame, 1 token
    mask[3, :T] = 1                                 # no text at all: the video slots stay
    labels = torch.full((B, S), -100, dtype=torch.long)
    labels[0, T + 2] = 7
    labels[2, T + 5] = 9                            # a label behind the last valid token keeps its row
    pk = Engine._make_packing(eng, mask, labels.view(-1), None, B, S, T)
    plen = (pk.row0[1:] - pk.row0[:-1]).tolist()
    assert plen == [T + 4, S, T + 6, T] and pk.n == sum(plen) and pk.row0[0] == 0
    assert pk.sel.tolist() == [b * S + s for b in range(B) for s in range(plen[b])]
    assert pk.pos.tolist() == [s for b in range(B) for s in range(plen[b])]
    assert torch.equal(pk.inv[pk.sel], torch.arange(pk.n)) and int((pk.inv >= 0).sum()) == pk.n
    # requested logit rows extend a sample as labels do
    rows = torch.tensor([2 * S + T + 7])
    pk2 = Engine._make_packing(eng, mask, None, rows, B, S, T)
    assert (pk2.row0[1:] - pk2.row0[:-1]).tolist() == [T + 4, S, T + 8, T] and int(pk2.inv[rows[0]]) >= 0
    # nothing to drop -> no packing
    assert Engine._make_packing(eng, torch.ones(B, S, dtype=torch.int32), None, rows, B, S, T) is None


def test_bench_supervisor_restarts_a_measuring_process_that_died_silently(capfd):
    """bench.py at --gpus 1 measures in a child process: one that is killed before it printed its JSON line is started once
    more (and only once), whatever its exit status was (the failure this guards against came back as a silent status 1), and
    the second attempt is told how and where the first one ended."""
    import sys

    import bench

    dies_first = ("import os, signal, json\n"
                  f"a = os.environ['{bench.CHILD_MARK}']\n"
                  "print('starting', a, flush=True)\n"
                  f"open(os.environ['{bench.PHASE_FILE}'], 'w').write('phase-' + a)\n"
                  "if a == '1': os.kill(os.getpid(), signal.SIGKILL)\n"
                  "print(json.dumps({'metric': 'm', 'first': os.environ.get('FBL_BENCH_FIRST_RC'),\n"
                  "                  'where': os.environ.get('FBL_BENCH_FIRST_PHASE')}), flush=True)\n")
    assert bench.supervise_single_rank([sys.executable, "-c", dies_first]) == 0
    out, err = capfd.readouterr()
    assert out.count("starting") == 2 and out.count('"metric"') == 1 and '"first": "-9"' in out and '"where": "phase-1"' in out
    assert "once more" in err and "phase-1" in err
    always_dies = "import os, signal\nprint('x', flush=True)\nos.kill(os.getpid(), signal.SIGSEGV)\n"
    assert bench.supervise_single_rank([sys.executable, "-c", always_dies]) == 128 + 11
    out, _ = capfd.readouterr()
    assert out.count("x") == 2
    raises = "print('y', flush=True)\nraise SystemExit(1)\n"
    assert bench.supervise_single_rank([sys.executable, "-c", raises]) == 1
    out, _ = capfd.readouterr()
    assert out.count("y") == 2
    # who supervises: the plain single-GPU call only
    ns = types.SimpleNamespace(gpus=1, no_retry=False)
    import os

    saved = {k: os.environ.pop(k) for k in list(os.environ) if k in ("WORLD_SIZE", bench.CHILD_MARK) or k.startswith(("ROCP_", "ROCPROF"))}
    try:
        assert bench._wants_supervisor(ns)
        assert not bench._wants_supervisor(types.SimpleNamespace(gpus=2, no_retry=False))
        assert not bench._wants_supervisor(types.SimpleNamespace(gpus=1, no_retry=True))
        for k in ("WORLD_SIZE", bench.CHILD_MARK, "ROCP_TOOL_LIBRARIES"):
            os.environ[k] = "1"
            assert not bench._wants_supervisor(ns)
            del os.environ[k]
    finally:
        os.environ.update(saved)


def test_bench_supervisor_takes_the_measuring_process_with_it(tmp_path):
    """Whoever stops `bench.py` stops the measurement: SIGTERM is passed on to the measuring process, and a SIGKILL of the
    supervisor (a time-out) takes it along too (PR_SET_PDEATHSIG) -- no orphan keeps the GPU busy under the next command."""
    import os
    import signal
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sig in (signal.SIGTERM, signal.SIGKILL):
        pidfile = tmp_path / f"pid{int(sig)}"
        child = f"import os, time\nopen({str(pidfile)!r}, 'w').write(str(os.getpid()))\ntime.sleep(60)\n"
        sup = subprocess.Popen([sys.executable, "-c",
                                f"import sys; sys.path.insert(0, {root!r}); import bench; "
                                f"sys.exit(bench.supervise_single_rank([sys.executable, '-c', {child!r}]))"],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t0 = time.time()
        while not (pidfile.exists() and pidfile.read_text()) and time.time() - t0 < 60:
            time.sleep(0.05)
        pid = int(pidfile.read_text())
        sup.send_signal(sig)
        sup.wait(timeout=30)
        t0 = time.time()
        alive = True
        while alive and time.time() - t0 < 10:
            try:
                os.kill(pid, 0)
                # (a zombie re-parented to init answers kill(0) until it is reaped: look at its state)
                alive = open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] not in ("Z", "X")
            except (ProcessLookupError, FileNotFoundError):
                alive = False
            time.sleep(0.05)
        assert not alive, f"the measuring process survived its supervisor's {sig.name}"
