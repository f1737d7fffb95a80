"""train-mode parity over many mask streams (torch seeds): worst tensor per seed, fp32 oracle and bf16-operand / gate-matched oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import deberta_oracle as O
from tests.golden.make_goldens import _tiny_cfg, synth_batch
from tests.test_gpu_model import build, to_dev, _gates_of, _rel_fro
from tests.dropout_replay import ReplayedMasks

cfg = _tiny_cfg()
B, Lt = 4, 40
P = O.synth_params(cfg, seed=43, std=0.05, ln_jitter=0.1)
batch = synth_batch(cfg, B=B, L=Lt, seed=9)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    torch.manual_seed(1000 + seed * 7919)
    m = build(cfg, P, train=True)
    out = m(**to_dev(batch))
    run = out.__dict__["_run"]
    c = m.config
    masks = ReplayedMasks(run, cfg, cfg.num_attention_heads, c.hidden_dropout_prob, c.attention_probs_dropout_prob, m.adapter_dropout)
    masks2 = ReplayedMasks(run, cfg, cfg.num_attention_heads, c.hidden_dropout_prob, c.attention_probs_dropout_prob, m.adapter_dropout)
    gates = _gates_of(run, cfg)
    out.loss.backward()
    torch.cuda.synchronize()
    got = {n: p.grad.float().cpu().clone() for n, p in m.named_parameters() if p.requires_grad}
    res = []
    for tag, ctxs in (("fp32", [O.dropout_masks(masks)]),
                      ("bf16+gates", [O.bf16_operands(), O.adapter_gates([g_.view(B, -1, g_.shape[-1]) for g_ in gates]), O.dropout_masks(masks2)])):
        for k, v in P.items():
            v.requires_grad_(O.is_trainable(k)); v.grad = None
        import contextlib
        with contextlib.ExitStack() as st:
            for cx in ctxs: st.enter_context(cx)
            ref = O.forward(P, cfg, **batch); ref["loss"].backward()
        worst_fro = max(((_rel_fro(got[n], P[n].grad), n) for n in got))
        worst_max = max((((got[n] - P[n].grad).abs().max().item() / max(P[n].grad.abs().max().item(), 1e-6)), n) for n in got)
        res.append(f"{tag}: loss d={abs(out.loss.item()-ref['loss'].item()):.2e} fro {worst_fro[0]:.3e} {worst_fro[1].split('deberta.')[-1]} | maxrel {worst_max[0]:.3e} {worst_max[1].split('deberta.')[-1]}")
    print(f"seed {seed}: " + " || ".join(res), flush=True)
    del m, out, run
