"""An nn.Module face for the CPU oracle (TEST INFRASTRUCTURE ONLY): same call convention and outputs as the model of
the path, so the product's host-side loops (main / videoqa / mc) can be driven on CPU and compared with the reference's
golden results.  Parameters follow the freeze policy of model/deberta.py:1152-1158,1334-1339; eval-mode arithmetic
(dropout 0) in both modes.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import deberta_oracle as O


class OracleModel(nn.Module):
    def __init__(self, cfg: O.OracleConfig, P: O.Params, a2tok=None):
        super().__init__()
        self.cfg = cfg
        self._names = list(P)
        for i, (n, v) in enumerate(P.items()):
            self.register_parameter(f"p{i}", nn.Parameter(v.clone(), requires_grad=O.is_trainable(n)))
        if a2tok is not None:
            self.set_answer_embeddings(a2tok)

    def _P(self):
        return {n: getattr(self, f"p{i}") for i, n in enumerate(self._names)}

    def named_ref_parameters(self):
        return self._P()

    def set_answer_embeddings(self, a2tok, freeze_last=True):
        P = self._P()
        table = O.answer_embeddings(a2tok, P, self.cfg).detach()
        # model/deberta.py:1371-1377: the table is replaced; ``answer_bias.weight = ...`` only sets an attribute, so the
        # bias keeps whatever value it had (zeros from the ctor, or what a checkpoint loaded)
        for n, v in (("answer_embeddings.weight", table), ("answer_bias", torch.zeros(len(table)))):
            if n in self._names:
                if n == "answer_embeddings.weight":
                    P[n].data = v
            else:
                self.register_parameter(f"p{len(self._names)}", nn.Parameter(v, requires_grad=False))
                self._names.append(n)
        self.cfg.n_ans = len(table)

    def forward(self, video=None, video_mask=None, input_ids=None, attention_mask=None, labels=None, mlm=False, **_):
        return O.forward(self._P(), self.cfg, input_ids, attention_mask, video, video_mask, labels, mlm)
