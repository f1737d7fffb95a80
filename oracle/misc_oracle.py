"""CPU oracle of the integer/host helpers of util/misc.py (TEST INFRASTRUCTURE ONLY).

get_mask           util/misc.py:6-11
mask_tokens        util/misc.py:14-56   (RNG order: bernoulli(p), bernoulli(.8), bernoulli(.5), randint)
adjust_learning_rate util/misc.py:59-78
"""
from __future__ import annotations

import torch


def get_mask(lengths: torch.Tensor, max_length: int) -> torch.Tensor:
    """int64 [B, max_length]: 1 where position < length."""
    pos = torch.arange(max_length, device=lengths.device)
    return (pos[None, :] < lengths.reshape(-1, 1)).to(torch.int64)


def mask_tokens(inputs: torch.Tensor, special_mask: torch.Tensor, pad_id, mask_id: int, vocab_len: int, p: float,
                generator=None):
    """BERT 80/10/10 masking; consumes the default CPU generator in the reference's order.

    ``special_mask`` bool [B,L] = tokenizer.get_special_tokens_mask per row; ``pad_id`` None if no pad token.
    Mutates ``inputs`` in place like the reference and returns (inputs, labels).
    """
    labels = inputs.clone()
    prob = torch.full(labels.shape, p)
    prob.masked_fill_(special_mask, 0.0)
    if pad_id is not None:
        prob.masked_fill_(labels.eq(pad_id), 0.0)
    chosen = torch.bernoulli(prob, generator=generator).bool()
    labels[~chosen] = -100
    repl = torch.bernoulli(torch.full(labels.shape, 0.8), generator=generator).bool() & chosen
    inputs[repl] = mask_id
    rnd = torch.bernoulli(torch.full(labels.shape, 0.5), generator=generator).bool() & chosen & ~repl
    words = torch.randint(vocab_len, labels.shape, dtype=torch.long, generator=generator)
    inputs[rnd] = words[rnd]
    return inputs, labels


def lr_at(step: int, total: int, lr: float, schedule: str, fraction_warmup_steps: float) -> float:
    warm = round(fraction_warmup_steps * total)
    if schedule == "linear_with_warmup":
        if step < warm:
            g = float(step) / float(max(1, warm))
        else:
            g = max(0.0, float(total - step) / float(max(1, total - warm)))
    else:
        g = 1
    return lr * g
