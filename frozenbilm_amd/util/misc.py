"""Step helpers with the reference's signatures (util/misc.py): host-side integer work, reproduced exactly."""
from __future__ import annotations

import torch


def get_mask(lengths, max_length):
    """[B, max_length] int64 padding mask, 1 where position < length (util/misc.py:6-11)."""
    steps = torch.arange(max_length, device=lengths.device)
    return (steps.unsqueeze(0) < lengths.reshape(-1, 1)).long()


def mask_tokens(inputs, tokenizer, mlm_probability):
    """BERT masking for MLM: 15% of the non-special, non-pad tokens are selected; of those 80% become [MASK], 10% a
    random id, 10% stay (util/misc.py:14-56).  Same RNG consumption order as the reference (default CPU generator:
    bernoulli(p), bernoulli(0.8), bernoulli(0.5), randint) so a seeded run yields bit-identical indices.
    ``inputs`` is modified in place, exactly like the reference."""
    if tokenizer.mask_token is None:
        raise ValueError(
            "This tokenizer does not have a mask token which is necessary for masked language modeling. "
            "Remove the --mlm flag if you want to use this tokenizer.")
    labels = inputs.clone()
    special = torch.tensor(
        [tokenizer.get_special_tokens_mask(row, already_has_special_tokens=True) for row in labels.tolist()],
        dtype=torch.bool)
    select_p = torch.full(labels.shape, float(mlm_probability))
    select_p.masked_fill_(special, 0.0)
    if tokenizer._pad_token is not None:
        select_p.masked_fill_(labels.eq(tokenizer.pad_token_id), 0.0)
    picked = torch.bernoulli(select_p).bool()
    labels[~picked] = -100
    to_mask = torch.bernoulli(torch.full(labels.shape, 0.8)).bool() & picked
    inputs[to_mask] = tokenizer.convert_tokens_to_ids(tokenizer.mask_token)
    to_random = torch.bernoulli(torch.full(labels.shape, 0.5)).bool() & picked & ~to_mask
    random_ids = torch.randint(len(tokenizer), labels.shape, dtype=torch.long)
    inputs[to_random] = random_ids[to_random]
    return inputs, labels


def special_token_ids(tokenizer) -> torch.Tensor:
    """ids `get_special_tokens_mask(..., already_has_special_tokens=True)` flags, plus padding (util/misc.py:27-35)"""
    ids = set(getattr(tokenizer, "all_special_ids", None) or [])
    for name in ("pad_token_id", "cls_token_id", "sep_token_id", "mask_token_id", "unk_token_id", "bos_token_id", "eos_token_id"):
        v = getattr(tokenizer, name, None)
        if v is not None:
            ids.add(int(v))
    return torch.tensor(sorted(ids), dtype=torch.long)


def mask_tokens_device(inputs, tokenizer, mlm_probability, seed: int):
    """`mask_tokens` as one kernel on ids that already live on the GPU (same distribution, counter-based RNG keyed by
    `seed`; the host version above is the one that reproduces the reference's CPU generator bit for bit).  In place on
    `inputs` like the reference; returns (inputs, labels)."""
    from .. import lib as L

    if getattr(tokenizer, "mask_token_id", None) is None:
        raise ValueError(
            "This tokenizer does not have a mask token which is necessary for masked language modeling. "
            "Remove the --mlm flag if you want to use this tokenizer.")
    if not inputs.is_cuda:
        raise RuntimeError("mask_tokens_device needs the ids on the GPU (use mask_tokens for host tensors)")
    inputs = inputs if inputs.is_contiguous() else inputs.contiguous()
    labels = torch.empty_like(inputs)
    L.mask_tokens(inputs, labels, special_token_ids(tokenizer).to(inputs.device), mlm_probability, tokenizer.mask_token_id,
                  len(tokenizer), seed)
    return inputs, labels


def adjust_learning_rate(optimizer, curr_step: int, num_training_steps: int, args):
    """constant, or linear warm-up then linear decay (util/misc.py:59-78); writes param_groups[0]['lr']."""
    warmup = round(args.fraction_warmup_steps * num_training_steps)
    if args.schedule == "linear_with_warmup":
        if curr_step < warmup:
            factor = float(curr_step) / float(max(1, warmup))
        else:
            factor = max(0.0, float(num_training_steps - curr_step) / float(max(1, num_training_steps - warmup)))
    else:
        factor = 1
    optimizer.param_groups[0]["lr"] = args.lr * factor
