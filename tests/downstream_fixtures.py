"""Synthetic inputs of the downstream loops (BASELINE configs 4 and 5), shared by the golden generator (which feeds them
to the REFERENCE's videoqa.py / mc.py loops) and by the tests (which feed them to the product's loops).

No tokenizer model is available offline (tokenization parity is unpinned, the path starts at integer ids),
so the "text" of a sample is its token ids written as a space-separated string and ``StubTokenizer`` parses it back --
same call signature and special-token attributes as the HuggingFace tokenizer the loops expect.
"""
from __future__ import annotations

import torch


class StubTokenizer:
    pad_token_id, cls_token_id, sep_token_id, mask_token_id = 0, 1, 2, 4
    mask_token, _pad_token = "[MASK]", "[PAD]"

    def convert_tokens_to_ids(self, token):
        return {"[MASK]": self.mask_token_id, "[PAD]": self.pad_token_id}[token]

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size

    def __len__(self):
        return self.vocab_size

    def __call__(self, text, add_special_tokens=True, max_length=None, padding="longest", truncation=True,
                 return_tensors="pt"):
        rows = [[int(t) for t in s.split()] for s in text]
        if add_special_tokens:
            rows = [[self.cls_token_id] + r + [self.sep_token_id] for r in rows]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        L = max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        att = torch.zeros(len(rows), L, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r)
            att[i, : len(r)] = 1
        return {"input_ids": ids, "attention_mask": att}

    def get_special_tokens_mask(self, val, already_has_special_tokens=True):
        return [1 if v in (self.pad_token_id, self.cls_token_id, self.sep_token_id, self.mask_token_id) else 0 for v in val]


class _DS:
    def __init__(self, n, mc=None):
        self.n, self.mc = n, mc

    def __len__(self):
        return self.n


class ListLoader:
    """Stands in for the DataLoader: iterable of ready batches with ``len()`` and a ``.dataset``."""

    def __init__(self, batches, mc=None):
        self.batches = batches
        self.dataset = _DS(sum(len(b["qid"]) for b in batches), mc)

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def _text_with_mask(g, vocab, lo, n_tok, mask_id):
    toks = torch.randint(lo, vocab, (n_tok,), generator=g).tolist()
    pos = int(torch.randint(0, n_tok, (1,), generator=g))
    toks[pos] = mask_id
    return " ".join(str(t) for t in toks)


def make_videoqa_batches(vocab, T, F, n_ans, n_batches, B, seed, dataset_name="msrvtt", min_tok=3, max_tok=9):
    """Question text with exactly one [MASK]; answer ids in [0, n_ans) (iVQA/VQA: per-answer annotator counts)."""
    g = torch.Generator().manual_seed(seed)
    out, q = [], 0
    for _ in range(n_batches):
        video = torch.randn(B, T, F, generator=g).half().float()
        vlen = torch.randint(1, T + 1, (B,), generator=g)
        for b in range(B):
            video[b, vlen[b]:] = 0
        text = [_text_with_mask(g, vocab, 5, int(torch.randint(min_tok, max_tok + 1, (1,), generator=g)), StubTokenizer.mask_token_id)
                for _ in range(B)]
        if dataset_name in ("ivqa", "vqa"):
            answer_id = torch.zeros(B, n_ans)
            for b in range(B):
                ks = torch.randint(0, n_ans, (3,), generator=g)
                for k in ks:
                    answer_id[b, k] += int(torch.randint(1, 4, (1,), generator=g))
        else:
            answer_id = torch.randint(0, n_ans, (B,), generator=g)
        out.append(dict(video=video, video_len=vlen, text=text, answer_id=answer_id, qid=[f"q{q + i}" for i in range(B)],
                        type=torch.randint(0, 2, (B,), generator=g)))
        q += B
    return out


def make_mc_batches(vocab, T, F, n_choices, n_batches, B, seed, min_tok=4, max_tok=12, with_gt=True):
    """`text[aid]` is the batch of candidate `aid` (mc.py:44-50): same question prefix, different candidate suffix."""
    g = torch.Generator().manual_seed(seed)
    out, q = [], 0
    for _ in range(n_batches):
        video = torch.randn(B, T, F, generator=g).half().float()
        vlen = torch.randint(1, T + 1, (B,), generator=g)
        for b in range(B):
            video[b, vlen[b]:] = 0
        prefix = [torch.randint(5, vocab, (int(torch.randint(min_tok, max_tok + 1, (1,), generator=g)),), generator=g).tolist()
                  for _ in range(B)]
        text = []
        for _ in range(n_choices):
            cand = []
            for b in range(B):
                suffix = torch.randint(5, vocab, (int(torch.randint(1, 4, (1,), generator=g)),), generator=g).tolist()
                cand.append(" ".join(str(t) for t in prefix[b] + suffix + [StubTokenizer.mask_token_id]))
            text.append(cand)
        answer_id = torch.randint(0, n_choices, (B,), generator=g) if with_gt else torch.full((B,), -1)
        out.append(dict(video=video, video_len=vlen, text=text, answer_id=answer_id, qid=[f"m{q + i}" for i in range(B)],
                        type=torch.zeros(B, dtype=torch.long)))
        q += B
    return out


def make_videotext_batches(vocab, T, F, n_batches, B, seed, min_tok=6, max_tok=20):
    """main.py batches: clip features + plain caption ids (the loop itself applies mask_tokens)"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_batches):
        video = torch.randn(B, T, F, generator=g).half().float()
        vlen = torch.randint(1, T + 1, (B,), generator=g)
        for b in range(B):
            video[b, vlen[b]:] = 0
        text = [" ".join(str(t) for t in torch.randint(5, vocab, (int(torch.randint(min_tok, max_tok + 1, (1,), generator=g)),),
                                                       generator=g).tolist()) for _ in range(B)]
        out.append(dict(video=video, video_len=vlen, text=text, qid=list(range(B))))
    return out


class Args:
    """The handful of ``args`` fields the loops read."""

    def __init__(self, **kw):
        self.max_tokens, self.max_feats, self.use_video = 64, 4, True
        self.suffix, self.use_context, self.print_freq = "", False, 1000
        self.epochs, self.lr, self.schedule, self.fraction_warmup_steps = 1, 1e-3, "", 0.1
        self.mlm_prob = 0.15
        self.__dict__.update(kw)
