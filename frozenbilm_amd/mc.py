"""Multiple-choice VideoQA loops with the reference's signatures and return values (mc.py:25-231) -- BASELINE config 5.

One forward per answer candidate over the same video prefix (S up to 512 with ASR context), the 2-way Yes/No answer
head read at the ``[MASK]`` row, ``softmax[:, 0]`` as the candidate's score; balanced BCE for training.  Under data
parallelism the gradient exchange waits for the last candidate's backward (``GradReducer.accumulate``).
"""
from __future__ import annotations

from functools import reduce

import torch
import torch.nn.functional as F

from .loops import EpochRunner, LossLog, frozen_weights, logged_step, tokenize, video_inputs
from .util import dist
from .videoqa import answer_logits


def candidate_scores(model, tokenizer, batch_dict, device, args):
    """mc.py:44-72,140-165: text[aid] is the batch of candidate `aid`; returns scores [B, n_candidates].

    The reference runs one forward per candidate over the same video prefix.  Here the candidates of a batch go through
    ONE forward of C.B samples (each candidate's token batch padded to the longest one: padding is masked, and a sample's
    output does not depend on the rest of its batch), which fills the chip at B = 8 and, in training, needs one backward
    instead of C.  ``args.mc_sequential = True`` keeps the reference's loop."""
    video, video_mask = video_inputs(batch_dict, device)
    text = batch_dict["text"]
    C = len(text)
    enc = [tokenize(tokenizer, text[aid], args) for aid in range(C)]
    if getattr(args, "mc_sequential", False) or C == 1:
        scores = []
        for e in enc:  # one forward per answer candidate id
            logits = answer_logits(model, tokenizer, e["input_ids"], args, video=video, video_mask=video_mask,
                                   input_ids=e["input_ids"].to(device), attention_mask=e["attention_mask"].to(device))
            scores.append(logits.softmax(-1)[:, 0])
        return torch.stack(scores, 1)
    Lmax = max(e["input_ids"].size(1) for e in enc)
    B = video.size(0)
    ids = torch.full((C * B, Lmax), tokenizer.pad_token_id, dtype=enc[0]["input_ids"].dtype)
    att = torch.zeros((C * B, Lmax), dtype=enc[0]["attention_mask"].dtype)
    for aid, e in enumerate(enc):
        ids[aid * B:(aid + 1) * B, : e["input_ids"].size(1)] = e["input_ids"]
        att[aid * B:(aid + 1) * B, : e["attention_mask"].size(1)] = e["attention_mask"]
    logits = answer_logits(model, tokenizer, ids, args, video=video.repeat(C, 1, 1), video_mask=video_mask.repeat(C, 1),
                           input_ids=ids.to(device), attention_mask=att.to(device))
    return logits.softmax(-1)[:, 0].view(C, B).t()  # one [MASK] per text (mc.py:166-172)


def mc_loss(scores, gt, n_choices):
    """mc.py:75-92: balanced BCE over the positive and the negative candidates (plain BCE for a single candidate)."""
    if n_choices > 1:
        pos = scores[torch.arange(len(scores), device=scores.device), gt]
        neg_mask = torch.ones_like(scores)
        neg_mask.scatter_(1, gt.unsqueeze(-1), 0)
        neg = scores[neg_mask.bool()].view(len(scores), n_choices - 1).view(-1)
        pos_loss = F.binary_cross_entropy(pos, torch.ones_like(pos))
        neg_loss = F.binary_cross_entropy(neg, torch.zeros_like(neg))
        return (pos_loss + neg_loss) / 2
    return F.binary_cross_entropy(scores.squeeze(1), gt.float())


def train_one_epoch(model, tokenizer, data_loader, optimizer, device, epoch, args, max_norm: float = 0):
    model.train()
    run = EpochRunner(data_loader, args, "Epoch: [{}]".format(epoch), epoch)
    # several forwards feed one step: under data parallelism the gradient exchange waits for the last backward pass
    reducer = getattr(model.engine(), "reducer", None) if hasattr(model, "engine") else None
    log = LossLog(run, "cls_loss", delayed=getattr(args, "delayed_loss_check", False), reducer=getattr(model, "_reducer", None))
    for i_batch, batch_dict in run:
        scores = candidate_scores(model, tokenizer, batch_dict, device, args)
        loss = mc_loss(scores, batch_dict["answer_id"].to(device), data_loader.dataset.mc)
        logged_step(log, loss, optimizer, model, max_norm, reducer=reducer)
        run.schedule(optimizer, i_batch)
        run.log(lr=optimizer.param_groups[0]["lr"])
    return run.finish()


@torch.no_grad()
def evaluate(model, tokenizer, data_loader, device, dataset_name, args, split="test", type_map={0: "all"}):
    model.eval()
    if getattr(args, "inference_graphs", False) and hasattr(model, "inference_graphs"):
        model.inference_graphs = True  # replay the per-batch forward as one hipGraph (fixed batch shapes pay off most)
    if getattr(args, "packed_rows", False) and hasattr(model, "packed_rows"):
        model.packed_rows = True  # ragged batches without the padding rows behind each sample's last token (takes precedence)
    run = EpochRunner(data_loader, args, f"{split}:")
    res = {}
    with frozen_weights(model):  # nothing writes to the parameters during an evaluation: packed operands are reused
        for _, batch_dict in run:
            scores = candidate_scores(model, tokenizer, batch_dict, device, args)
            preds = scores.round().long().squeeze(1) if scores.shape[1] == 1 else scores.max(1).indices
            qids, types = batch_dict["qid"], batch_dict["type"]
            if batch_dict["answer_id"][0].item() != -1:
                answer_id = batch_dict["answer_id"].to(device)
                agreeings = preds == answer_id
                # one device-to-host copy per tensor instead of three .item() synchronisations per question
                preds_h, gts_h, agr_h = preds.tolist(), answer_id.tolist(), agreeings.tolist()
                for i, (qid, type_) in enumerate(zip(qids, types)):
                    res[qid] = {"pred": preds_h[i], "gt": gts_h[i]}
                    if type_map is not None and len(type_map) > 1:
                        res[qid]["type"] = int(type_)
                    res[qid]["acc"] = agr_h[i]
                run.log(acc=dist.reduce_dict({"acc": agreeings.sum() / len(qids)})["acc"].item())
            else:  # hidden test set: predictions only (mc.py:205-207)
                for qid, pred in zip(qids, preds.tolist()):
                    res[str(qid)] = int(pred)
    all_res = dist.all_gather(res)
    results = reduce(lambda a, b: a.update(b) or a, all_res, {})
    assert len(results) == len(data_loader.dataset)
    if isinstance(next(iter(results.values())), dict):
        acc = sum(int(results[qid]["acc"]) for qid in results) / len(results)
        acc_type = None
        if type_map is not None and len(type_map) > 1:
            acc_type = {type_map[i]: sum(results[qid]["acc"] for qid in results if results[qid]["type"] == i)
                        / len([x for x in results.values() if x["type"] == i]) for i in type_map}
        if dist.is_main_process():
            print(dataset_name)
            print(f"{split} acc: {acc: .2%}")
            if acc_type is not None:
                for x in acc_type:
                    print(f"acc {x}: {acc_type[x]: .2%}")
        return results, acc
    return results, 0
