"""Open-ended VideoQA loops with the reference's signatures and return values (videoqa.py:25-245) -- BASELINE config 4.

The model is the same masked-LM forward; the prediction head scores the answer vocabulary (``set_answer_embeddings``)
and the row of the ``[MASK]`` token is read out.  Fine-tuning back-propagates a loss computed by the caller on the
returned ``logits`` (they are a differentiable output of the single autograd node of the engine).
"""
from __future__ import annotations

from functools import reduce

import torch
import torch.nn.functional as F

from .loops import EpochRunner, LossLog, frozen_weights, logged_step, tokenize, video_inputs
from .util import dist


def mask_row_logits(output_logits, encoded_ids, tokenizer, args):
    """videoqa.py:66-69,163-166: rows of the text part (after the `max_feats` visual slots) where the input is [MASK]."""
    delay = args.max_feats if args.use_video else 0
    ids = encoded_ids.to(output_logits.device)
    return output_logits[:, delay: ids.size(1) + delay][ids == tokenizer.mask_token_id]


def mask_rows(encoded_ids, tokenizer, args, device):
    """Flat indices b*S + s (S = max_feats + L) of the [MASK] tokens: the rows mask_row_logits reads."""
    delay = args.max_feats if args.use_video else 0
    b, l = torch.nonzero(encoded_ids == tokenizer.mask_token_id, as_tuple=True)
    return (b * (encoded_ids.size(1) + delay) + delay + l).to(device)


def answer_logits(model, tokenizer, encoded_ids, args, **feed):
    """Logits of the [MASK] rows, [n_mask, n_ans].  Without gradients the HIP model runs its prediction head on those rows
    only (``logit_rows``); any other model -- or a training pass -- takes the reference's route: full logits, then
    the row selection of videoqa.py:164-168."""
    if hasattr(model, "engine") and not torch.is_grad_enabled():
        rows = mask_rows(encoded_ids, tokenizer, args, feed["input_ids"].device)
        return model(logit_rows=rows, **feed)["logits"]
    return mask_row_logits(model(**feed)["logits"], encoded_ids, tokenizer, args)


def vqa_loss(logits, answer_id, dataset_name):
    """videoqa.py:70-80: soft-label NLL for iVQA / VQA (answer counts /2 resp. /3, clamped to 1), CE otherwise."""
    if dataset_name in ("ivqa", "vqa"):
        a = (answer_id / (2 if dataset_name == "ivqa" else 3)).clamp(max=1)
        nll = -F.log_softmax(logits, 1)
        return (nll * a / a.sum(1, keepdim=True).clamp(min=1)).sum(dim=1).mean()
    return F.cross_entropy(logits, answer_id)


def topk_agreement(logits, answer_id, dataset_name, thresholds):
    """videoqa.py:167-195: softmax -> top-k answer ids; exact match, or the iVQA/VQA soft score of the best hit."""
    probs = logits.softmax(-1)
    topk_aids = torch.topk(probs, max(thresholds), -1).indices
    agreeings = {}
    if dataset_name not in ("ivqa", "vqa"):
        expanded = answer_id.view(-1, 1).expand_as(topk_aids)
        gt = answer_id
        for x in thresholds:
            agreeings[x] = topk_aids[:, :x] == expanded[:, :x]
    else:
        gt = (answer_id / (2 if dataset_name == "ivqa" else 3)).clamp(max=1)
        for x in thresholds:
            predicted = F.one_hot(topk_aids[:, :x], num_classes=gt.shape[-1]).sum(1)
            agreeings[x] = (predicted * gt).max(1)[0]
    return topk_aids, gt, agreeings


def train_one_epoch(model, tokenizer, data_loader, optimizer, device, epoch, dataset_name, args, max_norm: float = 0):
    model.train()
    run = EpochRunner(data_loader, args, "Epoch: [{}]".format(epoch), epoch)
    log = LossLog(run, "cls_loss", delayed=getattr(args, "delayed_loss_check", False), reducer=getattr(model, "_reducer", None))
    for i_batch, batch_dict in run:
        video, video_mask = video_inputs(batch_dict, device)
        encoded = tokenize(tokenizer, batch_dict["text"], args)
        output = model(video=video, video_mask=video_mask, input_ids=encoded["input_ids"].to(device),
                       attention_mask=encoded["attention_mask"].to(device))
        logits = mask_row_logits(output["logits"], encoded["input_ids"], tokenizer, args)
        loss = vqa_loss(logits, batch_dict["answer_id"].to(device), dataset_name)
        logged_step(log, loss, optimizer, model, max_norm)
        run.schedule(optimizer, i_batch)
        run.log(lr=optimizer.param_groups[0]["lr"])
    return run.finish()


@torch.no_grad()
def evaluate(model, tokenizer, data_loader, device, dataset_name, args, thresholds=[1, 10], split="test",
             type_map={0: "all"}):
    model.eval()
    if getattr(args, "inference_graphs", False) and hasattr(model, "inference_graphs"):
        model.inference_graphs = True  # replay the per-batch forward as one hipGraph (fixed batch shapes pay off most)
    if getattr(args, "packed_rows", False) and hasattr(model, "packed_rows"):
        model.packed_rows = True  # ragged batches without the padding rows behind each sample's last token (takes precedence)
    run = EpochRunner(data_loader, args, f"{split}:")
    res = {}
    with frozen_weights(model):  # nothing writes to the parameters during an evaluation: packed operands are reused
        for _, batch_dict in run:
            video, video_mask = video_inputs(batch_dict, device)
            encoded = tokenize(tokenizer, batch_dict["text"], args)
            input_ids = encoded["input_ids"].to(device)
            attention_mask = encoded["attention_mask"].to(device)
            if not args.suffix and not args.use_context:  # remove sep token if not using the suffix (videoqa.py:152-156)
                attention_mask[input_ids == tokenizer.sep_token_id] = 0
                input_ids[input_ids == tokenizer.sep_token_id] = tokenizer.pad_token_id
            logits = answer_logits(model, tokenizer, encoded["input_ids"], args, video=video, video_mask=video_mask,
                                   input_ids=input_ids, attention_mask=attention_mask)
            answer_id, qids = batch_dict["answer_id"].to(device), batch_dict["qid"]
            types = batch_dict["type"]
            subs = batch_dict["sub"] if "sub" in batch_dict else [0] * len(types)
            topk_aids, gts, agreeings = topk_agreement(logits, answer_id, dataset_name, thresholds)
            # one device-to-host copy per tensor instead of (2 + thresholds) synchronisations per question
            preds_h, gts_h = topk_aids.tolist(), gts.tolist()
            acc_h = {x: agreeings[x].reshape(len(qids), -1).sum(1).tolist() for x in thresholds}
            for i, (qid, type_, sub) in enumerate(zip(qids, types, subs)):
                res[qid] = {"pred": preds_h[i], "gt": gts_h[i], "type": int(type_), "sub": sub}
                for x in thresholds:
                    res[qid][f"acc{x}"] = acc_h[x][i]
            run.log(acc=dist.reduce_dict({"acc": agreeings[1].sum() / len(qids)})["acc"].item())

    all_res = dist.all_gather(res)
    results = reduce(lambda a, b: a.update(b) or a, all_res, {})
    assert len(results) == len(data_loader.dataset)
    out = {}
    for x in thresholds:
        out[f"acc{x}"] = sum(results[qid][f"acc{x}"] for qid in results) / len(results)
    acc_type = None
    if type_map is not None and len(type_map) > 1:
        acc_type = {type_map[i]: sum(results[qid]["acc1"] for qid in results if results[qid]["type"] == i)
                    / len([x for x in results.values() if x["type"] == i]) for i in type_map}
    n_sub = len([x for x in results.values() if x["sub"]])
    acc_sub = sum(results[qid]["acc1"] for qid in results if results[qid]["sub"]) / n_sub if n_sub else None
    if dist.is_main_process():
        print(dataset_name)
        for x in thresholds:
            print(f"{split} acc{x}: {out[f'acc{x}']: .2%}")
        if acc_type is not None:
            for x in acc_type:
                print(f"acc {x}: {acc_type[x]: .2%}")
            out.update(acc_type)
        if n_sub:
            print(f"acc sub: {acc_sub: .2%}; proportion {n_sub / len(results): .2%}")
            out["acc_sub"] = acc_sub
    return results, out
