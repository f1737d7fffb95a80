"""Model factory with the reference's signatures (model/__init__.py:14-112)."""
from .adapter import Adapter
from .config import DebertaV2Config
from .deberta import DebertaV2ForMaskedLM, MaskedLMOutput


def build_model(args, config=None):
    """``build_model(args)`` of the reference for the DeBERTa branch (model/__init__.py:15-47).

    No hub / checkpoint access exists on the MI355X box, so the configuration is the literal DeBERTa-v2-XLarge one
    unless ``config`` (ours, a transformers config, or a dict) is given; weights come from ``load_state_dict``
    (reference key names) or stay at their seeded initialisation (``--scratch`` behaviour).
    """
    name = getattr(args, "model_name", "deberta-v2-xlarge")
    if "deberta" not in name:
        raise NotImplementedError(f"only the DeBERTa-v2 path is implemented (model_name={name!r})")
    cfg = DebertaV2Config.from_any(config) if config is not None else DebertaV2Config()
    return DebertaV2ForMaskedLM(
        cfg,
        features_dim=args.features_dim if getattr(args, "use_video", True) else 0,
        max_feats=args.max_feats,
        freeze_lm=getattr(args, "freeze_lm", True),
        freeze_mlm=getattr(args, "freeze_mlm", True),
        ft_ln=getattr(args, "ft_ln", True),
        ds_factor_attn=args.ds_factor_attn,
        ds_factor_ff=args.ds_factor_ff,
        dropout=args.dropout,
        n_ans=getattr(args, "n_ans", 0),
        freeze_last=getattr(args, "freeze_last", True),
    )


def get_tokenizer(args):
    """Pass-through to transformers' DebertaV2Tokenizer (model/__init__.py:94-98); needs a local spm.model."""
    from transformers import DebertaV2Tokenizer

    return DebertaV2Tokenizer.from_pretrained(args.model_name, local_files_only=True)
