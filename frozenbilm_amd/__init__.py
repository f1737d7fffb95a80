"""frozenbilm_amd -- MI355X-native (gfx950) implementation of FrozenBiLM's masked-LM forward/backward hot path.

Only what the path needs: ``csrc/`` (HIP kernels + the C ABI of include/fbl.h, built into libfbl.so), ``lib`` (ctypes
binding), ``engine`` (explicit forward/backward pipelines), ``model`` (reference-shaped DebertaV2ForMaskedLM / Adapter),
``util`` + ``main`` (train_one_epoch / evaluate with the reference signatures), ``optim`` (fused Adam + clip),
``parallel`` (RCCL gradient all-reduce of the trainable set).
"""
__version__ = "0.1.0"
