"""CPU oracle for the FrozenBiLM masked-LM hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It is a plain torch-fp32 / numpy CPU
restatement of the reference algorithm (antoyang/FrozenBiLM, model/deberta.py,
model/adapter.py, util/misc.py) used ONLY as the checker by `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`.
Nothing under `frozenbilm_amd/` imports it; the product path fails loudly when
the HIP library is missing instead of falling back to this code.

Parity pin: the reference ships no tests or golden vectors, so
the oracle is pinned against outputs of the reference itself, imported in the
build container with the shims of SURVEY.md App. B by
`tests/golden/make_goldens.py`; the resulting input/output vectors are
committed under `tests/golden/*.npz` and `tests/test_oracle_golden.py` checks
the oracle against every one of them.
"""
