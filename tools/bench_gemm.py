"""Micro-benchmark (+ optional correctness check) of fbl_gemm_bf16_nt on the hot-path shapes, HIP events on the launch stream.
usage: python tools/bench_gemm.py [--iters 20] [--check] [--set hot|square|all]
Kernel selection switches are environment variables read once per process (FBL_GEMM8, FBL_GEMM8_VAR, FBL_GEMM_NO224, ...),
so A/B comparisons run this script once per setting.  Operands are uniform random in [-1, 1) (never zero-filled: the chip
clocks higher on zeros).  --check compares against an fp32 torch matmul of the same bf16 operands (tool only)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frozenbilm_amd import lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--check", action="store_true")
ap.add_argument("--set", default="hot")
args = ap.parse_args()
dev = "cuda"
L.load()
SQUARE = [(4096, 4096, 4096, "bf16"), (8192, 8192, 8192, "bf16")]
HOT = [  # (M, N, K, variant)
    (8512, 1536, 1536, "f32+bf16"), (9024, 4608, 1536, "bf16"),
    (8512, 6144, 1536, "gelugrad"), (8512, 1536, 6144, "f32+bf16"),
    (8512, 6144, 1536, "mulbf16"), (8512, 1536, 6144, "addf32"), (8512, 1536, 4608, "addf32"), (8512, 1536, 1536, "bf16"),
    (8512, 128100, 1536, "logits"), (8512, 192, 1536, "relu"), (8512, 1536, 192, "addf32"), (8512, 1536, 192, "addbf16"),
]
EDGE = [(4100, 3500, 256, "f32+bf16"), (4100, 3584, 384, "gelugrad"), (8512, 6144, 256, "addf32"), (8512, 6144, 256, "mulbf16"),
        (8512, 6144, 256, "addbf16"), (8512, 6144, 256, "dgelu"), (8512, 6144, 256, "gelu")]
# the step's large GEMMs at the row count of a packed ragged batch (model.packed_rows: 5322 of 8512 rows at the bench batch)
PACKED = [(5834, 4608, 1536, "bf16"), (5322, 1728, 1536, "bf16"), (5322, 6144, 1536, "gelugrad"), (5322, 1728, 6144, "bf16"),
          (5322, 6144, 1536, "mulbf16"), (5322, 1536, 6144, "addf32"), (5322, 1536, 4608, "addf32"), (5322, 1536, 1792, "bf16"),
          (4100, 1536, 6144, "addf32"), (6900, 1536, 6144, "addf32"), (3000, 1536, 6144, "addf32")]
# the vocabulary GEMM of the loss on the labelled rows only (a few hundred to ~1300 rows at the bench batch)
HEAD = [(691, 128100, 1536, "logits"), (768, 128100, 1536, "logits"), (1280, 128100, 1536, "logits"),
        (691, 1536, 128128, "splitk"), (1280, 1536, 128128, "splitk")]  # ... and of its backward dh = dlogits . E (split-K, accumulating)
shapes = {"head": HEAD, "hot": SQUARE + HOT, "square": SQUARE, "all": SQUARE + HOT + EDGE, "edge": EDGE, "packed": PACKED}[args.set]
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FBL_"))
print(f"# {tag or 'default switches'}", flush=True)
for M, N, K, var in shapes:
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dev).to(torch.bfloat16)
    B = (torch.rand(N, K, generator=g) * 2 - 1).to(dev).to(torch.bfloat16)
    ldc = (N + 63) // 64 * 64
    o16 = torch.empty(M, ldc, dtype=torch.bfloat16, device=dev) if var != "logits" else None
    o32 = torch.empty(M, ldc, dtype=torch.float32, device=dev) if var in ("f32+bf16", "addf32", "logits", "splitk") else None
    if var == "splitk": o16 = None
    bias = torch.rand(N, device=dev)
    aux = None
    kw = dict(bias=bias, N=N)
    if var in ("bf16", "relu"): kw.update(out_bf16=o16, act=L.ACT_RELU if var == "relu" else L.ACT_NONE)
    elif var == "f32+bf16": kw.update(out_f32=o32, out_bf16=o16)
    elif var == "logits": kw.update(out_f32=o32)
    elif var == "gelu": kw.update(out_bf16=o16, out_pre=torch.empty_like(o16), act=L.ACT_GELU)
    elif var == "gelugrad": kw.update(out_bf16=o16, out_pre=torch.empty_like(o16), act=L.ACT_GELU_GRAD)
    elif var in ("dgelu", "mulbf16", "addbf16"):
        aux = torch.randn(M, ldc, device=dev).to(torch.bfloat16)
        kw.update(out_bf16=o16, aux=aux, aux_kind={"dgelu": L.AUX_MUL_DGELU_BF16, "mulbf16": L.AUX_MUL_BF16,
                                                    "addbf16": L.AUX_ADD_BF16}[var]); kw.pop("bias")
    elif var == "splitk":
        kw = dict(out_f32=o32, N=N, splitk=max(2, min(16, K // 8192)), ws=torch.empty(48 << 20, dtype=torch.float32, device=dev))
    elif var == "addf32":
        aux = torch.randn(M, ldc, device=dev)
        kw.update(out_f32=o32, aux=aux, aux_kind=L.AUX_ADD_F32); kw.pop("bias")
    for _ in range(3): L.gemm(A, B, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters): L.gemm(A, B, **kw)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / args.iters
    msg = ""
    if args.check and var not in ("logits", "splitk"):
        base = A.float() @ B.float().t()
        if "bias" in kw: base = base + bias
        x = aux[:, :N].float() if aux is not None else None
        if var == "relu": ref = torch.relu(base)
        elif var in ("gelu", "gelugrad"): ref = torch.nn.functional.gelu(base)
        elif var == "dgelu":
            xx = x.clone().requires_grad_(True); torch.nn.functional.gelu(xx).sum().backward(); ref = base * xx.grad
        elif var == "mulbf16": ref = base * x
        elif var in ("addbf16", "addf32"): ref = base + x
        else: ref = base
        errs = []
        if o32 is not None: errs.append(("f32", (o32[:, :N] - ref).abs().max().item(), 1e-3 * K ** 0.5 * 0 + 2e-3 * max(1.0, ref.abs().max().item())))
        if o16 is not None and kw.get("out_bf16") is not None: errs.append(("bf16", (o16[:, :N].float() - ref).abs().max().item(), 1.2e-2 * max(1.0, ref.abs().max().item())))
        if var == "gelugrad":
            bb = base.clone().requires_grad_(True); torch.nn.functional.gelu(bb).sum().backward()
            errs.append(("gelu'", (kw["out_pre"][:, :N].float() - bb.grad).abs().max().item(), 1.2e-2))
        ok = all(er <= tol for _, er, tol in errs)
        msg = ("  OK " if ok else "  MISMATCH ") + " ".join(f"{n}:{er:.2e}" for n, er, _ in errs)
        del base, ref
    print(f"M={M:6d} N={N:6d} K={K:5d} {var:9s} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s{msg}", flush=True)
    del A, B, o16, o32, kw, aux
