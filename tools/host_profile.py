"""cProfile of the host side of a few training steps (where does the Python time go?)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["bench.py"]
import bench
from frozenbilm_amd.model import DebertaV2Config, DebertaV2ForMaskedLM
from frozenbilm_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
model = DebertaV2ForMaskedLM(DebertaV2Config(), max_feats=10, features_dim=1024).to(dev).train()
opt = FusedAdam(model, lr=3e-5)
batch = bench.synth_batch(32, 10, 1024, 256, 128100, 1, dev)
def step():
    opt.zero_grad(set_to_none=False); loss = model(**batch).loss; loss.backward(); opt.step(clip_max_norm=0.1)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28); print(st.getvalue()[:5000])
