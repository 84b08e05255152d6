import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
os.environ["RECALGO_SPARSE_DBG_BUF"] = str(buf.data_ptr())
from recalgorithm_amd import sparse as sp
from recalgorithm_amd.io import synth
from recalgorithm_amd.variables import EmbeddingArena
spec = synth.SynthSpec(n_fields=26, max_vocab=1_000_000)
ar = EmbeddingArena("t", 16, dev, seed=1)
for n, v in zip(spec.names, spec.vocabs): ar.add_table(n, v)
ar.materialize()
names = sorted(spec.names)
rb = torch.tensor([ar.tables[n][0] for n in names], dtype=torch.int64, device=dev)
feats, _, _ = synth.device_features(spec, 4096, dev, batch_index=0)
ids = torch.stack([feats[n] for n in names], 1).contiguous()
g = torch.randn(4096, 26 * 16, device=dev)
class Store:
    opt_state = {"step": torch.ones(1, dtype=torch.int64, device=dev)}; arenas = {"t": ar}
st = Store()
for it in range(3):
    src = sp.begin_lookup(ar, st, ids, None, rb, 0, 4096, 26, True); src.set_grad(g)
    sp.apply(ar, False, st.opt_state["step"], 0.001, 0.9, 0.999, 1e-8)
torch.cuda.synchronize()
t = buf.view(-1, 8)[:1024].cpu()
tot = (t[:, 3] - t[:, 0]); i = int(tot.argmax())
# wall_clock64 ticks at 100 MHz
f = 1e6 / 100e6
print("block with the longest time:", i, "n", int(t[i, 4]), "n_long", int(t[i, 5]), "us: count", (t[i,1]-t[i,0]).item()*f, "group", (t[i,2]-t[i,1]).item()*f, "process", (t[i,3]-t[i,2]).item()*f)
med = tot.median().item(); j = int((tot - med).abs().argmin())
print("median block:", j, "n", int(t[j, 4]), "us: count", (t[j,1]-t[j,0]).item()*f, "group", (t[j,2]-t[j,1]).item()*f, "process", (t[j,3]-t[j,2]).item()*f)
srt = torch.sort(tot, descending=True)
print("top 8 blocks (us, n, n_long):", [(round(tot[k].item()*f,1), int(t[k,4]), int(t[k,5])) for k in srt.indices[:8].tolist()])
print("kernel span us:", (t[:,3].max() - t[:,0].min()).item()*f)
