import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from xnrs_amd import hip, synth
dev = torch.device("cuda", 0)
w = bench.WORKLOAD
model, _ = bench.build_model(w, dev)
gen = torch.Generator(device=dev); gen.manual_seed(31)
n_news = 65536
tx, tm = synth.device_tokens(gen, n_news + 1, w["S"], w["D"], dev)
tx[0] = 0; tm[0] = 0
tm = tm.reshape(n_news + 1, w["S"])
rng = np.random.default_rng(5)
z = np.minimum(rng.zipf(1.1, size=(w["B"], w["H"] + w["C"])), n_news).astype(np.int32)
n_hist = rng.integers(1, w["H"] + 1, size=(w["B"], 1))
z[:, :w["H"]][np.arange(w["H"])[None, :] >= n_hist] = 0
ids = torch.from_numpy(z).to(dev)
hist_ids, cand_ids = ids[:, :w["H"]].contiguous(), ids[:, w["H"]:].contiguous()
# materialised dense inputs of the SAME ids
g = hist_ids.long(); c = cand_ids.long()
hist = (tx[g], tm[g].unsqueeze(-1)); cand = (tx[c], tm[c].unsqueeze(-1))
with torch.no_grad():
    fa = lambda: model.forward_ids(tx, tm, hist_ids, cand_ids)
    fb = lambda: model._forward(hist, cand)
    # the gather kernel on SEQUENTIAL rows: a table that is the materialised batch itself, ids = 0..n-1 (separates the
    # cost of the per-lane 64-bit row pointers from the cost of the random rows)
    stab = torch.cat([hist[0].reshape(-1, w["S"], w["D"]), cand[0].reshape(-1, w["S"], w["D"])])
    smask = torch.cat([hist[1].reshape(-1, w["S"]), cand[1].reshape(-1, w["S"])])
    nh = w["B"] * w["H"]
    sh = torch.arange(nh, device=dev, dtype=torch.int32).reshape(w["B"], w["H"])
    sc = (nh + torch.arange(w["B"] * w["C"], device=dev, dtype=torch.int32)).reshape(w["B"], w["C"])
    fc = lambda: model.forward_ids(stab, smask, sh, sc)
    res = {"ids": [], "dense": [], "seq_ids": []}
    for rnd in range(4):
        for name, fn in (("ids", fa), ("dense", fb), ("seq_ids", fc)):
            res[name].append(bench.timed(fn, 5, 2, False) / 5 * 1e3)
    print({k: [round(x, 2) for x in v] for k, v in res.items()})
    for name, fn in (("ids", fa), ("dense", fb), ("seq_ids", fc)):
        hip.profile_enable(hip.PROFILE_ALL); fn(); torch.cuda.synchronize(); st = hip.profile_read(); hip.profile_enable(0)
        print(name, {k: round(v[0], 3) for k, v in st.items()})
    print("equal", torch.equal(fa(), fb()), torch.equal(fc(), fb()))
