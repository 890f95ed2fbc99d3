"""where one filtered evaluation spends its time (FB15k-shaped: 14 951 entities, 50 000 test triples x 2 modes, 592 k known triples):
host filter lists (numpy) vs device ranking (kge_rank_eval).  usage: python tools/eval_timing.py [model] [batch]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dgl-ke_amd"))
import __graft_entry__  # noqa: E402

__graft_entry__.build()
from dglke_amd import eval as E  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "TransE_l2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n_ent, n_rel, D, gamma = 14951, 1345, 400, 19.9
rng = np.random.RandomState(0)
known = tuple(rng.randint(0, n, 592213) for n in (n_ent, n_rel, n_ent))
test = tuple(k[:50000] for k in known)
dev = "cuda:0"
torch.manual_seed(0)
emb_init = (gamma + 2.0) / D
ent = torch.empty(n_ent, D, device=dev).uniform_(-emb_init, emb_init)
rel = torch.empty(n_rel, D // 2 if model == "RotatE" else D, device=dev).uniform_(-emb_init, emb_init)
for it in range(3):
    t0 = time.perf_counter()
    filts = [E.build_filter(known[0], known[1], known[2], test[0], test[1], test[2], nh, n_rel) for nh in (True, False)]
    t1 = time.perf_counter()
    rk = E.Ranker(model, ent, rel, gamma, emb_init, batch)
    out = []
    for nh, f in zip((True, False), filts):
        out.append(rk.ranks(test[0], test[1], test[2], nh, f))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    m = E.metrics_from_ranks(torch.cat(out))
    t3 = time.perf_counter()
    print("%s batch %d pass %d: filter lists (host) %.3f s | ranking 2 x 50 000 x %d (device, incl. H2D of ids and lists) %.3f s = %.1f TFLOP/s | "
          "metrics %.3f s | MRR %.4f" % (model, batch, it, t1 - t0, n_ent, t2 - t1, 2 * 50000 * n_ent * D * 2 / (t2 - t1) / 1e12, t3 - t2, m["MRR"]))
print("rank checksum", int(torch.cat(out).to(torch.int64).sum()))
# what dglke_train's validations run: evaluate() with a cache - lists built on the device at the first call, reused afterwards
cache = {}
for it in range(3):
    t0 = time.perf_counter()
    m = E.evaluate(model, ent, rel, gamma, emb_init, test, known, batch=batch, cache=cache)
    torch.cuda.synchronize()
    print("evaluate(cache) call %d: %.3f s  MRR %.4f" % (it, time.perf_counter() - t0, m["MRR"]))

