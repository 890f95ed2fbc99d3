"""the sampled-candidate evaluation protocol (--neg_sample_size_eval n: every chunk of --batch_size_eval test triples against its own n
uniformly drawn candidates, filtered): seconds per evaluation at FB15k's shape.  usage: python tools/eval_sampled_timing.py [n_cand] [chunk]"""
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

n_cand = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_ent, n_rel, D, gamma = 14951, 1345, 400, 19.9
rng = np.random.RandomState(0)
known = tuple(rng.randint(0, n, 592213) for n in (n_ent, n_rel, n_ent))
test = tuple(k[:50000] for k in known)
torch.manual_seed(0)
emb_init = (gamma + 2.0) / D
ent = torch.empty(n_ent, D, device="cuda").uniform_(-emb_init, emb_init)
rel = torch.empty(n_rel, D, device="cuda").uniform_(-emb_init, emb_init)
cache = {}
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = E.evaluate("TransE_l2", ent, rel, gamma, emb_init, test, known, batch=4096, n_cand=n_cand, chunk=chunk, seed=3, cache=cache)
    torch.cuda.synchronize()
    print("sampled evaluation (2 x 50 000 triples, %d candidates per chunk of %d, filtered) call %d: %.3f s  MRR %.4f"
          % (n_cand, chunk, it, time.perf_counter() - t0, m["MRR"]))
