#!/usr/bin/env python3
"""Golden vectors for the ranking evaluation: runs the UNMODIFIED reference `KEModel.forward_test`
(models/general_models.py:436-485) on CPU through the same dgl/ogb stubs as gen_golden.py.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference).  Each case: a
small random KG, a randomly initialised model, a batch of test triples scored against ALL entities
as corrupted heads and tails (one chunk, like EvalSampler with neg_sample_size_eval = -1,
dataloader/sampler.py:492-495), with and without the false-negative filter
(`neg_g.edata['bias']`, sampler.py:586-587).  Recorded: tables, triples, bias, scores, rankings."""
import json
import os
import sys

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run_case(name, case):
    from dglke.models import KEModel
    th.manual_seed(case["seed"])
    rng = np.random.RandomState(case["seed"])
    n_ent, n_rel = case["n_ent"], case["n_rel"]
    # random KG without duplicate triples
    trip = set()
    while len(trip) < case["n_triples"]:
        trip.add((int(rng.randint(n_ent)), int(rng.randint(n_rel)), int(rng.randint(n_ent))))
    trip = np.array(sorted(trip), np.int64)
    rng.shuffle(trip)
    test = trip[:case["E"]]
    args = G.make_args(dict(lr=0.1, reg_coef=0.0, reg_norm=3, adv=False, adv_temp=1.0))
    model = KEModel(args, case["model"], n_ent, n_rel, case["hidden"], case["gamma"],
                    double_entity_emb=case["de"], double_relation_emb=case["dr"])
    if case.get("scale"):       # spread the scores so that fewer candidates tie
        with th.no_grad():
            model.entity_emb.emb.mul_(case["scale"])
            model.relation_emb.emb.mul_(case["scale"])
    out = {"entity": model.entity_emb.emb.numpy().copy(), "relation": model.relation_emb.emb.numpy().copy(),
           "projection": (model.score_func.projection_emb.emb.numpy().copy() if case["model"] == "TransR"
                          else np.zeros(0, np.float32)),
           "emb_init": np.float64(model.emb_init), "known": trip, "test": test,
           "case_json": np.array(json.dumps(case))}
    known = set(map(tuple, trip.tolist()))
    h, r, t = test[:, 0], test[:, 1], test[:, 2]
    nid, inv = np.unique(np.concatenate([h, t]), return_inverse=True)
    E = test.shape[0]
    for mode in ("head", "tail"):
        neg_head = mode == "head"
        false_neg = np.zeros((E, n_ent), np.float32)
        for i in range(E):
            for e in range(n_ent):
                c = (e, int(r[i]), int(t[i])) if neg_head else (int(h[i]), int(r[i]), e)
                false_neg[i, e] = 1.0 if c in known else 0.0
        for filt in (True, False):
            args.eval_filter = filt
            pos_g = G.PosG(th.from_numpy(nid), th.from_numpy(inv[:E].astype(np.int64)),
                           th.from_numpy(inv[E:].astype(np.int64)), th.from_numpy(r.copy()))
            neg_g = G.NegG(th.arange(n_ent), 1, E, n_ent, neg_head)
            neg_g.edata["bias"] = th.from_numpy(-false_neg.reshape(-1))
            logs = []
            with th.no_grad():
                model.forward_test(pos_g, neg_g, logs, -1)
            ranks = np.array([int(round(l["MR"])) for l in logs], np.int64)
            out["%s_ranks_%s" % (mode, "filtered" if filt else "raw")] = ranks
        with th.no_grad():
            pos_g.ndata["emb"] = model.entity_emb(pos_g.ndata["id"], -1, False)
            pos_g.edata["emb"] = model.relation_emb(pos_g.edata["id"], -1, False)
            model.score_func.prepare(pos_g, -1, False)
            out[mode + "_pos_score"] = model.predict_score(pos_g).numpy().copy()
            out[mode + "_neg_score"] = model.predict_neg_score(pos_g, neg_g, trace=False).numpy().reshape(E, n_ent).copy()
        out[mode + "_false_neg"] = false_neg
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, out["tail_ranks_filtered"][:6], out["head_ranks_raw"][:6])


def base(model, **kw):
    c = dict(model=model, n_ent=45, n_rel=4, n_triples=220, E=14, hidden=16, gamma=8.0, de=False, dr=False, seed=31)
    c.update(kw)
    return c


CASES = {
    "eval_transe_l2": base("TransE_l2", scale=4.0),
    "eval_transe_l1": base("TransE_l1", scale=4.0, seed=32),
    "eval_distmult": base("DistMult", scale=6.0, seed=33),
    "eval_complex": base("ComplEx", de=True, dr=True, scale=6.0, seed=34),
    "eval_rotate": base("RotatE", de=True, scale=4.0, seed=35),
    "eval_simple": base("SimplE", de=True, dr=True, scale=6.0, seed=37),
    "eval_rescal": base("RESCAL", hidden=8, scale=3.0, seed=38),
    "eval_transr": base("TransR", hidden=8, scale=3.0, seed=39),
    # ragged: candidate count / dims that are not tile multiples
    "eval_transe_l2_ragged": base("TransE_l2", n_ent=37, hidden=20, E=9, scale=4.0, seed=36),
}


def main():
    G.install_stubs()
    sys.path.insert(0, G.REF)
    only = sys.argv[1:]
    for name, case in CASES.items():
        if only and name not in only:
            continue
        run_case(name, case)


if __name__ == "__main__":
    main()
