#!/usr/bin/env python3
"""FB15k-SHAPED planted knowledge graph (there is no network, so no real FB15k): 14 951 entities,
1 345 relations, 483 142 / 50 000 / 59 071 train / valid / test triples (docs/source/benchmarks.rst:10
of the reference).

Planted model (typed, exactly translational): entity = (type k, index j) with a hidden vector
c_k + u_j; relation r goes from type a_r to another type b_r with vector c_b - c_a; (h, r, t) is a
triple iff type(h) = a_r, type(t) = b_r and index(t) = index(h).  TransE can represent it exactly, the
relations are 1-to-1 (so both corruption modes have a unique answer) and the structure is shared across
types and indices, like the type structure of real knowledge graphs.  `--noise` replaces that fraction
of the tails by uniformly random entities, which bounds the reachable MRR below 1; entity ids are
shuffled so that they carry no structure; relation frequencies follow a long tail.
Written in the reference's `udd_hrt` id format:
    dglke_train --format udd_hrt --dataset fb15k_planted --data_path <out> \\
        --data_files entities.dict relations.dict train.txt valid.txt test.txt ...
usage: make_planted_fb15k.py OUT_DIR [--noise 0.1] [--types 30] [--seed 0]"""
import argparse
import os

import numpy as np

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--n_ent", type=int, default=14951)
    ap.add_argument("--n_rel", type=int, default=1345)
    ap.add_argument("--train", type=int, default=483142)
    ap.add_argument("--valid", type=int, default=50000)
    ap.add_argument("--test", type=int, default=59071)
    ap.add_argument("--noise", type=float, default=0.1)
    ap.add_argument("--types", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.RandomState(a.seed)
    K = a.types
    J = (a.n_ent + K - 1) // K
    slot = rng.permutation(K * J)[:a.n_ent]            # entity id -> (type, index) slot, shuffled
    etype, eidx = slot // J, slot % J
    ent_of_slot = np.full(K * J, -1, np.int64)
    ent_of_slot[slot] = np.arange(a.n_ent)
    # relation r: source type a_r -> target type b_r != a_r (several relations may share a type pair)
    ra = rng.randint(0, K, a.n_rel)
    rb = (ra + rng.randint(1, K, a.n_rel)) % K
    rels_of_type = [np.nonzero(ra == k)[0] for k in range(K)]
    total = a.train + a.valid + a.test
    # every valid (h, r): r leaves type(h) and the slot (b_r, index(h)) is occupied
    hs, rs, ts = [], [], []
    for k in range(K):
        ents = np.nonzero(etype == k)[0]
        for r in rels_of_type[k]:
            t = ent_of_slot[rb[r] * J + eidx[ents]]
            ok = t >= 0
            hs.append(ents[ok]); rs.append(np.full(int(ok.sum()), r)); ts.append(t[ok])
    trip = np.stack([np.concatenate(hs), np.concatenate(rs), np.concatenate(ts)], 1)
    assert len(trip) >= total, "only %d valid triples: lower --types" % len(trip)
    # long-tailed relation frequencies: sample without replacement with weight ~ 1/(1+rank)^0.7
    wgt = 1.0 / (1.0 + rng.permutation(a.n_rel)) ** 0.7
    keys = rng.rand(len(trip)) ** (1.0 / wgt[trip[:, 1]])
    trip = trip[np.argsort(-keys)[:total]]
    rng.shuffle(trip)
    noisy = rng.rand(total) < a.noise
    trip[noisy, 2] = rng.randint(0, a.n_ent, int(noisy.sum()))
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, "entities.dict"), "w") as f:
        f.writelines("%d\t/m/e%d\n" % (i, i) for i in range(a.n_ent))
    with open(os.path.join(a.out, "relations.dict"), "w") as f:
        f.writelines("%d\t/r/%d\n" % (i, i) for i in range(a.n_rel))
    o = 0
    for name, n in (("train.txt", a.train), ("valid.txt", a.valid), ("test.txt", a.test)):
        np.savetxt(os.path.join(a.out, name), trip[o:o + n], fmt="%d", delimiter="\t")
        o += n
    print("wrote %s: %d entities, %d relations, %d/%d/%d triples, noise %.2f" % (
        a.out, a.n_ent, a.n_rel, a.train, a.valid, a.test, a.noise))


if __name__ == "__main__":
    main()
