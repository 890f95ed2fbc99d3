#!/usr/bin/env python3
"""Golden vectors for `--rel_part`: the UNMODIFIED reference's SoftRelationPartition (dataloader/sampler.py:32-148 - the function
its TrainDataset calls for `--rel_part`, sampler.py:363-365) run on seeded relation-id lists.

TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference), writes tests/golden/relpart/relpart_*.npz, which
tests/test_relpart.py holds dglke_amd.dist.soft_relation_partition against.

The reference permutes the triples in place and returns index RANGES; the heads passed in are the edge numbers 0 .. E-1, so that the
heads of range k afterwards ARE the original edge numbers of partition k (in the order the reference put them there).
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = os.environ.get("DGLKE_REFERENCE", "/root/reference/python")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
from oracle.ref_stub import install_stubs      # noqa: E402

CASES = {
    # name: (edges, relations, ranks, exponent of the relation weights 1 / (k + 1)^a, seed)
    "relpart_zipf_w2": (3000, 40, 2, 1.0, 1),
    "relpart_zipf_w4": (5000, 200, 4, 1.1, 2),
    "relpart_zipf_w8": (6000, 300, 8, 0.9, 3),
    "relpart_flat_w4": (4000, 25, 4, 0.1, 4),          # nearly uniform relations: 1 / ranks of the edges is the smaller bound
    "relpart_hub_w8": (4000, 12, 8, 2.0, 5),           # one relation with most of the edges, a few edges per rank in the tail
    "relpart_small_w3": (97, 7, 3, 0.7, 6),            # odd sizes, shares that do not divide
    # many relations with EQUAL counts: the order among ties is the reference's np.flip(np.argsort(cnts)) - it decides who gets which rank
    "relpart_ties_w2": (240, 60, 2, 0.3, 7),
    "relpart_ties_w4": (400, 90, 4, 0.2, 8),
    "relpart_ties_w5": (333, 45, 5, 0.5, 9),
    "relpart_ties_w8": (900, 150, 8, 0.4, 10),
}


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from dglke.dataloader.sampler import SoftRelationPartition
    for name, (E, R, W, a, seed) in CASES.items():
        rng = np.random.RandomState(seed)
        p = 1.0 / np.arange(1, R + 1) ** a
        rels = rng.choice(R, size=E, p=p / p.sum()).astype(np.int64)
        heads = np.arange(E, dtype=np.int64)
        tails = rng.randint(0, 1000, size=E).astype(np.int64)
        rels_in = rels.copy()
        with contextlib.redirect_stdout(io.StringIO()):
            parts, rel_parts, cross, cross_rels = SoftRelationPartition((heads, rels, tails), W)
        part_of_edge = np.full(E, -1, np.int64)
        for k, rng_k in enumerate(parts):
            assert (np.diff(heads[rng_k]) > 0).all()          # a partition keeps the edge list's order
            part_of_edge[heads[rng_k]] = k
        assert (part_of_edge >= 0).all() and (rels_in[heads] == rels).all()
        out = dict(rels=rels_in, world=np.int64(W), part=part_of_edge, cross=np.bool_(cross),
                   cross_rels=np.asarray(cross_rels, np.int64))
        for k in range(W):
            out["rel_parts_%d" % k] = np.asarray(rel_parts[k], np.int64)
        np.savez_compressed(os.path.join(OUT, "relpart", name + ".npz"), **out)
        print(name, "edges per rank", np.bincount(part_of_edge, minlength=W).tolist(), "cross", cross_rels.tolist())


if __name__ == "__main__":
    main()
