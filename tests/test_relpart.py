"""`--rel_part`: dglke_amd.dist.soft_relation_partition against the reference's own SoftRelationPartition (dataloader/sampler.py:32-148,
what TrainDataset calls for `--rel_part`, sampler.py:363-365) - golden vectors from the unmodified reference
(tests/golden/gen_golden_relpart.py), edge for edge; plus the properties the sharded trainer relies on."""
import glob
import os

import numpy as np
import pytest

from dglke_amd import dist as kd

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "relpart", "relpart_*.npz")))


def test_golden_files_present():
    assert len(GOLD) >= 10


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_soft_partition_matches_reference(path):
    g = np.load(path)
    W = int(g["world"])
    part, rel_parts, cross = kd.soft_relation_partition(g["rels"], W)
    assert np.array_equal(part, g["part"])                     # every edge on the rank the reference puts it on
    assert np.array_equal(cross, g["cross_rels"]) and bool(len(cross)) == bool(g["cross"])
    for k in range(W):
        assert np.array_equal(rel_parts[k], g["rel_parts_%d" % k])


@pytest.mark.parametrize("world", [1, 2, 5, 8])
def test_soft_partition_properties(world):
    rng = np.random.RandomState(world)
    p = 1.0 / np.arange(1, 61) ** 1.2
    rels = rng.choice(60, size=7001, p=p / p.sum())
    part, rel_parts, cross = kd.soft_relation_partition(rels, world)
    cnt = np.bincount(part, minlength=world)
    assert cnt.sum() == len(rels) and cnt.max() <= 1.1 * cnt.mean() + 1
    cross = set(cross.tolist())
    for r in np.unique(rels):
        ranks = np.unique(part[rels == r])
        assert (len(ranks) > 1) <= (r in cross)                # only the relations named as cross live on more than one rank
        for k in ranks:
            assert r in rel_parts[k]
    # a cross relation's edges go to the ranks in the order of the edge list, cnt // world + 1 at a time
    for r in cross:
        pr = part[rels == r]
        assert (np.diff(pr) >= 0).all()
        share = len(pr) // world + 1
        assert (np.bincount(pr, minlength=world)[:-1] <= share).all()


def test_whole_relation_partition_and_policy():
    rels = np.array([0] * 50 + [1] * 30 + [2] * 10 + [3] * 10)
    owner, part = kd.relation_partition(rels, 2)
    assert all(len(np.unique(part[rels == r])) == 1 for r in range(4))
    # policy: whole relations while they balance, the reference's split of the large ones when they do not
    mode, part2, owner2, cross2 = kd.choose_relation_partition(rels, 2, "auto")
    assert mode == "whole" and np.array_equal(part2, part) and len(cross2) == 0
    skew = np.array([0] * 90 + [1] * 5 + [2] * 5)
    mode, part3, owner3, cross3 = kd.choose_relation_partition(skew, 2, "auto")
    assert mode == "soft" and cross3.tolist() == [0] and abs(int((part3 == 0).sum()) - 50) <= 2
    assert owner3[0] == -2 and owner3[1] >= 0                   # a split relation has no single owner
    mode, part4, owner4, cross4 = kd.choose_relation_partition(skew, 2, "whole")
    assert mode == "whole" and len(cross4) == 0 and (part4[skew == 0] == part4[0]).all()
    mode, _, _, cross5 = kd.choose_relation_partition(rels, 2, "soft")
    assert mode == "soft" and 0 in cross5.tolist()
