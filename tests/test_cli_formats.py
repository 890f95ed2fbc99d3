"""Host logic of the dglke_train entry point: dataset formats (reference dataloader/KGDataset.py) and
the flag surface (utils.py:199-297, train.py:40-60).  CPU only."""
import json
import os

import numpy as np
import pytest

from dglke_amd import kgdataset as K


def _write(path, rows, delim="\t"):
    with open(path, "w") as f:
        for r in rows:
            f.write(delim.join(str(x) for x in r) + "\n")


def test_built_in_layout_roundtrip(tmp_path):
    tr = (np.array([0, 1, 2, 3]), np.array([0, 1, 0, 1]), np.array([1, 2, 3, 0]))
    va = (np.array([0]), np.array([1]), np.array([2]))
    te = (np.array([3]), np.array([0]), np.array([1]))
    K.write_built_in_layout(str(tmp_path), "FB15k", 4, 2, tr, va, te)
    ds = K.get_dataset(str(tmp_path), "FB15k", "built_in")
    assert (ds.n_entities, ds.n_relations) == (4, 2)
    for got, want in ((ds.train, tr), (ds.valid, va), (ds.test, te)):
        for g, w in zip(got, want):
            assert g.dtype == np.int64 and np.array_equal(g, w)
    assert ds.emap_fname == "entities.dict" and ds.rmap_fname == "relations.dict"
    with pytest.raises(FileNotFoundError):
        K.get_dataset(str(tmp_path), "wn18", "built_in")          # not unpacked, no download here
    with pytest.raises(ValueError):
        K.get_dataset(str(tmp_path), "Freebase", "built_in")


def test_raw_udd_builds_id_maps_in_order_of_appearance(tmp_path):
    _write(tmp_path / "tr.tsv", [("a", "likes", "b"), ("b", "likes", "c"), ("c", "knows", "a")])
    _write(tmp_path / "va.tsv", [("a", "knows", "d")])
    _write(tmp_path / "te.tsv", [("d", "likes", "a")])
    ds = K.get_dataset(str(tmp_path), "toy", "raw_udd_hrt", "\t", ["tr.tsv", "va.tsv", "te.tsv"])
    assert ds.entity2id == {"a": 0, "b": 1, "c": 2, "d": 3} and ds.relation2id == {"likes": 0, "knows": 1}
    assert np.array_equal(ds.train[0], [0, 1, 2]) and np.array_equal(ds.train[1], [0, 0, 1])
    assert np.array_equal(ds.valid[2], [3]) and np.array_equal(ds.test[0], [3])
    assert open(tmp_path / "entities.tsv").read().split("\n")[0] == "0\ta"
    # column order: 'raw_udd_trh' = tail, relation, head
    _write(tmp_path / "p.tsv", [("b", "likes", "a")])
    ds2 = K.get_dataset(str(tmp_path), "toy", "raw_udd_trh", "\t", ["p.tsv"])
    assert ds2.entity2id == {"a": 0, "b": 1} and np.array_equal(ds2.train[0], [0]) and np.array_equal(ds2.train[2], [1])
    with pytest.raises(ValueError):
        K.get_dataset(str(tmp_path), "FB15k", "raw_udd_hrt", "\t", ["tr.tsv"])


def test_udd_ids_edge_importance_and_checks(tmp_path):
    _write(tmp_path / "e.txt", [(i, "e%d" % i) for i in range(5)], "|")
    _write(tmp_path / "r.txt", [(i, "r%d" % i) for i in range(2)], "|")
    _write(tmp_path / "tr.txt", [(0, 1, 4, 0.5), (3, 0, 2, 2.0)], "|")
    ds = K.get_dataset(str(tmp_path), "toy", "udd_hrt", "|", ["e.txt", "r.txt", "tr.txt"], has_edge_importance=True)
    assert (ds.n_entities, ds.n_relations) == (5, 2) and ds.valid is None and ds.test is None
    assert np.array_equal(ds.train[0], [0, 3]) and np.allclose(ds.train[3], [0.5, 2.0])
    _write(tmp_path / "bad.txt", [(0, 1, 9)], "|")
    with pytest.raises(ValueError):
        K.get_dataset(str(tmp_path), "toy", "udd_hrt", "|", ["e.txt", "r.txt", "bad.txt"])
    _write(tmp_path / "bad2.txt", [("x", 1, 2)], "|")
    with pytest.raises(ValueError):
        K.get_dataset(str(tmp_path), "toy", "udd_hrt", "|", ["e.txt", "r.txt", "bad2.txt"])
    with pytest.raises(ValueError):
        K.get_dataset(str(tmp_path), "toy", "udd_hrt", "|", ["e.txt", "r.txt"])
    # id triples are parsed as integers straight away (round 6: no Python string per id); column orders, extra columns, padded
    # tokens (which only the string path accepts) and a multi-character delimiter give the same arrays
    _write(tmp_path / "trh.txt", [(4, 1, 0, "x"), (2, 0, 3, "y")], "|")
    ds = K.get_dataset(str(tmp_path), "toy", "udd_trh", "|", ["e.txt", "r.txt", "trh.txt"])
    assert np.array_equal(ds.train[0], [0, 3]) and np.array_equal(ds.train[1], [1, 0]) and np.array_equal(ds.train[2], [4, 2])
    with open(tmp_path / "pad.txt", "w") as f:
        f.write("0| 1|4\n3|0 |2\n")
    ds = K.get_dataset(str(tmp_path), "toy", "udd_hrt", "|", ["e.txt", "r.txt", "pad.txt"])
    assert np.array_equal(ds.train[0], [0, 3]) and np.array_equal(ds.train[1], [1, 0]) and ds.train[0].dtype == np.int64
    _write(tmp_path / "e2.txt", [(i, "e%d" % i) for i in range(5)], "||")
    _write(tmp_path / "r2.txt", [(i, "r%d" % i) for i in range(2)], "||")
    _write(tmp_path / "tr2.txt", [(0, 1, 4), (3, 0, 2)], "||")
    ds = K.get_dataset(str(tmp_path), "toy", "udd_hrt", "||", ["e2.txt", "r2.txt", "tr2.txt"])
    assert np.array_equal(ds.train[2], [4, 2])


def test_flag_surface_matches_reference_defaults():
    from dglke_amd import train as T
    a = T.ArgParser().parse_args([])
    want = dict(model_name="TransE", data_path="data", dataset="FB15k", format="built_in", save_path="ckpts",
                max_step=80000, batch_size=1024, batch_size_eval=8, neg_sample_size=256, neg_sample_size_eval=-1,
                eval_percent=1, log_interval=1000, eval_interval=10000, num_proc=1, num_thread=1,
                force_sync_interval=-1, hidden_dim=400, lr=0.01, gamma=12.0, adversarial_temperature=1.0,
                regularization_coef=0.000002, regularization_norm=3, loss_genre="Logsigmoid", margin=1.0, gpu=[-1])
    for k, v in want.items():
        assert getattr(a, k) == v, k
    for flag in ("no_save_emb", "neg_deg_sample", "neg_deg_sample_eval", "no_eval_filter", "test", "double_ent",
                 "double_rel", "neg_adversarial_sampling", "pairwise", "mix_cpu_gpu", "valid", "rel_part",
                 "async_update", "has_edge_importance"):
        assert getattr(a, flag) is False, flag
    b = T.ArgParser().parse_args("--model_name RotatE -de -adv -a 0.5 -rc 1e-7 -g 12 --gpu 0 -log 100 -pw -m 2".split())
    assert b.double_ent and b.neg_adversarial_sampling and b.adversarial_temperature == 0.5 and b.pairwise and b.margin == 2
    assert T.get_compatible_batch_size(1000, 200) == 1000 and T.get_compatible_batch_size(1000, 256) == 1024
    assert T.get_compatible_batch_size(100, 256) == 100


def test_cpu_run_is_refused_loudly(tmp_path):
    from dglke_amd import train as T
    from dglke_amd._lib import KgeError
    _write(tmp_path / "e.txt", [(i, i) for i in range(5)])
    _write(tmp_path / "r.txt", [(0, 0)])
    _write(tmp_path / "tr.txt", [(0, 0, 1), (1, 0, 2)])
    with pytest.raises(KgeError):
        T.main(["--format", "udd_hrt", "--dataset", "toy", "--data_path", str(tmp_path), "--data_files", "e.txt",
                "r.txt", "tr.txt", "--save_path", str(tmp_path / "ck"), "--max_step", "2"])     # --gpu defaults to -1


def test_dglke_eval_flags_and_gpu_only():
    """dglke_eval takes the reference's flags (eval.py:39-110) with its defaults, and refuses to run without a GPU."""
    from dglke_amd import eval_cli
    from dglke_amd._lib import KgeError
    a = eval_cli.ArgParser().parse_args([])
    assert (a.model_name, a.dataset, a.format, a.model_path) == ("TransE", "FB15k", "built_in", "ckpts")
    assert (a.batch_size_eval, a.neg_sample_size_eval, a.hidden_dim, a.gamma, a.eval_percent) == (8, -1, 256, 12.0, 1)
    assert a.gpu == [-1] and not a.double_ent and not a.no_eval_filter and a.num_proc == 1
    with pytest.raises(KgeError):
        eval_cli.main(["--gpu", "-1"])
