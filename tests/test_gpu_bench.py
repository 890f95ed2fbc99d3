"""bench.py contract (the driver runs `python bench.py --gpus 1 --steps K --warmup W` and reads ONE JSON line)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_skewed_triples_have_fb15k_like_hubs():
    """bench.synth_triples(skew=True): heavy-tailed ids - the most frequent relation ~3.6 % of the edges, the hub entity ~1 % of
    the heads and of the tails (what uniform ids never produce: rows with 20 - 40 contributions per 1000-edge batch)."""
    import bench
    w = bench.WORKLOADS["transe_l2_fb15k"]
    h, r, t = bench.synth_triples(w, 0, skew=True)
    n = len(h)
    top_r = np.bincount(r, minlength=w["n_rel"]).max() / n
    top_h = np.bincount(h, minlength=w["n_ent"]).max() / n
    top_t = np.bincount(t, minlength=w["n_ent"]).max() / n
    assert 0.025 < top_r < 0.05 and 0.006 < top_h < 0.015 and 0.006 < top_t < 0.015, (top_r, top_h, top_t)
    h2, r2, t2 = bench.synth_triples(w, 0, skew=True)
    assert np.array_equal(h, h2) and np.array_equal(r, r2) and np.array_equal(t, t2)
    hu, ru, tu = bench.synth_triples(w, 0)
    assert np.bincount(ru, minlength=w["n_rel"]).max() / n < 0.002


def test_sharded_workload_of_an_invocation():
    """`bench.py --gpus N` without --workload measures BASELINE configs[4] (Freebase-scale RotatE); the FB15k-shaped graph of
    configs[1] runs on the sharded engines only when it is asked for - `transe_l2_fb15k` is also bench.py's argparse default - and is
    not weak-scaled: the same 14 951 entities and 483 142 triples at every world size."""
    import types
    import bench_dist as bd
    mk = lambda wl, ex: types.SimpleNamespace(workload=wl, workload_explicit=ex)
    assert bd.dist_workload_name(mk("transe_l2_fb15k", False)) == "rotate_freebase"
    assert bd.dist_workload_name(mk("transe_l2_fb15k", True)) == "transe_l2_fb15k"
    assert bd.dist_workload_name(mk("transe_l2_freebase", False)) == "transe_l2_freebase"
    assert bd.dist_workload_name(mk("distmult_fb15k", True)) == "rotate_freebase"
    fb, fr = bd.DIST_WORKLOADS["transe_l2_fb15k"], bd.DIST_WORKLOADS["rotate_freebase"]
    old = {k: os.environ.pop(k, None) for k in ("KGE_DIST_ENTITIES", "KGE_DIST_TRIPLES")}
    try:
        assert [bd.dist_entities(fb, n) for n in (1, 2, 8)] == [14951] * 3
        assert [bd.dist_triples(fb, n) for n in (1, 2, 8)] == [483142, 241571, 60392]
        assert bd.dist_entities(fr, 8) == 86054152 and bd.dist_entities(fr, 1) == 10756769
        assert bd.dist_triples(fr, 8) == 338586276 // 8 and bd.dist_triples(fr, 1) == 48_000_000
    finally:
        os.environ.update({k: v for k, v in old.items() if v is not None})


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline", "--hogwild", "0", "--no-async-update", "--no-configs"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["unit"] == "edges/s" and d["dtype"] == "f32"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert 5e6 < d["value"] < 1e9 and abs(d["value"] - 1000 * 1e3 / d["ms_per_step"]) / d["value"] < 0.02
    # round 5: what ran untimed is said, and the 20 timed steps build the next 20 batches on their own launches (no sampler launch)
    assert d["warmup_effective"] == 120 and "kge_step_fused_sampling" in d["config"]["launch"]
    assert r.get("traffic_ratio") is None or r["traffic_ratio"] > 1.0
    # round 6: the launch description is built from what the TIMED groups did, and the line says whether the committed
    # profile summaries it quotes were taken from this build's source set
    assert d["config"]["sampler_groups_timed"] == {"fused": 1, "launch": 0, "none": 0} and "1 of the 1 timed groups: NO sampler launch" in d["config"]["launch"]
    assert isinstance(r["profile_matches_build"], bool) and len(r["build_source_hash"]) == 16


@pytest.mark.gpu
def test_bench_time_to_mrr_leg_reports_the_second_half_of_the_metric():
    """round 6: the bounded time-to-MRR@0.65 leg of the default line - dglke_train with the reference's FB15k
    TransE_l2 recipe on real FB15k when supplied, else on the labelled PLANTED graph."""
    import bench
    d = bench.time_to_mrr(timeout_s=280.0)
    assert "error" not in d, d
    assert d["graph"] in ("planted", "fb15k") and d["target_mrr"] == 0.65
    assert d["reached"] and d["mrr"] >= 0.65 and 500 <= d["steps"] <= 24000 and d["steps"] % 500 == 0
    assert 0 < d["train_seconds"] < 60 and d["eval_seconds"] > 0 and d["validations"] == d["steps"] // 500
    assert d["test_mrr"] is not None and ("PLANTED" in d["note"]) == (d["graph"] == "planted")


@pytest.mark.gpu
def test_bench_rank_eval_leg():
    """the default line's `rank_eval` object: one filtered evaluation at FB15k's shape as the trainers run it, with its rate against the
    fp32-MFMA peak (checked here on a smaller test set)."""
    import bench
    d = bench.rank_eval_leg(n_test=6000)
    assert d["triples"] == 12000 and d["candidates"] == 14951 and 0 < d["cached_call_s"] <= d["first_call_s"]
    assert 0.05 < d["frac_fp32_mfma_peak"] < 1.0 and abs(d["tflops"] - d["frac_fp32_mfma_peak"] * 157.3) < 0.2


@pytest.mark.gpu
def test_forced_exchange_line_carries_per_rank_diagnostics():
    """round 6: the multi-GPU worker's line explains itself - communicator creation time, bucket capacity and
    growth, microseconds per phase of the synchronous step (here: the N > 1 code path at world 1 with its RCCL exchanges kept)."""
    env = dict(os.environ)
    env.update({"KGE_DIST_MODE": "a2a", "KGE_DIST_FORCE_COLL": "1", "KGE_DIST_PIPELINE": "0", "KGE_DIST_OTHER_LEG": "0",
                "KGE_DIST_ENTITIES": "2000000", "KGE_DIST_TRIPLES": "2000000", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29700 + os.getpid() % 200)})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "20",
                          "--workload", "rotate_freebase", "--no-cpu-baseline"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    dg = d["config"]["diagnostics"]
    assert len(dg) == 1 and dg[0]["rank"] == 0 and dg[0]["communicator"] == "RcclComm" and dg[0]["communicator_create_s"] > 0
    ph = dg[0]["phase_us_per_step"]
    assert set(ph) >= {"route", "ids_a2a", "gather", "rows_a2a", "compute", "push", "apply", "steps"}, ph
    assert ph["compute"] > 20.0 and ph["steps"] == 20 and dg[0]["bucket_rows"] > 0


@pytest.mark.gpu
def test_fb15k_shaped_graph_through_the_sharded_path():
    """north_star: FB15k-shaped triples at 1/2/4/8 GPUs.  (a) `--workload transe_l2_fb15k` through the all-to-all engine with its RCCL
    exchanges kept at world 1 (the `transe_l2_fb15k_a2a_forced_exchange_relpart` leg of the default line); (b) the `fb15k_shaped` leg
    of a Freebase-scale line, with the N = 1 point of its own curve beside it."""
    env = dict(os.environ)
    env.update({"KGE_FORCE_DIST": "1", "KGE_DIST_MODE": "a2a", "KGE_DIST_FORCE_COLL": "1", "KGE_DIST_PIPELINE": "0",
                "KGE_DIST_REL_PART": "force", "KGE_DIST_OTHER_LEG": "0", "KGE_DIST_DIAG": "0", "WORLD_SIZE": "1", "RANK": "0",
                "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29900 + os.getpid() % 90)})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "20",
                          "--workload", "transe_l2_fb15k", "--no-cpu-baseline"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "FB15k-shaped" in d["config"]["workload"] and "n_ent=14951" in d["config"]["workload"] and d["config"]["mode"] == "a2a"
    assert d["config"]["relation_partition"] is True and d["config"]["bucket_overflows"] == 0 and d["config"]["bucket_rows"] > 2000
    assert d["steps"] == 40 and 20.0 < 1e3 * d["ms_per_step"] < 1000.0 and 0.3 < d["mean_loss"] < 2.0
    assert d["n1_same_workload"].endswith("--workload transe_l2_fb15k")
    # (b) a small Freebase-shaped headline with its secondary legs forced at world 1
    env.update({"KGE_DIST_OTHER_LEG": "force", "KGE_DIST_ENTITIES": "2000000", "KGE_DIST_TRIPLES": "2000000",
                "MASTER_PORT": str(29990 - os.getpid() % 90)})
    env.pop("KGE_FORCE_DIST")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "20",
                          "--workload", "rotate_freebase", "--no-cpu-baseline"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    fb = d["fb15k_shaped"]
    assert "error" not in fb, fb
    assert fb["steps"] == 40 and fb["bucket_overflows"] == 0 and fb["value"] > 1e6 and "n_ent=14951" in fb["workload"]
    assert 15.0 < fb["per_gpu_step_without_exchange"]["us_per_step"] < fb["us_per_step"]
    assert fb["schedule"] == "synchronous" and fb["other_schedule"]["schedule"] == "overlapped" and fb["other_schedule"]["steps"] == 40
    assert "error" not in fb["p2p"] and fb["p2p"]["steps"] == 40 and fb["p2p"]["us_per_step"] < fb["us_per_step"], fb["p2p"]


@pytest.mark.gpu
def test_bench_line_carries_the_other_baseline_configs():
    """the default line's `configs` object (BASELINE.json configs[2..4] as bounded legs in their own processes): checked here on the
    two FB15k / wikikg2 legs' machinery with a short leg (the Freebase-shard legs need 40 GB and run in the real bench only)."""
    import bench
    legs = bench.other_configs(steps=40, timeout_s=280.0, only=("distmult_fb15k",))
    d = legs["distmult_fb15k"]
    assert "error" not in d, d
    assert d["steps"] == 40 and 10.0 < d["us_per_step"] < 500.0 and d["algorithmic_bytes_per_step"] > 1e7 and 0 < d["frac"] < 1
