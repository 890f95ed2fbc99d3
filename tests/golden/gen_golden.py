#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference (awslabs/dgl-ke) on CPU.

TEST INFRASTRUCTURE ONLY.  This script runs in the build container only (it needs
/root/reference, which does not exist on the GPU box); the .npz files it writes are committed
under tests/golden/ and are what the oracle (oracle/kge_oracle.py) and the HIP path are pinned
against.

How the reference is made importable here: `dgl` and `ogb` are not
installed, so stub modules are registered in sys.modules before `from dglke.models import
KEModel`; the stub only provides the thin tensor shim `dgl.backend` and placeholder classes, no
arithmetic.  The batches are duck-typed objects exposing the members the reference reads
(general_models.py:376-427, 548-569): PosG / NegG below.

Every case runs `KEModel.forward -> loss.backward() -> KEModel.update` (train_pytorch.py:141-152)
for a few steps and records, per step: the id batch, pos_score, neg_score, the log dict, the three
trace gradients (pos-entity [U,D], relation [B,D], negative-entity [C*N,D]); plus the initial and
final tables and state_sum.
"""
import os
import sys
import types
import json

import numpy as np
import torch as th

REF = os.environ.get("DGLKE_REFERENCE", "/root/reference/python")
OUT = os.path.dirname(os.path.abspath(__file__))


# the dgl / ogb stubs and the duck-typed subgraphs live in oracle/ref_stub.py (shared with bench.py's `cpu_baseline` of kind
# "reference", which drives the same unmodified reference files staged by oracle/make_ref.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
from oracle.ref_stub import Args, NegG, PosG, install_stubs      # noqa: E402


def make_args(case):
    a = Args()
    a.gpu = [-1]
    a.mix_cpu_gpu = False
    a.has_edge_importance = bool(case.get("impts", False))
    a.strict_rel_part = False
    a.soft_rel_part = False
    a.lr = case["lr"]
    a.neg_deg_sample = bool(case.get("neg_deg", False))      # general_models.py:396-402, 424-432
    a.neg_deg_sample_eval = False
    a.eval_filter = False
    a.regularization_coef = case["reg_coef"]
    a.regularization_norm = case["reg_norm"]
    a.loss_genre = case.get("loss_genre", "Logsigmoid")
    a.neg_adversarial_sampling = case["adv"]
    a.adversarial_temperature = case["adv_temp"]
    a.pairwise = case.get("pairwise", False)
    a.margin = case.get("margin", 1.0)
    a.num_thread = 1
    return a


def make_batch(rng, case, step):
    """Seeded id batch: uniform h,t,r; C*N negatives with replacement (sampler.py:376-419
    semantics restated in SURVEY.md 8c); odd steps corrupt tails, even steps heads
    (sampler.py:853-859, step counter starts at 1)."""
    B, N, chunk = case["B"], case["N"], case["chunk"]
    C = B // chunk
    h = rng.randint(0, case["n_ent"], size=B).astype(np.int64)
    t = rng.randint(0, case["n_ent"], size=B).astype(np.int64)
    r = rng.randint(0, case["n_rel"], size=B).astype(np.int64)
    neg = rng.randint(0, case["n_ent"], size=C * N).astype(np.int64)
    neg_head = (step % 2 == 0)
    nid, inv = np.unique(np.concatenate([h, t]), return_inverse=True)
    h_local, t_local = inv[:B].astype(np.int64), inv[B:].astype(np.int64)
    w = rng.uniform(0.5, 1.5, size=B).astype(np.float32) if case.get("impts", False) else None
    return dict(h=h, t=t, r=r, neg=neg, neg_head=neg_head, nid=nid.astype(np.int64),
                h_local=h_local, t_local=t_local, C=C, w=w)


class HeldQueue(object):
    """stands in for the `mp.Queue(1)` that `ExternalEmbedding.create_async_update` makes (tensor_models.py:364-368): `put` parks
    the item; nothing is applied until `land()` hands the parked items to the reference's own, unmodified `async_update` loop
    (tensor_models.py:136-175) - with the helper PROCESS the moment is a race (<= 1 step late), here it is fixed at its bound."""

    def __init__(self):
        self.items = []

    def put(self, item):
        self.items.append(item)


class _DrainQueue(object):
    def __init__(self, items):
        self.items = list(items) + [(None, None, None)]      # ... then the sentinel of finish_async_update (:370-375)

    def get(self):
        return self.items.pop(0)


def land(model, args):
    """run the reference's async_update body on everything parked in the entity table's queue"""
    from dglke.models.pytorch.tensor_models import async_update
    q = model.entity_emb.async_q
    if q.items:
        async_update(args, model.entity_emb, _DrainQueue(q.items))
        q.items = []


def run_case(name, case, fp64=False):
    """fp64=True (`--noise`): the SAME reference code on the same batches with its tables cast to float64 after construction
    (same float32-rounded initial values) - nothing is written; returns the final tables so that main() can record how far the
    float32 reference itself is from exact arithmetic for this case (tests/golden/noise.json: the yardstick of the GPU row
    tolerance, tests/test_gpu_parity.py)."""
    from dglke.models import KEModel
    th.manual_seed(case["seed"])
    rng = np.random.RandomState(case["seed"])
    args = make_args(case)
    model = KEModel(args, case["model"], case["n_ent"], case["n_rel"], case["hidden"],
                    case["gamma"], double_entity_emb=case["de"], double_relation_emb=case["dr"])
    if fp64:
        embs = [model.entity_emb, model.relation_emb] + ([model.score_func.projection_emb] if case["model"] == "TransR" else [])
        for e in embs:
            e.emb = e.emb.double()
            e.state_sum = e.state_sum.double()
    out = {}
    out["init_entity"] = model.entity_emb.emb.numpy().copy()
    out["init_relation"] = model.relation_emb.emb.numpy().copy()
    out["emb_init"] = np.float64(model.emb_init)
    transr = case["model"] == "TransR"
    if transr:      # third table: per-relation projection matrices owned by the score function (score_fun.py:114-118)
        out["init_projection"] = model.score_func.projection_emb.emb.numpy().copy()
    use_async = bool(case.get("async", False))
    if use_async:
        # --async_update (train_pytorch.py:120-121 -> KEModel.create_async_update, general_models.py:639-647: the ENTITY table
        # only): ExternalEmbedding.update then hands its traces to the queue (tensor_models.py:325-328) instead of applying them
        model.entity_emb.async_q = HeldQueue()
    for s in range(1, case["steps"] + 1):
        b = make_batch(rng, case, s)
        pos_g = PosG(th.from_numpy(b["nid"]), th.from_numpy(b["h_local"]),
                     th.from_numpy(b["t_local"]), th.from_numpy(b["r"]),
                     th.from_numpy(b["w"]) if b["w"] is not None else None)
        neg_g = NegG(th.from_numpy(b["neg"]), b["C"], case["chunk"], case["N"], b["neg_head"])
        loss, log = model.forward(pos_g, neg_g, -1)
        loss.backward()
        p = "s%d_" % s
        for k in ("h", "t", "r", "neg", "nid", "h_local", "t_local"):
            out[p + k] = b[k]
        if b["w"] is not None:
            out[p + "w"] = b["w"]
        out[p + "neg_head"] = np.int64(b["neg_head"])
        out[p + "pos_score"] = pos_g.edata["score"].detach().numpy().copy()
        # neg score recomputed without trace for recording (same tensors, no side effect)
        with th.no_grad():
            if case.get("neg_deg", False):    # predict_neg_score mutates neg_g.neg_sample_size in this mode: fresh object
                ng2 = NegG(th.from_numpy(b["neg"]), b["C"], case["chunk"], case["N"], b["neg_head"])
                ns = model.predict_neg_score(pos_g, ng2, trace=False, neg_deg_sample=True)
            else:
                ns = model.predict_neg_score(pos_g, neg_g, trace=False)
        out[p + "neg_score"] = ns.detach().numpy().copy()
        out[p + "log"] = np.array([log.get("pos_loss", np.nan), log.get("neg_loss", np.nan),
                                   log["loss"], log.get("regularization", 0.0)], dtype=np.float64)
        out[p + "loss_total"] = np.float64(loss.item())
        et = model.entity_emb.trace
        rt = model.relation_emb.trace
        assert len(et) == 2 and len(rt) == 1
        if transr:  # two traces: prepare() (positive projections) then the neg-prepare closure
            pt = model.score_func.projection_emb.trace
            assert len(pt) == 2
            out[p + "g_proj0"] = pt[0][1].grad.numpy().copy()
            out[p + "g_proj1"] = pt[1][1].grad.numpy().copy()
        out[p + "g_pos_ent"] = et[0][1].grad.numpy().copy()
        out[p + "g_neg"] = et[1][1].grad.numpy().copy()
        out[p + "g_rel"] = rt[0][1].grad.numpy().copy()
        if use_async:
            land(model, args)        # the helper finished step s-1's entity update while step s was being scored
        model.update(-1)
        out[p + "entity_state"] = model.entity_emb.state_sum.numpy().copy()
        out[p + "relation_state"] = model.relation_emb.state_sum.numpy().copy()
        if transr:
            out[p + "projection_state"] = model.score_func.projection_emb.state_sum.numpy().copy()
            out[p + "projection"] = model.score_func.projection_emb.emb.numpy().copy()
        if case.get("save_tables_each_step", True):
            out[p + "entity"] = model.entity_emb.emb.numpy().copy()
            out[p + "relation"] = model.relation_emb.emb.numpy().copy()
    if use_async:
        land(model, args)            # finish_async_update: the last pending update lands
    out["final_entity"] = model.entity_emb.emb.numpy().copy()
    out["final_relation"] = model.relation_emb.emb.numpy().copy()
    out["final_entity_state"] = model.entity_emb.state_sum.numpy().copy()
    out["final_relation_state"] = model.relation_emb.state_sum.numpy().copy()
    if fp64:
        return out
    out["case_json"] = np.array(json.dumps(case))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "loss", out["s%d_loss_total" % case["steps"]])


def base(model, **kw):
    c = dict(model=model, n_ent=60, n_rel=7, hidden=16, gamma=12.0, de=False, dr=False,
             B=16, N=4, chunk=4, lr=0.1, adv=True, adv_temp=1.0, reg_coef=1e-3, reg_norm=3,
             steps=3, seed=7)
    c.update(kw)
    return c


CASES = {
    # the five in-scope score functions (score_fun.py:40,222,289,451), both corruption modes
    # (3 steps: tail, head, tail), -adv on, regularisation on.
    "transe_l2_small": base("TransE_l2"),
    "transe_l1_small": base("TransE_l1", seed=8),
    "distmult_small": base("DistMult", gamma=6.0, hidden=16, lr=0.08, seed=9),
    "complex_small": base("ComplEx", gamma=6.0, de=True, dr=True, seed=10),
    "rotate_small": base("RotatE", gamma=12.0, de=True, seed=11),
    # SimplE (score_fun.py:556): [x_i | x_j] halves, clamp(+-20); the large-scale case saturates the clamp
    "simple_small": base("SimplE", gamma=6.0, de=True, dr=True, seed=41),
    "simple_ragged": base("SimplE", hidden=10, de=True, dr=True, B=30, N=7, chunk=10, seed=42),
    "simple_dups": base("SimplE", n_ent=9, n_rel=2, de=True, dr=True, steps=4, seed=43),
    "simple_plain": base("SimplE", hidden=16, adv=False, reg_coef=0.0, seed=44),
    "simple_clamped": base("SimplE", hidden=8, gamma=22.0, de=True, dr=True, lr=0.01, seed=45),
    # TransR (score_fun.py:110): per-relation projection matrices [ent_dim x rel_dim], L1 distance in relation space
    "transr_small": base("TransR", gamma=8.0, hidden=8, seed=61),
    "transr_ragged": base("TransR", gamma=8.0, hidden=6, B=30, N=7, chunk=10, seed=62),
    "transr_dups": base("TransR", gamma=8.0, hidden=8, n_ent=9, n_rel=2, steps=4, seed=63),
    "transr_mid": base("TransR", n_ent=200, n_rel=12, hidden=32, gamma=12.0, B=64, N=32, chunk=32, lr=0.05,
                       reg_coef=1e-6, steps=2, seed=64, save_tables_each_step=False),
    # RESCAL (score_fun.py:378): relation rows are [rel_dim x ent_dim] matrices
    "rescal_small": base("RESCAL", gamma=6.0, hidden=8, seed=51),
    "rescal_ragged": base("RESCAL", gamma=6.0, hidden=6, B=30, N=7, chunk=10, seed=52),
    "rescal_dups": base("RESCAL", gamma=6.0, hidden=8, n_ent=9, n_rel=2, steps=4, seed=53),
    "rescal_mid": base("RESCAL", n_ent=200, n_rel=12, hidden=32, gamma=12.0, B=64, N=32, chunk=32, lr=0.05,
                       reg_coef=1e-6, steps=2, seed=54, save_tables_each_step=False),
    # no adversarial weighting, no regularisation
    "transe_l2_noadv": base("TransE_l2", adv=False, reg_coef=0.0, seed=12),
    "distmult_noadv": base("DistMult", adv=False, reg_coef=0.0, seed=13),
    # duplicate-heavy: 9 entities / 2 relations -> exercises ExternalEmbedding.update
    # duplicate semantics (tensor_models.py:352-361)
    "transe_l2_dups": base("TransE_l2", n_ent=9, n_rel=2, steps=4, seed=14),
    "complex_dups": base("ComplEx", n_ent=9, n_rel=2, de=True, dr=True, steps=4, seed=15),
    "rotate_dups": base("RotatE", n_ent=9, n_rel=2, de=True, steps=4, seed=16),
    # ragged tile shapes: chunk != N, dims not multiples of 16/64
    "transe_l2_ragged": base("TransE_l2", hidden=20, B=30, N=7, chunk=10, seed=17),
    "transe_l1_ragged": base("TransE_l1", hidden=20, B=30, N=7, chunk=10, seed=18),
    "distmult_ragged": base("DistMult", hidden=20, B=30, N=7, chunk=10, seed=19),
    "complex_ragged": base("ComplEx", hidden=10, de=True, dr=True, B=30, N=7, chunk=10, seed=20),
    "rotate_ragged": base("RotatE", hidden=10, de=True, B=30, N=7, chunk=10, seed=21),
    # edge-importance weights (general_models.py:568, loss.py:72-75)
    "transe_l2_impts": base("TransE_l2", impts=True, seed=22),
    # a mid-size case shaped like the FB15k config (chunk == N, several MFMA tiles, D=64)
    "transe_l2_mid": base("TransE_l2", n_ent=400, n_rel=30, hidden=64, gamma=19.9, B=96, N=32,
                          chunk=32, lr=0.25, reg_coef=1e-9, steps=2, seed=23,
                          save_tables_each_step=False),
    "distmult_mid": base("DistMult", n_ent=400, n_rel=30, hidden=64, gamma=30.0, B=96, N=32,
                         chunk=32, lr=0.08, reg_coef=2e-6, steps=2, seed=24,
                         save_tables_each_step=False),
    "complex_mid": base("ComplEx", n_ent=400, n_rel=30, hidden=32, gamma=14.0, de=True, dr=True,
                        B=96, N=32, chunk=32, lr=0.1, reg_coef=2e-6, steps=2, seed=25,
                        save_tables_each_step=False),
    "rotate_mid": base("RotatE", n_ent=400, n_rel=30, hidden=32, gamma=12.0, de=True,
                       B=96, N=32, chunk=32, lr=0.01, reg_coef=1e-7, steps=2, seed=26,
                       save_tables_each_step=False),
    "transe_l1_mid": base("TransE_l1", n_ent=400, n_rel=30, hidden=64, gamma=16.0, B=96, N=32,
                          chunk=32, lr=0.01, reg_coef=1e-7, steps=2, seed=27,
                          save_tables_each_step=False),
    # other loss genres (loss.py:10-38, 44-61), pointwise
    "transe_l2_logistic": base("TransE_l2", loss_genre="Logistic", adv=False, seed=28),
    "transe_l2_hinge": base("TransE_l2", loss_genre="Hinge", margin=2.0, adv=False, seed=29),
    "distmult_bce": base("DistMult", loss_genre="BCE", gamma=1.0, adv=False, seed=30),
    "transe_l2_hinge_pairwise": base("TransE_l2", loss_genre="Hinge", margin=2.0, adv=False,
                                     pairwise=True, seed=31),
    "distmult_logistic_pairwise": base("DistMult", loss_genre="Logistic", adv=False,
                                       pairwise=True, seed=32),
    # --neg_deg_sample (general_models.py:396-402, 424-432): the chunk's own positives join the negatives, the
    # diagonal is masked to score 0; their gradients land in the POSITIVE trace
    "nd_transe_l2_small": base("TransE_l2", neg_deg=True, seed=71),
    "nd_transe_l1_small": base("TransE_l1", neg_deg=True, seed=72),
    "nd_distmult_small": base("DistMult", gamma=6.0, lr=0.08, neg_deg=True, seed=73),
    "nd_complex_small": base("ComplEx", gamma=6.0, de=True, dr=True, neg_deg=True, seed=74),
    "nd_rotate_small": base("RotatE", gamma=12.0, de=True, neg_deg=True, seed=75),
    "nd_simple_small": base("SimplE", gamma=6.0, de=True, dr=True, neg_deg=True, seed=76),
    "nd_transe_l2_dups": base("TransE_l2", n_ent=9, n_rel=2, steps=4, neg_deg=True, seed=77),
    "nd_rotate_ragged": base("RotatE", hidden=10, de=True, B=30, N=7, chunk=10, neg_deg=True, seed=78),
    "nd_distmult_noadv": base("DistMult", adv=False, reg_coef=0.0, neg_deg=True, seed=79),
    "nd_transe_l2_mid": base("TransE_l2", n_ent=400, n_rel=30, hidden=64, gamma=19.9, B=96, N=32,
                             chunk=32, lr=0.25, reg_coef=1e-9, steps=2, neg_deg=True, seed=80,
                             save_tables_each_step=False),
    "nd_rotate_mid": base("RotatE", n_ent=400, n_rel=30, hidden=32, gamma=12.0, de=True,
                          B=96, N=32, chunk=32, lr=0.01, reg_coef=1e-7, steps=2, neg_deg=True, seed=81,
                          save_tables_each_step=False),
    "nd_transe_l1_mid": base("TransE_l1", n_ent=400, n_rel=30, hidden=64, gamma=16.0, B=96, N=32,
                             chunk=32, lr=0.01, reg_coef=1e-7, steps=2, neg_deg=True, seed=82,
                             save_tables_each_step=False),
    # round 6: --neg_deg_sample with the relation-matrix models (the concat-and-mask of general_models.py:396-402, 417-432 sits in front
    # of head_neg_prepare / tail_neg_prepare, so TransR projects the in-batch rows like every other negative)
    "nd_transr_small": base("TransR", gamma=8.0, hidden=8, neg_deg=True, seed=83),
    "nd_transr_ragged": base("TransR", gamma=8.0, hidden=6, B=30, N=7, chunk=10, neg_deg=True, seed=84),
    "nd_transr_dups": base("TransR", gamma=8.0, hidden=8, n_ent=9, n_rel=2, steps=4, neg_deg=True, seed=85),
    "nd_transr_mid": base("TransR", n_ent=200, n_rel=12, hidden=32, gamma=12.0, B=64, N=32, chunk=32, lr=0.05,
                          reg_coef=1e-7, steps=2, neg_deg=True, seed=86),
    "nd_rescal_small": base("RESCAL", gamma=6.0, hidden=8, neg_deg=True, seed=87),
    "nd_rescal_ragged": base("RESCAL", gamma=6.0, hidden=6, B=30, N=7, chunk=10, neg_deg=True, seed=88),
    # --async_update: the reference's own async_update body applies the entity traces one step late (HeldQueue above); the
    # duplicate-heavy cases make the staleness visible in every row
    "async_transe_l2_small": base("TransE_l2", steps=4, seed=91, **{"async": True}),
    "async_transe_l2_dups": base("TransE_l2", n_ent=9, n_rel=2, steps=5, seed=92, **{"async": True}),
    "async_distmult_dups": base("DistMult", gamma=6.0, lr=0.08, n_ent=9, n_rel=2, steps=5, seed=93, **{"async": True}),
    "async_complex_small": base("ComplEx", gamma=6.0, de=True, dr=True, steps=4, seed=94, **{"async": True}),
    "async_rotate_dups": base("RotatE", n_ent=9, n_rel=2, de=True, steps=5, seed=95, **{"async": True}),
    "async_transe_l1_ragged": base("TransE_l1", hidden=20, B=30, N=7, chunk=10, steps=4, seed=96, **{"async": True}),
    "async_transe_l2_mid": base("TransE_l2", n_ent=400, n_rel=30, hidden=64, gamma=19.9, B=96, N=32, chunk=32, lr=0.25,
                                reg_coef=1e-9, steps=4, seed=97, save_tables_each_step=False, **{"async": True}),
}


def main():
    install_stubs()
    sys.path.insert(0, REF)
    th.set_num_threads(1)
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--noise" in sys.argv:
        # distance of the float32 reference (the committed goldens) from the same code in float64, per case: max |difference| of
        # the final tables and of every recorded per-step table, in units of lr
        noise = {}
        if only and os.path.exists(os.path.join(OUT, "noise.json")):      # named cases only: the other entries stay as they are
            with open(os.path.join(OUT, "noise.json")) as f:
                noise = json.load(f)
        for name, case in CASES.items():
            if only and name not in only:
                continue
            z = np.load(os.path.join(OUT, name + ".npz"))
            o64 = run_case(name, case, fp64=True)
            rec = {}
            for tab in ("entity", "relation"):
                keys = [k for k in z.files if (k == "final_" + tab or (k.startswith("s") and k.endswith("_" + tab)))]
                rec[tab] = max(float(np.abs(z[k].astype(np.float64) - o64[k]).max()) for k in keys) / case["lr"]
            noise[name] = rec
            print("%-40s entity %.3e lr   relation %.3e lr" % (name, rec["entity"], rec["relation"]))
        with open(os.path.join(OUT, "noise.json"), "w") as f:
            json.dump(noise, f, indent=1, sort_keys=True)
        return
    for name, case in CASES.items():
        if only and name not in only:
            continue
        run_case(name, case)


if __name__ == "__main__":
    main()
