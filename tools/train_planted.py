#!/usr/bin/env python3
"""Train TransE_l2 on a planted knowledge graph with the fused HIP step and report filtered MRR over
time (time-to-MRR; real FB15k is not available offline).  usage: train_planted.py [--steps N] ..."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dgl-ke_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n_ent", type=int, default=2000); ap.add_argument("--n_rel", type=int, default=12)
    ap.add_argument("--edges", type=int, default=40000); ap.add_argument("--true_dim", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=64); ap.add_argument("--gamma", type=float, default=8.0)
    ap.add_argument("--lr", type=float, default=0.25); ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--neg", type=int, default=64); ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--eval_every", type=int, default=1000); ap.add_argument("--rc", type=float, default=1e-7)
    ap.add_argument("--model", default="TransE_l2")
    a = ap.parse_args()
    from dglke_amd.dataloader import UniformChunkedSampler
    from dglke_amd.general_models import KEModel
    from planted_kg import make_planted
    from test_gpu_end_to_end import Args, evaluate_mrr
    train, test = make_planted(a.n_ent, a.n_rel, a.edges, dim=a.true_dim, seed=1)
    test = test[:400]
    all_trip = np.concatenate([train, test])
    args = Args(gpu=[0], lr=a.lr, regularization_coef=a.rc, regularization_norm=3, neg_adversarial_sampling=True,
                adversarial_temperature=1.0, loss_genre="Logsigmoid", eval_filter=True, neg_deg_sample_eval=False)
    torch.manual_seed(0)
    de = a.model in ("ComplEx", "RotatE"); dr = a.model == "ComplEx"
    model = KEModel(args, a.model, a.n_ent, a.n_rel, a.hidden, a.gamma, de, dr)
    sampler = UniformChunkedSampler(train[:, 0], train[:, 1], train[:, 2], a.n_ent, a.batch, a.neg, "cuda:0", seed=5)
    print("train %d test %d triples; %s hidden %d gamma %g lr %g batch %d neg %d" % (
        len(train), len(test), a.model, a.hidden, a.gamma, a.lr, a.batch, a.neg))
    print("step 0 MRR %.4f" % evaluate_mrr(model, all_trip, test, a.n_ent))
    t_train = 0.0
    for s0 in range(0, a.steps, a.eval_every):
        batches = sampler.next_batches(a.eval_every)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in batches:
            model.engine.step(b)
        torch.cuda.synchronize(); t_train += time.perf_counter() - t0
        loss = model.engine.read_loss_sums()[2] / a.eval_every
        print("step %d train_time %.3fs loss %.4f MRR %.4f" % (s0 + a.eval_every, t_train, loss,
                                                              evaluate_mrr(model, all_trip, test, a.n_ent)))


if __name__ == "__main__":
    main()
