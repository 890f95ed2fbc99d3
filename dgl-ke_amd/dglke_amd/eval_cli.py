"""dglke_eval: rank the test triples of a dataset with saved embeddings - the reference's evaluation entry point
(python/dglke/eval.py:39-110 flags; test() of train_pytorch.py:199-253 and KEModel.forward_test,
general_models.py:436-485, replaced by the on-device filtered ranking kge_rank_eval).  Reads the files dglke_train
(this one or the reference's) writes: <model_path>/<dataset>_<model>_{entity,relation}.npy (+ ...projection.npy for
TransR)."""
import argparse
import os
import sys
import time

import numpy as np
import torch as th

from ._lib import KgeError
from .kgdataset import get_dataset


class ArgParser(argparse.ArgumentParser):
    def __init__(self):
        super(ArgParser, self).__init__()
        a = self.add_argument
        a('--model_name', default='TransE', choices=['TransE', 'TransE_l1', 'TransE_l2', 'TransR', 'RESCAL', 'DistMult',
                                                     'ComplEx', 'RotatE', 'SimplE'])
        a('--data_path', type=str, default='data')
        a('--dataset', type=str, default='FB15k')
        a('--format', type=str, default='built_in')
        a('--data_files', type=str, default=None, nargs='+')
        a('--delimiter', type=str, default='\t')
        a('--model_path', type=str, default='ckpts')
        a('--batch_size_eval', type=int, default=8)
        a('--neg_sample_size_eval', type=int, default=-1)
        a('--neg_deg_sample_eval', action='store_true')
        a('--hidden_dim', type=int, default=256)
        a('-g', '--gamma', type=float, default=12.0)
        a('--eval_percent', type=float, default=1)
        a('--no_eval_filter', action='store_true')
        a('--gpu', type=int, default=[-1], nargs='+')
        a('--mix_cpu_gpu', action='store_true')           # accepted, no effect: the tables live in HBM
        a('-de', '--double_ent', action='store_true')
        a('-dr', '--double_rel', action='store_true')
        a('--num_proc', type=int, default=1)              # accepted, no effect: one process ranks on the GPU
        a('--num_thread', type=int, default=1)
        a('--seed', type=int, default=0)


def main(argv=None):
    from . import eval as kev
    args = ArgParser().parse_args(argv)
    args.eval_filter = not args.no_eval_filter
    if args.neg_deg_sample_eval:
        raise KgeError("--neg_deg_sample_eval is not available (uniform --neg_sample_size_eval or all entities)")
    if args.gpu[0] < 0:
        raise KgeError("dglke_eval ranks on the GPU only: pass --gpu <id> (there is no CPU fallback)")
    if not os.path.isdir(args.model_path):
        raise KgeError("No existing model_path: {}".format(args.model_path))
    dev = th.device("cuda", args.gpu[0])
    th.cuda.set_device(dev)
    ds = get_dataset(args.data_path, args.dataset, args.format, args.delimiter, args.data_files)
    if ds.test is None:
        raise KgeError("the dataset has no test split")
    model = 'TransE_l2' if args.model_name == 'TransE' else args.model_name
    stem = os.path.join(args.model_path, '{}_{}_'.format(args.dataset, args.model_name))

    def load(name):
        f = stem + name + '.npy' if name != 'projection' else stem[:-1] + 'projection.npy'
        if not os.path.exists(f):
            raise KgeError("missing embedding file {}".format(f))
        return th.from_numpy(np.load(f)).to(dev, th.float32).contiguous()
    ent, rel = load('entity'), load('relation')
    proj = load('projection') if model == 'TransR' else None
    d_e = args.hidden_dim * (2 if args.double_ent else 1)
    if ent.shape != (ds.n_entities, d_e):
        raise KgeError("entity embeddings are {} but the dataset / flags say {}".format(tuple(ent.shape), (ds.n_entities, d_e)))
    emb_init = (args.gamma + 2.0) / args.hidden_dim
    h, r, t = (np.asarray(x) for x in ds.test[:3])
    if args.eval_percent < 1:
        keep = np.random.RandomState(args.seed + 17).permutation(len(h))[:max(1, int(len(h) * args.eval_percent))]
        h, r, t = h[keep], r[keep], t[keep]
    known = None
    if args.eval_filter:
        parts = [p for p in (ds.train, ds.valid, ds.test) if p is not None]
        known = tuple(np.concatenate([np.asarray(p[k]) for p in parts]) for k in range(3))
    Eb = int(max(1, min(max(args.batch_size_eval, 4096), (1 << 31) // (4 * ds.n_entities), len(h))))
    if proj is not None:
        Eb = min(Eb, 64)
    start = time.time()
    metrics = kev.evaluate(model, ent, rel, args.gamma, emb_init, (h, r, t), known, batch=Eb, proj=proj,
                           n_cand=args.neg_sample_size_eval if args.neg_sample_size_eval > 0 else None,
                           chunk=args.batch_size_eval, seed=args.seed + 29)
    for k, v in metrics.items():
        print('[{}]{} average {}: {}'.format(0, 'Test', k, v))          # train_pytorch.py:236-247 format
    print('Test takes {:.3f} seconds'.format(time.time() - start))
    return metrics


if __name__ == '__main__':
    try:
        main()
    except KgeError as e:
        sys.exit("dglke_eval: %s" % e)
