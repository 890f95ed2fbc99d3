"""`dglke_train`-compatible entry point (reference: python/dglke/train.py:40-330 + the common flags of
python/dglke/utils.py:199-297): same flag names, same dataset formats, same log-line formats, same
output files (`<save_path>/<model>_<dataset>_<n>/<dataset>_<model>_{entity,relation}.npy` +
config.json, utils.py:35-49) so that the reference's `dglke_eval` / `dglke_predict` can consume the
embeddings.

What is different underneath: the per-step loop `sample -> forward -> backward -> update`
(train_pytorch.py:132-152) is ONE fused HIP step (`kge_step_fused`) fed by the on-device sampler,
replayed from a hipGraph; validation / test are one `kge_rank_eval` call per corruption mode instead
of the per-triple Python loop.  There is no CPU training path: `--gpu` must name a GPU.

    python -m dglke_amd.train --model_name TransE_l2 --dataset FB15k --data_path data --gpu 0 \\
        --batch_size 1000 --neg_sample_size 200 --hidden_dim 400 --gamma 19.9 --lr 0.25 \\
        --max_step 24000 --log_interval 1000 --batch_size_eval 16 -adv --regularization_coef 1e-9 --test
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch as th

from . import _lib
from ._lib import KgeError
from .kgdataset import get_dataset


class ArgParser(argparse.ArgumentParser):
    """flags of utils.py:199-297 (CommonArgParser) + train.py:40-60 (ArgParser)."""

    def __init__(self):
        super(ArgParser, self).__init__(prog="dglke_train")
        a = self.add_argument
        a('--model_name', default='TransE', choices=['TransE', 'TransE_l1', 'TransE_l2', 'TransR', 'RESCAL',
                                                     'DistMult', 'ComplEx', 'RotatE', 'SimplE'])
        a('--data_path', type=str, default='data')
        a('--dataset', type=str, default='FB15k')
        a('--format', type=str, default='built_in')
        a('--data_files', type=str, default=None, nargs='+')
        a('--delimiter', type=str, default='\t')
        a('--save_path', type=str, default='ckpts')
        a('--no_save_emb', action='store_true')
        a('--max_step', type=int, default=80000)
        a('--batch_size', type=int, default=1024)
        a('--batch_size_eval', type=int, default=8)
        a('--neg_sample_size', type=int, default=256)
        a('--neg_deg_sample', action='store_true')
        a('--neg_deg_sample_eval', action='store_true')
        a('--neg_sample_size_eval', type=int, default=-1)
        a('--eval_percent', type=float, default=1)
        a('--no_eval_filter', action='store_true')
        a('-log', '--log_interval', type=int, default=1000)
        a('--eval_interval', type=int, default=10000)
        a('--test', action='store_true')
        a('--num_proc', type=int, default=1)
        a('--num_thread', type=int, default=1)
        a('--force_sync_interval', type=int, default=-1,
          help='multi-GPU: every trainer waits at a barrier every this many steps (reference flag; -1: never)')
        a('--hidden_dim', type=int, default=400)
        a('--lr', type=float, default=0.01)
        a('-g', '--gamma', type=float, default=12.0)
        a('-de', '--double_ent', action='store_true')
        a('-dr', '--double_rel', action='store_true')
        a('-adv', '--neg_adversarial_sampling', action='store_true')
        a('-a', '--adversarial_temperature', default=1.0, type=float)
        a('-rc', '--regularization_coef', type=float, default=0.000002)
        a('-rn', '--regularization_norm', type=int, default=3)
        a('-pw', '--pairwise', action='store_true')
        a('--loss_genre', default='Logsigmoid', choices=['Hinge', 'Logistic', 'Logsigmoid', 'BCE'])
        a('-m', '--margin', type=float, default=1.0)
        a('--gpu', type=int, default=[-1], nargs='+')
        a('--mix_cpu_gpu', action='store_true')
        a('--valid', action='store_true')
        a('--rel_part', action='store_true',
          help='multi-GPU a2a mode: split the training triples BY RELATION over the trainers (see --rel_part_policy); while every '
               'relation lives on one trainer its row is updated only there - no relation exchange')
        a('--rel_part_policy', default='auto', choices=['auto', 'soft', 'whole'],
          help='whole: whole relations, most frequent first, to the trainer with the fewest edges (never split: a relation with more '
               'than 1 / trainers of the edges unbalances the split).  soft: the reference\'s partition (SoftRelationPartition: '
               'relations with more than min(5 %%, 1 / trainers) of the edges are dealt evenly over all trainers) - even edge shares; '
               'the relation gradients are then all-gathered and applied by every trainer.  auto (default): whole while the fullest '
               'trainer stays within 1.1 x the mean edge share, soft otherwise')
        a('--async_update', action='store_true')
        a('--has_edge_importance', action='store_true',
          help='train.txt carries a 4th column of edge weights (reference flag).  Batches then come from the host sampler; with several '
               'GPUs both modes take them step by step (eager launches)')
        # additions of this build
        a('--async_update_rel', action='store_true',
          help='with --async_update: defer the relation-table update by one step as well (the reference defers the entity '
               'table only); the next step\'s gather then shares a launch with this step\'s backward (fastest mode)')
        a('--async_update_pipeline', action='store_true',
          help='with --async_update (entity table only, the reference\'s semantics): run the one-step-stale pipeline anyway.  By '
               'default the flag maps onto the STRICT step, which is faster on this GPU than a pipeline that must still land the '
               'relation trace between two steps (profiles/r03_v3_workloads_kernels.txt: 32.9 vs 30.7 us per cfg-T step) and is '
               'within the licence of the flag (staleness <= 1 step; here 0)')
        a('--dist_slack', type=float, default=None,
          help='--dist_mode a2a: initial capacity of an owner bucket as a multiple of the mean share of a batch\'s unique entities '
               '(default 1.5, or KGE_DIST_SLACK); buckets grow by themselves when a group of batches needs more')
        a('--dist_mode', default='a2a', choices=['a2a', 'p2p'],
          help='multi-GPU training (--gpu g0 g1 ...): a2a = entity table range-sharded, relation table replicated, RCCL '
               'all-to-all pull / push with owner-side Adagrad (parameter-server semantics); p2p = both tables sharded and '
               'mapped peer to peer (hipIpc), Hogwild across the trainers, no collective.  TransR and RESCAL train in both modes '
               'with the entity table sharded, relation rows / matrices and the projection table local to the trainers and the '
               'triples partitioned by relation (the reference\'s --rel_part layout)')
        a('--dist_schedule', default=None, choices=['sync', 'pull', 'overlap'],
          help='--dist_mode a2a: where a step\'s exchanges run.  sync: on the compute stream, every step pulls after its predecessor\'s '
               'update has landed (default without --async_update); pull / overlap (need --async_update: one-step-stale entity rows): the '
               'next step\'s pull, or every exchange, on a side stream next to the compute (overlap: default with --async_update).  '
               'With --graph_steps > 0 the group - kernels and RCCL collectives - replays from one hipGraph; --graph_steps 0 launches '
               'eagerly (the way out if a recorded schedule ever stalls on a new software stack)')
        a('--seed', type=int, default=0, help='seed of the table initialisation and of the device sampler')
        a('--graph_steps', type=int, default=100, help='steps per captured hipGraph (0: eager launches)')
        a('--target_mrr', type=float, default=None,
          help='with --valid: stop as soon as the validation MRR reaches this value and report the time')


def get_compatible_batch_size(batch_size, neg_sample_size):
    """utils.py:27-33"""
    if neg_sample_size < batch_size and batch_size % neg_sample_size != 0:
        old = batch_size
        batch_size = int(math.ceil(batch_size / neg_sample_size) * neg_sample_size)
        print('batch size ({}) is incompatible to the negative sample size ({}). Change the batch size to {}'.format(
            old, neg_sample_size, batch_size))
    return batch_size


def prepare_save_path(args):
    """train.py:62-72"""
    os.makedirs(args.save_path, exist_ok=True)
    folder = '{}_{}_'.format(args.model_name, args.dataset)
    n = len([x for x in os.listdir(args.save_path) if x.startswith(folder)])
    args.save_path = os.path.join(args.save_path, folder + str(n))
    os.makedirs(args.save_path, exist_ok=True)


def save_model(args, model, emap_file=None, rmap_file=None):
    """utils.py:35-49 (same keys, including the reference's 'emp_file' spelling)"""
    os.makedirs(args.save_path, exist_ok=True)
    print('Save model to {}'.format(args.save_path))
    model.save_emb(args.save_path, args.dataset)
    conf = dict(vars(args))
    conf.update({'emp_file': emap_file, 'rmap_file': rmap_file})
    with open(os.path.join(args.save_path, 'config.json'), 'w') as f:
        json.dump(conf, f, indent=4)


class _Lane(object):
    """one trainer: a StepEngine (sharing the tables), its sampler over its share of the training triples,
    its HIP stream and - device sampler - a hipGraph of [1 sampler launch + G steps].  `--num_proc K` on
    one GPU = K lanes running concurrently, lock-free on the same tables: the reference's multi-process
    Hogwild training (train.py:298-317, RandomPartition of the edges sampler.py:256-290) with the
    processes replaced by streams."""

    def __init__(self, trainer, k, engine, triples, weights):
        from .dataloader import DeviceSampler, UniformChunkedSampler
        a = trainer.args
        self.t, self.k, self.engine = trainer, k, engine
        self.stream = th.cuda.current_stream() if trainer.n_lanes == 1 else th.cuda.Stream(device=trainer.dev)
        B, N, chunk = a.batch_size, a.neg_sample_size, trainer.chunk
        n_ent = trainer.dataset.n_entities
        if trainer.device_sampler:
            self.sampler = DeviceSampler(triples[0], triples[1], triples[2], n_ent, B, N, trainer.dev,
                                         n_slots=max(2, a.graph_steps or 2), neg_chunk_size=chunk, seed=a.seed + 1000 * k)
        else:
            self.sampler = UniformChunkedSampler(triples[0], triples[1], triples[2], n_ent, B, N, trainer.dev,
                                                 neg_chunk_size=chunk, seed=a.seed + 1000 * k, edge_importance=weights)
        self._graph = None
        self._rem_graphs = {}
        self.dropin_logs = {}
        # --async_update (reference: tensor_models.py:136-175, general_models.py:639-647): the one-step-stale pipeline of
        # kge_step_async - the entity update of step s-1 shares a launch with the backward of step s; every captured /
        # enqueued group of steps ends with a flush
        self.async_update = bool(getattr(a, 'async_update', False)) and trainer.async_ok and trainer.async_pipeline

    def _steps(self, batches):
        eng = self.engine
        if self.async_update:
            eng.steps_async(batches)
        else:
            for b in batches:
                eng.step(b)

    def timed_step(self):
        """one strict step with the reference's four timers (train_pytorch.py:132-152): returns seconds per phase"""
        import time as _t
        with th.cuda.stream(self.stream):
            t0 = _t.time()
            if self.t.device_sampler:
                e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
                e0.record()
                b = self.sampler.sample(1)[0]
                e1.record()
                d = self.engine.step_timed(b)
                # one sampler launch builds a whole group of batches in the training loop (its latency does not depend on the
                # count): this step's share of a launch is 1 / group
                smp = e0.elapsed_time(e1) * 1e-3 / max(1, int(getattr(self.t.args, 'graph_steps', 1) or 1))
            else:
                b = self.sampler.next_batches(1)[0]
                smp = _t.time() - t0
                d = self.engine.step_timed(b)
        if d is not None:
            d['sample'] = smp
        return d

    def enqueue(self, n):
        """enqueue n steps on this lane's stream (no synchronisation)."""
        t, eng = self.t, self.engine
        with th.cuda.stream(self.stream):
            if not t.fused:
                for _ in range(n):           # drop-in path (autograd Functions over the modular kernels)
                    pos_g, neg_g = next(self.sampler)
                    loss, log = t.model.forward(pos_g, neg_g, t.args.gpu[0])
                    loss.backward()
                    t.model.update(t.args.gpu[0])
                    for k_, v_ in log.items():          # reference: logs.append(log), averaged at the log marks
                        self.dropin_logs[k_] = self.dropin_logs.get(k_, 0.0) + float(v_)
                return
            if not t.device_sampler:
                self._steps(self.sampler.next_batches(n))
                return
            G, smp, done = t.args.graph_steps, self.sampler, 0
            if G >= 2 and G % 2 == 0 and n >= G and smp.host_step % 2 == 1:
                if self._graph is None:
                    self._steps(smp.sample(G))        # eager warm-up (also allocates the workspace)
                    done += G
                    if n - done >= G:
                        self.stream.synchronize()
                        g = th.cuda.CUDAGraph()
                        with _lib.graph_capture(g, stream=self.stream if t.n_lanes > 1 else None):
                            self._steps(smp.sample(G))
                        self._graph = g
                        smp.host_step -= G            # the capture itself did not run the steps
                while self._graph is not None and n - done >= G:
                    self._graph.replay()
                    smp.host_step += G
                    done += G
            while done < n:                           # remainder (< G steps, or the parity of a full group is off)
                k = min(smp.n_slots, n - done)
                key = (k, smp.host_step % 2)          # neg_head of slot j is baked into the graph: same parity only
                g = self._rem_graphs.get(key)
                if g is None and self._graph is not None and len(self._rem_graphs) < 8:
                    # the log / eval marks repeat: record [1 sampler launch + k steps] once instead of paying
                    # ~6 eager launches per step every time (56 -> 44 us/step at log_interval 1000, eval 500)
                    self.stream.synchronize()
                    g = th.cuda.CUDAGraph()
                    with _lib.graph_capture(g, stream=self.stream if t.n_lanes > 1 else None):
                        self._steps(smp.sample(k))
                    smp.host_step -= k                # the capture itself did not run the steps
                    self._rem_graphs[key] = g
                if g is not None:
                    g.replay()
                    smp.host_step += k
                else:
                    self._steps(smp.sample(k))
                done += k


class Trainer(object):
    """the training loop of train_pytorch.py:105-196 over the fused step."""

    def __init__(self, args, dataset):
        from .engine import StepEngine
        from .general_models import KEModel
        self.args = args
        self.dataset = dataset
        if args.gpu[0] < 0:
            raise KgeError("dglke_amd trains on the GPU only: pass --gpu <id> (there is no CPU fallback)")
        th.cuda.set_device(args.gpu[0])
        self.dev = th.device("cuda", args.gpu[0])
        th.manual_seed(args.seed)
        self.model = KEModel(args, args.model_name, dataset.n_entities, dataset.n_relations, args.hidden_dim,
                             args.gamma, double_entity_emb=args.double_ent, double_relation_emb=args.double_rel)
        tr = dataset.train
        B, N = args.batch_size, args.neg_sample_size
        self.chunk = N if N <= B else B
        C = B // self.chunk
        # --neg_deg_sample runs on the fused step (KGE_FLAG_NEG_DEG_SAMPLE) for every model - round 6: TransR and RESCAL too (the
        # reference's concat-and-mask is model-agnostic, general_models.py:396-402, 417-432), also with --num_proc lanes
        self.fused = True
        self.step_flags = _lib.FLAG_NEG_DEG_SAMPLE if args.neg_deg_sample else 0
        if getattr(args, 'async_update', False) and getattr(args, 'async_update_rel', False):
            self.step_flags |= _lib.FLAG_ASYNC_REL
            self.model.engine.hp.flags |= _lib.FLAG_ASYNC_REL
        self.device_sampler = self.fused and not args.has_edge_importance and 2 * B + C * N <= 8192
        self.n_lanes = max(1, int(args.num_proc))
        self.async_ok = self.fused and args.model_name not in ('TransR', 'RESCAL')
        # entity-only deferral (the reference's --async_update) runs as the strict step unless the pipeline is asked for: with the
        # relation trace landing between two steps the pipeline needs one more launch than it hides
        self.async_pipeline = bool(getattr(args, 'async_update_rel', False) or getattr(args, 'async_update_pipeline', False))
        if getattr(args, 'async_update', False) and not self.async_ok:
            print('--async_update: not available for this model / option combination; running the strict step')
        elif getattr(args, 'async_update', False) and not self.async_pipeline:
            print('--async_update: running the strict step (no staleness; faster on this GPU than the entity-only one-step-stale '
                  'pipeline - pass --async_update_rel to defer the relation trace too, or --async_update_pipeline to force it)')
        if self.n_lanes > 1 and not self.fused:
            raise KgeError("--num_proc > 1 needs the fused step (not available with --neg_deg_sample)")
        # the lanes share the tables; every lane trains on its own random share of the triples
        m = self.model
        tables = (m.entity_emb.emb, m.entity_emb.state_sum, m.relation_emb.emb, m.relation_emb.state_sum)
        if args.model_name == 'TransR':       # the lanes share the projection table too (general_models.py:97-100)
            tables += (m.score_func.projection_emb.emb, m.score_func.projection_emb.state_sum)
        parts = np.array_split(np.random.RandomState(args.seed).permutation(len(tr[0])), self.n_lanes)
        if min(len(p) for p in parts) < B:
            # both samplers work on whole batches (the reference drops partial ones, dataloader/sampler.py:503-504): a trainer
            # whose share of the triples is smaller than one batch could never step
            raise KgeError("every trainer needs at least batch_size training triples: %d triples over %d trainer(s) < batch_size %d "
                           "- lower --batch_size or --num_proc" % (len(tr[0]), self.n_lanes, B))
        self.lanes = []
        for k in range(self.n_lanes):
            eng = m.engine if k == 0 else StepEngine(
                args.model_name, dataset.n_entities, dataset.n_relations, args.hidden_dim, args.gamma, args.lr,
                self.dev, args.double_ent, args.double_rel, args.neg_adversarial_sampling,
                args.adversarial_temperature, args.regularization_coef, args.regularization_norm, args.loss_genre,
                args.pairwise, args.margin, flags=self.step_flags, tables=tables)
            sel = parts[k] if self.n_lanes > 1 else slice(None)
            trip = tuple(np.asarray(x)[sel] for x in tr[:3])
            w = np.asarray(tr[3])[sel] if args.has_edge_importance else None
            self.lanes.append(_Lane(self, k, eng, trip, w))

    def _run(self, n):
        """n steps on every lane, concurrently (no synchronisation here)."""
        for lane in self.lanes:
            lane.enqueue(n)

    def evaluate(self, which, mode):
        from . import eval as kev
        args, ds = self.args, self.dataset
        trip = getattr(ds, which)
        if trip is None:
            raise KgeError("the dataset has no %s split" % which)
        h, r, t = (np.asarray(x) for x in trip[:3])
        if args.eval_percent < 1:
            rng = np.random.RandomState(args.seed + 17)
            keep = rng.permutation(len(h))[:max(1, int(len(h) * args.eval_percent))]
            h, r, t = h[keep], r[keep], t[keep]
        known = None
        if args.eval_filter:
            parts = [p for p in (ds.train, ds.valid, ds.test) if p is not None]
            known = tuple(np.concatenate([np.asarray(p[k]) for p in parts]) for k in range(3))
        m = self.model
        n_cand = m.n_entities
        Eb = int(max(1, min(max(args.batch_size_eval, 4096), (1 << 31) // (4 * n_cand), len(h))))
        proj = m.score_func.projection_emb.emb if args.model_name == 'TransR' else None
        if proj is not None:
            Eb = min(Eb, 64)                  # TransR projects every candidate with every test triple's matrix
        # (the split and the known set do not change during a run: filter lists and test ids stay on the device between validations)
        cache = self.__dict__.setdefault('_eval_cache', {}).setdefault(which, {})
        metrics = kev.evaluate(args.model_name, m.entity_emb.emb, m.relation_emb.emb, args.gamma, m.emb_init,
                               (h, r, t), known, batch=Eb, proj=proj, n_cand=args.neg_sample_size_eval,
                               chunk=args.batch_size_eval, seed=args.seed + 29, cache=cache)   # sampled candidates if < n_entities
        for k, v in metrics.items():
            print('[{}]{} average {}: {}'.format(0, mode, k, v))
        return metrics

    def train(self):
        args = self.args
        eng = self.model.engine
        keys = ['loss'] if args.pairwise else ['pos_loss', 'neg_loss', 'loss']
        if args.regularization_coef > 0 and args.regularization_norm > 0:
            keys.append('regularization')
        idx = {'pos_loss': 0, 'neg_loss': 1, 'loss': 2, 'regularization': 3}
        th.cuda.synchronize()
        train_start = start = time.time()
        step, t_train, reached = 0, 0.0, None
        marks = set()
        for iv in (args.log_interval, args.eval_interval if args.valid else 0):
            if iv and iv > 0:
                marks.update(range(iv, args.max_step + 1, iv))
        marks.add(args.max_step)
        # --force_sync_interval with --num_proc K lanes on one GPU: the lanes are streams of this process - the synchronise at a mark
        # IS the reference's barrier between its trainer processes (train_pytorch.py:181-187)
        fsi = int(getattr(args, 'force_sync_interval', -1) or -1)
        if fsi > 0 and self.n_lanes > 1:
            marks.update(range(fsi, args.max_step + 1, fsi))
        since_log = 0
        timed = None
        t_interval = 0.0
        for nxt in sorted(marks):
            n = nxt - step
            at_log = args.log_interval > 0 and nxt % args.log_interval == 0
            if n > 0:
                t0 = time.time()
                # the reference's four timers (train_pytorch.py:127-177): the LAST step before a log mark runs as four
                # phase groups with HIP events in between (same kernels, same result).  That one step is launched eagerly and
                # synchronised, so its absolute times are not those of the graph-replayed steps: only the SPLIT is taken from
                # it - the printed totals are the interval's measured training time divided in that step's proportions
                want_timers = at_log and self.fused and self.n_lanes == 1 and not self.lanes[0].async_update
                self._run(n - 1 if want_timers else n)
                if want_timers:
                    th.cuda.synchronize()
                    timed = self.lanes[0].timed_step()
                th.cuda.synchronize()
                t_train += time.time() - t0
                t_interval += time.time() - t0
                step, since_log = nxt, since_log + n
            if at_log and since_log:
                for lane in self.lanes:
                    if self.fused:
                        sums = lane.engine.read_loss_sums()
                        for k in keys:
                            print('[proc {}][Train]({}/{}) average {}: {}'.format(lane.k, step, args.max_step, k,
                                                                                sums[idx[k]] / since_log))
                    else:
                        for k, v in lane.dropin_logs.items():
                            print('[proc {}][Train]({}/{}) average {}: {}'.format(lane.k, step, args.max_step, k, v / since_log))
                        lane.dropin_logs = {}
                    print('[proc {}][Train] {} steps take {:.3f} seconds'.format(lane.k, since_log, time.time() - start))
                if timed is not None:
                    tot = sum(timed[k] for k in ('sample', 'forward', 'backward', 'update')) or 1.0
                    sc = t_interval / tot
                    print('[proc {}]sample: {:.3f}, forward: {:.3f}, backward: {:.3f}, update: {:.3f}'.format(
                        0, timed['sample'] * sc, timed['forward'] * sc, timed['backward'] * sc, timed['update'] * sc))
                    print('[proc {}](split of {:.3f} s over {} steps in the proportions of one phase-timed step)'.format(
                        0, t_interval, since_log))
                    timed = None
                t_interval = 0.0
                print('[proc {}]sample+forward+backward+update (fused HIP step{}{}): {:.3f}'.format(
                    0, ', --async_update pipeline' if self.lanes[0].async_update else '',
                    '' if self.n_lanes == 1 else ', %d concurrent trainers' % self.n_lanes, t_train))
                since_log, start = 0, time.time()
            if args.valid and step % args.eval_interval == 0 and step > 1 and self.dataset.valid is not None:
                valid_start = time.time()
                m = self.evaluate('valid', 'Valid')
                print('[proc {}]validation take {:.3f} seconds:'.format(0, time.time() - valid_start))
                if args.target_mrr is not None and reached is None and m['MRR'] >= args.target_mrr:
                    reached = (step, t_train)
                    print('[proc 0]validation MRR {:.4f} >= {:.4f} after {} steps, {:.3f} s of training'.format(
                        m['MRR'], args.target_mrr, step, t_train))
                    break
                # (no reset of `start` here: the reference's '[Train] N steps take' interval includes a validation that falls
                #  inside it, train_pytorch.py:168-176)
        print('proc {} takes {:.3f} seconds'.format(0, time.time() - train_start))
        return reached


class ShardedTrainer(object):
    """one of the `--gpu g0 g1 ...` trainer processes (reference: train.py:298-317, one process per GPU on
    tables in shared host memory).  Here the shared tables are the union of the GPUs' HBM, mapped peer to peer
    (dglke_amd/p2p.py): every process trains on its random share of the triples with the fused step
    (`kge_step_sharded`), lock-free across processes like the reference; the process group (gloo) is only
    used to exchange the hipIpc handles and for barriers."""

    def __init__(self, args, dataset, rank, world):
        from . import p2p
        from .engine import StepEngine
        self.args, self.dataset, self.rank, self.world = args, dataset, rank, world
        th.cuda.set_device(args.gpu[rank])
        self.dev = th.device("cuda", args.gpu[rank])
        B, N = args.batch_size, args.neg_sample_size
        self.chunk = N if N <= B else B
        self.fused, self.n_lanes, self.async_ok = True, 1, False
        # (edge importance / more than 8192 ids per batch: host-built batches, like the single-GPU trainer's host sampler path)
        self.device_sampler = 2 * B + (B // self.chunk) * N <= 8192 and not args.has_edge_importance
        if args.neg_deg_sample and args.model_name in ('TransR', 'RESCAL'):
            raise KgeError("--neg_deg_sample is not available for %s on sharded tables" % args.model_name)
        d_e = args.hidden_dim * (2 if args.double_ent else 1)
        d_r = args.hidden_dim * (2 if args.double_rel else 1)
        self.emb_init = (args.gamma + 2.0) / args.hidden_dim
        # TransR / RESCAL (round 6): the entity table is spread over the GPUs like every model's; the relation-side tables - relation
        # rows / matrices and TransR's projection table - are whole tables LOCAL to every trainer and the triples are partitioned BY
        # RELATION (whole relations, dist.relation_partition), so that a relation's rows are trained on exactly one GPU: the
        # reference's --rel_part layout, which its own multi-GPU TransR recipe passes (examples/freebase/multi_gpu.sh:80-89,
        # general_models.py:590-637).  The owners' rows are collected when the tables are read (sync_tables).
        self.rel_side_local = args.model_name in ('TransR', 'RESCAL')
        self.rel_owner, part = None, None
        if self.rel_side_local:
            from . import dist as kd
            self.rel_owner, edge_rank = kd.relation_partition(dataset.train[1], world)
            part = np.nonzero(edge_rank == rank)[0]
            cnt = np.bincount(edge_rank, minlength=world)
            if rank == 0:
                print("%s on %d GPUs: entity table sharded peer to peer, relation-side tables local, triples partitioned by relation "
                      "(whole relations; edges per trainer %s)" % (args.model_name, world, cnt.tolist()))
            if cnt.min() < B:
                raise KgeError("relation partition: trainer %d gets %d training triples, fewer than --batch_size %d"
                               % (int(cnt.argmin()), int(cnt.min()), B))
        self.tabs = p2p.ShardedTables(dataset.n_entities, dataset.n_relations, d_e,
                                      d_r * d_e if args.model_name == 'RESCAL' else d_r, self.dev, world, rank,
                                      rel_local=self.rel_side_local,
                                      proj_dim=d_e * d_r if args.model_name == 'TransR' else 0)
        if not self.tabs.probe():
            raise KgeError("peer mappings do not reach the other GPUs' memory")
        self.tabs.init_uniform(self.emb_init, args.seed)
        self.engine = StepEngine(args.model_name, dataset.n_entities, dataset.n_relations, args.hidden_dim, args.gamma,
                                 args.lr, self.dev, args.double_ent, args.double_rel, args.neg_adversarial_sampling,
                                 args.adversarial_temperature, args.regularization_coef, args.regularization_norm,
                                 args.loss_genre, args.pairwise, args.margin,
                                 flags=_lib.FLAG_NEG_DEG_SAMPLE if args.neg_deg_sample else 0, shards=self.tabs)
        tr = dataset.train
        if part is None:
            part = np.array_split(np.random.RandomState(args.seed).permutation(len(tr[0])), world)[rank]
        self.lane = _Lane(self, rank, self.engine, tuple(np.asarray(x)[part] for x in tr[:3]),
                          np.asarray(tr[3])[part] if args.has_edge_importance else None)

    def _enqueue(self, n):
        self.lane.enqueue(n)

    def sync_tables(self):
        """collective point in front of rank 0's validation / test / save: the peer-mapped tables need nothing; relation-side tables
        that are local to the trainers (TransR / RESCAL) are collected from the relations' owners into rank 0's."""
        if self.rel_side_local:
            th.cuda.synchronize()
            self.tabs.collect_relations(self.rel_owner)

    def close(self):
        self.tabs.close()

    def projection(self):
        """TransR: the projection table as rank 0 holds it after sync_tables()"""
        return self.tabs.proj_tab

    def full_tables(self):
        """the whole entity / relation tables read through the shard map (rank-local copies)."""
        ds = self.dataset
        ent = self.tabs.gather("ent", th.arange(ds.n_entities, device=self.dev))
        rel = self.tabs.rel_tab if self.tabs.rel_local else self.tabs.gather("rel", th.arange(ds.n_relations, device=self.dev))
        return ent, rel

    def evaluate(self, which, mode):
        from . import eval as kev
        args, ds = self.args, self.dataset
        trip = getattr(ds, which)
        if trip is None:
            raise KgeError("the dataset has no %s split" % which)
        h, r, t = (np.asarray(x) for x in trip[:3])
        if args.eval_percent < 1:
            keep = np.random.RandomState(args.seed + 17).permutation(len(h))[:max(1, int(len(h) * args.eval_percent))]
            h, r, t = h[keep], r[keep], t[keep]
        known = None
        if args.eval_filter:
            parts = [p for p in (ds.train, ds.valid, ds.test) if p is not None]
            known = tuple(np.concatenate([np.asarray(p[k]) for p in parts]) for k in range(3))
        ent, rel = self.full_tables()
        Eb = int(max(1, min(max(args.batch_size_eval, 4096), (1 << 31) // (4 * ds.n_entities), len(h))))
        proj = self.projection() if args.model_name == 'TransR' else None
        if proj is not None:
            Eb = min(Eb, 64)                  # TransR projects every candidate with every test triple's matrix
        cache = self.__dict__.setdefault('_eval_cache', {}).setdefault(which, {})
        metrics = kev.evaluate(args.model_name, ent, rel, args.gamma, self.emb_init, (h, r, t), known, batch=Eb, proj=proj,
                               n_cand=args.neg_sample_size_eval, chunk=args.batch_size_eval, seed=args.seed + 29, cache=cache)
        for k, v in metrics.items():
            print('[{}]{} average {}: {}'.format(self.rank, mode, k, v))
        return metrics

    def train(self):
        import torch.distributed as dist
        args, rank = self.args, self.rank
        keys = ['loss'] if args.pairwise else ['pos_loss', 'neg_loss', 'loss']
        if args.regularization_coef > 0 and args.regularization_norm > 0:
            keys.append('regularization')
        idx = {'pos_loss': 0, 'neg_loss': 1, 'loss': 2, 'regularization': 3}
        marks = set()
        for iv in (args.log_interval, args.eval_interval if args.valid else 0):
            if iv and iv > 0:
                marks.update(range(iv, args.max_step + 1, iv))
        marks.add(args.max_step)
        # --force_sync_interval (reference train_pytorch.py:181-187: every trainer waits at a barrier every so many steps, so that no
        # process of the lock-free shared-table mode runs far ahead of the others); the all-to-all mode is in step by construction
        fsi = int(getattr(args, 'force_sync_interval', -1) or -1)
        if fsi > 0:
            marks.update(range(fsi, args.max_step + 1, fsi))
        th.cuda.synchronize()
        dist.barrier()
        train_start = start = time.time()
        step = since_log = 0
        for nxt in sorted(marks):
            n = nxt - step
            if n > 0:
                self._enqueue(n)
                th.cuda.synchronize()
                step, since_log = nxt, since_log + n
            if fsi > 0 and step % fsi == 0:
                dist.barrier()
            if args.log_interval > 0 and step % args.log_interval == 0 and since_log:
                sums = self.engine.read_loss_sums()
                for k in keys:
                    print('[proc {}][Train]({}/{}) average {}: {}'.format(rank, step, args.max_step, k,
                                                                        sums[idx[k]] / since_log))
                print('[proc {}][Train] {} steps take {:.3f} seconds'.format(rank, since_log, time.time() - start))
                since_log, start = 0, time.time()
            if args.valid and step % args.eval_interval == 0 and step > 1 and self.dataset.valid is not None:
                dist.barrier()                   # like the reference: all trainers stop for the validation
                self.sync_tables()
                if rank == 0:
                    valid_start = time.time()
                    self.evaluate('valid', 'Valid')
                    print('[proc {}]validation take {:.3f} seconds:'.format(rank, time.time() - valid_start))
                dist.barrier()           # (`start` is not reset: the interval includes the validation, train_pytorch.py:168-176)
        th.cuda.synchronize()
        print('proc {} takes {:.3f} seconds'.format(rank, time.time() - train_start))
        dist.barrier()


class A2ATrainer(ShardedTrainer):
    """`--gpu g0 g1 ... --dist_mode a2a`: the partitioning BASELINE.json's north_star names - the entity table range-sharded
    over the GPUs, the relation table REPLICATED (every relation row local to every trainer: what the reference's relation
    partitioning is after, general_models.py:590-637), parameter-server semantics as collectives (dglke_amd/dist.py
    DistEngine: pull -> compute -> push, the owner applies the sparse Adagrad in rank order; reference:
    general_models.py:650-680, kvserver.py:41-51).  Collectives: librccl called directly when every rank has its own GPU;
    ranks that share a GPU (`--gpu 0 0`) exchange through the gloo group (dist.HostStagedComm).  Rank 0 gathers the shards
    for validation / test / saving."""

    def __init__(self, args, dataset, rank, world):
        from . import dist as kd
        from .dataloader import DeviceSampler
        from .engine import StepEngine
        self.args, self.dataset, self.rank, self.world = args, dataset, rank, world
        th.cuda.set_device(args.gpu[rank])
        self.dev = th.device("cuda", args.gpu[rank])
        B, N = args.batch_size, args.neg_sample_size
        self.chunk = N if N <= B else B
        self.fused, self.n_lanes, self.async_ok = True, 1, False
        # the on-device sampler builds the batches of a whole group ahead (group routing, one id exchange, group graphs); batches it
        # cannot build - edge importance, or more than 8192 ids per batch - come from the host sampler: plans built on the host,
        # routed and exchanged step by step, eager launches (the same sharded step; slower: the host builds a plan per step)
        self.device_sampler = 2 * B + (B // self.chunk) * N <= 8192 and not args.has_edge_importance
        # TransR / RESCAL (round 6): the relation side - relation rows / matrices, TransR's projection table - is applied IN PLACE on the
        # trainer that holds the relation's edges, so the triples are always partitioned by whole relations for these two (the
        # reference's own multi-GPU TransR recipe passes --rel_part, examples/freebase/multi_gpu.sh:80-89); only entity messages travel
        self.rel_side_local = args.model_name in ('RESCAL', 'TransR')
        if self.rel_side_local and args.neg_deg_sample:
            raise KgeError("--neg_deg_sample is not available for %s on more than one GPU" % args.model_name)
        d_e = args.hidden_dim * (2 if args.double_ent else 1)
        self.emb_init = (args.gamma + 2.0) / args.hidden_dim
        self.spec = kd.ShardSpec(dataset.n_entities, world, rank)
        th.manual_seed(args.seed + 7919 * (rank + 1))
        self.ent = th.empty(self.spec.n_local, d_e, dtype=th.float32, device=self.dev).uniform_(-self.emb_init, self.emb_init)
        self.ent_state = th.zeros(self.spec.n_local, dtype=th.float32, device=self.dev)
        th.manual_seed(args.seed)                       # identical relation replicas on every rank
        self.engine = StepEngine(args.model_name, 1, dataset.n_relations, args.hidden_dim, args.gamma, args.lr, self.dev,
                                 args.double_ent, args.double_rel, args.neg_adversarial_sampling,
                                 args.adversarial_temperature, args.regularization_coef, args.regularization_norm,
                                 args.loss_genre, args.pairwise, args.margin,
                                 flags=_lib.FLAG_NEG_DEG_SAMPLE if args.neg_deg_sample else 0)
        own_gpu = len(set(args.gpu)) == world
        self.comm = kd.make_comm() if own_gpu else kd.HostStagedComm()
        slack = args.dist_slack if getattr(args, 'dist_slack', None) else float(os.environ.get("KGE_DIST_SLACK", "1.5"))
        # --rel_part (the reference's multi-GPU recipes pass it, examples/freebase/multi_gpu.sh): the triples are split BY RELATION
        # (dist.choose_relation_partition: whole relations while that balances, else the reference's SoftRelationPartition with its
        # large relations dealt over all trainers).  While every relation lives on ONE trainer its row is updated there and nowhere
        # else - no relation exchange (dist.DistEngine rel_local); with split relations the relation gradients are all-gathered
        # and applied by every trainer like without --rel_part (exact for any edge split), only the edge shares are the reference's
        self.rel_part = bool(getattr(args, 'rel_part', False)) or self.rel_side_local
        tr = dataset.train
        self.rel_owner, self.rel_local, part = None, False, None
        if self.rel_part:
            mode, edge_rank, self.rel_owner, cross = kd.choose_relation_partition(
                tr[1], world, 'whole' if self.rel_side_local else getattr(args, 'rel_part_policy', 'auto'))
            self.rel_local = len(cross) == 0
            part = np.nonzero(edge_rank == rank)[0]
            cnt = np.bincount(edge_rank, minlength=world)
            if rank == 0:
                print("relation partition (%s): %d relations over %d trainers, edges per trainer %s%s" % (
                    mode, int((self.rel_owner != -1).sum()), world, cnt.tolist(),
                    "" if self.rel_local else "; %d relations split over the trainers: relation gradients all-gathered" % len(cross)))
                if cnt.max() > 1.5 * cnt.mean():
                    # (--rel_part_policy whole only: a relation with more than 1 / world of the edges unbalances the trainers; every
                    # trainer runs max_step steps, so the light trainers revisit their edges more often than under the reference's split)
                    print("WARNING: --rel_part leaves trainer %d with %.2f x the mean edge share (whole relations only; the most "
                          "frequent relation holds %.1f %% of the edges; --rel_part_policy soft splits it)"
                          % (int(cnt.argmax()), cnt.max() / cnt.mean(), 100.0 * np.bincount(np.asarray(tr[1])).max() / len(tr[1])))
            if cnt.min() < B:            # the partition is the same on every rank: every rank sees the short one and stops HERE, before
                raise KgeError("--rel_part: trainer %d gets %d training triples, fewer than --batch_size %d (%d relations over %d "
                               "trainers)" % (int(cnt.argmin()), int(cnt.min()), B, int((self.rel_owner != -1).sum()), world))   # any collective
        self.de = kd.DistEngine(self.engine, self.spec, self.ent, self.ent_state, comm=self.comm, slack=slack,
                                rel_local=self.rel_local,
                                ue_bound=None if self.device_sampler else 2 * B + (B // self.chunk) * N)
        # exchanges may overlap the steps only under the staleness --async_update licenses (tensor_models.py:136-175): then push,
        # owner-side apply and the pull of step s+2 run on a side stream next to step s+1 (DistEngine._steps_overlapped: entity rows
        # exactly one step stale, relation rows current; KGE_DIST_PIPELINE=1: the pull only, the same tables bit for bit);
        # without the flag every step gathers after its predecessor's update has landed, like the reference
        self.pipelined = bool(getattr(args, 'async_update', False))
        if self.pipelined and os.environ.get("KGE_DIST_PIPELINE", "overlap") == "overlap":
            self.pipelined = "overlap"
        # --dist_schedule (the schedule used to be reachable through an environment variable only): sync = every step pulls
        # after its predecessor's update (what runs without --async_update), pull = the pull of step s+1 next to step s, overlap = every
        # exchange on a side stream (the default with --async_update); the two stale schedules need the flag's licence
        sched = getattr(args, 'dist_schedule', None)
        if sched:
            if sched != 'sync' and not getattr(args, 'async_update', False):
                raise KgeError("--dist_schedule %s computes on one-step-stale entity rows: it needs --async_update" % sched)
            self.pipelined = {'sync': False, 'pull': True, 'overlap': 'overlap'}[sched]
        if part is None:
            part = np.array_split(np.random.RandomState(args.seed).permutation(len(tr[0])), world)[rank]
        if len(part) < B:
            raise KgeError("--batch_size %d is larger than a trainer's share of the training triples (%d over %d trainers)"
                           % (B, len(tr[0]), world))
        h, r, t = (np.asarray(x)[part] for x in tr[:3])
        if self.device_sampler:
            self.sampler = DeviceSampler(h, r, t, dataset.n_entities, B, N, self.dev, n_slots=max(2, args.graph_steps or 2),
                                         neg_chunk_size=self.chunk, seed=args.seed + 1000 * rank)
        else:
            from .dataloader import UniformChunkedSampler
            w = np.asarray(tr[3])[part] if args.has_edge_importance else None
            self.sampler = UniformChunkedSampler(h, r, t, dataset.n_entities, B, N, self.dev, neg_chunk_size=self.chunk,
                                                 seed=args.seed + 1000 * rank, edge_importance=w)
        self._full = None
        if rank == 0:
            print("multi-GPU mode a2a: entity rows %d per GPU, relations replicated, collectives: %s"
                  % (self.spec.shard, type(self.comm).__name__))
            if self.rel_side_local:
                print("%s on %d GPUs (a2a): entity messages exchanged, relation-side tables (relation rows%s) applied in place on the "
                      "trainer that holds the relation's edges" % (args.model_name, world,
                                                                    ", projection matrices" if args.model_name == 'TransR' else " = matrices"))

    def _enqueue(self, n):
        """n sharded steps, eagerly (every rank issues the same collectives in the same order); inside a group of sampled
        batches the pull of step s+1 overlaps step s."""
        smp, done = self.sampler, 0
        log = (lambda m: print('[proc {}] {}'.format(self.rank, m))) if self.rank == 0 else None
        while not self.device_sampler and done < n:          # host-built plans: groups of <= 16 steps, each routed by its own step
            k = min(16, n - done)
            bs = smp.next_batches(k)
            self.de.ensure_capacity(bs, log)                 # (one device read per group; every rank decides alike)
            self.de.run_steps(bs, self.pipelined)
            done += k
        while done < n:
            k = min(smp.n_slots, n - done)
            dbs = smp.sample(k)
            # owner buckets are sized BEFORE the group runs (one small device read per group; every rank takes the same decision);
            # then the routing of the whole group, ONE exchange of its request ids and its steps - with their collectives - replay
            # from one hipGraph (dist.DistEngine.run_group; --graph_steps 0 or a host-staged transport: eager launches)
            self.de.run_group(dbs, log=log, graph=bool(self.args.graph_steps), pipelined=self.pipelined)
            done += k
        lost = self.de.check_overflow()
        if lost:                                 # cannot happen behind ensure_capacity: a bug, not a tuning matter
            raise KgeError('[proc {}] {} entities did not fit their owner bucket although the capacity was checked'.format(self.rank, lost))

    def sync_tables(self):
        import torch.distributed as dist
        th.cuda.synchronize()
        parts = [None] * self.world if self.rank == 0 else None
        dist.gather_object(self.ent.cpu(), parts, dst=0)
        if self.rel_local:                   # every replica holds the current rows of ITS relations only: collect them on rank 0
            from . import dist as kd
            kd.relation_rows_from_owners(self.engine.rel, self.engine.rel_state, self.rel_owner)
            if self.engine.proj is not None:     # TransR: the projection rows live with their relation
                kd.relation_rows_from_owners(self.engine.proj, self.engine.proj_state, self.rel_owner)
        if self.rank == 0:
            self._full = (th.cat(parts).to(self.dev), self.engine.rel)

    def full_tables(self):
        return self._full

    def projection(self):
        return self.engine.proj

    def close(self):
        self.de.close()              # the group graphs first, then the communicator (dist.RcclComm.close)


def _mp_worker(rank, args, port):
    import torch.distributed as dist
    world = len(args.gpu)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        init_time_start = time.time()
        if rank != 0:                            # one copy of the loader messages is enough
            sys.stdout = open(os.devnull, "w")
        dataset = get_dataset(args.data_path, args.dataset, args.format, args.delimiter, args.data_files,
                              args.has_edge_importance)
        sys.stdout = sys.__stdout__
        a2a = args.dist_mode == 'a2a'
        trainer = (A2ATrainer if a2a else ShardedTrainer)(args, dataset, rank, world)
        if rank == 0:
            print('Total initialize time {:.3f} seconds'.format(time.time() - init_time_start))
        start = time.time()
        trainer.train()
        failure = None
        trainer.sync_tables()
        if rank == 0:
            # whatever happens in rank 0's save / test section, every rank must still reach the barrier below
            # (a rank-0-only exception used to leave the others waiting for the gloo timeout)
            try:
                print('training takes {} seconds'.format(time.time() - start))
                ent, rel = trainer.full_tables()
                if not args.no_save_emb:
                    print('Save model to {}'.format(args.save_path))
                    np.save(os.path.join(args.save_path, '%s_%s_entity.npy' % (args.dataset, args.model_name)), ent.cpu().numpy())
                    np.save(os.path.join(args.save_path, '%s_%s_relation.npy' % (args.dataset, args.model_name)), rel.cpu().numpy())
                    if args.model_name == 'TransR':      # TransRScore.save (score_fun.py:190-191): <dataset>_<model>projection.npy
                        np.save(os.path.join(args.save_path, '%s_%sprojection.npy' % (args.dataset, args.model_name)),
                                trainer.projection().cpu().numpy())
                    conf = dict(vars(args))
                    conf.update({'emp_file': dataset.emap_fname, 'rmap_file': dataset.rmap_fname})
                    with open(os.path.join(args.save_path, 'config.json'), 'w') as f:
                        json.dump(conf, f, indent=4)
                if args.test:
                    start = time.time()
                    trainer.evaluate('test', 'Test')
                    print('testing takes {:.3f} seconds'.format(time.time() - start))
            except Exception as e:      # noqa: BLE001 - re-raised after the barrier
                failure = e
        dist.barrier()
        trainer.close()
        if failure is not None:
            raise failure
    finally:
        dist.destroy_process_group()


def launch_multi_gpu(args):
    """`--gpu g0 g1 ...`: one trainer process per listed GPU (the same GPU may be listed twice: the processes
    then share it, which is how the path is tested on a one-GPU box)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_mp_worker, args=(args, port), nprocs=len(args.gpu), join=True)


def main(argv=None):
    args = ArgParser().parse_args(argv)
    prepare_save_path(args)
    init_time_start = time.time()
    if len(args.gpu) > 1:                        # multi-GPU: one process per GPU on peer-to-peer shared tables
        if min(args.gpu) < 0:
            raise KgeError("dglke_amd trains on the GPU only: pass --gpu <ids> (there is no CPU fallback)")
        if args.log_interval <= 0:
            raise KgeError("--log_interval must be positive")
        if args.num_proc > len(args.gpu):            # reference: several trainer processes per GPU (train.py:94-100, 115-119)
            if args.num_proc % len(args.gpu):
                raise KgeError("--num_proc should be a multiple of the number of GPUs")
            args.gpu = [g for g in args.gpu for _ in range(args.num_proc // len(args.gpu))]
        args.batch_size = get_compatible_batch_size(args.batch_size, args.neg_sample_size)
        args.eval_filter = not args.no_eval_filter
        args.soft_rel_part = args.strict_rel_part = False
        launch_multi_gpu(args)
        return None
    if args.log_interval <= 0:
        raise KgeError("--log_interval must be positive")
    dataset = get_dataset(args.data_path, args.dataset, args.format, args.delimiter, args.data_files,
                          args.has_edge_importance)
    if args.test and dataset.test is None:
        raise KgeError("--test: the dataset has no test split")
    if args.neg_sample_size_eval < 0:
        args.neg_sample_size_eval = dataset.n_entities
    args.batch_size = get_compatible_batch_size(args.batch_size, args.neg_sample_size)
    args.batch_size_eval = get_compatible_batch_size(args.batch_size_eval, args.neg_sample_size_eval)
    args.eval_filter = not args.no_eval_filter
    if args.neg_deg_sample_eval:
        assert not args.eval_filter, "if negative sampling based on degree, we can't filter positive edges."
    args.soft_rel_part = args.strict_rel_part = False     # one replicated HBM relation table
    trainer = Trainer(args, dataset)
    print('Total initialize time {:.3f} seconds'.format(time.time() - init_time_start))
    start = time.time()
    trainer.train()
    print('training takes {} seconds'.format(time.time() - start))
    if not args.no_save_emb:
        save_model(args, trainer.model, emap_file=dataset.emap_fname, rmap_file=dataset.rmap_fname)
    if args.test:
        start = time.time()
        trainer.evaluate('test', 'Test')
        print('testing takes {:.3f} seconds'.format(time.time() - start))
    return trainer


if __name__ == '__main__':
    try:
        main()
    except KgeError as e:
        sys.exit("dglke_train: %s" % e)
