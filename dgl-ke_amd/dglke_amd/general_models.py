"""`KEModel` with the reference's public surface (models/general_models.py:183-680), running on
libkge_hip.  Two ways to run a training step:

* drop-in (reference loop, train_pytorch.py:141-152):
      loss, log = model.forward(pos_g, neg_g, gpu_id); loss.backward(); model.update(gpu_id)
  every op (gather, scores, loss, their backward, sparse Adagrad) is one C-ABI call; torch only
  carries the autograd graph between them.
* fused:  model.train_step(pos_g, neg_g)  ->  one C-ABI call (`kge_step_fused`) for the whole
  step, no host synchronisation; `model.engine.capture()` records many steps into a HIP graph.
"""
import torch as th

from . import ops
from . import _lib
from .engine import StepEngine
from .loss import LossGenerator
from .score_fun import (ComplExScore, DistMultScore, RESCALScore, RotatEScore, SimplEScore, TransEScore,
                        TransRScore)
from .tensor_models import (ExternalEmbedding, cuda, get_dev, get_device, get_scalar, norm,
                            reshape)
from ._lib import KgeError

EMB_INIT_EPS = 2.0


class KEModel(object):
    def __init__(self, args, model_name, n_entities, n_relations, hidden_dim, gamma,
                 double_entity_emb=False, double_relation_emb=False):
        self.args = args
        self.has_edge_importance = getattr(args, 'has_edge_importance', False)
        self.n_entities = n_entities
        self.n_relations = n_relations
        self.model_name = model_name
        self.hidden_dim = hidden_dim
        self.eps = EMB_INIT_EPS
        self.emb_init = (gamma + self.eps) / hidden_dim
        entity_dim = 2 * hidden_dim if double_entity_emb else hidden_dim
        relation_dim = 2 * hidden_dim if double_relation_emb else hidden_dim
        device = get_device(args)
        if device.type != 'cuda':
            raise KgeError("dglke_amd.KEModel needs args.gpu = [k >= 0]: the tables live in HBM "
                           "and every op is a HIP kernel (no CPU fallback).")
        self.device = device
        self.loss_gen = LossGenerator(args,
                                      getattr(args, 'loss_genre', None) or 'Logsigmoid',
                                      bool(getattr(args, 'neg_adversarial_sampling', False)),
                                      getattr(args, 'adversarial_temperature', None) or 1.0,
                                      bool(getattr(args, 'pairwise', False)))
        self.entity_emb = ExternalEmbedding(args, n_entities, entity_dim, device)
        if model_name == 'RESCAL':        # relation_emb = relation_dim * entity_dim (general_models.py:232-236)
            if relation_dim != entity_dim:
                raise KgeError("RESCAL needs relation_dim == entity_dim (the reference's edge_func multiplies "
                               "head [ent_dim] with M tail [rel_dim] element-wise, score_fun.py:387-394)")
            relation_dim = relation_dim * entity_dim
        self.rel_dim = relation_dim
        self.entity_dim = entity_dim
        self.strict_rel_part = bool(getattr(args, 'strict_rel_part', False))
        self.soft_rel_part = bool(getattr(args, 'soft_rel_part', False))
        # relation table: replicated in HBM (supersedes the per-process GPU copy +
        # CPU global table of general_models.py:242-246, 590-637)
        self.relation_emb = ExternalEmbedding(args, n_relations, relation_dim, device)
        self.global_relation_emb = self.relation_emb

        if model_name in ('TransE', 'TransE_l2'):
            self.score_func = TransEScore(gamma, 'l2')
        elif model_name == 'TransE_l1':
            self.score_func = TransEScore(gamma, 'l1')
        elif model_name == 'DistMult':
            self.score_func = DistMultScore()
        elif model_name == 'ComplEx':
            self.score_func = ComplExScore()
        elif model_name == 'RotatE':
            self.score_func = RotatEScore(gamma, self.emb_init)
        elif model_name == 'SimplE':
            self.score_func = SimplEScore()
        elif model_name == 'RESCAL':
            self.score_func = RESCALScore(relation_dim // entity_dim, entity_dim)
        elif model_name == 'TransR':
            projection_emb = ExternalEmbedding(args, n_relations, entity_dim * relation_dim, device)
            self.score_func = TransRScore(gamma, projection_emb, relation_dim, entity_dim)
        else:
            raise KgeError("unknown model %s" % model_name)
        self.head_neg_score = self.score_func.create_neg(True)
        self.tail_neg_score = self.score_func.create_neg(False)
        self.head_neg_prepare = self.score_func.create_neg_prepare(True)
        self.tail_neg_prepare = self.score_func.create_neg_prepare(False)
        self.reset_parameters()
        self.engine = StepEngine(
            model_name, n_entities, n_relations, hidden_dim, gamma, args.lr, device,
            double_entity_emb, double_relation_emb,
            bool(getattr(args, 'neg_adversarial_sampling', False)),
            getattr(args, 'adversarial_temperature', None) or 1.0,
            getattr(args, 'regularization_coef', 0.0) or 0.0,
            getattr(args, 'regularization_norm', 3) or 0,
            getattr(args, 'loss_genre', None) or 'Logsigmoid',
            bool(getattr(args, 'pairwise', False)), getattr(args, 'margin', 1.0),
            tables=(self.entity_emb.emb, self.entity_emb.state_sum, self.relation_emb.emb,
                    self.relation_emb.state_sum) + ((self.score_func.projection_emb.emb,
                                                     self.score_func.projection_emb.state_sum)
                                                    if model_name == 'TransR' else ()),
            flags=_lib.FLAG_NEG_DEG_SAMPLE if getattr(args, 'neg_deg_sample', False) and
            model_name not in ('TransR', 'RESCAL') else 0)

    # ---- bookkeeping (general_models.py:278-330) ----------------------------------------
    def share_memory(self):
        self.entity_emb.share_memory()
        self.relation_emb.share_memory()

    def save_emb(self, path, dataset):
        self.entity_emb.save(path, dataset + '_' + self.model_name + '_entity')
        self.relation_emb.save(path, dataset + '_' + self.model_name + '_relation')
        self.score_func.save(path, dataset + '_' + self.model_name)

    def load_emb(self, path, dataset):
        self.entity_emb.load(path, dataset + '_' + self.model_name + '_entity')
        self.relation_emb.load(path, dataset + '_' + self.model_name + '_relation')
        self.score_func.load(path, dataset + '_' + self.model_name)
        self._rebind()

    def _rebind(self):
        e = self.engine
        e.ent, e.ent_state = self.entity_emb.emb, self.entity_emb.state_sum
        e.rel, e.rel_state = self.relation_emb.emb, self.relation_emb.state_sum
        if self.model_name == 'TransR':
            e.proj, e.proj_state = self.score_func.projection_emb.emb, self.score_func.projection_emb.state_sum
        e._bind_tables()

    def reset_parameters(self):
        self.entity_emb.init(self.emb_init)
        self.score_func.reset_parameters()
        self.relation_emb.init(self.emb_init)

    # ---- scores (general_models.py:332-434) ---------------------------------------------
    def predict_score(self, g):
        self.score_func(g)
        return g.edata['score']

    def predict_neg_score(self, pos_g, neg_g, to_device=None, gpu_id=-1, trace=False,
                          neg_deg_sample=False):
        num_chunks = neg_g.num_chunks
        chunk_size = neg_g.chunk_size
        neg_sample_size = neg_g.neg_sample_size
        head_ids, tail_ids = pos_g.all_edges(order='eid')
        rel = pos_g.edata['emb']
        if neg_g.neg_head:
            neg_ids = neg_g.ndata['id'][neg_g.head_nid]
            neg = self.entity_emb(neg_ids, gpu_id, trace)
            pos_side = ops.gather_local(pos_g.ndata['emb'], tail_ids)
            other = head_ids
        else:
            neg_ids = neg_g.ndata['id'][neg_g.tail_nid]
            neg = self.entity_emb(neg_ids, gpu_id, trace)
            pos_side = ops.gather_local(pos_g.ndata['emb'], head_ids)
            other = tail_ids
        if neg_deg_sample:
            # in-batch positives of the corrupted side are used as extra negatives, the true
            # edge is masked out (general_models.py:396-402, 417-423, 429-432)
            extra = ops.gather_local(pos_g.ndata['emb'], other).reshape(num_chunks, chunk_size, -1)
            neg = th.cat([extra, neg.reshape(num_chunks, neg_sample_size, -1)], 1)
            neg_sample_size = chunk_size + neg_sample_size
            neg = neg.reshape(num_chunks * neg_sample_size, -1)
        if neg_g.neg_head:
            neg, pos_side = self.head_neg_prepare(pos_g.edata['id'], num_chunks, neg, pos_side,
                                                  gpu_id, trace)
            neg_score = self.head_neg_score(neg, rel, pos_side, num_chunks, chunk_size,
                                            neg_sample_size)
        else:
            pos_side, neg = self.tail_neg_prepare(pos_g.edata['id'], num_chunks, pos_side, neg,
                                                  gpu_id, trace)
            neg_score = self.tail_neg_score(pos_side, rel, neg, num_chunks, chunk_size,
                                            neg_sample_size)
        if neg_deg_sample:
            neg_g.neg_sample_size = neg_sample_size
            return ops.mask_diag(neg_score, num_chunks, chunk_size, neg_sample_size)     # mask[:, 0::(N'+1)] = 0
        return neg_score

    # ---- evaluation (general_models.py:436-485), rank computed on the GPU --------------
    def forward_test(self, pos_g, neg_g, logs, gpu_id=-1):
        with th.no_grad():
            pos_g.ndata['emb'] = self.entity_emb(pos_g.ndata['id'], gpu_id, False)
            pos_g.edata['emb'] = self.relation_emb(pos_g.edata['id'], gpu_id, False)
            self.score_func.prepare(pos_g, gpu_id, False)
            batch_size = pos_g.number_of_edges()
            pos_scores = reshape(self.predict_score(pos_g), batch_size, -1)
            neg_scores = self.predict_neg_score(
                pos_g, neg_g, to_device=cuda, gpu_id=gpu_id, trace=False,
                neg_deg_sample=getattr(self.args, 'neg_deg_sample_eval', False))
            neg_scores = reshape(neg_scores, batch_size, -1)
            bias = None
            if getattr(self.args, 'eval_filter', False):
                bias = reshape(neg_g.edata['bias'], batch_size, -1)
            rankings = ops.rank_from_scores(neg_scores, pos_scores, bias).tolist()
        for ranking in rankings:
            logs.append({'MRR': 1.0 / ranking, 'MR': float(ranking),
                         'HITS@1': 1.0 if ranking <= 1 else 0.0,
                         'HITS@3': 1.0 if ranking <= 3 else 0.0,
                         'HITS@10': 1.0 if ranking <= 10 else 0.0})

    # ---- training step, drop-in form (general_models.py:529-588) ------------------------
    def forward(self, pos_g, neg_g, gpu_id=-1):
        pos_g.ndata['emb'] = self.entity_emb(pos_g.ndata['id'], gpu_id, True)
        pos_g.edata['emb'] = self.relation_emb(pos_g.edata['id'], gpu_id, True)
        self.score_func.prepare(pos_g, gpu_id, True)
        pos_score = self.predict_score(pos_g)
        neg_score = self.predict_neg_score(pos_g, neg_g, to_device=cuda, gpu_id=gpu_id, trace=True,
                                           neg_deg_sample=getattr(self.args, 'neg_deg_sample', False))
        neg_score = reshape(neg_score, -1, neg_g.neg_sample_size)
        edge_weight = pos_g.edata['impts'] if self.has_edge_importance else None
        loss, log = self.loss_gen.get_total_loss(pos_score, neg_score, edge_weight)
        coef = getattr(self.args, 'regularization_coef', 0.0) or 0.0
        nm = getattr(self.args, 'regularization_norm', 0) or 0
        if coef > 0.0 and nm > 0:
            # x.norm(p) ** p of the concatenated traces = the sum over the traces (no th.cat copy)
            terms = [ops.pnorm_pow(data, nm) for _, data in self.entity_emb.trace + self.relation_emb.trace]
            reg = coef * sum(terms[1:], terms[0])
            log['regularization'] = get_scalar(reg)
            loss = loss + reg
        return loss, log

    def update(self, gpu_id=-1):
        self.entity_emb.update(gpu_id)
        self.relation_emb.update(gpu_id)
        self.score_func.update(gpu_id)

    # ---- training step, fused form ------------------------------------------------------
    def train_step(self, pos_g, neg_g=None, sync_log=False):
        """forward + backward + update in one kernel sequence.  Returns the log dict if
        sync_log (one 16-byte D2H copy) else None; running sums are kept in
        self.engine.loss_accum."""
        batch = pos_g.batch if hasattr(pos_g, 'batch') else pos_g
        self.engine.step(batch, per_step_loss=sync_log)
        if not sync_log:
            return None
        v = self.engine.read_loss()
        if self.engine.hp.pairwise:
            log = {'loss': v[2]}
        else:
            log = {'pos_loss': v[0], 'neg_loss': v[1], 'loss': v[2]}
        if self.engine.hp.reg_coef > 0 and self.engine.hp.reg_norm > 0:
            log['regularization'] = v[3]
        return log

    # ---- relation-partition API (general_models.py:590-637): with a replicated HBM relation
    # table these are bookkeeping no-ops kept for interface compatibility --------------------
    def prepare_relation(self, device=None):
        return None

    def prepare_cross_rels(self, cross_rels):
        self.relation_emb.setup_cross_rels(cross_rels, self.global_relation_emb)

    def writeback_relation(self, rank=0, rel_parts=None):
        return None

    def load_relation(self, device=None):
        return None

    def create_async_update(self):
        self.entity_emb.create_async_update()

    def finish_async_update(self):
        self.entity_emb.finish_async_update()

    def pull_model(self, client, pos_g, neg_g):
        raise KgeError("the DGL KVStore parameter server (general_models.py:650-680) is replaced "
                       "by range-sharded tables + RCCL all-to-all: see dglke_amd.dist")

    def push_gradient(self, client):
        raise KgeError("the DGL KVStore parameter server (general_models.py:650-680) is replaced "
                       "by range-sharded tables + RCCL all-to-all: see dglke_amd.dist")
