"""HIP-backed score-function plugins with the reference's plugin surface
(models/pytorch/score_fun.py): `edge_func`, `infer`, `prepare`, `create_neg_prepare`, `forward`,
`create_neg`, `update`, `reset_parameters`, `save`, `load`.

In scope (hand-written kernels, forward and analytic backward): TransE_l1 / TransE_l2
(score_fun.py:40), DistMult (:222), ComplEx (:289), RotatE (:451), SimplE (:556).
"""
import torch as th

from . import ops


class _HipScore(object):
    """shared plumbing: every method routes to libkge_hip through ops.score_pos / ops.score_neg."""
    model_name = None
    gamma = 0.0
    emb_init = 1.0
    flags = 0

    def __call__(self, g):
        return self.forward(g)

    def edge_func(self, edges):
        return {'score': ops.score_pos(self.model_name, edges.src['emb'], edges.data['emb'],
                                       edges.dst['emb'], self.gamma, self.emb_init)}

    def forward(self, g):
        g.apply_edges(lambda edges: self.edge_func(edges))

    def infer(self, head_emb, rel_emb, tail_emb):
        """[H,R,T] scores of every (head, rel, tail) combination (e.g. score_fun.py:61-66): one
        chunk of H*R positives against T 'negatives' in tail-corruption mode."""
        H, R, T = head_emb.shape[0], rel_emb.shape[0], tail_emb.shape[0]
        heads = head_emb.unsqueeze(1).expand(H, R, head_emb.shape[1]).reshape(H * R, -1).contiguous()
        rels = rel_emb.unsqueeze(0).expand(H, R, rel_emb.shape[1]).reshape(H * R, -1).contiguous()
        s = ops.score_neg(self.model_name, False, heads, rels, tail_emb.contiguous(), 1, H * R, T,
                          self.gamma, self.emb_init, self.flags)
        return s.reshape(H, R, T)

    def prepare(self, g, gpu_id, trace=False):
        pass

    def create_neg_prepare(self, neg_head):
        def fn(rel_id, num_chunks, head, tail, gpu_id, trace=False):
            return head, tail
        return fn

    def update(self, gpu_id=-1):
        pass

    def reset_parameters(self):
        pass

    def save(self, path, name):
        pass

    def load(self, path, name):
        pass

    def create_neg(self, neg_head):
        """closure with the reference signature fn(heads, relations, tails, num_chunks,
        chunk_size, neg_sample_size) -> [C, chunk, N] (score_fun.py:91-108)."""
        def fn(heads, relations, tails, num_chunks, chunk_size, neg_sample_size):
            if neg_head:      # heads are the corrupt entities, tails the positive side
                return ops.score_neg(self.model_name, True, tails, relations, heads, num_chunks,
                                     chunk_size, neg_sample_size, self.gamma, self.emb_init,
                                     self.flags)
            return ops.score_neg(self.model_name, False, heads, relations, tails, num_chunks,
                                 chunk_size, neg_sample_size, self.gamma, self.emb_init, self.flags)
        return fn


class TransEScore(_HipScore):
    """score_fun.py:40-108"""
    def __init__(self, gamma, dist_func='l2'):
        self.gamma = gamma
        if dist_func == 'l1':
            self.dist_ord, self.model_name = 1, 'TransE_l1'
        else:
            self.dist_ord, self.model_name = 2, 'TransE_l2'


class DistMultScore(_HipScore):
    """score_fun.py:222-286"""
    model_name = 'DistMult'


class ComplExScore(_HipScore):
    """score_fun.py:289-376"""
    model_name = 'ComplEx'


class SimplEScore(_HipScore):
    """score_fun.py:556-641: rows are [x_i | x_j] halves, relation rows [rel | rel_inv];
    score = clamp(1/2 (<h_i, rel, t_j> + <t_i, rel_inv, h_j>), -20, 20)."""
    model_name = 'SimplE'

    def infer(self, head_emb, rel_emb, tail_emb):
        """the reference's `infer` does NOT clamp (score_fun.py:570-577): run the un-clamped dot-product
        kernel on the SimplE pos-side vectors 1/2 [rel_inv * h_j | h_i * rel]."""
        H, R, T = head_emb.shape[0], rel_emb.shape[0], tail_emb.shape[0]
        hd = head_emb.shape[1] // 2
        hi, hj = head_emb[:, :hd].unsqueeze(1), head_emb[:, hd:].unsqueeze(1)
        rel, rinv = rel_emb[:, :hd].unsqueeze(0), rel_emb[:, hd:].unsqueeze(0)
        a = (0.5 * th.cat([rinv * hj, hi * rel], -1)).reshape(H * R, -1).contiguous()
        s = ops.score_neg('DistMult', False, a, th.ones_like(a), tail_emb.contiguous(), 1, H * R, T, 0.0, 1.0,
                          self.flags)
        return s.reshape(H, R, T)


class TransRScore(_HipScore):
    """score_fun.py:110-220: owns the third table, projection_emb [n_rel, entity_dim * relation_dim].  The HIP
    build runs TransR through the fused step (`KEModel.train_step` / `dglke_train`) and `kge_rank_eval_ex`; the
    per-op autograd route of the other score functions (edge_func / create_neg closures) is not provided."""
    model_name = 'TransR'

    def __init__(self, gamma, projection_emb, relation_dim, entity_dim):
        self.gamma = gamma
        self.projection_emb = projection_emb
        self.relation_dim, self.entity_dim = relation_dim, entity_dim

    def _no_modular(self, *a, **k):
        from ._lib import KgeError
        raise KgeError("TransR runs through the fused step (KEModel.train_step / dglke_train) and kge_rank_eval_ex; "
                       "the per-op drop-in route is not built for it")

    edge_func = infer = prepare = _no_modular

    def create_neg(self, neg_head):
        return self._no_modular

    def create_neg_prepare(self, neg_head):
        return self._no_modular

    def reset_parameters(self):
        self.projection_emb.init(1.0)                     # score_fun.py:170-171

    def update(self, gpu_id=-1):
        self.projection_emb.update(gpu_id)

    def save(self, path, name):
        self.projection_emb.save(path, name + 'projection')

    def load(self, path, name):
        self.projection_emb.load(path, name + 'projection')

    def share_memory(self):
        self.projection_emb.share_memory()


class RESCALScore(_HipScore):
    """score_fun.py:378-449: relation rows are [relation_dim x entity_dim] matrices M; p = h . (M t); both
    corruption modes score (M x) . neg with x the uncorrupted entity."""
    model_name = 'RESCAL'

    def __init__(self, relation_dim, entity_dim):
        self.relation_dim, self.entity_dim = relation_dim, entity_dim

    def infer(self, head_emb, rel_emb, tail_emb):
        """score_fun.py:396-401: [H, R, T] = h . (M t): the head-corruption kernel with every (r, t) pair as a
        'positive' and the heads as candidates, transposed."""
        H, R, T = head_emb.shape[0], rel_emb.shape[0], tail_emb.shape[0]
        tails = tail_emb.unsqueeze(0).expand(R, T, tail_emb.shape[1]).reshape(R * T, -1).contiguous()
        rels = rel_emb.unsqueeze(1).expand(R, T, rel_emb.shape[1]).reshape(R * T, -1).contiguous()
        s = ops.score_neg(self.model_name, True, tails, rels, head_emb.contiguous(), 1, R * T, H, 0.0, 1.0,
                          self.flags)
        return s.reshape(R, T, H).permute(2, 0, 1).contiguous()


class RotatEScore(_HipScore):
    """score_fun.py:451-554"""
    model_name = 'RotatE'

    def __init__(self, gamma, emb_init):
        self.gamma = gamma
        self.emb_init = emb_init
