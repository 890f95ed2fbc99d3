"""HIP-backed score-function plugins with the reference's plugin surface
(models/pytorch/score_fun.py): `edge_func`, `infer`, `prepare`, `create_neg_prepare`, `forward`,
`create_neg`, `update`, `reset_parameters`, `save`, `load`.

Hand-written kernels, forward and analytic backward: TransE_l1 / TransE_l2 (score_fun.py:40), DistMult (:222),
ComplEx (:289), RotatE (:451), SimplE (:556), RESCAL (:378); TransR (:110) per-op = library GEMM projections +
the TransE_l1 kernels in relation space (its fast path is the fused step).
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
    """score_fun.py:110-220: owns the third table, projection_emb [n_rel, entity_dim * relation_dim].

    Per-op route (the reference's own decomposition): `prepare` projects head and tail of every positive edge with its
    relation's matrix (projection trace 0), the `create_neg_prepare` closures gather the matrices AGAIN (trace 1), project
    the uncorrupted side and EVERY negative of the chunk with EVERY positive's matrix - batched matrix products on the
    fp32-MFMA tile routine of kge_transr.hip (`kge_transr_project*`, analytic backward; the reference: th.matmul + autograd)
    - and `edge_func` / `create_neg` are L1 distances in
    relation space: the TransE_l1 HIP kernels on the projected rows (one 'chunk' per positive for the negatives, whose
    projections differ per positive).  The fast path for TransR is the fused step (`KEModel.train_step`, kge_transr.hip),
    which never materialises the [C, chunk, N, relation_dim] tensor this route passes around like the reference."""
    model_name = 'TransR'

    def __init__(self, gamma, projection_emb, relation_dim, entity_dim):
        self.gamma = gamma
        self.projection_emb = projection_emb
        self.relation_dim, self.entity_dim = relation_dim, entity_dim

    def edge_func(self, edges):
        # gamma - ||head_emb + emb - tail_emb||_1 (score_fun.py:121-126) = the TransE_l1 edge score on projected rows
        return {'score': ops.score_pos('TransE_l1', edges.data['head_emb'], edges.data['emb'], edges.data['tail_emb'],
                                       self.gamma, 1.0)}

    def infer(self, head_emb, rel_emb, tail_emb):
        pass                                              # like the reference (score_fun.py:128-129)

    def prepare(self, g, gpu_id, trace=False):
        head_ids, tail_ids = g.all_edges(order='eid')
        projection = self.projection_emb(g.edata['id'], gpu_id, trace)        # [B, entity_dim * relation_dim]
        emb = g.ndata['emb']
        De, Dr = self.entity_dim, self.relation_dim
        g.edata['head_emb'] = ops.transr_project(ops.gather_local(emb, head_ids), projection, De, Dr)
        g.edata['tail_emb'] = ops.transr_project(ops.gather_local(emb, tail_ids), projection, De, Dr)

    def create_neg_prepare(self, neg_head):
        def project(rel_id, num_chunks, pos, neg, gpu_id, trace):
            projection = self.projection_emb(rel_id, gpu_id, trace)           # gathered AGAIN: the second trace entry
            De, Dr = self.entity_dim, self.relation_dim
            pos = ops.transr_project(pos.reshape(-1, De), projection, De, Dr).reshape(num_chunks, -1, Dr)
            chunk = pos.shape[1]
            neg = neg.reshape(-1, De)
            # (num_chunks, num_rel, num_neg_nodes, rel_dim): every negative through every positive's matrix
            neg = ops.transr_project_neg(neg, projection, num_chunks, chunk, neg.shape[0] // num_chunks, De, Dr)
            return pos, neg
        if neg_head:
            def fn(rel_id, num_chunks, head, tail, gpu_id, trace=False):
                tail, head = project(rel_id, num_chunks, tail, head, gpu_id, trace)
                return head, tail
        else:
            def fn(rel_id, num_chunks, head, tail, gpu_id, trace=False):
                head, tail = project(rel_id, num_chunks, head, tail, gpu_id, trace)
                return head, tail
        return fn

    def create_neg(self, neg_head):
        gamma = self.gamma

        def fn(heads, relations, tails, num_chunks, chunk_size, neg_sample_size):
            # BOTH closures of the reference subtract the relation (score_fun.py:203-204, 212-213):
            #   head mode: gamma - ||heads' - (tails - r)||_1      tail mode: gamma - ||(heads - r) - tails'||_1
            # = TransE_l1 negative scores with B 'chunks' of one positive each (the projected negatives differ per
            #   positive); tail mode feeds -r so that the kernel's pos-side vector h + (-r) is h - r
            B = num_chunks * chunk_size
            if neg_head:
                s = ops.score_neg('TransE_l1', True, tails.reshape(B, -1), relations.reshape(B, -1),
                                  heads.reshape(B * neg_sample_size, -1), B, 1, neg_sample_size, gamma, 1.0, self.flags)
            else:
                s = ops.score_neg('TransE_l1', False, heads.reshape(B, -1), -relations.reshape(B, -1),
                                  tails.reshape(B * neg_sample_size, -1), B, 1, neg_sample_size, gamma, 1.0, self.flags)
            return s.reshape(num_chunks, chunk_size, neg_sample_size)
        return fn

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
