"""HIP-backed `ExternalEmbedding` - same surface as the reference class
(models/pytorch/tensor_models.py:210-407): `emb`, `state_sum`, `trace`, `init`, `__call__`,
`update`, `curr_emb`, `save`, `load`, `share_memory`, async-update and cross-relation methods.

The table and its Adagrad state live in HBM.  `__call__` is the coalesced row gather kernel,
`update` the lock-free scatter Adagrad kernel (one call per trace, reference order).
"""
import os

import numpy as np
import torch as th

from . import ops
from ._lib import KgeError

# helper lambdas with the reference's names (tensor_models.py:53-57)
none = lambda x: x
norm = lambda x, p: x.norm(p=p) ** p
get_scalar = lambda x: x.detach().item()
reshape = lambda arr, x, y: arr.view(x, y)
cuda = lambda arr, gpu: arr.cuda(gpu)


def get_dev(gpu):
    return th.device('cpu') if gpu < 0 else th.device('cuda:' + str(gpu))


def get_device(args):
    return th.device('cpu') if args.gpu[0] < 0 else th.device('cuda:' + str(args.gpu[0]))


class ExternalEmbedding(object):
    """Sparse embedding table resident in HBM (tensor_models.py:210)."""

    def __init__(self, args, num, dim, device):
        device = th.device(device)
        if device.type != 'cuda':
            raise KgeError("dglke_amd.ExternalEmbedding lives in GPU HBM; device %s is not "
                           "supported (no CPU fallback; --mix_cpu_gpu is superseded by the "
                           "HBM-resident / range-sharded tables)." % device)
        self.gpu = args.gpu
        self.args = args
        self.num = num
        self.dim = dim
        self.trace = []
        self.emb = th.empty(num, dim, dtype=th.float32, device=device)
        self.state_sum = th.zeros(num, dtype=th.float32, device=device)
        self.state_step = 0
        self.has_cross_rel = False
        self.async_q = None
        self.async_p = None

    def init(self, emb_init):
        """uniform(-emb_init, emb_init), zero state (tensor_models.py:240-249)."""
        th.nn.init.uniform_(self.emb, -emb_init, emb_init)
        self.state_sum.zero_()

    def setup_cross_rels(self, cross_rels, global_emb):
        """tensor_models.py:251-257.  With an HBM-resident (replicated) relation table there is no
        CPU-side global copy to refresh from; the bitmap is kept so get_noncross_idx works."""
        bitmap = th.zeros((self.num,), dtype=th.bool)
        for rel in cross_rels:
            bitmap[rel] = 1
        self.cpu_bitmap = bitmap
        self.has_cross_rel = False
        self.global_emb = global_emb

    def get_noncross_idx(self, idx):
        mask = ~self.cpu_bitmap[idx.cpu()]
        return idx[mask.to(idx.device)]

    def share_memory(self):
        """tensor_models.py:264-268: the reference shares a CPU table between trainer processes.
        Here each process owns its HBM shard, so there is nothing to share."""
        return None

    def __call__(self, idx, gpu_id=-1, trace=True):
        """row gather (tensor_models.py:270-302)."""
        idx = idx.to(self.emb.device)
        s = ops.gather_rows(self.emb, idx)
        if trace:
            data = s.requires_grad_(True)          # s is already a fresh copy
            self.trace.append((idx, data))
        else:
            data = s
        return data

    def update(self, gpu_id=-1):
        """row-sparse Adagrad per trace, in trace order (tensor_models.py:304-362)."""
        self.state_step += 1
        with th.no_grad():
            for idx, data in self.trace:
                if data.grad is None:
                    continue
                ops.adagrad_scatter(self.emb, self.state_sum, idx, data.grad, self.args.lr, 1e-10)
        self.trace = []

    def create_async_update(self):
        """tensor_models.py:364-369: the reference overlaps a CPU update process with GPU compute.
        Kernels on a HIP stream are already asynchronous w.r.t. the host, so this is a no-op."""
        self.async_q = None

    def finish_async_update(self):
        if th.cuda.is_available():
            th.cuda.current_stream().synchronize()

    def curr_emb(self):
        return th.cat([data for _, data in self.trace], 0)

    def save(self, path, name):
        np.save(os.path.join(path, name + '.npy'), self.emb.detach().cpu().numpy())

    def load(self, path, name):
        arr = np.load(os.path.join(path, name + '.npy'))
        self.emb = th.tensor(arr, dtype=th.float32, device=self.emb.device)
        self.num, self.dim = self.emb.shape
