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
        """tensor_models.py:264-268: the reference moves the CPU table into shared memory so that the trainer
        processes see one table.  The HBM-resident table is already visible to every trainer of this process
        (streams / lanes) and, across processes, through the hipIpc shard map (dglke_amd.p2p): nothing to move.
        Returns self (the reference's tensors return themselves from share_memory_())."""
        return self

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

    def _apply(self, traces):
        with th.no_grad():
            for idx, grad in traces:
                ops.adagrad_scatter(self.emb, self.state_sum, idx, grad, self.args.lr, 1e-10)

    def update(self, gpu_id=-1):
        """row-sparse Adagrad per trace, in trace order (tensor_models.py:304-362).  After create_async_update():
        the traces of THIS step are parked and the traces of the PREVIOUS step land now - the next forward gathers
        rows that contain every update but this step's, the deterministic form of the reference's helper-process
        update (tensor_models.py:136-175, :325-328): one step of staleness, no race."""
        self.state_step += 1
        traces = [(idx, data.grad.detach()) for idx, data in self.trace if data.grad is not None]
        self.trace = []
        if self.async_q is None:
            self._apply(traces)
        else:
            pending, self.async_q = self.async_q, traces
            self._apply(pending)

    def create_async_update(self):
        """tensor_models.py:364-369 (starts the helper process there): from now on `update` defers by one step."""
        self.async_q = []

    def finish_async_update(self):
        """tensor_models.py:371-375: land what is still parked and return to synchronous updates."""
        if self.async_q is not None:
            self._apply(self.async_q)
        self.async_q = None

    def curr_emb(self):
        return th.cat([data for _, data in self.trace], 0)

    def save(self, path, name):
        np.save(os.path.join(path, name + '.npy'), self.emb.detach().cpu().numpy())

    def load(self, path, name):
        """tensor_models.py:399-407.  The table keeps its storage when the shape matches (engines and graphs hold its
        address); a file of another shape replaces table AND state (a fresh, zero Adagrad state of the new row count)."""
        arr = np.load(os.path.join(path, name + '.npy'))
        if tuple(arr.shape) == tuple(self.emb.shape):
            self.emb.copy_(th.as_tensor(arr, dtype=th.float32))
        else:
            self.emb = th.tensor(arr, dtype=th.float32, device=self.emb.device)
            self.num, self.dim = self.emb.shape
            self.state_sum = th.zeros(self.num, dtype=th.float32, device=self.emb.device)
