"""Batch container + "plan" for the fused HIP step.

A batch is the content of the reference's (pos_g, neg_g) pair (dataloader/sampler.py:421-457,
862-876; members read at models/general_models.py:376-427, 548-549) as flat id arrays.  The plan
groups duplicate rows so that the update kernel can give every table row to exactly one
wavefront (no atomics, reference trace order - see include/kge_hip.h `kge_batch`).

This is host-side integer bookkeeping (numpy); it is the part of the sampler that DGL's C++
EdgeSampler does when it relabels the nodes of the positive subgraph.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_I64_FIELDS = ("h_gid", "t_gid", "rel_ids", "neg_ids", "ue_id", "ur_id", "nid", "h_local", "t_local")
_I32_FIELDS = ("ue_pos_ptr", "ue_pos_adj", "ue_neg_ptr", "ue_neg_slot", "ur_ptr", "ur_edge", "ue_rec",
               "ur_rec")


def build_plan(h, t, r, neg, chunk, N, neg_head, edge_w=None):
    """h,t,r: int64 [B] global ids; neg: int64 [C*N] corrupt entity ids (chunk c of positives,
    rows c*chunk.., shares negatives c*N..).  Returns a dict of numpy arrays + scalars."""
    h = np.ascontiguousarray(h, dtype=np.int64)
    t = np.ascontiguousarray(t, dtype=np.int64)
    r = np.ascontiguousarray(r, dtype=np.int64)
    neg = np.ascontiguousarray(neg, dtype=np.int64)
    B = h.shape[0]
    if B % chunk != 0:
        raise ValueError("batch size %d is not a multiple of the chunk size %d" % (B, chunk))
    Cn = B // chunk
    if neg.shape[0] != Cn * N:
        raise ValueError("need C*N = %d negative ids, got %d" % (Cn * N, neg.shape[0]))
    ends = np.concatenate([h, t])
    nid, inv = np.unique(ends, return_inverse=True)         # pos_g.ndata['id'], local ids
    ue_id = np.unique(np.concatenate([nid, neg]))
    UE = ue_id.shape[0]
    # positive adjacency: code = edge*2 + side, grouped by union entry, ascending code inside
    upos = np.searchsorted(ue_id, ends)
    codes = np.concatenate([np.arange(B, dtype=np.int64) * 2, np.arange(B, dtype=np.int64) * 2 + 1])
    order = np.lexsort((codes, upos))
    ue_pos_adj = codes[order].astype(np.int32)
    ue_pos_ptr = np.zeros(UE + 1, np.int32)
    np.cumsum(np.bincount(upos, minlength=UE), out=ue_pos_ptr[1:])
    # negative slots grouped by union entry, ascending slot inside (index_add_ order)
    uneg = np.searchsorted(ue_id, neg)
    ue_neg_slot = np.argsort(uneg, kind="stable").astype(np.int32)
    ue_neg_ptr = np.zeros(UE + 1, np.int32)
    np.cumsum(np.bincount(uneg, minlength=UE), out=ue_neg_ptr[1:])
    # relations
    ur_id, rinv = np.unique(r, return_inverse=True)
    UR = ur_id.shape[0]
    ur_edge = np.argsort(rinv, kind="stable").astype(np.int32)
    ur_ptr = np.zeros(UR + 1, np.int32)
    np.cumsum(np.bincount(rinv, minlength=UR), out=ur_ptr[1:])
    # packed 32-byte records (include/kge_hip.h): id, list bounds and the first list entries
    def _lohi(ids):
        ids = ids.astype(np.int64)
        return (ids & 0xFFFFFFFF).astype(np.uint32).view(np.int32), (ids >> 32).astype(np.int32)
    ue_rec = np.zeros((UE, 8), np.int32)
    ue_rec[:, 0], ue_rec[:, 1] = _lohi(ue_id)
    ue_rec[:, 2], ue_rec[:, 3] = ue_pos_ptr[:-1], ue_pos_ptr[1:]
    ue_rec[:, 4], ue_rec[:, 5] = ue_neg_ptr[:-1], ue_neg_ptr[1:]
    hp = ue_pos_ptr[1:] > ue_pos_ptr[:-1]
    hn = ue_neg_ptr[1:] > ue_neg_ptr[:-1]
    ue_rec[:, 6] = np.where(hp, ue_pos_adj[np.minimum(ue_pos_ptr[:-1], 2 * B - 1)], -1)
    ue_rec[:, 7] = np.where(hn, ue_neg_slot[np.minimum(ue_neg_ptr[:-1], Cn * N - 1)], -1)
    ur_rec = np.zeros((UR, 8), np.int32)
    ur_rec[:, 0], ur_rec[:, 1] = _lohi(ur_id)
    ur_rec[:, 2], ur_rec[:, 3] = ur_ptr[:-1], ur_ptr[1:]
    ur_rec[:, 4] = ur_edge[ur_ptr[:-1]]
    out = dict(B=B, C=Cn, chunk=int(chunk), N=int(N), neg_head=int(bool(neg_head)),
               U=int(nid.shape[0]), UE=int(UE), UR=int(UR),
               h_gid=h, t_gid=t, rel_ids=r, neg_ids=neg, ue_id=ue_id.astype(np.int64),
               ur_id=ur_id.astype(np.int64), nid=nid.astype(np.int64),
               h_local=inv[:B].astype(np.int64), t_local=inv[B:].astype(np.int64),
               ue_pos_ptr=ue_pos_ptr, ue_pos_adj=ue_pos_adj, ue_neg_ptr=ue_neg_ptr,
               ue_neg_slot=ue_neg_slot, ur_ptr=ur_ptr, ur_edge=ur_edge,
               ue_rec=ue_rec.reshape(-1), ur_rec=ur_rec.reshape(-1),
               edge_w=None if edge_w is None else np.ascontiguousarray(edge_w, np.float32))
    return out


def _pack(plans):
    """Pack the arrays of several plans into one byte buffer; returns (bytes, per-plan offsets)."""
    chunks, offs, pos = [], [], 0

    def put(a):
        nonlocal pos
        pad = (-pos) % 32
        if pad:
            chunks.append(np.zeros(pad, np.uint8))
            pos += pad
        o = pos
        b = a.view(np.uint8).reshape(-1)
        chunks.append(b)
        pos += b.shape[0]
        return o

    for p in plans:
        o = {}
        for k in _I64_FIELDS + _I32_FIELDS:
            o[k] = put(p[k])
        if p["edge_w"] is not None:
            o["edge_w"] = put(p["edge_w"])
        offs.append(o)
    return np.concatenate(chunks) if chunks else np.zeros(0, np.uint8), offs


class Batch(object):
    """One device-resident batch: flat ids + plan, and the ctypes `kge_batch` that points at them."""

    def __init__(self, plan, dev_buf, offs):
        self.p = plan
        self.buf = dev_buf          # keeps the device memory alive
        base = dev_buf.data_ptr()
        self.B, self.C, self.chunk, self.N = plan["B"], plan["C"], plan["chunk"], plan["N"]
        self.neg_head = bool(plan["neg_head"])
        self.U, self.UE, self.UR = plan["U"], plan["UE"], plan["UR"]
        self._offs = offs
        kb = _lib.KgeBatch()
        for k in ("B", "C", "chunk", "N", "neg_head", "U", "UE", "UR"):
            setattr(kb, k, plan[k])
        for k in ("h_gid", "t_gid", "rel_ids", "neg_ids", "ue_id", "ue_pos_ptr", "ue_pos_adj",
                  "ue_neg_ptr", "ue_neg_slot", "ur_id", "ur_ptr", "ur_edge", "ue_rec", "ur_rec"):
            setattr(kb, k, base + offs[k])
        kb.edge_w = (base + offs["edge_w"]) if "edge_w" in offs else None
        if "edge_w" in offs:
            # the batch's MEAN importance, once per batch (ABI 8 kge_batch.edge_w_mean: the weight of every positive edge in the
            # reference's loss, loss.py:75,82); accumulated in fp64, rounded to fp32 like the reference's mean over a float32 tensor
            kb.edge_w_mean = float(np.float32(np.asarray(plan["edge_w"], np.float64).mean()))
        self.c = kb

    def view(self, name):
        """torch view of one packed array (int64 / int32 / float32)."""
        if name == "edge_w":
            n, dt, isz = self.p["edge_w"].shape[0], torch.float32, 4
        elif name in _I64_FIELDS:
            n, dt, isz = self.p[name].shape[0], torch.int64, 8
        else:
            n, dt, isz = self.p[name].shape[0], torch.int32, 4
        o = self._offs[name]
        return self.buf[o:o + n * isz].view(dt)


def upload(plans, device):
    """One H2D copy for a list of plans -> list of Batch."""
    host, offs = _pack(plans)
    dev = torch.from_numpy(host).to(device)
    return [Batch(p, dev, o) for p, o in zip(plans, offs)]


def make_batch(h, t, r, neg, chunk, N, neg_head, device, edge_w=None):
    return upload([build_plan(h, t, r, neg, chunk, N, neg_head, edge_w)], device)[0]
