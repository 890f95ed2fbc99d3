"""ctypes binding of libkge_hip.so (C ABI: include/kge_hip.h).

The library is the product: every op in this package runs through it.  There is NO CPU / eager
fallback - if the shared object is missing (not built) importing an op raises, and calling an op
with a non-CUDA tensor raises.
"""
import contextlib
import ctypes as C
import gc
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KGE_LIB") or os.path.join(_HERE, "libkge_hip.so")   # KGE_LIB: A/B builds

KGE_ABI_VERSION = 8
MODEL_IDS = {"TransE_l1": 0, "TransE_l2": 1, "TransE": 1, "DistMult": 2, "ComplEx": 3, "RotatE": 4, "SimplE": 5, "RESCAL": 6, "TransR": 7}
LOSS_IDS = {"Logsigmoid": 0, "Logistic": 1, "Hinge": 2, "BCE": 3}
FLAG_FORCE_PAIRWISE = 1
FLAG_NO_TRANSE_FAST = 2
FLAG_DENSE_NEG = 4
FLAG_FUSED_LOSS = 8
FLAG_TWO_PASS_PAIR = 16
FLAG_NEG_DEG_SAMPLE = 32
FLAG_ASYNC_REL = 64
FLAG_DENSE_BWD = 256        # merged first launch: backward GEMM reads a dense copy of the negative rows (A-B aid)
FLAG_FWD_DIRECT = 512       # merged first launch: direct fragment loads instead of the LDS pos-side tile (A-B aid)
FLAG_LOSS_IN_FWD = 1024     # strict step: LossGenerator inside the first launch (3 launches; opt-in: measured slower, A-B aid)
TICKET_INTS = 4096          # kge_step_out.tickets: int32 words, zero when first handed over
FLAG_SPLIT_FWD = 128        # strict step: edge-forward and forward GEMM as two launches (validation / A-B aid)
PHASE_GATHER, PHASE_FORWARD, PHASE_BACKWARD, PHASE_UPDATE = 1, 2, 4, 8
ACC_SLOTS = 4096

c_f = C.c_float
c_i = C.c_int
c_i32 = C.c_int32
c_i64 = C.c_int64
c_p = C.c_void_p
c_sz = C.c_size_t
c_u = C.c_uint


class KgeBatch(C.Structure):
    _fields_ = [("B", c_i32), ("C", c_i32), ("chunk", c_i32), ("N", c_i32), ("neg_head", c_i32),
                ("U", c_i32), ("UE", c_i32), ("UR", c_i32),
                ("h_gid", c_p), ("t_gid", c_p), ("rel_ids", c_p), ("neg_ids", c_p), ("edge_w", c_p),
                ("ue_id", c_p), ("ue_pos_ptr", c_p), ("ue_pos_adj", c_p), ("ue_neg_ptr", c_p),
                ("ue_neg_slot", c_p), ("ur_id", c_p), ("ur_ptr", c_p), ("ur_edge", c_p),
                ("ue_rec", c_p), ("ur_rec", c_p), ("counts_dev", c_p),
                ("edge_w_mean", c_f), ("reserved_", c_i32)]        # ABI 8


class KgeHParams(C.Structure):
    _fields_ = [("model", c_i32), ("d_e", c_i32), ("d_r", c_i32), ("loss_genre", c_i32),
                ("adv", c_i32), ("pairwise", c_i32), ("reg_norm", c_i32), ("flags", C.c_uint32),
                ("gamma", c_f), ("emb_init", c_f), ("lr", c_f), ("adv_temp", c_f), ("margin", c_f),
                ("reg_coef", c_f), ("eps", c_f)]


class KgeTables(C.Structure):
    _fields_ = [("ent", c_p), ("ent_state", c_p), ("rel", c_p), ("rel_state", c_p),
                ("n_ent", c_i64), ("n_rel", c_i64), ("proj", c_p), ("proj_state", c_p)]


class KgeStepOut(C.Structure):
    _fields_ = [("loss4", c_p), ("loss_accum", c_p), ("pos_score", c_p), ("neg_score", c_p),
                ("g_pos_ent", c_p), ("g_neg", c_p), ("g_rel", c_p), ("tickets", c_p)]


class KgeSamplerJob(C.Structure):
    """kge_sampler_job (ABI 7): one batch of the NEXT group, built by tail workgroups of a training step's launches"""
    _fields_ = [("heads", c_p), ("rels", c_p), ("tails", c_p), ("perm", c_p), ("n_train", c_i64), ("n_ent", c_i64),
                ("B", c_i32), ("C", c_i32), ("chunk", c_i32), ("N", c_i32), ("seed", C.c_uint64), ("state", c_p), ("slot", c_p),
                ("scratch", c_p), ("scratch_bytes", c_sz), ("k", c_i32), ("advance", c_i32), ("pre_permuted", c_i32),
                ("reserved", c_i32), ("prev_slot", c_p)]


class KgeEmit(C.Structure):
    _fields_ = [("g0", c_p), ("gs0", c_p), ("g1", c_p), ("gs1", c_p), ("gr", c_p), ("gsr", c_p),
                ("ld_e", c_i32), ("ld_r", c_i32), ("rid", c_p), ("ent_by_id", c_i32), ("reserved", c_i32),
                ("msg_rows", c_p), ("msg_cap", c_i32), ("msg_cap_extra", c_i32)]      # ABI 8: packed single-trace entity messages


class KgeMergeJob(C.Structure):
    _fields_ = [("table", c_p), ("state_sum", c_p), ("n_rows", c_i64), ("dim", c_i32), ("nsrc", c_i32), ("cap", c_i32),
                ("ld", c_i32), ("ntraces", c_i32), ("cap_extra", c_i32), ("id_words", c_p), ("id_stride_words", c_i64),
                ("id_offset", c_i64), ("msg", c_p)]


class KgeShards(C.Structure):
    _fields_ = [("n_shards", c_i32), ("reserved", c_i32), ("ent_rows_per_shard", c_i64),
                ("rel_rows_per_shard", c_i64), ("ent_rows", c_p), ("ent_state", c_p),
                ("rel_rows", c_p), ("rel_state", c_p), ("n_ent", c_i64), ("n_rel", c_i64),
                ("rel_local", c_p), ("rel_state_local", c_p), ("proj_local", c_p), ("proj_state_local", c_p)]   # ABI 8


IPC_HANDLE_BYTES = 64

_SIGNATURES = {
    "kge_abi_version": (c_i, []),
    "kge_last_error": (C.c_char_p, []),
    "kge_gather_rows": (c_i, [c_p, c_i64, c_i, c_p, c_i64, c_p, c_p]),
    "kge_score_pos": (c_i, [c_i, c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_f, c_p, c_p]),
    "kge_score_pos_bwd": (c_i, [c_i, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_f, c_p, c_p, c_p,
                                c_p]),
    "kge_score_neg_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "kge_score_neg_fwd": (c_i, [c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_p,
                                c_p, c_sz, c_u, c_p]),
    "kge_score_neg_bwd": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f,
                                c_f, c_p, c_p, c_p, c_p, c_sz, c_u, c_p]),
    "kge_loss_fwd_bwd": (c_i, [c_i, c_i, c_f, c_i, c_f, c_p, c_p, c_p, c_i64, c_i, c_p, c_p, c_p,
                               c_p, c_sz, c_p]),
    "kge_reduce_loss": (c_i, [c_p, c_p, c_i, c_p]),
    "kge_adagrad_scatter": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_i64, c_f, c_f, c_p]),
    "kge_scatter_add_rows": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_i64, c_p]),
    "kge_pnorm_pow": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p, c_sz, c_p]),
    "kge_pnorm_pow_bwd": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p, c_p]),
    "kge_mask_diag": (c_i, [c_p, c_i, c_i, c_i, c_p]),
    "kge_rank_from_scores": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_p, c_p]),
    "kge_sampler_slot_bytes": (c_sz, [c_i, c_i, c_i]),
    "kge_sample_batches": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, C.c_uint64, c_p, c_p, c_sz,
                                 c_i, c_p]),
    "kge_batch_from_slot": (c_i, [c_p, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, C.POINTER(KgeBatch)]),
    "kge_adagrad_apply_packed": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_i, c_i64, c_i, c_f, c_f, c_p]),
    "kge_transr_project": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p, c_p]),
    "kge_transr_project_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_p, c_i, c_p]),
    "kge_transr_project_neg": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "kge_transr_project_neg_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p]),
    "kge_route_fill": (c_i, [c_p, c_i, c_sz, c_i, c_i64, c_p, c_p]),
    "kge_route_build": (c_i, [c_p, c_i, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    "kge_route_build_group": (c_i, [c_p, c_i, c_sz, c_i, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p, c_i, c_p, c_p]),
    "kge_batch_localized": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "kge_gather_rows_req": (c_i, [c_p, c_i64, c_i, c_p, c_i64, c_i64, c_p, c_p]),
    "kge_adagrad_apply_merged": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_f, c_f, c_p]),
    "kge_adagrad_apply_merged_pair": (c_i, [C.POINTER(KgeMergeJob), C.POINTER(KgeMergeJob), c_f, c_f, c_p]),
    "kge_adagrad_apply_rows": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_p, c_i64, c_f, c_f, c_p]),
    "kge_step_workspace_bytes": (c_sz, [C.POINTER(KgeHParams), c_i, c_i, c_i, c_i, c_i, c_i]),
    "kge_step_fused": (c_i, [C.POINTER(KgeHParams), C.POINTER(KgeTables), C.POINTER(KgeBatch),
                             C.POINTER(KgeStepOut), c_p, c_sz, c_p]),
    "kge_sampler_tail_scratch_bytes": (c_sz, [c_i, c_i, c_i, c_i64]),
    "kge_step_fused_sampling": (c_i, [C.POINTER(KgeHParams), C.POINTER(KgeTables), C.POINTER(KgeBatch),
                                      C.POINTER(KgeStepOut), c_p, c_sz, C.POINTER(KgeSamplerJob), c_p]),
    "kge_pipe_create": (c_i, [C.POINTER(c_p)]),
    "kge_pipe_destroy": (c_i, [c_p]),
    "kge_step_async_workspace_bytes": (c_sz, [C.POINTER(KgeHParams), c_i, c_i, c_i, c_i, c_i, c_i]),
    "kge_step_async": (c_i, [c_p, C.POINTER(KgeHParams), C.POINTER(KgeTables), C.POINTER(KgeBatch), C.POINTER(KgeBatch),
                             C.POINTER(KgeStepOut), c_p, c_sz, c_p]),
    "kge_step_async_flush": (c_i, [c_p, c_p]),
    "kge_step_phase": (c_i, [C.POINTER(KgeHParams), C.POINTER(KgeTables), C.POINTER(KgeBatch), C.POINTER(KgeStepOut), c_p,
                             c_sz, c_i, c_p]),
    "kge_step_grads": (c_i, [C.POINTER(KgeHParams), C.POINTER(KgeTables), C.POINTER(KgeBatch),
                             C.POINTER(KgeStepOut), C.POINTER(KgeEmit), c_p, c_sz, c_p]),
    "kge_step_sharded": (c_i, [C.POINTER(KgeHParams), C.POINTER(KgeShards), C.POINTER(KgeBatch),
                               C.POINTER(KgeStepOut), c_p, c_sz, c_p]),
    "kge_gather_rows_sharded": (c_i, [c_p, c_i, c_i64, c_i, c_p, c_i64, c_p, c_p]),
    "kge_rank_workspace_bytes": (c_sz, [c_i, c_i64, c_i]),
    "kge_rank_eval": (c_i, [c_i, c_i, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_f, c_p, c_i64,
                            c_p, c_p, c_i, c_p, c_p, c_p, c_sz, c_u, c_p]),
    "kge_rank_eval_ex": (c_i, [c_i, c_i, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_f, c_p, c_i64,
                               c_p, c_p, c_i, c_p, c_p, c_p, c_sz, c_u, c_p]),
    "kge_ipc_export": (c_i, [c_p, c_p, C.POINTER(c_i64)]),
    "kge_ipc_open": (c_i, [c_p, C.POINTER(c_p)]),
    "kge_ipc_close": (c_i, [c_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())

_lib = None


class KgeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KgeError(
                "libkge_hip.so is missing (%s). Build it with `python __graft_entry__.py` or "
                "`make -C dgl-ke_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.kge_abi_version() != KGE_ABI_VERSION:
            raise KgeError("libkge_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise KgeError("libkge_hip: %s (status %d)" % (lib().kge_last_error().decode(), rc))


def ptr(t):
    """device pointer of a CUDA tensor (None -> NULL); refuses host tensors: no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise KgeError("libkge_hip ops need CUDA (HIP) tensors; got a %s tensor. There is no CPU "
                       "fallback in the product path." % t.device)
    if not t.is_contiguous():
        raise KgeError("libkge_hip ops need contiguous tensors")
    return t.data_ptr()


_STREAM_TLS = threading.local()       # per-thread: trainer lanes / evaluation threads keep their own streams


def stream_ptr():
    """the stream the library calls are enqueued on: torch's current stream, or the one set by use_stream() IN THIS THREAD"""
    ov = getattr(_STREAM_TLS, "ptr", None)
    if ov is not None:
        return ov
    return torch.cuda.current_stream().cuda_stream


def use_stream(ptr):
    """make every following library call OF THE CALLING THREAD go to the HIP stream `ptr` (None: back to torch's current
    stream).  A cheap stand-in for `with torch.cuda.stream(...)` around ctypes calls (dist.DistEngine's pull pipeline);
    returns the previous value."""
    prev = getattr(_STREAM_TLS, "ptr", None)
    _STREAM_TLS.ptr = ptr
    return prev


_hip = None


def hip():
    """the HIP runtime by ctypes (already loaded by torch): only for the three event calls of RawEvent"""
    global _hip
    if _hip is None:
        h = C.CDLL("libamdhip64.so")
        h.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        h.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        h.hipEventDestroy.argtypes = [C.c_void_p]
        _hip = h
    return _hip


class RawEvent(object):
    """hipEvent without timing, recorded on / waited for by raw stream pointers: ~1 us of host time per call where
    torch.cuda.Event.record / Stream.wait_event cost 4-5 us each - the eager multi-GPU step issues six of them per step and is
    host-bound (dist.DistEngine.step_pipelined)."""

    def __init__(self):
        self._e = C.c_void_p()
        if hip().hipEventCreateWithFlags(C.byref(self._e), 0x2) != 0:        # hipEventDisableTiming
            raise KgeError("hipEventCreateWithFlags failed")

    def record(self, stream):
        if hip().hipEventRecord(self._e, getattr(stream, "cuda_stream", stream)) != 0:
            raise KgeError("hipEventRecord failed")

    def wait(self, stream):
        if hip().hipStreamWaitEvent(getattr(stream, "cuda_stream", stream), self._e, 0) != 0:
            raise KgeError("hipStreamWaitEvent failed")

    def __del__(self):
        try:
            if self._e:
                hip().hipEventDestroy(self._e)
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass


class TorchEvent(object):
    """the same two calls on a torch.cuda.Event (communicators / ops that follow torch's current stream)"""

    def __init__(self):
        self._e = torch.cuda.Event()

    def record(self, stream):
        self._e.record(stream)

    def wait(self, stream):
        stream.wait_event(self._e)


@contextlib.contextmanager
def graph_capture(g, stream=None):
    """`with torch.cuda.graph(g, stream=stream)` with Python's cyclic garbage collector held off for the duration of the capture: a
    collection that runs DURING capture may destroy an older engine's graphs / tensors (hipFree, hipGraphDestroy), which
    invalidates the capture and aborts the process (seen when several trainers are built one after the other in one process)."""
    was = gc.isenabled()          # (torch.cuda.graph.__enter__ runs one full collection itself: no second one here - it costs
    gc.disable()                  #  tens of milliseconds and the CLI captures inside its timed training loop)
    try:
        with torch.cuda.graph(g, stream=stream):
            yield
    finally:
        if was:
            gc.enable()


def model_id(name):
    if name not in MODEL_IDS:
        raise KgeError("model %r has no HIP kernel (supported: %s)" % (name, sorted(MODEL_IDS)))
    return MODEL_IDS[name]
