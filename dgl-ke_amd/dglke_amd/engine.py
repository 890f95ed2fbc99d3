"""Fused-step driver: one C-ABI call (`kge_step_fused`) = KEModel.forward + loss.backward() +
KEModel.update of the reference (train_pytorch.py:141-152) for one batch, enqueued on the current
HIP stream with no host synchronisation.  `StepEngine.capture()` records a sequence of steps over
pre-staged batches into a HIP graph so that the launch-bound inner loop is replayed without host
work (one step is only a few microseconds of HBM traffic, SURVEY.md 7 'hard parts')."""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import check, lib, model_id, ptr, stream_ptr


class StepEngine(object):
    def __init__(self, model_name, n_entities, n_relations, hidden_dim, gamma, lr, device,
                 double_entity_emb=False, double_relation_emb=False, neg_adversarial_sampling=False,
                 adversarial_temperature=1.0, regularization_coef=0.0, regularization_norm=3,
                 loss_genre='Logsigmoid', pairwise=False, margin=1.0, flags=0,
                 tables=None, shards=None):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.KgeError("StepEngine needs a CUDA (HIP) device; there is no CPU fallback")
        self.model_name = model_name
        self.n_entities, self.n_relations = n_entities, n_relations
        self.hidden_dim = hidden_dim
        self.gamma = gamma
        self.emb_init = (gamma + 2.0) / hidden_dim          # general_models.py:217-218
        self.d_e = 2 * hidden_dim if double_entity_emb else hidden_dim
        self.d_r = 2 * hidden_dim if double_relation_emb else hidden_dim
        if model_name == 'RESCAL':                          # relation row = [rel_dim x ent_dim] matrix, general_models.py:232-236
            self.d_r = self.d_r * self.d_e
        hp = _lib.KgeHParams()
        hp.model = model_id(model_name)
        hp.d_e, hp.d_r = self.d_e, self.d_r
        hp.loss_genre = _lib.LOSS_IDS[loss_genre]
        hp.adv = int(bool(neg_adversarial_sampling))
        hp.pairwise = int(bool(pairwise))
        hp.reg_norm = int(regularization_norm)
        hp.flags = int(flags)
        hp.gamma = float(gamma)
        hp.emb_init = float(self.emb_init)
        hp.lr = float(lr)
        hp.adv_temp = float(adversarial_temperature)
        hp.margin = float(margin if margin is not None else 1.0)
        hp.reg_coef = float(regularization_coef)
        hp.eps = 1e-10
        self.hp = hp
        import os
        if (shards is not None and shards.n_shards == 1 and not shards.emulate and tables is None and
                os.environ.get("KGE_P2P_LOCAL_SHORTCUT", "1") != "0"):
            # ONE shard (world 1): every row is a row of this process's own arena - the plain single-table step on it (round 6:
            # no shard-map division per row, no dense copy of the negative rows; the emulated-shard tests keep the map)
            if (shards.d_e, shards.d_r) != (self.d_e, self.d_r):
                raise _lib.KgeError("sharded tables have row widths (%d, %d), model needs (%d, %d)"
                                    % (shards.d_e, shards.d_r, self.d_e, self.d_r))
            tables = (shards.ent(0), shards.ent_state(0), shards.rel(0), shards.rel_state(0))
            self.shards_local = shards     # (keeps the arena alive)
            shards = None
        self.shards = shards           # p2p.ShardedTables: the tables live in the peers' HBM
        if shards is not None:
            if (shards.d_e, shards.d_r) != (self.d_e, self.d_r):
                raise _lib.KgeError("sharded tables have row widths (%d, %d), model needs (%d, %d)"
                                    % (shards.d_e, shards.d_r, self.d_e, self.d_r))
            self.ent, self.ent_state = shards.ent(0), shards.ent_state(0)
            self.rel, self.rel_state = shards.rel(0), shards.rel_state(0)
        self.proj = self.proj_state = None      # TransR: projection_emb of the score function (score_fun.py:114-118)
        if shards is not None:
            if model_name in ('TransR', 'RESCAL') and not getattr(shards, 'rel_local', False):
                raise _lib.KgeError("%s on sharded tables needs ShardedTables(rel_local=True%s): relation-side tables local to the rank"
                                    % (model_name, ", proj_dim=d_e*d_r" if model_name == 'TransR' else ""))
            if model_name == 'TransR':
                if shards.proj_tab is None or shards.proj_tab.shape[1] != self.d_e * self.d_r:
                    raise _lib.KgeError("TransR on sharded tables needs ShardedTables(proj_dim=%d)" % (self.d_e * self.d_r))
                self.proj, self.proj_state = shards.proj_tab, shards.proj_state_tab
        elif tables is None:
            if model_name == 'TransR':
                self.proj = torch.empty(n_relations, self.d_e * self.d_r, dtype=torch.float32, device=self.device)
                self.proj_state = torch.zeros(n_relations, dtype=torch.float32, device=self.device)
            self.ent = torch.empty(n_entities, self.d_e, dtype=torch.float32, device=self.device)
            self.ent_state = torch.zeros(n_entities, dtype=torch.float32, device=self.device)
            self.rel = torch.empty(n_relations, self.d_r, dtype=torch.float32, device=self.device)
            self.rel_state = torch.zeros(n_relations, dtype=torch.float32, device=self.device)
            self.reset_parameters()
        else:
            self.ent, self.ent_state, self.rel, self.rel_state = tables[:4]
            if model_name == 'TransR':
                if len(tables) < 6:
                    raise _lib.KgeError("TransR needs tables = (ent, ent_state, rel, rel_state, proj, proj_state)")
                self.proj, self.proj_state = tables[4], tables[5]
        self._bind_tables()
        self.loss4 = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.loss_accum = torch.zeros(4 * _lib.ACC_SLOTS, dtype=torch.float32, device=self.device)
        self._sum4 = torch.zeros(4, dtype=torch.float32, device=self.device)
        # hand-off words of the strict step's first launch (kge_step_out.tickets): zero now, every step returns them to zero
        self.tickets = torch.zeros(_lib.TICKET_INTS, dtype=torch.int32, device=self.device)
        self._ws = None
        self._ws_bytes = 0
        self._graphs = 0               # graphs captured with this engine's workspace baked in (the workspace may not move then)
        self._pipe = None              # kge_pipe of the --async_update pipeline (side stream + events)
        self._aws = None
        self._aws_bytes = 0

    def _bind_tables(self):
        tb = _lib.KgeTables()
        tb.ent, tb.ent_state = ptr(self.ent), ptr(self.ent_state)
        tb.rel, tb.rel_state = ptr(self.rel), ptr(self.rel_state)
        tb.n_ent, tb.n_rel = self.ent.shape[0], self.rel.shape[0]
        tb.proj, tb.proj_state = ptr(self.proj), ptr(self.proj_state)
        self.tb = tb

    def reset_parameters(self):
        torch.nn.init.uniform_(self.ent, -self.emb_init, self.emb_init)
        torch.nn.init.uniform_(self.rel, -self.emb_init, self.emb_init)
        self.ent_state.zero_()
        self.rel_state.zero_()
        if self.proj is not None:                 # TransRScore.reset_parameters: projection_emb.init(1.0)
            torch.nn.init.uniform_(self.proj, -1.0, 1.0)
            self.proj_state.zero_()

    def load_tables(self, ent, rel, ent_state=None, rel_state=None):
        self.ent.copy_(torch.as_tensor(ent))
        self.rel.copy_(torch.as_tensor(rel))
        if ent_state is None:
            self.ent_state.zero_()
        else:
            self.ent_state.copy_(torch.as_tensor(ent_state))
        if rel_state is None:
            self.rel_state.zero_()
        else:
            self.rel_state.copy_(torch.as_tensor(rel_state))

    def workspace_for(self, b):
        need = lib().kge_step_workspace_bytes(C.byref(self.hp), b.B, b.C, b.chunk, b.N, b.UE, b.UR)
        if need > self._ws_bytes:
            if self._graphs:
                raise _lib.KgeError("workspace would be re-allocated after graph capture")
            self._ws = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=self.device)
            self._ws_bytes = self._ws.numel()
        return self._ws

    def step(self, batch, want=None, emit=None, per_step_loss=False, sample_job=None):
        """enqueue one fused step.  `want`: dict of optional output tensors (pos_score, neg_score,
        g_pos_ent, g_neg, g_rel).  `emit`: KgeEmit for the sharded path.  `sample_job` (KgeSamplerJob, DeviceSampler.tail_jobs):
        the step's launches also BUILD that batch of the next group (kge_step_fused_sampling; strict step on local tables)."""
        ws = self.workspace_for(batch)
        out = _lib.KgeStepOut()
        if want is not None or per_step_loss:
            out.loss4 = ptr(self.loss4)
        out.loss_accum = ptr(self.loss_accum)
        out.tickets = ptr(self.tickets)
        if want:
            for k, t in want.items():
                setattr(out, k, ptr(t))
        if sample_job is not None and (self.shards is not None or emit is not None):
            raise _lib.KgeError("the sampler tail rides on the strict single-table step only")
        if self.shards is not None:
            if emit is not None:
                raise _lib.KgeError("gradient emission and peer-to-peer sharding are different multi-GPU modes")
            check(lib().kge_step_sharded(C.byref(self.hp), C.byref(self.shards.c), C.byref(batch.c),
                                         C.byref(out), ptr(ws), self._ws_bytes, stream_ptr()))
        elif emit is None and sample_job is not None:
            check(lib().kge_step_fused_sampling(C.byref(self.hp), C.byref(self.tb), C.byref(batch.c),
                                                C.byref(out), ptr(ws), self._ws_bytes, C.byref(sample_job), stream_ptr()))
        elif emit is None:
            check(lib().kge_step_fused(C.byref(self.hp), C.byref(self.tb), C.byref(batch.c),
                                       C.byref(out), ptr(ws), self._ws_bytes, stream_ptr()))
        else:
            check(lib().kge_step_grads(C.byref(self.hp), C.byref(self.tb), C.byref(batch.c),
                                       C.byref(out), C.byref(emit), ptr(ws), self._ws_bytes,
                                       stream_ptr()))

    def step_timed(self, batch):
        """ONE strict step issued as its four phase groups (kge_step_phase: gather+positive / negative scores+loss /
        gradients / Adagrad - together exactly kge_step_fused) with HIP events in between; returns the four durations in
        seconds (forward = gather + scores + loss, like the reference's timer around model.forward).  Synchronises."""
        if self.shards is not None:
            self.step(batch)
            return None
        ws = self.workspace_for(batch)
        out = _lib.KgeStepOut()
        out.loss_accum = ptr(self.loss_accum)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        for k, ph in enumerate((_lib.PHASE_GATHER, _lib.PHASE_FORWARD, _lib.PHASE_BACKWARD, _lib.PHASE_UPDATE)):
            check(lib().kge_step_phase(C.byref(self.hp), C.byref(self.tb), C.byref(batch.c), C.byref(out), ptr(ws),
                                       self._ws_bytes, ph, stream_ptr()))
            ev[k + 1].record()
        ev[4].synchronize()
        t = [ev[k].elapsed_time(ev[k + 1]) * 1e-3 for k in range(4)]
        return dict(forward=t[0] + t[1], backward=t[2], update=t[3])

    # ---- --async_update: UPDATE(s-1) in the same launch as backward(s) (include/kge_hip.h, kge_step_async) ----
    def async_workspace_for(self, b):
        need = lib().kge_step_async_workspace_bytes(C.byref(self.hp), b.B, b.C, b.chunk, b.N, b.UE, b.UR)
        if need > self._aws_bytes:
            if self._graphs:
                raise _lib.KgeError("workspace would be re-allocated after graph capture")
            self._aws = torch.empty(int(need * 1.25) + 8192, dtype=torch.uint8, device=self.device)
            self._aws_bytes = self._aws.numel()
        return self._aws

    def step_async(self, batch, want=None, per_step_loss=False, next_batch=None):
        """enqueue one step of the --async_update pipeline: gathers rows that contain every update up to step s-2's
        (inside one flush-to-flush group), applies step s-1's update in the same launch as this step's forward tiles.
        `next_batch`: the batch of the following step_async call (optional): with KGE_FLAG_ASYNC_REL its PREP shares
        this step's backward launch.  `flush_async()` must follow the last step (before the tables are read / before a
        stream capture ends)."""
        if self.shards is not None:
            raise _lib.KgeError("--async_update is not available on peer-to-peer sharded tables")
        if self._pipe is None:
            h = C.c_void_p()
            check(lib().kge_pipe_create(C.byref(h)))
            self._pipe = h
        ws = self.async_workspace_for(batch)
        out = _lib.KgeStepOut()
        if want is not None or per_step_loss:
            out.loss4 = ptr(self.loss4)
        out.loss_accum = ptr(self.loss_accum)
        if want:
            for k, t in want.items():
                setattr(out, k, ptr(t))
        check(lib().kge_step_async(self._pipe, C.byref(self.hp), C.byref(self.tb), C.byref(batch.c),
                                   C.byref(next_batch.c) if next_batch is not None else None, C.byref(out),
                                   ptr(ws), self._aws_bytes, stream_ptr()))
        self._async_keep = (batch, next_batch)       # the pending update / the prefetched PREP read their plan arrays

    def steps_async(self, batches):
        """a group of steps of the pipeline, every step naming its successor, flushed at the end"""
        for k, b in enumerate(batches):
            self.step_async(b, next_batch=batches[k + 1] if k + 1 < len(batches) else None)
        self.flush_async()

    def flush_async(self):
        if self._pipe is not None:
            check(lib().kge_step_async_flush(self._pipe, stream_ptr()))

    def __del__(self):
        try:
            if self._pipe is not None:
                lib().kge_pipe_destroy(self._pipe)
                self._pipe = None
        except Exception:
            pass

    def alloc_outputs(self, batch):
        dev = self.device
        # neg_deg_sample: every chunk is scored against its own positives too (N' = chunk + N rows)
        Np = batch.N + (batch.chunk if self.hp.flags & _lib.FLAG_NEG_DEG_SAMPLE else 0)
        return dict(pos_score=torch.empty(batch.B, device=dev),
                    neg_score=torch.empty(batch.C, batch.chunk, Np, device=dev),
                    g_pos_ent=torch.empty(batch.UE, self.d_e, device=dev),
                    g_neg=torch.empty(batch.C * Np, self.d_e, device=dev),
                    g_rel=torch.empty(batch.B, self.d_r, device=dev))

    def capture(self, batches, stream=None, async_update=False):
        """record `len(batches)` consecutive steps into one HIP graph (on `stream` if given);
        returns the graph.  async_update: the --async_update pipeline, flushed at the end of the group."""
        for b in batches:
            (self.async_workspace_for if async_update else self.workspace_for)(b)
        # warm-up on a side stream is not needed: the library allocates nothing
        g = torch.cuda.CUDAGraph()
        with _lib.graph_capture(g, stream=stream):
            if async_update:
                self.steps_async(batches)
            else:
                for b in batches:
                    self.step(b)
        g._kge_batches = batches       # the graph reads the batches' arrays: they live as long as the graph does
        self._graphs += 1
        return g

    def read_loss(self):
        """[pos_loss, neg_loss, loss, regularization] of the last step run with `want` or
        `per_step_loss` (one 16-byte D2H copy)."""
        return self.loss4.tolist()

    def read_loss_sums(self, zero=True):
        """sums of the four loss terms over all steps since the last reset (one reduction
        kernel + one 16-byte D2H copy); the caller divides by its step count."""
        check(lib().kge_reduce_loss(ptr(self.loss_accum), ptr(self._sum4), int(zero), stream_ptr()))
        return self._sum4.tolist()
