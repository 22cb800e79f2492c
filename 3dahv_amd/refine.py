"""Coarse-to-fine verification captured as ONE hipGraph (BASELINE.json configs[4]).

Build-defined: the reference scores one flat set of 50 000 random rotations
(test_objaverse.py:17, modules/model.py:184).  Here stage 1 scores N1 coarse hypotheses, stage 2
scores N2 refinements ``R* @ D[n]`` around each sample's stage-1 winner, where ``D`` is a fixed set of
small rotations (``rotations.refine_rotations(I, N2, max_angle)``; D[0] = I, so stage 2 can never
score below stage 1).  Everything between the two stages stays on the device: the winner index is
decoded from the packed key by ``ahv_compose_rotations_f32``; no host round trip, so the whole step
(2 fused scorer launches -- the first builds the target features in-launch --, compose, 2 selects) replays from a graph.
On one rank the step also exists as ONE launch (``fused=True``: ``ahv_coarse_to_fine_f32``, the workgroups meet at a
device-wide counter between the stages).

Multi-rank (one process per GPU): both hypothesis sets are sharded contiguously (``dist.shard_range``);
each stage ends in the 8*B-byte packed-key all-reduce(max) (int64 MAX on the key as the kernel packs it) -- two collectives per
step; the winner's rotation row needs no exchange (every rank composes the full refinement set of the
winner).  The verify semantics per stage
are those of modules/model.py:183-196.  With the ``nccl`` backend (= RCCL) the collectives are enqueued on
the capturing stream like any kernel, so the two stages AND their all-reduces replay from one hipGraph
(SURVEY.md section 8(d) cfg 5); other backends (gloo rehearsals, CPU tests) run the step eagerly.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from .dist import KEY_EMPTY, shard_range
from .rotations import refine_rotations


class CoarseToFine:
    """``backend`` provides ``verify_pair, score_hypotheses, compose_rotations, select_rotation`` with the signatures of
    ``3dahv_amd.ops`` (the default and the only product backend: HIP kernels, no CPU path); CPU tests inject an
    oracle-backed object to execute the multi-rank control flow under gloo.

    One step = FIVE launches: the coarse stage as one ``verify_pair`` launch (the target features are built inside it and
    kept for the fine stage), ``compose_rotations``, the fine stage, and one ``select_rotation`` per stage, each of which
    also hands its key back empty for the next step (no clearing launches).
    ``fused=True`` (one rank, no collectives, the HIP backend): the step is ONE launch, ``ops.coarse_to_fine`` -- both
    stages, a device-wide meeting point between them and the decoding of both keys inside ``ahv_coarse_to_fine_f32``.
    Same results bit for bit; measured 1.5 % faster (189.7 against 192.6 us for 10 000 + 1 000 hypotheses: the queue
    already hides the launches it removes), and its meeting point assumes nothing else holds compute units for long --
    hence opt-in.
    Scores do not depend on how the hypothesis sets are split over ranks, bit for bit (a team's score is a lone wave's);
    ``no_teams`` is the scheduling knob of ``ops.score_hypotheses``.  ``check()`` (host sync) raises if a one-launch step
    had to give its meeting point up -- such a step's outputs are poisoned (NaN, -1), never plausible.
    ``use_graph``: None = captured when the step carries collectives, eager otherwise (see __init__); ``run_many`` replays
    several steps from one graph."""

    def __init__(self, W1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor, R_coarse: torch.Tensor,
                 D: Optional[torch.Tensor] = None, n_fine: int = 1000, max_angle_deg: float = 10.0,
                 batch: int = 1, use_graph: Optional[bool] = None, group=None, seed: int = 0, backend=None,
                 want_scores: bool = False, force_collectives: bool = False, no_teams: bool = False,
                 fused: Optional[bool] = None):
        dev = R_coarse.device
        self.ops = ops if backend is None else backend
        self.W1, self.W2, self.b2 = W1, W2, b2
        self.R_coarse = R_coarse.contiguous()
        if D is None:
            g = torch.Generator(device="cpu").manual_seed(seed)
            D = refine_rotations(torch.eye(3), n_fine, max_angle_deg, generator=g)
        self.D = D.to(dev).contiguous()
        self.B = batch
        self.group = group
        self.want_scores = want_scores
        self.no_teams = no_teams
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        # a 1-rank RCCL group with force_collectives exercises "collectives inside the captured graph" on one GPU
        self.collectives = self.world > 1 or (force_collectives and inited)
        self.c_lo, self.c_hi = shard_range(self.R_coarse.shape[0], self.rank, self.world)
        self.f_lo, self.f_hi = shard_range(self.D.shape[0], self.rank, self.world)
        capturable = (not self.collectives) or (inited and dist.get_backend(group) == "nccl")
        # Default (use_graph=None), from the kernel-trace timelines of profiles/r06_graph_timeline.txt: a hipGraphLaunch idles
        # the device ~9 us between two replays, plain launches none -- so WITHOUT collectives a single step is issued eagerly
        # (200.9 against 207.9 us; run_many captures several steps per graph and replays at 199.9); WITH collectives the
        # eager step idles ~20 us inside itself (the event packets around each all-reduce) and the graph wins (207.8
        # against 220.0 us), so a multi-rank step is captured whenever the backend can be (RCCL).
        if use_graph is None:
            use_graph = self.collectives
        self.use_graph = bool(use_graph and dev.type == "cuda" and capturable)
        # the two keys live with the object: every step's select hands them back empty
        self._keys = [torch.full((batch,), KEY_EMPTY, dtype=torch.int64, device=dev) for _ in range(2)]
        self._R_fine = torch.empty((batch, self.D.shape[0], 3, 3), dtype=torch.float32, device=dev)
        self._graph = None
        self._static = None
        can_fuse = backend is None and not self.collectives and self.world == 1 and dev.type == "cuda"
        if fused and not can_fuse:
            raise RuntimeError("the one-launch step needs one rank, no collectives and the HIP backend")
        self.fused = bool(fused)
        self._fused_state = ops.CoarseToFineState(batch, dev) if self.fused else None
        self._fused_out = {}     # slot -> the one-launch step's output buffers (a slot = one step of a multi-step graph)
        self._many = None        # (K, static inputs, captured graph, outputs) of run_many

    def _merge(self, key):
        if self.collectives:  # world > 1, or forced on a 1-rank group: same call, same captured node
            dist.all_reduce(key, op=dist.ReduceOp.MAX, group=self.group)
        return key

    @property
    def buffers(self):
        """The static input volumes ``(vol_src, vol_tgt)`` of the captured step (allocated on first use).  A producer that
        writes its volumes straight into them -- ``forward_2d3d(..., out=c2f.buffers)`` -- and then calls ``c2f()`` with no
        arguments replays the graph with no staging copy."""
        if self._static is None:
            dev = self.R_coarse.device
            self._static = tuple(torch.zeros((self.B, 16, 8, 8, 8), dtype=torch.float32, device=dev) for _ in range(2))
        return self._static

    # ---- the step, written once; runs eagerly or under capture
    def _step(self, vol_src, vol_tgt, slot: int = 0):
        o = self.ops
        if self.fused:
            r = o.coarse_to_fine(vol_src, vol_tgt, self.R_coarse, self.D, self.W1, self.W2, self.b2, state=self._fused_state,
                                 want_scores=self.want_scores, no_teams=self.no_teams, out=self._fused_out.setdefault(slot, {}))
            # (the refinement set is never materialised here: R_fine stays None)
            self.last = {"coarse_scores": r.get("coarse_scores"), "fine_scores": r.get("fine_scores"), "R_fine": None}
            return r["fine_score"], r["fine_idx"], r["R_pred"], r["coarse_score"], r["coarse_idx"]
        key1, key2 = self._keys
        kw = {"no_teams": True} if self.no_teams else {}
        Rc = self.R_coarse[self.c_lo:self.c_hi]
        s1, _, f_tgt = o.verify_pair(vol_src, vol_tgt, Rc, self.W1, self.W2, self.b2, n_offset=self.c_lo,
                                     want_scores=self.want_scores, best_key=key1, reset_best=False, want_feat_tgt=True, **kw)
        self._merge(key1)
        # Every rank holds the whole coarse set AND the whole refinement set D, so after the key all-reduce each
        # rank composes ALL N2 refinements of the winner locally (N2 * B threads) and scores its own slice of them.
        # After the second key all-reduce every rank knows both winning indices and already holds the winning
        # row: R_pred is a local gather -- two collectives per step, not three.
        R_fine_all = o.compose_rotations(key1, self.R_coarse, self.D, out=self._R_fine)
        R_fine = R_fine_all if self.world == 1 else R_fine_all[:, self.f_lo:self.f_hi]
        s2, _ = o.score_hypotheses(vol_src, f_tgt, R_fine, self.W1, self.W2, self.b2, n_offset=self.f_lo,
                                   want_scores=self.want_scores, best_key=key2, reset_best=False, **kw)
        self._merge(key2)
        score, idx, R_pred = o.select_rotation(key2, R_fine_all, n_offset=0, reset_key=True)
        coarse_score, coarse_idx, _ = o.select_rotation(key1, self.R_coarse, n_offset=0, reset_key=True)
        # this rank's slices of the two score sets and of the refinement set (None unless want_scores)
        self.last = {"coarse_scores": s1, "fine_scores": s2, "R_fine": R_fine if self.want_scores else None}
        return score, idx, R_pred, coarse_score, coarse_idx

    def check(self):
        """Host sync.  Raises if a one-launch step abandoned its device-wide meeting point (another kernel held compute units
        for about a second: the stage-1 scores were then taken against an incomplete coarse winner).  The kernel poisons such a
        step's outputs (NaN scores, index -1, NaN rotation), so the failure cannot be mistaken for a result; this names it."""
        if self.fused and self._fused_state.gave_up():
            raise RuntimeError("ahv_coarse_to_fine_f32: a workgroup gave up the meeting point (the device was shared with "
                               "another kernel for ~1 s); the outputs of that step AND of every step since are poisoned "
                               "(NaN / -1): the error word is sticky. Call clear_error() to use this object again, or use "
                               "fused=False when other work runs on the GPU.")

    def clear_error(self):
        """After a give-up (``check()`` raised): reset the one-launch step's state -- keys, meeting-point counters, the sticky
        error word -- so that the next step runs clean."""
        if self._fused_state is not None:
            self._fused_state.clear_error()

    def _reset_keys(self):
        """After an exception inside a step the persistent keys may hold half a step's winners: hand them back empty."""
        for k in self._keys:
            k.fill_(KEY_EMPTY)
        if self._fused_state is not None:
            self._fused_state.keys.fill_(KEY_EMPTY)
            self._fused_state.sync[:-1].zero_()

    @torch.no_grad()
    def __call__(self, vol_src: Optional[torch.Tensor] = None, vol_tgt: Optional[torch.Tensor] = None):
        """vol_src, vol_tgt (B,16,8,8,8) -> (fine score (B,), fine index (B,), R_pred (B,3,3),
        coarse score (B,), coarse index (B,)).  With ``use_graph`` the outputs are static buffers that
        the next call overwrites; called with no arguments the step runs on ``self.buffers`` as they are."""
        if (vol_src is None) != (vol_tgt is None):
            raise RuntimeError("pass both volumes or neither")
        if not self.use_graph:
            if vol_src is None:
                vol_src, vol_tgt = self.buffers
            try:
                return self._step(vol_src, vol_tgt)
            except Exception:
                self._reset_keys()  # a step that raised midway must not leave its winners to the next one
                raise
        static = self.buffers
        if vol_src is not None and vol_src.data_ptr() != static[0].data_ptr():
            static[0].copy_(vol_src)
        if vol_tgt is not None and vol_tgt.data_ptr() != static[1].data_ptr():
            static[1].copy_(vol_tgt)
        if self._graph is None:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):  # warm-up outside capture (lazy initialisation inside the launchers / RCCL)
                for _ in range(2):
                    self._step(*static)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._out = self._step(*static)
        self._graph.replay()
        return self._out

    @torch.no_grad()
    def run_many(self, vol_src: Optional[torch.Tensor] = None, vol_tgt: Optional[torch.Tensor] = None, steps: int = 8):
        """``steps`` consecutive verify steps -- volumes ``(steps, B, 16, 8, 8, 8)`` -- as ONE hipGraph launch; returns a list
        of ``steps`` result tuples (static buffers with ``use_graph``).  Why it exists: a hipGraphLaunch leaves the device
        idle for ~9 us between the last kernel of one replay and the first of the next (ROCm 7.2, `rocprofv3
        --kernel-trace`, profiles/r06_graph_timeline.txt) while plain launches follow each other with no gap, so a graph of
        ONE 0.2-ms step replays 4 % slower than the same step issued eagerly; with several steps per graph the gap is paid
        once per replay and the captured steps run back to back.  Without ``use_graph`` the steps are issued eagerly.
        Called with no volumes it runs on the static inputs as they are (``many_buffers(steps)``)."""
        if (vol_src is None) != (vol_tgt is None):
            raise RuntimeError("pass both volumes or neither")
        if not self.use_graph:
            if vol_src is None:
                vol_src, vol_tgt = self.many_buffers(steps)
            try:
                return [self._step(vol_src[k], vol_tgt[k], slot=k) for k in range(steps)]
            except Exception:
                self._reset_keys()
                raise
        static = self.many_buffers(steps)
        if vol_src is not None and vol_src.data_ptr() != static[0].data_ptr():
            static[0].copy_(vol_src)
            static[1].copy_(vol_tgt)
        if self._many[2] is None:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):  # warm-up outside capture
                for k in range(min(2, steps)):
                    self._step(static[0][k], static[1][k], slot=k)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = [self._step(static[0][k], static[1][k], slot=k) for k in range(steps)]
            self._many[2], self._many[3] = graph, outs
        self._many[2].replay()
        return self._many[3]

    def many_buffers(self, steps: int):
        """Static inputs ``(vol_src, vol_tgt)``, each ``(steps, B, 16, 8, 8, 8)``, of ``run_many``."""
        if self._many is None or self._many[0] != steps:
            dev = self.R_coarse.device
            bufs = tuple(torch.zeros((steps, self.B, 16, 8, 8, 8), dtype=torch.float32, device=dev) for _ in range(2))
            self._many = [steps, bufs, None, None]
        return self._many[1]
