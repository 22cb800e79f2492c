"""Evaluation-harness counterpart of the reference's ``test_co3d.py`` / ``test_linemod.py`` (SURVEY.md
section 8a row H1) with a synthetic-pair mode (no dataset or checkpoint exists offline).

What is reproduced from the reference scripts:
* seeding ``torch.manual_seed(0); np.random.seed(0)``                      test_co3d.py:24-25
* ONE proposal set per category, reused for every pair                      test_co3d.py:106
* ``key_frames = np.random.choice(n, num_frames, replace=False)``            test_co3d.py:112
* ordered pairs (i, j), i != j                                               test_co3d.py:47-53
* ``R_gt = R_i^T R_j`` (pytorch3d row-vector convention)                     test_co3d.py:121-124
* per pair: model -> volumes -> verify step -> ``R_pred`` -> angular error   test_co3d.py:133-152
* per category: mean error, ``100*mean(err<30)``, ``100*mean(err<15)``        test_co3d.py:180-182
* "mean" over categories, ``repeats`` repetitions averaged, result lines
  ``f"{category:>10s}{err:6.02f}{acc15:6.02f}{acc30:6.02f}"`` appended to
  ``models/<RUN_NAME>/co3d_result.txt``                                      test_co3d.py:186-252
* LINEMOD: per-object loop, ``np.savetxt(linemod_pred_Rs_%06d.txt)``          test_linemod.py:20-86

The verify step (rotate + project + score + arg-max over all proposals) is the fused HIP launch
(``model.verify``).  ``verify_fn`` can be injected: the CPU-tier tests pass the oracle there to
exercise this file's plumbing without a GPU -- the product default has no CPU path.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, Optional

import numpy as np
import torch

from .rotations import geodesic_deg, random_rotations


def get_permutations(num_frames: int) -> torch.Tensor:
    return torch.tensor([(i, j) for i in range(num_frames) for j in range(num_frames) if i != j])


class SyntheticSequences:
    """Synthetic stand-in for ``Co3dDataset``: ``n_seq`` sequences of ``n_frames`` views.  Each item is a
    dict ``{"n", "model_id", "R" (n,3,3)}`` plus either ``"image" (n,3,256,256)`` or, when
    ``layer4=True``, ``"layer4" (n,768,8,8)`` backbone features (for models without a backbone)."""

    def __init__(self, n_seq: int = 4, n_frames: int = 6, layer4: bool = True, seed: int = 0, in_channel: int = 768):
        self.n_seq, self.n_frames, self.layer4, self.seed, self.in_channel = n_seq, n_frames, layer4, seed, in_channel

    def __len__(self):
        return self.n_seq

    def __iter__(self):
        for s in range(self.n_seq):
            g = torch.Generator().manual_seed(self.seed * 1000 + s)
            item = {"n": self.n_frames, "model_id": "synthetic_%03d" % s,
                    "R": random_rotations(self.n_frames, generator=g)}
            if self.layer4:
                item["layer4"] = torch.randn(self.n_frames, self.in_channel, 8, 8, generator=g)
            else:
                item["image"] = torch.rand(self.n_frames, 3, 256, 256, generator=g) * 2 - 1
            yield item


class _FlushUploader:
    """Host -> device staging of one flush of the evaluation loop: everything a flush needs (the key frames of all queued
    sequences, their rotations, the pair indices) travels as ONE pinned fp32 buffer in ONE non-blocking copy, so the host
    never waits for the GPU -- a pageable ``.to(device)`` is ordered on the compute stream and blocks the host until
    everything in front of it has run (13 % of the batched per-pair time in round 3, profiles/r03g_evaluation_loop.jsonl).
    Large flushes go through a copy stream of their own and overlap the previous flush's kernels; small ones (a pair or
    two) stay on the compute stream: a cross-stream event wait costs that queue ~30 us on this stack
    (profiles/r04_forced_pg_timeline.txt), more than their 20-us copy.  Two buffer sets alternate; one uploader per
    device lives for the process (pinned allocations cost milliseconds)."""
    SIDE_STREAM_BYTES = 2 << 20

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device)
        self.slots = [None, None]
        self.turn = 0
        self.last = None

    def upload(self, parts):
        """[[host fp32 arrays to be laid end to end], ...] -> [device fp32 tensor per part, flat]; the compute stream
        waits for the copy, the host does not (it only waits for the copy that used this slot two flushes ago).  The
        host side is numpy on purpose: torch's CPU operators wake one thread per VISIBLE core (256 here, inside a 16-CPU
        cgroup) and cost milliseconds."""
        sizes = [sum(int(a.size) for a in part) for part in parts]
        total = sum(sizes)
        slot = self.slots[self.turn]
        cur = torch.cuda.current_stream(self.device)
        if slot is None or slot[0].numel() < total:
            cap = 1 << max(20, int(total - 1).bit_length())   # >= 4 MB, powers of two: pinned allocations cost milliseconds
            slot = (torch.empty(cap, dtype=torch.float32).pin_memory(),
                    torch.empty(cap, dtype=torch.float32, device=self.device), torch.cuda.Event(), torch.cuda.Event())
            slot[3].record(cur)
            self.slots[self.turn] = slot
        self.last = slot
        self.turn ^= 1
        pin, dev_buf, ev, used = slot
        ev.synchronize()
        host = pin.numpy()
        o = 0
        for part in parts:
            for a in part:
                host[o:o + a.size] = a.reshape(-1)
                o += a.size
        if total * 4 >= self.SIDE_STREAM_BYTES:
            self.stream.wait_event(used)          # the flush that last read this device buffer (two flushes ago) is past it
            with torch.cuda.stream(self.stream):
                dev_buf[:total].copy_(pin[:total], non_blocking=True)
                ev.record(self.stream)
            cur.wait_event(ev)
        else:
            dev_buf[:total].copy_(pin[:total], non_blocking=True)   # stream-ordered: no event packets
            ev.record(cur)
        out, o = [], 0
        for n in sizes:
            out.append(dev_buf[o:o + n])
            o += n
        return out

    def release(self):
        """The compute stream has issued its last read of the buffer handed out by the latest ``upload``."""
        self.last[3].record(torch.cuda.current_stream(self.device))


_UPLOADERS = {}


def _uploader(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _UPLOADERS:
        _UPLOADERS[key] = _FlushUploader(device)
    return _UPLOADERS[key]


@torch.no_grad()
def evaluate_category(cfg, model, sequences: Iterable[dict], num_frames: int = 2, device=None,
                      proposals: Optional[torch.Tensor] = None, verify_fn: Optional[Callable] = None,
                      return_details: bool = False, batch_pairs: Optional[bool] = None,
                      encoder_fn: Optional[Callable] = None, batch_sequences: Optional[int] = None):
    """Counterpart of ``evaluate_category`` (test_co3d.py:93-154).  Returns the array of angular errors.

    Per-pair results stay on the device and are fetched once per category (the reference synchronises per pair with
    ``.item()``).  ``encoder_fn(layer4_src, layer4_tgt)`` replaces ``model.forward_features`` for ``layer4`` inputs,
    e.g. ``model.feature_aligner.graphed_forward_2d3d(2)`` to replay the encoder's 63 launches from one hipGraph.

    ``batch_pairs`` (default: on the GPU) runs the ordered pairs of a sequence -- (0,1) and (1,0) for two frames --
    as ONE batch through the encoder and ONE fused verify launch instead of one by one as the reference does, and
    ``batch_sequences`` (default: 16 on the GPU, 1 on the CPU) queues that many sequences per batch: at B = 1 the encoder
    is launch-latency bound (0.31 ms for one pair, 0.07 ms per pair at 32).  Results are the same (every kernel on the
    path is row-independent) and ``np.random.choice`` is drawn per sequence in the reference's order, so a seed selects
    the reference's pairs whatever the batching."""
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    if batch_pairs is None:
        batch_pairs = on_gpu
    if batch_sequences is None:
        batch_sequences = 16 if on_gpu else 1
    permutations = get_permutations(num_frames)
    perm_np = permutations.numpy()
    if proposals is None:
        proposals = random_rotations(cfg["DATA"]["NUM_ROTA"])
    proposals = proposals.to(device)
    if verify_fn is None:
        verify_fn = lambda vs, vt, P: model.verify(vs, vt, P)[1:]  # (best, idx, R_pred): one fused launch + one select
    uploader = _uploader(device) if on_gpu else None
    details, pending = [], []
    queue = []  # per queued pair group: (frames (F,...) host, R (F,3,3) host, src ids, tgt ids, model_id, uses layer4, pairs)

    def flush():
        """One upload, one encoder call, one fused verify launch and one select for every queued ordered pair."""
        if not queue:
            return
        n_frames = sum(q[0].shape[0] for q in queue)
        frame_shape = queue[0][0].shape[1:]
        base = np.cumsum([0] + [q[0].shape[0] for q in queue[:-1]])
        pair = np.stack([np.concatenate([b + q[2] for b, q in zip(base, queue)]),
                         np.concatenate([b + q[3] for b, q in zip(base, queue)])])          # (2, P): source / target frame
        if uploader is not None:
            frames, rots, pair_f = uploader.upload([[q[0] for q in queue], [q[1] for q in queue], [pair.astype(np.float32)]])
            frames, rots = frames.view((n_frames,) + frame_shape), rots.view(n_frames, 3, 3)
            pair_d = pair_f.view(2, -1).to(torch.int64)   # frame numbers are small integers: exact in fp32
            src, tgt = pair_d[0], pair_d[1]
        else:
            frames = torch.from_numpy(np.concatenate([q[0] for q in queue]))
            rots = torch.from_numpy(np.concatenate([q[1] for q in queue]))
            src, tgt = torch.from_numpy(pair[0].astype(np.int64)), torch.from_numpy(pair[1].astype(np.int64))
        R_gt = torch.bmm(rots.index_select(0, src).transpose(1, 2), rots.index_select(0, tgt))
        embed = (encoder_fn or model.forward_features) if queue[0][5] else model
        vol_src, vol_tgt = embed(frames.index_select(0, src), frames.index_select(0, tgt))
        if uploader is not None:
            uploader.release()
        res = verify_fn(vol_src, vol_tgt, proposals)
        best, idx = res[0], res[1]
        R_pred = res[2] if len(res) > 2 else proposals[idx.reshape(-1)]
        err = geodesic_deg(R_pred, R_gt).reshape(-1)
        pending.append(err)                                                # stays on the device: no sync per pair
        if return_details:
            bs, ids, el = best.reshape(-1).tolist(), idx.reshape(-1).tolist(), err.tolist()
            k = 0
            for q in queue:
                for pr in q[6]:
                    details.append({"model_id": q[4], "pair": pr, "best": float(bs[k]), "idx": int(ids[k]),
                                    "R_pred": R_pred[k].cpu().numpy(), "err": el[k]})
                    k += 1
        queue.clear()

    pairs_per_flush = (len(permutations) if batch_pairs else 1) * max(int(batch_sequences), 1)
    for meta in sequences:
        key_frames = np.random.choice(meta["n"], num_frames, replace=False)
        if "get_data" in meta:  # lazy source (co3d.Co3dSequences): decode only the key frames (test_co3d.py:112)
            meta = dict(meta, **meta["get_data"](key_frames))
            key_frames = np.arange(num_frames)
        uses_l4 = "layer4" in meta
        # the key frames are picked on the HOST with numpy (torch's CPU indexing spins up one thread per visible core,
        # which costs milliseconds per call inside a small cgroup) and uploaded once per flush
        kf = np.asarray(key_frames)
        data = meta["layer4"] if uses_l4 else meta["image"]
        frames = data.cpu().numpy()[kf].astype(np.float32, copy=False)
        rot = meta["R"].cpu().numpy()[kf].astype(np.float32, copy=False)
        if batch_pairs:
            queue.append((frames, rot, perm_np[:, 0], perm_np[:, 1], meta["model_id"], uses_l4,
                          [tuple(p) for p in perm_np.tolist()]))
            if sum(len(q[2]) for q in queue) >= pairs_per_flush:
                flush()
        else:
            for i in range(len(perm_np)):  # the reference's order: one ordered pair at a time
                queue.append((frames, rot, perm_np[i:i + 1, 0], perm_np[i:i + 1, 1], meta["model_id"], uses_l4,
                              [tuple(perm_np[i].tolist())]))
                if sum(len(q[2]) for q in queue) >= pairs_per_flush:
                    flush()
    flush()
    errors = torch.cat(pending).tolist() if pending else []               # ONE host synchronisation per category
    errors = np.array(errors)
    return (errors, details) if return_details else errors


def evaluate_pairwise(cfg, model, categories: Dict[str, Iterable[dict]], num_frames: int = 2, print_results=True,
                      **kw):
    """Counterpart of ``evaluate_pairwise`` (test_co3d.py:157-198)."""
    errors, errors_15, errors_30 = {}, {}, {}
    for category, sequences in categories.items():
        e = evaluate_category(cfg, model, sequences, num_frames=num_frames, **kw)
        errors[category] = np.mean(e)
        errors_15[category] = 100 * np.mean(e < 15)
        errors_30[category] = 100 * np.mean(e < 30)
        if print_results:
            print(category + " err: %.2f || acc_30: %.2f || acc_15: %.2f " % (errors[category], errors_30[category],
                                                                                errors_15[category]))
    errors["mean"] = np.mean(list(errors.values()))
    errors_15["mean"] = np.mean(list(errors_15.values()))
    errors_30["mean"] = np.mean(list(errors_30.values()))
    return errors, errors_30, errors_15


def format_result_line(category: str, err: float, acc15: float, acc30: float) -> str:
    return f"{category:>10s}{err:6.02f}{acc15:6.02f}{acc30:6.02f}"


def run_co3d(cfg, model, categories: Dict[str, Iterable[dict]], repeats: int = 5, out_dir: Optional[str] = None,
             **kw):
    """Counterpart of ``test_co3d.py.__main__`` (:201-253): seeds, ``repeats`` runs averaged, result file."""
    torch.manual_seed(0)
    np.random.seed(0)
    acc = {}
    for _ in range(repeats):
        e, e30, e15 = evaluate_pairwise(cfg, model, categories, print_results=False, **kw)
        for c in e:
            acc.setdefault(c, []).append((e[c], e15[c], e30[c]))
    lines = []
    for c, vals in acc.items():
        m = np.asarray(vals).mean(axis=0)
        lines.append(format_result_line(c, m[0], m[1], m[2]))
    if out_dir is None:
        out_dir = os.path.join("models", cfg["RUN_NAME"])
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "co3d_result.txt"), "a") as f:
        for line in lines:
            f.write(line + " \n")
    return lines


@torch.no_grad()
def test_linemod_category(cfg, model, loader: Iterable[dict], clsID: int, out_dir: Optional[str] = None,
                          proposals_fn: Optional[Callable] = None):
    """Evident intent of ``test_category`` (test_linemod.py:20-86; the shipped script does not run, SURVEY.md
    section 2 row 7): per batch a fresh codebook, verify, angular error (<, not <=), ``linemod_pred_Rs`` file."""
    pred_Rs, errs = [], []
    for data in loader:
        mask_src, mask_tgt = data["src_mask"], data["ref_mask"]
        thr = cfg["DATA"]["SIZE_THR"]
        if torch.any(mask_src.flatten(1).sum(dim=-1) < thr) or torch.any(mask_tgt.flatten(1).sum(dim=-1) < thr):
            print("Skip bad case")
            continue
        dev = data["src_img"].device
        codebook = proposals_fn() if proposals_fn else random_rotations(model.num_rota, device=dev)
        vol_src, vol_tgt = model(data["src_img"], mask_src, data["ref_img"], mask_tgt)
        gt = torch.bmm(data["ref_R"], torch.inverse(data["src_R"]))
        _, _, _, R_pred = model.verify(vol_src, vol_tgt, codebook.to(dev))
        errs.append(geodesic_deg(R_pred, gt))
        pred_Rs.append(R_pred.cpu().numpy().reshape(-1))
    err = torch.cat(errs)
    acc30, acc15 = 100 * (err < 30).float().mean().item(), 100 * (err < 15).float().mean().item()
    if out_dir is None:
        out_dir = os.path.join("models", cfg["RUN_NAME"])
    os.makedirs(out_dir, exist_ok=True)
    np.savetxt(os.path.join(out_dir, "linemod_pred_Rs_%06d.txt" % clsID), np.asarray(pred_Rs))
    return err.mean().item(), acc30, acc15


class SyntheticTrainingPairs:
    """Synthetic stand-in for the CO3D training loader (modules/model_co3d.py:101-134): batches with the keys
    ``training_step`` reads -- ``image (B,2,3,S,S)`` and ``relative_rotation (B,1,3,3)``."""

    def __init__(self, batch_size: int, steps: int, size: int = 256, seed: int = 0):
        self.batch_size, self.steps, self.size, self.seed = batch_size, steps, size, seed

    def __len__(self):
        return self.steps

    def __iter__(self):
        for s in range(self.steps):
            g = torch.Generator().manual_seed(self.seed * 100003 + s)
            yield {"image": torch.randn(self.batch_size, 2, 3, self.size, self.size, generator=g),
                   "relative_rotation": random_rotations(self.batch_size, generator=g)[:, None]}


class FitResult(list):
    """The per-step losses of ``fit`` (a plain list) plus the training state to continue from."""
    optimizer = None
    scheduler = None
    epoch = 0
    global_step = 0


def save_training_checkpoint(path: str, model, optimizer, scheduler, epoch: int, global_step: int) -> None:
    """Lightning-shaped checkpoint: ``state_dict`` with the reference's prefixes (readable by ``checkpoint`` and by
    the reference's ``load_from_checkpoint``) plus optimizer / scheduler state for ``fit(..., ckpt_path=)``."""
    from . import checkpoint
    checkpoint.save_lightning_style(path, model, extra={
        "epoch": epoch, "global_step": global_step, "optimizer_states": [optimizer.state_dict()],
        "lr_schedulers": [scheduler.state_dict()]})


def fit(cfg, model, loader: Iterable[dict], device=None, max_steps: Optional[int] = None, group=None,
        epochs: int = 1, optimizer=None, scheduler=None, ckpt_path: Optional[str] = None,
        save_path: Optional[str] = None) -> FitResult:
    """Plain-loop counterpart of ``trainer.fit(model, train_dataloader, ckpt_path=...)``
    (modules/model_co3d.py:130-145; ``max_epochs`` = 250 in the reference): ONE AdamW and ONE StepLR for the whole
    run (``configure_optimizers``, or the pair passed in to continue a run); per batch ``training_step`` ->
    ``backward`` -> data-parallel gradient averaging (when a process group exists) -> optimizer step; the
    scheduler steps once per epoch (= one pass over ``loader``).  ``ckpt_path``: resume model, optimizer,
    scheduler and epoch counter from a checkpoint written by ``save_path`` (skipped when the file does not
    exist, like the reference's ``os.path.exists`` guard); ``save_path``: written after every epoch.
    ``max_steps`` bounds the total number of optimizer steps.  Returns the losses (``FitResult``: a list with
    ``.optimizer``, ``.scheduler``, ``.epoch``, ``.global_step``)."""
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if optimizer is None:
        (optimizer,), (sched_new,) = model.configure_optimizers()
        scheduler = sched_new if scheduler is None else scheduler
    elif scheduler is None:
        raise ValueError("pass the scheduler that belongs to the optimizer (or neither)")
    start_epoch, global_step = 0, 0
    if ckpt_path is not None and os.path.exists(ckpt_path):
        from . import checkpoint
        blob = torch.load(ckpt_path, map_location="cpu", weights_only=True)  # tensors and plain containers only
        checkpoint.load_into(model, blob["state_dict"])
        optimizer.load_state_dict(blob["optimizer_states"][0])
        scheduler.load_state_dict(blob["lr_schedulers"][0])
        start_epoch, global_step = int(blob["epoch"]), int(blob["global_step"])
    model.train()
    out = FitResult()
    # The encoder's skinny fp32 GEMMs (M = 64*B rows) run ~25 % faster through rocBLAS than through hipBLASLt's
    # default picks on gfx950 (18.6 vs 24.3 ms per encoder forward+backward at B = 12): prefer it while training.
    prev_blas = None
    if torch.device(device).type == "cuda" and hasattr(torch.backends.cuda, "preferred_blas_library"):
        prev_blas = torch.backends.cuda.preferred_blas_library()
        torch.backends.cuda.preferred_blas_library("cublas")  # = rocBLAS on ROCm
    epoch = start_epoch
    try:
        for epoch in range(start_epoch, start_epoch + epochs):
            left = None if max_steps is None else max_steps - len(out)
            if left is not None and left <= 0:
                break
            out.extend(_fit_loop(model, loader, optimizer, device, left, group))
            scheduler.step()
            if save_path is not None:
                save_training_checkpoint(save_path, model, optimizer, scheduler, epoch + 1, global_step + len(out))
        else:
            epoch = start_epoch + epochs
    finally:
        if prev_blas is not None:
            torch.backends.cuda.preferred_blas_library(prev_blas)
    out.optimizer, out.scheduler, out.epoch, out.global_step = optimizer, scheduler, epoch, global_step + len(out)
    return out


def _fit_loop(model, loader, opt, device, max_steps, group):
    from . import dist as adist
    losses = []
    for step, batch in enumerate(loader):
        if max_steps is not None and step >= max_steps:
            break
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch, step)
        loss.backward()
        adist.all_reduce_gradients(model.parameters(), group=group)
        opt.step()
        losses.append(float(loss.item()))
    return losses


class GraphedTrainStep:
    """One whole ``EstimatorCo3d`` training iteration -- backbone, encoder, InfoNCE through the HIP scorer,
    backward, AdamW -- captured ONCE in a hipGraph and replayed per batch.

    Why: at the reference's batch (12 pairs) the iteration is ~1 850 kernel launches of a few microseconds each;
    issued eagerly the host cannot keep the GPU fed and the step is launch-bound.  The graph needs static shapes
    and buffers, so the batch is copied into fixed tensors, and the per-step hypothesis set (ground truth at index
    0 + fresh Haar samples, modules/model_co3d.py:84-86) is written into a fixed ``(B, num_rota, 3, 3)`` buffer
    just before each replay.  ``random_masking`` draws from torch's graph-safe Philox state, so every replay masks
    differently.  The learning rate is a device scalar (``set_lr``) because a captured AdamW cannot see Python
    floats change.

    The warm-up iterations that must precede a capture (lazy initialisation of libraries and of the optimizer
    state) run on placeholder data; parameters, buffers and the optimizer state (moments, step counts) are put
    back IN PLACE afterwards, so the first replay starts from exactly the model and the fresh AdamW it was given.
    With a process group of more than one rank the bucketed gradient all-reduce (``dist.all_reduce_gradients``)
    is captured between backward and the optimizer step -- RCCL collectives are stream operations; a backend that
    cannot be captured (gloo) is refused instead of silently training diverging replicas."""

    def __init__(self, model, batch_size: int, image_size: int = 256, device=None, warmup: int = 3, group=None):
        from . import dist as adist, ops
        import torch.distributed as tdist
        self.model, self.ops = model, ops
        self.group = group
        self.world = tdist.get_world_size(group) if (tdist.is_available() and tdist.is_initialized()) else 1
        if self.world > 1 and tdist.get_backend(group) != "nccl":
            raise RuntimeError("GraphedTrainStep with %d ranks needs the nccl (RCCL) backend: %s collectives cannot be "
                               "captured in a hipGraph; use harness.fit" % (self.world, tdist.get_backend(group)))
        self._all_reduce = (lambda: adist.all_reduce_gradients(model.parameters(), group=group)) if self.world > 1 \
            else (lambda: None)
        dev = torch.device(device if device is not None else "cuda")
        self.images = torch.zeros(batch_size, 2, 3, image_size, image_size, device=dev)
        self.gt = torch.eye(3, device=dev).repeat(batch_size, 1, 1)
        self.R = torch.eye(3, device=dev).repeat(batch_size, model.num_rota, 1, 1)
        self.lr = torch.tensor(float(model.cfg["TRAIN"]["LR"]), device=dev)
        groups = [{"params": list(model.feature_aligner.parameters()), "lr": self.lr}]
        if isinstance(model.feature_extractor, torch.nn.Module):
            groups.append({"params": list(model.feature_extractor.parameters()), "lr": self.lr})
        self.optimizer = torch.optim.AdamW(groups, eps=1e-5, capturable=True)
        self._draws = 0
        model.train()
        prev_blas = torch.backends.cuda.preferred_blas_library()
        torch.backends.cuda.preferred_blas_library("cublas")  # rocBLAS: see fit()
        try:
            # what the placeholder warm-up must not leave behind
            saved = [t.detach().clone() for t in list(model.parameters()) + list(model.buffers())]
            rng = torch.cuda.get_rng_state(dev)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):  # lazy initialisation (libraries, optimizer state) must not be captured
                    self._fill_rotations()
                    self.optimizer.zero_grad(set_to_none=True)
                    self._loss().backward()
                    self._all_reduce()
                    self.optimizer.step()
                with torch.no_grad():
                    for t, s0 in zip(list(model.parameters()) + list(model.buffers()), saved):
                        t.copy_(s0)
                    for st in self.optimizer.state.values():  # fresh AdamW: zero moments and step counts, in place
                        for v in st.values():
                            if torch.is_tensor(v):
                                v.zero_()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.set_rng_state(rng, dev)
            self._draws = 0
            del saved
            self.graph = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self.loss = self._loss()
                self.loss.backward()
                self._all_reduce()
                self.optimizer.step()
        finally:
            torch.backends.cuda.preferred_blas_library(prev_blas)

    def _loss(self):
        m = self.model
        vol_src, vol_tgt = m.feature_aligner.forward_2d3d(
            m.feature_extraction(self.images[:, 0]), m.feature_extraction(self.images[:, 1]),
            random_mask=m.cfg["TRAIN"]["MASK"], mask_ratio=m.cfg["TRAIN"]["MASK_RATIO"])
        return m.infoNCE_loss(vol_src, vol_tgt, self.R, self.gt, reduce_mean=True)

    def _fill_rotations(self):
        B, n = self.R.shape[0], self.R.shape[1] - 1
        self._draws += 1
        self.R[:, 0].copy_(self.gt)
        self.R[:, 1:].copy_(self.ops.random_rotations(B * n, seed=torch.initial_seed() + self._draws,
                                                      device=self.R.device).reshape(B, n, 3, 3))

    def set_lr(self, lr: float):
        self.lr.fill_(lr)

    def __call__(self, batch: dict) -> torch.Tensor:
        """Runs one iteration on ``batch`` (keys as ``training_step``); returns the loss as a device scalar that the
        next call overwrites (no host synchronisation here)."""
        self.images.copy_(batch["image"], non_blocking=True)
        self.gt.copy_(batch["relative_rotation"].reshape(-1, 3, 3), non_blocking=True)
        self._fill_rotations()
        self.graph.replay()
        return self.loss
