"""TimeT objective, SwavOptimizer and the command line - the reference's ``time_tuning.py`` surface
(``time_tuning.py:80-302``, ``:379-429``, ``:508-717``) on the MI355X kernels.

What is the same: class and method names, argument names and defaults, state_dict keys, the flag set
and its quirks (see ``build_parser``), the numerical result of one training iteration.
What is different by design: ``get_loss`` runs as ONE fused forward+backward launch sequence
(``engine`` / ``_run_step``) instead of autograd over ATen ops; ``loss.backward()`` only hands the
already-computed gradients to the parameters.
"""
from __future__ import annotations

import argparse
import copy
import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine
from . import hip_ops as ops
from .models import DistributedDataParallelModel, FeatureExtractor
from .my_utils import cosine_scheduler

world_size = 1  # module global, as in the reference (time_tuning.py:75,511-512)


class PatchPrototypeSimilarity(nn.Module):
    """Patch-token x prototype scores + Sinkhorn-Knopp assignment: the north-star's name for what the
    reference spreads over ``get_feature_prototype_similarity`` / ``find_optimal_assignment`` / ``get_scores``
    (``time_tuning.py:130-141,157-168,195-217``).  It owns nothing: prototypes, teacher prototypes and the queue
    stay attributes of the ``TimeT`` it is attached to (so the state_dict layout is the reference's).

    ``forward(features[bs,n,dim], use_teacher=False) -> (q[bs,n,K], scores[bs,n,K])``; inference only - the
    training gradient flows through ``TimeT.get_loss``'s fused step."""

    def __init__(self, owner: "TimeT", epsilon: float = 0.05, sinkhorn_iterations: int = 10):
        super().__init__()
        object.__setattr__(self, "_owner", owner)  # not a sub-module: avoids a parameter cycle
        self.epsilon = epsilon
        self.sinkhorn_iterations = sinkhorn_iterations

    def similarity(self, x: torch.Tensor, use_teacher: bool = False) -> torch.Tensor:
        """``normalize(x, dim=-1) @ prototypes.T`` (time_tuning.py:130-141).  With autograd enabled the scores carry grad to
        ``x`` and to the prototypes, as in the reference (``_SimilarityFunction``: HIP forward and backward)."""
        o = self._owner
        protos = o.teacher_prototypes if use_teacher else o.prototypes
        if torch.is_grad_enabled() and (x.requires_grad or protos.requires_grad):
            return _SimilarityFunction.apply(x, protos)
        return engine.prototype_scores(x.detach().contiguous(), protos.detach())

    @torch.no_grad()
    def forward(self, features: torch.Tensor, use_teacher: bool = False, epsilon: Optional[float] = None,
                sinkhorn_iterations: Optional[int] = None):
        o = self._owner
        bs, n, dim = features.shape
        eps = self.epsilon if epsilon is None else epsilon
        iters = self.sinkhorn_iterations if sinkhorn_iterations is None else sinkhorn_iterations
        if engine.exchange_group() is None and features.is_cuda and not ops.fine_grained():
            # one rank: scores of the batch (and of a full queue) + the assignment as ONE call (tt_scores_sinkhorn)
            protos = (o.teacher_prototypes if use_teacher else o.prototypes).detach()
            queue = o.queue if o.queue is not None and o.queue_is_full() else None
            q, scores = ops.scores_sinkhorn(features.detach().reshape(bs * n, dim).contiguous().float(), protos, queue, int(iters), eps, bs * n)
            return q.view(bs, n, -1), scores[: bs * n].view(bs, n, -1)
        batch_scores = self.similarity(features.reshape(bs * n, dim), use_teacher)
        scores = batch_scores
        if o.queue is not None and o.queue_is_full():
            scores = torch.cat([batch_scores, self.similarity(o.queue, use_teacher)], dim=0)
        q = engine.global_sinkhorn(scores, bs * n, eps, int(iters))
        return q.view(bs, n, -1), batch_scores.view(bs, n, -1)


class _SimilarityFunction(torch.autograd.Function):
    """scores = normalize(x) @ prototypes.T with a HIP backward: d_prototypes = d_scores.T @ normalize(x),
    d_x = l2norm_bwd(d_scores @ prototypes)."""

    @staticmethod
    def forward(ctx, x, protos):
        sv: Dict[str, torch.Tensor] = {}
        scores = engine.prototype_scores(x.detach().contiguous(), protos.detach(), sv)
        ctx.save_for_backward(sv["zn"], sv["inv"], protos.detach())
        return scores

    @staticmethod
    def backward(ctx, dscores):
        zn, inv, protos = ctx.saved_tensors
        ds = dscores.contiguous()
        dprotos = dx = None
        if ctx.needs_input_grad[1]:
            dprotos, _ = ops.linear_bwd_weight(ds, zn, need_bias=False)
        if ctx.needs_input_grad[0]:
            dx = ops.l2norm_bwd(ops.linear_bwd_data(ds, protos), zn, inv)
        return dx, dprotos


class _FusedLoss(torch.autograd.Function):
    """Bridges the fused step into autograd: forward runs the whole HIP forward+backward sequence and keeps the
    parameter gradients; backward scales them by the incoming gradient (1 for ``loss.backward()``)."""

    @staticmethod
    def forward(ctx, model, x, hp, need_grad, *params):
        loss, grads = model._run_step(x, hp, need_grad)
        ctx.grads = [grads.get(p) for p in params] if need_grad else None
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        if ctx.grads is None:
            raise RuntimeError("the fused TimeT step was run without gradients")
        out, ctx.grads = ctx.grads, None  # hand the buffers over: with no other owner AccumulateGrad adopts them instead of cloning
        live = [g for g in out if g is not None]
        ops.scale_tensors_(live, gout.reshape(1).to(torch.float32).contiguous())   # chain rule: d loss_out / d loss (1 for loss.backward())
        return (None, None, None, None, *out)


# Launches of ONE captured step graph allowed in flight (0 = unbounded).  Defensive: the cause of round 5's diverging replays is ROCm 7.2's
# replay of AQL packets captured at instantiation, which this package switches off (timetuning_amd/__init__.py) - with the capture ON the
# replays went wrong once the host was more than three launches ahead (bound 1, 2, 3: exact; unbounded: loss 3.69 instead of 4.27454,
# gpurun_out/r06a/bis4_*).  The bound costs nothing (the eager tail of a step - loss copy, AdamW, EMA - is queued behind its replay and
# runs while the host issues the next one) and keeps the host from running an epoch ahead of the device.
STEP_GRAPHS_IN_FLIGHT = int(os.environ.get("TT_STEP_GRAPHS_IN_FLIGHT", "2"))   # (the environment variable: sweeps only)


class _GraphLoss(torch.autograd.Function):
    """``_FusedLoss`` for a step that lives in a captured hipGraph (``TimeT.enable_step_graph``): forward replays the graph - the whole
    launch sequence of forward AND backward in one host call - and hands out the loss; backward hands out the graph's static gradient
    buffers (fresh views: autograd adopts them)."""

    @staticmethod
    def forward(ctx, model, rec, *params):
        # a bounded number of launches in flight (STEP_GRAPHS_IN_FLIGHT).  The wait costs nothing: the eager tail of the previous step (loss
        # copy, AdamW, EMA) is queued behind its replay and runs while the host issues this one.
        evs = rec.setdefault("_events", [])
        if STEP_GRAPHS_IN_FLIGHT > 0 and len(evs) >= STEP_GRAPHS_IN_FLIGHT:
            evs.pop(0).synchronize()
        rec["graph"].replay()
        if STEP_GRAPHS_IN_FLIGHT > 0:
            ev = torch.cuda.Event()
            ev.record()
            evs.append(ev)
        ctx.grads = [rec["grads"].get(p) for p in params]
        ctx.params = params
        return rec["loss"].clone().view(())

    @staticmethod
    def backward(ctx, gout):
        if ctx.grads is None:
            raise RuntimeError("the captured TimeT step hands its gradient buffers out once per forward (no second backward / retain_graph)")
        # The gradients ARE the graph's static buffers (autograd adopts the views; the next replay overwrites them).  With the usual
        # zero_grad() -> backward() -> step() that is what one wants; a .grad that still aliases a buffer HERE was adopted from an earlier
        # step and never reset - gradient accumulation across steps - and this step's replay has already overwritten what it held
        # (ADVICE r5): fail loudly instead of handing the optimizer 2 x the newest gradient.
        for p, g in zip(ctx.params, ctx.grads):
            if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                raise RuntimeError("gradient accumulation across captured TimeT steps: .grad still holds the previous step's graph buffer, which "
                                   "this step's replay overwrote.  Call zero_grad() between steps (SwavOptimizer.step does), or "
                                   "enable_step_graph(False) to accumulate.")
        out, ctx.grads = [None if g is None else g.view(g.shape) for g in ctx.grads], None
        live = [g for g in out if g is not None]
        ops.scale_tensors_(live, gout.reshape(1).to(torch.float32).contiguous())
        return (None, None, *out)


class TimeT(nn.Module):
    """``TimeT(feature_extractor, prototype_number=10, prototype_init=None)`` (``time_tuning.py:80-93``)."""

    def __init__(self, feature_extractor: FeatureExtractor, prototype_number=10, prototype_init=None):
        super().__init__()
        self.feature_extractor = feature_extractor
        self.teacher = None
        self.max_epochs = None
        self.train_iters_per_epoch = None
        self.teacher_prototypes = None
        self.queue = None
        self.momentum_schedule = None
        self.data_parallel = False
        self._queue_rows_pushed = 0
        self._queue_seen = None
        self._ema_flat = None
        self._frame_maps: Dict[tuple, torch.Tensor] = {}
        if prototype_init is None:
            prototype_init = F.normalize(torch.randn((prototype_number, feature_extractor.feature_dim)), dim=-1, p=2)
        self.prototypes = nn.Parameter(prototype_init)
        self.similarity = PatchPrototypeSimilarity(self)
        self.last_aux: Dict[str, torch.Tensor] = {}
        self.register_load_state_dict_post_hook(TimeT._after_load)

    # -- teacher / queue state (time_tuning.py:96-128) ---------------------------------------------
    def init_momentum_teacher(self, teacher=None, prototypes=None):
        if teacher is None:
            self.teacher = copy.deepcopy(self.feature_extractor)
            self.teacher.requires_grad_(False)
            self.teacher_prototypes = nn.Parameter(self.prototypes.detach().clone())
            self.teacher_prototypes.requires_grad_(False)
        else:
            self.teacher = teacher
            self.teacher_prototypes = prototypes
        # The EMA rewrites EVERY teacher tensor through raw pointers (no version bump) unless the teacher still shares the student's
        # frozen tensors - a teacher passed in from outside normally does not: derived copies of teacher tensors (the bf16 planes of
        # the plane-GEMM modes, engine.weight_planes) must never be cached, whichever branch built the teacher.
        for p in self.teacher.parameters():
            p._tt_static = False
            if hasattr(p, "_tt_planes"):
                del p._tt_planes
        self._ema_flat = None

    def init_queue(self, queue_size):
        self.queue = torch.zeros((queue_size, self.feature_extractor.feature_dim), device=self.prototypes.device)
        self._queue_rows_pushed = 0
        self._queue_seen = self._queue_signature()

    def set_queue(self, rows: torch.Tensor) -> None:
        """Replace the queue's contents (a restored checkpoint, a pre-filled benchmark queue): ``rows`` [queue_size, feature_dim].
        Equivalent to ``model.queue.copy_(rows)``, which is detected too (``queue_is_full``); allocates the queue if there is none
        or its size differs."""
        rows = rows.detach()
        if self.queue is None or self.queue.shape != rows.shape:
            self.queue = torch.zeros(tuple(rows.shape), dtype=torch.float32, device=self.prototypes.device)
        self.queue.copy_(rows)
        self._queue_seen = None   # fullness is read from the device at the next use

    def _queue_signature(self):
        return None if self.queue is None else (self.queue.data_ptr(), self.queue._version, tuple(self.queue.shape))

    def _queue_pushed(self, m: int) -> None:
        """Bookkeeping after this module's own ``tt_queue_push`` (a raw-pointer write: no version bump)."""
        if self._queue_rows_pushed is not None:
            self._queue_rows_pushed += m
        self._queue_seen = self._queue_signature() if self._queue_rows_pushed is not None else None

    def queue_is_full(self) -> bool:
        """``self.queue[-1].count_nonzero() != 0`` (time_tuning.py:207).  While only this module writes the queue the answer is
        tracked on the host - the FIFO's last row becomes non-zero exactly when ``queue_size`` rows have been pushed - so the
        training loop pays no device sync.  When somebody else has written the tensor (``model.queue.copy_(...)``, ``set_queue``,
        a loaded checkpoint, a replaced tensor: the storage pointer / version / shape no longer match what this module left
        behind) the reference's own device check runs, once if it says full (a full FIFO stays full), else at every call until
        it does."""
        if self.queue is None:
            return False
        if self._queue_seen is None or self._queue_seen != self._queue_signature():
            full = bool(self.queue[-1].count_nonzero().item() != 0)
            self._queue_rows_pushed = self.queue.shape[0] if full else None   # None: contents unknown, keep asking the device
            self._queue_seen = self._queue_signature() if full else None
            return full
        return self._queue_rows_pushed >= self.queue.shape[0]

    def probe_state(self) -> dict:
        """What a forward + backward of the training step changes besides ``.grad``: the queue (contents and host bookkeeping) and torch's CPU
        generator (the queue permutation, time_tuning.py:259).  ``restore_probe_state`` puts it back - a start-up probe
        (``engine.autotune_exchange``) then leaves the run seed for seed what it would have been."""
        return dict(queue=None if self.queue is None else self.queue.clone(), pushed=self._queue_rows_pushed, known=self._queue_seen is not None,
                    rng=torch.get_rng_state())

    def restore_probe_state(self, st: dict) -> None:
        if st["queue"] is not None:
            self.queue.copy_(st["queue"])
            self._queue_rows_pushed = st["pushed"]
            self._queue_seen = self._queue_signature() if st["known"] else None   # (the copy bumped the version: re-sign what this module knows)
        torch.set_rng_state(st["rng"])

    def set_momentum_teacher_schedular_params(self, momentum_teacher, momentum_teacher_end, max_epochs, train_iter_per_epoch):
        self.momentum_schedule = cosine_scheduler(momentum_teacher, momentum_teacher_end, max_epochs, train_iter_per_epoch)

    def _flatten_for_ema(self):
        """Re-homes student and teacher extractor parameters into two flat buffers (same order: trainable tensors first, frozen
        ones after) so that the EMA is one launch instead of ~150, and decides ONCE whether the teacher's frozen tensors still
        equal the student's bit for bit (``shared``).  They do whenever the teacher is the reference's ``deepcopy`` of the
        student (time_tuning.py:96): a frozen parameter never changes, and its EMA ``t*(1-m) + s*m`` blends two identical
        tensors (:113-114) - mathematically the identity.  With ``shared`` the EMA therefore runs over the trainable tensors
        only (5.7 M of 23.8 M floats at ViT-S/16) and the teacher pass reuses the student's frozen-block activations
        (``_run_step``).  A teacher passed in from outside, or loaded from a checkpoint whose frozen tensors differ, is
        ``shared = False`` and takes the full pass and the full EMA."""
        sp, tp = list(self.feature_extractor.parameters()), list(self.teacher.parameters())
        order = [i for i, p in enumerate(sp) if p.requires_grad] + [i for i, p in enumerate(sp) if not p.requires_grad]
        sig = tuple(p.requires_grad for p in sp)
        ok = self._ema_flat is not None and self._ema_flat["sig"] == sig
        if ok:
            c = self._ema_flat   # still re-homed?  (.to() / .cuda() re-allocate every tensor: the two ends tell)
            ends = ((order[0], c["offs"][0]), (order[-1], c["offs"][-1]))
            ok = all(sp[i].data_ptr() == c["s"].data_ptr() + 4 * o and tp[i].data_ptr() == c["t"].data_ptr() + 4 * o for i, o in ends)
        if ok:
            return self._ema_flat
        offs, total, n_train = [], 0, 0
        for i in order:
            offs.append(total)
            total += (sp[i].numel() + 3) // 4 * 4
            if sp[i].requires_grad:
                n_train = total
        dev = self.prototypes.device
        fs_, ft_ = torch.zeros(total, device=dev), torch.zeros(total, device=dev)
        for plist, flat in ((sp, fs_), (tp, ft_)):
            for i, o in zip(order, offs):
                p = plist[i]
                view = flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        shared = total == n_train or ops.count_mismatch(fs_[n_train:], ft_[n_train:]) == 0
        self._ema_flat = dict(s=fs_, t=ft_, offs=offs, n_train=n_train, shared=shared, sig=sig)
        return self._ema_flat

    def invalidate_teacher_cache(self):
        """Call after changing teacher or student FROZEN tensors by hand (``load_state_dict`` does it by itself, also through a
        wrapping module): the next
        step re-checks whether the teacher may share the student's frozen-block activations."""
        self._ema_flat = None

    @staticmethod
    def _after_load(module, incompatible_keys) -> None:
        """load_state_dict post-hook: runs for a direct ``model.load_state_dict`` AND when the state arrives through a parent
        module (``DistributedDataParallelModel.load_state_dict`` recurses with ``_load_from_state_dict`` and never calls a child's
        ``load_state_dict``): the loaded teacher may differ from the student's frozen tensors, so the sharing decision is retaken."""
        module._ema_flat = None

    def teacher_shares_frozen_blocks(self) -> bool:
        return self.teacher is not None and self.prototypes.is_cuda and self._flatten_for_ema()["shared"]

    def update_momentum_teacher(self, step, writer=None):
        """teacher <- teacher * (1 - m) + student * m with m = momentum_schedule[step] (time_tuning.py:109-118;
        the student gets weight m ~ 0.995 - the reference's convention, kept)."""
        with torch.no_grad():
            momentum = float(self.momentum_schedule[step])
            if writer is not None:
                writer.add_scalar("momentum", momentum, step)
            c = self._flatten_for_ema()
            n = c["n_train"] if c["shared"] else c["s"].numel()
            if n:
                ops.ema_update_(c["t"][:n], c["s"][:n], momentum)
            ops.ema_update_(self.teacher_prototypes.data, self.prototypes.data, momentum)
            ops.normalize_rows_(self.teacher_prototypes.data)

    def normalize_prototypes(self):
        with torch.no_grad():
            ops.normalize_rows_(self.prototypes.data)

    def train_update(self, optimizer: "SwavOptimizer", loss, step: int, writer=None) -> None:
        """The tail of a training iteration (time_tuning.py:659-663) - ``optimizer.step(loss)``, ``normalize_prototypes()`` and,
        with a teacher, ``update_momentum_teacher(step)`` - with the device work of all three enqueued by ONE call
        (tt_adamw_ema_step).  Same results as calling the three reference-named methods one after the other."""
        tail = dict(prototypes=self.prototypes.data)
        if self.teacher is not None:
            momentum = float(self.momentum_schedule[step])
            if writer is not None:
                writer.add_scalar("momentum", momentum, step)
            c = self._flatten_for_ema()
            n = c["n_train"] if c["shared"] else c["s"].numel()
            tail.update(teacher_flat=c["t"][:n] if n else None, student_flat=c["s"][:n] if n else None,
                        teacher_prototypes=self.teacher_prototypes.data, momentum=momentum)
        optimizer.step(loss, tail=tail)
        if ops.pairs():
            ops.poll_pair_range(self.prototypes.device)   # the "f16x3" mode's range contract, without a synchronisation (one step late)

    def check_pair_range(self) -> None:
        """The "f16x3" arithmetic has fp32's precision, not its range: raises ``hip_ops.PairRangeError`` if an operand of a step since the
        last check lay beyond fp16's range (|x| > 65504) or was not finite - where the reference's fp32 arithmetic would have carried on
        and this mode's products are inf / NaN.  Synchronises: the driver calls it where it reads the loss anyway."""
        ops.check_pair_range(self.prototypes.device)

    # -- reference method names, delegating to the HIP path ------------------------------------------
    def get_feature_prototype_similarity(self, x, use_teacher=False):
        return self.similarity.similarity(x, use_teacher)

    def find_optimal_assignment(self, scores, epsilon, sinkhorn_iterations):
        with torch.no_grad():
            return engine.global_sinkhorn(scores.contiguous(), scores.shape[0], epsilon, int(sinkhorn_iterations))

    def get_scores(self, features, epsilon, sinkhorn_iterations, use_teacher=False):
        return self.similarity(features, use_teacher, epsilon, sinkhorn_iterations)

    def make_seg_maps(self, first_frame_segmentation, orig_x, n_last_frames, size_mask_neighborhood, topk, features_exist=False):
        """``time_tuning.py:143-154``: first_frame_segmentation [n, K] (a Sinkhorn assignment used as soft labels), orig_x the
        clip's backbone tokens [fs, n, D] (``features_exist=True``, the training path :294) or its frames [fs, 3, H, W] (the
        reference's default: each frame goes through the extractor without head, mask_propagation.py:462-466).  Returns the STACK
        of all fs - 1 propagated maps [fs-1, K, g, g] fp64 as the reference does (only [-1] is read on the training path)."""
        from .mask_propagation import propagate_labels

        g = self.feature_extractor.spatial_resolution
        scores = first_frame_segmentation.view(g, g, -1).permute(2, 0, 1)
        maps = propagate_labels(n_last_frames, size_mask_neighborhood, topk, self.feature_extractor, orig_x, scores.unsqueeze(0), features_exist)
        return torch.stack(maps)

    def reshape_to_spatial_resolution(self, x, spatial_resolution):
        return x.view(spatial_resolution, spatial_resolution, -1).permute(2, 0, 1)

    def save(self, path):
        torch.save(self.state_dict(), path)

    def forward(self, x, annotations=None, train=False, mask_features=False, use_head=True):
        if not train:
            with torch.no_grad():
                return self.feature_extractor(x, use_head=use_head)
        return self.get_loss(x, annotations=annotations, mask_features=mask_features)

    # -- the objective -----------------------------------------------------------------------------
    def get_loss(self, x, annotations=None, n_last_frames=7, size_mask_neighborhood=6, topk=5, epsilon=0.05,
                 sinkhorn_iterations=10, mask_features=False, queue_perm=None, target_labels=None):
        """``time_tuning.py:224-302``.  Two arguments that are not in the reference pin its two discrete decisions so that
        runs can be reproduced exactly: ``queue_perm`` injects the permutation ``torch.randperm`` draws at :259, and
        ``target_labels`` (int64 [bs, n]) replaces the hard labels ``argmax_K`` of the propagated map (:294-295) in the
        cross entropy - the propagation still runs and ``last_aux["labels"]`` still reports its own result."""
        hp = dict(n_last_frames=n_last_frames, radius=size_mask_neighborhood, topk=topk, epsilon=epsilon,
                  iters=int(sinkhorn_iterations), queue_perm=queue_perm, mask_features=bool(mask_features),
                  target_labels=target_labels)
        params = [p for p in self.parameters() if p.requires_grad]
        need_grad = torch.is_grad_enabled() and len(params) > 0
        if getattr(self, "_step_graph_on", False) and need_grad and x.is_cuda and engine.exchange_group() is None and (
                self._step_graph_max_rows is None
                or x.shape[0] * x.shape[1] * (1 + self.feature_extractor.spatial_resolution ** 2) <= self._step_graph_max_rows):
            out = self._graph_step(x, hp, params)
            if out is not None:
                return out
        return _FusedLoss.apply(self, x, hp, need_grad, *params)

    # -- the step as ONE captured hipGraph (launch-bound regimes: BASELINE C1's 4 frames are ~230 launches of 5 - 15 us each) ---------
    # ``--step_graph auto``: a step is replayed from its captured graph while it is LAUNCH-bound - the host needs ~3.5 ms to issue a step's
    # ~600 launches whatever their size, so below ~10 k token rows (12 clips x 4 frames of ViT-S/16) the replay wins (C1: 2.0 against 3.5 ms)
    # and above it the two are equal to +- 1 % (round 6, one box, ms graph / eager: 8 clips 3.14 / 3.72, 12: 3.93 / 3.96, 16: 4.69 / 4.68,
    # 32 = C2: 7.40 / 7.34, C4 18.24 / 18.22, C5 16.89 / 16.94): large steps stay on the launch-by-launch path, which is also what a
    # multi-GPU rank runs.
    STEP_GRAPH_AUTO_MAX_ROWS = 10000

    def enable_step_graph(self, on: bool = True, max_token_rows: Optional[int] = None) -> None:
        """From the second training step of a given shape on, ``get_loss`` replays a captured hipGraph of the step's whole launch
        sequence (forward + backward, ~230 launches at C1) instead of issuing it launch by launch: the host then costs one replay
        (~15 us) instead of ~10 us per launch.  The first step of a shape runs eagerly (lazily made operands, pinned buffers and the
        K-split workspace exist afterwards), the second one is captured and replayed; a new input shape, a changed set of trainable
        tensors, arithmetic mode, hyper-parameter or queue state (filling -> full) captures again.  What stays outside the graph: the
        optimizer (its learning rate / weight decay / step count are kernel ARGUMENTS that change every step) and the host's own
        bookkeeping - the queue permutation is drawn on the host into the pinned buffer the graph's copy node reads.  Needs ABI 7 (no
        allocation on a launch path).  One process per GPU without an exchange (W = 1); memory: the graph keeps the step's activations."""
        if on:
            from . import GRAPH_FLAG, step_graph_safe
            if not step_graph_safe():
                raise RuntimeError(f"TimeT.enable_step_graph: {GRAPH_FLAG}={os.environ.get(GRAPH_FLAG)!r} - ROCm 7.2 does not replay the captured step reliably "
                                   f"with its AQL packet capture on (timetuning_amd/__init__.py); export {GRAPH_FLAG}=0, or import timetuning_amd before the "
                                   "process's first HIP call (it was imported after torch had initialised the GPU), or run without the step graph")
        self._step_graph_on = bool(on)
        self._step_graph_max_rows = max_token_rows   # None: every step; else only steps of at most this many token rows (``auto``)
        self._step_graphs = {}
        self._step_graph_seen = set()
        self._step_graph_failed = set()

    def _graph_step(self, x, hp, params):
        bs, fs = x.shape[0], x.shape[1]
        n = self.feature_extractor.spatial_resolution ** 2
        # the queue's state as the step will see it AFTER its push (time_tuning.py:207: the scores take the queue rows once it is full)
        full, m = False, 0
        if self.queue is not None:
            if self._queue_rows_pushed is None or self._queue_seen is None or self._queue_seen != self._queue_signature():
                # somebody else wrote the queue (set_queue / queue.copy_ / a checkpoint) and it is not known to be full: the eager step asks
                # the device (a synchronisation no capture can hold) - and keeps asking until the FIFO is full
                if not self.queue_is_full():
                    return None
            m = min(bs * 10, self.queue.shape[0])
            full = self._queue_rows_pushed + m >= self.queue.shape[0]
        labels_in = hp["target_labels"] is not None
        key = (tuple(x.shape), str(x.device), ops.get_gemm_precision(), ops.PAIRS_MIN_ROWS, hp["n_last_frames"], hp["radius"], hp["topk"],
               float(hp["epsilon"]), hp["iters"], hp["mask_features"], self.teacher is not None, None if self.queue is None else self.queue.shape[0],
               full, labels_in, tuple(id(p) for p in params), self.teacher_shares_frozen_blocks() if self.teacher is not None else None,
               # the STATIC frozen tensors' contents: their pair operands are made once and never refreshed by a replay (everything else is
               # converted inside the captured step); an in-place rewrite (load_state_dict, a hand edit) bumps torch's version counter
               sum(p_._version for p_ in self.parameters() if not p_.requires_grad))
        if key in self._step_graph_failed:
            return None
        rec = self._step_graphs.get(key)
        if rec is None and key not in self._step_graph_seen:   # the first step of a shape: eager (it creates what a capture may not create)
            self._step_graph_seen.add(key)
            return None
        # the host's share of the step (what _run_step does outside its launches): the queue permutation travels to the device buffer the
        # captured push reads, target labels (a test's pinned decision) to theirs
        rng = torch.get_rng_state() if rec is None else None
        hp = dict(hp)
        if self.queue is not None:
            hp["_perm_dev"] = self._stage_queue_perm(bs * n, m, hp["queue_perm"], x.device)
        if rec is None:
            if len(self._step_graphs) >= 4:
                self._step_graphs.clear()
            # capture.  The host work of the step runs here, once, as it would in an eager step (queue bookkeeping); the kernels are only
            # recorded - the replay below executes them.  A replay cannot retake a host decision, so: every cached operand derived from a
            # tensor the optimizer / EMA may rewrite is made stale first (the captured step then converts them all, whether or not an
            # update happened since the last step - ADVICE r5: two get_loss calls per update), and a capture that fails (a host
            # synchronisation inside it) puts the host state back and leaves this signature to the eager path for good.
            xs = x.clone()
            lab = torch.as_tensor(hp["target_labels"]).to(device=x.device, dtype=torch.int64).reshape(bs, n).contiguous().clone() if labels_in else None
            hp["target_labels"] = lab
            g = torch.cuda.CUDAGraph()
            keep = (self._queue_rows_pushed, self._queue_seen, self.last_aux)
            ops._bump_param_epoch()
            torch.cuda.synchronize(x.device)
            try:
                with torch.cuda.graph(g):
                    loss, grads = self._run_step(xs, hp, True)
            except Exception as e:   # noqa: BLE001 - whatever broke the capture, the eager path is the answer
                import warnings
                warnings.warn(f"TimeT step graph: capture failed ({type(e).__name__}: {e}); this step signature stays on the eager path")
                self._step_graph_failed.add(key)
                self._queue_rows_pushed, self._queue_seen, self.last_aux = keep
                ops._bump_param_epoch()   # operands 'converted' by launches that never ran are stale
                torch.set_rng_state(rng)
                return None
            rec = self._step_graphs[key] = dict(graph=g, x=xs, loss=loss, grads=grads, aux=self.last_aux, m=m, labels=lab)
        else:
            rec["x"].copy_(x)
            if labels_in:
                rec["labels"].copy_(torch.as_tensor(hp["target_labels"]).reshape(bs, n))
            if self.queue is not None:
                self._queue_pushed(rec["m"])
            self.last_aux = rec["aux"]
        return _GraphLoss.apply(self, rec, *params)

    _PERM_RING = 8

    def _stage_queue_perm(self, total: int, m: int, perm, dev) -> torch.Tensor:
        """The queue's permutation (time_tuning.py:259: ``torch.randperm`` from torch's CPU generator - kept, for seed-for-seed
        reproducibility) on its way to the device: drawn into a slot of a small ring of PINNED buffers, its first ``m`` entries copied
        asynchronously into the one persistent device buffer the push kernel reads (the same address every step: a captured step reads it
        too).  A slot is rewritten only after the copy that last read it has run (its event) - the host may be several steps ahead of the
        device, with graph replays by the whole step."""
        ring = getattr(self, "_queue_perm_ring", None)
        if ring is None or ring["total"] != total or ring["dev"].device != torch.device(dev):
            cuda = torch.device(dev).type == "cuda"
            ring = self._queue_perm_ring = dict(
                total=total, i=0, events=[None] * self._PERM_RING,
                pins=[torch.empty(total, dtype=torch.int64).pin_memory() if cuda else torch.empty(total, dtype=torch.int64) for _ in range(self._PERM_RING)],
                dev=torch.empty(total, dtype=torch.int64, device=dev))
        i = ring["i"]
        ring["i"] = (i + 1) % self._PERM_RING
        if ring["events"][i] is not None:
            ring["events"][i].synchronize()
        pin = ring["pins"][i]
        if perm is None:
            torch.randperm(total, out=pin)
        else:
            pin.copy_(torch.as_tensor(perm).to(torch.int64).reshape(-1)[:total])
        idx = ring["dev"][:m]
        idx.copy_(pin[:m], non_blocking=True)
        if idx.is_cuda:
            ring["events"][i] = torch.cuda.Event()
            ring["events"][i].record()
        return idx

    def _frame_map(self, bs, fs, device, only_t: Optional[int] = None):
        key = (bs, fs, only_t, str(device))
        if key not in self._frame_maps:
            m = torch.arange(bs * fs, dtype=torch.int32).view(bs, fs).t().contiguous()  # [fs, bs]: t-major
            m = m.view(-1) if only_t is None else m[only_t].contiguous()
            self._frame_maps[key] = m.to(device)
        return self._frame_maps[key]

    def _run_step(self, x: torch.Tensor, hp: dict, need_grad: bool):
        try:
            return self._run_step_impl(x, hp, need_grad)
        finally:
            ops.wgrad_join()   # (an exception inside the backward must not leave the side stream armed for the next caller)

    def _run_step_impl(self, x: torch.Tensor, hp: dict, need_grad: bool):
        fe = self.feature_extractor
        vit = fe.backbone
        bs, fs, c, h, w = x.shape
        if fs < 2:
            raise ValueError("TimeT needs clips of at least 2 frames")
        keep = getattr(self, "_debug_tensors", None)   # tools/graph_vs_eager.py: {name: tensor} of the step's intermediates, or None
        dbg = (lambda name, t: keep.__setitem__(name, t)) if keep is not None else (lambda name, t: None)
        Fr = bs * fs
        xf = vit._check(x.reshape(Fr, c, h, w))
        dev = xf.device
        depth = len(vit.blocks)
        all_train_ids = fe.trainable_block_ids()
        train_ids = all_train_ids if need_grad else []
        first = min(train_ids) if train_ids else None
        save = {i: {} for i in range(first, depth)} if first is not None else None
        # EMA teacher that still shares the student's frozen tensors: its blocks [0, t_first) would reproduce the student's
        # activations of frame 0, so they are tapped from the student's pass instead of recomputed
        t_first = (min(all_train_ids) if all_train_ids else depth) if self.teacher_shares_frozen_blocks() else 0
        tap: Optional[dict] = {"block": t_first, "rows": bs} if t_first > 0 else None

        if ops.pairs():   # "f16x3": every pair operand the last optimizer / EMA update made stale, in one launch
            # (rebuilt per step - a walk over ~150 parameters: a teacher attached or a head module swapped after the first step is in the
            # batched refresh at once)
            mods = [self] + ([self.teacher] if getattr(self, "teacher", None) is not None else [])
            seen, lst = set(), []
            for m in mods:
                for p_ in m.parameters():
                    if p_.dim() == 2 and id(p_) not in seen:
                        seen.add(id(p_)); lst.append(p_)
            engine.refresh_pair_operands(lst)

        # ---- student: one pass over all frames, time-major
        use_mask = hp.get("mask_features", False)
        g = fe.spatial_resolution
        s_aux: Optional[dict] = {} if use_mask else None
        # Only the target frames (time-major: the last bs) carry a gradient (time_tuning.py:296-302): the trainable blocks keep
        # their activations for those frames only and run the other frames as a second stream that keeps nothing.
        f0 = (fs - 1) * bs
        # (in the default f32 mode the two streams would run the same kernels on smaller launches: split only when a bf16-plane
        # mode gives the stream that keeps nothing a faster path)
        # The EMA teacher's own blocks + head on frame 0 continue from the tap (the student's frozen-block activations): an independent chain
        # of 6 304-row launches - with two streams it starts on side stream 1 as soon as the tap exists, beside the student's trainable blocks
        teach: dict = {}

        def teacher_chain(tap_):
            tvit_ = self.teacher.backbone
            t_tok_ = engine.vit_blocks(tvit_, tap_["x"], t_first)
            t_feats_ = ops.layernorm_fwd(t_tok_, tvit_.norm.weight, tvit_.norm.bias, drop_first_token=True)
            teach["z_q"] = engine.head_forward(t_feats_, self.teacher.head) if self.teacher.head is not None else t_feats_

        def on_tap(tap_):
            s1 = engine.side_stream(dev, 1)
            s1.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s1):
                teacher_chain(tap_)
            teach["stream"] = s1

        early_teacher = tap is not None and not use_mask and t_first < depth and engine.two_streams(dev, Fr)
        tok, _ = engine.vit_tokens(vit, xf, self._frame_map(bs, fs, dev), save, last_block_aux=s_aux, tap=tap,
                                   save_from_frame=f0 if ops.plane_count_for(f0 * (1 + fe.spatial_resolution ** 2)) else 0,
                                   on_tap=on_tap if early_teacher else None)
        tok_lo, tok_hi = tok if isinstance(tok, tuple) else (tok[:f0], tok[f0:])
        dbg("tok_lo", tok_lo); dbg("tok_hi", tok_hi)
        if keep is not None and save:
            for i_, sv_ in save.items():
                for k_, v_ in sv_.items():
                    if isinstance(v_, torch.Tensor):
                        dbg(f"save{i_}.{k_}", v_)
        N, D = tok_hi.shape[1], tok_hi.shape[2]
        n = N - 1
        # --use_mask (time_tuning.py:244-246 -> models.py:93-144): foreground masks from the last block's cls attention.
        # Only the frames whose head features are consumed need one: the target frames and, without a teacher, frame 0.
        if use_mask:
            qkv_tgt = s_aux["qkv_hi"] if "qkv_hi" in s_aux else s_aux["qkv"][f0:]
            qkv_src = (s_aux["qkv_lo"] if "qkv_lo" in s_aux else s_aux["qkv"])[:bs]
            mask_tgt = ops.foreground_mask(qkv_tgt, vit.num_heads, g).view(-1)
        else:
            mask_tgt = qkv_src = None
        feats = torch.empty((Fr * n, D), dtype=torch.float32, device=dev)
        ops.layernorm_fwd(tok_lo, vit.norm.weight, vit.norm.bias, drop_first_token=True, out=feats[: f0 * n])
        if need_grad:   # statistics of the target frames' rows, for the backward of the final norm
            _, mean_f, rstd_f = ops.layernorm_fwd(tok_hi, vit.norm.weight, vit.norm.bias, save_stats=True, drop_first_token=True,
                                                  out=feats[f0 * n:])
        else:
            ops.layernorm_fwd(tok_hi, vit.norm.weight, vit.norm.bias, drop_first_token=True, out=feats[f0 * n:])
        xn_bb = ops.l2norm_fwd(feats).view(fs, bs, n, D)           # label-propagation features (pre-head tokens)
        src_rows, tgt_rows = feats[: bs * n], feats[(fs - 1) * bs * n:]
        dbg("feats", feats); dbg("xn_bb", xn_bb)

        # ---- target frames: head + scores (with grad).  Independent of the assignment chain below (source head, queue, scores, Sinkhorn,
        # label propagation) until the cross entropy: with two streams (engine.TWO_STREAMS) it runs on the side stream beside that chain -
        # both are 6 272-row launch sequences that leave a quarter of the chip idle on their own
        def target_chain():
            sv_h: Optional[dict] = {} if need_grad else None
            z_t = engine.head_forward(tgt_rows, fe.head, sv_h) if fe.head is not None else tgt_rows
            if use_mask:
                z_t = ops.scale_rows_(z_t if fe.head is not None else z_t.clone(), mask_tgt)  # features * mask (models.py:142)
            sv_s: Optional[dict] = {} if need_grad else None
            return z_t, engine.prototype_scores(z_t, self.prototypes.data, sv_s), sv_h, sv_s   # scores [bs*n, K]

        tgt = None
        lp_sims = lp_ready = None
        side = engine.side_stream(dev) if engine.two_streams(dev, Fr) else None
        if side is not None:
            if fe.head is not None and engine._head_pairs_ok(engine.head_linears(fe.head), tgt_rows.shape[0]):
                for lin_ in engine.head_linears(fe.head):     # lazily made, cached operands: made on THIS stream before the fork
                    engine.weight_planes(lin_.weight, 2)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                # the label propagation's cosine similarities (independent of the assignment: 60 us at C2) first, beside the source head /
                # scores / Sinkhorn of the other stream, which ends in the propagation that reads them
                lp_sims = ops.label_propagate_sims(xn_bb, self.prototypes.shape[0], hp["n_last_frames"]) if engine.LP_SIMS_ON_SIDE else None
                if lp_sims is not None:
                    lp_ready = torch.cuda.Event()
                    lp_ready.record()
                tgt = target_chain()

        # ---- assignment source: teacher on frame 0 if present, else the student's frame 0 (no grad either way)
        if self.teacher is not None:
            tvit = self.teacher.backbone
            t_aux: Optional[dict] = {} if use_mask else None
            if "stream" in teach:   # (launched from inside the student's pass, on side stream 1: see teacher_chain)
                torch.cuda.current_stream().wait_stream(teach["stream"])
                z_q = teach["z_q"]
            else:
                if tap is not None:   # frame 0 = the first bs frames of the time-major pass
                    t_tok = engine.vit_blocks(tvit, tap["x"], t_first, last_block_aux=t_aux)
                    if use_mask and t_first == depth:
                        t_aux["qkv"] = qkv_src
                else:
                    t_tok, _ = engine.vit_tokens(tvit, xf, self._frame_map(bs, fs, dev, only_t=0), last_block_aux=t_aux)
                t_feats = ops.layernorm_fwd(t_tok, tvit.norm.weight, tvit.norm.bias, drop_first_token=True)
                z_q = engine.head_forward(t_feats, self.teacher.head) if self.teacher.head is not None else t_feats
            protos_q = self.teacher_prototypes.data
            mask_q = ops.foreground_mask(t_aux["qkv"], tvit.num_heads, g).view(-1) if use_mask else None  # time_tuning.py:235-236
        else:
            z_q = engine.head_forward(src_rows, fe.head) if fe.head is not None else src_rows
            protos_q = self.prototypes.data
            mask_q = ops.foreground_mask(qkv_src, vit.num_heads, g).view(-1) if use_mask else None
            if use_mask and fe.head is None:
                z_q = z_q.clone()  # src_rows is a view of the propagation features, which stay unmasked (time_tuning.py:285)
        if use_mask:
            ops.scale_rows_(z_q, mask_q)

        if self.queue is not None:  # time_tuning.py:250-261 (before scoring, so the batch is also in the queue)
            m = min(bs * 10, self.queue.shape[0])
            # (a captured step: the host staged the permutation before the capture / each replay, _graph_step)
            idx = hp.get("_perm_dev")
            if idx is None:
                idx = self._stage_queue_perm(bs * n, m, hp["queue_perm"], dev)
            foreign = self._queue_seen is None or self._queue_seen != self._queue_signature()
            if foreign:
                self.queue_is_full()   # somebody else wrote the queue: read its state from the device before this push hides it
            ops.queue_push_(self.queue, z_q, idx)
            self._queue_pushed(m)

        if self.queue_is_full():   # time_tuning.py:207-211: batch rows, then the queue rows - both products write ONE score matrix
            scores_q = torch.empty((z_q.shape[0] + self.queue.shape[0], protos_q.shape[0]), dtype=torch.float32, device=dev)
            engine.prototype_scores(z_q, protos_q, out=scores_q[: z_q.shape[0]])
            engine.prototype_scores(self.queue, protos_q, out=scores_q[z_q.shape[0]:])
        else:
            scores_q = engine.prototype_scores(z_q, protos_q)
        dbg("z_q", z_q); dbg("scores_q", scores_q)
        gather = engine.global_sinkhorn_begin(scores_q)  # W > 1: the score rows travel while the target head runs

        if side is None:   # one stream: the target chain here, between the two halves of the exchange (W > 1: the score rows travel under it)
            tgt = target_chain()
        K = self.prototypes.shape[0]
        q = engine.global_sinkhorn_end(gather, bs * n, hp["epsilon"], hp["iters"])              # [bs*n, K]
        if lp_ready is not None:
            torch.cuda.current_stream().wait_event(lp_ready)
        labels = ops.label_propagate(xn_bb, q.view(bs, n, K), hp["n_last_frames"], hp["radius"], hp["topk"], 0.1, sims=lp_sims)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        z_tgt, scores_t, sv_head, sv_sc = tgt
        ce_labels = labels
        if hp.get("target_labels") is not None:
            ce_labels = torch.as_tensor(hp["target_labels"]).to(device=dev, dtype=torch.int64).reshape(bs, n).contiguous()
        loss, dscores = ops.ce_loss_fwd_bwd(scores_t, ce_labels.view(-1), 0.1, need_grad, row_weight=mask_tgt)  # :296-300
        dbg("z_tgt", z_tgt); dbg("scores_t", scores_t); dbg("q", q); dbg("labels", labels); dbg("loss", loss); dbg("dscores", dscores)
        self.last_aux = dict(q=q.view(bs, n, K), target_scores=scores_t.view(bs, n, K), labels=labels)
        if use_mask:
            self.last_aux.update(target_mask=mask_tgt.view(bs, n), source_mask=mask_q.view(bs, n))
        if not need_grad:
            return loss, {}

        # ---- backward on the target frames only
        grads: Dict[nn.Parameter, torch.Tensor] = {}
        # the data-parallel exchange: bucketed all-reduce (mean) over RCCL, overlapped with backward; from the second step on the backward
        # kernels write their dw / db straight into its persistent flat buckets (``exchange.out``)
        if getattr(self, "_grad_arena", None) is None:
            self._grad_arena = engine.GradArena()
        exchange = engine.GradExchange(self._grad_arena)
        out = exchange.out
        prescaled = exchange.prescale_(dscores)   # 1 / W once, on the 5 MB loss gradient, instead of on every bucket
        if engine.two_streams(dev, bs * fs) and ops.pairs():
            ops.wgrad_fork(engine.side_stream(dev, 3))   # the weight-gradient products beside the data-gradient chain (joined below)
        zn_t = sv_sc["zn"]
        grads[self.prototypes], _ = ops.wgrad_call(lambda: ops.linear_bwd_weight(dscores, zn_t, need_bias=False, dw_out=out(self.prototypes)),
                                                   keep=(dscores, zn_t))   # (a leaf too: beside the head's backward)
        # "f16x3": every dy of a Linear is scaled by a power of two before its pair split; the kernel that PRODUCES the dy publishes its
        # max |.| into a slot of this pool (one 64-float fill per step instead of a max pass per dy: 12 launches)
        pool = ops.AmaxPool.get(dev) if (ops.pairs() and ops.GRAD_SCALE and ops.AMAX_FROM_PRODUCERS) else None
        if pool is not None:
            pool.reset()
        take = pool.take if pool is not None else (lambda: None)
        a_dz = take()
        dz = ops.l2norm_bwd(ops.linear_bwd_data(dscores, self.prototypes.data), sv_sc["zn"], sv_sc["inv"], amax_out=a_dz)
        if use_mask:
            ops.scale_rows_(dz, mask_tgt)  # backward of features * mask (rows times a number in [0, 1]: a_dz stays an upper bound)
        d_feats = engine.head_backward(dz, fe.head, sv_head, grads, out=out, dz_amax=a_dz, amax_pool=pool) if fe.head is not None else dz
        dbg("dz", dz); dbg("d_feats", d_feats)
        exchange.push(grads)  # prototypes + head
        #  the final norm's backward also runs when ONLY that norm is trainable, and its parameter gradients are built when
        # either of them asks for one
        wg = vit.norm.weight.requires_grad or vit.norm.bias.requires_grad
        if train_ids or wg:
            a_tok = take()
            dx, dg, db = ops.layernorm_bwd(d_feats, tok_hi, vit.norm.weight, mean_f, rstd_f, need_wgrad=wg, drop_first_token=True,
                                           dg_out=out(vit.norm.weight) if wg else None, db_out=out(vit.norm.bias) if wg else None, amax_out=a_tok)
            if wg:
                grads[vit.norm.weight], grads[vit.norm.bias] = dg, db
        if train_ids:
            dx = dx.view(bs * N, D)
            # kept activations: the target frames only (two-stream pass: rows [0, bs)) or all frames (rows [f0, Fr))
            kept = save[first]["x_in"].shape[0]
            b0, b1 = (0, bs) if kept == bs else (f0, Fr)
            for i in range(len(vit.blocks) - 1, first - 1, -1):
                # the LAST block of the backward has nothing after it to hide its bucket behind: its MLP gradients (two thirds of
                # the block) leave as soon as they exist, so only the attention third is exposed
                dx = engine.block_backward(dx, vit.blocks[i], vit.num_heads, save[i], b0, b1, grads, need_dx=i > first,
                                           after_mlp=(lambda: exchange.push(grads)) if i == first else None, out=out,
                                           dx_out_amax=a_tok, amax_pool=pool)
                if pool is not None:
                    dx, a_tok = dx
                if dx is not None:
                    dbg(f"dx_in{i}", dx)
                if i > first:
                    exchange.push(grads)  # this block's gradients travel while the next block's backward runs
        grads = {p: g for p, g in grads.items() if p.requires_grad}
        ops.wgrad_join()
        return loss, exchange.finish(grads, scale=not prescaled)


# ------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------

class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (betas (0.9, 0.999), eps 1e-8, decoupled decay), one HIP launch per step for
    all tensors.  State layout matches torch's (``step``, ``exp_avg``, ``exp_avg_sq``), so optimizer state dicts
    are interchangeable with the reference's."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, tail: Optional[dict] = None):
        """``tail`` (TimeT.train_update): what follows the parameter update in a training iteration - prototype
        renormalisation and the EMA teacher - enqueued by the same C call as the update (tt_adamw_ema_step)."""
        by_step: Dict[int, list] = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (int(st["step"].item()), b1, b2, group["eps"])
                by_step.setdefault(key, []).append((p.data, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], group["weight_decay"]))
        groups = list(by_step.items())
        for i, ((step, b1, b2, eps), entries) in enumerate(groups):
            if tail is not None and i == len(groups) - 1:
                ops.adamw_ema_step_(entries, step, b1, b2, eps, **tail)
            else:
                ops.adamw_step_(entries, step, b1, b2, eps)
        if tail is not None and not groups:
            ops.adamw_ema_step_([], 1, **tail)
        return None


class SwavOptimizer:
    """``time_tuning.py:379-429``: AdamW over (prototypes | head | backbone) x (decayed | bias-and-1-D) groups,
    CosineAnnealingLR, per-step weight-decay reschedule.  ``lr_scheduler == "CosineAnnealingLR"`` is compared by
    value (the reference uses ``is``, which is True only for the in-file default, :383)."""

    def __init__(self, model, optimizer, use_projection_head, backbone_lr, lr, lr_scheduler, wd_schedule, num_itr=None,
                 num_epochs=None, exclude_bias_norm=True, writer=None):
        self.optimizer = self.configure_optimizer(model, optimizer, use_projection_head, backbone_lr, lr, wd_schedule[0],
                                                  exclude_bias_norm=exclude_bias_norm)
        if lr_scheduler == "CosineAnnealingLR":
            self.lr_scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=num_itr * num_epochs, eta_min=0)
        else:
            self.lr_scheduler = None
        self.wd_schedule = wd_schedule
        self.writer = writer
        self.global_step = 0

    def get_optimization_dict(self, model, filter_name, exclude_decay=True, weight_decay=0.0001, learning_rate=0.001):
        params, excluded = [], []
        for name, param in model.named_parameters():
            if param.requires_grad and (filter_name in name):
                if exclude_decay and (name.endswith(".bias") or param.dim() == 1):
                    excluded.append(param)
                else:
                    params.append(param)
        return [{"params": params, "weight_decay": weight_decay, "lr": learning_rate},
                {"params": excluded, "weight_decay": 0.0, "lr": learning_rate}]

    def configure_optimizer(self, model, optimizer, use_projection_head, backbone_lr, lr, weight_decay, exclude_bias_norm=True):
        if optimizer != "AdamW":
            raise ValueError("only AdamW is built (the reference leaves `opt` undefined for anything else, :413-415)")
        target = model.get_non_ddp_model() if isinstance(model, DistributedDataParallelModel) else model
        groups = self.get_optimization_dict(target, "prototypes", exclude_bias_norm, weight_decay, lr)
        if use_projection_head:
            groups += self.get_optimization_dict(target, "feature_extractor.head", exclude_bias_norm, weight_decay, lr)
        groups += self.get_optimization_dict(target, "feature_extractor.backbone", exclude_bias_norm, weight_decay, backbone_lr)
        return FusedAdamW(groups, lr)

    def state_dict(self):
        return self.optimizer.state_dict(), self.global_step

    def step(self, loss, tail: Optional[dict] = None):
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step(tail=tail) if tail is not None else self.optimizer.step()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        self.global_step += 1
        for param_group in self.optimizer.param_groups:
            if param_group["weight_decay"] != 0:
                # the reference indexes wd_schedule[global_step] and raises IndexError on the very last step
                # (:427-429); the schedule's last value is held instead
                param_group["weight_decay"] = float(self.wd_schedule[min(self.global_step, len(self.wd_schedule) - 1)])


# ------------------------------------------------------------------------------------------------
# checkpoints (time_tuning.py:460-505)
# ------------------------------------------------------------------------------------------------

def save_checkpoint(model, optimizer: SwavOptimizer, epch_num: int, filename: str) -> None:
    opt_state, global_step = optimizer.state_dict()
    torch.save({"epoch": epch_num, "global_step": global_step, "model": model.state_dict(), "optimizer": opt_state,
                "scheduler": optimizer.lr_scheduler.state_dict() if optimizer.lr_scheduler is not None else None}, filename)


def load_checkpoint(model, swav_optimizer: SwavOptimizer, filename: str) -> int:
    if not os.path.isfile(filename):
        print(f"No checkpoint found at {filename}")
        return 0
    # (the optimizer / scheduler state holds plain Python and NumPy scalars: the reference's torch.load predates weights_only)
    state = torch.load(filename, map_location="cpu", weights_only=False)
    model.load_state_dict(state["model"])
    inner = model.get_non_ddp_model() if hasattr(model, "get_non_ddp_model") else model
    if hasattr(inner, "invalidate_teacher_cache"):
        inner.invalidate_teacher_cache()   # (the post-hook has done it already; kept explicit for wrappers that bypass hooks)
    swav_optimizer.optimizer.load_state_dict(state["optimizer"])
    swav_optimizer.global_step = state["global_step"]
    if swav_optimizer.lr_scheduler is not None and state.get("scheduler") is not None:
        swav_optimizer.lr_scheduler.load_state_dict(state["scheduler"])
    return state["epoch"]


# ------------------------------------------------------------------------------------------------
# command line + driver (time_tuning.py:508-717)
# ------------------------------------------------------------------------------------------------

def build_parser() -> argparse.ArgumentParser:
    """Flag names, types and defaults of ``time_tuning.py:674-713``.

    Reproduced quirks: the ``type=bool`` flags treat ANY non-empty string as True (``--use_queue False`` turns the
    queue ON, as in the reference's README command); ``--epsilon --sinkhorn_iterations --n_last_frames --topk
    --size_mask_neighborhood --epochs --dataset_path --destination_path`` are parsed but never reach ``get_loss``,
    which uses its signature defaults (eps 0.05, 10 iterations, 7 frames, radius 6, top-5).
    Added (not in the reference): ``--dataset synthetic`` / ``--steps_per_epoch`` / ``--eval_clips`` because the dataset
    loaders are out of scope here, ``--eval_every`` (the reference hard-codes 4) and ``--precision`` (the arithmetic mode of the
    matrix products; the library-level default of ``hip_ops.set_gemm_precision`` stays "f32", the training driver's is "f16x3")."""
    p = argparse.ArgumentParser()
    p.add_argument("--architecture", type=str, default="dino-s16")
    p.add_argument("--model_path", type=str, default="vits16_800ep.pth.tar")
    p.add_argument("--dataset", type=str, default="ytvos")
    p.add_argument("--dataset_path", type=str, default="../data")
    p.add_argument("--destination_path", type=str, default="ytvos")
    p.add_argument("--evaluation_protocol", type=str, default="dataset-wise")
    p.add_argument("--visualization_directory", type=str, default="visualizations")
    p.add_argument("--logging_directory", type=str, default="logs")
    p.add_argument("--EMA_decay", type=float, default=0.995)
    p.add_argument("--lr_scheduler", type=str, default="CosineAnnealingLR")
    p.add_argument("--head_lr", type=float, default=0.0001)
    p.add_argument("--batch_size", type=int, default=128)
    p.add_argument("--num_epochs", type=int, default=100)
    p.add_argument("--num_workers", type=int, default=10)
    p.add_argument("--num_clusters", type=int, default=200)
    p.add_argument("--input_resolution", type=int, default=224)
    p.add_argument("--many_to_one", type=bool, default=False)
    p.add_argument("--precision_based", type=bool, default=False)
    p.add_argument("--num_frames", type=int, default=4)
    p.add_argument("--n_last_frames", type=int, default=6)
    p.add_argument("--uvos", type=int, default=False)
    p.add_argument("--topk", type=int, default=5)
    p.add_argument("--size_mask_neighborhood", default=6, type=int)
    p.add_argument("--epsilon", default=0.05, type=float)
    p.add_argument("--sinkhorn_iterations", default=3, type=float)
    p.add_argument("--use_projection_head", type=bool, default=True)
    p.add_argument("--use_queue", type=bool, default=False)
    p.add_argument("--queue_size", type=int, default=16384)
    p.add_argument("--use_mask", type=bool, default=False)
    p.add_argument("--use_teacher", type=bool, default=True)
    p.add_argument("--load_checkpoint", type=bool, default=False)
    p.add_argument("--regular_step", type=int, default=3)
    p.add_argument("-n", "--nodes", default=1, type=int, metavar="N")
    p.add_argument("-g", "--gpus", default=1, type=int)
    p.add_argument("-nr", "--nr", default=0, type=int)
    p.add_argument("--epochs", default=3000, type=int, metavar="N")
    p.add_argument("--steps_per_epoch", default=8, type=int, help="synthetic data only")
    p.add_argument("--eval_every", default=4, type=int, help="epochs between rank-0 evaluations (the reference hard-codes 4; 0 = never)")
    p.add_argument("--eval_clips", default=8, type=int, help="synthetic evaluation set size")
    p.add_argument("--precision", default="f16x3", choices=["f16x3", "f32", "bf16x6", "bf16x3", "bf16"],
                   help="arithmetic of the matrix products (hip_ops.set_gemm_precision): f16x3 = the fp32-accurate fp16-pair split (per-op error "
                        "under the exact-f32 MFMA kernels', 1.7x their speed), f32 = exact fp32 MFMA, bf16 = BASELINE C4's bf16 path")
    p.add_argument("--sinkhorn_exchange", default="auto", choices=["auto", "allgather", "allreduce"],
                   help="W > 1: allgather = one all-gather of the score rows, every rank solves the global problem; allreduce = the "
                        "reference's own pattern (my_utils.py:250-272): columns stay on their rank, the K row sums are all-reduced per iteration; "
                        "auto (default) = both - and the 4-bucket against the 1-bucket gradient exchange - are timed on the first batch and the "
                        "faster is kept (engine.autotune_exchange)")
    p.add_argument("--step_graph", default="auto", choices=["auto", "on", "off"],
                   help="replay the training step's launch sequence as ONE captured HIP graph from its second occurrence on "
                        "(TimeT.enable_step_graph: the host then issues one replay instead of ~250 - 600 launches; C1 -22 %, C2 -2.4 %); "
                        "auto (default) = on with one process per node (no exchange inside the step), off otherwise")
    return p


class SyntheticClips:
    """Stand-in for ``make_loader`` (data_loader.py:1047-1110): yields ``(data[bs,1,fs,3,H,W], annotations, label)``
    like the real loader (data_loader.py:743-767), from the portable generator, already on the device."""

    def __init__(self, batch_size, num_frames, resolution, steps, device, rank=0):
        from . import synth

        self.steps, self.bs, self.fs, self.res, self.device, self.rank = steps, batch_size, num_frames, resolution, device, rank
        self._synth = synth

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            clips = self._synth.make_clips(self.bs, self.fs, self.res, seed=1000 * self.rank + i + 1)
            data = torch.from_numpy(clips).unsqueeze(1).to(self.device, non_blocking=True)
            yield data, torch.zeros(self.bs, 1, self.fs, 1), torch.zeros(self.bs)


class SyntheticFrameClips(SyntheticClips):
    """As ``SyntheticClips`` but starting from raw uint8 frames ``[fs, H, W, 3]`` that go through the GPU input pipeline
    (``video_transformations.training_transforms``: colour jitter / grayscale / blur, Resize, RandomResizedCrop, ClipToTensor) -
    the work the reference's DataLoader workers do with Pillow on the host (``time_tuning.py:588-593``)."""

    def __init__(self, batch_size, num_frames, resolution, steps, device, rank=0, raw_size=(360, 480)):
        super().__init__(batch_size, num_frames, resolution, steps, device, rank)
        from . import video_transformations as VT

        self.raw_size = raw_size
        self.frame_transform, self.video_transform = VT.training_transforms(resolution)

    def __iter__(self):
        H, W = self.raw_size
        for i in range(self.steps):
            clips = []
            for b in range(self.bs):
                z = self._synth.normal(f"raw.{self.rank}.{i}.{b}", (self.fs, H // 8, W // 8, 3), 60.0, 127.0)
                raw = torch.from_numpy(np.clip(np.kron(z, np.ones((1, 8, 8, 1), np.float32)), 0, 255).astype(np.uint8)).to(self.device)
                clips.append(self.video_transform(self.frame_transform(raw)))
            yield torch.stack(clips).unsqueeze(1), torch.zeros(self.bs, 1, self.fs, 1), torch.zeros(self.bs)


class SyntheticEvalClips:
    """Stand-in for the reference's evaluation loader (``pascal_loader(60, ..., "val", 112, train_size=224)``, time_tuning.py:596):
    a fixed set of synthetic frames with planted integer masks, batches ``(data [bs,1,fs,3,R,R], annotations [bs,1,fs,R,R])``."""

    def __init__(self, num_clips, num_frames, resolution, batch_size, device):
        from . import mask_propagation as MP

        self.batches = []
        for b0 in range(0, num_clips, batch_size):
            clips = [MP.synthetic_tracking_clip(num_frames, resolution, seed=900 + i) for i in range(b0, min(num_clips, b0 + batch_size))]
            data = torch.stack([c for c, _ in clips]).unsqueeze(1).to(device)
            ann = torch.stack([m for _, m in clips]).unsqueeze(1).to(device)
            self.batches.append((data, ann))

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def seed_everything(seed: int = 1) -> None:
    """The reference seeds python / numpy / torch with 1 at import time (time_tuning.py:66-69): prototype initialisation and
    the queue permutations (``torch.randperm``, :259) are reproducible from run to run."""
    import random

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def time_tuning(gpu=0, args=None):
    """One process per GPU (``time_tuning.py:508-666``): model, optimiser, epoch loop with the rank-0 evaluation every
    ``--eval_every`` (reference: 4) epochs and the barrier behind it (:634-648).  Datasets are synthetic (the readers are out of
    scope): training clips from the portable generator, evaluation frames with planted masks."""
    import torch.distributed as dist

    from .evaluation import Evaluator

    global world_size
    seed_everything(1)
    world_size = args.gpus * args.nodes
    rank = args.nr * args.gpus + gpu
    # test aids (not reference flags): TT_SHARE_DEVICE=1 puts every rank on cuda:0 of a 1-GPU box, TT_DIST_BACKEND=gloo carries
    # the exchange there (RCCL refuses two ranks on one device); the default is one GPU per rank over RCCL ("nccl")
    if os.environ.get("TT_SHARE_DEVICE"):
        gpu = 0
    device = torch.device("cuda", gpu)
    torch.cuda.set_device(device)
    ops.set_gemm_precision(getattr(args, "precision", "f16x3"))   # the driver's default: the fp32-accurate fp16-pair split
    sk_exchange = getattr(args, "sinkhorn_exchange", "auto")
    engine.SINKHORN_EXCHANGE = "allgather" if sk_exchange == "auto" else sk_exchange
    if world_size > 1 and not dist.is_initialized():
        dist.init_process_group(backend=os.environ.get("TT_DIST_BACKEND", "nccl"), init_method="env://", world_size=world_size, rank=rank)
    if args.use_projection_head:
        fe = FeatureExtractor(args.architecture, args.model_path, [1024, 1024, 512, 256], unfreeze_layers=["blocks.11", "blocks.10"],
                              return_attention=False)
    else:
        fe = FeatureExtractor(args.architecture, args.model_path, return_attention=False)
    model = TimeT(fe, args.num_clusters).to(device)
    if world_size > 1:
        model = DistributedDataParallelModel(model, gpu)
    elif getattr(args, "step_graph", "auto") in ("auto", "on"):
        from . import step_graph_safe
        if args.step_graph == "on" or step_graph_safe():     # ("on" raises where the runtime's packet capture could not be switched off)
            model.enable_step_graph(max_token_rows=TimeT.STEP_GRAPH_AUTO_MAX_ROWS if args.step_graph == "auto" else None)
        elif rank == 0:
            print("step graph: off (the runtime's hipGraph packet capture could not be switched off in this process: timetuning_amd/__init__.py)")
    if args.dataset == "synthetic":
        loader = SyntheticClips(args.batch_size, args.num_frames, args.input_resolution, args.steps_per_epoch, device, rank)
    elif args.dataset == "synthetic_frames":  # raw uint8 frames through the GPU input pipeline
        loader = SyntheticFrameClips(args.batch_size, args.num_frames, args.input_resolution, args.steps_per_epoch, device, rank)
    else:
        raise NotImplementedError("the dataset readers (data_loader.py: directory scanning, JPEG decoding) are out of scope for this "
                                  "build; run with --dataset synthetic or --dataset synthetic_frames")
    num_itr = len(loader)
    opt = SwavOptimizer(model, "AdamW", args.use_projection_head, args.head_lr / 10, args.head_lr, args.lr_scheduler,
                        cosine_scheduler(0.04, 0.4, args.num_epochs, num_itr), num_itr, args.num_epochs)
    if args.use_teacher:
        model.init_momentum_teacher()
        model.set_momentum_teacher_schedular_params(args.EMA_decay, 1.0, args.num_epochs, num_itr)
    if args.use_queue:
        model.init_queue(args.queue_size // world_size)
    os.makedirs(args.logging_directory, exist_ok=True)
    if args.load_checkpoint:
        load_checkpoint(model, opt, os.path.join(args.logging_directory, "checkpoint.pth"))
    # evaluation (time_tuning.py:596-604): rank 0 only, on the bare model, k-means over backbone features, matched mIoU
    eval_model = model.get_non_ddp_model() if isinstance(model, DistributedDataParallelModel) else model
    evaluator = None
    if args.eval_every > 0 and rank == 0:
        eval_loader = SyntheticEvalClips(args.eval_clips, 1, args.input_resolution, 4, device)
        evaluator = Evaluator(eval_model, eval_loader, 21, device, clustering_algorithm="k-means", uvos_flag=bool(args.uvos))
    eval_resolution = args.input_resolution // 2 if args.evaluation_protocol == "dataset-wise" else args.input_resolution  # :603
    previous_score, scores = 0.0, []
    last = num_itr * args.num_epochs - 1
    for epoch in range(args.num_epochs):
        if rank == 0:
            save_checkpoint(model, opt, epoch, os.path.join(args.logging_directory, "checkpoint.pth"))
        if evaluator is not None and epoch % args.eval_every == 0:        # :634-646
            with torch.no_grad():
                eval_model.eval()
                eval_score = float(evaluator.evaluate(many_to_one=args.many_to_one, evaluation_protocol=args.evaluation_protocol,
                                                      eval_resolution=eval_resolution, num_clusters=21, use_annotations=False,
                                                      use_mask=False, precision_based=args.precision_based))
                if eval_score > previous_score:
                    previous_score = eval_score
                    eval_model.save(os.path.join(args.logging_directory, f"{previous_score}_{epoch}.pth"))
                scores.append((epoch, eval_score))
                print("Epoch: {} Scores/localization {:.4f}".format(epoch, eval_score))
            eval_model.train()
        if world_size > 1:
            dist.barrier()                                                  # :647-648: the other ranks wait for rank 0
        model.train()
        for i, (data, annotations, label) in enumerate(loader):
            data = data.squeeze(1)
            if world_size > 1 and sk_exchange == "auto" and engine.EXCHANGE_CHOICE is None:
                # the first batch decides how this communicator exchanges (forward + backward only: no parameter or teacher update; the queue,
                # its bookkeeping and the host generator are put back before every timed variant and afterwards - TimeT.probe_state)
                def _probe_step():
                    model.zero_grad(set_to_none=True)
                    model(data, annotations, True, args.use_mask).backward()

                engine.autotune_exchange(_probe_step, device, log=print, state=(eval_model.probe_state, eval_model.restore_probe_state))
                model.zero_grad(set_to_none=True)
            loss = model(data, annotations, True, args.use_mask)
            # optimizer.step(loss); model.normalize_prototypes(); model.update_momentum_teacher(global_step) (:659-663) as one call
            model.train_update(opt, loss, min(opt.global_step + 1, last))
            if rank == 0:
                print("Iteration: {}/{} loss {:.4f}".format(i, num_itr, loss.item()))
                model.check_pair_range()   # the "f16x3" mode's range contract (the host has just synchronised for the loss)
    model.eval_scores = scores
    return model


def main(argv=None):
    import torch.multiprocessing as mp

    args = build_parser().parse_args(argv)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if args.gpus == 1:
        time_tuning(0, args)
    else:
        mp.spawn(time_tuning, nprocs=args.gpus, args=(args,))


if __name__ == "__main__":
    main()
