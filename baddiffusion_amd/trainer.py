"""Fused BadDiffusion train step (e): the loop body of /root/reference/baddiffusion.py:590-615
as one launch sequence on the GPU, data-parallel over one process per GPU.

    poison-blend + q_sample  ->  UNet forward  ->  MSE + dL/dpred  ->  UNet backward (segment by segment,
    each finished gradient range all-reduced over RCCL while the next segment computes)  ->
    global-norm clip + Adam on the flat parameter buffer.

No host synchronisation inside a step (the loss stays a device scalar; call .item() when you log).
Replaces: torch.optim.Adam (:320), clip_grad_norm_ (:612), get_cosine_schedule_with_warmup
(diffusers/optimization.py:109-140), nn.DataParallel (:325) -> RCCL all-reduce of the flat gradient.
"""
import ctypes
import math
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops


def cosine_schedule_with_warmup(step, num_warmup_steps, num_training_steps, num_cycles=0.5):
    """LR multiplier of diffusers/optimization.py:134-138."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


class DPReducer:
    """Sum-all-reduce of finished ranges of one flat gradient buffer, issued asynchronously as backward produces them.  The
    loss gradient is pre-scaled by 1/world, so the sum is the global-batch mean gradient.  Two transports:
      * `rccl` (an rccl.RcclComm; the product path on GPUs): ncclAllReduce enqueued directly on the caller's communication
        stream -- ordering is plain HIP stream order, nothing else runs per collective;
      * torch.distributed (c10d) on the current stream: gloo in the CPU / shared-GPU tests, or BD_DP_TRANSPORT=c10d for A/B."""

    def __init__(self, flat, process_group=None, force=False, shadow=None, rccl=None, stream=None):
        self.flat, self.pg = flat, process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rccl, self.stream = rccl, stream
        self.active = self.world > 1 or (force and (rccl is not None or (dist.is_available() and dist.is_initialized())))
        self.shadow = shadow          # ordering check (TrainEngine(dp_check=True)): what the collective's stream saw, see there
        self.works = []
        self.bytes = 0

    def reduce_ranges(self, ranges):
        """all ranges a backward segment has finalised; call with the communication stream current (its waits already issued)"""
        if not self.active:
            return
        ranges = [(lo, hi) for lo, hi in ranges if hi > lo]
        if self.rccl == "none":      # BD_DP_TRANSPORT=none (measurement only): the whole path without the collective call itself
            self.bytes += sum(hi - lo for lo, hi in ranges) * 4
        elif self.rccl is not None:
            self.bytes += self.rccl.all_reduce_ranges_(self.flat, ranges, self.stream or torch.cuda.current_stream())
        else:
            for lo, hi in ranges:
                self.works.append(dist.all_reduce(self.flat[lo:hi], group=self.pg, async_op=True))
                self.bytes += 4 * (hi - lo)
        if self.shadow is not None:
            # snapshot the ranges on the stream that ISSUED the collective, right behind it: it reads the gradient at the
            # point in time the collective's kernels may read it
            for lo, hi in ranges:
                self.shadow[lo:hi].copy_(self.flat[lo:hi])

    def reduce_range(self, lo, hi):
        self.reduce_ranges([(lo, hi)])

    def finish(self):
        for w in self.works:
            w.wait()
        self.works = []


def ensure_single_rank_group(backend=None):
    """A 1-rank process group in this process (BD_FORCE_DP=1: the data-parallel code path -- comm stream, side-stream event
    wait, async RCCL all-reduce per finished gradient range -- on ONE GPU, so that its stream ordering runs on hardware before
    a multi-GPU node exists).  No-op when a group is already initialised."""
    if dist.is_available() and dist.is_initialized():
        return
    import socket
    backend = backend or os.environ.get("BD_DIST_BACKEND", "nccl")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    kw = {"device_id": torch.device("cuda", torch.cuda.current_device())} if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, **kw)


def plan_segments(model):
    """[(lo, hi)] main ranges of the flat gradient in the order backward finalises them."""
    return [r[0] for r in plan_segment_ranges(model)]


def merge_ranges(ranges, pads=None):
    """union of [lo, hi) ranges as a sorted list of disjoint ranges (adjacent / overlapping ones fused): consecutive backward
    segments own adjacent stretches of the flat gradient -- alignment pads between tensors are zero and ride along.  A gap is
    bridged only when it IS padding: with `pads` (model._pads, the exact [lo, hi) pad stretches of the flat buffer) the gap must lie
    inside one of them; without, it must be shorter than one alignment unit (tensors start on 32-float boundaries, so a real pad is
    <= 31 floats -- a wider tolerance could swallow a small foreign tensor whose backward segment has not run yet)."""
    def is_pad(a, b):
        if pads is None:
            return b - a <= 31
        return any(plo <= a and b <= phi for plo, phi in pads)
    out = []
    for lo, hi in sorted(r for r in ranges if r[1] > r[0]):
        if out and (lo <= out[-1][1] or is_pad(out[-1][1], lo)):
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


def plan_buckets(seg_ranges, bucket_bytes, tail_bytes=8 << 20, pads=None):
    """Group the backward segments (in the order they finish) into communication buckets: a bucket closes with the first segment
    that brings it to `bucket_bytes`.  The LAST bucket is the only collective nothing can hide (it is issued behind the last
    data-gradient kernel), so it holds just the trailing segments that fit `tail_bytes`; a short remainder in front of it joins the
    bucket before.  Returns [(last segment index, [merged ranges])].  bucket_bytes <= 0: one bucket per segment."""
    sizes = [sum(hi - lo for lo, hi in rs) * 4 for rs in seg_ranges]
    n = len(seg_ranges)
    if bucket_bytes <= 0:
        return [(s, merge_ranges(rs, pads)) for s, rs in enumerate(seg_ranges)]
    tail, acc = n, 0
    while tail > 1 and acc + sizes[tail - 1] <= tail_bytes:
        tail -= 1
        acc += sizes[tail]
    groups, cur, size = [], [], 0
    for s in range(tail):
        cur.append(s)
        size += sizes[s]
        if size >= bucket_bytes:
            groups.append(cur)
            cur, size = [], 0
    if cur:
        if groups and size < bucket_bytes // 2:
            groups[-1] += cur
        else:
            groups.append(cur)
    if tail < n:
        groups.append(list(range(tail, n)))
    return [(g[-1], merge_ranges([r for s in g for r in seg_ranges[s]], pads)) for g in groups]


def plan_segment_ranges(model):
    """Per backward segment, ALL ranges of the flat gradient that are final once it has run: its own parameters and, for
    blocks with resnets, their rows of the batched time_emb_proj weight / bias."""
    lib = L.load()
    out = []
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    for s in range(lib.bd_unet_num_segments(model._plan)):
        rs = []
        for k in range(lib.bd_unet_segment_num_ranges(model._plan, s)):
            L.check(lib.bd_unet_segment_range_k(model._plan, s, k, ctypes.byref(lo), ctypes.byref(hi)), "bd_unet_segment_range_k")
            rs.append((lo.value, hi.value))
        out.append(rs)
    return out


class TrainEngine:
    def __init__(self, model, noise_sched, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0,
                 lr_warmup_steps=500, num_training_steps=None, loss_type="l2", process_group=None,
                 grad_accum_steps=1, use_graph=None, force_dp=None, dp_check=False):
        if not model.flat.is_cuda:
            raise RuntimeError("TrainEngine needs the model on a GPU")
        self.model, self.sched = model, noise_sched
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        self.warmup, self.total_steps = lr_warmup_steps, num_training_steps
        self.loss_type = loss_type
        self.pg = process_group
        # force_dp / BD_FORCE_DP=1: take the data-parallel path at world == 1 too (a 1-rank RCCL group is created when the
        # process has none).  dp_check: the optimizer consumes a SNAPSHOT of every gradient range taken on the collective's
        # stream right behind the all-reduce, and the gradient buffer is filled with NaN before every backward -- a collective
        # that is not ordered behind the kernels producing its range (side-stream weight gradients, the GroupNorm parameter
        # fold on the main stream) then ships NaNs / stale values into the weights instead of passing unnoticed (a sum over one
        # rank is the identity whenever it runs).
        if force_dp is None:
            force_dp = os.environ.get("BD_FORCE_DP", "0") == "1"
        self.force_dp = bool(force_dp)
        if self.force_dp and os.environ.get("BD_DP_TRANSPORT", "rccl") == "c10d":
            ensure_single_rank_group()      # c10d needs a process group even for one rank; the direct RCCL transport does not
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.dp = self.world > 1 or self.force_dp
        self.accum = int(grad_accum_steps)
        dev = model.flat.device
        n = model.num_flat
        self.grads = torch.zeros(n, device=dev)          # written by backward (padding stays 0)
        self.acc = torch.zeros(n, device=dev) if self.accum > 1 else None
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.sumsq = torch.zeros((), dtype=torch.float64, device=dev)
        self.grad_norm = torch.zeros((), device=dev)
        self.opt_step = 0      # optimizer steps taken
        self.micro = 0
        self._lib = L.load()
        self._nseg = self._lib.bd_unet_num_segments(model._plan)
        self._seg_ranges = plan_segment_ranges(model)
        # communication buckets: every collective has a fixed cost beside the backward kernels, xGMI rings
        # are per-link bound and want large messages -- segments are fused until a bucket holds BD_DP_BUCKET_MB (default 32 MB:
        # 5 buckets / 8 collectives for the CIFAR UNet's 143 MB instead of 12 / 30)
        self._buckets = plan_buckets(self._seg_ranges, int(float(os.environ.get("BD_DP_BUCKET_MB", "32")) * 2 ** 20), pads=model._pads)
        self._bucket_at = {s: rs for s, rs in self._buckets}
        # hipGraph replay of the whole step (fused q_sample -> forward -> loss -> backward -> clip + Adam): one launch per
        # step from the host instead of ~700.  Single-process, no gradient accumulation; BD_TRAIN_GRAPH=0/1 overrides.
        if use_graph is None:
            use_graph = os.environ.get("BD_TRAIN_GRAPH", "0") == "1"
        self.use_graph = bool(use_graph) and not self.dp and self.accum == 1
        self._graphs = {}
        # deferred join of the weight-gradient side stream (include/bd_hip.h: bd_unet_set_deferred_join); BD_DEFER_JOIN=0 = A/B
        self._defer = os.environ.get("BD_DEFER_JOIN", "1") != "0"
        L.check(self._lib.bd_unet_set_deferred_join(model._plan, 1 if self._defer else 0), "bd_unet_set_deferred_join")
        # transport of the gradient exchange: RCCL called directly on our own stream (rccl.py) whenever the ranks own one GPU each
        # (default group backend nccl, or the forced 1-rank path); c10d otherwise (gloo: CPU tests / ranks sharing a GPU)
        self._rccl = None
        self.transport_note = None       # why the direct RCCL transport is not in use (None: it is, or it was never asked for)
        if self.dp and os.environ.get("BD_DP_TRANSPORT", "rccl") == "none":
            self._rccl = "none"          # measurement only: at world > 1 the replicas drift apart (nothing is exchanged)
            if self.world > 1:           # a stray environment variable must not silently turn the gradient exchange off
                import warnings
                self.transport_note = "BD_DP_TRANSPORT=none: NO gradient exchange, the replicas drift apart (measurement only)"
                warnings.warn(self.transport_note)
        if self.dp and os.environ.get("BD_DP_TRANSPORT", "rccl") == "rccl":
            be = dist.get_backend(process_group) if (dist.is_available() and dist.is_initialized()) else "none"
            if process_group is None and (be == "nccl" or self.world == 1):
                self._rccl, self.transport_note = self._open_rccl(dev)
        self._comm = torch.cuda.Stream(device=dev) if (self.dp and (self._defer or self._rccl is not None)) else None
        self._dp_shadow = torch.zeros(n, device=dev) if (dp_check and self.dp) else None
        if dp_check and self.accum != 1:
            raise ValueError("dp_check needs grad_accum_steps == 1")
        self.collective_bytes = 0
        # measurement hooks (bench.py): events on the communication stream around every bucket's collective(s)
        self.time_buckets = False
        self._bucket_events = []
        self.alphas, self.alphas_cumprod = noise_sched.device_tables(dev)
        self.sync_state()

    def _open_rccl(self, dev):
        """(communicator or None, reason): the direct RCCL transport, agreed on COLLECTIVELY, stage by stage -- (1) every rank loads librccl,
        (2) rank 0 obtains the unique id, (3) after the id broadcast every rank runs ncclCommInitRank, (4) the self-test (an all-reduce of a
        rank-dependent vector against its closed form) -- with an all_reduce MIN of an ok flag over the launcher's group behind EACH stage,
        so a rank whose host-side call raised never reaches a torch.distributed collective the others are not in.  If any rank failed at
        any stage, EVERY rank destroys what it has and uses torch.distributed.all_reduce (ADVICE rounds 4, 5).  What this cannot cover: a
        rank that dies or blocks INSIDE ncclCommInitRank (a native rendezvous) leaves the others waiting there until RCCL's own timeout.
        Fault injection for the tests: BD_RCCL_FAIL_RANK=<rank>[:<stage>] (stage load | uid | init | selftest), honoured only with
        BD_TEST_HOOKS=1."""
        import warnings
        multi = self.world > 1

        def all_ok(ok):
            if not multi:
                return ok
            flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(int(flag.item()))

        me = dist.get_rank() if multi else 0
        inj = os.environ.get("BD_RCCL_FAIL_RANK", "") if os.environ.get("BD_TEST_HOOKS") == "1" else ""
        inj_rank, _, inj_stage = inj.partition(":")

        def inject(stage):
            if inj_rank == str(me) and (inj_stage or "load") == stage:
                raise RuntimeError(f"BD_RCCL_FAIL_RANK fault injection at stage {stage}")

        comm = None

        def give_up(reason):
            nonlocal comm
            if comm is not None:
                try:
                    torch.cuda.synchronize()
                    comm.destroy()
                except Exception:
                    pass
                comm = None
            warnings.warn(f"direct RCCL transport unavailable ({reason}); gradients go through torch.distributed.all_reduce on every rank")
            if not multi:
                ensure_single_rank_group()
            return None, reason

        def stage(name, fn, other):
            """run this rank's part of a stage, then agree: (ok on every rank, reason)"""
            err = None
            try:
                inject(name)
                fn()
            except (RuntimeError, OSError) as e:
                err = f"{name}: {type(e).__name__}: {e}"
            return all_ok(err is None), (err or other)

        from . import rccl
        ok, why = stage("load", rccl._load, "librccl failed to load on another rank")
        if not ok:
            return give_up(why if why.startswith("load") and "fault" in why else f"librccl not loadable ({why})")
        uid = [None]
        ok, why = stage("uid", lambda: uid.__setitem__(0, rccl.unique_id() if me == 0 else None), "ncclGetUniqueId failed on rank 0")
        if not ok:
            return give_up(why)
        if multi:
            dist.broadcast_object_list(uid, src=0)      # every rank is here: the stage above was agreed on

        def init():
            nonlocal comm
            comm = rccl.RcclComm(dev, uid=uid[0])
        ok, why = stage("init", init, "ncclCommInitRank failed on another rank")
        if not ok:
            return give_up(why)

        def selftest():
            r = comm.self_test()
            if r is not None:
                raise RuntimeError(r)
        ok, why = stage("selftest", selftest, "the communicator self-test failed on another rank")
        if not ok:
            return give_up(why)
        return comm, None

    # ---- measurement hooks (bench.py --gpus N) ---------------------------------------------------------------
    def set_buckets(self, bucket_mb):
        """re-plan the communication buckets (sweeps): same segments, another fusion size"""
        self._buckets = plan_buckets(self._seg_ranges, int(float(bucket_mb) * 2 ** 20), pads=self.model._pads)
        self._bucket_at = {s: rs for s, rs in self._buckets}

    def set_collectives(self, enabled):
        """False: every step runs the whole data-parallel path EXCEPT the collective calls (exposed-communication measurement: the
        replicas drift apart -- call sync_state() afterwards).  True: restore the transport chosen at construction."""
        if not enabled:
            if getattr(self, "_rccl_saved", None) is None:
                self._rccl_saved = (self._rccl,)
            self._rccl = "none"
        elif getattr(self, "_rccl_saved", None) is not None:
            self._rccl = self._rccl_saved[0]
            self._rccl_saved = None

    def bucket_times(self):
        """[(last segment of the bucket, bytes, mean ms, launches)] from the events recorded while time_buckets was on (synchronises)"""
        torch.cuda.synchronize()
        acc = {}
        for s, e0, e1, nbytes in self._bucket_events:
            a = acc.setdefault(s, [nbytes, 0.0, 0])
            a[1] += e0.elapsed_time(e1); a[2] += 1
        self._bucket_events = []
        return [(s, a[0], a[1] / a[2], a[2]) for s, a in sorted(acc.items())]

    def close(self):
        """Tear the gradient communicator down explicitly (while the HIP runtime is still up; otherwise it goes with the object)."""
        c, self._rccl = self._rccl, None
        if c not in (None, "none"):
            torch.cuda.synchronize()
            c.destroy()

    def sync_state(self):
        """Make every rank start from rank 0's parameters and Adam moments (replicas only ever exchange gradients:
        without this, ranks that initialised or resumed differently would apply the averaged gradient at different
        points and drift apart).  Call again after loading optimizer state on a resume."""
        if self.world > 1:
            for t in (self.model.flat.data, self.m, self.v):
                if self._rccl is not None and self._rccl != "none":
                    self._rccl.broadcast_(t, 0, torch.cuda.current_stream())
                else:
                    dist.broadcast(t, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)

    # ---- LR schedule (host scalar; baddiffusion.py:327-331) ------------------------------------------
    def current_lr(self):
        if self.total_steps is None:
            return self.lr
        return self.lr * cosine_schedule_with_warmup(self.opt_step, self.warmup, self.total_steps)

    # ---- pieces ---------------------------------------------------------------------------------------
    def forward_backward(self, xn, tg, t):
        """xn / tg NHWC [B,H,W,C] ; fills self.grads (already all-reduced when world > 1); returns loss."""
        model = self.model
        flat = model.flat.data
        pred, ws = model._run_forward(flat, xn, t, training=True)
        loss, dpred = ops.loss_fwd_bwd(pred, tg, self.loss_type, grad_scale=1.0 / (self.world * self.accum))
        B = xn.shape[0]
        if self._dp_shadow is not None:
            self.grads.fill_(float("nan"))
            for plo, phi in model._pads:
                self.grads[plo:phi].zero_()
        red = DPReducer(self.grads, self.pg, force=self.force_dp, shadow=self._dp_shadow, rccl=self._rccl, stream=self._comm)
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        main = torch.cuda.current_stream()
        for s in range(self._nseg):
            L.check(self._lib.bd_unet_backward_segment(
                model._plan, s, B, flat.data_ptr(), xn.data_ptr(), xn.shape[-1], dpred.data_ptr(), dpred.shape[-1],
                self.grads.data_ptr(), ws.data_ptr(), ws.numel(), L.stream(), ctypes.byref(lo), ctypes.byref(hi)),
                "bd_unet_backward_segment")
            if not self.dp or s not in self._bucket_at:
                continue
            ranges = self._bucket_at[s]
            # The segment's weight gradients may still be running on the plan's side stream (deferred join): the collective
            # is issued from a third stream that waits for (a) the main stream's position -- GroupNorm parameter fold, bias
            # rows -- and (b) the side stream's position, so the NEXT segment's kernels on the main stream are not held back.
            # RCCL orders itself after the issuing stream, then overlaps with the next segments; 143 MB per step for the
            # CIFAR UNet.
            if self._comm is not None:
                self._comm.wait_stream(main)
                if self._defer:     # (without the deferred join the segment call has already ordered `main` behind the side stream)
                    L.check(self._lib.bd_unet_stream_wait_aux(model._plan, self._comm.cuda_stream), "bd_unet_stream_wait_aux")
                with torch.cuda.stream(self._comm):
                    if self.time_buckets:     # `e0` fires once the waits above are satisfied, `e1` when the collective has finished
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    red.reduce_ranges(ranges)
                    if self.time_buckets:
                        e1.record()
                        self._bucket_events.append((s, e0, e1, sum(hi - lo for lo, hi in ranges) * 4))
            else:
                red.reduce_ranges(ranges)
        red.finish()          # the main stream waits for every collective (work.wait() orders the CURRENT stream)
        if self._comm is not None:
            main.wait_stream(self._comm)      # ... and for whatever else the issuing stream did behind them (dp_check snapshots)
        self.collective_bytes = red.bytes
        model._release_ws(ws)
        return loss

    def optimizer_step(self, grads):
        self.opt_step += 1
        lr = self.current_lr() if self.total_steps is None else \
            self.lr * cosine_schedule_with_warmup(self.opt_step - 1, self.warmup, self.total_steps)
        ops.sumsq(grads, out=self.sumsq)
        ops.adam_clip(self.model.flat.data, grads, self.m, self.v, self.sumsq, self.opt_step, lr, self.max_grad_norm,
                      self.betas, self.eps, grad_norm_out=self.grad_norm)

    # ---- hipGraph replay of the step ----------------------------------------------------------------------
    def _adam_hyper(self, step, lr):
        b1, b2 = self.betas
        return lr / (1.0 - b1 ** step), math.sqrt(1.0 - b2 ** step)

    def _graph_step(self, images, is_poison, trigger, target_img, noise, timesteps, row_index, flip):
        """Same launches as the eager step, captured once per input signature and replayed.  Everything that changes
        from step to step enters through static device buffers: the batch (or its row numbers into the resident
        dataset), noise, timesteps and the two Adam scalars {lr / (1 - b1^t), sqrt(1 - b2^t)} (bd_adam_clip_dev)."""
        key = (tuple(images.shape), images.dtype, images.data_ptr() if row_index is not None else 0, row_index is not None,
               flip is not None, tuple(noise.shape))
        ent = self._graphs.get(key)
        if ent is None:
            dev = noise.device
            st = {"is_poison": torch.empty_like(is_poison), "noise": torch.empty_like(noise), "t": torch.empty_like(timesteps),
                  "trigger": trigger.clone(), "target": target_img.clone(),
                  "images": images if row_index is not None else torch.empty_like(images),
                  "rows": torch.empty_like(row_index) if row_index is not None else None,
                  "flip": torch.empty_like(flip) if flip is not None else None,
                  "hyper": torch.zeros(2, device=dev), "loss": torch.zeros((), device=dev)}
            ring = torch.zeros(64, 2).pin_memory()

            def body():
                xn, tg = ops.poison_qsample(st["images"], st["is_poison"], st["trigger"], st["target"], st["noise"], st["t"],
                                            self.alphas, self.alphas_cumprod, row_index=st["rows"], flip=st["flip"])
                loss = self.forward_backward(xn, tg, st["t"])
                ops.sumsq(self.grads, out=self.sumsq)
                b1, b2 = self.betas
                L.check(self._lib.bd_adam_clip_dev(self.model.flat.data.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                                                   self.v.data_ptr(), self.grads.numel(), self.sumsq.data_ptr(), float(self.max_grad_norm),
                                                   st["hyper"].data_ptr(), float(b1), float(b2), float(self.eps),
                                                   self.grad_norm.data_ptr(), L.stream()), "bd_adam_clip_dev")
                st["loss"].copy_(loss)
            ent = {"st": st, "ring": ring, "graph": None, "body": body, "n": 0}
            self._graphs[key] = ent
        st = ent["st"]
        st["is_poison"].copy_(is_poison); st["noise"].copy_(noise); st["t"].copy_(timesteps)
        if row_index is None:
            st["images"].copy_(images)
        else:
            st["rows"].copy_(row_index)
        if flip is not None:
            st["flip"].copy_(flip)
        self.opt_step += 1
        self.micro += 1
        lr = self.lr if self.total_steps is None else self.lr * cosine_schedule_with_warmup(self.opt_step - 1, self.warmup, self.total_steps)
        slot = ent["ring"][ent["n"] % 64]
        slot[0], slot[1] = self._adam_hyper(self.opt_step, lr)
        st["hyper"].copy_(slot, non_blocking=True)
        ent["n"] += 1
        if ent["graph"] is None and ent["n"] >= 2:        # step 1 runs eagerly (lazy allocations, side stream creation)
            # Captured in the single-stream order: a replayed graph with the plan's ~350 fork / join edges to the side
            # stream runs 1.6x SLOWER than the eager two-stream schedule on ROCm 7.2 (40 vs 24 ms / step), the
            # single-stream graph equals the single-stream eager step.
            self.model.set_aux_stream(False)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                ent["body"]()
            ent["graph"] = g
            # the capture did not execute: replay below performs this step
        if ent["graph"] is None:
            ent["body"]()
        else:
            ent["graph"].replay()
        return st["loss"]

    # ---- the train step -----------------------------------------------------------------------------------
    def train_step(self, images, is_poison, trigger, target_img, noise, timesteps, row_index=None, flip=None):
        """images: uint8 [B,H,W,C] or float [B,C,H,W] on the GPU; is_poison bool [B]; trigger/target [C,H,W];
        noise [B,C,H,W]; timesteps int64 [B].  With row_index (int64 [B]) `images` is the whole HBM-resident dataset
        and the shuffled gather (and, with flip [B], the random horizontal flip) happens inside the fused kernel.
        Returns the (local) loss as a device scalar."""
        if self.use_graph and not self._lib.bd_prof_enabled():
            return self._graph_step(images, is_poison.to(torch.uint8) if is_poison.dtype == torch.bool else is_poison, trigger,
                                    target_img, noise, timesteps.to(torch.int64), row_index, flip)
        xn, tg = ops.poison_qsample(images, is_poison, trigger, target_img, noise, timesteps, self.alphas,
                                    self.alphas_cumprod, row_index=row_index, flip=flip)
        return self.step_from_noisy(xn, tg, timesteps)

    def train_step_batch(self, x_start, R, noise, timesteps):
        """The reference's calling convention (baddiffusion.py:593-607): collated x_start / R batches."""
        xn, tg = ops.qsample(x_start, R, noise, timesteps, self.alphas, self.alphas_cumprod)
        return self.step_from_noisy(xn, tg, timesteps)

    def step_from_noisy(self, xn, tg, timesteps):
        t = timesteps.to(torch.int64).contiguous()
        loss = self.forward_backward(xn, tg, t)
        self.micro += 1
        if self.accum > 1:
            lib = self._lib
            L.check(lib.bd_axpy(self.grads.data_ptr(), self.acc.data_ptr(), self.grads.numel(), 1.0,
                                int(self.micro % self.accum != 1), L.stream()), "bd_axpy")
            if self.micro % self.accum == 0:
                self.optimizer_step(self.acc)
        else:
            self.optimizer_step(self.grads if self._dp_shadow is None else self._dp_shadow)
        return loss
