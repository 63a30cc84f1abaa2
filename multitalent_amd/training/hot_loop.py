"""The device side of `run_iteration` (MultiTalent_Trainer_DDP.py:324-370, nnUNetTrainerV2.py:225-274):
forward -> loss -> backward (+ DDP gradient all-reduce overlapped with backward) -> clip_grad_norm_(12) ->
SGD-Nesterov step, all on the HIP engine with flat parameter / gradient / momentum buffers.

The network forward/backward bypass torch autograd entirely (engine.forward / engine.backward), and so does the loss when it offers
`fused_step` (statistics kernels -> mt_loss_combine -> backward kernels); autograd only runs for a loss callable without it or for
the cases `fused_step` declines (few-hundred-float combination on top of the fused statistics kernels)."""
import os

import torch
import torch.distributed as dist

from .. import ops


class GradAllReducer:
    """Bucketed gradient all-reduce (mean) over RCCL, launched on a side HIP stream as soon as a contiguous
    slice of the flat gradient buffer is final (the buffer is laid out in backward-completion order), i.e.
    overlapped with the rest of backward.  Replaces torch DDP's bucket hooks (nnUNetTrainerV2_DDP.py:200)."""

    def __init__(self, engine, bucket_bytes=32 << 20):
        self.eng = engine
        self.bucket = bucket_bytes // 4
        self.sent = 0
        self.stream = None
        self.handles = []
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # MT_FORCE_REDUCER=1 runs the full bucketed / side-stream code path even at world size 1 (single-GPU validation)
        import os
        self.force = bool(int(os.environ.get('MT_FORCE_REDUCER', '0'))) and dist.is_available() and dist.is_initialized()
        self.reset_stats()

    # ---- communication accounting (shown by bench.py as `comm`): is the all-reduce hidden behind backward? ----------------
    def reset_stats(self):
        self._steps = 0
        self._bytes = 0
        self._buckets = 0
        self._ev_wait = []        # (before, after) on the compute stream around wait_stream(side): what backward did NOT hide
        self._ev_comm = []        # (before, after) on the side stream around each bucket's scale + all-reduce

    def stats(self):
        """per-step averages since reset_stats(): bytes all-reduced, buckets, side-stream busy time and the time the compute
        stream had to wait for the side stream before clip / SGD (0 = fully overlapped)."""
        if self._steps == 0:
            return None
        out = {"allreduce_bytes_per_step": int(self._bytes / self._steps), "buckets_per_step": round(self._buckets / self._steps, 2),
               "bucket_bytes": int(self.bucket * 4), "world": self.world}
        if self._ev_wait:
            torch.cuda.synchronize()
            timed = len(self._ev_wait)          # steps whose events were kept (the first 4096: see begin()); averages are over THOSE steps
            out["exposed_wait_ms_per_step"] = round(sum(a.elapsed_time(b) for a, b in self._ev_wait) / timed, 3)
            out["side_stream_busy_ms_per_step"] = round(sum(a.elapsed_time(b) for a, b in self._ev_comm) / timed, 3)
            out["timed_steps"] = timed
        return out

    def begin(self):
        self.sent = 0
        self.handles = []
        self._timing = len(self._ev_wait) < 4096     # keep the events of the first 4096 steps only (one decision per step)
        self.on_gpu = self.eng.flat_grad.is_cuda
        active = self.world > 1 or self.force
        # gloo has no device path here: two ranks sharing ONE GPU (the single-GPU test box) or CPU tensors.  Device slices then
        # travel through the host at finish() — transport only, same bucketing and arithmetic.
        self.via_host = active and self.on_gpu and dist.get_backend() == 'gloo'
        if active and self.stream is None and self.on_gpu and not self.via_host:
            self.stream = torch.cuda.Stream()

    def ready(self, lo, hi):
        """all gradients in flat_grad[0:hi) are final."""
        if self.world <= 1 and not self.force:
            return
        n = self.eng.flat_grad.numel()
        final = hi >= n
        while hi - self.sent >= self.bucket or (final and self.sent < n):
            end = n if final and (n - self.sent) < 2 * self.bucket else self.sent + self.bucket
            sl = self.eng.flat_grad[self.sent:end]
            if self.via_host:
                self.handles.append(sl)
            elif self.on_gpu:
                # the side stream waits for the kernels that produced this slice (event on the compute stream), scales and
                # all-reduces it there; finish() makes the compute stream wait for the side stream before clip / SGD read it
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                evs = [ev]
                for st in getattr(self.eng, 'wstreams', ()):      # the weight gradients are produced on the engine's side streams
                    e2 = torch.cuda.Event()
                    e2.record(st)
                    evs.append(e2)
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(self.stream):
                    for e in evs:
                        self.stream.wait_event(e)
                    c0.record(self.stream)
                    sl.div_(self.world)
                    self.handles.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True))
                    self.handles[-1].wait()          # orders the side stream behind the collective (no host block for NCCL work)
                    c1.record(self.stream)
                if self._timing:
                    self._ev_comm.append((c0, c1))
            else:       # host tensors (gloo): used by the CPU tests of the bucketing logic
                sl.div_(self.world)
                self.handles.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True))
            self._bytes += int(sl.numel()) * 4
            self._buckets += 1
            self.sent = end

    def finish(self):
        if self.world <= 1 and not self.force:
            return
        self._steps += 1
        if self.via_host:
            for sl in self.handles:
                h = sl.cpu()
                h.div_(self.world)
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                sl.copy_(h)
            return
        if self.on_gpu:
            cur = torch.cuda.current_stream()
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0.record(cur)
            for h in self.handles:          # stream-level waits (RCCL work objects do not block the host)
                h.wait()
            cur.wait_stream(self.stream)
            w1.record(cur)
            if self._timing:
                self._ev_wait.append((w0, w1))
        else:
            for h in self.handles:
                h.wait()


class FusedTrainStep:
    def __init__(self, network, loss_fn, lr=1e-2, weight_decay=3e-5, momentum=0.99, max_norm=12.0, ddp=False):
        self.net = network
        self.eng = network.engine()
        self.loss_fn = loss_fn
        self.lr, self.wd, self.mom, self.max_norm = lr, weight_decay, momentum, max_norm
        self.buf = None
        self.first = True
        self.sumsq = None
        self.ws = None
        self.reducer = GradAllReducer(self.eng) if ddp else None
        self.head_params, self.head_opt = None, None
        self.fused_loss = os.environ.get('MT_FUSED_LOSS', '1') != '0'      # 0: always the autograd form of the loss (A/B, tests)
        self._loss_form_checked = False
        # MT_STEP_GRAPH=1 (default 0): the steady-state step — the same ~450 launches on two streams every iteration — is captured once as a
        # HIP graph per (shapes, hyper-parameters) and replayed: the host enqueues one graph instead of hundreds of launches.  Bit-identical
        # to the eager step (tests/test_step_graph_gpu.py), but on ROCm 7.2 the replay is SLOWER than the eager enqueue it replaces
        # [measured, round 6: Task009 fp32 29.74 -> 31.49 ms, residual encoder mixed 21.61 -> 23.14, Task009 mixed 10.72 -> 11.90]: the graph
        # executor loses more between dependent nodes than the host's 7 % of idle time it removes.  Kept opt-in.  Single process, fused SGD only.
        self.use_graph = os.environ.get('MT_STEP_GRAPH', '0') == '1'
        self._graph = None
        self._eager_seen = {}            # graph key -> eager steps run with it (planning, packing programs and buffers exist before a capture)
        self._eager_steps = 0
        self.last_logits = None      # full-resolution logits (NCDHW view of the engine's NDHWC buffer) of the latest forward

    def set_head_optimizer(self, params, lr=3e-3, weight_decay=3e-5):
        """Heads-only phase of the fine-tuning trainers (reference nnUNetTrainerV2_warmup.py:119-132): AdamW(amsgrad) on `params`
        (a few 1x1x1 conv tensors: torch glue, not a hot path) instead of the fused SGD over the flat buffer; forward, loss,
        backward and the clip norm over ALL parameters stay the same launches.  None switches back to SGD.  `self.lr` remains
        the knob the trainers' schedules turn."""
        if params is None:
            self.head_params, self.head_opt = None, None
            return
        self.head_params = list(params)
        self.head_opt = torch.optim.AdamW(self.head_params, lr, weight_decay=weight_decay, amsgrad=True)

    def reset_momentum(self):
        """a fresh optimizer instance in the reference = momentum buffers start from the next gradient."""
        if self.buf is not None:
            self.buf.zero_()
        self.first = True

    def _head_step(self):
        eng = self.eng
        # torch.nn.utils.clip_grad_norm_(network.parameters(), 12): coefficient from the norm of the WHOLE flat gradient
        coef = torch.clamp(self.max_norm / (self.sumsq.sqrt() + 1e-6), max=1.0)
        for p in self.head_params:
            p.grad = eng.grad_of(p) * coef
        for g in self.head_opt.param_groups:
            g['lr'] = self.lr
        self.head_opt.step()
        for p in self.head_params:
            p.grad = None

    def _state(self, dev):
        if self.buf is None or self.buf.device != dev or self.buf.numel() != self.eng.flat.numel():
            self.buf = torch.zeros_like(self.eng.flat)
            self.first = True
            self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
            self.ws = torch.empty(ops.sumsq_workspace(self.eng.flat.numel()) // 4 + 16, dtype=torch.float32, device=dev)

    def _loss_on(self, outs, loss_args):
        leaves = [o.permute(0, 4, 1, 2, 3).requires_grad_(True) for o in outs]
        res = self.loss_fn(leaves, *loss_args)
        return leaves, res

    def forward_loss(self, data, loss_args):
        """Forward + the loss in its autograd form (any callable with the reference's compute_loss signature)."""
        return self._loss_on(self.eng.forward(data, need_grad=True, all_heads=True), loss_args)

    def __call__(self, data, *loss_args, do_backprop=True):
        """Returns whatever loss_fn returns (loss first if a tuple), detached device tensors (no host sync)."""
        eng = self.eng
        if not do_backprop:
            with torch.no_grad():
                outs = eng.forward(data, need_grad=False, all_heads=True)
                res = self.loss_fn([o.permute(0, 4, 1, 2, 3) for o in outs], *loss_args)
                self.last_logits = outs[0].permute(0, 4, 1, 2, 3)       # online evaluation reads THIS forward's output, like the reference
            return res
        if self._graphable(data):
            done = self._graph_step(data, loss_args)
            if done is not None:
                return done
        self._eager_steps += 1
        return self._train_step(data, loss_args)

    # ---- HIP graph of the steady-state step -----------------------------------------------------------------------------------
    def _graphable(self, data):
        return (self.use_graph and data.is_cuda and self.reducer is None and self.head_opt is None and not self.first
                and self.fused_loss and hasattr(self.loss_fn, 'static_args') and not torch.cuda.is_current_stream_capturing())

    def invalidate_graph(self):
        self._graph = None

    def _graph_step(self, data, loss_args):
        """Replay (or capture, then replay) the graph of one training step.  The graph reads its inputs from STATIC tensors (the batch and
        the loss arguments are copied into them: device-to-device, a few MB) and leaves the loss values in static tensors that the next
        call overwrites.  Anything that changes what the step launches — shapes, precision, learning rate (a kernel argument), the
        engine's stream setup — is part of the key; a new key captures a new graph (the poly schedule: once per epoch)."""
        eng = self.eng
        sa = self.loss_fn.static_args(*loss_args)
        if sa is None:
            return None
        tensors, rebuild, skey = sa
        key = (tuple(data.shape), data.dtype, tuple((tuple(t.shape), t.dtype) for t in tensors), skey, float(self.lr), float(self.wd), float(self.mom),
               float(self.max_norm), eng.mma, eng.bwdw_streams, eng._planned, eng.flat.data_ptr())
        g = self._graph
        if (g is None or g['key'] != key) and self._eager_seen.get(key, 0) < 2:
            self._eager_seen[key] = self._eager_seen.get(key, 0) + 1        # (a new shape plans, allocates and uploads tables: not capturable)
            return None
        if g is None or g['key'] != key:
            sx = data.clone()
            st = [t.clone() for t in tensors]
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    res = self._train_step(sx, rebuild(st))
            except Exception as ex:            # noqa: BLE001 — a step that cannot be captured runs eagerly, loudly, from now on
                import sys
                sys.stderr.write("multitalent_amd: HIP graph capture of the training step failed (%s: %s); running eagerly\n" % (type(ex).__name__, str(ex)[:300]))
                self.use_graph = False
                self._graph = None
                torch.cuda.synchronize()
                return None
            g = self._graph = {'key': key, 'graph': graph, 'x': sx, 't': st, 'res': res, 'logits': self.last_logits}
        else:
            g['x'].copy_(data)
            for a, b in zip(g['t'], tensors):
                a.copy_(b)
        g['graph'].replay()
        self.last_logits = g['logits']
        eng.mark_params_dirty()
        return g['res']

    def _train_step(self, data, loss_args):
        eng = self.eng
        outs = eng.forward(data, need_grad=True, all_heads=True)
        # value + dLoss/dlogits straight from the loss kernels (mt_loss_combine) when the loss offers it: no autograd graph over
        # the [L, B, C] glue; None = a case only the autograd form covers
        fused = self.loss_fn.fused_step(outs, *loss_args) if self.fused_loss and hasattr(self.loss_fn, 'fused_step') else None
        if not self._loss_form_checked:
            # The fused form and the autograd form issue DIFFERENT numbers of collectives under batch Dice (one all-reduce of the
            # statistics with dLoss/dstats scaled by the world size, against the all-gather + all-reduce pairs of the autograd graph), and
            # the world-size scaling is only the all-gather's backward while every rank holds the same dLoss/d(dice sums) — the same
            # (L, B, C) and deep-supervision weights.  Both are properties of the configuration, not of a call; pinned here once
            # (ADVICE r5): a rank that decided differently raises instead of deadlocking.
            self._loss_form_checked = True
            from . import distributed_utils
            sig = [1.0 if fused is not None else 0.0, len(outs)] + [float(v) for o in outs for v in (o.shape[0], o.shape[-1])]
            sig += [float(w) for w in getattr(self.loss_fn, 'ds_loss_weights', [])]
            distributed_utils.assert_same_on_all_ranks(sig, "the loss form (fused / autograd), the levels' (B, C) or the deep-supervision weights")
        if fused is not None:
            res, dl = fused
            self.last_logits = outs[0].permute(0, 4, 1, 2, 3)
        else:
            leaves, res = self._loss_on(outs, loss_args)
            self.last_logits = leaves[0].detach()
            loss = res[0] if isinstance(res, (tuple, list)) else res
            loss.backward()
            dl = [None if l.grad is None else l.grad.permute(0, 2, 3, 4, 1).contiguous() for l in leaves]
        if self.reducer is not None:
            self.reducer.begin()
            eng.grad_ready_hook = self.reducer.ready
        eng.backward(dl)
        if self.reducer is not None:
            self.reducer.finish()
        self._state(eng.flat.device)
        ops.sumsq(eng.flat_grad, self.sumsq, self.ws)
        if self.head_opt is not None:
            self._head_step()
        else:
            ops.sgd_nesterov(eng.flat, eng.flat_grad, self.buf, self.lr, self.wd, self.mom, self.first, self.sumsq, self.max_norm)
            self.first = False
        eng.mark_params_dirty()
        if isinstance(res, (tuple, list)):
            return tuple(r.detach() for r in res)
        return res.detach()
