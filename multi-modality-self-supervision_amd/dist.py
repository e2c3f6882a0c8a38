"""Data-parallel exchange for the pretraining step: one process per GPU, RCCL over xGMI.

Replaces nn.DataParallel of models/train_origin.py:53-55 (per-step parameter broadcast, input
scatter, [B,L,V] logit gather and gradient reduce to GPU 0) by the only exchange the math
needs:
  * one tiny all-reduce of (#labelled tokens, local batch) BEFORE the backward, so that every
    rank back-propagates  mlm_nll_sum / n_labelled_global + itm_nll_sum / B_global  -- the
    gradient of the reference's global-batch mean losses;
  * sum all-reduce of the flat fp32 gradient, cut into contiguous buckets (heads, one per
    encoder layer, embeddings) that are launched on a side stream as soon as the backward has
    finished them, so the exchange overlaps the remaining backward kernels.
Parameters are never broadcast after construction: identical init + identical updates; `check_replicas` proves the
"identical init" half once, at TrainStep construction, with an all-reduced checksum (a per-rank seed would otherwise
diverge silently).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def bucket_ranges(layout, n_flat, layers):
    """name -> (start, end) element ranges of the flat buffer, in forward (= memory) order."""
    def off(name):
        return layout[name][0]
    r = {}
    first_layer = off("enc.encoder.layer.0.attention.self.query.weight") if layers > 0 else off("enc.pooler.dense.weight")
    r["embeddings"] = (0, first_layer)
    for l in range(layers):
        s = off(f"enc.encoder.layer.{l}.attention.self.query.weight")
        e = off(f"enc.encoder.layer.{l + 1}.attention.self.query.weight") if l + 1 < layers else off("enc.pooler.dense.weight")
        r[f"layer{l}"] = (s, e)
    r["heads"] = (off("enc.pooler.dense.weight"), n_flat)
    return r


class GradAllReducer:
    def __init__(self, flat_g: torch.Tensor, layout, n_flat: int, layers: int, group=None, merge_layers: int = 2, stream=None):
        self.flat_g = flat_g
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # rehearsal on a one-GPU box: MV_DP_FORCE=1 issues every collective even in a one-rank group, so that the RCCL calls,
        # the communication stream and the Work handles are exercised for real (tests/test_dp_gpu.py)
        self.force = os.environ.get("MV_DP_FORCE") == "1" and dist.is_initialized()
        self.ranges = bucket_ranges(layout, n_flat, layers)
        self.merge = max(1, merge_layers)      # layers per bucket: ~2 x 28 MB fp32 at BERT-base
        self.layers = layers
        self.cuda = flat_g.is_cuda
        # The stream the collectives are ISSUED from (the process group runs them on its own internal stream behind it).  TrainStep passes
        # the engine's side stream: a stream of our own meant one more queue user, and a stream whose head is an event wait blocks whatever
        # else shares its hardware queue -- the one-rank RCCL rehearsal ran 25.7-30.5 ms per step (box-dependent) against 24.2
        self.stream = stream if stream is not None else (torch.cuda.Stream(device=flat_g.device) if self.cuda else None)
        self.works = []
        self._pending_hi = None
        self._pending_ev = []
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.timing = False            # bench: record how long the compute stream waits for the exchange in finish()
        self._exposed = []

    def check_replicas(self, flat_p: torch.Tensor):
        """Raise unless every rank holds the same parameters (sum and sum of squares, all-reduced MIN and MAX)."""
        if self.world == 1 and not self.force:
            return
        d = flat_p.double()
        c = torch.stack([d.sum(), (d * d).sum()])
        lo, hi = c.clone(), c.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo, hi):
            raise RuntimeError("data-parallel replicas start from different parameters (seed torch identically on every rank "
                               f"before building the model, or load the same checkpoint): checksum range {lo.tolist()} .. {hi.tolist()}")

    def global_counts(self, n_labelled: int, batch: int, device):
        """-> f32[2] device tensor (n_labelled_global, B_global); one small all-reduce."""
        # two fill kernels, NOT torch.tensor([...], device=...): that is a pageable host-to-device copy, which blocks the launching thread
        # until everything already queued on the stream (the whole forward) has run -- the host lost its lead over the device every
        # step (one-rank RCCL rehearsal: 25.7-28.5 ms per step against 23.6-24.5 undistributed; profiles/r03_notes.txt)
        t = torch.empty(2, dtype=torch.float32, device=device)
        t[0].fill_(float(n_labelled))
        t[1].fill_(float(batch))
        if self.world > 1 or self.force:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _launch(self, s, e, events=()):
        if (self.world == 1 and not self.force) or e <= s:
            return
        view = self.flat_g[s:e]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                for x in events:                 # gradients produced on the engine's side stream
                    if x is not None:
                        self.stream.wait_event(x)
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def hook(self, name: str, event=None):
        """Called by Engine.encoder_backward when the gradients of bucket `name` are final (`event`: side-stream
        event after the bucket's weight-gradient GEMMs, or None when everything is on the current stream)."""
        if name == "heads":
            # mlm.predictions.* / pooler / itm are final; the tied E gradient is NOT (embedding scatter comes last)
            self._launch(*self.ranges["heads"], events=(event,))
        elif name.startswith("layer"):
            l = int(name[5:])
            if self._pending_hi is None:
                self._pending_hi = self.ranges[name][1]
                self._pending_ev = []
            self._pending_ev.append(event)
            if l % self.merge == 0 or l == 0:
                self._launch(self.ranges[name][0], self._pending_hi, self._pending_ev)
                self._pending_hi = None
        elif name == "embeddings":
            self._launch(*self.ranges["embeddings"])

    def finish(self):
        """Make the current stream wait for every outstanding bucket."""
        timed = self.timing and self.cuda and self.works
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        for w in self.works:
            w.wait()
        if timed:
            e1.record(torch.cuda.current_stream())
            self._exposed.append((e0, e1))
        self.works.clear()
        self._pending_hi = None

    def exposed_ms(self):
        """Mean time per step the compute stream spent waiting in finish() (events recorded while `timing`)."""
        if not self._exposed:
            return 0.0
        torch.cuda.synchronize()
        v = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed.clear()
        return sum(v) / len(v)
