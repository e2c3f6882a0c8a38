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


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_numa_node(local_rank: int, root: str = "/"):
    """NUMA node of HIP device `local_rank`, from sysfs alone (no GPU call: this runs BEFORE the HIP runtime starts).  The KFD topology
    lists the GPU agents in HIP's enumeration order (nodes with simd_count > 0); HIP_/ROCR_VISIBLE_DEVICES index lists are applied.
    -> (node or None, pci address or None)."""
    import glob
    gpus = []
    for d in sorted(glob.glob(os.path.join(root, "sys/class/kfd/kfd/topology/nodes/*")), key=lambda x: int(os.path.basename(x))):
        props = dict(l.split(None, 1) for l in (_read(os.path.join(d, "properties")) or "").splitlines() if " " in l)
        if int(props.get("simd_count", "0")) > 0:
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            gpus.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v and all(x.strip().isdigit() for x in v.split(",")):
            gpus = [gpus[int(x)] for x in v.split(",") if int(x) < len(gpus)]
    if not (0 <= local_rank < len(gpus)):
        return None, None
    node = _read(os.path.join(root, "sys/bus/pci/devices", gpus[local_rank], "numa_node"))
    return (int(node) if node not in (None, "") and int(node) >= 0 else None), gpus[local_rank]


def bind_to_gpu_numa(local_rank: int, root: str = "/"):
    """Pin this process (and the threads it starts later: the mask check workers, the DataLoader hand-over) to the CPUs of the NUMA node
    its GPU hangs off -- on an 8-GPU MI355X host the GPUs sit on two sockets, and a rank whose host threads run on the other socket
    pays the inter-socket hop on every launch and every pinned-memory copy.  Call before the first GPU call.  Best effort: returns a
    description (what was done or why not) and never raises.  MV_NUMA_BIND=0 switches it off."""
    info = {"local_rank": local_rank, "bound": False}
    if os.environ.get("MV_NUMA_BIND", "1") == "0":
        info["why"] = "MV_NUMA_BIND=0"
        return info
    try:
        node, pci = gpu_numa_node(local_rank, root)
        info["pci"], info["numa_node"] = pci, node
        if node is None:
            info["why"] = "no NUMA node recorded for this GPU (single-node host or sysfs not readable)"
            return info
        cpulist = _read(os.path.join(root, f"sys/devices/system/node/node{node}/cpulist"))
        cpus = set()
        for part in (cpulist or "").split(","):
            if "-" in part:
                lo, hi = part.split("-")
                cpus.update(range(int(lo), int(hi) + 1))
            elif part.strip():
                cpus.add(int(part))
        allowed = os.sched_getaffinity(0)
        want = cpus & allowed
        if not want:
            info["why"] = "the node's CPUs are outside this process's affinity mask"
            return info
        os.sched_setaffinity(0, want)
        info["bound"], info["cpus"] = True, len(want)
    except Exception as e:               # never cost a run
        info["why"] = repr(e)
    return info


def rank_environment(extra=None):
    """What every rank of the job sees of the settings that decide how streams and RCCL behave -- gathered to rank 0 (bench line)."""
    env = {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY", "HIP_FORCE_DEV_KERNARG", "NCCL_DEBUG",
                                          "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}
    env["rank"] = dist.get_rank() if dist.is_initialized() else 0
    if extra:
        env.update(extra)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [env]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, env)
    return out


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
        self._bucket_ev = []           # timing only: per step [[name, bytes, issue event, event after the compute stream's wait for it]]
        self._bucket_log = []

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

    def _launch(self, s, e, events=(), name=""):
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
                if self.timing:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(self.stream)       # the bucket's gradients are final here: the collective is issued now
                w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.works.append(w)
            if self.timing:
                # (the completion side is stamped in finish(), on the compute stream, as it passes each bucket's wait: a stream of its own
                #  that waits per bucket measured the buckets directly but cost 4.6 ms per step on the one-rank rehearsal -- its event
                #  waits shared a hardware queue with the side stream)
                self._bucket_ev.append([name, (e - s) * 4, t0, None])
        else:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def hook(self, name: str, event=None):
        """Called by Engine.encoder_backward when the gradients of bucket `name` are final (`event`: side-stream
        event after the bucket's weight-gradient GEMMs, or None when everything is on the current stream)."""
        if name == "heads":
            # mlm.predictions.* / pooler / itm are final; the tied E gradient is NOT (embedding scatter comes last)
            self._launch(*self.ranges["heads"], events=(event,), name="heads")
        elif name.startswith("layer"):
            l = int(name[5:])
            if self._pending_hi is None:
                self._pending_hi = self.ranges[name][1]
                self._pending_ev = []
            self._pending_ev.append(event)
            if l % self.merge == 0 or l == 0:
                self._launch(self.ranges[name][0], self._pending_hi, self._pending_ev, name=f"layers{l}..")
                self._pending_hi = None
        elif name == "embeddings":
            self._launch(*self.ranges["embeddings"], name="embeddings")

    def finish(self):
        """Make the current stream wait for every outstanding bucket."""
        timed = self.timing and self.cuda and self.works
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        for i, w in enumerate(self.works):
            w.wait()
            if timed and i < len(self._bucket_ev):
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record(torch.cuda.current_stream())          # the compute stream is past this bucket's wait
                self._bucket_ev[i][3] = t1
        if timed:
            e1.record(torch.cuda.current_stream())
            self._exposed.append((e0, e1))
            if self._bucket_ev:
                self._bucket_log.append((self._bucket_ev, e0, e1))
        self._bucket_ev = []
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

    def bucket_timeline(self):
        """Per bucket, averaged over the timed steps: when its all-reduce was ISSUED (its gradients final, on the issuing stream) and when
        the COMPUTE stream was past its wait for it (`done_at_ms`: an upper bound of its completion -- the compute stream only starts
        waiting at the end of the backward, `compute_stream_wait_from_ms`), in ms relative to the issue of the step's first bucket.
        Buckets whose done_at equals the wait's start finished under the backward; the ones after it are the exposed tail."""
        if not self._bucket_log:
            return None
        torch.cuda.synchronize()
        acc, n = {}, 0
        wait0 = wait1 = 0.0
        for evs, e0, e1 in self._bucket_log:
            base = evs[0][2]
            for i, (name, nbytes, t0, t1) in enumerate(evs):
                if t1 is None:
                    continue
                a = acc.setdefault(i, dict(bucket=name, mbytes=nbytes / 1e6, issued_at_ms=0.0, done_at_ms=0.0))
                a["issued_at_ms"] += base.elapsed_time(t0)
                a["done_at_ms"] += base.elapsed_time(t1)
            wait0 += base.elapsed_time(e0)
            wait1 += base.elapsed_time(e1)
            n += 1
        self._bucket_log.clear()
        out = []
        for i in sorted(acc):
            a = acc[i]
            a["issued_at_ms"], a["done_at_ms"] = a["issued_at_ms"] / n, a["done_at_ms"] / n
            out.append(a)
        return {"buckets": out, "compute_stream_wait_from_ms": wait0 / n, "compute_stream_wait_until_ms": wait1 / n, "steps": n}
