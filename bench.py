#!/usr/bin/env python3
"""Benchmark of the MedViLL / CXRBERT pretraining step on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 needs one process per GPU: when WORLD_SIZE is not set, this script starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...` itself as a CHILD process (before
anything here touches the GPU) and relays rank 0's JSON line; launched under torchrun it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.

One "step" = the loop body of models/train_origin.py:95-146 on one synthetic mini-batch that is already resident in
HBM: forward, CE(mlm, ignore -100) + CE(itm), backward, HF AdamW, the ITM/MLM accuracy counters, and (N > 1) the RCCL
gradient all-reduce.  Workload = BASELINE.json configs[1]: BERT-base (12L/12H/768), L = 512 (36 regions + 476 text),
bidirectional mask, 16-bit MFMA path, batch 64 per GPU (weak scaling), random-init weights, synthetic inputs.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     the dominant kernel (the FFN-up MFMA GEMM with its fused epilogue) timed alone with HIP events on its
               stream, against the dense bf16 MFMA peak; `roofline.step` carries the whole step: `frac` = EXECUTED FLOP
               rate / peak, `dense_frac` = SURVEY 8d's dense MFU convention (364.076 GFLOP/sample);
  cpu_baseline the CPU oracle (a port; the reference's Python cannot travel) timed on the host cores on a bounded
               sample (B = 2, same shape).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 / f16 MFMA (MI355X_MICROARCH.md); never the 2:1-sparse figure
CONFIGS = {
    "c2": dict(name="BERT-base L512 (36 regions + 476 text) bidirectional", N=36, S=473, family="full", max_pos=512),
    "c3": dict(name="BERT-base L512 Bi+Seq2Seq mixed 75/25", N=36, S=473, family="mixed", max_pos=512),
    "c4": dict(name="BERT-base L512 non-cross modality mask", N=36, S=473, family="noncross", max_pos=512),
    "c5": dict(name="BERT-base L768 (100 regions + 668 text) seq2seq", N=100, S=665, family="s2s", max_pos=768),
}


def flops_fwd_per_sample(H, I, V, D, layers, L, N):
    return 2.0 * N * D * H + layers * (L * (8.0 * H * H + 4.0 * H * I) + 4.0 * L * L * H) + 2.0 * H * H + L * (2.0 * H * H + 2.0 * H * V) + 4.0 * H


def cpu_baseline(cfgname, steps=3):
    """The oracle's training step (forward + both CE + autograd backward + HF AdamW, dropout on like
    the reference's train mode) on the host cores; B = 2."""
    import torch
    from oracle import cxrbert_oracle as O
    from oracle import synth
    c = CONFIGS[cfgname]
    ocfg = O.OracleConfig(max_pos=c["max_pos"])
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("MV_CPU_THREADS", "16"))))    # the GPU box's CPU share is 16 per GPU
    torch.set_num_threads(cores)
    P = {k: v.clone().requires_grad_(True) for k, v in O.make_params(ocfg, seed=3).items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    B = 2
    b = {k: torch.from_numpy(v) for k, v in synth.make_batch(ocfg, B, c["N"], c["S"], c["family"], seed=1234).items()}
    t00 = time.time()
    O.train_step(P, M, V, 1, ocfg, b, p_drop=0.1, training=True)
    print(f"[cpu_baseline] warm-up step {time.time() - t00:.1f}s on {cores} threads", file=sys.stderr, flush=True)
    t0 = time.time()
    for t in range(steps):
        O.train_step(P, M, V, t + 2, ocfg, b, p_drop=0.1, training=True)
    dt = (time.time() - t0) / steps
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(value=B / dt, unit="pairs/s", cores=cores, kind="port",
                sample=f"oracle/cxrbert_oracle.train_step, fp32, dropout 0.1, B={B}, {steps} timed steps after 1 warm-up, "
                       f"same shape ({c['name']}); host CPU: {model}")


HBM_PEAK_GBS = 8000.0           # MI355X HBM3E (MI355X_MICROARCH.md)
RIDGE_FLOP_PER_BYTE = PEAK_BF16_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)      # 312.5: below it a kernel is HBM-bound by the roofline


_PROFILE = None


def committed_profile(case):
    """profiles/r05_kernels.json[case] (profiles/tools/r05_profile_all.sh + r05_condense.py: HIP events, rocprofv3 --kernel-trace --stats and the
    FETCH_SIZE / WRITE_SIZE passes of profiles/tools/dominant.py <case>, all in ONE lease on one box) or None."""
    global _PROFILE
    if _PROFILE is None:
        try:
            with open(os.path.join(ROOT, "profiles", "r05_kernels.json")) as f:
                _PROFILE = json.load(f)
        except (OSError, ValueError):
            _PROFILE = {}
    return _PROFILE.get(case)


def pmc_traffic(case):
    """HBM bytes per call of a kernel family from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE runs of
    profiles/tools/dominant.py <case>, gfx950 corrections per MI355X_MICROARCH.md): an offline measurement of the same kernel on the same
    operands, not of this run."""
    pr = committed_profile(case)
    if pr and pr.get("hbm_bytes_per_call"):
        return float(pr["hbm_bytes_per_call"]), pr.get("note", "") + " [profiles/r05_kernels.json]"
    for rnd in ("r04", "r03"):
        p = os.path.join(ROOT, "profiles", f"{rnd}_{case}_pmc.json")
        try:
            with open(p) as f:
                d = json.load(f)
            return float(d["hbm_bytes_per_launch"]), d.get("note", "") + f" [profiles/{rnd}_{case}_pmc.json]"
        except (OSError, KeyError, ValueError):
            continue
    return None, "no committed PMC pass for this kernel"


def time_kernel_case(case):
    """One of the step's kernel families alone: HIP events on the stream it is launched on, 50 warm-up calls, then 200 timed calls in
    batches of 10 -- MEDIAN batch (min / max beside it) -- on the seeded operands of profiles/tools/dominant.py: the SAME command the
    committed rocprofv3 passes ran (profiles/r05_<case>_kernel_stats.txt, profiles/r05_kernels.json).  `profile` puts that lease's
    rocprofv3 figure beside this run's: the two agree within 5 % on the same box; boxes of the pool differ by more (clock under load)."""
    sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
    import dominant
    m = dominant.time_case(case)
    traffic, note = pmc_traffic(case)
    tfl = m["flop"] / m["ms"] / 1e9
    gbs = m["bytes"] / m["ms"] / 1e6
    ai = m["flop"] / m["bytes"]
    bound = "mfma" if ai >= RIDGE_FLOP_PER_BYTE else "hbm"
    pr = committed_profile(case)
    prof = None
    if pr and pr.get("rocprof_call_us_median_after_warmup"):
        ref_us = float(pr["rocprof_call_us_median_after_warmup"])
        prof = dict(file=f"profiles/r05_{case}_kernel_stats.txt (+ profiles/r05_kernels.json)",
                    rocprof_us_median_after_warmup=ref_us, rocprof_us_avg_after_warmup=pr.get("rocprof_call_us_avg_after_warmup"),
                    rocprof_us_avg_all_launches=pr.get("rocprof_call_us_avg_all_launches"),
                    hip_event_us_median_same_lease=pr.get("hip_event_median_us"),
                    tflops_from_profile=m["flop"] / ref_us / 1e6, frac_from_profile=m["flop"] / ref_us / 1e6 / PEAK_BF16_TFLOPS,
                    this_run_over_profile=m["ms"] * 1e3 / ref_us, agrees_within_5pct=abs(m["ms"] * 1e3 / ref_us - 1.0) <= 0.05)
    n = dominant.LAUNCHES_PER_STEP.get(case) or dominant.GEMM_FAMILY[case][1]
    return dict(name=case, kernel=m["kernel"], ms=m["ms"], us=m["ms"] * 1e3, us_min=m["ms_min"] * 1e3, us_max=m["ms_max"] * 1e3,
                timing=f"HIP events, median of {m['timed_calls'] // 10} batches of 10 calls after {m['warmup_calls']} warm-up calls",
                tflops=tfl, algorithmic_flop=m["flop"], algorithmic_bytes=m["bytes"],
                arithmetic_intensity_flop_per_byte=ai, bound=bound, mfma_frac=tfl / PEAK_BF16_TFLOPS, frac=tfl / PEAK_BF16_TFLOPS,
                hbm_frac=gbs / HBM_PEAK_GBS, algorithmic_gb_per_s=gbs, hbm_traffic_pmc_bytes=traffic,
                traffic_over_algorithmic=(traffic / m["bytes"]) if traffic else None, traffic_note=note,
                launches_per_step=n, ms_per_step_alone=n * m["ms"], profile=prof)


def drop_in_user_timings(mv, model, step, cfg, dev, B, N, S, n):
    """(i) CXRBERT_Trainer._run_epoch over a host-side loader of the reference's 9-tuples (dataset_origin.py:181: CPU tensors, int64
    [B,L,L] masks of 134 MB per batch) -- H2D copies, mask recognition and its sampled verification included, per step;
    (ii) the literal model API: CXRBERT.forward() -> [B,L,V] f32 logits -> torch CrossEntropyLoss -> backward() -> fused AdamW, B = 16."""
    import contextlib
    import time
    from types import SimpleNamespace
    import torch
    out = {}
    with contextlib.redirect_stdout(sys.stderr):       # the trainer prints like the reference's; stdout carries the ONE JSON line only
        _drop_in_user_timings(mv, model, step, cfg, dev, B, N, S, n, out, time, SimpleNamespace, torch)
    return out


def _drop_in_user_timings(mv, model, step, cfg, dev, B, N, S, n, out, time, SimpleNamespace, torch):
    args_t = SimpleNamespace(with_cuda=True, weight_load=False, bert_model="bert-base-scratch", lr=1e-5, log_freq=10, mlm_task=True,
                             itm_task=True, cuda_devices=None, dropout_prob=0.1)
    tr = None
    for fam in ("full", "bar"):
        host = []
        for i in range(2):
            b = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, fam, seed=4321 + i, device="cpu")
            host.append((b["cls_tok"], b["input_txt"], b["txt_labels"], b["attn_mask"], (b["img_feats"], b["img_pos"]), b["segment"],
                         b["is_aligned"], b["sep_tok"], torch.zeros(B)))
        if tr is None:
            torch.manual_seed(1234)
            tr = mv.CXRBERT_Trainer(args_t, host * 2, None, config=cfg)
            tr.model.train()
        tr._run_epoch(host * 2, 0, True)                # warm-up: workspaces, the first batches' every-entry mask checks
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 4 * (n + 1)                              # one epoch of 2 * reps steps: its first / last step (pipeline fill, counter read-back) amortised
        tr._run_epoch(host * reps, 0, True)
        torch.cuda.synchronize()
        out[f"trainer_hostloader_{fam}_ms_per_step"] = (time.perf_counter() - t0) / (2 * reps) * 1e3
    del tr
    b16 = mv.data.synthetic_batch(cfg.vocab_size, 16, N, S, "full", seed=999, device=dev)
    ce_m, ce_i = torch.nn.CrossEntropyLoss(ignore_index=-100), torch.nn.CrossEntropyLoss()
    eng = model.engine

    def one(t):
        mlm, itm = model(b16["cls_tok"], b16["input_txt"], b16["attn_mask"], b16["segment"], (b16["img_feats"], b16["img_pos"]), b16["sep_tok"])
        loss = ce_m(mlm.transpose(1, 2), b16["txt_labels"]) + ce_i(itm, b16["is_aligned"])
        model.zero_grad()                           # optim.zero_grad() of train_origin.py:129
        loss.backward()
        eng.adamw_step(t, lr=1e-5)
    step.sync()
    one(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        one(t + 2)
    torch.cuda.synchronize()
    out["dropin_forward_backward_b16_ms"] = (time.perf_counter() - t0) / n * 1e3
    # (iii) the same loop with ONE line changed: `loss = medvill_amd.losses.mlm_itm_loss(mlm, itm, labels, aligned)` under
    # model.lazy_logits = True (the MLM head runs on the labelled rows inside the loss; no [B,L,V] tensor): B = 16 and B = 64
    model.lazy_logits = True
    # ... and with the labels also handed to the forward (`model(..., txt_labels=txt_labels)`: last layer on the consumed rows only), B = 64
    for bsz, key in ((16, "dropin_lazy_loss_b16_ms"), (64, "dropin_lazy_loss_b64_ms"), (64, "dropin_lazy_loss_labels_in_forward_b64_ms")):
        bl = mv.data.synthetic_batch(cfg.vocab_size, bsz, N, S, "full", seed=998, device=dev)
        kw = {"txt_labels": bl["txt_labels"]} if "labels_in_forward" in key else {}

        def one_lazy(t):
            mlm, itm = model(bl["cls_tok"], bl["input_txt"], bl["attn_mask"], bl["segment"], (bl["img_feats"], bl["img_pos"]), bl["sep_tok"], **kw)
            loss = mv.losses.mlm_itm_loss(mlm, itm, bl["txt_labels"], bl["is_aligned"])
            model.zero_grad()                       # optim.zero_grad() of train_origin.py:129
            loss.backward()
            eng.adamw_step(t, lr=1e-5)
        one_lazy(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(n):
            one_lazy(t + 2)
        torch.cuda.synchronize()
        out[key] = (time.perf_counter() - t0) / n * 1e3
    model.lazy_logits = False
    return out


def spawn_ranks(args):
    """--gpus N > 1 without a torch.distributed environment: start the N ranks as a child job (never exec: nothing in
    this process has touched the GPU, and it stays that way) and relay its output."""
    port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU mini-batch")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary timings (padded / BAR / non-cross / full-length / bf16 operands)")
    ap.add_argument("--no-kernel-table", action="store_true", help="(internal) skip the stand-alone timing of the other kernel families")
    ap.add_argument("--dropin-only", action="store_true", help="(internal) print the drop-in user timings as one JSON line and exit: bench.py runs this "
                    "as a CHILD process, so that the trainer is measured in a process of its own like a user's (in a process that already holds "
                    "other models' streams the trainer's two streams can land on one hardware queue)")
    ap.add_argument("--fwd-operand", default=None, choices=["f16", "bf16"], help="encoding of the forward MFMA operands (default f16)")
    ap.add_argument("--grad-operand", default=None, choices=["f16", "bf16"], help="encoding of the gradient-product operands (default: f16 "
                    "under a loss scale with f16 forward operands, else bf16)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # before the HIP runtime starts: see medvill_amd/__init__.py.  Several ranks time-slicing ONE GPU (the gloo rehearsal) keep the
    # default pool: two processes with 8 hardware queues each oversubscribe the device's queue slots (7.7 s per step measured)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if os.environ.get("MV_SINGLE_DEVICE") == "1" else "8")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL (the hosts of this pool support nothing else)
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")            # kernel arguments in device memory: see medvill_amd/__init__.py (-0.3 ms per step)
    # each rank on the CPUs of its GPU's NUMA node, before anything touches the GPU (multi-GPU hosts: two sockets); importing the package
    # starts no HIP runtime
    import medvill_amd.dist as mvdist
    numa = mvdist.bind_to_gpu_numa(0 if os.environ.get("MV_SINGLE_DEVICE") == "1" else local_rank) if world > 1 or os.environ.get("MV_NUMA_BIND") == "1" \
        else {"bound": False, "why": "single-GPU run"}
    import torch
    # rehearsal hooks for a 1-GPU box: MV_DIST_BACKEND=gloo MV_SINGLE_DEVICE=1 run every rank on cuda:0 over gloo
    if os.environ.get("MV_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    rccl_ranks = 1
    # MV_DP_FORCE=1 under a one-rank launcher (torch.distributed.run --nproc-per-node 1): rehearse the RCCL branch on a one-GPU
    # box -- the process group is built and every collective of the step is issued over it (see dist.GradAllReducer)
    dist_on = world > 1 or (os.environ.get("MV_DP_FORCE") == "1" and "RANK" in os.environ)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MV_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                                       # the communicator really spans `world` ranks
        rccl_ranks = int(round(float(probe[0])))
        assert rccl_ranks == dist.get_world_size() == world

    import medvill_amd as mv
    c = CONFIGS[args.config]
    cfg = mv.ModelConfig(max_pos=c["max_pos"])
    torch.manual_seed(1234)                                 # identical init on every rank (checked by TrainStep's checksum)
    model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev, fwd_operand=args.fwd_operand, grad_operand=args.grad_operand)
    model.train()                                           # dropout 0.1 at every site, like the reference's train()
    step = mv.TrainStep(model, lr=1e-5, distributed=dist_on, overlap_optimizer=True)      # as CXRBERT_Trainer builds it
    B, N, S = args.batch, c["N"], c["S"]
    L = N + S + 3
    if args.dropin_only:
        print(json.dumps(drop_in_user_timings(mv, model, step, cfg, dev, B, N, S, max(2, min(args.steps, 5)))), flush=True)
        return

    def make_batches(family, lengths=None, n=4):
        # a few distinct resident batches so that successive steps do not see identical data
        return [mv.data.synthetic_batch(cfg.vocab_size, B, N, S, family, seed=1234 + 1000 * i + rank, device=dev, lengths=lengths)
                for i in range(n)]
    batches = make_batches(c["family"])

    def sync():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(st, bs, warm, n):
        """ms per step of `n` steps after `warm` untimed ones; barrier + device sync on both sides, max over ranks."""
        for i in range(warm):
            st(bs[i % len(bs)])
        sync()
        t0 = time.perf_counter()
        out = None
        for i in range(n):
            out = st(bs[i % len(bs)])
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t[0])
        return dt / n * 1e3, out

    step.time_exchange = dist_on
    ms_per_step, stats = timed(step, batches, args.warmup, args.steps)
    print(f"[bench] rank {rank}: {ms_per_step:.2f} ms/step", file=sys.stderr, flush=True)
    value = world * B / (ms_per_step / 1e3)
    st = stats.cpu()
    exposed_ms = step.exchange_exposed_ms() if dist_on else 0.0
    timeline = step.exchange_timeline() if dist_on else None
    rank_env = mvdist.rank_environment({"numa": numa}) if dist_on else [dict(mvdist.rank_environment({"numa": numa})[0])]
    packed = model.engine.S.get("cu") is not None
    tq_on = model.engine.S.get("tq") is not None          # last layer's attention ran with the consumed rows as its only queries

    extras = {}
    if not args.no_extras:
        n2 = max(2, min(args.steps, 5))
        if packed:
            # the same step with padding removal switched off (every padded position computed, like the reference does)
            step.pack_rows = False
            extras["padded_ms_per_step"], _ = timed(step, batches, 2, n2)
            step.pack_rows = True
        if args.config == "c2":
            # every sample at full length (vl = L): the headline's rows are ragged, these are not
            extras["full_length_ms_per_step"], _ = timed(step, make_batches(c["family"], lengths=[S] * B, n=2), 2, n2)
            # the reference's DEFAULT mask (BAR, main_origin.py:91) and config 4's non-cross mask: padding is visible in
            # both, so they always run the padded layout
            extras["bar_ms_per_step"], _ = timed(step, make_batches("bar", n=2), 2, n2)
            extras["noncross_ms_per_step"], _ = timed(step, make_batches("noncross", n=2), 2, n2)
            # BASELINE.json configs 3 and 5 on this build, so that they are witnessed by the driver's run too
            extras["c3_ms_per_step"], _ = timed(step, make_batches("mixed", n=2), 2, n2)
            if world == 1:
                import subprocess
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dropin-only", "--batch", str(B), "--steps", str(args.steps),
                                        "--config", args.config], capture_output=True, text=True, timeout=900)
                    extras.update(json.loads(r.stdout.strip().splitlines()[-1]))
                except Exception as e:          # the secondary figure must never cost the headline line
                    print(f"[bench] drop-in timings unavailable: {e!r}", file=sys.stderr, flush=True)
                # the data-parallel code path on this one GPU: a one-rank RCCL group with every collective issued (MV_DP_FORCE=1), as a child job
                # under torch.distributed.run -- a regression of the exchange's stream handling shows here before a multi-GPU node sees it
                try:
                    env = dict(os.environ, MV_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
                    port = 29500 + (os.getpid() + 7) % 2000
                    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                                        "--master-port", str(port), os.path.abspath(__file__), "--gpus", "1", "--batch", str(B), "--steps", str(args.steps),
                                        "--warmup", str(args.warmup), "--config", args.config, "--no-extras", "--no-cpu-baseline", "--no-kernel-table"],
                                       capture_output=True, text=True, timeout=900, env=env)
                    d_ = json.loads(r.stdout.strip().splitlines()[-1])
                    if d_["config"].get("dist_backend") == "nccl":
                        extras["rccl_rehearsal_ms_per_step"] = d_["ms_per_step"]
                        extras["rccl_rehearsal_exposed_ms"] = d_["config"].get("allreduce_exposed_ms")
                        extras["rccl_rehearsal_timeline"] = d_["config"].get("allreduce_timeline")
                except Exception as e:
                    print(f"[bench] one-rank RCCL rehearsal unavailable: {e!r}", file=sys.stderr, flush=True)
                c5 = CONFIGS["c5"]
                cfg5 = mv.ModelConfig(max_pos=c5["max_pos"])
                torch.manual_seed(1234)
                m5 = mv.CXRBERT(cfg5, None, dtype=torch.bfloat16, device=dev, fwd_operand=args.fwd_operand, grad_operand=args.grad_operand)
                m5.train()
                st5 = mv.TrainStep(m5, lr=1e-5, overlap_optimizer=True)
                b5 = [mv.data.synthetic_batch(cfg5.vocab_size, B, c5["N"], c5["S"], c5["family"], seed=77 + i, device=dev) for i in range(2)]
                extras["c5_ms_per_step"], _ = timed(st5, b5, 2, n2)
                del m5, st5, b5
    if rank == 0:
        f_fwd = flops_fwd_per_sample(cfg.hidden, cfg.intermediate, cfg.vocab_size, cfg.img_hidden, cfg.layers, L, N)
        f_step = 3.0 * f_fwd
        dense = value / world * f_step / 1e12
        # executed FLOPs: the MLM head runs on the labelled rows only (unlabelled rows have zero loss and gradient), and
        # with padding removal the encoder runs on the valid rows only (positions after the text [SEP] are invisible to
        # every valid query in the full / seq2seq families and carry no label)
        n_lab = float(st[1]) / B
        Hh, Ii, Vv = cfg.hidden, cfg.intermediate, cfg.vocab_size
        vls = torch.cat([b_["attn_desc"].host_desc()[:, 2] for b_ in batches]).double() if packed else torch.full((1,), float(L)).double()
        rows_mean = float(vls.mean())
        f_enc = cfg.layers * (rows_mean * (8.0 * Hh * Hh + 4.0 * Hh * Ii) + 4.0 * float((vls * vls).mean()) * Hh)
        if step.tail_rows:
            # the last layer's output projection and FFN run on the consumed rows only (labelled rows + one [CLS] row per sample)
            f_enc -= (rows_mean - (n_lab + 1.0)) * (2.0 * Hh * Hh + 4.0 * Hh * Ii)
        if tq_on:
            # ... and its attention scores / context only for those rows as queries (every row stays a key)
            f_enc -= 4.0 * (float((vls * vls).mean()) - (n_lab + 1.0) * rows_mean) * Hh
        f_exec = 3.0 * (2.0 * N * cfg.img_hidden * Hh + f_enc + 2.0 * Hh * Hh + n_lab * (2.0 * Hh * Hh + 2.0 * Hh * Vv) + 4.0 * Hh)
        executed = value / world * f_exec / 1e12
        # dominant kernel: the FFN-up GEMM with its fused epilogue (largest single launch of the step; its symbol, the 256x256 ring
        # kernel, and the persistent weight-gradient kernel each take ~20 % of the step's kernel time: profiles/r03_bench_kernel_stats.txt).
        # The weight-gradient GEMM is reported beside it: its call is two kernels (split-K partials + reduction), timed together.
        if os.environ.get("MV_BENCH_NO_KERNELS") == "1":      # (profiles/tools/r05_profile_all.sh: kernel statistics of the STEP alone)
            print(json.dumps({"metric": "image-text pairs/sec pretraining step, BERT-base seq512", "value": value, "unit": "pairs/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "note": "MV_BENCH_NO_KERNELS=1: no roofline objects"}), flush=True)
            return
        kern = time_kernel_case("ffn1")
        kern_dw = time_kernel_case("dw")
        # ... and every other MFMA kernel family of the step, each alone on the step's shapes (profiles/tools/dominant.py): attention (one
        # layer's forward + dQ + dK/dV) and the ten other GEMM calls of a layer.  `coverage` = their stand-alone times x launches per step
        # over the measured step (the two streams overlap, so the sum can exceed 1)
        sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
        import dominant
        kernels = [kern, kern_dw] + ([] if args.no_kernel_table else [time_kernel_case("attn")] + [time_kernel_case(c_) for c_ in dominant.GEMM_FAMILY])
        alone_ms = sum(k_["ms_per_step_alone"] for k_ in kernels)
        eng = model.engine
        pps = lambda ms: (world * B / (ms / 1e3)) if ms else None
        out = {
            "metric": "image-text pairs/sec pretraining step, BERT-base seq512", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16/bf16" if eng.dual else ("f16" if eng.fadt == torch.float16 else "bf16"), "data": "synthetic",
            "config": {"workload": c["name"], "per_gpu_batch": B, "global_batch": B * world, "seq_len": L, "regions": N,
                       "mask": c["family"], "layers": cfg.layers, "hidden": cfg.hidden, "vocab": cfg.vocab_size,
                       "parallelism": f"dp{world}", "optimizer": "HF AdamW fused", "dropout": cfg.dropout,
                       "dropout_realised": {"hidden_state_sites": 6554 / 65536, "attention_probabilities": mv.hip_ops.attn_drop_prob(cfg.dropout)},
                       "precision": ("16-bit MFMA operands, fp32 accumulate / statistics / master weights / optimizer; forward operands "
                                     "f16-encoded, gradient-product operands " + ("bf16-encoded (every stored activation in both encodings)" if eng.dual
                                                                                 else "f16-encoded under a dynamic loss scale (device-side overflow check, "
                                                                                      "overflowed steps skipped)")
                                     + "; encoder LayerNorm inputs (residual sums) stored "
                                     + ("f16" if eng.ln_in_16 else "fp32") + " (BERT-base logits " + ("3.8-4.6e-3" if eng.ln_in_16 else "3.2-3.4e-3")
                                     + " max-abs from the reference, tolerance 1e-2; bf16-encoded forward operands give 2.6e-2: "
                                     "profiles/r02_bf16_error.txt)") if eng.fadt == torch.float16
                       else "16-bit MFMA operands, all bf16-encoded",
                       "loss_scale": (float(eng.scaler[0]) if eng.scaler is not None else None),
                       "steps_skipped_by_overflow": (int(eng.scaler[5]) if eng.scaler is not None else None),
                       "optimizer_steps_applied": (int(eng.scaler[4]) if eng.scaler is not None else None),
                       "rows": (f"padding removed: encoder on the valid rows only (mean {rows_mean:.1f} of {L} positions per sample; "
                                "results equal the padded run)") if packed else "padded",
                       "last_layer": "output projection / FFN / LayerNorms of the last layer on the consumed rows only (labelled rows + "
                                     "[CLS] rows; the other rows' outputs are unused and their gradients exactly zero)"
                                     + ("; its attention takes those rows as the only queries (rows reordered with them first, per-sample query limits)"
                                        if tq_on else "") if step.tail_rows else "all rows",
                       "padded_ms_per_step": extras.get("padded_ms_per_step"), "padded_pairs_per_s": pps(extras.get("padded_ms_per_step")),
                       "full_length_ms_per_step": extras.get("full_length_ms_per_step"),
                       "full_length_pairs_per_s": pps(extras.get("full_length_ms_per_step")),
                       "bar_mask_ms_per_step": extras.get("bar_ms_per_step"), "bar_mask_pairs_per_s": pps(extras.get("bar_ms_per_step")),
                       "noncross_mask_ms_per_step": extras.get("noncross_ms_per_step"),
                       "noncross_mask_pairs_per_s": pps(extras.get("noncross_ms_per_step")),
                       "c3_mixed_mask_ms_per_step": extras.get("c3_ms_per_step"), "c3_pairs_per_s": pps(extras.get("c3_ms_per_step")),
                       "c5_L768_s2s_ms_per_step": extras.get("c5_ms_per_step"), "c5_pairs_per_s": pps(extras.get("c5_ms_per_step")),
                       # what a user of the reference's own interfaces gets (host-resident 9-tuples with int64 [B,L,L] masks through
                       # CXRBERT_Trainer; CXRBERT.forward + torch cross-entropy + backward at B = 16): see drop_in_user_timings
                       "trainer_hostloader_full_ms_per_step": extras.get("trainer_hostloader_full_ms_per_step"),
                       "trainer_hostloader_bar_ms_per_step": extras.get("trainer_hostloader_bar_ms_per_step"),
                       "dropin_forward_backward_b16_ms": extras.get("dropin_forward_backward_b16_ms"),
                       # ... and with the two CrossEntropyLoss calls replaced by medvill_amd.losses.mlm_itm_loss under model.lazy_logits
                       "dropin_lazy_loss_b16_ms": extras.get("dropin_lazy_loss_b16_ms"), "dropin_lazy_loss_b64_ms": extras.get("dropin_lazy_loss_b64_ms"),
                       "dropin_lazy_loss_labels_in_forward_b64_ms": extras.get("dropin_lazy_loss_labels_in_forward_b64_ms"),
                       "rccl_ranks": rccl_ranks, "dist_backend": backend, "allreduce_exposed_ms": exposed_ms,
                       # rank 0's view of the gradient exchange: when each bucket's all-reduce was issued / completed relative to the first
                       # bucket's issue, and when the compute stream waited for the rest (None undistributed)
                       "allreduce_timeline": timeline,
                       # GPU_MAX_HW_QUEUES / HSA_ENABLE_IPC_MODE_LEGACY / ... and the NUMA binding AS SEEN BY EVERY RANK
                       "rank_env": rank_env,
                       # the same step as a one-rank RCCL job with every collective issued (child process); at 8 ranks the model of DESIGN.md 7
                       # expects 0.5-0.6 ms of exposed all-reduce (the embeddings bucket) on top
                       "rccl_rehearsal_ms_per_step": extras.get("rccl_rehearsal_ms_per_step"),
                       "rccl_rehearsal_exposed_ms": extras.get("rccl_rehearsal_exposed_ms"),
                       "rccl_rehearsal_timeline": extras.get("rccl_rehearsal_timeline"),
                       "expected_exposed_allreduce_ms_at_8_ranks": "0.5-0.6 (model, DESIGN.md 7; unmeasured)",
                       "mlm_loss": float(st[0] / max(float(st[1]), 1.0)), "itm_loss": float(st[3] / max(float(st[4]), 1.0))},
            # dominant kernel (the FFN-up GEMM, largest single share of the step): algorithmic FLOPs per launch / its
            # median duration, HIP events over 200 launches after 50 warm-up launches on its own stream (time_kernel_case); traffic = HBM
            # bytes per launch from the committed rocprofv3 PMC passes of the same kernel and shape (profiles/), not of this run
            "roofline": {"bound": kern["bound"],
                         "achieved": kern["tflops"] if kern["bound"] == "mfma" else kern["algorithmic_gb_per_s"],
                         "peak": PEAK_BF16_TFLOPS if kern["bound"] == "mfma" else HBM_PEAK_GBS,
                         "unit": "TFLOP/s" if kern["bound"] == "mfma" else "GB/s",
                         "frac": kern["mfma_frac"] if kern["bound"] == "mfma" else kern["hbm_frac"],
                         "traffic": kern["hbm_traffic_pmc_bytes"],
                         # the committed rocprofv3 pass of the same command (same-lease HIP events beside it): frac recomputed from it
                         "profile": kern["profile"],
                         "ridge_flop_per_byte": RIDGE_FLOP_PER_BYTE,
                         "kernel": kern, "weight_gradient_gemm": kern_dw, "attention": kernels[2] if len(kernels) > 2 else None,
                         "kernels": kernels,
                         "coverage": {"stand_alone_ms_per_step": alone_ms, "of_step": alone_ms / ms_per_step,
                                      "note": "sum over the listed kernel families of (stand-alone time x launches per step) / measured step; the last "
                                              "layer's calls run on fewer rows and the weight-gradient GEMMs overlap the main chain on a second stream"},
                         "step": {"achieved": executed, "frac": executed / PEAK_BF16_TFLOPS,
                                  "convention": "EXECUTED FLOPs of the whole step (valid rows, labelled rows) / dense 16-bit MFMA peak",
                                  "dense_tflops": dense, "dense_frac": dense / PEAK_BF16_TFLOPS, "flop_per_sample_dense": f_step,
                                  "dense_convention": "dense model FLOPs, backward = 2 x forward (SURVEY 8d), whatever was skipped"}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
