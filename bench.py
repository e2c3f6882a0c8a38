#!/usr/bin/env python3
"""Benchmark of the MedViLL / CXRBERT pretraining step on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One "step" = the loop body of models/train_origin.py:95-146 on one synthetic mini-batch that is
already resident in HBM: forward, CE(mlm, ignore -100) + CE(itm), backward, HF AdamW, the
ITM/MLM accuracy counters, and (N > 1) the RCCL gradient all-reduce.  Workload = BASELINE.json
configs[1]: BERT-base (12L/12H/768), L = 512 (36 regions + 476 text), bidirectional mask, bf16
MFMA path, batch 64 per GPU (weak scaling), random-init weights, synthetic inputs.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     whole-step model FLOPs (dense MFU convention, 364.076 GFLOP/sample)
               against the dense bf16 MFMA peak, plus the dominant kernel (the MFMA GEMM) timed
               alone with HIP events on its own stream;
  cpu_baseline the CPU oracle (a port; the reference's Python cannot travel) timed on the host
               cores on a bounded sample (B = 2, same shape).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md); never the 2:1-sparse figure
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


def time_dominant_kernel(eng, B, L):
    """The FFN-up projection GEMM ([B*L,768] x [3072,768]^T + bias + GELU) alone, HIP events on its stream."""
    import medvill_amd.hip_ops as ops
    from medvill_amd._lib import EPI_BIAS_GELU_D as EPI_BIAS_GELU
    M, H, I = B * L, eng.cfg.hidden, eng.cfg.intermediate
    p = "enc.encoder.layer.0."
    x = eng._buf("x0", (M, H), eng.adt)
    out, z = eng._buf("i0", (M, I), eng.adt), eng._buf("dgelu0", (M, I), eng.adt)
    st = torch.cuda.current_stream()
    reps = 20
    for _ in range(3):
        ops.gemm(x, eng.w[p + "intermediate.dense.weight"], out, M=M, N=I, K=H, bias=eng.p[p + "intermediate.dense.bias"],
                 epi=EPI_BIAS_GELU, c2=z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        ops.gemm(x, eng.w[p + "intermediate.dense.weight"], out, M=M, N=I, K=H, bias=eng.p[p + "intermediate.dense.bias"],
                 epi=EPI_BIAS_GELU, c2=z)
    e1.record(st)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * H * I
    return dict(kernel="gemm_ring_kernel<NT, 256x256x64> 32768x3072x768 +bias+GELU (writes gelu(z) and gelu'(z))", ms=ms,
                tflops=fl / ms / 1e9, algorithmic_flop=fl,
                algorithmic_bytes=2.0 * (M * H + I * H + 2 * M * I),
                hbm_traffic_pmc_bytes=6.24e8,
                traffic_note="rocprofv3 --pmc, separate passes: FETCH_SIZE 110,887 KB x2 (gfx950 wide-read correction) = 221.8 MB "
                             "+ WRITE_SIZE 393,216 KB = 402.7 MB per launch (profiles/r01_gemm_pmc.txt)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU mini-batch")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs a torch.distributed launch with WORLD_SIZE={args.gpus} (got {world})")
    # rehearsal hooks for a 1-GPU box: MV_DIST_BACKEND=gloo MV_SINGLE_DEVICE=1 run every rank on cuda:0 over gloo
    if os.environ.get("MV_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MV_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import medvill_amd as mv
    c = CONFIGS[args.config]
    cfg = mv.ModelConfig(max_pos=c["max_pos"])
    torch.manual_seed(1234)                                 # identical init on every rank: no parameter broadcast
    model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
    model.train()                                           # dropout 0.1 at every site, like the reference's train()
    step = mv.TrainStep(model, lr=1e-5, distributed=(world > 1))
    B, N, S = args.batch, c["N"], c["S"]
    L = N + S + 3
    # a few distinct resident batches so that successive steps do not see identical data
    batches = [mv.data.synthetic_batch(cfg.vocab_size, B, N, S, c["family"], seed=1234 + 1000 * i + rank, device=dev)
               for i in range(4)]

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(batches[i % len(batches)])
    sync()
    print(f"[bench] rank {rank}: warm-up done", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    stats = None
    for i in range(args.steps):
        stats = step(batches[i % len(batches)])
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])
    ms_per_step = dt / args.steps * 1e3
    print(f"[bench] rank {rank}: {ms_per_step:.2f} ms/step", file=sys.stderr, flush=True)
    value = world * B * args.steps / dt
    st = stats.cpu()
    packed = model.engine.S.get("cu") is not None
    padded_ms = None
    if packed:
        # for transparency: the same step with padding removal switched off (every padded position computed, like the
        # reference does), timed after the headline region on the same batches; reported as config.padded_*
        step.pack_rows = False
        n2 = max(2, min(args.steps, 5))
        for i in range(2):
            step(batches[i % len(batches)])
        sync()
        t1 = time.perf_counter()
        for i in range(n2):
            step(batches[i % len(batches)])
        sync()
        d2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([d2], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            d2 = float(t[0])
        padded_ms = d2 / n2 * 1e3
        step.pack_rows = True
    if rank == 0:
        f_fwd = flops_fwd_per_sample(cfg.hidden, cfg.intermediate, cfg.vocab_size, cfg.img_hidden, cfg.layers, L, N)
        f_step = 3.0 * f_fwd
        achieved = value / world * f_step / 1e12
        # executed FLOPs: the MLM head runs on the labelled rows only (unlabelled rows have zero loss and gradient), and
        # with padding removal the encoder runs on the valid rows only (positions after the text [SEP] are invisible to
        # every valid query in the full / seq2seq families and carry no label)
        n_lab = float(st[1]) / B
        Hh, Ii, Vv = cfg.hidden, cfg.intermediate, cfg.vocab_size
        vls = torch.cat([b_["attn_desc"].host_desc()[:, 2] for b_ in batches]).double() if packed else torch.full((1,), float(L)).double()
        rows_mean = float(vls.mean())
        f_enc = cfg.layers * (rows_mean * (8.0 * Hh * Hh + 4.0 * Hh * Ii) + 4.0 * float((vls * vls).mean()) * Hh)
        f_exec = 3.0 * (2.0 * N * cfg.img_hidden * Hh + f_enc + 2.0 * Hh * Hh + n_lab * (2.0 * Hh * Hh + 2.0 * Hh * Vv) + 4.0 * Hh)
        kern = time_dominant_kernel(model.engine, B, L)
        out = {
            "metric": "image-text pairs/sec pretraining step, BERT-base seq512", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": c["name"], "per_gpu_batch": B, "global_batch": B * world, "seq_len": L, "regions": N,
                       "mask": c["family"], "layers": cfg.layers, "hidden": cfg.hidden, "vocab": cfg.vocab_size,
                       "parallelism": f"dp{world}", "optimizer": "HF AdamW fused", "dropout": cfg.dropout,
                       "rows": (f"padding removed: encoder on the valid rows only (mean {rows_mean:.1f} of {L} positions per sample; "
                                "results equal the padded run)") if packed else "padded",
                       "padded_ms_per_step": padded_ms, "padded_pairs_per_s": (world * B / (padded_ms / 1e3)) if padded_ms else None,
                       "mlm_loss": float(st[0] / max(float(st[1]), 1.0)), "itm_loss": float(st[3] / max(float(st[4]), 1.0))},
            # dominant kernel (the FFN-up GEMM, largest single share of the step): algorithmic FLOPs per launch / its
            # average duration, HIP events around 20 launches on its own stream (time_dominant_kernel); traffic = HBM bytes
            # per launch from rocprofv3 PMC passes of the same kernel (profiles/r01_gemm_pmc.txt)
            "roofline": {"bound": "mfma", "achieved": kern["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": kern["tflops"] / PEAK_BF16_TFLOPS, "traffic": kern["hbm_traffic_pmc_bytes"],
                         "kernel": kern,
                         "step": {"achieved": achieved, "frac": achieved / PEAK_BF16_TFLOPS, "flop_per_sample": f_step,
                                  "convention": "dense model FLOPs of the whole step, backward = 2 x forward (SURVEY 8d)",
                                  "executed_tflops": value / world * f_exec / 1e12}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
