"""Data-parallel step on the GPU box: two ranks (gloo transport, both on cuda:0 -- the box has one GPU; RCCL needs
one device per rank) run TrainStep(distributed=True) on the two halves of a global mini-batch; the result must equal
the single-process step on the whole mini-batch (global loss normalisation + summed gradients)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dtype):
    import medvill_amd as mv
    cfg = mv.ModelConfig(vocab_size=2048, hidden=128, layers=3, heads=2, intermediate=512, max_pos=128)
    m = mv.CXRBERT(cfg, None, dtype=dtype, device="cuda:0")
    m.reset_parameters(seed=11)
    m.eval()                      # dropout off: the comparison is between identical functions
    return mv, cfg, m


def _worker(rank, world, port, out, dtype, train_mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    mv, cfg, m = _model(dtype)
    if train_mode:
        m.train()                 # dropout on: every rank draws its own masks, replicas must still stay identical
    full = mv.data.synthetic_batch(cfg.vocab_size, 8, 6, 40, "mixed", seed=5, device="cuda:0")
    sl = slice(rank * 4, rank * 4 + 4)
    half = {k: v[sl] for k, v in full.items() if k not in ("label_rows", "label_ids")}
    ts = mv.TrainStep(m, lr=1e-3, distributed=True)
    for _ in range(2):
        stats = ts(half, train=True)
    ev = ts(half, train=False)            # eval: no collective is issued (ranks may hold different numbers of eval batches)
    torch.cuda.synchronize()
    out[rank] = (m.engine.flat_p.cpu(), stats.cpu(), m.engine.drop_seed, m.engine.S["cu"] is not None, ev.cpu())
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_rank_step_equals_single_process_step(dtype):
    """fp32: exact path, padded rows.  bf16: the 16-bit path with packed rows (mixed full / seq2seq masks)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, dtype, False), nprocs=world, join=True)
    mv, cfg, m = _model(dtype)
    full = mv.data.synthetic_batch(cfg.vocab_size, 8, 6, 40, "mixed", seed=5, device="cuda:0")
    ts = mv.TrainStep(m, lr=1e-3)
    for _ in range(2):
        stats = ts(full, train=True)
    ref = m.engine.flat_p.cpu()
    p0, s0 = out[0][:2]
    p1, s1 = out[1][:2]
    assert torch.equal(p0, p1)                                     # replicas stay identical without any broadcast
    # 2 AdamW steps of size 1e-3 on the same gradient (16-bit path: sign flips of near-zero gradient entries move 2e-3)
    assert float((p0 - ref).abs().max()) < (2e-5 if dtype == torch.float32 else 4.1e-3)
    assert float((p0 - ref).abs().mean()) < (1e-6 if dtype == torch.float32 else 5e-5)
    tot = s0 + s1
    assert float(tot[1]) == float(stats[1]) and abs(float(tot[0]) - float(stats[0])) < 1e-2 * float(stats[0])
    assert out[0][3] == (dtype == torch.bfloat16)                  # packed rows ran under data parallelism


def test_two_rank_training_mode_draws_rank_local_dropout_and_keeps_replicas_identical():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, torch.bfloat16, True), nprocs=world, join=True)
    assert torch.equal(out[0][0], out[1][0]) and bool(torch.isfinite(out[0][0]).all())
    assert out[0][2] != out[1][2]                                  # per-rank dropout keys


def _bad_seed_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import medvill_amd as mv
    cfg = mv.ModelConfig(vocab_size=2048, hidden=128, layers=1, heads=2, intermediate=512, max_pos=128)
    m = mv.CXRBERT(cfg, None, dtype=torch.float32, device="cuda:0")
    m.reset_parameters(seed=11 + rank)                             # a per-rank seed: replicas differ
    try:
        mv.TrainStep(m, lr=1e-3, distributed=True)
        out[rank] = "no error"
    except RuntimeError as e:
        out[rank] = str(e)
    dist.destroy_process_group()


def test_replicas_with_different_parameters_are_refused():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bad_seed_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all("different parameters" in out[r] for r in range(world)), dict(out)


def _rccl_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import medvill_amd as mv
    cfg = mv.ModelConfig(vocab_size=2048, hidden=128, layers=3, heads=2, intermediate=512, max_pos=128)
    m = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=f"cuda:{rank}")
    m.reset_parameters(seed=11)
    m.train()
    full = mv.data.synthetic_batch(cfg.vocab_size, 8, 6, 40, "mixed", seed=5, device=f"cuda:{rank}")
    sl = slice(rank * 4, rank * 4 + 4)
    half = {k: v[sl] for k, v in full.items() if k not in ("label_rows", "label_ids")}
    ts = mv.TrainStep(m, lr=1e-3, distributed=True)
    for _ in range(3):
        stats = ts(half, train=True)
    torch.cuda.synchronize()
    out[rank] = (m.engine.flat_p.cpu(), stats.cpu(), float(ts.exchange_exposed_ms()))
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: runs on nodes with >= 2 GPUs")
def test_two_rank_step_over_rccl():
    """The same two-rank step over the nccl backend (= RCCL over xGMI), one device per rank, bf16 path, dropout on, packed rows,
    bucketed gradient all-reduce on the side stream: replicas must stay bit-identical and finite."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert torch.equal(out[0][0], out[1][0]) and bool(torch.isfinite(out[0][0]).all())
    assert float(out[0][1][1]) > 0


def _rccl_one_rank_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MV_DP_FORCE"] = "1"               # issue every collective although the group has one rank
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import medvill_amd as mv
    cfg = mv.ModelConfig(vocab_size=2048, hidden=128, layers=3, heads=2, intermediate=512, max_pos=128)
    res = []
    for distributed in (True, False):
        torch.manual_seed(3)
        m = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device="cuda:0")
        m.reset_parameters(seed=11)
        m.eval()                      # dropout off: a distributed step draws rank-specific masks, the comparison needs the same function
        ts = mv.TrainStep(m, lr=1e-3, distributed=distributed, overlap_optimizer=True)
        ts.time_exchange = True
        for i in range(3):
            batch = mv.data.synthetic_batch(cfg.vocab_size, 8, 6, 40, "mixed", seed=5 + i, device="cuda:0")
            stats = ts(batch, train=True)
        ts.sync()
        torch.cuda.synchronize()
        res.append((m.engine.flat_p.cpu(), stats.cpu(), float(ts.exchange_exposed_ms()), None if ts.dp is None else len(ts.dp.ranges),
                    ts.exchange_timeline(), mv.dist.rank_environment({"numa": mv.dist.bind_to_gpu_numa(0)}) if distributed else None))
    out[0] = res
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_every_collective():
    """The nccl (= RCCL) code path on a one-GPU box: a one-rank process group with MV_DP_FORCE=1, so that the parameter checksum,
    the per-step count all-reduce and the bucketed gradient all-reduces on the communication stream all go through RCCL.  Summing
    over one rank changes nothing: the parameters must follow the undistributed step (up to the float-atomic noise of the
    gradients, see test_overlapped_optimizer_equals_the_plain_step)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_one_rank_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    (p_d, s_d, exposed, nb, timeline, env), (p_s, s_s, _, _, tl_s, _) = out[0]
    assert bool(torch.isfinite(p_d).all()) and nb is not None and nb >= 3
    # the per-bucket timeline the bench line carries: heads first, the embeddings bucket last, the compute stream
    # passes every bucket's wait after the bucket was issued, and its wait ends after the last bucket was issued
    assert tl_s is None and timeline is not None and timeline["steps"] == 3
    names = [b_["bucket"] for b_ in timeline["buckets"]]
    assert names[0] == "heads" and names[-1] == "embeddings" and len(names) >= 3
    assert all(b_["done_at_ms"] >= b_["issued_at_ms"] >= 0.0 and b_["mbytes"] > 0 for b_ in timeline["buckets"])
    assert abs(sum(b_["mbytes"] for b_ in timeline["buckets"]) - 4e-6 * p_d.numel()) < 1e-3          # the buckets tile the flat gradient
    assert timeline["compute_stream_wait_until_ms"] >= timeline["buckets"][-1]["issued_at_ms"]
    assert len(env) == 1 and env[0]["rank"] == 0 and env[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "bound" in env[0]["numa"]
    d = (p_d - p_s).abs()
    assert float(d.max()) < 3 * 2e-3 + 1e-4 and float(d.mean()) < 5e-5
    assert float(s_d[1]) == float(s_s[1]) and exposed >= 0.0


def _trainer_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from types import SimpleNamespace
    import medvill_amd as mv
    V, B, N, S = 2048, 4, 6, 41
    cfgd = dict(vocab_size=V, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, max_position_embeddings=64)

    def tup(seed, flip=False):
        b = mv.data.synthetic_batch(V, B, N, S, "full", seed=seed, device="cpu")
        m = b["attn_mask"].clone()
        if flip:
            m[2, N + 9, N + 5] ^= 1               # deep inside the matrix: the recognition probes still say "full"
        return (b["cls_tok"], b["input_txt"], b["txt_labels"], m, (b["img_feats"], b["img_pos"]), b["segment"], b["is_aligned"], b["sep_tok"],
                torch.zeros(B))
    train = [tup(100 * rank + i, flip=(rank == 1 and i == 1)) for i in range(3)]
    evals = [tup(900 + 10 * rank + i) for i in range(1 + rank)]             # uneven: rank 0 holds one eval batch, rank 1 two
    args = SimpleNamespace(with_cuda=True, weight_load=False, bert_model="custom", lr=1e-3, log_freq=2, mlm_task=True, itm_task=True,
                           cuda_devices=[0], dropout_prob=0.1)
    torch.manual_seed(3)
    tr = mv.CXRBERT_Trainer(args, train, evals, config=cfgd, dtype=torch.bfloat16)
    for ep in range(2):
        res = tr.train(ep)
    torch.cuda.synchronize()
    out[rank] = (tr.model.engine.flat_p.cpu(), tr.n_recognised, tr.n_rejected, tr.distributed, float(res["avg_loss"]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_one_rank_failing_its_mask_check_changes_no_collective():
    """the mask check's outcome is rank-local and known BEFORE the step is enqueued (host-side every-entry check one
    batch ahead), and the matrix path issues the same collectives as the descriptor path -- so the rank whose batch fails simply runs
    that step on the matrix while its peer runs on descriptors.  Uneven evaluation loaders (no collective in evaluation) and two
    epochs: nothing hangs, the replicas stay identical."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trainer_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (p0, rec0, rej0, d0, l0), (p1, rec1, rej1, d1, l1) = out[0], out[1]
    assert d0 and d1
    assert torch.equal(p0, p1) and bool(torch.isfinite(p0).all())
    assert rej0 == 0 and rej1 == 2                     # the flipped batch, once per epoch
    assert rec0 == 2 * (3 + 1) and rec1 == 2 * (3 + 2)


def _comm_abi_worker(rank, world, path, out):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    import time
    import medvill_amd as mv
    if rank == 0:
        uid = mv.hip_ops.RcclComm.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
    else:
        for _ in range(600):
            if os.path.exists(path):
                break
            time.sleep(0.05)
        uid = open(path, "rb").read()
    comm = mv.hip_ops.RcclComm(rank, world, uid)
    dev = torch.device("cuda", rank)
    side = torch.cuda.Stream(device=dev)
    res = []
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        t = (torch.arange(4096, device=dev, dtype=torch.float32) * (rank + 1) / 64.0).to(dt)
        want = (torch.arange(4096, dtype=torch.float32) / 64.0).to(dt).float() * sum(r + 1 for r in range(world))
        side.wait_stream(torch.cuda.current_stream())
        comm.allreduce_async(t, stream=side)            # a bucket's exchange on the side stream ...
        comm.wait()                                     # ... the compute stream waits on the device before it reads the sum
        res.append(float((t.float().cpu() - want).abs().max() / want.abs().max()))
    torch.cuda.synchronize()
    comm.destroy()
    out[rank] = res


def test_c_abi_rccl_communicator_on_one_rank(tmp_path):
    """mv_comm_* (SURVEY 8b's export list): the C ABI's own RCCL binding, bound at run time.  A one-rank communicator on this one-GPU box
    (the sum over one rank is the identity) exercises id creation, init, asynchronous all-reduce on a side stream, the device-side wait and
    destroy through RCCL itself; the two-rank form below runs wherever two GPUs are visible."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_comm_abi_worker, args=(1, str(tmp_path / "uid0"), out), nprocs=1, join=True)
    assert max(out[0]) < 1e-6, out[0]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: runs on nodes with >= 2 GPUs")
def test_c_abi_rccl_communicator_on_two_ranks(tmp_path):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_comm_abi_worker, args=(2, str(tmp_path / "uid"), out), nprocs=2, join=True)
    assert max(out[0]) < 1e-2 and max(out[1]) < 1e-2, dict(out)


_TWO_ON_ONE = r"""
import os, sys, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, sys.argv[3])
import torch
rank, path = int(sys.argv[1]), sys.argv[2]
torch.cuda.set_device(0)                      # BOTH ranks on the one GPU of the box
import medvill_amd as mv
if rank == 0:
    uid = mv.hip_ops.RcclComm.unique_id()
    open(path + ".tmp", "wb").write(uid)
    os.replace(path + ".tmp", path)
else:
    for _ in range(1200):
        if os.path.exists(path):
            break
        time.sleep(0.05)
    uid = open(path, "rb").read()
try:
    comm = mv.hip_ops.RcclComm(rank, 2, uid)
except RuntimeError as e:
    print("REFUSED", e)
    sys.exit(0)
t = torch.full((1024,), float(rank + 1), device="cuda:0")
comm.allreduce_async(t)
comm.wait()
torch.cuda.synchronize()
print("SUM", float(t[0]), float(t[-1]))
comm.destroy()
"""


@pytest.mark.timeout(300)
def test_c_abi_rccl_two_processes_on_the_one_gpu_enter_the_two_rank_path(tmp_path):
    """the TWO-rank mv_comm_* path on a one-GPU box.  Two processes build a two-rank communicator on the same device.
    RCCL either runs it (then the sum must be 1 + 2) or refuses it at init ("Duplicate GPU detected", ncclInvalidUsage) -- which the C ABI
    must hand back as a negative MV_E_COMM_BASE - ncclResult_t status that medvill_amd raises as `ncclResult_t <n>`: either way
    mv_comm_unique_id / mv_comm_init were entered with world = 2 and nothing hangs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "uid2")
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_ON_ONE, str(r), path, root], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p_ in procs:
        try:
            o, e = p_.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a two-rank communicator on one device neither ran nor was refused within 240 s")
        outs.append((p_.returncode, o.strip().splitlines()[-1] if o.strip() else "", e[-400:]))
    assert all(rc == 0 for rc, _, _ in outs), outs
    kinds = {o.split()[0] for _, o, _ in outs}
    assert kinds in ({"SUM"}, {"REFUSED"}), outs
    if kinds == {"SUM"}:
        assert all(o.split()[1:] == ["3.0", "3.0"] for _, o, _ in outs), outs
    else:
        assert all("ncclResult_t" in o for _, o, _ in outs), outs
