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


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    mv, cfg, m = _model(torch.float32)
    full = mv.data.synthetic_batch(cfg.vocab_size, 8, 6, 40, "mixed", seed=5, device="cuda:0")
    sl = slice(rank * 4, rank * 4 + 4)
    half = {k: v[sl] for k, v in full.items() if k not in ("label_rows", "label_ids")}
    ts = mv.TrainStep(m, lr=1e-3, distributed=True)
    for _ in range(2):
        stats = ts(half, train=True)
    torch.cuda.synchronize()
    out[rank] = (m.engine.flat_p.cpu(), stats.cpu())
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    mv, cfg, m = _model(torch.float32)
    full = mv.data.synthetic_batch(cfg.vocab_size, 8, 6, 40, "mixed", seed=5, device="cuda:0")
    ts = mv.TrainStep(m, lr=1e-3)
    for _ in range(2):
        stats = ts(full, train=True)
    ref = m.engine.flat_p.cpu()
    p0, s0 = out[0]
    p1, s1 = out[1]
    assert torch.equal(p0, p1)                                     # replicas stay identical without any broadcast
    assert float((p0 - ref).abs().max()) < 2e-5                    # 2 AdamW steps of size 1e-3 on the same gradient
    tot = s0 + s1
    assert float(tot[1]) == float(stats[1]) and abs(float(tot[0]) - float(stats[0])) < 1e-2 * float(stats[0])
