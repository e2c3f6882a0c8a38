"""The N > 1 path on CPU: two gloo ranks drive the gradient all-reducer with the engine's bucket
hook order and the global loss normalisation, no GPU needed."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import medvill_amd as mv
from medvill_amd.dist import GradAllReducer


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = mv.ModelConfig(vocab_size=512, hidden=64, layers=3, heads=1, intermediate=128, max_pos=64, img_hidden=64)
    lay, n = mv.param_layout(cfg)
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = GradAllReducer(g, lay, n, cfg.layers, merge_layers=2)
    cnt = red.global_counts(10 + rank, 4, "cpu")
    # the order Engine.encoder_backward reports finished buckets in
    red.hook("heads", None)
    for l in reversed(range(cfg.layers)):
        red.hook(f"layer{l}", None)
    red.hook("embeddings", None)
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = bool(torch.equal(g, expect)) and cnt.tolist() == [float(sum(10 + r for r in range(world))), 4.0 * world]
    # loss normalisation: each rank scales by the global counts; the summed gradient is the global-batch mean
    local_nll = torch.tensor([3.0 * (rank + 1)])
    contrib = local_nll / cnt[0]
    dist.all_reduce(contrib)
    tot_nll, tot_cnt = 3.0 * sum(r + 1 for r in range(world)), float(sum(10 + r for r in range(world)))
    ok = ok and abs(float(contrib) - tot_nll / tot_cnt) < 1e-6
    out[rank] = ok
    dist.destroy_process_group()





@pytest.mark.parametrize("world", [2, 4])
def test_gradient_allreduce_and_global_normalisation(world):
    """world 4: uneven labelled-token counts per rank (10, 11, 12, 13) and the merged two-layer buckets of a 3-layer model."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out[r] for r in range(world))
