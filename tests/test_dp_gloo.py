"""Data-parallel plumbing on CPU: world_size-2 gloo processes exercise the bucketed gradient reducer
(sum over ranks, bucket coverage, union of active days) and the batch sharding rule."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "nejm-brain-to-text_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rnn_model import GRUDecoder
        from b2t_train_step import GradReducer, bucket_spans
        torch.manual_seed(0)
        m = GRUDecoder(16, 32, 5, 41, 0, 0, 2, 0, 0)
        lay = m.layout()
        buckets = bucket_spans(lay, 2)
        g = torch.Generator().manual_seed(100 + rank)
        arena = torch.randn(lay["total"], generator=g)
        mine = arena.clone()
        red = GradReducer(arena, buckets)
        assert red.world == world
        for name, _, _ in buckets:            # launch in backward order, asynchronously
            red.launch(name)
        red.finish()
        others = [torch.randn(lay["total"], generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        want = sum(others)
        ok_sum = bool(torch.allclose(arena, want, atol=1e-6))
        # union of active flags: rank 0 saw days {0,1}, rank 1 saw days {1,4}
        nseg = len(lay["names"])
        active = torch.ones(nseg, dtype=torch.int32)
        seen = {0: {0, 1}, 1: {1, 4}}[rank]
        for s, n in enumerate(lay["names"]):
            if n.startswith("day_"):
                active[s] = int(int(n.split(".")[1]) in seen)
        red.union_active(active)
        got_days = sorted({int(n.split(".")[1]) for s, n in enumerate(lay["names"]) if n.startswith("day_") and active[s]})
        # batch sharding rule of the trainer: batch i goes to rank i % world
        my_batches = [i for i in range(10) if i % world == rank]
        q.put((rank, ok_sum, got_days, my_batches, float((arena - mine).abs().max()) > 0))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_sum, days, mine, changed in res:
        assert ok_sum and changed
        assert days == [0, 1, 4]
    assert sorted(res[0][3] + res[1][3]) == list(range(10)) and not set(res[0][3]) & set(res[1][3])
