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


def _sparse_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "nejm-brain-to-text_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import b2t_ops as ops
        from rnn_model import GRUDecoder
        from b2t_train_step import GradReducer, bucket_spans
        torch.manual_seed(0)
        F, D = 16, 9
        m = GRUDecoder(F, 32, D, 41, 0, 0, 2, 0, 0)
        lay = m.layout()
        names = lay["names"]
        buckets = bucket_spans(lay, 2)
        seg = torch.tensor([names.index(f"day_weights.{d}") for d in range(D)], dtype=torch.int64)
        w0, b0 = lay["spans"][names.index("day_weights.0")][0], lay["spans"][names.index("day_biases.0")][0]
        ws, bs = ops.pad_to(F * F), ops.pad_to(F)
        res = {}
        for case, seen, cap in (("disjoint", {0: {1, 7}, 1: {3, 4}}, 4), ("overlap", {0: {0, 8}, 1: {8, 2}}, 4), ("overflow", {0: {0, 1, 2}, 1: {3, 4, 5}}, 4)):
            arenas = []
            for r in range(world):      # what each rank would hold: gradients only in the day records it saw (the rest zeroed)
                a = torch.randn(lay["total"], generator=torch.Generator().manual_seed(7 + r))
                for d in range(D):
                    if d not in seen[r]:
                        a[w0 + d * ws: w0 + (d + 1) * ws] = 0; a[b0 + d * bs: b0 + (d + 1) * bs] = 0
                arenas.append(a)
            arena = arenas[rank].clone()
            active = torch.ones(len(names), dtype=torch.int32)
            for s_, n in enumerate(names):
                if n.startswith("day_"):
                    active[s_] = int(int(n.split(".")[1]) in seen[rank])
            status = torch.zeros(1)
            red = GradReducer(arena, buckets)
            red.set_sparse_days(active, seg, w0, ws, b0, bs, D, cap, status)
            red.union_active(active)
            for name, _, _ in buckets:
                red.launch(name)
            red.finish()
            want = sum(arenas)
            res[case] = (bool(torch.allclose(arena, want, atol=1e-6)), float(status[0]), tuple(red.sparse["stage_w"].shape), red.n_collectives)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_sparse_day_reduction_world2():
    """Round 4 (verdict item 7b): only the union of the active day layers is all-reduced -- ranks with disjoint and with
    overlapping day sets end with the dense sum; the staging buffer holds `capacity` day records, not all of them; more active
    days than it holds set status 3 (the step is refused) instead of silently dropping a day's gradient."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in out:
        assert res["disjoint"][0] and res["disjoint"][1] == 0.0 and res["disjoint"][2][0] == 4
        # round 6: the five buckets of this two-layer model (head, layer1, layer0, day, h0) leave as THREE collectives
        # (head + layer1 | layer0 | day records + h0)
        assert res["disjoint"][3] == 3, res["disjoint"][3]
        assert res["overlap"][0] and res["overlap"][1] == 0.0
        assert res["overflow"][1] == 3.0                                 # six active days, four slots: the step is refused ...
    assert not all(res["overflow"][0] for _, res in out)                  # ... because a rank would otherwise miss a day's gradient


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


def _loop_worker(rank, world, port, q):
    """The trainer's data-parallel loop control (rnn_trainer.py: rank_batches + broadcast_val_metrics + the early-stopping
    rule of train()) with a stub step: one all-reduce per global step (a rank that took a different number of steps, or
    broke out alone, would hang here), validation on rank 0 only."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "nejm-brain-to-text_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from rnn_trainer import rank_batches, broadcast_val_metrics
        n_batches, val_every, patience = 23, 2, 2          # 23 batches over 2 ranks: 11 global steps, one batch dropped
        mine = list(rank_batches(n_batches, world, rank))
        last_step = n_batches // world - 1
        pers = iter([0.9, 0.5, 0.6, 0.7, 0.1, 0.1])         # improvement stops after the 2nd validation -> stop at the 4th
        best, since, stopped_at, steps, vals = float("inf"), 0, None, 0, []
        for i, b in enumerate(mine):
            t = torch.tensor([float(b)])
            dist.all_reduce(t)                              # the step's gradient exchange
            steps += 1
            assert float(t) == sum(range(i * world, (i + 1) * world))
            if i % val_every == 0 or i == last_step:
                vm = dict(avg_PER=next(pers), avg_loss=1.0, day_PERs={}) if rank == 0 else None   # only rank 0 validates
                vm = broadcast_val_metrics(vm, rank == 0, world, torch.device("cpu"))
                vals.append(vm["avg_PER"])
                if vm["avg_PER"] < best:
                    best, since = vm["avg_PER"], 0
                else:
                    since += 1
                if since >= patience:
                    stopped_at = i
                    break
            if i >= last_step:
                break
        q.put((rank, len(mine), steps, stopped_at, best, vals))
    finally:
        dist.destroy_process_group()


def test_trainer_loop_control_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, s0, stop0, best0, v0), (r1, n1, s1, stop1, best1, v1) = res
    assert n0 == n1 == 11
    assert s0 == s1 == 7 and stop0 == stop1 == 6           # validations at steps 0,2,4,6: 0.9, 0.5, 0.6, 0.7 -> stop
    assert best0 == best1 == 0.5 and v0 == v1 == [0.9, 0.5, 0.6, 0.7]


def test_bench_rendezvous_path_world2():
    """bench.py under `python -m torch.distributed.run` exactly as the driver launches it for N > 1 (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the environment, --gpus N must equal WORLD_SIZE, max-over-ranks of the time, rank 0 prints the
    one JSON line) -- with --dry-run, which swaps RCCL for gloo and the step for a sleep: the plumbing, not a measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # only rank 0 prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["config"]["global_batch"] == 128 and out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak"
    assert out["ms_per_step"] >= 20.0                      # rank 1 sleeps 20 ms per "step": the MAX over ranks is reported
    # a mismatch between --gpus and the launched world is an error, not a silent single-GPU run
    cmd[cmd.index("--gpus") + 1] = "4"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_spawns_its_own_ranks(scaling):
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment (the driver's N > 1 form, VERDICT
    round 2 weak #4): bench.py starts the two ranks itself (rendezvous on 127.0.0.1, free port), still prints exactly one
    JSON line, reports the world size the process group saw, and `--scaling strong` shards the one 64-sentence batch."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run",
           "--scaling", scaling]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_seen"] == 2 and out["scaling"] == scaling
    assert out["config"]["global_batch"] == (128 if scaling == "weak" else 64)
    assert out["config"]["rows_per_rank"] == (64 if scaling == "weak" else 32)
    assert out["ms_per_step"] >= 20.0


def test_bench_spawn_propagates_a_rank_failure():
    """A rank that dies must end the job with a non-zero code (and take the other ranks with it), not hang the driver."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["B2T_BENCH_DRY_FAIL_RANK"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=root, env=env)
    assert r.returncode != 0
