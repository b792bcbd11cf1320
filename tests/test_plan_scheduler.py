"""Host logic of the pass scheduler (csrc/exec.cpp: schedule_plan, exported as b2t_plan_schedule_host): the executor
places the task graph of a pass on four in-order queues with this function.  No GPU: the invariants every schedule must
satisfy for the issued stream/event program to be correct, plus a few shapes with a known answer."""
import ctypes as C

import numpy as np
import pytest

import b2t_native as N

HOP = 20.0     # exec.cpp HOP_US


def schedule(est, deps, nq, qmask=None):
    lib = N.load()
    n = len(est)
    est = np.asarray(est, np.float32)
    qm = np.full(n, 0xffffffff, np.uint32) if qmask is None else np.asarray(qmask, np.uint32)
    off = np.zeros(n + 1, np.int32)
    flat = []
    for i, d in enumerate(deps):
        flat += list(d); off[i + 1] = len(flat)
    flat = np.asarray(flat if flat else [0], np.int32)
    q = np.zeros(n, np.int32); start = np.zeros(n, np.float32); order = np.zeros(n, np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    N.check(lib.b2t_plan_schedule_host(n, P(est), P(qm), P(off), P(flat), nq, P(q), P(start), P(order)), "b2t_plan_schedule_host")
    return q, start, order


def check_invariants(est, deps, nq, q, start, order, qmask=None):
    n = len(est)
    end = start + np.asarray(est, np.float32)
    assert sorted(order.tolist()) == list(range(n))                       # a permutation
    pos = np.empty(n, int); pos[order] = np.arange(n)
    for i in range(n):
        assert 0 <= q[i] < nq
        if qmask is not None and nq > 1:
            assert (int(qmask[i]) >> int(q[i])) & 1, f"task {i} on a queue outside its mask"
        for d in deps[i]:
            assert pos[d] < pos[i], "a dependency must be issued first (its event is recorded before it is waited on)"
            assert start[i] + 1e-3 >= end[d] + (HOP if q[d] != q[i] else 0.0)
    for qq in range(nq):                                                   # in-order queues: no overlap, issue order = start order
        ids = [i for i in order if q[i] == qq]
        for a, b in zip(ids, ids[1:]):
            assert start[b] + 1e-3 >= end[a], f"tasks {a} and {b} overlap on queue {qq}"


def test_chain_stays_on_one_queue():
    est = [10.0] * 6
    deps = [[]] + [[i] for i in range(5)]
    q, start, order = schedule(est, deps, 4)
    assert len(set(q.tolist())) == 1 and order.tolist() == list(range(6))
    np.testing.assert_allclose(start, np.arange(6) * 10.0)


def test_fork_join_uses_all_queues():
    est = [0.0] + [100.0] * 4 + [0.0]
    deps = [[]] + [[0]] * 4 + [[1, 2, 3, 4]]
    q, start, order = schedule(est, deps, 4)
    check_invariants(est, deps, 4, q, start, order)
    assert sorted(q[1:5].tolist()) == [0, 1, 2, 3]
    assert start[5] <= 100.0 + 2 * HOP + 1.0                               # one hop out, one hop back (+ the 0.5 us slot of a bookkeeping task)
    q1, s1, _ = schedule(est, deps, 1)
    assert set(q1.tolist()) == {0} and abs(s1[5] - 400.0) < 1.0


def test_pinned_tasks_and_masks():
    est = [0.0, 50.0, 50.0, 50.0, 0.0]
    deps = [[], [0], [0], [0], [1, 2, 3]]
    mask = [1, 0b1110, 0b1110, 0b0010, 1]                                   # start / end on queue 0, workers elsewhere
    q, start, order = schedule(est, deps, 4, mask)
    check_invariants(est, deps, 4, q, start, order, mask)
    assert q[0] == 0 and q[4] == 0 and q[3] == 1 and 0 not in q[1:4].tolist()


def test_gaps_are_filled_by_low_priority_work():
    # a critical chain with a hop-sized hole on its queue and an independent short task: the short task must not extend the makespan
    est = [100.0, 100.0, 100.0, 30.0]
    deps = [[], [0], [1], []]
    q, start, order = schedule(est, deps, 2)
    check_invariants(est, deps, 2, q, start, order)
    assert max(start + np.asarray(est, np.float32)) <= 300.0 + 1e-3


@pytest.mark.parametrize("seed", range(12))
def test_random_dags_satisfy_the_invariants_and_are_deterministic(seed):
    rs = np.random.RandomState(seed)
    n = int(rs.randint(5, 120)); nq = int(rs.randint(1, 6))
    est = rs.choice([0.0, 5.0, 40.0, 170.0, 500.0, 800.0], size=n).astype(np.float32)
    deps = [sorted(set(rs.randint(0, i, size=rs.randint(0, 4)).tolist())) if i else [] for i in range(n)]
    mask = [int(rs.choice([0xffffffff, 1, (1 << nq) - 2 if nq > 1 else 1, 0xffffffff])) for _ in range(n)]
    mask = [m if m & ((1 << nq) - 1) else 0xffffffff for m in mask]
    q, start, order = schedule(est, deps, nq, mask)
    check_invariants(est, deps, nq, q, start, order, mask)
    q2, s2, o2 = schedule(est, deps, nq, mask)
    assert np.array_equal(q, q2) and np.array_equal(start, s2) and np.array_equal(order, o2)
    # never worse than running everything in sequence, never better than the critical path
    crit = np.zeros(n)
    for i in range(n):
        crit[i] = est[i] + max([crit[d] for d in deps[i]], default=0.0)
    makespan = float(np.max(start + est))
    assert crit.max() - 1e-2 <= makespan <= float(est.sum()) + HOP * n


def test_pass_shaped_graph_pipelines_the_layers():
    """The forward pass's shape: L layers x nc chunks, gi GEMM -> sweep, sweep(l,c) after sweep(l,c-1) and gi(l,c) after
    sweep(l-1,c).  On four queues the planned makespan must be close to the wavefront bound, far below the serial sum."""
    L, nc, sweep, gemm = 5, 6, 500.0, 120.0
    est, deps, ids = [0.0], [[]], {}
    for l in range(L):
        for c in range(nc):
            est.append(gemm); deps.append([0] if l == 0 else [ids[("s", l - 1, c)]]); ids[("g", l, c)] = len(est) - 1
            est.append(sweep); deps.append([ids[("g", l, c)]] + ([ids[("s", l, c - 1)]] if c else [])); ids[("s", l, c)] = len(est) - 1
    q, start, order = schedule(est, deps, 4)
    check_invariants(est, deps, 4, q, start, order)
    makespan = float(np.max(start + np.asarray(est, np.float32)))
    wavefront = (nc + L - 1) * (sweep + gemm + 2 * HOP)
    assert makespan <= 1.25 * wavefront and makespan < 0.45 * sum(est)
    assert schedule(est, deps, 1)[1].max() >= sum(est) - sweep - 1e-3       # one queue: strictly serial


def test_rejects_non_topological_input():
    lib = N.load()
    est = np.array([1.0, 1.0], np.float32); qm = np.full(2, 0xffffffff, np.uint32)
    off = np.array([0, 1, 1], np.int32); deps = np.array([1], np.int32)
    q = np.zeros(2, np.int32); s = np.zeros(2, np.float32); o = np.zeros(2, np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.b2t_plan_schedule_host(2, P(est), P(qm), P(off), P(deps), 4, P(q), P(s), P(o)) != 0
    assert "topological" in N.last_error()


def schedule_admission(est, deps, cls, nq):
    lib = N.load()
    n = len(est)
    est = np.asarray(est, np.float32)
    qm = np.full(n, 0xffffffff, np.uint32)
    off = np.zeros(n + 1, np.int32)
    flat = []
    for i, d in enumerate(deps):
        flat += list(d); off[i + 1] = len(flat)
    flat = np.asarray(flat if flat else [0], np.int32)
    cl = np.asarray(cls, np.int32)
    q = np.zeros(n, np.int32); start = np.zeros(n, np.float32); end = np.zeros(n, np.float32); order = np.zeros(n, np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    N.check(lib.b2t_plan_admission_host(n, P(est), P(qm), P(off), P(flat), P(cl), nq, P(q), P(start), P(end), P(order)),
            "b2t_plan_admission_host")
    return q, start, end, order


def _wavefront(L, nc, sweep_us, gemm_us):
    """The forward pass's shape: sweep (l, c) after its projection gi (l, c) and after sweep (l, c-1); gi (l, c) after sweep
    (l-1, c).  Returns est, deps, cls (layer parity for the sweeps, -1 for the GEMMs)."""
    est, deps, cls, sw = [], [], [], {}
    for l in range(L):
        for c in range(nc):
            est.append(gemm_us); deps.append([sw[(l - 1, c)]] if l > 0 else []); cls.append(-1)
            gi = len(est) - 1
            est.append(sweep_us); deps.append([gi] + ([sw[(l, c - 1)]] if c > 0 else [])); cls.append(l & 1)
            sw[(l, c)] = len(est) - 1
    return est, deps, cls


@pytest.mark.parametrize("L,nc,sweep_us,gemm_us", [(5, 6, 450.0, 120.0), (5, 4, 800.0, 60.0), (7, 8, 300.0, 10.0), (3, 6, 500.0, 100.0), (8, 3, 200.0, 400.0)])
def test_admission_never_more_than_two_of_a_class_in_flight(L, nc, sweep_us, gemm_us):
    """Sweeps of one layer parity share an XCD set that holds the row groups of exactly two sweeps (exec.cpp: admission edges).
    In the planned schedule of the wavefront no instant has three sweeps of a parity running -- without the edges five layers
    put layers 0, 2 and 4 in flight together -- and, by construction (k-th waits for the (k-2)-th to FINISH), neither can
    the executed one.  The schedule stays valid: dependencies, hops, in-order queues."""
    est, deps, cls = _wavefront(L, nc, sweep_us, gemm_us)
    q, start, end, order = schedule_admission(est, deps, cls, 4)
    n = len(est)
    pos = np.empty(n, int); pos[order] = np.arange(n)
    for i in range(n):
        for d in deps[i]:
            assert pos[d] < pos[i] and start[i] + 1e-3 >= end[d] + (HOP if q[d] != q[i] else 0.0)
    for qq in range(4):
        ids = [i for i in order if q[i] == qq]
        for a, b in zip(ids, ids[1:]):
            assert start[b] + 1e-3 >= end[a]
    for k in (0, 1):
        ids = sorted((i for i in range(n) if cls[i] == k), key=lambda i: (start[i], i))
        for a, b in zip(ids, ids[2:]):
            assert start[b] + 1e-3 >= end[a], f"class {k}: task {b} planned to start while {a} and its successor are still in flight"
        events = sorted([(start[i], 1) for i in ids] + [(end[i] - 1e-3, -1) for i in ids])
        live = peak = 0
        for _, dlt in events:
            live += dlt; peak = max(peak, live)
        assert peak <= 2
    if L >= 5 and gemm_us < sweep_us:      # the unconstrained plan does overlap three of a parity (what the edges are for)
        q0, s0, o0 = schedule(est, deps, 4)
        e0 = s0 + np.asarray(est, np.float32)
        ids = [i for i in range(n) if cls[i] == 0]
        ev = sorted([(s0[i], 1) for i in ids] + [(e0[i] - 1e-3, -1) for i in ids])
        live = peak = 0
        for _, dlt in ev:
            live += dlt; peak = max(peak, live)
        assert peak >= 2


def test_admission_without_classes_is_the_plain_schedule():
    est = [0.0] + [100.0] * 4 + [0.0]
    deps = [[]] + [[0]] * 4 + [[1, 2, 3, 4]]
    q, start, order = schedule(est, deps, 4)
    q2, start2, end2, order2 = schedule_admission(est, deps, [-1] * 6, 4)
    assert q.tolist() == q2.tolist() and order.tolist() == order2.tolist()
    np.testing.assert_allclose(start, start2)


def test_admission_fuzz_random_estimates():
    """Random wavefront graphs with random per-task estimates (the advisor's round-2 fuzz: ranks and placement used to
    follow task ids, admission edges can point from a later-created task to an earlier-created one, and a dependant could
    be placed before its admission dependency): every data dependency AND every admission edge must hold in the planned
    schedule, i.e. never three sweeps of a parity in flight, and the issue order must respect them."""
    rng = np.random.RandomState(1234)
    for trial in range(400):
        L, nc = int(rng.randint(2, 9)), int(rng.randint(1, 9))
        est, deps, cls = _wavefront(L, nc, 1.0, 1.0)
        est = [float(rng.choice([0.0, 5.0, 40.0, 120.0, 450.0, 900.0]) * rng.uniform(0.5, 1.5)) for _ in est]
        nq = int(rng.choice([2, 3, 4]))
        q, start, end, order = schedule_admission(est, deps, cls, nq)
        n = len(est)
        assert sorted(order.tolist()) == list(range(n))
        pos = np.empty(n, int); pos[order] = np.arange(n)
        for i in range(n):
            for d in deps[i]:
                assert pos[d] < pos[i] and start[i] + 1e-3 >= end[d] + (HOP if q[d] != q[i] else 0.0), (trial, i, d)
        for qq in range(nq):
            ids = [i for i in order if q[i] == qq]
            for a, b in zip(ids, ids[1:]):
                assert start[b] + 1e-3 >= end[a]
        for k in (0, 1):
            ids = [i for i in range(n) if cls[i] == k]
            events = sorted([(start[i], 1) for i in ids] + [(end[i] - 1e-3, -1) for i in ids])
            live = peak = 0
            for _, dlt in events:
                live += dlt; peak = max(peak, live)
            assert peak <= 2, (trial, L, nc, nq, k, peak)


@pytest.mark.parametrize("L,nc,sweep_us,gemm_us", [(5, 4, 800.0, 60.0), (7, 6, 300.0, 10.0), (8, 3, 200.0, 400.0)])
def test_admission_capacity_one_classes(L, nc, sweep_us, gemm_us):
    """Round 5: the paired backward sweeps (one 512-thread workgroup per CU on the two XCDs of their set) are admission classes
    2..5 = 2 + (layer & 3) with room for ONE task in flight (exec.cpp cls_capacity): in the planned schedule no two tasks of such a
    class overlap, classes 0 / 1 keep their capacity of two, and the schedule stays valid."""
    est, deps, cls = _wavefront(L, nc, sweep_us, gemm_us)
    layer = 0
    cls1 = []
    k = 0
    for c in cls:                      # sweeps come in layer-major order: re-class them by layer & 3
        if c >= 0:
            cls1.append(2 + ((k // nc) & 3)); k += 1
        else:
            cls1.append(-1)
    q, start, end, order = schedule_admission(est, deps, cls1, 4)
    n = len(est)
    pos = np.empty(n, int); pos[order] = np.arange(n)
    for i in range(n):
        for d in deps[i]:
            assert pos[d] < pos[i] and start[i] + 1e-3 >= end[d] + (HOP if q[d] != q[i] else 0.0)
    for kcls in range(2, 6):
        ids = sorted((i for i in range(n) if cls1[i] == kcls), key=lambda i: (start[i], i))
        for a, b in zip(ids, ids[1:]):
            assert start[b] + 1e-3 >= end[a], f"class {kcls}: tasks {a} and {b} planned in flight together"
