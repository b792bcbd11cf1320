"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/b2t.h declares; the
host logic (LR schedule, parameter groups, arena layout, bucket plan, dataset sampler, model init) matches the
reference's behaviour as captured in the golden fixtures; the product path refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest
import torch


def test_library_loads_and_exports_header_symbols():
    import b2t_native as N
    lib = N.load()
    syms = N.header_symbols()
    assert len(syms) >= 20 and "b2t_gemm_f32" in syms and "b2t_ctc_loss_f32" in syms
    for s in syms:
        assert hasattr(lib, s), s
        assert s in N._SIGNATURES, f"no ctypes signature for {s}"
    assert lib.b2t_version() == 2
    # error convention: non-zero return + message (null descriptor needs no GPU)
    rc = lib.b2t_gemm_f32(None, None)
    assert rc != 0 and b"null descriptor" in lib.b2t_last_error()


def test_header_cites_reference_for_every_entry_point():
    import b2t_native as N
    txt = open(N.HEADER_PATH).read()
    assert txt.count("rnn_trainer.py") >= 6 and "rnn_model.py" in txt and "lm_decoder.cc" in txt
    assert "extern \"C\"" in txt and "torch" not in re.sub(r"/\*.*?\*/", "", txt, flags=re.S)   # no torch types in the ABI


def test_no_cpu_fallback():
    from rnn_model import GRUDecoder
    from data_augmentations import gauss_smooth
    m = GRUDecoder(16, 32, 2, 41, 0, 0, 1, 0, 0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 10, 16), torch.tensor([0]))
    with pytest.raises(RuntimeError, match="no CPU path"):
        gauss_smooth(torch.zeros(1, 20, 16), "cpu")


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nejm-brain-to-text_amd")
    for fn in os.listdir(root):
        if fn.endswith(".py"):
            src = open(os.path.join(root, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle|__import__\(.oracle|import_module\(.oracle", src, flags=re.M), fn


@pytest.mark.parametrize("tag", ["a", "b"])
def test_model_init_matches_reference(golden_dir, tag):
    """Same seed -> the reference class's initial weights, names and shapes: bit-identical, except that the recurrent weights come
    out of nn.init.orthogonal_'s QR factorisation, whose last bits depend on the host's LAPACK kernels (on the EPYC of the GPU
    boxes they differ from this container's): those to 1e-6."""
    from rnn_model import GRUDecoder
    z = np.load(os.path.join(golden_dir, f"init_{tag}.npz"), allow_pickle=False)
    cfg = z["cfg"]
    torch.manual_seed(10)
    m = GRUDecoder(int(cfg[0]), int(cfg[1]), int(cfg[2]), int(cfg[3]), float(cfg[4]), float(cfg[5]), int(cfg[6]),
                   int(cfg[7]), int(cfg[8]))
    assert [n for n, _ in m.named_parameters()] == list(z["names"])
    sd = m.state_dict()
    gold = {k[4:]: z[k] for k in z.files if k.startswith("sd::")}
    assert set(sd) == set(gold)
    for k in gold:
        assert tuple(sd[k].shape) == gold[k].shape, k
        if "weight_hh" in k:
            np.testing.assert_allclose(sd[k].numpy(), gold[k], atol=1e-6, err_msg=k)
        else:
            assert np.array_equal(sd[k].numpy(), gold[k]), k


def test_arena_pack_preserves_values_and_aliases():
    from rnn_model import GRUDecoder
    import b2t_ops as ops
    torch.manual_seed(1)
    m = GRUDecoder(16, 32, 3, 41, 0, 0, 2, 0, 0)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    m.pack()
    lay = m.layout()
    assert lay["total"] % ops.ARENA_ALIGN == 0
    for (o, n) in lay["spans"]:
        assert o % ops.ARENA_ALIGN == 0
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    # parameters alias the arena: in-place update through the arena is visible in state_dict
    o, n = lay["spans"][lay["names"].index("h0")]
    m.arena()[o:o + n] += 1.0
    assert torch.allclose(m.h0.reshape(-1), before["h0"].reshape(-1) + 1.0)
    # load_state_dict writes through into the arena
    m.load_state_dict(before)
    assert torch.equal(m.arena()[o:o + n], before["h0"].reshape(-1))
    # checkpoints with torch.compile / DataParallel prefixes load after stripping (evaluate_model.py:74-76)
    from rnn_trainer import _strip_prefix
    m.load_state_dict(_strip_prefix({"_orig_mod." + k: v for k, v in before.items()}))


def test_lr_schedule_and_groups(golden_dir):
    from b2t_train_step import cosine_lr_factor, param_group_of
    z = np.load(os.path.join(golden_dir, "lr_table.npz"), allow_pickle=False)
    for s, fac in zip(z["steps"], z["factors"]):
        assert abs(cosine_lr_factor(int(s), 0.0001 / 0.005, 120000, 1000) - fac[0]) < 1e-15
    assert cosine_lr_factor(0, 0.02, 120000, 1000) == 0.0
    assert list(z["groups"]) == ["bias", "day_layer", "other"]
    assert param_group_of("gru.bias_ih_l0") == 0 and param_group_of("out.bias") == 0
    assert param_group_of("day_weights.3") == 1 and param_group_of("day_biases.0") == 1
    assert param_group_of("gru.weight_hh_l4") == 2 and param_group_of("h0") == 2 and param_group_of("out.weight") == 2


def test_bucket_plan_covers_arena():
    from rnn_model import GRUDecoder
    from b2t_train_step import bucket_spans
    m = GRUDecoder(16, 32, 4, 41, 0, 0, 3, 0, 0)
    lay = m.layout()
    spans = bucket_spans(lay, 3)
    assert [n for n, _, _ in spans] == ["head", "layer2", "layer1", "layer0", "h0", "day"]    # backward order
    covered = sorted((a, b) for _, a, b in spans)
    assert covered[0][0] == 0 and covered[-1][1] == lay["total"]
    for (a0, b0), (a1, b1) in zip(covered, covered[1:]):
        assert b0 == a1                                                                        # disjoint + complete


def test_taps_match_reference(golden_dir):
    import b2t_ops as ops
    z = np.load(os.path.join(golden_dir, "smooth.npz"), allow_pickle=False)
    np.testing.assert_allclose(ops.gauss_taps(2, 100), z["taps"], atol=1e-8)


def test_text_helpers(golden_dir):
    from evaluate_model_helpers import remove_punctuation, rearrange_speech_logits_pt, LOGIT_TO_PHONEME, greedy_phonemes
    for line in open(os.path.join(golden_dir, "remove_punctuation.txt")):
        src, want = line.rstrip("\n").split("\t")
        assert remove_punctuation(src) == want
    z = np.load(os.path.join(golden_dir, "greedy.npz"), allow_pickle=False)
    np.testing.assert_array_equal(rearrange_speech_logits_pt(z["logits"]), z["rearranged"])
    assert len(LOGIT_TO_PHONEME) == 41 and LOGIT_TO_PHONEME[0] == "BLANK" and LOGIT_TO_PHONEME[40] == " | "
    for b in range(z["logits"].shape[0]):
        assert greedy_phonemes(z["logits"][b]) == [LOGIT_TO_PHONEME[i] for i in z[f"evaluate_{b}"]]


def test_dataset_sampler_semantics():
    from dataset import BrainToTextDataset, SyntheticTrials
    trials = {d: {"trials": list(range(10 + d)), "session_path": f"/nonexistent/t15.2023.01.{d:02d}/x.hdf5"} for d in range(6)}
    ds = BrainToTextDataset(trials, n_batches=20, split="train", batch_size=10, days_per_batch=4, random_seed=1)
    assert len(ds) == 20
    for bi in range(20):
        picks = ds.batch_index[bi]
        assert len(picks) == 4 and sum(len(v) for v in picks.values()) == 10          # day-balanced, exact size
        for d, tl in picks.items():
            assert all(t in trials[d]["trials"] for t in tl)
    ds2 = BrainToTextDataset(trials, n_batches=20, split="train", batch_size=10, days_per_batch=4, random_seed=1)
    assert all(np.array_equal(ds.batch_index[3][d], ds2.batch_index[3][d]) for d in ds.batch_index[3])   # seeded
    te = BrainToTextDataset(trials, n_batches=None, split="test", batch_size=4, days_per_batch=None, random_seed=1)
    seen = {d: [] for d in trials}
    for bi in range(len(te)):
        (d, tl), = te.batch_index[bi].items()
        assert len(tl) <= 4
        seen[d] += list(tl)
    assert all(seen[d] == trials[d]["trials"] for d in trials)                         # every trial exactly once
    syn = SyntheticTrials(3, 8, 5, 16, 41, 4, max_T=50, min_T=30, max_S=6, seed=2)
    b = syn[1]
    assert set(b) == {"input_features", "seq_class_ids", "n_time_steps", "phone_seq_lens", "day_indicies",
                      "transcriptions", "block_nums", "trial_nums"}
    assert b["input_features"].dtype == torch.float32 and b["input_features"].shape[0] == 8
    assert int(b["n_time_steps"].max()) == b["input_features"].shape[1]
    for i in range(8):       # zero padding beyond each trial's length
        assert torch.all(b["input_features"][i, int(b["n_time_steps"][i]):] == 0)
        assert torch.all(b["seq_class_ids"][i, int(b["phone_seq_lens"][i]):] == 0)
    assert torch.equal(syn[1]["input_features"], b["input_features"])


def test_use_amp_selects_the_bf16_mode(monkeypatch):
    """rnn_args.yaml:19 `use_amp: true` -> autocast(bfloat16) (rnn_trainer.py:527,704; evaluate_model_helpers.py:90): the
    yaml's switch selects the bf16 mode here too; `amd_bf16_matmul` in the args and B2T_AMP override it in that order."""
    import b2t_ops as ops
    monkeypatch.delenv("B2T_AMP", raising=False)
    assert ops.precision_from_args({"use_amp": True}) is True
    assert ops.precision_from_args({"use_amp": False}) is False
    assert ops.precision_from_args({}) is False
    assert ops.precision_from_args({"use_amp": True, "amd_bf16_matmul": False}) is False     # the opt-out
    assert ops.precision_from_args({"use_amp": False, "amd_bf16_matmul": True}) is True
    # the reference's entry scripts pass an OmegaConf DictConfig: a Mapping that is NOT a dict subclass
    import collections.abc

    class Cfg(collections.abc.Mapping):
        def __init__(self, d): self._d = dict(d)
        def __getitem__(self, k): return self._d[k]
        def __iter__(self): return iter(self._d)
        def __len__(self): return len(self._d)
    assert not isinstance(Cfg({}), dict)
    assert ops.precision_from_args(Cfg({"use_amp": True})) is True
    assert ops.precision_from_args(Cfg({"use_amp": False})) is False
    assert ops.precision_from_args(Cfg({"use_amp": True, "amd_bf16_matmul": False})) is False
    assert ops.precision_from_args(Cfg({"amd_bf16_matmul": True})) is True
    assert ops.precision_from_args(None) is False
    monkeypatch.setenv("B2T_AMP", "0")
    assert ops.precision_from_args({"use_amp": True}) is False
    assert ops.precision_from_args({"use_amp": True, "amd_bf16_matmul": True}) is True       # the args win over the environment
    monkeypatch.setenv("B2T_AMP", "1")
    assert ops.precision_from_args({"use_amp": False}) is True


def test_time_chunk_plan_rule():
    """Layer pipelining only when two sweeps can be resident together (b2t_ops.time_chunks): C2 pipelines over 6 chunks,
    the H = 768 shape runs the layers in sequence, short sequences are not cut below 16 steps per chunk."""
    import b2t_ops as ops
    assert "B2T_CHUNKS" not in os.environ
    assert ops.time_chunks(500, 64, 512) == 6
    assert ops.time_chunks(122, 64, 768) == 1
    assert ops.time_chunks(122, 64, 768, amp=True) == 3       # bf16 operands: GEMMs and sweeps of different layers overlap
    assert ops.time_chunks_bwd(122, 64, 768, True, 3) == 3 and ops.time_chunks_bwd(500, 64, 512, False, 6) == 4
    assert ops.time_chunks_bwd(122, 64, 768, False, 1) == 0
    assert ops.time_chunks(500, 128, 512) == 6
    assert ops.time_chunks(500, 192, 512) == 1          # 2 x 384 workgroups > 512 slots
    assert ops.time_chunks(40, 64, 512) == 2 and ops.time_chunks(1, 32, 512) == 1


def test_sampler_reproduces_reference_batch_index(golden_dir):
    """BrainToTextDataset's batch index under a seed == the reference sampler's (dataset.py:162-242), captured by
    tests/golden/make_golden.py:make_sampler_index: day-balanced random training batches (with and without
    must_include_days, incl. the surplus-trial removal) and the sequential test batches."""
    import dataset as ds
    z = np.load(os.path.join(golden_dir, "sampler_index.npz"), allow_pickle=False)
    trial_idx = {int(d): {"trials": [int(t) for t in z[f"trials_{int(d)}"]], "session_path": f"/nonexistent/day{int(d)}.hdf5"}
                 for d in z["days"]}
    for tag, kw in (("a", dict(batch_size=10, days_per_batch=3, must_include_days=None)),
                    ("b", dict(batch_size=16, days_per_batch=4, must_include_days=[1, -1]))):
        tr = ds.BrainToTextDataset(trial_idx, n_batches=12, split="train", random_seed=7, **kw)
        assert len(tr) == 12
        for bi in range(12):
            days = [int(d) for d in tr.batch_index[bi].keys()]
            assert days == [int(d) for d in z[f"train_{tag}_{bi}_days"]], (tag, bi)
            for d in days:
                np.testing.assert_array_equal(np.asarray(tr.batch_index[bi][d]), z[f"train_{tag}_{bi}_{d}"])
            assert sum(len(v) for v in tr.batch_index[bi].values()) == kw["batch_size"]
    te = ds.BrainToTextDataset(trial_idx, n_batches=None, split="test", batch_size=16, random_seed=7)
    assert len(te) == int(z["test_n"])
    for bi in range(len(te)):
        (d, t), = te.batch_index[bi].items()
        assert int(d) == int(z[f"test_{bi}_day"])
        np.testing.assert_array_equal(np.asarray(t), z[f"test_{bi}"])


def test_rank_batches_shard_the_batch_index():
    """Data-parallel sharding of the pre-generated batch index: every rank takes the same number of steps, global step g
    uses batches g*world .. g*world+world-1, the tail that does not fill a step is dropped."""
    from rnn_trainer import rank_batches
    for n, world in ((12, 1), (12, 4), (13, 4), (7, 8), (120000, 8)):
        shards = [list(rank_batches(n, world, r)) for r in range(world)]
        assert len(set(len(s) for s in shards)) == 1
        steps = n // world
        assert all(len(s) == steps for s in shards)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(steps * world))
        for g in range(min(steps, 5)):
            assert [s[g] for s in shards] == list(range(g * world, (g + 1) * world))


def test_lr_schedules_host_side(golden_dir):
    from b2t_train_step import cosine_lr_factor, linear_lr_factor
    z = np.load(os.path.join(golden_dir, "lr_table.npz"), allow_pickle=False)
    tot = int(z["linear_total"])
    for i, lrs in enumerate(z["linear_lrs"]):
        np.testing.assert_allclose(0.005 * linear_lr_factor(i, 0.0001 / 0.005, tot), lrs[0], rtol=1e-12)
    for s_, fac in zip(z["steps"], z["factors"]):
        assert abs(cosine_lr_factor(int(s_), 0.0001 / 0.005, 120000, 1000) - fac[0]) < 1e-15


def test_convert_all_to_inputs_matches_the_scalar_form():
    """wfst_decoder.convert_all_to_inputs (one vectorised pass over an utterance's n-best alignments) against
    convert_to_inputs (CtcWfstBeamSearch::ConvertToInputs, ctc_wfst_beam_search.cc:162-188) entry by entry: blanks dropped,
    repeats merged, times of the runs' last positions, the frame map only for alignments that cover every decoded frame;
    empty entries, entries that end in the label the next one starts with, all-blank entries."""
    import numpy as np
    from wfst_decoder import convert_all_to_inputs, convert_to_inputs
    rng = np.random.default_rng(5)
    for trial in range(200):
        F = int(rng.integers(1, 40))
        n = int(rng.integers(0, 12))
        mapping = np.sort(rng.choice(200, size=F, replace=False)).astype(np.int32)
        parts = []
        for k in range(n):
            kind = rng.integers(0, 6)
            ln = F if kind < 3 else (0 if kind == 3 else int(rng.integers(1, 50)))
            parts.append(rng.choice([1, 1, 1, 2, 2, 3, 7, 41], size=ln).astype(np.int32) if kind != 5 else np.ones(ln, np.int32))
        off = np.zeros(n + 1, np.int32)
        for k, pp in enumerate(parts):
            off[k + 1] = off[k] + len(pp)
        ali = np.concatenate(parts + [np.full(5, 9, np.int32)]) if parts else np.full(5, 9, np.int32)   # trailing garbage beyond off[n]
        inps, tms = convert_all_to_inputs(ali, off, n, mapping)
        assert len(inps) == n and len(tms) == n
        for k, pp in enumerate(parts):
            want = convert_to_inputs(pp, mapping if len(pp) == F else np.arange(len(pp)))
            assert (inps[k], tms[k]) == (want[0], want[1]), (trial, k)


@pytest.mark.parametrize("cls,ctype", [("GemmDesc", "b2t_gemm_desc"), ("ModelDesc", "b2t_model_t"), ("PassDesc", "b2t_pass_t"),
                                       ("WfstGraph", "b2t_wfst_graph_t"), ("WfstOpts", "b2t_wfst_opts_t"), ("LexLmDesc", "b2t_lexlm_t"), ("WaveDesc", "b2t_wave_t")])
def test_struct_layouts_match_the_header(tmp_path, cls, ctype):
    """The ctypes mirrors in b2t_native against the C compiler's view of include/b2t.h: size and the offset of every
    (scalar or pointer) field -- a field added to one side only would silently shift everything behind it."""
    import ctypes as C
    import shutil
    import subprocess
    import b2t_native as Nn
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    K = getattr(Nn, cls)
    fields = [f[0] for f in K._fields_]
    src = tmp_path / "lay.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "b2t.h"\nint main(void){ printf("%zu\\n", sizeof(' + ctype + '));\n' +
                   "".join(f'printf("{f} %zu\\n", offsetof({ctype}, {f}));\n' for f in fields) + "return 0; }\n")
    exe = tmp_path / "lay"
    subprocess.run([gcc, "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    assert int(out[0]) == C.sizeof(K)
    for line in out[1:]:
        if line.strip():
            name, off = line.split()
            assert getattr(K, name).offset == int(off), name


def test_packed_gemm_workspace_sizes_are_host_arithmetic():
    """b2t_gemm_bf16p_ws_bytes / _ws_bytes_z (include/b2t.h) are plain arithmetic and run without a GPU: operands padded to 128 rows and
    64 k, two bytes per element, + 512; a Z-batched descriptor packs every matrix of the batch (Z times the operands of one)."""
    import b2t_native as N
    lib = N.load()
    pad = lambda v, a: (v + a - 1) // a * a
    for (M, Nn, K) in ((498, 512, 512), (512, 512, 498), (1, 1, 1), (2304, 7168, 7808)):
        one = lib.b2t_gemm_bf16p_ws_bytes(M, Nn, K)
        assert one == (pad(M, 128) + pad(Nn, 128)) * pad(K, 64) * 2 + 512
        assert lib.b2t_gemm_bf16p_ws_bytes_z(M, Nn, K, 1) == one
        assert lib.b2t_gemm_bf16p_ws_bytes_z(M, Nn, K, 64) == 64 * (one - 512) + 512
    assert lib.b2t_gemm_bf16p_ws_bytes(0, 4, 4) == 0 and lib.b2t_gemm_bf16p_ws_bytes_z(4, 4, 4, 0) == 0
