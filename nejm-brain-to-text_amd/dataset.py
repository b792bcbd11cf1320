"""Batch sources with the reference's batch-dict contract (model_training/dataset.py:100-159):
    input_features [B,T,F] f32 zero-padded · seq_class_ids [B,S] zero-padded · n_time_steps [B] ·
    phone_seq_lens [B] · day_indicies [B] · transcriptions [B,*] · block_nums [B] · trial_nums [B]

`BrainToTextDataset` / `train_test_split_indicies` keep the reference's constructor arguments and
sampling semantics (day-balanced random training batches drawn with replacement; validation batches
that visit every trial once, one day per batch) and read the same HDF5 layout
(groups `trial_%04d` with datasets input_features / seq_class_ids / transcription and attrs
n_time_steps / seq_len / block_num / trial_num).  h5py is imported lazily: it is only needed when real
session files are used.  `SyntheticTrials` produces the same dict from seeded random data so the
trainer, tests and benchmarks run without the Dryad download.
`ResidentDataset` keeps every trial of a split in HBM as flat arrays (one-time conversion, optional
flat-binary file) and assembles a batch on the device (b2t_batch_gather_b32): no per-batch file reads, no host padding,
no PCIe copy of the features.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset


def _collate(feats, labels, trans, n_steps, seq_lens, days, blocks, trials):
    return {
        'input_features': pad_sequence(feats, batch_first=True, padding_value=0),
        'seq_class_ids': pad_sequence(labels, batch_first=True, padding_value=0),
        'n_time_steps': torch.tensor(n_steps),
        'phone_seq_lens': torch.tensor(seq_lens),
        'day_indicies': torch.tensor(days),
        'transcriptions': torch.stack(trans) if trans else torch.zeros((0, 1), dtype=torch.int64),
        'block_nums': torch.tensor(blocks),
        'trial_nums': torch.tensor(trials),
    }


class BrainToTextDataset(Dataset):
    """Each item is a whole batch.  split='train': n_batches random batches, each made of
    `days_per_batch` distinct days with ceil(batch_size/days_per_batch) trials per day sampled with
    replacement, surplus trials removed from random days; split='test': consecutive slices of each day."""

    def __init__(self, trial_indicies, n_batches, split='train', batch_size=64, days_per_batch=1, random_seed=-1,
                 must_include_days=None, feature_subset=None):
        if random_seed != -1:
            np.random.seed(random_seed)
            torch.manual_seed(random_seed)
        if split not in ('train', 'test'):
            raise ValueError(f'split must be either "train" or "test". Received {split}')
        self.split, self.batch_size, self.days_per_batch = split, batch_size, days_per_batch
        self.trial_indicies = trial_indicies
        self.n_days = len(trial_indicies)
        self.n_trials = sum(len(v['trials']) for v in trial_indicies.values())
        self.feature_subset = feature_subset
        if must_include_days is not None:
            if len(must_include_days) > (days_per_batch or 0):
                raise ValueError('must_include_days must be less than or equal to days_per_batch')
            must_include_days = [d if d >= 0 else self.n_days + d for d in must_include_days]
        self.must_include_days = must_include_days
        if split == 'train':
            if days_per_batch > self.n_days:
                raise ValueError(f'Requested days_per_batch: {days_per_batch} is greater than available days {self.n_days}.')
            self.n_batches = n_batches
            self.batch_index = self._index_train()
        else:
            self.batch_index = self._index_test()
            self.n_batches = len(self.batch_index)

    def __len__(self):
        return self.n_batches

    def _index_train(self):
        days_all = list(self.trial_indicies.keys())
        fixed = list(self.must_include_days) if self.must_include_days else []
        free = [d for d in days_all if d not in fixed]
        per_day = math.ceil(self.batch_size / self.days_per_batch)
        index = {}
        for bi in range(self.n_batches):
            if fixed:
                extra = np.random.choice(free, size=self.days_per_batch - len(fixed), replace=False)
                days = np.concatenate((fixed, extra))
            else:
                days = np.random.choice(days_all, size=self.days_per_batch, replace=False)
            picks = {d: np.random.choice(self.trial_indicies[d]['trials'], size=per_day, replace=True) for d in days}
            surplus = per_day * len(days) - self.batch_size
            while surplus > 0:
                d = np.random.choice(days)
                picks[d] = picks[d][:-1]
                surplus -= 1
            index[bi] = picks
        return index

    def _index_test(self):
        index, bi = {}, 0
        for d, info in self.trial_indicies.items():
            trials = info['trials']
            for s in range(0, len(trials), self.batch_size):
                index[bi] = {d: trials[s:s + self.batch_size]}
                bi += 1
        return index

    def read_trials(self, d, tlist):
        """The trials `tlist` of day `d` from that day's session file, in order: a list of
        (trial id, features [T,F], labels, transcription, n_time_steps, seq_len, block_num, trial_num); unreadable trials
        are logged and skipped like the reference does (dataset.py:144-146)."""
        import h5py
        out = []
        with h5py.File(self.trial_indicies[d]['session_path'], 'r') as f:
            for t in tlist:
                try:
                    g = f[f'trial_{t:04d}']
                    x = torch.from_numpy(g['input_features'][:])
                    if self.feature_subset:
                        x = x[:, self.feature_subset]
                    out.append((int(t), x, torch.from_numpy(g['seq_class_ids'][:]), torch.from_numpy(g['transcription'][:]),
                                g.attrs['n_time_steps'], g.attrs['seq_len'], g.attrs['block_num'], g.attrs['trial_num']))
                except Exception as e:
                    print(f'Error loading trial {t} from session {self.trial_indicies[d]["session_path"]}: {e}')
        return out

    def __getitem__(self, idx):
        feats, labels, trans, n_steps, seq_lens, days, blocks, trials = [], [], [], [], [], [], [], []
        for d, tlist in self.batch_index[idx].items():
            for (_, x, lab, tr, nt, sl, bn, tn) in self.read_trials(d, tlist):
                feats.append(x); labels.append(lab); trans.append(tr); n_steps.append(nt); seq_lens.append(sl)
                days.append(int(d)); blocks.append(bn); trials.append(tn)
        return _collate(feats, labels, trans, n_steps, seq_lens, days, blocks, trials)


def train_test_split_indicies(file_paths, test_percentage=0.1, seed=-1, bad_trials_dict=None):
    """Per-day trial lists for the train / test splits: {day: {'trials': [...], 'session_path': path}}."""
    if seed != -1:
        np.random.seed(seed)
    per_day = {}
    for i, path in enumerate(file_paths):
        sess = [s for s in path.split('/') if s.startswith('t15.20') or s.startswith('t12.20')]
        session = sess[0] if sess else os.path.basename(os.path.dirname(path))
        good = []
        if os.path.exists(path):
            import h5py
            with h5py.File(path, 'r') as f:
                for t in range(len(list(f.keys()))):
                    g = f[f'trial_{t:04d}']
                    bn, tn = g.attrs['block_num'], g.attrs['trial_num']
                    bad = (bad_trials_dict is not None and session in bad_trials_dict
                           and str(bn) in bad_trials_dict[session] and tn in bad_trials_dict[session][str(bn)])
                    if not bad:
                        good.append(t)
        per_day[i] = (good, path)
    train, test = {}, {}
    for day, (good, path) in per_day.items():
        if test_percentage == 0:
            tr, te = good, []
        elif test_percentage == 1:
            tr, te = [], good
        else:
            n_test = max(1, int(len(good) * test_percentage))
            te = np.random.choice(good, size=n_test, replace=False).tolist()
            tr = [t for t in good if t not in te]
        train[day] = {'trials': tr, 'session_path': path}
        test[day] = {'trials': te, 'session_path': path}
    return train, test


class SyntheticTrials(Dataset):
    """Seeded synthetic stand-in with the same batch dict: class-dependent templates + noise so that a
    short training run measurably lowers the CTC loss / PER (used by tests and examples)."""

    def __init__(self, n_batches, batch_size, n_days, n_features, n_classes, days_per_batch, max_T=120, min_T=60,
                 max_S=12, seed=0, one_day_per_batch=False):
        self.n_batches, self.B, self.D, self.F, self.C = n_batches, batch_size, n_days, n_features, n_classes
        self.dpb, self.max_T, self.min_T, self.max_S, self.seed = days_per_batch, max_T, min_T, max_S, seed
        self.one_day = one_day_per_batch
        g = np.random.default_rng(1234)
        self.templates = g.standard_normal((n_classes, n_features)).astype(np.float32)
        self.day_gain = (1.0 + 0.1 * g.standard_normal((n_days, n_features))).astype(np.float32)

    def __len__(self):
        return self.n_batches

    def __getitem__(self, idx):
        g = np.random.default_rng(self.seed * 1000003 + idx)
        if self.one_day:
            days = np.full(self.B, idx % self.D)
        else:
            chosen = g.choice(self.D, size=min(self.dpb, self.D), replace=False)
            days = np.repeat(chosen, math.ceil(self.B / len(chosen)))[:self.B]
        feats, labels, n_steps, seq_lens = [], [], [], []
        for b in range(self.B):
            S = int(g.integers(3, self.max_S + 1))
            lab = g.integers(1, self.C, S)
            T = int(g.integers(max(self.min_T, 6 * S), self.max_T + 1))
            bounds = np.sort(g.choice(np.arange(1, T), size=S - 1, replace=False)) if S > 1 else np.array([], int)
            seg = np.searchsorted(bounds, np.arange(T), side='right')
            x = self.templates[lab[seg]] * self.day_gain[days[b]] + 0.5 * g.standard_normal((T, self.F)).astype(np.float32)
            feats.append(torch.from_numpy(x.astype(np.float32)))
            labels.append(torch.from_numpy(lab.astype(np.int64)))
            n_steps.append(T)
            seq_lens.append(S)
        trans = [torch.zeros(4, dtype=torch.int64) for _ in range(self.B)]
        return _collate(feats, labels, trans, n_steps, seq_lens, [int(d) for d in days], [0] * self.B,
                        list(range(self.B)))


class ResidentFormatError(RuntimeError):
    pass


class ResidentDataset:
    """All trials of a split resident on the device: features [sum_T, F] f32 and labels [sum_S] i32 back to back with
    row offsets, plus the per-trial scalars.  `batch(rows)` returns the reference's batch dict (dataset.py:100-159) for
    the given trial rows, assembled by one gather kernel per array; `from_dataset(dataset)` flattens a source once
    (unique trials only when the source has a trial index); `batch_of(i)` replays the source's i-th batch (its rows are
    indices into the trial table: batch_rows[batch_off[i]:batch_off[i+1]]).
    `save(path)` / `load(path, device)`: a flat binary (.npz, uncompressed) so that later runs skip the conversion."""

    KEYS = ('feat', 'feat_off', 'lab', 'lab_off', 'lab_n', 'trans', 'n_time_steps', 'seq_len', 'day', 'block', 'trial',
            'batch_rows', 'batch_off')
    MAX_BYTES = int(float(os.environ.get("B2T_RESIDENT_MAX_GB", "200")) * 1e9)   # refuse conversions that cannot fit in HBM

    def __init__(self, arrays: dict, device='cuda:0'):
        self.device = torch.device(device)
        self.host = {k: np.ascontiguousarray(arrays[k]) for k in self.KEYS}
        h = self.host
        self.n_trials = int(h['n_time_steps'].shape[0])
        self.F = int(h['feat'].shape[1])
        to = lambda a, dt: torch.from_numpy(a).to(self.device, dtype=dt).contiguous()
        self.feat = to(h['feat'], torch.float32)
        self.lab = to(h['lab'], torch.int32)
        self.feat_off = to(h['feat_off'], torch.int64)
        self.lab_off = to(h['lab_off'], torch.int64)
        self.n_time_steps = to(h['n_time_steps'], torch.int32)
        self.seq_len = to(h['seq_len'], torch.int32)
        self.lab_n = to(h['lab_n'], torch.int32)   # stored label-row lengths (the session files pad them with zeros)
        self.day = to(h['day'], torch.int64)
        self.block = to(h['block'], torch.int64)
        self.trial = to(h['trial'], torch.int64)
        self.trans = to(h['trans'], torch.int64)
        # dtypes the source's label / transcription rows had (the reference's collate keeps them: dataset.py:136-137)
        self.lab_dtype = getattr(torch, str(arrays.get('lab_dtype', 'int64')).replace('torch.', ''))
        self.trans_dtype = getattr(torch, str(arrays.get('trans_dtype', 'int64')).replace('torch.', ''))
        self.host['lab_dtype'] = np.array(str(self.lab_dtype))
        self.host['trans_dtype'] = np.array(str(self.trans_dtype))

    # ---- construction -------------------------------------------------------------------------------------------
    @classmethod
    def from_dataset(cls, dataset, device='cuda:0'):
        """Flatten a batch source.  A BrainToTextDataset (it has `trial_indicies` + `batch_index`) stores every UNIQUE
        (day, trial) once -- the training split draws its 120,000 x 64 batch rows with replacement from ~8 k trials, so
        replaying the batches would copy each trial ~1000 times -- and keeps the source's batch composition as row
        indices into that table.  Other sources (SyntheticTrials: every batch is new data) go through from_batches."""
        if not (hasattr(dataset, 'trial_indicies') and hasattr(dataset, 'batch_index')):
            return cls.from_batches(dataset, device)
        feats, labs, foff, loff = [], [], [0], [0]
        nts, sls, days, blocks, trials, trans = [], [], [], [], [], []
        row_of = {}
        dts = {}
        for d, info in dataset.trial_indicies.items():
            wanted = sorted(set(int(t) for t in info['trials']))
            for (t, x, lab, tr, nt, sl, bn, tn) in dataset.read_trials(d, wanted):
                row_of[(d, t)] = len(nts)
                dts.setdefault('lab_dtype', str(lab.dtype)); dts.setdefault('trans_dtype', str(tr.dtype))
                T, S = int(nt), int(sl)
                # label rows are kept as stored (the session files zero-pad them to max_seq_elements): pad_sequence in the
                # reference's collate then yields the same [B, S_stored] array
                feats.append(x[:T].numpy().astype(np.float32)); labs.append(lab.numpy().astype(np.int32))
                foff.append(foff[-1] + T); loff.append(loff[-1] + int(lab.shape[0]))
                nts.append(T); sls.append(S); days.append(int(d)); blocks.append(int(bn)); trials.append(int(tn))
                trans.append(tr.numpy().astype(np.int64))
            if feats and foff[-1] * feats[0].shape[1] * 4 > cls.MAX_BYTES:
                raise RuntimeError(f"ResidentDataset: split exceeds {cls.MAX_BYTES / 1e9:.0f} GB (B2T_RESIDENT_MAX_GB)")
        rows, off = [], [0]
        for bi in range(len(dataset)):
            for d, tlist in dataset.batch_index[bi].items():
                rows.extend(row_of[(d, int(t))] for t in tlist if (d, int(t)) in row_of)   # unreadable trials were skipped
            off.append(len(rows))
        return cls(dict(cls._arrays(feats, labs, foff, loff, trans, nts, sls, days, blocks, trials, rows, off), **dts), device)

    @staticmethod
    def _arrays(feats, labs, foff, loff, trans, nts, sls, days, blocks, trials, rows, off):
        lab_n = np.diff(np.asarray(loff, np.int64)).astype(np.int32)
        wt = max(len(t) for t in trans)
        tr = np.zeros((len(trans), wt), dtype=np.int64)
        for i, t in enumerate(trans):
            tr[i, :len(t)] = t
        return dict(feat=np.concatenate(feats, 0), feat_off=np.asarray(foff, np.int64), lab=np.concatenate(labs, 0),
                    lab_off=np.asarray(loff, np.int64), lab_n=lab_n, trans=tr, n_time_steps=np.asarray(nts, np.int32),
                    seq_len=np.asarray(sls, np.int32), day=np.asarray(days, np.int64), block=np.asarray(blocks, np.int64),
                    trial=np.asarray(trials, np.int64), batch_rows=np.asarray(rows, np.int64),
                    batch_off=np.asarray(off, np.int64))

    @classmethod
    def from_batches(cls, dataset, device='cuda:0'):
        """Replay a source batch by batch (each sampled row stored as its own trial): for sources whose batches are all
        new data.  The estimated size is checked against MAX_BYTES before anything is copied."""
        if len(dataset) > 0:
            b0 = dataset[0]
            est = len(dataset) * int(b0['input_features'].numel()) * 4
            if est > cls.MAX_BYTES:
                raise RuntimeError(f"ResidentDataset.from_batches: ~{est / 1e9:.0f} GB of replayed batches exceed "
                                   f"{cls.MAX_BYTES / 1e9:.0f} GB (B2T_RESIDENT_MAX_GB); use from_dataset on a source with a "
                                   "trial index, or keep the DataLoader path")
        feats, labs, foff, loff = [], [], [0], [0]
        nts, sls, days, blocks, trials, trans, rows, off = [], [], [], [], [], [], [], [0]
        dts = {}
        for bi in range(len(dataset)):
            b = dataset[bi]
            dts.setdefault('lab_dtype', str(b['seq_class_ids'].dtype)); dts.setdefault('trans_dtype', str(b['transcriptions'].dtype))
            first = len(nts)
            B = int(b['n_time_steps'].shape[0])
            for i in range(B):
                T, S = int(b['n_time_steps'][i]), int(b['phone_seq_lens'][i])
                feats.append(b['input_features'][i, :T].numpy().astype(np.float32))
                labs.append(b['seq_class_ids'][i, :S].numpy().astype(np.int32))
                foff.append(foff[-1] + T); loff.append(loff[-1] + S)
                nts.append(T); sls.append(S); days.append(int(b['day_indicies'][i]))
                blocks.append(int(b['block_nums'][i])); trials.append(int(b['trial_nums'][i]))
                trans.append(b['transcriptions'][i].numpy().astype(np.int64))
            rows.extend(range(first, first + B)); off.append(len(rows))
        return cls(dict(cls._arrays(feats, labs, foff, loff, trans, nts, sls, days, blocks, trials, rows, off), **dts), device)

    FORMAT = 3      # 1: before lab_n / batch_rows / batch_off (one stored row per batch row); 2: before the source fingerprint

    @staticmethod
    def source_fingerprint(dataset) -> str:
        """What a cached flat binary was built FROM: the trial lists, the pre-generated batch index and the arguments that shape
        them (sessions / days, seed, batch size, number of batches, days per batch, feature subset, split, bad-trial filter as far
        as it shows in the trial lists).  A cache whose fingerprint differs from the source's is rebuilt, never reused."""
        import hashlib
        h = hashlib.sha256()

        def put(x):
            if isinstance(x, dict):
                h.update(b"{")
                for k in sorted(x, key=lambda v: str(v)):
                    put(k); put(x[k])
                h.update(b"}")
            elif isinstance(x, (list, tuple)):
                a = None
                try:
                    a = np.asarray(x)
                except Exception:
                    a = None
                if a is not None and a.dtype.kind in "iuf" and a.ndim >= 1:
                    h.update(str(a.shape).encode()); h.update(np.ascontiguousarray(a.astype(np.float64 if a.dtype.kind == "f" else np.int64)).tobytes())
                else:
                    h.update(b"[")
                    for v in x:
                        put(v)
                    h.update(b"]")
            elif isinstance(x, np.ndarray):
                h.update(str(x.shape).encode()); h.update(np.ascontiguousarray(x).tobytes())
            elif torch.is_tensor(x):
                put(x.detach().cpu().numpy())
            else:
                h.update(repr(x).encode()); h.update(b";")

        put(type(dataset).__name__); put(len(dataset))
        for name in ("split", "days_per_batch", "batch_size", "n_batches", "seed", "random_seed", "feature_subset", "n_days", "n_features",
                     "n_classes", "max_T", "min_T", "max_S", "one_day_per_batch", "must_include_days", "B", "D", "F", "C", "dpb", "one_day"):
            if hasattr(dataset, name):
                put(name); put(getattr(dataset, name))
        if hasattr(dataset, "trial_indicies"):
            put({int(d): dict(trials=[int(t) for t in info["trials"]], path=str(info.get("session_path", ""))) for d, info in dataset.trial_indicies.items()})
        if hasattr(dataset, "batch_index"):
            bi = dataset.batch_index
            put({int(i): {int(d): [int(t) for t in tl] for d, tl in bi[i].items()} for i in (bi.keys() if isinstance(bi, dict) else range(len(bi)))})
        return h.hexdigest()

    def save(self, path, fingerprint: str = ""):
        """Atomic: written next to `path` and renamed over it, so a reader never sees a torn file."""
        tmp = f"{path}.tmp.{os.getpid()}.npz"
        np.savez(tmp, format=np.int32(self.FORMAT), fingerprint=np.array(str(fingerprint)), **self.host)
        os.replace(tmp, path)

    @classmethod
    def load(cls, path, device='cuda:0', fingerprint=None):
        """Raises ResidentFormatError for a file an older version of this class wrote, one that lacks a table, or (when
        `fingerprint` is given) one built from a different source: `load_or_build` then re-converts the source."""
        with np.load(path) as z:
            fmt = int(z['format']) if 'format' in z.files else 1
            missing = [k for k in cls.KEYS if k not in z.files]
            if fmt != cls.FORMAT or missing:
                raise ResidentFormatError(f"{path}: resident-dataset format {fmt} (this build reads {cls.FORMAT})"
                                          + (f", missing tables {missing}" if missing else ""))
            have = str(z['fingerprint']) if 'fingerprint' in z.files else ""
            if fingerprint is not None and have != fingerprint:
                raise ResidentFormatError(f"{path}: built from a different source (sessions / seed / batch index / filters changed)")
            return cls({k: z[k] for k in cls.KEYS + tuple(m for m in ('lab_dtype', 'trans_dtype') if m in z.files)}, device)

    @classmethod
    def load_or_build(cls, path, dataset, device='cuda:0', writer=True):
        """The cached flat binary if it is there, readable and built from THIS source (fingerprint), else converted from
        `dataset` and -- by the one process with writer=True (rank 0 under data parallel) -- written back atomically.  Any failure
        to read the cache (torn or foreign file included) means 'rebuild', never a crash and never a silently stale batch index."""
        fp = cls.source_fingerprint(dataset) if path else None
        if path and os.path.exists(path):
            try:
                return cls.load(path, device, fingerprint=fp)
            except Exception as e:   # noqa: BLE001 -- ResidentFormatError, BadZipFile, ValueError, OSError ...: all mean rebuild
                import warnings
                warnings.warn(f"{type(e).__name__}: {e}; rebuilding from the source dataset")
        rd = cls.from_dataset(dataset, device)
        if path and writer:
            try:
                rd.save(path, fp)
            except OSError as e:
                import warnings
                warnings.warn(f"could not write the resident-dataset cache {path}: {e}")
        return rd

    # ---- batches ---------------------------------------------------------------------------------------------------
    def __len__(self):
        return int(self.host['batch_off'].shape[0]) - 1

    def batch_of(self, i):
        a, e = int(self.host['batch_off'][i]), int(self.host['batch_off'][i + 1])
        return self.batch(torch.from_numpy(self.host['batch_rows'][a:e]))

    def batch(self, rows):
        import ctypes as C
        import b2t_native as N
        import b2t_ops as ops
        rows_h = torch.as_tensor(rows, dtype=torch.int64).cpu()
        B = int(rows_h.shape[0])
        T = int(self.host['n_time_steps'][rows_h.numpy()].max())       # pad_sequence pads to the longest trial of the batch
        S = int(self.host['lab_n'][rows_h.numpy()].max())
        r = rows_h.to(self.device)
        nts, sls, labn = self.n_time_steps[r].contiguous(), self.seq_len[r].contiguous(), self.lab_n[r].contiguous()
        x = torch.empty((B, T, self.F), dtype=torch.float32, device=self.device)
        y = torch.empty((B, max(S, 1)), dtype=torch.int32, device=self.device)
        lib = N.load()
        with torch.cuda.device(self.device):
            N.check(lib.b2t_batch_gather_b32(ops._p(self.feat), ops._p(self.feat_off[r].contiguous()), ops._p(nts), ops._p(x),
                                             B, T, self.F, ops._stream()), "b2t_batch_gather_b32")
            N.check(lib.b2t_batch_gather_b32(ops._p(self.lab), ops._p(self.lab_off[r].contiguous()), ops._p(labn), ops._p(y),
                                             B, max(S, 1), 1, ops._stream()), "b2t_batch_gather_b32")
        # tensors the step consumes stay on the device; the bookkeeping fields are host tensors, as a DataLoader yields them
        hn = rows_h.numpy()
        host = lambda k: torch.from_numpy(self.host[k][hn])
        return {'input_features': x, 'seq_class_ids': y.to(self.lab_dtype), 'n_time_steps': nts.to(torch.int64),
                'phone_seq_lens': sls.to(torch.int64), 'day_indicies': host('day'), 'transcriptions': host('trans').to(self.trans_dtype),
                'block_nums': host('block'), 'trial_nums': host('trial')}


def make_synthetic_datasets(args):
    dsa = args['dataset']
    syn = dsa['synthetic'] if isinstance(dsa['synthetic'], dict) else {}
    kw = dict(batch_size=dsa['batch_size'], n_days=len(dsa['sessions']), n_features=args['model']['n_input_features'],
              n_classes=dsa['n_classes'], days_per_batch=dsa['days_per_batch'], max_T=syn.get('max_T', 120),
              min_T=syn.get('min_T', 60), max_S=syn.get('max_S', 12))
    train = SyntheticTrials(args['num_training_batches'], seed=dsa.get('seed', 1), **kw)
    val = SyntheticTrials(syn.get('val_batches', 4), seed=10_000 + dsa.get('seed', 1), one_day_per_batch=True, **kw)
    return train, val
