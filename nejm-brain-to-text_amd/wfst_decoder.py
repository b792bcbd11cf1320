"""Host side of the WFST decoder: drives the batched token-passing kernels (csrc/wfst.hip) for U utterances and turns
what they leave in HBM into the reference's outputs.

    WfstSearch.search(logp)      CtcWfstBeamSearch::Search        ctc_wfst_beam_search.cc:70-121  (partial best path)
    WfstSearch.finalize()        CtcWfstBeamSearch::FinalizeSearch ctc_wfst_beam_search.cc:123-160 (n-best)

The per-frame work (blank skipping, ProcessEmitting / ProcessNonemitting, cut-offs, token hash) and the backward lattice
pruning run on the GPU; the n-best extraction from the pruned lattice (a few thousand arcs per utterance, once per
utterance) is the C++ host function b2t_lattice_nbest_host.
"""
from __future__ import annotations

import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor
from typing import List

import numpy as np
import torch

import b2t_native as N
import b2t_ops as ops


def _host_threads() -> int:
    """Cores this process may run on (affinity mask), overridable with B2T_HOST_THREADS."""
    env = os.environ.get("B2T_HOST_THREADS")
    if env:
        return max(1, int(env))
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def _pool_threads() -> int:
    """Threads of the n-best pool: a quarter of the usable cores stays free for the thread that drives the GPU (with every
    core busy in b2t_lattice_nbest_host the next batch's launches and copies crawl: measured 67 vs 43 ms per pipelined batch),
    and never more than 16 -- the C search of 32 lattices takes 7.2 ms on 12 threads and 6.7 on 16 (EPYC 9575F), while every
    further thread only queues for the interpreter lock around its call."""
    n = _host_threads()
    return max(1, min(16, n - max(1, n // 4))) if not os.environ.get("B2T_HOST_THREADS") else n


def _pow2_at_least(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


class WfstSearch:
    def __init__(self, graph, opts, U: int = 1, device="cuda:0", max_frames: int = 1024, max_tokens: int = 1 << 19,
                 max_links: int = 1 << 21, hash_size: int = 0, prune_interval: int = 25, prune_scale: float = 0.1,
                 prune_min_fill: float = 0.0, prune_after_read: bool = False):
        """graph: wfst.DecodeGraph.  opts: an object with the reference's DecodeOptions fields (max_active, min_active, beam,
        lattice_beam, acoustic_scale, ctc_blank_skip_threshold, length_penalty, nbest)."""
        self.g, self.U, self.device = graph, int(U), torch.device(device)
        self.lib = N.load()
        d = graph.to_device(self.device)
        self._keep = d
        if getattr(graph, "compact", False):
            self.cg = N.WfstGraph(d["row"].data_ptr(), None, None, None, d["next"].data_ptr(), d["n_eps"].data_ptr(), d["final"].data_ptr(),
                                  graph.n_states, graph.start, d["labels"].data_ptr(), d["weight_f16"].data_ptr(), 1)
        else:
            self.cg = N.WfstGraph(d["row"].data_ptr(), d["ilabel"].data_ptr(), d["olabel"].data_ptr(), d["weight"].data_ptr(),
                                  d["next"].data_ptr(), d["n_eps"].data_ptr(), d["final"].data_ptr(), graph.n_states, graph.start,
                                  None, None, 0)
        if hash_size <= 0:
            # a frame can hold at most one token per graph state; graphs of the reference's size class (10^6+ states) put
            # 10-30 k tokens into a frame with the production beam: 2^18 slots keep the linear probing short there
            cap = 1 << 18 if graph.n_states > 500000 else 1 << 16
            hash_size = max(1024, min(cap, _pow2_at_least(2 * min(graph.n_states, cap // 2))))
        self.caps = (int(max_frames), int(max_tokens), int(max_links), int(hash_size))
        # LatticeFasterDecoderConfig::prune_interval / prune_scale (lattice-faster-decoder.h:62-72): PruneActiveTokens every
        # prune_interval frames; 0 = never (everything is pruned once, in finalize -- same lattice, more memory)
        self.prune_interval = int(os.environ.get("B2T_WFST_PRUNE_INTERVAL", prune_interval))
        self.prune_scale = float(prune_scale)
        # > 0: a pass is skipped for an utterance whose token / link arrays are filled below this fraction (memory-pressure
        # policy: the passes only bound memory and cost about as much as the frames between them); 0 = the reference's policy
        self.prune_min_fill = float(os.environ.get("B2T_WFST_PRUNE_MIN_FILL", prune_min_fill))
        self._since_prune = 0
        # Streaming: a PruneActiveTokens pass that falls due is NOT run inside the frame that made it due but enqueued behind the
        # frame's partial result (best_path's read-back), i.e. it runs while the host waits for the next 80 ms frame -- the
        # frame's latency is the search's, the lattice is the same (the pass sits between the same two frames as before).
        self.prune_after_read = bool(prune_after_read)
        self._prune_due = False
        self.set_opts(opts)
        self.state_bytes = self.lib.b2t_wfst_state_bytes(*self.caps)
        self.state = torch.zeros((self.U * self.state_bytes,), dtype=torch.uint8, device=self.device)
        off = (N.LL * 16)()
        N.check(self.lib.b2t_wfst_state_offsets(*self.caps, off), "b2t_wfst_state_offsets")
        self.off = list(off)
        self.reset()

    def set_opts(self, opts):
        self.nbest = int(getattr(opts, "nbest", 10))
        self.acoustic_scale = float(opts.acoustic_scale)
        self.lattice_beam = float(opts.lattice_beam)
        mf, mt, ml, hs = self.caps
        self.co = N.WfstOpts(float(opts.beam), float(opts.lattice_beam), float(getattr(opts, "beam_delta", 0.5)), float(opts.acoustic_scale),
                             float(getattr(opts, "length_penalty", 0.0)), float(opts.ctc_blank_skip_threshold),
                             int(min(opts.max_active, 2 ** 31 - 1)), int(opts.min_active), mf, mt, ml, hs)

    def _s(self):
        return ops._stream()

    def reset(self):
        with torch.cuda.device(self.device):
            N.check(self.lib.b2t_wfst_reset(C.byref(self.cg), C.byref(self.co), ops._p(self.state), self.U, self._s()), "b2t_wfst_reset")
        self.finalized = False
        self._since_prune = 0
        self._prune_due = False

    def prune(self):
        """PruneActiveTokens(lattice_beam * prune_scale) for every utterance + compaction of their token / link arrays."""
        with torch.cuda.device(self.device):
            N.check(self.lib.b2t_wfst_prune(C.byref(self.cg), C.byref(self.co), ops._p(self.state), self.U,
                                            C.c_float(self.lattice_beam * self.prune_scale), C.c_float(self.prune_min_fill), self._s()),
                    "b2t_wfst_prune")
        self._since_prune = 0
        self._prune_due = False

    # ---- Search ------------------------------------------------------------------------------------------------------
    def search(self, logp: torch.Tensor, lens=None):
        """logp [U, T, C] fp32 on the device (after the DecodeNumpy prologue); may be called chunk after chunk."""
        ops._need(logp, name="logp")
        U, T, Cc = logp.shape
        if U != self.U:
            raise ValueError(f"expected {self.U} utterances")
        lens_t = None if lens is None else self._lens_on_device(lens)
        iv = self.prune_interval
        if self._prune_due:                        # (nobody read a result in between)
            self.prune()
        t0 = 0
        while t0 < T:
            # the reference prunes whenever NumFramesDecoded() % prune_interval == 0 (lattice-faster-decoder.cc:592-630): a
            # long call is cut into pieces of at most prune_interval input frames with a PruneActiveTokens pass between them
            t1 = T if iv <= 0 else min(T, t0 + max(1, iv - self._since_prune))
            piece = logp if (t0 == 0 and t1 == T) else logp[:, t0:t1].contiguous()
            ln = lens_t if (lens_t is None or (t0 == 0 and t1 == T)) else (lens_t - t0).clamp(0, t1 - t0).to(torch.int32).contiguous()
            with torch.cuda.device(self.device):
                N.check(self.lib.b2t_wfst_search_f32(C.byref(self.cg), C.byref(self.co), ops._p(self.state), ops._p(piece),
                                                     ops._p(ln), U, t1 - t0, Cc, self._s()), "b2t_wfst_search_f32")
            self._since_prune += t1 - t0
            if iv > 0 and self._since_prune >= iv:
                if self.prune_after_read and t1 == T:
                    self._prune_due = True        # behind this call's result (best_path), or in front of the next search
                else:
                    self.prune()
            t0 = t1

    def _lens_on_device(self, lens):
        """int32 [U] on the device.  A streamed utterance passes the same few length vectors frame after frame (all ones until
        utterances end): the uploads are cached by value (an upload per frame is a staged host-to-device copy, ~30 us)."""
        if isinstance(lens, torch.Tensor) and lens.is_cuda:
            return lens.to(device=self.device, dtype=torch.int32).contiguous()
        arr = np.ascontiguousarray(np.asarray(lens, dtype=np.int32))
        key = arr.tobytes()
        cache = self.__dict__.setdefault("_lens_cache", {})
        t = cache.get(key)
        if t is None:
            if len(cache) >= 64:
                cache.clear()
            t = cache[key] = torch.from_numpy(arr.copy()).to(self.device)
        return t

    def best_path(self, use_final: bool = False, max_len: int = 0):
        """[(inputs, times, words, lm_score, ac_score)] per utterance: GetBestPath + ConvertToInputs + the likelihood pair.
        Called once per frame by a streaming decoder (the partial result of CtcWfstBeamSearch::Search, ctc_wfst_beam_search.cc:
        100-121), so its host side is one launch + one gather + ONE copy into pinned memory + vectorised numpy: the six
        result arrays and the utterances' header words live in one device buffer (the first version allocated and zeroed six
        tensors, read each back with its own synchronising copy and converted the alignments in Python loops: 0.6 of the
        0.79 ms a streamed frame took)."""
        max_len = int(max_len or self.caps[0] * 2)
        U, dev = self.U, self.device
        buf = self.__dict__.get("_bp_buf")
        if buf is None or buf["max_len"] != max_len:
            n_int = 3 * U * max_len + 4 * U + 18 * U                   # ali | frames | words | n_ali | n_words | costs[2] | header[18]
            buf = self._bp_buf = dict(max_len=max_len, dev=torch.empty((n_int,), dtype=torch.int32, device=dev),
                                      host=torch.empty((n_int,), dtype=torch.int32).pin_memory())
        d = buf["dev"]
        o_ali, o_fr, o_wd = 0, U * max_len, 2 * U * max_len
        o_na = 3 * U * max_len; o_nw = o_na + U; o_cs = o_nw + U; o_hd = o_cs + 2 * U
        base = d.data_ptr()
        P = lambda off: C.c_void_p(base + 4 * off)
        with torch.cuda.device(dev):
            N.check(self.lib.b2t_wfst_best_path(C.byref(self.cg), C.byref(self.co), ops._p(self.state), U, int(use_final), max_len,
                                                P(o_ali), P(o_fr), P(o_na), P(o_wd), P(o_nw), P(o_cs), self._s()),
                    "b2t_wfst_best_path")
            st = self.state.view(U, self.state_bytes)
            d[o_hd:o_hd + 18 * U].view(U, 18).copy_(st[:, self.off[0]:self.off[0] + 72].view(torch.int32))
            buf["host"].copy_(d, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            if self._prune_due and not use_final:
                self.prune()                       # asynchronous: runs while the caller handles this frame's result
        h = buf["host"].numpy()
        self._raise_on_overflow(h[o_hd:o_hd + 18 * U].reshape(U, 18))
        na, nw = h[o_na:o_na + U], h[o_nw:o_nw + U]
        cs = h[o_cs:o_cs + 2 * U].view(np.float32).reshape(U, 2)
        inps, tms = convert_rows_to_inputs(h[o_ali:o_ali + U * max_len].reshape(U, max_len), h[o_fr:o_fr + U * max_len].reshape(U, max_len), na)
        wd = h[o_wd:o_wd + U * max_len].reshape(U, max_len)
        return [(inps[u], tms[u], wd[u, :nw[u]].tolist(), -float(cs[u, 0]), -float(cs[u, 1])) for u in range(U)]

    def _header(self):
        st = self.state.view(self.U, self.state_bytes)
        return st[:, self.off[0]:self.off[0] + 72].contiguous().view(torch.int32).cpu().numpy()

    def memory_stats(self):
        """Per utterance: tokens / links held now, their high-water marks, PruneActiveTokens passes so far."""
        h = self._header()
        return [dict(tokens=int(r[1]), links=int(r[2]), peak_tokens=max(int(r[14]), int(r[1])), peak_links=max(int(r[15]), int(r[2])),
                     prunes=int(r[13]), created_tokens=int(r[1]) + int(r[16]), created_links=int(r[2]) + int(r[17])) for r in h]

    def _check_overflow(self):
        self._raise_on_overflow(self._header())

    def _raise_on_overflow(self, h):
        if h[:, 3].any():
            bits = int(np.bitwise_or.reduce(h[:, 3]))
            if bits & 64:
                raise RuntimeError("WFST cluster search: the workgroups of an utterance were not placed on one XCD (block b is "
                                   "expected on XCD b % 8); nothing was decoded -- call b2t_wfst_set_cluster(1) / set "
                                   "B2T_WFST_CLUSTER=1 to use one workgroup per utterance")
            if bits & 32:
                raise RuntimeError("WFST cluster search: a cluster barrier timed out (a workgroup of the cluster was not resident: "
                                   "is another kernel holding CUs?) or a frame's epsilon closure did not converge (an epsilon cycle of negative "
                                   "weight in the graph); results are invalid -- reset() and retry, or B2T_WFST_CLUSTER=1")
            what = [n for b, n in ((1, "max_tokens"), (2, "max_links"), (4, "hash_size"), (8, "max_frames"), (16, "epsilon work list / heavy-token list of a frame (524288 / 131072)")) if bits & b]
            raise RuntimeError(f"WFST search: a capacity was exhausted ({', '.join(what)} of {self.caps}; peak tokens "
                               f"{int(h[:, 1].max())}, links {int(h[:, 2].max())}); results are invalid -- construct WfstSearch "
                               "with larger capacities")

    def arcs_expanded(self):
        """Emitting graph arcs examined so far, per utterance (16 B of graph each: the search's algorithmic traffic)."""
        h = self._header().astype(np.int64)
        return [int((h[u, 11] & 0xffffffff) << 32 | (h[u, 10] & 0xffffffff)) for u in range(self.U)]

    def frames_decoded(self):
        return [int(v) for v in self._header()[:, 0]]

    # ---- FinalizeSearch ------------------------------------------------------------------------------------------------
    def finalize(self):
        """FinalizeDecoding on the GPU, then per utterance the n-best word sequences of the pruned lattice:
        [[(inputs, times, words, lm_score, ac_score), ...] best first]."""
        return self.finalize_async().result()

    _pool = None

    def finalize_async(self):
        """finalize() split where the GPU's part ends: FinalizeDecoding, lattice compaction and ONE copy to the host happen
        here; the n-best extraction (host C++, one thread per utterance) runs on a background pool and is returned as a
        future -- the caller may reset() and decode the next batch of utterances while it runs (the state block is no longer
        needed), which takes the host n-best off the decode loop's critical path."""
        self.finalize_begin()
        return self.finalize_collect()

    def finalize_begin(self):
        """The first half of finalize_async(): FinalizeDecoding is launched, nothing is waited for.  `finalize_event` is
        recorded behind it: a SECOND searcher working on another stream makes that stream wait for it
        (`torch.cuda.current_stream().wait_event(first.finalize_event)`), resets and searches the next batch of utterances at
        once -- two cluster searches must never be in flight together (neither would get all its workgroups resident), but
        the lattice extraction, the copy to the host and the host's share of this batch then run under the next search
        (tools/bench_wfst.py, `pipelined`)."""
        self._prune_due = False                    # (FinalizeDecoding prunes everything)
        with torch.cuda.device(self.device):
            N.check(self.lib.b2t_wfst_finalize(C.byref(self.cg), C.byref(self.co), ops._p(self.state), self.U, self._s()), "b2t_wfst_finalize")
            ev = self.__dict__.get("finalize_event")
            if ev is None:
                ev = self.finalize_event = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        self.finalized = True

    def finalize_collect(self):
        """The second half of finalize_async() (call it under the stream finalize_begin() ran on): waits for the GPU, copies the
        lattices out, starts the host n-best and returns its future."""
        from concurrent.futures import Future
        if not self.finalized:
            raise RuntimeError("finalize_collect() before finalize_begin()")
        self._check_overflow()
        fut = Future()
        if self.nbest == 1:
            fut.set_result([[r] if self.frames_decoded()[u] > 0 else [] for u, r in enumerate(self.best_path(True))])
            return fut
        hdr = self._header()
        prev = self.__dict__.get("_last_nbest")
        if prev is not None and not prev.done():
            try:
                prev.exception()          # waits; whatever it raised was (or will be) delivered through that future
            except BaseException:         # noqa: BLE001
                pass
        cn, host = self._lattices()
        mapping_all = self.state.view(self.U, self.state_bytes)[:, self.off[1]:self.off[1] + 4 * (self.caps[0] + 1)].contiguous().view(torch.int32).cpu().numpy()
        if WfstSearch._pool is None:
            WfstSearch._pool = ThreadPoolExecutor(max_workers=_pool_threads())
        nbest = self.nbest

        def job():
            try:
                fut.set_result(self._nbest_host(nbest, hdr, cn, host, mapping_all))
            except BaseException as e:          # noqa: BLE001 -- delivered through the future
                fut.set_exception(e)
        import threading
        self._last_nbest = fut
        threading.Thread(target=job, daemon=True).start()
        return fut

    def _pinned(self, rows: int, cols: int) -> torch.Tensor:
        """Page-locked staging for the lattice copy, owned by this searcher and reused from utterance batch to utterance batch
        (a fresh pinned allocation is a ~50 ms driver call; torch's caching host allocator hands one out whenever its blocks
        are still referenced -- seen as one 80 ms finalize in six).  The previous batch's host n-best reads these buffers:
        finalize_async waits for it before they are overwritten."""
        pool = self.__dict__.setdefault("_pin_pool", {})
        t = pool.get(rows)
        if t is None or t.numel() < rows * cols:
            t = torch.empty(rows * max(1024, int(cols * 1.5)), dtype=torch.int32, pin_memory=True)
            pool[rows] = t
        return t[:rows * cols].view(rows, cols)

    def _lattices(self, cap_arcs: int = 1 << 18, cap_final: int = 1 << 13):
        """The pruned lattices of all utterances, compacted on the GPU (b2t_wfst_lattice) and copied out once."""
        U, dev = self.U, self.device
        while True:
            counts = torch.zeros((U, 5), dtype=torch.int32, device=dev)
            arcs = torch.empty((6, U, cap_arcs), dtype=torch.int32, device=dev)      # src, dst, ilabel, olabel | graph, acoustic (fp32 bits)
            fins = torch.empty((2, U, cap_final), dtype=torch.int32, device=dev)     # final state | final cost (fp32 bits)
            with torch.cuda.device(dev):
                N.check(self.lib.b2t_wfst_lattice(C.byref(self.cg), C.byref(self.co), ops._p(self.state), U, cap_arcs, cap_final,
                                                  ops._p(counts), ops._p(arcs[0]), ops._p(arcs[1]), ops._p(arcs[2]), ops._p(arcs[3]),
                                                  ops._p(arcs[4]), ops._p(arcs[5]), ops._p(fins[0]), ops._p(fins[1]), self._s()),
                        "b2t_wfst_lattice")
            cn = counts.cpu().numpy()
            if not cn[:, 4].any():
                break
            cap_arcs, cap_final = cap_arcs * 4, cap_final * 4      # a lattice did not fit: retry with more room
        # ONE compact copy: the utterances' arcs back to back (the slabs are U x cap_arcs, of which a few per cent are used:
        # copying U x max-arcs rows moved 134 MB for 29 MB of lattice and was most of finalize's GPU half).  One gather for
        # the six arc arrays and one for the finals, indices from the counts already on the host (no data-dependent sizes on
        # the device), one copy each into pinned memory.
        a_off = np.concatenate([[0], np.cumsum(cn[:, 1].astype(np.int64))]); f_off = np.concatenate([[0], np.cumsum(cn[:, 2].astype(np.int64))])

        def compact(slab, cap, off):
            tot = int(off[-1])
            if tot == 0:
                return np.zeros((slab.shape[0], 0), dtype=np.int32)
            n = torch.from_numpy(np.diff(off)).to(dev, non_blocking=True)
            base = torch.from_numpy(np.arange(U, dtype=np.int64) * cap - off[:-1]).to(dev, non_blocking=True)
            idx = torch.arange(tot, device=dev) + torch.repeat_interleave(base, n, output_size=tot)
            host = self._pinned(slab.shape[0], tot)
            host.copy_(slab.view(slab.shape[0], -1).index_select(1, idx), non_blocking=True)
            return host

        ha, hf = compact(arcs, cap_arcs, a_off), compact(fins, cap_final, f_off)
        torch.cuda.current_stream(dev).synchronize()
        ha, hf = (x.numpy() if isinstance(x, torch.Tensor) else x for x in (ha, hf))
        flat = [ha[0], ha[1], ha[2], ha[3], ha[4].view(np.float32), ha[5].view(np.float32), hf[0], hf[1].view(np.float32)]
        return cn, (flat, a_off, f_off)

    def _nbest_all(self, nbest: int, rescore=None):
        """rescore = (old grammar, new grammar, back-off label), wfst.HostFst handles arc-sorted by ilabel: the n-best of the lattice
        composed with both (BrainSpeechDecoder::Rescore, b2t_lattice_rescore_nbest_host) instead of the lattice's own."""
        hdr = self._header()
        prev = self.__dict__.get("_last_nbest")
        if prev is not None and not prev.done():
            try:
                prev.exception()          # waits; whatever it raised was (or will be) delivered through that future
            except BaseException:         # noqa: BLE001
                pass
        cn, host = self._lattices()
        mapping_all = self.state.view(self.U, self.state_bytes)[:, self.off[1]:self.off[1] + 4 * (self.caps[0] + 1)].contiguous().view(torch.int32).cpu().numpy()
        return self._nbest_host(nbest, hdr, cn, host, mapping_all, rescore)

    def _nbest_host(self, nbest, hdr, cn, host, mapping_all, rescore=None):
        """Host half of FinalizeSearch: nothing here touches the device or this object's state block."""
        (src, dst, il, ol, gr, ac, fs, fc), a_off, f_off = host
        # Everything a worker does outside the two C calls holds the interpreter lock, i.e. is serial across the utterances'
        # threads: addresses are computed once here (no per-utterance slices / ctypes casts), the output buffers of all
        # utterances are one allocation each, and Python only cuts flat lists into the n-best tuples at the end.
        U = self.U
        Fs = [int(hdr[u, 0]) for u in range(U)]
        cap = nbest * (2 * max(Fs + [0]) + 16) + 16
        ow = np.empty((U, cap), dtype=np.int32); oa = np.empty((U, cap), dtype=np.int32)
        ii = np.empty((U, cap), dtype=np.int32); it = np.empty((U, cap), dtype=np.int32)
        woff = np.zeros((U, nbest + 1), dtype=np.int32); aoff = np.zeros((U, nbest + 1), dtype=np.int32); io = np.zeros((U, nbest + 1), dtype=np.int32)
        costs = np.empty((U, 2 * nbest), dtype=np.float32)
        mapping_all = np.ascontiguousarray(mapping_all, dtype=np.int32)
        base = {k: v.ctypes.data for k, v in dict(src=src, dst=dst, il=il, ol=ol, gr=gr, ac=ac, fs=fs, fc=fc, ow=ow, oa=oa, ii=ii, it=it,
                                                  woff=woff, aoff=aoff, io=io, costs=costs, mp=mapping_all).items()}
        for k, v in dict(src=src, dst=dst, il=il, ol=ol, gr=gr, ac=ac, fs=fs, fc=fc).items():
            assert v.flags.c_contiguous and v.itemsize == 4, k
        a_off_l, f_off_l = [int(x) for x in a_off], [int(x) for x in f_off]
        cn_l = cn[:, :4].tolist()
        row = lambda name, u, width: base[name] + 4 * u * width
        beam = C.c_float(self.lattice_beam)
        mp_w = mapping_all.shape[1]

        def one(u):
            F = Fs[u]
            n_states, n_arcs, n_final, start = cn_l[u]
            if F == 0 or n_states == 0 or start < 0:
                return []
            ao, fo = 4 * a_off_l[u], 4 * f_off_l[u]
            if rescore is None:
                n = self.lib.b2t_lattice_nbest_host(n_states, start, n_arcs, base["src"] + ao, base["dst"] + ao, base["il"] + ao, base["ol"] + ao,
                                                    base["gr"] + ao, base["ac"] + ao, n_final, base["fs"] + fo, base["fc"] + fo, nbest, beam,
                                                    row("ow", u, cap), row("woff", u, nbest + 1), cap, row("oa", u, cap),
                                                    row("aoff", u, nbest + 1), cap, row("costs", u, 2 * nbest))
            else:
                n = self.lib.b2t_lattice_rescore_nbest_host(n_states, start, n_arcs, base["src"] + ao, base["dst"] + ao, base["il"] + ao,
                                                            base["ol"] + ao, base["gr"] + ao, base["ac"] + ao, n_final, base["fs"] + fo,
                                                            base["fc"] + fo, rescore[0]._h, rescore[1]._h, int(rescore[2]), nbest, beam,
                                                            row("ow", u, cap), row("woff", u, nbest + 1), cap, row("oa", u, cap),
                                                            row("aoff", u, nbest + 1), cap, row("costs", u, 2 * nbest), None)
            if n < 0:
                raise RuntimeError("b2t_lattice_nbest_host failed: " + N.last_error())
            # ConvertToInputs (ctc_wfst_beam_search.cc:162-188) for the n entries in one host call as well
            if self.lib.b2t_nbest_convert_to_inputs(row("oa", u, cap), row("aoff", u, nbest + 1), n, row("mp", u, mp_w), F, row("ii", u, cap),
                                                    row("it", u, cap), row("io", u, nbest + 1), cap) != 0:
                raise RuntimeError("b2t_nbest_convert_to_inputs failed: " + N.last_error())
            io_l, wo_l = io[u, :n + 1].tolist(), woff[u, :n + 1].tolist()
            inl, tl, wl, cl = ii[u, :io_l[n]].tolist(), it[u, :io_l[n]].tolist(), ow[u, :wo_l[n]].tolist(), costs[u, :2 * n].tolist()
            return [(inl[io_l[k]:io_l[k + 1]], tl[io_l[k]:io_l[k + 1]], wl[wo_l[k]:wo_l[k + 1]], -cl[2 * k], -cl[2 * k + 1]) for k in range(n)]

        # utterances are independent and the C call releases the GIL: one host thread per utterance up to the usable cores
        workers = min(self.U, _host_threads())
        if workers <= 1:
            return [one(u) for u in range(self.U)]
        if WfstSearch._pool is None:
            WfstSearch._pool = ThreadPoolExecutor(max_workers=_pool_threads())
        # largest lattice first: the call takes max(longest utterance, sum / threads), and a long one started last is all tail
        by_size = sorted(range(self.U), key=lambda u: -int(cn[u, 1]))
        res = dict(zip(by_size, WfstSearch._pool.map(one, by_size)))
        return [res[u] for u in range(self.U)]

    def _nbest_of(self, u, h, nbest=None):
        return self._nbest_all(nbest or self.nbest)[u]


class WfstPipeline:
    """Offline decoding of a stream of utterance batches at the GPU's pace: two WfstSearch instances on two HIP streams.  Batch
    b + 1 is reset and searched on the other stream -- behind batch b's FinalizeDecoding kernel (`finalize_event`): two
    cluster searches are never in flight together -- BEFORE the host waits for batch b's lattices, so lattice extraction, the
    copy to the host, the host's Python and the n-best extraction (host threads) of a batch all run under the next batch's
    search.  32 utterances x 111 frames with the production options: 18.5 ms per batch against 22.9 ms with one searcher and
    finalize_async().  No reference counterpart (the reference decodes one utterance per process, one CPU thread).

        pipe = WfstPipeline(graph, opts, U=32, max_frames=T + 8)
        for nbest_lists in pipe.decode(batches):      # batches: iterable of (logp [U, T, C] on the device, lens [U])
            ...                                        # the lists WfstSearch.finalize() returns, in batch order
    """

    def __init__(self, graph, opts, U: int, **kw):
        self.S = [WfstSearch(graph, opts, U=U, **kw), WfstSearch(graph, opts, U=U, **kw)]
        dev = self.S[0].device
        self.streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

    def _enqueue(self, i, batch, after=None):
        logp, lens = batch
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream(st.device))    # whoever produced logp
        with torch.cuda.stream(st):
            if after is not None:
                st.wait_event(after)
            self.S[i].reset(); self.S[i].search(logp, lens)

    def decode(self, batches):
        it = iter(batches)
        first = next(it, None)
        if first is None:
            return
        self._enqueue(0, first)
        i, pend = 0, None
        while True:
            cur = i % 2
            nxt = next(it, None)
            with torch.cuda.stream(self.streams[cur]):
                self.S[cur].finalize_begin()
            if nxt is not None:
                self._enqueue(1 - cur, nxt, after=self.S[cur].finalize_event)
            with torch.cuda.stream(self.streams[cur]):
                fut = self.S[cur].finalize_collect()
            if pend is not None:
                yield pend.result()
            pend = fut
            i += 1
            if nxt is None:
                break
        yield pend.result()


def convert_all_to_inputs(ali, off, n, mapping):
    """convert_to_inputs for the n alignments ali[off[k]:off[k+1]] of one utterance in one vectorised pass (100 n-best entries
    x ~100 frames in a Python loop per utterance were most of what was left of finalize): runs of equal labels, cut at the
    entries' boundaries; a run of a non-blank label gives (label - 1, frame of the run's last position).  Frames come from
    `mapping` (decoded frame -> input frame) for alignments that cover all decoded frames, else they are positions."""
    inps, tms = [[] for _ in range(n)], [[] for _ in range(n)]
    tot = int(off[n])
    if n == 0 or tot == 0:
        return inps, tms
    a = np.asarray(ali[:tot])
    offs = np.asarray(off[:n + 1], dtype=np.int64)
    brk = np.empty(tot, dtype=bool)
    brk[:-1] = a[1:] != a[:-1]
    brk[-1] = True
    last = offs[1:][offs[1:] > offs[:-1]] - 1        # last position of every non-empty entry
    brk[last] = True
    ends = np.flatnonzero(brk)
    labs = a[ends]
    keep = labs != 1                                  # ilabel 1 = blank
    ends, labs = ends[keep], labs[keep] - 1
    owner = np.searchsorted(offs[1:], ends, side="right")
    pos = ends - offs[owner]
    F = len(mapping)
    full = (offs[1:] - offs[:-1]) == F
    times = np.where(full[owner], np.asarray(mapping, dtype=np.int64)[np.minimum(pos, max(F - 1, 0))] if F else pos, pos)
    cut = np.searchsorted(owner, np.arange(n + 1))
    labs_l, times_l = labs.tolist(), times.tolist()
    for k in range(n):
        inps[k] = labs_l[cut[k]:cut[k + 1]]; tms[k] = times_l[cut[k]:cut[k + 1]]
    return inps, tms


def convert_rows_to_inputs(ali, frames, n):
    """convert_to_inputs for every row u of ali [U, max_len] / frames [U, max_len] (the first n[u] entries count), in one
    vectorised pass: runs of equal labels inside a row; a run of a non-blank label gives (label - 1, frame of its last entry)."""
    U = ali.shape[0]
    inps, tms = [[] for _ in range(U)], [[] for _ in range(U)]
    n = np.asarray(n, dtype=np.int64)
    tot = int(n.sum())
    if tot == 0:
        return inps, tms
    mask = np.arange(ali.shape[1])[None, :] < n[:, None]
    a, f = ali[mask], frames[mask]                       # row-major: the rows' prefixes back to back
    offs = np.concatenate([[0], np.cumsum(n)])
    brk = np.empty(tot, dtype=bool)
    brk[:-1] = a[1:] != a[:-1]
    brk[-1] = True
    last = offs[1:][n > 0] - 1
    brk[last] = True                                      # a run never continues into the next row
    ends = np.flatnonzero(brk)
    labs = a[ends]
    keep = labs != 1                                      # ilabel 1 = blank
    ends, labs = ends[keep], labs[keep] - 1
    owner = np.searchsorted(offs[1:], ends, side="right")
    cut = np.searchsorted(owner, np.arange(U + 1))
    labs_l, times_l = labs.tolist(), f[ends].tolist()
    for u in range(U):
        inps[u] = labs_l[cut[u]:cut[u + 1]]; tms[u] = times_l[cut[u]:cut[u + 1]]
    return inps, tms


def convert_to_inputs(alignment, frames):
    """CtcWfstBeamSearch::ConvertToInputs (ctc_wfst_beam_search.cc:162-188): drop blanks (ilabel 1), merge repeats, ilabel - 1;
    the time of a unit is the input frame of its LAST repeated label."""
    inp: List[int] = []; tm: List[int] = []
    cur, n = 0, len(alignment)
    while cur < n:
        while cur < n and alignment[cur] - 1 == 0:
            cur += 1
        while cur + 1 < n and alignment[cur + 1] == alignment[cur]:
            cur += 1
        if cur < n:
            inp.append(int(alignment[cur]) - 1); tm.append(int(frames[cur])); cur += 1
    return inp, tm
