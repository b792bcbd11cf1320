#!/usr/bin/env python3
"""Where the GPU half of WfstSearch.finalize_async goes (tools/bench_wfst.py workload): finalize kernel, header read, lattice
kernel, counts read, gathers, copies.  Prints ms per piece (best of 3)."""
import ctypes as C
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import b2t_native as N
import b2t_ops as ops
import bench_wfst as BW
from wfst_decoder import WfstSearch


def main():
    lib = N.load(); dev = torch.device("cuda:0")
    prons, words, arpa, g, seqs, logits, lens, build_s = BW.make()
    U, T, Cc = logits.shape
    lg, pri, lp = BW._logp(logits, dev, lib)
    S = WfstSearch(g, BW.Opt, U=U, prune_interval=25, prune_min_fill=0.5, max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24)
    sync = torch.cuda.synchronize
    best = {}

    def tick(name, t0):
        sync(); dt = (time.perf_counter() - t0) * 1e3
        best[name] = min(best.get(name, 1e9), dt)
        return time.perf_counter()

    for rep in range(4):
        S.reset(); S.search(lp, lens); sync()
        if rep == 0:
            st = S.state.view(S.U, S.state_bytes)
            F0 = int(S._header()[0, 0])
            lo = st[0, S.off[3]:S.off[3] + 4 * (2 * F0 + 3)].contiguous().view(torch.int32).cpu().numpy()
            to = st[0, S.off[2]:S.off[2] + 4 * (F0 + 2)].contiguous().view(torch.int32).cpu().numpy()
            eps = lo[1:2 * F0 + 2:2] - lo[0:2 * F0 + 1:2]; emi = lo[2:2 * F0 + 2:2] - lo[1:2 * F0 + 1:2]
            print("utt 0: frames", F0, "tokens/frame mean", np.diff(to[:F0 + 2]).mean().round(0), "max", np.diff(to[:F0 + 2]).max(),
                  "| eps links/frame mean", eps.mean().round(0), "max", eps.max(), "| emitting links/frame mean", emi.mean().round(0), "max", emi.max())
        t = time.perf_counter()
        N.check(lib.b2t_wfst_finalize(C.byref(S.cg), C.byref(S.co), ops._p(S.state), S.U, S._s()), "fin"); t = tick("finalize_kernel", t)
        S.finalized = True
        S._check_overflow(); t = tick("check_overflow", t)
        hdr = S._header(); t = tick("header", t)
        cn, host = S._lattices(); t = tick("lattices_total", t)
        mapping_all = S.state.view(S.U, S.state_bytes)[:, S.off[1]:S.off[1] + 4 * (S.caps[0] + 1)].contiguous().view(torch.int32).cpu().numpy()
        t = tick("mapping", t)
        # pieces of _lattices again, separately
        cap_arcs, cap_final = 1 << 18, 1 << 13
        counts = torch.zeros((U, 5), dtype=torch.int32, device=dev)
        arcs = torch.empty((6, U, cap_arcs), dtype=torch.int32, device=dev); fins = torch.empty((2, U, cap_final), dtype=torch.int32, device=dev)
        t = tick("  alloc", t)
        N.check(lib.b2t_wfst_lattice(C.byref(S.cg), C.byref(S.co), ops._p(S.state), U, cap_arcs, cap_final, ops._p(counts),
                                     ops._p(arcs[0]), ops._p(arcs[1]), ops._p(arcs[2]), ops._p(arcs[3]), ops._p(arcs[4]), ops._p(arcs[5]),
                                     ops._p(fins[0]), ops._p(fins[1]), S._s()), "lat")
        t = tick("  lattice_kernel", t)
        cn = counts.cpu().numpy(); t = tick("  counts", t)
        a_off = np.concatenate([[0], np.cumsum(cn[:, 1].astype(np.int64))]); tot = int(a_off[-1])
        n = torch.from_numpy(np.diff(a_off)).to(dev); base = torch.from_numpy(np.arange(U, dtype=np.int64) * cap_arcs - a_off[:-1]).to(dev)
        idx = torch.arange(tot, device=dev) + torch.repeat_interleave(base, n, output_size=tot); t = tick("  index", t)
        sel = arcs.view(6, -1).index_select(1, idx); t = tick("  gather", t)
        hostb = torch.empty((6, tot), dtype=torch.int32, pin_memory=True); t = tick("  pinned_alloc", t)
        hostb.copy_(sel, non_blocking=True); t = tick("  copy", t)
        fut = S._nbest_host(S.nbest, hdr, cn, host, mapping_all); t = tick("nbest_host", t)
    print({k: round(v, 3) for k, v in best.items()}, "arcs", tot, "MB", tot * 24 / 1e6)
    if os.environ.get("B2T_DUMP_LAT"):
        (src, dst, il, ol, gr, ac, fs, fc), a_off, f_off = host
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "lat32.npz"), src=src, dst=dst, il=il, ol=ol, gr=gr, ac=ac, fs=fs, fc=fc,
                            a_off=a_off, f_off=f_off, cn=cn, hdr=hdr, mapping=mapping_all, lattice_beam=np.float32(S.lattice_beam))


if __name__ == "__main__":
    main()
