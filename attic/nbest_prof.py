#!/usr/bin/env python3
"""Times b2t_lattice_nbest_host on the lattices dumped by fin_probe.py (gpurun_out/lat32.npz), one thread, per utterance."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import b2t_native as N


def main():
    lib = N.load()
    z = np.load(os.path.join(ROOT, "gpurun_out", "lat32.npz"))
    a_off, f_off, cn, hdr = z["a_off"], z["f_off"], z["cn"], z["hdr"]
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    nbest = int(os.environ.get("NBEST", "100"))
    tot = 0.0
    rows = []
    for u in range(len(cn)):
        F = int(hdr[u, 0]); n_states, n_arcs, n_final, start = (int(v) for v in cn[u, :4])
        a = [np.ascontiguousarray(z[k][a_off[u]:a_off[u] + n_arcs]) for k in ("src", "dst", "il", "ol", "gr", "ac")]
        f_s = np.ascontiguousarray(z["fs"][f_off[u]:f_off[u] + n_final]); f_c = np.ascontiguousarray(z["fc"][f_off[u]:f_off[u] + n_final])
        w_cap = a_cap = nbest * (2 * F + 16) + 16
        ow = np.zeros(w_cap, np.int32); oa = np.zeros(a_cap, np.int32); woff = np.zeros(nbest + 1, np.int32); aoff = np.zeros(nbest + 1, np.int32)
        costs = np.zeros(2 * nbest, np.float32)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            n = lib.b2t_lattice_nbest_host(n_states, start, n_arcs, P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(a[4]), P(a[5]), n_final, P(f_s), P(f_c),
                                           nbest, C.c_float(float(z["lattice_beam"])), P(ow), P(woff), w_cap, P(oa), P(aoff), a_cap, P(costs))
            best = min(best, time.perf_counter() - t0)
        tot += best
        rows.append((u, n_states, n_arcs, n, round(best * 1e3, 2), __import__("zlib").crc32(ow[:woff[n]].tobytes() + oa[:aoff[n]].tobytes() + costs[:2 * n].tobytes()), round(float(costs[0]), 4), round(float(costs[2 * (n - 1)]), 4)))
    for r in rows:
        print(r)
    print("total ms (1 thread):", round(tot * 1e3, 1))


if __name__ == "__main__":
    main()
