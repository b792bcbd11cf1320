import sys, time, os, ctypes as C
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '/root/repo/nejm-brain-to-text_amd')
import b2t_native as N
lib = N.load()
z = np.load('/root/repo/build/lat32.npz')
a_off, f_off, cn, hdr = z["a_off"], z["f_off"], z["cn"], z["hdr"]
arrs = {k: z[k] for k in ("src", "dst", "il", "ol", "gr", "ac", "fs", "fc")}
P = lambda x: x.ctypes.data_as(C.c_void_p)
def prep(u):
    F = int(hdr[u, 0]); n_states, n_arcs, n_final, start = (int(v) for v in cn[u, :4])
    a = [np.ascontiguousarray(arrs[k][a_off[u]:a_off[u] + n_arcs]) for k in ("src", "dst", "il", "ol", "gr", "ac")]
    f_s = np.ascontiguousarray(arrs["fs"][f_off[u]:f_off[u] + n_final]); f_c = np.ascontiguousarray(arrs["fc"][f_off[u]:f_off[u] + n_final])
    nb = 100; cap = nb * (2 * F + 16) + 16
    outs = [np.zeros(cap, np.int32), np.zeros(nb + 1, np.int32), np.zeros(cap, np.int32), np.zeros(nb + 1, np.int32), np.zeros(2 * nb, np.float32)]
    return (n_states, start, n_arcs, a, n_final, f_s, f_c, cap, outs)
preps = [prep(u) for u in range(32)]
def call(u):
    n_states, start, n_arcs, a, n_final, f_s, f_c, cap, o = preps[u]
    return lib.b2t_lattice_nbest_host(n_states, start, n_arcs, P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(a[4]), P(a[5]), n_final, P(f_s), P(f_c), 100,
                                      C.c_float(8.0), P(o[0]), P(o[1]), cap, P(o[2]), P(o[3]), cap, P(o[4]))
for n in (1, 2, 4, 8, 12, 16):
    with ThreadPoolExecutor(n) as ex:
        list(ex.map(call, range(32)))
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); list(ex.map(call, range(32))); best = min(best, time.perf_counter() - t0)
    print(n, "threads:", round(best * 1e3, 1), "ms")
