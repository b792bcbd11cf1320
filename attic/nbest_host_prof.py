import sys, time, os
import numpy as np
sys.path.insert(0, '/root/repo/nejm-brain-to-text_amd')
import b2t_native as N
import wfst_decoder as W
z = np.load('/root/repo/build/lat32.npz')
class Fake: pass
f = Fake(); f.lib = N.load(); f.U = 32; f.lattice_beam = float(z['lattice_beam']); f.nbest = 100
host = ([z[k] for k in ('src','dst','il','ol','gr','ac','fs','fc')], z['a_off'], z['f_off'])
for rep in range(4):
    t0 = time.perf_counter()
    r = W.WfstSearch._nbest_host(f, 100, z['hdr'], z['cn'], host, z['mapping'])
    print('nbest_host all', round((time.perf_counter() - t0) * 1e3, 2), 'ms', os.cpu_count(), W._host_threads(), W._pool_threads())
import cProfile, pstats
W.WfstSearch._pool = None
orig = W._host_threads
W._host_threads = lambda: 1
cProfile.run("W.WfstSearch._nbest_host(f, 100, z['hdr'], z['cn'], host, z['mapping'])", '/tmp/nb.prof')
pstats.Stats('/tmp/nb.prof').sort_stats('cumtime').print_stats(14)
W._host_threads = orig
W.WfstSearch._pool = None
import threading
real = f.lib.b2t_lattice_nbest_host
log = []
class Wrap:
    def __getattr__(self, k):
        fn = getattr(N.load(), k)
        if k != "b2t_lattice_nbest_host": return fn
        def g(*a):
            t0 = time.perf_counter(); r = fn(*a); log.append((threading.get_ident() % 1000, round((t0 - T0) * 1e3, 1), round((time.perf_counter() - T0) * 1e3, 1))); return r
        return g
f.lib = Wrap()
T0 = time.perf_counter()
W.WfstSearch._nbest_host(f, 100, z['hdr'], z['cn'], host, z['mapping'])
print("total", round((time.perf_counter() - T0) * 1e3, 1))
for l in sorted(log, key=lambda x: x[1])[:14]: print(l)
