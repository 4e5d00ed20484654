import sys, os, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import prographmsa_amd as pg
import faulthandler; faulthandler.dump_traceback_later(15, exit=True)
ctx = pg.Context(0)
case = sys.argv[1]
rng = np.random.default_rng(1)
dim = 20
score = rng.integers(-4, 8, (21, 21)).astype(np.int32).reshape(-1)
if case == 'a': lens = [5, 5]
elif case == 'b': lens = [100, 90]
elif case == 'c': lens = [5, 5, 7, 9]
elif case == 'd': lens = [0, 5]
elif case == 'e': lens = [64, 65, 129]
seqs = [rng.integers(0, 21, L).astype(np.int8) for L in lens]
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
syms = np.concatenate(seqs).astype(np.int8)
pairs = [(i, j) for i in range(len(lens)) for j in range(len(lens)) if i != j]
pi = np.array([p[0] for p in pairs], np.uint32); pj = np.array([p[1] for p in pairs], np.uint32)
counts = np.zeros(len(pairs) * 400, np.int32); gaps = np.zeros(len(pairs), np.uint32)
P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
t0 = time.time()
rc = pg.lib.pgm_nw_pairs_batch(ctx.handle, dim, P(score, C.c_int32), -10, -2, len(lens), P(syms, C.c_int8), P(offs, C.c_uint32), len(pairs), P(pi, C.c_uint32), P(pj, C.c_uint32), P(counts, C.c_int32), P(gaps, C.c_uint32))
print(case, 'rc', rc, 'time', time.time() - t0, 'gaps', gaps, 'sum', counts.sum(), flush=True)
import oracle_lib
co, go = oracle_lib.nw_pairs(dim, score, -10, -2, syms, offs, pi, pj)
print('match', np.array_equal(co.reshape(-1), counts), np.array_equal(go, gaps), flush=True)
