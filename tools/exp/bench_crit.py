# bench.py against the -DZT_CRIT library (tools/build_crit.sh), then the hub-hop stamps of the LAST T-PPR launch of the run:
# hop times inside the pipeline (beside the aggregation, on the T-PPR stream's CUs), for the edges of the launch's first batch
#   python3 tools/exp/bench_crit.py --steps 40 --cpu-edges 0
import os, sys, runpy, ctypes as C
os.environ["ZT_CRIT_MULTI"] = "1"
import numpy as np
sys.path.insert(0, '/root/repo')
from zebra_amd import _capi
_capi.LIB_PATH = '/root/repo/tools/out/libzebra_crit.so'
sys.argv = ['bench.py'] + sys.argv[1:]
try:
    runpy.run_path('/root/repo/bench.py', run_name='__main__')
except SystemExit:
    pass
lib = _capi.lib()
c2 = np.zeros((8200, 16), np.int64)
lib.zt_debug_crit(c2.ctypes.data_as(C.c_void_p), C.c_int(8200))
p = lambda a: np.percentile(a, [10, 50, 90]).round(0)
for mo in (0, 1):
    c = c2[mo * 4096:(mo + 1) * 4096]
    c = c[(c[:, 3] > 0) & (c[:, 0] > 0)]
    c = c[np.argsort(c[:, 3])]
    c = c[c[:, 3] > c[:, 3].max() - 4_200_000]        # stamps of the last launch over 3+ batches only (its first batch's edges)
    ch = c[:, 15] // 100000
    big = np.bincount(ch.astype(np.int64)).argmax()   # the hub's chain: the one with the most hops
    c = c[ch == big]
    c = c[np.argsort(c[:, 15])]
    c = c[np.concatenate([[True], np.diff(c[:, 15]) == 1])]
    d = np.diff(c[:, 3])
    d = d[d < 60000]                                  # (hops of other batches' edges in between are not stamped)
    lean = c[c[:, 7] == 1]
    non = c[c[:, 7] != 1]
    print("model %d: hops that did not take the lean section, by reason (1 partner side not prepared, 2 norm, 3 key match, 4 new key / NaN, 5 row not sorted, 6 table clash): %s" % (
        mo, dict(zip(*np.unique(non[:, 11], return_counts=True)))))
    print("model %d: %d stamped hops; publication to publication %s mean %.0f; critical section %s; lean %d: preparation %s slack %s tail %s" % (
        mo, len(c), p(d), d.mean(), p(c[:, 3] - c[:, 0]), len(lean), p(lean[:, 5] - lean[:, 4]), p(lean[:, 0] - lean[:, 5]), p(lean[:, 6] - lean[:, 3])))
    # slow hops of the hub's chain: a long preparation = the partner's row was not there (a general task or another chain
    # had to write it first); a normal preparation with no slack = every wave of the chain was busy
    dd = np.diff(c[:, 3]); slow = np.where(dd > 4500)[0] + 1
    prep = c[:, 5] - c[:, 4]
    kind_wait = [t for t in slow if c[t, 7] == 1 and prep[t] > 9000]
    kind_busy = [t for t in slow if c[t, 7] == 1 and prep[t] <= 9000]
    kind_gen = [t for t in slow if c[t, 7] != 1]
    ex = lambda ts: sum(dd[t - 1] - np.median(dd) for t in ts)
    print("model %d: slow hops %d of %d (excess over the median %.0f clocks = %.0f %% of the chain): partner's row late %d (%.0f), waves busy %d (%.0f), general code %d (%.0f)" % (
        mo, len(slow), len(c), ex(slow), 100 * ex(slow) / max(1, dd.sum()), len(kind_wait), ex(kind_wait), len(kind_busy), ex(kind_busy), len(kind_gen), ex(kind_gen)))
    head = (c[:, 13] // 100000) / 2.0; edge = c[:, 13] % 100000          # (two models: task index / 2 = edge)
    okh = c[:, 13] > 0
    print("model %d: general queue head - this hop's edge, when the hop was ready to wait for its row (edges; > 0: the queue is ahead): %s" % (mo, p((head - edge)[okh])))
    print("model %d: partner's side not prepared because (7 no norm prediction, 8 both slot functions clash, 9 NaN): %s; lean section left after the old front half would have succeeded: %d" % (
        mo, dict(zip(*np.unique(non[:, 9], return_counts=True))), int((non[:, 11] == 0).sum())))
    print("model %d: lean section left at (1 not sorted / norm, 2 no prune, 3 alternate may be in the row, 4 key match / NaN, 5 picked member kept): %s" % (
        mo, dict(zip(*np.unique(non[non[:, 10] < 100][:, 10], return_counts=True)))))
r = c2[8199]
print("model 0, all chains and launches since the library was loaded: lean %d; left at 1 not sorted / norm %d, 2 no prune %d, 3 alternate may be in the row %d, 4 key match / NaN %d, 5 picked member kept %d; lean not tried: prepared but left (sum of 1-5) %d, no norm prediction %d, slot functions clash %d, NaN / k %d" % tuple(int(x) for x in (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9])))
