import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 1] > 0]
t0, t1 = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64)
base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0
samples = (a[:, 2] >> np.uint64(32)).astype(np.int64)
work = samples > 0
print("waves", len(a), "with samples", work.sum(), "without", (~work).sum())
print("kernel end %.1f us; last wave WITH samples ends %.1f; first wave WITHOUT samples starts %.1f; sum time of empty waves %.0f us (%.2f %% of all wave time)" % (
    t1.max(), t1[work].max(), t0[~work].min(), (t1 - t0)[~work].sum(), 100 * (t1 - t0)[~work].sum() / (t1 - t0).sum()))
for lo in np.linspace(0, t1.max(), 11)[:-1]:
    hi = lo + t1.max() / 10
    ov = lambda m: np.clip(np.minimum(t1[m], hi) - np.maximum(t0[m], lo), 0, None).sum() / (hi - lo)
    print("  t=%7.1f-%7.1f  resident waves with samples %7.1f  empty %7.1f   empty waves started in the bucket %d" % (lo, hi, ov(work), ov(~work), ((t0 >= lo) & (t0 < hi) & ~work).sum()))
