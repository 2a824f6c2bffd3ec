"""analyze an OVR_HIP_TRACE dump: per-wave residency intervals (s_memrealtime ticks, 100 MHz) -> concurrency over time"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 1] > 0]
t0, t1 = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64)
base = t0.min()
t0 -= base; t1 -= base
dur = (t1 - t0) / 100.0  # us
samples = (a[:, 2] >> np.uint64(32)).astype(np.int64); shaded = (a[:, 2] & np.uint64(0xffffffff)).astype(np.int64); shadow = a[:, 3].astype(np.int64)
T = t1.max() / 100.0
print(f"waves {len(a)}  kernel span {T:.1f} us  sum(wave time) {dur.sum():.0f} us  avg concurrency {dur.sum()/T:.1f} waves")
print("wave duration us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
edges = np.linspace(0, T, 21)
for i in range(20):
    lo, hi = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(t1 / 100.0, hi) - np.maximum(t0 / 100.0, lo), 0, None).sum() / (hi - lo)
    print(f"  t={lo:8.1f}-{hi:8.1f} us  resident waves {ov:8.1f}")
k = np.argsort(-dur)[:8]
for i in k:
    print(f"  heavy wave: {dur[i]:.1f} us  samples {samples[i]}  shaded {shaded[i]}  shadow {shadow[i]}")
print("work totals: samples %d shaded %d shadow %d" % (samples.sum(), shaded.sum(), shadow.sum()))
