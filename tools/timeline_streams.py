#!/usr/bin/env python3
"""Development aid: per-stream summary of a bench.py --timeline dump (all kernels of six timed steps): busy time per scan, span, and
the kernels of every stream in launch order for the middle scan."""
import sys, collections
rows = []
for l in open(sys.argv[1]):
    n, a, b, s = l.strip().split(",")
    rows.append((float(a) * 1e3, float(b) * 1e3, n, s))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
by = collections.defaultdict(list)
for r in rows:
    by[r[3]].append(r)
t0 = rows[0][0]
for s, rs in by.items():
    busy = sum(b - a for a, b, n, _ in rs)
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    small = [g for g in gaps if g < 20]
    print(f"stream {s}: {len(rs)} kernels, busy {busy / steps:7.1f} us/scan, span {rs[-1][1] - rs[0][0]:8.1f} us, "
          f"{len(rs) / steps:5.1f} kernels/scan, median in-chain gap {sorted(small)[len(small) // 2] if small else 0:5.1f} us, sum of gaps < 20 us {sum(small) / steps:6.1f} us/scan")
    names = collections.Counter(n for _, _, n, _ in rs)
    print("    " + ", ".join(f"{n} x{c // steps if c >= steps else c}" for n, c in names.most_common(14)))
if len(sys.argv) > 3:
    for s, rs in by.items():
        print("stream", s)
        for a, b, n, _ in rs[: int(sys.argv[3])]:
            print(f"   {a - t0:9.1f} {b - a:7.1f}  {n}")
