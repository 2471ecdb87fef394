#!/usr/bin/env python3
"""Per-stream chain latency from bench.py --timeline FILE --timeline-kernels first,last,...: for every stream, the time from the
start of a chain's first kernel to the end of its last kernel, and the period between chains (unprofiled run, HIP events)."""
import sys, collections, statistics as st
rows = []
for l in open(sys.argv[1]):
    n, a, b, s = l.strip().split(",")
    rows.append((float(a) * 1e3, float(b) * 1e3, n, s))
rows.sort()
pairs = [p.split(":") for p in sys.argv[2:]]  # label:first:last
for label, first, last in pairs:
    streams = collections.Counter(s for a, b, n, s in rows if n == first)
    for s in streams:
        F = [(a, b) for a, b, n, ss in rows if n == first and ss == s]
        L = [(a, b) for a, b, n, ss in rows if n == last and ss == s]
        d = []
        j = 0
        for a, b in F:
            while j < len(L) and L[j][1] < a:
                j += 1
            if j < len(L):
                d.append(L[j][1] - a)
        per = [F[i + 1][0] - F[i][0] for i in range(len(F) - 1)]
        if d and per:
            print(f"{label:8s} n {len(d):4d}  chain median {st.median(d):6.0f} p10 {sorted(d)[len(d)//10]:6.0f} p90 {sorted(d)[len(d)*9//10]:6.0f}   period median {st.median(per):6.0f}")
