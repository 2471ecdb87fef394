"""Development aid: section timings inside k_lm_solve_map's rounds (library built with `make STAMPS=1`, SCALOAM_LIB pointing at it).
Sections per round: evaluate | wave reduce + publish | collect (poll) | sum of partials | serial trust-region step."""
import ctypes, sys, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'sc-a-loam_amd', 'python'))
sys.path.insert(0, os.path.join(HERE, 'synth'))
import scaloam as S
import scansynth
lib = S.lib()
world = scansynth.World(scansynth.HDL64, 205, threads=16)
reg = S.ScanRegistration(S.HDL64, 5.0)
od, mp = S.LaserOdometry(), S.LaserMapping(0.4, 0.8)
buf = (ctypes.c_longlong * 32)()
names = ['eval', 'reduce+publish', 'collect', 'sum', 'serial']
for k in range(12):
    reg.laserCloudHandler(world.scan(10 + k))
    _, _, qw, tw, _ = od.step_features(reg)
    _, _, st = mp.process_features(reg, qw, tw)
    lib.scal_debug_stamps_map(buf)
    sv = np.array(buf[:24], dtype=np.int64)
    ex = np.array(buf[24:32], dtype=np.int64)
    if k < 4:
        continue
    rows = []
    for r in range(4):
        s = sv[6 * r:6 * r + 6]
        nxt = sv[6 * (r + 1)] if r < 3 else s[5]
        rows.append(' '.join(f'{n}={(s[i + 1] - s[i]) * 0.01:.1f}' for i, n in enumerate(names)) + f' to-next-round={(nxt - s[5]) * 0.01:.1f}')
    print(f'scan {k}: blocks {st.n_edge[1]}+{st.n_plane[1]} iters {list(st.lm_iters)}')
    print(f'   serial step of round 1: copy in {(ex[1] - ex[0]) * 0.01:.2f}, trust-region logic + candidate {(ex[6] - ex[1]) * 0.01:.2f} (of which the Cholesky solve of the last round: {(ex[3] - ex[2]) * 0.01:.2f}), copy out {(ex[7] - ex[6]) * 0.01:.2f} us')
    if False: print(f'   k_assoc_fit: edge thread {(ex[1] - ex[0]) * 0.01:.1f} us, plane thread {(ex[3] - ex[2]) * 0.01:.1f} us; knn5_half: edge query {(ex[5] - ex[4]) * 0.01:.1f} us')
    for r, row in enumerate(rows):
        print(f'   round {r}: {row}   round total {(sv[6 * r + 5] - sv[6 * r]) * 0.01:.1f} us')
