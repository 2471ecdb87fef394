"""Development aid: per-round section timings of the map's k_lm_solve (library built with `make STAMPS=1`)."""
import ctypes, sys, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'sc-a-loam_amd', 'python'))
sys.path.insert(0, os.path.join(HERE, 'synth'))
import scaloam as S
import scansynth
lib = S.lib()
world = scansynth.World(scansynth.HDL64, 205)
reg = S.ScanRegistration(S.HDL64, 5.0)
od = S.LaserOdometry()
mp = S.LaserMapping(0.4, 0.8)
buf = (ctypes.c_longlong * 32)()
names = ['eval', 'blockred', 'barrier', 'sum', 'tail']
for k in range(8):
    reg.laserCloudHandler(world.scan(k))
    qlc, tlc, qw, tw, st = od.step_features(reg)
    qm, tm, ms = mp.process_features(reg, qw, tw)
    lib.scal_debug_stamps_map(buf)
    st = np.array(buf[:24], dtype=np.int64).reshape(4, 6)
    if k >= 4:
        for r in range(4):
            d = np.diff(st[r]) * 0.01
            print(k, 'round', r, ' '.join(f'{n}={v:.2f}' for n, v in zip(names, d)), 'total', (st[r, 5] - st[r, 0]) * 0.01, 'blocks', list(ms.n_edge), list(ms.n_plane), list(ms.lm_iters))
