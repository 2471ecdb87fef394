"""Development aid: section timings of k_merge_prepare's surf workgroup (library built with `make STAMPS=1`)."""
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
names = ['fill', 'sort', 'heads', 'search', 'flags', 'scan+store']
for k in range(8):
    reg.laserCloudHandler(world.scan(k))
    qlc, tlc, qw, tw, st = od.step_features(reg)
    qm, tm, ms = mp.process_features(reg, qw, tw)
    lib.scal_debug_stamps_map(buf)
    st = np.array(buf[26:32], dtype=np.int64)
    if k >= 3:
        print(k, ' '.join(f'{n}={v:.2f}' for n, v in zip(names, np.diff(st) * 0.01)), 'total', (st[5] - st[0]) * 0.01, 'stack', ms.n_surf_stack, 'map', ms.n_map_surf_total, 'path', ms.insert_path)
