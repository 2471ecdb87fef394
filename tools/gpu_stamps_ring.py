"""Development aid: section timings inside k_ring for ring 32 (library built with `make STAMPS=1`, SCALOAM_LIB pointing at it)."""
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
buf = (ctypes.c_longlong * 32)()
names = ['load ring + gap flags', 'six segment sorts', 'greedy picks', 'lessFlat count + bbox', 'voxel keys', 'block sort', 'centroids']
for k in range(5):
    f = reg.laserCloudHandler(world.scan(10 + k))
    lib.scal_debug_stamps_features(buf)
    sv = np.array(buf[:8], dtype=np.int64)
    order = [7, 0, 1, 2, 3, 4, 5, 6]
    t = [sv[i] for i in order]
    print(f'scan {k}: ' + ' | '.join(f'{n} {(t[i + 1] - t[i]) * 0.01:.1f}' for i, n in enumerate(names)) + f' | total {(t[-1] - t[0]) * 0.01:.1f} us; lessFlat {len(f["less_flat"])}')
