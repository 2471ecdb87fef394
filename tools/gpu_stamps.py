"""Development aid: section timings inside k_ring / k_vox_small (library built with `make STAMPS=1`)."""
import ctypes, sys, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'sc-a-loam_amd', 'python'))
sys.path.insert(0, os.path.join(HERE, 'synth'))
import scaloam as S
import scansynth

lib = S.lib()
world = scansynth.World(scansynth.HDL64, 205)
reg = S.ScanRegistration(S.HDL64 if hasattr(S, 'HDL64') else 2, 0.1)
vg = S.VoxelGrid()
buf = (ctypes.c_longlong * 32)()
for k in range(4):
    xyz = world.scan(10 + k)
    f = reg.laserCloudHandler(xyz)
    lib.scal_debug_stamps_features(buf)
    st = np.array(buf[:8], dtype=np.int64)
    names = ['init', 'sort6', 'picks', 'lf_bbox', 'lf_keys', 'lf_sort', 'lf_centroids']
    order = [7, 0, 1, 2, 3, 4, 5, 6]
    d = [(st[order[i + 1]] - st[order[i]]) * 0.01 for i in range(7)]
    print('k_ring ring32 us:', ' '.join(f'{n}={v:.1f}' for n, v in zip(names, d)), 'total', (st[6] - st[7]) * 0.01)
    cl = f['cloud'] if 'cloud' in f else None
    ls = cl[f['less_sharp']] if cl is not None else None
    if ls is not None:
        out = vg.filter(ls, 0.4)
        lib.scal_debug_stamps_voxel(buf)
        sv = np.array(buf[:6], dtype=np.int64)
        nv = ['bbox', 'keys', 'sort', 'preload+heads', 'centroids']
        print('k_vox_small n=%d us:' % len(ls), ' '.join(f'{n}={(sv[i+1]-sv[i])*0.01:.1f}' for i, n in enumerate(nv)), 'total', (sv[5] - sv[0]) * 0.01)
