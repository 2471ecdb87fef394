"""Development aid: section timings inside k_rs_scatter (library built with `make STAMPS=1`, SCALOAM_LIB pointing at it)."""
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
vg = S.VoxelGrid()
buf = (ctypes.c_longlong * 32)()
for k in range(4):
    f = reg.laserCloudHandler(world.scan(10 + k))
    for name, cloud, leaf in (('keyframe 0.4', f['cloud'], 0.4), ('lessFlat 0.8', f['less_flat'], 0.8)):
        out = vg.filter(cloud, leaf)
        lib.scal_debug_stamps_radix(buf)
        sv = np.array(buf[:6], dtype=np.int64)
        nv = ['bases', 'clear', 'sync', 'load+rank', 'sync2', 'scatter']
        print(f'k_rs_scatter pass 0, {name}, n={len(cloud)} -> {len(out)} us:', ' '.join(f'{n}={(sv[i+1]-sv[i])*0.01:.1f}' for i, n in enumerate(nv[:5])), 'total', (sv[5] - sv[0]) * 0.01)
