import sys, os, time
import numpy as np
sys.path.insert(0, 'sc-a-loam_amd/python'); sys.path.insert(0, 'tools/synth'); sys.path.insert(0, 'oracle')
import scaloam as S, scansynth
world = scansynth.World(scansynth.HDL64, 205, threads=16)
reg = S.ScanRegistration(S.HDL64, 5.0)
od, mp = S.LaserOdometry(), S.LaserMapping(0.4, 0.8)
tA = tB = tC = 0.0
paths = []
for k in range(40):
    xyz = world.scan(k)
    t0 = time.perf_counter(); f = reg.laserCloudHandler(xyz); t1 = time.perf_counter()
    c = f['cloud']
    r = od.step(c[f['sharp']], c[f['less_sharp']], c[f['flat']], f['less_flat']); t2 = time.perf_counter()
    qw, tw = r[2], r[3]
    out = mp.process(c[f['less_sharp']], f['less_flat'], c, qw, tw); t3 = time.perf_counter()
    st = out[2]
    if k >= 5:
        tA += t1 - t0; tB += t2 - t1; tC += t3 - t2
    paths.append(st.insert_path)
print('A %.2f B %.2f C %.2f ms per scan; insert paths %s' % (tA / 35 * 1e3, tB / 35 * 1e3, tC / 35 * 1e3, paths))
