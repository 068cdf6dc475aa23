import os
for k in ("OMP_NUM_THREADS","OPENBLAS_NUM_THREADS","MKL_NUM_THREADS"): os.environ[k]="1"
import sys, time, cProfile, pstats
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from blackbox_amd import overscan, synth, settings
rs=np.random.RandomState(1)
ysz,xsz=5280,1320; dy=5300; dx=1500; hos_rows=dy-ysz-10
mean_vos=(1000+0.001*np.arange(dy)+rs.normal(0,0.8,dy)).astype(np.float64)
hos=(1000+rs.normal(0,8,(hos_rows,dx))).astype(np.float32)
hos[:, :30]+= np.linspace(20,0,30)[None,:]
args=(3,mean_vos,hos,ysz,xsz,settings.voscan_poldeg,'ML1',2000,'f32seq')
overscan.channel_solve(args)
t=time.perf_counter()
for _ in range(50): overscan.channel_solve(args)
print('ms/channel', (time.perf_counter()-t)/50*1e3)
pr=cProfile.Profile(); pr.enable()
for _ in range(50): overscan.channel_solve(args)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(18)
