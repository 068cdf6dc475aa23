import sys,time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
if __name__=='__main__':
    from blackbox_amd import overscan
    from blackbox_amd.pipeline import HostPool
    rs=np.random.RandomState(0)
    mvc=6400+rs.normal(0,0.7,5300)+1e-4*np.arange(5300)
    hos=(6400+rs.normal(0,8,(10,1500))).astype(np.float32)
    hos[:,:1320]+= (20*np.exp(-np.arange(1320)/30.)).astype(np.float32)
    args=(0,mvc,hos,5280,1320,3,'ML1',2000,'f32seq')
    t=time.time(); 
    for i in range(16): overscan.channel_solve(args)
    print('serial 16: %.1f ms'%((time.time()-t)*1e3))
    import cProfile,pstats
    pr=cProfile.Profile(); pr.enable()
    for i in range(16): overscan.channel_solve(args)
    pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(8)
    pool=HostPool(int(sys.argv[1]) if len(sys.argv)>1 else 6)
    for rep in range(3):
        t=time.time()
        rs_=[pool.submit(overscan.channel_solve,[args]*16) for k in range(4)]
        for r in rs_: r.get()
        print('pool 64 tasks: %.1f ms'%((time.time()-t)*1e3))
    pool.close()
