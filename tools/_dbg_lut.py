import sys, numpy as np
sys.path.insert(0,'/root/repo')
import libgwaspp_amd as gw
from libgwaspp_amd import capi
for n,m in ((600,48),(600,200),(3000,500)):
    t=gw.GwTable(n,m); t.simulate(5,0,0); t.simulate_phenotype(5)
    for lut in (1,0):
        t.set_option(capi.OPT_BOOST_LUT, lut)
        for thr in (30.0, 5.0, 1e9, -1e9):
            h,nh,npairs,_,_=t.boost_screen(0,m,0,m,threshold=thr,extremes=False)
            print(n,m,'lut',lut,'thr',thr,'hits',nh,'pairs',npairs, m*(m-1)//2, t.last_boost_kernel(), t.get_option(capi.OPT_BOOST_PREFILTER_TRIPS))
    t.close()
