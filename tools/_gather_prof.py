import time, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import synth, api
P = synth.config("C2"); o,i = P.as_lists()
def t(fn,n=500):
    fn(); t0=time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter()-t0)/n*1e6
print("gather %.1f (no planar check %.1f)"%(t(lambda: api._gather_views(o,i)), t(lambda: api._gather_views(o,i,planar_check=False))))
print("types both %.1f"%t(lambda: (set(map(type,o))=={np.ndarray}, set(map(type,i))=={np.ndarray})))
print("concat o %.1f  i %.1f"%(t(lambda: np.concatenate(o,axis=0)), t(lambda: np.concatenate(i,axis=0))))
print("lens both %.1f"%t(lambda: (np.fromiter(map(len,o),dtype=np.int64,count=len(o)), np.fromiter(map(len,i),dtype=np.int64,count=len(i)))))
rt=np.random.rand(500,6)
print("tuples %.1f"%t(lambda: api._poses_to_tuples(rt)))
co=np.concatenate(o,axis=0)
print("astype/reshape %.1f"%t(lambda: co.astype(np.float32,copy=False).reshape(-1,3)))
cnt=np.fromiter(map(len,o),dtype=np.int64,count=len(o))
print("checks %.1f"%t(lambda: (np.any((cnt*3)%3), np.any((cnt*2)%2), np.any(cnt!=cnt), cnt.min(), np.cumsum(cnt))))
