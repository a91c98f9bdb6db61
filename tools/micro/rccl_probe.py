import os, sys, time
sys.path.insert(0, "deep-neuroevolution_amd")
from dne_hip import _lib
t=time.time()
e = _lib.Engine(_lib.KIND_ES, 18, max_members=16, ref_count=16)
uid = _lib.comm_unique_id(); t1=time.time()
e.comm_init(0, 1, uid); t2=time.time()
e.barrier(); t3=time.time()
print("unique_id %.2f s, comm_init %.2f s, barrier %.2f s" % (t1-t, t2-t1, t3-t2))
