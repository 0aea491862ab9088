import numpy as np, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from origin_amd import thresholds as T, pipeline
rng=np.random.default_rng(0)
o2=rng.chisquare(3681,size=(600,600))/3681
areamap=np.zeros((600,600),int); k=1
for i in range(6):
    for j in range(6):
        areamap[i*100:(i+1)*100,j*100:(j+1)*100]=k;k+=1
flat=areamap.reshape(-1)
spx=[np.nonzero(flat==i)[0].astype(np.int32) for i in range(1,37)]
for r in range(3):
    t=time.perf_counter(); thr=pipeline.pca_threshold(o2,areamap,36,0.01,spx=spx); print("total ms",(time.perf_counter()-t)*1e3)
t=time.perf_counter(); tests=[o2.reshape(-1)[s] for s in spx]; print("gather ms",(time.perf_counter()-t)*1e3)
for r in range(2):
    t=time.perf_counter(); H=T.clipped_histograms(tests); print("hist ms",(time.perf_counter()-t)*1e3)
t=time.perf_counter()
for a in range(36): T.compute_thresh_gaussfit(tests[a],0.01,_hist=H[a])
print("fits ms",(time.perf_counter()-t)*1e3)
import scipy; print(scipy.__version__, os.cpu_count())
