import sys, numpy as np
sys.path.insert(0, '.')
from origin_amd import pipeline, synth
from origin_amd.device import Context
ctx = Context(0)
for N in (200, 300, 600):
    f = synth.SyntheticField(3681, N, N)
    raw, var, mask = f.arrays()
    d_raw, d_var, d_mask = ctx.to_device(raw), ctx.to_device(var), ctx.to_device(mask.astype(np.uint8))
    del raw, var
    pre = pipeline.preprocess(ctx, d_raw, d_var, d_mask, want_cont=False)
    thr = pipeline.pca_threshold(pre["o2_host"], f.areamap, f.nbAreas, 0.01)
    spx = pipeline.area_lists(f.areamap, f.nbAreas)
    F, mapO2, nstop, drv = pipeline.greedy_pca(ctx, pre["cube_std"], f.areamap, f.nbAreas, thr["thresO2"], thr["testO2"], 50, 100, spx=spx, o2_dev=pre["o2"])
    print(N, "areas", f.nbAreas, "trace (areas, nuisance):", drv.trace)
    del d_raw, d_var, d_mask, pre, F
ctx.close()
