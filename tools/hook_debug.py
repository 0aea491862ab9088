import sys, numpy as np
sys.path.insert(0, '.')
from origin_amd import pipeline, synth
from origin_amd.device import Context
ctx = Context(0)
N = 600
f = synth.SyntheticField(3681, N, N)
raw, var, mask = f.arrays()
d_raw, d_var, d_mask = ctx.to_device(raw), ctx.to_device(var), ctx.to_device(mask.astype(np.uint8))
del raw, var
pre = pipeline.preprocess(ctx, d_raw, d_var, d_mask, want_cont=False)
thr = pipeline.pca_threshold(pre["o2_host"], f.areamap, f.nbAreas, 0.01)
spx = pipeline.area_lists(f.areamap, f.nbAreas)
rows = [(int(s.min()) // N, int(s.max()) // N) for s in spx]
def hook(areas):
    print("hook: active areas", areas, "rows", [rows[a] for a in areas])
ctx.set_pca_tail_hook(hook, 2)
F, mapO2, nstop, drv = pipeline.greedy_pca(ctx, pre["cube_std"], f.areamap, f.nbAreas, thr["thresO2"], thr["testO2"], 50, 100, spx=spx, o2_dev=pre["o2"])
ctx.set_pca_tail_hook(None)
it = [int(mapO2.reshape(-1)[s].max()) for s in spx]
print("iterations per area:", it)
print("trace:", drv.trace[:14])
ctx.close()
