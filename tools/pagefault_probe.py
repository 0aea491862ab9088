import numpy as np, time, os, ctypes as C, sys
sys.path.insert(0, os.getcwd())
print("THP enabled:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
print("THP defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1325_000_000 // 4
for rep in range(2):
    a = np.empty(n, np.float64)
    t = time.perf_counter(); a[:] = 1.0; t1 = time.perf_counter() - t
    t = time.perf_counter(); a[:] = 2.0; t2 = time.perf_counter() - t
    print(f"{a.nbytes/1e9:.2f} GB: first touch {t1:.3f} s ({a.nbytes/t1/1e9:.1f} GB/s), second write {t2:.3f} s ({a.nbytes/t2/1e9:.1f} GB/s)")
    del a
from origin_amd.device import Context
ctx = Context(0)
d = ctx.empty((n,), np.float32)
d.fill_bytes(0)
for rep in range(3):
    out = np.empty(n, np.float64)
    t = time.perf_counter(); d.to_host_f64(out); t1 = time.perf_counter() - t
    t = time.perf_counter(); d.to_host_f64(out); t2 = time.perf_counter() - t
    print(f"to_host_f64 of {d.nbytes/1e9:.2f} GB: fresh destination {t1:.3f} s ({d.nbytes/t1/1e9:.1f} GB/s of device bytes), mapped destination {t2:.3f} s ({d.nbytes/t2/1e9:.1f} GB/s)")
    h = np.empty(n, np.float32)
    t = time.perf_counter(); d.to_host(h); t3 = time.perf_counter() - t
    t = time.perf_counter(); d.to_host(h); t4 = time.perf_counter() - t
    print(f"to_host (float32, pageable): fresh {t3:.3f} s ({d.nbytes/t3/1e9:.1f} GB/s), mapped {t4:.3f} s ({d.nbytes/t4/1e9:.1f} GB/s)")
