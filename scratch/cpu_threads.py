import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for n in [8, 16, 32, 64, 128]:
    os.environ['HUGS_CPU_THREADS'] = str(n)
    t0 = time.time(); r = bench.cpu_baseline(1); print(n, r['value'], round(time.time() - t0, 1), flush=True)
