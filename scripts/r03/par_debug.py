import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
if __name__ == "__main__":
    t = time.time()
    print(bench.cpu_baseline_parallel(int(sys.argv[1]) if len(sys.argv) > 1 else 50, 16, budget_s=400), "in %.1f s" % (time.time() - t), flush=True)
