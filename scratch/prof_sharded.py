import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--force_sharded", "--no_cpu_baseline", "--steps", "256"]
sys.path.insert(0, ".")
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(28)
print(s.getvalue()[:6000])
