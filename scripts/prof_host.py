"""cProfile of drop-in calls (host-side cost split).  usage: python scripts/prof_host.py C3 C4 ..."""
import cProfile, os, pstats, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, "scripts", "bench_api.py"), run_name="__main__")
finally:
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
