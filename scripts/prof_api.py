import cProfile, pstats, sys, os, io
sys.argv=["bench_api.py","C4"]
sys.path.insert(0, os.path.join(os.getcwd(),"scripts"))
pr=cProfile.Profile()
pr.enable()
exec(open("scripts/bench_api.py").read())
pr.disable()
s=io.StringIO(); ps=pstats.Stats(pr,stream=s).sort_stats("cumulative"); ps.print_stats(45); print(s.getvalue()[:9000])
