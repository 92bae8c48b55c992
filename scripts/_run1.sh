cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short < /dev/null 2>&1 | tail -8 > gpurun_out/gpu_suite.log
timeout 300 python bench.py < /dev/null > gpurun_out/bench.log 2>&1
for s in 41 42 43 44 45 46 47 48 49 50 51 52; do SOAK_DENSE=1 timeout 400 python tests/soak_scoring.py $s 200 < /dev/null 2>&1 | tail -8; done > gpurun_out/soak.log 2>&1
