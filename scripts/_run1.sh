cd $GRAFT_REPO_ROOT
for s in 31 32 33 34 35 36; do timeout 900 python tests/soak_pointwise.py $s 600 < /dev/null 2>&1 | tail -8 | cut -c1-400; done > gpurun_out/soak2.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -k "graph" -q -x < /dev/null 2>&1 | tail -3 >> gpurun_out/soak2.log
