# usage: bash scripts/profile_graph.sh <tag>   (on the GPU box through gpurun)
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pg_${TAG} -o graph -- python $R/scripts/bench_graph.py > $R/gpurun_out/pg_${TAG}.log 2>&1
cd $R && python scripts/rocpd_summary.py gpurun_out/pg_${TAG}/graph_results.db > gpurun_out/profile_graph_${TAG}.txt
