# usage: bash scripts/profile_api.sh <tag> <config...>   (on the GPU box through gpurun)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pa_${TAG} -o api -- python $R/scripts/bench_api.py "$@" > $R/gpurun_out/pa_${TAG}.log 2>&1
cd $R && python scripts/rocpd_summary.py gpurun_out/pa_${TAG}/api_results.db > gpurun_out/profile_api_${TAG}.txt
