# usage: bash scripts/profile_labelling.sh <tag> <config...>   (on the GPU box through gpurun)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pl_${TAG} -o lab -- python $R/scripts/bench_labelling.py "$@" > $R/gpurun_out/pl_${TAG}.log 2>&1
cd $R && python scripts/rocpd_summary.py gpurun_out/pl_${TAG}/lab_results.db > gpurun_out/profile_labelling_${TAG}.txt
