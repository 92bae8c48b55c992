# usage: bash scripts/profile_sq.sh <tag> [env assignments...]   SQ counter pass of the default bench (on the GPU box)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs"
env "$@" rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/psq_${TAG} -o bench -- $B > $R/gpurun_out/psq_${TAG}.log 2>&1
cd $R && python scripts/rocpd_summary.py gpurun_out/psq_${TAG}/bench_results.db | grep "score_" > gpurun_out/profile_sq_${TAG}.txt
rm -rf gpurun_out/psq_${TAG}
