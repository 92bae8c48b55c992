# usage: bash scripts/profile_sq2.sh <tag>   second SQ counter pass of the default bench: LDS / wait breakdown (GPU box)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs"
env "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/psq2_${TAG} -o bench -- $B > $R/gpurun_out/psq2_${TAG}.log 2>&1
env "$@" rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/psq3_${TAG} -o bench -- $B > $R/gpurun_out/psq3_${TAG}.log 2>&1
cd $R && python scripts/rocpd_summary.py gpurun_out/psq2_${TAG}/bench_results.db gpurun_out/psq3_${TAG}/bench_results.db | grep "score_" > gpurun_out/profile_sq2_${TAG}.txt
tail -3 gpurun_out/psq2_${TAG}.log gpurun_out/psq3_${TAG}.log | grep -i "error\|invalid\|not" | head
rm -rf gpurun_out/psq2_${TAG} gpurun_out/psq3_${TAG}
