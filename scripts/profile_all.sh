# usage (on the GPU box, through gpurun): bash scripts/profile_all.sh <tag>
# kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) of the default bench; summaries -> gpurun_out/,
# and gpurun_out/pmc_<tag>.json = the constants bench.py quotes (copy to profiles/pmc_current.json)
set -x
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-legs"   # the default steps / warm-up: the kernel averages are those of the bench line
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_${TAG}_stats -o bench -- $B > $R/gpurun_out/p_${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/p_${TAG}_fetch -o bench -- $B > $R/gpurun_out/p_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/p_${TAG}_write -o bench -- $B > $R/gpurun_out/p_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/p_${TAG}_sq -o bench -- $B > $R/gpurun_out/p_${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $R/gpurun_out/p_${TAG}_lds -o bench -- $B > $R/gpurun_out/p_${TAG}_lds.log 2>&1
cd $R && python scripts/rocpd_summary.py gpurun_out/p_${TAG}_stats/bench_results.db gpurun_out/p_${TAG}_fetch/bench_results.db gpurun_out/p_${TAG}_write/bench_results.db gpurun_out/p_${TAG}_sq/bench_results.db gpurun_out/p_${TAG}_lds/bench_results.db > gpurun_out/profile_${TAG}.txt
python scripts/pmc_json.py gpurun_out/profile_${TAG}.txt profiles/round6_bench_${TAG}.txt > gpurun_out/pmc_${TAG}.json
rm -rf gpurun_out/p_${TAG}_stats gpurun_out/p_${TAG}_fetch gpurun_out/p_${TAG}_write gpurun_out/p_${TAG}_sq gpurun_out/p_${TAG}_lds
