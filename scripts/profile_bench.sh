set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o bench -- $B > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o bench -- $B > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o bench -- $B > $R/gpurun_out/prof_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/prof_sq -o bench -- $B > $R/gpurun_out/prof_sq.log 2>&1
ls -R $R/gpurun_out | head -50
