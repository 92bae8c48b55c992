# usage (GPU box, through gpurun): bash scripts/profile_labelling_pmc.sh <tag>      (counters over the labelling)
# One expansion from zeros per config (scripts/ab_expansion.py --reps 1: memo off, every min-cut solved) under rocprofv3: a --stats pass
# and separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with another trace domain).  scripts/pmc_labelling_json.py sums the
# counters over every dispatch of the min-cut kernels (mf_k_*, t_move_kernel, r_*_kernel, energy) -> gpurun_out/pmc_labelling_<tag>.json
# (copy to profiles/pmc_labelling.json) and a text summary gpurun_out/profile_labelling_pmc_<tag>.txt (copy to profiles/).
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for CFG in C3 C5 C4; do
  B="python $R/scripts/ab_expansion.py $CFG --reps 1"
  rocprofv3 --kernel-trace --stats -d /tmp/pl_${CFG}_stats -o lab -- $B > $R/gpurun_out/pl_${TAG}_${CFG}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pl_${CFG}_fetch -o lab -- $B > $R/gpurun_out/pl_${TAG}_${CFG}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pl_${CFG}_write -o lab -- $B > $R/gpurun_out/pl_${TAG}_${CFG}_write.log 2>&1
done
cd $R && python scripts/pmc_labelling_json.py $TAG /tmp/pl_ > gpurun_out/profile_labelling_pmc_${TAG}.txt 2> gpurun_out/profile_labelling_pmc_${TAG}.err
