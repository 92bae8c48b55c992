# usage (GPU box): bash scripts/ab_schedule.sh > gpurun_out/ab_schedule.txt   - the expansion legs under schedule variants (same cut in all)
R=$GRAFT_REPO_ROOT
run() { echo "== $*"; env PGX_MF_MEMO=0 "$@" python $R/scripts/bench_labelling.py C4 C3 C5 --no-oracle 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'gpu_expansion_ms' in d: print('   %-40s %8.1f ms  relabels %4d levels %5d sweeps %5d waves %3d crc %d' % (d['config'][:40], d['gpu_expansion_ms'], d['global_relabels'], d['bfs_levels'], d['sweeps'], d['wave_passes'], d['labels_crc']))
"; }
run PGX_X=0
run PGX_MF_WAVE_MAX=0
run PGX_MF_WAVE_MAX=0 PGX_MF_SWEEPS_LIST=32
run PGX_MF_WAVE_MAX=0 PGX_MF_SWEEPS_LIST=16
run PGX_MF_WAVE_FROM=2
run PGX_MF_WAVE_FROM=2 PGX_MF_SWEEPS_LIST=32
run PGX_MF_SWEEPS_LIST=48
run PGX_MF_SWEEPS_LIST=192
run PGX_MF_STALL=4 PGX_MF_WAVE_MAX=0 PGX_MF_SWEEPS_LIST=32
