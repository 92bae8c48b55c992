# usage (on the GPU box): bash scripts/calibrate_fetch.sh     -> gpurun_out/fetch_calibration.txt
# FETCH_SIZE / WRITE_SIZE of known-bytes kernels in this repo's access patterns (scripts/micro/fetch_calib.hip), one --pmc pass each
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/scripts/micro/fetch_calib.hip || exit 1
/tmp/fetch_calib 1 > $R/gpurun_out/fetch_calibration.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$C
  rocprofv3 --pmc $C --kernel-trace -d /tmp/fc_$C -o calib -- /tmp/fetch_calib 3 > /tmp/fc_$C.log 2>&1
  echo "== --pmc $C" >> $R/gpurun_out/fetch_calibration.txt
  python $R/scripts/rocpd_summary.py /tmp/fc_$C/calib_results.db >> $R/gpurun_out/fetch_calibration.txt 2>&1 || (ls -R /tmp/fc_$C | head -20 >> $R/gpurun_out/fetch_calibration.txt)
done
