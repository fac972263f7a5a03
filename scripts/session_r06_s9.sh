cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R="$PWD"; O=gpurun_out/s9; mkdir -p $O
python scripts/register_time.py B1 host_threads=3 2>&1 | tail -2 > $O/register_untraced.txt
python scripts/register_time.py --robust B1 host_threads=3 2>&1 | tail -2 >> $O/register_untraced.txt
cat $O/register_untraced.txt
for mode in gn robust; do
  rm -rf gpurun_out/api_tmp
  flag=""; [ $mode = robust ] && flag="--robust"
  (cd /tmp && timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d "$R/gpurun_out/api_tmp" -o t -- python "$R/scripts/register_api_trace.py" run $flag) > $O/run_$mode.log 2>&1
  a=$(find gpurun_out/api_tmp -name "*hip_api_trace.csv" | head -1); k=$(find gpurun_out/api_tmp -name "*kernel_trace.csv" | head -1)
  grep "Register x" $O/run_$mode.log
  python scripts/register_api_trace.py report $a $k > $O/register_hip_api_trace_$mode.txt 2>&1
  head -60 $O/register_hip_api_trace_$mode.txt
  rm -rf gpurun_out/api_tmp
done
