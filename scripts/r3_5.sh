#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 60 gpurun_out/pytest_gpu.log | cut -c1-220
for wl in B1 C; do for p in 1 0; do
  CTGN_PERSISTENT=$p timeout 300 python bench.py --workload $wl --sub none --no-pmc --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl persistent=$p step_ms=%.4f fps_reg=%s robust=%s parity=%s' % (d['ms_per_step'], d.get('frames_per_sec',{}).get('ms_per_frame'), d.get('robust_route',{}).get('ms_per_frame'), d.get('parity_m_rad')))"
done; done
