#!/bin/bash
# A/B timing on ONE box (box-to-box noise is +-2..4 %): alternates builds and / or tuning tables, REPS times each, on the fresh-solve
# loop of a bench workload. A variant is  name[:lib[:tuning]]  — lib = path of a libctgn build (default: the in-tree one),
# tuning = a CTGN_TUNING string. Examples:
#   scripts/ab.sh base:ct_icp_amd/libctgn_base.so cur                       two builds
#   scripts/ab.sh on off::stop_poll=0                                       one build, two tuning tables
#   MODE=iter WORKLOADS="B2 D" scripts/ab.sh cur nosums:.ab/libctgn_nosums.so
# MODE=step (default): bench.py's headline loop (ms per step, search-kernel ms first / later). MODE=iter: scripts/iter_times.py (search-kernel
# time of each iteration + un-profiled step time). MODE=ablate MASKS="0 1 2 ...": kernel ms under ablation masks (results invalid, timings only).
# Replaces ab.sh / ab3.sh / ab_bench.sh / ablate.sh / ablate2.sh of rounds 1-5.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
variants=("$@"); [ ${#variants[@]} -eq 0 ] && variants=(cur)
for w in ${WORKLOADS:-B2}; do
  for rep in $(seq 1 ${REPS:-2}); do
    for v in "${variants[@]}"; do
      IFS=: read -r name lib tun <<< "$v"
      [ -z "$lib" ] && lib="ct_icp_amd/libctgn.so"
      case "$lib" in /*) ;; *) lib="$PWD/$lib";; esac
      case "${MODE:-step}" in
        iter)
          echo "$w $name: $(CTGN_TUNING="$tun" CTGN_LIB_PATH=$lib timeout 600 python scripts/iter_times.py $w ${MASKS:-0} 2>&1 | grep '^{' | tr '\n' ' ')";;
        ablate)
          for m in ${MASKS:-0 1 2 4 16}; do
            CTGN_TUNING="$tun" CTGN_LIB_PATH=$lib python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --sub none --ablate $m ${BENCH_ARGS:-} 2>/dev/null | tail -1 | \
              sed -E "s/.*\"kernel_ms_avg\":([0-9.]+).*/$w $name ablate=$m kernel_ms=\1/"
          done;;
        *)
          CTGN_TUNING="$tun" CTGN_LIB_PATH=$lib python bench.py --workload $w --steps ${STEPS:-100} --warmup 10 --clock-warm ${WARM:-100} --no-cpu-baseline --no-pmc --no-extras --sub none ${BENCH_ARGS:-} 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $name step_ms=%.4f kernel_ms=%.4f first=%.4f later=%.4f parity=%s' % (d['ms_per_step'], r['kernel_ms_avg'], r['first_iteration']['kernel_ms'], r['later_iterations']['kernel_ms'], d.get('parity_m_rad')))";;
      esac
    done
  done
done
