#!/bin/bash
# A/B of compile-time knobs on the per-cycle trace: tools/ab_trace.sh "<flags A>" "<flags B>" ...   (run on the GPU box)
set -e
i=0
for flags in "$@"; do
  GSE_CXXFLAGS="$flags" python gpu_se_amd/build.py --force > gpurun_out/ab_build_$i.log 2>&1
  echo "== [$flags]"
  python tools/kernel_meta.py "pf_step_kernel<true, true, false, true, 2, 3" || true
  timeout -k 10 200 python tools/cycle_trace.py 20 ${AB_CYCLES:-600} gpurun_out/ab_trace_$i.csv > gpurun_out/ab_trace_$i.log 2>&1; grep -A1 "cycles from" gpurun_out/ab_trace_$i.log
  i=$((i+1))
done
