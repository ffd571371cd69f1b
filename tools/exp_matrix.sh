#!/bin/bash
# A/B matrix on the GPU box: gather mapping (compile time) x first-level search (run time)
for S in 0 1; do
  GSE_CXXFLAGS="-DGSE_GATHER_STRIDED=$S" python gpu_se_amd/build.py --force > /dev/null 2>&1 || exit 1
  for TOPS in 0 1; do
    echo "== strided=$S tops=$TOPS"
    GSE_TOPS=$TOPS tools/prof.sh exp_${S}_${TOPS} tools/kbench.py 20 200 > /dev/null
    python tools/kstats.py gpurun_out/prof_exp_${S}_${TOPS} 2>/dev/null | head -3
    grep cycle gpurun_out/prof_exp_${S}_${TOPS}.log
  done
done
python gpu_se_amd/build.py --force > /dev/null 2>&1
