#!/bin/bash
# A/B matrix on the GPU box (several flags of one configuration are joined with commas): compile-time knobs of the fused step kernel, timed with the realistic cycle of kbench
for FLAGS in ${GSE_EXP_LIST:-"-DGSE_STAGGER=1"}; do
  FLAGS=${FLAGS//,/ }
  GSE_CXXFLAGS="$FLAGS" python gpu_se_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $FLAGS"; continue; }
  tag=$(echo "$FLAGS" | tr -cd 'A-Za-z0-9_=' )
  tools/prof.sh exp_$tag tools/kbench.py 20 200 > /dev/null
  echo "== $FLAGS: $(python tools/kstats.py gpurun_out/prof_exp_$tag 2>/dev/null | grep 'true, true, false, true' | head -1 | sed 's/gse::pf_step_kernel//')"
done
python gpu_se_amd/build.py --force > /dev/null 2>&1
