#!/bin/bash
# A/B of compile-time knobs of the GS-UKF kernels at 2^18 Gaussians (see exp_matrix.sh)
for FLAGS in ${GSE_EXP_LIST:-"-DGSE_GSUKF_PRED_WAVES=1"}; do
  FLAGS=${FLAGS//,/ }
  GSE_CXXFLAGS="$FLAGS" python gpu_se_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $FLAGS"; continue; }
  tag=$(echo "$FLAGS" | tr -cd 'A-Za-z0-9_=' )
  tools/prof.sh gsf_$tag tools/gsf_bench.py 262144 > /dev/null
  echo "== $FLAGS"; python tools/kstats.py gpurun_out/prof_gsf_$tag 2>/dev/null | sed -n 1,4p
done
python gpu_se_amd/build.py --force > /dev/null 2>&1
