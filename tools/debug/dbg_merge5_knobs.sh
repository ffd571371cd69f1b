#!/bin/bash
# timing of the fused merge kernel under tuning knobs (rocprofv3 kernel stats)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "2 0" "1 0" "2 1" "1 1"; do
  set -- $cfg
  export GSE_MOMRES_SPLIT=$1
  if [ "$2" = "1" ]; then export GSE_MERGE_NOTAIL=1; else unset GSE_MERGE_NOTAIL; fi
  rm -rf $R/gpurun_out/prof_knob
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_knob -- python3 $R/tools/debug/dbg_merge5_prof.py 20 > /dev/null 2>&1
  echo "split=$1 notail=$2: $(grep -h resampled $R/gpurun_out/prof_knob/*/*kernel_stats.csv | sed 's/.*",\([0-9]*,[0-9]*,[0-9.]*,[0-9.]*,[0-9]*,[0-9]*\),.*/\1/')"
done
