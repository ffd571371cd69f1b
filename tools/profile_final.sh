#!/bin/bash
# usage: tools/profile_final.sh <tag>  (on the GPU box, from the repo root): what has to be re-taken when the cycle kernel's sources
# changed after tools/profile_round.sh ran -- the kernel statistics of bench.py, the four PMC passes (2^20, 2^21) and the traffic
# json, the default bench line, the two-rank rehearsal of the N > 1 line.
tag=$1
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_bench -- python3 $R/bench.py --steps 200 --warmup 20 > $O/${tag}_bench_N2p20.json.log 2> $O/${tag}_bench.err
cp $O/prof_${tag}_bench/*/*kernel_stats.csv $O/${tag}_bench_N2p20_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${tag}_$C -- python3 $R/bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $O/pmc_${tag}_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc21_${tag}_$C -- python3 $R/bench.py --steps 50 --warmup 10 --log2-particles 21 --no-extras --no-cpu-baseline > $O/pmc21_${tag}_$C.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE > $O/${tag}_pmc_fetch_write.txt
python tools/pmc_summary.py gpurun_out/pmc21_${tag}_FETCH_SIZE gpurun_out/pmc21_${tag}_WRITE_SIZE > $O/${tag}_pmc_fetch_write_N2p21.txt
rm -f gpurun_out/pmc_traffic.json
python tools/traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE "profiles/${tag}_pmc_fetch_write.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 50)" 20 > $O/${tag}_traffic.log 2>&1
python tools/traffic.py gpurun_out/pmc21_${tag}_FETCH_SIZE gpurun_out/pmc21_${tag}_WRITE_SIZE "profiles/${tag}_pmc_fetch_write_N2p21.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 50 --log2-particles 21)" 21 > $O/${tag}_traffic_N2p21.log 2>&1
cp gpurun_out/pmc_traffic.json $O/${tag}_pmc_traffic.json
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json       # (so that the lines below carry the traffic of THESE sources)
python bench.py > $O/${tag}_bench_default_line.json.log 2> $O/${tag}_bench_default.err
GSE_BENCH_REHEARSAL=1 timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/${tag}_bench_rehearsal_2ranks_gloo.json.log 2> $O/${tag}_bench_rehearsal.err
echo "final profiles done"
