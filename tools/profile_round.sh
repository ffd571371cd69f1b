#!/bin/bash
# usage: tools/profile_round.sh <tag>   (on the GPU box, from the repo root): everything profiles/ keeps for a round, under
# gpurun_out/<tag>_*: kernel-trace statistics of bench.py, of the GS-UKF cycle and of the resident closed loop; the two PMC
# passes (FETCH_SIZE, WRITE_SIZE) of bench.py and the traffic json; the shader counters; the open-loop run sequences.
tag=$1
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_bench -- python3 $R/bench.py --steps 200 --warmup 20 > $O/${tag}_bench_N2p20.json.log 2> $O/${tag}_bench.err
cp $O/prof_${tag}_bench/*/*kernel_stats.csv $O/${tag}_bench_N2p20_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_gsf -- python3 $R/tools/gsf_bench.py 64 262144 > $O/${tag}_gsukf.log 2>&1
cp $O/prof_${tag}_gsf/*/*kernel_stats.csv $O/${tag}_gsukf_kernel_stats.csv
GSE_RESIDENT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_cl -- python3 $R/tools/closedloop_bench.py 32 18 50 0.1 > $O/${tag}_closedloop.log 2>&1
cp $O/prof_${tag}_cl/*/*kernel_stats.csv $O/${tag}_closedloop_resident_32x2p18_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${tag}_$C -- python3 $R/bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $O/pmc_${tag}_$C.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE > $O/${tag}_pmc_fetch_write.txt
python tools/traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE "profiles/${tag}_pmc_fetch_write.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 50)" > /dev/null
tools/pmc_sq.sh $tag bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $O/${tag}_pmc_sq.txt 2>&1
python tools/run_seq.py pf 24 50 > $O/${tag}_run_seq_pf.txt 2>&1
python tools/run_seq.py gsf 18 50 > $O/${tag}_run_seq_gsf.txt 2>&1
tools/sweep.sh > $O/${tag}_sweep.txt 2>&1
echo done
