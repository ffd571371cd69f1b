#!/bin/bash
# usage: tools/profile_round.sh <tag> [part]  (on the GPU box, from the repo root): everything profiles/ keeps for a round,
# under gpurun_out/<tag>_*.  part 1: kernel-trace statistics of bench.py, of the GS-UKF cycle and of the resident closed loops;
# the two PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py and the traffic json; the tail profile (first 300 cycles from the
# prior + the adversarial patterns) and the per-cycle trace.  part 2: the shader counters (PF cycle and GS-UKF lanes), the
# open-loop run sequences, the size sweep, the large-bank ancestors kernel on the adversarial weight patterns.
tag=$1
part=${2:-all}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
if [ "$part" = "1" ] || [ "$part" = "all" ]; then
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_bench -- python3 $R/bench.py --steps 200 --warmup 20 > $O/${tag}_bench_N2p20.json.log 2> $O/${tag}_bench.err
cp $O/prof_${tag}_bench/*/*kernel_stats.csv $O/${tag}_bench_N2p20_kernel_stats.csv
echo "bench profiled"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_gsf -- python3 $R/tools/gsf_bench.py 64 262144 > $O/${tag}_gsukf.log 2>&1
cp $O/prof_${tag}_gsf/*/*kernel_stats.csv $O/${tag}_gsukf_kernel_stats.csv
GSE_RESIDENT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_cl -- python3 $R/tools/closedloop_bench.py 32 18 50 0.1 > $O/${tag}_closedloop.log 2>&1
cp $O/prof_${tag}_cl/*/*kernel_stats.csv $O/${tag}_closedloop_resident_32x2p18_kernel_stats.csv
GSE_RESIDENT=1 GSE_GSF=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_clg -- python3 $R/tools/closedloop_bench.py 32 6 50 0.1 > $O/${tag}_closedloop_gsukf.log 2>&1
cp $O/prof_${tag}_clg/*/*kernel_stats.csv $O/${tag}_closedloop_gsukf_persistent_32x64_kernel_stats.csv
echo "loops profiled"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_tail -- python3 $R/tools/tail_profile.py 300 > $O/${tag}_tail.log 2>&1
cd $R
python tools/tail_summary.py gpurun_out/prof_${tag}_tail > $O/${tag}_tail_summary.txt
cp $O/prof_${tag}_tail/*/*kernel_stats.csv $O/${tag}_tail_kernel_stats.csv
python tools/cycle_trace.py 20 600 $O/${tag}_cycle_trace_N2p20.csv > $O/${tag}_cycle_trace_N2p20.txt 2>&1
echo "tail profiled"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${tag}_$C -- python3 $R/bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $O/pmc_${tag}_$C.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE > $O/${tag}_pmc_fetch_write.txt
python tools/traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE "profiles/${tag}_pmc_fetch_write.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 50)" > $O/${tag}_traffic.log 2>&1
# ... and the same two passes at 2^21 particles (the per-rank size of BASELINE config 4), so that a line of that size has its traffic
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc21_${tag}_$C -- python3 $R/bench.py --steps 50 --warmup 10 --log2-particles 21 --no-extras --no-cpu-baseline > $O/pmc21_${tag}_$C.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc21_${tag}_FETCH_SIZE gpurun_out/pmc21_${tag}_WRITE_SIZE > $O/${tag}_pmc_fetch_write_N2p21.txt
python tools/traffic.py gpurun_out/pmc21_${tag}_FETCH_SIZE gpurun_out/pmc21_${tag}_WRITE_SIZE "profiles/${tag}_pmc_fetch_write_N2p21.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 50 --log2-particles 21)" 21 > $O/${tag}_traffic_N2p21.log 2>&1
cp gpurun_out/pmc_traffic.json $O/${tag}_pmc_traffic.json
echo "part 1 done"
fi
if [ "$part" = "2" ] || [ "$part" = "all" ]; then
cd $R
tools/pmc_sq.sh $tag bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $O/${tag}_pmc_sq.txt 2>&1
tools/pmc_sq.sh ${tag}gsf tools/gsf_bench.py 262144 > $O/${tag}_gsukf_pmc_sq.txt 2>&1
python tools/run_seq.py pf 24 50 > $O/${tag}_run_seq_pf.txt 2>&1
python tools/run_seq.py gsf 18 50 > $O/${tag}_run_seq_gsf.txt 2>&1
tools/sweep.sh > $O/${tag}_sweep.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof_${tag}_anc -- python3 $R/tools/gsf_ancestors_tail.py > $O/${tag}_anc.log 2>&1
GSE_ANCESTORS_BY_SEARCH=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof_${tag}_anc_search -- python3 $R/tools/gsf_ancestors_tail.py > $O/${tag}_anc_search.log 2>&1
cd $R
python tools/gsf_ancestors_tail.py gpurun_out/prof_${tag}_anc > $O/${tag}_gsukf_ancestors_tail.txt
python tools/gsf_ancestors_tail.py gpurun_out/prof_${tag}_anc_search > $O/${tag}_gsukf_ancestors_tail_by_search.txt
echo "part 2 done"
fi
