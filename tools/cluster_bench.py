"""Cycle time of mid-size particle filters with four workgroups per tile (GSE_OP_CLUSTER) against one workgroup per tile:
device microseconds per cycle (HIP events), a filter drawn from its prior and after 500 cycles of tracking.
usage: cluster_bench.py [log2N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpu_se_amd
from gpu_se_amd import sim_base
import bench

sizes = [float(v) for v in sys.argv[1:]] or [12, 14, 15, 16, 17, 18]
io = bench.make_inputs(1200, 1.0)
bench.device_warm_up(torch, 1 << 20, 1.0, cycles=300)
for log2n in sizes:
    n = int(round(2 ** log2n))
    row = []
    for cluster in (False, True):
        gpu_se_amd.seed(2)
        _, _, _, pf = sim_base.get_parts(N_particles=n)
        pf.cluster = cluster and pf.cluster
        pf.weight_scale = os.environ.get('GSE_BENCH_NO_SCALE') != '1'
        at = [0]

        def cycle(_i):
            u, z = io[at[0] % len(io)]
            at[0] += 1
            pf.predict(u, 1.0); pf.update(u, z); pf.resample()
        for i in range(10):
            cycle(i)
        fresh = bench.time_method_events(torch, cycle, 200)
        for i in range(300):
            cycle(i)
        steady = bench.time_method_events(torch, cycle, 300)
        row.append((fresh * 1e6, steady * 1e6))
    print("N=2^%-5g  tile: from the prior %6.2f us, tracking %6.2f us   cluster: %6.2f us, %6.2f us" % (
        log2n, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
