"""HBM traffic of the cycle kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc.sh) -> profiles/pmc_traffic.json.
usage: traffic.py <fetch dir> <write dir> <source note> [log2 N = 20]      (one entry per size: {"by_n": {"1048576": {...}, ...}})

Corrections as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: both counters are in KB; WRITE_SIZE is exact for the streaming
stores of this kernel; FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950 -- x2 for wide coalesced reads, uncalibrated for
narrower ones (the gathered state rows are dword reads), so fetch x 2 is an upper bound; the moment kernel of the same run (24 B per
particle of 16-byte-per-lane reads, a known byte count) is listed as the check of the factor."""
import csv, glob, hashlib, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def averages(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][0] += float(row["Counter_Value"]); acc[name][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


fetch, write = averages(sys.argv[1]), averages(sys.argv[2])
log2n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
cycle = next(k for k in fetch if "pf_step_kernel<true, true, false, true, 2, 3>" in k)
moments = next((k for k in fetch if k.endswith("pf_moments_kernel")), None)
n = 1 << log2n
out = {
    "source": sys.argv[3] if len(sys.argv) > 3 else "",
    "csrc_sha256_16": bench.csrc_hash(),
    "kernel": "pf_step_kernel<gather,predict,update,scan in+out>, N = 2^%d, one launch = one cycle" % log2n,
    "launches": fetch[cycle][1],
    "WRITE_SIZE_KB": round(write[cycle][0], 1),
    "FETCH_SIZE_KB_raw": round(fetch[cycle][0], 1),
    "fetch_correction": "x2 (gfx950 counts 128-byte requests at 64 bytes; exact for 16-byte-per-lane reads, an upper bound for the "
                        "dword reads of the gathered rows)",
    "fetch_bytes_corrected": int(fetch[cycle][0] * 1024 * 2),
    "write_bytes": int(write[cycle][0] * 1024),
    "traffic_bytes_per_launch": int(fetch[cycle][0] * 1024 * 2 + write[cycle][0] * 1024),
    "algorithmic_bytes_per_launch": 140 * n,
}
if moments:
    out["check_pf_moments_kernel"] = {"FETCH_SIZE_KB_raw": round(fetch[moments][0], 1), "x2_bytes": int(fetch[moments][0] * 2048),
                                       "known_bytes": 24 * n}
# merge into the per-size file (the scratch copy if an earlier pass of this run wrote one, else the committed one)
path = os.path.join(ROOT, "gpurun_out", "pmc_traffic.json")
doc = {"by_n": {}}
for cand in (path, os.path.join(ROOT, "profiles", "pmc_traffic.json")):
    if os.path.exists(cand):
        old = json.load(open(cand))
        doc = old if "by_n" in old else {"by_n": {str(1 << 20): old}}
        break
doc["by_n"][str(n)] = out
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
