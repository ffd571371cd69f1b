"""Print a compact per-kernel table from a rocprofv3 *_kernel_stats.csv (or a directory containing one)."""
import csv, glob, os, signal, sys
signal.signal(signal.SIGPIPE, signal.SIG_DFL)
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
for row in csv.DictReader(open(path)):
    name = row["Name"].split("(")[0].replace("void ", "")
    if len(name) > 60: name = name[:57] + "..."
    print("%-60s calls %5s  avg %8.2f us  min %8.2f  max %8.2f" % (name, row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
