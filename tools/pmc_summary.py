"""Per-kernel average of rocprofv3 --pmc counters (one directory per counter pass)."""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(d, ": no counter_collection.csv; files:", glob.glob(os.path.join(d, "**", "*"), recursive=True)[:8])
        continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(files[0])):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (name, row["Counter_Name"])
        acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
    for (name, ctr), (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
        print("%-58s %-11s launches %5d  avg %14.1f" % (name[:58], ctr, n, tot / n))
