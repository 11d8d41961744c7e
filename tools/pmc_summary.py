"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter: sum and per-dispatch mean."""
import csv, sys, glob, collections
for path in sys.argv[1:]:
    for f in sorted(glob.glob(path + "/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: [0.0, set()])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")
            if "anonymous" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("::")[1].split("(")[0]
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
        print("#", f)
        for (k, c), (v, d) in sorted(acc.items()):
            print("%-16s %-22s total %16.0f  dispatches %5d  per_dispatch %14.1f" % (k, c, v, len(d), v / len(d)))
