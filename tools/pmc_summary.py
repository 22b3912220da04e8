"""Developer tool: print per-launch counter means of render_kernel from rocprofv3 --pmc CSV dirs."""
import collections, csv, glob, os, sys
KERNEL = os.environ.get("PMC_KERNEL", "render_kernel")
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in sorted(acc):
            print("%-28s %-32s %.6g (x%d)" % (d.rstrip("/").split("/")[-1][:28], k, acc[k] / n[k], n[k]))
