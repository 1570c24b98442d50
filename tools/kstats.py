"""Print rocprofv3 *_kernel_stats.csv files side by side: name (short), calls, avg us, total ms, %."""
import csv
import sys

for path in sys.argv[1:]:
    print("==", path)
    rows = list(csv.DictReader(open(path)))
    for r in rows[:16]:
        name = r["Name"].split("(")[0].replace("void ", "")
        print(f"  {name[:52]:52s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:9.2f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms  {float(r['Percentage']):5.1f} %")
