"""Print rocprofv3 kernel_stats.csv compactly: template arguments kept, namespaces dropped."""
import csv
import re
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    n = re.sub(r"\(anonymous namespace\)::|q3::|void ", "", n)
    n = re.sub(r"\(.*\)$", "", n)
    print(f"{n[:70]:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs']) / 1e3:9.2f} us total {float(r['TotalDurationNs']) / 1e6:9.2f} ms {float(r['Percentage']):6.2f}%")
