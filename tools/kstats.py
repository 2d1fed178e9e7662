"""print a rocprofv3 kernel_stats.csv:  python3 tools/kstats.py <dir> [rows]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{r['Name'][:64]:64s} n={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:8.1f} pct={r['Percentage']}")
