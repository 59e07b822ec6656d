"""Print the kernel stats CSV(s) rocprofv3 --kernel-trace --stats --output-format csv left under a directory."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f"# {f}")
    print(f"{'kernel':58s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
        print(f"{r['Name'][:58]:58s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e3:12.1f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
