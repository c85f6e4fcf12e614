"""Average duration per (kernel, grid size) of a rocprofv3 kernel trace (*_kernel_trace.csv): launches of different sizes apart.
usage: python tools/summarize_trace_by_grid.py <kernel_trace.csv>"""
import csv, re, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
	m = re.search(r"(k_\w+(?:<\w+>)?)", row["Kernel_Name"])
	name = m.group(1) if m else row["Kernel_Name"][:48]
	grid = int(row.get("Grid_Size_X", row.get("Grid_Size", 0)))
	acc[(name, grid)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
for (name, grid), v in sorted(acc.items()):
	print("%-50s grid %8d  n %4d  avg %8.3f ms  min %8.3f  max %8.3f" % (name, grid, len(v), sum(v) / len(v), min(v), max(v)))
