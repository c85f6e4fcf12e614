#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel trace stats + separate FETCH_SIZE / WRITE_SIZE PMC passes) into profiles/.

usage: summarize_rocprof.py OUT_DIR TRACE_DIR [FETCH_DIR WRITE_DIR] [--traffic-key=C3@1] [--fetch-factor=F]
The traffic entry records the content key of the kernels it was measured on (panvc3_amd.build.kernel_key() of the tree the
script runs in, which must be the tree that ran): bench.py quotes the number only for that same library.
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 bytes, so the read side is doubled ("corrected"); both raw and corrected are stored.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def short(name):
	m = re.search(r"(k_\w+)(?:<[^>]*>)?\(", name)		# (k_project_fast is a template: "k_project_fast<false>(...)")
	return m.group(1) if m else name.split("(")[0]


def main():
	args = [a for a in sys.argv[1:] if not a.startswith("--")]
	key = None
	fetch_factor = None
	for a in sys.argv[1:]:
		if a.startswith("--traffic-key="):
			key = a.split("=", 1)[1]
		if a.startswith("--fetch-factor="):
			fetch_factor = float(a.split("=", 1)[1])
	out_dir, trace_dir = args[0], args[1]
	os.makedirs(out_dir, exist_ok=True)
	stats = sorted(glob.glob(os.path.join(trace_dir, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)		# the newest run of a directory that has seen several
	if stats:
		shutil.copy(stats[0], os.path.join(out_dir, "kernel_stats.csv"))
	summary = {}
	if stats:
		for r in csv.DictReader(open(stats[0])):
			summary.setdefault(short(r["Name"]), {}).update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), pct=float(r["Percentage"]))
	if len(args) >= 4:
		for d, ctr in ((args[2], "FETCH_SIZE"), (args[3], "WRITE_SIZE")):
			f = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
			acc = collections.defaultdict(list)
			meta = {}
			for r in csv.DictReader(open(f)):
				if r["Counter_Name"] == ctr and "pvc3::" in r["Kernel_Name"]:
					k = short(r["Kernel_Name"])
					acc[k].append(float(r["Counter_Value"]))
					meta[k] = dict(vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]), grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]))
			for k, v in acc.items():
				s = summary.setdefault(k, {})
				s[ctr + "_KiB_avg"] = sum(v) / len(v)
				s.update(meta[k])
		for k, s in summary.items():
			if "FETCH_SIZE_KiB_avg" in s or "WRITE_SIZE_KiB_avg" in s:
				f = s.get("FETCH_SIZE_KiB_avg", 0.0) * 1024
				w = s.get("WRITE_SIZE_KiB_avg", 0.0) * 1024
				s["hbm_bytes_raw"] = f + w
				s["hbm_bytes_corrected"] = 2 * f + w
	json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1, sort_keys=True)
	if key and "k_project_fast" in summary and "hbm_bytes_corrected" in summary["k_project_fast"]:
		p = os.path.join(os.path.dirname(os.path.abspath(out_dir)), "pmc_traffic.json")
		cur = json.load(open(p)) if os.path.exists(p) else {}
		sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
		from panvc3_amd import build as pvc3_build
		kf = summary["k_project_fast"]
		fetch = kf.get("FETCH_SIZE_KiB_avg", 0.0) * 1024
		write = kf.get("WRITE_SIZE_KiB_avg", 0.0) * 1024
		# The guide's correction: FETCH_SIZE tallies 128-byte requests at 64 bytes.  Round 1 used 1.30 for this kernel, from a
		# calibration against the USEFUL bytes of its access patterns; the lines that move are what the counter doubled says
		# (tools/calib/random_access saturates at the copy rate only when its bytes are counted in 128-byte lines).
		factor = fetch_factor if fetch_factor is not None else 2.0
		cur[key] = {
			"k_project_fast_hbm_bytes_per_launch": factor * fetch + write,
			"k_project_fast_hbm_bytes_raw": fetch + write,
			"k_project_fast_hbm_bytes_fetch_doubled": 2 * fetch + write,
			"fetch_factor": factor,
			"source": os.path.basename(os.path.abspath(out_dir)),
			"kernel_key": pvc3_build.kernel_key(),
		}
		json.dump(cur, open(p, "w"), indent=1, sort_keys=True)
	for k, s in sorted(summary.items(), key=lambda kv: -kv[1].get("avg_ns", 0)):
		print("%-22s avg %10.3f ms  %5.1f%%  hbm(corrected) %s" % (k, s.get("avg_ns", 0) / 1e6, s.get("pct", 0), ("%.2f GB" % (s["hbm_bytes_corrected"] / 1e9)) if "hbm_bytes_corrected" in s else "-"))


if __name__ == "__main__":
	main()
