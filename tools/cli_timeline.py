#!/usr/bin/env python3
"""Where the wall clock of panvc3_project_alignments goes, start to exit: its --debugging-output timeline, the time before its
clock starts and the time between its last line and wait() returning (a process that has used the GPU pays for the memory it
holds when it exits: tools/calib/exit_cost_hip.py).  usage: tools/cli_timeline.py [--repeat 4] [--runs 3] [extra options of the tool]"""
import argparse
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cli_e2e

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--repeat", type=int, default=4)
ap.add_argument("--runs", type=int, default=3)
ap.add_argument("--workdir", default="/tmp/pvc3_e2e")
args, extra = ap.parse_known_args()
P = cli_e2e.prepare(args.workdir, args.scale, args.repeat)
full = cli_e2e.project_command(P) + extra
print("records %d, in.bam %.1f MB" % (P["n"], P["in_size"] / 1e6))
for _ in range(args.runs):
	e0 = time.time()
	p = subprocess.run(full, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
	e1 = time.time()
	if p.returncode:
		raise SystemExit(p.stderr.decode()[-2000:])
	print("wall %.3f s = %.2f M records/s" % (e1 - e0, P["n"] / (e1 - e0) / 1e6))
	for line in p.stderr.decode().split("\n"):
		if "wall clock at the start" in line:
			a = float(line.split("wall clock at the start ")[1].split(",")[0])
			b = float(line.split(", now ")[1].split(";")[0])
			cpu = float(line.split("CPU time (s): ")[1]) if "CPU time (s): " in line else 0.0
			print("    before main's clock starts %.3f s, main %.3f s, after its last line %.3f s; CPU time %.2f s = %.1f cores busy" % (a - e0, b - a, e1 - b, cpu, cpu / (b - a)))
		if "memory given back" in line:
			import re
			m = re.search(r"last batch rewritten ([0-9.e-]+), writer thread joined ([0-9.e-]+), output closed ([0-9.e-]+), memory given back ([0-9.e-]+)", line)
			if m:
				t = [float(x) for x in m.groups()]
				print("    after the last batch: writer thread %.3f s, closing the output %.3f s, still giving memory back %.3f s" % (t[1] - t[0], t[2] - t[1], t[3] - t[2]))
		if "Stages" in line or "Timeline" in line or "BGZF on" in line or "BGZF input" in line or "teardown" in line:
			print("   ", line.strip())
