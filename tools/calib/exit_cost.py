#!/usr/bin/env python3
"""Runs tools/calib/exit_cost.cpp: seconds between the program's last line and wait() returning, per mode."""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(HERE, "exit_cost")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(HERE, "exit_cost.cpp")])
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| cpus", os.cpu_count())
for gb, mode, threads in ((1, "plain", 16), (3, "plain", 16), (3, "plain", 16), (3, "thp", 16), (3, "free", 16), (3, "free", 4), (3, "dontneed", 16), (3, "free", 1)):
	p = subprocess.Popen([exe, str(gb), mode, str(threads)], stdout=subprocess.PIPE)
	line = p.stdout.readline().decode()
	p.wait()
	t = time.time()
	print("%d GB %-8s %2d threads: exit took %.3f s | %s" % (gb, mode, threads, t - float(line.split()[0]), " ".join(line.split()[1:])))
