#!/usr/bin/env python3
"""Runs tools/calib/exit_cost_hip.hip: seconds between the program's last line and wait() returning."""
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(HERE, "exit_cost_hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-o", exe, os.path.join(HERE, "exit_cost_hip.hip")])
for args in (("0", "0"), ("0", "0"), ("1", "0"), ("2", "0"), ("0", "2"), ("0", "8"), ("2", "8"), ("2", "8", "free"), ("1", "0", "free"),
			 ("0", "0", "plain", "1"), ("0", "0", "plain", "3"), ("0", "0", "thp", "3"), ("0", "0", "dontneed", "3"), ("0", "0", "unmap", "3")):
	p = subprocess.Popen([exe] + list(args), stdout=subprocess.PIPE)
	line = p.stdout.readline().decode()
	p.wait()
	t = time.time()
	print("pinned %s GB, device %s GB %s: exit took %.3f s | %s" % (args[0], args[1], " ".join(args[2:]), t - float(line.split()[0]), " ".join(line.split()[1:])))
