#!/usr/bin/env python3
"""Builds a second copy of libpvc3_hip.so with extra -D flags (ablation / instrumentation experiments) into
panvc3_amd/_build/diag_NAME/.  usage: tools/diag_build.py NAME -DFLAG [-DFLAG ...]; then run with
PVC3_LIB_PATH=panvc3_amd/_build/diag_NAME/libpvc3_hip.so python3 bench.py --no-build ...  (results of such a build are not
meant to be correct: only its timings are read)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panvc3_amd import build as b		# noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "panvc3_amd", "_build", "diag_" + name)
os.makedirs(out, exist_ok=True)
objs = []
procs = []
for src in b.SOURCES:
	o = os.path.join(out, os.path.splitext(src)[0] + ".o")
	objs.append(o)
	procs.append(subprocess.Popen([b.hipcc(), "-O3", "-std=c++17", "-fPIC", "--offload-arch=" + b.ARCH, "-ffp-contract=off", "-fvisibility=hidden", "-x", "hip"] + flags + ["-c", os.path.join(b.CSRC, src), "-o", o]))
for p in procs:
	if p.wait():
		raise SystemExit("compile failed")
subprocess.check_call([b.hipcc(), "-shared", "-fPIC", "--offload-arch=" + b.ARCH, "-o", os.path.join(out, "libpvc3_hip.so")] + objs)
print(os.path.join(out, "libpvc3_hip.so"))
