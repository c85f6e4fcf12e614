#!/usr/bin/env python3
"""Diagnostic: where the cycles of the packed alignment kernels go (set-up / alignment steps / traceback), per class.

Builds a second library with -DPVC3_DP_STAMPS (s_memtime stamps around the three phases of k_realign_dp; the stamps cost a
few percent themselves) into panvc3_amd/_build/diag/ and runs one C5 batch through it.  Usage, on a GPU box:
	python3 tools/dp_phase_cycles.py build      (here, no GPU needed)
	PVC3_LIB_PATH=panvc3_amd/_build/diag/libpvc3_hip.so python3 tools/dp_phase_cycles.py run [scale]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIAG = os.path.join(ROOT, "panvc3_amd", "_build", "diag")


def build():
	from panvc3_amd import build as b
	os.makedirs(DIAG, exist_ok=True)
	objs = []
	for src in b.SOURCES:
		o = os.path.join(DIAG, os.path.splitext(src)[0] + ".o")
		objs.append(o)
		subprocess.check_call([b.hipcc(), "-O3", "-std=c++17", "-fPIC", "--offload-arch=" + b.ARCH, "-ffp-contract=off", "-fvisibility=hidden", "-DPVC3_DP_STAMPS", "-x", "hip",
							   "-c", os.path.join(b.CSRC, src), "-o", o])
	subprocess.check_call([b.hipcc(), "-shared", "-fPIC", "--offload-arch=" + b.ARCH, "-o", os.path.join(DIAG, "libpvc3_hip.so")] + objs)


def run(scale):
	import numpy as np
	import panvc3_amd
	from panvc3_amd import _lib
	from tests.synth import synth
	gp, rp, desc = synth.config("C5", scale=scale)
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	index = panvc3_amd.Index()
	genome.fill_index(index)
	ctx = panvc3_amd.Context(0)
	ctx.upload_index(index)
	db = panvc3_amd.DeviceBatch(ctx, reads.batch())
	for _ in range(2):
		db.project()
		st = db.finish()
	L = _lib.lib()
	out = (C.c_ulonglong * 104)()
	L.pvc3_debug_wave_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
	assert 0 == L.pvc3_debug_wave_stats(db._h, out, 104)
	v = np.array(out[:]).reshape(13, 8)
	names = ["%d lanes, %s" % (w, o) for w in (3, 4, 6, 8, 12, 16) for o in ("query", "ref")]
	print("%s, scale %g: %d runs" % (desc, scale, st["n_runs"]))
	print("%-16s %10s %12s %12s %12s   cycles per task" % ("class", "tasks", "set-up", "alignment", "traceback"))
	for c, name in enumerate(names):
		t = max(1, int(v[c][3]))
		print("%-16s %10d %12.0f %12.0f %12.0f" % (name, v[c][3], v[c][0] / t, v[c][1] / t, v[c][2] / t))


if __name__ == "__main__":
	if sys.argv[1] == "build":
		build()
	else:
		run(float(sys.argv[2]) if len(sys.argv) > 2 else 0.25)
