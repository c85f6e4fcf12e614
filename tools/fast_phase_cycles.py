#!/usr/bin/env python3
"""Diagnostic: where a wavefront of k_project_fast spends its cycles, phase by phase (s_memtime stamps, each behind a full
s_waitcnt: a phase is charged with the latency of the loads it consumes; the stamps serialise what the kernel would
overlap, so the sum is larger than the kernel's real time per group — read the proportions).
	python3 tools/diag_build.py stamps -DPVC3_FAST_STAMPS      (here, no GPU needed)
	PVC3_LIB_PATH=panvc3_amd/_build/diag_stamps/libpvc3_hip.so python3 tools/fast_phase_cycles.py [config] [scale] [sorted]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np		# noqa: E402
import panvc3_amd		# noqa: E402
from panvc3_amd import _lib		# noqa: E402
from synth import synth		# noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "C3"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
gp, rp, desc = synth.config(config, scale=scale)
genome = synth.Genome(gp, n_threads=8)
reads = synth.Reads(genome, rp, n_threads=os.cpu_count())
batch = reads.batch()
if len(sys.argv) > 3:
	batch = synth.reorder_batch(batch, np.lexsort((batch["src_pos"], batch["src_entry"])))
index = panvc3_amd.Index()
genome.fill_index(index)
ctx = panvc3_amd.Context(0)
ctx.upload_index(index)
ctx.set_timing(True)
db = panvc3_amd.DeviceBatch(ctx, batch)
for _ in range(3):
	db.project()
	st = db.finish()
n = int(reads.n)
groups = (n + 63) // 64
n_waves = 4 * min(256 * 8, max(1, (groups + 3) // 4))
L = _lib.lib()
out = (C.c_ulonglong * (16 * n_waves))()
L.pvc3_debug_fast_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
assert 0 == L.pvc3_debug_fast_stamps(db._h, out, n_waves)
v = np.array(out[:], dtype=np.float64).reshape(n_waves, 16)
names = ["descriptors (+ next group's requested)", "entries + CIGAR", "window headers", "header arithmetic", "mate", "compare", "run count, scan, arena", "emission", "work list, outputs"]
g = v[:, 12].sum()
print("%s%s: k_project_fast %.3f ms with stamps; %d waves, %d groups" % (desc, " (sorted)" if len(sys.argv) > 3 else "", st["ms_project"], n_waves, g))
tot = v[:, :9].sum()
for k, name in enumerate(names):
	print("%-40s %9.0f cycles (100 MHz ticks x clock ratio) per group  %5.1f %%" % (name, v[:, k].sum() / g, 100 * v[:, k].sum() / tot))
print("%-40s %9.0f" % ("sum", tot / g))
