#!/usr/bin/env python3
"""Where the time of the host-array entry points goes (pvc3_dbatch_create / _project / _finish / _download / _destroy and
pvc3_batch_project) for ordinary and for page-locked arrays: seconds per call, two rounds each.
usage: tools/host_buffers_probe.py [--records 4000000]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--records", type=int, default=4_000_000)
	ap.add_argument("--scale", type=float, default=0.1)
	args = ap.parse_args()
	import panvc3_amd
	from panvc3_amd import api
	from tests.synth import synth
	gp, rp, desc = synth.config("C3", scale=args.scale)
	genome = synth.Genome(gp, n_threads=8)
	reads = synth.Reads(genome, rp, n_threads=16)
	batch = reads.batch()
	k = min(int(reads.n), args.records)
	sb = synth.slice_batch(batch, 0, k)
	index = panvc3_amd.Index(); genome.fill_index(index)
	ctx = panvc3_amd.Context(0); ctx.upload_index(index)
	params = api.make_params(0, -1, 0)
	panvc3_amd.project_batch(ctx, synth.slice_batch(batch, 0, 100_000))

	def phases(b, label, out=None, pinned_out=False):
		t = [time.perf_counter()]
		db = panvc3_amd.DeviceBatch(ctx, b); t.append(time.perf_counter())
		db.project(params); t.append(time.perf_counter())
		db.finish(); t.append(time.perf_counter())
		res = db.download(pinned=pinned_out, out=out); t.append(time.perf_counter())
		db.close(); t.append(time.perf_counter())
		d = [t[i + 1] - t[i] for i in range(5)]
		print("%-28s create %.4f project %.4f finish %.4f download %.4f destroy %.4f | total %.4f s = %.1f M records/s" % ((label,) + tuple(d) + (t[-1] - t[0], k / (t[-1] - t[0]) / 1e6)), flush=True)
		return res

	for rnd in range(2):
		phases(sb, "ordinary arrays, round %d" % rnd)
	pb, owners = panvc3_amd.pinned_batch(sb)
	for rnd in range(2):
		phases(pb, "page-locked in, round %d" % rnd)
	# the one-call entry point (keeps its device batch)
	keep_out = None
	for rnd in range(3):
		t0 = time.perf_counter()
		res, _ = api.batch_project_call(ctx, sb, params, out=keep_out)
		keep_out = res["_arrays"]
		dt = time.perf_counter() - t0
		print("pvc3_batch_project, ordinary arrays, round %d: %.4f s = %.1f M records/s" % (rnd, dt, k / dt / 1e6), flush=True)


if __name__ == "__main__":
	main()
