"""The N > 1 path on CPU: two processes (gloo), each owning a contiguous shard of the record stream, no data-path collective.

What bench.py does on GPUs is reproduced with the oracle as the stand-in compute (test infrastructure; no GPU here): shard
ranges tile the batch, per-shard results concatenated in rank order equal the unsharded result byte for byte, and the
collectives used for bookkeeping (max of elapsed times, sum of record counts, wrapping sum of checksums) behave.
"""
import os
import sys

import numpy as np
import pytest

from helpers import ROOT

from panvc3_amd.shard import reduce_job, shard_range


def test_shard_ranges_tile_the_batch():
	for n in (0, 1, 7, 64, 1000, 40_000_001):
		for world in (1, 2, 3, 8):
			prev = 0
			for r in range(world):
				b, e = shard_range(n, r, world)
				assert b == prev and e >= b
				prev = e
			assert prev == n
	with pytest.raises(ValueError):
		shard_range(10, 2, 2)


def _checksum(res, first_record):
	"""Host restatement of pvc3_dbatch_checksum (k_checksum in panvc3_amd/csrc/kernels.hip)."""
	M = (1 << 64) - 1

	def mix(x):
		x = (x + 0x9E3779B97F4A7C15) & M
		x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
		x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
		return x ^ (x >> 31)
	total = 0
	n = res["dst_pos"].size
	for i in range(n):
		h = mix(first_record + i)
		h = mix(h ^ int(res["dst_pos"][i]))
		h = mix(h ^ int(res["mate_dst_pos"][i]))
		h = mix(h ^ (int(res["status"][i]) & 0xFFFFFFFF))
		c0, c1 = int(res["out_cigar_off"][i]), int(res["out_cigar_off"][i + 1])
		h = mix(h ^ (c1 - c0))
		for v in res["out_cigar"][c0:c1]:
			h = mix(h ^ int(v))
		r0, r1 = int(res["run_off"][i]), int(res["run_off"][i + 1])
		h = mix(h ^ (r1 - r0))
		for rr in res["runs"][r0:r1]:
			for f in ("query_location", "query_length", "ref_position", "ref_length"):
				h = mix(h ^ int(rr[f]))
		total = (total + h) & M
	return total


def _worker(rank, world, port, out_dir):
	sys.path.insert(0, ROOT)
	sys.path.insert(0, os.path.join(ROOT, "tests"))
	import torch.distributed as dist
	from oracle import oracle as orc
	from tests.synth import synth
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group(backend="gloo", rank=rank, world_size=world)
	try:
		gp, rp, _ = synth.config("C5", scale=0.0003)		# every rank builds the same index and the same read stream
		genome = synth.Genome(gp, n_threads=2)
		reads = synth.Reads(genome, rp, n_threads=2)
		batch = reads.batch()
		lo, hi = shard_range(reads.n, rank, world)
		o = orc.Oracle()
		genome.fill_index(o)
		res = o.batch_project(synth.slice_batch(batch, lo, hi))
		np.savez(os.path.join(out_dir, "shard%d.npz" % rank), lo=lo, hi=hi, **{k: v for k, v in res.items()})
		elapsed, n_total, csum = reduce_job(dist, "cpu", 0.5 + rank, hi - lo, _checksum(res, lo))
		assert elapsed == pytest.approx(0.5 + world - 1)		# MAX over ranks
		assert n_total == reads.n							# SUM over ranks
		if rank == 0:
			whole = o.batch_project(batch)
			np.savez(os.path.join(out_dir, "whole.npz"), csum=np.uint64(csum), whole_csum=np.uint64(_checksum(whole, 0)), **{k: v for k, v in whole.items()})
		dist.barrier()
	finally:
		dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
	import socket
	import torch.multiprocessing as mp
	s = socket.socket()
	s.bind(("127.0.0.1", 0))
	port = s.getsockname()[1]
	s.close()
	world = 2
	mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
	whole = np.load(tmp_path / "whole.npz")
	shards = [np.load(tmp_path / ("shard%d.npz" % r)) for r in range(world)]
	assert int(shards[0]["lo"]) == 0 and int(shards[0]["hi"]) == int(shards[1]["lo"])
	# concatenation in rank order == the unsharded result, byte for byte
	for k in ("dst_pos", "mate_dst_pos", "status", "out_cigar", "runs"):
		assert np.array_equal(np.concatenate([s[k] for s in shards]), whole[k]), k
	for off in ("out_cigar_off", "run_off"):
		cat = np.concatenate([shards[0][off][:-1], shards[1][off] + shards[0][off][-1]])
		assert np.array_equal(cat, whole[off]), off
	# the checksum is additive over shards
	assert int(whole["csum"]) == int(whole["whole_csum"])
