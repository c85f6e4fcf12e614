"""The N > 1 path on CPU: two processes (gloo), each owning a contiguous shard of the record stream, no data-path collective.

What bench.py does on GPUs is reproduced with the oracle as the stand-in compute (test infrastructure; no GPU here): shard
ranges tile the batch, per-shard results concatenated in rank order equal the unsharded result byte for byte, and the
collectives used for bookkeeping (max of elapsed times, sum of record counts, wrapping sum of checksums) behave.
"""
import json
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


def _whole_oracle(scale):
	from oracle import oracle as orc
	from tests.synth import synth
	gp, rp, _ = synth.config("C5", scale=scale)
	genome = synth.Genome(gp, n_threads=2)
	reads = synth.Reads(genome, rp, n_threads=2)
	o = orc.Oracle()
	genome.fill_index(o)
	return o.batch_project(reads.batch())


def check_shards_against_whole(out_dir, world, whole):
	"""Concatenation of the ranks' results in rank order == the unsharded result, byte for byte; the job checksum (summed
	through reduce_job) == the checksum of the whole."""
	shards = [np.load(os.path.join(out_dir, "shard%d.npz" % r)) for r in range(world)]
	assert int(shards[0]["lo"]) == 0
	for a, b in zip(shards, shards[1:]):
		assert int(a["hi"]) == int(b["lo"])
	for k in ("dst_pos", "mate_dst_pos", "status", "out_cigar", "runs"):
		assert np.array_equal(np.concatenate([s[k] for s in shards]), whole[k]), k
	for off in ("out_cigar_off", "run_off"):
		parts, base = [], 0
		for s in shards:
			parts.append(s[off][:-1].astype(np.uint64) + np.uint64(base))
			base += int(s[off][-1])
		assert np.array_equal(np.concatenate(parts + [np.array([base], dtype=np.uint64)]), whole[off]), off
	job = np.load(os.path.join(out_dir, "job.npz"))
	assert int(job["csum"]) == _checksum(whole, 0)
	assert int(job["n"]) == whole["dst_pos"].size


def test_launch_ranks_gloo(tmp_path):
	"""The launcher bench.py --gpus N uses (panvc3_amd.shard.launch_ranks), driven with gloo on CPU: three ranks."""
	from panvc3_amd.shard import launch_ranks
	world = 3
	codes = launch_ranks([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), "oracle", str(tmp_path)], world, timeout=600)
	assert codes == [0] * world
	check_shards_against_whole(str(tmp_path), world, _whole_oracle(0.0003))


def test_launch_ranks_failure_does_not_hang(tmp_path):
	"""A rank that dies takes the job down: the launcher terminates the ranks waiting at the barrier and reports non-zero."""
	from panvc3_amd.shard import launch_ranks
	import time
	t0 = time.time()
	codes = launch_ranks([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), "fail", str(tmp_path)], 2, timeout=300)
	assert codes[1] == 3 and codes[0] != 0
	assert time.time() - t0 < 120


def test_bench_parent_starts_ranks_and_reports_their_failure():
	"""`python bench.py --gpus 2` with no launcher: the parent starts two ranks itself (here, without a GPU, both stop with the
	"needs a GPU" message) and exits non-zero; with WORLD_SIZE that contradicts --gpus it refuses to run."""
	import subprocess
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.0001", "--steps", "1", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
	if r.returncode == 0:
		line = json.loads(r.stdout.strip().splitlines()[-1])		# a box with GPUs: the job ran on two ranks
		assert line["n_gpus"] == 2
	else:
		assert "ranks failed" in r.stderr and "rank 0" in r.stderr and "rank 1" in r.stderr
		assert r.stderr.count("bench.py needs a GPU") == 2
	env.update(RANK="0", WORLD_SIZE="3", LOCAL_RANK="0")
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=600)
	assert r.returncode != 0 and "does not match --gpus" in r.stderr


def test_bench_setup_of_eight_ranks_is_bounded():
	"""`python bench.py --gpus 8 --setup-only`: the parent starts eight ranks through launch_ranks with bench.py's own argument
	path; they meet over gloo, rank 0 generates the genome once into /dev/shm, all map it, every rank samples its own reads and
	builds the host side of the index, prints its set-up seconds and peak resident set, and stops before a GPU is touched.  The
	genome file is gone afterwards, no rank takes long, and no rank holds more than a small multiple of its own reads."""
	import glob
	import re
	import subprocess
	import time
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
	before = set(glob.glob("/dev/shm/pvc3_bench_*"))
	t0 = time.time()
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--scale", "0.002", "--setup-only"], env=env, capture_output=True, text=True, timeout=900)
	assert r.returncode == 0, r.stderr[-2000:]
	line = json.loads(r.stdout.strip().splitlines()[-1])
	assert line["setup_only"] and line["n_ranks"] == 8 and line["records_per_rank"] == 80000
	ranks = re.findall(r"\[bench\] rank (\d+) of 8: set-up ([0-9.]+) s \(genome ([0-9.]+), reads ([0-9.]+), index on the host ([0-9.]+); (\d+) threads\), peak resident set ([0-9.]+) MB", r.stderr)
	assert sorted(int(x[0]) for x in ranks) == list(range(8))
	for x in ranks:
		assert float(x[1]) < 120.0, x			# seconds of set-up per rank (2 Mbp genome, 80 000 records)
		assert float(x[6]) < 2500.0, x			# MB: torch + the mapped genome + this rank's reads; never eight genomes
	assert set(glob.glob("/dev/shm/pvc3_bench_*")) == before		# the shared genome's name was removed
	assert time.time() - t0 < 600
