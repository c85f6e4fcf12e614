"""One rank of a multi-process test, started by panvc3_amd.shard.launch_ranks (the launcher bench.py --gpus N uses).

usage: rank_worker.py MODE OUT_DIR [SCALE]
  MODE oracle  the oracle stands in for the compute (CPU, gloo): what tests/test_shard_gloo.py checks without a GPU
  MODE hip     the HIP path on device (rank mod visible devices), gloo for the bookkeeping collectives: two processes
               share the one GPU of a test box, exactly as two ranks would each own a GPU of a node
  MODE fail    rank 1 exits with code 3 at once, the others wait at a barrier (the launcher must not hang)
Every rank builds the same index and the same read stream, projects its shard_range slice, saves it, and takes part in
reduce_job (max elapsed, sum of counts, wrapping sum of checksums).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
	if p not in sys.path:
		sys.path.insert(0, p)


def main():
	mode, out_dir = sys.argv[1], sys.argv[2]
	scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0003
	rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
	if "fail" == mode and 1 == rank:
		sys.exit(3)
	import numpy as np
	import torch.distributed as dist
	from panvc3_amd.shard import reduce_job, shard_range
	from tests.synth import synth
	from test_shard_gloo import _checksum
	dist.init_process_group(backend="gloo", rank=rank, world_size=world)
	try:
		if "fail" == mode:
			dist.barrier()		# never completes: rank 1 is gone; the launcher terminates us
			return
		gp, rp, _ = synth.config("C5", scale=scale)
		genome = synth.Genome(gp, n_threads=2)
		reads = synth.Reads(genome, rp, n_threads=2)
		batch = reads.batch()
		lo, hi = shard_range(reads.n, rank, world)
		shard = synth.slice_batch(batch, lo, hi)
		if "oracle" == mode:
			from oracle import oracle as orc
			o = orc.Oracle()
			genome.fill_index(o)
			res = o.batch_project(shard)
			csum = _checksum(res, lo)
		else:
			import panvc3_amd
			n_dev = panvc3_amd.device_count()
			assert n_dev > 0, "no GPU visible: the HIP path has no CPU fallback"
			index = panvc3_amd.Index()
			genome.fill_index(index)
			ctx = panvc3_amd.Context(rank % n_dev)
			ctx.upload_index(index)
			db = panvc3_amd.DeviceBatch(ctx, shard)
			db.project()
			stats = db.finish()
			assert stats["n_failed"] == 0
			csum = db.checksum(first_record=lo)		# computed on the device
			res = db.download()
			assert csum == _checksum(res, lo)
			db.close()
			ctx.close()
		np.savez(os.path.join(out_dir, "shard%d.npz" % rank), lo=lo, hi=hi, **{k: v for k, v in res.items()})
		elapsed, n_total, total = reduce_job(dist, "cpu", 0.5 + rank, hi - lo, csum)
		assert abs(elapsed - (0.5 + world - 1)) < 1e-9		# MAX over ranks
		assert n_total == reads.n							# SUM over ranks
		if 0 == rank:
			np.savez(os.path.join(out_dir, "job.npz"), csum=np.uint64(total), n=np.uint64(n_total))
		dist.barrier()
	finally:
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
