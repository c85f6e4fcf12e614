"""Multi-GPU sharding of the hot path: records are independent, so each rank owns a contiguous slice of the record stream,
the index is replicated, and there is no collective on the data path.  Collectives are used only to agree
on timing (max over ranks) and to add up counts / checksums."""


def shard_range(n_total, rank, world_size):
	"""Contiguous record range [begin, end) of `rank`; ranges of all ranks tile [0, n_total) in order."""
	if world_size < 1 or not (0 <= rank < world_size):
		raise ValueError("bad rank / world size")
	return n_total * rank // world_size, n_total * (rank + 1) // world_size


def reduce_job(dist, device, elapsed_s, n_records, checksum=0):
	"""(max elapsed over ranks, total records, wrapping sum of checksums); `dist` is torch.distributed or None."""
	if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
		return elapsed_s, n_records, checksum & 0xFFFFFFFFFFFFFFFF
	import torch
	t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
	dist.all_reduce(t, op=dist.ReduceOp.MAX)
	# counts and the 64-bit checksum as 16-bit limbs in int64 so that the SUM cannot overflow
	limbs = [n_records & 0xFFFF, (n_records >> 16) & 0xFFFF, (n_records >> 32) & 0xFFFF, (n_records >> 48) & 0xFFFF,
			 checksum & 0xFFFF, (checksum >> 16) & 0xFFFF, (checksum >> 32) & 0xFFFF, (checksum >> 48) & 0xFFFF]
	v = torch.tensor(limbs, dtype=torch.int64, device=device)
	dist.all_reduce(v, op=dist.ReduceOp.SUM)
	v = [int(x) for x in v.cpu()]
	n = v[0] + (v[1] << 16) + (v[2] << 32) + (v[3] << 48)
	c = (v[4] + (v[5] << 16) + (v[6] << 32) + (v[7] << 48)) & 0xFFFFFFFFFFFFFFFF
	return float(t.item()), n, c
