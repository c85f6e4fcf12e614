"""Multi-GPU sharding of the hot path: records are independent, so each rank owns a contiguous slice of the record stream,
the index is replicated, and there is no collective on the data path.  Collectives are used only to agree
on timing (max over ranks) and to add up counts / checksums."""


def shard_range(n_total, rank, world_size):
	"""Contiguous record range [begin, end) of `rank`; ranges of all ranks tile [0, n_total) in order."""
	if world_size < 1 or not (0 <= rank < world_size):
		raise ValueError("bad rank / world size")
	return n_total * rank // world_size, n_total * (rank + 1) // world_size


def reduce_job(dist, device, elapsed_s, n_records, checksum=0, force=False):
	"""(max elapsed over ranks, total records, wrapping sum of checksums); `dist` is torch.distributed or None.  With `force`
	the collectives run even in a group of one rank (to exercise the backend)."""
	if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
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


def free_port():
	import socket
	s = socket.socket()
	s.bind(("127.0.0.1", 0))
	port = s.getsockname()[1]
	s.close()
	return port


def launch_ranks(cmd, world_size, env=None, timeout=None, poll_s=0.2):
	"""Starts `cmd` (argv list) once per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one process
	per GPU of this node, and waits for all of them.  Returns the list of exit codes in rank order.  When a rank fails, the
	others are terminated (their exit code is then the signal's) so that nobody waits at a barrier forever.

	The caller must not have initialised the GPU: the children are fresh processes (fork + exec of the interpreter), and a
	process that has touched the GPU is never exec'ed over.  Nothing here imports torch."""
	import os
	import subprocess
	import time
	if world_size < 1:
		raise ValueError("world_size must be positive")
	base = dict(os.environ if env is None else env)
	base.setdefault("MASTER_ADDR", "127.0.0.1")
	base.setdefault("MASTER_PORT", str(free_port()))
	base["WORLD_SIZE"] = str(world_size)
	base["LOCAL_WORLD_SIZE"] = str(world_size)
	procs = []
	for r in range(world_size):
		e = dict(base)
		e["RANK"] = str(r)
		e["LOCAL_RANK"] = str(r)
		procs.append(subprocess.Popen(list(cmd), env=e))
	t0 = time.time()
	failed = False
	while True:
		codes = [p.poll() for p in procs]
		if all(c is not None for c in codes):
			break
		failed = failed or any(c not in (None, 0) for c in codes)
		timed_out = timeout is not None and time.time() - t0 > timeout
		if failed or timed_out:
			for p in procs:
				if p.poll() is None:
					p.terminate()
			for p in procs:
				try:
					p.wait(timeout=20)
				except subprocess.TimeoutExpired:
					p.kill()
					p.wait()
			codes = [p.returncode for p in procs]
			if timed_out and not failed:
				codes = [c if c else -1 for c in codes]
			break
		time.sleep(poll_s)
	return codes
