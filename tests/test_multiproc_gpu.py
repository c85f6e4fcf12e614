"""The N > 1 path with the HIP library in every rank: two processes started by the launcher of bench.py --gpus N
(panvc3_amd.shard.launch_ranks), each projecting its shard_range slice through libpvc3_hip.so.  A test box has one GPU, so
both ranks use device 0 (RCCL refuses two ranks on one device, hence gloo for the three bookkeeping collectives; the data
path has no collective).  Concatenated results and the summed device checksum are compared with the oracle on the whole."""
import os
import sys

import pytest

from helpers import ROOT
from test_shard_gloo import _whole_oracle, check_shards_against_whole

pytestmark = pytest.mark.gpu


def test_two_ranks_hip_path(tmp_path, oracle_mod):
	from panvc3_amd.shard import launch_ranks
	scale = 0.002		# 80 000 records of config C5 (long indel events: projection, realignment and packing all take part)
	codes = launch_ranks([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), "hip", str(tmp_path), str(scale)], 2, timeout=600)
	assert codes == [0, 0]
	check_shards_against_whole(str(tmp_path), 2, _whole_oracle(scale))


def test_bench_through_rccl_with_one_rank(tmp_path):
	"""bench.py's N > 1 code path (RCCL process group, barriers around the timed region, MAX / SUM all-reduces, teardown) needs
	one GPU per rank, which a test box does not have; with PVC3_BENCH_FORCE_DIST=1 the same calls run in a group of one rank.
	Also: `bench.py --gpus 2` on a one-GPU box must fail cleanly through the launcher (rank 1 has no device)."""
	import json
	import subprocess
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
	env["PVC3_BENCH_FORCE_DIST"] = "1"
	cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C5", "--scale", "0.002", "--steps", "2", "--warmup", "1", "--cpu-baseline-seconds", "0", "--skip-host-buffers", "--no-build"]
	r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stderr[-2000:]
	line = json.loads(r.stdout.strip().splitlines()[-1])
	assert line["n_gpus"] == 1 and line["value"] > 0 and line["batch"]["indel_runs"] > 0
	# the same workload without the process group: same checksum
	env.pop("PVC3_BENCH_FORCE_DIST")
	r2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
	assert r2.returncode == 0, r2.stderr[-2000:]
	assert json.loads(r2.stdout.strip().splitlines()[-1])["batch"]["checksum"] == line["batch"]["checksum"]
