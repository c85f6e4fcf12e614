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
