import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, HERE):
	if p not in sys.path:
		sys.path.insert(0, p)


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
	from oracle import oracle as orc
	orc.build()
	return orc


# Which projection kernels the GPU tests send the records through.  "default" is the product configuration (lane-parallel
# fast path, then the lane-per-record walk, then the cooperative column walk for what is left); the other two force every
# record through one of the slower kernels so that each of them is compared with the oracle on every test input
# ("walk_only" also sends every indel run through the striped alignment kernel instead of the packed ones).  "probes" is the
# product configuration of an index without position windows (one that does not fit the device): k_project_fast_probes
# looks the records up through the select samples and the paired blocks.
PATH_MODES = {
	"default": {}, "probes": {"position_windows": 0}, "mid_only": {"fast_path": 0},
	"walk_only": {"fast_path": 0, "mid_path": 0, "packed_dp": 0},
}


@pytest.fixture(scope="module", params=list(PATH_MODES))
def pvc3(request):
	import types
	import panvc3_amd
	assert panvc3_amd.device_count() > 0, "no GPU visible: the HIP path has no CPU fallback"
	options = PATH_MODES[request.param]

	class Context(panvc3_amd.Context):
		def __init__(self, *a, **kw):
			super().__init__(*a, **kw)
			for k, v in options.items():
				self.set_option(k, v)

	mod = types.SimpleNamespace(**{k: getattr(panvc3_amd, k) for k in dir(panvc3_amd) if not k.startswith("__")})
	mod.Context = Context
	mod.paths_mode = request.param
	return mod
