"""Builds libpvc3_hip.so (hand-written HIP for gfx950 + the C ABI) in-tree with hipcc.

hipcc cross-compiles gfx950 without a GPU.  The shared object lands next to this file so that it travels to the GPU
box with the repository snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libpvc3_hip.so")
SOURCES = ["api.cpp", "records_api.cpp", "split_api.cpp", "index.cpp", "kernels.hip", "project_fast.hip", "project_probes.hip", "project_mid.hip", "project_general.hip", "realign.hip", "mapq.hip", "records.hip", "split_ops.hip"]
ARCH = "gfx950"


def hipcc():
	for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
		if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
			return c
	raise RuntimeError("hipcc not found")


def _key(deps, cmd):
	"""Content key of a build step: the bytes of everything it reads and the command line.  (File times are useless here:
	the snapshot that carries the tree to a GPU box does not keep them, and a rebuild by several ranks at once would race.)"""
	h = hashlib.sha256()
	for d in sorted(os.path.abspath(x) for x in deps):
		h.update(os.path.basename(d).encode())
		with open(d, "rb") as f:
			h.update(f.read())
	# The tree lives at another path on the GPU box: the key must not depend on where the repository is.
	root = os.path.abspath(os.path.join(HERE, ".."))
	h.update("\0".join(os.path.normpath(c).replace(root, "<repo>") if os.path.isabs(c) or root in c else c for c in cmd).encode())
	return h.hexdigest()


def _stale(target, key):
	kp = target + ".key"
	return not (os.path.exists(target) and os.path.exists(kp) and open(kp).read() == key)


def _done(target, key):
	with open(target + ".key", "w") as f:
		f.write(key)


def _plan():
	"""(compile steps [(object, key, command)], link step (marker, key, command))."""
	headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
	headers.append(os.path.join(HERE, "..", "include", "pvc3.h"))
	headers.append(os.path.join(HERE, "..", "include", "pvc3_records.h"))
	flags = [
		"-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=off", "-fvisibility=hidden",
		"-Wall", "-Wextra", "-Wno-unused-parameter", "-x", "hip",
	]
	steps, keys = [], []
	for src in SOURCES:
		sp = os.path.join(CSRC, src)
		op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
		cmd = [hipcc()] + flags + ["-c", sp, "-o", op]
		key = _key([sp] + headers, cmd[1:])
		steps.append((op, key, cmd))
		keys.append(key)
	objs = [s[0] for s in steps]
	cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
	root = os.path.abspath(os.path.join(HERE, ".."))
	h = hashlib.sha256("\0".join(keys + [os.path.normpath(c).replace(root, "<repo>") if root in c else c for c in cmd[1:]]).encode()).hexdigest()
	return steps, (os.path.join(OBJ, "libpvc3_hip.so"), h, cmd)


def kernel_key():
	"""Content key of libpvc3_hip.so (every kernel source, header and flag): profiles that report per-kernel numbers record it,
	and bench.py only quotes such a number when the key matches the library that is running."""
	return _plan()[1][1]


def is_stale():
	"""True if build() would compile or link anything (nothing is built here)."""
	steps, (marker, link_key, _) = _plan()
	return any(_stale(op, key) for op, key, _ in steps) or not os.path.exists(LIB) or _stale(marker, link_key)


def build(force=False, verbose=False):
	os.makedirs(OBJ, exist_ok=True)
	steps, (marker, link_key, link_cmd) = _plan()
	for op, key, cmd in steps:
		if force or _stale(op, key):
			if verbose:
				print(" ".join(cmd), file=sys.stderr)
			subprocess.check_call(cmd)
			_done(op, key)
	if force or not os.path.exists(LIB) or _stale(marker, link_key):
		if verbose:
			print(" ".join(link_cmd), file=sys.stderr)
		subprocess.check_call(link_cmd)
		open(marker, "w").close()		# marker whose .key holds the link key (keeps the key file out of the package directory)
		_done(marker, link_key)
	build_bgzf(force=force, verbose=verbose)
	return LIB


BGZF_SRC = [os.path.join(HERE, "bgzf", "deflate.hip"), os.path.join(HERE, "bgzf", "inflate.hip")]
BGZF_LIB = os.path.join(HERE, "libpvc3_bgzf.so")


def build_bgzf(force=False, verbose=False):
	"""libpvc3_bgzf.so: the BGZF block compressor (panvc3_amd/bgzf/deflate.hip and inflate.hip, C ABI include/pvc3_bgzf.h).  A library of its own:
	the BAM writer's, not the projection's, and kernel_key() above stays the key of the projection kernels alone."""
	os.makedirs(OBJ, exist_ok=True)
	cmd = [hipcc(), "-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fvisibility=hidden", "-Wall", "-Wextra", "-shared", "-o", BGZF_LIB] + BGZF_SRC
	key = _key(BGZF_SRC + [os.path.join(HERE, "bgzf", "crc32_device.hpp"), os.path.join(HERE, "..", "include", "pvc3_bgzf.h")], cmd[1:])
	marker = os.path.join(OBJ, "libpvc3_bgzf.so")
	if force or not os.path.exists(BGZF_LIB) or _stale(marker, key):
		if verbose:
			print(" ".join(cmd), file=sys.stderr)
		subprocess.check_call(cmd)
		open(marker, "w").close()
		_done(marker, key)
	return BGZF_LIB


HOST = os.path.join(HERE, "host")
BIN = os.path.join(HERE, "..", "bin")


def build_host(force=False, verbose=False):
	"""The C++ host pipeline (g++): bin/panvc3_index_msa and bin/panvc3_project_alignments (the latter links libpvc3_hip.so)."""
	build(force=False, verbose=verbose)
	os.makedirs(BIN, exist_ok=True)
	headers = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [os.path.join(HERE, "..", "include", f) for f in ("pvc3.h", "pvc3_bgzf.h", "pvc3_records.h")]
	cxx = os.environ.get("CXX", "g++")
	flags = ["-O2", "-g", "-std=c++17", "-Wall", "-Wextra", "-pthread"]
	out = []
	for name, extra in (
		("panvc3_index_msa", ["-lz"]),
		("panvc3_project_alignments", ["-L" + HERE, "-lpvc3_hip", "-lpvc3_bgzf", "-lz", "-Wl,-rpath,$ORIGIN/../panvc3_amd", "-Wl,-rpath-link,/opt/rocm/lib"]),
		("panvc3_rewrite_cigar", ["-L" + HERE, "-lpvc3_hip", "-lz", "-Wl,-rpath,$ORIGIN/../panvc3_amd", "-Wl,-rpath-link,/opt/rocm/lib"]),
		("panvc3_convert_bed_positions", ["-L" + HERE, "-lpvc3_hip", "-lz", "-Wl,-rpath,$ORIGIN/../panvc3_amd", "-Wl,-rpath-link,/opt/rocm/lib"]),
		("panvc3_recalculate_mapq", ["-L" + HERE, "-lpvc3_hip", "-lpvc3_bgzf", "-lz", "-Wl,-rpath,$ORIGIN/../panvc3_amd", "-Wl,-rpath-link,/opt/rocm/lib"]),
	):
		src = os.path.join(HOST, name.replace("panvc3_", "") + "_main.cpp")
		dst = os.path.join(BIN, name)
		out.append(dst)
		cmd = [cxx] + flags + ["-o", dst, src] + extra
		key = _key([src, os.path.join(OBJ, "libpvc3_hip.so.key"), os.path.join(OBJ, "libpvc3_bgzf.so.key")] + headers, cmd[1:])
		if force or _stale(dst, key):
			if verbose:
				print(" ".join(cmd), file=sys.stderr)
			subprocess.check_call(cmd)
			_done(dst, key)
	return out


def build_cpp_tests(force=False, verbose=False):
	"""tests/cpp/*.cpp: programs that use the C++ mirror header include/pvc3.hpp (they link libpvc3_hip.so) or a header of the host pipeline."""
	build(force=False, verbose=verbose)
	src_dir = os.path.join(HERE, "..", "tests", "cpp")
	os.makedirs(BIN, exist_ok=True)
	out = []
	for name in sorted(os.listdir(src_dir)):
		if not name.endswith(".cpp"):
			continue
		src = os.path.join(src_dir, name)
		dst = os.path.join(BIN, "test_" + os.path.splitext(name)[0])
		out.append(dst)
		cmd = [os.environ.get("CXX", "g++"), "-O2", "-g", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(HERE, "..", "include"), "-o", dst, src,
			   "-L" + HERE, "-lpvc3_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN/../panvc3_amd", "-Wl,-rpath-link,/opt/rocm/lib"]
		host_headers = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp"))
		key = _key([src, os.path.join(HERE, "..", "include", "pvc3.hpp"), os.path.join(HERE, "..", "include", "pvc3.h"), os.path.join(OBJ, "libpvc3_hip.so.key")] + host_headers, cmd[1:])
		if force or _stale(dst, key):
			if verbose:
				print(" ".join(cmd), file=sys.stderr)
			subprocess.check_call(cmd)
			_done(dst, key)
	return out


if __name__ == "__main__":
	print(build(force="--force" in sys.argv, verbose=True))
	print(build_host(force="--force" in sys.argv, verbose=True))
	print(build_cpp_tests(force="--force" in sys.argv, verbose=True))
