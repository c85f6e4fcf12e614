"""Builds libpvc3_hip.so (hand-written HIP for gfx950 + the C ABI) in-tree with hipcc.

hipcc cross-compiles gfx950 without a GPU.  The shared object lands next to this file so that it travels to the GPU
box with the repository snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libpvc3_hip.so")
SOURCES = ["api.cpp", "index.cpp", "kernels.hip", "project_fast.hip", "project_mid.hip", "project_general.hip", "realign.hip"]
ARCH = "gfx950"


def hipcc():
	for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
		if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
			return c
	raise RuntimeError("hipcc not found")


def _newer(target, deps):
	if not os.path.exists(target):
		return True
	t = os.path.getmtime(target)
	return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
	os.makedirs(OBJ, exist_ok=True)
	headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
	headers.append(os.path.join(HERE, "..", "include", "pvc3.h"))
	flags = [
		"-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=off", "-fvisibility=hidden",
		"-Wall", "-Wextra", "-Wno-unused-parameter", "-x", "hip",
	]
	objs = []
	for src in SOURCES:
		sp = os.path.join(CSRC, src)
		op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
		objs.append(op)
		if force or _newer(op, [sp] + headers):
			cmd = [hipcc()] + flags + ["-c", sp, "-o", op]
			if verbose:
				print(" ".join(cmd), file=sys.stderr)
			subprocess.check_call(cmd)
	if force or _newer(LIB, objs):
		cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
		if verbose:
			print(" ".join(cmd), file=sys.stderr)
		subprocess.check_call(cmd)
	return LIB


HOST = os.path.join(HERE, "host")
BIN = os.path.join(HERE, "..", "bin")


def build_host(force=False, verbose=False):
	"""The C++ host pipeline (g++): bin/panvc3_index_msa and bin/panvc3_project_alignments (the latter links libpvc3_hip.so)."""
	build(force=False, verbose=verbose)
	os.makedirs(BIN, exist_ok=True)
	headers = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [os.path.join(HERE, "..", "include", "pvc3.h")]
	cxx = os.environ.get("CXX", "g++")
	flags = ["-O2", "-g", "-std=c++17", "-Wall", "-Wextra", "-pthread"]
	out = []
	for name, extra in (
		("panvc3_index_msa", ["-lz"]),
		("panvc3_project_alignments", ["-L" + HERE, "-lpvc3_hip", "-lz", "-Wl,-rpath,$ORIGIN/../panvc3_amd", "-Wl,-rpath-link,/opt/rocm/lib"]),
		("panvc3_convert_bed_positions", ["-L" + HERE, "-lpvc3_hip", "-lz", "-Wl,-rpath,$ORIGIN/../panvc3_amd", "-Wl,-rpath-link,/opt/rocm/lib"]),
	):
		src = os.path.join(HOST, name.replace("panvc3_", "") + "_main.cpp")
		dst = os.path.join(BIN, name)
		out.append(dst)
		if force or _newer(dst, [src, LIB] + headers):
			cmd = [cxx] + flags + ["-o", dst, src] + extra
			if verbose:
				print(" ".join(cmd), file=sys.stderr)
			subprocess.check_call(cmd)
	return out


if __name__ == "__main__":
	print(build(force="--force" in sys.argv, verbose=True))
	print(build_host(force="--force" in sys.argv, verbose=True))
