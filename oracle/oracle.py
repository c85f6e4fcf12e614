"""ctypes binding of the CPU oracle (oracle/pvc3_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under panvc3_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpvc3_oracle.so")

CIGAR_OPS = "MIDNSHP=X"
SCORE_FP64 = 0
SCORE_INT32_TRUNC = 1
FLAG_HAS_QUAL = 1
FLAG_HAS_MATE = 2
NO_MATE = 0xFFFFFFFF

STATUS_NAMES = {0: "ok", 1: "src_range", 2: "cigar_op", 3: "query_range", 4: "ref_range", 5: "capacity"}


def build(force=False):
	"""Compile the oracle with gcc.  Keyed by the content of the sources and the command line (file times do not survive the
	snapshot that carries the tree to a GPU box), so a fresh library is never rebuilt there."""
	from panvc3_amd.build import _done, _key, _stale
	srcs = [os.path.join(_HERE, f) for f in ("pvc3_oracle.c", "pvc3_oracle.h", "Makefile")]
	cmd = ["make", "-C", _HERE, "-s", "-B"]
	key = _key(srcs, cmd)
	if force or _stale(_LIB_PATH, key):
		subprocess.check_call(cmd)
		_done(_LIB_PATH, key)
	return _LIB_PATH


def is_stale():
	from panvc3_amd.build import _key, _stale
	srcs = [os.path.join(_HERE, f) for f in ("pvc3_oracle.c", "pvc3_oracle.h", "Makefile")]
	return _stale(_LIB_PATH, _key(srcs, ["make", "-C", _HERE, "-s", "-B"]))


class Run(C.Structure):
	_fields_ = [("query_location", C.c_uint32), ("query_length", C.c_uint32), ("ref_position", C.c_uint32), ("ref_length", C.c_uint32)]


RUN_DTYPE = np.dtype([("query_location", "<u4"), ("query_length", "<u4"), ("ref_position", "<u4"), ("ref_length", "<u4")])


class BatchIn(C.Structure):
	_fields_ = [
		("n", C.c_uint64),
		("src_pos", C.c_void_p), ("src_entry", C.c_void_p), ("dst_entry", C.c_void_p), ("flags", C.c_void_p),
		("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
		("seq_off", C.c_void_p), ("l_seq", C.c_void_p), ("seq4", C.c_void_p),
		("qual_off", C.c_void_p), ("qual", C.c_void_p),
		("mate_src_entry", C.c_void_p), ("mate_dst_entry", C.c_void_p), ("mate_pos", C.c_void_p),
	]


class BatchOut(C.Structure):
	_fields_ = [
		("dst_pos", C.c_void_p), ("mate_dst_pos", C.c_void_p), ("status", C.c_void_p),
		("out_cigar_off", C.c_void_p), ("out_cigar", C.c_void_p), ("out_cigar_cap", C.c_uint64),
		("run_off", C.c_void_p), ("runs", C.c_void_p), ("run_cap", C.c_uint64),
	]


_lib = None


def lib():
	global _lib
	if _lib is None:
		build()
		L = C.CDLL(_LIB_PATH)
		L.orc_index_new.restype = C.c_void_p
		L.orc_index_free.argtypes = [C.c_void_p]
		L.orc_index_add_aligned.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
		L.orc_index_add_aligned.restype = C.c_uint32
		L.orc_index_add_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
		L.orc_index_add_bits.restype = C.c_uint32
		L.orc_index_set_reference.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
		for f in (L.orc_entry_cols, L.orc_entry_zeros):
			f.argtypes = [C.c_void_p, C.c_uint32]
			f.restype = C.c_uint64
		for f in (L.orc_rank0, L.orc_select0, L.orc_rank0_naive, L.orc_select0_naive):
			f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
			f.restype = C.c_uint64
		L.orc_bit.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
		L.orc_project_position.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
		L.orc_project_position.restype = C.c_uint64
		L.orc_rewrite_cigar.argtypes = [
			C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32,
			C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
		L.orc_indel_runs.argtypes = [
			C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
			C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
		L.orc_align_global.argtypes = [
			C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_void_p, C.c_int32, C.c_int32, C.c_int,
			C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
		L.orc_collapse_cigar.argtypes = [C.c_void_p, C.c_uint32]
		L.orc_collapse_cigar.restype = C.c_uint32
		L.orc_project_alignment.argtypes = [
			C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_void_p,
			C.c_int32, C.c_int32, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
			C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
		L.orc_malde_score.argtypes = [C.c_int, C.c_int, C.c_int]
		L.orc_malde_score.restype = C.c_double
		L.orc_batch_project.argtypes = [C.c_void_p, C.POINTER(BatchIn), C.POINTER(BatchOut), C.c_int32, C.c_int32, C.c_int, C.c_int]
		_lib = L
	return _lib


# --- CIGAR helpers -------------------------------------------------------------------------------

def cigar_encode(text):
	"""'3=1D' -> uint32 array in BAM encoding (count << 4 | op)."""
	out = []
	num = ""
	for ch in text:
		if ch.isdigit():
			num += ch
		else:
			out.append((int(num) << 4) | CIGAR_OPS.index(ch))
			num = ""
	assert num == "", "trailing digits in CIGAR %r" % text
	return np.array(out, dtype=np.uint32)


def cigar_decode(arr):
	return "".join("%d%s" % (int(v) >> 4, CIGAR_OPS[int(v) & 0xf]) for v in arr)


def _ptr(a):
	return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
	"""One index (a list of gap bit-vector entries) + the reference algorithms over it."""

	def __init__(self):
		self._L = lib()
		self._h = C.c_void_p(self._L.orc_index_new())
		self._keep = []

	def __del__(self):
		try:
			self._L.orc_index_free(self._h)
		except Exception:
			pass

	# index -------------------------------------------------------------------------------------
	def add_aligned(self, aligned):
		b = aligned.encode() if isinstance(aligned, str) else bytes(aligned)
		return self._L.orc_index_add_aligned(self._h, b, len(b))

	def add_bits(self, words, n_cols):
		words = np.ascontiguousarray(words, dtype=np.uint64)
		assert words.size >= (n_cols + 63) // 64
		return self._L.orc_index_add_bits(self._h, _ptr(words), n_cols)

	def set_reference(self, entry, text):
		if isinstance(text, str):
			text = text.encode()
		buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, dtype=np.uint8)
		self._keep.append(buf)	# borrowed by the C side
		self._L.orc_index_set_reference(self._h, entry, _ptr(buf), buf.size)

	def cols(self, e): return self._L.orc_entry_cols(self._h, e)
	def zeros(self, e): return self._L.orc_entry_zeros(self._h, e)
	def rank0(self, e, i): return self._L.orc_rank0(self._h, e, i)
	def select0(self, e, k): return self._L.orc_select0(self._h, e, k)
	def rank0_naive(self, e, i): return self._L.orc_rank0_naive(self._h, e, i)
	def select0_naive(self, e, k): return self._L.orc_select0_naive(self._h, e, k)
	def bit(self, e, i): return self._L.orc_bit(self._h, e, i)
	def project_position(self, src, dst, pos): return self._L.orc_project_position(self._h, src, dst, pos)

	# algorithms --------------------------------------------------------------------------------
	def rewrite_cigar(self, src, dst, src_pos, cigar, query):
		cig = cigar_encode(cigar) if isinstance(cigar, str) else np.ascontiguousarray(cigar, dtype=np.uint32)
		q = query.encode() if isinstance(query, str) else bytes(query)
		cap = 64
		while True:
			out = np.zeros(cap, dtype=np.uint32)
			n_out = C.c_uint32()
			dst_pos = C.c_uint64()
			rc = self._L.orc_rewrite_cigar(self._h, src, dst, src_pos, _ptr(cig), cig.size, q, len(q), _ptr(out), cap, C.byref(n_out), C.byref(dst_pos))
			if rc == -5:
				cap = n_out.value + 16
				continue
			if rc:
				raise ValueError("orc_rewrite_cigar: %s" % STATUS_NAMES.get(-rc, rc))
			return dst_pos.value, out[:n_out.value].copy()

	def indel_runs(self, cigar, ref_pos=0):
		cig = cigar_encode(cigar) if isinstance(cigar, str) else np.ascontiguousarray(cigar, dtype=np.uint32)
		cap = max(1, cig.size)
		runs = np.zeros(cap, dtype=RUN_DTYPE)
		ref_loc = np.zeros(cap, dtype=np.uint32)
		first = np.zeros(cap, dtype=np.uint32)
		second = np.zeros(cap, dtype=np.uint32)
		n = C.c_uint32()
		fq = C.c_uint64()
		fr = C.c_uint64()
		rc = self._L.orc_indel_runs(_ptr(cig), cig.size, ref_pos, _ptr(runs), _ptr(ref_loc), _ptr(first), _ptr(second), cap, C.byref(n), C.byref(fq), C.byref(fr))
		if rc:
			raise ValueError("orc_indel_runs: %d" % rc)
		k = n.value
		return runs[:k].copy(), ref_loc[:k].copy(), first[:k].copy(), second[:k].copy(), fq.value, fr.value

	def align_global(self, ref, query, qual=None, gap_open=0, gap_ext=-1, score_mode=SCORE_FP64, tiebreak_mode=0):
		score_mode = int(score_mode) | (int(tiebreak_mode) << 8)
		r = ref.encode() if isinstance(ref, str) else bytes(ref)
		q = query.encode() if isinstance(query, str) else bytes(query)
		qa = None if qual is None else np.ascontiguousarray(qual, dtype=np.uint8)
		cap = len(r) + len(q) + 1
		out = np.zeros(cap, dtype=np.uint32)
		n = C.c_uint32()
		score = C.c_double()
		rc = self._L.orc_align_global(r, len(r), q, len(q), _ptr(qa), gap_open, gap_ext, score_mode, _ptr(out), cap, C.byref(n), C.byref(score))
		if rc:
			raise ValueError("orc_align_global: %d" % rc)
		return out[:n.value].copy(), score.value

	def project_alignment(self, src, dst, src_pos, cigar, query, qual=None, gap_open=0, gap_ext=-1, score_mode=SCORE_FP64, tiebreak_mode=0):
		"""Returns (dst_pos, cigar u32 array, runs structured array)."""
		score_mode = int(score_mode) | (int(tiebreak_mode) << 8)
		cig = cigar_encode(cigar) if isinstance(cigar, str) else np.ascontiguousarray(cigar, dtype=np.uint32)
		q = query.encode() if isinstance(query, str) else bytes(query)
		qa = None if qual is None else np.ascontiguousarray(qual, dtype=np.uint8)
		cap, rcap = 256, 16
		while True:
			out = np.zeros(cap, dtype=np.uint32)
			runs = np.zeros(rcap, dtype=RUN_DTYPE)
			n = C.c_uint32()
			nr = C.c_uint32()
			dp = C.c_uint64()
			rc = self._L.orc_project_alignment(
				self._h, src, dst, src_pos, _ptr(cig), cig.size, q, len(q), _ptr(qa), gap_open, gap_ext, score_mode,
				_ptr(out), cap, C.byref(n), _ptr(runs), rcap, C.byref(nr), C.byref(dp))
			if rc == -5:
				cap = max(cap, n.value + 16)
				rcap = max(rcap, nr.value + 4)
				continue
			if rc:
				raise ValueError("orc_project_alignment: %s" % STATUS_NAMES.get(-rc, rc))
			return dp.value, out[:n.value].copy(), runs[:nr.value].copy()

	def malde_score(self, matches, lhs_q, rhs_q):
		return self._L.orc_malde_score(int(matches), lhs_q, rhs_q)

	def collapse(self, cigar):
		cig = cigar_encode(cigar) if isinstance(cigar, str) else np.array(cigar, dtype=np.uint32)
		n = self._L.orc_collapse_cigar(_ptr(cig), cig.size)
		return cig[:n].copy()

	def batch_project(self, batch, gap_open=0, gap_ext=-1, score_mode=SCORE_FP64, n_threads=1, tiebreak_mode=0):
		"""batch: dict of numpy arrays with the field names of include/pvc3.h's pvc3_batch_in.
		Returns dict(dst_pos, mate_dst_pos, status, out_cigar_off, out_cigar, run_off, runs)."""
		score_mode = int(score_mode) | (int(tiebreak_mode) << 8)
		n = int(batch["src_pos"].size)
		bi = BatchIn()
		bi.n = n
		keep = []

		def put(name, dtype, optional=False):
			a = batch.get(name)
			if a is None:
				assert optional, name
				setattr(bi, name, None)
				return
			a = np.ascontiguousarray(a, dtype=dtype)
			keep.append(a)
			setattr(bi, name, a.ctypes.data)
		put("src_pos", np.uint32); put("src_entry", np.uint32); put("dst_entry", np.uint32); put("flags", np.uint32)
		put("cigar_off", np.uint64); put("cigar", np.uint32)
		put("seq_off", np.uint64); put("l_seq", np.uint32); put("seq4", np.uint8)
		put("qual_off", np.uint64, True); put("qual", np.uint8, True)
		put("mate_src_entry", np.uint32, True); put("mate_dst_entry", np.uint32, True); put("mate_pos", np.uint32, True)

		dst_pos = np.zeros(n, dtype=np.uint32)
		mate = np.zeros(n, dtype=np.uint32)
		status = np.zeros(n, dtype=np.int32)
		coff = np.zeros(n + 1, dtype=np.uint64)
		roff = np.zeros(n + 1, dtype=np.uint64)
		ccap = max(64, int(batch["cigar"].size) * 4 + 16 * n)
		rcap = max(16, n // 8)
		while True:
			ocig = np.zeros(ccap, dtype=np.uint32)
			runs = np.zeros(rcap, dtype=RUN_DTYPE)
			bo = BatchOut()
			bo.dst_pos = dst_pos.ctypes.data; bo.mate_dst_pos = mate.ctypes.data; bo.status = status.ctypes.data
			bo.out_cigar_off = coff.ctypes.data; bo.out_cigar = ocig.ctypes.data; bo.out_cigar_cap = ccap
			bo.run_off = roff.ctypes.data; bo.runs = runs.ctypes.data; bo.run_cap = rcap
			rc = self._L.orc_batch_project(self._h, C.byref(bi), C.byref(bo), gap_open, gap_ext, score_mode, n_threads)
			if rc == -2:
				ccap = int(coff[n]) + 16
				rcap = int(roff[n]) + 16
				continue
			if rc:
				raise ValueError("orc_batch_project: %d" % rc)
			break
		return dict(
			dst_pos=dst_pos, mate_dst_pos=mate, status=status,
			out_cigar_off=coff, out_cigar=ocig[:int(coff[n])].copy(),
			run_off=roff, runs=runs[:int(roff[n])].copy(),
		)
