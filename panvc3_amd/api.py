"""Host-side Python mirror of the C ABI (include/pvc3.h) and of the reference's alignment_projector interface.

`AlignmentProjector.project_alignment()` takes the arguments of panvc3::alignment_projector::project_alignment
(include/panvc3/alignment_projector.hh:58-68) and exposes alignment(), realigned_reference_ranges() and
realigned_query_ranges() (:70-74), so that parity tests read like tests/alignment_projector.cc.  Everything is computed by
the HIP kernels; nothing here falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BatchIn, BatchOut, BatchStats, MapqBatch, MapqStats, Params, PipelineCaps, RescoreParams, check

CIGAR_OPS = "MIDNSHP=X"
SCORE_FP64 = 0
SCORE_INT32_TRUNC = 1
FLAG_HAS_QUAL = 1
FLAG_HAS_MATE = 2
NO_MATE = 0xFFFFFFFF
RUN_DTYPE = np.dtype([("query_location", "<u4"), ("query_length", "<u4"), ("ref_position", "<u4"), ("ref_length", "<u4")])
_SEQ4 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def cigar_encode(text):
	out, num = [], ""
	for ch in text:
		if ch.isdigit():
			num += ch
		else:
			out.append((int(num) << 4) | CIGAR_OPS.index(ch))
			num = ""
	if num:
		raise ValueError("trailing digits in CIGAR %r" % text)
	return np.array(out, dtype=np.uint32)


def cigar_decode(arr):
	return "".join("%d%s" % (int(v) >> 4, CIGAR_OPS[int(v) & 0xf]) for v in arr)


def seq4_encode(seq):
	"""ASCII bases -> BAM 4-bit packing (high nibble first); unknown characters become N (15)."""
	codes = np.fromiter((_SEQ4.get(c.upper(), 15) for c in seq), dtype=np.uint8, count=len(seq))
	if codes.size & 1:
		codes = np.append(codes, np.uint8(0))
	return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)


def device_count():
	n = C.c_int()
	check(_lib.lib().pvc3_device_count(C.byref(n)))
	return n.value


class Index:
	"""The MSA gap bit-vector index (msa_index, include/panvc3/msa_index.hh): one entry per (chromosome, sequence)."""

	def __init__(self):
		self._L = _lib.lib()
		self._h = C.c_void_p()
		check(self._L.pvc3_index_create(C.byref(self._h)))

	def __del__(self):
		if getattr(self, "_h", None):
			self._L.pvc3_index_destroy(self._h)
			self._h = None

	def add_aligned(self, aligned):
		b = aligned.encode() if isinstance(aligned, str) else bytes(aligned)
		e = C.c_uint32()
		check(self._L.pvc3_index_add_aligned(self._h, b, len(b), C.byref(e)))
		return e.value

	def add_bits(self, words, n_cols):
		words = np.ascontiguousarray(words, dtype=np.uint64)
		if words.size < (n_cols + 63) // 64:
			raise ValueError("too few words")
		e = C.c_uint32()
		check(self._L.pvc3_index_add_bits(self._h, words.ctypes.data, n_cols, C.byref(e)))
		return e.value

	def set_reference(self, entry, text):
		if isinstance(text, str):
			text = text.encode()
		buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, dtype=np.uint8)
		check(self._L.pvc3_index_set_reference(self._h, entry, buf.ctypes.data, buf.size))

	def set_destination(self, entry, dst_entry):
		"""pvc3_index_set_destination: the entry that `entry`'s records are projected to."""
		check(self._L.pvc3_index_set_destination(self._h, entry, dst_entry))

	def entry_count(self):
		n = C.c_uint32()
		check(self._L.pvc3_index_entry_count(self._h, C.byref(n)))
		return n.value

	def entry_info(self, entry):
		a, b = C.c_uint64(), C.c_uint64()
		check(self._L.pvc3_index_entry_info(self._h, entry, C.byref(a), C.byref(b)))
		return a.value, b.value

	def entry_layout(self, entry):
		"""(blocks [n_blocks, 8] uint64, sel uint32): host copy of the HBM layout."""
		bp, sp = C.c_void_p(), C.c_void_p()
		nb, ns = C.c_uint64(), C.c_uint64()
		check(self._L.pvc3_index_entry_layout(self._h, entry, C.byref(bp), C.byref(nb), C.byref(sp), C.byref(ns)))
		blocks = np.ctypeslib.as_array(C.cast(bp, C.POINTER(C.c_uint64)), shape=(nb.value * 8,)).copy().reshape(nb.value, 8)
		sel = np.ctypeslib.as_array(C.cast(sp, C.POINTER(C.c_uint32)), shape=(ns.value,)).copy()
		return blocks, sel


def make_sequence_entry_pair(index, lhs, rhs):
	"""panvc3::make_sequence_entry_pair (msa_index.hh:214-236): two gapped strings -> (src entry, dst entry)."""
	if len(lhs) != len(rhs):
		raise ValueError("aligned sequences differ in length")
	return index.add_aligned(lhs), index.add_aligned(rhs)


_BATCH_FIELDS = [
	("src_pos", np.uint32, False), ("src_entry", np.uint32, False), ("dst_entry", np.uint32, False), ("flags", np.uint32, False),
	("cigar_off", np.uint64, False), ("cigar", np.uint32, False),
	("seq_off", np.uint64, False), ("l_seq", np.uint32, False), ("seq4", np.uint8, False),
	("qual_off", np.uint64, True), ("qual", np.uint8, True),
	("mate_src_entry", np.uint32, True), ("mate_dst_entry", np.uint32, True), ("mate_pos", np.uint32, True),
]


def _pack_batch_in(batch):
	bi = BatchIn()
	keep = []
	bi.n = int(np.asarray(batch["src_pos"]).size)
	for name, dtype, optional in _BATCH_FIELDS:
		a = batch.get(name)
		if a is None:
			if not optional:
				raise KeyError(name)
			setattr(bi, name, None)
			continue
		a = np.ascontiguousarray(a, dtype=dtype)
		keep.append(a)
		setattr(bi, name, a.ctypes.data if a.size else None)
	# zero-length but required arrays still need a non-null offset array
	for name in ("cigar_off", "seq_off"):
		a = np.ascontiguousarray(batch[name], dtype=np.uint64)
		keep.append(a)
		setattr(bi, name, a.ctypes.data)
	return bi, keep


def _stats_dict(st):
	return {k: getattr(st, k) for k, _ in BatchStats._fields_ if k != "reserved"}


class Context:
	"""One GPU: stream, index replica in HBM, workspaces."""

	def __init__(self, device=0):
		self._L = _lib.lib()
		self._h = C.c_void_p()
		check(self._L.pvc3_context_create(device, C.byref(self._h)))
		self.device = device

	def close(self):
		if getattr(self, "_h", None):
			self._L.pvc3_context_destroy(self._h)
			self._h = None

	def __del__(self):
		self.close()

	def upload_index(self, index):
		check(self._L.pvc3_context_upload_index(self._h, index._h))

	def set_stream(self, hip_stream_ptr):
		check(self._L.pvc3_context_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

	def set_timing(self, enabled=True):
		check(self._L.pvc3_context_set_timing(self._h, int(enabled)))

	def set_option(self, key, value):
		check(self._L.pvc3_context_set_option(self._h, key.encode(), int(value)))

	def synchronize(self):
		check(self._L.pvc3_context_synchronize(self._h))

	def hbm_bytes(self):
		a, b = C.c_uint64(), C.c_uint64()
		check(self._L.pvc3_context_hbm_bytes(self._h, C.byref(a), C.byref(b)))
		return a.value, b.value

	def project_positions(self, src_entry, dst_entry, pos):
		"""Batched msa_index::sequence_entry::project_position (msa_index.hh:253-258)."""
		pos = np.ascontiguousarray(pos, dtype=np.uint32)
		src = np.ascontiguousarray(np.broadcast_to(np.asarray(src_entry, dtype=np.uint32), pos.shape))
		dst = np.ascontiguousarray(np.broadcast_to(np.asarray(dst_entry, dtype=np.uint32), pos.shape))
		out = np.zeros(pos.size, dtype=np.uint32)
		check(self._L.pvc3_project_positions(self._h, pos.size, src.ctypes.data, dst.ctypes.data, pos.ctypes.data, out.ctypes.data))
		return out


class Reference:
	"""Reference sequences in HBM (pvc3_reference_*) for Context.split_match_ops."""

	def __init__(self, ctx, sequences):
		self._L = _lib.lib()
		self._ctx = ctx
		self._keep = [np.frombuffer(s if isinstance(s, (bytes, bytearray)) else s.encode(), dtype=np.uint8) for s in sequences]
		ptrs = (C.c_void_p * max(1, len(self._keep)))(*[a.ctypes.data if a.size else None for a in self._keep])
		lens = np.array([a.size for a in self._keep], dtype=np.uint64)
		h = C.c_void_p()
		check(self._L.pvc3_reference_create(ctx._h, len(self._keep), ptrs, lens.ctypes.data, C.byref(h)))
		self._h = h

	def close(self):
		if self._h:
			self._L.pvc3_reference_destroy(self._h)
			self._h = None

	def __del__(self):
		self.close()


def split_match_ops(ctx, reference, sequence, pos, cigars, seqs4, l_seq):
	"""pvc3_split_match_ops (rewrite-cigar/main.cc:191-284): cigars[i] = array of BAM operations, seqs4[i] = the record's packed bases.
	Returns (list of operation arrays, list of N-position maps as bytes, statuses)."""
	L = _lib.lib()
	n = len(cigars)
	cigar_off = np.zeros(n + 1, dtype=np.uint64); cigar_off[1:] = np.cumsum([len(c) for c in cigars])
	cigar = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint32) for c in cigars]) if n and cigar_off[n] else np.zeros(0, dtype=np.uint32))
	seq_off = np.zeros(n, dtype=np.uint64)
	if n: seq_off[1:] = np.cumsum([len(b) for b in seqs4])[:-1]
	seq4 = np.frombuffer(b"".join(bytes(b) for b in seqs4), dtype=np.uint8)
	sequence = np.ascontiguousarray(sequence, dtype=np.uint32); pos = np.ascontiguousarray(pos, dtype=np.uint32); l_seq = np.ascontiguousarray(l_seq, dtype=np.uint32)
	si = _lib.SplitIn(n, sequence.ctypes.data, pos.ctypes.data, cigar_off.ctypes.data, cigar.ctypes.data if cigar.size else None, seq_off.ctypes.data,
					  seq4.ctypes.data if seq4.size else None, seq4.size, l_seq.ctypes.data)
	out_off = np.zeros(n + 1, dtype=np.uint64); map_off = np.zeros(n + 1, dtype=np.uint64); status = np.zeros(max(1, n), dtype=np.int32)
	cap_c, cap_m = 0, 0		# the first call asks
	for _ in range(2):
		out_c = np.zeros(max(1, cap_c), dtype=np.uint32); out_m = np.zeros(max(1, cap_m), dtype=np.uint8)
		so = _lib.SplitOut(out_off.ctypes.data, out_c.ctypes.data, cap_c, map_off.ctypes.data, out_m.ctypes.data, cap_m, status.ctypes.data)
		nc, nm = C.c_uint64(), C.c_uint64()
		rc = L.pvc3_split_match_ops(ctx._h, reference._h, C.byref(si), C.byref(so), C.byref(nc), C.byref(nm))
		if -5 == rc:		# PVC3_E_CAPACITY: the sizes needed are in nc, nm
			cap_c, cap_m = nc.value, nm.value
			continue
		check(rc)
		break
	else:
		raise RuntimeError("pvc3_split_match_ops asked for capacity twice")
	return ([out_c[int(out_off[i]):int(out_off[i + 1])].copy() for i in range(n)], [out_m[int(map_off[i]):int(map_off[i + 1])].tobytes() for i in range(n)], status[:n].copy())


class DeviceBatch:
	"""A batch resident in HBM (pvc3_dbatch_*)."""

	def __init__(self, ctx, batch):
		self._L = _lib.lib()
		self.ctx = ctx
		self._h = C.c_void_p()
		bi, keep = _pack_batch_in(batch)
		check(self._L.pvc3_dbatch_create(ctx._h, C.byref(bi), C.byref(self._h)))
		self.n = bi.n

	def close(self):
		if getattr(self, "_h", None):
			self._L.pvc3_dbatch_destroy(self._h)
			self._h = None

	def __del__(self):
		self.close()

	def project(self, params=None):
		"""Enqueues the kernels on the context's stream (asynchronous)."""
		p = params if isinstance(params, Params) else make_params(**(params or {}))
		check(self._L.pvc3_dbatch_project(self.ctx._h, self._h, C.byref(p)))

	def finish(self):
		st = BatchStats()
		check(self._L.pvc3_dbatch_finish(self.ctx._h, self._h, C.byref(st)))
		self.stats = _stats_dict(st)
		return self.stats

	def download(self, pinned=False, out=None):
		"""Copies the packed results to host arrays: fresh ones, fresh page-locked ones, or the arrays of `out` (a previous
		result of the same sizes, e.g. page-locked arrays that are reused from batch to batch)."""
		n = self.n
		ccap = int(self.stats["out_cigar_total"])
		rcap = int(self.stats["n_runs"])
		shapes = dict(dst_pos=(n, np.uint32), mate_dst_pos=(n, np.uint32), status=(n, np.int32), out_cigar_off=(n + 1, np.uint64), run_off=(n + 1, np.uint64),
					  out_cigar=(max(1, ccap), np.uint32), runs=(max(1, rcap), RUN_DTYPE))
		if out is not None:
			res = {k: out[k] for k in shapes}
			for k, (cnt, dt) in shapes.items():
				if res[k].size < cnt or res[k].dtype != np.dtype(dt):
					raise ValueError("download: array %s of `out` is too small" % k)
		elif pinned:
			self._pinned_out = [PinnedArray((cnt,), dt) for cnt, dt in shapes.values()]
			res = {k: pa.array for k, pa in zip(shapes, self._pinned_out)}
		else:
			res = {k: np.zeros(cnt, dtype=dt) for k, (cnt, dt) in shapes.items()}
		bo = BatchOut()
		for k in ("dst_pos", "mate_dst_pos", "status", "out_cigar_off", "out_cigar", "run_off", "runs"):
			setattr(bo, k, res[k].ctypes.data)
		bo.out_cigar_cap = res["out_cigar"].size
		bo.run_cap = res["runs"].size
		check(self._L.pvc3_dbatch_download(self.ctx._h, self._h, C.byref(bo)))
		res = dict(res)
		for k, (cnt, dt) in shapes.items():
			res[k] = res[k][:cnt]
		res["out_cigar"] = res["out_cigar"][:ccap]
		res["runs"] = res["runs"][:rcap]
		return res

	def checksum(self, first_record=0):
		v = C.c_uint64()
		check(self._L.pvc3_dbatch_checksum(self.ctx._h, self._h, first_record, C.byref(v)))
		return v.value


class PinnedArray:
	"""A numpy array in page-locked host memory (pvc3_host_alloc): `.array`; freed by close() or on collection."""

	def __init__(self, shape, dtype):
		self._L = _lib.lib()
		dt = np.dtype(dtype)
		n = int(np.prod(shape))
		self._p = C.c_void_p()
		check(self._L.pvc3_host_alloc(max(1, n * dt.itemsize), C.byref(self._p)))
		buf = (C.c_uint8 * max(1, n * dt.itemsize)).from_address(self._p.value)
		self.array = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

	def close(self):
		if getattr(self, "_p", None) and self._p.value:
			self.array = None
			self._L.pvc3_host_free(self._p)
			self._p = C.c_void_p()

	def __del__(self):
		self.close()


def pinned_batch(batch):
	"""A copy of a batch (dict of arrays) in page-locked memory.  Returns (batch of arrays, owners to keep alive)."""
	out, owners = {}, []
	for k, v in batch.items():
		if not isinstance(v, np.ndarray):
			continue
		pa = PinnedArray(v.shape, v.dtype)
		pa.array[...] = v
		out[k] = pa.array
		owners.append(pa)
	return out, owners


NO_POS = 0xFFFFFFFF
SCORE_MIN = -1.7976931348623157e308		# the score of a record without an AS tag (recalculate-mapq/main.cc:50)
_MAPQ_FIELDS = ("ref_id", "pos", "mate_ref_id", "mate_pos", "orig_ref_id", "orig_pos", "orig_mate_ref_id", "orig_mate_pos", "l_seq", "seq_class")


def rescore_alignments(ctx, batch, xn_off, xn, min_mismatch_penalty=2, max_mismatch_penalty=6, n_penalty=1, gap_opening_penalty=5, gap_extension_penalty=3):
	"""pvc3_rescore_alignments: cigar_alignment_scorer::calculate_score (recalculate-mapq/main.cc:189-289) for every record of a
	batch (cigar_off / cigar / seq_off / seq4 / l_seq / qual_off / qual as in pvc3_batch_in) -> (scores float64, status int32)."""
	n = int(np.asarray(batch["l_seq"]).size)
	a = {k: np.ascontiguousarray(batch[k], dtype=dt) for k, dt in (("cigar_off", np.uint64), ("cigar", np.uint32), ("seq_off", np.uint64), ("seq4", np.uint8), ("l_seq", np.uint32))}
	has_qual = batch.get("qual") is not None and batch.get("qual_off") is not None
	qo = np.ascontiguousarray(batch["qual_off"], dtype=np.uint64) if has_qual else None
	q = np.ascontiguousarray(batch["qual"], dtype=np.uint8) if has_qual else None
	xo = np.ascontiguousarray(xn_off, dtype=np.uint64)
	x = np.ascontiguousarray(xn, dtype=np.uint8)
	score = np.zeros(n, dtype=np.float64)
	status = np.zeros(n, dtype=np.int32)
	prm = RescoreParams(min_mismatch_penalty, max_mismatch_penalty, n_penalty, gap_opening_penalty, gap_extension_penalty)
	ptr = lambda v: v.ctypes.data if v is not None and v.size else None
	check(_lib.lib().pvc3_rescore_alignments(ctx._h, n, a["cigar_off"].ctypes.data, ptr(a["cigar"]), a["seq_off"].ctypes.data, ptr(a["seq4"]), ptr(a["l_seq"]),
		qo.ctypes.data if has_qual else None, ptr(q) if has_qual else None, xo.ctypes.data, ptr(x), C.byref(prm), ptr(score), ptr(status)))
	return score, status


def recalculate_mapq(ctx, group_off, fields):
	"""pvc3_recalculate_mapq: mapq_scorer::process_alignment_group (recalculate-mapq/main.cc:962-1180) for a batch of QNAME
	groups.  fields: dict of per-record arrays (ref_id, pos, mate_ref_id, mate_pos, orig_ref_id, orig_pos, orig_mate_ref_id,
	orig_mate_pos, l_seq, seq_class as uint32 with NO_POS for unset values; score as float64 with SCORE_MIN for a missing AS).
	-> (mapq uint8 [records], group_emitted uint8 [groups], statistics dict)."""
	go = np.ascontiguousarray(group_off, dtype=np.uint64)
	ng = go.size - 1
	n = int(go[-1]) if go.size else 0
	mb = MapqBatch()
	keep = [go]
	mb.n_groups = ng
	mb.group_off = go.ctypes.data
	for k in _MAPQ_FIELDS:
		v = np.ascontiguousarray(fields[k], dtype=np.uint32)
		keep.append(v)
		setattr(mb, k, v.ctypes.data if v.size else None)
	sc = np.ascontiguousarray(fields["score"], dtype=np.float64)
	keep.append(sc)
	mb.score = sc.ctypes.data if sc.size else None
	mapq = np.zeros(max(1, n), dtype=np.uint8)
	emitted = np.zeros(max(1, ng), dtype=np.uint8)
	st = MapqStats()
	check(_lib.lib().pvc3_recalculate_mapq(ctx._h, C.byref(mb), mapq.ctypes.data, emitted.ctypes.data, C.byref(st)))
	return mapq[:n], emitted[:ng], {k: getattr(st, k) for k, _ in MapqStats._fields_}


class Pipeline:
	"""pvc3_pipeline_*: a stream of batches with the upload of batch k + 1, the kernels of batch k and the download of batch
	k - 1 overlapping.  submit() takes the batch arrays and the output arrays (see output_arrays()); both should be
	page-locked (PinnedArray / pinned_batch) and must stay untouched until wait() has returned the batch."""

	def __init__(self, ctx, max_records, max_cigar_ops, max_seq_bytes, max_qual_bytes=0, depth=3):
		self._L = _lib.lib()
		self.ctx = ctx
		self._h = C.c_void_p()
		caps = PipelineCaps(max_records, max_cigar_ops, max_seq_bytes, max_qual_bytes, depth)
		check(self._L.pvc3_pipeline_create(ctx._h, C.byref(caps), C.byref(self._h)))
		self._keep = []

	def close(self):
		if getattr(self, "_h", None):
			self._L.pvc3_pipeline_destroy(self._h)
			self._h = None

	def __del__(self):
		self.close()

	@staticmethod
	def output_arrays(max_records, max_cigar_ops, max_runs, pinned=True):
		"""A set of output arrays for one batch in flight: dict of numpy arrays (+ "_owner" keeping page-locked memory alive)."""
		shapes = dict(dst_pos=(max_records, np.uint32), mate_dst_pos=(max_records, np.uint32), status=(max_records, np.int32),
					  out_cigar_off=(max_records + 1, np.uint64), run_off=(max_records + 1, np.uint64), out_cigar=(max(1, max_cigar_ops), np.uint32), runs=(max(1, max_runs), RUN_DTYPE))
		if pinned:
			owners = [PinnedArray((cnt,), dt) for cnt, dt in shapes.values()]
			res = {k: pa.array for k, pa in zip(shapes, owners)}
			res["_owner"] = owners
		else:
			res = {k: np.zeros(cnt, dtype=dt) for k, (cnt, dt) in shapes.items()}
		return res

	def submit(self, batch, out, params=None):
		bi, keep = _pack_batch_in(batch)
		bo = BatchOut()
		for k in ("dst_pos", "mate_dst_pos", "status", "out_cigar_off", "out_cigar", "run_off", "runs"):
			setattr(bo, k, out[k].ctypes.data)
		bo.out_cigar_cap = out["out_cigar"].size
		bo.run_cap = out["runs"].size
		p = params if isinstance(params, Params) else make_params(**(params or {}))
		check(self._L.pvc3_pipeline_submit(self._h, C.byref(bi), C.byref(bo), C.byref(p)))
		self._keep.append((keep, out, bi.n))

	def wait(self):
		"""(results of the oldest batch as views of its output arrays, stats)."""
		st = BatchStats()
		keep, out, n = self._keep.pop(0)
		check(self._L.pvc3_pipeline_wait(self._h, C.byref(st)))
		stats = _stats_dict(st)
		res = {k: out[k][:n] for k in ("dst_pos", "mate_dst_pos", "status")}
		res["out_cigar_off"] = out["out_cigar_off"][:n + 1]
		res["run_off"] = out["run_off"][:n + 1]
		res["out_cigar"] = out["out_cigar"][:int(stats["out_cigar_total"])]
		res["runs"] = out["runs"][:int(stats["n_runs"])]
		if "_owner" in out:
			res["_owner"] = out["_owner"]		# the views above do not keep page-locked memory alive
		return res, stats

	def in_flight(self):
		v = C.c_uint32()
		check(self._L.pvc3_pipeline_in_flight(self._h, C.byref(v)))
		return v.value


def pipeline_bench(ctx, batch, params, batch_records=2_000_000, total_records=16_000_000, depth=3):
	"""PCIe-inclusive throughput of the pipelined entry point (SURVEY.md §8d protocol ii): the batch is cut into slices of
	`batch_records`, copied once into page-locked arrays (the state a caller that fills its arrays directly is in), and the
	slices are streamed through pvc3_pipeline_submit / _wait.  Returns the dict bench.py prints as "pipeline"."""
	import time
	n_all = int(np.asarray(batch["src_pos"]).size)
	n_use = min(n_all, total_records)
	n_slices = max(1, n_use // batch_records)
	per = n_use // n_slices

	def cut(lo, hi):
		out = {}
		for name in ("src_pos", "src_entry", "dst_entry", "flags", "l_seq", "mate_src_entry", "mate_dst_entry", "mate_pos"):
			out[name] = batch[name][lo:hi]
		for off, data in (("cigar_off", "cigar"), ("seq_off", "seq4"), ("qual_off", "qual")):
			o = batch[off][lo:hi + 1]
			out[off] = (o - o[0]).astype(np.uint64)
			out[data] = batch[data][int(o[0]):int(o[-1])]
		return out
	slices, owners = [], []
	for i in range(n_slices):
		pb, own = pinned_batch(cut(i * per, (i + 1) * per))
		slices.append(pb)
		owners.append(own)
	max_cig = max(int(s["cigar"].size) for s in slices)
	max_seq = max(int(s["seq4"].size) for s in slices)
	pipe = Pipeline(ctx, per, max_cig, max_seq, 0, depth)
	outs = [Pipeline.output_arrays(per, 4 * max_cig + 8 * per, per // 4 + 1024) for _ in range(depth)]
	h2d = sum(sum(int(v.nbytes) for k, v in s.items() if k not in ("qual", "qual_off")) for s in slices)
	d2h = 0
	# warm-up: one slice through the pipeline (first-use allocations, workspaces)
	pipe.submit(slices[0], outs[0], params)
	pipe.wait()
	t0 = time.perf_counter()
	done = 0
	for i, sl in enumerate(slices):
		if pipe.in_flight() == depth:
			res, st = pipe.wait()
			d2h += 28 * per + 16 + 4 * int(st["out_cigar_total"]) + 16 * int(st["n_runs"])
			done += 1
		pipe.submit(sl, outs[i % depth], params)
	while pipe.in_flight():
		res, st = pipe.wait()
		d2h += 28 * per + 16 + 4 * int(st["out_cigar_total"]) + 16 * int(st["n_runs"])
		done += 1
	dt = time.perf_counter() - t0
	pipe.close()
	return {
		"value": n_slices * per / dt, "unit": "alignments/s", "records": n_slices * per, "batch_records": per, "depth": depth,
		"GBps_h2d": h2d / dt / 1e9, "GBps_d2h": d2h / dt / 1e9,
		"what": "pvc3_pipeline_submit / _wait over page-locked arrays: upload of batch k + 1, kernels of batch k and download of batch k - 1 overlap; QUAL is fetched by the kernels for realigned records only",
	}


def make_params(gap_open=0, gap_ext=-1, score_mode=SCORE_FP64, tiebreak_mode=0):
	return Params(gap_open, gap_ext, score_mode, tiebreak_mode)


def project_batch(ctx, batch, gap_open=0, gap_ext=-1, score_mode=SCORE_FP64, pinned_out=False, out=None, tiebreak_mode=0):
	"""Host arrays in, host arrays out through pvc3_dbatch_* (upload, kernels, pack, download).  Returns (results, stats).
	With pinned_out the result arrays live in page-locked memory owned by the returned dict's "_owner"."""
	db = DeviceBatch(ctx, batch)
	try:
		db.project(make_params(gap_open, gap_ext, score_mode, tiebreak_mode))
		stats = db.finish()
		res = db.download(pinned=pinned_out, out=out)
		if pinned_out and out is None:
			res["_owner"] = db._pinned_out
		return res, stats
	finally:
		db.close()


def batch_project_call(ctx, batch, params=None, out=None):
	"""pvc3_batch_project, the one call of the C ABI with host arrays in and host arrays out: the context keeps
	its device batch from call to call.  `out`: result arrays of a previous call to use again (their sizes are the capacities);
	otherwise arrays with room for four times the input's CIGAR operations are made, and made larger if the call asks for it.
	Returns (results, stats)."""
	L = _lib.lib()
	p = params if isinstance(params, Params) else make_params(**(params or {}))
	bi, keep = _pack_batch_in(batch)
	n = bi.n
	n_cig_in = int(np.asarray(batch["cigar_off"])[-1]) if n else 0
	if out is None:
		out = dict(dst_pos=np.empty(n, np.uint32), mate_dst_pos=np.empty(n, np.uint32), status=np.empty(n, np.int32), out_cigar_off=np.empty(n + 1, np.uint64), run_off=np.empty(n + 1, np.uint64),
				   out_cigar=np.empty(max(16, 4 * n_cig_in + 16 * n), np.uint32), runs=np.empty(max(16, n // 8 + 1024), RUN_DTYPE))
	for attempt in range(2):
		bo = BatchOut()
		for k in ("dst_pos", "mate_dst_pos", "status", "out_cigar_off", "out_cigar", "run_off", "runs"):
			setattr(bo, k, out[k].ctypes.data)
		bo.out_cigar_cap = out["out_cigar"].size
		bo.run_cap = out["runs"].size
		st = BatchStats()
		rc = L.pvc3_batch_project(ctx._h, C.byref(bi), C.byref(bo), C.byref(p), C.byref(st))
		if -5 == rc and 0 == attempt:		# PVC3_E_CAPACITY: the sizes needed are in the offset arrays' last elements
			out["out_cigar"] = np.empty(int(out["out_cigar_off"][n]) + 16, np.uint32)
			out["runs"] = np.empty(int(out["run_off"][n]) + 16, RUN_DTYPE)
			continue
		check(rc)
		break
	stats = _stats_dict(st)
	res = dict(out)
	res["out_cigar"] = out["out_cigar"][:int(stats["out_cigar_total"])]
	res["runs"] = out["runs"][:int(stats["n_runs"])]
	res["_arrays"] = out
	return res, stats


class AlignmentProjector:
	"""Mirror of panvc3::alignment_projector (include/panvc3/alignment_projector.hh:32-75) over the GPU path.

	The reference takes `msa_index::sequence_entry` objects; here entries are ids in an `Index` that has been uploaded to
	the `Context`, and `ref_seq` is the text attached to the destination entry with Index.set_reference().
	"""

	def __init__(self, ctx):
		self.ctx = ctx
		self.reset()

	def reset(self):
		self._alignment = np.zeros(0, dtype=np.uint32)
		self._ref_ranges = []
		self._query_ranges = []

	def project_alignment(self, src_pos, src_seq_entry, dst_seq_entry, query_seq, cigar_seq, base_qualities=None, gap_opening_cost=0, gap_extension_cost=-1, score_mode=SCORE_FP64):
		"""Returns the projected position; raises ValueError if the record is rejected (the reference would assert)."""
		cig = cigar_encode(cigar_seq) if isinstance(cigar_seq, str) else np.ascontiguousarray(cigar_seq, dtype=np.uint32)
		seq4 = seq4_encode(query_seq)
		batch = dict(
			src_pos=[src_pos], src_entry=[src_seq_entry], dst_entry=[dst_seq_entry],
			flags=[FLAG_HAS_QUAL if base_qualities is not None and len(base_qualities) else 0],
			cigar_off=[0, cig.size], cigar=cig, seq_off=[0, seq4.size], l_seq=[len(query_seq)], seq4=seq4,
		)
		if base_qualities is not None and len(base_qualities):
			batch["qual_off"] = [0, len(base_qualities)]
			batch["qual"] = np.asarray(base_qualities, dtype=np.uint8)
		res, _ = project_batch(self.ctx, batch, gap_opening_cost, gap_extension_cost, score_mode)
		if res["status"][0]:
			raise ValueError("record rejected, status %d" % res["status"][0])
		self._alignment = res["out_cigar"]
		self._ref_ranges = [(int(r["ref_position"]), int(r["ref_length"])) for r in res["runs"]]
		self._query_ranges = [(int(r["query_location"]), int(r["query_length"])) for r in res["runs"]]
		return int(res["dst_pos"][0])

	def alignment(self):
		return self._alignment

	def realigned_reference_ranges(self):
		return self._ref_ranges

	def realigned_query_ranges(self):
		return self._query_ranges
