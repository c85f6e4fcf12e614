"""ctypes binding of the synthetic workload generator (tests/synth/synth.cpp).  Test / benchmark infrastructure.

`config(name)` returns the (genome, reads) parameters of BASELINE.json's configurations; `scale` shrinks
genome and read count together for tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpvc3_synth.so")


def _build_key():
	from panvc3_amd.build import _key
	cmd = ["make", "-C", HERE, "-s", "-B"]
	return _key([os.path.join(HERE, "synth.cpp"), os.path.join(HERE, "Makefile")], cmd), cmd


def build(force=False):
	"""Keyed by the content of the sources (file times do not survive the snapshot that carries the tree to a GPU box)."""
	from panvc3_amd.build import _done, _stale
	key, cmd = _build_key()
	if force or _stale(LIB_PATH, key):
		subprocess.check_call(cmd)
		_done(LIB_PATH, key)
	return LIB_PATH


def is_stale():
	from panvc3_amd.build import _stale
	return _stale(LIB_PATH, _build_key()[0])


class GenomeParams(C.Structure):
	_fields_ = [
		("seed", C.c_uint64), ("genome_len", C.c_uint64), ("n_contigs", C.c_uint32), ("n_founders", C.c_uint32),
		("snv_rate", C.c_double), ("indel_rate", C.c_double), ("complex_rate", C.c_double), ("complex_max", C.c_uint32),
		("long_rate", C.c_double), ("long_min", C.c_uint32), ("long_max", C.c_uint32), ("carrier_prob", C.c_double),
		("lowercase_every", C.c_uint32), ("indel_mean_len", C.c_double), ("indel_max_len", C.c_uint32),
	]


class ReadsParams(C.Structure):
	_fields_ = [
		("seed", C.c_uint64), ("n_records", C.c_uint64), ("read_len", C.c_uint32), ("paired", C.c_uint32),
		("insert_mean", C.c_double), ("insert_sd", C.c_double), ("sub_rate", C.c_double), ("indel_read_frac", C.c_double),
		("clip_read_frac", C.c_double), ("with_qual", C.c_uint32), ("n_base_rate", C.c_double),
	]


_lib = None


def lib():
	global _lib
	if _lib is None:
		build()
		L = C.CDLL(LIB_PATH)
		L.synth_genome_create.argtypes = [C.POINTER(GenomeParams), C.c_int]
		L.synth_genome_create.restype = C.c_void_p
		L.synth_genome_from_aligned.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.c_uint64]
		L.synth_genome_from_aligned.restype = C.c_void_p
		L.synth_genome_free.argtypes = [C.c_void_p]
		L.synth_genome_save.argtypes = [C.c_void_p, C.c_char_p]
		L.synth_genome_save.restype = C.c_int
		L.synth_genome_map.argtypes = [C.c_char_p]
		L.synth_genome_map.restype = C.c_void_p
		for f in (L.synth_genome_entry_count, L.synth_genome_founders, L.synth_genome_contigs):
			f.argtypes = [C.c_void_p]
			f.restype = C.c_uint32
		L.synth_genome_entry.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
		L.synth_reads_create.argtypes = [C.c_void_p, C.POINTER(ReadsParams), C.c_int]
		L.synth_reads_create.restype = C.c_void_p
		L.synth_reads_free.argtypes = [C.c_void_p]
		L.synth_reads_count.argtypes = [C.c_void_p]
		L.synth_reads_count.restype = C.c_uint64
		L.synth_reads_array.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
		L.synth_reads_array.restype = C.c_void_p
		_lib = L
	return _lib


def default_threads():
	return max(1, min(32, os.cpu_count() or 1))


class Genome:
	"""Random founder MSA: entry e = contig * (1 + F) + row, row 0 = REF (destination), rows 1..F = founders."""

	def __init__(self, params=None, n_threads=None, aligned_rows=None, mapped_from=None):
		self._L = lib()
		if mapped_from is not None:
			h = self._L.synth_genome_map(os.fsencode(mapped_from))
			if not h:
				raise OSError("%s is not a genome saved by Genome.save()" % mapped_from)
			self._h = C.c_void_p(h)
		elif aligned_rows is not None:
			rows = [r if isinstance(r, bytes) else r.encode() for r in aligned_rows]
			arr = (C.c_char_p * len(rows))(*rows)
			self._h = C.c_void_p(self._L.synth_genome_from_aligned(arr, len(rows), len(rows[0])))
		else:
			self._h = C.c_void_p(self._L.synth_genome_create(C.byref(params), n_threads or default_threads()))
		self.n_entries = self._L.synth_genome_entry_count(self._h)
		self.n_founders = self._L.synth_genome_founders(self._h)
		self.n_contigs = self._L.synth_genome_contigs(self._h)

	def __del__(self):
		if getattr(self, "_h", None):
			self._L.synth_genome_free(self._h)
			self._h = None

	def save(self, path):
		"""The genome as one file that Genome(mapped_from=path) maps (shared pages: several processes, one copy)."""
		if self._L.synth_genome_save(self._h, os.fsencode(path)):
			raise OSError("unable to write %s" % path)

	def entry(self, e):
		"""(gap_words uint64 view, n_cols, text uint8 view) — views into the generator's memory."""
		wp, tp = C.c_void_p(), C.c_void_p()
		nc, tl = C.c_uint64(), C.c_uint64()
		self._L.synth_genome_entry(self._h, e, C.byref(wp), C.byref(nc), C.byref(tp), C.byref(tl))
		words = np.ctypeslib.as_array(C.cast(wp, C.POINTER(C.c_uint64)), shape=((nc.value + 63) // 64,))
		text = np.ctypeslib.as_array(C.cast(tp, C.POINTER(C.c_uint8)), shape=(tl.value,)) if tl.value else np.zeros(0, dtype=np.uint8)
		return words, nc.value, text

	def is_destination(self, e):
		return e % (1 + self.n_founders) == 0

	def fill_index(self, index, with_all_texts=False):
		"""Adds every entry to an index object that has add_bits / set_reference (the product's or the oracle's)."""
		for e in range(self.n_entries):
			words, n_cols, text = self.entry(e)
			got = index.add_bits(words, n_cols)
			assert got == e
			if self.is_destination(e) or with_all_texts:
				index.set_reference(e, text)


_ARRAYS = [
	("src_pos", np.uint32), ("src_entry", np.uint32), ("dst_entry", np.uint32), ("flags", np.uint32), ("cigar_off", np.uint64),
	("cigar", np.uint32), ("seq_off", np.uint64), ("l_seq", np.uint32), ("seq4", np.uint8), ("qual_off", np.uint64), ("qual", np.uint8),
	("mate_src_entry", np.uint32), ("mate_dst_entry", np.uint32), ("mate_pos", np.uint32),
]


class Reads:
	def __init__(self, genome, params, n_threads=None):
		self._L = lib()
		self._genome = genome
		self._h = C.c_void_p(self._L.synth_reads_create(genome._h, C.byref(params), n_threads or default_threads()))
		self.n = self._L.synth_reads_count(self._h)

	def __del__(self):
		if getattr(self, "_h", None):
			self._L.synth_reads_free(self._h)
			self._h = None

	def batch(self, copy=False):
		"""The SoA batch of include/pvc3.h as numpy views (valid while this object lives) or copies."""
		out = {}
		for i, (name, dtype) in enumerate(_ARRAYS):
			cnt = C.c_uint64()
			p = self._L.synth_reads_array(self._h, i, C.byref(cnt))
			if cnt.value == 0:
				a = np.zeros(0, dtype=dtype)
			else:
				a = np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(cnt.value,))
			out[name] = a.copy() if copy else a
		out["_owner"] = self
		return out


def slice_batch(batch, lo, hi):
	"""Records [lo, hi) of a batch as a self-contained batch (offsets rebased)."""
	out = {}
	for name in ("src_pos", "src_entry", "dst_entry", "flags", "l_seq", "mate_src_entry", "mate_dst_entry", "mate_pos"):
		out[name] = batch[name][lo:hi]
	for off, data in (("cigar_off", "cigar"), ("seq_off", "seq4"), ("qual_off", "qual")):
		o = batch[off][lo:hi + 1]
		out[off] = (o - o[0]).astype(np.uint64)
		out[data] = batch[data][int(o[0]):int(o[-1])]
	return out


def reorder_batch(batch, order, chunk=2_000_000):
	"""The batch with its records permuted by `order` (variable-length fields are re-packed, a few million records at a time so
	that the index arrays of the gather stay small next to the data)."""
	order = np.asarray(order, dtype=np.int64)
	out = {}
	for name in ("src_pos", "src_entry", "dst_entry", "flags", "l_seq", "mate_src_entry", "mate_dst_entry", "mate_pos"):
		out[name] = batch[name][order]
	n = order.size
	for off, data in (("cigar_off", "cigar"), ("seq_off", "seq4"), ("qual_off", "qual")):
		o = batch[off].astype(np.int64)
		lens = (o[1:] - o[:-1])[order]
		new_off = np.concatenate([[0], np.cumsum(lens)])
		starts = o[:-1][order]
		src = batch[data]
		dst = np.empty(int(new_off[-1]), dtype=src.dtype)
		for a in range(0, n, chunk):
			b = min(n, a + chunk)
			# for every output element of the chunk its source index
			idx = np.repeat(starts[a:b] - new_off[a:b], lens[a:b]) + np.arange(new_off[a], new_off[b])
			dst[new_off[a]:new_off[b]] = src[idx]
		out[off] = new_off.astype(np.uint64)
		out[data] = dst
	return out


def config(name, scale=1.0, seed=None):
	"""BASELINE.json configurations.  Returns (GenomeParams, ReadsParams, description)."""
	name = name.upper()
	base = dict(snv_rate=1e-3, indel_rate=1e-4, complex_rate=5e-5, complex_max=15, long_rate=0.0, long_min=20, long_max=60,
				carrier_prob=0.5, lowercase_every=0, indel_mean_len=0.0, indel_max_len=0)
	reads = dict(read_len=150, paired=1, insert_mean=400.0, insert_sd=50.0, sub_rate=0.005, indel_read_frac=0.02, clip_read_frac=0.03,
				 with_qual=1, n_base_rate=0.0)
	if name == "C2":
		g = dict(genome_len=100_000_000, n_contigs=1, n_founders=2); r = dict(n_records=1_000_000, paired=0); num = 2
		desc = "C2: 1 M synthetic 150 bp single-end reads, REF + 2 founders"
	elif name == "C3":
		g = dict(genome_len=1_000_000_000, n_contigs=8, n_founders=8); r = dict(n_records=40_000_000); num = 3
		desc = "C3: 20 M 2x150 bp pairs (40 M records), REF + 8 founders, 1 Gbp in 8 contigs"
	elif name == "C3G":
		# C3 on an MSA with the gap statistics of the reference's shipped test MSA (test-workflow/index-input: 21 - 26 % gap columns
		# per row, gap runs of mean length 16, longest 63): an insertion or deletion of mean length 16 every 22 bases
		g = dict(genome_len=1_000_000_000, n_contigs=8, n_founders=8, indel_rate=0.045, indel_mean_len=15.0, indel_max_len=62); r = dict(n_records=40_000_000); num = 3
		desc = "C3G: 20 M 2x150 bp pairs (40 M records), REF + 8 founders, 1 Gbp in 8 contigs, gap-dense MSA (a gap run of mean length 16 every 22 bases)"
	elif name == "C4":
		g = dict(genome_len=1_000_000_000, n_contigs=8, n_founders=8); r = dict(n_records=400_000_000); num = 4
		desc = "C4: 200 M 2x150 bp pairs sharded over 8 GPUs"
	elif name == "C5":
		# long events tuned so that about 5 % of 250 bp reads overlap one: rate * (250 + ~40) * carrier 0.5 ~ 0.05
		g = dict(genome_len=1_000_000_000, n_contigs=8, n_founders=8, long_rate=3.4e-4); r = dict(n_records=40_000_000, read_len=250); num = 5
		desc = "C5: 20 M 2x250 bp pairs, 5 % of reads over a long (20-60 bp) indel event"
	else:
		raise ValueError(name)
	base.update(g); reads.update(r)
	base["genome_len"] = max(20_000, int(base["genome_len"] * scale))
	reads["n_records"] = max(2, int(reads["n_records"] * scale))
	s = (42 + num) if seed is None else seed
	return GenomeParams(seed=s, **base), ReadsParams(seed=s, **reads), desc
