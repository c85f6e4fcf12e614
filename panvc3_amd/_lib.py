"""Loads libpvc3_hip.so (the HIP kernels + C ABI).  There is no fallback: if the library is missing, loading fails."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PVC3_LIB_PATH") or os.path.join(HERE, "libpvc3_hip.so")		# the override is for diagnostic builds (tools/)

_lib = None


class Error(RuntimeError):
	def __init__(self, code, message):
		super().__init__("pvc3 error %d: %s" % (code, message))
		self.code = code


class Run(C.Structure):
	_fields_ = [("query_location", C.c_uint32), ("query_length", C.c_uint32), ("ref_position", C.c_uint32), ("ref_length", C.c_uint32)]


class Params(C.Structure):
	_fields_ = [("gap_open", C.c_int32), ("gap_ext", C.c_int32), ("score_mode", C.c_int32), ("tiebreak_mode", C.c_int32)]


class SplitIn(C.Structure):
	_fields_ = [("n", C.c_uint64), ("sequence", C.c_void_p), ("pos", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("seq_off", C.c_void_p),
				("seq4", C.c_void_p), ("seq4_bytes", C.c_uint64), ("l_seq", C.c_void_p)]


class SplitOut(C.Structure):
	_fields_ = [("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("cigar_cap", C.c_uint64), ("n_map_off", C.c_void_p), ("n_map", C.c_void_p), ("n_map_cap", C.c_uint64),
				("status", C.c_void_p)]


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


class BatchStats(C.Structure):
	_fields_ = [
		("n_records", C.c_uint64), ("n_failed", C.c_uint64), ("n_realigned_records", C.c_uint64), ("n_runs", C.c_uint64),
		("out_cigar_total", C.c_uint64), ("algorithmic_bytes", C.c_uint64), ("algorithmic_bytes_project", C.c_uint64),
		("n_fast_path", C.c_uint64), ("n_mid_path", C.c_uint64), ("ms_project", C.c_float), ("ms_mid", C.c_float), ("ms_general", C.c_float),
		("ms_realign", C.c_float), ("ms_pack", C.c_float), ("attempts", C.c_uint32),
	]


class RescoreParams(C.Structure):
	_fields_ = [("min_mismatch_penalty", C.c_int32), ("max_mismatch_penalty", C.c_int32), ("n_penalty", C.c_int32), ("gap_opening_penalty", C.c_int32), ("gap_extension_penalty", C.c_int32)]


class MapqBatch(C.Structure):
	_fields_ = [("n_groups", C.c_uint64), ("group_off", C.c_void_p)] + [(k, C.c_void_p) for k in (
		"ref_id", "pos", "mate_ref_id", "mate_pos", "orig_ref_id", "orig_pos", "orig_mate_ref_id", "orig_mate_pos", "score", "l_seq", "seq_class")]


class MapqStats(C.Structure):
	_fields_ = [(k, C.c_uint64) for k in ("total_alignments", "unpaired_alignments", "reads_with_and_without_mate", "mate_not_found", "reads_without_valid_position")]


class PipelineCaps(C.Structure):
	_fields_ = [("max_records", C.c_uint64), ("max_cigar_ops", C.c_uint64), ("max_seq_bytes", C.c_uint64), ("max_qual_bytes", C.c_uint64), ("depth", C.c_uint32)]


def lib():
	"""The loaded library (raises OSError if it has not been built: there is no CPU fallback)."""
	global _lib
	if _lib is not None:
		return _lib
	if not os.path.exists(LIB_PATH):
		raise OSError("%s is missing: build it with `python -m panvc3_amd.build` (hipcc, gfx950)" % LIB_PATH)
	# PyTorch-ROCm wheels bundle their own libamdhip64.so.  Loading torch first makes our library bind to that same
	# runtime (same SONAME), so a process never ends up with two HIP runtimes.
	try:
		import torch  # noqa: F401
	except ImportError:
		pass
	L = C.CDLL(LIB_PATH)
	vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
	P = C.POINTER
	sig = {
		"pvc3_version": ([], i32),
		"pvc3_strerror": ([i32], C.c_char_p),
		"pvc3_last_error": ([], C.c_char_p),
		"pvc3_device_count": ([P(i32)], i32),
		"pvc3_index_create": ([P(vp)], i32),
		"pvc3_index_destroy": ([vp], None),
		"pvc3_index_add_aligned": ([vp, C.c_char_p, u64, P(u32)], i32),
		"pvc3_index_add_bits": ([vp, vp, u64, P(u32)], i32),
		"pvc3_index_set_reference": ([vp, u32, vp, u64], i32),
		"pvc3_index_set_destination": ([vp, u32, u32], i32),
		"pvc3_index_entry_count": ([vp, P(u32)], i32),
		"pvc3_index_entry_info": ([vp, u32, P(u64), P(u64)], i32),
		"pvc3_index_entry_layout": ([vp, u32, P(vp), P(u64), P(vp), P(u64)], i32),
		"pvc3_context_create": ([i32, P(vp)], i32),
		"pvc3_context_destroy": ([vp], None),
		"pvc3_context_set_stream": ([vp, vp], i32),
		"pvc3_context_set_timing": ([vp, i32], i32),
		"pvc3_context_set_option": ([vp, C.c_char_p, C.c_int64], i32),
		"pvc3_context_synchronize": ([vp], i32),
		"pvc3_context_upload_index": ([vp, vp], i32),
		"pvc3_context_hbm_bytes": ([vp, P(u64), P(u64)], i32),
		"pvc3_batch_project": ([vp, P(BatchIn), P(BatchOut), P(Params), P(BatchStats)], i32),
		"pvc3_dbatch_create": ([vp, P(BatchIn), P(vp)], i32),
		"pvc3_dbatch_destroy": ([vp], None),
		"pvc3_dbatch_project": ([vp, vp, P(Params)], i32),
		"pvc3_dbatch_finish": ([vp, vp, P(BatchStats)], i32),
		"pvc3_dbatch_download": ([vp, vp, P(BatchOut)], i32),
		"pvc3_dbatch_checksum": ([vp, vp, u64, P(u64)], i32),
		"pvc3_project_positions": ([vp, u64, vp, vp, vp, vp], i32),
		"pvc3_reference_create": ([vp, u32, vp, vp, P(vp)], i32),
		"pvc3_reference_destroy": ([vp], None),
		"pvc3_split_match_ops": ([vp, vp, P(SplitIn), P(SplitOut), P(u64), P(u64)], i32),
		"pvc3_rescore_alignments": ([vp, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, P(RescoreParams), vp, vp], i32),
		"pvc3_recalculate_mapq": ([vp, P(MapqBatch), vp, vp, P(MapqStats)], i32),
		"pvc3_pipeline_create": ([vp, P(PipelineCaps), P(vp)], i32),
		"pvc3_pipeline_destroy": ([vp], None),
		"pvc3_pipeline_submit": ([vp, P(BatchIn), P(BatchOut), P(Params)], i32),
		"pvc3_pipeline_wait": ([vp, P(BatchStats)], i32),
		"pvc3_pipeline_in_flight": ([vp, P(u32)], i32),
		"pvc3_pipeline_synchronize": ([vp], i32),
		"pvc3_host_alloc": ([u64, P(vp)], i32),
		"pvc3_host_free": ([vp], None),
		# include/pvc3_records.h (the record path on the GPU; its caller is the C++ command line, the binding is for tests)
		"pvc3_records_create": ([vp, vp, P(vp)], i32),
		"pvc3_records_destroy": ([vp], None),
		"pvc3_records_buffer": ([vp, u32, P(vp), P(u64)], i32),
		"pvc3_records_upload": ([vp, u32, u64, vp, u64], i32),
		"pvc3_records_frame": ([vp, u32, u64, vp, u32, u64, i32, vp, P(i32), P(u64)], i32),
		"pvc3_records_project": ([vp, P(Params), P(BatchStats)], i32),
		"pvc3_records_rewrite": ([vp, vp, u64, u32, u32, u32, P(u64), P(u64), P(i32), P(u64)], i32),
		"pvc3_records_rewrite_to_host": ([vp, vp, u64, P(u64), P(u64), P(i32), P(u64)], i32),
		"pvc3_records_runs": ([vp, vp, u64], i32),
		"pvc3_records_removed_tags": ([vp, vp], i32),
		"pvc3_records_batch_sizes": ([vp, P(u64), P(u64)], i32),
		"pvc3_records_download_batch": ([vp] + [vp] * 14, i32),
	}
	for name, (argtypes, restype) in sig.items():
		f = getattr(L, name)
		f.argtypes = argtypes
		f.restype = restype
	L._pvc3_signatures = sig
	_lib = L
	return L


def check(rc):
	if rc != 0:
		raise Error(rc, lib().pvc3_last_error().decode(errors="replace"))
