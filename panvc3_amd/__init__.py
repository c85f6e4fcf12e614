"""panvc3_amd — MI355X-native project_alignments hot path of PanVC 3 (hand-written HIP for gfx950 behind a C ABI).

This Python layer is only a binding of include/pvc3.h (ctypes) plus a mirror of the reference's
panvc3::alignment_projector interface for tests and benchmarks; all computation happens in libpvc3_hip.so on the GPU.
"""
from .api import (AlignmentProjector, Context, DeviceBatch, Index, Params, SCORE_FP64, SCORE_INT32_TRUNC, FLAG_HAS_MATE,
				  FLAG_HAS_QUAL, NO_MATE, RUN_DTYPE, PinnedArray, Pipeline, cigar_decode, cigar_encode, device_count, pinned_batch, pipeline_bench, project_batch, recalculate_mapq, rescore_alignments, seq4_encode)
from ._lib import Error

__all__ = [
	"AlignmentProjector", "Context", "DeviceBatch", "Index", "Params", "Error", "SCORE_FP64", "SCORE_INT32_TRUNC",
	"FLAG_HAS_MATE", "FLAG_HAS_QUAL", "NO_MATE", "RUN_DTYPE", "cigar_decode", "cigar_encode", "device_count",
	"project_batch", "seq4_encode", "PinnedArray", "pinned_batch", "Pipeline", "pipeline_bench", "recalculate_mapq", "rescore_alignments",
]
