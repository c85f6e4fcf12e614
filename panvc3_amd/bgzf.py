"""ctypes mirror of include/pvc3_bgzf.h: BGZF blocks compressed on the GPU (panvc3_amd/bgzf/deflate.hip).

No CPU path behind it: without libpvc3_bgzf.so or without a GPU, BlockCompressor() raises."""
import ctypes as C
import os
import struct
import zlib

import numpy as np

BLOCK = 65280
IN_PAD = 64
IN_STRIDE = 65536
OUT_STRIDE = 65536
SYMBOLS = ("pvc3_bgzf_create", "pvc3_bgzf_destroy", "pvc3_bgzf_max_blocks", "pvc3_bgzf_slot_input", "pvc3_bgzf_slot_lengths", "pvc3_bgzf_submit",
		   "pvc3_bgzf_wait", "pvc3_bgzf_deflate", "pvc3_bgzf_last_error")
_L = None


def lib():
	global _L
	if _L is None:
		from . import build
		path = build.BGZF_LIB
		if not os.path.exists(path):
			raise RuntimeError("libpvc3_bgzf.so has not been built (python -m panvc3_amd.build)")
		L = C.CDLL(path)
		L.pvc3_bgzf_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
		L.pvc3_bgzf_destroy.argtypes = [C.c_void_p]
		L.pvc3_bgzf_destroy.restype = None
		L.pvc3_bgzf_max_blocks.argtypes = [C.c_void_p]
		L.pvc3_bgzf_max_blocks.restype = C.c_uint32
		L.pvc3_bgzf_slot_input.argtypes = [C.c_void_p, C.c_uint32]
		L.pvc3_bgzf_slot_input.restype = C.c_void_p
		L.pvc3_bgzf_slot_lengths.argtypes = [C.c_void_p, C.c_uint32]
		L.pvc3_bgzf_slot_lengths.restype = C.c_void_p
		L.pvc3_bgzf_submit.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
		L.pvc3_bgzf_wait.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_float)]
		L.pvc3_bgzf_deflate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]
		L.pvc3_bgzf_last_error.restype = C.c_char_p
		_L = L
	return _L


def _check(rc):
	if rc:
		raise RuntimeError("pvc3_bgzf: %s (%d)" % (lib().pvc3_bgzf_last_error().decode(), rc))


class BlockCompressor:
	def __init__(self, device=0, max_blocks=1024, n_slots=2):
		self._L = lib()
		h = C.c_void_p()
		_check(self._L.pvc3_bgzf_create(device, max_blocks, n_slots, C.byref(h)))
		self._h = h
		self.max_blocks = max_blocks

	def close(self):
		if self._h:
			self._L.pvc3_bgzf_destroy(self._h)
			self._h = None

	def __del__(self):
		self.close()

	def deflate(self, data, piece=BLOCK):
		"""The raw DEFLATE streams of data cut into pieces of `piece` bytes (at least one stream: empty data gives the empty block's)."""
		src = np.frombuffer(bytes(data), dtype=np.uint8)
		n_pieces = max(1, -(-len(src) // piece))
		if 0 == len(src):
			return [bytes([3, 0])]
		dst = np.empty(n_pieces * OUT_STRIDE, dtype=np.uint8)
		lens = np.zeros(n_pieces, dtype=np.uint32)
		_check(self._L.pvc3_bgzf_deflate(self._h, src.ctypes.data, len(src), piece, dst.ctypes.data, lens.ctypes.data))
		return [dst[i * OUT_STRIDE:i * OUT_STRIDE + int(lens[i])].tobytes() for i in range(n_pieces)]

	def timed(self, data, piece=BLOCK, repeats=3):
		"""(kernel ms of the best of `repeats` groups, compressed bytes) for data's first max_blocks pieces, through the slots."""
		src = np.frombuffer(bytes(data), dtype=np.uint8)
		n_blocks = min(self.max_blocks, -(-len(src) // piece))
		base = self._L.pvc3_bgzf_slot_input(self._h, 0)
		inp = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_uint8)), shape=(self.max_blocks * IN_STRIDE,))
		lens = np.ctypeslib.as_array(C.cast(self._L.pvc3_bgzf_slot_lengths(self._h, 0), C.POINTER(C.c_uint32)), shape=(self.max_blocks,))
		for i in range(n_blocks):
			chunk = src[i * piece:(i + 1) * piece]
			inp[i * IN_STRIDE + IN_PAD:i * IN_STRIDE + IN_PAD + len(chunk)] = chunk
			lens[i] = len(chunk)
		best, size = None, 0
		for _ in range(repeats):
			_check(self._L.pvc3_bgzf_submit(self._h, 0, n_blocks))
			out, out_len, ms = C.c_void_p(), C.c_void_p(), C.c_float()
			_check(self._L.pvc3_bgzf_wait(self._h, 0, C.byref(out), C.byref(out_len), C.byref(ms)))
			ol = np.ctypeslib.as_array(C.cast(out_len, C.POINTER(C.c_uint32)), shape=(n_blocks,))
			size = int(ol.sum())
			best = ms.value if best is None else min(best, ms.value)
		return best, size, int(lens[:n_blocks].sum())


def bgzf_member(stream, data):
	"""The BGZF block (gzip member with the BC field, SAMv1 §4.1.1) around a raw DEFLATE stream of `data`."""
	bsize = 18 + len(stream) + 8
	return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1) + stream
			+ struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
