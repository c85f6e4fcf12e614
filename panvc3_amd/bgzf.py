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
		   "pvc3_bgzf_wait", "pvc3_bgzf_deflate", "pvc3_bgzf_last_error",
		   "pvc3_bgzf_inflater_create", "pvc3_bgzf_inflater_destroy", "pvc3_bgzf_inflater_blocks_in_flight", "pvc3_bgzf_inflater_limit_in_flight", "pvc3_bgzf_inflater_input", "pvc3_bgzf_inflater_output", "pvc3_bgzf_inflater_blocks",
		   "pvc3_bgzf_inflater_submit", "pvc3_bgzf_inflater_wait", "pvc3_bgzf_inflater_submit_device", "pvc3_bgzf_inflater_wait_device",
		   "pvc3_bgzf_dev_create", "pvc3_bgzf_dev_destroy", "pvc3_bgzf_dev_input", "pvc3_bgzf_dev_capacity", "pvc3_bgzf_dev_submit", "pvc3_bgzf_dev_wait", "pvc3_bgzf_dev_fetch")
BLOCK_DTYPE = np.dtype([("in_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4"), ("out_off", "<u8")])
_L = None


def lib():
	global _L
	if _L is None:
		from . import build
		path = os.environ.get("PVC3_BGZF_LIB", build.BGZF_LIB)		# (an experimental build of the kernels: tools/calib)
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
		L.pvc3_bgzf_inflater_create.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
		L.pvc3_bgzf_inflater_destroy.argtypes = [C.c_void_p]
		L.pvc3_bgzf_inflater_destroy.restype = None
		for name in ("pvc3_bgzf_inflater_input", "pvc3_bgzf_inflater_output", "pvc3_bgzf_inflater_blocks"):
			getattr(L, name).argtypes = [C.c_void_p, C.c_uint32]
			getattr(L, name).restype = C.c_void_p
		L.pvc3_bgzf_inflater_submit.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t]
		L.pvc3_bgzf_inflater_wait.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_float)]
		L.pvc3_bgzf_inflater_limit_in_flight.argtypes = [C.c_void_p, C.c_uint32]
		L.pvc3_bgzf_inflater_blocks_in_flight.argtypes = [C.c_void_p]
		L.pvc3_bgzf_inflater_blocks_in_flight.restype = C.c_uint32
		L.pvc3_bgzf_inflater_submit_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p, C.c_size_t]
		L.pvc3_bgzf_inflater_wait_device.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_float)]
		L.pvc3_bgzf_dev_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
		L.pvc3_bgzf_dev_destroy.argtypes = [C.c_void_p]
		L.pvc3_bgzf_dev_destroy.restype = None
		L.pvc3_bgzf_dev_input.argtypes = [C.c_void_p, C.c_uint32]
		L.pvc3_bgzf_dev_input.restype = C.c_void_p
		L.pvc3_bgzf_dev_capacity.argtypes = [C.c_void_p]
		L.pvc3_bgzf_dev_capacity.restype = C.c_uint64
		L.pvc3_bgzf_dev_submit.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
		L.pvc3_bgzf_dev_wait.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
		L.pvc3_bgzf_dev_fetch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
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


class BlockInflater:
	"""Raw DEFLATE streams inflated on the GPU (pvc3_bgzf_inflater_*)."""

	def __init__(self, device=0, max_in=8 << 20, max_out=32 << 20, max_blocks=4096, n_slots=2):
		self._L = lib()
		h = C.c_void_p()
		_check(self._L.pvc3_bgzf_inflater_create(device, max_in, max_out, max_blocks, n_slots, C.byref(h)))
		self._h = h
		self.max_in, self.max_out, self.max_blocks, self.n_slots = max_in, max_out, max_blocks, n_slots

	def close(self):
		if self._h:
			self._L.pvc3_bgzf_inflater_destroy(self._h)
			self._h = None

	def __del__(self):
		self.close()

	def _views(self, slot):
		inp = np.ctypeslib.as_array(C.cast(self._L.pvc3_bgzf_inflater_input(self._h, slot), C.POINTER(C.c_uint8)), shape=(self.max_in,))
		out = np.ctypeslib.as_array(C.cast(self._L.pvc3_bgzf_inflater_output(self._h, slot), C.POINTER(C.c_uint8)), shape=(self.max_out,))
		raw = np.ctypeslib.as_array(C.cast(self._L.pvc3_bgzf_inflater_blocks(self._h, slot), C.POINTER(C.c_uint8)), shape=(self.max_blocks * BLOCK_DTYPE.itemsize,))
		return inp, out, raw.view(BLOCK_DTYPE)

	def inflate(self, streams, sizes, gap=0, slot=0, repeats=1):
		"""streams[i] inflated to sizes[i] bytes; the streams lie `gap` bytes apart in the input (any alignment), the outputs one
		behind the other.  Returns (list of bytes or None per stream, statuses, kernel ms)."""
		inp, out, blocks = self._views(slot)
		pos, opos = 1, 0		# the first stream at an odd address
		for i, (s, n) in enumerate(zip(streams, sizes)):
			inp[pos:pos + len(s)] = np.frombuffer(s, dtype=np.uint8)
			blocks[i] = (pos, len(s), n, opos)
			pos += len(s) + gap
			opos += n
		out[:opos] = 0xAB
		best = None
		for _ in range(repeats):
			_check(self._L.pvc3_bgzf_inflater_submit(self._h, slot, len(streams), pos, opos))
			st, ms = C.c_void_p(), C.c_float()
			_check(self._L.pvc3_bgzf_inflater_wait(self._h, slot, C.byref(st), C.byref(ms)))
			best = ms.value if best is None else min(best, ms.value)
		status = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_uint32)), shape=(len(streams),)).copy()
		got, o = [], 0
		for i, n in enumerate(sizes):
			got.append(out[o:o + n].tobytes() if 0 == status[i] else None)
			o += n
		return got, status, best

	def inflate_slots(self, groups):
		"""groups[k] = (streams, sizes) for slot k: every slot is submitted before any is waited for (each has a stream of its own, so
		their kernels are on the GPU together).  Returns a list of (list of bytes or None, statuses) per slot."""
		laid = []
		for slot, (streams, sizes) in enumerate(groups):
			inp, out, blocks = self._views(slot)
			pos, opos = 2, 0
			for i, (s, n) in enumerate(zip(streams, sizes)):
				inp[pos:pos + len(s)] = np.frombuffer(s, dtype=np.uint8)
				blocks[i] = (pos, len(s), n, opos)
				pos += len(s) + 1
				opos += n
			out[:opos] = 0xCD
			laid.append((out, pos, opos))
		for slot, (streams, _) in enumerate(groups):
			_check(self._L.pvc3_bgzf_inflater_submit(self._h, slot, len(streams), laid[slot][1], laid[slot][2]))
		results = []
		for slot, (streams, sizes) in enumerate(groups):
			st, ms = C.c_void_p(), C.c_float()
			_check(self._L.pvc3_bgzf_inflater_wait(self._h, slot, C.byref(st), C.byref(ms)))
			status = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_uint32)), shape=(len(streams),)).copy()
			got, o = [], 0
			for i, n in enumerate(sizes):
				got.append(laid[slot][0][o:o + n].tobytes() if 0 == status[i] else None)
				o += n
			results.append((got, status))
		return results

	def inflate_to_device(self, streams, sizes, slot=0):
		"""pvc3_bgzf_inflater_submit_device: the streams inflated into device memory (allocated here), one behind the other; returns
		(the bytes copied back, statuses, the CRC-32 of every block as the device computed them)."""
		inp, out, blocks = self._views(slot)
		pos, opos = 3, 0
		for i, (s, n) in enumerate(zip(streams, sizes)):
			inp[pos:pos + len(s)] = np.frombuffer(s, dtype=np.uint8)
			blocks[i] = (pos, len(s), n, opos)
			pos += len(s) + 5
			opos += n
		H = hip_runtime()
		d = C.c_void_p()
		assert 0 == H.hipMalloc(C.byref(d), opos + 4096)
		try:
			assert 0 == H.hipMemset(d, 0xAB, opos + 4096)
			_check(self._L.pvc3_bgzf_inflater_submit_device(self._h, slot, len(streams), pos, d, opos))
			st, crc, ms = C.c_void_p(), C.c_void_p(), C.c_float()
			_check(self._L.pvc3_bgzf_inflater_wait_device(self._h, slot, C.byref(st), C.byref(crc), C.byref(ms)))
			status = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_uint32)), shape=(len(streams),)).copy()
			crcs = np.ctypeslib.as_array(C.cast(crc, C.POINTER(C.c_uint32)), shape=(len(streams),)).copy()
			host = (C.c_uint8 * max(1, opos))()
			assert 0 == H.hipMemcpy(host, d, opos, D2H)
			return bytes(host)[:opos], status, crcs
		finally:
			H.hipFree(d)


_HIP = None


def hip_runtime():
	"""The HIP runtime the library is linked with, for the few raw calls the device-resident forms need in tests and tools
	(hipMalloc / hipMemcpy / hipFree of buffers that kernels of the caller would write in the product)."""
	global _HIP
	if _HIP is None:
		lib()
		H = C.CDLL("libamdhip64.so")
		H.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
		H.hipFree.argtypes = [C.c_void_p]
		H.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
		H.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
		_HIP = H
	return _HIP


H2D, D2H = 1, 2


class DeviceCompressor:
	"""pvc3_bgzf_dev_*: the compressor whose input lies in device memory (here filled with hipMemcpy; in the product by the record
	path's kernels).  deflate() returns the bytes of the BGZF file: blocks with header, stream, CRC-32 and ISIZE."""

	def __init__(self, device=0, max_blocks=256, n_slots=2):
		self._L = lib()
		h = C.c_void_p()
		_check(self._L.pvc3_bgzf_dev_create(device, max_blocks, n_slots, C.byref(h)))
		self._h = h
		self.max_blocks = max_blocks

	def close(self):
		if self._h:
			self._L.pvc3_bgzf_dev_destroy(self._h)
			self._h = None

	def __del__(self):
		self.close()

	def deflate(self, data, slot=0):
		data = bytes(data)
		assert 0 < len(data) <= self._L.pvc3_bgzf_dev_capacity(self._h)
		d_in = self._L.pvc3_bgzf_dev_input(self._h, slot)
		H = hip_runtime()
		for b in range(0, -(-len(data) // BLOCK)):
			chunk = data[b * BLOCK:(b + 1) * BLOCK]
			assert 0 == H.hipMemcpy(d_in + b * IN_STRIDE + IN_PAD, chunk, len(chunk), H2D)
		_check(self._L.pvc3_bgzf_dev_submit(self._h, slot, len(data)))
		total, ms = C.c_uint64(), C.c_float()
		_check(self._L.pvc3_bgzf_dev_wait(self._h, slot, C.byref(total), C.byref(ms)))
		blob, off = b"", 0
		while off < total.value:		# (BAM records come back in one piece; incompressible bytes need two)
			out, n = C.c_void_p(), C.c_uint64()
			_check(self._L.pvc3_bgzf_dev_fetch(self._h, slot, off, C.byref(out), C.byref(n)))
			blob += C.string_at(out, n.value)
			off += n.value
		return blob, ms.value


def bgzf_member(stream, data):
	"""The BGZF block (gzip member with the BC field, SAMv1 §4.1.1) around a raw DEFLATE stream of `data`."""
	bsize = 18 + len(stream) + 8
	return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1) + stream
			+ struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
