"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/pvc3.h declares, the host-side
index builder produces the documented HBM layout, and the library refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import ROOT, load_test_msa

import panvc3_amd
from panvc3_amd import _lib


def declared_symbols():
	text = open(os.path.join(ROOT, "include", "pvc3.h")).read() + open(os.path.join(ROOT, "include", "pvc3_records.h")).read()
	return sorted(set(re.findall(r"PVC3_API\s+[^;(]*?\b(pvc3_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
	from panvc3_amd import build
	build.build()
	L = ctypes.CDLL(_lib.LIB_PATH)
	names = declared_symbols()
	assert len(names) >= 25
	for name in names:
		assert hasattr(L, name), name
	# and the ctypes binding covers all of them
	assert set(names) == set(_lib.lib()._pvc3_signatures)
	assert _lib.lib().pvc3_version() == 100


def test_no_cpu_fallback():
	if panvc3_amd.device_count() > 0:
		pytest.skip("a GPU is visible")
	with pytest.raises(panvc3_amd.Error) as ei:
		panvc3_amd.Context(0)
	assert ei.value.code == -4		# PVC3_E_NO_DEVICE
	assert "no CPU path" in str(ei.value)


def _check_layout(idx, e, aligned):
	n = len(aligned)
	gaps = np.frombuffer(aligned if isinstance(aligned, bytes) else aligned.encode(), dtype=np.uint8) == ord("-")
	n_cols, n_zeros = idx.entry_info(e)
	assert n_cols == n and n_zeros == int((~gaps).sum())
	blocks, sel = idx.entry_layout(e)
	n_blocks = (n + 447) // 448
	assert blocks.shape == (n_blocks + 2, 8)
	# unpack the 448 data bits of every block
	bits = np.unpackbits(blocks[:, 1:].copy().view(np.uint8), bitorder="little").reshape(n_blocks + 2, 448).reshape(-1)
	assert np.array_equal(bits[:n].astype(bool), gaps)
	assert bits[n:].all()		# padding columns and blocks are gaps
	per_block_zeros = 448 - bits.reshape(-1, 448).sum(axis=1)
	zeros_before = np.concatenate([[0], np.cumsum(per_block_zeros)[:-1]])
	assert np.array_equal(blocks[:, 0] & np.uint64(0xFFFFFFFF), zeros_before.astype(np.uint64))
	# the in-block directory: zeros in data words 0, 0..2 and 0..4
	word_zeros = 64 - bits.reshape(-1, 7, 64).sum(axis=2)
	cum = np.cumsum(word_zeros, axis=1)
	for k in range(3):
		assert np.array_equal((blocks[:, 0] >> np.uint64(32 + 9 * k)) & np.uint64(0x1FF), cum[:, 2 * k].astype(np.uint64))
	assert not (blocks[:, 0] >> np.uint64(59)).any()
	zero_cols = np.flatnonzero(~gaps)
	assert sel.size == (n_zeros + 4095) // 4096 + 1		# one select sample per 4096 non-gap columns + sentinel
	assert np.array_equal(sel[:-1], zero_cols[::4096].astype(np.uint32))
	assert sel[-1] == n


def test_index_layout_small_and_ragged():
	rng = np.random.default_rng(11)
	idx = panvc3_amd.Index()
	for n in (1, 63, 64, 447, 448, 449, 896, 1000, 5000, 20000):
		for p_gap in (0.0, 0.3, 1.0):
			s = "".join("-" if x else "ACGT"[i & 3] for i, x in enumerate(rng.random(n) < p_gap))
			_check_layout(idx, idx.add_aligned(s), s)
	# bits form agrees with the aligned form
	s = "".join("-" if x else "A" for x in rng.random(3000) < 0.2)
	gaps = np.frombuffer(s.encode(), dtype=np.uint8) == ord("-")
	words = np.packbits(np.concatenate([gaps, np.ones((-len(gaps)) % 64, dtype=bool)]), bitorder="little").view(np.uint64)
	e1, e2 = idx.add_aligned(s), idx.add_bits(words, len(s))
	b1, s1 = idx.entry_layout(e1)
	b2, s2 = idx.entry_layout(e2)
	assert np.array_equal(b1, b2) and np.array_equal(s1, s2)


def test_index_layout_shipped_msa_matches_oracle(oracle_mod):
	rows = load_test_msa()
	idx = panvc3_amd.Index()
	o = oracle_mod.Oracle()
	for k, row in rows.items():
		e = idx.add_aligned(row)
		oe = o.add_aligned(row)
		_check_layout(idx, e, row)
		blocks, sel = idx.entry_layout(e)
		# rank words and select samples agree with the oracle's rank0 / select0
		for b in range(0, blocks.shape[0] - 2, 3):
			assert int(blocks[b, 0]) & 0xFFFFFFFF == o.rank0(oe, b * 448)
		for j in range(0, sel.size - 1):
			assert sel[j] == o.select0(oe, 4096 * j + 1)


def test_set_reference_validates_length():
	idx = panvc3_amd.Index()
	e = idx.add_aligned("AC-GT")
	with pytest.raises(panvc3_amd.Error):
		idx.set_reference(e, "ACG")		# 4 non-gap columns need at least 4 characters
	idx.set_reference(e, "ACGT")
	with pytest.raises(panvc3_amd.Error):
		idx.set_reference(e + 1, "ACGT")
