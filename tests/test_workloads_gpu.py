"""HIP path vs oracle on the synthetic workloads of BASELINE.json (seeded, tests/synth) and on the edge cases the
reference's tests and source single out.  Bit-exact comparison of every output array; full-size configurations are
checked through size-independent properties plus a sampled oracle comparison."""
import numpy as np
import pytest

from helpers import BatchBuilder, assert_results_equal, cigar_decode, load_test_msa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def synth():
	from tests.synth import synth as s
	s.build()
	return s


def _run_both(pvc3, oracle_mod, genome, batch, ctx=None, score_mode=0, gap=(0, -1), threads=8, with_all_texts=False):
	index = pvc3.Index()
	genome.fill_index(index, with_all_texts)
	ctx = ctx or pvc3.Context(0)
	ctx.upload_index(index)
	got, stats = pvc3.project_batch(ctx, batch, gap[0], gap[1], score_mode)
	o = oracle_mod.Oracle()
	genome.fill_index(o, with_all_texts)
	want = o.batch_project(batch, gap[0], gap[1], score_mode, n_threads=threads)
	assert_results_equal(batch, got, want)
	return got, stats, ctx, o


@pytest.mark.parametrize("name,scale", [("C2", 0.2), ("C3", 0.005), ("C3G", 0.005), ("C5", 0.005)])
def test_scaled_configs(pvc3, oracle_mod, synth, name, scale):
	gp, rp, desc = synth.config(name, scale=scale)
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	batch = reads.batch()
	for mode in (0, 1):
		# Position windows sized for the reads (256-byte windows for the 250-base reads of C5: k_project_fast <true>) in the
		# second pass, the default (128 bytes: reads of more than 158 bases are left to k_project_mid) in the first.
		ctx = pvc3.Context(0)
		if mode: ctx.set_option("window_read_length", rp.read_len)
		got, stats, ctx, o = _run_both(pvc3, oracle_mod, genome, batch, ctx=ctx, score_mode=mode)
		assert stats["n_failed"] == 0
		assert stats["attempts"] == 1
		if mode and "default" == pvc3.paths_mode and "C3G" != name:
			assert stats["n_fast_path"] > 0.5 * reads.n
	if name == "C5":
		assert stats["n_realigned_records"] > 0.02 * reads.n		# long events: a few percent of the reads are realigned
	# mate symmetry: both reads of a pair see each other's projected position
	if rp.paired:
		a, b = got["dst_pos"][0::2], got["dst_pos"][1::2]
		assert np.array_equal(got["mate_dst_pos"][0::2][:b.size], b) and np.array_equal(got["mate_dst_pos"][1::2], a[:b.size])


@pytest.mark.parametrize("name,scale", [("C2", 0.03), ("C3", 0.002), ("C5", 0.002)])
def test_other_seeds_n_bases_and_soft_masking(pvc3, oracle_mod, synth, name, scale):
	"""Fresh genomes and reads (seeds other than the configuration's own), some with N in the reads and lower-case stretches
	in the reference: every output array against the oracle."""
	for seed in (101, 102, 103):
		gp, rp, _ = synth.config(name, scale=scale, seed=seed)
		rp.n_base_rate = 0.01 if seed % 2 else 0.0
		gp.lowercase_every = 7 if 0 == seed % 3 else 0
		genome = synth.Genome(gp)
		reads = synth.Reads(genome, rp)
		got, stats, ctx, o = _run_both(pvc3, oracle_mod, genome, reads.batch())
		assert stats["n_failed"] == 0
		ctx.close()


def test_device_checksum_matches_host_restatement(pvc3, oracle_mod, synth):
	"""pvc3_dbatch_checksum (used for the additive multi-GPU / full-size checks) against its host restatement."""
	from test_shard_gloo import _checksum
	gp, rp, _ = synth.config("C5", scale=0.0005)
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	index = pvc3.Index()
	genome.fill_index(index)
	ctx = pvc3.Context(0)
	ctx.upload_index(index)
	db = pvc3.DeviceBatch(ctx, reads.batch())
	db.project()
	db.finish()
	got = db.download()
	for first in (0, 12345678901):
		assert db.checksum(first) == _checksum(got, first)
	db.close()


def test_c2_full_size_bit_exact(pvc3, oracle_mod, synth):
	"""BASELINE.json configs[1]: 1 M synthetic 150 bp single-end reads, 2 founders, bit-exact vs the CPU path."""
	gp, rp, desc = synth.config("C2")
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	assert reads.n == 1_000_000
	got, stats, ctx, o = _run_both(pvc3, oracle_mod, genome, reads.batch(), threads=16)
	assert stats["n_failed"] == 0 and stats["n_realigned_records"] > 0


def test_shipped_msa_reads(pvc3, oracle_mod, synth):
	"""C1 stand-in: reads sampled from the reference's shipped 5-row test MSA (bowtie2 is not available to make the real input)."""
	rows = load_test_msa()
	order = ["REF", "1", "2", "3", "4"]
	genome = synth.Genome(aligned_rows=[rows[k] for k in order])
	gp, rp, _ = synth.config("C3", scale=1e-6)
	rp.n_records = 2000
	rp.read_len = 100
	reads = synth.Reads(genome, rp)
	got, stats, ctx, o = _run_both(pvc3, oracle_mod, genome, reads.batch())
	assert stats["n_failed"] == 0


def test_soft_masked_reference_and_n_bases(pvc3, oracle_mod, synth):
	"""Quirk Q3: the rewrite step compares raw bytes (lower-case reference != upper-case read => X), the realignment
	compares dna5 ranks (case-insensitive, N == N)."""
	gp, rp, _ = synth.config("C5", scale=0.002)
	gp.lowercase_every = 7
	rp.n_base_rate = 0.01
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	got, stats, ctx, o = _run_both(pvc3, oracle_mod, genome, reads.batch())
	ops = got["out_cigar"] & 0xf
	assert (ops == 8).sum() > 0.1 * reads.n		# plenty of X from the case mismatch


def _edge_index(pvc3, oracle_mod, pairs):
	"""pairs: list of (src_aligned, dst_aligned, dst_text or None)."""
	idx, o = pvc3.Index(), oracle_mod.Oracle()
	ids = []
	for src, dst, text in pairs:
		s, d = idx.add_aligned(src), idx.add_aligned(dst)
		assert (o.add_aligned(src), o.add_aligned(dst)) == (s, d)
		if text is not None:
			idx.set_reference(d, text); o.set_reference(d, text)
		ids.append((s, d))
	ctx = pvc3.Context(0)
	ctx.upload_index(idx)
	return ctx, o, ids


def test_abi_call_sequence_capacity_and_empty_batch(pvc3, oracle_mod):
	"""Error behaviour of the C ABI on a GPU: wrong call order, output buffers that are too small, an empty batch, page-locked
	arrays in and out.  (Codes: include/pvc3.h.)"""
	import ctypes as C
	from panvc3_amd import _lib
	L = _lib.lib()
	E_STATE, E_CAPACITY = -6, -5
	ctx = pvc3.Context(0)
	bb = BatchBuilder()
	bb.add(0, 0, 1, "5M", "ACGTA", qual=[30] * 5)
	batch = bb.build()
	with pytest.raises(pvc3.Error) as ei:		# no index uploaded yet
		pvc3.DeviceBatch(ctx, batch)
	assert ei.value.code == E_STATE
	idx = pvc3.Index()
	s = idx.add_aligned("ACGTA-CGT"); d = idx.add_aligned("ACGTAACGT")
	idx.set_reference(d, "ACGTAACGT")
	ctx.upload_index(idx)
	db = pvc3.DeviceBatch(ctx, batch)
	with pytest.raises(pvc3.Error) as ei:		# finish before project
		db.finish()
	assert ei.value.code == E_STATE
	st = _lib.BatchStats()
	bo = _lib.BatchOut()
	assert L.pvc3_dbatch_download(ctx._h, db._h, C.byref(bo)) == E_STATE		# download before finish
	db.project()
	stats = db.finish()
	assert stats["n_records"] == 1 and stats["out_cigar_total"] == 1
	out = db.download()
	assert cigar_decode(out["out_cigar"]) == "5="
	# too little room for the operations: an error, nothing written past the buffers
	small = dict(dst_pos=np.zeros(1, np.uint32), mate_dst_pos=np.zeros(1, np.uint32), status=np.zeros(1, np.int32), out_cigar_off=np.zeros(2, np.uint64),
				 run_off=np.zeros(2, np.uint64), out_cigar=np.full(4, 0xdeadbeef, np.uint32), runs=np.zeros(1, pvc3.RUN_DTYPE))
	for k in ("dst_pos", "mate_dst_pos", "status", "out_cigar_off", "out_cigar", "run_off", "runs"):
		setattr(bo, k, small[k].ctypes.data)
	bo.out_cigar_cap = 0; bo.run_cap = 0
	assert L.pvc3_dbatch_download(ctx._h, db._h, C.byref(bo)) == E_CAPACITY
	assert (small["out_cigar"] == 0xdeadbeef).all()
	db.close()
	# an empty batch is not an error
	empty = BatchBuilder().build()
	got, stats = pvc3.project_batch(ctx, empty)
	assert stats["n_records"] == 0 and got["dst_pos"].size == 0 and got["out_cigar_off"].tolist() == [0]
	# page-locked arrays in and out give the same bytes
	pb, owners = pvc3.pinned_batch(batch)
	got2, _ = pvc3.project_batch(ctx, pb, pinned_out=True)
	assert cigar_decode(got2["out_cigar"]) == "5=" and got2["dst_pos"].tolist() == out["dst_pos"].tolist()
	ctx.close()


def test_edge_cases_and_statuses(pvc3, oracle_mod):
	rng = np.random.default_rng(1)
	src = "GATTACA" * 40
	dst = "GAT-ACA" * 40
	ctx, o, ids = _edge_index(pvc3, oracle_mod, [
		(src, dst, dst.replace("-", "")),
		("ACGT" * 30, "ACGT" * 30, None),							# destination without text
		("AC-GT", "ACGGT", "ACGGT"),
	])
	(s0, d0), (s1, d1), (s2, d2) = ids
	q = lambda n: "".join("ACGT"[x] for x in rng.integers(0, 4, size=n))
	bb = BatchBuilder()
	bb.add(0, s0, d0, "", "")											# empty CIGAR and SEQ
	bb.add(5, s0, d0, "10S", q(10))										# clipping only
	bb.add(5, s0, d0, "3H2S5M1P5M3N4M2I4D6=3X2S1H", q(2 + 5 + 5 + 4 + 2 + 6 + 3 + 2), qual=[30] * 29)	# every operation type
	bb.add(0, s0, d0, "280M", src, qual=[20] * 280)						# whole sequence
	bb.add(279, s0, d0, "1M", "A")										# last base
	bb.add(280, s0, d0, "1M", "A")										# src_pos == length        -> status 1
	bb.add(270, s0, d0, "20M", q(20))									# runs past the end        -> status 1
	bb.add(0, s0, d0, "5M", q(3))										# CIGAR longer than SEQ    -> status 3
	bb.add(0, s0, d0, np.array([(5 << 4) | 9], dtype=np.uint32), q(5))	# operation code 9         -> status 2
	bb.add(0, s0, d0, np.array([(5 << 4) | 15, (5 << 4) | 0], dtype=np.uint32), q(5))
	bb.add(0, s1, d1, "10M", q(10))										# no destination text      -> status 4
	bb.add(0, s1, d1, "4S3I", q(7))										# ... but nothing to compare: fine
	bb.add(0, s0, d2, "3M", q(3))										# column counts differ     -> status 1
	bb.add(0, s2, d2, "0M2M0I2M", q(4))									# zero-length operations vanish (cigar.cc:65-66)
	bb.add(3, s0, s0, "50M", src[3:53])									# identity projection needs the text of s0 -> status 4
	bb.add(0, s0, d0, "5M", q(5), mate=(s0, d0, 279))
	bb.add(0, s0, d0, "5M", q(5), mate=(s0, d0, 280))					# mate beyond the sequence: PVC3_NO_MATE
	bb.add(0, s0, d0, "5M", q(5), mate=(s2, d2, 1))
	batch = bb.build()
	got, stats = pvc3.project_batch(ctx, batch)
	want = o.batch_project(batch)
	assert_results_equal(batch, got, want)
	assert list(got["status"][5:11]) == [1, 1, 3, 2, 2, 4]
	assert got["status"][12] == 1 and got["status"][14] == 4
	assert got["mate_dst_pos"][16] == 0xFFFFFFFF
	assert stats["n_failed"] == 8
	# empty batch
	empty = BatchBuilder().build()
	got, stats = pvc3.project_batch(ctx, empty)
	assert got["out_cigar_off"].tolist() == [0] and stats["n_records"] == 0


def test_many_operations_spill_and_big_dp(pvc3, oracle_mod):
	"""More runs than the lane buffers hold (arena spill), more than 64 input operations, and an indel run whose DP needs
	several 64-row stripes and the global scratch."""
	rng = np.random.default_rng(2)
	n = 400
	base = "".join("ACGT"[x] for x in rng.integers(0, 4, size=n))
	zig_src = "".join(c + "-" for c in base)		# A-C-G-...
	zig_dst = "".join("-" + c for c in base)		# -A-C-G...
	ctx, o, ids = _edge_index(pvc3, oracle_mod, [
		(base, base, base),
		(zig_src, zig_dst, base),
	])
	(s0, d0), (s1, d1) = ids
	bb = BatchBuilder()
	# alternating mismatches: 300 runs of length 1
	alt = "".join(c if i % 2 == 0 else "ACGT"[("ACGT".index(c) + 1) % 4] for i, c in enumerate(base[:300]))
	bb.add(0, s0, d0, "300M", alt, qual=[37] * 300)
	# 150 input operations
	bb.add(10, s0, d0, "1M1=" * 75, base[10:160])
	bb.add(10, s0, d0, "1M1I" * 100, "".join(base[10 + i] + "T" for i in range(100)), qual=[30] * 200)
	# zig-zag MSA: every base becomes 1I1D => one indel run of 300 x 300
	bb.add(20, s1, d1, "300M", base[20:320], qual=[int(x) for x in rng.integers(2, 42, size=300)])
	bb.add(0, s1, d1, "70M", base[0:70], qual=None)
	bb.add(100, s1, d1, "30M5S", base[100:130] + "AAAAA", qual=[41] * 35)
	batch = bb.build()
	got, stats = pvc3.project_batch(ctx, batch)
	want = o.batch_project(batch)
	assert_results_equal(batch, got, want)
	assert int(got["out_cigar_off"][1]) == 300
	assert stats["n_runs"] == 3


def test_long_gap_runs(pvc3, oracle_mod):
	"""Gap runs far longer than the two-block window: select0 gallops over the rank words, the walk crosses them."""
	rng = np.random.default_rng(3)
	seg = lambda k: "".join("ACGT"[x] for x in rng.integers(0, 4, size=k))
	a, b, c = seg(3000), seg(100_000), seg(3000)
	# src lacks b (100 kbp deletion in the founder); dst lacks an insertion of the founder elsewhere
	ins = seg(50_000)
	src = a + "-" * len(b) + c + ins + a
	dst = a + b + c + "-" * len(ins) + a
	ctx, o, ids = _edge_index(pvc3, oracle_mod, [(src, dst, dst.replace("-", ""))])
	s, d = ids[0]
	stext = src.replace("-", "")
	bb = BatchBuilder()
	for p in (2900, 2990, 2999, 3000, 5900, 5990, 6001, 30000, 55990, 56010):
		bb.add(p, s, d, "150M", stext[p:p + 150], qual=[30] * 150, mate=(s, d, p + 300))
	bb.add(2500, s, d, "3000M", stext[2500:5500])
	batch = bb.build()
	got, stats = pvc3.project_batch(ctx, batch)
	assert_results_equal(batch, got, o.batch_project(batch))
	assert "100000D" in cigar_decode(got["out_cigar"][int(got["out_cigar_off"][0]):int(got["out_cigar_off"][1])])


def test_workspace_growth_and_scratch_paths(pvc3, oracle_mod, synth):
	"""Arena / runs / scratch overflow make the library grow the workspace and repeat on the GPU; forcing the DP out of LDS
	must not change a single byte."""
	gp, rp, _ = synth.config("C5", scale=0.003)
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	batch = reads.batch()
	base, stats0, ctx, o = _run_both(pvc3, oracle_mod, genome, batch)
	index = pvc3.Index()
	genome.fill_index(index)
	for opts in (dict(arena_initial=1000), dict(runs_initial=3), dict(realign_lds_bytes=64, packed_dp=0), dict(realign_lds_bytes=64, scratch_per_wave=512, packed_dp=0), dict(realign_waves=3), dict(big_waves=2, packed_dp=0), dict(trace_initial=4096), dict(fast_path=0)):
		c2 = pvc3.Context(0)
		for k, v in opts.items():
			c2.set_option(k, v)
		c2.upload_index(index)
		got, stats = pvc3.project_batch(c2, batch)
		for k in ("status", "dst_pos", "mate_dst_pos", "out_cigar_off", "out_cigar", "run_off", "runs"):
			assert np.array_equal(got[k], base[k]), (opts, k)
		if "arena_initial" in opts or "runs_initial" in opts or "scratch_per_wave" in opts or ("trace_initial" in opts and "walk_only" != pvc3.paths_mode):
			assert stats["attempts"] > 1, opts
		c2.close()


def test_fast_path_boundaries(pvc3, oracle_mod):
	"""The lane-parallel fast path (simple CIGAR, no gap column in the window) must agree with the general column walk and
	the oracle at its boundaries: 256 / 257 bases, clips of either kind, odd query offsets, gaps right next to the window."""
	rng = np.random.default_rng(9)
	seg = lambda k: "".join("ACGT"[x] for x in rng.integers(0, 4, size=k))
	a, b = seg(700), seg(700)
	src = a + "-" + b			# one dst-only column right after a
	dst = a + "T" + b
	pairs = [(src, dst, dst.replace("-", "")), (dst, src, src.replace("-", ""))]
	ctx, o, ids = _edge_index(pvc3, oracle_mod, pairs)
	bb = BatchBuilder()
	mut = lambda t: "".join(c if rng.random() > 0.1 else "ACGTN"[rng.integers(5)] for c in t)
	for (s, d), text in ((ids[0], src.replace("-", "")), (ids[1], dst)):
		for L in (1, 2, 7, 8, 9, 31, 32, 33, 63, 64, 65, 150, 255, 256, 257, 300):
			for p in (0, 1, 699 - L, 700 - L, 701 - L, 702, 705):
				if p < 0 or p + L > len(text):
					continue
				q = mut(text[p:p + L])
				bb.add(p, s, d, "%dM" % L, q, qual=[30] * L)
				bb.add(p, s, d, "3S%dM" % L, "ACG" + q)
				bb.add(p, s, d, "2H%d=4S" % L, q + "ACGT")
				bb.add(p, s, d, "1S%dX2H" % L, "N" + q)
				if L > 2:
					bb.add(p, s, d, "1M%d=1X" % (L - 2), q)
					bb.add(p, s, d, "1M1I%dM" % (L - 2), q[:1] + "G" + q[1:-1])		# not simple
	batch = bb.build()
	want = o.batch_project(batch)
	for fp in (1, 0):
		c2, _, _ = _edge_index(pvc3, oracle_mod, pairs)
		c2.set_option("fast_path", fp)
		got, stats = pvc3.project_batch(c2, batch)
		assert_results_equal(batch, got, want)
		c2.close()


def test_project_positions(pvc3, oracle_mod, synth):
	gp, rp, _ = synth.config("C3", scale=0.002)
	genome = synth.Genome(gp)
	index = pvc3.Index()
	genome.fill_index(index)
	ctx = pvc3.Context(0)
	ctx.upload_index(index)
	o = oracle_mod.Oracle()
	genome.fill_index(o)
	rng = np.random.default_rng(4)
	per = 1 + genome.n_founders
	src = rng.integers(0, genome.n_entries, size=20000).astype(np.uint32)
	dst = (src // per * per).astype(np.uint32)
	lens = np.array([o.zeros(int(e)) for e in range(genome.n_entries)])
	pos = (rng.random(20000) * lens[src]).astype(np.uint32)
	pos[:50] = lens[src[:50]] - 1
	pos[50:60] = lens[src[50:60]]			# out of range
	got = ctx.project_positions(src, dst, pos)
	want = np.array([o.project_position(int(s), int(d), int(p)) if p < lens[s] else 0xFFFFFFFF for s, d, p in zip(src, dst, pos)], dtype=np.uint32)
	assert np.array_equal(got, want)


def test_select_guess_on_uneven_gap_density(pvc3, oracle_mod):
	"""Rows whose gap density changes abruptly inside a select-sample stretch (4096 non-gap columns): the interpolated guess of
	the look-up then lies far before or far PAST the target, and the select has to walk forwards or backwards from the probe
	(wave.hpp dir_select0).  Every position of every row against the oracle, through pvc3_project_positions and through the
	whole path (reads laid over the same places), with and without the paired / windowed copies of the index."""
	rng = np.random.default_rng(77)
	n_cols = 60000
	def row(kind):
		g = np.zeros(n_cols, dtype=bool)
		if "front" == kind: g[1000:21000] = True						# 20 000 gap columns, then none
		elif "back" == kind: g[30000:52000] = True
		elif "bursts" == kind:
			for s in range(0, n_cols - 3000, 9000): g[s + 500:s + 500 + 2500] = True
		elif "dense" == kind: g[:] = rng.random(n_cols) < 0.6
		elif "mixed" == kind: g[:30000] = rng.random(30000) < 0.7; g[30000:] = rng.random(30000) < 0.01
		return g
	kinds = ["none", "front", "back", "bursts", "dense", "mixed"]
	rows = [row(k) for k in kinds]
	texts = ["".join("-" if x else "ACGT"[i & 3] for i, x in enumerate(r)) for r in rows]
	o = oracle_mod.Oracle()
	for opts in ({}, {"position_windows": 0}, {"paired_blocks": 0}, {"window_read_length": 250}, {"paired_blocks": 0, "text_window": 0, "position_windows": 0}):
		index = pvc3.Index()
		for t in texts:
			e = index.add_aligned(t)
			index.set_reference(e, t.replace("-", ""))
		ctx = pvc3.Context(0)
		for k, v in opts.items(): ctx.set_option(k, v)
		ctx.upload_index(index)
		if not opts:
			for t in texts:
				e = o.add_aligned(t)
				o.set_reference(e, t.replace("-", ""))
		for s in range(len(rows)):
			n_zero = int((~rows[s]).sum())
			pos = np.arange(0, n_zero, 7, dtype=np.uint32)
			for d in range(len(rows)):
				got = ctx.project_positions(np.full(pos.size, s, dtype=np.uint32), np.full(pos.size, d, dtype=np.uint32), pos)
				cols = np.flatnonzero(~rows[s])[pos]
				want = np.concatenate([[0], np.cumsum(~rows[d])])[cols].astype(np.uint32)		# rank0_dst(select0_src(pos + 1))
				assert np.array_equal(got, want), (kinds[s], kinds[d], opts)
		ctx.close()
	# spot check of the numpy expectation itself against the oracle
	for s, d, p in ((1, 0, 5), (4, 2, 12345), (5, 3, 20000)):
		cols = np.flatnonzero(~rows[s])[p]
		assert o.project_position(s, d, p) == int((~rows[d][:cols]).sum())
	# The whole path over the same rows: reads of 40 - 150 bases copied from every source row (a few substitutions), plain
	# CIGARs and some with a clip, an insertion or a deletion, every pair of rows, positions spread over the changes of density.
	bb = BatchBuilder()
	for s in range(len(rows)):
		seq = texts[s].replace("-", "")
		for d in range(len(rows)):
			for p0 in list(range(0, len(seq) - 200, 997)) + [len(seq) - 151]:
				L = int(rng.integers(40, 151))
				q = list(seq[p0:p0 + L])
				for x in rng.integers(0, L, size=2): q[x] = "ACGT"[int(rng.integers(0, 4))]
				kind = int(rng.integers(0, 4))
				if 0 == kind: cig, query = "%dM" % L, q
				elif 1 == kind: cig, query = "5S%dM" % L, list("ACGTA") + q
				elif 2 == kind: cig, query = "%dM3I%dM" % (L // 2, L - L // 2), q[:L // 2] + list("TTT") + q[L // 2:]
				else: cig, query = "%dM4D%dM" % (L // 2, L - L // 2 - 4), q[:L // 2] + q[L // 2 + 4:]
				bb.add(p0, s, d, cig, "".join(query))
	batch = bb.build()
	want = o.batch_project(batch, 0, -1, 0, n_threads=8)
	for opts in ({}, {"position_windows": 0}, {"paired_blocks": 0}, {"window_read_length": 250}, {"paired_blocks": 0, "text_window": 0, "position_windows": 0}):
		index = pvc3.Index()
		for t in texts:
			e = index.add_aligned(t)
			index.set_reference(e, t.replace("-", ""))
		ctx = pvc3.Context(0)
		for k, v in opts.items(): ctx.set_option(k, v)
		ctx.upload_index(index)
		got, stats = pvc3.project_batch(ctx, batch, 0, -1, 0)
		assert_results_equal(batch, got, want)
		ctx.close()


@pytest.mark.parametrize("window_read_length", [150, 250])
def test_every_start_position_against_window_boundaries(pvc3, oracle_mod, window_read_length):
	"""Position windows: a record is served from the window of its start position if the columns from the window's first to
	the record's last are free of gaps in both rows, and from the window's last clean column otherwise.  Records of many
	lengths (one base ... the longest the compare takes, across the 16-byte pieces of its loads) from EVERY source position of
	an MSA with single gap columns and short gap runs in either row, with leading soft clips of both parities (the compare is
	anchored at a byte boundary of SEQ) and mates on every position."""
	rng = np.random.default_rng(77)
	n_cols = 2600
	ref = "".join("ACGT"[x] for x in rng.integers(0, 4, size=n_cols))
	gs = np.zeros(n_cols, dtype=bool); gd = np.zeros(n_cols, dtype=bool)
	for c in (70, 71, 133, 190, 191, 192, 260, 321, 500, 511, 512, 513, 700, 1023, 1024, 1500, 1501, 2000, 2047, 2048, 2300):
		gs[c] = True
	for c in (64, 129, 130, 255, 256, 400, 640, 641, 642, 643, 900, 1100, 1536, 1537, 1900, 2200, 2201, 2500):
		gd[c] = True
	gd &= ~gs		# (a column that is a gap in both rows is no column of either sequence)
	src = "".join("-" if g else c for c, g in zip(ref, gs))
	dst = "".join("-" if g else c for c, g in zip(ref, gd))
	idx, o = pvc3.Index(), oracle_mod.Oracle()
	s_id, d_id = idx.add_aligned(src), idx.add_aligned(dst)
	assert (o.add_aligned(src), o.add_aligned(dst)) == (s_id, d_id)
	idx.set_reference(d_id, dst.replace("-", "")); o.set_reference(d_id, dst.replace("-", ""))
	ctx = pvc3.Context(0)
	ctx.set_option("window_read_length", window_read_length)
	ctx.upload_index(idx)
	text = src.replace("-", "")
	lengths = [1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 150, 151, 157, 158, 159, 160, 200, 249, 250, 251, 254, 255, 256, 300]
	bb = BatchBuilder()
	k = 0
	for p0 in range(len(text)):
		for rep in range(2):
			L = lengths[k % len(lengths)]; k += 1
			if p0 + L > len(text): L = len(text) - p0
			q = list(text[p0:p0 + L])
			if L > 4 and k % 3 == 0: q[L // 2] = "ACGT"[(("ACGT".index(q[L // 2])) + 1) % 4]		# one substitution
			lead = k % 4				# 0 .. 3 soft-clipped bases in front: both parities of the SEQ offset
			trail = (k // 4) % 3
			cig = ("%dS" % lead if lead else "") + "%dM" % L + ("%dS" % trail if trail else "")
			mate = (s_id, d_id, (p0 * 7 + rep) % len(text))
			bb.add(p0, s_id, d_id, cig, "T" * lead + "".join(q) + "G" * trail, qual=[30] * (lead + L + trail), mate=mate)
	batch = bb.build()
	got, stats = pvc3.project_batch(ctx, batch)
	assert_results_equal(batch, got, o.batch_project(batch, n_threads=8))
	assert stats["n_failed"] == 0
	if "default" == pvc3.paths_mode:
		assert stats["n_fast_path"] > 0.15 * batch["src_pos"].size		# (a gap column every hundred columns or so: most longer records meet one)
	ctx.close()


def test_many_entries_and_long_cigars(pvc3, oracle_mod):
	"""What k_project_fast keeps in LDS has a size: the first 256 entries of the table (others are read from global memory and
	handed over through LDS, their records left to k_project_mid) and the 256 CIGAR operations that follow the previous
	group's (a group with more fetches the rest itself).  170 MSAs of two rows each = 340 entries; records on all of them,
	with mates across the 256 boundary, plain CIGARs and CIGARs of nine and of seventy operations."""
	rng = np.random.default_rng(11)
	n_msa = 170
	pairs, texts = [], []
	for k in range(n_msa):
		L = 400 + 3 * k
		ref = "".join("ACGT"[x] for x in rng.integers(0, 4, size=L))
		gaps_s = np.zeros(L, dtype=bool); gaps_d = np.zeros(L, dtype=bool)
		if k % 3: gaps_s[100 + k:105 + k] = True			# a deletion of the founder ...
		if k % 2: gaps_d[250:253] = True					# ... and an insertion of it (gap in the reference row)
		src = "".join("-" if g else c for c, g in zip(ref, gaps_s))
		dst = "".join("-" if g else c for c, g in zip(ref, gaps_d))
		pairs.append((src, dst, dst.replace("-", "")))
		texts.append(src.replace("-", ""))
	ctx, o, ids = _edge_index(pvc3, oracle_mod, pairs)
	assert max(max(p) for p in ids) == 2 * n_msa - 1
	bb = BatchBuilder()
	for rep in range(3):
		for k in range(n_msa):
			s, d = ids[k]
			t = texts[k]
			p0 = int(rng.integers(0, len(t) - 160))
			L = int(rng.integers(80, 151))
			q = list(t[p0:p0 + L])
			for x in rng.integers(0, L, size=2): q[x] = "ACGT"[int(rng.integers(0, 4))]
			ko = (k + 97) % n_msa							# the mate lies on another MSA, often across the 256-entry boundary
			mate = (ids[ko][0], ids[ko][1], int(rng.integers(0, len(texts[ko]) - 1)))
			kind = (k + rep) % 3
			if 0 == kind:
				bb.add(p0, s, d, "%dM" % L, "".join(q), qual=[30] * L, mate=mate)
			elif 1 == kind:		# nine operations
				cig = "3S" + "".join("%d%s" % (c, op) for c, op in zip([L // 4, 1, L // 4, 1, L // 4, 2, L - 3 * (L // 4) - 4], "M=XM=M=")) + "2S"
				assert sum(int(x) for x in __import__("re").findall(r"(\d+)[M=X]", cig)) == L
				bb.add(p0, s, d, cig, "AAA" + "".join(q) + "CC", qual=[20] * (L + 5), mate=mate)
			else:				# seventy operations: 1M 1= 1X ... (every group of 64 such records has far more than 256)
				ops = "".join("1%s" % "M=X"[i % 3] for i in range(70))
				bb.add(p0, s, d, ops, "".join(q[:70]), qual=[40] * 70, mate=mate)
	batch = bb.build()
	got, stats = pvc3.project_batch(ctx, batch)
	assert_results_equal(batch, got, o.batch_project(batch))
	assert stats["n_failed"] == 0
	ctx.close()


def test_position_windows_left_out_when_they_do_not_fit(pvc3, oracle_mod, synth):
	"""pvc3_context_upload_index builds the position windows only if they leave `window_reserve` bytes of the device free; an
	index without them is served by k_project_fast_probes with the same results (and less HBM)."""
	gp, rp, _ = synth.config("C3", scale=0.004)
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	batch = reads.batch()
	sizes = {}
	for name, reserve in (("with", None), ("without", 1 << 50)):
		ctx = pvc3.Context(0)
		if reserve is not None: ctx.set_option("window_reserve", reserve)		# more than any device has
		got, stats, ctx, o = _run_both(pvc3, oracle_mod, genome, batch, ctx=ctx)
		assert stats["n_failed"] == 0
		sizes[name] = ctx.hbm_bytes()[0]
		ctx.close()
	if "default" == pvc3.paths_mode:
		assert sizes["with"] > sizes["without"] + 2 * 0.9 * sum(genome.entry(e)[2].size for e in range(genome.n_entries))		# two bytes per source position


def test_plain_index_layout_matches_oracle(pvc3, oracle_mod, synth):
	"""The same workloads without the position windows, the paired blocks and the windowed text (options at upload): the plain
	64-byte blocks and the plain packed text are then what k_project_fast_probes and k_project_mid read."""
	for name, scale in (("C3", 0.004), ("C3G", 0.002)):
		gp, rp, _ = synth.config(name, scale=scale)
		genome = synth.Genome(gp)
		reads = synth.Reads(genome, rp)
		ctx = pvc3.Context(0)
		ctx.set_option("paired_blocks", 0)
		ctx.set_option("text_window", 0)
		ctx.set_option("position_windows", 0)
		got, stats, ctx, o = _run_both(pvc3, oracle_mod, genome, reads.batch(), ctx=ctx)
		assert stats["n_failed"] == 0
		index_bytes, _ = ctx.hbm_bytes()
		ctx2 = pvc3.Context(0)
		index = pvc3.Index()
		genome.fill_index(index)
		ctx2.upload_index(index)
		assert ctx2.hbm_bytes()[0] > index_bytes		# the default upload carries the two copies
		ctx.close(); ctx2.close()


def test_destinations_named_by_the_caller(pvc3, oracle_mod, synth):
	"""Two chromosomes whose MSAs have the same number of columns: the replica's paired blocks and position windows of a row are made
	towards the nearest row with a text, which for half of the first chromosome's founders is the other chromosome's reference —
	their records are then projected correctly, but by the walk kernels.  pvc3_index_set_destination names the pairs (the command
	line knows them): the same results, and the records are finished by k_project_fast again."""
	gp, rp, _ = synth.config("C3", scale=0.002)
	gp.n_contigs = 1
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	batch = reads.batch()
	per = genome.n_entries		# REF + founders of the one contig
	def fill(index, name_pairs):
		for copy in range(2):		# the same MSA as two chromosomes: entries 0 .. per - 1 and per .. 2 per - 1
			for e in range(per):
				words, n_cols, text = genome.entry(e)
				got = index.add_bits(words, n_cols)
				assert got == copy * per + e
				if 0 == e:
					index.set_reference(got, text)
		if name_pairs:
			for copy in range(2):
				for e in range(per):
					index.set_destination(copy * per + e, copy * per)
	o = oracle_mod.Oracle()
	fill(o, False)
	want = o.batch_project(batch, n_threads=4)
	fast = {}
	for named in (False, True):
		index = pvc3.Index()
		fill(index, named)
		ctx = pvc3.Context(0)
		ctx.upload_index(index)
		got, stats = pvc3.project_batch(ctx, batch)
		assert_results_equal(batch, got, want)
		fast[named] = stats["n_fast_path"]
		ctx.close()
	if "default" == pvc3.paths_mode:
		assert fast[True] > 0.9 * reads.n and fast[False] < 0.8 * fast[True], fast
	with pytest.raises(pvc3.Error):
		index.set_destination(0, 2 * per)


@pytest.mark.parametrize("name", ["C3", "C4", "C5"])
def test_full_size_properties(pvc3, oracle_mod, synth, name):
	"""BASELINE.json configs[2] and configs[4] at full size (20 M pairs, 8 founders, 1 Gbp; 2x150 and 2x250 bp with long
	indels) and one GPU's shard of configs[3] (200 M pairs over 8 GPUs = 50 M records per GPU, as bench.py --config C4
	generates it): size-independent properties + sampled oracle parity."""
	if "default" != pvc3.paths_mode:
		pytest.skip("full size is run in the product configuration only")
	gp, rp, desc = synth.config(name)
	if "C4" == name:
		rp.n_records //= 8
		rp.seed += 3000		# the shard of rank 3
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	batch = reads.batch()
	n = reads.n
	assert n == (50_000_000 if "C4" == name else 40_000_000)
	index = pvc3.Index()
	genome.fill_index(index)
	ctx = pvc3.Context(0)
	ctx.set_option("window_read_length", rp.read_len)		# as bench.py does
	ctx.upload_index(index)
	db = pvc3.DeviceBatch(ctx, batch)
	db.project()
	stats = db.finish()
	whole = db.checksum(0)
	got = db.download()
	db.close()
	assert stats["n_failed"] == 0
	# 1. query-length conservation: = X I S counts add up to l_seq for every record
	ops = got["out_cigar"] & 0xf
	cnt = (got["out_cigar"] >> 4).astype(np.int64)
	qlen = np.where((ops == 7) | (ops == 8) | (ops == 1) | (ops == 4), cnt, 0)
	csum = np.concatenate([[0], np.cumsum(qlen)])
	per_rec = csum[got["out_cigar_off"][1:].astype(np.int64)] - csum[got["out_cigar_off"][:-1].astype(np.int64)]
	assert np.array_equal(per_rec, batch["l_seq"].astype(np.int64))
	# 2. adjacent operations differ (cigar_buffer / collapse merged everything), no zero counts
	same_as_prev = ops[1:] == ops[:-1]
	starts = np.zeros(ops.size, dtype=bool)
	starts[got["out_cigar_off"][:-1].astype(np.int64)[got["out_cigar_off"][:-1] < ops.size]] = True
	assert not (same_as_prev & ~starts[1:]).any() and (cnt > 0).all()
	# 3. mate symmetry
	assert np.array_equal(got["mate_dst_pos"][0::2], got["dst_pos"][1::2]) and np.array_equal(got["mate_dst_pos"][1::2], got["dst_pos"][0::2])
	# 4. the checksum is additive over shards (what the multi-GPU path relies on): 4 contiguous shards
	total = 0
	for k in range(4):
		lo, hi = n * k // 4, n * (k + 1) // 4
		sb = synth.slice_batch(batch, lo, hi)
		d2 = pvc3.DeviceBatch(ctx, sb)
		d2.project()
		d2.finish()
		total = (total + d2.checksum(lo)) & 0xFFFFFFFFFFFFFFFF
		if k == 2:
			# 5. sampled oracle parity inside the third shard
			o = oracle_mod.Oracle()
			genome.fill_index(o)
			sample = synth.slice_batch(batch, lo + 12345, lo + 12345 + 300_000)
			g2, _ = pvc3.project_batch(ctx, sample)
			assert_results_equal(sample, g2, o.batch_project(sample, n_threads=16))
		d2.close()
	assert total == whole


def test_indel_run_shapes_and_gap_scores(pvc3, oracle_mod):
	"""One indel run of m reference bases against n query bases for every packing shape of the alignment kernels (four lines of
	the matrix per lane, lane groups of 3 / 4 / 6 / 8 / 12 / 16 lanes — a shorter side of up to 12 / 16 / 24 / 32 / 48 / 64, on
	either side of every limit, with and without a partly used last lane — lines along the query or along the reference, runs of
	different lengths side by side in a wavefront, the striped kernel beyond 64 x ... or 128 steps),
	with related and unrelated slices (co-optimal paths), mixed base qualities, N bases, lower-case reference text, records
	without qualities, and gap scores with a positive, zero and negative opening score (open + extension > 0, == 0, < 0)."""
	rng = np.random.default_rng(77)
	seg = lambda k: "".join("ACGT"[x] for x in rng.integers(0, 4, size=k))
	shapes = [(1, 1), (1, 16), (16, 1), (16, 16), (17, 5), (5, 17), (15, 40), (40, 15), (16, 113), (113, 16), (17, 17), (32, 32), (32, 60),
			  (60, 32), (33, 32), (32, 33), (33, 33), (64, 64), (64, 65), (65, 64), (40, 64), (64, 40), (65, 10), (10, 65), (100, 20), (20, 100),
			  (100, 29), (100, 30), (30, 100), (120, 9), (9, 120), (3, 126), (126, 3), (127, 2), (2, 128), (70, 70), (150, 7), (1, 200),
			  (2, 7), (3, 3), (5, 5), (6, 9), (7, 7), (9, 9), (11, 40), (12, 30), (30, 12), (13, 30), (30, 13), (23, 23), (24, 50), (50, 24), (25, 50),
			  (50, 25), (47, 47), (48, 60), (60, 48), (49, 60), (60, 49), (63, 63), (64, 61), (61, 64), (12, 12), (13, 13), (24, 24), (25, 25), (48, 48), (49, 49)]
	pairs, recs = [], []
	for m, n in shapes:
		for variant in range(3):
			p, s = seg(int(rng.integers(3, 40))), seg(int(rng.integers(3, 40)))
			q = seg(n)
			if 0 == variant:
				r = seg(m)								# unrelated
			elif 1 == variant:
				r = (q * (m // max(1, n) + 2))[:m]		# the query slice repeated: many co-optimal alignments
			else:
				r = "".join(c if rng.random() < 0.7 else "ACGT"[rng.integers(4)] for c in (q + seg(m))[:m])
			if 2 == variant and m > 4:
				r = r[:2] + r[2:4].lower() + "N" + r[5:]	# soft-masked reference text and an N (the alignment compares dna5 ranks)
			src = p + q + "-" * m + s
			dst = p + "-" * n + r + s
			query = p + q + s
			if 1 == variant and n > 3:
				query = query[:len(p) + 1] + "N" + query[len(p) + 2:]
			qual = [int(x) for x in rng.integers(0, 46, size=len(query))] if variant != 0 or (m + n) % 2 else None
			pairs.append((src, dst, dst.replace("-", "")))
			recs.append((query, qual))
	ctx, o, ids = _edge_index(pvc3, oracle_mod, pairs)
	bb = BatchBuilder()
	for (s, d), (query, qual) in zip(ids, recs):
		bb.add(0, s, d, "%dM" % len(query), query, qual=qual)
		bb.add(1, s, d, "1S%dM" % (len(query) - 2), query[:-1], qual=qual[:-1] if qual else None)
	batch = bb.build()
	for go, ge in ((0, -1), (-3, -1), (2, -1), (1, -1), (3, -5), (5, -2), (0, -4), (4, 0)):
		for mode in (0, 1):
			got, stats = pvc3.project_batch(ctx, batch, go, ge, mode)
			assert_results_equal(batch, got, o.batch_project(batch, go, ge, mode))
			assert stats["n_failed"] == 0 and stats["n_runs"] >= len(recs)


def test_tiebreak_policies(pvc3, oracle_mod):
	"""pvc3_params::tiebreak_mode 0 .. 11 (which of the co-optimal alignments: the order of preference among equal candidates of
	a cell x whether a gap that could be extended or opened at the same score is extended or opened): every policy against the
	oracle's, both score types, on runs with many co-optimal paths; the scores do not depend on the policy, the paths do."""
	rng = np.random.default_rng(79)
	seg = lambda k: "".join("ACGT"[x] for x in rng.integers(0, 4, size=k))
	pairs, recs = [], []
	for m, n in [(1, 1), (3, 3), (3, 5), (7, 2), (12, 12), (16, 20), (33, 8), (8, 33), (40, 40), (70, 30), (5, 100)]:
		for variant in range(4):
			p, s = seg(int(rng.integers(3, 30))), seg(int(rng.integers(3, 30)))
			q = seg(n)
			r = [seg(m), (q * (m // max(1, n) + 2))[:m], "".join(c if rng.random() < 0.6 else "ACGT"[rng.integers(4)] for c in (q + seg(m))[:m]), ("AC" * m)[:m]][variant]
			if 3 == variant:
				q = ("CA" * n)[:n]
			src, dst, query = p + q + "-" * m + s, p + "-" * n + r + s, p + q + s
			qual = [int(x) for x in rng.choice([2, 12, 23, 37, 41], size=len(query))] if variant != 1 else None
			pairs.append((src, dst, dst.replace("-", "")))
			recs.append((query, qual))
	ctx, o, ids = _edge_index(pvc3, oracle_mod, pairs)
	bb = BatchBuilder()
	for (s, d), (query, qual) in zip(ids, recs):
		bb.add(0, s, d, "%dM" % len(query), query, qual=qual)
	batch = bb.build()
	different = set()
	for go, ge in ((0, -1), (-3, -1), (2, -1)):
		for score_mode in (0, 1):
			base = None
			for tb in range(12):
				got, stats = pvc3.project_batch(ctx, batch, go, ge, score_mode, tiebreak_mode=tb)
				assert_results_equal(batch, got, o.batch_project(batch, go, ge, score_mode, tiebreak_mode=tb))
				assert stats["n_failed"] == 0 and stats["n_runs"] >= len(recs)
				if 0 == tb:
					base = got["out_cigar"].copy()
				elif got["out_cigar"].size != base.size or not np.array_equal(got["out_cigar"], base):
					different.add(tb)
	assert len(different) >= 8		# (the policies are not the same thing under different names)
	with pytest.raises(pvc3.Error):
		pvc3.project_batch(ctx, batch, tiebreak_mode=12)


def test_one_large_alignment_among_many_small(pvc3, oracle_mod):
	"""A run whose alignment needs megabytes of scratch (a 250-base query slice against a 9 kb deletion of the source
	sequence) among many small ones: the scratch grows, but slots x waves stays within the budget instead of multiplying the
	large slot by thousands of waves; the results do not change."""
	rng = np.random.default_rng(78)
	seg = lambda k: "".join("ACGT"[x] for x in rng.integers(0, 4, size=k))
	pairs, recs = [], []
	for i in range(300):
		m, n = (9000, 250) if 17 == i else (int(rng.integers(1, 30)), int(rng.integers(1, 30)))
		p, s, q, r = seg(20), seg(20), seg(n), seg(m)
		pairs.append((p + q + "-" * m + s, p + "-" * n + r + s, p + r + s))
		recs.append(p + q + s)
	ctx, o, ids = _edge_index(pvc3, oracle_mod, pairs)
	ctx.set_option("scratch_budget", 16 << 20)
	ctx.set_option("trace_initial", 1 << 20)		# 4 MB of trace words for the packed kernels (the default is sized for millions of records)
	bb = BatchBuilder()
	for (s, d), query in zip(ids, recs):
		bb.add(0, s, d, "%dM" % len(query), query, qual=[int(x) for x in rng.integers(2, 42, size=len(query))])
	batch = bb.build()
	got, stats = pvc3.project_batch(ctx, batch)
	assert_results_equal(batch, got, o.batch_project(batch))
	assert stats["attempts"] == 2 and stats["n_runs"] == 300
	_, workspace = ctx.hbm_bytes()
	assert workspace <= (16 << 20) + (4 << 20) + (1 << 20)


def test_pipeline_matches_batch_project(pvc3, oracle_mod, synth):
	"""pvc3_pipeline_submit / _wait (uploads, kernels and downloads of consecutive batches overlapping; QUAL fetched from the
	page-locked host arrays for the realigned records only) returns byte for byte what pvc3_batch_project returns, batch by
	batch, in submission order; the same with pageable arrays (QUAL uploaded); full pipeline and capacity errors."""
	gp, rp, _ = synth.config("C5", scale=0.004)
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	batch = reads.batch()
	index = pvc3.Index()
	genome.fill_index(index)
	ctx = pvc3.Context(0)
	ctx.upload_index(index)
	o = oracle_mod.Oracle()
	genome.fill_index(o)
	n, per = reads.n, 20_000
	slices = [synth.slice_batch(batch, lo, min(n, lo + per)) for lo in range(0, n, per)]
	want = [o.batch_project(s, n_threads=8) for s in slices]
	max_cig = max(int(s["cigar"].size) for s in slices)
	max_seq = max(int(s["seq4"].size) for s in slices)
	max_qual = max(int(s["qual"].size) for s in slices)
	for pinned in (True, False):
		depth = 3
		pipe = pvc3.Pipeline(ctx, per, max_cig, max_seq, 0 if pinned else max_qual, depth)
		outs = [pvc3.Pipeline.output_arrays(per, 8 * max_cig + 16 * per, per, pinned=pinned) for _ in range(depth)]
		keep = []
		got = []
		subs = []
		for s in slices:
			if pinned:
				s, own = pvc3.pinned_batch(s)
				keep.append(own)
			subs.append(s)
		for i, s in enumerate(subs):
			if pipe.in_flight() == depth:
				res, st = pipe.wait()
				got.append({k: v.copy() for k, v in res.items()})
			pipe.submit(s, outs[i % depth])
		assert pipe.in_flight() == depth		# three batches in flight: a fourth is refused
		with pytest.raises(pvc3.Error) as ei:
			pipe.submit(subs[0], outs[0])
		assert ei.value.code == -6
		while pipe.in_flight():
			res, st = pipe.wait()
			got.append({k: v.copy() for k, v in res.items()})
		assert len(got) == len(slices)
		for s, g, w in zip(slices, got, want):
			assert_results_equal(s, g, w)
		# a batch beyond the capacities is refused, results larger than the output arrays are reported by wait()
		with pytest.raises(pvc3.Error) as ei:
			pipe.submit(synth.slice_batch(batch, 0, per + 1), outs[0])
		assert ei.value.code == -5
		small = pvc3.Pipeline.output_arrays(per, 10, 1, pinned=False)
		pipe.submit(subs[0], small)
		with pytest.raises(pvc3.Error) as ei:
			pipe.wait()
		assert ei.value.code == -5
		pipe.submit(subs[1], outs[1])		# the pipeline goes on after an error
		res, st = pipe.wait()
		assert_results_equal(slices[1], res, want[1])
		pipe.close()
	ctx.close()


def test_malformed_batches_are_refused(pvc3, oracle_mod):
	"""Offsets that decrease or run past their arrays, and records whose SEQ / QUAL slices are shorter than l_seq, make the
	kernels read outside the batch: such a batch is refused as a whole (PVC3_E_INVALID), also in the pipeline."""
	rng = np.random.default_rng(5)
	text = "".join("ACGT"[x] for x in rng.integers(0, 4, size=500))
	ctx, o, ids = _edge_index(pvc3, oracle_mod, [(text, text, text)])
	s, d = ids[0]
	bb = BatchBuilder()
	for p in range(0, 300, 10):
		bb.add(p, s, d, "100M", text[p:p + 100], qual=[30] * 100)
	good = bb.build()
	got, stats = pvc3.project_batch(ctx, good)
	assert_results_equal(good, got, o.batch_project(good))

	def broken(**changes):
		b = {k: v.copy() for k, v in good.items()}
		for k, (i, v) in changes.items():
			b[k][i] = v
		return b
	cases = [
		broken(cigar_off=(5, 3)),							# decreasing
		broken(cigar_off=(12, 1000)),						# past the array (the last offset itself is the caller's statement of the array length)
		broken(seq_off=(7, int(good["seq_off"][8]) - 1)),	# decreasing start: record 6 loses a byte ... and 7 starts early (fine), 6 is short
		broken(seq_off=(20, 10**9)),
		broken(l_seq=(4, 5000)),							# l_seq promises more SEQ than the record owns
		broken(qual_off=(9, int(good["qual_off"][9]) + 50)),	# record 8 keeps 150 bytes, record 9 only 50
		broken(qual_off=(17, 10**9)),
	]
	for bad in cases:
		with pytest.raises(pvc3.Error) as ei:
			pvc3.project_batch(ctx, bad)
		assert ei.value.code == -1, ei.value
	pipe = pvc3.Pipeline(ctx, 64, 256, 4096, 8192)
	out = pvc3.Pipeline.output_arrays(64, 4096, 64, pinned=False)
	pipe.submit(cases[0], out)
	pipe.submit(good, pvc3.Pipeline.output_arrays(64, 4096, 64, pinned=False))
	with pytest.raises(pvc3.Error) as ei:
		pipe.wait()
	assert ei.value.code == -1
	res, st = pipe.wait()
	assert_results_equal(good, res, o.batch_project(good))
	# The same through page-locked arrays, whose QUAL is not uploaded but fetched by the kernels through the caller's offsets.
	keep = []
	for bad in (cases[5], cases[6], good):
		pb, own = pvc3.pinned_batch(bad)
		keep.append(own)
		pipe.submit(pb, pvc3.Pipeline.output_arrays(64, 4096, 64))
	for _ in range(2):
		with pytest.raises(pvc3.Error) as ei:
			pipe.wait()
		assert ei.value.code == -1
	res, st = pipe.wait()
	assert_results_equal(good, res, o.batch_project(good))
	pipe.close()
