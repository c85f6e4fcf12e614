// synth.cpp — deterministic synthetic workloads for tests and bench.py (SURVEY.md §8d "Synthetic inputs").
// Test / benchmark infrastructure, not product code: builds a random founder-sequence MSA (REF + F founders per contig, with
// SNVs, small indels and complex del+ins events laid out as adjacent src-gap / dst-gap blocks) and samples reads from the
// founders with sequencing errors, producing the SoA batch of include/pvc3.h.
//
// PRNG: splitmix64 -> xoshiro256**; every contig and every read (pair) derives its own stream from (seed, index), so the
// output does not depend on the thread count.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct rng
{
	uint64_t s[4];

	static uint64_t splitmix(uint64_t &x)
	{
		uint64_t z = (x += UINT64_C(0x9E3779B97F4A7C15));
		z = (z ^ (z >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
		z = (z ^ (z >> 27)) * UINT64_C(0x94D049BB133111EB);
		return z ^ (z >> 31);
	}

	rng(uint64_t seed, uint64_t stream)
	{
		uint64_t x = seed ^ (stream * UINT64_C(0xD1342543DE82EF95) + UINT64_C(0x632BE59BD9B4E019));
		for (auto &v : s) v = splitmix(x);
	}

	static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

	uint64_t next()
	{
		uint64_t const result = rotl(s[1] * 5, 7) * 9;
		uint64_t const t = s[1] << 17;
		s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
		s[2] ^= t; s[3] = rotl(s[3], 45);
		return result;
	}

	double uniform() { return (double) (next() >> 11) * (1.0 / 9007199254740992.0); }
	uint64_t below(uint64_t n) { return (uint64_t) ((__uint128_t) next() * n >> 64); }
	uint64_t range(uint64_t lo, uint64_t hi) { return lo + below(hi - lo + 1); }	// inclusive

	// Number of failures before the first success, success probability p.
	uint64_t geometric(double p)
	{
		double const u = 1.0 - uniform();
		return (uint64_t) std::floor(std::log(u) / std::log1p(-p));
	}

	double normal()
	{
		double const u1 = 1.0 - uniform(), u2 = uniform();
		return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
	}
};

char const NUC[4] = {'A', 'C', 'G', 'T'};

struct text_view
{
	char const *p{};
	uint64_t n{};
	uint64_t size() const { return n; }
	char operator[](uint64_t i) const { return p[i]; }
};

struct row
{
	std::vector <uint64_t> gap_words;	// linear: bit (c & 63) of word (c >> 6) = 1 iff column c is a gap
	std::string text;					// unaligned characters
	// What the readers use: the two above, or the same arrays inside a file mapped by synth_genome_map (then the two above are empty).
	uint64_t const *gw{};
	uint64_t n_gw{};
	text_view tv;
	void finish() { gw = gap_words.data(); n_gw = gap_words.size(); tv = text_view{text.data(), text.size()}; }
};

struct contig
{
	uint64_t n_cols{};
	std::vector <row> rows;				// rows[0] = REF, rows[1 + f] = founder f
};

}	// anonymous namespace

extern "C" {

struct synth_genome_params
{
	uint64_t seed;
	uint64_t genome_len;		// total REF bases over all contigs
	uint32_t n_contigs;
	uint32_t n_founders;
	double snv_rate;			// per REF base
	double indel_rate;			// small insertion or deletion, length 1 + geometric (mean 3)
	double complex_rate;		// deletion of k + insertion of m at the same place, k, m in [1, complex_max]
	uint32_t complex_max;
	double long_rate;			// long complex events (config C5), k or m in [long_min, long_max]
	uint32_t long_min, long_max;
	double carrier_prob;		// each founder carries each ALT with this probability
	uint32_t lowercase_every;	// if non-zero, every n-th REF base is written in lower case (soft-masked reference)
	double indel_mean_len;		// mean of the geometric part of an indel's length (0: 3)
	uint32_t indel_max_len;		// its cap (0: 40)
};

struct synth_reads_params
{
	uint64_t seed;
	uint64_t n_records;			// paired: 2 records per pair
	uint32_t read_len;
	uint32_t paired;
	double insert_mean, insert_sd;
	double sub_rate;			// sequencing substitutions per base
	double indel_read_frac;		// reads with one 1-3 bp insertion or deletion
	double clip_read_frac;		// reads with a 1-10 bp soft clip at one end
	uint32_t with_qual;
	double n_base_rate;			// bases replaced by N
};

struct synth_genome
{
	std::vector <contig> contigs;
	uint32_t n_founders{};
	void *mapping{};			// synth_genome_map: the file's pages (shared by every process that maps it)
	uint64_t mapping_bytes{};
};

struct synth_reads
{
	std::vector <uint32_t> src_pos, src_entry, dst_entry, flags, cigar, l_seq, mate_src_entry, mate_dst_entry, mate_pos;
	std::vector <uint64_t> cigar_off, seq_off, qual_off;
	std::vector <uint8_t> seq4, qual;
};

static void set_gap(row &r, uint64_t col) { r.gap_words[col >> 6] |= UINT64_C(1) << (col & 63); }

namespace {

struct event
{
	uint64_t ref_pos;		// first REF base affected (deleted bases / SNV) or the base the insertion precedes
	uint32_t del_len;		// REF bases absent from the carriers
	uint32_t ins_len;		// inserted bases present in the carriers only
	uint32_t carriers;		// bit f = founder f carries the ALT
	uint64_t ins_seed;
	uint8_t is_snv;
	uint8_t alt_shift;		// 1..3
};

void build_contig(synth_genome_params const &p, uint32_t ci, uint64_t len, contig &c)
{
	rng g(p.seed, 0x1000 + ci);
	uint32_t const F = p.n_founders;

	// 1. REF bases.
	std::string ref(len, 'A');
	for (uint64_t i = 0; i < len; i += 32)
	{
		uint64_t bits = g.next();
		uint64_t const lim = std::min <uint64_t>(32, len - i);
		for (uint64_t k = 0; k < lim; ++k, bits >>= 2)
			ref[i + k] = NUC[bits & 3];
	}

	// 2. Events, left to right, non-overlapping.
	std::vector <event> events;
	double const total_rate = p.snv_rate + p.indel_rate + p.complex_rate + p.long_rate;
	uint64_t pos = 0;
	uint64_t extra_cols = 0;
	while (total_rate > 0)
	{
		pos += g.geometric(total_rate);
		if (pos + 1 >= len) break;
		event e{};
		e.ref_pos = pos;
		for (uint32_t f = 0; f < F; ++f)
			if (g.uniform() < p.carrier_prob) e.carriers |= 1u << f;
		e.ins_seed = g.next();
		double const u = g.uniform() * total_rate;
		if (u < p.snv_rate)
		{
			e.is_snv = 1; e.alt_shift = (uint8_t) g.range(1, 3);
			e.del_len = 1;	// occupies one base
		}
		else if (u < p.snv_rate + p.indel_rate)
		{
			double const mean = (p.indel_mean_len > 0.0) ? p.indel_mean_len : 3.0;
			uint32_t const cap = p.indel_max_len ? p.indel_max_len : 40;
			uint32_t const l = 1 + (uint32_t) std::min <uint64_t>(g.geometric(1.0 / mean), cap);
			if (g.next() & 1) e.ins_len = l; else e.del_len = l;
		}
		else if (u < p.snv_rate + p.indel_rate + p.complex_rate)
		{
			e.del_len = (uint32_t) g.range(1, p.complex_max);
			e.ins_len = (uint32_t) g.range(1, p.complex_max);
		}
		else
		{
			// long event: at least one side long
			uint32_t const a = (uint32_t) g.range(p.long_min, p.long_max), b = (uint32_t) g.range(1, p.long_max);
			if (g.next() & 1) { e.del_len = a; e.ins_len = b; } else { e.del_len = b; e.ins_len = a; }
		}
		if (e.ref_pos + e.del_len >= len) break;
		if (e.ins_len && 0 == e.carriers) e.ins_len = 0;		// nobody carries the insertion: no all-gap columns
		if (0 == e.ins_len && 0 == e.del_len) { pos += 1; continue; }
		extra_cols += e.ins_len;
		events.push_back(e);
		pos = e.ref_pos + std::max <uint32_t>(e.del_len, 1) + 1;	// keep one untouched base between events
	}

	// 3. Rows.
	c.n_cols = len + extra_cols;
	uint64_t const n_words = (c.n_cols + 63) / 64;
	c.rows.resize(1 + F);
	for (auto &r : c.rows) { r.gap_words.assign(n_words + 1, 0); r.text.reserve(len + extra_cols / 2 + 16); }
	row &R = c.rows[0];
	uint64_t col = 0, rp = 0;
	auto copy_plain = [&](uint64_t upto) {	// REF bases [rp, upto) are shared by everybody
		if (upto <= rp) return;
		uint64_t const n = upto - rp;
		R.text.append(ref, rp, n);
		for (uint32_t f = 0; f < F; ++f) c.rows[1 + f].text.append(ref, rp, n);
		col += n; rp = upto;
	};
	for (event const &e : events)
	{
		copy_plain(e.ref_pos);
		if (e.is_snv)
		{
			char const rc = ref[rp];
			uint32_t ri = 0;
			while (NUC[ri] != rc) ++ri;
			R.text.push_back(rc);
			for (uint32_t f = 0; f < F; ++f)
				c.rows[1 + f].text.push_back((e.carriers >> f & 1) ? NUC[(ri + e.alt_shift) & 3] : rc);
			++col; ++rp;
			continue;
		}
		// REF block: del_len columns, gaps in the carriers.
		for (uint32_t k = 0; k < e.del_len; ++k)
		{
			char const rc = ref[rp];
			R.text.push_back(rc);
			for (uint32_t f = 0; f < F; ++f)
			{
				if (e.carriers >> f & 1) set_gap(c.rows[1 + f], col);
				else c.rows[1 + f].text.push_back(rc);
			}
			++col; ++rp;
		}
		// ALT block: ins_len columns, gap in REF and in the non-carriers.
		if (e.ins_len)
		{
			uint64_t x = e.ins_seed;
			for (uint32_t k = 0; k < e.ins_len; ++k)
			{
				char const ic = NUC[rng::splitmix(x) & 3];
				set_gap(R, col);
				for (uint32_t f = 0; f < F; ++f)
				{
					if (e.carriers >> f & 1) c.rows[1 + f].text.push_back(ic);
					else set_gap(c.rows[1 + f], col);
				}
				++col;
			}
		}
	}
	copy_plain(len);
	if (p.lowercase_every)
		for (uint64_t i = 0; i < R.text.size(); i += p.lowercase_every)
			R.text[i] = (char) (R.text[i] | 0x20);
}

}	// anonymous namespace

synth_genome *synth_genome_create(synth_genome_params const *p, int n_threads)
{
	synth_genome *g = new synth_genome;
	g->n_founders = p->n_founders;
	g->contigs.resize(p->n_contigs);
	std::vector <uint64_t> lens(p->n_contigs);
	for (uint32_t i = 0; i < p->n_contigs; ++i)
		lens[i] = p->genome_len * (i + 1) / p->n_contigs - p->genome_len * i / p->n_contigs;
	if (n_threads < 1) n_threads = 1;
	std::vector <std::thread> th;
	for (int t = 0; t < n_threads; ++t)
		th.emplace_back([&, t] {
			for (uint32_t i = (uint32_t) t; i < p->n_contigs; i += (uint32_t) n_threads)
				build_contig(*p, i, lens[i], g->contigs[i]);
		});
	for (auto &x : th) x.join();
	for (auto &c : g->contigs) for (auto &r : c.rows) r.finish();
	return g;
}

// One contig from caller-supplied aligned rows (rows[0] = REF), e.g. the reference's shipped test MSA.
synth_genome *synth_genome_from_aligned(char const *const *rows, uint32_t n_rows, uint64_t n_cols)
{
	synth_genome *g = new synth_genome;
	g->n_founders = n_rows - 1;
	g->contigs.resize(1);
	contig &c = g->contigs[0];
	c.n_cols = n_cols;
	c.rows.resize(n_rows);
	for (uint32_t r = 0; r < n_rows; ++r)
	{
		c.rows[r].gap_words.assign((n_cols + 63) / 64 + 1, 0);
		for (uint64_t i = 0; i < n_cols; ++i)
		{
			if ('-' == rows[r][i]) set_gap(c.rows[r], i);
			else c.rows[r].text.push_back(rows[r][i]);
		}
		c.rows[r].finish();
	}
	return g;
}

void synth_genome_free(synth_genome *g)
{
	if (g && g->mapping) munmap(g->mapping, g->mapping_bytes);
	delete g;
}

// The genome as one file (bench.py with several ranks on a node: one rank generates it into /dev/shm, the others map the same
// pages instead of generating or holding copies of their own).  Layout: u64 magic, n_contigs, n_founders; per contig u64 n_cols;
// per row u64 n_gap_words, n_text; then the arrays, each at a multiple of 64 bytes, in the same order.
static uint64_t const SYNTH_MAGIC = UINT64_C(0x31474E5933435650);	// "PVC3YNG1"

int synth_genome_save(synth_genome const *g, char const *path)
{
	FILE *f = std::fopen(path, "wb");
	if (!f) return -1;
	std::vector <uint64_t> head{SYNTH_MAGIC, (uint64_t) g->contigs.size(), (uint64_t) g->n_founders};
	for (auto const &c : g->contigs)
	{
		head.push_back(c.n_cols);
		for (auto const &r : c.rows) { head.push_back(r.n_gw); head.push_back(r.tv.n); }
	}
	bool ok = head.size() == std::fwrite(head.data(), 8, head.size(), f);
	uint64_t at = 8 * head.size();
	static char const zeros[64] = {0};
	auto put = [&](void const *p, uint64_t n) {
		uint64_t const pad = (64 - at % 64) % 64;
		ok = ok && pad == std::fwrite(zeros, 1, pad, f) && n == std::fwrite(p, 1, n, f);
		at += pad + n;
	};
	for (auto const &c : g->contigs)
		for (auto const &r : c.rows) { put(r.gw, 8 * r.n_gw); put(r.tv.p, r.tv.n); }
	ok = (0 == std::fclose(f)) && ok;
	return ok ? 0 : -1;
}

synth_genome *synth_genome_map(char const *path)
{
	int const fd = open(path, O_RDONLY);
	if (fd < 0) return nullptr;
	struct stat st{};
	if (fstat(fd, &st) || st.st_size < 24) { close(fd); return nullptr; }
	void *m = mmap(nullptr, (size_t) st.st_size, PROT_READ, MAP_SHARED, fd, 0);
	close(fd);
	if (MAP_FAILED == m) return nullptr;
	uint64_t const *h = (uint64_t const *) m;
	if (SYNTH_MAGIC != h[0]) { munmap(m, (size_t) st.st_size); return nullptr; }
	synth_genome *g = new synth_genome;
	g->mapping = m; g->mapping_bytes = (uint64_t) st.st_size;
	g->contigs.resize(h[1]);
	g->n_founders = (uint32_t) h[2];
	uint64_t k = 3;
	for (auto &c : g->contigs)
	{
		c.n_cols = h[k++];
		c.rows.resize(1 + g->n_founders);
		for (auto &r : c.rows) { r.n_gw = h[k++]; r.tv.n = h[k++]; }
	}
	uint64_t at = 8 * k;
	for (auto &c : g->contigs)
		for (auto &r : c.rows)
		{
			at = (at + 63) / 64 * 64; r.gw = (uint64_t const *) ((char const *) m + at); at += 8 * r.n_gw;
			at = (at + 63) / 64 * 64; r.tv.p = (char const *) m + at; at += r.tv.n;
		}
	if (at > g->mapping_bytes) { synth_genome_free(g); return nullptr; }
	return g;
}
uint32_t synth_genome_entry_count(synth_genome const *g) { return (uint32_t) (g->contigs.size() * (1 + g->n_founders)); }
uint32_t synth_genome_founders(synth_genome const *g) { return g->n_founders; }
uint32_t synth_genome_contigs(synth_genome const *g) { return (uint32_t) g->contigs.size(); }

// Entry e = contig * (1 + F) + row; row 0 is REF (the destination), rows 1..F are the founders.
void synth_genome_entry(synth_genome const *g, uint32_t e, uint64_t const **gap_words, uint64_t *n_cols, char const **text, uint64_t *text_len)
{
	uint32_t const per = 1 + g->n_founders;
	contig const &c = g->contigs[e / per];
	row const &r = c.rows[e % per];
	*gap_words = r.gw; *n_cols = c.n_cols; *text = r.tv.p; *text_len = r.tv.n;
}

namespace {

uint8_t code_of(char c)
{
	switch (c) { case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'G': case 'g': return 4; case 'T': case 't': return 8; default: return 15; }
}

struct read_out
{
	uint32_t src_pos, src_entry, dst_entry, flags, l_seq, mate_src_entry, mate_dst_entry, mate_pos;
	uint32_t cigar[6]; uint32_t n_cigar;
	std::vector <uint8_t> codes, qual;
};

// One read of length L starting at founder position `start`.
void make_read(synth_genome const &g, synth_reads_params const &p, rng &r, uint32_t contig_i, uint32_t founder, uint64_t start, read_out &out)
{
	uint32_t const per = 1 + g.n_founders;
	text_view const text = g.contigs[contig_i].rows[1 + founder].tv;
	uint32_t const L = p.read_len;
	out.src_entry = contig_i * per + 1 + founder;
	out.dst_entry = contig_i * per;
	out.l_seq = L;
	out.codes.clear(); out.qual.clear();
	out.n_cigar = 0;

	uint32_t clip_l = 0, clip_r = 0;
	if (r.uniform() < p.clip_read_frac)
	{
		uint32_t const n = (uint32_t) r.range(1, 10);
		if (r.next() & 1) clip_l = n; else clip_r = n;
	}
	uint32_t const body = L - clip_l - clip_r;
	int indel = 0;	// +k insertion, -k deletion
	uint32_t indel_at = 0;
	if (r.uniform() < p.indel_read_frac && body > 20)
	{
		uint32_t const k = (uint32_t) r.range(1, 3);
		indel = (r.next() & 1) ? (int) k : -(int) k;
		indel_at = (uint32_t) r.range(5, body - 5 - (indel > 0 ? k : 0));
	}

	auto push_cigar = [&](uint32_t op, uint32_t n) { if (n) out.cigar[out.n_cigar++] = (n << 4) | op; };
	auto random_base = [&] { out.codes.push_back(code_of(NUC[r.next() & 3])); };
	uint64_t fp = start;
	auto founder_base = [&] {
		char c = (fp < text.size()) ? text[fp] : 'A';
		++fp;
		uint8_t code = code_of(c);
		if (r.uniform() < p.sub_rate)
		{
			uint32_t ni = 0;
			while (ni < 4 && NUC[ni] != (c & ~0x20)) ++ni;
			code = code_of(NUC[(ni + 1 + r.below(3)) & 3]);
		}
		if (p.n_base_rate > 0 && r.uniform() < p.n_base_rate) code = 15;
		out.codes.push_back(code);
	};

	out.src_pos = (uint32_t) start;
	push_cigar(4, clip_l);
	for (uint32_t i = 0; i < clip_l; ++i) random_base();
	if (0 == indel)
	{
		push_cigar(0, body);
		for (uint32_t i = 0; i < body; ++i) founder_base();
	}
	else if (indel > 0)
	{
		uint32_t const k = (uint32_t) indel;
		push_cigar(0, indel_at); push_cigar(1, k); push_cigar(0, body - indel_at - k);
		for (uint32_t i = 0; i < indel_at; ++i) founder_base();
		for (uint32_t i = 0; i < k; ++i) random_base();
		for (uint32_t i = 0; i < body - indel_at - k; ++i) founder_base();
	}
	else
	{
		uint32_t const k = (uint32_t) -indel;
		push_cigar(0, indel_at); push_cigar(2, k); push_cigar(0, body - indel_at);
		for (uint32_t i = 0; i < indel_at; ++i) founder_base();
		fp += k;
		for (uint32_t i = 0; i < body - indel_at; ++i) founder_base();
	}
	push_cigar(4, clip_r);
	for (uint32_t i = 0; i < clip_r; ++i) random_base();

	out.flags = 0;
	if (p.with_qual)
	{
		out.flags |= 1;
		static uint8_t const bins[4] = {2, 12, 23, 37};
		for (uint32_t i = 0; i < L; ++i)
		{
			double const u = r.uniform();
			out.qual.push_back(bins[u < 0.02 ? 0 : u < 0.05 ? 1 : u < 0.15 ? 2 : 3]);
		}
	}
}

}	// anonymous namespace

synth_reads *synth_reads_create(synth_genome const *g, synth_reads_params const *pp, int n_threads)
{
	synth_reads_params const p = *pp;
	uint64_t const n = p.n_records;
	uint32_t const L = p.read_len;
	synth_reads *R = new synth_reads;
	if (n_threads < 1) n_threads = 1;
	uint64_t const n_units = p.paired ? (n + 1) / 2 : n;

	// Contig choice proportional to founder-0 length.
	std::vector <uint64_t> cum(g->contigs.size() + 1, 0);
	for (size_t i = 0; i < g->contigs.size(); ++i)
		cum[i + 1] = cum[i] + g->contigs[i].rows[1].tv.size();

	// Per-thread contiguous unit ranges; each thread writes thread-local vectors that are concatenated in order.
	struct tl { synth_reads part; };
	std::vector <tl> parts(n_threads);
	std::vector <std::thread> th;
	for (int t = 0; t < n_threads; ++t)
		th.emplace_back([&, t] {
			synth_reads &P = parts[t].part;
			uint64_t const u0 = n_units * (uint64_t) t / (uint64_t) n_threads, u1 = n_units * (uint64_t) (t + 1) / (uint64_t) n_threads;
			read_out ro[2];
			for (uint64_t u = u0; u < u1; ++u)
			{
				rng r(p.seed, 0x100000 + u);
				// contig, founder, start
				uint64_t const x = r.below(cum.back());
				uint32_t ci = (uint32_t) (std::upper_bound(cum.begin(), cum.end(), x) - cum.begin() - 1);
				uint32_t const f = (uint32_t) r.below(g->n_founders);
				uint64_t const flen = g->contigs[ci].rows[1 + f].tv.size();
				uint32_t n_here = 1;
				uint64_t span = L + 8;
				uint64_t insert = 0;
				if (p.paired)
				{
					double ins = p.insert_mean + p.insert_sd * r.normal();
					if (ins < L + 10) ins = L + 10;
					if (ins > p.insert_mean + 4 * p.insert_sd) ins = p.insert_mean + 4 * p.insert_sd;
					insert = (uint64_t) ins;
					span = insert + 8;
					n_here = (2 * u + 1 < n) ? 2 : 1;
				}
				uint64_t const start = (flen > span + 1) ? r.below(flen - span) : 0;
				make_read(*g, p, r, ci, f, start, ro[0]);
				if (p.paired)
				{
					uint64_t const mate_start = start + insert - L;
					if (2 == n_here) make_read(*g, p, r, ci, f, mate_start, ro[1]);
					ro[0].flags |= 2; ro[0].mate_src_entry = ro[0].src_entry; ro[0].mate_dst_entry = ro[0].dst_entry; ro[0].mate_pos = (uint32_t) mate_start;
					if (2 == n_here)
					{
						// Soft clipping shifts nothing: src_pos is the first aligned base in both reads.
						ro[1].flags |= 2; ro[1].mate_src_entry = ro[0].src_entry; ro[1].mate_dst_entry = ro[0].dst_entry; ro[1].mate_pos = ro[0].src_pos;
						ro[0].mate_pos = ro[1].src_pos;
					}
				}
				else
				{
					ro[0].mate_src_entry = 0; ro[0].mate_dst_entry = 0; ro[0].mate_pos = 0;
				}
				for (uint32_t k = 0; k < n_here; ++k)
				{
					read_out const &o = ro[k];
					P.src_pos.push_back(o.src_pos); P.src_entry.push_back(o.src_entry); P.dst_entry.push_back(o.dst_entry);
					P.flags.push_back(o.flags); P.l_seq.push_back(o.l_seq);
					P.mate_src_entry.push_back(o.mate_src_entry); P.mate_dst_entry.push_back(o.mate_dst_entry); P.mate_pos.push_back(o.mate_pos);
					P.cigar_off.push_back(P.cigar.size());
					P.cigar.insert(P.cigar.end(), o.cigar, o.cigar + o.n_cigar);
					P.seq_off.push_back(P.seq4.size());
					for (size_t i = 0; i < o.codes.size(); i += 2)
						P.seq4.push_back((uint8_t) ((o.codes[i] << 4) | ((i + 1 < o.codes.size()) ? o.codes[i + 1] : 0)));
					P.qual_off.push_back(P.qual.size());
					P.qual.insert(P.qual.end(), o.qual.begin(), o.qual.end());
				}
			}
		});
	for (auto &x : th) x.join();

	// Concatenate.
	uint64_t tot_n = 0, tot_c = 0, tot_s = 0, tot_q = 0;
	for (auto const &pt : parts) { tot_n += pt.part.src_pos.size(); tot_c += pt.part.cigar.size(); tot_s += pt.part.seq4.size(); tot_q += pt.part.qual.size(); }
	R->src_pos.reserve(tot_n); R->src_entry.reserve(tot_n); R->dst_entry.reserve(tot_n); R->flags.reserve(tot_n); R->l_seq.reserve(tot_n);
	R->mate_src_entry.reserve(tot_n); R->mate_dst_entry.reserve(tot_n); R->mate_pos.reserve(tot_n);
	R->cigar_off.reserve(tot_n + 1); R->seq_off.reserve(tot_n + 1); R->qual_off.reserve(tot_n + 1);
	R->cigar.reserve(tot_c); R->seq4.reserve(tot_s + 8); R->qual.reserve(tot_q);
	for (auto &pt : parts)
	{
		synth_reads &P = pt.part;
		uint64_t const c0 = R->cigar.size(), s0 = R->seq4.size(), q0 = R->qual.size();
		R->src_pos.insert(R->src_pos.end(), P.src_pos.begin(), P.src_pos.end());
		R->src_entry.insert(R->src_entry.end(), P.src_entry.begin(), P.src_entry.end());
		R->dst_entry.insert(R->dst_entry.end(), P.dst_entry.begin(), P.dst_entry.end());
		R->flags.insert(R->flags.end(), P.flags.begin(), P.flags.end());
		R->l_seq.insert(R->l_seq.end(), P.l_seq.begin(), P.l_seq.end());
		R->mate_src_entry.insert(R->mate_src_entry.end(), P.mate_src_entry.begin(), P.mate_src_entry.end());
		R->mate_dst_entry.insert(R->mate_dst_entry.end(), P.mate_dst_entry.begin(), P.mate_dst_entry.end());
		R->mate_pos.insert(R->mate_pos.end(), P.mate_pos.begin(), P.mate_pos.end());
		for (uint64_t v : P.cigar_off) R->cigar_off.push_back(c0 + v);
		for (uint64_t v : P.seq_off) R->seq_off.push_back(s0 + v);
		for (uint64_t v : P.qual_off) R->qual_off.push_back(q0 + v);
		R->cigar.insert(R->cigar.end(), P.cigar.begin(), P.cigar.end());
		R->seq4.insert(R->seq4.end(), P.seq4.begin(), P.seq4.end());
		R->qual.insert(R->qual.end(), P.qual.begin(), P.qual.end());
		P = synth_reads();
	}
	R->cigar_off.push_back(R->cigar.size());
	R->seq_off.push_back(R->seq4.size());
	R->qual_off.push_back(R->qual.size());
	return R;
}

void synth_reads_free(synth_reads *r) { delete r; }
uint64_t synth_reads_count(synth_reads const *r) { return r->src_pos.size(); }

// which: 0 src_pos 1 src_entry 2 dst_entry 3 flags 4 cigar_off 5 cigar 6 seq_off 7 l_seq 8 seq4 9 qual_off 10 qual
//        11 mate_src_entry 12 mate_dst_entry 13 mate_pos
void const *synth_reads_array(synth_reads const *r, int which, uint64_t *count)
{
#define RET(v) do { *count = (v).size(); return (v).data(); } while (0)
	switch (which)
	{
		case 0: RET(r->src_pos); case 1: RET(r->src_entry); case 2: RET(r->dst_entry); case 3: RET(r->flags);
		case 4: RET(r->cigar_off); case 5: RET(r->cigar); case 6: RET(r->seq_off); case 7: RET(r->l_seq); case 8: RET(r->seq4);
		case 9: RET(r->qual_off); case 10: RET(r->qual); case 11: RET(r->mate_src_entry); case 12: RET(r->mate_dst_entry);
		case 13: RET(r->mate_pos);
		default: *count = 0; return nullptr;
	}
#undef RET
}

}	// extern "C"
