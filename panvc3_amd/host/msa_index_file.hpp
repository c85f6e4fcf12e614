// msa_index_file.hpp — on-disk MSA index of this implementation.
//
// The reference serialises {chr_id, [{seq_id, sdsl::rrr_vector<15>, rank_0, select_0}]} with cereal
// (include/panvc3/msa_index.hh:158-192); neither library is available here and the format cannot be verified offline, so
// panvc3_index_msa writes its own format: plain gap bit words (bit = 1 iff the column is a gap), which is what the GPU
// index is built from.  Entries are sorted by id like the reference's (index-msa/input_processor.cc:189,198).
//
//   "PVC3IDX\1"  u32 n_chr  { str chr_id  u32 n_seq  { str seq_id  u64 n_cols  u64 words[ceil(n_cols / 64)] } }
//   str = u32 length + bytes; little endian.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bgzf.hpp"

#include <zlib.h>

namespace pvc3host {

struct msa_sequence_entry
{
	std::string seq_id;
	uint64_t n_cols{};
	std::vector <uint64_t> gap_words;
};

struct msa_chr_entry
{
	std::string chr_id;
	std::vector <msa_sequence_entry> sequence_entries;
};

struct msa_index_file
{
	std::vector <msa_chr_entry> chr_entries;

	void sort()
	{
		std::sort(chr_entries.begin(), chr_entries.end(), [](auto const &a, auto const &b) { return a.chr_id < b.chr_id; });
		for (auto &c : chr_entries)
			std::sort(c.sequence_entries.begin(), c.sequence_entries.end(), [](auto const &a, auto const &b) { return a.seq_id < b.seq_id; });
	}

	msa_chr_entry const *find_chr(std::string_view id) const
	{
		auto const it = std::lower_bound(chr_entries.begin(), chr_entries.end(), id, [](auto const &c, std::string_view v) { return c.chr_id < v; });
		return (chr_entries.end() != it && it->chr_id == id) ? &*it : nullptr;
	}

	static msa_sequence_entry const *find_seq(msa_chr_entry const &c, std::string_view id)
	{
		auto const it = std::lower_bound(c.sequence_entries.begin(), c.sequence_entries.end(), id, [](auto const &s, std::string_view v) { return s.seq_id < v; });
		return (c.sequence_entries.end() != it && it->seq_id == id) ? &*it : nullptr;
	}

	void save(char const *path) const
	{
		FILE *f = std::fopen(path, "wb");
		if (!f) throw io_error(std::string("unable to open ") + path + " for writing");
		auto w = [&](void const *p, std::size_t n) { if (n && 1 != std::fwrite(p, n, 1, f)) { std::fclose(f); throw io_error("write error"); } };
		auto wstr = [&](std::string const &s) { uint32_t const l = (uint32_t) s.size(); w(&l, 4); w(s.data(), s.size()); };
		w("PVC3IDX\1", 8);
		uint32_t const n_chr = (uint32_t) chr_entries.size();
		w(&n_chr, 4);
		for (auto const &c : chr_entries)
		{
			wstr(c.chr_id);
			uint32_t const n_seq = (uint32_t) c.sequence_entries.size();
			w(&n_seq, 4);
			for (auto const &s : c.sequence_entries)
			{
				wstr(s.seq_id);
				w(&s.n_cols, 8);
				w(s.gap_words.data(), 8 * ((s.n_cols + 63) / 64));
			}
		}
		if (0 != std::fclose(f)) throw io_error("write error");
	}

	void load(char const *path)
	{
		FILE *f = std::fopen(path, "rb");
		if (!f) throw io_error(std::string("unable to open ") + path);
		auto r = [&](void *p, std::size_t n) { if (n && 1 != std::fread(p, n, 1, f)) { std::fclose(f); throw io_error(std::string("truncated MSA index ") + path); } };
		auto rstr = [&]() { uint32_t l; r(&l, 4); std::string s(l, '\0'); r(s.data(), l); return s; };
		char magic[8];
		r(magic, 8);
		if (0 != std::memcmp(magic, "PVC3IDX\1", 8))
		{
			std::fclose(f);
			throw io_error(std::string(path) + " is not an MSA index written by this panvc3_index_msa (the reference's cereal / sdsl-lite archives cannot be read; rebuild the index with this tool)");
		}
		uint32_t n_chr; r(&n_chr, 4);
		chr_entries.assign(n_chr, {});
		for (auto &c : chr_entries)
		{
			c.chr_id = rstr();
			uint32_t n_seq; r(&n_seq, 4);
			c.sequence_entries.assign(n_seq, {});
			for (auto &s : c.sequence_entries)
			{
				s.seq_id = rstr();
				r(&s.n_cols, 8);
				s.gap_words.assign((s.n_cols + 63) / 64, 0);
				r(s.gap_words.data(), 8 * s.gap_words.size());
			}
		}
		std::fclose(f);
	}
};

// Whole FASTA (plain, gzip or bgzip) into memory: name (up to the first white space) -> raw sequence bytes, case preserved
// (the reference reads the same bytes through .gzi / .fai random access, libpanvc3/compressed_fasta_reader.cc:94-143).
// ---- random access to the contigs of a (BGZF-compressed) FASTA through its .fai and .gzi indexes ----------------------------
// Replaces panvc3::compressed_fasta_reader (libpanvc3/compressed_fasta_reader.cc:94-248): the reference opens
// <fasta>.gzi and <fasta>.fai and inflates, per contig, from the BGZF block that holds the contig's first base.  Here the
// blocks that cover a contig are inflated in parallel.  Line terminators are dropped; bytes are kept as they are (case matters
// downstream, quirk Q3).
struct fai_entry { std::string name; uint64_t length{}, offset{}, line_bases{}, line_width{}; };

inline bool file_exists(std::string const &p) { FILE *f = std::fopen(p.c_str(), "rb"); if (f) std::fclose(f); return nullptr != f; }

inline std::vector <fai_entry> read_fai(std::string const &path)
{
	std::vector <fai_entry> out;
	byte_reader in;
	in.open(path.c_str());
	std::string line;
	while (in.getline(line))
	{
		if (line.empty()) continue;
		std::vector <std::string> f(1);
		for (char c : line) { if ('\t' == c) f.emplace_back(); else f.back().push_back(c); }
		if (f.size() < 5) throw io_error("malformed .fai line: " + line);
		fai_entry e;
		e.name = f[0]; e.length = std::stoull(f[1]); e.offset = std::stoull(f[2]); e.line_bases = std::stoull(f[3]); e.line_width = std::stoull(f[4]);
		if (0 == e.line_bases || e.line_width < e.line_bases) throw io_error("malformed .fai line: " + line);
		out.push_back(std::move(e));
	}
	return out;
}

// Bytes [begin, end) of the uncompressed FASTA.
class fasta_random_access
{
	FILE *m_file{};
	bool m_bgzf{};
	std::vector <std::pair <uint64_t, uint64_t>> m_blocks;		// (compressed offset, uncompressed offset) of every BGZF block, ascending
	unsigned m_threads{1};

public:
	fasta_random_access(std::string const &path, unsigned threads): m_threads(threads ? threads : 1)
	{
		m_file = std::fopen(path.c_str(), "rb");
		if (!m_file) throw io_error("unable to open " + path);
		uint8_t magic[4]{};
		std::size_t const got = std::fread(magic, 1, 4, m_file);
		m_bgzf = (4 == got && 0x1f == magic[0] && 0x8b == magic[1] && 8 == magic[2] && (magic[3] & 4));
		if (4 == got && 0x1f == magic[0] && 0x8b == magic[1] && !m_bgzf) throw io_error("plain gzip (not BGZF) cannot be read at random: " + path);
		if (m_bgzf)
		{
			// <fasta>.gzi: u64 n, then n pairs (compressed offset, uncompressed offset) of the blocks after the first
			FILE *g = std::fopen((path + ".gzi").c_str(), "rb");
			if (!g) throw io_error("unable to open " + path + ".gzi");
			uint64_t n = 0;
			if (1 != std::fread(&n, 8, 1, g)) { std::fclose(g); throw io_error("truncated .gzi"); }
			m_blocks.assign(1, {0, 0});
			for (uint64_t i = 0; i < n; ++i)
			{
				uint64_t pr[2];
				if (2 != std::fread(pr, 8, 2, g)) { std::fclose(g); throw io_error("truncated .gzi"); }
				m_blocks.emplace_back(pr[0], pr[1]);
			}
			std::fclose(g);
		}
	}
	fasta_random_access(fasta_random_access const &) = delete;
	~fasta_random_access() { if (m_file) std::fclose(m_file); }

	void read(uint64_t begin, uint64_t end, std::vector <char> &dst)
	{
		dst.clear();
		if (end <= begin) return;
		if (!m_bgzf)
		{
			dst.resize(end - begin);
			if (0 != fseeko(m_file, (off_t) begin, SEEK_SET) || dst.size() != std::fread(dst.data(), 1, dst.size(), m_file)) throw io_error("FASTA shorter than its .fai says");
			return;
		}
		// blocks b0 .. b1 cover [begin, end)
		auto const cmp = [](uint64_t v, std::pair <uint64_t, uint64_t> const &b) { return v < b.second; };
		std::size_t const b0 = (std::size_t) (std::upper_bound(m_blocks.begin(), m_blocks.end(), begin, cmp) - m_blocks.begin()) - 1;
		std::size_t b1 = (std::size_t) (std::upper_bound(m_blocks.begin(), m_blocks.end(), end - 1, cmp) - m_blocks.begin()) - 1;
		uint64_t const c_begin = m_blocks[b0].first;
		std::vector <uint8_t> comp;
		if (b1 + 1 < m_blocks.size()) comp.resize(m_blocks[b1 + 1].first - c_begin);
		else
		{
			if (0 != fseeko(m_file, 0, SEEK_END)) throw io_error("seek error");
			comp.resize((uint64_t) ftello(m_file) - c_begin);
		}
		if (0 != fseeko(m_file, (off_t) c_begin, SEEK_SET) || comp.size() != std::fread(comp.data(), 1, comp.size(), m_file)) throw io_error("read error");
		std::size_t const nb = b1 - b0 + 1;
		std::vector <std::vector <uint8_t>> plain(nb);
		parallel_for(nb, m_threads, [&](std::size_t i) {
			uint8_t const *p = comp.data() + (m_blocks[b0 + i].first - c_begin);
			std::size_t const avail = comp.size() - (m_blocks[b0 + i].first - c_begin);
			if (avail < 18 + 8) throw io_error("corrupt BGZF block");
			std::size_t data_off = 0;
			std::size_t const bsize = byte_reader::bgzf_block_size(p, avail, data_off);		// walks the extra field: BC need not be its first subfield
			std::size_t const xlen = data_off - 12;
			if (0 == bsize || bsize > avail || bsize < 12 + xlen + 8) throw io_error("corrupt BGZF block");
			uint32_t isize; std::memcpy(&isize, p + bsize - 4, 4);
			plain[i].resize(isize);
			z_stream zs{};
			if (Z_OK != inflateInit2(&zs, -15)) throw io_error("inflateInit2 failed");
			zs.next_in = const_cast <uint8_t *>(p + 12 + xlen); zs.avail_in = (uInt) (bsize - 12 - xlen - 8);
			zs.next_out = plain[i].data(); zs.avail_out = (uInt) isize;
			int const rc = isize ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
			inflateEnd(&zs);
			if (Z_STREAM_END != rc) throw io_error("corrupt BGZF block");
		});
		dst.reserve(end - begin);
		for (std::size_t i = 0; i < nb; ++i)
		{
			uint64_t const u0 = m_blocks[b0 + i].second;
			uint64_t const lo = std::max(begin, u0), hi = std::min <uint64_t>(end, u0 + plain[i].size());
			if (lo < hi) dst.insert(dst.end(), plain[i].begin() + (std::ptrdiff_t) (lo - u0), plain[i].begin() + (std::ptrdiff_t) (hi - u0));
		}
		if (dst.size() != end - begin) throw io_error("FASTA shorter than its .fai says");
	}
};

// The named contigs only, through <path>.fai (+ <path>.gzi for BGZF); returns false if there is no .fai (or no .gzi next to a
// BGZF file), in which case the caller reads the file as a whole.
inline bool read_fasta_contigs(char const *path, std::vector <std::string> const &names, unsigned threads, std::vector <std::pair <std::string, std::vector <char>>> &out)
{
	std::string const p(path);
	if (!file_exists(p + ".fai")) return false;
	{
		FILE *f = std::fopen(path, "rb");
		if (!f) throw io_error("unable to open " + p);
		uint8_t m[2]{}; std::size_t const got = std::fread(m, 1, 2, f); std::fclose(f);
		if (2 == got && 0x1f == m[0] && 0x8b == m[1] && !file_exists(p + ".gzi")) return false;
	}
	auto const fai = read_fai(p + ".fai");
	fasta_random_access ra(p, threads);
	std::vector <char> raw;
	for (auto const &name : names)
	{
		auto const it = std::find_if(fai.begin(), fai.end(), [&](fai_entry const &e) { return e.name == name; });
		if (fai.end() == it) continue;		// the caller reports what is missing
		// Up to the contig's last base: the terminator of its last line is not asked for (a FASTA may end without one).
		uint64_t const full_lines = it->length / it->line_bases, rest = it->length % it->line_bases;
		uint64_t const bytes = rest ? (full_lines * it->line_width + rest) : (full_lines ? ((full_lines - 1) * it->line_width + it->line_bases) : 0);
		ra.read(it->offset, it->offset + bytes, raw);
		std::vector <char> seq;
		seq.reserve(it->length);
		for (uint64_t at = 0; at < raw.size() && seq.size() < it->length; at += it->line_width)
		{
			uint64_t const n = std::min <uint64_t>(it->line_bases, std::min <uint64_t>(raw.size() - at, it->length - seq.size()));
			seq.insert(seq.end(), raw.begin() + (std::ptrdiff_t) at, raw.begin() + (std::ptrdiff_t) (at + n));
		}
		if (seq.size() != it->length) throw io_error("contig " + name + " is shorter than its .fai entry says");
		out.emplace_back(name, std::move(seq));
	}
	return true;
}

inline void read_fasta(char const *path, std::vector <std::pair <std::string, std::vector <char>>> &out)
{
	byte_reader in;
	in.open(path);
	std::string line;
	while (in.getline(line))
	{
		if (!line.empty() && '>' == line[0])
		{
			std::size_t e = 1;
			while (e < line.size() && ' ' != line[e] && '\t' != line[e]) ++e;
			out.emplace_back(line.substr(1, e - 1), std::vector <char>());
		}
		else if (!out.empty())
			out.back().second.insert(out.back().second.end(), line.begin(), line.end());
	}
}

}
