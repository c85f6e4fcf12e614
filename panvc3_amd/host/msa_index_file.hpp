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
