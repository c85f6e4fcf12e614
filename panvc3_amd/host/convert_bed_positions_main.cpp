// panvc3_convert_bed_positions — converts the co-ordinates of BED intervals from one sequence of the MSA to another
// (SURVEY.md §8 f4).  Same command line as the reference tool (convert-bed-positions/cmdline.ggo):
//   --msa-index=index.dat --chr=chr --src-seq=id --dst-seq=id [--dst-chr=chr] [--bed=input.bed] < input.bed
// For every line whose first field equals --chr both bounds are converted with the reference's convert_position
// (convert-bed-positions/main.cc:60-73): the column of the srcpos-th non-gap character of the source row, then the number
// of non-gap characters of the destination row before that column (for both values of the destination bit the
// reference's rank0(1 + column) + bit - 1 equals rank0(column)), i.e. project_position — here pvc3_project_positions on
// the GPU, all positions of the file in one batch per 4 M lines.  Lines of other chromosomes are counted and dropped
// (:84-90).  The BED parser of the reference lives in libbio, which is absent: only the first three tab-separated fields
// are read; empty lines and lines starting with '#', "track" or "browser" are skipped.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/pvc3.h"

#include "msa_index_file.hpp"

using namespace pvc3host;

namespace {

[[noreturn]] void die(std::string const &msg)
{
	std::cerr << "ERROR: " << msg << '\n';
	std::exit(EXIT_FAILURE);
}

void check(int rc, char const *what)
{
	if (rc) die(std::string(what) + ": " + pvc3_last_error());
}

}

int main(int argc, char **argv)
{
	char const *msa_index_path{}, *bed_path{}, *chr{}, *dst_chr{}, *src_seq{}, *dst_seq{};
	int device = 0;
	static option const opts[] = {
		{"msa-index", required_argument, nullptr, 'm'}, {"bed", required_argument, nullptr, 'b'}, {"chr", required_argument, nullptr, 'c'},
		{"dst-chr", required_argument, nullptr, 't'}, {"src-seq", required_argument, nullptr, 's'}, {"dst-seq", required_argument, nullptr, 'd'},
		{"device", required_argument, nullptr, 1000}, {"help", no_argument, nullptr, 'h'}, {"version", no_argument, nullptr, 'V'}, {nullptr, 0, nullptr, 0}};
	int c;
	while (-1 != (c = getopt_long(argc, argv, "m:b:c:t:s:d:hV", opts, nullptr)))
	{
		switch (c)
		{
			case 'm': msa_index_path = optarg; break;
			case 'b': bed_path = optarg; break;
			case 'c': chr = optarg; break;
			case 't': dst_chr = optarg; break;
			case 's': src_seq = optarg; break;
			case 'd': dst_seq = optarg; break;
			case 1000: device = std::atoi(optarg); break;
			case 'V': std::cout << "convert-bed-positions 0.1 (panvc3-mi355x)\n"; return 0;
			case 'h': std::cout << "Usage: panvc3_convert_bed_positions --msa-index=index.dat --chr=chr --src-seq=src-seq-id --dst-seq=dst-seq-id [--dst-chr=chr] [--bed=input.bed] < input.bed\n"; return 0;
			default: return EXIT_FAILURE;
		}
	}
	if (!msa_index_path || !chr || !src_seq || !dst_seq) die("--msa-index, --chr, --src-seq and --dst-seq are required");

	try
	{
		std::cerr << "Loading the MSA index…\n";
		msa_index_file msa;
		msa.load(msa_index_path);
		auto find_chr = [&](char const *id) -> msa_chr_entry const & {
			auto const *ce = msa.find_chr(id);
			if (!ce) die(std::string("Did not find an entry for chromosome ID “") + id + "” in the MSA index.");	// main.cc:24-26
			return *ce;
		};
		auto find_seq = [&](msa_chr_entry const &ce, char const *id) -> msa_sequence_entry const & {
			auto const *se = msa_index_file::find_seq(ce, id);
			if (!se) die(std::string("Did not find an entry for sequence ID “") + id + "” in the MSA index.");		// main.cc:39-41
			return *se;
		};
		msa_chr_entry const &src_chr = find_chr(chr);
		msa_chr_entry const &dst_chr_entry = (dst_chr && *dst_chr) ? find_chr(dst_chr) : src_chr;
		msa_sequence_entry const &src = find_seq(src_chr, src_seq);
		msa_sequence_entry const &dst = find_seq(dst_chr_entry, dst_seq);
		if (src.n_cols != dst.n_cols) die("the source and destination sequences have different numbers of columns");

		pvc3_index *index{};
		check(pvc3_index_create(&index), "pvc3_index_create");
		uint32_t e_src = 0, e_dst = 0;
		check(pvc3_index_add_bits(index, src.gap_words.data(), src.n_cols, &e_src), "pvc3_index_add_bits");
		check(pvc3_index_add_bits(index, dst.gap_words.data(), dst.n_cols, &e_dst), "pvc3_index_add_bits");
		pvc3_context *ctx{};
		check(pvc3_context_create(device, &ctx), "pvc3_context_create");
		check(pvc3_context_upload_index(ctx, index), "pvc3_context_upload_index");

		std::cerr << "Processing the BED input…\n";
		byte_reader in;
		in.open(bed_path);
		std::string line;
		std::vector <uint32_t> pos, out, se, de;
		uint64_t matches = 0, mismatches = 0, lineno = 0;
		std::string const chr_id(chr);
		auto flush = [&]() {
			if (pos.empty()) return;
			se.assign(pos.size(), e_src); de.assign(pos.size(), e_dst); out.resize(pos.size());
			check(pvc3_project_positions(ctx, pos.size(), se.data(), de.data(), pos.data(), out.data()), "pvc3_project_positions");
			for (std::size_t i = 0; i < pos.size(); i += 2)
			{
				if (PVC3_NO_MATE == out[i] || PVC3_NO_MATE == out[i + 1])
					die("interval " + std::to_string(pos[i]) + "-" + std::to_string(pos[i + 1]) + " lies outside the source sequence");
				std::cout << chr_id << '\t' << out[i] << '\t' << out[i + 1] << '\n';
			}
			pos.clear();
		};
		bool eof = false;
		while (!eof)
		{
			if (!in.getline(line)) { eof = true; break; }
			++lineno;
			if (line.empty() || '#' == line[0] || 0 == line.compare(0, 5, "track") || 0 == line.compare(0, 7, "browser")) continue;
			std::size_t const t1 = line.find('\t');
			std::size_t const t2 = (std::string::npos == t1) ? t1 : line.find('\t', t1 + 1);
			if (std::string::npos == t2) die("Parse error in BED input on line " + std::to_string(lineno) + '.');		// main.cc:78-80
			if (0 != line.compare(0, t1, chr_id)) { ++mismatches; continue; }		// :86-90
			std::size_t const t3 = line.find('\t', t2 + 1);
			char *end{};
			unsigned long long const lb = std::strtoull(line.c_str() + t1 + 1, &end, 10);
			if (end != line.c_str() + t2) die("Parse error in BED input on line " + std::to_string(lineno) + '.');
			unsigned long long const rb = std::strtoull(line.c_str() + t2 + 1, &end, 10);
			if (end != line.c_str() + ((std::string::npos == t3) ? line.size() : t3)) die("Parse error in BED input on line " + std::to_string(lineno) + '.');
			if (lb > 0xFFFFFFFEull || rb > 0xFFFFFFFEull) die("position too large on line " + std::to_string(lineno));
			++matches;
			pos.push_back((uint32_t) lb); pos.push_back((uint32_t) rb);
			if (pos.size() >= (std::size_t) 8 << 20) flush();
		}
		flush();
		std::cout << std::flush;
		pvc3_context_destroy(ctx);
		pvc3_index_destroy(index);
		std::cerr << "Done. Chromosome ID matches: " << matches << " mismatches: " << mismatches << ".\n";	// :104
	}
	catch (std::exception const &e)
	{
		die(e.what());
	}
	return 0;
}
