// The reference's tests/alignment_projector.cc:62-84, 157-247 through include/pvc3.hpp (the C++ mirror of
// panvc3::alignment_projector over the C ABI): for every "Q" row of a fixture TSV build the sequence entry pair from the two
// gapped strings of the preceding "S" row (make_sequence_entry_pair), project the alignment with all-maximum qualities and gap
// costs (0, -1), and print  line <tab> projected position <tab> CIGAR <tab> realigned reference ranges <tab> query ranges.
// The GPU test compares the output with the fixture's expectations (cigar_eq) and with the oracle.  Test infrastructure.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "pvc3.hpp"

namespace {

std::vector <std::string> split_tabs(std::string const &line)
{
	std::vector <std::string> out;
	std::string cur;
	for (char c : line) { if ('\t' == c) { out.push_back(cur); cur.clear(); } else cur.push_back(c); }
	out.push_back(cur);
	return out;
}

pvc3::cigar_vector parse_cigar(std::string const &text)
{
	static std::string const ops("MIDNSHP=X");
	pvc3::cigar_vector out;
	std::uint32_t n(0);
	for (char c : text)
	{
		if ('0' <= c && c <= '9') n = 10 * n + std::uint32_t(c - '0');
		else { out.push_back((n << 4) | std::uint32_t(ops.find(c))); n = 0; }
	}
	return out;
}

std::string format_cigar(pvc3::cigar_vector const &cig)
{
	std::ostringstream os;
	for (auto const v : cig) os << (v >> 4) << "MIDNSHP=X"[v & 0xf];
	return os.str();
}

std::string without_gaps(std::string const &s)
{
	std::string out;
	for (char c : s) if ('-' != c) out.push_back(c);
	return out;
}

}

int main(int argc, char **argv)
{
	if (argc < 2) { std::fprintf(stderr, "usage: %s fixture.tsv\n", argv[0]); return 2; }
	try
	{
		std::ifstream in(argv[1]);
		std::string line, src, dst;
		std::size_t lineno(0);
		while (std::getline(in, line))
		{
			++lineno;
			if (line.empty()) continue;
			auto const f(split_tabs(line));
			if ("S" == f[0]) { src = f[2]; dst = f[3]; continue; }
			// Q  query  src_pos  expected_dst_pos  cigar  expected_cigar
			pvc3::index idx;
			auto const pair(pvc3::make_sequence_entry_pair(idx, src, dst));		// msa_index.hh:214-236
			idx.set_reference(pair.second, without_gaps(dst));						// alignment_projector.cc:223: dst without gaps is ref_seq
			pvc3::context ctx(0);
			ctx.upload_index(idx);
			pvc3::alignment_projector projector(ctx);
			std::string const &query(f[1]);
			std::vector <std::uint8_t> const qual(query.size(), 41);				// alignment_projector.cc:223: all-maximum qualities
			auto const pos(projector.project_alignment(std::stoul(f[2]), pair.first, pair.second, query, parse_cigar(f[4]), qual, 0, -1));
			std::cout << lineno << '\t' << pos << '\t' << format_cigar(projector.alignment()) << '\t';
			for (auto const &r : projector.realigned_reference_ranges()) std::cout << r.location << ',' << r.length << ';';
			std::cout << '\t';
			for (auto const &r : projector.realigned_query_ranges()) std::cout << r.location << ',' << r.length << ';';
			std::cout << '\n';
		}
	}
	catch (pvc3::error const &e)
	{
		std::fprintf(stderr, "pvc3 error %d: %s\n", e.code, e.what());
		return 1;
	}
	return 0;
}
