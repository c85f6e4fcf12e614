// panvc3_index_msa — builds the MSA co-ordinate transformation index.
//
// Same command line as the reference tool (index-msa/cmdline.ggo) for the modes the workflows use:
//   --build-index --sequence-inputs=inputs.tsv [--pipe-input='bzip2 -d -c'] --msa-index-output=index.dat [--output-fasta
//   [--fasta-line-width=N]]      (test-workflow/Snakefile:36, bowtie2-workflow/Snakefile:96-101)
//   --build-index --sequences=msa.a2m ...      --list-contents --msa-index-input=index.dat
//   --query --msa-index-input=index.dat --chr-id=chr      (interactive project_position queries, index-msa/main.cc:88-186)
// One gap bit-vector per (chromosome, sequence): bit i set iff byte i of the aligned sequence is '-' (every input byte is a
// column, index-msa/index_vector_builder.cc:29-52); with --output-fasta the non-gap bytes go to stdout as ">chr/seq".
// The index file format is this implementation's own (msa_index_file.hpp).
#include <getopt.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "msa_index_file.hpp"

using namespace pvc3host;

namespace {

struct builder
{
	bool output_seq{};
	long wrap{};
	uint64_t non_gap{};

	void begin(std::string const &chr_id, std::string const &seq_id, msa_sequence_entry &e)
	{
		if (output_seq) std::cout << '>' << chr_id << '/' << seq_id << '\n';
		e.seq_id = seq_id; e.n_cols = 0; e.gap_words.clear();
		non_gap = 0;
	}

	void update(char const *buf, std::size_t n, msa_sequence_entry &e)
	{
		e.gap_words.resize((e.n_cols + n + 63) / 64, 0);
		for (std::size_t i = 0; i < n; ++i)
		{
			uint64_t const c = e.n_cols + i;
			if ('-' == buf[i]) e.gap_words[c >> 6] |= UINT64_C(1) << (c & 63);
			else if (output_seq)
			{
				std::cout << buf[i];
				++non_gap;
				if (wrap && 0 == non_gap % (uint64_t) wrap) std::cout << '\n';
			}
		}
		e.n_cols += n;
	}

	void end()
	{
		if (output_seq && (0 == wrap || 0 != non_gap % (uint64_t) wrap)) std::cout << '\n';
	}
};

std::string shell_quote(std::string const &s)
{
	std::string q = "'";
	for (char c : s) { if ('\'' == c) q += "'\\''"; else q += c; }
	return q + "'";
}

[[noreturn]] void die(std::string const &msg)
{
	std::cerr << "ERROR: " << msg << '\n';
	std::exit(EXIT_FAILURE);
}

msa_chr_entry &chr_entry_for(msa_index_file &idx, std::string const &chr_id)
{
	for (auto &c : idx.chr_entries) if (c.chr_id == chr_id) return c;
	idx.chr_entries.emplace_back();
	idx.chr_entries.back().chr_id = chr_id;
	return idx.chr_entries.back();
}

}

// Host-side rank0 / select0 over the plain gap words of one index entry (the --query mode needs a handful of look-ups, no GPU).
struct entry_rank_select
{
	msa_sequence_entry const *e{};
	std::vector <uint64_t> zeros_before;		// per 64-column word

	explicit entry_rank_select(msa_sequence_entry const &entry): e(&entry)
	{
		std::size_t const n_words = (entry.n_cols + 63) / 64;
		zeros_before.resize(n_words + 1, 0);
		for (std::size_t w = 0; w < n_words; ++w)
			zeros_before[w + 1] = zeros_before[w] + (uint64_t) __builtin_popcountll(~word(w));
	}

	uint64_t word(std::size_t w) const		// columns past the end count as gaps
	{
		uint64_t v = e->gap_words[w];
		if (w == e->n_cols / 64 && (e->n_cols & 63)) v |= ~UINT64_C(0) << (e->n_cols & 63);
		return v;
	}

	uint64_t n_zeros() const { return zeros_before.back(); }

	uint64_t rank0(uint64_t col) const		// non-gap columns before col
	{
		std::size_t const w = col / 64;
		if (w >= zeros_before.size() - 1) return n_zeros();
		return zeros_before[w] + (uint64_t) __builtin_popcountll(~word(w) & ((UINT64_C(1) << (col & 63)) - 1));
	}

	uint64_t select0(uint64_t k) const		// column of the k-th (0-based) non-gap character; k < n_zeros()
	{
		std::size_t const w = (std::size_t) (std::upper_bound(zeros_before.begin(), zeros_before.end(), k) - zeros_before.begin()) - 1;
		uint64_t z = ~word(w);
		for (uint64_t skip = k - zeros_before[w]; skip; --skip) z &= z - 1;
		return 64 * (uint64_t) w + (uint64_t) __builtin_ctzll(z);
	}
};

// The interactive query mode (index-msa/main.cc:88-186): project_position of typed co-ordinates between two sequences of
// one chromosome; 's' / 'd' switch the source / destination sequence; ends with the input.
int query_index(char const *path, char const *chr_id)
{
	msa_index_file idx;
	idx.load(path);
	msa_chr_entry const *ce = idx.find_chr(chr_id);
	if (!ce)
	{
		std::cerr << "ERROR: No entry for chromosome '" << chr_id << "'.\n";
		return EXIT_FAILURE;
	}
	std::string buffer;
	auto read_seq = [&](char const *msg) -> msa_sequence_entry const * {
		for (;;)
		{
			std::cout << msg << std::flush;
			if (!(std::cin >> buffer)) return nullptr;
			msa_sequence_entry const *se = msa_index_file::find_seq(*ce, buffer);
			if (se) return se;
			std::cerr << "No entry for sequence '" << buffer << "'.\n";
		}
	};
	msa_sequence_entry const *src = read_seq("Source sequence identifier? ");
	if (!src) return 0;
	msa_sequence_entry const *dst = read_seq("Destination sequence identifier? ");
	if (!dst) return 0;
	entry_rank_select rs_src(*src), rs_dst(*dst);
	for (;;)
	{
		std::cout << "[" << src->seq_id << " → " << dst->seq_id << "] Source co-ordinate or 's' or 'd' to switch sequence? ([0, " << rs_src.n_zeros() << ")) " << std::flush;
		if (!(std::cin >> buffer)) return 0;
		if ("s" == buffer)
		{
			src = read_seq("Source sequence identifier? ");
			if (!src) return 0;
			rs_src = entry_rank_select(*src);
		}
		else if ("d" == buffer)
		{
			dst = read_seq("Destination sequence identifier? ");
			if (!dst) return 0;
			rs_dst = entry_rank_select(*dst);
		}
		else
		{
			char *end{};
			unsigned long long const pos = std::strtoull(buffer.c_str(), &end, 10);
			if (buffer.empty() || end != buffer.c_str() + buffer.size() || !(buffer[0] >= '0' && buffer[0] <= '9')) continue;		// :178-180
			if (!(pos < rs_src.n_zeros())) continue;																		// :182-183
			std::cout << rs_dst.rank0(rs_src.select0(pos)) << '\n';		// msa_index.hh:253-258
		}
	}
}

// (split helper without pulling in sam_bam.hpp)
static std::vector <std::string> split_tabs(std::string const &s)
{
	std::vector <std::string> out;
	std::size_t b = 0;
	for (;;)
	{
		std::size_t const e = s.find('\t', b);
		out.push_back(s.substr(b, (std::string::npos == e) ? e : e - b));
		if (std::string::npos == e) break;
		b = e + 1;
	}
	return out;
}

int main(int argc, char **argv)
{
	char const *msa_index_input{}, *sequence_inputs{}, *a2m_input{}, *msa_index_output{}, *pipe_input{}, *chr_id{};
	bool build_index = false, list_contents = false, query = false, output_fasta = false;
	long fasta_line_width = 0;
	static option const opts[] = {
		{"msa-index-input", required_argument, nullptr, 'd'}, {"build-index", no_argument, nullptr, 'B'},
		{"sequence-inputs", required_argument, nullptr, 'i'}, {"sequences", required_argument, nullptr, 'a'},
		{"msa-index-output", required_argument, nullptr, 'm'}, {"pipe-input", required_argument, nullptr, 'p'},
		{"output-fasta", no_argument, nullptr, 'f'}, {"fasta-line-width", required_argument, nullptr, 'w'},
		{"list-contents", no_argument, nullptr, 'L'}, {"query", no_argument, nullptr, 'Q'}, {"chr-id", required_argument, nullptr, 1000},
		{"help", no_argument, nullptr, 'h'}, {"version", no_argument, nullptr, 'V'}, {nullptr, 0, nullptr, 0}};
	int c;
	while (-1 != (c = getopt_long(argc, argv, "d:Bi:a:m:p:fw:LQhV", opts, nullptr)))
	{
		switch (c)
		{
			case 'd': msa_index_input = optarg; break;
			case 'B': build_index = true; break;
			case 'i': sequence_inputs = optarg; break;
			case 'a': a2m_input = optarg; break;
			case 'm': msa_index_output = optarg; break;
			case 'p': pipe_input = optarg; break;
			case 'f': output_fasta = true; break;
			case 'w': fasta_line_width = std::atol(optarg); break;
			case 'L': list_contents = true; break;
			case 'Q': query = true; break;
			case 1000: chr_id = optarg; break;
			case 'V': std::cout << "index-msa 0.1 (panvc3-mi355x)\n"; return 0;
			case 'h':
				std::cout << "Usage: panvc3_index_msa --build-index --sequence-inputs=inputs.tsv [--pipe-input=cmd] --msa-index-output=msa-index.dat [--output-fasta [--fasta-line-width=N]] > combined.fa\n"
							 "       panvc3_index_msa --build-index --sequences=msa.a2m --msa-index-output=msa-index.dat\n"
							 "       panvc3_index_msa --list-contents --msa-index-input=msa-index.dat\n";
				return 0;
			default: return EXIT_FAILURE;
		}
	}

	try
	{
		if (list_contents)
		{
			if (!msa_index_input) die("--list-contents needs --msa-index-input");
			msa_index_file idx;
			idx.load(msa_index_input);
			for (auto const &ce : idx.chr_entries)		// index-msa/main.cc:61-88
			{
				uint64_t const seq_len = ce.sequence_entries.empty() ? 0 : ce.sequence_entries.front().n_cols;
				std::cout << ce.chr_id << " (" << ce.sequence_entries.size() << " sequences of " << seq_len << " characters)\n";
				for (auto const &se : ce.sequence_entries)
				{
					if (se.n_cols != seq_len) die("Sequence " + se.seq_id + " has " + std::to_string(se.n_cols) + " characters, expected " + std::to_string(seq_len) + ".");
					std::cout << '\t' << se.seq_id << '\n';
				}
			}
			return 0;
		}
		if (query)
		{
			if (!msa_index_input || !chr_id) die("--query needs --msa-index-input and --chr-id");
			return query_index(msa_index_input, chr_id);
		}
		if (!build_index) die("one of --build-index, --list-contents is required");
		if (!msa_index_output) die("--msa-index-output is required");
		if (!sequence_inputs && !a2m_input) die("--sequence-inputs or --sequences is required");

		msa_index_file idx;
		builder b;
		b.output_seq = output_fasta; b.wrap = fasta_line_width;
		std::vector <char> buf(1 << 20);

		if (sequence_inputs)
		{
			byte_reader list;
			list.open(sequence_inputs);
			std::string line;
			while (list.getline(line))		// index-msa/input_processor.cc:77-98: chr_id \t seq_id \t path
			{
				if (line.empty()) continue;
				auto const f = split_tabs(line);
				if (f.size() < 3) die("expected chr_id, seq_id and path separated by tabulators: " + line);
				msa_chr_entry &ce = chr_entry_for(idx, f[0]);
				ce.sequence_entries.emplace_back();
				msa_sequence_entry &se = ce.sequence_entries.back();
				b.begin(f[0], f[1], se);
				FILE *in = pipe_input ? popen((std::string(pipe_input) + " " + shell_quote(f[2])).c_str(), "r") : std::fopen(f[2].c_str(), "rb");
				if (!in) die("unable to read " + f[2]);
				std::size_t n;
				while ((n = std::fread(buf.data(), 1, buf.size(), in)) > 0)
					b.update(buf.data(), n, se);
				int const rc = pipe_input ? pclose(in) : std::fclose(in);
				if (pipe_input && 0 != rc) die("the pipe command failed for " + f[2]);
				b.end();
			}
		}
		else
		{
			// A2M: ">chr_id<tab or space>seq_id" headers (index-msa/input_processor.cc:243-), sequence lines concatenated.
			byte_reader in;
			in.open(a2m_input);
			std::string line;
			msa_sequence_entry *cur{};
			std::string chr_id, seq_id;
			while (in.getline(line))
			{
				if (!line.empty() && '>' == line[0])
				{
					if (cur) b.end();
					std::size_t const sp = line.find_first_of(" \t");
					if (std::string::npos == sp) die("A2M header without a sequence identifier: " + line);
					chr_id = line.substr(1, sp - 1);
					seq_id = line.substr(sp + 1);
					msa_chr_entry &ce = chr_entry_for(idx, chr_id);
					ce.sequence_entries.emplace_back();
					cur = &ce.sequence_entries.back();
					b.begin(chr_id, seq_id, *cur);
				}
				else if (cur)
					b.update(line.data(), line.size(), *cur);
			}
			if (cur) b.end();
		}

		// All sequences of a chromosome must have the same number of columns (index-msa/main.cc:76-80 checks it on listing).
		for (auto const &ce : idx.chr_entries)
			for (auto const &se : ce.sequence_entries)
				if (se.n_cols != ce.sequence_entries.front().n_cols)
					die("Sequence " + ce.chr_id + "/" + se.seq_id + " has " + std::to_string(se.n_cols) + " characters, expected " + std::to_string(ce.sequence_entries.front().n_cols) + ".");
		idx.sort();
		idx.save(msa_index_output);
		std::cout.flush();
	}
	catch (std::exception const &e)
	{
		die(e.what());
	}
	return 0;
}
