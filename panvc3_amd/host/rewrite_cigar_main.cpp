// panvc3_rewrite_cigar — rewrites the CIGAR of every record.  Same command line as the reference tool
// (rewrite-cigar/cmdline.ggo); no MSA and no projection is involved, each record is compared with its own reference sequence —
// on the host's threads, or with --device=N on that GPU (pvc3_split_match_ops, include/pvc3.h: the reference sequences in HBM,
// the records' operations and bases in batches of 2^18).
//   --output-alignment-match-ops (-M)   = and X become M, adjacent equal operations are merged by cigar_buffer
//                                        (rewrite-cigar/main.cc:151-188, libpanvc3/cigar.cc:54-95)
//   --output-sequence-match-ops (-S) --reference=ref.fa[.gz] [--ref-n-positions-tag=XN]
//                                        every M is split into = / X by comparing dna5::to_char of the read base with the raw
//                                        reference byte (:191-284; runs inside one M are NOT merged with neighbouring
//                                        operations), and the positions of 'N' in the reference, relative to the alignment
//                                        start, are stored as a bit map in a B:C tag (:126-148, :417-419)
// Records without a reference id or position pass through unchanged in -S mode (:385-400).  As everywhere in this code base
// tags are written in tag-key order and SEQ / QUAL are normalised as seqan3's dna5 / phred42 alphabets do.
#include <getopt.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <regex>
#include <string>
#include <vector>

#include "../../include/pvc3.h"
#include "msa_index_file.hpp"
#include "sam_bam.hpp"

using namespace pvc3host;

namespace {

[[noreturn]] void die(std::string const &msg)
{
	std::cerr << "ERROR: " << msg << '\n';
	std::exit(EXIT_FAILURE);
}

inline uint16_t tag_key(char const t[2]) { return (uint16_t) ((uint8_t) t[0] * 256 + (uint8_t) t[1]); }

// BAM base code -> seqan3::dna5::to_char()
inline char dna5_char_of_code(unsigned code)
{
	switch (code) { case 1: return 'A'; case 2: return 'C'; case 4: return 'G'; case 8: return 'T'; default: return 'N'; }
}

// cigar_buffer (libpanvc3/cigar.cc:54-95): equal neighbours are merged, zero counts dropped.
struct cigar_merger
{
	std::vector <uint32_t> ops;
	uint32_t cur_op{0xf}, cur_cnt{};
	void push(uint32_t op, uint32_t cnt)
	{
		if (0 == cnt) return;
		if (op == cur_op) { cur_cnt += cnt; return; }
		if (cur_cnt) ops.push_back((cur_cnt << 4) | cur_op);
		cur_op = op; cur_cnt = cnt;
	}
	void finish() { if (cur_cnt) ops.push_back((cur_cnt << 4) | cur_op); cur_op = 0xf; cur_cnt = 0; }
};

}

int main(int argc, char **argv)
{
	char const *alignments{}, *output_path{}, *reference{};
	std::string n_tag = "XN";
	bool bam_input = false, output_bam = false, mode_m = false, mode_s = false, print_invocation = false, print_pid = false;
	unsigned threads = default_io_threads();
	long status_interval = 5;
	int device = -1;		// --device: the GPU that splits the M operations (-S); none: the host does
	static option const opts[] = {
		{"alignments", required_argument, nullptr, 'a'}, {"bam-input", no_argument, nullptr, 'B'}, {"output-path", required_argument, nullptr, 'o'},
		{"output-bam", no_argument, nullptr, 'b'}, {"status-output-interval", required_argument, nullptr, 1000},
		{"output-alignment-match-ops", no_argument, nullptr, 'M'}, {"output-sequence-match-ops", no_argument, nullptr, 'S'},
		{"reference", required_argument, nullptr, 'r'}, {"ref-n-positions-tag", required_argument, nullptr, 1001},
		{"print-invocation", no_argument, nullptr, 1002}, {"print-pid", no_argument, nullptr, 1003}, {"threads", required_argument, nullptr, 1004},
		{"device", required_argument, nullptr, 1005},
		{"help", no_argument, nullptr, 'h'}, {"version", no_argument, nullptr, 'V'}, {nullptr, 0, nullptr, 0}};
	int c;
	while (-1 != (c = getopt_long(argc, argv, "a:Bo:bMSr:hV", opts, nullptr)))
	{
		switch (c)
		{
			case 'a': alignments = optarg; break;
			case 'B': bam_input = true; break;
			case 'o': output_path = optarg; break;
			case 'b': output_bam = true; break;
			case 1000: status_interval = std::atol(optarg); break;
			case 'M': mode_m = true; break;
			case 'S': mode_s = true; break;
			case 'r': reference = optarg; break;
			case 1001: n_tag = optarg; break;
			case 1002: print_invocation = true; break;
			case 1003: print_pid = true; break;
			case 1004: threads = (unsigned) std::max(1, std::atoi(optarg)); break;
			case 1005: device = std::atoi(optarg); break;
			case 'V': std::cout << "rewrite-cigar 0.1 (panvc3-mi355x)\n"; return 0;
			case 'h': std::cout << "Usage: panvc3_rewrite_cigar --alignments=alignments.bam --output-sequence-match-ops --reference=ref.fa.gz > rewritten.sam\n"
								   "       panvc3_rewrite_cigar --alignments=alignments.bam --output-alignment-match-ops > rewritten.sam\n"; return 0;
			default: return EXIT_FAILURE;
		}
	}
	if (print_invocation) { std::cerr << "Invocation:"; for (int i = 0; i < argc; ++i) std::cerr << ' ' << argv[i]; std::cerr << '\n'; }
	if (print_pid) std::cerr << "PID: " << getpid() << '\n';
	if (alignments && bam_input) die("--bam-input has no effect when reading input from a file.");		// main.cc:437-441
	if (mode_m == mode_s) die("exactly one of --output-alignment-match-ops and --output-sequence-match-ops is required");
	if (mode_s && !reference) die("--reference is required with --output-sequence-match-ops");
	if (device >= 0 && !mode_s) die("--device has an effect with --output-sequence-match-ops only");
	if (!std::regex_match(n_tag, std::regex("^[XYZ][A-Za-z0-9]$"))) die("The given tag '" + n_tag + "' does not match the expected format.");		// :476-480
	if (status_interval < 0) die("Status output interval must be non-negative.");		// :489-493

	try
	{
		alignment_reader reader;
		reader.set_threads(threads);
		{
			bool is_bam = bam_input;
			if (alignments)
			{
				std::string const p(alignments);
				is_bam = p.size() >= 4 && 0 == p.compare(p.size() - 4, 4, ".bam");
			}
			reader.open(alignments, is_bam);
		}
		sam_header const &in_header = reader.header;
		sam_header out_header = in_header;
		append_sam_program_info(in_header, out_header, "panvc3.rewrite-cigar.", "PanVC 3 rewrite_cigar", argc, argv);		// :109-122
		alignment_writer writer;
		writer.set_threads(threads);
		writer.open(output_path, output_bam, out_header);

		std::map <std::string, std::vector <char>> ref_by_name;
		if (mode_s)
		{
			std::cerr << "Loading the reference sequences…\n";
			std::vector <std::pair <std::string, std::vector <char>>> fasta;
			read_fasta(reference, fasta);
			for (auto &kv : fasta) ref_by_name.emplace(std::move(kv.first), std::move(kv.second));
		}
		std::vector <std::vector <char> const *> ref_of_id(in_header.refs.size(), nullptr);

		std::cerr << "Processing the alignments…\n";
		bam_record w;
		std::map <uint16_t, std::vector <uint8_t>> tags;
		// -- the output record: header fields, new CIGAR, SEQ / QUAL through dna5 / phred42, tags in key order (+ the N bit map)
		auto write_record = [&](bam_record const &r, std::vector <uint32_t> const &new_cigar, std::vector <uint8_t> const &n_positions, bool rewrite) {
			if (new_cigar.size() > 65535) die("more than 65535 CIGAR operations after rewriting");

			std::vector <uint32_t> const *cig = &new_cigar;
			std::vector <uint32_t> old;
			if (!rewrite)
			{
				for (uint16_t k = 0; k < r.n_cigar(); ++k) old.push_back(r.get <uint32_t>(r.cigar_offset() + 4u * k));
				cig = &old;
			}
			tags.clear();
			for (auto const &f : parse_aux(r.data.data() + r.aux_offset(), r.data.data() + r.data.size()))
			{
				std::vector <uint8_t> &v = tags[tag_key(f.tag)];
				v.assign({(uint8_t) f.tag[0], (uint8_t) f.tag[1], (uint8_t) f.type});
				v.insert(v.end(), f.value, f.value + f.size);
			}
			if (mode_s && rewrite && !n_positions.empty())		// :417-419
			{
				std::vector <uint8_t> &v = tags[tag_key(n_tag.c_str())];
				v.assign({(uint8_t) n_tag[0], (uint8_t) n_tag[1], 'B', 'C'});
				uint32_t const cnt = (uint32_t) n_positions.size(); uint8_t b[4]; std::memcpy(b, &cnt, 4);
				v.insert(v.end(), b, b + 4);
				v.insert(v.end(), n_positions.begin(), n_positions.end());
			}
			std::size_t const name_end = 32 + r.l_read_name();
			w.data.assign(r.data.begin(), r.data.begin() + (std::ptrdiff_t) name_end);
			w.set <uint16_t>(12, (uint16_t) cig->size());
			int64_t span = 0;
			for (uint32_t cc : *cig) { uint32_t const op = cc & 0xf; if (0 == op || 2 == op || 3 == op || 7 == op || 8 == op) span += cc >> 4; }
			if (r.pos() >= 0) w.set <uint16_t>(10, reg2bin(r.pos(), (int64_t) r.pos() + (span ? span : 1)));
			w.data.insert(w.data.end(), (uint8_t const *) cig->data(), (uint8_t const *) (cig->data() + cig->size()));
			std::size_t const seq_at = w.data.size();
			w.data.insert(w.data.end(), r.seq4(), r.seq4() + (r.l_seq() + 1) / 2);
			for (uint32_t k = 0; k < r.l_seq(); ++k)
			{
				uint8_t &byte = w.data[seq_at + (k >> 1)];
				unsigned const shift = (~k & 1u) << 2;
				unsigned const code = (byte >> shift) & 0xf;
				if (1 != code && 2 != code && 4 != code && 8 != code) byte = (uint8_t) ((byte & ~(0xf << shift)) | (15u << shift));
			}
			std::size_t const qual_at = w.data.size();
			w.data.insert(w.data.end(), r.qual(), r.qual() + r.l_seq());
			if (r.has_qual()) for (uint32_t k = 0; k < r.l_seq(); ++k) if (w.data[qual_at + k] > 41) w.data[qual_at + k] = 41;
			for (auto const &kv : tags) w.data.insert(w.data.end(), kv.second.begin(), kv.second.end());
			writer.write(w);
		};

		if (device >= 0)
		{
			// ---- the M operations split on the GPU -------------------------------------------------------------------------------------
			auto check = [](int rc, char const *what) { if (rc) die(std::string(what) + ": " + pvc3_last_error()); };
			pvc3_context *ctx{};
			check(pvc3_context_create(device, &ctx), "pvc3_context_create");
			std::vector <uint8_t const *> seq_bytes; std::vector <uint64_t> seq_len;
			std::vector <uint32_t> sequence_of_id(in_header.refs.size(), UINT32_MAX);
			for (std::size_t i = 0; i < in_header.refs.size(); ++i)
			{
				auto const it = ref_by_name.find(in_header.refs[i].name);
				if (ref_by_name.end() == it) continue;		// (an error only if a record lies on it, as in the loop below)
				sequence_of_id[i] = (uint32_t) seq_bytes.size();
				seq_bytes.push_back((uint8_t const *) it->second.data()); seq_len.push_back(it->second.size());
			}
			pvc3_reference *ref_dev{};
			check(pvc3_reference_create(ctx, (uint32_t) seq_bytes.size(), seq_bytes.data(), seq_len.data(), &ref_dev), "pvc3_reference_create");
			constexpr std::size_t BATCH = std::size_t(1) << 18;
			std::vector <bam_record> recs(BATCH);
			std::vector <uint32_t> which, sequence, pos, l_seq, cigar, out_cigar;
			std::vector <uint64_t> cigar_off, seq_off, out_off, map_off;
			std::vector <uint8_t> seq4, out_map;
			std::vector <int32_t> status;
			std::vector <uint32_t> one_cigar; std::vector <uint8_t> one_map;
			for (bool more = true; more; )
			{
				std::size_t n_recs = 0;
				while (n_recs < BATCH && (more = reader.next(recs[n_recs]))) ++n_recs;
				which.clear(); sequence.clear(); pos.clear(); l_seq.clear(); cigar.clear(); seq4.clear(); cigar_off.assign(1, 0); seq_off.clear();
				for (std::size_t k = 0; k < n_recs; ++k)
				{
					bam_record const &r = recs[k];
					if (r.pos() < 0 || r.ref_id() < 0) continue;		// :385-400: passes through
					std::size_t const ref_id = (std::size_t) r.ref_id();
					if (ref_id >= sequence_of_id.size()) die("record with a reference id outside the header");
					if (UINT32_MAX == sequence_of_id[ref_id]) die("Unable to load sequence ‘" + in_header.refs[ref_id].name + "’ from the input FASTA.");		// :410-413
					which.push_back((uint32_t) k); sequence.push_back(sequence_of_id[ref_id]); pos.push_back((uint32_t) r.pos()); l_seq.push_back(r.l_seq());
					for (uint16_t x = 0; x < r.n_cigar(); ++x) cigar.push_back(r.get <uint32_t>(r.cigar_offset() + 4u * x));
					cigar_off.push_back(cigar.size());
					seq_off.push_back(seq4.size());
					seq4.insert(seq4.end(), r.seq4(), r.seq4() + (r.l_seq() + 1) / 2);
				}
				std::size_t const n = which.size();
				out_off.assign(n + 1, 0); map_off.assign(n + 1, 0); status.assign(n ? n : 1, 0);
				pvc3_split_in in{};
				in.n = n; in.sequence = sequence.data(); in.pos = pos.data(); in.cigar_off = cigar_off.data(); in.cigar = cigar.data(); in.seq_off = seq_off.data();
				in.seq4 = seq4.data(); in.seq4_bytes = seq4.size(); in.l_seq = l_seq.data();
				for (int attempt = 0; ; ++attempt)
				{
					pvc3_split_out out{};
					out.cigar_off = out_off.data(); out.cigar = out_cigar.data(); out.cigar_cap = out_cigar.size();
					out.n_map_off = map_off.data(); out.n_map = out_map.data(); out.n_map_cap = out_map.size(); out.status = status.data();
					uint64_t n_cigar = 0, n_map = 0;
					int const rc = pvc3_split_match_ops(ctx, ref_dev, &in, &out, &n_cigar, &n_map);
					if (PVC3_E_CAPACITY == rc && 0 == attempt) { out_cigar.resize(n_cigar + n_cigar / 4 + 1024); out_map.resize(n_map + n_map / 4 + 1024); continue; }
					check(rc, "pvc3_split_match_ops");
					break;
				}
				std::size_t next = 0;
				for (std::size_t k = 0; k < n_recs; ++k)
				{
					bam_record const &r = recs[k];
					one_cigar.clear(); one_map.clear();
					bool const rewrite = next < n && which[next] == k;
					if (rewrite)
					{
						switch (status[next])
						{
							case PVC3_SPLIT_OK: break;
							case PVC3_SPLIT_BAD_OP: die(std::string("Unexpected CIGAR operation in record ") + r.read_name());
							case PVC3_SPLIT_QUERY_RANGE: die(std::string("the CIGAR of ") + r.read_name() + " consumes more bases than SEQ has");
							case PVC3_SPLIT_EMPTY_M: die(std::string("empty M operation in record ") + r.read_name() + " (undefined in the reference tool)");
							default: die(std::string("the alignment of ") + r.read_name() + " extends past the end of its reference sequence");
						}
						one_cigar.assign(out_cigar.begin() + (std::ptrdiff_t) out_off[next], out_cigar.begin() + (std::ptrdiff_t) out_off[next + 1]);
						one_map.assign(out_map.begin() + (std::ptrdiff_t) map_off[next], out_map.begin() + (std::ptrdiff_t) map_off[next + 1]);
						++next;
					}
					write_record(r, one_cigar, one_map, rewrite);
				}
			}
			pvc3_reference_destroy(ref_dev);
			pvc3_context_destroy(ctx);
			writer.close();
			return 0;
		}

		bam_record rec;
		cigar_merger merger;
		std::vector <uint32_t> new_cigar;
		std::vector <uint8_t> n_positions;
		while (reader.next(rec))
		{
			bam_record const &r = rec;
			new_cigar.clear();
			n_positions.clear();
			bool rewrite = true;
			if (mode_m)
			{
				merger.ops.clear();
				for (uint16_t k = 0; k < r.n_cigar(); ++k)
				{
					uint32_t const cc = r.get <uint32_t>(r.cigar_offset() + 4u * k);
					uint32_t const op = cc & 0xf, cnt = cc >> 4;
					if (op > 8) die(std::string("Unexpected CIGAR operation in record ") + r.read_name());
					merger.push((7 == op || 8 == op) ? 0u : op, cnt);		// = X -> M (:176-179)
				}
				merger.finish();
				new_cigar = merger.ops;
			}
			else if (r.pos() < 0 || r.ref_id() < 0)
				rewrite = false;		// :385-400
			else
			{
				std::size_t const ref_id = (std::size_t) r.ref_id();
				if (ref_id >= ref_of_id.size()) die("record with a reference id outside the header");
				if (!ref_of_id[ref_id])
				{
					auto const it = ref_by_name.find(in_header.refs[ref_id].name);
					if (ref_by_name.end() == it) die("Unable to load sequence ‘" + in_header.refs[ref_id].name + "’ from the input FASTA.");		// :410-413
					ref_of_id[ref_id] = &it->second;
				}
				std::vector <char> const &ref = *ref_of_id[ref_id];
				std::size_t const base = (std::size_t) r.pos();
				std::size_t query_pos = 0, ref_pos = 0;
				auto fill_n = [&](std::size_t count) {		// :126-148
					if (base + ref_pos + count > ref.size()) die(std::string("the alignment of ") + r.read_name() + " extends past the end of its reference sequence");
					n_positions.resize((ref_pos + count + 7) / 8, 0);
					for (std::size_t i = 0; i < count; ++i)
						if ('N' == ref[base + ref_pos + i])
							n_positions[(ref_pos + i) / 8] |= (uint8_t) (1u << ((ref_pos + i) % 8));
				};
				for (uint16_t k = 0; k < r.n_cigar(); ++k)
				{
					uint32_t const cc = r.get <uint32_t>(r.cigar_offset() + 4u * k);
					uint32_t const op = cc & 0xf, cnt = cc >> 4;
					switch (op)
					{
						case 5: case 6: new_cigar.push_back(cc); break;								// H P
						case 1: case 4: new_cigar.push_back(cc); query_pos += cnt; break;			// I S
						case 7: case 8: new_cigar.push_back(cc); fill_n(cnt); query_pos += cnt; ref_pos += cnt; break;	// = X
						case 2: case 3: new_cigar.push_back(cc); fill_n(cnt); ref_pos += cnt; break;	// D N
						case 0:		// M (:247-281)
						{
							if (query_pos + cnt > r.l_seq()) die(std::string("the CIGAR of ") + r.read_name() + " consumes more bases than SEQ has");
							if (0 == cnt) die(std::string("empty M operation in record ") + r.read_name() + " (undefined in the reference tool)");
							fill_n(cnt);
							auto is_match = [&](std::size_t i) {
								std::size_t const qi = query_pos + i;
								unsigned const code = (r.seq4()[qi >> 1] >> ((~qi & 1u) << 2)) & 0xf;
								return dna5_char_of_code(code) == ref[base + ref_pos + i];
							};
							uint32_t prev_cnt = 1, prev_op = is_match(0) ? 7u : 8u;
							for (std::size_t i = 1; i < cnt; ++i)
							{
								uint32_t const cur = is_match(i) ? 7u : 8u;
								if (cur == prev_op) ++prev_cnt;
								else { new_cigar.push_back((prev_cnt << 4) | prev_op); prev_cnt = 1; prev_op = cur; }
							}
							new_cigar.push_back((prev_cnt << 4) | prev_op);
							ref_pos += cnt; query_pos += cnt;
							break;
						}
						default: die(std::string("Unexpected CIGAR operation in record ") + r.read_name());
					}
				}
			}
			write_record(r, new_cigar, n_positions, rewrite);
		}
		writer.close();
	}
	catch (std::exception const &e)
	{
		die(e.what());
	}
	return 0;
}
