// panvc3_recalculate_mapq — the step after project_alignments in both workflows (test-workflow/Snakefile:76-86,
// bowtie2-workflow/Snakefile:143-153).  Same command line as the reference tool
// (recalculate-mapq/cmdline.ggo).  Takes projected alignments grouped by QNAME, re-derives every record's MAPQ from the
// alignment scores of its group the way Bowtie 2 (version 2) does, using the original positions project_alignments left in the
// XR / XP / YR / YP tags to pair the mates, and writes the records of every group in input order.
//
// Host side: SAM / BAM I/O, grouping, tag extraction and tag rewriting (recalculate-mapq/main.cc:291-335, 1186-1278).
// GPU side through the C ABI (include/pvc3.h): pvc3_rescore_alignments (--rescore-alignments: score from the CIGAR, the
// base qualities and the XN bit map, main.cc:189-289) and pvc3_recalculate_mapq (main.cc:441-566, 962-1180).
// Not reproduced: the per-record "mate not found" lines of --verbose (the count is printed at the end as in the reference)
// and the periodic status line.
#include <getopt.h>
#include <unistd.h>

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <regex>
#include <string>
#include <vector>

#include <memory>

#include "../../include/pvc3.h"
#include "sam_bam.hpp"
#include "gpu_deflate.hpp"

using namespace pvc3host;

namespace {

[[noreturn]] void die(std::string const &msg)
{
	std::cerr << "ERROR: " << msg << '\n';
	std::exit(EXIT_FAILURE);
}

void check(int rc) { if (rc) die(pvc3_last_error()); }

inline uint16_t tag_key(char const t[2]) { return (uint16_t) ((uint8_t) t[0] * 256 + (uint8_t) t[1]); }

// Integer value of an aux field (any BAM integer type); false if the field is not an integer.
bool aux_int(aux_field const &f, int64_t &out)
{
	switch (f.type)
	{
		case 'c': out = (int8_t) f.value[0]; return true;
		case 'C': out = f.value[0]; return true;
		case 's': { int16_t v; std::memcpy(&v, f.value, 2); out = v; return true; }
		case 'S': { uint16_t v; std::memcpy(&v, f.value, 2); out = v; return true; }
		case 'i': { int32_t v; std::memcpy(&v, f.value, 4); out = v; return true; }
		case 'I': { uint32_t v; std::memcpy(&v, f.value, 4); out = v; return true; }
		default: return false;
	}
}

// Smallest BAM integer type for a value, as everywhere in this code base.
void put_int_tag(std::vector <uint8_t> &v, char const tag[2], int64_t x)
{
	v.assign({(uint8_t) tag[0], (uint8_t) tag[1]});
	auto put = [&](char type, void const *p, std::size_t n) { v.push_back((uint8_t) type); v.insert(v.end(), (uint8_t const *) p, (uint8_t const *) p + n); };
	if (x >= 0)
	{
		if (x <= 0xff) { uint8_t const y = (uint8_t) x; put('C', &y, 1); }
		else if (x <= 0xffff) { uint16_t const y = (uint16_t) x; put('S', &y, 2); }
		else { uint32_t const y = (uint32_t) x; put('I', &y, 4); }
	}
	else
	{
		if (x >= -128) { int8_t const y = (int8_t) x; put('c', &y, 1); }
		else if (x >= -32768) { int16_t const y = (int16_t) x; put('s', &y, 2); }
		else { int32_t const y = (int32_t) x; put('i', &y, 4); }
	}
}

struct pending_record
{
	bam_record rec;
	bool has_as{};
	int64_t as_value{};
	std::vector <uint8_t> xn;		// bytes of the XN bit map
};

}

int main(int argc, char **argv)
{
	char const *alignments{}, *output_path{};
	bool bam_input = false, output_bam = false, verbose = false, rescore = false, print_invocation = false, print_pid = false, print_reference_names = false;
	long status_interval = 5, device = 0;
	pvc3_rescore_params scoring{2, 6, 1, 5, 3};
	std::string tag_xn = "XN", tag_xr = "XR", tag_xp = "XP", tag_yr = "YR", tag_yp = "YP", tag_xs = "XS", tag_ys = "YS";
	unsigned threads = default_io_threads();
	std::size_t batch_records = 1u << 20;
	bool gpu_deflate = true;		// --bgzf-deflate=gpu|host: BAM output's blocks are compressed on the GPU (as panvc3_project_alignments)
	static option const opts[] = {
		{"alignments", required_argument, nullptr, 'a'}, {"bam-input", no_argument, nullptr, 'B'}, {"output-path", required_argument, nullptr, 'o'},
		{"output-bam", no_argument, nullptr, 'b'}, {"status-output-interval", required_argument, nullptr, 1000}, {"verbose", no_argument, nullptr, 1001},
		{"rescore-alignments", no_argument, nullptr, 1002}, {"min-mismatch-penalty", required_argument, nullptr, 1003}, {"max-mismatch-penalty", required_argument, nullptr, 1004},
		{"n-penalty", required_argument, nullptr, 1005}, {"gap-opening-penalty", required_argument, nullptr, 1006}, {"gap-extension-penalty", required_argument, nullptr, 1007},
		{"ref-n-positions-tag", required_argument, nullptr, 1008}, {"original-rname-tag", required_argument, nullptr, 1009}, {"original-pos-tag", required_argument, nullptr, 1010},
		{"original-rnext-tag", required_argument, nullptr, 1011}, {"original-pnext-tag", required_argument, nullptr, 1012}, {"original-alignment-score-tag", required_argument, nullptr, 1013},
		{"new-alignment-score-tag", required_argument, nullptr, 1014}, {"print-invocation", no_argument, nullptr, 1015}, {"print-pid", no_argument, nullptr, 1016},
		{"print-reference-names", no_argument, nullptr, 1017}, {"threads", required_argument, nullptr, 1018}, {"device", required_argument, nullptr, 1019},
		{"batch-records", required_argument, nullptr, 1020}, {"bgzf-deflate", required_argument, nullptr, 1021}, {"help", no_argument, nullptr, 'h'}, {"version", no_argument, nullptr, 'V'}, {nullptr, 0, nullptr, 0}};
	int c;
	while (-1 != (c = getopt_long(argc, argv, "a:Bo:bhV", opts, nullptr)))
	{
		switch (c)
		{
			case 'a': alignments = optarg; break;
			case 'B': bam_input = true; break;
			case 'o': output_path = optarg; break;
			case 'b': output_bam = true; break;
			case 1000: status_interval = std::atol(optarg); break;
			case 1001: verbose = true; break;
			case 1002: rescore = true; break;
			case 1003: scoring.min_mismatch_penalty = (int32_t) std::atol(optarg); break;
			case 1004: scoring.max_mismatch_penalty = (int32_t) std::atol(optarg); break;
			case 1005: scoring.n_penalty = (int32_t) std::atol(optarg); break;
			case 1006: scoring.gap_opening_penalty = (int32_t) std::atol(optarg); break;
			case 1007: scoring.gap_extension_penalty = (int32_t) std::atol(optarg); break;
			case 1008: tag_xn = optarg; break;
			case 1009: tag_xr = optarg; break;
			case 1010: tag_xp = optarg; break;
			case 1011: tag_yr = optarg; break;
			case 1012: tag_yp = optarg; break;
			case 1013: tag_xs = optarg; break;
			case 1014: tag_ys = optarg; break;
			case 1015: print_invocation = true; break;
			case 1016: print_pid = true; break;
			case 1017: print_reference_names = true; break;
			case 1018: threads = (unsigned) std::max(1, std::atoi(optarg)); break;
			case 1019: device = std::atol(optarg); break;
			case 1020: batch_records = (std::size_t) std::max(1L, std::atol(optarg)); break;
			case 1021:
				if (0 == std::strcmp(optarg, "gpu")) gpu_deflate = true;
				else if (0 == std::strcmp(optarg, "host")) gpu_deflate = false;
				else die("--bgzf-deflate expects gpu or host");
				break;
			case 'V': std::cout << "recalculate-mapq 0.1 (panvc3-mi355x)\n"; return 0;
			case 'h':
				std::cout << "Usage: panvc3_recalculate_mapq --alignments=alignments.bam > projected.sam\n"
							 "  (the options of the reference's recalculate-mapq/cmdline.ggo; in addition --device=INT, --threads=INT, --batch-records=INT and\n"
							 "  --bgzf-deflate=gpu|host: where the blocks of BAM output are compressed, on the GPU by default)\n";
				return 0;
			default: return EXIT_FAILURE;
		}
	}
	(void) verbose;
	if (print_invocation) { std::cerr << "Invocation:"; for (int i = 0; i < argc; ++i) std::cerr << ' ' << argv[i]; std::cerr << '\n'; }
	if (print_pid) std::cerr << "PID: " << getpid() << '\n';
	if (status_interval < 0) die("Status output interval must be non-negative.");			// main.cc:1312-1316
	if (scoring.min_mismatch_penalty < 0) die("Minimum mismatch penalty must be non-negative.");		// :1318-1322
	if (scoring.max_mismatch_penalty < 0) die("Maximum mismatch penalty must be non-negative.");
	if (scoring.n_penalty < 0) die("N character penalty must be non-negative.");
	if (scoring.gap_opening_penalty < 0) die("Gap opening penalty must be non-negative.");
	if (scoring.gap_extension_penalty < 0) die("Gap extension penalty must be non-negative.");
	for (std::string const *t : {&tag_xn, &tag_xr, &tag_xp, &tag_yr, &tag_yp, &tag_xs, &tag_ys})		// cmdline.ggo: "SAM tags need to match /[XYZ][A-Za-z0-9]/"
		if (!std::regex_match(*t, std::regex("^[XYZ][A-Za-z0-9]$"))) die("The given tag '" + *t + "' does not match the expected format.");

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
		append_sam_program_info(in_header, out_header, "panvc3.recalculate-mapq.", "PanVC 3 recalculate_mapq", argc, argv);		// main.cc:1291-1304
		std::unique_ptr <gpu_group_deflater> block_deflater;		// (before the writer, which uses it until it is closed)
		alignment_writer writer;
		writer.set_threads(threads);
		writer.open(output_path, output_bam, out_header);
		if (print_reference_names)
		{
			std::cerr << "Reference IDs:\n";
			for (std::size_t i = 0; i < in_header.refs.size(); ++i) std::cerr << i << '\t' << in_header.refs[i].name << '\n';
		}

		pvc3_context *ctx{};
		check(pvc3_context_create((int) device, &ctx));
		if (output_bam && gpu_deflate)
		{
			block_deflater = std::make_unique <gpu_group_deflater>((int) device, 512, 2);
			writer.attach(block_deflater.get());
		}
		std::cerr << "Processing the alignments…\n";

		uint16_t const k_xr = tag_key(tag_xr.c_str()), k_xp = tag_key(tag_xp.c_str()), k_yr = tag_key(tag_yr.c_str()), k_yp = tag_key(tag_yp.c_str());
		uint16_t const k_xn = tag_key(tag_xn.c_str()), k_as = tag_key("AS");
		pvc3_mapq_stats totals{};
		std::vector <pending_record> recs;
		std::vector <uint64_t> group_off{0};

		auto flush = [&]() {
			std::size_t const n = recs.size();
			if (0 == n) return;
			std::vector <uint32_t> ref_id(n), pos(n), mref(n), mpos(n), xr(n, PVC3_NO_POS), xp(n, PVC3_NO_POS), yr(n, PVC3_NO_POS), yp(n, PVC3_NO_POS), l_seq(n), seq_class(n);
			std::vector <double> score(n, -DBL_MAX);
			for (std::size_t i = 0; i < n; ++i)
			{
				bam_record const &r = recs[i].rec;
				// position::from_fields (main.cc:652-669): unset fields give the invalid position, negative values are an input error
				ref_id[i] = (r.ref_id() < 0) ? PVC3_NO_POS : (uint32_t) r.ref_id();
				pos[i] = (r.pos() < 0) ? PVC3_NO_POS : (uint32_t) r.pos();
				mref[i] = (r.next_ref_id() < 0) ? PVC3_NO_POS : (uint32_t) r.next_ref_id();
				mpos[i] = (r.next_pos() < 0) ? PVC3_NO_POS : (uint32_t) r.next_pos();
				l_seq[i] = r.l_seq();
				recs[i].has_as = false;
				recs[i].xn.clear();
				for (auto const &f : parse_aux(r.data.data() + r.aux_offset(), r.data.data() + r.data.size()))
				{
					uint16_t const k = tag_key(f.tag);
					int64_t v;
					auto tag_value = [&](std::vector <uint32_t> &dst) {
						if (!aux_int(f, v)) die(std::string("Record ‘") + r.read_name() + "’ had unexpected type for tag ‘" + f.tag[0] + f.tag[1] + "’.");		// tag_type_mismatch
						if (v < 0) die(std::string("Value of tag ‘") + f.tag[0] + f.tag[1] + "’ out of bounds in record with ID ‘" + r.read_name() + "’.");	// tag_value_out_of_bounds
						dst[i] = (uint32_t) v;
					};
					if (k == k_xr) tag_value(xr);
					else if (k == k_xp) tag_value(xp);
					else if (k == k_yr) tag_value(yr);
					else if (k == k_yp) tag_value(yp);
					else if (k == k_as) { if (aux_int(f, v)) { recs[i].has_as = true; recs[i].as_value = v; } }
					else if (k == k_xn && 'B' == f.type && f.size >= 5) recs[i].xn.assign(f.value + 5, f.value + f.size);		// B, subtype, count, then the bytes
				}
				if (recs[i].has_as) score[i] = (double) recs[i].as_value;
			}
			// equal SEQ inside a group (consulted for groups of unpaired records)
			for (std::size_t g = 0; g + 1 < group_off.size(); ++g)
			{
				std::map <std::string, uint32_t> classes;
				for (uint64_t i = group_off[g]; i < group_off[g + 1]; ++i)
				{
					bam_record const &r = recs[i].rec;
					std::string const key((char const *) r.seq4(), (r.l_seq() + 1) / 2);
					seq_class[i] = classes.emplace(key + '\0' + std::to_string(r.l_seq()), (uint32_t) classes.size()).first->second;
				}
			}
			if (rescore)
			{
				std::vector <uint64_t> cigar_off(n + 1, 0), seq_off(n + 1, 0), qual_off(n + 1, 0), xn_off(n + 1, 0);
				std::vector <uint32_t> cigar;
				std::vector <uint8_t> seq4, qual, xn;
				for (std::size_t i = 0; i < n; ++i)
				{
					bam_record const &r = recs[i].rec;
					for (uint16_t k = 0; k < r.n_cigar(); ++k) cigar.push_back(r.get <uint32_t>(r.cigar_offset() + 4u * k));
					seq4.insert(seq4.end(), r.seq4(), r.seq4() + (r.l_seq() + 1) / 2);
					if (r.has_qual()) qual.insert(qual.end(), r.qual(), r.qual() + r.l_seq());
					xn.insert(xn.end(), recs[i].xn.begin(), recs[i].xn.end());
					cigar_off[i + 1] = cigar.size(); seq_off[i + 1] = seq4.size(); qual_off[i + 1] = qual.size(); xn_off[i + 1] = xn.size();
				}
				std::vector <double> new_score(n);
				std::vector <int32_t> status(n);
				check(pvc3_rescore_alignments(ctx, n, cigar_off.data(), cigar.data(), seq_off.data(), seq4.data(), l_seq.data(), qual_off.data(), qual.data(), xn_off.data(), xn.data(), &scoring, new_score.data(), status.data()));
				for (std::size_t i = 0; i < n; ++i)
				{
					if (PVC3_RESCORE_CIGAR_OP == status[i]) die(std::string("Unexpected CIGAR operation in record ") + recs[i].rec.read_name() + " (rescoring needs = and X in place of M)");
					if (status[i]) die(std::string("the ") + tag_xn + " bit map or SEQ of record " + recs[i].rec.read_name() + " does not cover its CIGAR");
					score[i] = new_score[i];
				}
			}
			std::vector <uint8_t> mapq(n), emitted(group_off.size() - 1);
			pvc3_mapq_batch mb{};
			mb.n_groups = group_off.size() - 1; mb.group_off = group_off.data();
			mb.ref_id = ref_id.data(); mb.pos = pos.data(); mb.mate_ref_id = mref.data(); mb.mate_pos = mpos.data();
			mb.orig_ref_id = xr.data(); mb.orig_pos = xp.data(); mb.orig_mate_ref_id = yr.data(); mb.orig_mate_pos = yp.data();
			mb.score = score.data(); mb.l_seq = l_seq.data(); mb.seq_class = seq_class.data();
			pvc3_mapq_stats st{};
			check(pvc3_recalculate_mapq(ctx, &mb, mapq.data(), emitted.data(), &st));
			totals.total_alignments += st.total_alignments; totals.unpaired_alignments += st.unpaired_alignments;
			totals.reads_with_and_without_mate += st.reads_with_and_without_mate; totals.mate_not_found += st.mate_not_found;
			totals.reads_without_valid_position += st.reads_without_valid_position;

			bam_record w;
			std::map <uint16_t, std::vector <uint8_t>> tags;
			for (std::size_t g = 0; g + 1 < group_off.size(); ++g)
			{
				if (!emitted[g])
				{
					std::cerr << "WARNING: Read ‘" << recs[group_off[g]].rec.read_name() << "’ has both paired and unpaired alignment records; skipping.\n";		// main.cc:1042
					continue;
				}
				for (uint64_t i = group_off[g]; i < group_off[g + 1]; ++i)
				{
					bam_record &r = recs[i].rec;
					r.data[9] = mapq[i];
					if (!rescore) { writer.write(r); continue; }
					// cigar_alignment_scorer::operator() (main.cc:291-335): AS takes the new score; a previous AS moves to the
					// original-alignment-score tag and the new score is also stored as a float
					tags.clear();
					for (auto const &f : parse_aux(r.data.data() + r.aux_offset(), r.data.data() + r.data.size()))
					{
						std::vector <uint8_t> &v = tags[tag_key(f.tag)];
						v.assign({(uint8_t) f.tag[0], (uint8_t) f.tag[1], (uint8_t) f.type});
						v.insert(v.end(), f.value, f.value + f.size);
					}
					if (recs[i].has_as)
					{
						put_int_tag(tags[tag_key(tag_xs.c_str())], tag_xs.c_str(), recs[i].as_value);
						float const fs = (float) score[i];
						std::vector <uint8_t> &v = tags[tag_key(tag_ys.c_str())];
						v.assign({(uint8_t) tag_ys[0], (uint8_t) tag_ys[1], 'f'});
						v.insert(v.end(), (uint8_t const *) &fs, (uint8_t const *) &fs + 4);
					}
					put_int_tag(tags[k_as], "AS", (int64_t) (int32_t) score[i]);
					w.data.assign(r.data.begin(), r.data.begin() + (std::ptrdiff_t) r.aux_offset());
					for (auto const &kv : tags) w.data.insert(w.data.end(), kv.second.begin(), kv.second.end());
					writer.write(w);
				}
			}
			recs.clear();
			group_off.assign(1, 0);
		};

		bam_record rec;
		std::string current_name;
		std::size_t in_group = 0;
		while (reader.next(rec))
		{
			if (rec.flag() & 0x4) continue;		// unmapped: main.cc:1236-1238
			std::string const name(rec.read_name());
			if (in_group && name != current_name)
			{
				group_off.push_back(recs.size());
				in_group = 0;
				if (recs.size() >= batch_records) flush();
			}
			current_name = name;
			recs.emplace_back();
			recs.back().rec = rec;
			++in_group;
		}
		if (in_group) group_off.push_back(recs.size());
		flush();
		writer.close();
		pvc3_context_destroy(ctx);

		std::cerr << "Done.\n";		// main.cc:1266-1277
		std::cerr << "\tTotal alignments: " << totals.total_alignments << '\n';
		std::cerr << "\tUnpaired alignments: " << totals.unpaired_alignments << '\n';
		std::cerr << "\tRecords with mate missing: " << totals.mate_not_found << '\n';
		std::cerr << "\tReads with and without a mate: " << totals.reads_with_and_without_mate << '\n';
		std::cerr << "\tReads without valid positions: " << totals.reads_without_valid_position << '\n';
	}
	catch (std::exception const &e)
	{
		die(e.what());
	}
	return 0;
}
