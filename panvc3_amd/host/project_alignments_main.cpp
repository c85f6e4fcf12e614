// panvc3_project_alignments — the reference's command line and SAM/BAM-in, SAM/BAM-out surface over the MI355X hot path.
//
// Mirrors project-alignments/project_alignments.cc of the reference around the C ABI (include/pvc3.h):
//   options            project-alignments/cmdline.ggo:10-49
//   record filters     project_alignments.cc:528-570        (unmapped / QC-fail / duplicate / no RNAME / no POS / no SEQ are dropped)
//   header rewrite     :1092-1253 (process_headers), :1256-1268 + include/panvc3/utility.hh:92-143 (@PG)
//   entry lookup       :1271-1316 (fill_sequence_entries)
//   per record         :700-919: projection (GPU), tag filter :775-802, OA :805-844, XB / YB :848-861, XI :864-869,
//                      XR / XP :872-877, mate :880-911, POS / CIGAR / RNAME :915-918
// Differences: records are projected in large batches on the GPU instead of four at a time on one CPU thread (the XI value
// still follows the four-record chunks); the MSA index is this implementation's own file format; the whole destination FASTA
// is read up front.  SEQ / QUAL are normalised the way seqan3's default dna5 / phred42 alphabets do.
#include <getopt.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <future>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <regex>
#include <set>

#include <memory>

#include <sys/resource.h>

#include "../../include/pvc3.h"
#include "msa_index_file.hpp"
#include "sam_bam.hpp"
#include "gpu_deflate.hpp"
#include "device_records.hpp"

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

char const *const PRESERVED_TAGS[] = {	// project_alignments.cc:53-87
	"AM", "AS", "BC", "BZ", "CB", "CO", "CR", "CS", "CT", "CY", "E2", "FZ", "LB", "MI", "ML", "MM", "OA", "OQ", "OX", "PG", "PU", "QT", "QX", "RX", "TS"};

inline uint16_t tag_key(char const t[2]) { return (uint16_t) ((uint8_t) t[0] * 256 + (uint8_t) t[1]); }	// include/panvc3/sam_tag.hh:34-44

// The tags of one output record: what std::map <uint16_t, std::vector <uint8_t>> would hold, without its node per tag and
// record (the byte vectors keep their capacity from record to record).
class tag_set
{
	struct entry { uint16_t key; std::vector <uint8_t> data; };
	std::vector <entry> m_entries;
	std::size_t m_used{};
	std::vector <uint16_t> m_order;

public:
	void clear() { m_used = 0; }

	std::vector <uint8_t> &operator[](uint16_t key)
	{
		for (std::size_t i = 0; i < m_used; ++i)
			if (m_entries[i].key == key) return m_entries[i].data;
		if (m_used == m_entries.size()) m_entries.emplace_back();
		entry &e = m_entries[m_used++];
		e.key = key; e.data.clear();
		return e.data;
	}

	template <typename t_fn>
	void for_each_sorted(t_fn &&fn)
	{
		m_order.resize(m_used);
		for (std::size_t i = 0; i < m_used; ++i) m_order[i] = (uint16_t) i;
		std::sort(m_order.begin(), m_order.end(), [&](uint16_t a, uint16_t b) { return m_entries[a].key < m_entries[b].key; });
		for (uint16_t const i : m_order) fn(m_entries[i].data);
	}
};

// A byte of BAM SEQ read through seqan3::dna5: a code that is not A, C, G or T (1, 2, 4, 8) becomes N (15).
struct dna5_byte_table
{
	uint8_t v[256];
	dna5_byte_table()
	{
		auto canon = [](unsigned c) -> unsigned { return (1 == c || 2 == c || 4 == c || 8 == c) ? c : 15u; };
		for (unsigned b = 0; b < 256; ++b) v[b] = (uint8_t) ((canon(b >> 4) << 4) | canon(b & 0xf));
	}
};
static dna5_byte_table const dna5_bytes;

struct options
{
	char const *alignments{}, *msa_index{}, *reference{}, *reference_msa_id{}, *ref_order{}, *output_path{}, *realigned_ranges{};
	std::string separator{"_"};
	bool bam_input{}, output_bam{}, keep_duplicate_ranges{}, print_invocation{}, print_pid{}, io_only{}, debugging_output{};
	bool gpu_deflate{true};				// --bgzf-deflate=gpu|host: BAM output with the built-in encoder is compressed on the first GPU
	int record_path{-1};				// --record-path=gpu|host: -1 = gpu where it applies (BGZF BAM file in, BAM out with the built-in encoder on the GPU, one GPU)
	bool inflate_given{}, deflate_given{};		// --bgzf-inflate / --bgzf-deflate was given
	bool gpu_inflate{false};			// --bgzf-inflate=gpu|host: the blocks of BGZF input are inflated on the first GPU (7 GB/s: sixteen host threads are faster)
	std::vector <std::string> preserve_tags;
	std::string rname_tag{"XR"}, pos_tag{"XP"}, rnext_tag{"YR"}, pnext_tag{"YP"}, query_ranges_tag{"XB"}, ref_ranges_tag{"YB"}, rec_idx_tag;
	int gap_open{0}, gap_ext{-1}, compression_level{-1};
	int score_mode{PVC3_SCORE_FP64}, tiebreak_mode{0};		// --realign-score-type, --realign-tiebreak: what seqan3 does inside align_pairwise and no reference test pins (tests/parity_kit/)
	long status_interval{5};			// --status-output-interval, minutes; 0 disables
	std::vector <int> devices{0};		// --device / --devices: batches go to the GPUs in turn, output stays in input order
	uint64_t batch_size{UINT64_C(1) << 18};		// records per GPU batch: small enough for reading, projecting and writing of consecutive batches to overlap
	unsigned threads{default_io_threads()};		// BGZF inflate / deflate
};

double peak_rss_mb()
{
	struct rusage ru{};
	getrusage(RUSAGE_SELF, &ru);
	return (double) ru.ru_maxrss / 1024.0;
}

double cpu_seconds()		// user + system time of the whole process so far
{
	struct rusage ru{};
	getrusage(RUSAGE_SELF, &ru);
	return (double) ru.ru_utime.tv_sec + (double) ru.ru_stime.tv_sec + 1e-6 * ((double) ru.ru_utime.tv_usec + (double) ru.ru_stime.tv_usec);
}

std::string rss_kinds()		// resident now, by kind (/proc/self/status), and the CPU time split
{
	std::string out;
	if (FILE *f = std::fopen("/proc/self/status", "r"))
	{
		char line[256];
		while (std::fgets(line, sizeof(line), f))
			if (0 == std::strncmp(line, "RssAnon", 7) || 0 == std::strncmp(line, "RssFile", 7) || 0 == std::strncmp(line, "RssShmem", 8) || 0 == std::strncmp(line, "VmLck", 5) || 0 == std::strncmp(line, "VmPin", 5))
			{
				std::string l(line);
				while (!l.empty() && ('\n' == l.back() || ' ' == l.back())) l.pop_back();
				for (auto &c : l) if ('\t' == c) c = ' ';
				out += (out.empty() ? "" : ", ") + l;
			}
		std::fclose(f);
	}
	struct rusage ru{};
	getrusage(RUSAGE_SELF, &ru);
	out += "; user " + std::to_string((double) ru.ru_utime.tv_sec + 1e-6 * (double) ru.ru_utime.tv_usec) + " s, system " + std::to_string((double) ru.ru_stime.tv_sec + 1e-6 * (double) ru.ru_stime.tv_usec) + " s";
	return out;
}

double rss_mb()		// resident now (/proc/self/statm, pages)
{
	double mb = 0.0;
	if (FILE *f = std::fopen("/proc/self/statm", "r"))
	{
		unsigned long size = 0, resident = 0;
		if (2 == std::fscanf(f, "%lu %lu", &size, &resident)) mb = (double) resident * (double) sysconf(_SC_PAGESIZE) / (1024.0 * 1024.0);
		std::fclose(f);
	}
	return mb;
}

void usage()
{
	std::cout <<
		"Usage: panvc3_project_alignments --alignments=alignments.bam --msa-index=msa-index.dat --reference=ref.fa.gz --reference-msa-id=REF > projected.sam\n"
		"  -a, --alignments=filename            Alignment file path (stdin by default)\n"
		"  -B, --bam-input                      Read BAM from stdin\n"
		"  -m, --msa-index=filename             MSA index path (built by panvc3_index_msa of this package)\n"
		"  -r, --reference=filename             Reference sequences FASTA path (plain, gzip or bgzip)\n"
		"  -d, --reference-msa-id=STRING        Reference sequence ID in MSA index\n"
		"  -s, --ref-id-separator=STRING        Reference ID separator in input (default `_')\n"
		"  -u, --reference-order-input=filename Read output reference order from the given path, one identifier per line\n"
		"  -o, --output-path=path               Output path (stdout by default)\n"
		"  -b, --output-bam                     Output BAM instead of SAM\n"
		"      --status-output-interval=SHORT   Status output interval in minutes, zero to disable (default 5)\n"
		"  -l, --output-realigned-ranges=filename  Output the re-aligned ranges as TSV to the given path\n"
		"  -t, --preserve-tag=tag               Preserve the given SAM tag (may be repeated)\n"
		"      --original-rname-tag=XR --original-pos-tag=XP --original-rnext-tag=YR --original-pnext-tag=YP\n"
		"      --realigned-query-ranges-tag=XB --realigned-ref-ranges-tag=YB --record-index-tag=TAG\n"
		"  -q, --use-read-base-qualities        accepted, no effect (as in the reference)\n"
		"  -g, --gap-opening-cost=INT           (default 0)     -e, --gap-extension-cost=INT (default -1)\n"
		"      --device=INT                     GPU to use (default 0)\n"
		"      --devices=LIST                   GPUs to use, e.g. 0,1,2,3: batches go to them in turn (output order = input order)\n"
		"      --threads=INT                    host threads for (de)compression, parsing and rewriting\n"
		"      --compression-level=-1..9        BAM output: -1 (default) the built-in deflate encoders, 0..9 zlib at that level\n"
		"      --bgzf-deflate=gpu|host          where the built-in encoder runs: on the first GPU (default) or on the host threads\n"
		"      --bgzf-inflate=gpu|host          where the blocks of BGZF input (BAM, bgzip'ed SAM) are inflated: on the host threads (default)\n"
		"                                       or on the first GPU once it is up (for hosts with few cores: 7 GB/s)\n"
		"      --realign-score-type=double|int32, --realign-tiebreak=0..11\n"
		"                                       the two things seqan3 decides inside the realignment and no test of the reference pins: the\n"
		"                                       score type (default double) and which of several equally good alignments is returned\n"
		"                                       (default 0; include/pvc3.h, tests/parity_kit/README.md)\n"
		"      --record-path=gpu|host           where the records are cut out of the input and written anew: on the GPU (default for a BAM file\n"
		"                                       in and BAM out on one GPU: the blocks are inflated, the records found, projected, rewritten\n"
		"                                       and compressed without leaving it; there the GPU inflates as many blocks of a piece of the\n"
		"                                       file as it works on at a time and the host's threads the rest meanwhile, --bgzf-inflate=gpu\n"
		"                                       gives it all blocks and --bgzf-inflate=host none) or by the host's threads (every other\n"
		"                                       input and output)\n";
}

// process_headers (project_alignments.cc:1092-1253)
void process_headers(sam_header const &in, options const &opt, sam_header &out, std::vector <std::size_t> &ref_id_mapping)
{
	out.comments = in.comments;
	out.read_groups = in.read_groups;
	ref_id_mapping.assign(in.refs.size(), SIZE_MAX);

	std::vector <std::pair <std::string, std::string>> chr_seq;
	for (auto const &sq : in.refs)
	{
		auto const pos = sq.name.find_first_of(opt.separator);		// :1124 (find_first_of, quirk Q13)
		if (std::string::npos == pos) die("Separator '" + opt.separator + "' not found in reference ID '" + sq.name + "'.");
		chr_seq.emplace_back(sq.name.substr(0, pos), sq.name.substr(pos + 1));
	}

	std::map <std::string, std::pair <std::size_t, bool>> order;		// chr id -> (index, used)
	if (opt.ref_order)
	{
		std::set <std::string> unique_chr;
		for (auto const &cs : chr_seq) unique_chr.insert(cs.first);
		byte_reader in_order;
		in_order.open(opt.ref_order);
		std::string line;
		std::size_t idx = 0;
		while (in_order.getline(line))
		{
			std::string const id = line.substr(0, line.find('\t'));		// first tabulator: .fai input works
			if (!unique_chr.count(id)) { std::cerr << "WARNING: Identifier ‘" << id << "’ specified in reference name order but does not appear in the alignments.\n"; continue; }
			if (!order.emplace(id, std::make_pair(idx, false)).second) { std::cerr << "WARNING: Identifier ‘" << id << "’ specified in reference name order more than once.\n"; continue; }
			++idx;
		}
	}

	std::map <std::string, std::size_t> unique_ref_ids;
	std::vector <std::size_t> ref_seq_idxs;
	std::size_t next_idx = order.size();
	for (std::size_t i = 0; i < chr_seq.size(); ++i)
	{
		auto const &[chr_id, seq_id] = chr_seq[i];
		auto const it = unique_ref_ids.find(chr_id);
		std::size_t out_idx;
		if (unique_ref_ids.end() == it)
		{
			auto const oit = order.find(chr_id);
			if (order.end() == oit)
			{
				out_idx = next_idx++;
				if (!order.empty()) std::cerr << "WARNING: Output reference ID ‘" << chr_id << "’ not found in the output order, placing in the end.\n";
			}
			else { out_idx = oit->second.first; oit->second.second = true; }
			unique_ref_ids.emplace(chr_id, out_idx);
		}
		else out_idx = it->second;
		ref_id_mapping[i] = out_idx;
		if (seq_id == opt.reference_msa_id) ref_seq_idxs.push_back(i);
	}

	out.refs.assign(unique_ref_ids.size(), {});
	for (auto const &kv : unique_ref_ids) out.refs[kv.second].name = kv.first;
	for (auto const i : ref_seq_idxs)		// length + extra info of chr<sep><reference-msa-id> (:1223-1240)
	{
		auto &dst = out.refs[ref_id_mapping[i]];
		dst.length = in.refs[i].length;
		dst.extra = in.refs[i].extra;
	}
	for (auto const &kv : order)
		if (!kv.second.second) std::cerr << "WARNING: Identifier ‘" << kv.first << "’ was given in the output reference name order but was not found in the input.\n";
	out.index_refs();
}

void append_program_info(sam_header const &in, sam_header &out, int argc, char **argv)
{
	append_sam_program_info(in, out, "panvc3.project-alignments.", "PanVC 3 project_alignments", argc, argv);
}

// The records of one batch: their bytes back to back in one arena (no allocation per record).
struct batch_store
{
	std::vector <uint8_t> raw;				// the records' bytes (with whatever lay between them in the input: length fields, filtered records)
	std::vector <uint64_t> begin;			// [n] where record i starts in raw
	std::vector <uint32_t> len;				// [n]
	std::vector <uint64_t> input_index;		// index among all input records, filtered ones included
	std::size_t size() const { return input_index.size(); }
	bool empty() const { return input_index.empty(); }
	bam_view operator[](std::size_t i) const { return bam_view{raw.data() + begin[i], (std::size_t) len[i]}; }
	void clear() { raw.clear(); begin.clear(); len.clear(); input_index.clear(); }
	void swap(batch_store &o) { raw.swap(o.raw); begin.swap(o.begin); len.swap(o.len); input_index.swap(o.input_index); }
};

// A growable array in page-locked host memory (pvc3_host_alloc): what the pipelined entry point copies from and to without
// staging.  Contents are not kept when it grows.
template <typename t_value>
struct pinned_array
{
	t_value *p{};
	std::size_t cap{};
	pinned_array() = default;
	pinned_array(pinned_array const &) = delete;
	pinned_array &operator=(pinned_array const &) = delete;
	~pinned_array() { if (p) pvc3_host_free(p); }
	void reserve(std::size_t n)
	{
		if (n <= cap) return;
		if (p) { pvc3_host_free(p); p = nullptr; cap = 0; }
		std::size_t const want = n + n / 4 + 64;
		void *q{};
		check(pvc3_host_alloc(want * sizeof(t_value), &q), "pvc3_host_alloc");
		p = (t_value *) q; cap = want;
	}
	t_value *data() { return p; }
	t_value const *data() const { return p; }
	t_value &operator[](std::size_t i) { return p[i]; }
	t_value const &operator[](std::size_t i) const { return p[i]; }
};

// Finished batches on their way to the output file: the formatted bytes of the slices, in order.
struct write_job
{
	std::vector <std::string> slices;
};

}	// anonymous namespace


int main(int argc, char **argv)
{
	// The record path on the GPU keeps seven streams busy (the context's, three of the inflater, the compressor's ...); the HIP runtime
	// spreads a process's streams over four hardware queues unless told otherwise, and two streams that land on one queue take
	// turns: the compressed bytes then came back at 10 GB/s behind an inflate kernel instead of 40 (every other run, as the threads
	// that make the streams happened to be scheduled).  Read by the runtime when it starts; a value in the environment wins.
	setenv("GPU_MAX_HW_QUEUES", "8", 0);
	options opt;
	static option const long_opts[] = {
		{"alignments", required_argument, nullptr, 'a'}, {"bam-input", no_argument, nullptr, 'B'}, {"msa-index", required_argument, nullptr, 'm'},
		{"reference", required_argument, nullptr, 'r'}, {"reference-msa-id", required_argument, nullptr, 'd'}, {"ref-id-separator", required_argument, nullptr, 's'},
		{"reference-order-input", required_argument, nullptr, 'u'}, {"output-path", required_argument, nullptr, 'o'}, {"output-bam", no_argument, nullptr, 'b'},
		{"status-output-interval", required_argument, nullptr, 1000}, {"output-realigned-ranges", required_argument, nullptr, 'l'},
		{"keep-duplicate-ranges", no_argument, nullptr, 1001}, {"preserve-tag", required_argument, nullptr, 't'},
		{"original-rname-tag", required_argument, nullptr, 1002}, {"original-pos-tag", required_argument, nullptr, 1003},
		{"original-rnext-tag", required_argument, nullptr, 1004}, {"original-pnext-tag", required_argument, nullptr, 1005},
		{"realigned-query-ranges-tag", required_argument, nullptr, 1006}, {"realigned-ref-ranges-tag", required_argument, nullptr, 1007},
		{"record-index-tag", required_argument, nullptr, 1008}, {"use-read-base-qualities", no_argument, nullptr, 'q'},
		{"gap-opening-cost", required_argument, nullptr, 'g'}, {"gap-extension-cost", required_argument, nullptr, 'e'},
		{"print-invocation", no_argument, nullptr, 1009}, {"print-pid", no_argument, nullptr, 1010}, {"debugging-output", no_argument, nullptr, 1011},
		{"device", required_argument, nullptr, 1012}, {"batch-size", required_argument, nullptr, 1013}, {"io-only", no_argument, nullptr, 1014}, {"threads", required_argument, nullptr, 1015},
		{"devices", required_argument, nullptr, 1016}, {"compression-level", required_argument, nullptr, 1017}, {"bgzf-deflate", required_argument, nullptr, 1018}, {"bgzf-inflate", required_argument, nullptr, 1019}, {"record-path", required_argument, nullptr, 1020}, {"realign-score-type", required_argument, nullptr, 1021}, {"realign-tiebreak", required_argument, nullptr, 1022},
		{"help", no_argument, nullptr, 'h'}, {"version", no_argument, nullptr, 'V'}, {nullptr, 0, nullptr, 0}};
	int c;
	while (-1 != (c = getopt_long(argc, argv, "a:Bm:r:d:s:u:o:bl:t:qg:e:hV", long_opts, nullptr)))
	{
		switch (c)
		{
			case 'a': opt.alignments = optarg; break;
			case 'B': opt.bam_input = true; break;
			case 'm': opt.msa_index = optarg; break;
			case 'r': opt.reference = optarg; break;
			case 'd': opt.reference_msa_id = optarg; break;
			case 's': opt.separator = optarg; break;
			case 'u': opt.ref_order = optarg; break;
			case 'o': opt.output_path = optarg; break;
			case 'b': opt.output_bam = true; break;
			case 1000: opt.status_interval = std::atol(optarg); break;
			case 'l': opt.realigned_ranges = optarg; break;
			case 1001: opt.keep_duplicate_ranges = true; break;
			case 't': opt.preserve_tags.emplace_back(optarg); break;
			case 1002: opt.rname_tag = optarg; break;
			case 1003: opt.pos_tag = optarg; break;
			case 1004: opt.rnext_tag = optarg; break;
			case 1005: opt.pnext_tag = optarg; break;
			case 1006: opt.query_ranges_tag = optarg; break;
			case 1007: opt.ref_ranges_tag = optarg; break;
			case 1008: opt.rec_idx_tag = optarg; break;
			case 'q': break;		// stored and never consulted in the reference (quirk Q1)
			case 'g': opt.gap_open = std::atoi(optarg); break;
			case 'e': opt.gap_ext = std::atoi(optarg); break;
			case 1009: opt.print_invocation = true; break;
			case 1010: opt.print_pid = true; break;
			case 1011: opt.debugging_output = true; break;
			case 1012: opt.devices.assign(1, std::atoi(optarg)); break;
			case 1016:
			{
				opt.devices.clear();
				for (char const *q = optarg; *q; ) { char *end{}; opt.devices.push_back((int) std::strtol(q, &end, 10)); if (end == q) die("--devices expects a comma-separated list of GPU numbers"); q = (',' == *end) ? end + 1 : end; }
				if (opt.devices.empty()) die("--devices expects a comma-separated list of GPU numbers");
				break;
			}
			case 1017: opt.compression_level = std::max(-1, std::min(9, std::atoi(optarg))); break;
			case 1018:
				if (0 == std::strcmp(optarg, "gpu")) opt.gpu_deflate = true;
				else if (0 == std::strcmp(optarg, "host")) opt.gpu_deflate = false;
				else die("--bgzf-deflate expects gpu or host");
				opt.deflate_given = true;
				break;
			case 1019:
				if (0 == std::strcmp(optarg, "gpu")) opt.gpu_inflate = true;
				else if (0 == std::strcmp(optarg, "host")) opt.gpu_inflate = false;
				else die("--bgzf-inflate expects gpu or host");
				opt.inflate_given = true;
				break;
			case 1020:
				if (0 == std::strcmp(optarg, "gpu")) opt.record_path = 1;
				else if (0 == std::strcmp(optarg, "host")) opt.record_path = 0;
				else die("--record-path expects gpu or host");
				break;
			case 1021:
				if (0 == std::strcmp(optarg, "double")) opt.score_mode = PVC3_SCORE_FP64;
				else if (0 == std::strcmp(optarg, "int32")) opt.score_mode = PVC3_SCORE_INT32_TRUNC;
				else die("--realign-score-type expects double or int32");
				break;
			case 1022:
				opt.tiebreak_mode = std::atoi(optarg);
				if (opt.tiebreak_mode < 0 || opt.tiebreak_mode >= PVC3_TIEBREAK_MODES) die("--realign-tiebreak expects 0 .. 11");
				break;
			case 1013: opt.batch_size = std::strtoull(optarg, nullptr, 10); break;
			case 1014: opt.io_only = true; break;
			case 1015: opt.threads = (unsigned) std::max(1, std::atoi(optarg)); break;
			case 'h': usage(); return 0;
			case 'V': std::cout << "project-alignments 0.1 (panvc3-mi355x)\n"; return 0;
			default: return EXIT_FAILURE;
		}
	}
	if (opt.print_invocation) { std::cerr << "Invocation:"; for (int i = 0; i < argc; ++i) std::cerr << ' ' << argv[i]; std::cerr << '\n'; }
	if (opt.print_pid) std::cerr << "PID: " << getpid() << '\n';
	if (!opt.io_only && (!opt.msa_index || !opt.reference || !opt.reference_msa_id))
		die("--msa-index, --reference and --reference-msa-id are required");

	// Tag checks (project_alignments.cc:1362-1376).
	{
		std::regex const xyz("^[XYZ][A-Za-z0-9]$"), any("^[A-Za-z0-9]{2}$");
		for (auto const *t : {&opt.rname_tag, &opt.pos_tag, &opt.rnext_tag, &opt.pnext_tag, &opt.query_ranges_tag, &opt.ref_ranges_tag, &opt.rec_idx_tag})
			if (!t->empty() && !std::regex_match(*t, xyz)) die("The given tag “" + *t + "” does not match the expected format.");
		for (auto const &t : opt.preserve_tags)
			if (!std::regex_match(t, any)) die("The given tag “" + t + "” does not match the expected format.");
	}
	std::set <uint16_t> preserved;
	for (auto const *t : PRESERVED_TAGS) preserved.insert(tag_key(t));
	for (auto const &t : opt.preserve_tags) preserved.insert(tag_key(t.c_str()));

	try
	{
		auto const t_start = std::chrono::steady_clock::now();
		double const epoch_start = std::chrono::duration <double>(std::chrono::system_clock::now().time_since_epoch()).count();
		// ---- input + headers -------------------------------------------------------------------------------------------------
		// (declared before the reader, which may be using it until it is closed)
		std::unique_ptr <gpu_group_inflater> block_inflater;
		alignment_reader reader;
		std::future <void> inflater_coming;		// (behind the reader: the task that makes the inflater attaches it to the reader)
		reader.set_threads(opt.threads);
		{
			bool is_bam = opt.bam_input;
			if (opt.alignments)
			{
				std::string const p(opt.alignments);
				is_bam = p.size() >= 4 && 0 == p.compare(p.size() - 4, 4, ".bam");
			}
			// Where the record path on the GPU may apply the file is read anew, raw, by that path's own reader: this one is then only
			// there for the header, and reads little of the file (pieces of 1 MB on one thread, not of 32 MB on all).
			bool header_only = is_bam && opt.alignments && 0 != opt.record_path && opt.output_bam && opt.compression_level < 0 && opt.gpu_deflate && 1 == opt.devices.size() && !opt.io_only;
			if (header_only)
			{
				// (what byte_reader::open takes for a BGZF file: a gzip member whose only extra field is BC)
				uint8_t first[18]{};
				FILE *f = std::fopen(opt.alignments, "rb");
				header_only = f && sizeof(first) == std::fread(first, 1, sizeof(first), f) && 0x1f == first[0] && 0x8b == first[1] && 8 == first[2] && 4 == first[3] && 6 == first[10] && 0 == first[11]
					&& 'B' == first[12] && 'C' == first[13] && 2 == first[14] && 0 == first[15];
				if (f) std::fclose(f);
			}
			if (header_only) { reader.set_threads(1); reader.set_piece_bytes(std::size_t(1) << 20); }
			reader.open(opt.alignments, is_bam);
		}
		sam_header const &in_header = reader.header;
		sam_header out_header;
		std::vector <std::size_t> ref_id_mapping;

		if (opt.io_only)
		{
			// Format conversion only (SAM <-> BAM round trips of the I/O layer; no index, no GPU).
			out_header = in_header;
			alignment_writer writer;
			writer.set_threads(opt.threads);
			writer.open(opt.output_path, opt.output_bam, out_header);
			bam_record rec;
			while (reader.next(rec)) writer.write(rec);
			writer.close();
			return 0;
		}

		process_headers(in_header, opt, out_header, ref_id_mapping);
		append_program_info(in_header, out_header, argc, argv);

		// ---- index, reference, GPUs: prepare_devices(), called once the reader thread is at work ------------------------------------
		pvc3_index *index{};
		std::vector <uint32_t> src_entries(in_header.refs.size()), dst_entries(out_header.refs.size(), UINT32_MAX);
		std::vector <pvc3_context *> device_contexts;
		// BAM output with the built-in encoder: the blocks are compressed on the first GPU, 1024 at a time (64 MiB), one group
		// being filled while the other is on the GPU (bgzf.hpp group_deflater; include/pvc3_bgzf.h).
		std::unique_ptr <gpu_group_deflater> block_deflater;
		std::future <std::unique_ptr <gpu_group_deflater>> deflater_coming;
		bool const want_gpu_deflate = opt.gpu_deflate && opt.output_bam && opt.compression_level < 0;
		double t_startup = 0.0, t_contexts_up = 0.0, t_deflater_up = 0.0, t_inflater_up = 0.0, t_files_read = 0.0, inflater_kernel_s = 0.0;
		uint64_t inflater_groups = 0;
		bool aux_codecs = true;		// the reader's GPU inflater and the writer's GPU compressor (not for the record path on the GPU, which has its own)
		std::function <void()> while_contexts_come;		// run on this thread once the files are read, before the contexts are waited for
		std::function <void()> after_contexts;		// run on the contexts' thread once they are up (the record path on the GPU makes its codecs there)
		auto prepare_devices = [&]() {
			// The GPU contexts come up on a thread of their own (the HIP runtime takes a few tenths of a second to start) while the
			// index file and the reference are read.
			std::future <std::vector <pvc3_context *>> contexts_coming = std::async(std::launch::async, [&]() {
				std::vector <pvc3_context *> made;
				for (int const dev : opt.devices)
				{
					pvc3_context *ctx{};
					if (pvc3_context_create(dev, &ctx)) throw std::runtime_error(std::string("pvc3_context_create: ") + pvc3_last_error());
					made.push_back(ctx);
				}
				t_contexts_up = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
				if (after_contexts) after_contexts();
				// The BGZF compressor's page-locked staging takes a while to allocate and is not needed before the first batch has been
				// rewritten: it comes up on a thread of its own, behind the contexts (the HIP runtime is up by now).
				// The inflater of the input's blocks likewise; the reader, which has been inflating on the host threads since the process
				// began, shares its pieces with it from the next one on: pieces of 32 MB of the file, 160 MB of page-locked output each.
				if (aux_codecs && opt.gpu_inflate)
					inflater_coming = std::async(std::launch::async, [&]() {
						block_inflater = std::make_unique <gpu_group_inflater>(opt.devices.front(), (std::size_t) 65 << 20, (std::size_t) 160 << 20, 4096, 2);
						reader.attach(block_inflater.get());
						t_inflater_up = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
					});
				if (aux_codecs && want_gpu_deflate)
					deflater_coming = std::async(std::launch::async, [&]() {
						char const *group = std::getenv("PVC3_BGZF_GROUP_BLOCKS");		// tests: small groups, many of them
						std::size_t const blocks = group ? (std::size_t) std::max(1, std::atoi(group)) : 512;
						auto made_deflater = std::make_unique <gpu_group_deflater>(opt.devices.front(), blocks, 2);
						t_deflater_up = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
						return made_deflater;
					});
				return made;
			});
			msa_index_file msa;
			msa.load(opt.msa_index);
			check(pvc3_index_create(&index), "pvc3_index_create");
			std::map <std::pair <std::string, std::string>, uint32_t> entry_ids;
			auto entry_for = [&](std::string const &chr_id, std::string const &seq_id) -> uint32_t {
				auto const key = std::make_pair(chr_id, seq_id);
				auto const it = entry_ids.find(key);
				if (entry_ids.end() != it) return it->second;
				auto const *ce = msa.find_chr(chr_id);
				if (!ce) die("Did not find an entry for chromosome ID “" + chr_id + "” in the MSA index.");
				auto const *se = msa_index_file::find_seq(*ce, seq_id);
				if (!se) die("Did not find an entry for sequence ID “" + seq_id + "” in the MSA index.");
				uint32_t id{};
				check(pvc3_index_add_bits(index, se->gap_words.data(), se->n_cols, &id), "pvc3_index_add_bits");
				entry_ids.emplace(key, id);
				return id;
			};
			// fill_sequence_entries (:1271-1316); the separator is matched as a substring here (quirk Q13).
			for (std::size_t i = 0; i < in_header.refs.size(); ++i)
			{
				auto const &name = in_header.refs[i].name;
				auto const pos = name.find(opt.separator);
				if (std::string::npos == pos) die("Unable to find the separator “" + opt.separator + "” in the RNAME “" + name + "”.");
				std::string const chr_id = name.substr(0, pos), seq_id = name.substr(pos + 1);
				std::size_t const out_idx = ref_id_mapping[i];
				if (UINT32_MAX == dst_entries[out_idx]) dst_entries[out_idx] = entry_for(chr_id, opt.reference_msa_id);
				src_entries[i] = entry_for(chr_id, seq_id);
				// (the pairs are known here: the replicas' paired blocks and position windows are made for exactly them)
				check(pvc3_index_set_destination(index, src_entries[i], dst_entries[out_idx]), "pvc3_index_set_destination");
			}
			{
				// The destination contigs only, through <reference>.fai (+ .gzi) when the indexes are there
				// (libpanvc3/compressed_fasta_reader.cc:163-248 requires them); the whole file otherwise.
				std::vector <std::pair <std::string, std::vector <char>>> fasta;
				std::vector <std::string> wanted;
				for (std::size_t o = 0; o < out_header.refs.size(); ++o)
					if (UINT32_MAX != dst_entries[o]) wanted.push_back(out_header.refs[o].name);
				bool const indexed = read_fasta_contigs(opt.reference, wanted, opt.threads, fasta);
				if (!indexed) read_fasta(opt.reference, fasta);
				if (opt.debugging_output) std::cerr << "Reference: " << fasta.size() << " sequences read " << (indexed ? "through the .fai index" : "from the whole file") << ".\n";
				for (std::size_t o = 0; o < out_header.refs.size(); ++o)
				{
					if (UINT32_MAX == dst_entries[o]) continue;
					auto const it = std::find_if(fasta.begin(), fasta.end(), [&](auto const &kv) { return kv.first == out_header.refs[o].name; });
					if (fasta.end() == it) die("Unable to load sequence ‘" + out_header.refs[o].name + "’ from the input FASTA.");
					check(pvc3_index_set_reference(index, dst_entries[o], it->second.data(), it->second.size()), "pvc3_index_set_reference");
				}
			}
			t_files_read = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
			if (while_contexts_come) while_contexts_come();
			device_contexts = contexts_coming.get();
		};

		alignment_writer writer;
		writer.set_threads(opt.threads);
		writer.open(opt.output_path, opt.output_bam, out_header, opt.compression_level);

		// ---- the record path on the GPU (device_records.hpp): BGZF BAM file in, BAM out, one GPU --------------------------------------------
		bool const device_records_apply = reader.is_bgzf_file() && opt.alignments && opt.output_bam && opt.compression_level < 0 && opt.gpu_deflate && 1 == opt.devices.size();
		if (1 == opt.record_path && !device_records_apply)
			die("--record-path=gpu needs a BAM file as input (--alignments=x.bam), BAM output (--output-bam) with the built-in encoder on the GPU, and one GPU");
		if (0 != opt.record_path && device_records_apply)
		{
			uint64_t const header_bytes = reader.bam_header_bytes();
			reader.release();		// the header has been read: the file is read anew, raw, by the record path's own reader
			aux_codecs = false;
			device_records_options dro;
			dro.inflate_mode = !opt.inflate_given ? 0 : (opt.gpu_inflate ? 1 : 2);
			if (char const *e = std::getenv("PVC3_BGZF_CHUNK")) { std::size_t const v = (std::size_t) std::strtoull(e, nullptr, 10); if (v >= 1024) dro.piece_compressed = v; }
			if (char const *e = std::getenv("PVC3_RECORDS_STREAM")) { std::size_t const v = (std::size_t) std::strtoull(e, nullptr, 10); if (v >= 65536) dro.piece_stream = v; }
			if (char const *e = std::getenv("PVC3_RECORDS_BLOCKS")) { dro.piece_blocks = (uint32_t) std::max(1, std::atoi(e)); }
			if (char const *e = std::getenv("PVC3_RECORDS_INFLATE_IN_FLIGHT")) { dro.inflate_in_flight = (uint32_t) std::max(1, std::atoi(e)); }		// (experiments: room for the other kernels next to the inflater's waves)
			if (char const *e = std::getenv("PVC3_RECORDS_GPU_BLOCKS")) { dro.gpu_blocks = (uint32_t) std::max(1, std::atoi(e)); }		// (tests: a host's share in small files)
			// The two codecs are mostly page-locked memory, which takes the runtime a tenth of a second or two to hand out: they are made
			// on threads of their own as soon as the GPU is up, while the index file and the FASTA are read.
			device_records_sizes const sizes = device_records_sizes::of(dro.piece_compressed, dro.piece_stream);
			std::future <pvc3_bgzf_inflater *> records_inflater;
			std::future <pvc3_bgzf_dev *> records_deflater;
			std::string inflater_message, deflater_message;
			int const first_device = opt.devices.front();
			after_contexts = [&]() {
				records_inflater = std::async(std::launch::async, [&]() { pvc3_bgzf_inflater *h{}; if (pvc3_bgzf_inflater_create(first_device, sizes.max_in, 1, sizes.max_blocks_in, PVC3_RECORDS_SLOTS, &h)) inflater_message = pvc3_bgzf_last_error(); return h; });
				records_deflater = std::async(std::launch::async, [&]() { pvc3_bgzf_dev *h{}; if (pvc3_bgzf_dev_create(first_device, sizes.max_blocks_out, device_records_out_slots, &h)) deflater_message = pvc3_bgzf_last_error(); return h; });
			};
			// the longest read decides between 128- and 256-byte position windows: the first records of the file are looked at on the host
			// (while the HIP runtime is still starting)
			uint32_t longest = 0;
			while_contexts_come = [&]() {
				alignment_reader peek;
				peek.set_threads(1);
				peek.set_piece_bytes(std::size_t(1) << 20);
				peek.open(opt.alignments, true);
				bam_record rec;
				for (unsigned k = 0; k < 2000 && peek.next(rec); ++k) longest = std::max(longest, rec.l_seq());
			};
			prepare_devices();
			double const t_prepared = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
			double const rss_prepared = rss_mb();
			check(pvc3_context_set_option(device_contexts[0], "window_read_length", longest ? longest : 150), "pvc3_context_set_option");
			double const t_peeked = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
			if (std::getenv("PVC3_RECORDS_TRACE")) std::cerr << "resident " << rss_mb() << " MB before the index goes to the GPU\n";
			if (pvc3_context_upload_index(device_contexts[0], index)) die(std::string("pvc3_context_upload_index: ") + pvc3_last_error());
			t_startup = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
			if (std::getenv("PVC3_RECORDS_TRACE")) std::cerr << "resident " << rss_mb() << " MB with the index on the GPU\n";

			dro.path = opt.alignments; dro.device = opt.devices.front(); dro.ctx = device_contexts[0];
			dro.take_inflater = [&]() { pvc3_bgzf_inflater *h = records_inflater.get(); if (!h) die("pvc3_bgzf_inflater_create: " + inflater_message); return h; };
			dro.take_deflater = [&]() { pvc3_bgzf_dev *h = records_deflater.get(); if (!h) die("pvc3_bgzf_dev_create: " + deflater_message); return h; };
			dro.params = pvc3_params{opt.gap_open, opt.gap_ext, opt.score_mode, opt.tiebreak_mode};
			dro.header_bytes = header_bytes; dro.want_ranges = nullptr != opt.realigned_ranges; dro.threads = opt.threads;
			std::vector <uint32_t> r_dst(in_header.refs.size()), name_off(in_header.refs.size() + 1, 0);
			std::vector <int32_t> r_out(in_header.refs.size());
			std::string names;
			for (std::size_t i = 0; i < in_header.refs.size(); ++i)
			{
				r_dst[i] = dst_entries[ref_id_mapping[i]]; r_out[i] = (int32_t) ref_id_mapping[i];
				names += in_header.refs[i].name; name_off[i + 1] = (uint32_t) names.size();
			}
			std::vector <uint16_t> kept_tags(preserved.begin(), preserved.end());
			pvc3_records_config &cfg = dro.config;
			cfg.n_refs = (uint32_t) in_header.refs.size(); cfg.src_entry = src_entries.data(); cfg.dst_entry = r_dst.data(); cfg.out_ref_id = r_out.data();
			cfg.ref_names = names.data(); cfg.ref_name_off = name_off.data(); cfg.preserved_tags = kept_tags.data(); cfg.n_preserved_tags = (uint32_t) kept_tags.size();
			auto set_tag = [](char (&dst)[2], std::string const &name) { dst[0] = name.empty() ? 0 : name[0]; dst[1] = name.empty() ? 0 : name[1]; };
			set_tag(cfg.rname_tag, opt.rname_tag); set_tag(cfg.pos_tag, opt.pos_tag); set_tag(cfg.rnext_tag, opt.rnext_tag); set_tag(cfg.pnext_tag, opt.pnext_tag);
			set_tag(cfg.query_ranges_tag, opt.query_ranges_tag); set_tag(cfg.ref_ranges_tag, opt.ref_ranges_tag); set_tag(cfg.rec_idx_tag, opt.rec_idx_tag);
			writer.flush_blocks();		// the header: blocks of its own
			dro.write_blocks = [&](void const *p, std::size_t n) { writer.write_blocks(p, n); };
			double next_status = 60.0 * (double) opt.status_interval;
			dro.progress = [&](uint64_t records_done) {
				if (opt.status_interval <= 0) return;
				double const running = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
				if (running < next_status) return;
				std::cerr << "Time spent processing: " << (uint64_t) running << " s; processed " << records_done << " records";
				if (records_done) std::cerr << " (in " << (running * 1e6 / (double) records_done) << " µs / record)";
				std::cerr << ".\n";
				next_status = running + 60.0 * (double) opt.status_interval;
			};
			dro.leave_objects = true;		// (this function ends in _Exit)
			double const rss_begin = rss_mb();
			double const t_records_begin = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
			device_records_result const dr = run_device_records(dro);
			double const t_batches_done = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
			writer.close();
			double const t_closed = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
			if (opt.realigned_ranges)
			{
				auto ranges = dr.ranges;
				if (!opt.keep_duplicate_ranges)
				{
					std::sort(ranges.begin(), ranges.end());
					ranges.erase(std::unique(ranges.begin(), ranges.end()), ranges.end());
				}
				FILE *f = std::fopen(opt.realigned_ranges, "w");
				if (!f) die(std::string("unable to open ") + opt.realigned_ranges);
				std::fputs("Location\tLength\n", f);
				for (auto const &rr : ranges) std::fprintf(f, "%u\t%u\n", rr.first, rr.second);
				std::fclose(f);
			}
			double const secs = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
			std::cerr << "Done. Time spent: " << secs << " s.\n";
			std::cerr << "\tMatched reads:         " << dr.matched << '\n';
			std::cerr << "\tReference ID missing:  " << dr.ref_id_missing << '\n';
			std::cerr << "\tFlags not matched:     " << dr.flags_not_matched << '\n';
			std::cerr << "\tSequence missing:      " << dr.seq_missing << '\n';
			std::cerr << "\tRe-aligned ranges:     " << dr.n_runs << '\n';
			if (opt.debugging_output)
			{
				std::cerr << "\tRecord path on the GPU: " << dr.records << " records in " << dr.batches << " pieces; " << dr.blocks_in << " blocks in (" << dr.blocks_host << " of them the host's share, " << dr.blocks_refused << " inflated on the host instead), "
						  << dr.bytes_in << " bytes of records in, " << dr.bytes_out << " bytes of file out; kernels: inflate " << dr.inflate_kernel_s << " s, deflate " << dr.deflate_kernel_s << " s\n";
				std::cerr << "\tStages (s): start-up " << t_startup << ", reader thread " << dr.t_reader << ", codecs and buffers " << dr.t_setup << ", main: waiting for the reader " << dr.t_wait_read << ", waiting for the inflaters " << dr.t_inflate_wait << " (host threads at work " << dr.t_host_inflate << ")"
						  << ", records found and scattered " << dr.t_frame << ", projected " << dr.t_project << ", rewritten " << dr.t_rewrite << ", waiting for the writer " << dr.t_wait_writer << ", writer thread " << dr.t_writer
						  << " (waiting for the compressor " << dr.t_writer_wait << ", fetching " << dr.t_writer_fetch << ", writing " << dr.t_writer_file << ")\n";
				std::cerr << "\tTimeline (s): GPU contexts " << t_contexts_up << ", index file and FASTA " << t_files_read << ", both " << t_prepared << ", first reads looked at " << t_peeked << ", index on the GPUs (start-up done) " << t_startup
						  << ", record path begun " << t_records_begin << " (its last pieces written in " << dr.t_drain << ", all of it " << dr.t_total << "), last batch rewritten " << t_batches_done << ", output closed " << t_closed << "; wall clock at the start " << std::fixed << epoch_start << ", now "
						  << std::chrono::duration <double>(std::chrono::system_clock::now().time_since_epoch()).count() << std::defaultfloat
						  << "; resident set (MB): with the contexts up " << rss_prepared << ", before the record path " << rss_begin << ", with its objects made " << dr.rss_setup_mb << ", peak " << peak_rss_mb() << "; CPU time (s): " << cpu_seconds() << " (" << rss_kinds() << ")\n";
			}
			if (opt.debugging_output && std::getenv("PVC3_RECORDS_TRACE"))
			{
				std::cerr << "\tPieces (s since the record path began: to the inflater, taken up, inflated, records found, projected, rewritten):\n";
				for (auto const &t : dr.piece_times) std::cerr << "\t\t" << t[0] << ' ' << t[1] << ' ' << t[2] << ' ' << t[3] << ' ' << t[4] << ' ' << t[5] << '\n';
			}
			bool any_removed = false;
			for (auto const c : dr.removed_tags) any_removed = any_removed || c;
			if (any_removed)
			{
				std::cerr << "\tRemoved tags:\n";
				for (std::size_t key = 0; key < dr.removed_tags.size(); ++key)
					if (dr.removed_tags[key]) std::cerr << "\t\t" << (char) (key >> 8) << (char) (key & 0xff) << ": " << dr.removed_tags[key] << '\n';
			}
			std::cout.flush(); std::cerr.flush(); std::fflush(nullptr);
			if (std::getenv("PVC3_EXIT_PROFILE")) std::exit(0);		// (under rocprofv3: its output is written by an exit handler, which _Exit would skip)
			std::_Exit(0);
		}

		// The output file is written by a thread of its own (BGZF deflate is most of the CPU time of a BAM-to-BAM run): batch k
		// is compressed and written while batch k + 1 is rewritten and batch k + 2 is read.
		std::mutex wq_mutex;
		std::condition_variable wq_cv;
		std::deque <write_job> wq;
		bool wq_done = false;
		std::exception_ptr writer_error;
		double tm_write = 0.0;
		std::thread writer_thread([&]() {
			try
			{
				for (;;)
				{
					write_job job;
					{
						std::unique_lock <std::mutex> lock(wq_mutex);
						wq_cv.wait(lock, [&]() { return wq_done || !wq.empty(); });
						if (wq.empty()) break;
						job = std::move(wq.front());
						wq.pop_front();
					}
					wq_cv.notify_all();
					auto const t_write = std::chrono::steady_clock::now();
					for (auto const &bytes : job.slices) writer.write_formatted(bytes.data(), bytes.size());
					tm_write += std::chrono::duration <double>(std::chrono::steady_clock::now() - t_write).count();
				}
			}
			catch (...) { writer_error = std::current_exception(); std::lock_guard <std::mutex> lock(wq_mutex); wq.clear(); wq_done = true; wq_cv.notify_all(); }
		});
		struct join_writer { std::thread &t; std::mutex &m; std::condition_variable &cv; bool &done; ~join_writer() { { std::lock_guard <std::mutex> l(m); done = true; } cv.notify_all(); if (t.joinable()) t.join(); } } writer_joiner{writer_thread, wq_mutex, wq_cv, wq_done};

		// ---- records ----------------------------------------------------------------------------------------------------------------
		// Three kinds of threads work on different batches at the same time (project_alignments.cc:600-627 is one loop):
		//   the reader      inflates the input and frames the records of batch k + 2 .. (read_batch, below);
		//   this thread     fills the page-locked arrays of batch k and hands it to its GPU (pvc3_pipeline_submit: upload, kernels
		//                   and download are asynchronous), then takes the results of batch k - 2 x GPUs and rewrites its records;
		//   the writer      deflates and writes batch k - 2 x GPUs - 1.
		// Batch k goes to GPU k mod D, every GPU has two batches of its own in flight, and the batches are retired in the order they
		// were read, so the output does not depend on the number of GPUs or threads.
		struct { uint64_t flags_not_matched{}, ref_id_missing{}, seq_missing{}, matched_reads{}; } st;
		std::map <uint16_t, uint64_t> removed_tag_counts;
		std::vector <std::pair <uint32_t, uint32_t>> realigned_ranges;
		uint64_t n_realigned_ranges = 0;
		pvc3_params const params{opt.gap_open, opt.gap_ext, opt.score_mode, opt.tiebreak_mode};

		// --debugging-output: where the wall clock goes (seconds, summed over the batches)
		struct stage_times { double read{}, wait_read{}, fill{}, gpu{}, rewrite{}, wait_write{}, write{}; } tm;
		auto now = []() { return std::chrono::steady_clock::now(); };
		auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration <double>(std::chrono::steady_clock::now() - t0).count(); };
		uint64_t records_done = 0;
		double next_status = 60.0 * (double) opt.status_interval;
		uint64_t input_index = 0;
		bool eof = false;
		bam_record scratch;
		struct slice_state		// per-thread scratch and results of one slice of a batch
		{
			std::string bytes, oa;
			tag_set tags;											// written in the order of the tag keys, as seqan3's sam_tag_dictionary
			std::vector <aux_field> aux;
			std::vector <std::pair <uint32_t, uint32_t>> ranges;
			std::map <uint16_t, uint64_t> removed;
			bam_record w;
			uint64_t n_ranges{};
		};
		std::vector <slice_state> slices;

		// Fills a batch with records that pass the filters; a multiple of the reference's CHUNK_SIZE (4) unless the input ends.
		// Runs on a helper thread while the previous batch is on the GPU / being rewritten.
		auto read_batch = [&](batch_store &dst) {
			auto const t_read = now();
			struct add_time { double &acc; std::chrono::steady_clock::time_point t0; ~add_time() { acc += std::chrono::duration <double>(std::chrono::steady_clock::now() - t0).count(); } } timer{tm.read, t_read};
			dst.clear();
			// Whether a record passes the filters (project_alignments.cc:528-570); counts it either way.
			auto keeps = [&](bam_view const &rec) -> bool {
				uint16_t const fl = rec.flag();
				if (fl & (0x4 | 0x200 | 0x400)) { ++st.flags_not_matched; return false; }		// unmapped, QC fail, duplicate (:528-537)
				if (rec.ref_id() < 0) { ++st.ref_id_missing; return false; }						// :546-551
				if (rec.pos() < 0) { ++st.flags_not_matched; return false; }						// :553-564
				if (0 == rec.l_seq()) { ++st.seq_missing; return false; }							// :566-570
				if ((std::size_t) rec.ref_id() >= src_entries.size()) throw std::runtime_error("record with a reference id outside the header");
				++st.matched_reads;
				return true;
			};
			auto wanted = [&]() { return dst.size() < opt.batch_size || (dst.size() & 3); };
			while (!eof && wanted())
			{
				// BAM from a BGZF file: the whole records that lie in the inflated piece at hand are framed in place and copied
				// to the arena in one go (length fields and filtered records included; they are just not listed).
				uint8_t const *p{};
				std::size_t avail = 0;
				if (reader.bam_span(p, avail))
				{
					std::size_t used = 0;
					std::size_t const base = dst.raw.size();
					while (wanted() && avail - used >= 4)
					{
						int32_t block_size;
						std::memcpy(&block_size, p + used, 4);
						if (block_size < 32) throw io_error("truncated BAM record");
						if (avail - used < 4 + (std::size_t) block_size) break;
						bam_view const rec{p + used + 4, (std::size_t) block_size};
						if (rec.aux_offset() > rec.size()) throw io_error("malformed BAM record: its name, CIGAR, SEQ and QUAL exceed block_size");
						uint64_t const idx = input_index++;
						if (keeps(rec))
						{
							dst.begin.push_back(base + used + 4); dst.len.push_back((uint32_t) block_size);
							dst.input_index.push_back(idx);
						}
						used += 4 + (std::size_t) block_size;
					}
					if (used)
					{
						dst.raw.insert(dst.raw.end(), p, p + used);
						reader.consume(used);
						continue;
					}
				}
				// one record the slow way: text input, or a record that straddles two pieces
				std::size_t const at = dst.raw.size();
				if (!reader.next_append(dst.raw, scratch)) { eof = true; break; }
				bam_view const rec{dst.raw.data() + at, dst.raw.size() - at};
				uint64_t const idx = input_index++;
				if (!keeps(rec)) { dst.raw.resize(at); continue; }
				dst.begin.push_back(at); dst.len.push_back((uint32_t) (dst.raw.size() - at));
				dst.input_index.push_back(idx);
			}
		};

		// The reader runs ahead by up to two batches; the stores it fills are handed back when their records have been rewritten.
		// (At most two filled stores wait in the queue: a reader far ahead of the GPUs would only hold memory.)
		std::mutex rq_mutex;
		std::condition_variable rq_cv;
		std::deque <std::unique_ptr <batch_store>> rq, rq_free;
		bool rq_done = false;
		std::exception_ptr reader_error;
		for (std::size_t i = 0; i < 2 * opt.devices.size() + 3; ++i) rq_free.emplace_back(new batch_store);		// two per GPU in flight, two read ahead, one being read
		std::thread reader_thread([&]() {
			try
			{
				for (;;)
				{
					std::unique_ptr <batch_store> bs;
					{
						std::unique_lock <std::mutex> lock(rq_mutex);
						rq_cv.wait(lock, [&]() { return rq_done || (!rq_free.empty() && rq.size() < 2); });
						if (rq_done) return;
						bs = std::move(rq_free.front()); rq_free.pop_front();
					}
					read_batch(*bs);
					bool const last = bs->empty();
					{
						std::lock_guard <std::mutex> lock(rq_mutex);
						if (!last) rq.push_back(std::move(bs));
						if (last) rq_done = true;
					}
					rq_cv.notify_all();
					if (last) return;
				}
			}
			catch (...) { reader_error = std::current_exception(); std::lock_guard <std::mutex> lock(rq_mutex); rq_done = true; rq_cv.notify_all(); }
		});
		struct join_reader { std::thread &t; std::mutex &m; std::condition_variable &cv; bool &done; ~join_reader() { { std::lock_guard <std::mutex> l(m); done = true; } cv.notify_all(); if (t.joinable()) t.join(); } } reader_joiner{reader_thread, rq_mutex, rq_cv, rq_done};

		// ---- index, reference, GPUs (the reader is already at work) ---------------------------------------------------------------
		prepare_devices();
		double const t_prepared = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
		double const rss_prepared = rss_mb();
		{
			// The position windows of the index replicas are sized for the reads: the longest of the first batch (which the
			// reader has had the whole start-up for) decides between 128- and 256-byte windows.
			uint32_t longest = 0;
			{
				std::unique_lock <std::mutex> lock(rq_mutex);
				rq_cv.wait(lock, [&]() { return rq_done || !rq.empty(); });
				if (!rq.empty())
					for (std::size_t i = 0; i < rq.front()->size(); ++i) longest = std::max(longest, (*rq.front())[i].l_seq());
			}
			if (reader_error) std::rethrow_exception(reader_error);
			if (longest)
				for (pvc3_context *ctx : device_contexts) check(pvc3_context_set_option(ctx, "window_read_length", longest), "pvc3_context_set_option");
		}
		double const t_first_batch = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
		// The index is replicated on every GPU, all of them at the same time.
		{
			std::vector <std::future <int>> uploads;
			std::vector <std::string> errors(device_contexts.size());
			for (std::size_t d = 0; d < device_contexts.size(); ++d)
				uploads.push_back(std::async(std::launch::async, [&, d]() { int const rc = pvc3_context_upload_index(device_contexts[d], index); if (rc) errors[d] = pvc3_last_error(); return rc; }));
			for (std::size_t d = 0; d < uploads.size(); ++d)
				if (uploads[d].get()) die("pvc3_context_upload_index: " + errors[d]);
		}

		std::vector <pvc3_context *> const &contexts = device_contexts;
		std::size_t const n_dev = contexts.size();
		t_startup = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
		double const rss_startup = rss_mb();

		// One slot per batch in flight: the records, the page-locked SoA arrays (include/pvc3.h) and the page-locked results.
		struct slot_state
		{
			std::unique_ptr <batch_store> store;
			pinned_array <uint32_t> b_src_pos, b_src_entry, b_dst_entry, b_flags, b_cigar, b_l_seq, b_mate_se, b_mate_de, b_mate_pos;
			pinned_array <uint64_t> b_cigar_off, b_seq_off, b_qual_off;
			pinned_array <uint8_t> b_seq4, b_qual;
			pinned_array <uint32_t> o_dst_pos, o_mate, o_cigar;
			pinned_array <int32_t> o_status;
			pinned_array <uint64_t> o_cigar_off, o_run_off;
			pinned_array <pvc3_run> o_runs;
			uint64_t n{};
			std::size_t device{};
			bool in_flight{}, pipelined{};
		};
		std::vector <slot_state> slot_states(2 * n_dev);
		std::vector <pvc3_pipeline *> pipelines(n_dev, nullptr);
		struct destroy_pipelines { std::vector <pvc3_pipeline *> &v; ~destroy_pipelines() { for (auto *p : v) if (p) pvc3_pipeline_destroy(p); } } pipeline_guard{pipelines};
		pvc3_pipeline_caps pipe_caps{};

		// Batch arrays of the slot's records (the reference builds one task per record, project_alignments.cc:627, 699).
		auto fill = [&](slot_state &sl) {
			batch_store const &batch = *sl.store;
			uint64_t const n = sl.n;
			auto &b_src_pos = sl.b_src_pos; auto &b_src_entry = sl.b_src_entry; auto &b_dst_entry = sl.b_dst_entry; auto &b_flags = sl.b_flags; auto &b_cigar = sl.b_cigar;
			auto &b_l_seq = sl.b_l_seq; auto &b_mate_se = sl.b_mate_se; auto &b_mate_de = sl.b_mate_de; auto &b_mate_pos = sl.b_mate_pos;
			auto &b_cigar_off = sl.b_cigar_off; auto &b_seq_off = sl.b_seq_off; auto &b_qual_off = sl.b_qual_off; auto &b_seq4 = sl.b_seq4; auto &b_qual = sl.b_qual;
			// SoA batch (include/pvc3.h): offsets first (one cheap pass), then the slices are filled by all threads.
			b_src_pos.reserve(n); b_src_entry.reserve(n); b_dst_entry.reserve(n); b_flags.reserve(n); b_l_seq.reserve(n);
			b_mate_se.reserve(n); b_mate_de.reserve(n); b_mate_pos.reserve(n);
			b_cigar_off.reserve(n + 1); b_seq_off.reserve(n + 1); b_qual_off.reserve(n + 1);
			// The sizes are read out of the records by all threads (a record header is a cache miss of its own), the prefix sums
			// are one pass over three dense arrays.
			std::size_t const n_fill = std::min <std::size_t>(n, 4 * (std::size_t) opt.threads);
			b_cigar_off[0] = b_seq_off[0] = b_qual_off[0] = 0;
			parallel_for(n_fill, opt.threads, [&](std::size_t si) {
				for (uint64_t i = n * si / n_fill; i < n * (si + 1) / n_fill; ++i)
				{
					bam_view const r = batch[i];
					b_cigar_off[i + 1] = r.n_cigar();
					b_seq_off[i + 1] = (r.l_seq() + 1) / 2;
					b_qual_off[i + 1] = r.has_qual() ? r.l_seq() : 0;
				}
			});
			for (uint64_t i = 0; i < n; ++i)
			{
				b_cigar_off[i + 1] += b_cigar_off[i];
				b_seq_off[i + 1] += b_seq_off[i];
				b_qual_off[i + 1] += b_qual_off[i];
			}
			b_cigar.reserve(b_cigar_off[n] + 64); b_seq4.reserve(b_seq_off[n] + 64); b_qual.reserve(b_qual_off[n] + 64);
			parallel_for(n_fill, opt.threads, [&](std::size_t si) {
				for (uint64_t i = n * si / n_fill; i < n * (si + 1) / n_fill; ++i)
				{
					bam_view const r = batch[i];
					b_src_pos[i] = (uint32_t) r.pos();
					b_src_entry[i] = src_entries[(std::size_t) r.ref_id()];
					b_dst_entry[i] = dst_entries[ref_id_mapping[(std::size_t) r.ref_id()]];
					b_l_seq[i] = r.l_seq();
					uint32_t fl = r.has_qual() ? PVC3_FLAG_HAS_QUAL : 0;
					b_mate_se[i] = 0; b_mate_de[i] = 0; b_mate_pos[i] = 0;
					if (r.next_ref_id() >= 0 && r.next_pos() >= 0)
					{
						if ((std::size_t) r.next_ref_id() >= src_entries.size()) throw std::runtime_error("record with a mate reference id outside the header");
						fl |= PVC3_FLAG_HAS_MATE;
						b_mate_se[i] = src_entries[(std::size_t) r.next_ref_id()];
						b_mate_de[i] = dst_entries[ref_id_mapping[(std::size_t) r.next_ref_id()]];
						b_mate_pos[i] = (uint32_t) r.next_pos();
					}
					b_flags[i] = fl;
					uint8_t const *p = r.bytes();
					std::memcpy(b_cigar.data() + b_cigar_off[i], p + r.cigar_offset(), 4 * (std::size_t) r.n_cigar());
					std::memcpy(b_seq4.data() + b_seq_off[i], p + r.seq_offset(), (r.l_seq() + 1) / 2);
					if (r.has_qual()) std::memcpy(b_qual.data() + b_qual_off[i], p + r.qual_offset(), r.l_seq());
				}
			});
		};
		auto batch_in_of = [&](slot_state &sl) {
			pvc3_batch_in in{};
			in.n = sl.n; in.src_pos = sl.b_src_pos.data(); in.src_entry = sl.b_src_entry.data(); in.dst_entry = sl.b_dst_entry.data(); in.flags = sl.b_flags.data();
			in.cigar_off = sl.b_cigar_off.data(); in.cigar = sl.b_cigar.data(); in.seq_off = sl.b_seq_off.data(); in.l_seq = sl.b_l_seq.data(); in.seq4 = sl.b_seq4.data();
			in.qual_off = sl.b_qual_off.data(); in.qual = sl.b_qual.data();
			in.mate_src_entry = sl.b_mate_se.data(); in.mate_dst_entry = sl.b_mate_de.data(); in.mate_pos = sl.b_mate_pos.data();
			return in;
		};
		auto batch_out_of = [&](slot_state &sl) {
			pvc3_batch_out out{};
			out.dst_pos = sl.o_dst_pos.data(); out.mate_dst_pos = sl.o_mate.data(); out.status = sl.o_status.data();
			out.out_cigar_off = sl.o_cigar_off.data(); out.out_cigar = sl.o_cigar.data(); out.out_cigar_cap = sl.o_cigar.cap;
			out.run_off = sl.o_run_off.data(); out.runs = sl.o_runs.data(); out.run_cap = sl.o_runs.cap;
			return out;
		};
		// The batch on its own, start to finish, with device arrays made for it: for a batch that does not fit the pipeline's
		// capacities or whose results do not fit the arrays they were to land in.
		auto project_alone = [&](slot_state &sl) {
			pvc3_context *ctx = contexts[sl.device];
			// The same GPU's next batch is already queued on its pipeline, and all batches of a context share its alignment scratch
			// and trace buffers: that batch's kernels run first.
			if (pipelines[sl.device]) check(pvc3_pipeline_synchronize(pipelines[sl.device]), "pvc3_pipeline_synchronize");
			pvc3_batch_in const in = batch_in_of(sl);
			pvc3_dbatch *db{};
			pvc3_batch_stats stats{};
			check(pvc3_dbatch_create(ctx, &in, &db), "pvc3_dbatch_create");
			check(pvc3_dbatch_project(ctx, db, &params), "pvc3_dbatch_project");
			check(pvc3_dbatch_finish(ctx, db, &stats), "pvc3_dbatch_finish");
			sl.o_cigar.reserve(stats.out_cigar_total + 1); sl.o_runs.reserve(stats.n_runs + 1);
			pvc3_batch_out out = batch_out_of(sl);
			check(pvc3_dbatch_download(ctx, db, &out), "pvc3_dbatch_download");
			pvc3_dbatch_destroy(db);
		};
		auto submit = [&](slot_state &sl) {
			uint64_t const n = sl.n;
			pvc3_batch_in const in = batch_in_of(sl);
			if (!pipelines[0])
			{
				// Capacities from the first batch, with room: twice its CIGAR operations and SEQ bytes per record.
				pipe_caps.max_records = std::max <uint64_t>(n, opt.batch_size + 4);
				pipe_caps.max_cigar_ops = std::max <uint64_t>(2 * in.cigar_off[n] * pipe_caps.max_records / std::max <uint64_t>(n, 1), 8 * pipe_caps.max_records) + 1024;		// (the first batch's operations per record, scaled to a full batch — not the full batch's floor scaled again: a first batch of a dozen records asked for 180 GB)
				pipe_caps.max_seq_bytes = 2 * in.seq_off[n] * pipe_caps.max_records / std::max <uint64_t>(n, 1) + 4096;
				pipe_caps.max_qual_bytes = 0;		// QUAL stays in the page-locked arrays and is fetched for the realigned records only
				pipe_caps.depth = 2;
				for (std::size_t d = 0; d < n_dev; ++d)
					check(pvc3_pipeline_create(contexts[d], &pipe_caps, &pipelines[d]), "pvc3_pipeline_create");
			}
			sl.o_dst_pos.reserve(n); sl.o_mate.reserve(n); sl.o_status.reserve(n); sl.o_cigar_off.reserve(n + 1); sl.o_run_off.reserve(n + 1);
			sl.o_cigar.reserve(std::max <uint64_t>(4 * in.cigar_off[n], 16 * n)); sl.o_runs.reserve(n + 1024);
			static bool const tiny_results = nullptr != std::getenv("PVC3_CLI_TINY_RESULTS");		// tests: every batch's results outgrow their arrays
			sl.pipelined = n <= pipe_caps.max_records && in.cigar_off[n] <= pipe_caps.max_cigar_ops && in.seq_off[n] <= pipe_caps.max_seq_bytes;
			if (sl.pipelined)
			{
				pvc3_batch_out out = batch_out_of(sl);
				if (tiny_results) { out.out_cigar_cap = std::min <uint64_t>(out.out_cigar_cap, n); out.run_cap = 0; }
				check(pvc3_pipeline_submit(pipelines[sl.device], &in, &out, &params), "pvc3_pipeline_submit");
			}
			sl.in_flight = true;
		};

		// Results of the slot's batch (waits for them), its records rewritten, the bytes on their way to the file.
		auto retire = [&](slot_state &sl) {
			auto t_stage = now();
			if (sl.pipelined)
			{
				pvc3_batch_stats stats{};
				int const rc = pvc3_pipeline_wait(pipelines[sl.device], &stats);
				if (PVC3_E_CAPACITY == rc) project_alone(sl);		// more operations or runs than the arrays were sized for
				else check(rc, "pvc3_pipeline_wait");
			}
			else
				project_alone(sl);
			sl.in_flight = false;
			tm.gpu += since(t_stage); t_stage = now();
			batch_store const &batch = *sl.store;
			uint64_t const n = sl.n;
			auto const &o_dst_pos = sl.o_dst_pos; auto const &o_mate = sl.o_mate; auto const &o_cigar = sl.o_cigar; auto const &o_status = sl.o_status;
			auto const &o_cigar_off = sl.o_cigar_off; auto const &o_run_off = sl.o_run_off; auto const &o_runs = sl.o_runs;

			// Rewrite the records: contiguous slices of the batch on separate threads, each into its own byte string; the strings
			// are written in slice order, so the output does not depend on the number of threads.
			std::size_t const n_slices = std::min <std::size_t>(n, 4 * (std::size_t) opt.threads);
			if (slices.size() < n_slices) slices.resize(n_slices);
			parallel_for(n_slices, opt.threads, [&](std::size_t si) {
			slice_state &S = slices[si];
			S.bytes.clear(); S.ranges.clear(); S.removed.clear(); S.n_ranges = 0;
			tag_set &tags = S.tags;
			std::vector <aux_field> &aux = S.aux;
			std::string &oa = S.oa;
			for (uint64_t i = n * si / n_slices; i < n * (si + 1) / n_slices; ++i)
			{
				bam_view const r = batch[i];
				if (o_status[i])
					throw std::runtime_error(std::string("record ") + r.read_name() + " (input record " + std::to_string(batch.input_index[i]) + ") cannot be projected: status " + std::to_string(o_status[i])
						+ " (the reference asserts here: position or CIGAR inconsistent with the index)");
				int32_t const ref_id = r.ref_id();
				uint32_t const *cig = o_cigar.data() + o_cigar_off[i];
				uint64_t const n_cig = o_cigar_off[i + 1] - o_cigar_off[i];
				if (n_cig > 65535) throw std::runtime_error("more than 65535 CIGAR operations after projection");
				pvc3_run const *runs = o_runs.data() + o_run_off[i];
				uint64_t const n_runs = o_run_off[i + 1] - o_run_off[i];
				S.n_ranges += n_runs;
				if (opt.realigned_ranges)
					for (uint64_t k = 0; k < n_runs; ++k) S.ranges.emplace_back(runs[k].ref_position, runs[k].ref_length);

				// -- tags: keep the preserved ones (:775-802), remember NM
				tags.clear();
				bool have_nm = false; int64_t nm = 0;
				std::string old_oa;
				parse_aux(r.bytes() + r.aux_offset(), r.bytes() + r.size(), aux);
				for (auto const &f : aux)
				{
					uint16_t const key = tag_key(f.tag);
					if ('N' == f.tag[0] && 'M' == f.tag[1] && aux_scalar_size(f.type) && 'A' != f.type && 'f' != f.type)
					{
						have_nm = true;
						switch (f.type)
						{
							case 'c': nm = (int8_t) f.value[0]; break; case 'C': nm = f.value[0]; break;
							case 's': { int16_t v; std::memcpy(&v, f.value, 2); nm = v; break; } case 'S': { uint16_t v; std::memcpy(&v, f.value, 2); nm = v; break; }
							case 'i': { int32_t v; std::memcpy(&v, f.value, 4); nm = v; break; } default: { uint32_t v; std::memcpy(&v, f.value, 4); nm = v; break; }
						}
					}
					if (!preserved.count(key)) { ++S.removed[key]; continue; }
					if ('O' == f.tag[0] && 'A' == f.tag[1] && 'Z' == f.type) { old_oa.assign((char const *) f.value, f.size - 1); continue; }
					std::vector <uint8_t> &v = tags[key];
					v.assign({(uint8_t) f.tag[0], (uint8_t) f.tag[1], (uint8_t) f.type});
					v.insert(v.end(), f.value, f.value + f.size);
				}
				auto put_int = [&](std::string const &name, int64_t v) {
					if (name.empty()) return;
					std::vector <uint8_t> &d = tags[tag_key(name.c_str())];
					d.clear();
					append_int_aux(d, name.c_str(), v);
				};
				auto put_ranges = [&](std::string const &name, bool query) {
					if (name.empty() || 0 == n_runs) return;
					std::vector <uint8_t> &d = tags[tag_key(name.c_str())];
					d.assign({(uint8_t) name[0], (uint8_t) name[1], 'B', 'I'});
					uint32_t const cnt = (uint32_t) (2 * n_runs); uint8_t b[4]; std::memcpy(b, &cnt, 4); d.insert(d.end(), b, b + 4);
					for (uint64_t k = 0; k < n_runs; ++k)
					{
						uint32_t const lo = query ? runs[k].query_location : runs[k].ref_position;
						uint32_t const hi = lo + (query ? runs[k].query_length : runs[k].ref_length);
						std::memcpy(b, &lo, 4); d.insert(d.end(), b, b + 4); std::memcpy(b, &hi, 4); d.insert(d.end(), b, b + 4);
					}
				};
				// -- OA (:805-844): RNAME,POS (0-based as the reference prints it),strand,CIGAR,MAPQ,NM;
				oa = old_oa;
				oa += in_header.refs[(std::size_t) ref_id].name; oa += ','; append_decimal(oa, r.pos()); oa += ',';
				oa += (r.flag() & 0x10) ? '-' : '+'; oa += ',';
				for (uint16_t k = 0; k < r.n_cigar(); ++k)
				{
					uint32_t const cc = r.get <uint32_t>(r.cigar_offset() + 4u * k);
					append_decimal(oa, cc >> 4); oa += CIGAR_CHARS[(cc & 0xf) < 9 ? (cc & 0xf) : 0];
				}
				oa += ','; append_decimal(oa, r.mapq()); oa += ',';
				if (have_nm) append_decimal(oa, nm);
				oa += ';';
				{
					std::vector <uint8_t> &d = tags[tag_key("OA")];
					d.assign({'O', 'A', 'Z'});
					d.insert(d.end(), oa.begin(), oa.end()); d.push_back(0);
				}
				put_ranges(opt.query_ranges_tag, true);		// :848-861
				put_ranges(opt.ref_ranges_tag, false);
				if (!opt.rec_idx_tag.empty())				// :864-869: the same value for the (up to) four records of a chunk
				{
					uint64_t const chunk_begin = i & ~UINT64_C(3), chunk_end = std::min <uint64_t>(chunk_begin + 4, n);
					uint64_t const xi = batch.input_index[chunk_end - 1] - (chunk_end - chunk_begin) + 1;
					if (xi <= INT32_MAX) put_int(opt.rec_idx_tag, (int64_t) xi);
				}
				put_int(opt.rname_tag, ref_id);				// :872-877
				put_int(opt.pos_tag, r.pos());

				// -- mate (:880-911)
				int32_t new_next_ref = -1, new_next_pos = -1;
				if (r.next_ref_id() >= 0)
				{
					new_next_ref = (int32_t) ref_id_mapping[(std::size_t) r.next_ref_id()];
					put_int(opt.rnext_tag, r.next_ref_id());
					if (r.next_pos() >= 0)
					{
						if (PVC3_NO_MATE == o_mate[i]) throw std::runtime_error(std::string("the mate position of ") + r.read_name() + " lies outside its sequence");
						new_next_pos = (int32_t) o_mate[i];
						put_int(opt.pnext_tag, r.next_pos());
					}
				}

				// -- the new record (:915-918)
				bam_record &w = S.w;
				std::size_t const name_end = 32 + r.l_read_name();
				w.data.assign(r.bytes(), r.bytes() + name_end);
				w.set <int32_t>(0, (int32_t) ref_id_mapping[(std::size_t) ref_id]);
				w.set <int32_t>(4, (int32_t) o_dst_pos[i]);
				w.set <uint16_t>(12, (uint16_t) n_cig);
				w.set <int32_t>(20, new_next_ref);
				w.set <int32_t>(24, new_next_pos);
				int64_t span = 0;
				for (uint64_t k = 0; k < n_cig; ++k) { uint32_t const op = cig[k] & 0xf; if (0 == op || 2 == op || 3 == op || 7 == op || 8 == op) span += cig[k] >> 4; }
				w.set <uint16_t>(10, reg2bin(o_dst_pos[i], (int64_t) o_dst_pos[i] + (span ? span : 1)));
				w.data.insert(w.data.end(), (uint8_t const *) cig, (uint8_t const *) (cig + n_cig));
				// SEQ through seqan3::dna5 (non-ACGT becomes N), QUAL through phred42 (saturates at 41)
				std::size_t const seq_at = w.data.size();
				{
					std::size_t const n_bytes = (r.l_seq() + 1) / 2;
					w.data.resize(seq_at + n_bytes);
					uint8_t const *src = r.seq4();
					uint8_t *dst = w.data.data() + seq_at;
					for (std::size_t k = 0; k < n_bytes; ++k) dst[k] = dna5_bytes.v[src[k]];
					if (r.l_seq() & 1) dst[n_bytes - 1] = (uint8_t) ((dst[n_bytes - 1] & 0xf0) | (src[n_bytes - 1] & 0x0f));	// the unused low nibble stays as it was
				}
				std::size_t const qual_at = w.data.size();
				w.data.insert(w.data.end(), r.qual(), r.qual() + r.l_seq());
				if (r.has_qual())
				{
					uint8_t *q = w.data.data() + qual_at;
					for (uint32_t k = 0; k < r.l_seq(); ++k) q[k] = (q[k] > 41) ? 41 : q[k];
				}
				tags.for_each_sorted([&](std::vector <uint8_t> const &v) { w.data.insert(w.data.end(), v.begin(), v.end()); });
				writer.format(w, S.bytes);
			}
			});
			write_job job;
			job.slices.resize(n_slices);
			for (std::size_t si = 0; si < n_slices; ++si)
			{
				slice_state &S = slices[si];
				job.slices[si].swap(S.bytes);
				n_realigned_ranges += S.n_ranges;
				realigned_ranges.insert(realigned_ranges.end(), S.ranges.begin(), S.ranges.end());
				for (auto const &kv : S.removed) removed_tag_counts[kv.first] += kv.second;
			}
			tm.rewrite += since(t_stage); t_stage = now();
			if (deflater_coming.valid())		// the first batch for the writer thread: from here on the GPU compresses the blocks
			{
				try
				{
					block_deflater = deflater_coming.get();
					writer.attach(block_deflater.get());
				}
				catch (std::exception const &e)
				{
					// asked for: an error; the default: the host's encoder writes the same records
					if (opt.deflate_given) throw;
					std::cerr << "WARNING: the BGZF compressor on the GPU could not be made (" << e.what() << "); the output's blocks are compressed by the host's threads.\n";
				}
			}
			{
				std::unique_lock <std::mutex> lock(wq_mutex);
				wq_cv.wait(lock, [&]() { return wq_done || wq.size() < 2; });		// at most two finished batches wait for the file
				if (writer_error) std::rethrow_exception(writer_error);
				wq.push_back(std::move(job));
			}
			wq_cv.notify_all();
			tm.wait_write += since(t_stage);
			// the store goes back to the reader
			sl.store->clear();
			{
				std::lock_guard <std::mutex> lock(rq_mutex);
				rq_free.push_back(std::move(sl.store));
			}
			rq_cv.notify_all();
		};

		uint64_t batch_no = 0;
		for (;; ++batch_no)
		{
			auto t_stage = now();
			std::unique_ptr <batch_store> bs;
			{
				std::unique_lock <std::mutex> lock(rq_mutex);
				rq_cv.wait(lock, [&]() { return rq_done || !rq.empty(); });
				if (!rq.empty()) { bs = std::move(rq.front()); rq.pop_front(); }
			}
			rq_cv.notify_all();		// room in the queue: the reader may go on
			if (reader_error) std::rethrow_exception(reader_error);
			tm.wait_read += since(t_stage);
			if (!bs) break;
			// (--bgzf-inflate=gpu and no inflater: an error as soon as it is known, not after the last batch)
			if (inflater_coming.valid() && std::future_status::ready == inflater_coming.wait_for(std::chrono::seconds(0))) inflater_coming.get();
			slot_state &sl = slot_states[batch_no % slot_states.size()];
			if (sl.in_flight) retire(sl);		// batch_no - 2 x GPUs: the same GPU's older batch
			// The reference's periodic status line (project_alignments.cc:446-481): time spent, records processed, µs per record.
			if (opt.status_interval > 0)
			{
				double const running = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
				if (running >= next_status)
				{
					std::cerr << "Time spent processing: " << (uint64_t) running << " s; processed " << records_done << " records";
					if (records_done) std::cerr << " (in " << (running * 1e6 / (double) records_done) << " µs / record)";
					std::cerr << ".\n";
					next_status = running + 60.0 * (double) opt.status_interval;
				}
			}
			sl.store = std::move(bs);
			sl.n = sl.store->size();
			sl.device = batch_no % n_dev;
			records_done += sl.n;
			t_stage = now();
			fill(sl);
			tm.fill += since(t_stage); t_stage = now();
			submit(sl);
			tm.gpu += since(t_stage);
		}
		// what is still in flight, oldest first
		for (uint64_t k = (batch_no > slot_states.size()) ? (batch_no - slot_states.size()) : 0; k < batch_no; ++k)
		{
			slot_state &sl = slot_states[k % slot_states.size()];
			if (sl.in_flight) retire(sl);
		}
		double const t_batches_done = since(t_start);
		double const rss_batches_done = rss_mb();
		for (auto *&pl : pipelines) if (pl) { pvc3_pipeline_destroy(pl); pl = nullptr; }
		{
			std::lock_guard <std::mutex> lock(wq_mutex);
			wq_done = true;
		}
		wq_cv.notify_all();
		// While the writer finishes the last batches and the file is closed, everything else this process holds is given back on a
		// thread of its own: a process that has used the GPU pays at exit for the memory it still holds (0.07-0.09 s per GB of
		// ordinary memory: tools/calib/exit_cost_hip.py), and that is time the caller waits.
		bool const profile_exit = nullptr != std::getenv("PVC3_EXIT_PROFILE");
		if (reader_thread.joinable()) reader_thread.join();		// (it has returned: the queue was empty and done)
		if (inflater_coming.valid()) inflater_coming.get();		// (--bgzf-inflate=gpu and no inflater: an error, not a silent run on the host threads)
		std::thread teardown_thread;
		if (!profile_exit)
			teardown_thread = std::thread([&]() {
				// Ordinary memory only, two things at a time.  (Page-locked memory is left to the exit: giving it back here takes as
				// long as the exit would, and hipHostFree holds the runtime up for the writer, whose last group is on the GPU.)
				auto quietly = [](auto &&f) { return std::async(std::launch::async, [f]() { try { f(); } catch (...) {} }); };
				auto a = quietly([&]() {
					{
						std::lock_guard <std::mutex> lock(rq_mutex);
						rq.clear(); rq_free.clear();
					}
					std::vector <slice_state>().swap(slices);
				});
				auto b = quietly([&]() {
					reader.release();
					if (block_inflater) { inflater_groups = block_inflater->groups(); inflater_kernel_s = block_inflater->kernel_seconds(); }
					pvc3_index_destroy(index); index = nullptr;		// the host's copy of the index: the contexts are not used again
				});
				a.get(); b.get();
			});
		struct join_teardown { std::thread &t; ~join_teardown() { if (t.joinable()) t.join(); } } teardown_joiner{teardown_thread};
		writer_thread.join();
		if (writer_error) std::rethrow_exception(writer_error);
		double const t_writer_joined = since(t_start);
		writer.close();
		double const t_closed = since(t_start);
		if (teardown_thread.joinable()) teardown_thread.join();
		double const t_torn_down = since(t_start);
		(void) contexts; (void) index;		// not taken down one by one: see the end of this block

		// ---- finish (:1033-1073) ---------------------------------------------------------------------------------------------------------
		if (opt.realigned_ranges)
		{
			if (!opt.keep_duplicate_ranges)
			{
				std::sort(realigned_ranges.begin(), realigned_ranges.end());
				realigned_ranges.erase(std::unique(realigned_ranges.begin(), realigned_ranges.end()), realigned_ranges.end());
			}
			FILE *f = std::fopen(opt.realigned_ranges, "w");
			if (!f) die(std::string("unable to open ") + opt.realigned_ranges);
			std::fputs("Location\tLength\n", f);
			for (auto const &rr : realigned_ranges) std::fprintf(f, "%u\t%u\n", rr.first, rr.second);
			std::fclose(f);
		}
		double const secs = std::chrono::duration <double>(std::chrono::steady_clock::now() - t_start).count();
		std::cerr << "Done. Time spent: " << secs << " s.\n";
		std::cerr << "\tMatched reads:         " << st.matched_reads << '\n';
		std::cerr << "\tReference ID missing:  " << st.ref_id_missing << '\n';
		std::cerr << "\tFlags not matched:     " << st.flags_not_matched << '\n';
		std::cerr << "\tSequence missing:      " << st.seq_missing << '\n';
		std::cerr << "\tRe-aligned ranges:     " << n_realigned_ranges << '\n';
		if (opt.debugging_output)
			std::cerr << "\tStages (s): start-up " << t_startup << ", reader thread " << tm.read << ", main: waiting for it " << tm.wait_read << ", batch arrays " << tm.fill
					  << ", GPU " << tm.gpu << ", rewriting " << tm.rewrite << ", waiting for the writer " << tm.wait_write << ", writer thread " << tm_write << '\n';
		if (opt.debugging_output)
			std::cerr << "\tTimeline (s): GPU contexts " << t_contexts_up << " and BGZF compressor " << t_deflater_up << " (their thread), index file and FASTA " << t_files_read << ", both " << t_prepared << ", first batch read " << t_first_batch << ", index on the GPUs (start-up done) " << t_startup << ", last batch rewritten " << t_batches_done << ", writer thread joined " << t_writer_joined
					  << ", output closed " << t_closed << ", memory given back " << t_torn_down << "; wall clock at the start " << std::fixed << epoch_start << ", now "
					  << std::chrono::duration <double>(std::chrono::system_clock::now().time_since_epoch()).count() << std::defaultfloat
					  << "; resident set (MB): contexts up " << rss_prepared << ", index uploaded " << rss_startup << ", last batch rewritten " << rss_batches_done
					  << ", peak " << peak_rss_mb() << "; CPU time (s): " << cpu_seconds() << '\n';
		if (opt.debugging_output && opt.gpu_inflate)
			std::cerr << "\tBGZF input on the GPU: up after " << t_inflater_up << " s, " << reader.group_blocks() << " blocks in " << inflater_groups << " pieces, kernel "
					  << inflater_kernel_s << " s; inflated on the host instead " << reader.group_refused() << '\n';
		if (opt.debugging_output && block_deflater)
			std::cerr << "\tBGZF on the GPU: " << block_deflater->blocks() << " blocks in " << block_deflater->groups() << " groups, kernel " << block_deflater->kernel_seconds() << " s\n";
		if (!removed_tag_counts.empty())
		{
			std::cerr << "\tRemoved tags:\n";
			for (auto const &kv : removed_tag_counts)
				std::cerr << "\t\t" << (char) (kv.first >> 8) << (char) (kv.first & 0xff) << ": " << kv.second << '\n';
		}
		// Every output is written and closed.  The process ends here without taking the GPU contexts, the page-locked arrays,
		// the worker pool and the HIP runtime down one by one: that took as long as projecting a million records, and the
		// driver frees all of it with the process (the reference ends with exit(0) from its main queue as well,
		// project_alignments.cc:1079-1089).
		if (profile_exit)		// what the exit would cost, step by step (tools/cli_timeline.py); nothing was given back above
		{
			auto lap = [&, t = now()](char const *what) mutable { std::cerr << "\tteardown: " << what << ' ' << since(t) << " s\n"; t = now(); };
			slot_states.clear(); lap("page-locked batch arrays");
			block_deflater.reset(); lap("BGZF compressor");
			for (pvc3_context *ctx : device_contexts) pvc3_context_destroy(ctx);
			lap("GPU contexts");
			pvc3_index_destroy(index); lap("index");
			std::cerr << "\twall clock after the teardown " << std::fixed << std::chrono::duration <double>(std::chrono::system_clock::now().time_since_epoch()).count() << std::defaultfloat << '\n';
		}
		std::cout.flush(); std::cerr.flush(); std::fflush(nullptr);
		std::_Exit(0);
	}
	catch (std::exception const &e)
	{
		die(e.what());
	}
	return 0;
}
