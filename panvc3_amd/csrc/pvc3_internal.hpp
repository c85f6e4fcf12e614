// pvc3_internal.hpp — layouts shared by the host side and the HIP kernels (not part of the ABI).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pvc3.h"

namespace pvc3 {

// ---- HBM layout of one gap bit-vector ---------------------------------------------------------------------------
// A sequence entry (msa_index::sequence_entry, include/panvc3/msa_index.hh:47-74) is stored as 64-byte blocks:
//   word 0      bits 0-31: number of non-gap columns ("zeros") before the block  -> rank0 in one cache line
//               bits 32-40 / 41-49 / 50-58: zeros in the block's data words 0 / 0..2 / 0..4 — the directory of the block's
//               16-byte pairs (word 0 + data 0, data 1 + 2, data 3 + 4, data 5 + 6): a rank or a select inside the block
//               needs word 0 and ONE pair (24 bytes in 6 registers instead of 64 bytes in 16)
//   words 1..7  448 gap bits (1 = gap), bit (c & 63) of word 1 + ((c % 448) >> 6)
// plus one select sample per SELECT_SAMPLE zeros (the column of zero number SELECT_SAMPLE * j).  Two all-gap padding blocks follow the
// last real block so that a two-block window load never leaves the allocation.
constexpr uint32_t BLOCK_WORDS = 8;
constexpr uint32_t BLOCK_DATA_WORDS = 7;
constexpr uint32_t BLOCK_COLS = 448;
constexpr uint32_t PAD_BLOCKS = 2;
constexpr uint32_t DIR_SHIFT = 32, DIR_BITS = 9;	// directory fields of word 0: zeros before pair p (1, 2, 3) at bit 32 + 9 (p - 1)
#ifndef PVC3_SELECT_SHIFT
#define PVC3_SELECT_SHIFT 12
#endif
constexpr uint32_t SELECT_SAMPLE_SHIFT = PVC3_SELECT_SHIFT;
constexpr uint32_t SELECT_SAMPLE = 1u << SELECT_SAMPLE_SHIFT;	// one select sample per that many non-gap columns
constexpr uint32_t TEXT_PAD = 512;		// bytes readable past the end of a device text
constexpr uint32_t TEXT_FRONT_PAD = 16;	// bytes readable before its start
constexpr uint32_t TEXT4_FRONT = 16;		// packed text: nibbles in front of base 0 (base i = nibble i + TEXT4_FRONT, low nibble first)

struct host_entry;
// dev_entry::paired for entry e and its partner o (same number of columns): blocks of 2 x 8 words.
void build_paired_blocks(host_entry const &e, host_entry const &o, std::vector <uint64_t> &out);

struct host_entry
{
	uint64_t n_cols{};
	uint64_t n_zeros{};
	uint64_t n_blocks{};				// without padding
	std::vector <uint64_t> blocks;		// (n_blocks + PAD_BLOCKS) * 8
	std::vector <uint32_t> sel;			// ceil(n_zeros / 64) + 1
	std::vector <char> text;			// unaligned text, may be empty
	uint32_t destination{0xFFFFFFFFu};	// pvc3_index_set_destination: the entry this one's records are projected to (none: the nearest entry with a text and as many columns)
};

struct dev_entry
{
	uint64_t const *blocks;
	uint32_t const *sel;
	uint8_t const *text;
	uint8_t const *text4;				// 4-bit copy for the fast path (A C G T N -> 1 2 4 8 15, anything else 0), see k_pack_text4
	uint32_t n_cols;
	uint32_t n_zeros;
	uint32_t text_len;
	uint32_t n_blocks;
	// Paired copy (null if the entry has no partner): a block of this entry and the block of entry `partner` over the same
	// columns side by side in one aligned 128-byte line, block b starting at column PAIRED_STEP * b.  HBM is fetched in
	// 128-byte lines, so a projection between the two entries reads ONE line per record for both rows instead of two
	// (tools/calib/random_access: 12.7 instead of 10.2 G records/s for the access pattern of k_project_fast).  The partner is
	// the entry with a text (a projection's destination) of the same MSA.
	uint64_t const *paired;
	uint32_t partner;
	uint32_t pad;
	// Windowed copy of text4 (null without a text): 128-byte lines, line k = the 256 nibbles of text4 from nibble
	// TEXT_WINDOW_STEP * k on (the lines overlap).  Any run of up to TEXT_WINDOW_MAX nibbles lies inside ONE line, so the
	// compare of a read of up to 150 bases fetches one 128-byte line of text instead of 1.6 on average.
	uint8_t const *text4w;
	// Position windows (null without one): what a projection from this entry to entry `pwin_partner` needs of a record that
	// starts at source position p, in ONE line addressed by p alone — no select sample, no rank block, no separate text line.
	// Window k covers the source positions from P = k << pwin_shift on and is 2 << pwin_shift bytes long (128 or 256):
	//   u32 c0      column of source position P (select0(P + 1))
	//   u32 r0      rank0 of the partner at c0 (the projected position of P)
	//   u32 clean   number of consecutive columns from c0 on that are non-gap in BOTH rows (capped at PWIN_CLEAN_CAP)
	//   u32 0
	//   bytes 16..  the partner's packed text (4-bit codes, low nibble first) from base r0 - PWIN_TEXT_BACK on
	// For p = P + o with o < clean the column is c0 + o and the projected position r0 + o; an alignment of R
	// reference-consuming bases has no gap column in either row iff o + R <= clean, and its text is then nibbles
	// o + PWIN_TEXT_BACK .. of the same line.  HBM is sized for this: two bytes per source position of every row.
	uint8_t const *pwin;
	uint32_t pwin_partner;
	uint32_t pwin_shift;
};
static_assert(88 == sizeof(dev_entry), "dev_entry layout");
constexpr uint32_t PWIN_HEADER = 16;			// bytes
constexpr uint32_t PWIN_TEXT_BACK = 2;			// nibbles of text before base r0 in a window (the compare may start one nibble early)
constexpr uint32_t PWIN_CLEAN_CAP = 1024;		// columns
// mate_dst_pos of a record whose mate is the neighbouring record (rec ^ 1) while that record's own start is still to be
// resolved (resolve_start): the walk kernel that resolves it stores the projected position here as well.
constexpr uint32_t MATE_FROM_NEIGHBOUR = 0xFFFFFFFEu;
constexpr uint32_t PWIN_NO_ANCHOR = 0xFFFFFFFFu;	// `skip` of a record whose start column has to be looked up from the select samples
// The compare reads the run in 16-byte pieces from the byte of its first nibble: piece floor(U / 32) is the last one, so a
// run of U <= 159 nibbles that starts in the first 96 nibbles of a line ends at byte 47 + 80 = 127 of it.
constexpr uint32_t TEXT_WINDOW_STEP = 96;		// nibbles
constexpr uint32_t TEXT_WINDOW_MAX = 159;		// longest run (query positions, including the phase nibble) served from the windowed copy
constexpr uint32_t NO_PARTNER = 0xFFFFFFFFu;
// Columns between the starts of consecutive blocks of a paired copy; a multiple of 64.  Less than BLOCK_COLS makes the blocks
// overlap, so that both ends of a span lie in one line; measured with 256 on config C3 (mates 250 columns apart): slower
// than 448, because two mates then never share a line (3.88 against 3.70 ms).
#ifndef PVC3_PAIRED_STEP
#define PVC3_PAIRED_STEP 448
#endif
constexpr uint32_t PAIRED_STEP = PVC3_PAIRED_STEP;

// ---- device batch ------------------------------------------------------------------------------------------------
struct dev_batch_in
{
	uint64_t n;
	uint32_t const *src_pos;
	uint32_t const *src_entry;
	uint32_t const *dst_entry;
	uint32_t const *flags;
	uint64_t const *cigar_off;
	uint32_t const *cigar;
	uint64_t const *seq_off;
	uint32_t const *l_seq;
	uint8_t const *seq4;
	uint64_t const *qual_off;
	uint8_t const *qual;
	uint32_t const *mate_src_entry;
	uint32_t const *mate_dst_entry;
	uint32_t const *mate_pos;
};

// ---- indel-run realignment ----------------------------------------------------------------------------------------
// One work item per indel run that has both insertions and deletions (k_realign_scan writes them next to the pvc3_run of the
// same index in the runs arena).  The alignment kernels leave the run's new operations in its slot of the CIGAR arena;
// k_realign_splice puts the record's CIGAR together.
struct run_work
{
	uint32_t rec;
	uint32_t ref_loc;			// run start in destination coordinates (pvc3_run::ref_position is the position after the terminator)
	uint32_t query_loc;
	uint32_t m;					// reference bases (horizontal)
	uint32_t n;					// query bases (vertical)
	uint32_t n_out;				// operations produced by the alignment
	uint32_t op_first, op_end;	// the run's operations in the rewritten CIGAR: [op_first, op_end)
	uint64_t ops_loc;			// arena offset of the first produced operation
	uint64_t splice_loc;		// first run of a record: arena space reserved for the record's final CIGAR
	uint8_t const *text;		// reference slice of the run (destination text + ref_loc)
	uint8_t const *seq4;		// SEQ of the record (4-bit codes; the run's bases start at query_loc)
	uint8_t const *qual;		// QUAL of the record or null
	uint64_t trace_loc;			// k_realign_dp: where the task of this run left its trace words (TRACE_NONE: it did not)
};
static_assert(80 == sizeof(run_work), "run_work layout");
constexpr uint64_t TRACE_NONE = ~UINT64_C(0);

// Alignment classes: the lines of the matrix along the run's shorter side go four to a lane, so a run takes (shorter + 3) / 4 lanes;
// the classes are lane groups of 3, 4, 6, 8, 12 and 16 lanes (a shorter side of up to 12, 16, 24, 32, 48, 64) x orientation (0:
// lines = query rows, 1: lines = reference columns), plus the runs that need the striped one-wave-per-run kernel (shorter side >
// 64 or longer + shorter - 1 > DP_MAX_STEPS).
constexpr uint32_t DP_LINES_PER_LANE = 4;
constexpr uint32_t DP_WIDTHS = 6;
constexpr uint32_t DP_CLASSES = 2 * DP_WIDTHS + 1;
constexpr uint32_t DP_CLASS_BIG = 2 * DP_WIDTHS;
__host__ __device__ constexpr uint32_t dp_lanes_of_class(uint32_t cls) { uint32_t const w = cls >> 1; return (0 == w) ? 3u : (1 == w) ? 4u : (2 == w) ? 6u : (3 == w) ? 8u : (4 == w) ? 12u : 16u; }
constexpr uint32_t DP_MAX_STEPS = 128;		// anti-diagonal steps of a packed alignment (long + short - 1)

// Global counters of one projection attempt (zeroed before every launch sequence).
struct dev_counters
{
	unsigned long long arena_head;		// next free uint32 in the CIGAR arena
	unsigned long long runs_head;		// next free pvc3_run in the runs arena
	unsigned int arena_overflow;
	unsigned int runs_overflow;
	unsigned int scratch_needed;		// max bytes of DP scratch a single wave asked for
	unsigned int scratch_overflow;
	unsigned long long n_failed;
	unsigned long long n_realigned_records;
	unsigned long long n_runs;
	unsigned long long bitvec_bytes;	// sum over records of 2 * ceil(columns walked / 8)
	unsigned long long ref_bytes;		// sum over records of destination characters spanned
	unsigned long long qual_bytes;		// sum of l_seq over realigned records
	unsigned long long n_fast;			// records finished by the lane-parallel fast path
	unsigned long long fast_bytes;		// algorithmic bytes of k_project_fast
	unsigned long long n_mid;			// records finished by k_project_mid
	unsigned long long general_bytes;	// algorithmic bytes of k_project_general (data the fast kernel had not read yet)
	unsigned int dp_count[DP_CLASSES];	// indel runs per alignment class (k_realign_scan appends, the k_realign_dp launches consume)
	unsigned int run_records;			// records with an indel run (sum of the per-region work lists, k_run_prefix)
	unsigned int qual_overflow;			// the compact QUAL buffer of a pipelined batch was too small
	unsigned int trace_overflow;		// the trace buffer of the packed alignment kernels was too small
	unsigned long long trace_head;		// next free word of it
	unsigned long long qual_head;		// next free byte of that buffer
};

// What k_project_mid needs of a record that k_project_fast could not finish, written by the lane that had it all in
// registers, one item behind the other in the wave's part of the list: k_project_mid then reads 48 contiguous bytes per
// record instead of nine fields from nine arrays at a scattered record number (a 128-byte line each: half of the lines the
// kernel moved on config C3, where one record in 26 comes here).  A wave whose first group of 64 leaves more than half of its
// records (a gap-dense MSA) does not write items: its list is then nearly every record in turn, the nine arrays are read
// as streams, and the items would be 48 more bytes to write for each.
struct slow_item
{
	uint32_t rec, col0, r0, src_e;
	uint32_t dst_e, n_cigar, l_seq, skip;		// skip: see resolve_start (wave.hpp); 0 = (col0, r0) are the record's own
	uint64_t cig_begin, seq_begin;
};
static_assert(48 == sizeof(slow_item), "slow_item layout");

// Per-record results before packing.  rec_flags: bits 0-7 status; REC_SLOW while a record waits for k_project_mid /
// k_project_general, which leave the plain status (the array then doubles as the status output).
struct dev_batch_work
{
	uint32_t *dst_pos;
	uint32_t *mate_dst_pos;
	uint32_t *rec_flags;
	uint64_t *cig_loc;					// arena offset of the record's CIGAR
	uint32_t *cig_n;
	uint64_t *run_loc;
	uint32_t *run_n;
	uint32_t *arena;
	uint64_t arena_cap;
	pvc3_run *runs_arena;
	uint64_t runs_cap;
	uint32_t *trace;					// trace words of the packed alignment kernels (context-wide, like the scratch)
	uint64_t trace_cap;
	uint8_t *scratch;					// DP scratch, scratch_per_wave bytes per realign wave
	uint64_t scratch_per_wave;
	uint32_t *slow_list;				// records left by the fast path: wave w of k_project_fast fills [64 * first group of w ...)
	slow_item *slow_items;				// the same records, same places, with what k_project_mid needs of them
	uint32_t *slow_count;				// [waves of k_project_fast]; k_project_fast sets SLOW_ITEMS when it wrote slow_items as well
	uint32_t *run_list;					// records with an indel run, same partition: k_project_mid appends, then k_project_general
	uint32_t *run_count;				// [waves of k_project_fast]
	uint32_t *run_prefix;				// [waves of k_project_fast + 1] exclusive prefix sums of run_count (k_run_prefix)
	run_work *run_items;				// [runs_cap], parallel to runs_arena
	uint32_t *dp_list;					// [DP_CLASSES][runs_cap] indices into run_items
	// Pipelined batches leave QUAL in the caller's page-locked host arrays; k_gather_qual fetches it over PCIe for the records
	// with an indel run only (a fraction of a percent of them) into qual_compact and notes the place in qual_off_dev[record];
	// dev_batch_in::qual / qual_off then point to these two.
	uint8_t const *host_qual;			// device-visible host pointers, null when QUAL was uploaded with the batch
	uint64_t const *host_qual_off;
	uint8_t *qual_compact;
	uint64_t qual_compact_cap;
	uint64_t *qual_off_dev;				// [records]
	unsigned long long *wave_stats;		// [3 kernels][waves][WAVE_STATS]: per-wave sums, folded into `counters` by k_reduce_stats
	dev_counters *counters;
	uint32_t const *first_bad;			// k_validate_batch: 0xFFFFFFFF, or the first record with a malformed layout (then nothing of the batch is read)
};

struct dev_batch_packed
{
	uint64_t *out_cigar_off;			// [n + 1]
	uint32_t *out_cigar;
	uint64_t out_cigar_cap;
	uint64_t *run_off;					// [n + 1]
	pvc3_run *runs;
	uint64_t run_cap;
	int32_t *status;					// [n]
};

// Per-wave statistics of the three projection kernels (plain stores; thousands of same-address atomics at the end of a
// kernel serialise into a visible tail).  Rows: 0 k_project_fast, 1 k_project_mid, 2 k_project_general.
constexpr uint32_t WAVE_STATS = 8;
constexpr uint32_t WS_BITVEC = 0, WS_REF = 1, WS_FAILED = 2, WS_DONE = 3, WS_BYTES = 4, WS_WITH_RUNS = 5;

constexpr uint32_t SLOW_ITEMS = 0x80000000u;
constexpr uint32_t REC_STATUS_MASK = 0xffu;
constexpr uint32_t REC_HAS_RUN = 0x100u;
constexpr uint32_t REC_SLOW = 0x200u;		// left by k_project_fast for k_project_general

struct dev_params
{
	int32_t gap_open;
	int32_t gap_ext;
	int32_t score_mode;
	int32_t tiebreak_mode;				// pvc3_params::tiebreak_mode: other than 0 only k_realign_big knows (every run goes there then)
	double s_match[42];					// Malde scores vs. reference quality 41, indexed by read phred
	double s_mismatch[42];
};

// Launch wrappers (kernels.hip).  All enqueue on `stream` and return a hipError_t as int.
int launch_project_fast(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, uint32_t use_fast_path, bool wide_windows, bool wary, void *stream);
int launch_project_mid(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, void *stream);
int launch_project_general(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, void *stream);
int launch_realign(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, dev_params const *params, uint32_t n_waves, uint32_t big_waves, uint32_t lds_budget, uint32_t packed, void *stream);	// consumes run_list / run_count
int launch_reduce_stats(uint64_t n, dev_batch_work const &work, void *stream);
int launch_pack(uint64_t n, dev_batch_work const &work, dev_batch_packed const &packed, uint64_t *scan_tmp, void *stream);
int launch_checksum(uint64_t n, uint64_t first_record, dev_batch_work const &work, dev_batch_packed const &packed, unsigned long long *out, void *stream);
int launch_pack_text4(uint8_t const *text, uint64_t len, uint8_t *out, void *stream);
int launch_window_text4(uint8_t const *text4, uint64_t n_lines, uint8_t *out, void *stream);	// out[128 k + j] = text4[TEXT_WINDOW_STEP / 2 * k + j]
int launch_pair_blocks(uint64_t const *a, uint64_t const *b, uint64_t n_have, uint64_t n_paired, uint64_t zeros_a, uint64_t zeros_b, uint64_t *out, void *stream);
int launch_build_pwin(dev_entry const &src, dev_entry const &dst, uint32_t shift, uint64_t n_windows, uint8_t *out, void *stream);	// dev_entry::pwin of `src` towards `dst`
int launch_project_probes(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, uint32_t use_fast_path, void *stream);	// k_project_fast's predecessor, for an index without position windows
int launch_project_positions(dev_entry const *entries, uint32_t n_entries, uint64_t n, uint32_t const *src_entry, uint32_t const *dst_entry, uint32_t const *pos, uint32_t *out, void *stream);
// Layout check of a freshly loaded batch; *first_bad = smallest malformed record (left at 0xFFFFFFFF if there is none).
int launch_validate(dev_batch_in const &in, uint64_t n_cigar, uint64_t n_seq, uint64_t n_qual, uint64_t const *qual_off, uint32_t *first_bad, void *stream);
int launch_rescore(
	uint64_t n, uint64_t const *cigar_off, uint32_t const *cigar, uint64_t const *seq_off, uint8_t const *seq4, uint32_t const *l_seq,
	uint64_t const *qual_off, uint8_t const *qual, uint64_t const *xn_off, uint8_t const *xn, pvc3_rescore_params const &prm, double *score, int32_t *status, void *stream);
int launch_mapq_groups(pvc3_mapq_batch const &in, double *pw, double *best_total, uint32_t *mate_len, uint8_t *mapq, uint8_t *emitted, unsigned long long *stats, void *stream);
uint32_t realign_lds_bytes();
uint32_t project_wave_count(uint64_t n);
uint64_t pack_scan_tmp_words(uint64_t n);

}
