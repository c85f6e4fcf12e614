// pvc3.hpp — header-only C++ host-side mirror of the reference's libpanvc3 interface over the C ABI (pvc3.h).
//
// pvc3::alignment_projector has the member functions of panvc3::alignment_projector
// (include/panvc3/alignment_projector.hh:32-75 in the reference): project_alignment(), alignment(),
// realigned_reference_ranges(), realigned_query_ranges(), reset().  Differences forced by the device boundary:
//   - sequence entries are ids in a pvc3::index that has been uploaded to the pvc3::context (the reference passes
//     msa_index::sequence_entry const &), and the destination text (ref_seq) is attached to the destination entry with
//     index::set_reference() instead of being passed with every call;
//   - the query is given as BAM 4-bit codes or as characters; CIGARs are BAM-encoded uint32 (count << 4 | op);
//   - errors are exceptions of type pvc3::error carrying the C ABI's code and message.
// A record pipeline should prefer project_batch(): one call per batch of records instead of one per record.
#ifndef PVC3_HPP
#define PVC3_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "pvc3.h"

namespace pvc3 {

struct error : public std::runtime_error
{
	int code;
	error(int code_, char const *msg): std::runtime_error(msg), code(code_) {}
};

inline void check(int rc) { if (rc) throw error(rc, pvc3_last_error()); }

struct range		// panvc3::range (include/panvc3/range.hh:18-37)
{
	std::size_t location{};
	std::size_t length{};
	bool operator==(range const &o) const { return location == o.location && length == o.length; }
};

typedef std::vector <range> range_vector;
typedef std::vector <std::uint32_t> cigar_vector;

class index
{
	pvc3_index *m_handle{};

public:
	index() { check(pvc3_index_create(&m_handle)); }
	~index() { pvc3_index_destroy(m_handle); }
	index(index const &) = delete;
	index &operator=(index const &) = delete;

	// msa_index::sequence_entry(seq_id, gap_positions): column i is a gap iff aligned[i] == '-'.
	std::uint32_t add_aligned(std::string_view aligned)
	{
		std::uint32_t e{};
		check(pvc3_index_add_aligned(m_handle, aligned.data(), aligned.size(), &e));
		return e;
	}

	std::uint32_t add_bits(std::uint64_t const *gap_words, std::uint64_t n_cols)
	{
		std::uint32_t e{};
		check(pvc3_index_add_bits(m_handle, gap_words, n_cols, &e));
		return e;
	}

	void set_reference(std::uint32_t entry, std::string_view text) { check(pvc3_index_set_reference(m_handle, entry, text.data(), text.size())); }
	pvc3_index const *handle() const { return m_handle; }
};

// panvc3::make_sequence_entry_pair (include/panvc3/msa_index.hh:214-236)
inline std::pair <std::uint32_t, std::uint32_t> make_sequence_entry_pair(index &idx, std::string_view lhs, std::string_view rhs)
{
	if (lhs.size() != rhs.size()) throw std::invalid_argument("aligned sequences differ in length");
	auto const a(idx.add_aligned(lhs));
	auto const b(idx.add_aligned(rhs));
	return {a, b};
}

class context
{
	pvc3_context *m_handle{};

public:
	explicit context(int device = 0) { check(pvc3_context_create(device, &m_handle)); }
	~context() { pvc3_context_destroy(m_handle); }
	context(context const &) = delete;
	context &operator=(context const &) = delete;

	void upload_index(index const &idx) { check(pvc3_context_upload_index(m_handle, idx.handle())); }
	pvc3_context *handle() const { return m_handle; }

	// msa_index::sequence_entry::project_position for many positions at once (msa_index.hh:253-258).
	std::vector <std::uint32_t> project_positions(std::vector <std::uint32_t> const &src, std::vector <std::uint32_t> const &dst, std::vector <std::uint32_t> const &pos)
	{
		std::vector <std::uint32_t> out(pos.size());
		check(pvc3_project_positions(m_handle, pos.size(), src.data(), dst.data(), pos.data(), out.data()));
		return out;
	}
};

// Owning storage for one output batch.
struct batch_result
{
	std::vector <std::uint32_t> dst_pos, mate_dst_pos, out_cigar;
	std::vector <std::int32_t> status;
	std::vector <std::uint64_t> out_cigar_off, run_off;
	std::vector <pvc3_run> runs;
	pvc3_batch_stats stats{};
};

inline batch_result project_batch(context &ctx, pvc3_batch_in const &in, pvc3_params const &params)
{
	batch_result res;
	pvc3_dbatch *b{};
	check(pvc3_dbatch_create(ctx.handle(), &in, &b));
	struct guard { pvc3_dbatch *b; ~guard() { pvc3_dbatch_destroy(b); } } g{b};
	check(pvc3_dbatch_project(ctx.handle(), b, &params));
	check(pvc3_dbatch_finish(ctx.handle(), b, &res.stats));
	res.dst_pos.resize(in.n); res.mate_dst_pos.resize(in.n); res.status.resize(in.n);
	res.out_cigar_off.resize(in.n + 1); res.run_off.resize(in.n + 1);
	res.out_cigar.resize(res.stats.out_cigar_total + 1); res.runs.resize(res.stats.n_runs + 1);
	pvc3_batch_out out{};
	out.dst_pos = res.dst_pos.data(); out.mate_dst_pos = res.mate_dst_pos.data(); out.status = res.status.data();
	out.out_cigar_off = res.out_cigar_off.data(); out.out_cigar = res.out_cigar.data(); out.out_cigar_cap = res.out_cigar.size();
	out.run_off = res.run_off.data(); out.runs = res.runs.data(); out.run_cap = res.runs.size();
	check(pvc3_dbatch_download(ctx.handle(), b, &out));
	res.out_cigar.resize(res.stats.out_cigar_total); res.runs.resize(res.stats.n_runs);
	return res;
}

class alignment_projector
{
	context *m_ctx{};
	cigar_vector m_cigar_realigned;
	range_vector m_realigned_reference_ranges;
	range_vector m_realigned_query_ranges;
	int m_score_mode{PVC3_SCORE_FP64};

	static std::uint8_t seq4_code(char c)
	{
		switch (c)
		{
			case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'G': case 'g': return 4; case 'T': case 't': return 8;
			default: return 15;
		}
	}

public:
	explicit alignment_projector(context &ctx): m_ctx(&ctx) {}
	void set_score_mode(int mode) { m_score_mode = mode; }

	void reset()
	{
		m_cigar_realigned.clear();
		m_realigned_reference_ranges.clear();
		m_realigned_query_ranges.clear();
	}

	// alignment_projector::project_alignment (alignment_projector.hh:58-68); returns the projected position.
	std::size_t project_alignment(
		std::size_t const src_pos,
		std::uint32_t const src_seq_entry,
		std::uint32_t const dst_seq_entry,
		std::string_view const query_seq,
		cigar_vector const &cigar_seq,
		std::vector <std::uint8_t> const &base_qualities,	// empty => QUAL-less scoring
		std::int32_t const gap_opening_cost,
		std::int32_t const gap_extension_cost
	)
	{
		std::vector <std::uint8_t> seq4((query_seq.size() + 1) / 2, 0);
		for (std::size_t i(0); i < query_seq.size(); ++i)
			seq4[i >> 1] |= std::uint8_t(seq4_code(query_seq[i]) << ((i & 1) ? 0 : 4));
		std::uint32_t const sp(src_pos), se(src_seq_entry), de(dst_seq_entry), ls(query_seq.size());
		std::uint32_t const fl(base_qualities.empty() ? 0 : PVC3_FLAG_HAS_QUAL);
		std::uint64_t const coff[2]{0, cigar_seq.size()}, soff[2]{0, seq4.size()}, qoff[2]{0, base_qualities.size()};
		pvc3_batch_in in{};
		in.n = 1; in.src_pos = &sp; in.src_entry = &se; in.dst_entry = &de; in.flags = &fl;
		in.cigar_off = coff; in.cigar = cigar_seq.data(); in.seq_off = soff; in.l_seq = &ls; in.seq4 = seq4.data();
		if (!base_qualities.empty()) { in.qual_off = qoff; in.qual = base_qualities.data(); }
		pvc3_params const params{gap_opening_cost, gap_extension_cost, m_score_mode, 0};
		auto res(project_batch(*m_ctx, in, params));
		if (res.status[0])
			throw error(PVC3_E_INVALID, "record rejected");
		reset();
		m_cigar_realigned = std::move(res.out_cigar);
		for (auto const &r : res.runs)
		{
			m_realigned_reference_ranges.push_back({r.ref_position, r.ref_length});
			m_realigned_query_ranges.push_back({r.query_location, r.query_length});
		}
		return res.dst_pos[0];
	}

	cigar_vector const &alignment() const { return m_cigar_realigned; }
	range_vector const &realigned_reference_ranges() const { return m_realigned_reference_ranges; }
	range_vector const &realigned_query_ranges() const { return m_realigned_query_ranges; }
};

}

#endif
