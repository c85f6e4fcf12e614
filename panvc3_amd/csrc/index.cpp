// index.cpp — host side of the MSA gap bit-vector index: builds the HBM block layout.
// Replaces msa_index::sequence_entry construction (include/panvc3/msa_index.hh:58-65: rrr_vector<15> + rank_0 /
// select_0 supports) and index-msa/index_vector_builder.cc:29-52 (bit i set iff byte i == '-').
#include "pvc3_internal.hpp"
#include "host_state.hpp"

#include <cstring>

namespace pvc3 {

static void finish_entry(host_entry &e)
{
	// Padding columns of the last block and the padding blocks are gaps (not zeros).
	uint64_t const total_blocks = e.n_blocks + PAD_BLOCKS;
	for (uint64_t c = e.n_cols; c < e.n_blocks * BLOCK_COLS; ++c)
	{
		uint64_t const b = c / BLOCK_COLS, r = c % BLOCK_COLS;
		e.blocks[b * BLOCK_WORDS + 1 + (r >> 6)] |= UINT64_C(1) << (r & 63);
	}
	for (uint64_t b = e.n_blocks; b < total_blocks; ++b)
		for (uint32_t w = 1; w < BLOCK_WORDS; ++w)
			e.blocks[b * BLOCK_WORDS + w] = ~UINT64_C(0);

	// Rank words (with the in-block directory, see pvc3_internal.hpp) + select samples in one pass.
	uint64_t zeros = 0;
	e.sel.clear();
	for (uint64_t b = 0; b < total_blocks; ++b)
	{
		uint64_t const zeros_before_block = zeros;
		uint64_t dir = 0;
		for (uint32_t w = 0; w < BLOCK_DATA_WORDS; ++w)
		{
			if (1 == (w & 1)) dir |= (zeros - zeros_before_block) << (DIR_SHIFT + DIR_BITS * (w >> 1));		// before pairs 1, 2, 3
			uint64_t z = ~e.blocks[b * BLOCK_WORDS + 1 + w];
			uint32_t const cnt = (uint32_t) __builtin_popcountll(z);
			// Samples falling into this word: zero numbers k with k % 64 == 0 and zeros <= k < zeros + cnt.
			uint64_t next_sample = (zeros + SELECT_SAMPLE - 1) / SELECT_SAMPLE * SELECT_SAMPLE;
			while (next_sample < zeros + cnt)
			{
				uint64_t zz = z;
				for (uint64_t skip = next_sample - zeros; skip; --skip)
					zz &= zz - 1;
				e.sel.push_back((uint32_t) (b * BLOCK_COLS + w * 64 + (uint64_t) __builtin_ctzll(zz)));
				next_sample += SELECT_SAMPLE;
			}
			zeros += cnt;
		}
		e.blocks[b * BLOCK_WORDS] = zeros_before_block | dir;
	}
	e.n_zeros = zeros;
	e.sel.push_back((uint32_t) e.n_cols);	// sentinel so that sel[n_zeros / SELECT_SAMPLE] is always readable
}

void build_paired_blocks(host_entry const &e, host_entry const &o, std::vector <uint64_t> &out)
{
	// Data word j of block b of the paired copy is word 4 b + j of the row's flat bit-vector (PAIRED_STEP = 4 x 64 columns).
	static_assert(0 == PAIRED_STEP % 64, "paired blocks start at word boundaries");
	uint64_t const words_per_step = PAIRED_STEP / 64;
	uint64_t const n_flat = e.n_blocks * BLOCK_DATA_WORDS;
	uint64_t const n_paired = e.n_cols / PAIRED_STEP + 1 + PAD_BLOCKS;
	out.assign(n_paired * 2 * BLOCK_WORDS, 0);
	host_entry const *rows[2] = {&e, &o};
	for (uint32_t r = 0; r < 2; ++r)
	{
		host_entry const &h = *rows[r];
		auto flat = [&](uint64_t i) -> uint64_t { return (i < n_flat) ? h.blocks[(i / BLOCK_DATA_WORDS) * BLOCK_WORDS + 1 + i % BLOCK_DATA_WORDS] : ~UINT64_C(0); };
		uint64_t zeros = 0, next_word = 0;		// zeros before flat word next_word
		for (uint64_t b = 0; b < n_paired; ++b)
		{
			for (; next_word < words_per_step * b; ++next_word)
				zeros += (uint64_t) __builtin_popcountll(~flat(next_word));
			uint64_t *dst = &out[(2 * b + r) * BLOCK_WORDS];
			uint64_t dir = 0, in_block = 0;
			for (uint32_t w = 0; w < BLOCK_DATA_WORDS; ++w)
			{
				if (1 == (w & 1)) dir |= in_block << (DIR_SHIFT + DIR_BITS * (w >> 1));
				uint64_t const x = flat(words_per_step * b + w);
				dst[1 + w] = x;
				in_block += (uint64_t) __builtin_popcountll(~x);
			}
			dst[0] = zeros | dir;
		}
	}
}

static int new_entry(pvc3_index *idx, uint64_t n_cols, host_entry **out)
{
	if (n_cols + 3 * (uint64_t) BLOCK_COLS >= (UINT64_C(1) << 32))
		return set_error(PVC3_E_INVALID, "pvc3_index: %llu columns exceed the 32-bit column limit", (unsigned long long) n_cols);
	idx->entries.emplace_back();
	host_entry &e = idx->entries.back();
	e.n_cols = n_cols;
	e.n_blocks = (n_cols + BLOCK_COLS - 1) / BLOCK_COLS;
	e.blocks.assign((e.n_blocks + PAD_BLOCKS) * BLOCK_WORDS, 0);
	*out = &e;
	return PVC3_OK;
}

int index_add_aligned(pvc3_index *idx, char const *aligned, uint64_t n_cols, uint32_t *entry)
{
	host_entry *e{};
	if (int const rc = new_entry(idx, n_cols, &e)) return rc;
	for (uint64_t b = 0; b < e->n_blocks; ++b)
	{
		uint64_t const base = b * BLOCK_COLS;
		uint64_t const lim = (base + BLOCK_COLS <= n_cols) ? BLOCK_COLS : (n_cols - base);
		uint64_t *words = &e->blocks[b * BLOCK_WORDS + 1];
		for (uint64_t r = 0; r < lim; ++r)
			if ('-' == aligned[base + r])
				words[r >> 6] |= UINT64_C(1) << (r & 63);
	}
	finish_entry(*e);
	*entry = (uint32_t) (idx->entries.size() - 1);
	return PVC3_OK;
}

int index_add_bits(pvc3_index *idx, uint64_t const *gap_words, uint64_t n_cols, uint32_t *entry)
{
	host_entry *e{};
	if (int const rc = new_entry(idx, n_cols, &e)) return rc;
	uint64_t const n_words = (n_cols + 63) / 64;
	for (uint64_t b = 0; b < e->n_blocks; ++b)
	{
		uint64_t const first = b * BLOCK_DATA_WORDS;
		uint64_t const cnt = (first + BLOCK_DATA_WORDS <= n_words) ? BLOCK_DATA_WORDS : (n_words - first);
		std::memcpy(&e->blocks[b * BLOCK_WORDS + 1], gap_words + first, cnt * sizeof(uint64_t));
	}
	// Clear whatever the caller had past n_cols in the last word; finish_entry() then marks it as gaps.
	if (n_cols & 63)
	{
		uint64_t const c = n_cols - 1;
		uint64_t const b = c / BLOCK_COLS, r = c % BLOCK_COLS;
		e->blocks[b * BLOCK_WORDS + 1 + (r >> 6)] &= (UINT64_C(1) << (n_cols & 63)) - 1;
	}
	finish_entry(*e);
	*entry = (uint32_t) (idx->entries.size() - 1);
	return PVC3_OK;
}

int index_set_reference(pvc3_index *idx, uint32_t entry, char const *text, uint64_t len)
{
	if (entry >= idx->entries.size())
		return set_error(PVC3_E_INVALID, "pvc3_index_set_reference: no entry %u", entry);
	host_entry &e = idx->entries[entry];
	if (len < e.n_zeros)
		return set_error(PVC3_E_INVALID, "pvc3_index_set_reference: text of entry %u has %llu characters but the entry has %llu non-gap columns",
			entry, (unsigned long long) len, (unsigned long long) e.n_zeros);
	if (len + 1024 >= (UINT64_C(1) << 32))		// nibble numbers of the packed text (TEXT4_FRONT + position) stay 32-bit
		return set_error(PVC3_E_INVALID, "pvc3_index_set_reference: text too long");
	e.text.assign(text, text + len);
	return PVC3_OK;
}

}
