// gpu_deflate.hpp — the BGZF writer's blocks compressed and the BGZF reader's blocks inflated on a GPU: group_deflater and
// group_inflater (bgzf.hpp) over the C ABI of libpvc3_bgzf.so (include/pvc3_bgzf.h; the kernels are panvc3_amd/bgzf/deflate.hip
// and inflate.hip).
// No reference code corresponds to this file: the reference writes BAM through seqan3 / zlib.
#pragma once

#include <string>

#include "bgzf.hpp"
#include "../../include/pvc3_bgzf.h"

namespace pvc3host {

class gpu_group_deflater final : public group_deflater
{
	pvc3_bgzf *m_handle{};
	unsigned m_slots{};
	std::size_t m_max_blocks{};
	double m_kernel_ms{};
	uint64_t m_groups{}, m_blocks{};

	[[noreturn]] static void fail(char const *call) { throw io_error(std::string(call) + ": " + pvc3_bgzf_last_error()); }

public:
	// Page-locked staging for `slots` groups of `max_blocks` blocks each (64 KiB in and out per block) and the device buffers.
	gpu_group_deflater(int device, std::size_t max_blocks, unsigned slots): m_slots(slots), m_max_blocks(max_blocks)
	{
		if (pvc3_bgzf_create(device, (uint32_t) max_blocks, slots, &m_handle)) fail("pvc3_bgzf_create");
	}
	gpu_group_deflater(gpu_group_deflater const &) = delete;
	~gpu_group_deflater() override { pvc3_bgzf_destroy(m_handle); }

	unsigned slots() const override { return m_slots; }
	std::size_t max_blocks() const override { return m_max_blocks; }
	std::size_t in_stride() const override { return PVC3_BGZF_IN_STRIDE; }
	std::size_t in_pad() const override { return PVC3_BGZF_IN_PAD; }
	std::size_t out_stride() const override { return PVC3_BGZF_OUT_STRIDE; }
	uint8_t *input(unsigned slot) override { return pvc3_bgzf_slot_input(m_handle, slot); }
	uint32_t *lengths(unsigned slot) override { return pvc3_bgzf_slot_lengths(m_handle, slot); }

	void submit(unsigned slot, std::size_t n_blocks) override
	{
		if (pvc3_bgzf_submit(m_handle, slot, (uint32_t) n_blocks)) fail("pvc3_bgzf_submit");
		++m_groups; m_blocks += n_blocks;
	}

	void wait(unsigned slot, uint8_t const *&out, uint32_t const *&out_len) override
	{
		float ms = 0;
		if (pvc3_bgzf_wait(m_handle, slot, &out, &out_len, &ms)) fail("pvc3_bgzf_wait");
		m_kernel_ms += ms;
	}

	double kernel_seconds() const { return 1e-3 * m_kernel_ms; }
	uint64_t groups() const { return m_groups; }
	uint64_t blocks() const { return m_blocks; }
};


class gpu_group_inflater final : public group_inflater
{
	pvc3_bgzf_inflater *m_handle{};
	unsigned m_slots{};
	std::size_t m_max_in{}, m_max_out{}, m_max_blocks{};
	double m_kernel_ms{};
	uint64_t m_groups{};
	static_assert(sizeof(group_inflater::block) == sizeof(pvc3_bgzf_block), "the same layout");

	[[noreturn]] static void fail(char const *call) { throw io_error(std::string(call) + ": " + pvc3_bgzf_last_error()); }

public:
	gpu_group_inflater(int device, std::size_t max_in, std::size_t max_out, std::size_t max_blocks, unsigned slots):
		m_slots(slots), m_max_in(max_in), m_max_out(max_out), m_max_blocks(max_blocks)
	{
		if (pvc3_bgzf_inflater_create(device, max_in, max_out, (uint32_t) max_blocks, slots, &m_handle)) fail("pvc3_bgzf_inflater_create");
	}
	gpu_group_inflater(gpu_group_inflater const &) = delete;
	~gpu_group_inflater() override { pvc3_bgzf_inflater_destroy(m_handle); }

	unsigned slots() const override { return m_slots; }
	std::size_t max_in() const override { return m_max_in; }
	std::size_t max_out() const override { return m_max_out; }
	std::size_t max_blocks() const override { return m_max_blocks; }
	std::size_t blocks_in_flight() const override { return pvc3_bgzf_inflater_blocks_in_flight(m_handle); }
	uint8_t *input(unsigned slot) override { return pvc3_bgzf_inflater_input(m_handle, slot); }
	uint8_t *output(unsigned slot) override { return pvc3_bgzf_inflater_output(m_handle, slot); }
	block *blocks(unsigned slot) override { return reinterpret_cast <block *>(pvc3_bgzf_inflater_blocks(m_handle, slot)); }

	void submit(unsigned slot, std::size_t n_blocks, std::size_t in_bytes, std::size_t out_bytes) override
	{
		if (pvc3_bgzf_inflater_submit(m_handle, slot, (uint32_t) n_blocks, in_bytes, out_bytes)) fail("pvc3_bgzf_inflater_submit");
		++m_groups;
	}

	uint32_t const *wait(unsigned slot) override
	{
		uint32_t const *status{};
		float ms = 0;
		if (pvc3_bgzf_inflater_wait(m_handle, slot, &status, &ms)) fail("pvc3_bgzf_inflater_wait");
		m_kernel_ms += ms;
		return status;
	}

	double kernel_seconds() const { return 1e-3 * m_kernel_ms; }
	uint64_t groups() const { return m_groups; }
};

}
