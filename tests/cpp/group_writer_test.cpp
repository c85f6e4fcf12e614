// Test driver for byte_writer's group_deflater path (panvc3_amd/host/bgzf.hpp) without a GPU: a group_deflater on the host
// (fast_deflate.hpp behind the slot interface, three blocks per group) stands in for gpu_deflate.hpp.
// usage: group_writer_test OUT (group|plain|late) SLOTS < data      "late": attached after the first 100 000 bytes were written
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../panvc3_amd/host/bgzf.hpp"

using namespace pvc3host;

namespace {

class host_group_deflater final : public group_deflater
{
	static constexpr std::size_t STRIDE = 65536, PAD = 64, BLOCKS = 3;
	unsigned m_slots;
	std::vector <std::vector <uint8_t>> m_in, m_out;
	std::vector <std::vector <uint32_t>> m_len, m_out_len;
	std::vector <int> m_state;		// 0 free, 1 submitted
public:
	explicit host_group_deflater(unsigned slots): m_slots(slots), m_in(slots, std::vector <uint8_t>(BLOCKS * STRIDE, 0xaa)), m_out(slots, std::vector <uint8_t>(BLOCKS * STRIDE)),
		m_len(slots, std::vector <uint32_t>(BLOCKS)), m_out_len(slots, std::vector <uint32_t>(BLOCKS)), m_state(slots, 0) {}
	unsigned slots() const override { return m_slots; }
	std::size_t max_blocks() const override { return BLOCKS; }
	std::size_t in_stride() const override { return STRIDE; }
	std::size_t in_pad() const override { return PAD; }
	std::size_t out_stride() const override { return STRIDE; }
	uint8_t *input(unsigned slot) override { return m_in[slot].data(); }
	uint32_t *lengths(unsigned slot) override { return m_len[slot].data(); }
	void submit(unsigned slot, std::size_t n) override
	{
		if (m_state[slot] || 0 == n || n > BLOCKS) throw io_error("submit: slot busy or bad count");
		fast_deflater enc;
		for (std::size_t i = 0; i < n; ++i)
		{
			std::size_t const c = enc.compress(m_in[slot].data() + i * STRIDE + PAD, m_len[slot][i], m_out[slot].data() + i * STRIDE, STRIDE);
			if (0 == c) throw io_error("compress failed");
			m_out_len[slot][i] = (uint32_t) c;
		}
		m_state[slot] = 1;
	}
	void wait(unsigned slot, uint8_t const *&out, uint32_t const *&out_len) override
	{
		if (1 != m_state[slot]) throw io_error("wait: nothing submitted");
		m_state[slot] = 0;
		out = m_out[slot].data(); out_len = m_out_len[slot].data();
	}
};

}

int main(int argc, char **argv)
{
	if (argc < 4) return 2;
	std::vector <uint8_t> in;
	uint8_t buf[1 << 16];
	for (std::size_t got; (got = std::fread(buf, 1, sizeof(buf), stdin)) > 0; )
		in.insert(in.end(), buf, buf + got);
	try
	{
		host_group_deflater g((unsigned) std::atoi(argv[3]));
		byte_writer w;
		w.set_threads(3);
		w.open(argv[1], true, -1);
		bool const group = 0 == std::strcmp(argv[2], "group"), late = 0 == std::strcmp(argv[2], "late");
		if (group) w.attach(&g);
		uint32_t lcg = 12345;
		std::size_t off = 0;
		bool attached = group;
		while (off < in.size())
		{
			lcg = lcg * 1664525u + 1013904223u;
			std::size_t const n = std::min <std::size_t>(in.size() - off, 1 + (lcg >> 8) % 200000);
			w.write(in.data() + off, n);
			off += n;
			if (late && !attached && off >= 100000) { w.attach(&g); attached = true; }
		}
		w.close();
	}
	catch (std::exception const &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
	return 0;
}
