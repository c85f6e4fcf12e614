// Test driver for byte_reader's group_inflater path (panvc3_amd/host/bgzf.hpp) without a GPU: a group_inflater on the host stands
// in for gpu_deflate.hpp.  It inflates with zlib, and on request refuses every third block or returns wrong bytes for every
// fifth (the reader must inflate those itself / find them by their checksum).
// usage: group_reader_test FILE (plain|group|late|refuse|garble) > bytes       PVC3_BGZF_CHUNK makes the pieces small
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../panvc3_amd/host/bgzf.hpp"

using namespace pvc3host;

namespace {

class host_group_inflater final : public group_inflater
{
	std::size_t m_max_in, m_max_out, m_max_blocks;
	std::vector <std::vector <uint8_t>> m_in, m_out;
	std::vector <std::vector <block>> m_blocks;
	std::vector <std::vector <uint32_t>> m_status;
	std::vector <int> m_state;
	int m_mode;		// 0 as asked, 1 refuse every third block, 2 garble every fifth
	uint64_t m_seen{};
public:
	host_group_inflater(std::size_t max_in, std::size_t max_out, std::size_t max_blocks, int mode): m_max_in(max_in), m_max_out(max_out), m_max_blocks(max_blocks),
		m_in(2, std::vector <uint8_t>(max_in + 64)), m_out(2, std::vector <uint8_t>(max_out)), m_blocks(2, std::vector <block>(max_blocks)), m_status(2, std::vector <uint32_t>(max_blocks)),
		m_state(2, 0), m_mode(mode) {}
	unsigned slots() const override { return 2; }
	std::size_t max_in() const override { return m_max_in; }
	std::size_t max_out() const override { return m_max_out; }
	std::size_t max_blocks() const override { return m_max_blocks; }
	std::size_t blocks_in_flight() const override { return 5; }		// (pieces of more blocks are shared with the reader's own threads)
	uint8_t *input(unsigned slot) override { return m_in[slot].data(); }
	uint8_t *output(unsigned slot) override { return m_out[slot].data(); }
	block *blocks(unsigned slot) override { return m_blocks[slot].data(); }
	void submit(unsigned slot, std::size_t n, std::size_t in_bytes, std::size_t out_bytes) override
	{
		if (m_state[slot] || 0 == n || n > m_max_blocks || in_bytes > m_max_in || out_bytes > m_max_out) throw io_error("submit: bad call");
		for (std::size_t i = 0; i < n; ++i, ++m_seen)
		{
			block const &b = m_blocks[slot][i];
			if (b.in_off + b.in_len > in_bytes || b.out_off + b.out_len > out_bytes) throw io_error("submit: block outside the slot");
			m_status[slot][i] = 0;
			if (1 == m_mode && 0 == m_seen % 3) { m_status[slot][i] = 4; continue; }
			z_stream zs{};
			inflateInit2(&zs, -15);
			zs.next_in = m_in[slot].data() + b.in_off; zs.avail_in = b.in_len;
			zs.next_out = m_out[slot].data() + b.out_off; zs.avail_out = b.out_len;
			int const rc = inflate(&zs, Z_FINISH);
			inflateEnd(&zs);
			if (Z_STREAM_END != rc || zs.total_out != b.out_len) m_status[slot][i] = 8;
			if (2 == m_mode && 0 == m_seen % 5 && b.out_len) m_out[slot][b.out_off + b.out_len / 2] ^= 0x20;
		}
		m_state[slot] = 1;
	}
	uint32_t const *wait(unsigned slot) override
	{
		if (1 != m_state[slot]) throw io_error("wait: nothing submitted");
		m_state[slot] = 0;
		return m_status[slot].data();
	}
};

}

int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	try
	{
		char const *mode = argv[2];
		host_group_inflater g(1u << 20, 8u << 20, 256, 0 == std::strcmp(mode, "refuse") ? 1 : 0 == std::strcmp(mode, "garble") ? 2 : 0);
		byte_reader r;
		r.set_threads(3);
		r.open(argv[1]);
		bool const plain = 0 == std::strcmp(mode, "plain"), late = 0 == std::strcmp(mode, "late");
		if (!plain && !late) r.attach(&g);
		std::vector <uint8_t> buf(70001);
		std::size_t total = 0;
		for (std::size_t got; (got = r.read(buf.data(), 1 + total % buf.size())) > 0; )
		{
			std::fwrite(buf.data(), 1, got, stdout);
			total += got;
			if (late && total > 300000) r.attach(&g);
		}
		std::fprintf(stderr, "blocks through the group_inflater %llu, inflated here instead %llu\n", (unsigned long long) r.group_blocks(), (unsigned long long) r.group_refused());
	}
	catch (std::exception const &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
	return 0;
}
