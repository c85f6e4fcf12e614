// bgzf.hpp — BGZF (blocked gzip, SAMv1 §4.1) reader and writer over zlib, plus a plain / gzip text line reader.
// Host-side I/O of the MI355X project_alignments pipeline (the reference uses seqan3 + Boost.Iostreams here).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pvc3host {

struct io_error : public std::runtime_error
{
	using std::runtime_error::runtime_error;
};

// Sequential byte source: a file or stdin, transparently inflating gzip / BGZF (zlib handles multi-member gzip).
class byte_reader
{
	gzFile m_file{};

public:
	byte_reader() = default;
	byte_reader(byte_reader const &) = delete;
	~byte_reader() { close(); }

	void open(char const *path)		// nullptr or "-" = stdin
	{
		close();
		if (!path || 0 == std::strcmp(path, "-"))
			m_file = gzdopen(0, "rb");
		else
			m_file = gzopen(path, "rb");
		if (!m_file)
			throw io_error(std::string("unable to open ") + (path ? path : "stdin"));
		gzbuffer(m_file, 1 << 20);
	}

	void close() { if (m_file) { gzclose(m_file); m_file = nullptr; } }

	// Reads exactly n bytes unless the stream ends; returns the number read.
	std::size_t read(void *dst, std::size_t n)
	{
		std::size_t got = 0;
		while (got < n)
		{
			unsigned const chunk = (unsigned) std::min <std::size_t>(n - got, 1u << 30);
			int const r = gzread(m_file, (char *) dst + got, chunk);
			if (r < 0) throw io_error("read error (corrupt gzip / BGZF stream?)");
			if (0 == r) break;
			got += (std::size_t) r;
		}
		return got;
	}

	// One text line without its terminator; false at the end of the stream.
	bool getline(std::string &line)
	{
		line.clear();
		char buf[1 << 16];
		for (;;)
		{
			if (!gzgets(m_file, buf, sizeof(buf)))
				return !line.empty();
			std::size_t const len = std::strlen(buf);
			line.append(buf, len);
			if (len && '\n' == buf[len - 1])
			{
				line.pop_back();
				if (!line.empty() && '\r' == line.back()) line.pop_back();
				return true;
			}
		}
	}

	int peek()
	{
		int const c = gzgetc(m_file);
		if (c >= 0) gzungetc(c, m_file);
		return c;
	}
};

// Sequential byte sink: plain (SAM text) or BGZF blocks (BAM).
class byte_writer
{
	FILE *m_file{};
	bool m_owns{};
	bool m_bgzf{};
	int m_level{6};
	std::vector <uint8_t> m_block;		// pending uncompressed bytes of the current BGZF block
	std::vector <uint8_t> m_out;

	static constexpr std::size_t BLOCK_DATA = 0xff00;	// uncompressed bytes per block (as htslib)

	void write_raw(void const *p, std::size_t n)
	{
		if (n && 1 != std::fwrite(p, n, 1, m_file))
			throw io_error("write error");
	}

	void flush_block()
	{
		// BGZF block = gzip member with the BC extra field (SAMv1 §4.1.1).
		m_out.resize(18 + compressBound((uLong) m_block.size()) + 8);
		z_stream zs{};
		if (Z_OK != deflateInit2(&zs, m_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY))
			throw io_error("deflateInit2 failed");
		zs.next_in = m_block.data(); zs.avail_in = (uInt) m_block.size();
		zs.next_out = m_out.data() + 18; zs.avail_out = (uInt) (m_out.size() - 18 - 8);
		if (Z_STREAM_END != deflate(&zs, Z_FINISH)) { deflateEnd(&zs); throw io_error("deflate failed"); }
		std::size_t const clen = zs.total_out;
		deflateEnd(&zs);
		std::size_t const bsize = 18 + clen + 8;
		static uint8_t const head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
		std::memcpy(m_out.data(), head, 16);
		m_out[16] = (uint8_t) ((bsize - 1) & 0xff); m_out[17] = (uint8_t) ((bsize - 1) >> 8);
		uint32_t const crc = (uint32_t) crc32(crc32(0L, Z_NULL, 0), m_block.data(), (uInt) m_block.size());
		uint32_t const isize = (uint32_t) m_block.size();
		std::memcpy(m_out.data() + 18 + clen, &crc, 4);
		std::memcpy(m_out.data() + 18 + clen + 4, &isize, 4);
		write_raw(m_out.data(), bsize);
		m_block.clear();
	}

public:
	byte_writer() = default;
	byte_writer(byte_writer const &) = delete;
	~byte_writer() { try { close(); } catch (...) {} }

	void open(char const *path, bool bgzf, int level = 6)		// nullptr or "-" = stdout
	{
		close();
		m_bgzf = bgzf; m_level = level;
		if (!path || 0 == std::strcmp(path, "-")) { m_file = stdout; m_owns = false; }
		else
		{
			m_file = std::fopen(path, "wb");
			if (!m_file) throw io_error(std::string("unable to open ") + path + " for writing");
			m_owns = true;
		}
		static_cast <void>(std::setvbuf(m_file, nullptr, _IOFBF, 1 << 20));
		m_block.reserve(BLOCK_DATA);
	}

	void write(void const *p, std::size_t n)
	{
		if (!m_bgzf) { write_raw(p, n); return; }
		uint8_t const *src = (uint8_t const *) p;
		while (n)
		{
			std::size_t const take = std::min(n, BLOCK_DATA - m_block.size());
			m_block.insert(m_block.end(), src, src + take);
			src += take; n -= take;
			if (m_block.size() == BLOCK_DATA) flush_block();
		}
	}

	void write(std::string const &s) { write(s.data(), s.size()); }

	void close()
	{
		if (!m_file) return;
		if (m_bgzf)
		{
			if (!m_block.empty()) flush_block();
			// The empty EOF marker block.
			static uint8_t const eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
			write_raw(eof, sizeof(eof));
		}
		std::fflush(m_file);
		if (m_owns) std::fclose(m_file);
		m_file = nullptr;
	}
};

}
