// bgzf.hpp — BGZF (blocked gzip, SAMv1 §4.1) reader and writer over zlib, plus a plain / gzip text line reader.
// Host-side I/O of the MI355X project_alignments pipeline (the reference uses seqan3 + Boost.Iostreams here).
#pragma once

#include <unistd.h>
#include <zlib.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fast_crc32.hpp"
#include "fast_inflate.hpp"
#include "fast_deflate.hpp"

namespace pvc3host {

struct io_error : public std::runtime_error
{
	using std::runtime_error::runtime_error;
};

// A pool of worker threads shared by everything in the process that calls parallel_for (the BGZF reader, the writer, the
// stages of panvc3_project_alignments, each from a thread of its own).  Starting threads per call — several calls per batch,
// dozens of threads each — was a tenth of the tool's CPU time.
class worker_pool
{
	struct job
	{
		std::function <void(std::size_t)> const *fn{};
		std::size_t n{};
		std::atomic <std::size_t> next{0};
		std::atomic <unsigned> running{0};		// tickets not finished yet
		std::mutex m;
		std::condition_variable done;
		std::exception_ptr error;

		void work()
		{
			try { for (std::size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n; ) (*fn)(i); }
			catch (...) { std::lock_guard <std::mutex> l(m); if (!error) error = std::current_exception(); next.store(n); }
		}
	};
	std::mutex m_mutex;
	std::condition_variable m_cv;
	std::deque <job *> m_tickets;
	std::vector <std::thread> m_threads;
	bool m_stop{};

	void loop()
	{
		for (;;)
		{
			job *j{};
			{
				std::unique_lock <std::mutex> l(m_mutex);
				m_cv.wait(l, [&]() { return m_stop || !m_tickets.empty(); });
				if (m_tickets.empty()) return;
				j = m_tickets.front(); m_tickets.pop_front();
			}
			j->work();
			// (Notified under the job's lock: the caller, who owns the job, cannot wake up and destroy it before this thread is
			// done with it.)
			{ std::lock_guard <std::mutex> l(j->m); if (0 == --j->running) j->done.notify_all(); }
		}
	}

public:
	~worker_pool()
	{
		{ std::lock_guard <std::mutex> l(m_mutex); m_stop = true; }
		m_cv.notify_all();
		for (auto &t : m_threads) t.join();
	}

	static worker_pool &instance() { static worker_pool p; return p; }

	// fn(i) for i in [0, n) on the calling thread and up to helpers pool threads.
	void run(std::size_t n, unsigned helpers, std::function <void(std::size_t)> const &fn)
	{
		job j;
		j.fn = &fn; j.n = n;
		{
			std::lock_guard <std::mutex> l(m_mutex);
			while (m_threads.size() < helpers && m_threads.size() < 256) m_threads.emplace_back([this]() { loop(); });
			j.running = helpers;
			for (unsigned t = 0; t < helpers; ++t) m_tickets.push_back(&j);
		}
		m_cv.notify_all();
		j.work();
		{
			// Tickets nobody has taken yet are taken back (the work is done); the others are waited for.
			std::lock_guard <std::mutex> l(m_mutex);
			for (auto it = m_tickets.begin(); it != m_tickets.end(); )
				if (*it == &j) { it = m_tickets.erase(it); std::lock_guard <std::mutex> lj(j.m); --j.running; }
				else ++it;
		}
		{
			std::unique_lock <std::mutex> l(j.m);
			j.done.wait(l, [&]() { return 0 == j.running; });
		}
		if (j.error) std::rethrow_exception(j.error);
	}
};

// Runs fn(i) for i in [0, n) on up to n_threads threads: the caller and n_threads - 1 threads of the shared pool (the pieces
// are tens of kilobytes of zlib work, or a slice of a batch, each).
template <typename t_fn>
inline void parallel_for(std::size_t n, unsigned n_threads, t_fn &&fn)
{
	if (n_threads <= 1 || n <= 1)
	{
		for (std::size_t i = 0; i < n; ++i) fn(i);
		return;
	}
	unsigned const k = (unsigned) std::min <std::size_t>(n_threads, n);
	std::function <void(std::size_t)> const f(std::ref(fn));
	worker_pool::instance().run(n, k - 1, f);
}

// The CPUs this process may use: the machine's, or fewer under a cgroup CPU quota (a container on a large host — more
// threads than the quota only take each other's time slices).
inline unsigned default_io_threads()
{
	unsigned hc = std::thread::hardware_concurrency();
	if (0 == hc) hc = 4;
	if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r"))		// cgroup v2: "<quota> <period>" or "max <period>"
	{
		char quota[32] = "";
		long period = 0;
		if (2 == std::fscanf(f, "%31s %ld", quota, &period) && period > 0 && 0 != std::strcmp(quota, "max"))
		{
			long const q = std::atol(quota);
			if (q > 0) hc = std::min <unsigned>(hc, (unsigned) std::max <long>(1, (q + period - 1) / period));
		}
		std::fclose(f);
	}
	return std::min(hc, 32u);
}

// An inflater of whole groups of BGZF blocks that lives elsewhere (the GPU: gpu_deflate.hpp over include/pvc3_bgzf.h).  The reader
// reads a piece of the file into input(), lists the piece's blocks (block: the layout of pvc3_bgzf_block), and finds their
// bytes in output(slot) after wait(); a block whose status is not 0 it inflates itself.  Both calls throw io_error.
class group_inflater
{
public:
	struct block { uint64_t in_off; uint32_t in_len; uint32_t out_len; uint64_t out_off; };
	virtual ~group_inflater() = default;
	virtual unsigned slots() const = 0;					// at least 2: a piece is consumed while the next one is made
	virtual std::size_t max_in() const = 0;
	virtual std::size_t max_out() const = 0;
	virtual std::size_t max_blocks() const = 0;
	virtual std::size_t blocks_in_flight() const = 0;		// a group of that many blocks takes as long as a group of one
	virtual uint8_t *input(unsigned slot) = 0;
	virtual uint8_t *output(unsigned slot) = 0;
	virtual block *blocks(unsigned slot) = 0;
	virtual void submit(unsigned slot, std::size_t n_blocks, std::size_t in_bytes, std::size_t out_bytes) = 0;
	virtual uint32_t const *wait(unsigned slot) = 0;
};

// Sequential byte source: a file or stdin.  BGZF input (BAM, bgzip'ed SAM) is inflated block-parallel; plain gzip and
// uncompressed input go through zlib's gz* interface.
class byte_reader
{
	gzFile m_file{};
	// BGZF path
	FILE *m_raw{};
	unsigned m_threads{1};
	std::vector <uint8_t> m_cbuf;		// compressed bytes not yet inflated (whole blocks + a partial tail)
	std::size_t m_clen{};
	// Inflated bytes: the piece being consumed and the next one, which a second thread reads and inflates meanwhile.
	// (ptr: data.data(), or the page-locked output of the group_inflater's slot `slot` when that made the piece)
	struct piece { std::vector <uint8_t> data; uint8_t *ptr{}; std::size_t len{}; unsigned slot{}; };
	piece m_cur, m_next;
	std::atomic <group_inflater *> m_group{nullptr};		// attach(): pieces are inflated there from the next one on
	group_inflater *m_group_in_use{};						// the producing thread's: the compressed bytes lie in its input(m_in_slot)
	unsigned m_in_slot{};
	uint64_t m_group_blocks{}, m_group_refused{};
	std::size_t m_dpos{};
	bool m_eof{};
	bool m_started{};
	std::future <bool> m_ahead;

	// Compressed bytes read (and inflated) per piece; PVC3_BGZF_CHUNK overrides it so that a test can make records straddle pieces.
	std::size_t CHUNK = []() -> std::size_t {
		char const *e = std::getenv("PVC3_BGZF_CHUNK");
		std::size_t const v = e ? (std::size_t) std::strtoull(e, nullptr, 10) : 0;
		return v ? v : (std::size_t) (32u << 20);
	}();

	struct block { std::size_t in_off, in_len, out_off, out_len; uint32_t crc; };

	// Length of the BGZF block at p (n bytes available) or 0 if incomplete; throws if p is not a BGZF block.
	static std::size_t block_size(uint8_t const *p, std::size_t n, std::size_t &data_off)
	{
		if (n < 18) return 0;
		if (!(0x1f == p[0] && 0x8b == p[1] && 8 == p[2] && (p[3] & 4))) throw io_error("corrupt BGZF stream (bad block header)");
		std::size_t const xlen = p[10] | ((std::size_t) p[11] << 8);
		if (n < 12 + xlen) return 0;
		for (std::size_t o = 12; o + 4 <= 12 + xlen; )
		{
			std::size_t const slen = p[o + 2] | ((std::size_t) p[o + 3] << 8);
			if ('B' == p[o] && 'C' == p[o + 1] && 2 == slen)
			{
				data_off = 12 + xlen;
				return (std::size_t) (p[o + 4] | ((std::size_t) p[o + 5] << 8)) + 1;
			}
			o += 4 + slen;
		}
		throw io_error("corrupt BGZF stream (no BC field)");
	}

	// One block on this thread: the block's own decoder first (fast_inflate.hpp); whatever it does not accept, or whose checksum
	// does not come out, goes through zlib, which then decides what is corrupt.
	static void inflate_block(uint8_t const *src, std::size_t in_len, uint8_t *dst, std::size_t out_len, uint32_t crc)
	{
		if (0 == out_len) return;
		static thread_local fast_inflater quick;
		if (quick.inflate(src, in_len, dst, out_len) && fast_crc32(0, dst, out_len) == crc)
			return;
		z_stream zs{};
		if (Z_OK != inflateInit2(&zs, -15)) throw io_error("inflateInit2 failed");
		zs.next_in = const_cast <uint8_t *>(src); zs.avail_in = (uInt) in_len;
		zs.next_out = dst; zs.avail_out = (uInt) out_len;
		int const rc = inflate(&zs, Z_FINISH);
		inflateEnd(&zs);
		if (Z_STREAM_END != rc || zs.total_out != out_len) throw io_error("corrupt BGZF block");
		if (fast_crc32(0, dst, out_len) != crc) throw io_error("BGZF block checksum mismatch");
	}

	// The next piece of the file into `dst` (false at the end); touches only m_raw, m_cbuf, m_clen, m_eof and m_group_in_use.
	bool produce(piece &dst)
	{
		dst.len = 0;
		// A group_inflater attached since the last piece: the compressed bytes at hand move to its page-locked input, where
		// the file is read to from now on.
		if (group_inflater *g = m_group.load(std::memory_order_acquire); g && g != m_group_in_use)
		{
			// (pieces of at most half the inflater's input from here on: the page-locked output is sized for them)
			CHUNK = std::min(CHUNK, g->max_in() / 2);
			if (CHUNK >= (1u << 16) && m_clen + CHUNK + (1u << 17) <= g->max_in())
			{
				std::memcpy(g->input(dst.slot), m_cbuf.data(), m_clen);
				m_group_in_use = g; m_in_slot = dst.slot;
			}
			else
				m_group.store(nullptr);		// (a piece would not fit: PVC3_BGZF_CHUNK larger than the inflater was made for)
		}
		while (0 == dst.len)
		{
			group_inflater *g = m_group_in_use;
			if (!g) m_cbuf.resize(std::max(m_cbuf.size(), m_clen + CHUNK));
			if (g && m_in_slot != dst.slot)		// the input of a slot goes with its output: what the last piece left over moves along
			{
				std::memcpy(g->input(dst.slot), g->input(m_in_slot), m_clen);
				m_in_slot = dst.slot;
			}
			uint8_t *cbuf = g ? g->input(m_in_slot) : m_cbuf.data();
			if (!m_eof)
			{
				std::size_t const got = std::fread(cbuf + m_clen, 1, CHUNK, m_raw);
				if (got < CHUNK) { if (std::ferror(m_raw)) throw io_error("read error"); m_eof = true; }
				m_clen += got;
			}
			std::vector <block> blocks;
			std::size_t off = 0, out = 0;
			for (;;)
			{
				std::size_t data_off = 0;
				std::size_t const bs = block_size(cbuf + off, m_clen - off, data_off);
				if (0 == bs || off + bs > m_clen) break;
				if (bs < data_off + 8) throw io_error("corrupt BGZF stream (short block)");
				uint32_t crc, isize;
				std::memcpy(&crc, cbuf + off + bs - 8, 4);
				std::memcpy(&isize, cbuf + off + bs - 4, 4);
				blocks.push_back(block{off + data_off, bs - data_off - 8, out, isize, crc});
				out += isize; off += bs;
			}
			if (blocks.empty())
			{
				if (m_eof) { if (m_clen) throw io_error("truncated BGZF stream"); return false; }
				continue;
			}
			dst.len = out;
			if (g && out <= g->max_out())
			{
				// As many of the piece's blocks as the GPU works on at a time go there (a group of them takes the GPU as long as one
				// block), the rest are inflated by the threads here meanwhile; all bytes land in the page-locked output of this piece's
				// slot.  What the GPU refuses is inflated here afterwards, and every block's checksum is verified here.
				std::size_t const n_group = std::min(blocks.size(), std::max <std::size_t>(1, std::min(g->blocks_in_flight(), g->max_blocks())));
				group_inflater::block *gb = g->blocks(dst.slot);
				for (std::size_t i = 0; i < n_group; ++i)
					gb[i] = group_inflater::block{blocks[i].in_off, (uint32_t) blocks[i].in_len, (uint32_t) blocks[i].out_len, blocks[i].out_off};
				std::size_t const out_group = (n_group < blocks.size()) ? blocks[n_group].out_off : out;
				std::size_t const in_group = (n_group < blocks.size()) ? blocks[n_group].in_off : off;
				g->submit(dst.slot, n_group, in_group, out_group);
				uint8_t *dbuf = g->output(dst.slot);
				dst.ptr = dbuf;
				std::exception_ptr here_error;
				try
				{
					parallel_for(blocks.size() - n_group, m_threads, [&](std::size_t k) {
						block const &b = blocks[n_group + k];
						inflate_block(cbuf + b.in_off, b.in_len, dbuf + b.out_off, b.out_len, b.crc);
					});
				}
				catch (...) { here_error = std::current_exception(); }
				uint32_t const *status = g->wait(dst.slot);		// (also when a block failed here: the slot must not stay busy)
				if (here_error) std::rethrow_exception(here_error);
				std::atomic <uint64_t> refused{0};
				std::size_t const n_tasks = std::min <std::size_t>(n_group, 8 * (std::size_t) m_threads);
				parallel_for(n_tasks, m_threads, [&](std::size_t t) {
					for (std::size_t i = n_group * t / n_tasks; i < n_group * (t + 1) / n_tasks; ++i)
					{
						block const &b = blocks[i];
						if (0 == b.out_len) continue;
						if (0 == status[i] && fast_crc32(0, dbuf + b.out_off, b.out_len) == b.crc) continue;
						++refused;
						inflate_block(cbuf + b.in_off, b.in_len, dbuf + b.out_off, b.out_len, b.crc);
					}
				});
				m_group_blocks += n_group; m_group_refused += refused.load();
			}
			else
			{
				if (dst.data.size() < out) dst.data.resize(out);		// grows only: no zero-fill per piece
				uint8_t *dbuf = dst.data.data();
				dst.ptr = dbuf;
				parallel_for(blocks.size(), m_threads, [&](std::size_t i) {
					block const &b = blocks[i];
					inflate_block(cbuf + b.in_off, b.in_len, dbuf + b.out_off, b.out_len, b.crc);
				});
			}
			std::memmove(cbuf, cbuf + off, m_clen - off);
			m_clen -= off;
			if (0 == dst.len && m_eof && 0 == m_clen) return false;		// only the EOF marker was left
		}
		return true;
	}

	// Makes the piece read ahead the current one and starts reading the one after it.
	bool refill()
	{
		m_cur.len = 0; m_dpos = 0;
		if (!m_started)
		{
			m_started = true;
			m_ahead = std::async(std::launch::async, [this]() { return produce(m_next); });
		}
		if (!m_ahead.valid())
			return false;
		if (!m_ahead.get())		// rethrows what the reading thread threw
			return false;
		std::swap(m_cur, m_next);
		m_ahead = std::async(std::launch::async, [this]() { return produce(m_next); });
		return true;
	}

public:
	// Length of the BGZF block at p and where its deflate data begin (the BC subfield is found by walking the extra field).
	static std::size_t bgzf_block_size(uint8_t const *p, std::size_t n, std::size_t &data_off) { return block_size(p, n, data_off); }

	byte_reader() = default;
	byte_reader(byte_reader const &) = delete;
	~byte_reader() { close(); }

	void set_threads(unsigned n) { m_threads = n ? n : 1; }
	bool is_bgzf_file() const { return nullptr != m_raw; }
	void set_piece_bytes(std::size_t n) { if (!m_started && n >= (1u << 16)) CHUNK = n; }		// compressed bytes read and inflated at a time (before the first read)

	// From the next piece on the blocks of BGZF input are inflated by `g` (which must outlive this reader's use).  May be called
	// from another thread than the one that reads.
	void attach(group_inflater *g) { m_group.store(g, std::memory_order_release); }
	uint64_t group_blocks() const { return m_group_blocks; }
	uint64_t group_refused() const { return m_group_refused; }

	void open(char const *path)		// nullptr or "-" = stdin
	{
		close();
		bool const is_stdin = !path || 0 == std::strcmp(path, "-");
		if (!is_stdin)
		{
			// BGZF?  Then inflate the blocks ourselves, several at a time.
			FILE *f = std::fopen(path, "rb");
			if (!f) throw io_error(std::string("unable to open ") + path);
			uint8_t head[18];
			std::size_t const got = std::fread(head, 1, sizeof(head), f);
			// The usual layout (htslib, this writer): BC is the only extra field.  Anything else is left to zlib.
			bool const bgzf = 18 == got && 0x1f == head[0] && 0x8b == head[1] && 8 == head[2] && 4 == head[3] && 6 == head[10] && 0 == head[11]
				&& 'B' == head[12] && 'C' == head[13] && 2 == head[14] && 0 == head[15];
			if (bgzf)
			{
				std::rewind(f);
				m_raw = f; m_clen = 0; m_eof = false; m_cur.len = 0; m_dpos = 0; m_started = false;
				m_cur.slot = 0; m_next.slot = 1; m_group_in_use = nullptr;
				return;
			}
			std::fclose(f);
		}
		m_file = is_stdin ? gzdopen(0, "rb") : gzopen(path, "rb");
		if (!m_file)
			throw io_error(std::string("unable to open ") + (path ? path : "stdin"));
		gzbuffer(m_file, 1 << 20);
	}

	void close()
	{
		if (m_ahead.valid()) { try { (void) m_ahead.get(); } catch (...) {} }
		if (m_file) { gzclose(m_file); m_file = nullptr; }
		if (m_raw) { std::fclose(m_raw); m_raw = nullptr; }
	}

	// close(), and the buffers given back (a process that has used the GPU pays at exit for the memory it still holds).
	void release()
	{
		close();
		std::vector <uint8_t>().swap(m_cbuf);
		std::vector <uint8_t>().swap(m_cur.data);
		std::vector <uint8_t>().swap(m_next.data);
		m_cur.ptr = m_next.ptr = nullptr; m_cur.len = m_next.len = 0; m_dpos = 0; m_clen = 0;
		m_group.store(nullptr); m_group_in_use = nullptr;
	}

	// Reads exactly n bytes unless the stream ends; returns the number read.
	std::size_t read(void *dst, std::size_t n)
	{
		std::size_t got = 0;
		if (m_raw)
		{
			while (got < n)
			{
				if (m_dpos == m_cur.len && !refill()) break;
				std::size_t const take = std::min(n - got, m_cur.len - m_dpos);
				std::memcpy((char *) dst + got, m_cur.ptr + m_dpos, take);
				m_dpos += take; got += take;
			}
			return got;
		}
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

	// The unread bytes of the current inflated piece (BGZF input only; false for other inputs and at the end of the stream), for
	// callers that frame records in place, and consume() to step over what they used.
	bool span(uint8_t const *&p, std::size_t &n)
	{
		if (!m_raw) return false;
		if (m_dpos == m_cur.len && !refill()) return false;
		p = m_cur.ptr + m_dpos;
		n = m_cur.len - m_dpos;
		return true;
	}

	void consume(std::size_t n) { m_dpos += n; }

	// Appends exactly n bytes to dst unless the stream ends; returns the number appended.  (No zero-fill before the copy, which
	// resize() + read() would do: a BAM reader appends every record of a batch to one arena.)
	std::size_t read_append(std::vector <uint8_t> &dst, std::size_t n)
	{
		if (!m_raw)
		{
			std::size_t const at = dst.size();
			dst.resize(at + n);
			std::size_t const got = read(dst.data() + at, n);
			dst.resize(at + got);
			return got;
		}
		std::size_t got = 0;
		while (got < n)
		{
			if (m_dpos == m_cur.len && !refill()) break;
			std::size_t const take = std::min(n - got, m_cur.len - m_dpos);
			uint8_t const *p = m_cur.ptr + m_dpos;
			dst.insert(dst.end(), p, p + take);
			m_dpos += take; got += take;
		}
		return got;
	}

	// One text line without its terminator; false at the end of the stream.
	bool getline(std::string &line)
	{
		line.clear();
		if (m_raw)
		{
			bool any = false;
			for (;;)
			{
				if (m_dpos == m_cur.len && !refill()) break;
				any = true;
				uint8_t const *b = m_cur.ptr + m_dpos;
				std::size_t const left = m_cur.len - m_dpos;
				uint8_t const *nl = (uint8_t const *) std::memchr(b, '\n', left);
				if (nl)
				{
					line.append((char const *) b, (std::size_t) (nl - b));
					m_dpos += (std::size_t) (nl - b) + 1;
					if (!line.empty() && '\r' == line.back()) line.pop_back();
					return true;
				}
				line.append((char const *) b, left);
				m_dpos += left;
			}
			return any && !line.empty();
		}
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
		if (m_raw)
		{
			if (m_dpos == m_cur.len && !refill()) return -1;
			return m_cur.ptr[m_dpos];
		}
		int const c = gzgetc(m_file);
		if (c >= 0) gzungetc(c, m_file);
		return c;
	}
};

// A compressor of whole groups of BGZF blocks that lives elsewhere (the GPU: gpu_deflate.hpp over include/pvc3_bgzf.h).  The
// writer fills the input of a slot — block i's bytes at i * in_stride() + in_pad(), their number in lengths()[i] — submits it and
// goes on with the next slot; wait() gives the raw DEFLATE stream of block i at out + i * out_stride(), out_len[i] bytes.
// Both calls throw io_error.
class group_deflater
{
public:
	virtual ~group_deflater() = default;
	virtual unsigned slots() const = 0;
	virtual std::size_t max_blocks() const = 0;
	virtual std::size_t in_stride() const = 0;
	virtual std::size_t in_pad() const = 0;
	virtual std::size_t out_stride() const = 0;
	virtual uint8_t *input(unsigned slot) = 0;
	virtual uint32_t *lengths(unsigned slot) = 0;
	virtual void submit(unsigned slot, std::size_t n_blocks) = 0;
	virtual void wait(unsigned slot, uint8_t const *&out, uint32_t const *&out_len) = 0;
};

// Sequential byte sink: plain (SAM text) or BGZF blocks (BAM).  Blocks are cut every 0xff00 bytes whatever the number of
// threads, so the file does not depend on it; they are deflated a group at a time, in parallel.
class byte_writer
{
	FILE *m_file{};
	bool m_owns{};
	bool m_bgzf{};
	int m_level{-1};
	unsigned m_threads{1};
	std::vector <uint8_t> m_pending;	// uncompressed bytes not yet written
	std::vector <std::vector <uint8_t>> m_out, m_out_writing;		// deflated blocks: being made / on their way to the file
	std::future <void> m_writing;

	// With a group_deflater attached the bytes go straight to its slots instead of m_pending.
	group_deflater *m_group{};
	unsigned m_slot{};							// the slot being filled
	std::size_t m_blk{}, m_fill{};				// whole blocks in it, bytes of the block after them
	std::deque <std::pair <unsigned, std::size_t>> m_in_flight;		// (slot, blocks) submitted, oldest first
	std::vector <uint8_t> m_joined, m_joined_writing;				// a group's BGZF blocks back to back
	std::vector <std::size_t> m_offsets;

	static constexpr std::size_t BLOCK_DATA = 0xff00;	// uncompressed bytes per block (as htslib)

	void write_raw(void const *p, std::size_t n)
	{
		if (n && 1 != std::fwrite(p, n, 1, m_file))
			throw io_error("write error");
	}

	// BGZF block = gzip member with the BC extra field (SAMv1 §4.1.1).  Level < 0: the built-in encoder (fast_deflate.hpp),
	// otherwise zlib at that level.
	void deflate_block(uint8_t const *src, std::size_t len, std::vector <uint8_t> &out) const
	{
		std::size_t clen = 0;
		if (m_level < 0)
		{
			thread_local fast_deflater enc;
			out.resize(18 + fast_deflater::bound(len) + 8);
			clen = enc.compress(src, len, out.data() + 18, out.size() - 18 - 8);
			if (0 == clen) throw io_error("deflate failed");
		}
		else
		{
			out.resize(18 + compressBound((uLong) len) + 8);
			z_stream zs{};
			if (Z_OK != deflateInit2(&zs, m_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY))
				throw io_error("deflateInit2 failed");
			zs.next_in = const_cast <uint8_t *>(src); zs.avail_in = (uInt) len;
			zs.next_out = out.data() + 18; zs.avail_out = (uInt) (out.size() - 18 - 8);
			if (Z_STREAM_END != deflate(&zs, Z_FINISH)) { deflateEnd(&zs); throw io_error("deflate failed"); }
			clen = zs.total_out;
			deflateEnd(&zs);
		}
		frame_block(out.data(), clen, src, len);
		out.resize(18 + clen + 8);
	}

	// The gzip header with the BC field in front of the clen bytes of DEFLATE data at block + 18, CRC32 and ISIZE behind them.
	static void frame_block(uint8_t *block, std::size_t clen, uint8_t const *src, std::size_t len)
	{
		std::size_t const bsize = 18 + clen + 8;
		static uint8_t const head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
		std::memcpy(block, head, 16);
		block[16] = (uint8_t) ((bsize - 1) & 0xff); block[17] = (uint8_t) ((bsize - 1) >> 8);
		uint32_t const crc = fast_crc32(0, src, len);
		uint32_t const isize = (uint32_t) len;
		std::memcpy(block + 18 + clen, &crc, 4);
		std::memcpy(block + 18 + clen + 4, &isize, 4);
	}

	void group_write(uint8_t const *src, std::size_t n)
	{
		std::size_t const stride = m_group->in_stride(), pad = m_group->in_pad();
		while (n)
		{
			std::size_t const k = std::min(n, BLOCK_DATA - m_fill);
			std::memcpy(m_group->input(m_slot) + m_blk * stride + pad + m_fill, src, k);
			m_fill += k; src += k; n -= k;
			if (BLOCK_DATA == m_fill)
			{
				m_group->lengths(m_slot)[m_blk++] = (uint32_t) BLOCK_DATA;
				m_fill = 0;
				if (m_blk == m_group->max_blocks()) group_submit();
			}
		}
	}

	// The slot's blocks (a partial last one included: only close() calls this with one) go to the compressor, and the next slot
	// is made free: its group, submitted a round ago, is framed and written.
	void group_submit()
	{
		if (m_fill) { m_group->lengths(m_slot)[m_blk++] = (uint32_t) m_fill; m_fill = 0; }
		if (m_blk)
		{
			m_group->submit(m_slot, m_blk);
			m_in_flight.emplace_back(m_slot, m_blk);
		}
		m_blk = 0;
		m_slot = (m_slot + 1) % m_group->slots();
		while (std::any_of(m_in_flight.begin(), m_in_flight.end(), [this](auto const &g) { return g.first == m_slot; }))
			group_retire();
	}

	void group_retire()
	{
		auto const [slot, n] = m_in_flight.front();
		m_in_flight.pop_front();
		uint8_t const *out{};
		uint32_t const *out_len{};
		m_group->wait(slot, out, out_len);
		uint8_t const *in = m_group->input(slot);
		uint32_t const *len = m_group->lengths(slot);
		std::size_t const stride = m_group->in_stride(), pad = m_group->in_pad(), out_stride = m_group->out_stride();
		m_offsets.resize(n + 1);
		m_offsets[0] = 0;
		for (std::size_t i = 0; i < n; ++i)
		{
			if (out_len[i] > BLOCK_DATA + 5) throw io_error("deflate failed");
			m_offsets[i + 1] = m_offsets[i] + 18 + out_len[i] + 8;
		}
		if (m_joined.size() < m_offsets[n]) m_joined.resize(m_offsets[n]);
		parallel_for(n, m_threads, [&](std::size_t i) {
			uint8_t *block = m_joined.data() + m_offsets[i];
			std::memcpy(block + 18, out + i * out_stride, out_len[i]);
			frame_block(block, out_len[i], in + i * stride + pad, len[i]);
		});
		if (m_writing.valid()) m_writing.get();		// rethrows a write error
		m_joined.swap(m_joined_writing);
		std::size_t const total = m_offsets[n];
		m_writing = std::async(std::launch::async, [this, total]() { write_raw(m_joined_writing.data(), total); });
	}

	// Writes all whole blocks of m_pending (and the partial last one if `all`).
	void flush_blocks(bool all)
	{
		std::size_t const n_full = m_pending.size() / BLOCK_DATA;
		std::size_t const tail = m_pending.size() - n_full * BLOCK_DATA;
		std::size_t const n = n_full + ((all && tail) ? 1 : 0);
		if (0 == n) return;
		if (m_out.size() < n) m_out.resize(n);
		parallel_for(n, m_threads, [&](std::size_t i) {
			std::size_t const len = (i < n_full) ? BLOCK_DATA : tail;
			deflate_block(m_pending.data() + i * BLOCK_DATA, len, m_out[i]);
		});
		// The blocks go to the file on a thread of their own while the next group is deflated.
		if (m_writing.valid()) m_writing.get();		// rethrows a write error
		m_out.swap(m_out_writing);
		m_writing = std::async(std::launch::async, [this, n]() {
			for (std::size_t i = 0; i < n; ++i)
				write_raw(m_out_writing[i].data(), m_out_writing[i].size());
		});
		if (n == n_full && tail)
		{
			std::memmove(m_pending.data(), m_pending.data() + n_full * BLOCK_DATA, tail);
			m_pending.resize(tail);
		}
		else
			m_pending.clear();
	}

public:
	byte_writer() = default;
	byte_writer(byte_writer const &) = delete;
	~byte_writer() { try { close(); } catch (...) {} }

	void set_threads(unsigned n) { m_threads = n ? n : 1; }

	// From here on the blocks are compressed by `g` (which must outlive close()); what has been written and not yet compressed
	// moves to its first slot, so the blocks are cut at the same bytes as without it.  Only for BGZF output with the built-in
	// encoder (level < 0): a zlib level the user asked for stays zlib's.
	void attach(group_deflater *g)
	{
		if (!g || !m_file || !m_bgzf || m_level >= 0 || m_group) return;
		m_group = g; m_slot = 0; m_blk = 0; m_fill = 0;
		std::vector <uint8_t> held;
		held.swap(m_pending);
		group_write(held.data(), held.size());
	}

	void open(char const *path, bool bgzf, int level = -1)		// nullptr or "-" = stdout
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
	}

	void write(void const *p, std::size_t n)
	{
		if (!m_bgzf) { write_raw(p, n); return; }
		uint8_t const *src = (uint8_t const *) p;
		if (m_group) { group_write(src, n); return; }
		m_pending.insert(m_pending.end(), src, src + n);
		if (m_pending.size() >= BLOCK_DATA * 16 * (std::size_t) m_threads)
			flush_blocks(false);
	}

	void write(std::string const &s) { write(s.data(), s.size()); }

	// Everything written so far as BGZF blocks of its own (the last one short), on the file when this returns; and BGZF blocks made
	// elsewhere, written as they are.  (For the record path on the GPU, which makes the blocks of the records itself.)
	void flush_all()
	{
		if (!m_bgzf || m_group) throw io_error("flush_all: only for BGZF output without an attached compressor");
		flush_blocks(true);
		if (m_writing.valid()) m_writing.get();
	}
	void write_framed(void const *p, std::size_t n)
	{
		if (!m_bgzf || m_group || !m_pending.empty()) throw io_error("write_framed: unwritten bytes in front of the blocks");
		if (m_writing.valid()) m_writing.get();
		// A regular file takes the bytes from several threads at once, each at its own place (one thread copies into the page
		// cache at 4 - 5 GB/s, which was what a run of the record path on the GPU waited for); a pipe takes them in order.
		int const fd = fileno(m_file);
		off_t at = -1;
		if (n >= (std::size_t(8) << 20) && m_threads > 1 && 0 == std::fflush(m_file)) at = lseek(fd, 0, SEEK_CUR);
		if (at < 0) { write_raw(p, n); return; }
		// (four, eight or sixteen threads: the same 9.6 GB/s — buffered writes to one file take turns inside the kernel)
		unsigned const parts = std::min <unsigned>(m_threads, 8);
		std::vector <std::future <bool>> jobs;
		for (unsigned k = 0; k < parts; ++k)
		{
			std::size_t const lo = n * k / parts, hi = n * (k + 1) / parts;
			jobs.push_back(std::async(std::launch::async, [=]() {
				uint8_t const *src = (uint8_t const *) p + lo;
				std::size_t left = hi - lo;
				off_t o = at + (off_t) lo;
				while (left)
				{
					ssize_t const w = pwrite(fd, src, left, o);
					if (w <= 0) { if (w < 0 && EINTR == errno) continue; return false; }
					src += w; left -= (std::size_t) w; o += w;
				}
				return true;
			}));
		}
		bool ok = true;
		for (auto &j : jobs) ok = j.get() && ok;
		if (!ok) throw io_error("write error");
		if (lseek(fd, at + (off_t) n, SEEK_SET) < 0) throw io_error("write error");
	}

	void close()
	{
		if (!m_file) return;
		// Whatever happens below, this object is closed afterwards: the destructor must not run close() again over the same
		// pending data after an error.
		FILE *f = m_file;
		bool const owns = m_owns;
		struct closer
		{
			byte_writer &w; FILE *f; bool owns; bool done;
			~closer()
			{
				if (done) return;
				// an error on the way: a write that is still running on the file finishes first, and the groups on the compressor are waited for
				if (w.m_writing.valid()) { try { w.m_writing.get(); } catch (...) {} }
				if (w.m_group)
					for (auto const &g : w.m_in_flight) { try { uint8_t const *o{}; uint32_t const *l{}; w.m_group->wait(g.first, o, l); } catch (...) {} }
				w.m_pending.clear(); w.m_group = nullptr; w.m_in_flight.clear(); w.m_file = nullptr;
				if (owns) std::fclose(f);
			}
		} guard{*this, f, owns, false};
		if (m_bgzf)
		{
			if (m_group)
			{
				group_submit();
				while (!m_in_flight.empty()) group_retire();
				m_group = nullptr;
			}
			else
				flush_blocks(true);
			if (m_writing.valid()) m_writing.get();
			// The empty EOF marker block.
			static uint8_t const eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
			write_raw(eof, sizeof(eof));
		}
		// The last buffered bytes (the EOF block among them) reach the file only here: a full disk must not end in a truncated
		// file and exit code 0.
		bool const flushed = 0 == std::fflush(f);
		bool const closed = !owns || 0 == std::fclose(f);
		guard.done = true;
		m_pending.clear();
		m_file = nullptr;
		if (!flushed || !closed) throw io_error("write error while closing the output (disk full?)");
	}
};

}
