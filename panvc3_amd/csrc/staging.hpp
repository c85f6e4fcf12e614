// staging.hpp — copies between ORDINARY (pageable) host arrays and the device through a ring of page-locked chunks.
//
// pvc3_batch_project and pvc3_dbatch_create / _download take plain host arrays (include/pvc3.h: the call INTEGRATION.md shows
// with std::vector).  hipMemcpy from or to pageable memory goes through the runtime's own staging, whose speed was seen to
// differ twelvefold between boxes of one pool (6.6 .. 78 M records/s for the same call).  Here the copy is done in the open:
// a few threads move the bytes into page-locked chunks of a fixed size while the previous chunks travel, so that the rate
// is that of memcpy by `threads` cores or of PCIe, whichever is lower, on every box.  Page-locked arrays (pvc3_host_alloc)
// are recognised and copied directly.
#pragma once

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace pvc3 {

class copy_pool
{
	struct part { uint8_t *dst; uint8_t const *src; size_t n; };
	std::vector <std::thread> m_threads;
	std::mutex m_mutex;
	std::condition_variable m_cv, m_done_cv;
	std::vector <part> m_parts;
	uint64_t m_generation{};
	unsigned m_pending{};
	bool m_stop{};

	void loop(unsigned idx)
	{
		uint64_t seen = 0;
		for (;;)
		{
			part p{};
			{
				std::unique_lock <std::mutex> lock(m_mutex);
				m_cv.wait(lock, [&]() { return m_stop || m_generation != seen; });
				if (m_stop) return;
				seen = m_generation;
				p = m_parts[idx];
			}
			if (p.n) std::memcpy(p.dst, p.src, p.n);
			{
				std::lock_guard <std::mutex> lock(m_mutex);
				if (0 == --m_pending) m_done_cv.notify_one();
			}
		}
	}

public:
	explicit copy_pool(unsigned helpers)
	{
		m_parts.resize(helpers);
		for (unsigned i = 0; i < helpers; ++i) m_threads.emplace_back([this, i]() { loop(i); });
	}
	copy_pool(copy_pool const &) = delete;
	~copy_pool()
	{
		{ std::lock_guard <std::mutex> lock(m_mutex); m_stop = true; }
		m_cv.notify_all();
		for (auto &t : m_threads) t.join();
	}

	// memcpy(dst, src, n) by the calling thread and the helpers together.
	void copy(void *dst, void const *src, size_t n)
	{
		unsigned const helpers = (unsigned) m_threads.size();
		if (0 == helpers || n < (size_t(1) << 20)) { if (n) std::memcpy(dst, src, n); return; }
		size_t const per = (n / (helpers + 1) + 4095) & ~size_t(4095);
		size_t off = (per < n) ? per : n;		// the caller's own part comes first
		{
			std::lock_guard <std::mutex> lock(m_mutex);
			for (unsigned i = 0; i < helpers; ++i)
			{
				size_t const k = (off < n) ? ((n - off < per || i + 1 == helpers) ? (n - off) : per) : 0;
				m_parts[i] = part{(uint8_t *) dst + off, (uint8_t const *) src + off, k};
				off += k;
			}
			m_pending = helpers;
			++m_generation;
		}
		m_cv.notify_all();
		std::memcpy(dst, src, (per < n) ? per : n);
		std::unique_lock <std::mutex> lock(m_mutex);
		m_done_cv.wait(lock, [&]() { return 0 == m_pending; });
	}
};


class host_stager
{
public:
	static constexpr unsigned RING = 4;
	static constexpr size_t CHUNK = size_t(8) << 20;

private:
	uint8_t *m_buf[RING]{};
	hipEvent_t m_ev[RING]{};
	bool m_used[RING]{};
	unsigned m_next{};
	copy_pool *m_pool{};
	unsigned m_threads{4};

	hipError_t ensure()
	{
		if (m_buf[0]) return hipSuccess;
		for (unsigned i = 0; i < RING; ++i)
		{
			hipError_t e = hipHostMalloc((void **) &m_buf[i], CHUNK, hipHostMallocDefault);
			if (hipSuccess == e) e = hipEventCreateWithFlags(&m_ev[i], hipEventDisableTiming);
			if (hipSuccess != e) return e;
		}
		m_pool = new copy_pool(m_threads > 1 ? m_threads - 1 : 0);
		return hipSuccess;
	}

public:
	host_stager() = default;
	host_stager(host_stager const &) = delete;
	~host_stager()
	{
		delete m_pool;
		for (unsigned i = 0; i < RING; ++i)
		{
			if (m_ev[i]) (void) hipEventDestroy(m_ev[i]);
			if (m_buf[i]) (void) hipHostFree(m_buf[i]);
		}
	}

	void set_threads(unsigned n) { if (!m_pool) m_threads = n ? n : 1; }

	static bool device_visible(void const *p)
	{
		if (!p) return false;
		hipPointerAttribute_t attr{};
		if (hipSuccess != hipPointerGetAttributes(&attr, p)) { (void) hipGetLastError(); return false; }
		return hipMemoryTypeHost == attr.type;
	}

	// dst (device) <- src (host), enqueued on `s`.  When it returns, `src` may be changed if it is pageable; a page-locked
	// source must stay as it is until the copy has run (as with hipMemcpyAsync).
	hipError_t h2d(void *dst, void const *src, size_t bytes, hipStream_t s)
	{
		if (0 == bytes) return hipSuccess;
		if (bytes < 65536 || device_visible(src)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
		if (hipError_t const e = ensure()) return e;
		for (size_t off = 0; off < bytes; off += CHUNK)
		{
			size_t const k = (bytes - off < CHUNK) ? (bytes - off) : CHUNK;
			unsigned const r = m_next; m_next = (m_next + 1) % RING;
			if (m_used[r]) if (hipError_t const e = hipEventSynchronize(m_ev[r])) return e;
			m_pool->copy(m_buf[r], (uint8_t const *) src + off, k);
			if (hipError_t const e = hipMemcpyAsync((uint8_t *) dst + off, m_buf[r], k, hipMemcpyHostToDevice, s)) return e;
			if (hipError_t const e = hipEventRecord(m_ev[r], s)) return e;
			m_used[r] = true;
		}
		return hipSuccess;
	}

	// dst (host) <- src (device); returns when the bytes are in dst.  `s`: the stream whose earlier work produced src.
	hipError_t d2h(void *dst, void const *src, size_t bytes, hipStream_t s)
	{
		if (0 == bytes) return hipSuccess;
		if (bytes < 65536 || device_visible(dst))
		{
			if (hipError_t const e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s)) return e;
			return hipStreamSynchronize(s);
		}
		if (hipError_t const e = ensure()) return e;
		// chunks c .. c + RING - 1 travel while chunk c is copied out of its page-locked buffer
		size_t const n_chunks = (bytes + CHUNK - 1) / CHUNK;
		auto issue = [&](size_t c) -> hipError_t {
			unsigned const r = (unsigned) (c % RING);
			size_t const off = c * CHUNK, k = (bytes - off < CHUNK) ? (bytes - off) : CHUNK;
			if (hipError_t const e = hipMemcpyAsync(m_buf[r], (uint8_t const *) src + off, k, hipMemcpyDeviceToHost, s)) return e;
			return hipEventRecord(m_ev[r], s);
		};
		// (the ring may still hold chunks of an earlier h2d: wait for those first)
		for (unsigned r = 0; r < RING; ++r)
			if (m_used[r]) { if (hipError_t const e = hipEventSynchronize(m_ev[r])) return e; m_used[r] = false; }
		m_next = 0;
		for (size_t c = 0; c < n_chunks && c < RING; ++c) if (hipError_t const e = issue(c)) return e;
		for (size_t c = 0; c < n_chunks; ++c)
		{
			unsigned const r = (unsigned) (c % RING);
			size_t const off = c * CHUNK, k = (bytes - off < CHUNK) ? (bytes - off) : CHUNK;
			if (hipError_t const e = hipEventSynchronize(m_ev[r])) return e;
			m_pool->copy((uint8_t *) dst + off, m_buf[r], k);
			if (c + RING < n_chunks) if (hipError_t const e = issue(c + RING)) return e;
		}
		return hipSuccess;
	}
};

}
