// device_records.hpp — the record loop of panvc3_project_alignments with the records on the GPU from the file's compressed
// blocks to the output's compressed blocks (--record-path=gpu; BGZF BAM file in, BAM out).
//
// What the reference's loop does record by record on one thread (project-alignments/project_alignments.cc:484-655: seqan3's
// BAM reader, the filters, project_task::process(), seqan3's BAM writer) is here three threads that only move bytes:
//   the reader      reads pieces of the file into page-locked memory and lists their BGZF blocks;
//   this thread     has the blocks inflated on the GPU into a stream buffer (include/pvc3_bgzf.h, device-resident form), compares
//                   the checksums the GPU computed with those in the file, and runs the record path on that buffer
//                   (include/pvc3_records.h): records found, filtered, scattered into batch arrays, projected, written anew
//                   straight into the compressor's input; then starts the compressor;
//   the writer      takes the compressor's output — the bytes of the BGZF file, headers and checksums included — and writes it.
// The inflater works on the next piece while the records of this one are processed and the compressor works on the previous
// one: three independent streams on the GPU.  The host touches no record.
#pragma once

#include <zlib.h>

#include <array>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

#include "../../include/pvc3.h"
#include "../../include/pvc3_bgzf.h"
#include "../../include/pvc3_records.h"
#include "bgzf.hpp"

namespace pvc3host {

struct device_records_result
{
	uint64_t matched{}, flags_not_matched{}, ref_id_missing{}, seq_missing{}, records{}, batches{}, n_runs{};
	uint64_t blocks_in{}, blocks_refused{}, bytes_in{}, bytes_out{};
	double inflate_kernel_s{}, deflate_kernel_s{};
	double t_setup{}, t_wait_read{}, t_inflate_wait{}, t_frame{}, t_project{}, t_rewrite{}, t_wait_writer{}, t_reader{}, t_writer{}, t_host_inflate{};
	uint64_t blocks_host{};
	std::vector <std::pair <uint32_t, uint32_t>> ranges;
	double rss_setup_mb{};
	double t_writer_wait{}, t_writer_fetch{}, t_writer_file{};		// of t_writer: waiting for the compressor, its bytes fetched, written to the file
	double t_drain{}, t_total{};				// the writer's last pieces after the main thread's last; the whole call but for the objects' destruction
	std::vector <uint64_t> removed_tags;		// [65536]
	// per piece, seconds since the path began: handed to the inflater, taken up by the main thread, inflated, records found, projected, rewritten (and handed to the compressor)
	std::vector <std::array <double, 6>> piece_times;
};

struct device_records_options
{
	char const *path{};
	int device{};
	pvc3_context *ctx{};
	pvc3_records_config config{};			// tables filled by the caller
	pvc3_params params{};
	uint64_t header_bytes{};				// uncompressed bytes in front of the first record
	bool want_ranges{};
	// Who inflates the blocks of a piece: the GPU (0, the default, and 1) — with gpu_blocks set in mode 0, the GPU that many and the
	// host's threads the rest meanwhile, which was the default while the GPU's inflater did 12 GB/s — or the host's threads all (2).
	int inflate_mode{0};
	unsigned threads{1};
	// Pieces of 24 MB of the file / 80 MB of records: 1 200 blocks a launch keep the inflater near its rate, and the buffers (three
	// of each on the input's side, page-locked memory among them, which costs the process time when it is made and when it ends) are
	// half of what 48 / 160 MB take: 16 M records in 0.67 s instead of 0.74; with 12 / 40 MB the launches are too small (0.77 s).
	std::size_t piece_compressed{std::size_t(24) << 20};		// compressed bytes read at a time at most
	std::size_t piece_stream{std::size_t(80) << 20};		// uncompressed bytes of a piece at most
	uint32_t gpu_blocks{0};									// blocks of a piece that go to the GPU in mode 0; 0: all
	uint32_t inflate_in_flight{0};							// blocks the inflater works on at a time; 0: as many as the chip's LDS holds
	uint32_t piece_blocks{0};								// BGZF blocks of a piece at most; 0: what the sizes above allow
	// the two codecs, made ahead of time by the caller (they are page-locked memory, which takes a while to get); null: made here
	std::function <pvc3_bgzf_inflater *()> take_inflater;
	std::function <pvc3_bgzf_dev *()> take_deflater;
	std::function <void(void const *, std::size_t)> write_blocks;		// BGZF blocks to the output file, in order
	std::function <void(uint64_t)> progress;					// records so far (status lines)
	// The process ends right after the call (the command line): the three objects on the GPU are left to the operating system, which
	// takes back gigabytes of page-locked memory faster than the runtime hands them back one by one.
	bool leave_objects{};
};

namespace detail {

inline void zlib_inflate_block(uint8_t const *src, std::size_t in_len, uint8_t *dst, std::size_t out_len)
{
	z_stream zs{};
	if (Z_OK != inflateInit2(&zs, -15)) throw io_error("inflateInit2 failed");
	zs.next_in = const_cast <uint8_t *>(src); zs.avail_in = (uInt) in_len;
	zs.next_out = dst; zs.avail_out = (uInt) out_len;
	int const rc = inflate(&zs, Z_FINISH);
	inflateEnd(&zs);
	if (Z_STREAM_END != rc || zs.total_out != out_len) throw io_error("corrupt BGZF block");
}

}

// Slots of the compressor: one being filled and compressed, one whose bytes are coming back, one whose bytes are going to the file.
constexpr unsigned device_records_out_slots = 3;

// What the two codecs are made for, from the sizes of a piece (so that a caller can make them ahead of time).
struct device_records_sizes
{
	std::size_t max_in{};
	uint32_t max_blocks_in{}, max_blocks_out{};
	static device_records_sizes of(std::size_t piece_compressed, std::size_t piece_stream)
	{
		device_records_sizes s;
		s.max_in = piece_compressed + (std::size_t(1) << 17);		// what a piece left over of its last block + the next read
		s.max_blocks_in = (uint32_t) (s.max_in / 28 / 8 + 4096);
		s.max_blocks_out = (uint32_t) ((piece_stream + piece_stream / 4) / PVC3_BGZF_BLOCK + 64);		// the records grow by their new tags: a quarter is plenty for records of a hundred bases and more
		return s;
	}
};

inline device_records_result run_device_records(device_records_options const &opt)
{
	device_records_result res;
	res.removed_tags.assign(65536, 0);
	auto now = []() { return std::chrono::steady_clock::now(); };
	auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration <double>(std::chrono::steady_clock::now() - t0).count(); };
	auto bgzf_check = [](int rc, char const *what) { if (rc) throw io_error(std::string(what) + ": " + pvc3_bgzf_last_error()); };
	auto rec_check = [](int rc, char const *what) { if (rc) throw io_error(std::string(what) + ": " + pvc3_last_error()); };

	constexpr unsigned IN_SLOTS = PVC3_RECORDS_SLOTS, OUT_SLOTS = device_records_out_slots;
	auto const t_begin = now();
	device_records_sizes const sz = device_records_sizes::of(opt.piece_compressed, opt.piece_stream);
	std::size_t const max_in = sz.max_in;
	uint32_t const max_blocks_in = sz.max_blocks_in;

	// ---- the three objects on the GPU ---------------------------------------------------------------------------------------------
	pvc3_bgzf_inflater *inflater{};
	pvc3_bgzf_dev *deflater{};
	pvc3_records *records{};
	uint8_t *staging[IN_SLOTS]{};		// page-locked: what the host's threads inflate of a piece, one behind the other
	struct cleanup
	{
		pvc3_bgzf_inflater *&i; pvc3_bgzf_dev *&d; pvc3_records *&r; uint8_t *(&st)[IN_SLOTS];
		bool leave{};
		~cleanup() { if (leave) return; if (r) pvc3_records_destroy(r); if (d) pvc3_bgzf_dev_destroy(d); if (i) pvc3_bgzf_inflater_destroy(i); for (auto *p : st) if (p) pvc3_host_free(p); }
	} cleanup_guard{inflater, deflater, records, staging};
	auto rss_now = []() { double mb = 0; if (FILE *f = std::fopen("/proc/self/statm", "r")) { long pages = 0, resident = 0; if (2 == std::fscanf(f, "%ld %ld", &pages, &resident)) mb = (double) resident * (double) sysconf(_SC_PAGESIZE) / 1048576.0; std::fclose(f); } return mb; };
	bool const rss_trace = nullptr != std::getenv("PVC3_RECORDS_TRACE");
	if (rss_trace) std::fprintf(stderr, "record path: resident %.0f MB at its start\n", rss_now());
	if (opt.take_inflater) inflater = opt.take_inflater();
	else bgzf_check(pvc3_bgzf_inflater_create(opt.device, max_in, 1, max_blocks_in, IN_SLOTS, &inflater), "pvc3_bgzf_inflater_create");
	if (opt.take_deflater) deflater = opt.take_deflater();
	else bgzf_check(pvc3_bgzf_dev_create(opt.device, sz.max_blocks_out, OUT_SLOTS, &deflater), "pvc3_bgzf_dev_create");
	if (!inflater || !deflater) throw io_error("the BGZF codecs on the GPU could not be made");
	if (rss_trace) std::fprintf(stderr, "record path: resident %.0f MB with the codecs\n", rss_now());
	pvc3_records_config cfg = opt.config;
	cfg.max_stream_bytes = opt.piece_stream;
	rec_check(pvc3_records_create(opt.ctx, &cfg, &records), "pvc3_records_create");
	if (rss_trace) std::fprintf(stderr, "record path: resident %.0f MB with the records object\n", rss_now());
	uint64_t const out_capacity = pvc3_bgzf_dev_capacity(deflater);
	if (opt.inflate_in_flight) bgzf_check(pvc3_bgzf_inflater_limit_in_flight(inflater, opt.inflate_in_flight), "pvc3_bgzf_inflater_limit_in_flight");
	bool const shared = 0 == opt.inflate_mode && 0 != opt.gpu_blocks;		// a piece's blocks shared between the GPU and the host's threads
	uint32_t const piece_blocks = std::min <uint32_t>(max_blocks_in, opt.piece_blocks ? opt.piece_blocks : max_blocks_in);
	// the host's share of a piece at most (blocks of 64 KiB at most)
	std::size_t const staging_bytes = (2 == opt.inflate_mode) ? std::min <std::size_t>(opt.piece_stream, std::size_t(piece_blocks) * 65536)
		: (shared && piece_blocks > opt.gpu_blocks) ? std::min <std::size_t>(opt.piece_stream, std::size_t(piece_blocks - opt.gpu_blocks) * 65536) : 0;
	if (staging_bytes)
		for (auto &p : staging) { void *q{}; rec_check(pvc3_host_alloc(staging_bytes, &q), "pvc3_host_alloc"); p = (uint8_t *) q; }
	res.t_setup = since(t_begin);
	if (FILE *f = std::fopen("/proc/self/statm", "r")) { long pages = 0, resident = 0; if (2 == std::fscanf(f, "%ld %ld", &pages, &resident)) res.rss_setup_mb = (double) resident * (double) sysconf(_SC_PAGESIZE) / 1048576.0; std::fclose(f); }

	// ---- the reader ------------------------------------------------------------------------------------------------------------------
	struct piece
	{
		unsigned slot{};
		uint32_t n_blocks{};
		std::size_t in_bytes{}, out_bytes{};
		std::vector <uint64_t> block_off;		// [n_blocks + 1] where the blocks' bytes begin in the stream buffer, from its front
		std::vector <uint32_t> crc;
		bool last{};
		double t_submitted{};
		uint32_t n_gpu{};						// the first n_gpu blocks are the GPU's, the rest the host's threads'
		std::future <double> host_part;			// (its seconds)
	};
	std::mutex m;
	std::condition_variable cv;
	std::deque <piece> ready;
	bool slot_free[IN_SLOTS];
	for (auto &f : slot_free) f = true;
	bool reader_done = false, stop = false;
	std::exception_ptr reader_error;
	FILE *raw = std::fopen(opt.path, "rb");
	if (!raw) throw io_error(std::string("unable to open ") + opt.path);
	struct close_file { FILE *f; ~close_file() { if (f) std::fclose(f); } } raw_guard{raw};

	std::thread reader([&]() {
		try
		{
			std::vector <uint8_t> left;		// what the last piece did not use of its input (a partial block, or blocks beyond the piece's room)
			bool eof = false;
			for (unsigned k = 0; ; ++k)
			{
				unsigned const slot = k % IN_SLOTS;
				{
					std::unique_lock <std::mutex> lock(m);
					cv.wait(lock, [&]() { return stop || slot_free[slot]; });
					if (stop) return;
					slot_free[slot] = false;
				}
				auto const t0 = now();
				uint8_t *in = pvc3_bgzf_inflater_input(inflater, slot);
				pvc3_bgzf_block *blocks = pvc3_bgzf_inflater_blocks(inflater, slot);
				piece p;
				p.slot = slot;
				std::size_t have = left.size();
				if (have) std::memcpy(in, left.data(), have);
				left.clear();
				for (bool read_more = have < opt.piece_compressed; ; read_more = true)
				{
					if (read_more && !eof && have < max_in)
					{
						std::size_t const want = std::min(max_in - have, opt.piece_compressed);
						std::size_t const got = std::fread(in + have, 1, want, raw);
						if (got < want) { if (std::ferror(raw)) throw io_error("read error"); eof = true; }
						have += got;
					}
					// the whole blocks at hand, as many as the stream buffer has room for
					std::size_t off = 0, out = 0;
					p.n_blocks = 0; p.block_off.clear(); p.crc.clear();
					while (p.n_blocks < piece_blocks)
					{
						std::size_t data_off = 0;
						std::size_t const bs = byte_reader::bgzf_block_size(in + off, have - off, data_off);
						if (0 == bs || off + bs > have) break;
						if (bs < data_off + 8) throw io_error("corrupt BGZF stream (short block)");
						uint32_t crc, isize;
						std::memcpy(&crc, in + off + bs - 8, 4);
						std::memcpy(&isize, in + off + bs - 4, 4);
						if (isize > 65536) throw io_error("corrupt BGZF stream (ISIZE beyond 64 KiB)");
						if (out + isize > opt.piece_stream) break;
						if (isize)
						{
							blocks[p.n_blocks] = pvc3_bgzf_block{off + data_off, (uint32_t) (bs - data_off - 8), isize, out};
							p.block_off.push_back(out); p.crc.push_back(crc);
							++p.n_blocks;
						}
						out += isize; off += bs;
					}
					if (0 == off && have)
					{
						// not one whole block: more of the file, or the file is damaged
						if (eof) throw io_error("truncated BGZF stream");
						if (have >= max_in) throw io_error("a BGZF block larger than the reader's buffer");
						continue;
					}
					p.block_off.push_back(out);
					p.in_bytes = off; p.out_bytes = out;
					left.assign(in + off, in + have);
					p.last = eof && left.empty();
					break;
				}
				res.t_reader += since(t0);
				bool const last = p.last;
				{
					std::lock_guard <std::mutex> lock(m);
					ready.push_back(std::move(p));
					if (last) reader_done = true;
				}
				cv.notify_all();
				if (last) return;
			}
		}
		catch (...) { std::lock_guard <std::mutex> lock(m); reader_error = std::current_exception(); reader_done = true; cv.notify_all(); }
	});

	// ---- the writer ------------------------------------------------------------------------------------------------------------------
	std::deque <unsigned> to_write;
	bool out_free[OUT_SLOTS];
	for (auto &f : out_free) f = true;
	bool writer_done = false;
	std::exception_ptr writer_error;
	std::thread writer([&]() {
		// The bytes of one piece go to the file (on a thread of their own) while the next piece's come back from the GPU.
		std::future <void> writing;
		unsigned writing_slot = 0;
		bool writing_last = false;		// the write in flight is the last of its slot
		auto finish_write = [&]() {
			if (!writing.valid()) return;
			auto const t = now();
			writing.get();
			res.t_writer_file += since(t);
			if (writing_last)
			{
				{
					std::lock_guard <std::mutex> lock(m);
					out_free[writing_slot] = true;
				}
				cv.notify_all();
			}
		};
		try
		{
			for (;;)
			{
				unsigned slot;
				{
					std::unique_lock <std::mutex> lock(m);
					cv.wait(lock, [&]() { return writer_done || !to_write.empty(); });
					if (to_write.empty()) break;
					slot = to_write.front(); to_write.pop_front();
				}
				auto const t0 = now();
				uint64_t total = 0; float ms = 0;
				bgzf_check(pvc3_bgzf_dev_wait(deflater, slot, &total, &ms), "pvc3_bgzf_dev_wait");
				res.t_writer_wait += since(t0);
				for (uint64_t off = 0; off < total; )
				{
					if (writing.valid() && writing_slot == slot) finish_write();		// (a piece that comes back in several parts: they share the slot's page-locked memory)
					uint8_t const *bytes{}; uint64_t n = 0;
					auto const t1 = now();
					bgzf_check(pvc3_bgzf_dev_fetch(deflater, slot, off, &bytes, &n), "pvc3_bgzf_dev_fetch");
					res.t_writer_fetch += since(t1);
					finish_write();
					off += n;
					writing_slot = slot; writing_last = off >= total;
					writing = std::async(std::launch::async, [&opt, bytes, n]() { opt.write_blocks(bytes, n); });
				}
				if (0 == total)
				{
					finish_write();
					{
						std::lock_guard <std::mutex> lock(m);
						out_free[slot] = true;
					}
					cv.notify_all();
				}
				res.deflate_kernel_s += 1e-3 * ms; res.bytes_out += total;
				res.t_writer += since(t0);
			}
			finish_write();
		}
		catch (...)
		{
			if (writing.valid()) { try { writing.get(); } catch (...) {} }
			std::lock_guard <std::mutex> lock(m); writer_error = std::current_exception(); writer_done = true; cv.notify_all();
		}
	});
	struct join_threads
	{
		std::thread &a, &b; std::mutex &m; std::condition_variable &cv; bool &stop, &writer_done;
		~join_threads() { { std::lock_guard <std::mutex> l(m); stop = true; writer_done = true; } cv.notify_all(); if (a.joinable()) a.join(); if (b.joinable()) b.join(); }
	} joiner{reader, writer, m, cv, stop, writer_done};

	// ---- this thread: pieces through the GPU ------------------------------------------------------------------------------------------
	void *base[IN_SLOTS]; uint64_t front = 0;
	for (unsigned s = 0; s < IN_SLOTS; ++s) rec_check(pvc3_records_buffer(records, s, &base[s], &front), "pvc3_records_buffer");
	std::deque <piece> inflating;		// submitted to the inflater, oldest first
	bool first = true, seen_last = false;
	unsigned out_slot = 0;
	std::vector <uint8_t> scratch;
	std::vector <pvc3_run> runs;

	auto start_inflate = [&](piece &p) {
		if (0 == p.n_blocks) return;
		p.n_gpu = (2 == opt.inflate_mode) ? 0 : (shared ? std::min(p.n_blocks, opt.gpu_blocks) : p.n_blocks);
		pvc3_bgzf_block const *blocks = pvc3_bgzf_inflater_blocks(inflater, p.slot);
		if (p.n_gpu)
		{
			std::size_t const in_gpu = (p.n_gpu < p.n_blocks) ? (std::size_t) blocks[p.n_gpu].in_off : p.in_bytes;
			bgzf_check(pvc3_bgzf_inflater_submit_device(inflater, p.slot, p.n_gpu, in_gpu, (uint8_t *) base[p.slot] + front, opt.piece_stream), "pvc3_bgzf_inflater_submit_device");
		}
		if (p.n_gpu < p.n_blocks)
		{
			// the rest on the host's threads meanwhile, into page-locked memory: uploaded in one piece when this thread comes to the piece
			uint8_t const *in = pvc3_bgzf_inflater_input(inflater, p.slot);
			uint8_t *out = staging[p.slot];
			uint64_t const out0 = blocks[p.n_gpu].out_off;
			if (p.out_bytes - out0 > staging_bytes) throw io_error("the host's share of a piece exceeds its staging buffer");
			uint32_t const first_host = p.n_gpu, n_host = p.n_blocks - p.n_gpu;
			uint32_t const *crc = p.crc.data();		// (the vector's storage moves along with the piece)
			unsigned const threads = opt.threads;
			p.host_part = std::async(std::launch::async, [in, out, out0, blocks, first_host, n_host, crc, threads]() -> double {
				auto const t0 = std::chrono::steady_clock::now();
				parallel_for(n_host, threads, [&](std::size_t k) {
					static thread_local fast_inflater quick;
					pvc3_bgzf_block const &b = blocks[first_host + k];
					uint8_t *dst = out + (b.out_off - out0);
					uint32_t const want = crc[first_host + k];
					if (!(quick.inflate(in + b.in_off, b.in_len, dst, b.out_len) && fast_crc32(0, dst, b.out_len) == want))
					{
						detail::zlib_inflate_block(in + b.in_off, b.in_len, dst, b.out_len);
						if (fast_crc32(0, dst, b.out_len) != want) throw io_error("BGZF block checksum mismatch");
					}
				});
				return std::chrono::duration <double>(std::chrono::steady_clock::now() - t0).count();
			});
			res.blocks_host += n_host;
		}
	};
	auto finish_inflate = [&](piece &p) {
		if (0 == p.n_blocks) return;
		pvc3_bgzf_block const *blocks = pvc3_bgzf_inflater_blocks(inflater, p.slot);
		std::exception_ptr host_error;
		if (p.host_part.valid())
		{
			try { res.t_host_inflate += p.host_part.get(); } catch (...) { host_error = std::current_exception(); }
			if (!host_error)
			{
				uint64_t const out0 = blocks[p.n_gpu].out_off;
				rec_check(pvc3_records_upload(records, p.slot, out0, staging[p.slot], p.out_bytes - out0), "pvc3_records_upload");
			}
		}
		if (p.n_gpu)
		{
			uint32_t const *status{}, *crc{};
			float ms = 0;
			bgzf_check(pvc3_bgzf_inflater_wait_device(inflater, p.slot, &status, &crc, &ms), "pvc3_bgzf_inflater_wait_device");		// (also after an error on the host's side: the slot must not stay busy)
			if (host_error) std::rethrow_exception(host_error);
			res.inflate_kernel_s += 1e-3 * ms;
			// whatever the GPU refused, or whose checksum is not the file's: inflated here (zlib decides what is corrupt) and put in its place
			uint8_t const *in = pvc3_bgzf_inflater_input(inflater, p.slot);
			for (uint32_t i = 0; i < p.n_gpu; ++i)
			{
				if (0 == status[i] && crc[i] == p.crc[i]) continue;
				++res.blocks_refused;
				pvc3_bgzf_block const &b = blocks[i];
				scratch.resize(b.out_len);
				detail::zlib_inflate_block(in + b.in_off, b.in_len, scratch.data(), b.out_len);
				if (fast_crc32(0, scratch.data(), b.out_len) != p.crc[i]) throw io_error("BGZF block checksum mismatch");
				rec_check(pvc3_records_upload(records, p.slot, b.out_off, scratch.data(), b.out_len), "pvc3_records_upload");
			}
		}
		if (host_error) std::rethrow_exception(host_error);
	};

	for (;;)
	{
		// up to two pieces with the inflater
		auto t0 = now();
		for (;;)
		{
			std::unique_lock <std::mutex> lock(m);
			if (inflating.size() >= 2 || seen_last) break;
			if (ready.empty())
			{
				if (!inflating.empty() && !reader_done) break;		// nothing more at hand: work on what is there
				cv.wait(lock, [&]() { return reader_done || !ready.empty(); });
				if (reader_error) std::rethrow_exception(reader_error);
				if (ready.empty()) break;
			}
			piece p = std::move(ready.front()); ready.pop_front();
			lock.unlock();
			seen_last = p.last;
			p.t_submitted = since(t_begin);
			start_inflate(p);
			inflating.push_back(std::move(p));
		}
		res.t_wait_read += since(t0);
		if (reader_error) std::rethrow_exception(reader_error);
		if (writer_error) std::rethrow_exception(writer_error);
		if (inflating.empty()) break;
		piece p = std::move(inflating.front()); inflating.pop_front();
		t0 = now();
		std::array <double, 6> times{p.t_submitted, since(t_begin), 0, 0, 0, 0};
		finish_inflate(p);
		times[2] = since(t_begin);
		res.t_inflate_wait += since(t0); t0 = now();
		res.blocks_in += p.n_blocks; res.bytes_in += p.out_bytes;
		// the compressed bytes are no longer needed: the reader may fill the slot again (the stream buffer of the same number
		// stays this thread's until the piece's records are written)
		uint64_t skip = 0;
		if (first)
		{
			if (opt.header_bytes > p.out_bytes) throw io_error("the BAM header is longer than the first piece of the file");
			skip = opt.header_bytes; first = false;
		}
		pvc3_records_frame_result fr{};
		int code = 0; uint64_t bad_record = 0;
		int rc = pvc3_records_frame(records, p.slot, p.out_bytes, p.block_off.data(), p.n_blocks, skip, p.last ? 1 : 0, &fr, &code, &bad_record);
		if (rc && code) throw io_error(std::string("input record ") + std::to_string(bad_record) + ": " + pvc3_last_error());
		rec_check(rc, "pvc3_records_frame");
		res.matched += fr.matched; res.flags_not_matched += fr.flags_not_matched; res.ref_id_missing += fr.ref_id_missing; res.seq_missing += fr.seq_missing;
		res.t_frame += since(t0); t0 = now();
		times[3] = since(t_begin);
		{
			std::lock_guard <std::mutex> lock(m);
			slot_free[p.slot] = true;		// (its input; see above)
		}
		cv.notify_all();
		if (fr.n_batch)
		{
			pvc3_batch_stats stats{};
			rec_check(pvc3_records_project(records, &opt.params, &stats), "pvc3_records_project");
			res.t_project += since(t0); t0 = now();
			times[4] = since(t_begin);
			{
				std::unique_lock <std::mutex> lock(m);
				cv.wait(lock, [&]() { return writer_done || out_free[out_slot]; });
				if (writer_error) std::rethrow_exception(writer_error);
				out_free[out_slot] = false;
			}
			res.t_wait_writer += since(t0); t0 = now();
			uint64_t n_bytes = 0, n_runs = 0;
			rc = pvc3_records_rewrite(records, pvc3_bgzf_dev_input(deflater, out_slot), out_capacity, PVC3_BGZF_BLOCK, PVC3_BGZF_IN_STRIDE, PVC3_BGZF_IN_PAD, &n_bytes, &n_runs, &code, &bad_record);
			if (rc && code)
				throw io_error(std::string("input record ") + std::to_string(bad_record) + " cannot be projected or written (" + pvc3_last_error()
					+ "; the reference asserts here: position or CIGAR inconsistent with the index)");
			rec_check(rc, "pvc3_records_rewrite");
			res.n_runs += n_runs;
			if (opt.want_ranges && n_runs)
			{
				runs.resize(n_runs);
				rec_check(pvc3_records_runs(records, runs.data(), n_runs), "pvc3_records_runs");
				for (auto const &r : runs) res.ranges.emplace_back(r.ref_position, r.ref_length);
			}
			bgzf_check(pvc3_bgzf_dev_submit(deflater, out_slot, n_bytes), "pvc3_bgzf_dev_submit");
			{
				std::lock_guard <std::mutex> lock(m);
				to_write.push_back(out_slot);
			}
			cv.notify_all();
			out_slot = (out_slot + 1) % OUT_SLOTS;
			res.t_rewrite += since(t0);
			times[5] = since(t_begin);
			res.records += fr.n_batch; ++res.batches;
			if (opt.progress) opt.progress(res.records);
		}
		res.piece_times.push_back(times);
		if (p.last) break;
	}
	{
		std::lock_guard <std::mutex> lock(m);
		writer_done = true;
	}
	cv.notify_all();
	auto const t_drain = now();
	writer.join();
	if (writer_error) std::rethrow_exception(writer_error);
	reader.join();
	if (reader_error) std::rethrow_exception(reader_error);
	rec_check(pvc3_records_removed_tags(records, res.removed_tags.data()), "pvc3_records_removed_tags");
	res.t_drain = since(t_drain);
	res.t_total = since(t_begin);
	cleanup_guard.leave = opt.leave_objects;
	return res;
}

}
