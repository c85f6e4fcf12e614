/*
 * pvc3_bgzf.h — C ABI of the BGZF block compressor on the GPU (libpvc3_bgzf.so).
 *
 * The BAM writer of panvc3_project_alignments spends most of the process's CPU time in DEFLATE.  BGZF blocks
 * are independent gzip members of at most 65 280 uncompressed bytes (SAMv1 §4.1), so a group of them is compressed by one
 * kernel launch, one wavefront per block.  What comes back is the raw DEFLATE stream of each block (RFC 1951, one final
 * dynamic-Huffman or stored block); the caller adds the 18-byte gzip/BC header, CRC32 and ISIZE.
 *
 * Reference interface replaced: none of panvc3's own — the reference writes BAM through seqan3's bgzf_ostream
 * (project-alignments/project_alignments.cc:1296-1330 opens the output; compression happens inside seqan3 / zlib).  Any inflate
 * reads the streams; they are not byte-identical to zlib's (no two DEFLATE encoders are).
 *
 * Staging is page-locked memory owned by the library, `n_slots` of it, so that one group is filled while another is
 * compressed:
 *   in    pvc3_bgzf_slot_input(h, s):  block i's bytes at  i * PVC3_BGZF_IN_STRIDE + PVC3_BGZF_IN_PAD
 *         pvc3_bgzf_slot_lengths(h, s)[i] = their number (0 .. PVC3_BGZF_BLOCK)
 *   out   after pvc3_bgzf_wait: block i's stream at out + i * PVC3_BGZF_OUT_STRIDE, out_len[i] bytes
 * One host thread per handle.  Functions return 0 or a negative code (the PVC3_E_* values of pvc3.h);
 * pvc3_bgzf_last_error() gives the message of the calling thread's last failure.
 */
#ifndef PVC3_BGZF_H
#define PVC3_BGZF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVC3_BGZF_API __attribute__((visibility("default")))

#define PVC3_BGZF_BLOCK 65280u			/* uncompressed bytes per block at most (0xff00, as htslib) */
#define PVC3_BGZF_IN_PAD 64u			/* unused bytes in front of a block's data in its slot */
#define PVC3_BGZF_IN_STRIDE 65536u
#define PVC3_BGZF_OUT_STRIDE 65536u

typedef struct pvc3_bgzf pvc3_bgzf;

/* device: HIP device ordinal; max_blocks: blocks per group at most; n_slots: 1 .. 4 staging areas. */
PVC3_BGZF_API int pvc3_bgzf_create(int device, uint32_t max_blocks, uint32_t n_slots, pvc3_bgzf **out);
PVC3_BGZF_API void pvc3_bgzf_destroy(pvc3_bgzf *h);
PVC3_BGZF_API uint32_t pvc3_bgzf_max_blocks(pvc3_bgzf const *h);
PVC3_BGZF_API uint8_t *pvc3_bgzf_slot_input(pvc3_bgzf *h, uint32_t slot);
PVC3_BGZF_API uint32_t *pvc3_bgzf_slot_lengths(pvc3_bgzf *h, uint32_t slot);

/* Starts copying the slot's first n_blocks blocks to the GPU, compressing them and copying the streams back; returns at once. */
PVC3_BGZF_API int pvc3_bgzf_submit(pvc3_bgzf *h, uint32_t slot, uint32_t n_blocks);
/* Waits for the slot's group.  *out / *out_len stay valid until the slot is submitted again.  kernel_ms (may be null): the
 * kernel's time on the GPU. */
PVC3_BGZF_API int pvc3_bgzf_wait(pvc3_bgzf *h, uint32_t slot, uint8_t const **out, uint32_t const **out_len, float *kernel_ms);

/* One call for callers without staging of their own (tests, small outputs): src[0 .. n) is cut into pieces of `piece` bytes
 * (1 .. PVC3_BGZF_BLOCK); piece i's stream goes to dst + i * PVC3_BGZF_OUT_STRIDE and its size to dst_len[i].  dst must hold
 * ceil(n / piece) * PVC3_BGZF_OUT_STRIDE bytes. */
PVC3_BGZF_API int pvc3_bgzf_deflate(pvc3_bgzf *h, uint8_t const *src, size_t n, uint32_t piece, uint8_t *dst, uint32_t *dst_len);

/* ---- the other direction: the blocks of a BGZF file inflated, one wave per block (panvc3_amd/bgzf/inflate.hip) --------------------
 * Replaces, like the compressor, nothing of panvc3's own: the reference reads BAM through seqan3 / zlib.  The caller reads a
 * piece of the file into the slot's page-locked input, lists its blocks (where the DEFLATE data of each begin and how long they
 * are, ISIZE, and where its bytes are to go in the slot's output: usually one behind the other), submits, and finds the inflated
 * bytes in the slot's page-locked output.  status[i] != 0: the kernel did not accept block i (its bytes in the output are
 * undefined) and the caller inflates it itself; the caller also verifies the CRC-32 of every block. */
typedef struct pvc3_bgzf_inflater pvc3_bgzf_inflater;
typedef struct pvc3_bgzf_block
{
	uint64_t in_off;		/* DEFLATE data of the block in the slot's input */
	uint32_t in_len;
	uint32_t out_len;		/* ISIZE, at most 65536 */
	uint64_t out_off;		/* where its bytes go in the slot's output */
} pvc3_bgzf_block;

PVC3_BGZF_API int pvc3_bgzf_inflater_create(int device, size_t max_in_bytes, size_t max_out_bytes, uint32_t max_blocks, uint32_t n_slots, pvc3_bgzf_inflater **out);
PVC3_BGZF_API void pvc3_bgzf_inflater_destroy(pvc3_bgzf_inflater *h);
/* How many blocks the GPU works on at the same time (one wave each): a launch of that many takes as long as a launch of one. */
PVC3_BGZF_API uint32_t pvc3_bgzf_inflater_blocks_in_flight(pvc3_bgzf_inflater const *h);
/* Fewer blocks at a time (0: as many as the chip holds again), for a caller whose other kernels run next to the inflater's. */
PVC3_BGZF_API int pvc3_bgzf_inflater_limit_in_flight(pvc3_bgzf_inflater *h, uint32_t blocks);
PVC3_BGZF_API uint8_t *pvc3_bgzf_inflater_input(pvc3_bgzf_inflater *h, uint32_t slot);
PVC3_BGZF_API uint8_t *pvc3_bgzf_inflater_output(pvc3_bgzf_inflater *h, uint32_t slot);
PVC3_BGZF_API pvc3_bgzf_block *pvc3_bgzf_inflater_blocks(pvc3_bgzf_inflater *h, uint32_t slot);
/* in_bytes / out_bytes: how much of the slot's input is uploaded and how much of its output comes back. */
PVC3_BGZF_API int pvc3_bgzf_inflater_submit(pvc3_bgzf_inflater *h, uint32_t slot, uint32_t n_blocks, size_t in_bytes, size_t out_bytes);
PVC3_BGZF_API int pvc3_bgzf_inflater_wait(pvc3_bgzf_inflater *h, uint32_t slot, uint32_t const **status, float *kernel_ms);

/* The device-resident form of the inflater, for a caller whose kernels read the inflated bytes (the record path,
 * include/pvc3_records.h): the slot's input is uploaded as before, but the blocks' bytes go to DEVICE memory of the caller — block
 * i's to d_out + blocks[i].out_off — and stay there.  What comes back is every block's status and the CRC-32 of its bytes,
 * computed on the device, for the caller to compare with the checksum in the file.  (max_out_bytes of the create call may be 1
 * for an inflater that is only used this way.)  When pvc3_bgzf_inflater_wait_device has returned, the bytes are in d_out. */
PVC3_BGZF_API int pvc3_bgzf_inflater_submit_device(pvc3_bgzf_inflater *h, uint32_t slot, uint32_t n_blocks, size_t in_bytes, void *d_out, size_t d_out_bytes);
PVC3_BGZF_API int pvc3_bgzf_inflater_wait_device(pvc3_bgzf_inflater *h, uint32_t slot, uint32_t const **status, uint32_t const **crc, float *kernel_ms);

/* The device-resident form of the compressor: the uncompressed bytes are written by kernels of the caller straight into the
 * compressor's device input (byte o of the stream at (o / PVC3_BGZF_BLOCK) * PVC3_BGZF_IN_STRIDE + PVC3_BGZF_IN_PAD + o %
 * PVC3_BGZF_BLOCK of pvc3_bgzf_dev_input(h, slot)), and what comes back is the bytes of the BGZF FILE: every block with its gzip /
 * BC header, stream, CRC-32 (computed on the device) and ISIZE, one behind the other.
 * pvc3_bgzf_dev_submit may only be called when the kernels that wrote the slot's input have finished. */
typedef struct pvc3_bgzf_dev pvc3_bgzf_dev;
PVC3_BGZF_API int pvc3_bgzf_dev_create(int device, uint32_t max_blocks, uint32_t n_slots, pvc3_bgzf_dev **out);
PVC3_BGZF_API void pvc3_bgzf_dev_destroy(pvc3_bgzf_dev *h);
PVC3_BGZF_API void *pvc3_bgzf_dev_input(pvc3_bgzf_dev *h, uint32_t slot);			/* device pointer */
PVC3_BGZF_API uint64_t pvc3_bgzf_dev_capacity(pvc3_bgzf_dev const *h);			/* uncompressed bytes per slot */
PVC3_BGZF_API int pvc3_bgzf_dev_submit(pvc3_bgzf_dev *h, uint32_t slot, uint64_t n_bytes);
/* Waits for the slot's blocks: *n_file_bytes of BGZF file.  They are fetched through page-locked memory in one or more pieces:
 * pvc3_bgzf_dev_fetch copies the bytes from `offset` on, as many as the staging holds (BAM records fit one fetch), and returns
 * where they are and how many; the bytes stay valid until the next fetch or submit of the slot. */
PVC3_BGZF_API int pvc3_bgzf_dev_wait(pvc3_bgzf_dev *h, uint32_t slot, uint64_t *n_file_bytes, float *kernel_ms);
PVC3_BGZF_API int pvc3_bgzf_dev_fetch(pvc3_bgzf_dev *h, uint32_t slot, uint64_t offset, uint8_t const **bytes, uint64_t *n_bytes);

PVC3_BGZF_API char const *pvc3_bgzf_last_error(void);

#ifdef __cplusplus
}
#endif

#endif
