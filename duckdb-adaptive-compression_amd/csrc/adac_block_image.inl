// adac_block_image.inl — persistent block images of packed segments, built and parsed in HBM.
// Included into adac_kernels.hip inside namespace adac::{anonymous}.
//
// The reference never persists a SUCCINCT segment (ColumnSegment::ConvertToPersistent returns early,
// src/storage/table/column_segment.cpp:529-533; the checkpoint-side slots print "SHOULD NOT HAPPEN",
// src/storage/compression/succinct.cpp:91-119).  The image defined in include/adacodec.h is the
// sdsl::int_vector<0> serialisation (uint64 bit_size, uint8 width, ceil(bit_size/64) words:
// third_party/sdsl/include/sdsl/int_vector.hpp:602-609,1565-1578) followed by a 16-byte trailer.  Because of the
// 9-byte header the words sit at byte offset 9 of the image: as a stream of aligned 8-byte units the image is
//     out[0] = bit_size,   out[k] = (X[k-2] >> 56) | (X[k-1] << 8)   for 1 <= k < W + 4
// with X[-1] = width << 56, X[0..W) = the packed words, X[W] = min, X[W+1] = flags | type << 8, 0 elsewhere —
// one funnel shift per unit, coalesced 8-byte loads and stores, no byte-granular access.  Parsing is the inverse:
//     word[j] = (in[j+1] >> 8) | (in[j+2] << 56).
// A whole checkpoint (thousands of segments) is then ONE kernel + ONE device-to-host copy instead of a copy per
// segment.

constexpr uint32_t kImageChunk = 2048; // 8-byte units per workgroup: 16 KiB of image

__device__ __forceinline__ uint64_t image_source(const BlockJob &j, const uint64_t *__restrict__ w, int64_t i) {
	if (i >= 0 && i < (int64_t)j.nwords) return w[i];
	if (i == -1) return (uint64_t)j.width << 56;
	if (i == (int64_t)j.nwords) return j.min;
	if (i == (int64_t)j.nwords + 1) return (uint64_t)j.flags | ((uint64_t)j.type << 8);
	return 0ull;
}

__global__ __launch_bounds__(kWorkgroup) void k_blocks_write(const BlockJob *__restrict__ jobs, uint32_t chunks_per_seg,
                                                            const uint64_t *__restrict__ words,
                                                            uint64_t *__restrict__ blocks) {
	const BlockJob j = jobs[blockIdx.x / chunks_per_seg];
	const uint32_t k0 = (blockIdx.x % chunks_per_seg) * kImageChunk;
	const uint32_t nout = j.nwords + 4u; // 8 * (W + 4) >= 9 + 8 W + 16 bytes: the image, zero-padded to 8 bytes
	if (k0 >= nout) return;
	const uint32_t k1 = k0 + kImageChunk < nout ? k0 + kImageChunk : nout;
	const uint64_t *__restrict__ w = words + j.word_off;
	uint64_t *__restrict__ out = blocks + (j.block_off >> 3);
	for (uint32_t k = k0 + threadIdx.x; k < k1; k += kWorkgroup) {
		out[k] = k == 0 ? j.bit_size
		                : (image_source(j, w, (int64_t)k - 2) >> 56) | (image_source(j, w, (int64_t)k - 1) << 8);
	}
}

// Parse: copies the words of every image into the arena (zeroing the allocation's tail words, as SDSL's resize does,
// memory_management.hpp:346-368) and checks the header against the descriptor the host derived from it; a mismatch
// (the device buffer does not hold the bytes the host peeked) is counted in *d_bad.
__global__ __launch_bounds__(kWorkgroup) void k_blocks_read(const BlockJob *__restrict__ jobs, uint32_t chunks_per_seg,
                                                           const uint64_t *__restrict__ blocks,
                                                           uint64_t *__restrict__ words, uint32_t *__restrict__ d_bad) {
	const BlockJob j = jobs[blockIdx.x / chunks_per_seg];
	const uint32_t chunk = blockIdx.x % chunks_per_seg;
	const uint64_t *__restrict__ in = blocks + (j.block_off >> 3);
	if (chunk == 0 && threadIdx.x == 0) {
		const uint64_t tail = (in[j.nwords + 1] >> 8) | (in[j.nwords + 2] << 56); // min
		const uint64_t meta = (in[j.nwords + 2] >> 8) | (in[j.nwords + 3] << 56); // flags, type
		if (in[0] != j.bit_size || (uint8_t)in[1] != j.width || tail != j.min || (uint8_t)meta != j.flags ||
		    (uint8_t)(meta >> 8) != j.type) {
			atomicAdd(d_bad, 1u);
		}
	}
	const uint32_t k0 = chunk * kImageChunk;
	if (k0 >= j.arena_words) return;
	const uint32_t k1 = k0 + kImageChunk < j.arena_words ? k0 + kImageChunk : j.arena_words;
	uint64_t *__restrict__ w = words + j.word_off;
	for (uint32_t k = k0 + threadIdx.x; k < k1; k += kWorkgroup) {
		w[k] = k < j.nwords ? (in[k + 1] >> 8) | (in[k + 2] << 56) : 0ull;
	}
}
