// C entry points over the reference's REAL fastpforlib (third_party/fastpforlib/bitpackinghelpers.h), compiled
// together with its bitpacking.cpp into oracle/_ref/libfastpfor_ref.so.  Test infrastructure only; contains no
// reference code — it only calls it.
#include <cstdint>

#include "bitpackinghelpers.h"

extern "C" {
__attribute__((visibility("default"))) void ref_pack8(const uint8_t *in, uint8_t *out, uint32_t w) {
	duckdb_fastpforlib::fastpack(in, out, w);
}
__attribute__((visibility("default"))) void ref_pack16(const uint16_t *in, uint16_t *out, uint32_t w) {
	duckdb_fastpforlib::fastpack(in, out, w);
}
__attribute__((visibility("default"))) void ref_pack32(const uint32_t *in, uint32_t *out, uint32_t w) {
	duckdb_fastpforlib::fastpack(in, out, w);
}
__attribute__((visibility("default"))) void ref_pack64(const uint64_t *in, uint32_t *out, uint32_t w) {
	duckdb_fastpforlib::fastpack(in, out, w);
}
__attribute__((visibility("default"))) void ref_unpack8(const uint8_t *in, uint8_t *out, uint32_t w) {
	duckdb_fastpforlib::fastunpack(in, out, w);
}
__attribute__((visibility("default"))) void ref_unpack16(const uint16_t *in, uint16_t *out, uint32_t w) {
	duckdb_fastpforlib::fastunpack(in, out, w);
}
__attribute__((visibility("default"))) void ref_unpack32(const uint32_t *in, uint32_t *out, uint32_t w) {
	duckdb_fastpforlib::fastunpack(in, out, w);
}
__attribute__((visibility("default"))) void ref_unpack64(const uint32_t *in, uint64_t *out, uint32_t w) {
	duckdb_fastpforlib::fastunpack(in, out, w);
}
}
