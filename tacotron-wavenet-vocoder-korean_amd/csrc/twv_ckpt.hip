// Host-side helper of the checkpoint reader/writer (checkpoint.py): CRC-32C (Castagnoli, reflected polynomial
// 0x82F63B78), the checksum TensorFlow's tensor bundles carry per tensor and per table block.  Slicing-by-8 tables,
// no device code.  Known answers (RFC 3720 B.4) are checked in tests/test_cpu.py.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xffu];
    }
};
const Crc32cTables g_crc;
}  // namespace

// crc = running value (0 for a fresh checksum); returns the CRC-32C of the bytes seen so far (final xor applied).
extern "C" uint32_t twv_crc32c(const void* data, size_t n, uint32_t crc) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = ~crc;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7u)) { c = g_crc.t[0][(c ^ *p++) & 0xffu] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;
        c = g_crc.t[7][w & 0xff] ^ g_crc.t[6][(w >> 8) & 0xff] ^ g_crc.t[5][(w >> 16) & 0xff] ^ g_crc.t[4][(w >> 24) & 0xff] ^
            g_crc.t[3][(w >> 32) & 0xff] ^ g_crc.t[2][(w >> 40) & 0xff] ^ g_crc.t[1][(w >> 48) & 0xff] ^ g_crc.t[0][(w >> 56) & 0xff];
        p += 8; n -= 8;
    }
    while (n--) c = g_crc.t[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
    return ~c;
}

// the build stamps the binary with the hash of ALL its sources (_lib.source_hash): tests assert that the library they loaded IS the
// tree.  This small host-only file is the only one compiled with the stamp (-DTWV_SRC_HASH), so the kernel files' objects can be
// cached per file (_lib.build).
#ifndef TWV_SRC_HASH
#define TWV_SRC_HASH "unstamped"
#endif
extern "C" const char* twv_version(void) { return "twv_amd 0.2 (gfx950) src:" TWV_SRC_HASH; }
