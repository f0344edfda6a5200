// Host-side helpers of the C ABI (no device code).
//   raft_crc32c : CRC-32C (Castagnoli), the checksum of TensorFlow's tensor-bundle checkpoint format
//                 (tf_raft_amd/checkpoint.py reads the reference's `checkpoints/model.{index,data-*}`,
//                 reference README.md:66-96, train_sintel.py:104-107).
#include <stddef.h>
#include <stdint.h>

#include "../../include/raft_hip.h"

namespace {
struct Crc32cTable {
    uint32_t t[8][256];
    Crc32cTable() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82f63b78u & (0u - (c & 1u)));   // reflected 0x1EDC6F41
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xffu];
    }
};
}   // namespace

extern "C" uint32_t raft_crc32c(uint32_t crc, const void *data, size_t n) {
    static const Crc32cTable tab;
    const unsigned char *p = (const unsigned char *)data;
    uint32_t c = ~crc;
    while (n && ((uintptr_t)p & 7u)) {
        c = (c >> 8) ^ tab.t[0][(c ^ *p++) & 0xffu];
        --n;
    }
    while (n >= 8) {   // slicing-by-8
        uint64_t w;
        __builtin_memcpy(&w, p, 8);
        w ^= c;
        c = tab.t[7][w & 0xff] ^ tab.t[6][(w >> 8) & 0xff] ^ tab.t[5][(w >> 16) & 0xff] ^ tab.t[4][(w >> 24) & 0xff] ^
            tab.t[3][(w >> 32) & 0xff] ^ tab.t[2][(w >> 40) & 0xff] ^ tab.t[1][(w >> 48) & 0xff] ^ tab.t[0][w >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ tab.t[0][(c ^ *p++) & 0xffu];
    return ~c;
}
