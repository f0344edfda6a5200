// Host-side helpers of the C ABI (no device code).
//   raft_crc32c : CRC-32C (Castagnoli), the checksum of TensorFlow's tensor-bundle checkpoint format
//                 (tf_raft_amd/checkpoint.py reads the reference's `checkpoints/model.{index,data-*}`,
//                 reference README.md:66-96, train_sintel.py:104-107).
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.h"

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

// ------------------------------------------------------------------------------------------------
// tuning switches (raft_set_option / raft_get_option)
// ------------------------------------------------------------------------------------------------
namespace {
const char *const kOptNames[RAFT_OPT_COUNT] = {
    "RAFT_CONV_WINO", "RAFT_SMALL_WINO", "RAFT_GRU_WINO", "RAFT_GRU_WINO4", "RAFT_WINO_TNW", "RAFT_WINO_SB",
    "RAFT_WINO_CK", "RAFT_WINO1D_TM",
    "RAFT_LOOKUP_FUSED", "RAFT_ONDEMAND_BLOCK", "RAFT_ENC_WINO", "RAFT_LOOP_GRAPH",
    "RAFT_WINO_KS", "RAFT_CONV_WINO4", "RAFT_WINO4_KS", "RAFT_MASK_FUSED", "RAFT_ENC_WINO4", "RAFT_CONVC2_KS", "RAFT_CONVF2_KS",
    "RAFT_EVENT_FENCE", "RAFT_CORR_XCD", "RAFT_CORR_POOL",
};
constexpr int kTileEntries = 16;
struct TileRule {
    int npad, taps, code;
};
struct Options {
    std::atomic<int> val[RAFT_OPT_COUNT];
    std::atomic<bool> set[RAFT_OPT_COUNT];
    int env_val[RAFT_OPT_COUNT];     // the load-time (environment) state: raft_set_option(name, NULL) returns to it
    bool env_set[RAFT_OPT_COUNT];
    std::atomic<int> generation{0};
    std::mutex tile_mu;
    char tile_text[256], tile_env[256];
    TileRule tile_rules[kTileEntries];
    int n_tile_rules, tile_all;      // tile_all: one code for every convolution it is valid for, -1 = none
    Options() {
        for (int i = 0; i < RAFT_OPT_COUNT; ++i) {
            const char *e = getenv(kOptNames[i]);
            env_set[i] = e && *e;
            env_val[i] = env_set[i] ? atoi(e) : 0;
            val[i].store(env_val[i]);
            set[i].store(env_set[i]);
        }
        const char *e = getenv("RAFT_CONV_TILE");
        snprintf(tile_env, sizeof(tile_env), "%s", e ? e : "");
        parse_tiles(tile_env);
    }
    void parse_tiles(const char *text) {   // caller holds tile_mu (or is the constructor)
        snprintf(tile_text, sizeof(tile_text), "%s", text ? text : "");
        n_tile_rules = 0;
        tile_all = -1;
        if (!tile_text[0]) return;
        if (!strchr(tile_text, ':')) {
            tile_all = atoi(tile_text);
            return;
        }
        const char *p = tile_text;
        while (*p && n_tile_rules < kTileEntries) {
            TileRule r;
            if (sscanf(p, "%d:%d:%d", &r.npad, &r.taps, &r.code) == 3) tile_rules[n_tile_rules++] = r;
            p = strchr(p, ',');
            if (!p) break;
            ++p;
        }
    }
};
Options &opts() {
    static Options o;   // constructed at first use (library load: the static initialiser below touches it)
    return o;
}
const int kOptsLoaded = (opts(), 0);
int opt_index(const char *name) {
    for (int i = 0; i < RAFT_OPT_COUNT; ++i)
        if (strcmp(name, kOptNames[i]) == 0) return i;
    return -1;
}
}   // namespace

namespace {
thread_local int tl_concurrency = 1;
}
int raft_concurrency() { return tl_concurrency; }
extern "C" int raft_set_thread_concurrency(int n) {
    const int prev = tl_concurrency;
    tl_concurrency = n < 1 ? 1 : (n > 64 ? 64 : n);
    return prev;
}

int raft_opt(int id, int dflt) {
    Options &o = opts();
    return o.set[id].load(std::memory_order_relaxed) ? o.val[id].load(std::memory_order_relaxed) : dflt;
}
bool raft_opt_is_set(int id) { return opts().set[id].load(std::memory_order_relaxed); }
int raft_opt_generation() { return opts().generation.load(std::memory_order_relaxed); }

int raft_opt_conv_tile(int npad, int taps, bool (*valid)(int, int)) {
    Options &o = opts();
    std::lock_guard<std::mutex> g(o.tile_mu);
    if (o.tile_all >= 0) return valid(o.tile_all, npad) ? o.tile_all : -1;
    for (int i = 0; i < o.n_tile_rules; ++i) {
        const TileRule &r = o.tile_rules[i];
        if (r.npad == npad && r.taps == taps && valid(r.code, npad)) return r.code;
    }
    return -1;
}

extern "C" int raft_set_option(const char *name, const char *value) {
    RAFT_REQUIRE_PTR(name);
    Options &o = opts();
    if (strcmp(name, "RAFT_CONV_TILE") == 0) {
        std::lock_guard<std::mutex> g(o.tile_mu);
        o.parse_tiles(value ? value : o.tile_env);
        o.generation.fetch_add(1);
        return RAFT_OK;
    }
    const int i = opt_index(name);
    if (i < 0) return RAFT_E_UNSUPPORTED;
    o.generation.fetch_add(1);
    if (value == nullptr) {   // back to the load-time state
        o.val[i].store(o.env_val[i]);
        o.set[i].store(o.env_set[i]);
    } else if (*value == 0) {   // "" = unset: the built-in default
        o.set[i].store(false);
    } else {
        o.val[i].store(atoi(value));
        o.set[i].store(true);
    }
    return RAFT_OK;
}

extern "C" int raft_get_option(const char *name, char *buf, size_t len) {
    RAFT_REQUIRE_PTR(name);
    RAFT_REQUIRE_PTR(buf);
    RAFT_REQUIRE(len > 0, RAFT_E_SHAPE);
    Options &o = opts();
    if (strcmp(name, "RAFT_CONV_TILE") == 0) {
        std::lock_guard<std::mutex> g(o.tile_mu);
        snprintf(buf, len, "%s", o.tile_text);
        return RAFT_OK;
    }
    const int i = opt_index(name);
    if (i < 0) return RAFT_E_UNSUPPORTED;
    if (o.set[i].load())
        snprintf(buf, len, "%d", o.val[i].load());
    else
        buf[0] = 0;
    return RAFT_OK;
}
