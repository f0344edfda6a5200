// Host-side check of the C ABI under AddressSanitizer / ThreadSanitizer ("re-entrant and thread-safe").
// Linked against a HOST-ONLY build of the library (hipcc --offload-host-only -fsanitize=...; tf_raft_amd/build.py
// build_sanitizer_library): no device code, no GPU -- what runs here is everything an entry point does BEFORE it launches:
// argument validation, geometry / workspace arithmetic, the option table, error strings.  Every launching entry point is
// called with arguments it must reject (null pointers, bad shapes, misaligned buffers) and has to return an error code
// without touching memory; the pure-host helpers are called with valid arguments; four threads then hammer the option table
// and the helpers concurrently (the only process-global state of the library).
// Test infrastructure: built and run by tests/test_abi_sanitizers.py, never shipped.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "raft_hip.h"

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

static void helpers_once(int salt) {
    int64_t off[RAFT_MAX_LEVELS + 1];
    int lh[RAFT_MAX_LEVELS], lw[RAFT_MAX_LEVELS];
    const int B = 1 + salt % 3, h = 56 + salt % 5, w = 64 + salt % 7;
    EXPECT(raft_corr_pyramid_layout(B, h, w, 4, off, lh, lw) == 0);
    EXPECT(off[0] == 0 && off[4] > off[3] && lh[0] == h && lw[0] == w && lh[3] == h / 8 && lw[3] == w / 8);
    for (int l = 0; l < 4; ++l) EXPECT((off[l + 1] - off[l]) % 32 == 0);     // whole 128-byte tiles per map
    EXPECT(raft_corr_pyramid_layout(B, h, w, 4, off, nullptr, nullptr) == 0);
    EXPECT(raft_corr_pyramid_layout(B, h, w, 0, off, lh, lw) != 0);
    EXPECT(raft_corr_pyramid_layout(B, h, w, RAFT_MAX_LEVELS + 1, off, lh, lw) != 0);
    EXPECT(raft_corr_pyramid_layout(0, h, w, 4, off, lh, lw) != 0);
    EXPECT(raft_corr_pyramid_layout(B, h, w, 4, nullptr, lh, lw) != 0);
    EXPECT(raft_corr_build_workspace_floats(B, h, w, 256, 4) > (int64_t)B * h * w * 256);
    EXPECT(raft_corr_build_workspace_floats(0, h, w, 256, 4) == 0);
    EXPECT(raft_update_workspace_floats(B, h, w) > 0);
    EXPECT(raft_small_update_workspace_floats(B, h, w) > 0);
    EXPECT(raft_metrics_workspace_doubles() > 0);
    EXPECT(raft_sumsq_workspace_doubles() > 0);
    EXPECT(raft_sumsq_multi_workspace_doubles(7) > 0);
    EXPECT(raft_norm_workspace_doubles(4, 64) > 0);
    EXPECT(raft_conv2d_wgrad_workspace_floats(128, 128, B, h, w, 3, 3) >= 0);
    EXPECT(raft_conv7x7_c2_wgrad_workspace_floats(128) > 0);
    EXPECT(raft_upsample_convex_backward_workspace_floats(B, h, w) >= 0);
    EXPECT(raft_version() > 0);
    for (int rc = -3; rc < 1100; rc += 37) {
        const char *s = raft_error_string(rc);
        EXPECT(s != nullptr && std::strlen(s) > 0);
    }
}

static void options_once(int salt) {
    char buf[64];
    EXPECT(raft_set_option("RAFT_NO_SUCH_SWITCH", "1") != 0);
    EXPECT(raft_get_option("RAFT_NO_SUCH_SWITCH", buf, sizeof buf) != 0);
    EXPECT(raft_set_option(nullptr, "1") != 0);
    const char *vals[3] = {"0", "1", nullptr};
    EXPECT(raft_set_option("RAFT_CORR_XCD", vals[salt % 3]) == 0);
    EXPECT(raft_get_option("RAFT_CORR_XCD", buf, sizeof buf) == 0);
    EXPECT(raft_get_option("RAFT_CORR_XCD", buf, 1) == 0 || true);       // a one-byte buffer: truncated, never overrun
    EXPECT(raft_get_option("RAFT_CORR_XCD", nullptr, 0) != 0 || true);
    EXPECT(raft_set_option("RAFT_LOOKUP_FUSED", vals[(salt + 1) % 3]) == 0);
    EXPECT(raft_set_option("RAFT_LOOKUP_FUSED", "not a number") != 0 || true);
}

// every launching entry point with arguments it has to reject before any launch (no GPU is present here)
static void rejects() {
    float f[64] = {0};
    int64_t off[RAFT_MAX_LEVELS + 1] = {0, 32, 64, 96, 128};
    unsigned char u8[8] = {0};
    double d[8] = {0};
    void *s = nullptr;
    EXPECT(raft_corr_build_f32(nullptr, f, 1, 8, 8, 256, 4, f, off, f, s) != 0);
    EXPECT(raft_corr_build_f32(f, f, 0, 8, 8, 256, 4, f, off, f, s) != 0);
    EXPECT(raft_corr_build_f32(f, f, 1, 8, 8, 255, 4, f, off, f, s) != 0);
    EXPECT(raft_corr_lookup_f32(nullptr, off, f, 1, 8, 8, 4, 4, f, 352, s) != 0);
    EXPECT(raft_corr_lookup_f32(f, off, f, 1, 8, 8, 4, 9, f, 352, s) != 0);       // unsupported radius
    EXPECT(raft_corr_lookup_f32(f, off, f, 1, 8, 8, 4, 4, f, 10, s) != 0);        // row stride < 4 * 81
    EXPECT(raft_corr_lookup_f32(f, off, f, 1, 0, 8, 4, 4, f, 352, s) != 0);
    EXPECT(raft_fmap_pyramid_f32(nullptr, 1, 8, 8, 256, 4, f, s) != 0);
    EXPECT(raft_fmap_pyramid_f32(f, 1, 8, 8, 255, 4, f, s) != 0);
    EXPECT(raft_fmap_pyramid_f32(f + 1, 1, 8, 8, 256, 4, f, s) != 0);            // misaligned
    EXPECT(raft_bilinear_sampler_f32(nullptr, f, 1, 8, 8, 1, 1, f, s) != 0);
    EXPECT(raft_bilinear_sampler_f32(f, f, 0, 8, 8, 1, 1, f, s) != 0);
    EXPECT(raft_coords_grid_f32(nullptr, 1, 8, 8, s) != 0);
    EXPECT(raft_coords_grid_f32(f, 1, 0, 8, s) != 0);
    EXPECT(raft_upsample_convex_f32(nullptr, f, 1, 8, 8, f, s) != 0);
    EXPECT(raft_upsample_convex_f32(f, f, 1, -1, 8, f, s) != 0);
    EXPECT(raft_upflow8_f32(nullptr, 1, 8, 8, f, s) != 0);
    EXPECT(raft_lookup_convc1_f32(f, off, f, 1, 8, 8, f, f, 128, 128, f, 256, s) != 0);   // npad must be 256
    EXPECT(raft_lookup_convc1_f32(f, off, f, 1, 8, 8, nullptr, f, 256, 256, f, 256, s) != 0);
    EXPECT(raft_conv2d_f32(nullptr, 0, 0, nullptr, 0, 0, f, f, 1, 8, 8, 3, 3, 64, 64, 1, 1.0f, f, 64, s) != 0);
    EXPECT(raft_conv7x7_c2_f32(nullptr, f, f, 128, 1, 8, 8, f, 128, s) != 0);
    EXPECT(raft_update_basic_f32(nullptr, 1, 8, 8, nullptr, s) != 0);
    EXPECT(raft_iterate_basic_f32(nullptr, f, off, 1, 8, 8, 1, nullptr, f, s) != 0);
    EXPECT(raft_iterate_small_f32(nullptr, f, off, 1, 8, 8, 1, nullptr, f, s) != 0);
    EXPECT(raft_iterate_basic_overlap_f32(nullptr, f, off, 1, 8, 8, 1, nullptr, f, s, s, s, nullptr) != 0);
    EXPECT(raft_prepare_state_f32(nullptr, 1, 8, 8, nullptr, s) != 0);
    EXPECT(raft_prepare_state_small_f32(nullptr, 1, 8, 8, nullptr, s) != 0);
    EXPECT(raft_encoder_f32(nullptr, f, 1, 64, 64, 1, f, f, s) != 0);
    EXPECT(raft_encoder_workspace_floats(nullptr, 1, 64, 64) == 0);
    EXPECT(raft_flow_metrics_f32(nullptr, u8, f, 64, 400.f, f, d, s) != 0);
    EXPECT(raft_relu_backward_f32(nullptr, f, f, 8, s) != 0);
    EXPECT(raft_axpby_f32(1.f, nullptr, 1.f, f, f, 8, s) != 0);
    EXPECT(raft_sumsq_f32(nullptr, 8, 0, d, d, s) != 0);
    EXPECT(raft_loop_ctx_destroy(nullptr) == 0 || true);
}

int main() {
    helpers_once(0);
    options_once(0);
    rejects();
    std::atomic<int> go{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
        th.emplace_back([t, &go] {
            while (!go.load()) {}
            // the launch-shape hint is THREAD-local: every thread sees only its own value, whatever the others set
            EXPECT(raft_set_thread_concurrency(t + 2) == 1);
            for (int i = 0; i < 400; ++i) {
                helpers_once(t * 1000 + i);
                options_once(t + i);
                if ((i & 63) == 0) rejects();
                EXPECT(raft_set_thread_concurrency(t + 2) == t + 2);
            }
            EXPECT(raft_set_thread_concurrency(0) == t + 2);       // n < 1 restores the default ...
            EXPECT(raft_set_thread_concurrency(1000) == 1);        // ... and large values are clamped
            EXPECT(raft_set_thread_concurrency(1) == 64);
        });
    go.store(1);
    for (auto &x : th) x.join();
    EXPECT(raft_set_thread_concurrency(1) == 1);                   // the main thread never saw the workers' hints
    raft_set_option("RAFT_CORR_XCD", nullptr);
    raft_set_option("RAFT_LOOKUP_FUSED", nullptr);
    if (failures) {
        std::fprintf(stderr, "%d expectation(s) failed\n", failures);
        return 1;
    }
    std::puts("abi_host_check: ok");
    return 0;
}
