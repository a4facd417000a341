// abi_fault_harness.cpp -- test infrastructure for include/ssdr.h:11-14 ("never throws or aborts").
//
// Built by tests/test_lib_abi.py with g++ -rdynamic and run as a subprocess.  It replaces the global operator new of the
// process; libssdr.so's own allocations (std::vector in csrc/ssdr_api.cpp: their call sites are instantiated inside the
// library, so the return address lies in its text) bind to this replacement through the dynamic symbol table.  When armed,
// the K-th allocation made FROM libssdr.so throws std::bad_alloc -- the HIP runtime's allocations are left alone.
//
//   harness <libssdr.so> cpu      entry points that need no GPU, every library allocation failing
//   harness <libssdr.so> gpu      fault-injection sweep: for every scenario, fail allocation 1, 2, 3, ... until the call runs
//                                 through; each time the call must RETURN (SSDR_ENOMEM), the ctx must stay usable and a
//                                 re-run of the scenario without faults must give SSDR_OK
// Prints one line per scenario and "PASS" / "FAIL"; exit code 0 / 1.  A crossed C boundary would be std::terminate = SIGABRT.
#include "../include/ssdr.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <new>
#include <vector>

static volatile long g_fail_at = 0;      // 0: not armed; K > 0: the K-th library allocation throws; -1: every one
static volatile long g_seen = 0;         // library allocations since arming
static volatile long g_thrown = 0;
static char g_libname[512] = "libssdr.so";

static bool from_lib(void *ret)
{
    Dl_info info;
    if (!dladdr(ret, &info) || !info.dli_fname) return false;
    return strstr(info.dli_fname, g_libname) != nullptr;
}

static void *alloc_or_throw(std::size_t n, void *ret)
{
    if (g_fail_at != 0 && from_lib(ret)) {
        const long k = ++g_seen;
        if (g_fail_at < 0 || k == g_fail_at) { g_thrown++; throw std::bad_alloc(); }
    }
    void *p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new(std::size_t n) { return alloc_or_throw(n, __builtin_return_address(0)); }
void *operator new[](std::size_t n) { return alloc_or_throw(n, __builtin_return_address(0)); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, std::size_t) noexcept { free(p); }
void operator delete[](void *p, std::size_t) noexcept { free(p); }

#define SYM(name) static decltype(&::name) p_##name
SYM(ssdr_create); SYM(ssdr_destroy); SYM(ssdr_set_params); SYM(ssdr_default_params); SYM(ssdr_reset_state);
SYM(ssdr_compile_params); SYM(ssdr_compile_params_decim); SYM(ssdr_compile_params_rate); SYM(ssdr_table);
SYM(ssdr_synth_iq); SYM(ssdr_run_chain); SYM(ssdr_run_wf); SYM(ssdr_run_audio); SYM(ssdr_sync);
SYM(ssdr_get_consts); SYM(ssdr_get_state); SYM(ssdr_set_state); SYM(ssdr_checkpoint_size); SYM(ssdr_checkpoint_save);
SYM(ssdr_checkpoint_load); SYM(ssdr_set_post_channels); SYM(ssdr_set_decimation); SYM(ssdr_set_kiwi_rate);
SYM(ssdr_set_wf_zoom); SYM(ssdr_set_exact_bins); SYM(ssdr_feed_open); SYM(ssdr_feed_close); SYM(ssdr_output_checksum);
SYM(ssdr_set_wfdata_rows); SYM(ssdr_strerror); SYM(ssdr_version); SYM(ssdr_last_hip_error); SYM(ssdr_set_wf_center);
SYM(ssdr_run_db2col); SYM(ssdr_run_playbuffer); SYM(ssdr_set_averaging); SYM(ssdr_set_hop);

static int g_bad = 0;
#define CHECK(cond, ...) do { if (!(cond)) { printf("  FAILED: " __VA_ARGS__); printf("\n"); g_bad++; } } while (0)

static const uint32_t N_CH = 96, N_FRAMES = 8;

template <class F> static void sweep(const char *name, ssdr_ctx *c, F scenario)
{
    long k = 1, injected = 0;
    for (;; k++) {
        g_seen = 0; g_thrown = 0; g_fail_at = k;
        const int rc = scenario();
        g_fail_at = 0;
        if (g_thrown == 0) {                               // fewer than k allocations: the call ran undisturbed
            CHECK(rc == SSDR_OK, "%s: undisturbed run returned %d (%s)", name, rc, p_ssdr_last_hip_error());
            break;
        }
        injected++;
        CHECK(rc == SSDR_ENOMEM, "%s: allocation %ld failed, the call returned %d instead of SSDR_ENOMEM", name, k, rc);
        if (c) {                                           // the ctx survives: the same call goes through afterwards
            const int rc2 = scenario();
            CHECK(rc2 == SSDR_OK, "%s: after the failure of allocation %ld the call returns %d (%s)", name, k, rc2, p_ssdr_last_hip_error());
        }
        if (k > 64) { CHECK(false, "%s: more than 64 allocations in one call?", name); break; }
    }
    printf("%-28s %ld allocation(s) failed in turn, every time an error code\n", name, injected);
}

static int run_cpu()
{
    g_fail_at = -1;
    ssdr_chan_params p;
    ssdr_chan_consts k;
    std::vector<float> taps(SSDR_NTAP_MAX), tab(SSDR_NFFT);      // (the harness's own allocations are not the library's)
    CHECK(p_ssdr_version() != nullptr, "version");
    CHECK(strcmp(p_ssdr_strerror(SSDR_ENOMEM), "out of memory") == 0, "strerror");
    for (int mode = SSDR_MODE_AM; mode <= SSDR_MODE_IQ; mode++) {
        CHECK(p_ssdr_default_params(mode, &p) == SSDR_OK, "default_params(%d)", mode);
        p.f_shift_hz = 1234.5;
        CHECK(p_ssdr_compile_params(&p, &k, taps.data()) == SSDR_OK, "compile_params(%d)", mode);
        CHECK(p_ssdr_compile_params_decim(&p, 4, &k, taps.data()) == SSDR_OK, "compile_params_decim(%d)", mode);
        CHECK(p_ssdr_compile_params_rate(&p, 1, SSDR_RATE_WIDE, &k, taps.data()) == SSDR_OK, "compile_params_rate(%d)", mode);
    }
    CHECK(p_ssdr_table(SSDR_T_WINDOW, tab.data(), SSDR_NFFT) == SSDR_OK, "table");
    CHECK(p_ssdr_table(SSDR_T_WINDOW, tab.data(), 7) == SSDR_EINVAL, "table, bad size");
    ssdr_ctx *c = nullptr;
    const int rc = p_ssdr_create(0, 1u << 20, SSDR_NFFT, SSDR_FRAME, &c);   // no GPU: SSDR_ENODEV; with one: the ctx's vectors fail -> SSDR_ENOMEM
    CHECK(rc == SSDR_ENODEV || rc == SSDR_ENOMEM, "create under failing allocations returned %d", rc);
    CHECK(c == nullptr, "create left a ctx behind");
    g_fail_at = 0;
    printf("cpu: ctx-free entry points under failing allocations: %s\n", g_bad ? "FAIL" : "ok");
    return g_bad;
}

static int run_gpu()
{
    ssdr_ctx *c = nullptr;
    {   // ssdr_create itself: every host allocation on its way fails in turn; no ctx may leak out, the next attempt works
        long k = 1;
        for (;; k++) {
            g_seen = 0; g_thrown = 0; g_fail_at = k;
            ssdr_ctx *t = nullptr;
            const int rc = p_ssdr_create(0, N_CH, SSDR_NFFT, SSDR_FRAME, &t);
            g_fail_at = 0;
            if (g_thrown == 0) { CHECK(rc == SSDR_OK && t, "create: %d (%s)", rc, p_ssdr_last_hip_error()); c = t; break; }
            CHECK(rc == SSDR_ENOMEM && t == nullptr, "create: allocation %ld failed, rc %d ctx %p", k, rc, (void *)t);
            if (k > 64) { CHECK(false, "create: more than 64 allocations?"); break; }
        }
        printf("%-28s %ld allocation(s) failed in turn, every time an error code\n", "ssdr_create", k - 1);
    }
    if (!c) { printf("no ctx\n"); return 1; }
    std::vector<ssdr_chan_params> params(N_CH);
    for (uint32_t i = 0; i < N_CH; i++) {
        p_ssdr_default_params((int)(i % 5), &params[i]);         // every path of the audio stage
        params[i].f_shift_hz = ((int)((i * 37) % 97) - 48) * 100.0;
    }
    std::vector<ssdr_chan_consts> consts(N_CH);
    std::vector<float> taps((size_t)N_CH * SSDR_NTAP_MAX);
    std::vector<ssdr_chan_state> state(N_CH);
    std::vector<int16_t> hist((size_t)N_CH * SSDR_HIST * 2);
    std::vector<uint32_t> sel = {1, 5, 17, 40};
    std::vector<double> centres(N_CH, 500.0);
    uint64_t ck_bytes = 0;
    sweep("ssdr_set_params", c, [&] { return p_ssdr_set_params(c, 0, N_CH, params.data()); });
    sweep("ssdr_reset_state", c, [&] { return p_ssdr_reset_state(c, 0, N_CH); });
    sweep("ssdr_synth_iq+run_chain", c, [&] {
        int rc = p_ssdr_set_params(c, 0, N_CH, params.data());   // marks the channel list dirty: the run rebuilds it (a host vector)
        if (rc == SSDR_OK) rc = p_ssdr_synth_iq(c, N_FRAMES, 7, 0);
        uint32_t lines = 0; int fused = 0;
        if (rc == SSDR_OK) rc = p_ssdr_run_chain(c, &lines, &fused);
        if (rc == SSDR_OK) rc = p_ssdr_sync(c);
        return rc;
    });
    sweep("ssdr_run_wf+run_audio", c, [&] {
        int rc = p_ssdr_set_params(c, 0, N_CH, params.data());
        if (rc == SSDR_OK) rc = p_ssdr_synth_iq(c, N_FRAMES, 8, 0);
        uint32_t lines = 0;
        if (rc == SSDR_OK) rc = p_ssdr_run_wf(c, nullptr, &lines, 0);
        if (rc == SSDR_OK) rc = p_ssdr_run_audio(c, nullptr, nullptr, 0);
        if (rc == SSDR_OK) rc = p_ssdr_sync(c);
        return rc;
    });
    sweep("ssdr_get_consts", c, [&] { return p_ssdr_get_consts(c, 0, N_CH, consts.data(), taps.data()); });
    sweep("ssdr_get_state", c, [&] { return p_ssdr_get_state(c, 0, N_CH, state.data(), hist.data()); });
    sweep("ssdr_set_state", c, [&] { return p_ssdr_set_state(c, 0, N_CH, state.data(), hist.data()); });
    CHECK(p_ssdr_checkpoint_size(c, &ck_bytes) == SSDR_OK && ck_bytes > 0, "checkpoint_size");
    std::vector<uint8_t> blob(ck_bytes);
    sweep("ssdr_checkpoint_save", c, [&] { return p_ssdr_checkpoint_save(c, blob.data()); });
    sweep("ssdr_checkpoint_load", c, [&] { return p_ssdr_checkpoint_load(c, blob.data(), ck_bytes); });
    sweep("ssdr_set_post_channels", c, [&] { return p_ssdr_set_post_channels(c, sel.data(), (uint32_t)sel.size()); });
    sweep("ssdr_set_wfdata_rows", c, [&] { return p_ssdr_set_wfdata_rows(c, 4); });
    {
        std::vector<ssdr_db2col_chan> db(sel.size());
        for (auto &d : db) { memset(&d, 0, sizeof d); d.auto_scale = 1; }
        std::vector<float> color((size_t)(N_FRAMES / 2) * sel.size() * SSDR_NFFT);
        std::vector<ssdr_play_chan> pc(sel.size(), ssdr_play_chan{100.0, 0.0});
        std::vector<int16_t> play((size_t)sel.size() * N_FRAMES * 2048 * 2);
        sweep("ssdr_run_db2col", c, [&] { return p_ssdr_run_db2col(c, db.data(), color.data(), 0); });
        sweep("ssdr_run_playbuffer", c, [&] { return p_ssdr_run_playbuffer(c, pc.data(), play.data(), 0); });
    }
    sweep("ssdr_set_post_channels(all)", c, [&] { return p_ssdr_set_post_channels(c, nullptr, 0); });
    sweep("ssdr_feed_open/close", c, [&] {
        int rc = p_ssdr_feed_open(c, N_FRAMES, 2, 0);
        const int rc2 = p_ssdr_feed_close(c);
        return rc != SSDR_OK ? rc : rc2;
    });
    sweep("ssdr_set_wf_zoom", c, [&] {
        int rc = p_ssdr_set_wf_zoom(c, 2);
        if (rc == SSDR_OK) rc = p_ssdr_set_wf_center(c, 0, N_CH, centres.data());
        const int rc2 = p_ssdr_set_wf_zoom(c, 1);
        return rc != SSDR_OK ? rc : rc2;
    });
    sweep("ssdr_set_exact_bins", c, [&] { int rc = p_ssdr_set_exact_bins(c, 1); const int rc2 = p_ssdr_set_exact_bins(c, 0); return rc != SSDR_OK ? rc : rc2; });
    sweep("ssdr_set_kiwi_rate", c, [&] { int rc = p_ssdr_set_kiwi_rate(c, SSDR_RATE_WIDE); const int rc2 = p_ssdr_set_kiwi_rate(c, SSDR_RATE); return rc != SSDR_OK ? rc : rc2; });
    sweep("ssdr_set_decimation", c, [&] { int rc = p_ssdr_set_decimation(c, 2); const int rc2 = p_ssdr_set_decimation(c, 1); return rc != SSDR_OK ? rc : rc2; });
    // after all of that the ctx still computes: two fresh runs of the same input agree
    uint64_t s1[3] = {0, 0, 0}, s2[3] = {1, 1, 1};
    for (uint64_t *s : {s1, s2}) {
        int rc = p_ssdr_reset_state(c, 0, N_CH);
        if (rc == SSDR_OK) rc = p_ssdr_set_params(c, 0, N_CH, params.data());
        if (rc == SSDR_OK) rc = p_ssdr_synth_iq(c, N_FRAMES, 99, 0);
        uint32_t lines = 0;
        if (rc == SSDR_OK) rc = p_ssdr_run_chain(c, &lines, nullptr);
        if (rc == SSDR_OK) rc = p_ssdr_output_checksum(c, s);
        CHECK(rc == SSDR_OK, "final run: %d (%s)", rc, p_ssdr_last_hip_error());
    }
    CHECK(memcmp(s1, s2, sizeof s1) == 0, "two fresh runs of the same input disagree after the sweeps");
    p_ssdr_destroy(c);
    return g_bad;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <libssdr.so> cpu|gpu\n", argv[0]); return 2; }
    const char *slash = strrchr(argv[1], '/');
    snprintf(g_libname, sizeof g_libname, "%s", slash ? slash + 1 : argv[1]);
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
#define LOAD(name) do { p_##name = (decltype(p_##name))dlsym(h, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; } } while (0)
    LOAD(ssdr_create); LOAD(ssdr_destroy); LOAD(ssdr_set_params); LOAD(ssdr_default_params); LOAD(ssdr_reset_state);
    LOAD(ssdr_compile_params); LOAD(ssdr_compile_params_decim); LOAD(ssdr_compile_params_rate); LOAD(ssdr_table);
    LOAD(ssdr_synth_iq); LOAD(ssdr_run_chain); LOAD(ssdr_run_wf); LOAD(ssdr_run_audio); LOAD(ssdr_sync);
    LOAD(ssdr_get_consts); LOAD(ssdr_get_state); LOAD(ssdr_set_state); LOAD(ssdr_checkpoint_size); LOAD(ssdr_checkpoint_save);
    LOAD(ssdr_checkpoint_load); LOAD(ssdr_set_post_channels); LOAD(ssdr_set_decimation); LOAD(ssdr_set_kiwi_rate);
    LOAD(ssdr_set_wf_zoom); LOAD(ssdr_set_exact_bins); LOAD(ssdr_feed_open); LOAD(ssdr_feed_close); LOAD(ssdr_output_checksum);
    LOAD(ssdr_set_wfdata_rows); LOAD(ssdr_strerror); LOAD(ssdr_version); LOAD(ssdr_last_hip_error); LOAD(ssdr_set_wf_center);
    LOAD(ssdr_run_db2col); LOAD(ssdr_run_playbuffer); LOAD(ssdr_set_averaging); LOAD(ssdr_set_hop);
    const int bad = strcmp(argv[2], "gpu") == 0 ? run_gpu() : run_cpu();
    printf(bad ? "FAIL\n" : "PASS\n");
    return bad ? 1 : 0;
}
