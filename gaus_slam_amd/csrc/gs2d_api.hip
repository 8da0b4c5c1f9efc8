// C ABI of the rasterizer (include/gs2d_rasterizer.h): stage drivers for forward / backward.
// Replaces CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (RAST/cuda_rasterizer/rasterizer_impl.cu:141-153,201-350,354-460 of the reference).
#include "gs2d_common.h"
#include "../../include/gs2d_rasterizer.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>
#include <atomic>
#include <mutex>
#include <thread>
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#endif

namespace {

thread_local std::string g_err;

int fail(const char* what, hipError_t e)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "gs2d: %s: %s", what, hipGetErrorString(e));
    g_err = buf;
    return -1;
}
int fail_msg(const char* what)
{
    g_err = std::string("gs2d: ") + what;
    return -1;
}

#define GS2D_CHECK(expr, what)                        \
    do {                                              \
        hipError_t _e = (expr);                       \
        if (_e != hipSuccess) return fail(what, _e);  \
    } while (0)

// After every stage when `debug` is set (the reference's CHECK_CUDA, auxiliary.h:295-302); launch errors always.
#define GS2D_STAGE(what)                                                             \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) return fail(what, _e);                                 \
        if (debug) { _e = hipStreamSynchronize(s); if (_e != hipSuccess) return fail(what, _e); } \
    } while (0)

// rasterizer_impl.cu:35-50
uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// Optional per-stage device timing (hipEvents recorded on the SAME stream the kernels are launched on).
// Used by bench.py for the per-kernel roofline; off by default (no events are created or recorded).
enum { ST_PREPROCESS = 0, ST_SCAN, ST_DUPLICATE, ST_SORT, ST_RANGES, ST_BLEND_FWD, ST_BLEND_BWD, ST_PREPROCESS_BWD, ST_CULL, ST_COUNT };
struct StageTimer {
    bool enabled = false;
    hipEvent_t ev[ST_COUNT][2];
    bool have_ev = false;
    bool recorded[ST_COUNT] = {};
    void begin(int st, hipStream_t s)
    {
        if (!enabled) return;
        if (!have_ev) {
            for (int i = 0; i < ST_COUNT; i++) { (void)hipEventCreate(&ev[i][0]); (void)hipEventCreate(&ev[i][1]); }
            have_ev = true;
        }
        (void)hipEventRecord(ev[st][0], s);
    }
    void end(int st, hipStream_t s)
    {
        if (!enabled) return;
        (void)hipEventRecord(ev[st][1], s);
        recorded[st] = true;
    }
};
StageTimer g_timer;

// deterministic backward (opt-in, process-wide): must not change between a forward and its backward
std::atomic<int> g_deterministic{0};
std::atomic<int> g_reference_binning{0};
// 1 (default): a single-frame forward enqueues ALL its kernels before the host looks at num_rendered (the kernels behind
// duplicate read the count on the device, DevBin); 0: the round-3 order (duplicate, host wait, the rest) -- kept for A/B
// measurements and as the path debug mode, the deterministic mode and images of more than 4096 tiles take anyway.
// (GS2D_LAUNCH_AHEAD=0 in the environment: initial value 0 -- for A/B runs of one build)
static int launch_ahead_default() { const char* e = getenv("GS2D_LAUNCH_AHEAD"); return (e && e[0] == '0') ? 0 : 1; }
std::atomic<int> g_launch_ahead{launch_ahead_default()};

// pinned host word for the num_rendered read-back (one per host thread)
struct PinnedWord {
    uint32_t* p = nullptr;
    ~PinnedWord() { if (p) (void)hipHostFree(p); }
};
thread_local PinnedWord g_pinned;
// Forward records.  The three scratch chunks carry no header a host could read without a device round trip, and the
// backward's signature (the reference's, rasterizer.h:73-107) has no chunk sizes, so the library remembers what it needs
// about the forwards whose backward may still come, keyed by the geometry chunk's address:
//   * `det`, `R`, `bin`, `bin_bytes`: the mode the forward sized the binning chunk for and how large the allocator made
//     it.  The backward lays the chunk out with the FORWARD's mode and refuses to run (gs2d_last_error) when the live
//     gs2d_set_deterministic flag disagrees with it, when R or the binning pointer are not the forward's, or when the chunk
//     is smaller than the layout it is asked to use -- a deterministic backward never runs on a chunk this table does
//     not vouch for, so the 320 B x R of partial records can never land outside the chunk;
//   * `clean`: the gradient accumulator (grad_rec) inside the geometry chunk is known to be all zero -- the forward's
//     blend kernel clears it with spare store slots, so the first backward on that forward needs no memset (7 us on the
//     critical path at 500k Gaussians).  Contract (include/gs2d_rasterizer.h): the chunks are opaque and must not be
//     written, copied over or relocated between a forward and its backward.
// A record enters when its forward has launched the blend kernel and leaves when a later forward is handed the same
// geometry address; with more than 64 forwards in flight the oldest record is overwritten (its backward then clears the
// accumulator itself and cannot run in deterministic mode).
struct FwdRecord {
    const void* geom = nullptr;
    const void* bin = nullptr;
    size_t bin_bytes = 0;
    int det = 0, R = 0, P = 0;
    bool clean = false;
};
struct FwdTable {
    static constexpr int N = 64;
    std::mutex m;
    FwdRecord e[N];
    int next = 0;
    void drop(const void* geom) { std::lock_guard<std::mutex> l(m); for (auto& r : e) if (r.geom == geom) r = FwdRecord(); }
    void add(const FwdRecord& r) { std::lock_guard<std::mutex> l(m); e[next] = r; next = (next + 1) % N; }
    // copy of the record of `geom` (false: unknown chunk); take_clean also hands over the "accumulator is zero" token
    bool find(const void* geom, bool take_clean, FwdRecord* out)
    {
        std::lock_guard<std::mutex> l(m);
        for (auto& r : e)
            if (r.geom == geom && geom != nullptr) {
                *out = r;
                if (take_clean) r.clean = false;
                return true;
            }
        return false;
    }
};
FwdTable g_fwd;

// previous forward's num_rendered for the same problem shape: sizes the early binning allocation (a guess only: the
// chunk is re-requested with the exact size when the guess was too small)
struct LastCount { int P = -1, W = 0, H = 0; uint32_t R = 0; };
thread_local LastCount g_last[GS2D_MAX_BATCH];

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    _mm_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

}  // namespace

extern "C" {

const char* gs2d_last_error(void) { return g_err.c_str(); }

void gs2d_stage_timing_enable(int on)
{
    g_timer.enabled = on != 0;
    for (int i = 0; i < ST_COUNT; i++) g_timer.recorded[i] = false;
}

// ms[9]: preprocess, scan, duplicate, sort, ranges, blend_fwd, blend_bwd, preprocess_bwd, cull of the most recent calls
// (-1 where nothing was recorded).  Synchronises on the recorded events.
int gs2d_stage_timing_read(float ms[9])
{
    for (int i = 0; i < ST_COUNT; i++) {
        ms[i] = -1.f;
        if (!g_timer.recorded[i]) continue;
        if (hipEventSynchronize(g_timer.ev[i][1]) != hipSuccess) return -1;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_timer.ev[i][0], g_timer.ev[i][1]) != hipSuccess) return -1;
        ms[i] = t;
    }
    return 0;
}

// begin / end of every recorded stage in ms relative to the begin of the most recent preprocess stage (-1: not recorded): where
// the stream sat idle between stages (host reaction to num_rendered, the caller's work between forward and backward ...)
int gs2d_stage_timing_read_abs(float ms[18])
{
    for (int i = 0; i < 2 * ST_COUNT; i++) ms[i] = -1.f;
    if (!g_timer.recorded[ST_PREPROCESS]) return 0;
    for (int i = 0; i < ST_COUNT; i++) {
        if (!g_timer.recorded[i]) continue;
        if (hipEventSynchronize(g_timer.ev[i][1]) != hipSuccess) return -1;
        for (int e = 0; e < 2; e++) {
            float t = 0.f;
            // (a stage recorded before the base event gives a negative offset: the backward of the PREVIOUS step when called
            // right after a forward)
            if (hipEventElapsedTime(&t, g_timer.ev[ST_PREPROCESS][0], g_timer.ev[i][e]) != hipSuccess) {
                if (hipEventElapsedTime(&t, g_timer.ev[i][e], g_timer.ev[ST_PREPROCESS][0]) != hipSuccess) return -1;
                t = -t;
            }
            ms[2 * i + e] = t;
        }
    }
    return 0;
}

void gs2d_set_deterministic(int on) { g_deterministic.store(on != 0); }
int gs2d_get_deterministic(void) { return g_deterministic.load(); }
void gs2d_set_reference_binning(int on) { g_reference_binning.store(on != 0); }
void gs2d_set_launch_ahead(int on) { g_launch_ahead.store(on != 0); }
int gs2d_get_launch_ahead(void) { return g_launch_ahead.load(); }
int gs2d_get_reference_binning(void) { return g_reference_binning.load(); }

#ifndef GS2D_SOURCE_HASH
#define GS2D_SOURCE_HASH "unknown"   /* gaus_slam_amd/build.py passes the hash of csrc/ + the C-ABI header */
#endif
const char* gs2d_build_info(void) { return "gs2d-hip gfx950 strict-fp (fp-contract=off) " __DATE__ " src " GS2D_SOURCE_HASH; }

size_t gs2d_geometry_bytes(int P) { return geom_layout(P).total; }
size_t gs2d_image_bytes(int width, int height) { return img_layout(width, height).total; }
size_t gs2d_binning_bytes(int R) { return bin_layout(R, g_deterministic.load() != 0).total; }

void gs2d_geometry_layout(int P, size_t o[5])
{
    const GeomLayout L = geom_layout(P);
    o[0] = L.depths; o[1] = L.tiles_touched; o[2] = L.point_offsets; o[3] = L.rec; o[4] = L.clamped;
}
void gs2d_binning_layout(int R, size_t o[2])
{
    const BinLayout L = bin_layout(R);
    o[0] = L.point_list; o[1] = L.keys;  // (L.hits is internal)
}
void gs2d_image_layout(int width, int height, size_t o[2])
{
    const ImgLayout L = img_layout(width, height);
    o[0] = L.ranges; o[1] = L.pix;
}

}  // extern "C"

namespace {

// ----------------------------------------------------------------------------------------------------------------------
// The forward in three phases, so that K frames over the same Gaussians can share ONE blend launch (gs2d_forward_batch):
//   A  per frame: geometry chunk, preprocess
//   B  per frame: image + binning chunks, duplicate (its last workgroup sends num_rendered to pinned word `slot`), wait for it, sort
//   C  once:      blend_fwd over the tiles of all frames, forward records
// gs2d_forward[_posed] is A, B, C with K = 1.
struct FwdShared {   // what all frames of a call have in common
    int P, D, M, width, height, use_sa, debug;
    const float* background;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *transMat_precomp;
    float scale_modifier;
    hipStream_t s;
};
struct FwdFrame {    // one frame: inputs, then the state the phases hand on
    gs2d_alloc_fn geometry_alloc, binning_alloc, image_alloc;
    void *geometry_user, *binning_user, *image_user;
    const float *viewmatrix, *projmatrix, *cam_pos, *pose_Rt, *pose_quat;
    float* out_color; float* out_others; int* radii;
    // state
    char* geom = nullptr; char* bin = nullptr; char* img = nullptr;
    size_t bin_bytes = 0;
    GeomLayout GL; CamParams cam;
    volatile uint32_t* pinned = nullptr;
    bool fused_sort = false;     // the blend kernel sorts this frame's tile lists itself (phase -1) ...
    int sort_cap = 0;            // ... at this capacity (gs2d_fused_sort_cap)
    bool spec = false;           // duplicate runs before the host knows num_rendered (single-pass binning: tiles <= GS2D_BIN_MAX_TILES)
    bool store_pending = false;  // a kernel WILL store into `pinned`: every return path first waits for that store
    bool ahead = false;          // the stages behind duplicate were launched with the count on the device (fwd_phase_b), the host
                                 // has not looked at num_rendered yet: fwd_phase_d does, after the blend kernel is enqueued
    int cap = 0;                 // ahead: the capacity the binning chunk `bin` was sized for
    bool det = false;
    int R = 0;
    gs2d::BlendFwdFrame bf;
};

int fwd_validate(const FwdShared& c)
{
    if (c.P < 0 || c.width <= 0 || c.height <= 0) return fail_msg("bad sizes");
    if (c.P >= (1 << 28)) return fail_msg("more than 2^28 - 1 Gaussians (the backward packs the Gaussian id into 28 bits)");
    if (c.P == 0) return 0;
    if (c.colors_precomp == nullptr && c.shs == nullptr) return fail_msg("provide shs or colors_precomp");
    if (c.transMat_precomp == nullptr && (c.scales == nullptr || c.rotations == nullptr))
        return fail_msg("provide scales+rotations or transMat_precomp");
    if (c.colors_precomp == nullptr && (c.M <= 0 || (c.D + 1) * (c.D + 1) > c.M)) return fail_msg("sh degree exceeds coefficients");
    return 0;
}

// Poll, then yield: the thread spins with `pause`; if the stream is backed up behind earlier work for long (or several ranks
// share the host's cores) it hands its time slice back between looks instead of burning a core per rank.  Falls back to a
// real synchronise after ~2 s (surfacing any GPU error).
// The wait is typically as long as the previous backward (the host runs one step ahead: 0.25 ms at the bench size), so the
// calling thread spins through it; it starts yielding its core only after a millisecond.  (Yielding after 100 us, as until
// round 3, put every steady-state wait into sched_yield: on a box with busy neighbours the thread can lose its core for a
// whole time slice that way -- host steps of 0.6-0.9 ms were seen once in a few hundred, scripts/dev/window_jitter.py.)
#ifndef GS2D_SPIN_BEFORE_YIELD_US
#define GS2D_SPIN_BEFORE_YIELD_US 1000
#endif
// 1 (default): the launch-ahead forward's one wait (fwd_phase_d) SLEEPS through most of itself instead of spinning.  In steady
// state the host runs a step ahead and the wait is as long as what is left of the previous step's backward (~0.2 ms per step at
// the bench size); the thread keeps a running mean of its recent waits, sleeps once for that mean minus GS2D_NAP_MARGIN_US (a
// timer's slack plus a wake-up: a sleep that overshoots the count's arrival delays the host's next launches, measured -2 % at
// 640x480 / 200k with plain 20-us naps) and polls the rest.  Short waits (idle GPU, small scenes) never sleep.
// GS2D_NAP_WAIT=0 in the environment: spin only, as the round-3 order does.
static int nap_wait_default() { const char* e = getenv("GS2D_NAP_WAIT"); return (e && e[0] == '0') ? 0 : 1; }
std::atomic<int> g_nap_wait{nap_wait_default()};
#ifndef GS2D_NAP_MARGIN_US
#define GS2D_NAP_MARGIN_US 150.0
#endif
thread_local double g_wait_mean_us = 0.0;

bool wait_total(FwdFrame& f, hipStream_t s, bool may_nap = false)
{
    if (!f.store_pending) return true;
    f.store_pending = false;
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    bool yielding = false;
    if (may_nap && g_nap_wait.load() != 0) {
        const double nap_us = g_wait_mean_us - GS2D_NAP_MARGIN_US;
        if (nap_us > 20.0 && *f.pinned == 0xFFFFFFFFu)
            std::this_thread::sleep_for(std::chrono::nanoseconds((long long)(nap_us * 1e3)));
        while (*f.pinned == 0xFFFFFFFFu) {
            cpu_relax();
            if ((++spins & 0x3FFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
                return hipStreamSynchronize(s) == hipSuccess && *f.pinned != 0xFFFFFFFFu;
        }
        const double w = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        // running mean of the waits; one wait counts for at most twice the mean + 50 us, so that a single overslept or
        // disturbed wait cannot make the next ones oversleep, while a lasting change (the first steps of a loop find an idle
        // GPU, the later ones a full queue) still pulls the mean up by 10 % per step
        const double wc = w < 2.0 * g_wait_mean_us + 50.0 ? w : 2.0 * g_wait_mean_us + 50.0;
        g_wait_mean_us = g_wait_mean_us == 0.0 ? wc : 0.9 * g_wait_mean_us + 0.1 * wc;
        return true;
    }
    while (*f.pinned == 0xFFFFFFFFu) {
        if (yielding) std::this_thread::yield(); else cpu_relax();
        if ((++spins & (yielding ? 0x3Fu : 0x3FFu)) == 0) {
            const auto dt = std::chrono::steady_clock::now() - t0;
            if (dt > std::chrono::seconds(2)) return hipStreamSynchronize(s) == hipSuccess && *f.pinned != 0xFFFFFFFFu;
            if (dt > std::chrono::microseconds(GS2D_SPIN_BEFORE_YIELD_US)) yielding = true;
        }
    }
    return true;
}

int fwd_phase_a(const FwdShared& c, FwdFrame& f, int slot)
{
    const int debug = c.debug;
    hipStream_t s = c.s;
    if (f.pose_Rt == nullptr && f.pose_quat != nullptr) return fail_msg("pose_quat without pose_Rt");
    if (f.pose_Rt != nullptr && c.transMat_precomp != nullptr) return fail_msg("a pose cannot be combined with transMat_precomp");
    if (!f.geometry_alloc || !f.binning_alloc || !f.image_alloc) return fail_msg("allocator callbacks are required");
    f.GL = geom_layout(c.P);
    const GeomLayout& GL = f.GL;
    f.geom = (char*)f.geometry_alloc(f.geometry_user, GL.total);
    if (!f.geom) return fail_msg("geometry allocation failed");
    g_fwd.drop(f.geom);

    CamParams& cam = f.cam;
    cam.vm = f.viewmatrix; cam.pm = f.projmatrix; cam.campos = f.cam_pos;
    cam.W = c.width; cam.H = c.height;
    cam.gx = (c.width + GS2D_TILE - 1) / GS2D_TILE;
    cam.gy = (c.height + GS2D_TILE - 1) / GS2D_TILE;
    cam.tight = g_reference_binning.load() == 0;

    char* geom = f.geom;
    float* depths = (float*)(geom + GL.depths);
    uint32_t* tiles_touched = (uint32_t*)(geom + GL.tiles_touched);
    float4* rec = (float4*)(geom + GL.rec);
    uint8_t* clamped = (uint8_t*)(geom + GL.clamped);
    ushort4* rect = (ushort4*)(geom + GL.rect);
    uint32_t* scan_tmp = (uint32_t*)(geom + GL.scan_tmp);

    g_timer.begin(ST_PREPROCESS, s);
    gs2d::launch_preprocess_fwd(c.P, c.D, c.M, c.means3D, c.scales, c.scale_modifier, c.rotations, c.opacities, c.shs,
                                c.transMat_precomp, c.colors_precomp, cam, f.radii, depths, rec, tiles_touched, rect, clamped,
                                f.pose_Rt, f.pose_quat, scan_tmp, s);
    g_timer.end(ST_PREPROCESS, s);
    GS2D_STAGE("preprocess");

    // The one host sync of the forward (rasterizer_impl.cu:287): the binning chunk is sized by num_rendered.
    // Read-back without an OS-level wait and without a copy: the kernel that knows the total first stores it straight into a
    // pinned host word (system-scope store) that was pre-set to a sentinel and is polled in phase B.  On a loaded host a
    // blocking hipStreamSynchronize can cost milliseconds of scheduler latency per call; the poll returns within a
    // microsecond of the store landing.
    // coherent pinned memory: the device's system-scope store must become visible to the polling CPU without a sync
    if (!g_pinned.p) GS2D_CHECK(hipHostMalloc((void**)&g_pinned.p, 64, hipHostMallocCoherent), "hipHostMalloc");
    f.pinned = g_pinned.p + slot;  // 16 words: one per frame of a batch
    // Single-pass binning (every image up to 4096 tiles): nothing more here -- duplicate_kernel finishes the prefix sum itself
    // and its last workgroup is the one that stores the total, while the kernel still runs (phase B).  Otherwise (the 8-bit
    // radix passes place duplicate's output by the pass count's parity, i.e. inside arrays laid out by num_rendered): the
    // single-workgroup scan of the block sums, which stores the total.
    f.spec = img_layout(c.width, c.height).tiles <= GS2D_BIN_MAX_TILES;
    if (f.spec) return 0;
    const int nblk = (c.P + 255) / 256;  // scan_tmp[0..nblk) = per-workgroup sums, then exclusive block offsets
    uint32_t* total_dev = scan_tmp + nblk + 8;
    *f.pinned = 0xFFFFFFFFu;
    g_timer.begin(ST_SCAN, s);
    gs2d::launch_offsets_blocksums(c.P, scan_tmp, total_dev, g_pinned.p + slot, s);
    g_timer.end(ST_SCAN, s);
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail("scan", le);  // nothing was enqueued: nothing will store into the word
    }
    // From here on the scan kernel WILL store into the pinned word, so every return path first waits for that store
    // (a later forward on this thread would otherwise reset the sentinel and could pick up this call's stale total).
    f.store_pending = true;
    return 0;
}

int fwd_phase_b(const FwdShared& c, FwdFrame& f, int slot, bool allow_ahead = true)
{
    const int debug = c.debug;
    hipStream_t s = c.s;
    const int P = c.P, width = c.width, height = c.height;
    const GeomLayout& GL = f.GL;
    const CamParams& cam = f.cam;
    const ImgLayout IL = img_layout(width, height);
    // While the GPU works on the preprocess, ask for the image chunk and for the binning chunk, the latter sized from the
    // previous call's count for this problem shape (+12.5 %): the allocator callbacks (trips through the caller's runtime)
    // then cost nothing on the critical path.  If the guess turns out too small the callback is simply invoked a second
    // time with the exact size.
    f.img = (char*)f.image_alloc(f.image_user, IL.total);
    if (!f.img) { (void)wait_total(f, s); return fail_msg("image allocation failed"); }
    LastCount& last = g_last[slot];
    const bool same_shape = last.P == P && last.W == width && last.H == height;
    const size_t guess_R = same_shape ? (size_t)last.R + last.R / 8 + 4096 : (size_t)P * 3 + 4096;
    const int C0 = (int)(guess_R > 0x7fffffffull ? 0x7fffffffull : guess_R);
    const bool det = g_deterministic.load() != 0;
    char* geom = f.geom;
    char* img = f.img;
    float* depths = (float*)(geom + GL.depths);
    uint32_t* tiles_touched = (uint32_t*)(geom + GL.tiles_touched);
    uint32_t* point_offsets = (uint32_t*)(geom + GL.point_offsets);
    ushort4* rect = (ushort4*)(geom + GL.rect);
    uint32_t* scan_tmp = (uint32_t*)(geom + GL.scan_tmp);
    // the chunk holds C >= num_rendered instances: only the arrays duplicate_kernel writes are laid out by C (bin_layout)
    int C = C0;
    const size_t pre_bytes = bin_layout(C0, det, C0).total;
    char* bin_pre = (char*)f.binning_alloc(f.binning_user, pre_bytes);
    if (f.spec) {
        // duplicate_kernel BEFORE the host knows num_rendered: it writes the unsorted pairs into the guessed chunk (nothing at
        // or beyond its capacity) and its last workgroup stores the total into the pinned word while the kernel is still running;
        // the host's reaction -- this poll, the launches below -- overlaps with the kernel instead of following a scan kernel.
        if (!bin_pre) return fail_msg("binning allocation failed");
        const BinLayout B0 = bin_layout(C0, det, C0);
        *f.pinned = 0xFFFFFFFFu;
        // launch-ahead: the kernels behind duplicate take the count from `total_dev` (device word in the geometry chunk's scan
        // scratch, written by duplicate's last workgroup next to the pinned host word) and are enqueued right here -- the
        // GPU never waits for the host between duplicate and the blend; the host looks at the count in fwd_phase_d
        const bool ahead = allow_ahead && g_launch_ahead.load() != 0 && !debug && !det;
        uint32_t* total_dev = scan_tmp + (P + 255) / 256 + 8;
        g_timer.begin(ST_DUPLICATE, s);
        gs2d::launch_duplicate(P, rect, depths, tiles_touched, scan_tmp, 0, point_offsets, cam.gx, (uint64_t*)(bin_pre + B0.keys_alt),
                               (uint32_t*)(bin_pre + B0.vals_alt), (uint32_t)C0, (uint32_t*)f.pinned, s, ahead ? total_dev : nullptr);
        g_timer.end(ST_DUPLICATE, s);
        {
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) return fail("duplicate", le);  // nothing was enqueued: nothing will store into the word
        }
        f.store_pending = true;
        if (ahead) {
            const gs2d::DevBin db = {bin_pre, total_dev, (uint32_t)C0, 0};
            uint2* ranges = (uint2*)(img + IL.ranges);
            const int tile_bits = (int)higher_msb((uint32_t)(cam.gx * cam.gy));
            g_timer.begin(ST_SORT, s);
            gs2d::launch_bin_by_tile_dev(db, IL.tiles, tile_bits, ranges, s);
            // depth-sort capacity class from the count this shape had last time (any class is correct: longer lists take the
            // global-memory variant); the sort is fused into the blend kernel at the class its LDS holds anyway
            const long long Rg = same_shape ? (long long)last.R : (long long)C0;
            const int cap_class = gs2d::tile_sort_capacity(Rg, IL.tiles);
            f.sort_cap = gs2d::gs2d_fused_sort_cap(cap_class);
            f.fused_sort = f.sort_cap > 0;
            if (!f.fused_sort) gs2d::launch_tile_depth_sort_dev(db, IL.tiles, ranges, cap_class, 0, s);
            g_timer.end(ST_SORT, s);
            GS2D_STAGE("bin by tile (count on the device)");
            f.ahead = true; f.cap = C0; f.det = false;
            f.bin = bin_pre;
            f.bf.ranges = ranges; f.bf.point_list = nullptr; f.bf.rec = (const float4*)(geom + GL.rec);
            f.bf.out_color = f.out_color; f.bf.out_others = f.out_others; f.bf.pix_state = (float*)(img + IL.pix);
            f.bf.hits = nullptr; f.bf.hits4 = nullptr; f.bf.zero = (float4*)(geom + GL.grad_rec);
            f.bf.keys = nullptr; f.bf.keys_alt = nullptr; f.bf.vals_alt = nullptr;
            f.bf.dev = db;
            return 0;
        }
    }
    if (!wait_total(f, s)) return fail_msg("num_rendered read-back failed");
    if (debug) GS2D_CHECK(hipStreamSynchronize(s), f.spec ? "duplicate" : "scan");
    const uint32_t num_rendered_u = *f.pinned;
    if (num_rendered_u == 0xFFFFFFFFu) return fail_msg("num_rendered read-back failed");
    if (num_rendered_u > 0x7fffffffu) return fail_msg("num_rendered overflows int32");
    const int R = (int)num_rendered_u;
    f.R = R;

    last.P = P; last.W = width; last.H = height; last.R = num_rendered_u;
    const bool reuse_pre = bin_pre && R <= C0;
    if (!reuse_pre) C = R;
    const BinLayout BL = bin_layout(R, det, C);  // fixed offsets follow the true count, the unsorted pairs the chunk's capacity
    char* bin = reuse_pre ? bin_pre : (char*)f.binning_alloc(f.binning_user, BL.total);
    if (!bin) return fail_msg("binning allocation failed");
    f.bin = bin;
    f.bin_bytes = BL.total;
    uint32_t* point_list = (uint32_t*)(bin + BL.point_list);
    uint64_t* keys = (uint64_t*)(bin + BL.keys);
    uint32_t* vals_alt = (uint32_t*)(bin + BL.vals_alt);
    uint64_t* keys_alt = (uint64_t*)(bin + BL.keys_alt);
    uint32_t* hist = (uint32_t*)(bin + BL.hist);
    uint2* ranges = (uint2*)(img + IL.ranges);

    // Sort = (a) stable binning of the pairs by tile id (single counting-sort pass; 8-bit radix passes on the tile
    // bits when there are too many tiles for LDS), Gaussian order kept inside a tile; (b) tile ranges; (c) per-tile
    // stable depth sort in LDS.  Same final order as the reference's single 64-bit SortPairs on bits [0, 32+bit)
    // (rasterizer_impl.cu:306-314).
    const int tile_bits = (int)higher_msb((uint32_t)(cam.gx * cam.gy));
    const int end_bit = 32 + tile_bits;
    const bool one_pass = IL.tiles <= GS2D_BIN_MAX_TILES;
    const int passes = one_pass ? 1 : (end_bit - 32 + 7) / 8;
    // unsorted pairs go where the ping-pong needs them so that the result lands in (keys, point_list)
    uint64_t* k_unsorted = (passes & 1) ? keys_alt : keys;
    uint32_t* v_unsorted = (passes & 1) ? vals_alt : point_list;
    // always launched: it also completes point_offsets (all zeros when nothing is visible)
    if (!f.spec) {
        g_timer.begin(ST_DUPLICATE, s);
        gs2d::launch_duplicate(P, rect, depths, tiles_touched, scan_tmp, 1, point_offsets, cam.gx, k_unsorted, v_unsorted, 0xFFFFFFFFu,
                               nullptr, s);
        g_timer.end(ST_DUPLICATE, s);
    } else if (!reuse_pre) {
        // the guess was too small (first call for this shape, or the scene grew by more than 12.5 %): once more, into the
        // chunk of the right size; the total is known, nothing is published
        gs2d::launch_duplicate(P, rect, depths, tiles_touched, scan_tmp, 0, point_offsets, cam.gx, k_unsorted, v_unsorted, (uint32_t)C,
                               nullptr, s);
    }
    GS2D_STAGE("duplicate");
    if (R > 0) {
        g_timer.begin(ST_SORT, s);
        if (one_pass) {
            gs2d::launch_bin_by_tile(R, IL.tiles, tile_bits, k_unsorted, v_unsorted, keys, point_list, hist, ranges, s);
        } else {
            gs2d::launch_sort_pairs(R, keys, point_list, keys_alt, vals_alt, 32, end_bit, hist, BL.hist_elems, s);
        }
        GS2D_STAGE("bin by tile");
    }
    if (R == 0 || !one_pass) {
        g_timer.begin(ST_RANGES, s);
        gs2d::launch_tile_ranges(R, keys, ranges, IL.tiles, s);
        g_timer.end(ST_RANGES, s);
        GS2D_STAGE("ranges");
    }
    // The per-tile depth sort: as phase -1 of the blend kernel when the lists fit the LDS its workgroups hold anyway (packed
    // pairs from the single-pass binning, capacity classes up to 3072), else as the kernel of its own
    f.sort_cap = one_pass && R > 0 ? gs2d::gs2d_fused_sort_cap(gs2d::tile_sort_capacity(R, IL.tiles)) : 0;
    f.fused_sort = f.sort_cap > 0;
    if (R > 0) {
        if (!f.fused_sort)
            gs2d::launch_tile_depth_sort(R, IL.tiles, ranges, keys, point_list, keys_alt, vals_alt, one_pass ? 1 : 0, debug ? 1 : 0, s);
        g_timer.end(ST_SORT, s);
        GS2D_STAGE("tile depth sort");
    }
    f.bf.ranges = ranges; f.bf.point_list = point_list; f.bf.rec = (const float4*)(geom + GL.rec);
    f.bf.out_color = f.out_color; f.bf.out_others = f.out_others; f.bf.pix_state = (float*)(img + IL.pix);
    f.bf.hits = (uint8_t*)(bin + BL.hits); f.bf.hits4 = (uint8_t*)(bin + BL.hits4);
    f.bf.zero = (float4*)(geom + GL.grad_rec);
    f.bf.keys = keys; f.bf.keys_alt = keys_alt; f.bf.vals_alt = vals_alt;
    f.bf.dev = gs2d::DevBin{nullptr, nullptr, 0u, 0};
    return 0;
}

// Phases A and B for K frames with every stage batched (tiles <= GS2D_BIN_MAX_TILES, no pose).  Same per-frame results as
// fwd_phase_a + fwd_phase_b: the kernels' bodies are the single-frame kernels' own.
int fwd_batch_fused(const FwdShared& c, FwdFrame* f, int K)
{
    const int debug = c.debug;
    hipStream_t s = c.s;
    const int P = c.P, width = c.width, height = c.height;
    const GeomLayout GL = geom_layout(P);
    const ImgLayout IL = img_layout(width, height);
    gs2d::PreFwdFrames pre;
    gs2d::BinFrames bin;
    CamParams cam0;
    cam0.vm = nullptr; cam0.pm = nullptr; cam0.campos = nullptr;
    cam0.W = width; cam0.H = height;
    cam0.gx = (width + GS2D_TILE - 1) / GS2D_TILE;
    cam0.gy = (height + GS2D_TILE - 1) / GS2D_TILE;
    cam0.tight = g_reference_binning.load() == 0;
    if (!g_pinned.p) GS2D_CHECK(hipHostMalloc((void**)&g_pinned.p, 64, hipHostMallocCoherent), "hipHostMalloc");
    for (int k = 0; k < K; k++) {
        if (!f[k].geometry_alloc || !f[k].binning_alloc || !f[k].image_alloc) return fail_msg("allocator callbacks are required");
        f[k].GL = GL;
        f[k].geom = (char*)f[k].geometry_alloc(f[k].geometry_user, GL.total);
        if (!f[k].geom) return fail_msg("geometry allocation failed");
        g_fwd.drop(f[k].geom);
        f[k].cam = cam0;
        f[k].cam.vm = f[k].viewmatrix; f[k].cam.pm = f[k].projmatrix; f[k].cam.campos = f[k].cam_pos;
        char* geom = f[k].geom;
        gs2d::PreFwdFrame& q = pre.f[k];
        q.vm = f[k].viewmatrix; q.pm = f[k].projmatrix; q.campos = f[k].cam_pos;
        q.radii = f[k].radii; q.depths = (float*)(geom + GL.depths); q.rec = (float4*)(geom + GL.rec);
        q.tiles_touched = (uint32_t*)(geom + GL.tiles_touched); q.rect = (ushort4*)(geom + GL.rect);
        q.clamped = (uint8_t*)(geom + GL.clamped); q.block_sums = (uint32_t*)(geom + GL.scan_tmp);
        gs2d::BinFrame& b = bin.f[k];
        b.rect = q.rect; b.depths = q.depths; b.tiles_touched = q.tiles_touched; b.block_sums = q.block_sums;
        f[k].pinned = g_pinned.p + k;
        f[k].spec = true;
        b.point_offsets = (uint32_t*)(geom + GL.point_offsets);
        b.R = 0; b.nblocks = 0;
    }
    for (int k = K; k < GS2D_MAX_BATCH; k++) { pre.f[k] = pre.f[0]; bin.f[k] = bin.f[0]; }
    g_timer.begin(ST_PREPROCESS, s);
    gs2d::launch_preprocess_fwd_batch(P, K, c.D, c.M, c.means3D, c.scales, c.scale_modifier, c.rotations, c.opacities, c.shs,
                                      c.transMat_precomp, c.colors_precomp, cam0, pre, s);
    g_timer.end(ST_PREPROCESS, s);
    GS2D_STAGE("preprocess (batch)");
    // while the GPU works on the preprocess: image chunks and, sized from the previous call's counts, the binning chunks; then
    // duplicate for all K frames BEFORE the host knows the K totals (see fwd_phase_b): each frame's last workgroup stores its own
    const bool det = g_deterministic.load() != 0;
    char* bin_pre[GS2D_MAX_BATCH];
    int cap[GS2D_MAX_BATCH];
    for (int k = 0; k < K; k++) {
        f[k].img = (char*)f[k].image_alloc(f[k].image_user, IL.total);
        if (!f[k].img) return fail_msg("image allocation failed");
        LastCount& last = g_last[k];
        const bool same_shape = last.P == P && last.W == width && last.H == height;
        const size_t guess_R = same_shape ? (size_t)last.R + last.R / 8 + 4096 : (size_t)P * 3 + 4096;
        cap[k] = (int)(guess_R > 0x7fffffffull ? 0x7fffffffull : guess_R);
        const BinLayout B0 = bin_layout(cap[k], det, cap[k]);
        bin_pre[k] = (char*)f[k].binning_alloc(f[k].binning_user, B0.total);
        if (!bin_pre[k]) return fail_msg("binning allocation failed");
        gs2d::BinFrame& b = bin.f[k];
        b.keys_unsorted = (uint64_t*)(bin_pre[k] + B0.keys_alt); b.vals_unsorted = (uint32_t*)(bin_pre[k] + B0.vals_alt);
        b.capacity = (uint32_t)cap[k];
        b.total_host = (uint32_t*)f[k].pinned;
        *f[k].pinned = 0xFFFFFFFFu;
    }
    for (int k = K; k < GS2D_MAX_BATCH; k++) bin.f[k] = bin.f[0];
    g_timer.begin(ST_DUPLICATE, s);
    gs2d::launch_duplicate_batch(P, K, cam0.gx, bin, s);
    g_timer.end(ST_DUPLICATE, s);
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail("duplicate (batch)", le);  // nothing was enqueued: nothing will store into the words
    }
    for (int k = 0; k < K; k++) f[k].store_pending = true;
    const int tile_bits = (int)higher_msb((uint32_t)(cam0.gx * cam0.gy));
    bool redo = false;
    for (int k = 0; k < K; k++) {
        if (!wait_total(f[k], s)) return fail_msg("num_rendered read-back failed");
        const uint32_t num_rendered_u = *f[k].pinned;
        if (num_rendered_u == 0xFFFFFFFFu) return fail_msg("num_rendered read-back failed");
        if (num_rendered_u > 0x7fffffffu) return fail_msg("num_rendered overflows int32");
        f[k].R = (int)num_rendered_u;
        LastCount& last = g_last[k];
        last.P = P; last.W = width; last.H = height; last.R = num_rendered_u;
        redo = redo || f[k].R > cap[k];
    }
    for (int k = 0; k < K; k++) {
        const int R = f[k].R;
        char* bn = bin_pre[k];
        if (R > cap[k]) {  // this frame's guess was too small: a chunk of the exact size
            cap[k] = R;
            bn = (char*)f[k].binning_alloc(f[k].binning_user, bin_layout(R, det, R).total);
            if (!bn) return fail_msg("binning allocation failed");
        }
        const BinLayout BL = bin_layout(R, det, cap[k]);
        f[k].bin = bn;
        f[k].bin_bytes = BL.total;
        gs2d::BinFrame& b = bin.f[k];
        b.capacity = (uint32_t)cap[k];
        b.total_host = nullptr;
        b.R = R;
        b.point_list = (uint32_t*)(bn + BL.point_list);
        b.keys = (uint64_t*)(bn + BL.keys);
        b.vals_alt = (uint32_t*)(bn + BL.vals_alt);
        b.keys_alt = (uint64_t*)(bn + BL.keys_alt);
        // one counting-sort pass: the unsorted pairs sit in the "alt" buffers so that the result lands in (keys, point_list)
        b.keys_unsorted = b.keys_alt; b.vals_unsorted = b.vals_alt;
        b.hist = (uint32_t*)(bn + BL.hist);
        b.ranges = (uint2*)(f[k].img + IL.ranges);
        gs2d::BlendFwdFrame& o = f[k].bf;
        o.ranges = b.ranges; o.point_list = b.point_list; o.rec = (const float4*)(f[k].geom + GL.rec);
        o.out_color = f[k].out_color; o.out_others = f[k].out_others; o.pix_state = (float*)(f[k].img + IL.pix);
        o.hits = (uint8_t*)(bn + BL.hits); o.hits4 = (uint8_t*)(bn + BL.hits4);
        o.zero = (float4*)(f[k].geom + GL.grad_rec);
        o.keys = b.keys; o.keys_alt = b.keys_alt; o.vals_alt = b.vals_alt;
        o.dev = gs2d::DevBin{nullptr, nullptr, 0u, 0};
    }
    if (debug) GS2D_CHECK(hipStreamSynchronize(s), "duplicate (batch)");
    for (int k = K; k < GS2D_MAX_BATCH; k++) bin.f[k] = bin.f[0];
    if (redo) {  // (rare: first call for this shape, or a scene that grew by more than 12.5 %) all frames once more, nothing published
        gs2d::launch_duplicate_batch(P, K, cam0.gx, bin, s);
        GS2D_STAGE("duplicate (batch, second launch)");
    }
    long long max_R = 0;
    bool any_empty = false;
    for (int k = 0; k < K; k++) { max_R = f[k].R > max_R ? f[k].R : max_R; any_empty = any_empty || f[k].R == 0; }
    // (a frame without instances has no ranges to sort by: keep the stand-alone sort kernel for such a batch)
    const int sort_cap = any_empty ? 0 : gs2d::gs2d_fused_sort_cap(gs2d::tile_sort_capacity(max_R, IL.tiles));
    const bool fused_sort = sort_cap > 0;
    for (int k = 0; k < K; k++) { f[k].fused_sort = fused_sort; f[k].sort_cap = sort_cap; }
    g_timer.begin(ST_SORT, s);
    gs2d::launch_bin_sort_batch(P, K, IL.tiles, cam0.gx, tile_bits, bin, debug ? 1 : 0, /*depth_sort=*/!fused_sort, s);
    g_timer.end(ST_SORT, s);
    GS2D_STAGE("sort (batch)");
    return 0;
}

int fwd_phase_c(const FwdShared& c, FwdFrame* frames, int K)
{
    const int debug = c.debug;
    hipStream_t s = c.s;
    gs2d::BlendFwdFrame bf[GS2D_MAX_BATCH];
    for (int k = 0; k < K; k++) bf[k] = frames[k].bf;
    g_timer.begin(ST_BLEND_FWD, s);
    bool fused = true;  // (all frames or none: the batch paths decide for the batch as a whole)
    int fused_cap = 0;  // (any capacity sorts any list: longer ones go through global scratch)
    for (int k = 0; k < K; k++) { fused = fused && frames[k].fused_sort; fused_cap = frames[k].sort_cap > fused_cap ? frames[k].sort_cap : fused_cap; }
    gs2d::launch_blend_fwd(c.width, c.height, K, bf, c.background, c.use_sa, (size_t)c.P * (GS2D_GRAD_FLOATS / 4),
                           fused ? fused_cap : 0, debug ? 1 : 0, s);
    const bool det = g_deterministic.load() != 0;
    for (int k = 0; k < K; k++) {
        if (frames[k].ahead) continue;  // num_rendered not known yet: fwd_phase_d writes the record
        FwdRecord fr;
        fr.geom = frames[k].geom; fr.bin = frames[k].bin; fr.bin_bytes = frames[k].bin_bytes; fr.det = det ? 1 : 0;
        fr.R = frames[k].R; fr.P = c.P; fr.clean = true;
        g_fwd.add(fr);
    }
    g_timer.end(ST_BLEND_FWD, s);
    GS2D_STAGE("blend_fwd");
    return 0;
}

// Launch-ahead forward, last step: every kernel of the forward is enqueued; NOW the host looks at num_rendered (the word
// duplicate_kernel's last workgroup stored -- long there unless the GPU is backed up behind earlier work, in which case the
// wait costs the GPU nothing: its queue is full).  A count within the chunk's capacity: done, the record is written.  A count
// beyond it (first call of a shape, a scene that grew by more than 12.5 % between calls): the kernels behind duplicate saw the
// same number and touched nothing; duplicate, binning and blend run once more in a chunk of the exact size, in stream order.
int fwd_phase_d(const FwdShared& c, FwdFrame& f, int slot)
{
    hipStream_t s = c.s;
    if (!wait_total(f, s, /*may_nap=*/true)) return fail_msg("num_rendered read-back failed");
    const uint32_t num_rendered_u = *f.pinned;
    if (num_rendered_u == 0xFFFFFFFFu) return fail_msg("num_rendered read-back failed");
    if (num_rendered_u > 0x7fffffffu) return fail_msg("num_rendered overflows int32");
    const int R = (int)num_rendered_u;
    LastCount& last = g_last[slot];
    last.P = c.P; last.W = c.width; last.H = c.height; last.R = num_rendered_u;
    f.ahead = false;
    if (R <= f.cap) {
        f.R = R;
        f.bin_bytes = bin_layout(R, false, f.cap).total;
    } else {
        // once more with the count known: the round-3 order from duplicate on (fwd_phase_b does exactly that when the
        // launch-ahead switch is off; the pinned word is re-armed by it and written again by the new duplicate launch)
        // (last.R is the exact count now, so the second pass's chunk is large enough)
        if (fwd_phase_b(c, f, slot, /*allow_ahead=*/false) < 0) return -1;
        if (fwd_phase_c(c, &f, 1) < 0) return -1;
        return 0;  // (fwd_phase_c wrote the record)
    }
    FwdRecord fr;
    fr.geom = f.geom; fr.bin = f.bin; fr.bin_bytes = f.bin_bytes; fr.det = 0; fr.R = f.R; fr.P = c.P; fr.clean = true;
    g_fwd.add(fr);
    return 0;
}

}  // namespace

extern "C" {

int gs2d_forward_posed(gs2d_alloc_fn geometry_alloc, void* geometry_user, gs2d_alloc_fn binning_alloc, void* binning_user,
                       gs2d_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width,
                       int height, const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                       const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                       float* out_others, int* radii, int use_sa, int debug, const float* pose_Rt, const float* pose_quat,
                       void* stream)
{
    (void)tan_fovx; (void)tan_fovy; (void)prefiltered;  // unused by the reference forward kernels as well (forward.cu:165)
    FwdShared c;
    c.P = P; c.D = D; c.M = M; c.width = width; c.height = height; c.use_sa = use_sa; c.debug = debug;
    c.background = background; c.means3D = means3D; c.shs = shs; c.colors_precomp = colors_precomp; c.opacities = opacities;
    c.scales = scales; c.rotations = rotations; c.transMat_precomp = transMat_precomp; c.scale_modifier = scale_modifier;
    c.s = (hipStream_t)stream;
    if (pose_Rt == nullptr && pose_quat != nullptr) return fail_msg("pose_quat without pose_Rt");
    if (pose_Rt != nullptr && transMat_precomp != nullptr) return fail_msg("a pose cannot be combined with transMat_precomp");
    if (fwd_validate(c) < 0) return -1;
    if (P == 0) return 0;  // rasterize_points.cu:100-101: zero images, rendered = 0 (outputs pre-zeroed by the caller)
    // hipGetLastError is per host thread and keeps the last failure until it is read: drop whatever another library (or an
    // earlier, already reported call) left behind, so that the checks below only ever see this call's launches.  A device
    // that is truly lost fails the next launch again.
    (void)hipGetLastError();
    FwdFrame f;
    f.geometry_alloc = geometry_alloc; f.geometry_user = geometry_user; f.binning_alloc = binning_alloc;
    f.binning_user = binning_user; f.image_alloc = image_alloc; f.image_user = image_user;
    f.viewmatrix = viewmatrix; f.projmatrix = projmatrix; f.cam_pos = cam_pos; f.pose_Rt = pose_Rt; f.pose_quat = pose_quat;
    f.out_color = out_color; f.out_others = out_others; f.radii = radii;
    if (fwd_phase_a(c, f, 0) < 0) { (void)wait_total(f, c.s); return -1; }
    if (fwd_phase_b(c, f, 0) < 0) { (void)wait_total(f, c.s); return -1; }
    if (fwd_phase_c(c, &f, 1) < 0) { (void)wait_total(f, c.s); return -1; }
    if (f.ahead && fwd_phase_d(c, f, 0) < 0) { (void)wait_total(f, c.s); return -1; }
    return f.R;
}

int gs2d_forward_batch(int K, const gs2d_frame_io* io, int P, int D, int M, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                       const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
                       int use_sa, int debug, int* num_rendered, void* stream)
{
    if (K < 1 || K > GS2D_MAX_BATCH) return fail_msg("K must be between 1 and GS2D_MAX_FRAMES (8)");
    if (!io || !num_rendered) return fail_msg("null pointer");
    FwdShared c;
    c.P = P; c.D = D; c.M = M; c.width = width; c.height = height; c.use_sa = use_sa; c.debug = debug;
    c.background = background; c.means3D = means3D; c.shs = shs; c.colors_precomp = colors_precomp; c.opacities = opacities;
    c.scales = scales; c.rotations = rotations; c.transMat_precomp = transMat_precomp; c.scale_modifier = scale_modifier;
    c.s = (hipStream_t)stream;
    if (fwd_validate(c) < 0) return -1;
    for (int k = 0; k < K; k++) num_rendered[k] = 0;
    if (P == 0) return 0;
    (void)hipGetLastError();
    FwdFrame f[GS2D_MAX_BATCH];
    for (int k = 0; k < K; k++) {
        f[k].geometry_alloc = io[k].geometry_alloc; f[k].geometry_user = io[k].geometry_user;
        f[k].binning_alloc = io[k].binning_alloc; f[k].binning_user = io[k].binning_user;
        f[k].image_alloc = io[k].image_alloc; f[k].image_user = io[k].image_user;
        f[k].viewmatrix = io[k].viewmatrix; f[k].projmatrix = io[k].projmatrix; f[k].cam_pos = io[k].cam_pos;
        f[k].pose_Rt = nullptr; f[k].pose_quat = nullptr;
        f[k].out_color = io[k].out_color; f[k].out_others = io[k].out_others; f[k].radii = io[k].radii;
    }
    auto drain = [&]() { for (int k = 0; k < K; k++) (void)wait_total(f[k], c.s); };
    if (img_layout(width, height).tiles <= GS2D_BIN_MAX_TILES) {
        // every stage as ONE launch over the K frames (blockIdx.y = frame): 7 launches instead of 6 K + 1, and the
        // latency-bound stages (duplicate, tile histogram, row scan, scatter, per-tile depth sort) get K times the
        // workgroups for the same latency
        if (fwd_batch_fused(c, f, K) < 0) { drain(); return -1; }
    } else {
        // more tiles than the single-pass binning takes: frame by frame through the generic radix passes; all K preprocess +
        // scan pairs are still enqueued before the host looks at the first total
        for (int k = 0; k < K; k++)
            if (fwd_phase_a(c, f[k], k) < 0) { drain(); return -1; }
        for (int k = 0; k < K; k++)
            if (fwd_phase_b(c, f[k], k) < 0) { drain(); return -1; }
    }
    if (fwd_phase_c(c, f, K) < 0) return -1;
    for (int k = 0; k < K; k++) num_rendered[k] = f[k].R;
    return 0;
}

int gs2d_forward(gs2d_alloc_fn geometry_alloc, void* geometry_user, gs2d_alloc_fn binning_alloc, void* binning_user,
                 gs2d_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width,
                 int height, const float* means3D, const float* shs, const float* colors_precomp,
                 const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                 const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                 float* out_others, int* radii, int use_sa, int debug, void* stream)
{
    return gs2d_forward_posed(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M,
                              background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                              rotations, transMat_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
                              out_color, out_others, radii, use_sa, debug, nullptr, nullptr, stream);
}

int gs2d_backward_staged(int stages, int g_begin, int g_end, int P, int D, int M, int R, const float* background, int width,
                         int height, const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                         float scale_modifier, const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                         const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                         char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix,
                         const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor,
                         float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot, int use_sa,
                         int debug, const float* pose_Rt, const float* pose_quat, float* dL_dpose, void* stream)
{
    if ((stages & ~7) != 0 || (stages & 3) == 0)
        return fail_msg("stages must be a combination of GS2D_BWD_BLEND (1) and GS2D_BWD_PREPROCESS (2), optionally with GS2D_BWD_POSE_4X4 (4)");
    if ((stages & GS2D_BWD_POSE_4X4) != 0 && (stages & 3) != 3)
        return fail_msg("GS2D_BWD_POSE_4X4 needs both stages in one call");
    if (pose_Rt == nullptr && pose_quat != nullptr) return fail_msg("pose_quat without pose_Rt");
    if (pose_Rt != nullptr && dL_dpose == nullptr) return fail_msg("dL_dpose is required with a pose");
    if ((stages & 2) != 0) {   // pose-only call: all six per-Gaussian outputs NULL (needs a pose and no SH gradient), otherwise none of them
        const int nulls = !dL_dmean2D + !dL_dopacity + !dL_dcolor + !dL_dmean3D + !dL_dscale + !dL_drot;
        if (nulls != 0 && (nulls != 6 || pose_Rt == nullptr || shs != nullptr))
            return fail_msg("per-Gaussian gradient outputs may only be omitted all together, with a pose and without SH");
    }
    (void)colors_precomp; (void)transMat_precomp;
    hipStream_t s = (hipStream_t)stream;
    // The pose gradient is accumulated by the preprocess stage(s) and cleared once, by the call that runs the blend stage:
    // inside blend_bwd_kernel when that kernel runs (one launch less on the tracking loop's critical path), else by a memset.
    const int pose_floats = (stages & GS2D_BWD_POSE_4X4) != 0 ? 16 : 12;
    if (P <= 0) {
        if (dL_dpose != nullptr && (stages & 1) != 0)
            GS2D_CHECK(hipMemsetAsync(dL_dpose, 0, sizeof(float) * pose_floats, s), "memset dL_dpose");
        return 0;
    }
    if (g_begin < 0 || g_end > P || g_begin > g_end) return fail_msg("bad Gaussian range");
    if (!geom_buffer || !img_buffer || (R > 0 && !binning_buffer)) return fail_msg("missing forward state");
    // The mode the FORWARD ran in decides the layout of the binning chunk (see FwdTable): never the live flag alone.
    const bool live_det = g_deterministic.load() != 0;
    FwdRecord fr;
    const bool known = g_fwd.find(geom_buffer, /*take_clean=*/(stages & 1) != 0, &fr);
    if (known && fr.P != P) return fail_msg("P is not the Gaussian count of the forward that produced this geometry chunk");
    if ((stages & 1) != 0) {
        if (known) {
            if (fr.R != R) return fail_msg("R is not the num_rendered of the forward that produced this geometry chunk");
            if (R > 0 && fr.bin != (const void*)binning_buffer)
                return fail_msg("binning_buffer is not the chunk the forward of this geometry chunk was given");
            if ((fr.det != 0) != live_det)
                return fail_msg("gs2d_set_deterministic changed between this forward and its backward (the forward sizes the "
                                "binning chunk for the mode it runs in): restore the flag or rerun the forward");
        } else if (live_det) {
            return fail_msg("deterministic backward on a geometry chunk without a forward record (relocated chunk, or more "
                            "than 64 forwards in flight): its binning chunk cannot be vouched for");
        }
    }
    const bool det = known ? fr.det != 0 : false;
    // Pose-only fast path (tracking with every Gaussian parameter detached): both stages in this call, no per-Gaussian output
    // of any kind, non-deterministic mode.  The blend stage then accumulates only the three dL_dT components dL/dmean needs, in
    // a dense layout over the head of the gradient records, and a 40-B-per-Gaussian kernel reduces the pose gradient from them.
    const bool pose_fast = (stages & 3) == 3 && pose_Rt != nullptr && dL_dmean3D == nullptr && dL_dtransMat == nullptr &&
                           dL_dnormal == nullptr && dL_dsh == nullptr && shs == nullptr && scales != nullptr &&
                           rotations != nullptr && !det && !live_det;
    if (dL_dpose != nullptr && (stages & 1) != 0 && (R <= 0 || det))
        GS2D_CHECK(hipMemsetAsync(dL_dpose, 0, sizeof(float) * pose_floats, s), "memset dL_dpose");
    const GeomLayout GL = geom_layout(P);
    const BinLayout BL = bin_layout(R, det);
    if (known && (stages & 1) != 0 && R > 0 && BL.total > fr.bin_bytes)
        return fail_msg("binning chunk is smaller than the layout of this backward");
    const ImgLayout IL = img_layout(width, height);
    const float4* rec = (const float4*)(geom_buffer + GL.rec);
    const uint8_t* clamped = (const uint8_t*)(geom_buffer + GL.clamped);
    float* grad_rec = (float*)(geom_buffer + GL.grad_rec);
    const uint32_t* point_list = (const uint32_t*)(binning_buffer + BL.point_list);
    const uint8_t* hits = (const uint8_t*)(binning_buffer + BL.hits4);  // the backward reads the row bits (one byte per quadrant)
    const uint2* ranges = (const uint2*)(img_buffer + IL.ranges);
    const float* pix_state = (const float*)(img_buffer + IL.pix);

    if ((stages & 1) != 0 && !det) {
        if (!(known && fr.clean))  // else: cleared by the forward's blend kernel, nothing has touched it since
            GS2D_CHECK(hipMemsetAsync(grad_rec, 0, sizeof(float) * GS2D_GRAD_FLOATS * (size_t)P, s), "memset grad_rec");
        if (R > 0) {
            g_timer.begin(ST_BLEND_BWD, s);
            const gs2d::BlendBwdFrame bf = {ranges, point_list, rec, pix_state, hits, dL_dpix, dL_depths, grad_rec, nullptr,
                                            (const uint16_t*)(binning_buffer + BL.hits), pose_fast ? grad_rec + 4 * (size_t)P : nullptr};
            gs2d::launch_blend_bwd(width, height, 1, &bf, background, use_sa, dL_dpose, pose_floats, s);
            g_timer.end(ST_BLEND_BWD, s);
            GS2D_STAGE("blend_bwd");
        }
    }
    if ((stages & 1) != 0 && det) {
        // no atomics: per-(instance, quadrant) partial records, then a fixed-order sum per Gaussian (gs2d_det.hip)
        float* det_slots = (float*)(binning_buffer + BL.det_slots);
        uint32_t* det_inv = (uint32_t*)(binning_buffer + BL.det_inv);
        // (the record's clean token was taken above: det_reduce overwrites the accumulator, it is no longer "known zero")
        g_timer.begin(ST_BLEND_BWD, s);
        if (R > 0) {
            GS2D_CHECK(hipMemsetAsync(det_slots, 0, sizeof(float) * GS2D_GRAD_FLOATS * 4 * (size_t)R, s), "memset det_slots");
            const gs2d::BlendBwdFrame bf = {ranges, point_list, rec, pix_state, hits, dL_dpix, dL_depths, grad_rec, det_slots,
                                            (const uint16_t*)(binning_buffer + BL.hits), nullptr};
            gs2d::launch_blend_bwd(width, height, 1, &bf, background, use_sa, nullptr, 0, s);
        }
        gs2d::launch_det_reduce(P, R, width, height, ranges, point_list, (const ushort4*)(geom_buffer + GL.rect),
                                (const uint32_t*)(geom_buffer + GL.tiles_touched),
                                (const uint32_t*)(geom_buffer + GL.point_offsets), hits, det_inv, det_slots, grad_rec, s);
        g_timer.end(ST_BLEND_BWD, s);
        GS2D_STAGE("blend_bwd (deterministic)");
    }
    if ((stages & 2) != 0 && g_end > g_begin) {
        // rasterizer_impl.cu:396-397 + backward.cu:641-642: the backward rebuilds W,H from focal*tan in float32
        const float focal_y = height / (2.0f * tan_fovy);
        const float focal_x = width / (2.0f * tan_fovx);
        CamParams cam;
        cam.vm = viewmatrix; cam.pm = projmatrix; cam.campos = campos;
        cam.W = (int)(focal_x * tan_fovx * 2);
        cam.H = (int)(focal_y * tan_fovy * 2);
        cam.gx = (width + GS2D_TILE - 1) / GS2D_TILE;
        cam.gy = (height + GS2D_TILE - 1) / GS2D_TILE;
        cam.tight = 0;
        g_timer.begin(ST_PREPROCESS_BWD, s);
        if (pose_fast)
            gs2d::launch_preprocess_bwd_pose(P, means3D, radii, scales, rotations, cam, grad_rec, grad_rec + 4 * (size_t)P, pose_Rt,
                                             pose_quat, dL_dpose, s);
        else
        gs2d::launch_preprocess_bwd(g_begin, g_end, D, M, means3D, rec, radii, shs, clamped, scales, rotations, cam, grad_rec,
                                    dL_dtransMat, dL_dnormal, dL_dcolor, dL_dopacity, dL_dsh, dL_dmean2D, dL_dmean3D,
                                    dL_dscale, dL_drot, pose_Rt, pose_quat, pose_Rt ? dL_dpose : nullptr,
                                    /*need_record=*/scale_modifier != 1.0f,
                                    // deterministic mode: per-workgroup partials in the geometry chunk's depth array (4 B
                                    // per Gaussian, dead once the forward's duplicate stage has run; 48 B per 256 Gaussians needed)
                                    // (the mode of the FORWARD, like everything else this backward decides by mode)
                                    (det && pose_Rt) ? (float*)(geom_buffer + GL.depths) : nullptr, s);
        g_timer.end(ST_PREPROCESS_BWD, s);
        GS2D_STAGE("preprocess_bwd");
    }
    return 0;
}

int gs2d_backward_batch(int K, const gs2d_frame_grad* fr, int accumulate, int P, int D, int M, const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                        float scale_modifier, const float* rotations, const float* transMat_precomp, int use_sa, int debug,
                        void* stream)
{
    if (K < 1 || K > GS2D_MAX_BATCH) return fail_msg("K must be between 1 and GS2D_MAX_FRAMES (8)");
    if (!fr) return fail_msg("null pointer");
    if (g_deterministic.load() != 0) return fail_msg("the batched backward has no deterministic variant: call gs2d_backward per frame");
    if (P <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const GeomLayout GL = geom_layout(P);
    const ImgLayout IL = img_layout(width, height);
    gs2d::BlendBwdFrame bf[GS2D_MAX_BATCH];
    int nb = 0;  // frames with instances: only those take part in the blend launch
    for (int k = 0; k < K; k++) {
        const gs2d_frame_grad& f = fr[k];
        const int R = f.num_rendered;
        if (!f.geom_buffer || !f.img_buffer || (R > 0 && !f.binning_buffer)) return fail_msg("missing forward state");
        FwdRecord rec;
        const bool known = g_fwd.find(f.geom_buffer, /*take_clean=*/true, &rec);
        if (known) {
            if (rec.P != P) return fail_msg("P is not the Gaussian count of the forward that produced this geometry chunk");
            if (rec.R != R) return fail_msg("R is not the num_rendered of the forward that produced this geometry chunk");
            if (R > 0 && rec.bin != (const void*)f.binning_buffer)
                return fail_msg("binning_buffer is not the chunk the forward of this geometry chunk was given");
            if (rec.det != 0) return fail_msg("gs2d_set_deterministic changed between this forward and its backward");
        }
        float* grad_rec = (float*)(f.geom_buffer + GL.grad_rec);
        if (!(known && rec.clean))
            GS2D_CHECK(hipMemsetAsync(grad_rec, 0, sizeof(float) * GS2D_GRAD_FLOATS * (size_t)P, s), "memset grad_rec");
        if (R > 0) {
            const BinLayout BL = bin_layout(R, false);
            if (known && BL.total > rec.bin_bytes) return fail_msg("binning chunk is smaller than the layout of this backward");
            gs2d::BlendBwdFrame& b = bf[nb++];
            b.ranges = (const uint2*)(f.img_buffer + IL.ranges);
            b.point_list = (const uint32_t*)(f.binning_buffer + BL.point_list);
            b.rec = (const float4*)(f.geom_buffer + GL.rec);
            b.pix_state = (const float*)(f.img_buffer + IL.pix);
            b.hits = (const uint8_t*)(f.binning_buffer + BL.hits4);
            b.dL_dpix = f.dL_dpix; b.dL_dothers = f.dL_depths; b.grad_rec = grad_rec; b.det_slots = nullptr; b.dense_m2d = nullptr;
            b.hits16 = (const uint16_t*)(f.binning_buffer + BL.hits);
        }
    }
    if (nb > 0) {
        g_timer.begin(ST_BLEND_BWD, s);
        gs2d::launch_blend_bwd(width, height, nb, bf, background, use_sa, nullptr, 0, s);
        g_timer.end(ST_BLEND_BWD, s);
        GS2D_STAGE("blend_bwd (batch)");
    }
    {   // the per-Gaussian stage of all frames in one launch (each frame has its own outputs)
        gs2d::PreBwdFrames tab;
        for (int k = 0; k < K; k++) {
            const gs2d_frame_grad& f = fr[k];
            if (!f.dL_dmean2D || !f.dL_dopacity || !f.dL_dcolor || !f.dL_dmean3D || !f.dL_dscale || !f.dL_drot)
                return fail_msg("per-Gaussian gradient outputs are required for every frame of a batch");
            gs2d::PreBwdFrame& q = tab.f[k];
            q.vm = f.viewmatrix; q.pm = f.projmatrix; q.campos = f.campos;
            // rasterizer_impl.cu:396-397 + backward.cu:641-642: the backward rebuilds W,H from focal*tan in float32
            const float focal_y = height / (2.0f * f.tan_fovy), focal_x = width / (2.0f * f.tan_fovx);
            q.W = (int)(focal_x * f.tan_fovx * 2); q.H = (int)(focal_y * f.tan_fovy * 2);
            q.rec = (const float4*)(f.geom_buffer + GL.rec); q.radii = f.radii;
            q.clamped = (const uint8_t*)(f.geom_buffer + GL.clamped); q.grad_rec = (const float*)(f.geom_buffer + GL.grad_rec);
            q.dL_dtransMat = f.dL_dtransMat; q.dL_dnormal = f.dL_dnormal; q.dL_dcolor = f.dL_dcolor; q.dL_dopacity = f.dL_dopacity;
            q.dL_dsh = f.dL_dsh; q.dL_dmean2D = f.dL_dmean2D; q.dL_dmean3D = f.dL_dmean3D; q.dL_dscale = f.dL_dscale; q.dL_drot = f.dL_drot;
        }
        for (int k = K; k < GS2D_MAX_BATCH; k++) tab.f[k] = tab.f[0];
        g_timer.begin(ST_PREPROCESS_BWD, s);
        gs2d::launch_preprocess_bwd_batch(P, K, D, M, means3D, shs, scales, rotations, /*need_record=*/scale_modifier != 1.0f, tab, s);
        g_timer.end(ST_PREPROCESS_BWD, s);
        GS2D_STAGE("preprocess_bwd (batch)");
        if (accumulate && K > 1) {
            gs2d::launch_sum_frames(P, K, shs != nullptr ? M : 0, tab, s);
            GS2D_STAGE("sum over frames");
        }
    }
    (void)colors_precomp; (void)transMat_precomp;
    return 0;
}

int gs2d_backward_posed(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                        const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                        const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                        const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                        char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix,
                        const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor,
                        float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot, int use_sa,
                        int debug, const float* pose_Rt, const float* pose_quat, float* dL_dpose, void* stream)
{
    return gs2d_backward_staged(3, 0, P > 0 ? P : 0, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales,
                                scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy,
                                radii, geom_buffer, binning_buffer, img_buffer, dL_dpix, dL_depths, dL_dmean2D, dL_dnormal,
                                dL_dopacity, dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot, use_sa, debug, pose_Rt,
                                pose_quat, dL_dpose, stream);
}

int gs2d_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                  const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                  const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                  const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                  char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix,
                  const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor,
                  float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot, int use_sa,
                  int debug, void* stream)
{
    return gs2d_backward_posed(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier,
                               rotations, transMat_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii,
                               geom_buffer, binning_buffer, img_buffer, dL_dpix, dL_depths, dL_dmean2D, dL_dnormal,
                               dL_dopacity, dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot, use_sa, debug,
                               nullptr, nullptr, nullptr, stream);
}

int gs2d_pose_quat(const float* pose_Rt, float* quat_out, void* stream)
{
    if (!pose_Rt || !quat_out) return fail_msg("null pointer");
    gs2d::launch_pose_quat(pose_Rt, quat_out, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail_msg("pose_quat launch failed");
}

int gs2d_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                      void* stream)
{
    (void)projmatrix;
    hipStream_t s = (hipStream_t)stream;
    const int debug = 0;
    if (P <= 0) return 0;
    gs2d::launch_mark_visible(P, means3D, viewmatrix, present, s);
    GS2D_STAGE("mark_visible");
    return 0;
}

}  // extern "C"
