// ctgn_api.hip — the C ABI of libctgn.so (include/ctgn.h) over the gfx950 kernels in ctgn_kernels.hpp.
// Host side only: context, device residency of the voxel map (full / delta upload), keypoint staging, launch
// sequencing of the GN loop on one HIP stream. No CPU fallback exists for any query or solve entry point.
#include "../../include/ctgn.h"
#include "ctgn_internal.h"      // measurement / test hooks: exported, but not part of the contract

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types only: the library is bound at run time (dlopen), see rccl_api()

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ctgn_devmap.hpp"
#include "ctgn_hostpool.hpp"
#include "ctgn_kernels.hpp"
#include "ctgn_robust.hpp"

using namespace ctgn;

namespace {

struct DeviceLevel {
    Slot *slots = nullptr;
    double *blocks = nullptr;
    size_t slots_cap = 0;    // in slots
    size_t blocks_cap = 0;   // in doubles
    bool resident = false;
};

struct EventPair {
    hipEvent_t start = nullptr, stop = nullptr;
    int bounded = 0;         // the launch it brackets had a carried-over bound (not the first search of a solve)
};

}  // namespace

// phase counters (12, padded to 16) + {start, end, fast rounds, rounds} per wave of the last instrumented launch
constexpr size_t PROF_WORDS = 16 + 4 * (size_t) MAX_PARTIAL_BLOCKS * ROW_WAVES;

// doubles kept behind the 7 keypoint arrays: [0,16) the input pose of a ctgn_register call (it rides in with the keypoints),
// [16, ...) the final GnState (it rides out with the world points)
constexpr size_t KP_TAIL = 16 + (sizeof(GnState) + 7) / 8;
static_assert(sizeof(GnState) % 8 == 0 && sizeof(GnState) / 8 <= 256, "GnState is mirrored by one thread block as doubles");

// ---------------------------------------------------------------------------------------- tuning table
// The A/B switches of the measurement sessions in ONE place (round 5: they used to be 18 getenv calls scattered over this file). Process-wide,
// like the function-local statics they replace; set through ctgn_set_tuning (ctgn_internal.h) or, for a session script that cannot call into
// the library, ONE environment variable read once: CTGN_TUNING="key=value,key=value". Defaults = what the A/B sessions adopted. None of them
// changes WHICH neighbours a keypoint gets, which keypoints pass the gates, or any per-keypoint quantity; four of them select another
// (still fixed) ORDER in which the packed sums are added and therefore move a pose in its last bits (1e-12 relative on the system):
// order (position order instead of index order), xcd_reduce (per-XCD group sums), fuse_small (per-wave sums of the fused small-frame kernel),
// robust_fuse (the robust route's one-block evaluation — which ctgn_solve_robust also switches by size at 1 024 keypoints). With xcd_reduce
// the fall-back to the per-block sums depends on where the dispatcher placed the workgroups, so two runs of one process agree bit for bit
// only as long as the placement does (tests/test_gpu_parity.py::test_per_xcd_presums_change_nothing_but_the_summation_order pins the
// two sums to 1e-12 of each other). The switches whose A/B is closed (hipGraph capture of the loop, 512-thread residual blocks, the
// 4-wave solve block, the unfused frame call) are gone together with their code paths.
// Not synchronised: ctgn_set_tuning is for a measurement script or a test that owns the process, between calls — not for threads that
// are solving meanwhile. host_threads is latched by the first scan-sized frame call (the pool is sized then): setting it later is refused.
#ifndef CTGN_ROWS_ABL
#define CTGN_ROWS_ABL false        // the search kernel of a solve without an ablation mask is compiled without the mask's tests (true: as rounds 1-5, one instantiation)
#endif
#ifndef CTGN_ROWS_WPS
#define CTGN_ROWS_WPS 3            // waves per SIMD the default search-kernel instantiations are compiled for (A/B builds: 4 with CTGN_LCAP=80)
#endif
struct Tuning {
    double host_threads = 3;        // helper threads of the host-side staging loops (0 = none)
    double order = -1;              // home-voxel ordering when ctgn_set_ordering left it automatic: -1 = cost model, 0 / 1 = never / always
    double pool_min = 8192;         // neighbour pools from this many keypoints on (automatic mode)
    double res_small = -1;          // 64-thread residual blocks: -1 = up to 8 192 keypoints, 0 / 1 = never / always
    double res_grid_cap = 0;        // cap of the residual kernel's grid (0 = 3 blocks per CU)
    double guess_factor = 1.25;     // guessed first-search bound: factor on r_k ...
    double guess_maxfrac = 0.8;     // ... used when (factor r_k)^2 is below this fraction of the squared radius
    double split = -1;              // pool check as its own kernel: -1 = from 400 k keypoints, 0 / 1 = never / always
    double xcd_split = -1;          // one eighth of the tiles per XCD: -1 = incoherent ordered uploads only, 0 / 1
    double fuse_small = -1;         // k_search_residual for small frames: 1 = on (its sums run in another fixed order)
    double persistent = -1;         // overrides ctgn_set_persistent when >= 0
    double persist_times = 0;       // per-block timeline of the persistent kernel -> ctgn_wave_timeline
    double frame_timing = 0;        // host-clock marks of the frame pipeline on stderr
    double frame_no_direct = 0;     // always stage page-locked scan arrays
    double frame_defer_update = 1;  // ctgn_frame_update_map without an insert mask returns once the update is enqueued (0: waits for it)
    double stop_poll = 1;           // the host watches the solve's stop flag and does not enqueue the launches behind it (0: enqueues all num_iters_icp iterations)
    double robust_fuse = -1;        // robust route: evaluation + step in one launch (k_robust_eval_step): -1 = up to 1 024 keypoints, 0 / 1 = never / always
    double state_init_fused = 1;    // the solve's first search launch writes the GnState itself (0: k_state_init in front of it, as rounds 1-5)
    double stage_lds = 0;           // 1: the 125-voxel sweep streams a home-voxel group's candidates from LDS (GroupStage; prototype, use with tile_chunk 4-16)
    double tile_chunk = 0;          // consecutive rounds of a search tile that take consecutive positions: 0 = the default (1: rounds strided over the scan), else that many
    double xcd_reduce = -1;         // per-XCD pre-sums of the residual kernel's block records: -1 = automatic (from 128 block records on), 0 / 1 = never / always (1: from 32 on)
};
static double *tuning_slot(Tuning &t, const std::string &key) {
#define CTGN_TUNING_KEY(name) if (key == #name) return &t.name;
    CTGN_TUNING_KEY(host_threads) CTGN_TUNING_KEY(order) CTGN_TUNING_KEY(pool_min) CTGN_TUNING_KEY(res_small) CTGN_TUNING_KEY(res_grid_cap)
    CTGN_TUNING_KEY(guess_factor) CTGN_TUNING_KEY(guess_maxfrac) CTGN_TUNING_KEY(split) CTGN_TUNING_KEY(xcd_split) CTGN_TUNING_KEY(fuse_small)
    CTGN_TUNING_KEY(persistent) CTGN_TUNING_KEY(persist_times) CTGN_TUNING_KEY(frame_timing) CTGN_TUNING_KEY(frame_no_direct) CTGN_TUNING_KEY(frame_defer_update) CTGN_TUNING_KEY(stop_poll)
    CTGN_TUNING_KEY(tile_chunk) CTGN_TUNING_KEY(stage_lds) CTGN_TUNING_KEY(state_init_fused) CTGN_TUNING_KEY(xcd_reduce) CTGN_TUNING_KEY(robust_fuse)
#undef CTGN_TUNING_KEY
    return nullptr;
}
static Tuning &tuning() {
    static Tuning t = [] {
        Tuning v;
        if (const char *e = std::getenv("CTGN_TUNING")) {
            std::string all(e);
            size_t at = 0;
            while (at < all.size()) {
                const size_t end = std::min(all.find(',', at), all.size());
                const std::string item = all.substr(at, end - at);
                const size_t eq = item.find('=');
                if (eq != std::string::npos)
                    if (double *slot = tuning_slot(v, item.substr(0, eq))) *slot = std::atof(item.c_str() + eq + 1);
                at = end + 1;
            }
        }
        // the per-switch variables of rounds 1-4 are no longer read: say so once, instead of letting an old session script measure the same
        // configuration twice
        for (const char *legacy : {"CTGN_ORDER", "CTGN_GRAPH", "CTGN_POOL_MIN", "CTGN_PERSISTENT", "CTGN_HOST_THREADS", "CTGN_FRAME_TIMING",
                                   "CTGN_PERSIST_TIMES", "CTGN_RES_SMALL", "CTGN_SPLIT", "CTGN_XCD_SPLIT", "CTGN_FUSE_SMALL", "CTGN_GUESS_FACTOR"})
            if (std::getenv(legacy))
                std::fprintf(stderr, "[ctgn] %s is ignored: the measurement switches are CTGN_TUNING=\"key=value,...\" (ctgn_api.hip, struct Tuning)\n", legacy);
        return v;
    }();
    return t;
}

// how many helpers: tuning().host_threads (0 = none, default 3); never more than the CPUs this process may run on minus the caller's
static bool g_host_threads_latched = false;
static int host_helpers_wanted() {
    static const int n = [] {
        g_host_threads_latched = true;
        int want = (int) tuning().host_threads;
        cpu_set_t set;
        int cpus = 1;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        return std::max(0, std::min(want, cpus - 1));
    }();
    return n;
}

struct ctgn_context {
    int device = -1;                    // -1: host-only map mirror, every device entry point fails
    int num_cus = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    ctgn_map_options opts{};
    std::vector<VoxelLevel> levels;
    std::vector<DeviceLevel> dlevels;
    int update_mode = 0;                // 0: host mirror + delta upload, 1: device-resident maintenance (ctgn_devmap)
    std::vector<DevLevel> devlevels;
    DevMapScratch dm;
    OrderScratch ord;                   // keypoint order of the neighbour-search kernel (sorted by home voxel), see order_keypoints
    bool order_valid = false, order_stale = true;
    bool kp_coherent = false;           // the upload's own order is spatially coherent (scan order): consecutive keypoints mostly share or
                                        // neighbour their home voxel; an incoherent one (shuffled, config D) is what ordering is for
    bool kp_presorted = false;          // the resident keypoints ARE in home-voxel order (ctgn_set_keypoints_sharded): nothing to sort, and
                                        // the tiles are dealt per XCD like an ordered upload's
    int planned_iters = 0;              // iteration budget of the running solve (num_iters_icp)
    int ordering_mode = -1;             // ctgn_set_ordering: -1 automatic, 0 never, 1 always
    double *d_kp_sorted = nullptr;      // the 7 keypoint arrays in position order (working copy of the GN kernels when ordered)
    size_t sorted_cap = 0;

    // keypoints: one device allocation holding 7 arrays [rx ry rz t wx wy wz] of kp_stride (= n rounded up to 64) doubles
    // back to back, so that one copy moves the whole set (and one copy brings the three world arrays back)
    int n_kp = 0, cap_kp = 0, kp_stride = 0;
    bool prefetch_world = false;        // ctgn_register*: bring the world points back with the final state, one sync
    const double *pose_with_kp = nullptr;   // ctgn_register: pose to append to the keypoint upload (one copy instead of two)
    bool pose_on_device = false;            // ... and it has been uploaded behind the keypoint arrays
    double *d_kp = nullptr;
    uint32_t *d_res = nullptr;          // [cap_kp][SEL_STRIDE] row-phase -> lane-phase hand-over records
    double *h_kp = nullptr;             // pinned staging, same layout
    double t_min = 0, t_max = 0;
    bool keep_world0 = false;           // ctgn_set_rewind: every upload leaves a copy of its world arrays in d_world0
    bool world0_valid = false;          // ... and the copy belongs to the resident keypoints
    double *d_world0 = nullptr;         // [3][kp_stride]
    size_t world0_cap = 0;              // in doubles

    // undistortion staging (ctgn_transform_points): 7 x tp_cap doubles on the device, 4 x tp_cap pinned
    double *d_tp = nullptr, *h_tp = nullptr;
    size_t tp_cap = 0;
    std::vector<hipEvent_t> tp_events;  // per 32 k-point chunk of a host-view ctgn_transform_points: its result has arrived | its kernel is done
    hipStream_t stream_down = nullptr;  // results travel back on their own stream, beside the uploads of the chunks behind them
    hipEvent_t ev_frame = nullptr;      // frame pipeline: the undistorted frame is ready (the outputs' stream waits for it)
    HostPool pool;                      // helper threads of the scan-sized host loops (created on first use)

    // solver
    GnState *d_state = nullptr;
    GnState *h_state = nullptr;         // pinned
    double *d_sys = nullptr;            // packed system in use (own buffer or caller-owned)
    double *d_sys_own = nullptr;
    double *d_partials = nullptr;
    double *d_pose_in = nullptr;
    double *h_pose_in = nullptr;        // pinned
    bool pose_in_valid = false;         // d_pose_in[0..14) holds (in stream order) what h_pose_in[0..14) holds
    GnParams prm{};
    ctgn_options gn_opts{};
    int launched_iters = 0;
    // persistent small-frame kernel (k_gn_persistent): barrier counters (two slots, alternating per launch), enabled unless a barrier timed out
    unsigned int *d_bar = nullptr;
    int persist_slot = 0;
    bool persist_disabled = false;
    int persist_mode = 0;               // ctgn_set_persistent: 0 = the three-launch loop (default), 1 = one persistent launch for small frames
    bool world_final_done = false;      // the running solve's last launch already re-transformed the keypoints and mirrored the state
    // state initialisation of a solve is deferred to its first launch (the persistent kernel does it in its prologue)
    bool init_pending = false;
    const double *init_pose = nullptr;
    double init_tbe[2] = {0.0, 0.0};
    bool kth_fresh = false;             // the k-th distances on the device were written by the previous search of this solve
    int searches_in_solve = 0;          // neighbour searches launched since the solve began (the first one has no carried-over bound)
    int last_grid = 0;
    uint64_t path_counts[2] = {0, 0};   // ctgn_path_counters: residual launches with per-XCD pre-sums | the last solve launch summed group records (0 / 1)
    XcdReduce xr_last{nullptr, nullptr, 0u};   // what the last residual launch was given (ctl == nullptr: per-block records only): the solve launch behind it gets the same
    unsigned int xr_epoch = 0;          // launch number of the per-XCD pre-sums (never 0 on the device)
    double guess_factor = -1.0;         // ctgn_set_search_guess: < 0 automatic, 0 off, > 0 forced factor on r_k (launch_accumulate)
    int normals_mode = 0;               // ctgn_set_normals: 0 library default (hybrid), 1 exact, 2 hybrid, 3 fast
    int fail_slot = 0;                  // which of the two fail-list counters the next split launch counts in (the other one is zeroed by it)
    uint64_t last_upload_bytes = 0;     // host-to-device bytes of the last ctgn_set_keypoints_sharded call on this rank (bench detail)
    int rb_split_search[2] = {0, 0};    // resident blocks of the split path's search launch (27- / 125-voxel instantiation)
    bool stage_ok = false;            // the GroupStage instantiation of the search kernel got its 80 KB of dynamic LDS
    int rb_rows[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // resident blocks of k_accumulate_rows by (sweep, variant)
    int rb_check = 0, rb_fused[2] = {0, 0}, per_cu_persist[2] = {0, 0};   // the other occupancy queries, cached per handle (= per device) too
    bool fail_reset_pending = true;     // a new solve began: its first split launch zeroes both counters and starts from slot 0 (a solve that
                                        // stops early leaves the slot its last EXECUTED launch counted in non-zero: the launches behind the stop
                                        // test return before they zero anything, while the host keeps toggling the slot)
    // the host enqueues a solve's iterations ahead of the device; the solve kernel publishes "iteration i done / stopped" in host memory it can
    // write, the host polls two iterations behind and stops enqueueing launches that would only find the stop flag set (launch_stop_poll)
    unsigned int *h_stop = nullptr, *d_stop = nullptr;     // 64 words, page-locked + mapped
    unsigned int stop_epoch = 0;
    bool stop_poll_broken = false;      // a poll timed out once (the mapping is not coherent here?): the handle stops polling
    bool gn_active = false;
    std::chrono::steady_clock::time_point gn_t0;
    double init_ms = 0.0;               // host time from the call to the first launch (ICPSummary::duration_init)
    hipEvent_t ev_loop_start = nullptr, ev_loop_stop = nullptr;

    // debug / introspection
    bool debug = false;
    int dbg_cap = 0;
    int *d_nnb = nullptr;
    double *d_normal = nullptr, *d_a2d = nullptr, *d_far = nullptr;
    uint8_t *d_used = nullptr;
    Counters *d_counters = nullptr;

    // edit staging
    void *d_edit = nullptr;
    void *h_edit = nullptr;             // pinned
    size_t edit_cap = 0;

    // kernel timing
    bool profiling = false;
    std::vector<EventPair> events;
    int events_used = 0;
    int events_base = 0;                // events of earlier solves that were begun but not ended (all their launches did work)
    double acc_ms = 0.0;
    int acc_launches = 0;
    double acc_ms_split[2] = {0.0, 0.0};      // [0] first search of a solve (radius only), [1] searches with a carried-over bound
    int acc_launches_split[2] = {0, 0};

    // robust-loss route (ctgn_robust.hpp)
    RobustState *d_rstate = nullptr;
    RobustState *h_rstate = nullptr;    // pinned
    double *d_rbuf = nullptr;           // (5 + 3 K) x rb_cap doubles
    int *d_rrank = nullptr;
    int rb_cap = 0, rb_k = 0;
    RobustParams rprm{};
    ctgn_robust_options r_opts{};

    // keypoint-sharded mode (ctgn_dist_init): this rank's RCCL communicator
    ncclComm_t comm = nullptr;
    int dist_rank = 0, dist_world = 1;

    // frame pipeline (ctgn_frame_register / ctgn_frame_update_map): the scan and its undistorted images stay on the device
    struct FrameScratch {
        double *d_scan = nullptr;       // 16 doubles: the initial pose, then n records x y z t in processing order
        double *d_world = nullptr;      // [3][stride] every point undistorted with the final poses
        double *d_corr = nullptr;       // [3][stride] the sampled frame undistorted (compact: num_sampled points)
        uint8_t *d_flag1 = nullptr, *d_flag2 = nullptr;
        uint32_t *d_sel1 = nullptr, *d_sel2 = nullptr;
        int *d_counts = nullptr;
        double *h_scan = nullptr;       // pinned, same layout
        double *h_out = nullptr;        // pinned [6][cap]: world | corrected
        uint32_t *h_sel = nullptr;      // pinned [2][cap]
        int *h_counts = nullptr;        // pinned
        size_t cap = 0, stride = 0, n = 0, n1 = 0, n2 = 0;
        bool valid = false;             // a registered frame is resident (ctgn_frame_update_map may insert it)
        // the step-by-step form (ctgn_frame_begin / _try_register / _undistort): what the later steps need from the earlier ones
        bool staged = false;            // ctgn_frame_begin left a sampled scan on the device
        bool direct_in = false;         // ... read in place from page-locked caller arrays (h_scan holds no records then)
        double tmin = 0, tmax = 0;      // timestamp range of the staged scan
        double frame_voxel = 0, kp_voxel = 0;   // voxel sizes the two samplers last ran with (d_sel1 / d_sel2 belong to them)
        // device-side shuffle (ctgn_frame_options::shuffle_seed)
        double *d_scan_in = nullptr;    // n records in scan order, as uploaded; k_frame_permute writes d_scan from them
        uint32_t *d_order = nullptr;    // order[j] = scan index of the point at processing position j
        uint32_t *h_order = nullptr;    // pinned staging of a caller's order on its way to the device
        uint32_t *d_selx = nullptr;     // [2][cap] the sampled / keypoint positions translated to the caller's point numbers (k_frame_translate)
        bool device_shuffled = false;
        bool permuted = false;          // d_scan holds the records in another order than the caller's: d_order[j] = caller index of position j
        bool prestaged = false;         // ctgn_frame_stage uploaded a scan (scan order, d_scan_in) that ctgn_frame_begin has yet to take
    } fr;

    int res_grid_cap = MAX_PARTIAL_BLOCKS;             // blocks of k_residual_reduce = per-block partials the solve kernel has to sum
    int pool_mode = -1;                 // ctgn_set_pools: -1 automatic, 0 never, 1 always
    int ablate = 0;                     // measurement hook: bit mask of kernel phases to skip (results become invalid)
    int variant = 0;                    // 0 rows+hist, 1 lane, 2 rows without hist, 3 rows+hist with phase clocks
    unsigned long long *d_prof = nullptr;
    bool map_update_pending = false;       // ctgn_frame_update_map enqueued an update whose counters are still to be read (NEED_DEVICE does)
    uint64_t insert_calls_with_skips = 0;  // insert CALLS that met points outside the voxel key range / non-finite (those points: skipped, inserted = 0)
    std::string last_error;
};

static ctgn_status frame_finish_map_update(ctgn_handle h);

namespace {

// RCCL entry points, bound on first use with dlopen("librccl.so.1"): a process that already holds RCCL (torch.distributed's
// "nccl" backend ships a librccl with the same SONAME) shares that instance, a plain C++ host gets the ROCm one; and libctgn.so
// keeps loading on a box without RCCL as long as nobody asks for the sharded mode.
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
const RcclApi &rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
        void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(lib, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
        return a;
    }();
    return api;
}

const char *status_str(ctgn_status s) {
    switch (s) {
        case CTGN_OK: return "ok";
        case CTGN_ERR_INVALID_ARGUMENT: return "invalid argument";
        case CTGN_ERR_NO_DEVICE: return "no gfx950 HIP device (libctgn has no CPU fallback)";
        case CTGN_ERR_HIP: return "HIP runtime error";
        case CTGN_ERR_OUT_OF_MEMORY: return "out of memory";
        case CTGN_ERR_TIMESTAMP_RANGE: return "The timestamp cannot be interpolated between the two poses";
        case CTGN_ERR_VOXEL_RANGE: return "voxel coordinate out of range";
        case CTGN_ERR_UNSUPPORTED: return "unsupported configuration";
        case CTGN_ERR_SOLVER: return "Error During Optimization";
    }
    return "unknown";
}

ctgn_status fail(ctgn_handle h, ctgn_status s, const std::string &msg) {
    if (h) h->last_error = std::string(status_str(s)) + ": " + msg;
    return s;
}

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(h, e_ == hipErrorOutOfMemory ? CTGN_ERR_OUT_OF_MEMORY : CTGN_ERR_HIP,        \
                        std::string("[HIP] ") + #call + " -> " + hipGetErrorString(e_));             \
    } while (0)

#define DMCHK(h, call)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(h, e_ == hipErrorOutOfMemory ? CTGN_ERR_OUT_OF_MEMORY : CTGN_ERR_HIP,         \
                        std::string("[HIP] device map: ") + hipGetErrorString(e_));                   \
    } while (0)

// (a map update ctgn_frame_update_map left in flight is completed — its counters read, its errors reported — by whichever entry point
// comes next: frame_finish_map_update, declared in front of this namespace)
#define NEED_DEVICE(h)                                                                               \
    do {                                                                                             \
        if (!(h)) return CTGN_ERR_INVALID_ARGUMENT;                                                  \
        if ((h)->device < 0) return fail(h, CTGN_ERR_NO_DEVICE, "context was created host-only");    \
        HIPCHK(h, hipSetDevice((h)->device));                                                        \
        if ((h)->map_update_pending) {                                                               \
            const ctgn_status fs_ = frame_finish_map_update(h);                                      \
            if (fs_ != CTGN_OK) return fs_;                                                          \
        }                                                                                            \
    } while (0)

inline double read_elem(const void *base, size_t stride, ctgn_dtype dt, size_t i, int c) {
    const char *p = static_cast<const char *>(base) + i * stride;
    return dt == CTGN_F64 ? reinterpret_cast<const double *>(p)[c] : (double) reinterpret_cast<const float *>(p)[c];
}

// Stage n points of a host or device view as the device map's batch: SoA planes `S.stride` = n rounded up to 64 apart, so that a
// host view travels in ONE copy (three separate copies cost more than the bytes they move).
ctgn_status stage_batch(ctgn_handle h, const void *base, size_t stride, ctgn_dtype dt, size_t n);

// A view / output pointer may address device memory (e.g. a torch CUDA tensor): then nothing is staged through the host.
bool on_device(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void) hipGetLastError(); return false; }     // plain host memory
    return a.type == hipMemoryTypeDevice;
}

// host memory the runtime can DMA from / to as it lies (hipHostMalloc, hipHostRegister, a framework's pinned allocator): both ends of
// the range belong to a page-locked allocation
bool host_pinned(const void *p, size_t bytes) {
    if (!p || !bytes) return false;
    auto pinned = [](const void *q) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void) hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const void *last = static_cast<const char *>(p) + bytes - 1;
    if (!(pinned(p) && pinned(last))) return false;
    // both ends page-locked is not yet "the whole range": two separate page-locked allocations with pageable memory between them would
    // pass. Where the runtime can name the allocation an address belongs to, both ends must name the same one.
    hipDeviceptr_t b0 = nullptr, b1 = nullptr;
    size_t s0 = 0, s1 = 0;
    const bool k0 = hipMemGetAddressRange(&b0, &s0, const_cast<void *>(p)) == hipSuccess;
    const bool k1 = hipMemGetAddressRange(&b1, &s1, const_cast<void *>(last)) == hipSuccess;
    if (!(k0 && k1)) { (void) hipGetLastError(); return k0 == k1; }      // not answerable for this kind of memory (both ends alike): the two-ends test stands
    return b0 == b1 && s0 == s1;
}

inline int grid_for(size_t n) { return (int) std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 4096)); }

// device view -> SoA doubles on the handle's stream
ctgn_status gather_device_view(ctgn_handle h, const void *base, size_t stride, ctgn_dtype dt, int ncomp, double *dst, size_t dst_stride,
                               size_t n) {
    hipLaunchKernelGGL(k_view_gather, dim3(grid_for(n)), dim3(256), 0, h->stream, static_cast<const char *>(base), stride,
                       dt == CTGN_F64 ? 1 : 0, ncomp, dst, dst_stride, (int) n);
    HIPCHK(h, hipGetLastError());
    return CTGN_OK;
}

ctgn_status scatter_device_view(ctgn_handle h, const double *src, size_t src_stride, int ncomp, void *base, size_t stride, ctgn_dtype dt,
                                size_t n) {
    hipLaunchKernelGGL(k_view_scatter, dim3(grid_for(n)), dim3(256), 0, h->stream, src, src_stride, ncomp, static_cast<char *>(base),
                       stride, dt == CTGN_F64 ? 1 : 0, (int) n);
    HIPCHK(h, hipGetLastError());
    return CTGN_OK;
}

ctgn_status stage_batch(ctgn_handle h, const void *base, size_t stride, ctgn_dtype dt, size_t n) {
    DevMapScratch &S = h->dm;
    S.stride = std::min((n + 63) & ~(size_t) 63, S.cap);
    if (on_device(base)) return gather_device_view(h, base, stride, dt, 3, S.pts, S.stride, n);
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) S.h_pts[a * S.stride + i] = read_elem(base, stride, dt, i, a);
    HIPCHK(h, hipMemcpyAsync(S.pts, S.h_pts, (2 * S.stride + n) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return CTGN_OK;
}

// min / max of n device doubles -> host (synchronises)
ctgn_status device_minmax(ctgn_handle h, const double *d_t, size_t n, double *lo, double *hi) {
    hipLaunchKernelGGL(k_minmax, dim3(1), dim3(1024), 0, h->stream, d_t, (int) n, h->d_pose_in + 14);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->h_pose_in + 14, h->d_pose_in + 14, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *lo = h->h_pose_in[14];
    *hi = h->h_pose_in[15];
    return CTGN_OK;
}

ctgn_status ensure_edit_buffer(ctgn_handle h, size_t bytes) {
    if (bytes <= h->edit_cap) return CTGN_OK;
    size_t cap = std::max<size_t>(bytes * 2, 1 << 20);
    if (h->d_edit) HIPCHK(h, hipFree(h->d_edit));
    if (h->h_edit) HIPCHK(h, hipHostFree(h->h_edit));
    h->d_edit = nullptr; h->h_edit = nullptr; h->edit_cap = 0;
    HIPCHK(h, hipMalloc(&h->d_edit, cap));
    HIPCHK(h, hipHostMalloc(&h->h_edit, cap, hipHostMallocDefault));
    h->edit_cap = cap;
    return CTGN_OK;
}

// Bring level `li` up to date on the device: full copy after (re)allocation / rehash, else scatter the edit log.
ctgn_status sync_level(ctgn_handle h, int li) {
    VoxelLevel &L = h->levels[li];
    DeviceLevel &D = h->dlevels[li];
    const size_t nslots = (size_t) L.mask + 1;
    const size_t nblk_doubles = (size_t) L.nblocks_cap * 3 * L.blk;
    if ((size_t) L.nblocks_used * 3 * L.blk * sizeof(double) >= ((size_t) 1 << 32) || (size_t) L.nblocks_used >= ((size_t) 1 << 25))
        return fail(h, CTGN_ERR_UNSUPPORTED, "point blocks of one resolution exceed 4 GiB or 2^25 blocks (32-bit block offsets / packed block index in the kernels)");
    bool full = L.need_full_upload || !D.resident;
    if (!full && L.slot_edits.size() * 4 > nslots) full = true;
    if (full) {
        if (D.slots_cap < nslots) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (D.slots) HIPCHK(h, hipFree(D.slots));
            D.slots = nullptr; D.slots_cap = 0;
            HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&D.slots), nslots * sizeof(Slot)));
            D.slots_cap = nslots;
        }
        if (D.blocks_cap < nblk_doubles) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (D.blocks) HIPCHK(h, hipFree(D.blocks));
            D.blocks = nullptr; D.blocks_cap = 0;
            HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&D.blocks), nblk_doubles * sizeof(double)));
            D.blocks_cap = nblk_doubles;
        }
        HIPCHK(h, hipMemcpyAsync(D.slots, L.slots.data(), nslots * sizeof(Slot), hipMemcpyHostToDevice, h->stream));
        const size_t used_doubles = (size_t) L.nblocks_used * 3 * L.blk;
        if (used_doubles)
            HIPCHK(h, hipMemcpyAsync(D.blocks, L.blocks.data(), used_doubles * sizeof(double), hipMemcpyHostToDevice,
                                     h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));     // the source vectors are pageable and may change next
    } else if (!L.slot_edits.empty() || !L.point_edits.empty()) {
        // delta: re-read the CURRENT host value of every logged location, so duplicates in the log agree
        const size_t ns = L.slot_edits.size(), np = L.point_edits.size();
        const size_t bytes = ns * sizeof(SlotEdit) + np * sizeof(PointEdit);
        HIPCHK(h, hipStreamSynchronize(h->stream));     // previous use of the staging buffer
        ctgn_status st = ensure_edit_buffer(h, bytes);
        if (st != CTGN_OK) return st;
        SlotEdit *hs = static_cast<SlotEdit *>(h->h_edit);
        PointEdit *hp = reinterpret_cast<PointEdit *>(hs + ns);
        for (size_t i = 0; i < ns; ++i) {
            hs[i].slot = L.slot_edits[i].slot;
            hs[i]._pad = 0;
            hs[i].value = L.slots[L.slot_edits[i].slot];
        }
        for (size_t i = 0; i < np; ++i) {
            const PointEdit &e = L.point_edits[i];
            const double *bx = L.bx(e.block);
            hp[i] = PointEdit{e.block, e.index, bx[3 * e.index], bx[3 * e.index + 1], bx[3 * e.index + 2]};
        }
        HIPCHK(h, hipMemcpyAsync(h->d_edit, h->h_edit, bytes, hipMemcpyHostToDevice, h->stream));
        SlotEdit *ds = static_cast<SlotEdit *>(h->d_edit);
        PointEdit *dp = reinterpret_cast<PointEdit *>(ds + ns);
        if (np) hipLaunchKernelGGL(k_scatter_points, dim3((unsigned) ((np + 255) / 256)), dim3(256), 0, h->stream, D.blocks,
                                   L.blk, dp, (int) np);
        if (ns) hipLaunchKernelGGL(k_scatter_slots, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, h->stream, D.slots,
                                   ds, (int) ns);
        HIPCHK(h, hipGetLastError());
    }
    L.slot_edits.clear();
    L.point_edits.clear();
    L.need_full_upload = false;
    L.log_edits = true;
    D.resident = true;
    return CTGN_OK;
}

ctgn_status make_map_view(ctgn_handle h, double radius, MapView *mv) {
    if (radius <= 0) radius = h->opts.default_radius;
    int map_id, nb;
    double res;
    search_params(h->levels, radius, &map_id, &res, &nb);
    if (h->update_mode == 1) {
        const DevLevel &DL = h->devlevels[map_id];
        if ((size_t) DL.host.next_block * 3 * DL.blk * sizeof(double) >= ((size_t) 1 << 32) || (size_t) DL.host.next_block >= ((size_t) 1 << 25))
            return fail(h, CTGN_ERR_UNSUPPORTED, "point blocks of one resolution exceed 4 GiB or 2^25 blocks (32-bit block offsets / packed block index in the kernels)");
        mv->slots = DL.slots;
        mv->blocks = DL.blocks;
        mv->mask = (uint32_t) (DL.slots_cap - 1);
        mv->blk = DL.blk;
        mv->nb = nb;
        mv->resolution = res;
        mv->r2thr = radius_sq_threshold(radius);
        return CTGN_OK;
    }
    ctgn_status st = sync_level(h, map_id);
    if (st != CTGN_OK) return st;
    const VoxelLevel &L = h->levels[map_id];
    const DeviceLevel &D = h->dlevels[map_id];
    mv->slots = D.slots;
    mv->blocks = D.blocks;
    mv->mask = L.mask;
    mv->blk = L.blk;
    mv->nb = nb;
    mv->resolution = res;
    mv->r2thr = radius_sq_threshold(radius);
    return CTGN_OK;
}

// Should the GN kernels work through this upload in home-voxel order (sorted positions + position-ordered working copy)?
//  * the searched level exceeds the caches: yes, it pays at once — neighbouring waves then share their voxels in L2 instead of
//    each fetching them from HBM (config D, 1 M keypoints over a 0.4 GB level: 5.3 -> 3.6 ms per iteration for a 0.2 ms sort);
//  * the level sits in the caches: the order still puts more rounds on the shared-home fast path and makes the residual
//    kernel's gathers coherent, worth ~5e-5 us per keypoint and iteration (6-9 us at 132 k), against a sort + permute of
//    ~60 us + 1.5e-4 us per keypoint (seven short launches). Ordered when the caller's iteration budget covers that — a
//    132 k-keypoint scan from 13 iterations on; a loop that stops early on its threshold has then paid ~80 us for nothing.
// Never below 32 k keypoints. ctgn_set_ordering (or CTGN_TUNING="order=0" / "order=1" in the environment, for whole test-suite runs) forces it
// off / on.
bool want_order(ctgn_handle h, uint64_t level_points) {
    const int env_forced = (int) tuning().order;
    const int forced = h->ordering_mode >= 0 ? h->ordering_mode : env_forced;
    if (forced >= 0) return forced != 0 && h->n_kp > 0;
    if (h->n_kp < 32768) return false;
    // a level that exceeds the caches: order an upload whose own order has no locality (config D: 3.35 -> 2.57 ms per launch); a
    // sweep in firing order already walks the map coherently and gains nothing (B2 over the 270 MB map: 0.144 vs 0.145 ms)
    if (level_points * 24ull >= (128ull << 20) && !h->kp_coherent) return true;
    const double n = (double) h->n_kp;
    return (double) h->planned_iters * n * 5e-5 > 60.0 + 1.5e-4 * n;
}

// buffers of the ordered mode, sized for the current keypoint capacity; called at upload time for scans that may be ordered so
// that no allocation lands inside a solve
ctgn_status order_reserve(ctgn_handle h) {
    DMCHK(h, order_scratch_reserve(h->ord, (size_t) h->n_kp));
    if (h->sorted_cap < 7 * (size_t) h->kp_stride) {
        if (h->d_kp_sorted) HIPCHK(h, hipFree(h->d_kp_sorted));
        h->d_kp_sorted = nullptr; h->sorted_cap = 0;
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_kp_sorted), 7 * (size_t) h->cap_kp * sizeof(double)));
        h->sorted_cap = 7 * (size_t) h->cap_kp;
    }
    return CTGN_OK;
}

// first accumulate launch after an upload: sort the positions by home voxel at the search resolution, or decide not to
ctgn_status order_keypoints(ctgn_handle h, const MapView &mv) {
    h->order_stale = false;
    h->order_valid = false;
    int map_id, nb;
    double res;
    search_params(h->levels, h->opts.default_radius, &map_id, &res, &nb);
    const uint64_t level_points = h->update_mode == 1 ? h->devlevels[map_id].host.num_points : h->levels[map_id].num_points;
    if (h->kp_presorted || !want_order(h, level_points)) return CTGN_OK;
    const size_t c = (size_t) h->kp_stride;
    DMCHK(h, order_by_home_voxel(h->ord, h->d_kp + 4 * c, h->d_kp + 5 * c, h->d_kp + 6 * c, (size_t) h->n_kp, mv.resolution, h->stream));
    ctgn_status rs = order_reserve(h);
    if (rs != CTGN_OK) return rs;
    hipLaunchKernelGGL(k_kp_permute, dim3(grid_for((size_t) h->n_kp)), dim3(256), 0, h->stream, h->d_kp, c, h->ord.order, h->n_kp,
                       h->d_kp_sorted, c);
    HIPCHK(h, hipGetLastError());
    h->order_valid = true;
    return CTGN_OK;
}

// `working`: the view the GN kernels iterate on. When the upload was ordered and nobody reads per-keypoint results by index
// (debug capture, the robust route's kernels), that is the position-ordered copy with plain indexing — the hand-over records are
// then indexed by position too; otherwise the caller-order arrays, through `order` if the upload was ordered.
KpView kp_view(ctgn_handle h, bool working = false) {
    KpView v;
    const size_t c = (size_t) h->kp_stride;
    const bool sorted = working && h->order_valid && !h->debug;
    double *base = sorted ? h->d_kp_sorted : h->d_kp;
    v.rx = base; v.ry = base + c; v.rz = base + 2 * c; v.t = base + 3 * c;
    v.wx = base + 4 * c; v.wy = base + 5 * c; v.wz = base + 6 * c;
    v.sel = h->d_res;
    v.cnt = h->d_res + (size_t) h->cap_kp * SEL_STRIDE;
    v.kth = reinterpret_cast<float *>(h->d_res + (size_t) h->cap_kp * (SEL_STRIDE + 1));
    v.kth_valid = 0;
    // Pools pay where the search is throughput-bound: a pool check is an extra dependent phase in front of the searches that remain, and a
    // frame of a few thousand keypoints (a handful of waves per CU) is bound by exactly that chain (B1 / C: +3 % with pools, B2 -7 %, D -29 %).
    const int env_pool_min = (int) tuning().pool_min;
    v.pools = h->pool_mode >= 0 ? h->pool_mode : (h->n_kp >= env_pool_min ? 1 : 0);
    v.n = h->n_kp;
    v.order = (h->order_valid && !sorted) ? h->ord.order : nullptr;
    v.chunk = tuning().tile_chunk > 0 ? std::min((int) tuning().tile_chunk, 16) : 1;     // consecutive rounds per chunk, ordered B2 at sustained clocks: 1 / 2 / 3 -> 0.965 / 0.952 / 0.923 of the accounting    // measured on B2 (ordered): chunk 1 / 2 / 4 -> 0.81 / 0.83 / 0.76 of the accounting
    v.xcd_split = 0;                    // set per launch (needs the grid size)
    v.clk_iter_start = nullptr;         // set by launch_accumulate for the launch that opens an iteration
    v.fail_list = h->d_res + (size_t) h->cap_kp * (SEL_STRIDE + 3);
    v.fail_count = v.fail_count_next = nullptr;       // set per split launch
    v.n_dev = nullptr;
    v.resume = 0;
    v.guess2 = 0.f;                     // set per launch (launch_accumulate: first searches over a dense level)
    return v;
}

DebugView dbg_view(ctgn_handle h) {
    DebugView d{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (h->debug) { d.n_nb = h->d_nnb; d.normal = h->d_normal; d.a2d = h->d_a2d; d.farthest = h->d_far; d.used = h->d_used; }
    return d;
}

ctgn_status ensure_debug(ctgn_handle h) {
    if (!h->debug || h->dbg_cap >= h->cap_kp) return CTGN_OK;
    if (h->d_nnb) { hipFree(h->d_nnb); hipFree(h->d_normal); hipFree(h->d_a2d); hipFree(h->d_far); hipFree(h->d_used); }
    h->d_nnb = nullptr; h->dbg_cap = 0;
    const size_t c = (size_t) h->cap_kp;
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_nnb), c * sizeof(int)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_normal), c * 3 * sizeof(double)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_a2d), c * sizeof(double)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_far), c * 3 * sizeof(double)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_used), c));
    h->dbg_cap = h->cap_kp;
    return CTGN_OK;
}

// Tile shape of the row kernel: a wave owns 4 x rounds keypoints per tile. Pick `rounds` (1..16) so that the
// resident waves of the chip get equal work: with W resident waves, a wave runs ceil(tiles / W) tiles of
// (rounds + overhead) steps; minimise that. Large scans end up with 64-keypoint tiles, a 1-3 k keypoint frame with
// 4-8 keypoint tiles spread over all 256 CUs.
int pick_rounds(int n, int resident_waves) {
    int best = 16;
    double best_cost = 1e300;
    for (int r = 1; r <= 16; ++r) {
        const long long tiles = ((long long) n + 4 * r - 1) / (4 * r);
        const long long per_wave = (tiles + resident_waves - 1) / resident_waves;
        const double cost = (double) per_wave * (r + 2.0);      // ~2 rounds' worth of per-tile phases A, C, D
        if (cost < best_cost - 1e-9) { best_cost = cost; best = r; }
    }
    return best;
}

template <typename K>
int resident_blocks(ctgn_handle h, K kernel, int block, size_t smem) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, smem) != hipSuccess || per_cu < 1) per_cu = 1;
    return per_cu * h->num_cus;
}

// k_residual_reduce: 256-thread blocks for throughput, 64-thread blocks for small frames (the scattered gathers are bound by the
// per-CU texture path, so a small frame wants MORE CUs, not fuller ones). Returns the grid = number of per-block partials.
int launch_residual(ctgn_handle h, const MapView &mv, const KpView &kv, const DebugView &dv) {
    const int env_small = (int) tuning().res_small;
    const bool small = env_small >= 0 ? env_small != 0 : h->n_kp <= 8192;
    if (small) {
        const int grid = std::max(1, std::min((h->n_kp + 63) / 64, h->res_grid_cap));
        h->xr_last = XcdReduce{nullptr, nullptr, 0u};
        hipLaunchKernelGGL(k_residual_reduce<64>, dim3(grid), dim3(64), 0, h->stream, mv, kv, h->d_state, h->prm, h->d_partials, dv, h->ablate, h->xr_last);
        return grid;
    }
    // (512-thread blocks would halve the per-block records the solve kernel sums — 259 instead of 518 on a 132 k-keypoint sweep — but 259
    // blocks of 8 waves need a second round on 3 CUs at 3 waves per SIMD: the kernel lost 5 us where the solve kernel gained 3; removed)
    const int tiles256 = (h->n_kp + RES_BLOCK - 1) / RES_BLOCK;
    const int env_cap = (int) tuning().res_grid_cap;
    // no more blocks than are resident at once (3 per CU): a larger scan's blocks take several tiles each, and the solve kernel has 768
    // records to sum instead of 2 048 (config D: 0.6931 -> 0.6858 ms per iteration; 1 024: 0.6872, 1 536: 0.6885)
    const int grid = std::max(1, std::min(tiles256, env_cap > 0 ? env_cap : std::min(h->res_grid_cap, 3 * h->num_cus)));
    // per-XCD pre-sums of the block records (XcdReduce, ctgn_kernels.hpp) once there are enough records for the solve kernel's own
    // reduction to show: B2 (518 records) solve kernel 13.7 -> 8.3 us, residual kernel + 1.5 us, -4 us per iteration; D (768) 19.1 -> 10.7 us
    const int env_xr = (int) tuning().xcd_reduce;
    const bool xr_on = env_xr >= 0 ? (env_xr != 0 && grid >= XCD_GROUPS) : grid >= 4 * XCD_GROUPS;
    if (xr_on) {
        if (++h->xr_epoch == 0u) h->xr_epoch = 1u;
        h->path_counts[0]++;
        double *rec = h->d_partials + (size_t) MAX_PARTIAL_BLOCKS * SYS_N;
        h->xr_last = XcdReduce{reinterpret_cast<unsigned int *>(rec + XCD_GROUPS * SYS_N), rec, h->xr_epoch};
    } else {
        h->xr_last = XcdReduce{nullptr, nullptr, 0u};
    }
    hipLaunchKernelGGL(k_residual_reduce<RES_BLOCK>, dim3(grid), dim3(RES_BLOCK), 0, h->stream, mv, kv, h->d_state, h->prm, h->d_partials, dv,
                       h->ablate, h->xr_last);
    return grid;
}

ctgn_status flush_state_init(ctgn_handle h);

ctgn_status launch_accumulate(ctgn_handle h, const MapView &mv, bool first_iter, bool search_only = false) {
    // The state initialisation gn_begin deferred rides INSIDE the solve's first search launch when that is the row kernel (StateInit,
    // ctgn_kernels.hpp: the first search reads nothing of the state); every other first launch gets k_state_init in front of it.
    StateInit state_init{nullptr, 0.0, 0.0};
    if (h->init_pending && first_iter && !search_only && h->variant != 1 && (mv.nb == 1 || mv.nb == 2) && mv.blk <= 64 &&
        (int) tuning().fuse_small != 1 && tuning().state_init_fused != 0) {
        state_init = StateInit{h->init_pose, h->init_tbe[0], h->init_tbe[1]};
        h->init_pending = false;
    } else {
        ctgn_status fs = flush_state_init(h);
        if (fs != CTGN_OK) return fs;
    }
    if (h->order_stale) {
        ctgn_status os = order_keypoints(h, mv);
        if (os != CTGN_OK) return os;
    }
    KpView kv = kp_view(h, !search_only);
    h->xr_last = XcdReduce{nullptr, nullptr, 0u};      // set again by launch_residual when the block records get per-XCD pre-sums
    if (!search_only) kv.clk_iter_start = &h->d_state->clk_iter_start;
    // the previous search's k-th distances bound this one — only inside one solve (same keypoints, same map, world points untouched)
    kv.kth_valid = (!first_iter && h->searches_in_solve > 0 && h->kth_fresh) ? 1 : 0;
    h->searches_in_solve++;
    h->kth_fresh = false;               // set again below by the kernel that writes them
    DebugView dv = dbg_view(h);
    EventPair *ev = nullptr;
    if (h->profiling) {
        if (h->events_used == (int) h->events.size()) {
            EventPair p;
            HIPCHK(h, hipEventCreate(&p.start));
            HIPCHK(h, hipEventCreate(&p.stop));
            h->events.push_back(p);
        }
        ev = &h->events[h->events_used++];
        ev->bounded = kv.kth_valid;
        HIPCHK(h, hipEventRecord(ev->start, h->stream));
    }
    int grid;
    const bool rows_ok = (mv.nb == 1 || mv.nb == 2) && mv.blk <= 64;
    // A search without a carried-over bound (the first of a solve) over a level that is DENSE for its radius starts from a guess: with
    // ppv points per voxel on a surface, k neighbours fill a disc of radius r_k = res sqrt(k / (pi ppv)); when 1.25 r_k is inside
    // the radius the search admits and streams what lies within that instead of everything within the radius (config D: ~45 candidates
    // instead of ~175, an eighth of the sweep's volume), and the few keypoints the guess fails — fewer than k candidates inside it — are
    // searched again on the radius in the same launch (rows_tiles, pass 1). Exact either way. Bit 24 of the ablation mask: off.
    // (config D, step ms by factor: off 0.844 | 0.9: 0.936 | 1.05: 0.806 | 1.15: 0.766 | 1.3: 0.777 | 1.6: 0.796 | 2.0: 0.828;
    //  B2, first search ms: off 0.133 | 1.1: 0.127 | 1.2: 0.116 | 1.25: 0.112 | 1.4: 0.113)
    if (rows_ok && !kv.kth_valid && !(h->ablate & (1 << 24)) && h->variant != 1) {
        const double env_factor = tuning().guess_factor;
        const bool forced = h->guess_factor > 0.0;            // ctgn_set_search_guess: a test forces guesses that mostly fail
        const double factor = h->guess_factor >= 0.0 ? h->guess_factor : env_factor;
        int map_id, nb_;
        double res_;
        search_params(h->levels, h->opts.default_radius, &map_id, &res_, &nb_);
        const double npts = h->update_mode == 1 ? (double) h->devlevels[map_id].host.num_points : (double) h->levels[map_id].num_points;
        const double nvox = h->update_mode == 1 ? (double) h->devlevels[map_id].host.num_voxels : (double) h->levels[map_id].num_voxels;
        if (factor > 0.0 && nvox > 0.0 && npts > 0.0) {
            const double ppv = npts / nvox;
            const double g = factor * mv.resolution * std::sqrt((double) h->prm.max_nb / (3.14159265358979323846 * ppv));
            const double env_frac = tuning().guess_maxfrac;
            if (g * g < (forced ? 1.0 : env_frac) * mv.r2thr) kv.guess2 = (float) (g * g * (1.0 + 1e-6));
        }
    }
    // From the third search of a solve on (nearly) every keypoint has a pool: the pool check runs as a kernel of its own and the search
    // kernel only over the list of positions it could not certify (k_pool_check). Bit 19 of the ablation mask switches the split off (A/B),
    // bit 25 forces it on below the size threshold (tests).
    // (measured: the 132 k-keypoint sweep loses 25 us per launch to the split — its pool check is bound by the scattered gathers of the pool
    // members, which the fused kernel overlaps with the search rounds of the waves that are done checking; config D, whose 14 k waves queue
    // anyway, gains 2.6 % per step)
    constexpr int SPLIT_MIN_KEYPOINTS = 400000;
    constexpr int FUSE_SMALL_MAX = 4096;
    const int env_split = (int) tuning().split;
    const bool split = rows_ok && h->variant == 0 && kv.kth_valid && kv.pools && h->searches_in_solve >= 3 && kv.order == nullptr &&
                       h->prm.max_nb + 1 <= KMAX && (h->ablate & 0xffff) == 0 && (env_split >= 0 ? env_split != 0 : ((h->n_kp >= SPLIT_MIN_KEYPOINTS || (h->ablate & (1 << 25))) && !(h->ablate & (1 << 19))));
    if (search_only && (h->variant == 1 || !rows_ok))
        return fail(h, CTGN_ERR_UNSUPPORTED, "the robust route needs the row kernel: voxel_neighborhood 1 or 2, <= 64 points per voxel");
    if (h->variant == 1 || !rows_ok) {
        const int ntiles = (h->n_kp + LANE_BLOCK - 1) / LANE_BLOCK;
        grid = std::max(1, std::min(ntiles, MAX_PARTIAL_BLOCKS));
        hipLaunchKernelGGL(k_accumulate_lane, dim3(grid), dim3(LANE_BLOCK), lane_kernel_smem(), h->stream, mv, kv,
                           h->d_state, h->prm, h->d_partials, dv, first_iter ? 1 : 0);
    } else {
        // one instantiation per (sweep half-width, selection flavour, instrumentation, waves per SIMD)
        int rb_slot = (mv.nb == 1 ? 0 : 8) + (h->variant & 7);
        auto launch = [&](auto kernel, size_t smem, unsigned long long *prof) {
            int &rb = h->rb_rows[rb_slot];        // one occupancy query per handle and instantiation (it used to run on every launch)
            if (rb == 0) rb = std::min(resident_blocks(h, kernel, ROW_BLOCK, smem), MAX_PARTIAL_BLOCKS);
            const int rounds = pick_rounds(h->n_kp, rb * ROW_WAVES);
            const int ntiles = (h->n_kp + 4 * rounds - 1) / (4 * rounds);
            const int g1 = std::max(1, std::min((ntiles + ROW_WAVES - 1) / ROW_WAVES, rb));
            const int env_xcd = (int) tuning().xcd_split;
            // one contiguous eighth of the tiles per XCD: +4 % on config D (uniformly spread keypoints), -50 % on a sweep whose density
            // varies along the sort key (the eighths then differ in work: B2 over the 270 MB map 0.145 -> 0.215 ms) -> incoherent uploads only
            kv.xcd_split = (env_xcd >= 0 ? env_xcd != 0 : ((h->order_valid && !h->kp_coherent) || h->kp_presorted)) && g1 >= 64 ? 1 : 0;
            hipLaunchKernelGGL(kernel, dim3(g1), dim3(ROW_BLOCK), smem, h->stream, mv, kv, h->d_state, h->prm, h->d_partials,
                               dv, first_iter ? 1 : 0, rounds, prof, h->ablate, state_init);
            h->kth_fresh = true;
            if (ev) (void) hipEventRecord(ev->stop, h->stream);        // the HIP-event pair brackets the neighbour-search kernel
            ev = nullptr;
            if (search_only) { grid = g1; return; }
            // second half: lane per keypoint (neighbour sets -> normal, residual, Jacobian, packed block sums)
            grid = launch_residual(h, mv, kv, dv);
        };
        // small frames: search + residual part in ONE launch (k_search_residual) — opt-in (CTGN_FUSE_SMALL=1), measured and not adopted:
        // B1 (1 024 keypoints, 27-voxel sweep) 0.0327 -> 0.0343 ms per iteration, C (1 500 keypoints, 125-voxel sweep) 0.0385 -> 0.0366 ms;
        // its packed sums run per wave and search block, a different (fixed) order than the residual kernel's, so switching it by size
        // would make poses depend on the path in the last bit. Not under per-launch profiling: the event pair brackets the SEARCH kernel.
        const int env_fuse = (int) tuning().fuse_small;
        const bool fuse_small = !search_only && h->variant == 0 && h->n_kp <= FUSE_SMALL_MAX && !kv.pools && kv.order == nullptr && !h->profiling &&
                                (h->ablate & 0xffff) == 0 && env_fuse == 1;
        if (fuse_small) {
            auto go = [&](auto kernel, size_t smem) {
                int &rb = h->rb_fused[mv.nb == 1 ? 0 : 1];
                if (rb == 0) rb = std::min(resident_blocks(h, kernel, ROW_BLOCK, smem), MAX_PARTIAL_BLOCKS);
                const int rounds = pick_rounds(h->n_kp, rb * ROW_WAVES);
                const int ntiles = (h->n_kp + 4 * rounds - 1) / (4 * rounds);
                grid = std::max(1, std::min((ntiles + ROW_WAVES - 1) / ROW_WAVES, rb));
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(ROW_BLOCK), smem, h->stream, mv, kv, h->d_state, h->prm, h->d_partials, dv, first_iter ? 1 : 0, rounds);
            };
            if (mv.nb == 1) go(k_search_residual<1>, search_residual_smem<1>());
            else go(k_search_residual<2>, search_residual_smem<2>());
            h->kth_fresh = true;
            if (ev) (void) hipEventRecord(ev->stop, h->stream);
            ev = nullptr;
        } else if (split) {
            int &rb_check = h->rb_check;
            if (rb_check == 0) rb_check = std::min(resident_blocks(h, k_pool_check<true, CHECK_WPS>, ROW_BLOCK, sizeof(CheckScratch) * ROW_WAVES), 4 * MAX_PARTIAL_BLOCKS);
            const int rounds_c = pick_rounds(h->n_kp, rb_check * ROW_WAVES);
            const int ntiles_c = (h->n_kp + 4 * rounds_c - 1) / (4 * rounds_c);
            int *counters = reinterpret_cast<int *>(h->d_res + (size_t) h->cap_kp * (SEL_STRIDE + 4));
            if (h->fail_reset_pending) {
                HIPCHK(h, hipMemsetAsync(counters, 0, 8 * sizeof(int), h->stream));
                h->fail_slot = 0;
                h->fail_reset_pending = false;
            }
            kv.fail_count = counters + 4 * h->fail_slot;
            kv.fail_count_next = counters + 4 * (h->fail_slot ^ 1);
            h->fail_slot ^= 1;
            hipLaunchKernelGGL((k_pool_check<true, CHECK_WPS>), dim3(std::max(1, std::min((ntiles_c + ROW_WAVES - 1) / ROW_WAVES, rb_check))), dim3(ROW_BLOCK),
                               sizeof(CheckScratch) * ROW_WAVES, h->stream, mv, kv, h->d_state, h->prm, rounds_c, mv.nb);
            KpView ks = kv;
            ks.order = kv.fail_list;
            ks.n_dev = kv.fail_count;
            ks.resume = 1;
            ks.clk_iter_start = nullptr;
            auto search = [&](auto kernel, size_t smem) {
                int &rb = h->rb_split_search[mv.nb == 1 ? 0 : 1];                 // one occupancy query per handle and instantiation
                if (rb == 0) rb = std::min(resident_blocks(h, kernel, ROW_BLOCK, smem), MAX_PARTIAL_BLOCKS);
                hipLaunchKernelGGL(kernel, dim3(rb), dim3(ROW_BLOCK), smem, h->stream, mv, ks, h->d_state, h->prm, h->d_partials, dv, 0, 1,
                                   (unsigned long long *) nullptr, h->ablate, StateInit{nullptr, 0.0, 0.0});
            };
            if (h->ablate != 0) {                 // (bits above the low sixteen: the mask is honoured)
                if (mv.nb == 1) search(k_accumulate_rows<1, true, false, CTGN_ROWS_WPS>, rows_kernel_smem<1>());
                else search(k_accumulate_rows<2, true, false, CTGN_ROWS_WPS>, rows_kernel_smem<2>());
            }
            else if (mv.nb == 1) search(k_accumulate_rows<1, true, false, CTGN_ROWS_WPS, false, true, false, CTGN_ROWS_ABL>, rows_kernel_smem<1>());
            else search(k_accumulate_rows<2, true, false, CTGN_ROWS_WPS, false, true, false, CTGN_ROWS_ABL>, rows_kernel_smem<2>());
            h->kth_fresh = true;
            if (ev) (void) hipEventRecord(ev->stop, h->stream);
            ev = nullptr;
            kv.xcd_split = ((h->order_valid && !h->kp_coherent) || h->kp_presorted) ? 1 : 0;      // the residual kernel's tiles per XCD, as below
            if (search_only) grid = 1;
            else grid = launch_residual(h, mv, kv, dv);
        } else if (mv.nb == 1) {
            const size_t sm = rows_kernel_smem<1>();
            switch (h->variant) {
                case 2: launch(k_accumulate_rows<1, false, false, 3, false, false>, sm, nullptr); break;
                case 3: launch(k_accumulate_rows<1, true, true, 3>, sm, h->d_prof); break;
                case 4: launch(k_accumulate_rows<1, true, false, 4, false, false>, sm, nullptr); break;
                case 5: launch(k_accumulate_rows<1, true, false, 3, true, false>, sm, nullptr); break;      // with the shared-home-voxel path (A/B hook)
                default:
                    if (h->ablate == 0) { rb_slot = 6; launch(k_accumulate_rows<1, true, false, CTGN_ROWS_WPS, false, true, false, CTGN_ROWS_ABL>, sm, nullptr); }
                    else launch(k_accumulate_rows<1, true, false, CTGN_ROWS_WPS>, sm, nullptr);
                    break;
            }
        } else {
            const size_t sm = rows_kernel_smem<2>();
            if (h->variant == 2) launch(k_accumulate_rows<2, false, false, 3, false, false>, sm, nullptr);
            else if (h->variant == 3) launch(k_accumulate_rows<2, true, true, 3>, sm, h->d_prof);
            else if (h->variant == 0 && tuning().stage_lds != 0 && h->stage_ok) {
                rb_slot = 8 + 6;           // (its own occupancy entry)
                launch(k_accumulate_rows<2, true, false, 2, false, true, true>, sm + ROW_WAVES * sizeof(GroupStage), h->d_prof);
            }
            else if (h->ablate == 0 && h->variant == 0) { rb_slot = 8 + 7; launch(k_accumulate_rows<2, true, false, CTGN_ROWS_WPS, false, true, false, CTGN_ROWS_ABL>, sm, nullptr); }
            else launch(k_accumulate_rows<2, true, false, CTGN_ROWS_WPS>, sm, nullptr);
        }
    }
    HIPCHK(h, hipGetLastError());
    if (ev) HIPCHK(h, hipEventRecord(ev->stop, h->stream));
    h->last_grid = grid;
    return CTGN_OK;
}

// the state initialisation gn_begin deferred: launched in front of the first kernel that needs d_state (the persistent kernel does it itself)
ctgn_status flush_state_init(ctgn_handle h) {
    if (!h->init_pending) return CTGN_OK;
    h->init_pending = false;
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(64), 0, h->stream, h->d_state, h->init_pose, h->init_tbe[0], h->init_tbe[1]);
    HIPCHK(h, hipGetLastError());
    return CTGN_OK;
}

// Is this solve one for the persistent kernel? Small frames on the default row kernel, nothing that needs per-launch events, per-position
// ordering or kernel variants; never after a barrier of it has timed out on this handle.
bool persistent_ok(ctgn_handle h, const MapView &mv) {
    const int env = (int) tuning().persistent;      // measurement hook: forces it on / off
    const int mode = env >= 0 ? env : h->persist_mode;
    if (mode != 1 || h->persist_disabled || h->n_kp < 1 || h->n_kp > 4096) return false;
    if (h->variant != 0 || h->profiling || h->ablate || h->ordering_mode == 1 || h->order_valid || h->kp_presorted) return false;
    if (!((mv.nb == 1 || mv.nb == 2) && mv.blk <= 64)) return false;
    // every keypoint must get its own row in ONE round on the one XCD (2 blocks of 4 waves per CU, 32 CUs: 1024 keypoints): with a
    // second round per wave the three-launch loop, which spreads the rounds over all eight XCDs, is the faster one (measured on the
    // NCLT profile, 1500 keypoints x 125 voxels: 55 us per iteration persistent against 43 us)
    return (h->n_kp + 3) / 4 <= 2 * std::max(1, h->num_cus / 8) * ROW_WAVES;
}

// `iters` whole GN iterations in ONE launch (k_gn_persistent). final_transform: also the re-transform of gn_end and the state mirror.
ctgn_status launch_persistent(ctgn_handle h, const MapView &mv, int iters, bool final_transform, double *state_copy) {
    if (h->order_stale) {                       // n <= 4096 is never ordered automatically; the call settles the flags
        ctgn_status os = order_keypoints(h, mv);
        if (os != CTGN_OK) return os;
    }
    const bool env_times = tuning().persist_times != 0;      // measurement hook: per-block timeline -> ctgn_wave_timeline
    auto launch = [&](auto kernel, size_t smem) -> ctgn_status {
        int &per_cu = h->per_cu_persist[mv.nb == 1 ? 0 : 1];
        if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, ROW_BLOCK, smem) != hipSuccess || per_cu < 1)) per_cu = 1;
        const int mmax = std::min(per_cu * std::max(1, h->num_cus / 8), MAX_PARTIAL_BLOCKS);       // co-resident blocks on one XCD
        int rounds = 1;
        if ((h->n_kp + 3) / 4 > mmax * ROW_WAVES) rounds = pick_rounds(h->n_kp, mmax * ROW_WAVES);
        const int ntiles = (h->n_kp + 4 * rounds - 1) / (4 * rounds);
        const int nblk = std::max(1, std::min(mmax, (ntiles + ROW_WAVES - 1) / ROW_WAVES));
        KpView kv = kp_view(h, false);
        const int kth_valid0 = (h->launched_iters > 0 && h->searches_in_solve > 0 && h->kth_fresh) ? 1 : 0;
        const int init = h->init_pending ? 1 : 0;
        h->init_pending = false;
        hipLaunchKernelGGL(kernel, dim3(8 * nblk), dim3(ROW_BLOCK), smem, h->stream, mv, kv, h->d_state, h->init_pose ? h->init_pose : h->d_pose_in,
                           h->init_tbe[0], h->init_tbe[1], init, h->prm, h->d_partials, dbg_view(h), iters, h->launched_iters, kth_valid0, rounds, nblk,
                           h->d_bar, h->persist_slot, state_copy, CTGN_MIN_KEYPOINTS_USED, final_transform ? 1 : 0, h->d_sys,
                           env_times ? h->d_prof + 16 : (unsigned long long *) nullptr);
        HIPCHK(h, hipGetLastError());
        h->persist_slot ^= 1;
        h->searches_in_solve += iters;
        h->launched_iters += iters;
        h->kth_fresh = true;
        h->last_grid = nblk;
        h->world_final_done = final_transform;
        return CTGN_OK;
    };
    return mv.nb == 1 ? launch(k_gn_persistent<1>, persistent_kernel_smem<1>()) : launch(k_gn_persistent<2>, persistent_kernel_smem<2>());
}

ctgn_status launch_reduce_solve(ctgn_handle h, int mode, int stop_slot = -1) {
    // (a 4-wave block for <= 128 partial columns was measured slower on the B1 frame — 0.0425 vs 0.0404 ms per iteration: the reduce is one
    // trip to 96 freshly written lines, and four waves have a quarter of the loads in flight — and removed)
    {
        ctgn_status fs = flush_state_init(h);
        if (fs != CTGN_OK) return fs;
    }
    unsigned int *flag = (stop_slot >= 0 && h->d_stop) ? h->d_stop + (stop_slot & 63) : nullptr;
    hipLaunchKernelGGL(k_reduce_solve<SOLVE_BLOCK>, dim3(1), dim3(SOLVE_BLOCK), 0, h->stream, h->d_partials, h->last_grid, h->d_sys, h->d_state,
                       h->prm, mode, CTGN_MIN_KEYPOINTS_USED, h->xr_last, flag, h->stop_epoch << 2);
    HIPCHK(h, hipGetLastError());
    return CTGN_OK;
}

// The three-launch GN loop of a solve: iterations enqueued ahead of the device, with the stop flag watched from the host. A solve that
// converges after two of five iterations (the driving profile's threshold: ||x|| < 0.1) used to enqueue the other nine launches all the
// same — each returns at once on the device, and each still costs a dispatch (profiles/r06_register_hip_api_trace.txt: ~30 us of a
// 140 us Register). Before enqueueing iteration i (i >= 2) the host reads what the solve kernel of iteration i - 2 published: the device is
// at least one whole iteration behind the host at that point, so the wait is short and the queue never runs dry.
ctgn_status launch_gn_iterations(ctgn_handle h, const MapView &mv, int iterations) {
    const bool poll = tuning().stop_poll != 0 && !h->stop_poll_broken && h->d_stop != nullptr && !h->profiling;
    if (poll) {
        h->stop_epoch = (h->stop_epoch + 1u) & 0x3fffffffu;
        if (h->stop_epoch == 0u) h->stop_epoch = 1u;
    }
    ctgn_status st = CTGN_OK;
    for (int it = 0; st == CTGN_OK && it < iterations; ++it) {                  // ct_icp.cpp:745
        if (poll && it >= 2) {
            const volatile unsigned int *f = h->h_stop + ((it - 2) & 63);
            const auto t0 = std::chrono::steady_clock::now();
            unsigned int v;
            int spins = 0;
            while (((v = __atomic_load_n(f, __ATOMIC_ACQUIRE)) >> 2) != h->stop_epoch) {
                if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { h->stop_poll_broken = true; break; }
                __builtin_ia32_pause();
            }
            if (!h->stop_poll_broken && (v & 3u) == 2u) break;                  // stopped (converged or failed): nothing left to enqueue
        }
        st = launch_accumulate(h, mv, it == 0);
        if (st == CTGN_OK) { h->launched_iters++; st = launch_reduce_solve(h, 0, poll ? it : -1); }
    }
    return st;
}

void fill_params(ctgn_handle h, const ctgn_options *o, const ctgn_motion_prior *p) {
    h->gn_opts = *o;
    GnParams &g = h->prm;
    g.min_nb = o->min_number_neighbors;
    g.max_nb = o->max_number_neighbors;
    g.max_dist = o->max_dist_to_plane_ct_icp;
    g.thr_norm = o->threshold_orientation_norm;
    g.has_prior = p ? 1 : 0;
    g.beta_c = p ? p->beta_location_consistency : 0.0;
    g.beta_e = p ? p->beta_constant_velocity : 0.0;
    for (int c = 0; c < 3; ++c) {
        g.prev_b[c] = p ? p->previous_begin_tr[c] : 0.0;
        g.prev_e[c] = p ? p->previous_end_tr[c] : 0.0;
    }
    g.normals = h->normals_mode;
}

// Fold the HIP-event times of the accumulate launches that did real work into the running average.
void harvest_events(ctgn_handle h, int real_launches) {
    for (int i = 0; i < h->events_used; ++i) {
        if (i < real_launches) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->events[i].start, h->events[i].stop) == hipSuccess) {
                h->acc_ms += ms;
                h->acc_launches++;
                h->acc_ms_split[h->events[i].bounded ? 1 : 0] += ms;
                h->acc_launches_split[h->events[i].bounded ? 1 : 0]++;
            }
        }
    }
    h->events_used = 0;
    h->events_base = 0;
}

}  // namespace

// =================================================================================================
extern "C" {

int32_t ctgn_abi_version(void) { return CTGN_ABI_VERSION; }

const char *ctgn_status_string(ctgn_status s) { return status_str(s); }

const char *ctgn_last_error(ctgn_handle h) { return h ? h->last_error.c_str() : "null handle"; }

void ctgn_map_options_default(ctgn_map_options *o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->num_resolutions = 3;
    o->device = 0;
    o->default_radius = 0.8;
    o->resolutions[0] = ctgn_resolution_param{0.2, 0.03, 50, 0};
    o->resolutions[1] = ctgn_resolution_param{0.5, 0.1, 40, 0};
    o->resolutions[2] = ctgn_resolution_param{1.5, 0.15, 40, 0};
    o->initial_voxel_capacity = 0;
}

void ctgn_options_default(ctgn_options *o) {
    if (!o) return;
    o->num_iters_icp = 5;
    o->min_number_neighbors = 20;
    o->max_number_neighbors = 20;
    o->debug_print = 0;
    o->max_dist_to_plane_ct_icp = 0.3;
    o->threshold_orientation_norm = 0.0001;
}

ctgn_status ctgn_create(const ctgn_map_options *opts, ctgn_handle *out) {
    if (!opts || !out) return CTGN_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (opts->num_resolutions < 1 || opts->num_resolutions > CTGN_MAX_RESOLUTIONS) return CTGN_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < opts->num_resolutions; ++i) {
        const ctgn_resolution_param &r = opts->resolutions[i];
        if (!(r.resolution > 0) || r.max_num_points < 1 || r.max_num_points > 64) return CTGN_ERR_INVALID_ARGUMENT;
    }
    if (!(opts->default_radius > 0)) return CTGN_ERR_INVALID_ARGUMENT;
    ctgn_context *h = new (std::nothrow) ctgn_context();
    if (!h) return CTGN_ERR_OUT_OF_MEMORY;
    h->opts = *opts;
    h->device = opts->device;
    h->levels.resize(opts->num_resolutions);
    h->dlevels.resize(opts->num_resolutions);
    for (int i = 0; i < opts->num_resolutions; ++i)
        h->levels[i].init(opts->resolutions[i].resolution, opts->resolutions[i].min_distance_between_points,
                          opts->resolutions[i].max_num_points, opts->initial_voxel_capacity);
    if (h->device >= 0) {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= h->device) { delete h; return CTGN_ERR_NO_DEVICE; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            delete h;
            return CTGN_ERR_NO_DEVICE;
        }
        h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        bool ok = hipSetDevice(h->device) == hipSuccess &&
                  hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
        h->own_stream = ok;
        ok = ok && hipMalloc(reinterpret_cast<void **>(&h->d_state), sizeof(GnState)) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void **>(&h->h_state), sizeof(GnState), hipHostMallocDefault) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&h->d_sys_own), SYS_N * sizeof(double)) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&h->d_partials), ((size_t) (MAX_PARTIAL_BLOCKS + XCD_GROUPS) * SYS_N) * sizeof(double) + XCD_CTL_WORDS * sizeof(unsigned int)) == hipSuccess &&    // + the XCD group records + control words (XcdReduce)
             hipMemsetAsync(h->d_partials + (size_t) (MAX_PARTIAL_BLOCKS + XCD_GROUPS) * SYS_N, 0, XCD_CTL_WORDS * sizeof(unsigned int), h->stream) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&h->d_pose_in), 16 * sizeof(double)) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void **>(&h->h_pose_in), 16 * sizeof(double), hipHostMallocDefault) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&h->d_counters), sizeof(Counters)) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&h->d_bar), 4 * sizeof(unsigned int)) == hipSuccess &&           // 2 arrival counters | 2 XCC masks
             hipMemsetAsync(h->d_bar, 0, 4 * sizeof(unsigned int), h->stream) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&h->d_prof), PROF_WORDS * sizeof(unsigned long long)) == hipSuccess &&
             hipMemsetAsync(h->d_prof, 0, PROF_WORDS * sizeof(unsigned long long), h->stream) == hipSuccess &&
             hipEventCreate(&h->ev_loop_start) == hipSuccess && hipEventCreate(&h->ev_loop_stop) == hipSuccess &&
             hipMemsetAsync(h->d_state, 0, sizeof(GnState), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_sys_own, 0, SYS_N * sizeof(double), h->stream) == hipSuccess;
        h->d_sys = h->d_sys_own;
        // the stop words the solve kernel publishes for the host (launch_gn_iterations): mapped, coherent host memory; without it (an
        // allocation or mapping the runtime refuses) the loop simply enqueues every iteration as before
        if (ok) {
            void *dp = nullptr;
            if (hipHostMalloc(reinterpret_cast<void **>(&h->h_stop), 64 * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
                hipHostGetDevicePointer(&dp, h->h_stop, 0) == hipSuccess) {
                std::memset(h->h_stop, 0, 64 * sizeof(unsigned int));
                h->d_stop = static_cast<unsigned int *>(dp);
            } else {
                if (h->h_stop) (void) hipHostFree(h->h_stop);
                h->h_stop = nullptr;
                (void) hipGetLastError();
            }
        }
        // the kernels use up to ~37 KB of dynamic LDS (rows) / 62 KB (lane): allow it explicitly
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_accumulate_lane),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) lane_kernel_smem()) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_radius_search),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) lane_kernel_smem()) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gn_persistent<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int) persistent_kernel_smem<1>()) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gn_persistent<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int) persistent_kernel_smem<2>()) == hipSuccess;
        h->stage_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_accumulate_rows<2, true, false, 2, false, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int) (rows_kernel_smem<2>() + ROW_WAVES * sizeof(GroupStage))) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_robust_eval_step), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int) sizeof(FuseScratch)) == hipSuccess;
        if (!ok) { ctgn_destroy(h); return CTGN_ERR_HIP; }
    }
    *out = h;
    return CTGN_OK;
}

static void frame_scratch_free(ctgn_handle h) {
    auto &F = h->fr;
    if (F.d_scan) hipFree(F.d_scan);
    if (F.d_world) hipFree(F.d_world);
    if (F.d_corr) hipFree(F.d_corr);
    if (F.d_flag1) hipFree(F.d_flag1);
    if (F.d_flag2) hipFree(F.d_flag2);
    if (F.d_sel1) hipFree(F.d_sel1);
    if (F.d_sel2) hipFree(F.d_sel2);
    if (F.d_counts) hipFree(F.d_counts);
    if (F.h_scan) hipHostFree(F.h_scan);
    if (F.h_out) hipHostFree(F.h_out);
    if (F.h_sel) hipHostFree(F.h_sel);
    if (F.h_counts) hipHostFree(F.h_counts);
    if (F.d_scan_in) hipFree(F.d_scan_in);
    if (F.d_order) hipFree(F.d_order);
    if (F.h_order) hipHostFree(F.h_order);
    if (F.d_selx) hipFree(F.d_selx);
    F = ctgn_context::FrameScratch{};
}

void ctgn_destroy(ctgn_handle h) {
    if (!h) return;
    if (h->device >= 0) {
        hipSetDevice(h->device);
        if (h->stream) hipStreamSynchronize(h->stream);
        if (h->comm && rccl_api().ok) { rccl_api().CommDestroy(h->comm); h->comm = nullptr; }
        for (auto &d : h->dlevels) { if (d.slots) hipFree(d.slots); if (d.blocks) hipFree(d.blocks); }
        for (auto &d : h->devlevels) devmap_level_free(d);
        devmap_scratch_free(h->dm);
        order_scratch_free(h->ord);
        if (h->d_kp_sorted) hipFree(h->d_kp_sorted);
        if (h->d_world0) hipFree(h->d_world0);
        if (h->d_kp) hipFree(h->d_kp);
        if (h->d_tp) hipFree(h->d_tp);
        if (h->h_tp) hipHostFree(h->h_tp);
        for (auto &e : h->tp_events) hipEventDestroy(e);
        if (h->stream_down) hipStreamDestroy(h->stream_down);
        if (h->ev_frame) hipEventDestroy(h->ev_frame);
        if (h->d_res) hipFree(h->d_res);
        if (h->h_kp) hipHostFree(h->h_kp);
        if (h->d_state) hipFree(h->d_state);
        if (h->h_state) hipHostFree(h->h_state);
        if (h->d_sys_own) hipFree(h->d_sys_own);
        if (h->d_partials) hipFree(h->d_partials);
        if (h->d_pose_in) hipFree(h->d_pose_in);
        if (h->h_pose_in) hipHostFree(h->h_pose_in);
        if (h->h_stop) hipHostFree(h->h_stop);
        if (h->d_counters) hipFree(h->d_counters);
        if (h->d_bar) hipFree(h->d_bar);
        if (h->d_prof) hipFree(h->d_prof);
        if (h->d_rstate) hipFree(h->d_rstate);
        if (h->h_rstate) hipHostFree(h->h_rstate);
        if (h->d_rbuf) hipFree(h->d_rbuf);
        if (h->d_rrank) hipFree(h->d_rrank);
        if (h->d_nnb) { hipFree(h->d_nnb); hipFree(h->d_normal); hipFree(h->d_a2d); hipFree(h->d_far); hipFree(h->d_used); }
        if (h->d_edit) hipFree(h->d_edit);
        if (h->h_edit) hipHostFree(h->h_edit);
        frame_scratch_free(h);
        for (auto &e : h->events) { hipEventDestroy(e.start); hipEventDestroy(e.stop); }
        if (h->ev_loop_start) hipEventDestroy(h->ev_loop_start);
        if (h->ev_loop_stop) hipEventDestroy(h->ev_loop_stop);
        if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    }
    delete h;
}

// ---------------------------------------------------------------------------------------- map
ctgn_status ctgn_map_set_update_mode(ctgn_handle h, int32_t device_updates) {
    NEED_DEVICE(h);
    if (device_updates != 0 && device_updates != 1) return CTGN_ERR_INVALID_ARGUMENT;
    uint64_t npts = 0;
    ctgn_map_num_points(h, &npts);
    if (npts != 0) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "the update mode can only change on an empty map");
    if (device_updates == 1 && h->devlevels.empty()) {
        h->devlevels.resize(h->levels.size());
        for (size_t i = 0; i < h->levels.size(); ++i)
            DMCHK(h, devmap_level_init(h->devlevels[i], h->levels[i].resolution, h->levels[i].min_distance, h->levels[i].blk, h->stream));
    }
    if (device_updates == 1) {
        // the frame pipeline's second stream and its event: created here, not inside the first frame (a stream costs ~0.3 ms to create)
        if (!h->stream_down) HIPCHK(h, hipStreamCreateWithFlags(&h->stream_down, hipStreamNonBlocking));
        if (!h->ev_frame) HIPCHK(h, hipEventCreateWithFlags(&h->ev_frame, hipEventDisableTiming));
    }
    h->update_mode = device_updates;
    return CTGN_OK;
}

static ctgn_status devmap_insert_staged(ctgn_handle h, size_t n, uint8_t *out);
static ctgn_status devmap_insert(ctgn_handle h, const void *xyz_base, size_t stride, ctgn_dtype dt, size_t n, uint8_t *out) {
    if (n == 0) return CTGN_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    DMCHK(h, devmap_scratch_reserve(h->dm, n));
    {
        ctgn_status gs = stage_batch(h, xyz_base, stride, dt, n);
        if (gs != CTGN_OK) return gs;
    }
    return devmap_insert_staged(h, n, out);
}

// the batch is in h->dm.pts (stride h->dm.stride): insert into every level, copy the `inserted` mask out
static ctgn_status devmap_insert_staged(ctgn_handle h, size_t n, uint8_t *out) {
    DevMapScratch &S = h->dm;
    HIPCHK(h, hipMemsetAsync(S.inserted, 0, n, h->stream));
    bool range_error = false, overflow = false;
    for (auto &DL : h->devlevels) {                      // map.h:199-205: every resolution
        DMCHK(h, devmap_level_insert(DL, S, n, h->stream));
        range_error = range_error || DL.host.range_error;
        overflow = overflow || DL.host.overflow;
    }
    if (out && on_device(out)) {
        HIPCHK(h, hipMemcpyAsync(out, S.inserted, n, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else if (out) {
        HIPCHK(h, hipMemcpyAsync(S.h_inserted, S.inserted, n, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::memcpy(out, S.h_inserted, n);
    }
    if (overflow) return fail(h, CTGN_ERR_HIP, "device map capacity exhausted (internal sizing error)");
    if (range_error) {
        // the batch HAS been inserted (every in-range point) and `out` says 0 for the skipped ones: report it, do not fail the call —
        // a caller that retried after an error would insert the batch twice
        for (auto &DL : h->devlevels) (void) hipMemsetAsync(&DL.counters->range_error, 0, sizeof(unsigned int), h->stream);
        h->insert_calls_with_skips++;
        h->last_error = "a point fell outside the 21-bit voxel key range (or was not finite) and was skipped";
    }
    return CTGN_OK;
}

ctgn_status ctgn_map_insert(ctgn_handle h, const void *xyz_base, size_t stride, ctgn_dtype dt, size_t n, uint8_t *out) {
    if (!h || (!xyz_base && n)) return CTGN_ERR_INVALID_ARGUMENT;
    h->kth_fresh = false;               // the bound carried from one search to the next assumes an unchanged map
    if (h->update_mode == 1) return devmap_insert(h, xyz_base, stride, dt, n, out);
    if (n && on_device(xyz_base))
        return fail(h, CTGN_ERR_UNSUPPORTED, "device-memory points need the device-resident map (ctgn_map_set_update_mode(h, 1)); "
                                              "the host mirror inserts from host memory");
    bool range_error = false;
    for (size_t i = 0; i < n; ++i) {
        double x = read_elem(xyz_base, stride, dt, i, 0), y = read_elem(xyz_base, stride, dt, i, 1),
               z = read_elem(xyz_base, stride, dt, i, 2);
        int any = 0;
        for (auto &L : h->levels) {                 // map.h:199-205: every resolution
            int r = L.insert_point(x, y, z);
            if (r < 0) range_error = true;
            else any |= r;
        }
        if (out) out[i] = (uint8_t) any;
    }
    if (range_error) {                              // inserted = 0 for the skipped points; the map holds all the others: not an error
        h->insert_calls_with_skips++;
        h->last_error = "a point fell outside the 21-bit voxel key range (or was not finite) and was skipped";
    }
    return CTGN_OK;
}

ctgn_status ctgn_map_remove_far(ctgn_handle h, const double location[3], double distance) {
    if (!h || !location) return CTGN_ERR_INVALID_ARGUMENT;
    h->kth_fresh = false;
    if (h->update_mode == 1) {
        HIPCHK(h, hipSetDevice(h->device));
        for (auto &DL : h->devlevels) DMCHK(h, devmap_level_remove_far(DL, location, distance, h->stream));
        return CTGN_OK;
    }
    for (auto &L : h->levels) L.remove_far(location, distance);
    return CTGN_OK;
}

ctgn_status ctgn_map_clear(ctgn_handle h) {
    if (!h) return CTGN_ERR_INVALID_ARGUMENT;
    h->kth_fresh = false;
    if (h->update_mode == 1) {
        HIPCHK(h, hipSetDevice(h->device));
        for (auto &DL : h->devlevels) DMCHK(h, devmap_level_clear(DL, h->stream));
        return CTGN_OK;
    }
    for (auto &L : h->levels) {
        bool log = L.log_edits;
        L.clear();
        L.log_edits = log;
    }
    return CTGN_OK;
}

ctgn_status ctgn_map_num_points(ctgn_handle h, uint64_t *out) {
    if (!h || !out) return CTGN_ERR_INVALID_ARGUMENT;
    if (h->map_update_pending) { NEED_DEVICE(h); }     // reads the counters of an update still in flight
    uint64_t s = 0;
    if (h->update_mode == 1) for (auto &DL : h->devlevels) s += DL.host.num_points;
    else for (auto &L : h->levels) s += L.num_points;
    *out = s;
    return CTGN_OK;
}

ctgn_status ctgn_map_num_voxels(ctgn_handle h, int32_t li, uint64_t *out) {
    if (!h || !out || li < 0 || li >= (int) h->levels.size()) return CTGN_ERR_INVALID_ARGUMENT;
    if (h->map_update_pending) { NEED_DEVICE(h); }
    *out = h->update_mode == 1 ? h->devlevels[li].host.num_voxels : h->levels[li].num_voxels;
    return CTGN_OK;
}

ctgn_status ctgn_map_search_params(ctgn_handle h, double radius, int32_t *map_id, double *res, int32_t *nb) {
    if (!h) return CTGN_ERR_INVALID_ARGUMENT;
    if (radius <= 0) radius = h->opts.default_radius;
    int mi, n;
    double r;
    search_params(h->levels, radius, &mi, &r, &n);
    if (map_id) *map_id = mi;
    if (res) *res = r;
    if (nb) *nb = n;
    return CTGN_OK;
}

ctgn_status ctgn_map_export(ctgn_handle h, int32_t li, double *out_xyz, uint64_t cap, uint64_t *out_n) {
    if (!h || li < 0 || li >= (int) h->levels.size()) return CTGN_ERR_INVALID_ARGUMENT;
    if (h->update_mode == 1) {
        HIPCHK(h, hipSetDevice(h->device));
        DMCHK(h, devmap_level_export(h->devlevels[li], out_xyz, cap, out_n, h->stream));
        return CTGN_OK;
    }
    uint64_t n = h->levels[li].export_points(out_xyz, cap);
    if (out_n) *out_n = n;
    return CTGN_OK;
}

ctgn_status ctgn_map_sync(ctgn_handle h) {
    NEED_DEVICE(h);
    MapView mv;
    ctgn_status st = make_map_view(h, -1.0, &mv);
    if (st != CTGN_OK) return st;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CTGN_OK;
}

ctgn_status ctgn_map_radius_search(ctgn_handle h, const double *queries, size_t n, double radius, int32_t k,
                                   double *out_xyz, int32_t *out_count) {
    NEED_DEVICE(h);
    if ((!queries || !out_xyz || !out_count) && n) return CTGN_ERR_INVALID_ARGUMENT;
    if (k < 1 || k > CTGN_MAX_NEIGHBORS) return fail(h, CTGN_ERR_UNSUPPORTED, "max_num_neighbors must be in [1, 32]");
    if (n == 0) return CTGN_OK;
    MapView mv;
    ctgn_status st = make_map_view(h, radius, &mv);
    if (st != CTGN_OK) return st;
    struct DevBuf {                                   // freed on every exit path
        void *p = nullptr;
        ~DevBuf() { if (p) (void) hipFree(p); }
    } bq, bout, bcnt;
    HIPCHK(h, hipMalloc(&bq.p, n * 3 * sizeof(double)));
    HIPCHK(h, hipMalloc(&bout.p, n * (size_t) k * 3 * sizeof(double)));
    HIPCHK(h, hipMalloc(&bcnt.p, n * sizeof(int)));
    double *dq = static_cast<double *>(bq.p), *dout = static_cast<double *>(bout.p);
    int *dcnt = static_cast<int *>(bcnt.p);
    HIPCHK(h, hipMemcpyAsync(dq, queries, n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(dout, 0, n * (size_t) k * 3 * sizeof(double), h->stream));
    hipLaunchKernelGGL(k_radius_search, dim3((unsigned) ((n + LANE_BLOCK - 1) / LANE_BLOCK)), dim3(LANE_BLOCK),
                       lane_kernel_smem(), h->stream, mv, dq, (int) n, (int) k, dout, dcnt);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out_xyz, dout, n * (size_t) k * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(out_count, dcnt, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CTGN_OK;
}

// ---------------------------------------------------------------------------------------- keypoints
// (re)size the keypoint arrays for n keypoints and reset the per-upload state
static ctgn_status reserve_keypoints(ctgn_handle h, size_t n) {
    // a solve begun but not yet launched may still point at the pose behind the keypoint arrays (ctgn_register uploads it there, and the
    // state initialisation is deferred to the solve's first launch): consume it in stream order before the arrays are replaced
    { ctgn_status fs = flush_state_init(h); if (fs != CTGN_OK) return fs; }
    if ((int) n > h->cap_kp) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_kp) HIPCHK(h, hipFree(h->d_kp));
        if (h->d_res) HIPCHK(h, hipFree(h->d_res));
        if (h->h_kp) HIPCHK(h, hipHostFree(h->h_kp));
        h->d_kp = nullptr; h->d_res = nullptr; h->h_kp = nullptr; h->cap_kp = 0;
        size_t cap = std::max<size_t>(n + n / 4, 4096);
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_kp), (cap * 7 + KP_TAIL) * sizeof(double)));
        // records | counts | pool radius, k-th distance | fail list of the split launches | its two counters (alternating per launch)
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_res), (cap * SEL_STRIDE + 4 * cap + 16) * sizeof(uint32_t)));
        HIPCHK(h, hipMemsetAsync(h->d_res + cap * (SEL_STRIDE + 4), 0, 16 * sizeof(uint32_t), h->stream));
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_kp), (cap * 7 + KP_TAIL) * sizeof(double), hipHostMallocDefault));
        h->cap_kp = (int) cap;
    }
    h->n_kp = (int) n;
    h->kth_fresh = false;               // new keypoints: no search of theirs has left a k-th distance
    h->order_stale = true;
    h->order_valid = false;
    h->kp_coherent = false;             // probed per upload where the world points pass through the host (ctgn_set_keypoints)
    h->kp_presorted = false;
    h->world0_valid = false;
    h->kp_stride = (int) std::min<size_t>((n + 63) & ~(size_t) 63, (size_t) h->cap_kp);
    if (n >= 32768 && h->ordering_mode != 0) {       // this upload may be ordered (want_order): have the buffers ready
        ctgn_status rs = order_reserve(h);
        if (rs != CTGN_OK) return rs;
    }
    return CTGN_OK;
}

// ctgn_set_rewind: leave a copy of the freshly uploaded world arrays for ctgn_rewind_keypoints (one device-to-device copy, enqueued)
static ctgn_status save_world0(ctgn_handle h) {
    if (!h->keep_world0 || h->n_kp == 0) return CTGN_OK;
    const size_t c = (size_t) h->kp_stride;
    if (h->world0_cap < 3 * (size_t) h->cap_kp) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_world0) HIPCHK(h, hipFree(h->d_world0));
        h->d_world0 = nullptr; h->world0_cap = 0;
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_world0), 3 * (size_t) h->cap_kp * sizeof(double)));
        h->world0_cap = 3 * (size_t) h->cap_kp;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_world0, h->d_kp + 4 * c, (2 * c + (size_t) h->n_kp) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    h->world0_valid = true;
    return CTGN_OK;
}

ctgn_status ctgn_set_rewind(ctgn_handle h, int32_t enable) {
    if (!h) return CTGN_ERR_INVALID_ARGUMENT;
    h->keep_world0 = enable != 0;
    if (!h->keep_world0) h->world0_valid = false;
    return CTGN_OK;
}

ctgn_status ctgn_rewind_keypoints(ctgn_handle h) {
    NEED_DEVICE(h);
    if (!h->world0_valid)
        return fail(h, CTGN_ERR_INVALID_ARGUMENT, "no saved world points: call ctgn_set_rewind(h, 1) before ctgn_set_keypoints");
    const size_t c = (size_t) h->kp_stride;
    if (h->n_kp)
        HIPCHK(h, hipMemcpyAsync(h->d_kp + 4 * c, h->d_world0, (2 * c + (size_t) h->n_kp) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    h->kth_fresh = false;               // the k-th distances on the device belong to world points that are gone
    h->order_stale = !h->kp_presorted;  // the position-ordered working copy (if any) holds the previous solve's world points
    h->order_valid = false;
    return CTGN_OK;
}

ctgn_status ctgn_set_keypoints(ctgn_handle h, ctgn_view raw, ctgn_view world, ctgn_view ts, size_t n) {
    NEED_DEVICE(h);
    if (n > 0 && (!raw.base || !world.base || !ts.base)) return CTGN_ERR_INVALID_ARGUMENT;
    if (n > (size_t) 1 << 30) return fail(h, CTGN_ERR_UNSUPPORTED, "too many keypoints");
    h->pose_on_device = false;
    const bool dev = n > 0 && on_device(raw.base);
    if (n > 0 && (on_device(world.base) != dev || on_device(ts.base) != dev))
        return fail(h, CTGN_ERR_UNSUPPORTED, "the raw, world and timestamp views must all be host memory or all be device memory");
    {
        ctgn_status rs = reserve_keypoints(h, n);
        if (rs != CTGN_OK) return rs;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));      // staging reuse
    const size_t c = (size_t) h->kp_stride;
    double tmin = INFINITY, tmax = -INFINITY;
    if (dev) {                                       // device-resident views: gathered on the GPU, nothing staged
        ctgn_status gs = gather_device_view(h, raw.base, raw.stride_bytes, raw.dtype, 3, h->d_kp, c, n);
        if (gs == CTGN_OK) gs = gather_device_view(h, ts.base, ts.stride_bytes, ts.dtype, 1, h->d_kp + 3 * c, c, n);
        if (gs == CTGN_OK) gs = gather_device_view(h, world.base, world.stride_bytes, world.dtype, 3, h->d_kp + 4 * c, c, n);
        if (gs == CTGN_OK) gs = device_minmax(h, h->d_kp + 3 * c, n, &tmin, &tmax);
        if (gs != CTGN_OK) return gs;
        h->t_min = tmin; h->t_max = tmax;
        gs = save_world0(h);
        if (gs != CTGN_OK) return gs;
        return ensure_debug(h);
    }
    for (size_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) {
            h->h_kp[a * c + i] = read_elem(raw.base, raw.stride_bytes, raw.dtype, i, a);
            h->h_kp[(4 + a) * c + i] = read_elem(world.base, world.stride_bytes, world.dtype, i, a);
        }
        double t = read_elem(ts.base, ts.stride_bytes, ts.dtype, i, 0);
        h->h_kp[3 * c + i] = t;
        tmin = t < tmin ? t : tmin;
        tmax = t > tmax ? t : tmax;
        if (t != t) tmax = NAN;
    }
    h->t_min = tmin; h->t_max = tmax;
    h->pose_on_device = false;
    h->kp_coherent = false;
    if (n >= 32768) {
        // spatial coherence of the caller's order, probed on ~4 k consecutive pairs of the staged world points at the search
        // resolution: a LiDAR sweep in firing order is coherent (>90 % of consecutive returns fall into the same or a neighbouring
        // voxel), a shuffled cloud is not. Only an incoherent upload is ordered for the sake of the caches (want_order).
        int map_id, nb;
        double res;
        search_params(h->levels, h->opts.default_radius, &map_id, &res, &nb);
        const size_t step = std::max<size_t>(1, (n - 1) / 4096);
        size_t pairs = 0, near = 0;
        for (size_t i = 0; i + 1 < n; i += step, ++pairs) {
            bool ok = true;
            for (int a = 0; a < 3 && ok; ++a) {
                const int v0 = voxel_coord(h->h_kp[(4 + a) * c + i], res), v1 = voxel_coord(h->h_kp[(4 + a) * c + i + 1], res);
                ok = std::abs(v0 - v1) <= 1;
            }
            near += ok ? 1 : 0;
        }
        h->kp_coherent = pairs > 0 && 2 * near >= pairs;
    }
    size_t words = 7 * c;
    if (n && h->pose_with_kp) {                    // ctgn_register: the pose shares the upload
        for (int i = 0; i < 14; ++i) h->h_kp[7 * c + i] = h->pose_with_kp[i];
        words += 16;
        h->pose_on_device = true;
    }
    if (n) HIPCHK(h, hipMemcpyAsync(h->d_kp, h->h_kp, words * sizeof(double), hipMemcpyHostToDevice, h->stream));
    ctgn_status st = save_world0(h);
    if (st != CTGN_OK) return st;
    return ensure_debug(h);
}

static void scatter_world_from_staging(ctgn_handle h, void *world_base, size_t stride, ctgn_dtype dt, size_t n) {
    const size_t c = (size_t) h->kp_stride;
    for (size_t i = 0; i < n; ++i) {
        char *p = static_cast<char *>(world_base) + i * stride;
        for (int a = 0; a < 3; ++a) {
            double v = h->h_kp[(4 + a) * c + i];
            if (dt == CTGN_F64) reinterpret_cast<double *>(p)[a] = v;
            else reinterpret_cast<float *>(p)[a] = (float) v;
        }
    }
}

// world arrays (4, 5, 6 of the keypoint block) -> pinned staging, enqueued on the stream (no sync)
static ctgn_status enqueue_world_readback(ctgn_handle h) {
    const size_t c = (size_t) h->kp_stride, n = (size_t) h->n_kp;
    if (n) HIPCHK(h, hipMemcpyAsync(h->h_kp + 4 * c, h->d_kp + 4 * c, (2 * c + n) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return CTGN_OK;
}

ctgn_status ctgn_get_world_points(ctgn_handle h, void *world_base, size_t stride, ctgn_dtype dt, size_t n) {
    NEED_DEVICE(h);
    if (n > (size_t) h->n_kp || (!world_base && n)) return CTGN_ERR_INVALID_ARGUMENT;
    if (n == 0) return CTGN_OK;
    if (on_device(world_base)) {
        ctgn_status ds = scatter_device_view(h, h->d_kp + 4 * (size_t) h->kp_stride, (size_t) h->kp_stride, 3, world_base, stride, dt, n);
        if (ds != CTGN_OK) return ds;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return CTGN_OK;
    }
    ctgn_status st = enqueue_world_readback(h);
    if (st != CTGN_OK) return st;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    scatter_world_from_staging(h, world_base, stride, dt, n);
    return CTGN_OK;
}

// ---------------------------------------------------------------------------------------- GN loop
ctgn_status ctgn_gn_begin(ctgn_handle h, const double pose[14], const double tbe[2], const ctgn_options *opts,
                          const ctgn_motion_prior *prior) {
    NEED_DEVICE(h);
    if (!pose || !tbe || !opts) return CTGN_ERR_INVALID_ARGUMENT;
    if (opts->max_number_neighbors < 1 || opts->max_number_neighbors > CTGN_MAX_NEIGHBORS)
        return fail(h, CTGN_ERR_UNSUPPORTED, "max_number_neighbors must be in [1, 32]");
    // InterpolatePose CHECKs begin.dest_timestamp <= t <= end.dest_timestamp (types.h:456); the reference would
    // abort the process at ct_icp.cpp:965 — reported as an error instead, before anything is modified.
    if (h->n_kp > 0 && !(tbe[0] <= h->t_min && h->t_max <= tbe[1]))
        return fail(h, CTGN_ERR_TIMESTAMP_RANGE, "keypoint timestamps must lie in [t_begin, t_end]");
    h->gn_t0 = std::chrono::steady_clock::now();
    fill_params(h, opts, prior);
    // h_pose_in is reused: every other user synchronises before returning, only an unfinished stepwise loop can still
    // have a copy from it in flight
    const double *d_pose = h->d_pose_in;
    if (h->pose_on_device) {                          // already behind the keypoint arrays (ctgn_register)
        d_pose = h->d_kp + 7 * (size_t) h->kp_stride;
        h->pose_on_device = false;
    } else if (h->pose_in_valid && std::memcmp(h->h_pose_in, pose, 14 * sizeof(double)) == 0) {
        // the same initial pose as the last upload (a registration retried / repeated on the same keypoints): it is on the device already
    } else {
        if (h->gn_active) HIPCHK(h, hipStreamSynchronize(h->stream));
        h->pose_in_valid = false;                     // h_pose_in changes now; valid again only once the copy is enqueued
        for (int i = 0; i < 14; ++i) h->h_pose_in[i] = pose[i];
        HIPCHK(h, hipMemcpyAsync(h->d_pose_in, h->h_pose_in, 14 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        h->pose_in_valid = true;
    }
    // the state initialisation rides with the solve's first launch (flush_state_init / the persistent kernel's prologue)
    h->init_pending = true;
    h->init_pose = d_pose;
    h->init_tbe[0] = tbe[0]; h->init_tbe[1] = tbe[1];
    h->world_final_done = false;
    h->launched_iters = 0;
    h->searches_in_solve = 0;
    h->fail_reset_pending = true;
    h->init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h->gn_t0).count();
    h->planned_iters = opts->num_iters_icp;
    // a solve begun while the previous one was never ended (repeated solves enqueued back to back): its launches all did work and
    // their event pairs are harvested with this solve's
    if (h->gn_active && h->profiling) h->events_base = h->events_used;
    else { h->events_used = 0; h->events_base = 0; }
    h->gn_active = true;
    HIPCHK(h, hipEventRecord(h->ev_loop_start, h->stream));
    return CTGN_OK;
}

ctgn_status ctgn_gn_accumulate(ctgn_handle h) {
    NEED_DEVICE(h);
    if (!h->gn_active) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_gn_begin was not called");
    MapView mv;
    ctgn_status st = make_map_view(h, -1.0, &mv);
    if (st != CTGN_OK) return st;
    st = launch_accumulate(h, mv, h->launched_iters == 0);
    if (st != CTGN_OK) return st;
    h->launched_iters++;
    return launch_reduce_solve(h, 1);
}

// `iterations` whole GN iterations enqueued behind ctgn_gn_begin: the fused sequence ctgn_solve runs (search -> residual/reduce ->
// reduce + solve), or with `sharded` the sequence of ctgn_solve_sharded (... -> reduce -> ncclAllReduce -> solve). No synchronisation.
ctgn_status ctgn_gn_iterate(ctgn_handle h, int32_t iterations, int32_t sharded) {
    NEED_DEVICE(h);
    if (!h->gn_active) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_gn_begin was not called");
    if (sharded && !h->comm) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_dist_init was not called");
    MapView mv;
    ctgn_status st = make_map_view(h, -1.0, &mv);
    if (st == CTGN_OK && !sharded && iterations > 0 && persistent_ok(h, mv))
        return launch_persistent(h, mv, iterations, false, nullptr);        // a small frame: the whole batch of iterations in one launch
    for (int it = 0; st == CTGN_OK && it < iterations; ++it) {
        st = launch_accumulate(h, mv, h->launched_iters == 0);
        if (st != CTGN_OK) break;
        h->launched_iters++;
        if (!sharded) { st = launch_reduce_solve(h, 0); continue; }
        st = launch_reduce_solve(h, 1);
        if (st != CTGN_OK) break;
        const ncclResult_t r = rccl_api().AllReduce(h->d_sys, h->d_sys, CTGN_SYSTEM_DOUBLES, ncclDouble, ncclSum, h->comm, h->stream);
        if (r != ncclSuccess) { st = fail(h, CTGN_ERR_HIP, std::string("[RCCL] ncclAllReduce: ") + rccl_api().GetErrorString(r)); break; }
        st = launch_reduce_solve(h, 2);
    }
    return st;
}

ctgn_status ctgn_gn_system_device_ptr(ctgn_handle h, void **out) {
    NEED_DEVICE(h);
    if (!out) return CTGN_ERR_INVALID_ARGUMENT;
    *out = h->d_sys;
    return CTGN_OK;
}

ctgn_status ctgn_gn_set_system_buffer(ctgn_handle h, void *device_ptr) {
    NEED_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->d_sys = device_ptr ? static_cast<double *>(device_ptr) : h->d_sys_own;
    return CTGN_OK;
}

ctgn_status ctgn_gn_solve_update(ctgn_handle h) {
    NEED_DEVICE(h);
    if (!h->gn_active) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_gn_begin was not called");
    return launch_reduce_solve(h, 2);
}

ctgn_status ctgn_gn_done(ctgn_handle h, int32_t *done) {
    NEED_DEVICE(h);
    if (!done) return CTGN_ERR_INVALID_ARGUMENT;
    {
        ctgn_status fs = flush_state_init(h);
        if (fs != CTGN_OK) return fs;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_state, h->d_state, sizeof(GnState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *done = h->h_state->done;
    return CTGN_OK;
}

static ctgn_status gn_collect(ctgn_handle h, double pose_out[14], ctgn_summary *summary);
ctgn_status ctgn_gn_end(ctgn_handle h, double pose_out[14], ctgn_summary *summary) {
    NEED_DEVICE(h);
    if (!h->gn_active) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_gn_begin was not called");
    const bool merged = h->prefetch_world && h->n_kp > 0;      // world points + state in ONE device-to-host copy
    const size_t c = (size_t) h->kp_stride;
    {
        ctgn_status fs = flush_state_init(h);
        if (fs != CTGN_OK) return fs;
    }
    if (h->n_kp > 0 && !h->world_final_done) {                 // (the persistent kernel's epilogue has done both already)
        const int grid = std::max(1, std::min((h->n_kp + 255) / 256, 2048));
        hipLaunchKernelGGL(k_transform, dim3(grid), dim3(256), 0, h->stream, kp_view(h), h->d_state,
                           merged ? h->d_kp + 7 * c + 16 : nullptr);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipEventRecord(h->ev_loop_stop, h->stream));
    // (round 5, measured and removed: a zero-copy ending for small frames — k_transform writing the world points and the final state
    // straight into the page-locked staging block and publishing a completion word the host polled, no copy command, no stream
    // synchronisation. Register of 1 679 keypoints: 0.1530 / 0.1541 ms with the copy + synchronise below, 0.1531 / 0.1529 ms without
    // them (profiles/r05_ab_s2_register_zero_copy_end.txt): the end of the call waits for the device, not for the runtime.)
    if (merged) {
        HIPCHK(h, hipMemcpyAsync(h->h_kp + 4 * c, h->d_kp + 4 * c, (3 * c + KP_TAIL) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    } else {
        HIPCHK(h, hipMemcpyAsync(h->h_state, h->d_state, sizeof(GnState), hipMemcpyDeviceToHost, h->stream));
        if (h->prefetch_world) { ctgn_status ws = enqueue_world_readback(h); if (ws != CTGN_OK) return ws; }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (merged) std::memcpy(h->h_state, h->h_kp + 7 * c + 16, sizeof(GnState));
    return gn_collect(h, pose_out, summary);
}

// after the final state has arrived in h->h_state: pose + ICPSummary
static ctgn_status gn_collect(ctgn_handle h, double pose_out[14], ctgn_summary *summary) {
    h->gn_active = false;
    const GnState &s = *h->h_state;
    if (h->profiling) harvest_events(h, h->events_base + s.iter + (s.failed ? 1 : 0));
    if (h->gn_opts.debug_print > 1)
        std::fprintf(stderr, "[ctgn] k_reduce_solve clocks: reduce %llu factorise %llu substitute %llu update %llu\n",
                     s.solve_cycles[0], s.solve_cycles[1], s.solve_cycles[2], s.solve_cycles[3]);
    if (s.failed == GN_FAILED_BARRIER) {
        // the persistent kernel's blocks were not all running (the device is shared with something that holds its compute units): the
        // registration was abandoned before the pose changed; this handle goes back to the three-launch loop
        h->persist_disabled = true;
        if (summary) { std::memset(summary, 0, sizeof(*summary)); std::snprintf(summary->error_log, sizeof(summary->error_log), "[HIP] in-kernel barrier timed out; retry"); }
        return fail(h, CTGN_ERR_HIP, "[HIP] the persistent small-frame kernel's barrier timed out (device shared?); the handle now uses the three-launch loop — retry the call");
    }
    if (s.failed == GN_FAILED_PEER) {
        if (summary) { std::memset(summary, 0, sizeof(*summary)); std::snprintf(summary->error_log, sizeof(summary->error_log), "[RCCL] a peer rank failed before the exchange"); }
        return fail(h, CTGN_ERR_INVALID_ARGUMENT, "keypoint-sharded solve: another rank could not start its solve (see that rank's error); the pose is unchanged");
    }
    if (pose_out) for (int i = 0; i < 14; ++i) pose_out[i] = s.pose[i];
    if (summary) {
        std::memset(summary, 0, sizeof(*summary));
        summary->success = s.failed ? 0 : 1;                       // ct_icp.cpp:869 / :992
        summary->num_residuals_used = s.n_used;
        summary->num_iters = s.iter;
        summary->last_step_norm = s.step_norm;
        float ms = 0.f;
        hipEventElapsedTime(&ms, h->ev_loop_start, h->ev_loop_stop);
        summary->duration_device_ms = ms;
        // ICPSummary's timing fields (ct_icp.h:164-168), from the device's 100 MHz wall clock stamped by the kernels themselves
        const double per_iter = s.iter > 0 ? 1e-5 / (double) s.iter : 0.0;                 // 10 ns ticks -> ms, averaged
        summary->avg_duration_neighborhood_ms = (double) s.ticks_neighborhood * per_iter;
        summary->avg_duration_solve_ms = (double) s.ticks_solve * per_iter;
        summary->avg_duration_iter_ms = (double) s.ticks_iter * per_iter;
        summary->duration_init_ms = h->init_ms;
        summary->duration_total_ms =
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h->gn_t0).count();
        if (s.failed) {
            std::snprintf(summary->error_log, sizeof(summary->error_log),
                          "[CT_ICP]Error : not enough keypoints selected in ct-icp !\n[CT_ICP]Number_of_residuals : %d\n",
                          s.n_used);                                // same text as ct_icp.cpp:862-863
            if (h->gn_opts.debug_print) std::fputs(summary->error_log, stdout);
        }
    }
    return CTGN_OK;
}

ctgn_status ctgn_solve(ctgn_handle h, double pose_io[14], const double tbe[2], const ctgn_options *opts,
                       const ctgn_motion_prior *prior, ctgn_summary *summary) {
    ctgn_status st = ctgn_gn_begin(h, pose_io, tbe, opts, prior);
    if (st != CTGN_OK) {
        if (summary) { std::memset(summary, 0, sizeof(*summary)); std::snprintf(summary->error_log, sizeof(summary->error_log), "%s", ctgn_last_error(h)); }
        return st;
    }
    MapView mv;
    st = make_map_view(h, -1.0, &mv);
    // (a hipGraph captured from this loop was measured in round 1 and dropped: the loop is bound by the GPU-side latency of its dependent
    // launches, not by host launch cost — docs/history.md section 7)
    if (st == CTGN_OK && opts->num_iters_icp > 0 && persistent_ok(h, mv)) {
        // a small frame (the reference's own keypoint count): state init, all iterations and the final re-transform in ONE launch
        const bool merged = h->prefetch_world && h->n_kp > 0;
        st = launch_persistent(h, mv, opts->num_iters_icp, true, merged ? h->d_kp + 7 * (size_t) h->kp_stride + 16 : nullptr);
    } else if (st == CTGN_OK) {
        st = launch_gn_iterations(h, mv, opts->num_iters_icp);
    }
    if (st != CTGN_OK) {
        h->gn_active = false;
        hipStreamSynchronize(h->stream);
        if (summary) { std::memset(summary, 0, sizeof(*summary)); std::snprintf(summary->error_log, sizeof(summary->error_log), "%s", ctgn_last_error(h)); }
        return st;
    }
    return ctgn_gn_end(h, pose_io, summary);
}

/* -------------------------------------------------------------------------------------------------
 * Keypoint-sharded mode: the collective is issued from here, on the handle's stream
 * ---------------------------------------------------------------------------------------------- */
ctgn_status ctgn_dist_unique_id(uint8_t out[CTGN_DIST_ID_BYTES]) {
    static_assert(CTGN_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ctgn.h mirrors ncclUniqueId");
    if (!out) return CTGN_ERR_INVALID_ARGUMENT;
    if (!rccl_api().ok) return CTGN_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (rccl_api().GetUniqueId(&id) != ncclSuccess) return CTGN_ERR_HIP;
    std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return CTGN_OK;
}

ctgn_status ctgn_dist_init(ctgn_handle h, int32_t rank, int32_t world_size, const uint8_t id_bytes[CTGN_DIST_ID_BYTES]) {
    NEED_DEVICE(h);
    if (!id_bytes || world_size < 1 || rank < 0 || rank >= world_size) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "bad rank / world size / id");
    if (!rccl_api().ok) return fail(h, CTGN_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    if (h->comm) { rccl_api().CommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = rccl_api().CommInitRank(&h->comm, world_size, id, rank);
    if (r != ncclSuccess) { h->comm = nullptr; return fail(h, CTGN_ERR_HIP, std::string("[RCCL] ncclCommInitRank: ") + rccl_api().GetErrorString(r)); }
    h->dist_rank = rank;
    h->dist_world = world_size;
    return CTGN_OK;
}

ctgn_status ctgn_dist_shutdown(ctgn_handle h) {
    NEED_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm && rccl_api().ok) rccl_api().CommDestroy(h->comm);
    h->comm = nullptr;
    h->dist_rank = 0; h->dist_world = 1;
    return CTGN_OK;
}

// Host views, more than one rank (round 5): every rank needs the ORDER of the whole scan (the partition must be the same everywhere and the
// sort is the library's deterministic one), but only its own chunk of the seven keypoint arrays. So only the world points go up for the key
// pass (24 B per keypoint of the scan), the rank's chunk of the order comes back (4 B per keypoint of the chunk), and the chunk's rows are
// gathered from the caller's views straight into kernel order and uploaded (56 B per keypoint of the chunk): 24 N + 56 N / G bytes per rank
// instead of 56 N. Timestamps are range-checked over the whole scan on the host, so that every rank accepts or refuses the same scan.
static ctgn_status set_keypoints_sharded_lean(ctgn_handle h, ctgn_view raw, ctgn_view world, ctgn_view ts, size_t n, int32_t rank,
                                              int32_t world_size, uint32_t *shard_indices, size_t *shard_n) {
    if (n > (size_t) 1 << 30) return fail(h, CTGN_ERR_UNSUPPORTED, "too many keypoints");
    ctgn_status st = reserve_keypoints(h, n);
    if (st != CTGN_OK) return st;
    HIPCHK(h, hipStreamSynchronize(h->stream));                // staging reuse
    const size_t c = (size_t) h->kp_stride;
    double tmin = INFINITY, tmax = -INFINITY;
    for (size_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) h->h_kp[(4 + a) * c + i] = read_elem(world.base, world.stride_bytes, world.dtype, i, a);
        const double t = read_elem(ts.base, ts.stride_bytes, ts.dtype, i, 0);
        tmin = t < tmin ? t : tmin;
        tmax = t > tmax ? t : tmax;
        if (t != t) tmax = NAN;
    }
    h->t_min = tmin; h->t_max = tmax;
    HIPCHK(h, hipMemcpyAsync(h->d_kp + 4 * c, h->h_kp + 4 * c, 3 * c * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int map_id, nb;
    double res;
    search_params(h->levels, h->opts.default_radius, &map_id, &res, &nb);
    st = order_reserve(h);
    if (st != CTGN_OK) return st;
    DMCHK(h, order_by_home_voxel(h->ord, h->d_kp + 4 * c, h->d_kp + 5 * c, h->d_kp + 6 * c, n, res, h->stream));
    const size_t base = n / (size_t) world_size, rem = n % (size_t) world_size;        // contiguous, balanced chunks
    const size_t lo = (size_t) rank * base + std::min<size_t>((size_t) rank, rem), m = base + ((size_t) rank < rem ? 1 : 0);
    const size_t c2 = std::min((m + 63) & ~(size_t) 63, c);
    std::vector<uint32_t> own;
    uint32_t *idx = shard_indices;
    if (!idx) { own.resize(m); idx = own.data(); }
    HIPCHK(h, hipMemcpyAsync(idx, h->ord.order + lo, m * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));                // the order is here; the staging buffer is free again
    for (size_t j = 0; j < m; ++j) {
        const size_t i = idx[j];
        for (int a = 0; a < 3; ++a) {
            h->h_kp[a * c2 + j] = read_elem(raw.base, raw.stride_bytes, raw.dtype, i, a);
            h->h_kp[(4 + a) * c2 + j] = read_elem(world.base, world.stride_bytes, world.dtype, i, a);
        }
        h->h_kp[3 * c2 + j] = read_elem(ts.base, ts.stride_bytes, ts.dtype, i, 0);
    }
    for (int a = 0; a < 7; ++a)
        for (size_t j = m; j < c2; ++j) h->h_kp[a * c2 + j] = 0.0;
    HIPCHK(h, hipMemcpyAsync(h->d_kp, h->h_kp, 7 * c2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->last_upload_bytes = (3 * (uint64_t) c + 7 * (uint64_t) c2) * sizeof(double);
    h->n_kp = (int) m;
    h->kp_stride = (int) c2;
    h->pose_on_device = false;
    h->kp_coherent = false;
    h->kp_presorted = true;
    h->order_stale = false;
    h->order_valid = false;
    h->kth_fresh = false;
    if (shard_n) *shard_n = m;
    st = save_world0(h);
    if (st != CTGN_OK) return st;
    return ensure_debug(h);
}

ctgn_status ctgn_set_keypoints_sharded(ctgn_handle h, ctgn_view raw, ctgn_view world, ctgn_view ts, size_t n, int32_t rank, int32_t world_size,
                                       uint32_t *shard_indices, size_t *shard_n) {
    NEED_DEVICE(h);
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "bad rank / world size");
    if (shard_n) *shard_n = 0;
    h->last_upload_bytes = 0;
    if (world_size > 1 && n >= (size_t) 64 * (size_t) world_size && raw.base && world.base && ts.base && !on_device(raw.base) &&
        !on_device(world.base) && !on_device(ts.base))
        return set_keypoints_sharded_lean(h, raw, world, ts, n, rank, world_size, shard_indices, shard_n);
    const bool keep = h->keep_world0;
    h->keep_world0 = false;                                    // the copy for ctgn_rewind_keypoints is taken of the shard, below
    ctgn_status st = ctgn_set_keypoints(h, raw, world, ts, n); // the whole scan, resident for a moment; t_min / t_max are the scan's
    if (st == CTGN_OK && n > 0 && !on_device(raw.base)) h->last_upload_bytes = 7 * (uint64_t) h->kp_stride * sizeof(double);
    h->keep_world0 = keep;
    if (st != CTGN_OK || n == 0) return st;
    int map_id, nb;
    double res;
    search_params(h->levels, h->opts.default_radius, &map_id, &res, &nb);
    const size_t c = (size_t) h->kp_stride;
    st = order_reserve(h);
    if (st != CTGN_OK) return st;
    DMCHK(h, order_by_home_voxel(h->ord, h->d_kp + 4 * c, h->d_kp + 5 * c, h->d_kp + 6 * c, n, res, h->stream));
    const size_t base = n / (size_t) world_size, rem = n % (size_t) world_size;        // contiguous, balanced chunks
    const size_t lo = (size_t) rank * base + std::min<size_t>((size_t) rank, rem), m = base + ((size_t) rank < rem ? 1 : 0);
    const size_t c2 = std::min((m + 63) & ~(size_t) 63, (size_t) h->cap_kp);
    if (m) {
        hipLaunchKernelGGL(k_kp_permute, dim3(grid_for(m)), dim3(256), 0, h->stream, h->d_kp, c, h->ord.order + lo, (int) m, h->d_kp_sorted, c2);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(h->d_kp, h->d_kp_sorted, 7 * c2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    if (shard_indices && m) HIPCHK(h, hipMemcpyAsync(shard_indices, h->ord.order + lo, m * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n_kp = (int) m;
    h->kp_stride = (int) c2;
    h->pose_on_device = false;
    h->kp_presorted = true;
    h->order_stale = false;
    h->order_valid = false;
    if (shard_n) *shard_n = m;
    return save_world0(h);
}

// A rank that cannot go on (its gn_begin failed: a timestamp of ITS shard outside the frame; a launch or a map view failed mid-loop) while
// its peers, whose shards are fine, wait in the all-reduce: it still takes part in every remaining exchange, with a poisoned count — every
// rank's solve kernel sees the negative sum, stops before the pose changes and reports GN_FAILED_PEER. All ranks fail together.
ctgn_status ctgn_dist_overheads(ctgn_handle h, int32_t reps, double out_us[2]) {
    NEED_DEVICE(h);
    if (!out_us || reps < 1) return CTGN_ERR_INVALID_ARGUMENT;
    if (!h->comm) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_dist_init was not called");
    out_us[0] = out_us[1] = 0.0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double *scratch = h->d_partials;                   // any 96 doubles nobody reads between solves
    for (int pass = 0; pass < 2; ++pass) {             // pass 0 warms the communicator's channels up
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) {
            const ncclResult_t r = rccl_api().AllReduce(scratch, scratch, CTGN_SYSTEM_DOUBLES, ncclDouble, ncclSum, h->comm, h->stream);
            if (r != ncclSuccess) return fail(h, CTGN_ERR_HIP, std::string("[RCCL] ncclAllReduce: ") + rccl_api().GetErrorString(r));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        out_us[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    }
    // the chain of a sharded iteration with the stop flag set: every kernel returns at once, the all-reduce runs
    int one = 1, zero = 0;
    int *d_done = reinterpret_cast<int *>(reinterpret_cast<char *>(h->d_state) + offsetof(GnState, done));
    HIPCHK(h, hipMemcpyAsync(d_done, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->n_kp > 0) {
        MapView mv;
        ctgn_status st = make_map_view(h, -1.0, &mv);
        if (st != CTGN_OK) return st;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps && st == CTGN_OK; ++i) {
            h->init_pending = false;
            st = launch_accumulate(h, mv, false);
            if (st == CTGN_OK) st = launch_reduce_solve(h, 1);
            if (st == CTGN_OK && rccl_api().AllReduce(h->d_sys, h->d_sys, CTGN_SYSTEM_DOUBLES, ncclDouble, ncclSum, h->comm, h->stream) != ncclSuccess)
                st = fail(h, CTGN_ERR_HIP, "[RCCL] ncclAllReduce");
            if (st == CTGN_OK) st = launch_reduce_solve(h, 2);
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (st != CTGN_OK) return st;
        out_us[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    }
    HIPCHK(h, hipMemcpyAsync(d_done, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->kth_fresh = false;
    return CTGN_OK;
}

static void join_remaining_exchanges_poisoned(ctgn_handle h, int remaining) {
    if (remaining <= 0 || !h->d_sys || !h->comm) return;
    double poison[CTGN_SYSTEM_DOUBLES] = {0.0};
    poison[90] = GN_PEER_POISON;
    bool ok = true;
    for (int it = 0; ok && it < remaining; ++it) {
        // (re-poisoned every time: the sum of the previous exchange is what the buffer holds now, and it must stay negative whatever the peers add)
        ok = hipMemcpyAsync(h->d_sys, poison, sizeof(poison), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
             hipStreamSynchronize(h->stream) == hipSuccess &&
             rccl_api().AllReduce(h->d_sys, h->d_sys, CTGN_SYSTEM_DOUBLES, ncclDouble, ncclSum, h->comm, h->stream) == ncclSuccess;
    }
    hipStreamSynchronize(h->stream);
}

ctgn_status ctgn_solve_sharded(ctgn_handle h, double pose_io[14], const double tbe[2], const ctgn_options *opts,
                               const ctgn_motion_prior *prior, ctgn_summary *summary) {
    NEED_DEVICE(h);
    if (!h->comm) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_dist_init was not called");
    // the options are the same on every rank: rejecting them here stops all ranks before any of them waits for another
    if (!opts || !pose_io || !tbe || opts->num_iters_icp < 0) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_solve_sharded: pose, frame times and options are required");
    const int iters = opts->num_iters_icp;
    auto give_up = [&](ctgn_status st, int exchanges_left) {
        const std::string own_error = ctgn_last_error(h);
        join_remaining_exchanges_poisoned(h, exchanges_left);
        h->gn_active = false;
        fail(h, st, own_error);
        if (summary) { std::memset(summary, 0, sizeof(*summary)); std::snprintf(summary->error_log, sizeof(summary->error_log), "%s", own_error.c_str()); }
        return st;
    };
    ctgn_status st = ctgn_gn_begin(h, pose_io, tbe, opts, prior);
    if (st != CTGN_OK) return give_up(st, iters);
    MapView mv;
    st = make_map_view(h, -1.0, &mv);
    if (st != CTGN_OK) return give_up(st, iters);
    for (int it = 0; it < iters; ++it) {
        st = launch_accumulate(h, mv, it == 0);                    // this rank's shard -> per-block partials
        if (st != CTGN_OK) return give_up(st, iters - it);
        h->launched_iters++;
        st = launch_reduce_solve(h, 1);                             // -> packed system (96 doubles) in d_sys
        if (st != CTGN_OK) return give_up(st, iters - it);
        // the one exchange of the path: 78 J^T J | 12 J^T r | count | pad, summed over the ranks, in place, on this stream
        const ncclResult_t r = rccl_api().AllReduce(h->d_sys, h->d_sys, CTGN_SYSTEM_DOUBLES, ncclDouble, ncclSum, h->comm, h->stream);
        if (r != ncclSuccess) {                                     // the communicator itself failed: nothing left to take part in
            fail(h, CTGN_ERR_HIP, std::string("[RCCL] ncclAllReduce: ") + rccl_api().GetErrorString(r));
            return give_up(CTGN_ERR_HIP, 0);
        }
        st = launch_reduce_solve(h, 2);                             // identical input on every rank -> identical pose, no broadcast
        if (st != CTGN_OK) return give_up(st, iters - it - 1);
    }
    return ctgn_gn_end(h, pose_io, summary);
}

ctgn_status ctgn_grid_sampling(ctgn_handle h, ctgn_view xyz, size_t n, double voxel_size, uint32_t *out_indices, size_t *out_count) {
    NEED_DEVICE(h);
    if (!out_count || !(voxel_size > 0) || (n && (!xyz.base || !out_indices))) return CTGN_ERR_INVALID_ARGUMENT;
    *out_count = 0;
    if (n == 0) return CTGN_OK;
    if (n > (size_t) 1 << 30) return fail(h, CTGN_ERR_UNSUPPORTED, "too many points");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    DMCHK(h, devmap_scratch_reserve(h->dm, n));
    DevMapScratch &S = h->dm;
    {
        ctgn_status gs = stage_batch(h, xyz.base, xyz.stride_bytes, xyz.dtype, n);
        if (gs != CTGN_OK) return gs;
    }
    DMCHK(h, devmap_grid_sampling(S, n, voxel_size, out_indices, out_count, h->stream));   // out_indices: host or device memory
    return CTGN_OK;
}

void ctgn_adaptive_sampling_options_default(ctgn_adaptive_sampling_options *o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->num_points_per_voxel = 1;
    o->max_num_points = -1;
    o->num_bands = 6;
    const double d[6] = {0.5, 2.0, 4.0, 8.0, 16.0, 200.0}, v[6] = {0.1, 0.2, 0.4, 0.8, 1.6, -1.0};   // sampling.h:18-25
    for (int j = 0; j < 6; ++j) { o->distance[j] = d[j]; o->voxel_size[j] = v[j]; }
}

ctgn_status ctgn_adaptive_sampling(ctgn_handle h, ctgn_view xyz, size_t n, const ctgn_adaptive_sampling_options *opts,
                                   uint32_t *out_indices, size_t *out_count) {
    NEED_DEVICE(h);
    if (!out_count || !opts || (n && (!xyz.base || !out_indices))) return CTGN_ERR_INVALID_ARGUMENT;
    *out_count = 0;
    if (opts->num_bands < 2 || opts->num_bands > CTGN_ADAPTIVE_MAX_BANDS || opts->num_points_per_voxel < 1)
        return fail(h, CTGN_ERR_INVALID_ARGUMENT, "adaptive sampling: 2..16 bands and num_points_per_voxel >= 1 expected");
    AdaptiveBands bands{};
    bands.num_bands = opts->num_bands;
    bands.num_points_per_voxel = opts->num_points_per_voxel;
    for (int j = 0; j < opts->num_bands; ++j) { bands.distance[j] = opts->distance[j]; bands.voxel_size[j] = opts->voxel_size[j]; }
    for (int j = 0; j + 1 < opts->num_bands; ++j) {
        if (!(bands.distance[j] < bands.distance[j + 1]) || !(bands.voxel_size[j] > 0))
            return fail(h, CTGN_ERR_INVALID_ARGUMENT, "adaptive sampling: distances must ascend and used voxel sizes be positive");
        if (!(bands.distance[j + 1] / bands.voxel_size[j] < (double) (1 << 19)))
            return fail(h, CTGN_ERR_INVALID_ARGUMENT, "adaptive sampling: band voxel coordinates exceed 20 bits");
    }
    if (n == 0) return CTGN_OK;
    if (n > (size_t) 1 << 30) return fail(h, CTGN_ERR_UNSUPPORTED, "too many points");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    DMCHK(h, devmap_scratch_reserve(h->dm, n));
    DevMapScratch &S = h->dm;
    {
        ctgn_status gs = stage_batch(h, xyz.base, xyz.stride_bytes, xyz.dtype, n);
        if (gs != CTGN_OK) return gs;
    }
    DMCHK(h, devmap_adaptive_sampling(S, n, bands, opts->max_num_points, out_indices, out_count, h->stream));
    return CTGN_OK;
}

static void write_point(void *base, size_t stride, ctgn_dtype dt, size_t i, double x, double y, double z) {
    char *p = static_cast<char *>(base) + i * stride;
    if (dt == CTGN_F64) { double *q = reinterpret_cast<double *>(p); q[0] = x; q[1] = y; q[2] = z; }
    else { float *q = reinterpret_cast<float *>(p); q[0] = (float) x; q[1] = (float) y; q[2] = (float) z; }
}

ctgn_status ctgn_transform_points(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const double pose[14], const double tbe[2],
                                  void *out_base, size_t out_stride, ctgn_dtype out_dtype) {
    NEED_DEVICE(h);
    if (!pose || !tbe || (n && (!raw.base || !ts.base || !out_base))) return CTGN_ERR_INVALID_ARGUMENT;
    if (n == 0) return CTGN_OK;
    if (n > (size_t) 1 << 30) return fail(h, CTGN_ERR_UNSUPPORTED, "too many points");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n > h->tp_cap) {
        if (h->d_tp) HIPCHK(h, hipFree(h->d_tp));
        if (h->h_tp) HIPCHK(h, hipHostFree(h->h_tp));
        h->d_tp = nullptr; h->h_tp = nullptr; h->tp_cap = 0;
        const size_t cap = ((n + n / 4 + 1024) + 63) & ~(size_t) 63;
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_tp), (cap * 7 + 32) * sizeof(double)));      // pose | x y z t records | out x y z (device views: planes)
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_tp), (cap * 7 + 32) * sizeof(double), hipHostMallocDefault));
        h->tp_cap = cap;
    }
    // array stride: n rounded up to 64, so that the four input arrays + the pose travel in ONE copy and the three output arrays
    // come back in one (seven separate copies cost more than the bytes they move)
    const size_t c = std::min((n + 63) & ~(size_t) 63, h->tp_cap);
    const bool dev = on_device(raw.base);
    if (on_device(ts.base) != dev || on_device(out_base) != dev)
        return fail(h, CTGN_ERR_UNSUPPORTED, "the point, timestamp and output views must all be host memory or all be device memory");
    if (dev) {
        ctgn_status gs = gather_device_view(h, raw.base, raw.stride_bytes, raw.dtype, 3, h->d_tp, c, n);
        if (gs == CTGN_OK) gs = gather_device_view(h, ts.base, ts.stride_bytes, ts.dtype, 1, h->d_tp + 3 * c, c, n);
        double lo = 0, hi = 0;
        if (gs == CTGN_OK) gs = device_minmax(h, h->d_tp + 3 * c, n, &lo, &hi);
        if (gs != CTGN_OK) return gs;
        if (!(tbe[0] <= lo && hi <= tbe[1])) return fail(h, CTGN_ERR_TIMESTAMP_RANGE, "point timestamps must lie in [t_begin, t_end]");
        // a solve begun but not yet launched still points at d_pose_in (its state initialisation is deferred to its first launch):
        // consume that pose in stream order before it is overwritten
        { ctgn_status fs = flush_state_init(h); if (fs != CTGN_OK) return fs; }
        h->pose_in_valid = false;
        for (int i = 0; i < 14; ++i) h->h_pose_in[i] = pose[i];
        HIPCHK(h, hipMemcpyAsync(h->d_pose_in, h->h_pose_in, 14 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        h->pose_in_valid = true;
        hipLaunchKernelGGL(k_transform_points, dim3(grid_for(n)), dim3(256), 0, h->stream, h->d_tp, h->d_tp + 4 * c, (int) n, c,
                           h->d_pose_in, tbe[0], tbe[1]);
        HIPCHK(h, hipGetLastError());
        gs = scatter_device_view(h, h->d_tp + 4 * c, c, 3, out_base, out_stride, out_dtype, n);
        if (gs != CTGN_OK) return gs;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return CTGN_OK;
    }
    // Host views: a pipeline over 32 k-point chunks. The caller's strided records are gathered into pinned memory as x y z t records
    // behind the pose and go up chunk by chunk while the next chunk is being gathered; every chunk's kernel writes x y z records that
    // come back into pinned memory on a second stream, beside the uploads of the chunks behind them. Once everything is gathered (and
    // every timestamp checked: nothing reaches the caller's output before that) the chunks are handed over as they arrive. The step
    // moves 56 bytes per point across PCIe for a few flops. Measured on the B2 scan (132 k points, 7.4 MB; tuning frame_timing marks):
    // gathered and enqueued after 0.17 ms, handed over at 0.28 ms; on one stream 0.35 ms (the copies then queue behind one another);
    // 16 k chunks cost more in runtime calls than they gain in overlap (0.39 ms); helper threads for the hand-over do not pay (0.30 ms).
    constexpr size_t CHUNK = 32768;
    const size_t nchunks = (n + CHUNK - 1) / CHUNK;
    if (!h->stream_down) HIPCHK(h, hipStreamCreateWithFlags(&h->stream_down, hipStreamNonBlocking));
    while (h->tp_events.size() < 2 * nchunks) {
        hipEvent_t e = nullptr;
        HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->tp_events.push_back(e);
    }
    double *h_in = h->h_tp, *h_out = h->h_tp + 4 * h->tp_cap + 16;      // pinned: pose | records ... results
    double *d_in = h->d_tp, *d_out = h->d_tp + 4 * h->tp_cap + 16;
    for (int i = 0; i < 14; ++i) h_in[i] = pose[i];
    const bool f64 = raw.dtype == CTGN_F64, tf64 = ts.dtype == CTGN_F64;
    const char *rb = static_cast<const char *>(raw.base), *tb_ = static_cast<const char *>(ts.base);
    const bool tp_timing = tuning().frame_timing != 0;      // measurement hook: host-clock marks on stderr
    const auto tp_t0 = std::chrono::steady_clock::now();
    auto tp_now = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp_t0).count(); };
    double tp_gather = 0;
    bool in_range = true;
    // an error in the middle of the pipeline must not return while copies into the pinned staging (and from it into the caller's
    // buffers) are still in flight on either stream: drain both first
#define TPCHK(call)                                                                                                  \
    do {                                                                                                             \
        hipError_t e_ = (call);                                                                                      \
        if (e_ != hipSuccess) {                                                                                      \
            (void) hipStreamSynchronize(h->stream);                                                                  \
            (void) hipStreamSynchronize(h->stream_down);                                                             \
            return fail(h, e_ == hipErrorOutOfMemory ? CTGN_ERR_OUT_OF_MEMORY : CTGN_ERR_HIP,                        \
                        std::string("[HIP] ") + #call + " -> " + hipGetErrorString(e_));                             \
        }                                                                                                            \
    } while (0)
    for (size_t k = 0; k < nchunks && in_range; ++k) {
        const size_t j0 = k * CHUNK, j1 = std::min(n, j0 + CHUNK);
        const double tg = tp_timing ? tp_now() : 0;
        for (size_t j = j0; j < j1; ++j) {
            double *q = h_in + 16 + 4 * j;
            if (f64) { const double *p = reinterpret_cast<const double *>(rb + j * raw.stride_bytes); q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
            else { const float *p = reinterpret_cast<const float *>(rb + j * raw.stride_bytes); q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
            const double t = tf64 ? *reinterpret_cast<const double *>(tb_ + j * ts.stride_bytes) : (double) *reinterpret_cast<const float *>(tb_ + j * ts.stride_bytes);
            q[3] = t;
            in_range = in_range && (tbe[0] <= t && t <= tbe[1]);
        }
        if (tp_timing) tp_gather += tp_now() - tg;
        if (!in_range) break;
        const size_t lo = k == 0 ? 0 : 16 + 4 * j0, hi = 16 + 4 * j1;                       // the first chunk carries the pose
        TPCHK(hipMemcpyAsync(d_in + lo, h_in + lo, (hi - lo) * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_transform_points, dim3(grid_for(j1 - j0)), dim3(256), 0, h->stream, d_in + 16 + 4 * j0, d_out + 3 * j0, (int) (j1 - j0), (size_t) 1,
                           d_in, tbe[0], tbe[1], (const uint32_t *) nullptr, (size_t) 0, (size_t) 4, 1);
        TPCHK(hipGetLastError());
        // the result travels back on a second stream, beside the uploads of the chunks behind it
        TPCHK(hipEventRecord(h->tp_events[nchunks + k], h->stream));
        TPCHK(hipStreamWaitEvent(h->stream_down, h->tp_events[nchunks + k], 0));
        TPCHK(hipMemcpyAsync(h_out + 3 * j0, d_out + 3 * j0, 3 * (j1 - j0) * sizeof(double), hipMemcpyDeviceToHost, h->stream_down));
        TPCHK(hipEventRecord(h->tp_events[k], h->stream_down));
    }
#undef TPCHK
    const double tp_enq = tp_timing ? tp_now() : 0;
    if (!in_range) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream_down));
        return fail(h, CTGN_ERR_TIMESTAMP_RANGE, "point timestamps must lie in [t_begin, t_end]");
    }
    std::atomic<bool> wait_failed{false};
    for (size_t k = 0; k < nchunks; ++k) {
        const size_t j0 = k * CHUNK, j1 = std::min(n, j0 + CHUNK);
        if (hipEventSynchronize(h->tp_events[k]) != hipSuccess) { wait_failed.store(true); break; }
        if (out_dtype == CTGN_F64 && out_stride == 3 * sizeof(double)) {
            std::memcpy(static_cast<char *>(out_base) + j0 * out_stride, h_out + 3 * j0, 3 * (j1 - j0) * sizeof(double));
        } else {
            for (size_t j = j0; j < j1; ++j) write_point(out_base, out_stride, out_dtype, j, h_out[3 * j], h_out[3 * j + 1], h_out[3 * j + 2]);
        }
    }
    if (tp_timing)
        std::fprintf(stderr, "[ctgn] transform_points us: gather %.0f | all enqueued at %.0f | hand-over done at %.0f (n %zu, %zu chunks)\n",
                     tp_gather, tp_enq, tp_now(), n, nchunks);
    HIPCHK(h, hipStreamSynchronize(h->stream_down));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (wait_failed.load()) return fail(h, CTGN_ERR_HIP, "[HIP] hipEventSynchronize");
    return CTGN_OK;
}

ctgn_status ctgn_register(ctgn_handle h, ctgn_view raw, void *world_base, size_t world_stride, ctgn_dtype world_dtype,
                          ctgn_view ts, size_t n, double pose_io[14], const double tbe[2], const ctgn_options *opts,
                          const ctgn_motion_prior *prior, ctgn_summary *summary) {
    ctgn_view world{world_base, world_stride, world_dtype, 0};
    if (h) h->pose_with_kp = pose_io;              // host views: the pose rides in with the keypoints, one upload
    ctgn_status st = ctgn_set_keypoints(h, raw, world, ts, n);
    if (h) h->pose_with_kp = nullptr;
    if (st != CTGN_OK) { if (h) h->pose_on_device = false; return st; }
    const bool dev_world = n > 0 && on_device(world_base);
    h->prefetch_world = !dev_world;                // host views: world points ride back with the final state, one synchronisation
    st = ctgn_solve(h, pose_io, tbe, opts, prior, summary);
    h->pose_on_device = false;
    h->prefetch_world = false;
    if (st != CTGN_OK) return st;
    if (dev_world) return ctgn_get_world_points(h, world_base, world_stride, world_dtype, n);
    if (n) scatter_world_from_staging(h, world_base, world_stride, world_dtype, n);
    return CTGN_OK;
}



/* -------------------------------------------------------------------------------------------------
 * Frame pipeline: scan resident on the device from the samplers to the map update
 * ---------------------------------------------------------------------------------------------- */
void ctgn_frame_options_default(ctgn_frame_options *o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->frame_voxel_size = 0.5;               // OdometryOptions::voxel_size (odometry.h)
    o->sample_voxel_size = 1.5;              // OdometryOptions::sample_voxel_size
    o->max_num_keypoints = -1;
}

static ctgn_status frame_reserve(ctgn_handle h, size_t n) {
    auto &F = h->fr;
    if (n <= F.cap) return CTGN_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    frame_scratch_free(h);
    const size_t cap = ((n + n / 4 + 1024) + 63) & ~(size_t) 63;
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_scan), (4 * cap + 16) * sizeof(double)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_world), 3 * cap * sizeof(double)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_corr), 3 * cap * sizeof(double)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_flag1), cap));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_flag2), cap));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_sel1), cap * sizeof(uint32_t)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_sel2), cap * sizeof(uint32_t)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_counts), 4 * sizeof(int)));      // sampled, keypoints, bad order
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&F.h_scan), (4 * cap + 16) * sizeof(double), hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&F.h_out), 6 * cap * sizeof(double), hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&F.h_sel), 2 * cap * sizeof(uint32_t), hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&F.h_counts), 4 * sizeof(int), hipHostMallocDefault));
    F.cap = cap;
    return CTGN_OK;
}


// `fused_max_distance` (ctgn_frame on the GN route): the map update of this frame — far-voxel eviction round the NEW end pose, then
// the insertion of the undistorted sampled frame unless the registration failed — is enqueued right behind the undistortion, with the
// location and the gate read from the device's own state, while the frame's outputs travel to the host on a second stream and are
// handed over there: the update no longer waits for a host round trip, and the hand-over no longer waits for the update.
static ctgn_status frame_register_impl(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const uint32_t *order,
                                       const ctgn_frame_options *fo, double pose_io[14], const double tbe[2], const ctgn_options *opts,
                                       const ctgn_motion_prior *prior, const ctgn_robust_options *robust, const ctgn_robust_prior *robust_prior,
                                       ctgn_frame_outputs *out, ctgn_summary *summary, const double *fused_max_distance);

ctgn_status ctgn_frame_register(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const uint32_t *order,
                                const ctgn_frame_options *fo, double pose_io[14], const double tbe[2], const ctgn_options *opts,
                                const ctgn_motion_prior *prior, const ctgn_robust_options *robust, const ctgn_robust_prior *robust_prior,
                                ctgn_frame_outputs *out, ctgn_summary *summary) {
    return frame_register_impl(h, raw, ts, n, order, fo, pose_io, tbe, opts, prior, robust, robust_prior, out, summary, nullptr);
}

// the read-backs a fused map update deferred: counters of every level, then the checks devmap_insert_staged makes
static ctgn_status frame_finish_map_update(ctgn_handle h) {
    h->map_update_pending = false;
    bool range_error = false, overflow = false;
    hipError_t e = hipStreamSynchronize(h->stream);
    for (auto &DL : h->devlevels) {
        if (e == hipSuccess) e = devmap_level_read_counters(DL, h->stream);
        range_error = range_error || DL.host.range_error;
        overflow = overflow || DL.host.overflow;
    }
    if (e != hipSuccess) return fail(h, CTGN_ERR_HIP, std::string("[HIP] frame map update -> ") + hipGetErrorString(e));
    if (overflow) return fail(h, CTGN_ERR_HIP, "device map capacity exhausted (internal sizing error)");
    if (range_error) {
        for (auto &DL : h->devlevels) (void) hipMemsetAsync(&DL.counters->range_error, 0, sizeof(unsigned int), h->stream);
        h->insert_calls_with_skips++;
        h->last_error = "a point fell outside the 21-bit voxel key range (or was not finite) and was skipped";
    }
    return CTGN_OK;
}

// every scan point's world point from the pinned read-back (F.h_out: x y z rows in the CALLER's numbering — the undistortion kernel writes
// them through the frame's order) into the rows of the caller's array, a chunk per helper thread
static void frame_scatter_all(ctgn_handle h, size_t n, const ctgn_frame_outputs *out, bool plain_rows) {
    auto &F = h->fr;
    constexpr size_t CHUNK = 16384;
    char *ob = static_cast<char *>(out->all_world_base);
    const size_t os = out->all_world_stride_bytes;
    const bool o64 = out->all_world_dtype == CTGN_F64;
    const double *w = F.h_out;
    h->pool.run((n + CHUNK - 1) / CHUNK, [&](size_t k) {
        const size_t j0 = k * CHUNK, j1 = std::min(n, j0 + CHUNK);
        if (plain_rows) {
            std::memcpy(ob + j0 * os, w + 3 * j0, (j1 - j0) * 3 * sizeof(double));
        } else if (o64) {
            for (size_t j = j0; j < j1; ++j) {
                double *q = reinterpret_cast<double *>(ob + j * os);
                q[0] = w[3 * j]; q[1] = w[3 * j + 1]; q[2] = w[3 * j + 2];
            }
        } else {
            for (size_t j = j0; j < j1; ++j) {
                float *q = reinterpret_cast<float *>(ob + j * os);
                q[0] = (float) w[3 * j]; q[1] = (float) w[3 * j + 1]; q[2] = (float) w[3 * j + 2];
            }
        }
    });
}

// Stage one scan for the frame pipeline: the caller's records -> x y z t records in processing order behind the pose (F.d_scan), uploaded
// chunk by chunk while the next chunk is being staged; timestamp range -> F.tmin / F.tmax, checked against [t_begin, t_end]. Everything is
// enqueued on the handle's stream; nothing of an earlier frame is valid afterwards.
// phase 0: all of it. phase 1 (ctgn_frame_stage): the upload only — the records go to d_scan_in whatever comes, and the call returns with the
// copy in flight. phase 2 (ctgn_frame_begin on a pre-staged scan): the processing order only — permute (or copy) d_scan_in into d_scan.
static ctgn_status frame_stage(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const uint32_t *order, const ctgn_frame_options *fo,
                               const double pose_io[14], const double tbe[2], int phase = 0) {
    auto &F = h->fr;
    if (phase == 2) {
        if (!F.prestaged || n != F.n) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "no scan of this size was staged (ctgn_frame_stage)");
        F.prestaged = false;
            const bool dev_shuffle = n > 1 && !order && fo->shuffle_seed != 0;
        F.device_shuffled = dev_shuffle;
        if (n && !(tbe[0] <= F.tmin && F.tmax <= tbe[1])) {
            hipStreamSynchronize(h->stream);
            return fail(h, CTGN_ERR_TIMESTAMP_RANGE, "point timestamps must lie in [t_begin, t_end]");
        }
        for (int k = 0; k < 14; ++k) F.h_scan[k] = pose_io[k];                      // (the header's earlier copy has been waited for: see phase 1)
        HIPCHK(h, hipMemcpyAsync(F.d_scan, F.h_scan, 16 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(F.d_counts + 2, 0, sizeof(int), h->stream));
        if (!(dev_shuffle || (order && n))) {
            if (n) HIPCHK(h, hipMemcpyAsync(F.d_scan + 16, F.d_scan_in, 4 * n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            F.permuted = false;
            return CTGN_OK;
        }
        int half_bits = 1;
        while (((size_t) 1 << (2 * half_bits)) < n) ++half_bits;
        if (order) {
            std::memcpy(F.h_order, order, n * sizeof(uint32_t));
            HIPCHK(h, hipMemcpyAsync(F.d_order, F.h_order, n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemsetAsync(h->dm.idx, 0, n * sizeof(uint32_t), h->stream));
        }
        hipLaunchKernelGGL(k_frame_permute, dim3(grid_for(n)), dim3(256), 0, h->stream, (const double *) F.d_scan_in, F.d_scan + 16,
                           order ? (const uint32_t *) F.d_order : (const uint32_t *) nullptr, F.d_order, (int) n, half_bits,
                           (unsigned long long) fo->shuffle_seed, reinterpret_cast<unsigned int *>(h->dm.idx), F.d_counts + 2);
        HIPCHK(h, hipGetLastError());
        F.permuted = true;
        return CTGN_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));       // pinned staging reuse
    F.prestaged = false;
    {
        ctgn_status rs = frame_reserve(h, std::max<size_t>(n, 1));     // an empty scan still stages the pose
        if (rs != CTGN_OK) return rs;
        DMCHK(h, devmap_scratch_reserve(h->dm, std::max<size_t>(n, 1)));
    }
    // ---- stage: caller's records -> x y z t records in processing order behind the pose, uploaded chunk by chunk while the next
    // chunk is being staged
    const size_t c = std::min((n + 63) & ~(size_t) 63, F.cap);                      // plane stride of the undistorted outputs
    F.valid = false;
    F.staged = false;
    F.stride = c; F.n = n; F.n1 = 0; F.n2 = 0;
    // a processing order — the caller's `order`, or the shuffle made on the device (ctgn_frame_options::shuffle_seed) — is applied ON THE
    // DEVICE: the scan travels in scan order (staged by sequential reads; gathering 132 k records through a shuffled index on the host
    // cost 0.8 ms of the call) and k_frame_permute deals the records, checking a caller's order for being a permutation on the way
    const bool dev_shuffle = phase == 0 && n > 1 && !order && fo->shuffle_seed != 0;
    const bool permute = phase == 1 || dev_shuffle || (order && n);
    F.device_shuffled = dev_shuffle;
    if (permute && !F.d_scan_in) {
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_scan_in), 4 * F.cap * sizeof(double)));
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_order), F.cap * sizeof(uint32_t)));
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&F.h_order), F.cap * sizeof(uint32_t), hipHostMallocDefault));
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&F.d_selx), 2 * F.cap * sizeof(uint32_t)));
    }
    double *d_recs_up = permute ? F.d_scan_in : F.d_scan + 16;                       // where the uploaded records go
    for (int k = 0; k < 14; ++k) F.h_scan[k] = pose_io[k];
    double *hs = F.h_scan + 16;
    double tmin = INFINITY, tmax = -INFINITY;
    constexpr size_t CHUNK = 16384;
    const bool f64 = raw.dtype == CTGN_F64, tf64 = ts.dtype == CTGN_F64;
    const char *rb = static_cast<const char *>(raw.base), *tb_ = static_cast<const char *>(ts.base);
    const size_t nchunks = std::max<size_t>(1, (n + CHUNK - 1) / CHUNK);
    struct ChunkStat { double mn, mx; bool has_nan, bad_order; };
    std::vector<ChunkStat> stat(nchunks, ChunkStat{INFINITY, -INFINITY, false, false});
    // Page-locked caller arrays in the plain layout (x y z rows of doubles, timestamps as doubles, scan order) are not staged: the DMA
    // engine reads them where they lie and a kernel writes the x y z t records (a driver that fills such a buffer from its sensor
    // packets saves the 4 MB staging copy of a 132 k-point scan). tuning frame_no_direct = 1: always stage (measurement hook).
    const bool no_direct = tuning().frame_no_direct != 0;
    const bool direct_in = n && !no_direct && f64 && raw.stride_bytes == 3 * sizeof(double) &&
                           (fo->override_timestamps || (tf64 && ts.stride_bytes == sizeof(double))) && host_pinned(raw.base, n * 3 * sizeof(double)) &&
                           (fo->override_timestamps || host_pinned(ts.base, n * sizeof(double)));
    const std::function<void(size_t)> stage_chunk = [&](size_t k) {
        const size_t j0 = k * CHUNK, j1 = std::min(n, j0 + CHUNK);
        // four independent min / max chains: one chain is a 4-cycle dependency per point and was what the loop ran at
        double mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        bool has_nan = false;
        const bool bad_order = false;
        auto stage_one = [&](size_t j, int u) {
            const size_t i = j;
            double *q = hs + 4 * j;
            if (f64) { const double *p = reinterpret_cast<const double *>(rb + i * raw.stride_bytes); q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
            else { const float *p = reinterpret_cast<const float *>(rb + i * raw.stride_bytes); q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
            const double t = fo->override_timestamps ? fo->override_timestamp
                             : tf64 ? *reinterpret_cast<const double *>(tb_ + i * ts.stride_bytes)
                                    : (double) *reinterpret_cast<const float *>(tb_ + i * ts.stride_bytes);
            q[3] = t;
            mn[u] = t < mn[u] ? t : mn[u];
            mx[u] = t > mx[u] ? t : mx[u];
            has_nan = has_nan || t != t;
        };
        size_t j = j0;
        for (; j + 4 <= j1; j += 4) { stage_one(j, 0); stage_one(j + 1, 1); stage_one(j + 2, 2); stage_one(j + 3, 3); }
        for (; j < j1; ++j) stage_one(j, 0);
        stat[k] = ChunkStat{std::min(std::min(mn[0], mn[1]), std::min(mn[2], mn[3])), std::max(std::max(mx[0], mx[1]), std::max(mx[2], mx[3])), has_nan, bad_order};
    };
    // a group of chunks is staged by the helper threads + this one, then uploaded while the next group is being staged
    const int helpers = n >= 4 * CHUNK ? host_helpers_wanted() : 0;
    h->pool.ensure(helpers);
    const size_t group = (size_t) helpers + 1;
    bool has_nan = false;
    if (direct_in) {
        if (fo->override_timestamps) {
            tmin = tmax = fo->override_timestamp;
            has_nan = tmin != tmin;
        } else {
            const double *tp = reinterpret_cast<const double *>(tb_);
            h->pool.run(nchunks, [&](size_t k) {
                const size_t j0 = k * CHUNK, j1 = std::min(n, j0 + CHUNK);
                double mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                bool nan = false;
                size_t j = j0;
                for (; j + 4 <= j1; j += 4)
                    for (int u = 0; u < 4; ++u) {
                        const double t = tp[j + u];
                        mn[u] = t < mn[u] ? t : mn[u];
                        mx[u] = t > mx[u] ? t : mx[u];
                        nan = nan || t != t;
                    }
                for (; j < j1; ++j) { const double t = tp[j]; mn[0] = t < mn[0] ? t : mn[0]; mx[0] = t > mx[0] ? t : mx[0]; nan = nan || t != t; }
                stat[k] = ChunkStat{std::min(std::min(mn[0], mn[1]), std::min(mn[2], mn[3])), std::max(std::max(mx[0], mx[1]), std::max(mx[2], mx[3])), nan, false};
            });
            for (size_t k = 0; k < nchunks; ++k) {
                has_nan = has_nan || stat[k].has_nan;
                tmin = std::min(tmin, stat[k].mn);
                tmax = std::max(tmax, stat[k].mx);
            }
        }
        // the output planes are free until the undistortion: the rows and the timestamps land there, the records are written from them
        HIPCHK(h, hipMemcpyAsync(F.d_scan, F.h_scan, 16 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(F.d_world, raw.base, n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (!fo->override_timestamps) HIPCHK(h, hipMemcpyAsync(F.d_corr, ts.base, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_frame_records, dim3(grid_for(n)), dim3(256), 0, h->stream, F.d_world, fo->override_timestamps ? (const double *) nullptr : F.d_corr,
                           fo->override_timestamp, (int) n, d_recs_up);
        HIPCHK(h, hipGetLastError());
    } else
    for (size_t g0 = 0; g0 < nchunks; g0 += group) {
        const size_t g1 = std::min(nchunks, g0 + group);
        h->pool.run(g1 - g0, [&](size_t i) { stage_chunk(g0 + i); });
        for (size_t k = g0; k < g1; ++k) {
            has_nan = has_nan || stat[k].has_nan;
            tmin = std::min(tmin, stat[k].mn);
            tmax = std::max(tmax, stat[k].mx);
        }
        const size_t j0 = g0 * CHUNK, j1 = std::min(n, g1 * CHUNK);
        if (permute) {
            if (g0 == 0) HIPCHK(h, hipMemcpyAsync(F.d_scan, F.h_scan, 16 * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(d_recs_up + 4 * j0, F.h_scan + 16 + 4 * j0, 4 * (j1 - j0) * sizeof(double), hipMemcpyHostToDevice, h->stream));
            continue;
        }
        const size_t lo = g0 == 0 ? 0 : 16 + 4 * j0, hi = 16 + 4 * j1;              // the first group carries the pose
        HIPCHK(h, hipMemcpyAsync(F.d_scan + lo, F.h_scan + lo, (hi - lo) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipMemsetAsync(F.d_counts + 2, 0, sizeof(int), h->stream));            // "order is not a permutation"
    if (phase == 1) {                                  // the order comes with ctgn_frame_begin
        if (has_nan) tmax = NAN;
        F.tmin = tmin; F.tmax = tmax; F.direct_in = direct_in;
        F.prestaged = true;
        return CTGN_OK;
    }
    if (permute) {
        int half_bits = 1;
        while (((size_t) 1 << (2 * half_bits)) < n) ++half_bits;
        if (order) {                                   // the caller's order: to the device through the pinned copy, checked there
            std::memcpy(F.h_order, order, n * sizeof(uint32_t));
            HIPCHK(h, hipMemcpyAsync(F.d_order, F.h_order, n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemsetAsync(h->dm.idx, 0, n * sizeof(uint32_t), h->stream));     // free until the samplers run
        }
        hipLaunchKernelGGL(k_frame_permute, dim3(grid_for(n)), dim3(256), 0, h->stream, (const double *) F.d_scan_in, F.d_scan + 16,
                           order ? (const uint32_t *) F.d_order : (const uint32_t *) nullptr, F.d_order, (int) n, half_bits,
                           (unsigned long long) fo->shuffle_seed, reinterpret_cast<unsigned int *>(h->dm.idx), F.d_counts + 2);
        HIPCHK(h, hipGetLastError());
    }
    F.permuted = permute;
    if (has_nan) tmax = NAN;
    // every point is undistorted below: InterpolatePose CHECKs begin <= t <= end for each (types.h:456)
    if (n && !(tbe[0] <= tmin && tmax <= tbe[1])) {
        hipStreamSynchronize(h->stream);
        return fail(h, CTGN_ERR_TIMESTAMP_RANGE, "point timestamps must lie in [t_begin, t_end]");
    }
    F.tmin = tmin; F.tmax = tmax; F.direct_in = direct_in;
    return CTGN_OK;
}

static ctgn_status frame_register_body(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const uint32_t *order,
                                       const ctgn_frame_options *fo, double pose_io[14], const double tbe[2], const ctgn_options *opts,
                                       const ctgn_motion_prior *prior, const ctgn_robust_options *robust, const ctgn_robust_prior *robust_prior,
                                       ctgn_frame_outputs *out, ctgn_summary *summary, const double *fused_max_distance);
// An error return must not leave transfers in flight: with page-locked caller arrays the DMA engine reads and writes the CALLER's
// memory, which the caller may free as soon as the call is back.
static ctgn_status frame_register_impl(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const uint32_t *order,
                                       const ctgn_frame_options *fo, double pose_io[14], const double tbe[2], const ctgn_options *opts,
                                       const ctgn_motion_prior *prior, const ctgn_robust_options *robust, const ctgn_robust_prior *robust_prior,
                                       ctgn_frame_outputs *out, ctgn_summary *summary, const double *fused_max_distance) {
    const ctgn_status st = frame_register_body(h, raw, ts, n, order, fo, pose_io, tbe, opts, prior, robust, robust_prior, out, summary, fused_max_distance);
    if (st != CTGN_OK && h && h->device >= 0) {
        (void) hipStreamSynchronize(h->stream);
        if (h->stream_down) (void) hipStreamSynchronize(h->stream_down);
        (void) hipGetLastError();
    }
    return st;
}
static ctgn_status frame_register_body(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const uint32_t *order,
                                       const ctgn_frame_options *fo, double pose_io[14], const double tbe[2], const ctgn_options *opts,
                                       const ctgn_motion_prior *prior, const ctgn_robust_options *robust, const ctgn_robust_prior *robust_prior,
                                       ctgn_frame_outputs *out, ctgn_summary *summary, const double *fused_max_distance) {
    NEED_DEVICE(h);
    if (summary) std::memset(summary, 0, sizeof(*summary));
    if (out) { out->num_sampled = 0; out->num_keypoints = 0; }
    if (!fo || !pose_io || !tbe || (!opts && !robust)) return CTGN_ERR_INVALID_ARGUMENT;
    if (n && (!raw.base || (!ts.base && !fo->override_timestamps))) return CTGN_ERR_INVALID_ARGUMENT;
    if (n > (size_t) 1 << 30) return fail(h, CTGN_ERR_UNSUPPORTED, "too many points");
    if (n && (on_device(raw.base) || on_device(ts.base)))
        return fail(h, CTGN_ERR_UNSUPPORTED, "ctgn_frame_register takes host views (the stage entry points accept device memory)");
    const auto t_call = std::chrono::steady_clock::now();
    // CTGN_TUNING="frame_timing=1": host-clock marks of the call's phases on stderr (measurement hook; no extra synchronisation)
    const bool timing = tuning().frame_timing != 0;
    double marks[8] = {0};
    auto mark = [&](int k) { if (timing) marks[k] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count(); };
    auto &F = h->fr;
    {
        ctgn_status ss = frame_stage(h, raw, ts, n, order, fo, pose_io, tbe);
        if (ss != CTGN_OK) return ss;
    }
    const size_t c = F.stride;                                                      // plane stride of the undistorted outputs
    const double *hs = F.h_scan + 16;
    const double *d_recs = F.d_scan + 16;
    const double tmin = F.tmin, tmax = F.tmax;
    const bool direct_in = F.direct_in;
    const bool no_direct = tuning().frame_no_direct != 0;
    const char *rb = static_cast<const char *>(raw.base);
    mark(0);                                          // staged + upload enqueued
    // ---- both samplers, then the one read-back that sizes the launches
    DMCHK(h, devmap_frame_sampling(h->dm, d_recs, 1, 4, n, fo->frame_voxel_size, fo->sample_voxel_size, F.d_flag1, F.d_flag2, F.d_sel1, F.d_sel2,
                                   F.d_counts, h->stream));
    HIPCHK(h, hipMemcpyAsync(F.h_counts, F.d_counts, 3 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    mark(1);                                          // samplers enqueued
    HIPCHK(h, hipStreamSynchronize(h->stream));
    mark(2);                                          // counts on the host
    if (F.h_counts[2]) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "order must be a permutation of 0..n-1");
    const size_t n1 = (size_t) F.h_counts[0];
    size_t n2 = (size_t) F.h_counts[1];
    if (fo->max_num_keypoints > 0 && n2 > (size_t) fo->max_num_keypoints) n2 = (size_t) fo->max_num_keypoints;
    F.n1 = n1; F.n2 = n2;
    // ---- keypoints: gathered into the solver's arrays, world points from the initial estimate (odometry.cpp:374-378)
    h->pose_on_device = false;
    {
        ctgn_status rs = reserve_keypoints(h, n2);
        if (rs != CTGN_OK) return rs;
    }
    const size_t kc = (size_t) h->kp_stride;
    if (n2) {
        hipLaunchKernelGGL(k_frame_keypoints, dim3(grid_for(n2)), dim3(256), 0, h->stream, d_recs, F.d_sel2, (int) n2, h->d_kp, kc);
        hipLaunchKernelGGL(k_transform_points, dim3(grid_for(n2)), dim3(256), 0, h->stream, h->d_kp, h->d_kp + 4 * kc, (int) n2, kc,
                           F.d_scan, tbe[0], tbe[1]);
        HIPCHK(h, hipGetLastError());
    }
    h->t_min = tmin; h->t_max = tmax;
    h->kp_coherent = false;
    if (n2 >= 32768 && !F.permuted) {                   // (under a processing order the staged records are in scan order: a shuffle is incoherent)
        // spatial coherence of the keypoint order (see ctgn_set_keypoints), probed on the staged raw points — a rigid motion keeps
        // neighbours neighbours: pairs one keypoint spacing apart in processing order
        int map_id, nb;
        double res;
        search_params(h->levels, h->opts.default_radius, &map_id, &res, &nb);
        const size_t gap = std::max<size_t>(1, n / n2), step = std::max<size_t>(1, (n - gap) / 4096);
        size_t pairs = 0, near = 0;
        for (size_t i = 0; i + gap < n; i += step, ++pairs) {
            bool ok = true;
            for (int a = 0; a < 3 && ok; ++a) {
                const double p0 = direct_in ? reinterpret_cast<const double *>(rb)[3 * i + a] : hs[4 * i + a];
                const double p1 = direct_in ? reinterpret_cast<const double *>(rb)[3 * (i + gap) + a] : hs[4 * (i + gap) + a];
                ok = std::abs(voxel_coord(p0, res) - voxel_coord(p1, res)) <= 1;
            }
            near += ok ? 1 : 0;
        }
        h->kp_coherent = pairs > 0 && 2 * near >= pairs;
    }
    {
        ctgn_status ds = ensure_debug(h);
        if (ds != CTGN_OK) return ds;
    }
    // ---- registration
    ctgn_status st;
    if (robust) {
        st = ctgn_solve_robust(h, pose_io, tbe, robust, robust_prior, summary);      // host-driven LM loop; the final pose stays in d_state
        if (st != CTGN_OK) return st;
    } else {
        st = ctgn_gn_begin(h, pose_io, tbe, opts, prior);
        MapView mv;
        if (st == CTGN_OK) st = make_map_view(h, -1.0, &mv);
        if (st == CTGN_OK && opts->num_iters_icp > 0 && persistent_ok(h, mv)) {
            st = launch_persistent(h, mv, opts->num_iters_icp, false, nullptr);        // the keypoints' world points are not an output here
        } else if (st == CTGN_OK) {
            st = launch_gn_iterations(h, mv, opts->num_iters_icp);
        }
        if (st == CTGN_OK) st = flush_state_init(h);                                 // num_iters_icp 0: the state is still to be written
        if (st != CTGN_OK) {
            h->gn_active = false;
            hipStreamSynchronize(h->stream);
            if (summary) std::snprintf(summary->error_log, sizeof(summary->error_log), "%s", ctgn_last_error(h));
            return st;
        }
        HIPCHK(h, hipEventRecord(h->ev_loop_stop, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->h_state, h->d_state, sizeof(GnState), hipMemcpyDeviceToHost, h->stream));
    }
    mark(3);                                          // registration enqueued (GN) / done (robust)
    // ---- undistortion with the device's own copy of the final poses (odometry.cpp:461-486): the sampled frame (the map update's
    // input) always, every scan point when asked for
    const double *d_pose = reinterpret_cast<const double *>(reinterpret_cast<const char *>(h->d_state) + offsetof(GnState, pose));
    const bool want_all = out && out->all_world_base && n;
    if (n1) hipLaunchKernelGGL(k_transform_points, dim3(grid_for(n1)), dim3(256), 0, h->stream, d_recs, F.d_corr, (int) n1, (size_t) 1, d_pose,
                               tbe[0], tbe[1], F.d_sel1, c, (size_t) 4);
    // every scan point: as x y z rows in the caller's numbering (the kernel writes through the frame's order): when that is what the
    // caller's array holds (float64 rows of 24 bytes) the copy lands in the final layout and the hand-over is a straight memcpy
    const bool all_rows = want_all && out->all_world_dtype == CTGN_F64 && out->all_world_stride_bytes == 3 * sizeof(double);
    if (want_all) hipLaunchKernelGGL(k_transform_points, dim3(grid_for(n)), dim3(256), 0, h->stream, d_recs, F.d_world, (int) n, (size_t) 1, d_pose,
                                     tbe[0], tbe[1], (const uint32_t *) nullptr, c, (size_t) 4, 1, F.permuted ? (const uint32_t *) F.d_order : (const uint32_t *) nullptr);
    // the indices this call reports are the caller's point numbers: under a processing order the positions are translated on the device
    // (the host never needs the order — a device-made shuffle never leaves the device)
    const uint32_t *d_idx1 = F.d_sel1, *d_idx2 = F.d_sel2;
    if (F.permuted) {
        if (out && out->sampled_indices && n1) { hipLaunchKernelGGL(k_frame_translate, dim3(grid_for(n1)), dim3(256), 0, h->stream, (const uint32_t *) F.d_sel1, (const uint32_t *) F.d_order, (int) n1, F.d_selx); d_idx1 = F.d_selx; }
        if (out && out->keypoint_indices && n2) { hipLaunchKernelGGL(k_frame_translate, dim3(grid_for(n2)), dim3(256), 0, h->stream, (const uint32_t *) F.d_sel2, (const uint32_t *) F.d_order, (int) n2, F.d_selx + F.cap); d_idx2 = F.d_selx + F.cap; }
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipGetLastError());
    const bool fuse = fused_max_distance != nullptr && !robust && h->update_mode == 1;
    hipStream_t s_out = h->stream;
    if (fuse) {
        // the outputs go home on the second stream; the map update follows the undistortion on the first
        if (!h->stream_down) HIPCHK(h, hipStreamCreateWithFlags(&h->stream_down, hipStreamNonBlocking));
        if (!h->ev_frame) HIPCHK(h, hipEventCreateWithFlags(&h->ev_frame, hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->ev_frame, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream_down, h->ev_frame, 0));
        s_out = h->stream_down;
    }
    // ... and straight into the caller's array when that is page-locked
    const bool direct_out = all_rows && !no_direct && host_pinned(out->all_world_base, n * 3 * sizeof(double));
    if (want_all)
        HIPCHK(h, hipMemcpyAsync(direct_out ? static_cast<double *>(out->all_world_base) : F.h_out, F.d_world, 3 * n * sizeof(double),
                                 hipMemcpyDeviceToHost, s_out));
    if (out && out->sampled_world_base && n1)
        HIPCHK(h, hipMemcpyAsync(F.h_out + 3 * c, F.d_corr, (2 * c + n1) * sizeof(double), hipMemcpyDeviceToHost, s_out));
    if (out && out->sampled_indices && n1)
        HIPCHK(h, hipMemcpyAsync(F.h_sel, d_idx1, n1 * sizeof(uint32_t), hipMemcpyDeviceToHost, s_out));
    if (out && out->keypoint_indices && n2)
        HIPCHK(h, hipMemcpyAsync(F.h_sel + c, d_idx2, n2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s_out));
    bool update_pending = false;
    if (fuse) {
        // odometry.cpp:936-952 on the device's own copy of the new pose: eviction round its end translation, then the sampled frame —
        // unless the registration failed (GnState::failed, read by the insert kernel itself)
        const int *d_failed = reinterpret_cast<const int *>(reinterpret_cast<const char *>(h->d_state) + offsetof(GnState, failed));
        h->kth_fresh = false;
        update_pending = true;
        hipError_t e = hipSuccess;
        for (auto &DL : h->devlevels)
            if (e == hipSuccess) e = devmap_level_remove_far_enqueue(DL, d_pose + 11, *fused_max_distance, h->stream, d_failed);
        if (e == hipSuccess && n1) {
            DevMapScratch &S = h->dm;
            double *own_pts = S.pts;
            const size_t own_stride = S.stride;
            S.pts = F.d_corr;                          // the undistorted sampled frame is the batch: nothing is copied
            S.stride = F.stride;
            e = hipMemsetAsync(S.inserted, 0, n1, h->stream);
            for (auto &DL : h->devlevels)
                if (e == hipSuccess) e = devmap_level_insert_enqueue(DL, S, n1, d_failed, h->stream);
            S.pts = own_pts;
            S.stride = own_stride;
        }
        if (e != hipSuccess) {
            (void) hipStreamSynchronize(h->stream_down);
            (void) frame_finish_map_update(h);
            return fail(h, CTGN_ERR_HIP, std::string("[HIP] frame map update -> ") + hipGetErrorString(e));
        }
    }
    mark(4);                                          // undistortion + read-backs (+ map update) enqueued
    // the final state was enqueued before the undistortion: with its event complete it is on the host (a stream that holds nothing
    // but a wait is not a reliable thing to synchronise with); then the outputs
    if (fuse) HIPCHK(h, hipEventSynchronize(h->ev_frame));
    HIPCHK(h, hipStreamSynchronize(s_out));           // the outputs — and, enqueued before them, the final state — are on the host
    mark(5);
    if (!robust) {
        st = gn_collect(h, pose_io, summary);
        if (st != CTGN_OK) {
            if (update_pending) (void) frame_finish_map_update(h);
            return st;
        }
    }
    if (summary) summary->duration_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    if (out) {
        out->num_sampled = n1;
        out->num_keypoints = n2;
        out->num_keypoint_candidates = (uint64_t) F.h_counts[1];
        if (want_all && !direct_out) frame_scatter_all(h, n, out, all_rows);
        if (out->sampled_world_base)
            for (size_t k = 0; k < n1; ++k)
                write_point(out->sampled_world_base, out->sampled_world_stride_bytes, out->sampled_world_dtype, k, F.h_out[3 * c + k],
                            F.h_out[4 * c + k], F.h_out[5 * c + k]);
        if (out->sampled_indices) std::memcpy(out->sampled_indices, F.h_sel, n1 * sizeof(uint32_t));
        if (out->keypoint_indices) std::memcpy(out->keypoint_indices, F.h_sel + c, n2 * sizeof(uint32_t));
    }
    F.valid = true;
    if (update_pending) {                             // (the hand-over above ran beside it)
        st = frame_finish_map_update(h);
        if (st != CTGN_OK) return st;
    }
    if (timing) {
        mark(6);
        std::fprintf(stderr, "[ctgn] frame_register us: stage+upload enqueue %.0f | enqueue samplers %.0f | wait counts %.0f | keypoints+registration "
                             "enqueue %.0f | undistort enqueue %.0f | wait %.0f | scatter outputs %.0f | total %.0f (n %zu, sampled %zu, "
                             "keypoints %zu%s%s)\n", marks[0], marks[1] - marks[0], marks[2] - marks[1], marks[3] - marks[2], marks[4] - marks[3],
                     marks[5] - marks[4], marks[6] - marks[5], marks[6], n, n1, n2, direct_in ? ", page-locked scan read in place" : "",
                     direct_out ? ", page-locked output written in place" : "");
    }
    return CTGN_OK;
}

ctgn_status ctgn_frame_update_map(ctgn_handle h, const double location[3], double max_distance, int32_t add_points, uint8_t *inserted) {
    NEED_DEVICE(h);
    if (!location) return CTGN_ERR_INVALID_ARGUMENT;
    if (h->update_mode != 1)
        return fail(h, CTGN_ERR_UNSUPPORTED, "the frame pipeline updates the device-resident map (ctgn_map_set_update_mode(h, 1))");
    auto &F = h->fr;
    h->kth_fresh = false;
    if (add_points && !F.valid) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "no registered frame is resident (ctgn_frame_register)");
    const bool timing = tuning().frame_timing != 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (!inserted && tuning().frame_defer_update != 0) {
        // nobody waits for the insert mask: eviction and insertion are enqueued and the call returns; the counters are read (and a key-range
        // note or a capacity error reported) by the next call on the handle, which finds the work done — the host's time between two frames
        // (the caller's bookkeeping, the next scan's preparation) is the map update's
        h->map_update_pending = true;
        hipError_t e = hipSuccess;
        for (auto &DL : h->devlevels)
            if (e == hipSuccess) e = devmap_level_remove_far_value_enqueue(DL, location, max_distance, h->stream);               // odometry.cpp:938-940
        if (e == hipSuccess && add_points && F.n1) {
            e = devmap_scratch_reserve(h->dm, F.n1);
            DevMapScratch &S = h->dm;
            double *own_pts = S.pts;
            const size_t own_stride = S.stride;
            S.pts = F.d_corr;                          // the undistorted sampled frame is the batch: nothing is copied
            S.stride = F.stride;
            if (e == hipSuccess) e = hipMemsetAsync(S.inserted, 0, F.n1, h->stream);
            for (auto &DL : h->devlevels)
                if (e == hipSuccess) e = devmap_level_insert_enqueue(DL, S, F.n1, nullptr, h->stream);                           // odometry.cpp:943-949
            S.pts = own_pts;
            S.stride = own_stride;
        }
        if (e != hipSuccess) {
            (void) frame_finish_map_update(h);
            return fail(h, CTGN_ERR_HIP, std::string("[HIP] frame map update -> ") + hipGetErrorString(e));
        }
        return CTGN_OK;
    }
    for (auto &DL : h->devlevels) DMCHK(h, devmap_level_remove_far(DL, location, max_distance, h->stream));   // odometry.cpp:938-940
    const auto t1 = std::chrono::steady_clock::now();
    if (!add_points || F.n1 == 0) return CTGN_OK;
    DMCHK(h, devmap_scratch_reserve(h->dm, F.n1));
    DevMapScratch &S = h->dm;
    double *own_pts = S.pts;
    const size_t own_stride = S.stride;
    S.pts = F.d_corr;                                  // the undistorted sampled frame is the batch: nothing is copied
    S.stride = F.stride;
    const ctgn_status st = devmap_insert_staged(h, F.n1, inserted);                    // odometry.cpp:943-949
    S.pts = own_pts;
    S.stride = own_stride;
    if (timing)
        std::fprintf(stderr, "[ctgn] frame_update_map us: evict %.0f | insert %.0f\n", std::chrono::duration<double, std::micro>(t1 - t0).count(),
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count());
    return st;
}

ctgn_status ctgn_frame(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const uint32_t *order, const ctgn_frame_options *fo,
                       double pose_io[14], const double tbe[2], const ctgn_options *opts, const ctgn_motion_prior *prior,
                       const ctgn_robust_options *robust, const ctgn_robust_prior *robust_prior, double max_distance,
                       ctgn_frame_outputs *out, ctgn_summary *summary) {
    ctgn_summary local;
    if (!summary) summary = &local;
    if (h && h->update_mode != 1)
        return fail(h, CTGN_ERR_UNSUPPORTED, "the frame pipeline updates the device-resident map (ctgn_map_set_update_mode(h, 1))");
    if (!robust)                   // GN route: the map update is enqueued inside, behind the undistortion (frame_register_impl)
        return frame_register_impl(h, raw, ts, n, order, fo, pose_io, tbe, opts, prior, robust, robust_prior, out, summary, &max_distance);
    ctgn_status st = ctgn_frame_register(h, raw, ts, n, order, fo, pose_io, tbe, opts, prior, robust, robust_prior, out, summary);
    if (st != CTGN_OK) return st;
    return ctgn_frame_update_map(h, pose_io + 11, max_distance, summary->success, nullptr);
}

// ---------------------------------------------------------------------------------------- the frame pipeline, step by step
// The same stages as frame_register_body behind the entry points a caller needs when it keeps Odometry::DoRegister's own control flow:
// InitializeFrame (odometry.cpp:333-382) -> ctgn_frame_begin; TryRegister (:525-601), possibly several times on the same sampled frame (the
// robust retry loop, :794-845) -> ctgn_frame_try_register; the two undistortion loops (:461-486) with the poses the HOST settled on ->
// ctgn_frame_undistort; UpdateMap (:936-952) -> ctgn_frame_update_map. integration/odometry_gpu_arm.h is that caller.
static ctgn_status frame_sample(ctgn_handle h, double frame_voxel, double kp_voxel) {
    auto &F = h->fr;
    DMCHK(h, devmap_frame_sampling(h->dm, F.d_scan + 16, 1, 4, F.n, frame_voxel, kp_voxel, F.d_flag1, F.d_flag2, F.d_sel1, F.d_sel2, F.d_counts, h->stream));
    HIPCHK(h, hipMemcpyAsync(F.h_counts, F.d_counts, 3 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (F.h_counts[2]) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "order must be a permutation of 0..n-1");
    F.n1 = (size_t) F.h_counts[0];
    F.n2 = (size_t) F.h_counts[1];
    F.frame_voxel = frame_voxel;
    F.kp_voxel = kp_voxel;
    return CTGN_OK;
}

// the first 16 doubles of the scan block are the pose slot the undistortion kernels read
static ctgn_status frame_upload_pose(ctgn_handle h, const double pose[14]) {
    auto &F = h->fr;
    for (int k = 0; k < 14; ++k) F.h_scan[k] = pose[k];
    HIPCHK(h, hipMemcpyAsync(F.d_scan, F.h_scan, 16 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return CTGN_OK;
}

static ctgn_status frame_begin_body(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const uint32_t *order, const ctgn_frame_options *fo,
                                    const double pose[14], const double tbe[2], ctgn_frame_outputs *out) {
    NEED_DEVICE(h);
    if (out) { out->num_sampled = 0; out->num_keypoints = 0; }
    if (!fo || !pose || !tbe) return CTGN_ERR_INVALID_ARGUMENT;
    const bool prestaged = n && !raw.base;             // "the scan ctgn_frame_stage uploaded"
    if (n && !prestaged && (!ts.base && !fo->override_timestamps)) return CTGN_ERR_INVALID_ARGUMENT;
    if (n > (size_t) 1 << 30) return fail(h, CTGN_ERR_UNSUPPORTED, "too many points");
    if (n && !prestaged && (on_device(raw.base) || on_device(ts.base)))
        return fail(h, CTGN_ERR_UNSUPPORTED, "ctgn_frame_begin takes host views (the stage entry points accept device memory)");
    auto &F = h->fr;
    {
        ctgn_status ss = frame_stage(h, raw, ts, n, order, fo, pose, tbe, prestaged ? 2 : 0);
        if (ss != CTGN_OK) return ss;
    }
    {
        ctgn_status ss = frame_sample(h, fo->frame_voxel_size, fo->sample_voxel_size);
        if (ss != CTGN_OK) return ss;
    }
    const size_t c = F.stride, n1 = F.n1;
    if (out && out->sampled_indices && n1) {
        const uint32_t *d_idx = F.d_sel1;
        if (F.permuted) {                              // positions -> the caller's point numbers, on the device
            hipLaunchKernelGGL(k_frame_translate, dim3(grid_for(n1)), dim3(256), 0, h->stream, (const uint32_t *) F.d_sel1, (const uint32_t *) F.d_order, (int) n1, F.d_selx);
            HIPCHK(h, hipGetLastError());
            d_idx = F.d_selx;
        }
        HIPCHK(h, hipMemcpyAsync(F.h_sel, d_idx, n1 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    }
    if (out && out->sampled_world_base && n1) {      // the sampled frame under the initial estimate (odometry.cpp:371-375)
        hipLaunchKernelGGL(k_transform_points, dim3(grid_for(n1)), dim3(256), 0, h->stream, F.d_scan + 16, F.d_corr, (int) n1, (size_t) 1,
                           (const double *) F.d_scan, tbe[0], tbe[1], (const uint32_t *) F.d_sel1, c, (size_t) 4);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(F.h_out + 3 * c, F.d_corr, (2 * c + n1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    if (out && (out->sampled_indices || out->sampled_world_base) && n1) HIPCHK(h, hipStreamSynchronize(h->stream));
    if (out) {
        out->num_sampled = n1;
        out->num_keypoints = fo->max_num_keypoints > 0 ? std::min<size_t>(F.n2, (size_t) fo->max_num_keypoints) : F.n2;
        out->num_keypoint_candidates = F.n2;
        if (out->sampled_indices) std::memcpy(out->sampled_indices, F.h_sel, n1 * sizeof(uint32_t));
        if (out->sampled_world_base)
            for (size_t k = 0; k < n1; ++k)
                write_point(out->sampled_world_base, out->sampled_world_stride_bytes, out->sampled_world_dtype, k, F.h_out[3 * c + k],
                            F.h_out[4 * c + k], F.h_out[5 * c + k]);
    }
    F.staged = true;
    return CTGN_OK;
}

ctgn_status ctgn_host_alloc(ctgn_handle h, size_t bytes, void **out) {
    NEED_DEVICE(h);
    if (!out || bytes == 0) return CTGN_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    HIPCHK(h, hipHostMalloc(out, bytes, hipHostMallocDefault));
    return CTGN_OK;
}

ctgn_status ctgn_host_free(ctgn_handle h, void *p) {
    NEED_DEVICE(h);
    if (!p) return CTGN_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));       // no transfer of this handle is still using it
    HIPCHK(h, hipHostFree(p));
    return CTGN_OK;
}

ctgn_status ctgn_frame_stage(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const ctgn_frame_options *fo, const double pose[14],
                             const double tbe[2]) {
    NEED_DEVICE(h);
    if (!fo || !pose || !tbe) return CTGN_ERR_INVALID_ARGUMENT;
    if (n && (!raw.base || (!ts.base && !fo->override_timestamps))) return CTGN_ERR_INVALID_ARGUMENT;
    if (n > (size_t) 1 << 30) return fail(h, CTGN_ERR_UNSUPPORTED, "too many points");
    if (n && (on_device(raw.base) || on_device(ts.base)))
        return fail(h, CTGN_ERR_UNSUPPORTED, "ctgn_frame_stage takes host views (the stage entry points accept device memory)");
    ctgn_status st = frame_stage(h, raw, ts, n, nullptr, fo, pose, tbe, 1);
    // the staging block and (page-locked caller arrays) the caller's memory are read by the copy engine: they are the caller's again
    // when the call returns
    if (h->device >= 0) { if (hipStreamSynchronize(h->stream) != hipSuccess && st == CTGN_OK) st = fail(h, CTGN_ERR_HIP, "[HIP] upload of the scan"); }
    if (st != CTGN_OK) h->fr.prestaged = false;
    return st;
}

ctgn_status ctgn_frame_begin(ctgn_handle h, ctgn_view raw, ctgn_view ts, size_t n, const uint32_t *order, const ctgn_frame_options *fo,
                             const double pose_initial[14], const double tbe[2], ctgn_frame_outputs *out) {
    const ctgn_status st = frame_begin_body(h, raw, ts, n, order, fo, pose_initial, tbe, out);
    if (st != CTGN_OK && h && h->device >= 0) {       // no transfer from the caller's arrays stays in flight behind an error
        (void) hipStreamSynchronize(h->stream);
        (void) hipGetLastError();
    }
    return st;
}

ctgn_status ctgn_frame_try_register(ctgn_handle h, const ctgn_frame_options *fo, double pose_io[14], const double tbe[2], const ctgn_options *opts,
                                    const ctgn_motion_prior *prior, const ctgn_robust_options *robust, const ctgn_robust_prior *robust_prior,
                                    ctgn_frame_outputs *out, ctgn_summary *summary) {
    NEED_DEVICE(h);
    if (summary) std::memset(summary, 0, sizeof(*summary));
    if (out) out->num_keypoints = 0;
    if (!fo || !pose_io || !tbe || (!opts && !robust)) return CTGN_ERR_INVALID_ARGUMENT;
    auto &F = h->fr;
    if (!F.staged) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "no sampled scan is resident (ctgn_frame_begin)");
    if (out) out->num_sampled = F.n1;
    F.valid = false;                                   // d_corr is about to hold nothing the map may take
    if (F.kp_voxel != fo->sample_voxel_size) {         // another keypoint voxel than the one sampled with (a robust retry): sample again
        ctgn_status ss = frame_sample(h, F.frame_voxel, fo->sample_voxel_size);
        if (ss != CTGN_OK) return ss;
    }
    size_t n2 = F.n2;
    if (fo->max_num_keypoints > 0 && n2 > (size_t) fo->max_num_keypoints) n2 = (size_t) fo->max_num_keypoints;
    h->pose_on_device = false;
    {
        ctgn_status rs = reserve_keypoints(h, n2);
        if (rs != CTGN_OK) return rs;
    }
    const size_t kc = (size_t) h->kp_stride;
    if (n2) {
        ctgn_status ps = frame_upload_pose(h, pose_io);
        if (ps != CTGN_OK) return ps;
        hipLaunchKernelGGL(k_frame_keypoints, dim3(grid_for(n2)), dim3(256), 0, h->stream, F.d_scan + 16, F.d_sel2, (int) n2, h->d_kp, kc);
        hipLaunchKernelGGL(k_transform_points, dim3(grid_for(n2)), dim3(256), 0, h->stream, h->d_kp, h->d_kp + 4 * kc, (int) n2, kc,
                           (const double *) F.d_scan, tbe[0], tbe[1]);
        HIPCHK(h, hipGetLastError());
    }
    h->t_min = F.tmin; h->t_max = F.tmax;
    h->kp_coherent = false;                            // a sampled frame in shuffled or scan order: let the cost model order large ones
    {
        ctgn_status ds = ensure_debug(h);
        if (ds != CTGN_OK) return ds;
    }
    const bool want_world = out && out->keypoint_world_base && n2;
    if (want_world && on_device(out->keypoint_world_base)) return fail(h, CTGN_ERR_UNSUPPORTED, "ctgn_frame_try_register writes host memory");
    h->prefetch_world = want_world;                    // the keypoints' final world points ride home with the final state
    ctgn_status st = robust ? ctgn_solve_robust(h, pose_io, tbe, robust, robust_prior, summary) : ctgn_solve(h, pose_io, tbe, opts, prior, summary);
    h->prefetch_world = false;
    if (st != CTGN_OK) return st;
    if (want_world) scatter_world_from_staging(h, out->keypoint_world_base, out->keypoint_world_stride_bytes, out->keypoint_world_dtype, n2);
    if (out && out->keypoint_indices && n2) {
        const uint32_t *d_idx = F.d_sel2;
        if (F.permuted) {
            hipLaunchKernelGGL(k_frame_translate, dim3(grid_for(n2)), dim3(256), 0, h->stream, (const uint32_t *) F.d_sel2, (const uint32_t *) F.d_order, (int) n2, F.d_selx + F.cap);
            HIPCHK(h, hipGetLastError());
            d_idx = F.d_selx + F.cap;
        }
        HIPCHK(h, hipMemcpyAsync(F.h_sel + F.stride, d_idx, n2 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::memcpy(out->keypoint_indices, F.h_sel + F.stride, n2 * sizeof(uint32_t));
    }
    if (out) { out->num_keypoints = n2; out->num_keypoint_candidates = F.n2; }
    return CTGN_OK;
}

ctgn_status ctgn_frame_undistort(ctgn_handle h, const double pose[14], const double tbe[2], ctgn_frame_outputs *out) {
    NEED_DEVICE(h);
    if (!pose || !tbe) return CTGN_ERR_INVALID_ARGUMENT;
    auto &F = h->fr;
    if (!F.staged) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "no sampled scan is resident (ctgn_frame_begin)");
    const size_t n = F.n, n1 = F.n1, c = F.stride;
    if (out) { out->num_sampled = n1; }
    // every point is undistorted: InterpolatePose CHECKs begin <= t <= end for each (types.h:456)
    if (n && !(tbe[0] <= F.tmin && F.tmax <= tbe[1])) return fail(h, CTGN_ERR_TIMESTAMP_RANGE, "point timestamps must lie in [t_begin, t_end]");
    const double *d_recs = F.d_scan + 16;
    F.valid = false;
    {
        ctgn_status ps = frame_upload_pose(h, pose);
        if (ps != CTGN_OK) return ps;
    }
    if (n1) hipLaunchKernelGGL(k_transform_points, dim3(grid_for(n1)), dim3(256), 0, h->stream, d_recs, F.d_corr, (int) n1, (size_t) 1,
                               (const double *) F.d_scan, tbe[0], tbe[1], (const uint32_t *) F.d_sel1, c, (size_t) 4);
    const bool want_all = out && out->all_world_base && n;
    const bool all_rows = want_all && out->all_world_dtype == CTGN_F64 && out->all_world_stride_bytes == 3 * sizeof(double);
    if (want_all) hipLaunchKernelGGL(k_transform_points, dim3(grid_for(n)), dim3(256), 0, h->stream, d_recs, F.d_world, (int) n, (size_t) 1,
                                     (const double *) F.d_scan, tbe[0], tbe[1], (const uint32_t *) nullptr, c, (size_t) 4, 1,
                                     F.permuted ? (const uint32_t *) F.d_order : (const uint32_t *) nullptr);
    HIPCHK(h, hipGetLastError());
    const bool direct_out = all_rows && tuning().frame_no_direct == 0 && host_pinned(out->all_world_base, n * 3 * sizeof(double));
    // the sampled frame first: it is small, and the host scatters it while the scan-sized copy is still travelling
    const bool want_sampled = out && out->sampled_world_base && n1;
    if (want_sampled) {
        HIPCHK(h, hipMemcpyAsync(F.h_out + 3 * c, F.d_corr, (2 * c + n1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (!h->ev_frame) HIPCHK(h, hipEventCreateWithFlags(&h->ev_frame, hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->ev_frame, h->stream));
    }
    if (want_all)
        HIPCHK(h, hipMemcpyAsync(direct_out ? static_cast<double *>(out->all_world_base) : F.h_out, F.d_world, 3 * n * sizeof(double),
                                 hipMemcpyDeviceToHost, h->stream));
    if (want_sampled) {
        HIPCHK(h, hipEventSynchronize(h->ev_frame));
        for (size_t k = 0; k < n1; ++k)
            write_point(out->sampled_world_base, out->sampled_world_stride_bytes, out->sampled_world_dtype, k, F.h_out[3 * c + k], F.h_out[4 * c + k],
                        F.h_out[5 * c + k]);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (want_all && !direct_out) frame_scatter_all(h, n, out, all_rows);
    F.valid = true;                                    // d_corr = the sampled frame under these poses: ctgn_frame_update_map may insert it
    return CTGN_OK;
}

// ---------------------------------------------------------------------------------------- robust-loss route
void ctgn_robust_options_default(ctgn_robust_options *o) {
    if (!o) return;
    o->num_iters_icp = 5;                 // reference include/ct_icp/ct_icp.h:58-132
    o->min_number_neighbors = 20;
    o->max_number_neighbors = 20;
    o->debug_print = 0;
    o->max_num_residuals = -1;
    o->loss_function = CTGN_LOSS_CAUCHY;
    o->ls_max_num_iters = 1;
    o->num_closest_neighbors = 1;
    o->weight_alpha = 0.9;
    o->weight_neighborhood = 0.1;
    o->power_planarity = 2.0;
    o->max_dist_to_plane_ct_icp = 0.3;
    o->ls_sigma = 0.1;
    o->ls_tolerant_min_threshold = 0.05;
    o->threshold_orientation_norm = 0.0001;
    o->threshold_translation_norm = 0.001;
}

static ctgn_status ensure_robust(ctgn_handle h, int k) {
    if (!h->d_rstate) {
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_rstate), sizeof(RobustState)));
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_rstate), sizeof(RobustState), hipHostMallocDefault));
        HIPCHK(h, hipMemsetAsync(h->d_rstate, 0, sizeof(RobustState), h->stream));
    }
    if (h->rb_cap < h->cap_kp || h->rb_k < k) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_rbuf) HIPCHK(h, hipFree(h->d_rbuf));
        if (h->d_rrank) HIPCHK(h, hipFree(h->d_rrank));
        h->d_rbuf = nullptr; h->d_rrank = nullptr; h->rb_cap = 0; h->rb_k = 0;
        const size_t c = (size_t) std::max(h->cap_kp, 1);
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_rbuf), c * (5 + 3 * (size_t) k) * sizeof(double)));
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_rrank), c * sizeof(int)));
        h->rb_cap = (int) c;
        h->rb_k = k;
    }
    return CTGN_OK;
}

static RobustBuf robust_buf(ctgn_handle h) {
    RobustBuf b;
    const size_t c = (size_t) h->rb_cap;
    b.nx = h->d_rbuf; b.ny = h->d_rbuf + c; b.nz = h->d_rbuf + 2 * c; b.w = h->d_rbuf + 3 * c; b.alpha = h->d_rbuf + 4 * c;
    b.ref = h->d_rbuf + 5 * c;
    b.rank = h->d_rrank;
    b.cap = c;
    return b;
}

ctgn_status ctgn_solve_robust(ctgn_handle h, double pose_io[14], const double tbe[2], const ctgn_robust_options *o,
                              const ctgn_robust_prior *prior, ctgn_summary *summary) {
    NEED_DEVICE(h);
    if (summary) std::memset(summary, 0, sizeof(*summary));
    if (!pose_io || !tbe || !o) return CTGN_ERR_INVALID_ARGUMENT;
    if (o->max_number_neighbors < 1 || o->max_number_neighbors > CTGN_MAX_NEIGHBORS)
        return fail(h, CTGN_ERR_UNSUPPORTED, "max_number_neighbors must be in [1, 32]");
    if (o->num_closest_neighbors < 1 || o->num_closest_neighbors > o->min_number_neighbors ||
        o->num_closest_neighbors > o->max_number_neighbors)
        return fail(h, CTGN_ERR_INVALID_ARGUMENT, "num_closest_neighbors must be in [1, min(min_, max_number_neighbors)]");
    if (o->loss_function < CTGN_LOSS_STANDARD || o->loss_function > CTGN_LOSS_TRUNCATED)
        return fail(h, CTGN_ERR_INVALID_ARGUMENT, "unknown loss_function");
    const double wsum = std::fabs(o->weight_alpha) + std::fabs(o->weight_neighborhood);
    if (!(wsum > 0.0))                                                    // CHECK at ct_icp.cpp:529
        return fail(h, CTGN_ERR_INVALID_ARGUMENT, "weight_alpha + weight_neighborhood must be > 0");
    if (h->n_kp > 0 && !(tbe[0] <= h->t_min && h->t_max <= tbe[1]))
        return fail(h, CTGN_ERR_TIMESTAMP_RANGE, "keypoint timestamps must lie in [t_begin, t_end]");
    const auto t0 = std::chrono::steady_clock::now();
    ctgn_status st = ensure_robust(h, o->num_closest_neighbors);
    if (st != CTGN_OK) return st;
    h->r_opts = *o;
    RobustParams &r = h->rprm;
    r.min_nb = o->min_number_neighbors; r.max_nb = o->max_number_neighbors; r.num_closest = o->num_closest_neighbors;
    r.max_res = o->max_num_residuals; r.loss = o->loss_function; r.ls_max_iters = std::max(0, o->ls_max_num_iters);
    r.lambda_w = std::fabs(o->weight_alpha) / wsum; r.lambda_n = std::fabs(o->weight_neighborhood) / wsum;   // :525-532
    r.power = o->power_planarity;
    r.nbr_scale = o->max_dist_to_plane_ct_icp * o->min_number_neighbors;
    r.sigma = o->ls_sigma; r.tol_min = o->ls_tolerant_min_threshold;
    r.thr_rot_deg = o->threshold_orientation_norm; r.thr_trans = o->threshold_translation_norm;
    r.has_prior = prior ? 1 : 0;
    r.beta_loc = prior ? prior->beta_location_consistency : 0.0;
    r.beta_vel = prior ? prior->beta_constant_velocity : 0.0;
    r.beta_small = prior ? prior->beta_small_velocity : 0.0;
    r.beta_orient = prior ? prior->beta_orientation_consistency : 0.0;
    for (int c = 0; c < 3; ++c) { r.prev_b[c] = prior ? prior->previous_begin_tr[c] : 0.0; r.prev_e[c] = prior ? prior->previous_end_tr[c] : 0.0; }
    for (int c = 0; c < 4; ++c) r.prev_q[c] = prior ? prior->previous_end_quat[c] : (c == 3 ? 1.0 : 0.0);
    // the row kernel reads max_nb only
    h->prm = GnParams{};
    h->prm.min_nb = o->min_number_neighbors;
    h->prm.max_nb = o->max_number_neighbors;

    if (h->gn_active) { HIPCHK(h, hipStreamSynchronize(h->stream)); h->gn_active = false; }
    h->init_pending = false;                          // this route initialises the state itself, below
    const double *d_pose = h->d_pose_in;
    if (h->pose_on_device) {                          // already behind the keypoint arrays (ctgn_register_robust)
        d_pose = h->d_kp + 7 * (size_t) h->kp_stride;
        h->pose_on_device = false;
    } else {
        h->pose_in_valid = false;
        for (int i = 0; i < 14; ++i) h->h_pose_in[i] = pose_io[i];
        HIPCHK(h, hipMemcpyAsync(h->d_pose_in, h->h_pose_in, 14 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        h->pose_in_valid = true;
    }
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(64), 0, h->stream, h->d_state, d_pose, tbe[0], tbe[1]);    // :476-477
    hipLaunchKernelGGL(k_robust_init, dim3(1), dim3(64), 0, h->stream, h->d_state, h->d_rstate);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev_loop_start, h->stream));
    MapView mv;
    st = make_map_view(h, -1.0, &mv);
    if (st != CTGN_OK) return st;
    h->planned_iters = 0;                            // the robust route caps its residuals at max_num_residuals: small sets
    if (h->order_stale) {
        st = order_keypoints(h, mv);
        if (st != CTGN_OK) return st;
    }
    const RobustBuf rb = robust_buf(h);
    const KpView kv = kp_view(h);
    const int n = h->n_kp;
    const int grid_lane = std::max(1, std::min((n + 255) / 256, 2048));
    const int grid_eval = std::max(1, std::min((n + EVAL_BLOCK - 1) / EVAL_BLOCK, h->res_grid_cap));
    // evaluation + step (+ the end-of-iteration bookkeeping) of the inner solver in ONE launch while ONE block evaluates the frame as fast as
    // several do: measured, Register on the robust route, 1 024 keypoints 0.703 / 0.717 -> 0.684 / 0.685 ms (31 launches fewer), 1 679 keypoints
    // 0.764 / 0.759 -> 0.827 / 0.825 ms (four tiles of Jacobians on one compute unit cost more than the launches they save): up to 1 024
    // keypoints. tuning robust_fuse: -1 = that rule, 0 / 1 = never / always (profiles/r05_ab_s5_robust_fused_launch.txt)
    constexpr int ROBUST_FUSE_MAX = 1024;
    const int env_fuse_r = (int) tuning().robust_fuse;
    const bool fuse_eval_step = env_fuse_r >= 0 ? env_fuse_r != 0 : n <= ROBUST_FUSE_MAX;
    const bool saved_prof = h->profiling;
    h->profiling = false;
    h->searches_in_solve = 0;
    h->fail_reset_pending = true;
    for (int it = 0; st == CTGN_OK && it < o->num_iters_icp; ++it) {                     // :535
        st = launch_accumulate(h, mv, false, true);                                      // transform_keypoints + neighbourhoods
        if (st != CTGN_OK) break;
        hipLaunchKernelGGL(k_robust_prepare, dim3(grid_lane), dim3(256), 0, h->stream, mv, kv, h->d_state, r, rb);
        hipLaunchKernelGGL(k_robust_cap, dim3(1), dim3(CAP_BLOCK), 0, h->stream, h->d_state, h->d_rstate, r, rb, n);
        for (int j = 0; j <= r.ls_max_iters; ++j) {        // ceres::Solve, :627: evaluation 0 at x, then one per candidate
            if (fuse_eval_step) {                           // small frames: one launch, one block (k_robust_eval_step)
                hipLaunchKernelGGL(k_robust_eval_step, dim3(1), dim3(FUSE_BLOCK), sizeof(FuseScratch), h->stream, kv, h->d_state, h->d_rstate, r, rb,
                                   j == r.ls_max_iters ? 1 : 0);
                continue;
            }
            hipLaunchKernelGGL(k_robust_eval, dim3(grid_eval), dim3(EVAL_BLOCK), 0, h->stream, kv, h->d_state, h->d_rstate, r, rb,
                               h->d_partials);
            hipLaunchKernelGGL(k_robust_step, dim3(1), dim3(STEP_BLOCK), 0, h->stream, h->d_partials, grid_eval, h->d_state,
                               h->d_rstate, r);
        }
        if (!fuse_eval_step) hipLaunchKernelGGL(k_robust_outer, dim3(1), dim3(64), 0, h->stream, h->d_state, h->d_rstate, r);
        if (hipGetLastError() != hipSuccess) st = fail(h, CTGN_ERR_HIP, "[HIP] robust kernel launch failed");
    }
    h->profiling = saved_prof;
    if (st != CTGN_OK) { hipStreamSynchronize(h->stream); return st; }
    if (n > 0) {                                                                          // :685 (not after a failure)
        hipLaunchKernelGGL(k_transform, dim3(grid_lane), dim3(256), 0, h->stream, kv, h->d_state,
                           h->prefetch_world ? h->d_kp + 7 * (size_t) h->kp_stride + 16 : nullptr);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipEventRecord(h->ev_loop_stop, h->stream));
    const bool merged = h->prefetch_world && n > 0;            // world points + GnState in one device-to-host copy
    const size_t cs = (size_t) h->kp_stride;
    if (merged) HIPCHK(h, hipMemcpyAsync(h->h_kp + 4 * cs, h->d_kp + 4 * cs, (3 * cs + KP_TAIL) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    else HIPCHK(h, hipMemcpyAsync(h->h_state, h->d_state, sizeof(GnState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_rstate, h->d_rstate, sizeof(RobustState), hipMemcpyDeviceToHost, h->stream));
    if (h->prefetch_world && !merged) { ctgn_status ws = enqueue_world_readback(h); if (ws != CTGN_OK) return ws; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (merged) std::memcpy(h->h_state, h->h_kp + 7 * cs + 16, sizeof(GnState));
    const GnState &s = *h->h_state;
    const RobustState &rs = *h->h_rstate;
    if (rs.error) return fail(h, CTGN_ERR_SOLVER, "the inner solver reported an unusable solution");
    for (int i = 0; i < 14; ++i) pose_io[i] = s.pose[i];
    if (summary) {
        summary->success = s.failed ? 0 : 1;
        summary->num_residuals_used = rs.n_res;
        summary->num_iters = rs.icp_iter;
        summary->last_step_norm = rs.diff_trans;
        float ms = 0.f;
        hipEventElapsedTime(&ms, h->ev_loop_start, h->ev_loop_stop);
        summary->duration_device_ms = ms;
        // ICPSummary's timing fields (ct_icp.h:164-168), from the device's 100 MHz wall clock stamped by the kernels themselves
        const double per_iter = s.iter > 0 ? 1e-5 / (double) s.iter : 0.0;                 // 10 ns ticks -> ms, averaged
        summary->avg_duration_neighborhood_ms = (double) s.ticks_neighborhood * per_iter;
        summary->avg_duration_solve_ms = (double) s.ticks_solve * per_iter;
        summary->avg_duration_iter_ms = (double) s.ticks_iter * per_iter;
        summary->duration_init_ms = h->init_ms;
        summary->duration_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (s.failed) {
            std::snprintf(summary->error_log, sizeof(summary->error_log),
                          "[CT_ICP] Error : not enough keypoints selected in ct-icp !\n[CT_ICP] number_of_residuals : %d\n",
                          rs.n_res);                                 // same text as ct_icp.cpp:614-615
            if (o->debug_print) std::fputs(summary->error_log, stdout);
        }
    }
    return CTGN_OK;
}

ctgn_status ctgn_register_robust(ctgn_handle h, ctgn_view raw, void *world_base, size_t world_stride, ctgn_dtype world_dtype,
                                 ctgn_view ts, size_t n, double pose_io[14], const double tbe[2], const ctgn_robust_options *opts,
                                 const ctgn_robust_prior *prior, ctgn_summary *summary) {
    ctgn_view world{world_base, world_stride, world_dtype, 0};
    if (h) h->pose_with_kp = pose_io;
    ctgn_status st = ctgn_set_keypoints(h, raw, world, ts, n);
    if (h) h->pose_with_kp = nullptr;
    if (st != CTGN_OK) { if (h) h->pose_on_device = false; return st; }
    const bool dev_world = n > 0 && on_device(world_base);
    h->prefetch_world = !dev_world;
    st = ctgn_solve_robust(h, pose_io, tbe, opts, prior, summary);
    h->prefetch_world = false;
    h->pose_on_device = false;
    if (st != CTGN_OK) return st;
    if (dev_world) return ctgn_get_world_points(h, world_base, world_stride, world_dtype, n);
    if (n) scatter_world_from_staging(h, world_base, world_stride, world_dtype, n);
    return CTGN_OK;
}

ctgn_status ctgn_robust_get_report(ctgn_handle h, ctgn_robust_report *out) {
    NEED_DEVICE(h);
    if (!out) return CTGN_ERR_INVALID_ARGUMENT;
    if (!h->d_rstate) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_solve_robust was not called");
    HIPCHK(h, hipMemcpyAsync(h->h_rstate, h->d_rstate, sizeof(RobustState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const RobustState &rs = *h->h_rstate;
    out->cost = rs.x_cost; out->radius = rs.radius; out->diff_rot_deg = rs.diff_rot; out->diff_trans = rs.diff_trans;
    out->num_residuals = rs.n_res; out->ls_iterations = rs.ls_iters_total; out->ls_accepted = rs.ls_accepted_total;
    out->converged = rs.converged;
    for (int i = 0; i < 144; ++i) out->JtJ[i] = rs.H[i];
    for (int i = 0; i < 12; ++i) out->Jtr[i] = rs.g[i];
    for (int i = 0; i < 8; ++i) out->step_cycles[i] = rs.step_cycles[i];
    return CTGN_OK;
}

ctgn_status ctgn_robust_get_blocks(ctgn_handle h, double *normal, double *weight, double *alpha, double *reference, int32_t *rank,
                                   size_t n) {
    NEED_DEVICE(h);
    if (!h->d_rbuf || n > (size_t) h->n_kp) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "no robust solve on this many keypoints");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const RobustBuf rb = robust_buf(h);
    std::vector<double> tmp(n);
    auto pull = [&](const double *src, double *dst, int comp, int ncomp) -> hipError_t {
        hipError_t e = hipMemcpy(tmp.data(), src, n * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess) for (size_t i = 0; i < n; ++i) dst[ncomp * i + comp] = tmp[i];
        return e;
    };
    if (normal) { HIPCHK(h, pull(rb.nx, normal, 0, 3)); HIPCHK(h, pull(rb.ny, normal, 1, 3)); HIPCHK(h, pull(rb.nz, normal, 2, 3)); }
    if (weight) HIPCHK(h, pull(rb.w, weight, 0, 1));
    if (alpha) HIPCHK(h, pull(rb.alpha, alpha, 0, 1));
    if (reference) {
        const size_t ncap = (size_t) h->rprm.num_closest * rb.cap;
        for (int c = 0; c < 3; ++c) HIPCHK(h, pull(rb.ref + c * ncap, reference, c, 3));
    }
    if (rank) HIPCHK(h, hipMemcpy(rank, rb.rank, n * sizeof(int), hipMemcpyDeviceToHost));
    return CTGN_OK;
}

// ---------------------------------------------------------------------------------------- misc
ctgn_status ctgn_set_stream(ctgn_handle h, void *stream) {
    NEED_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    h->stream = static_cast<hipStream_t>(stream);
    h->own_stream = false;
    return CTGN_OK;
}

ctgn_status ctgn_get_stream(ctgn_handle h, void **stream) {
    NEED_DEVICE(h);
    if (!stream) return CTGN_ERR_INVALID_ARGUMENT;
    *stream = h->stream;
    return CTGN_OK;
}

ctgn_status ctgn_set_debug(ctgn_handle h, int32_t enable) {
    NEED_DEVICE(h);
    h->debug = enable != 0;
    return ensure_debug(h);
}

ctgn_status ctgn_get_debug(ctgn_handle h, int32_t *nnb, double *normal, double *a2d, double *farthest, uint8_t *used, size_t n) {
    NEED_DEVICE(h);
    if (!h->debug || !h->d_nnb) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "debug capture is off (ctgn_set_debug)");
    if (n > (size_t) h->n_kp) return CTGN_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (nnb) HIPCHK(h, hipMemcpy(nnb, h->d_nnb, n * sizeof(int), hipMemcpyDeviceToHost));
    if (normal) HIPCHK(h, hipMemcpy(normal, h->d_normal, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (a2d) HIPCHK(h, hipMemcpy(a2d, h->d_a2d, n * sizeof(double), hipMemcpyDeviceToHost));
    if (farthest) HIPCHK(h, hipMemcpy(farthest, h->d_far, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (used) HIPCHK(h, hipMemcpy(used, h->d_used, n, hipMemcpyDeviceToHost));
    return CTGN_OK;
}

ctgn_status ctgn_get_system(ctgn_handle h, double out[CTGN_SYSTEM_DOUBLES]) {
    NEED_DEVICE(h);
    if (!out) return CTGN_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->d_sys, SYS_N * sizeof(double), hipMemcpyDeviceToHost));
    return CTGN_OK;
}

// What the box's HBM actually delivers to plain streaming kernels ("verify on the box with a device-to-device copy /
// triad ... record both"): a float4 grid-stride copy (bytes read + bytes written) and the triad a = b + s c, `reps` launches each over
// `bytes`-sized arrays, HIP events on the handle's stream. out_gbs[0] = copy, [1] = triad, [2] = hipMemcpyAsync device to device, GB/s of read + written bytes.
namespace {
// shape 0: one float4 per thread (n4 / 256 blocks); shape 1: grid-stride over 8 blocks per CU. (Four loads in flight per thread and
// step were measured too: slower, 4.4 against 4.95 TB/s.)
__global__ __launch_bounds__(256) void k_hbm_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t) gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_hbm_triad(const float4 *__restrict__ b, const float4 *__restrict__ c, float4 *__restrict__ a, float s, size_t n4) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t) gridDim.x * blockDim.x) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}
}  // namespace
ctgn_status ctgn_measure_hbm(ctgn_handle h, uint64_t bytes, int32_t reps, double out_gbs[3]) {
    NEED_DEVICE(h);
    if (!out_gbs || bytes < (1u << 20) || reps < 1) return CTGN_ERR_INVALID_ARGUMENT;
    const size_t n4 = (size_t) bytes / sizeof(float4);
    float4 *buf[3] = {nullptr, nullptr, nullptr};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&] { for (auto *b : buf) if (b) hipFree(b); if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); };
    for (auto &b : buf)
        if (hipMalloc(reinterpret_cast<void **>(&b), n4 * sizeof(float4)) != hipSuccess) { cleanup(); return fail(h, CTGN_ERR_HIP, "[HIP] ctgn_measure_hbm: out of device memory"); }
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { cleanup(); return fail(h, CTGN_ERR_HIP, "[HIP] hipEventCreate"); }
    for (auto *b : buf) (void) hipMemsetAsync(b, 0, n4 * sizeof(float4), h->stream);
    float ms = 0.f;
    out_gbs[0] = out_gbs[1] = 0.0;
    for (int shape = 0; shape < 2; ++shape) {          // the better of the two launch shapes per kernel
        const unsigned grid = shape == 0 ? (unsigned) std::min<size_t>((n4 + 255) / 256, 0x7fffffffu) : (unsigned) (h->num_cus * 8);
        for (int pass = 0; pass < 2; ++pass) {
            for (int r = -2; r < reps; ++r) {          // two untimed launches first
                if (r == 0) (void) hipEventRecord(e0, h->stream);
                if (pass == 0) hipLaunchKernelGGL(k_hbm_copy, dim3(grid), dim3(256), 0, h->stream, buf[0], buf[1], n4);
                else hipLaunchKernelGGL(k_hbm_triad, dim3(grid), dim3(256), 0, h->stream, buf[0], buf[1], buf[2], 0.5f, n4);
            }
            (void) hipEventRecord(e1, h->stream);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) { cleanup(); return fail(h, CTGN_ERR_HIP, "[HIP] ctgn_measure_hbm: timing failed"); }
            out_gbs[pass] = std::max(out_gbs[pass], (double) (pass == 0 ? 2 : 3) * (double) (n4 * sizeof(float4)) * reps / ((double) ms * 1e-3) / 1e9);
        }
    }
    // the runtime's own device-to-device copy of the same arrays, as a third figure
    for (int r = -2; r < reps; ++r) {
        if (r == 0) (void) hipEventRecord(e0, h->stream);
        (void) hipMemcpyAsync(buf[1], buf[0], n4 * sizeof(float4), hipMemcpyDeviceToDevice, h->stream);
    }
    (void) hipEventRecord(e1, h->stream);
    out_gbs[2] = 0.0;
    if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f)
        out_gbs[2] = 2.0 * (double) (n4 * sizeof(float4)) * reps / ((double) ms * 1e-3) / 1e9;
    cleanup();
    return CTGN_OK;
}

ctgn_status ctgn_count_traffic(ctgn_handle h, uint64_t *probed, uint64_t *hit, uint64_t *points) {
    NEED_DEVICE(h);
    MapView mv;
    ctgn_status st = make_map_view(h, -1.0, &mv);
    if (st != CTGN_OK) return st;
    HIPCHK(h, hipMemsetAsync(h->d_counters, 0, sizeof(Counters), h->stream));
    if (h->n_kp > 0) {
        const int grid = std::max(1, std::min((h->n_kp + 255) / 256, 2048));
        hipLaunchKernelGGL(k_count_traffic, dim3(grid), dim3(256), 0, h->stream, mv, kp_view(h), h->d_counters);
        HIPCHK(h, hipGetLastError());
    }
    Counters c;
    HIPCHK(h, hipMemcpyAsync(&c, h->d_counters, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (probed) *probed = c.probed;
    if (hit) *hit = c.hit;
    if (points) *points = c.points;
    return CTGN_OK;
}

ctgn_status ctgn_set_profiling(ctgn_handle h, int32_t enable) {
    NEED_DEVICE(h);
    h->profiling = enable != 0;
    return CTGN_OK;
}

ctgn_status ctgn_kernel_timing(ctgn_handle h, double *avg_ms, int32_t *launches, int32_t reset) {
    NEED_DEVICE(h);
    if (avg_ms) *avg_ms = h->acc_launches ? h->acc_ms / h->acc_launches : 0.0;
    if (launches) *launches = h->acc_launches;
    if (reset) { h->acc_ms = 0.0; h->acc_launches = 0; for (int k = 0; k < 2; ++k) { h->acc_ms_split[k] = 0.0; h->acc_launches_split[k] = 0; } }
    return CTGN_OK;
}

ctgn_status ctgn_kernel_timing_split(ctgn_handle h, double avg_ms[2], int32_t launches[2], int32_t reset) {
    NEED_DEVICE(h);
    for (int k = 0; k < 2; ++k) {
        if (avg_ms) avg_ms[k] = h->acc_launches_split[k] ? h->acc_ms_split[k] / h->acc_launches_split[k] : 0.0;
        if (launches) launches[k] = h->acc_launches_split[k];
        if (reset) { h->acc_ms_split[k] = 0.0; h->acc_launches_split[k] = 0; }
    }
    return CTGN_OK;
}

ctgn_status ctgn_phase_cycles(ctgn_handle h, uint64_t out[12], int32_t reset) {
    NEED_DEVICE(h);
    if (!out) return CTGN_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->d_prof, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(h, hipMemset(h->d_prof, 0, 12 * sizeof(unsigned long long)));
    return CTGN_OK;
}

ctgn_status ctgn_traffic_counters(ctgn_handle h, uint64_t out[2], int32_t reset) {
    NEED_DEVICE(h);
    if (!out) return CTGN_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->d_prof + 12, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(h, hipMemset(h->d_prof + 12, 0, 2 * sizeof(unsigned long long)));
    return CTGN_OK;
}

ctgn_status ctgn_wave_timeline(ctgn_handle h, uint64_t *out, size_t max_waves, size_t *n_waves) {
    NEED_DEVICE(h);
    if (!out || !n_waves) return CTGN_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t n = std::min(max_waves, (size_t) MAX_PARTIAL_BLOCKS * ROW_WAVES);
    HIPCHK(h, hipMemcpy(out, h->d_prof + 16, 4 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *n_waves = n;
    return CTGN_OK;
}

ctgn_status ctgn_test_sort_pairs(ctgn_handle h, const uint64_t *keys, size_t n, int32_t key_bits, int32_t key_bytes, uint32_t *order_out) {
    NEED_DEVICE(h);
    if ((n && (!keys || !order_out)) || key_bits < 1 || key_bits > 64 || (key_bytes != 4 && key_bytes != 8) || n > ((size_t) 1 << 30))
        return CTGN_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    DMCHK(h, devmap_test_sort(keys, n, key_bits, key_bytes, order_out, h->stream));
    return CTGN_OK;
}

ctgn_status ctgn_set_tuning(const char *key, double value) {
    if (!key) return CTGN_ERR_INVALID_ARGUMENT;
    double *slot = tuning_slot(tuning(), key);
    if (!slot) return CTGN_ERR_INVALID_ARGUMENT;
    if (slot == &tuning().host_threads && g_host_threads_latched && *slot != value) return CTGN_ERR_UNSUPPORTED;     // the pool has been sized
    *slot = value;
    return CTGN_OK;
}

// measurement hook (scripts/pool_probe.py): the per-keypoint state the searches carry from one iteration to the next, in working order —
// kth[2 n] (pool completeness radius | k-th neighbour's distance, KpView::kth) and the record's count word (n | pool size << 8 | TIE_FLAG)
ctgn_status ctgn_debug_pool_state(ctgn_handle h, float *kth_out, uint32_t *cnt_out, size_t n) {
    NEED_DEVICE(h);
    if (n > (size_t) h->n_kp || (n && (!kth_out || !cnt_out))) return CTGN_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const KpView kv = kp_view(h, false);
    HIPCHK(h, hipMemcpy(kth_out, kv.kth, 2 * n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(cnt_out, kv.cnt, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CTGN_OK;
}

ctgn_status ctgn_path_counters(ctgn_handle h, uint64_t out[2]) {
    if (!h || !out) return CTGN_ERR_INVALID_ARGUMENT;
    if (h->device >= 0 && h->d_partials) {           // did the last solve launch sum the per-XCD group records? (drains the stream)
        unsigned int w = 0;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpy(&w, reinterpret_cast<unsigned int *>(h->d_partials + (size_t) (MAX_PARTIAL_BLOCKS + XCD_GROUPS) * SYS_N) + 32 * XCD_GROUPS + 1, sizeof(w), hipMemcpyDeviceToHost));
        h->path_counts[1] = w;
    }
    out[0] = h->path_counts[0];
    out[1] = h->path_counts[1];
    return CTGN_OK;
}

ctgn_status ctgn_last_upload_bytes(ctgn_handle h, uint64_t *bytes) {
    if (!h || !bytes) return CTGN_ERR_INVALID_ARGUMENT;
    *bytes = h->last_upload_bytes;
    return CTGN_OK;
}

ctgn_status ctgn_test_compact(ctgn_handle h, const uint8_t *flags, size_t n, uint32_t *out_indices, size_t *out_count) {
    NEED_DEVICE(h);
    if (!out_count || (n && (!flags || !out_indices)) || n > ((size_t) 1 << 30)) return CTGN_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    DMCHK(h, devmap_test_compact(flags, n, out_indices, out_count, h->stream));
    return CTGN_OK;
}

ctgn_status ctgn_set_ordering(ctgn_handle h, int32_t mode) {
    if (!h || mode < -1 || mode > 1) return CTGN_ERR_INVALID_ARGUMENT;
    h->ordering_mode = mode;
    return CTGN_OK;
}

ctgn_status ctgn_set_persistent(ctgn_handle h, int32_t mode) {
    if (!h || (mode != 0 && mode != 1)) return CTGN_ERR_INVALID_ARGUMENT;
    h->persist_mode = mode;
    return CTGN_OK;
}

ctgn_status ctgn_set_pools(ctgn_handle h, int32_t mode) {
    if (!h || mode < -1 || mode > 1) return CTGN_ERR_INVALID_ARGUMENT;
    h->pool_mode = mode;
    return CTGN_OK;
}

ctgn_status ctgn_set_search_guess(ctgn_handle h, double factor) {
    if (!h) return CTGN_ERR_INVALID_ARGUMENT;
    h->guess_factor = factor;
    return CTGN_OK;
}

ctgn_status ctgn_set_normals(ctgn_handle h, int32_t mode) {
    if (!h) return CTGN_ERR_INVALID_ARGUMENT;
    if (mode < 0 || mode > 3) return fail(h, CTGN_ERR_INVALID_ARGUMENT, "ctgn_set_normals: mode 0 (default), 1 exact, 2 hybrid, 3 fast");
    h->normals_mode = mode;
    h->prm.normals = mode;
    return CTGN_OK;
}

ctgn_status ctgn_set_ablation(ctgn_handle h, int32_t mask) {
    if (!h) return CTGN_ERR_INVALID_ARGUMENT;
    h->ablate = mask;
    return CTGN_OK;
}

ctgn_status ctgn_set_variant(ctgn_handle h, int32_t variant) {
    if (!h || variant < 0 || variant > 5) return CTGN_ERR_INVALID_ARGUMENT;
    h->variant = variant;
    return CTGN_OK;
}

}  // extern "C"
