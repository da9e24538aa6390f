// group.hip -- one process, several GPUs: the cell-sharded generate behind the C ABI (mm_group_*).
//
// The reference drives ONE device (CUDAPhotomosaicGenerator(const int device),
// src/Photomosaic/CUDA/CUDAPhotomosaicGenerator.h:31); the split over the GPUs of a node is new and follows
// the data dependencies of the path: base scores of different cells are independent, so the valid cells (raster
// order) are cut into contiguous shards of ceil(n / devices), library / masks / target are replicated, every
// device scores its shard into its own packed list buffer, ONE RCCL all-gather (ncclGroupStart / ncclAllGather per
// device / ncclGroupEnd, one communicator per device from ncclCommInitAll) hands every device every list, and
// every device runs the same verifying resolve -- identical grids, compared here as a guard against silent
// divergence.  The payload is K x 12 bytes per cell (a few KB to a few MB): latency-bound, far below the size
// where the 7 x ~153 GB/s xGMI links matter, hence a single collective, no bucketing, nothing to overlap.
// Cells whose short speculative list could not vouch for the pick are deepened by their owner and gathered again
// (the loop of mm_generate, engine.hip, with the devices in place of the batches).
//
// Built on the public entry points of include/mosaic_hip.h only (mm_create, mm_score_cells_partial with device
// outputs, mm_resolve_verify with device inputs): the group owns nothing of an engine's inside.
// RCCL refuses one device twice in a communicator; a group over such a list (what a one-GPU box can rehearse) --
// or any group whose communicator cannot be built -- gathers through host memory instead: same buffers, same
// layout, same resolve, same result (mm_group_uses_rccl tells which).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mosaic_hip.h"

namespace {

thread_local std::string g_group_create_error;

struct Member {
    int device = 0;
    mm_engine* engine = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    void* send = nullptr;        // [per][K] i32 | pad to 8 | [per][K] f64: what this device contributes
    void* recv = nullptr;        // [world] x the above
    void* idx_all = nullptr;     // [world * per][K] i32, unpacked for the resolve
    void* score_all = nullptr;   // [world * per][K] f64
    size_t send_bytes = 0, recv_bytes = 0, idx_bytes = 0, score_bytes = 0;
    std::vector<int64_t> grid;   // this device's resolve (device 0's goes to the caller)
    std::vector<double> score;
    std::vector<uint8_t> insufficient;
    mm_status status = MM_OK;
    std::string own_error;       // a worker's text for a failure that is not the engine's (raw HIP call)
};

}  // namespace

struct mm_group {
    std::vector<Member> members;
    bool rccl = false;
    int n_lib = 0;
    int spec_k = 4;
    std::string err;
    std::vector<char> host_all;  // host-merge fallback: every device's packed lists
};

namespace {

void set_err(mm_group* g, const char* fmt, ...)
{
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf; else g_group_create_error = buf;
}

mm_status hip_fail(mm_group* g, hipError_t code, const char* what, int line)
{
    set_err(g, "%s: %s (%d) at %s:%d", what, hipGetErrorName(code), (int)code, __FILE__, line);
    (void)hipGetLastError();
    return code == hipErrorOutOfMemory ? MM_OUT_OF_MEMORY : MM_DEVICE_ERROR;
}

#define GROUP_HIP_TRY(g, call, what)                                               \
    do {                                                                            \
        hipError_t code_ = (call);                                                  \
        if (code_ != hipSuccess) return hip_fail((g), code_, (what), __LINE__);     \
    } while (0)

mm_status grow(mm_group* g, void*& p, size_t& have, size_t want, const char* what)
{
    if (p && have >= want) return MM_OK;
    if (p) { (void)hipFree(p); p = nullptr; have = 0; }
    GROUP_HIP_TRY(g, hipMalloc(&p, std::max<size_t>(want, 16)), what);
    have = want;
    return MM_OK;
}

// Byte layout of one device's lists inside the all-gather buffer (the layout of mosaicmagnifique_amd/dist.py
// pack_layout: indices, padded to 8 bytes, then scores).
inline size_t score_offset(int per, int k) { return ((size_t)per * k * 4 + 7) / 8 * 8; }
inline size_t packed_bytes(int per, int k) { return score_offset(per, k) + (size_t)per * k * 8; }

// On every member in parallel (one host thread each: the engine calls block); false if any failed.
template <typename F>
bool on_all(mm_group* g, F&& fn)
{
    std::vector<std::thread> threads;
    for (size_t i = 0; i < g->members.size(); ++i)
        threads.emplace_back([&, i] {
            Member& m = g->members[i];
            m.own_error.clear();
            try { m.status = fn((int)i, m); }
            catch (...) { m.status = MM_OUT_OF_MEMORY; }      // (std::bad_alloc of a host vector: no exception leaves a worker)
        });
    for (std::thread& t : threads) t.join();
    for (size_t i = 0; i < g->members.size(); ++i) {
        const Member& m = g->members[i];
        if (m.status != MM_OK) {
            if (m.status != MM_CANCELLED && g->err.empty())
                set_err(g, "device %d (member %zu): %s", m.device, i, m.own_error.empty() ? mm_last_error(m.engine) : m.own_error.c_str());
            return false;
        }
    }
    return true;
}

mm_status first_failure(const mm_group* g)
{
    for (const Member& m : g->members) if (m.status != MM_OK) return m.status;
    return MM_DEVICE_ERROR;
}

}  // namespace

extern "C" {

mm_status mm_group_create(const int* devices, int n_devices, mm_group** out)
{
    if (!out) return MM_INVALID_ARGUMENT;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) {
        set_err(nullptr, "mm_group_create: need 1 <= n_devices <= 64 (got %d)", n_devices);
        return MM_INVALID_ARGUMENT;
    }
    mm_group* g = new (std::nothrow) mm_group();
    if (!g) { set_err(nullptr, "out of host memory"); return MM_OUT_OF_MEMORY; }
    g->members.resize((size_t)n_devices);
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) {
        Member& m = g->members[(size_t)i];
        m.device = devices[i];
        for (int j = 0; j < i; ++j) distinct = distinct && devices[j] != devices[i];
        const mm_status st = mm_create(devices[i], &m.engine);
        if (st != MM_OK) {
            set_err(nullptr, "mm_group_create: device %d: %s", devices[i], mm_last_error(nullptr));
            mm_group_destroy(g);
            return st;
        }
        hipError_t code = hipSetDevice(devices[i]);
        if (code == hipSuccess) code = hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking);
        if (code != hipSuccess) {
            const mm_status fs = hip_fail(nullptr, code, "Failed to create the group's stream", __LINE__);
            mm_group_destroy(g);
            return fs;
        }
    }
    // One communicator per device, all built by one call (single process).  RCCL refuses a device listed twice, and a
    // communicator that cannot be built (a driver without peer access ...) is no reason to fail the generate:
    // both gather through the host instead.
    if (distinct) {
        std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
        const ncclResult_t res = ncclCommInitAll(comms.data(), n_devices, devices);
        (void)hipGetLastError();         // (RCCL's probing of peer access may leave a sticky HIP error on this thread)
        if (res == ncclSuccess) {
            for (int i = 0; i < n_devices; ++i) g->members[(size_t)i].comm = comms[(size_t)i];
            g->rccl = true;
        } else {
            set_err(g, "ncclCommInitAll: %s -- gathering through host memory", ncclGetErrorString(res));
        }
    }
    *out = g;
    return MM_OK;
}

void mm_group_destroy(mm_group* g)
{
    if (!g) return;
    for (Member& m : g->members) {
        (void)hipSetDevice(m.device);
        if (m.stream) (void)hipStreamSynchronize(m.stream);
        if (m.comm) { (void)ncclCommDestroy(m.comm); (void)hipGetLastError(); }
        for (void* p : { m.send, m.recv, m.idx_all, m.score_all }) if (p) (void)hipFree(p);
        if (m.stream) (void)hipStreamDestroy(m.stream);
        if (m.engine) mm_destroy(m.engine);
    }
    delete g;
}

const char* mm_group_last_error(const mm_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }
int mm_group_size(const mm_group* g) { return g ? (int)g->members.size() : 0; }
int mm_group_uses_rccl(const mm_group* g) { return g && g->rccl ? 1 : 0; }
mm_engine* mm_group_engine(mm_group* g, int index)
{
    return (g && index >= 0 && index < (int)g->members.size()) ? g->members[(size_t)index].engine : nullptr;
}

// ---- replicated staging: the same call on every engine, one host thread each ----------------------------------

mm_status mm_group_set_library(mm_group* g, const float* images, int n_images, int size)
{
    if (!g) return MM_INVALID_ARGUMENT;
    g->err.clear();
    if (!on_all(g, [&](int, Member& m) { return mm_set_library(m.engine, images, n_images, size); })) return first_failure(g);
    g->n_lib = n_images;
    return MM_OK;
}

mm_status mm_group_set_library_u8(mm_group* g, const uint8_t* bgr, int n_images, int size, int detail_size, int diff_type)
{
    if (!g) return MM_INVALID_ARGUMENT;
    g->err.clear();
    if (!on_all(g, [&](int, Member& m) { return mm_set_library_u8_sized(m.engine, bgr, n_images, size, detail_size, diff_type); }))
        return first_failure(g);
    g->n_lib = n_images;
    return MM_OK;
}

mm_status mm_group_set_masks(mm_group* g, const uint8_t* masks, int size)
{
    if (!g) return MM_INVALID_ARGUMENT;
    g->err.clear();
    return on_all(g, [&](int, Member& m) { return mm_set_masks(m.engine, masks, size); }) ? MM_OK : first_failure(g);
}

mm_status mm_group_set_target(mm_group* g, const float* image, int rows, int cols)
{
    if (!g) return MM_INVALID_ARGUMENT;
    g->err.clear();
    return on_all(g, [&](int, Member& m) { return mm_set_target(m.engine, image, rows, cols); }) ? MM_OK : first_failure(g);
}

mm_status mm_group_set_target_u8(mm_group* g, const uint8_t* bgr, int rows, int cols, int diff_type)
{
    if (!g) return MM_INVALID_ARGUMENT;
    g->err.clear();
    return on_all(g, [&](int, Member& m) { return mm_set_target_u8(m.engine, bgr, rows, cols, diff_type); }) ? MM_OK : first_failure(g);
}

mm_status mm_group_set_option(mm_group* g, const char* name, int value)
{
    if (!g || !name) return MM_INVALID_ARGUMENT;
    g->err.clear();
    for (Member& m : g->members) {
        const mm_status st = mm_set_option(m.engine, name, value);
        if (st != MM_OK) { set_err(g, "%s", mm_last_error(m.engine)); return st; }
    }
    if (strcmp(name, "spec_k") == 0) g->spec_k = std::max(0, value);
    return MM_OK;
}

// ---- the sharded generate ------------------------------------------------------------------------------------------

mm_status mm_group_generate(mm_group* g, const mm_cell* cells, int n_cells, int grid_w, int grid_h, int diff_type,
                            int repeat_range, int repeat_addition, int64_t* best_fit_out, double* best_score_out,
                            mm_progress_cb progress, void* progress_user, volatile int* cancel)
{
    if (!g) return MM_INVALID_ARGUMENT;
    g->err.clear();
    if (n_cells < 0 || (n_cells > 0 && !cells) || grid_w <= 0 || grid_h <= 0 || !best_fit_out || repeat_range < 0) {
        set_err(g, "mm_group_generate: bad arguments");
        return MM_INVALID_ARGUMENT;
    }
    const int world = (int)g->members.size();
    int n_lib = 0;
    (void)mm_library_shape(g->members[0].engine, &n_lib, nullptr);
    if (n_lib <= 0) { set_err(g, "library, masks and target must be staged on every device before generating"); return MM_NOT_READY; }
    const int K = std::max(1, mm_candidates_per_cell(n_lib, repeat_range, repeat_addition));
    const int Ks = std::max(1, mm_speculative_candidates(n_lib, repeat_range, repeat_addition, g->spec_k));
    const bool speculate = Ks < K && n_cells > 0;
    const int per = n_cells > 0 ? (n_cells + world - 1) / world : 0;
    const int rows = std::max(per, 1);
    const size_t total = packed_bytes(rows, K), s_off = score_offset(rows, K);
    const size_t cells_total = (size_t)grid_w * grid_h;

    // buffers (kept between calls)
    for (Member& m : g->members) {
        GROUP_HIP_TRY(g, hipSetDevice(m.device), "Failed to select device");
        mm_status st;
        if ((st = grow(g, m.send, m.send_bytes, total, "Failed to allocate the device's candidate lists")) ||
            (st = grow(g, m.recv, m.recv_bytes, total * world, "Failed to allocate the gathered candidate lists")) ||
            (st = grow(g, m.idx_all, m.idx_bytes, (size_t)rows * world * K * 4, "Failed to allocate the gathered indices")) ||
            (st = grow(g, m.score_all, m.score_bytes, (size_t)rows * world * K * 8, "Failed to allocate the gathered scores")))
            return st;
        m.grid.resize(cells_total);
        m.score.resize(cells_total);
        m.insufficient.assign((size_t)std::max(n_cells, 1), 0);
    }
    auto shard = [&](int i, int& lo, int& hi) { lo = std::min(i * per, n_cells); hi = std::min(lo + per, n_cells); };

    // 1. every device scores its own shard, short lists, straight into its send buffer; padding rows stay inert
    if (!on_all(g, [&](int i, Member& m) -> mm_status {
            int lo, hi;
            shard(i, lo, hi);
            if (hipSetDevice(m.device) != hipSuccess) return MM_DEVICE_ERROR;
            int32_t* idx = (int32_t*)m.send;
            double* sc = (double*)((char*)m.send + s_off);
            if (hi - lo < rows) {            // rows past the shard's own cells: index -1, score +inf
                std::vector<int32_t> pi((size_t)(rows - (hi - lo)) * K, -1);
                std::vector<double> ps(pi.size(), __builtin_huge_val());
                if (hipMemcpy(idx + (size_t)(hi - lo) * K, pi.data(), pi.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(sc + (size_t)(hi - lo) * K, ps.data(), ps.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
                    return MM_DEVICE_ERROR;
            }
            if (hi <= lo) return MM_OK;
            return mm_score_cells_partial(m.engine, cells + lo, hi - lo, diff_type, K, Ks, idx, sc, 1, nullptr, nullptr, cancel);
        }))
        return first_failure(g);
    if (progress) progress(n_cells, progress_user);

    std::vector<uint8_t> deep((size_t)std::max(n_cells, 1), 0);
    for (;;) {
        if (cancel && *cancel) return MM_CANCELLED;
        // 2. ONE all-gather of the packed lists
        if (g->rccl) {
            ncclResult_t res = ncclGroupStart();
            for (int i = 0; i < world && res == ncclSuccess; ++i) {
                Member& m = g->members[(size_t)i];
                res = ncclAllGather(m.send, m.recv, total, ncclUint8, m.comm, m.stream);
            }
            const ncclResult_t end = ncclGroupEnd();
            (void)hipGetLastError();
            if (res == ncclSuccess) res = end;
            if (res != ncclSuccess) { set_err(g, "ncclAllGather: %s", ncclGetErrorString(res)); return MM_DEVICE_ERROR; }
        } else {
            g->host_all.resize(total * world);
            for (int i = 0; i < world; ++i) {
                Member& m = g->members[(size_t)i];
                GROUP_HIP_TRY(g, hipSetDevice(m.device), "Failed to select device");
                GROUP_HIP_TRY(g, hipMemcpy(g->host_all.data() + total * i, m.send, total, hipMemcpyDeviceToHost), "Error copying candidate lists to host");
            }
            for (Member& m : g->members) {
                GROUP_HIP_TRY(g, hipSetDevice(m.device), "Failed to select device");
                GROUP_HIP_TRY(g, hipMemcpyAsync(m.recv, g->host_all.data(), total * world, hipMemcpyHostToDevice, m.stream),
                              "Error copying candidate lists to device");
            }
        }
        // ... unpacked into [cells][K] arrays on every device (shards are contiguous blocks of `rows`)
        for (Member& m : g->members) {
            GROUP_HIP_TRY(g, hipSetDevice(m.device), "Failed to select device");
            GROUP_HIP_TRY(g, hipMemcpy2DAsync(m.idx_all, (size_t)rows * K * 4, m.recv, total, (size_t)rows * K * 4, (size_t)world,
                                              hipMemcpyDeviceToDevice, m.stream), "Error unpacking the gathered indices");
            GROUP_HIP_TRY(g, hipMemcpy2DAsync(m.score_all, (size_t)rows * K * 8, (char*)m.recv + s_off, total, (size_t)rows * K * 8,
                                              (size_t)world, hipMemcpyDeviceToDevice, m.stream), "Error unpacking the gathered scores");
        }
        for (Member& m : g->members) {
            GROUP_HIP_TRY(g, hipSetDevice(m.device), "Failed to select device");
            GROUP_HIP_TRY(g, hipStreamSynchronize(m.stream), "Failed to synchronise the group's stream");
        }

        // 3. the same verifying resolve on every device
        if (!on_all(g, [&](int, Member& m) -> mm_status {
                return mm_resolve_verify(m.engine, cells, n_cells, grid_w, grid_h, diff_type, repeat_range, repeat_addition, K, Ks,
                                         (const int32_t*)m.idx_all, (const double*)m.score_all, 1,
                                         speculate ? deep.data() : nullptr, speculate ? m.insufficient.data() : nullptr,
                                         m.grid.data(), m.score.data());
            }))
            return first_failure(g);
        const Member& first = g->members[0];
        for (int i = 1; i < world; ++i) {
            const Member& m = g->members[(size_t)i];
            // (the score grids bit for bit too: identical lists through identical kernels, NaN patterns of the empty cells included)
            if (memcmp(m.grid.data(), first.grid.data(), cells_total * sizeof(int64_t)) != 0 ||
                memcmp(m.score.data(), first.score.data(), cells_total * sizeof(double)) != 0 ||
                (speculate && memcmp(m.insufficient.data(), first.insufficient.data(), (size_t)n_cells) != 0)) {
                set_err(g, "devices %d and %d resolved the same lists differently: inputs or engines differ", first.device, m.device);
                return MM_DEVICE_ERROR;
            }
        }
        if (!speculate) break;

        // 4. the owners deepen their unverified cells to the full length, in place in their send buffers
        std::vector<std::vector<int>> todo((size_t)world);
        bool any = false;
        for (int c = 0; c < n_cells; ++c)
            if (first.insufficient[(size_t)c] && !deep[(size_t)c]) {
                todo[(size_t)std::min(c / per, world - 1)].push_back(c);
                deep[(size_t)c] = 1;
                any = true;
            }
        if (!any) break;
        if (!on_all(g, [&](int i, Member& m) -> mm_status {
                const std::vector<int>& mine = todo[(size_t)i];
                if (mine.empty()) return MM_OK;
                int lo, hi;
                shard(i, lo, hi);
                std::vector<mm_cell> sub;
                for (int c : mine) sub.push_back(cells[c]);
                std::vector<int32_t> si(mine.size() * (size_t)K);
                std::vector<double> ss(mine.size() * (size_t)K);
                const mm_status st = mm_score_cells_partial(m.engine, sub.data(), (int)sub.size(), diff_type, K, K, si.data(), ss.data(), 0,
                                                            nullptr, nullptr, cancel);
                if (st != MM_OK) return st;
                // (a raw HIP failure here is not the engine's: the worker leaves its own text, on_all reports it)
                hipError_t code = hipSetDevice(m.device);
                for (size_t r = 0; code == hipSuccess && r < mine.size(); ++r) {
                    const size_t row = (size_t)(mine[r] - lo);
                    code = hipMemcpyAsync((int32_t*)m.send + row * K, si.data() + r * K, (size_t)K * 4, hipMemcpyHostToDevice, m.stream);
                    if (code == hipSuccess)
                        code = hipMemcpyAsync((double*)((char*)m.send + s_off) + row * K, ss.data() + r * K, (size_t)K * 8, hipMemcpyHostToDevice, m.stream);
                }
                if (code == hipSuccess) code = hipStreamSynchronize(m.stream);      // si / ss go out of scope
                if (code != hipSuccess) {
                    m.own_error = std::string("Error copying deepened lists to device: ") + hipGetErrorName(code);
                    (void)hipGetLastError();
                    return MM_DEVICE_ERROR;
                }
                return MM_OK;
            }))
            return first_failure(g);
    }
    memcpy(best_fit_out, g->members[0].grid.data(), cells_total * sizeof(int64_t));
    if (best_score_out) memcpy(best_score_out, g->members[0].score.data(), cells_total * sizeof(double));
    return MM_OK;
}

}  // extern "C"
