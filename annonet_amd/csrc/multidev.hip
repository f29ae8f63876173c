// multidev.hip — see multidev.h.  RCCL is the collective library of the exchange steps; nothing here touches a tensor's values
// except the three tiny kernels at the bottom (fixed-order add of the rehearsal backend, rectangle gather / scatter).
#include <string>

#include "multidev.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <set>

namespace anh {

namespace {
thread_local std::vector<int> g_devices;

void nccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) fail(ANH_ERR_DEVICE, std::string(what) + ": " + ncclGetErrorString(r));
}
#define NCCL_CHECK(expr) nccl_check((expr), #expr)

__global__ __launch_bounds__(256) void add_inplace_kernel(float* a, const float* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] += b[i];
}

// one thread per packed element: binary search of the rectangle that holds it
template <bool PACK>
__global__ __launch_bounds__(256) void rect_move_kernel(float* planes, int k, int height, int width, const anh_rect* rects, const int64_t* offsets,
                                                        int n_rects, int64_t total, float* packed) {
    const int64_t plane = (int64_t)height * width;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int lo = 0, hi = n_rects - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (offsets[mid] <= i) lo = mid; else hi = mid - 1; }
        const anh_rect r = rects[lo];
        const int64_t local = i - offsets[lo];
        const int64_t w = r.right - r.left + 1;
        const int64_t y = r.top + local / w, x = r.left + local % w;
        for (int c = 0; c < k; ++c) {
            float* p = planes + c * plane + y * width + x;
            if (PACK) packed[c * total + i] = *p;
            else *p = packed[c * total + i];
        }
    }
}
}  // namespace

const std::vector<int>& selected_devices() { return g_devices; }

void select_devices(const int* devices, int n) {
    ANH_REQUIRE(n >= 0 && (n == 0 || devices), "device list");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
    for (int i = 0; i < n; ++i) ANH_REQUIRE(devices[i] >= 0 && devices[i] < count, "device index out of range");
    g_devices.assign(devices, devices + n);
}

std::vector<anh_rect> cross_replica_overlaps(const std::vector<anh_tile>& tiles, int world, int width, int height) {
    std::vector<int> owner(tiles.size(), 0);
    for (int r = 0; r < world; ++r) {
        int64_t lo, hi;
        shard_range((int64_t)tiles.size(), world, r, lo, hi);
        for (int64_t i = lo; i < hi; ++i) owner[(size_t)i] = r;
    }
    std::set<std::tuple<long, long, long, long>> seen;
    std::vector<anh_rect> out;
    for (size_t i = 0; i < tiles.size(); ++i)
        for (size_t j = i + 1; j < tiles.size(); ++j) {
            if (owner[i] == owner[j]) continue;
            const anh_rect& a = tiles[i].full_rect;
            const anh_rect& b = tiles[j].full_rect;
            const long l = std::max({a.left, b.left, 0L}), t = std::max({a.top, b.top, 0L});
            const long r = std::min({a.right, b.right, (long)width - 1}), bt = std::min({a.bottom, b.bottom, (long)height - 1});
            if (l > r || t > bt) continue;
            if (seen.insert({l, t, r, bt}).second) out.push_back(anh_rect{l, t, r, bt});
        }
    std::sort(out.begin(), out.end(), [](const anh_rect& x, const anh_rect& y) {
        return std::tie(x.left, x.top, x.right, x.bottom) < std::tie(y.left, y.top, y.right, y.bottom);
    });
    return out;
}

RectTable make_rect_table(const std::vector<anh_rect>& rects) {
    RectTable t;
    t.rects = rects;
    t.offset.assign(rects.size() + 1, 0);
    for (size_t i = 0; i < rects.size(); ++i)
        t.offset[i + 1] = t.offset[i] + (int64_t)(rects[i].right - rects[i].left + 1) * (int64_t)(rects[i].bottom - rects[i].top + 1);
    return t;
}

void launch_pack_rects(const float* planes, int k, int height, int width, const anh_rect* d_rects, const int64_t* d_offsets, int n_rects, int64_t total,
                       float* packed, hipStream_t s) {
    if (total <= 0) return;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(rect_move_kernel<true>, dim3(blocks), dim3(256), 0, s, const_cast<float*>(planes), k, height, width, d_rects, d_offsets, n_rects, total, packed);
    HIP_CHECK(hipGetLastError());
}
void launch_unpack_rects(float* planes, int k, int height, int width, const anh_rect* d_rects, const int64_t* d_offsets, int n_rects, int64_t total,
                         const float* packed, hipStream_t s) {
    if (total <= 0) return;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(rect_move_kernel<false>, dim3(blocks), dim3(256), 0, s, planes, k, height, width, d_rects, d_offsets, n_rects, total, const_cast<float*>(packed));
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
Collective::Collective(const std::vector<int>& devices) : devices_(devices) {
    ANH_REQUIRE(!devices.empty(), "collective over no device");
    std::vector<int> sorted = devices;
    std::sort(sorted.begin(), sorted.end());
    const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
    // Transport.  Distinct devices take RCCL; if its communicators cannot be created (ncclCommInitAll fails: a driver / IPC problem, a
    // device another process holds exclusively), or ANH_COLLECTIVE_TRANSPORT=peer asks for it, the SAME process goes on with the
    // peer-copy transport below (events + hipMemcpyPeerAsync + a fixed-order sum on replica 0) — slower (everything funnels through
    // one device's links) but a run that comes back with numbers and says which transport made them (anh_exchange_stats::uses_rccl).
    const char* want = getenv("ANH_COLLECTIVE_TRANSPORT");
    const bool force_peer = want && std::string(want) == "peer";
    if (distinct && devices.size() > 1 && !force_peer) {
        std::vector<ncclComm_t> comms(devices.size());
        const ncclResult_t r = ncclCommInitAll(comms.data(), (int)devices.size(), devices.data());
        if (r == ncclSuccess) for (ncclComm_t c : comms) comms_.push_back((void*)c);
        else {
            fprintf(stderr, "annonet_hip: ncclCommInitAll over %zu devices failed (%s); the exchange steps of this handle use peer copies instead\n",
                    devices.size(), ncclGetErrorString(r));
            (void)hipGetLastError();
        }
    }
    peer_copies_ = distinct && devices.size() > 1 && comms_.empty();
    if (comms_.empty()) {
        for (size_t i = 0; i < devices.size(); ++i) {
            DeviceScope scope(devices[i]);
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ready_.push_back(e);
        }
        DeviceScope scope(devices[0]);
        HIP_CHECK(hipEventCreateWithFlags(&done_, hipEventDisableTiming));
    }
}

int Collective::rccl_version() {
    int v = 0;
    return ncclGetVersion(&v) == ncclSuccess ? v : 0;
}

Collective::~Collective() {
    for (void* c : comms_) (void)ncclCommDestroy((ncclComm_t)c);
    for (hipEvent_t e : ready_) (void)hipEventDestroy(e);
    if (done_) (void)hipEventDestroy(done_);
}

void Collective::all_reduce_sum(const std::vector<float*>& bufs, size_t count, const std::vector<hipStream_t>& streams) {
    const size_t R = devices_.size();
    ANH_REQUIRE(bufs.size() == R && streams.size() == R, "collective: one buffer and one stream per replica");
    if (R == 1 || count == 0) return;
    if (uses_rccl()) {
        NCCL_CHECK(ncclGroupStart());
        for (size_t i = 0; i < R; ++i) {
            DeviceScope scope(devices_[i]);
            NCCL_CHECK(ncclAllReduce(bufs[i], bufs[i], count, ncclFloat, ncclSum, (ncclComm_t)comms_[i], streams[i]));
        }
        NCCL_CHECK(ncclGroupEnd());
        return;
    }
    // rehearsal backend: replica 0's stream waits for every producer, sums in replica order, hands the result back
    for (size_t i = 1; i < R; ++i) {
        DeviceScope scope(devices_[i]);
        HIP_CHECK(hipEventRecord(ready_[i], streams[i]));
    }
    DeviceScope scope(devices_[0]);
    const int blocks = (int)std::min<size_t>((count + 255) / 256, 256 * 8);
    for (size_t i = 1; i < R; ++i) {
        HIP_CHECK(hipStreamWaitEvent(streams[0], ready_[i], 0));
        const float* src = bufs[i];
        if (devices_[i] != devices_[0]) {
            scratch_.reserve(count * 4);
            HIP_CHECK(hipMemcpyPeerAsync(scratch_.p, devices_[0], bufs[i], devices_[i], count * 4, streams[0]));
            src = scratch_.as<float>();
        }
        hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, streams[0], bufs[0], src, count);
        HIP_CHECK(hipGetLastError());
    }
    for (size_t i = 1; i < R; ++i) {
        if (devices_[i] == devices_[0]) HIP_CHECK(hipMemcpyAsync(bufs[i], bufs[0], count * 4, hipMemcpyDeviceToDevice, streams[0]));
        else HIP_CHECK(hipMemcpyPeerAsync(bufs[i], devices_[i], bufs[0], devices_[0], count * 4, streams[0]));
    }
    HIP_CHECK(hipEventRecord(done_, streams[0]));
    for (size_t i = 1; i < R; ++i) {
        DeviceScope s2(devices_[i]);
        HIP_CHECK(hipStreamWaitEvent(streams[i], done_, 0));
    }
}

}  // namespace anh
