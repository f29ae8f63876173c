// multidev.h — one process driving several GPUs of a node (anh_set_devices): the device selection of a host thread, the
// exchange step between the replicas of a handle (RCCL over xGMI), and the host logic that decides who owns what.
//   training  (annonet_train_main.cpp:583-614): the mini-batch is split along N, every replica runs forward + backward on its
//             share with the loss scale of the WHOLE batch, ONE all-reduce (sum) of the flat gradient bucket, identical update;
//   inference (annonet_infer.cpp:46-165): the tile list is split into contiguous row-major chunks, every replica blends its
//             tiles into planes of its own, the plane sums are exchanged ONLY where tiles of different replicas overlap.
#pragma once
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace anh {

// HIP's current device is per host thread; every call into a replica runs under its device
struct DeviceScope {
    int prev = -1, cur = -1;
    explicit DeviceScope(int device) : cur(device) {
        if (device < 0) return;
        HIP_CHECK(hipGetDevice(&prev));
        if (prev != device) HIP_CHECK(hipSetDevice(device));
    }
    ~DeviceScope() { if (cur >= 0 && prev != cur) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

// anh_set_devices: the devices that handles created afterwards ON THIS THREAD drive (empty = the thread's current device)
const std::vector<int>& selected_devices();
void select_devices(const int* devices, int n);

// replica r of `world` takes the units [lo, hi) of n (mini-batch samples, tiles): contiguous chunks, sizes differ by at most 1
inline void shard_range(int64_t n, int world, int rank, int64_t& lo, int64_t& hi) { lo = n * rank / world; hi = n * (rank + 1) / world; }

// The rectangles (inclusive, clipped to the image) in which tiles owned by DIFFERENT replicas overlap: the only pixels where a
// replica's blended planes hold partial sums.  Pairwise intersections; they may overlap each other (the exchange is idempotent).
std::vector<anh_rect> cross_replica_overlaps(const std::vector<anh_tile>& tiles, int world, int width, int height);

// SUM over the replicas of one fp32 buffer each, the result left in every buffer; enqueued on the replicas' own streams.
//   distinct devices  -> RCCL: one communicator per device (ncclCommInitAll), one grouped ncclAllReduce — xGMI, no host copy;
//   a repeated device -> rehearsal backend for boxes with fewer GPUs than replicas (tests): fixed-order sum on replica 0's
//                        stream, then copies back.  Same result up to fp32 summation order.
class Collective {
  public:
    explicit Collective(const std::vector<int>& devices);
    ~Collective();
    void all_reduce_sum(const std::vector<float*>& bufs, size_t count, const std::vector<hipStream_t>& streams);
    bool uses_rccl() const { return !comms_.empty(); }
    int transport() const { return uses_rccl() ? 1 : peer_copies_ ? 2 : 0; }   // 0 repeated-device rehearsal, 1 RCCL, 2 peer copies between distinct devices
    static int rccl_version();   // ncclGetVersion (e.g. 22203), 0 if the call fails
    int world() const { return (int)devices_.size(); }

  private:
    std::vector<int> devices_;
    std::vector<void*> comms_;   // ncclComm_t
    std::vector<hipEvent_t> ready_;
    hipEvent_t done_ = nullptr;
    DevBuf scratch_;
    bool peer_copies_ = false;
};

// gather / scatter of plane pixels inside a list of rectangles: packed[k][offset_j + (y - top_j) * w_j + (x - left_j)] <-> planes[k][y][x]
struct RectTable {
    std::vector<anh_rect> rects;
    std::vector<int64_t> offset;   // first packed index of each rectangle; back() = total
    int64_t total() const { return offset.empty() ? 0 : offset.back(); }
};
RectTable make_rect_table(const std::vector<anh_rect>& rects);
void launch_pack_rects(const float* planes, int k, int height, int width, const anh_rect* d_rects, const int64_t* d_offsets, int n_rects, int64_t total,
                       float* packed, hipStream_t s);
void launch_unpack_rects(float* planes, int k, int height, int width, const anh_rect* d_rects, const int64_t* d_offsets, int n_rects, int64_t total,
                         const float* packed, hipStream_t s);

// fn(r) for every replica r of a handle: replica 0 on the calling thread, the others on worker threads.  One host thread enqueues a
// replica's ~50 launches per training step (or its share of an image's tiles) in 0.3-0.4 ms; eight replicas driven from one thread
// would be launch-bound at twice the GPU time of a step.
//
// Round 4: the workers are PERSISTENT.  ReplicaWorkers lives with the handle (anh_trainer / anh_runtime): replicas - 1 threads, each
// bound to one replica index, with its device selected once at thread start; a call is ONE wake-up (a generation counter under a
// mutex + condition variable) and one wait for the stragglers.  Round 3's for_each_replica created and joined replicas - 1
// std::threads on every call — at eight replicas and a 1.6 ms step, seven thread creations per StartTraining, each new thread paying
// its first hipSetDevice.  Every fn(r) touches only replica r's state; the first exception is rethrown on the caller.
class ReplicaWorkers {
  public:
    explicit ReplicaWorkers(const std::vector<int>& devices) : devices_(devices), failed_(devices.size()) {
        for (size_t r = 1; r < devices_.size(); ++r) threads_.emplace_back([this, r] { loop(r); });
    }
    ~ReplicaWorkers() {
        { std::lock_guard<std::mutex> lock(mu_); stop_ = true; ++generation_; }
        go_.notify_all();
        for (std::thread& t : threads_) t.join();
    }
    ReplicaWorkers(const ReplicaWorkers&) = delete;
    ReplicaWorkers& operator=(const ReplicaWorkers&) = delete;
    size_t replicas() const { return devices_.size(); }
    uint64_t calls() const { return calls_.load(std::memory_order_relaxed); }   // (tests: the pool was used, and how often)

    // One call at a time per pool: `call_mu_` is held for the whole call, so a second host thread on the same handle waits
    // instead of overwriting job_ / pending_ (a NESTED call is refused: it could never finish) — the contract stays "one thread per handle" (annonet_hip.h), this only
    // makes a violation safe.  The wait for the stragglers has a DEADLINE (ANH_REPLICA_TIMEOUT_S, default 180 s — the collective
    // timeout of the one-process-per-GPU host, bench.py --collective-timeout): a worker that never comes back is a replica stuck in
    // a HIP / RCCL call; its job references this frame, so the call cannot unwind — the process ends with a message and exit code 3
    // (positive, as find_max_mini-batch_size.cmd:49-53 expects of a failure that is not a crash).
    template <class F>
    void run(F&& fn) {
        const size_t R = devices_.size();
        if (R <= 1) { if (R == 1) fn((size_t)0); return; }
        // a job that itself calls run() — from the calling thread or from a worker — can never finish (the pool is busy with it): refuse
        ANH_REQUIRE(!inside_job(), "ReplicaWorkers::run called from inside a replica job (each_replica does not nest)");
        std::lock_guard<std::mutex> one_call(call_mu_);
        struct Inside { Inside() { inside_job() = true; } ~Inside() { inside_job() = false; } } inside;
        std::function<void(size_t)> job = [&fn](size_t r) { fn(r); };
        {
            std::lock_guard<std::mutex> lock(mu_);
            job_ = &job;
            pending_ = R - 1;
            for (auto& e : failed_) e = nullptr;
            ++generation_;
        }
        calls_.fetch_add(1, std::memory_order_relaxed);
        go_.notify_all();
        try { fn((size_t)0); } catch (...) { failed_[0] = std::current_exception(); }
        {
            std::unique_lock<std::mutex> lock(mu_);
            if (!done_.wait_for(lock, std::chrono::seconds(replica_timeout_s()), [this] { return pending_ == 0; })) {
                fprintf(stderr, "annonet_hip: %zu of %zu replica worker(s) did not finish within %d s (ANH_REPLICA_TIMEOUT_S): a replica is stuck "
                                "in a device or collective call; ending the process\n", pending_, R - 1, replica_timeout_s());
                fflush(stderr);
                _exit(3);
            }
            job_ = nullptr;
        }
        for (const std::exception_ptr& e : failed_) if (e) std::rethrow_exception(e);
    }
    static int replica_timeout_s() { return replica_timeout_seconds() + 30; }   // (the workers' own device waits are bounded: let them report first)

  private:
    static bool& inside_job() { static thread_local bool v = false; return v; }
    void loop(size_t r) {
        (void)hipSetDevice(devices_[r]);   // once: every DeviceScope of this replica's calls is then a no-op
        inside_job() = true;               // a worker thread only ever runs jobs
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)>* job = nullptr;
            {
                std::unique_lock<std::mutex> lock(mu_);
                go_.wait(lock, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                job = job_;
            }
            try { (*job)(r); } catch (...) { failed_[r] = std::current_exception(); }
            bool last;
            { std::lock_guard<std::mutex> lock(mu_); last = --pending_ == 0; }
            if (last) done_.notify_one();
        }
    }
    std::vector<int> devices_;
    std::vector<std::exception_ptr> failed_;
    std::vector<std::thread> threads_;
    std::mutex mu_, call_mu_;
    std::condition_variable go_, done_;
    const std::function<void(size_t)>* job_ = nullptr;
    uint64_t generation_ = 0;
    std::atomic<uint64_t> calls_{0};
    size_t pending_ = 0;
    bool stop_ = false;
};

// (the creating / joining form, kept for callers without a handle)
template <class F>
void for_each_replica(size_t replicas, F&& fn) {
    if (replicas <= 1) { if (replicas == 1) fn((size_t)0); return; }
    std::vector<std::exception_ptr> failed(replicas);
    std::vector<std::thread> workers;
    workers.reserve(replicas - 1);
    for (size_t r = 1; r < replicas; ++r)
        workers.emplace_back([&fn, &failed, r] { try { fn(r); } catch (...) { failed[r] = std::current_exception(); } });
    try { fn((size_t)0); } catch (...) { failed[0] = std::current_exception(); }
    for (std::thread& w : workers) w.join();
    for (const std::exception_ptr& e : failed) if (e) std::rethrow_exception(e);
}

}  // namespace anh
