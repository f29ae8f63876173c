// api.cpp — the C ABI (include/annonet_hip.h) over Engine.  Exceptions stop here: every entry point returns a
// status code and leaves the message in a thread-local string (anh_last_error).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <unordered_set>

#include "common.h"
#include "engine.h"
#include "hostlogic.h"
#include "multidev.h"

using namespace anh;

namespace {
thread_local std::string g_error;

template <typename F>
int guarded(F&& f) {
    try { f(); return ANH_OK; }
    catch (const Error& e) { g_error = e.what(); return e.code; }
    catch (const std::bad_alloc&) { g_error = "host allocation failed"; return ANH_ERR_OOM; }
    catch (const std::exception& e) { g_error = e.what(); return ANH_ERR_INTERNAL; }
}

const char kRuntimeMagic[8] = {'A', 'N', 'H', 'R', 'T', '0', '0', '1'};
const char kStateMagic[8] = {'A', 'N', 'H', 'T', 'S', '0', '0', '2'};   // 002: + bn update counters, schedule counter, unrecorded losses

struct BlobHeader {
    char magic[8];
    int32_t levels, in_channels, classes, min_filters;
    double width_scaler;
    int64_t n_params, n_running;
};
}  // namespace

namespace anh {
void set_last_error(const std::string& message) { g_error = message; }

int replica_timeout_seconds() {
    static const int s = getenv("ANH_REPLICA_TIMEOUT_S") && atoi(getenv("ANH_REPLICA_TIMEOUT_S")) > 0 ? atoi(getenv("ANH_REPLICA_TIMEOUT_S")) : 180;
    return s;
}
namespace {
template <class Query>
void poll_until_done(Query&& query, const char* what) {
    hipError_t e = query();
    if (e == hipSuccess) return;   // the common case (the staging set of step k - 2): no clock read
    const auto t0 = std::chrono::steady_clock::now();
    const auto deadline = t0 + std::chrono::seconds(replica_timeout_seconds());
    for (int spins = 0; e == hipErrorNotReady; ++spins) {
        if (spins < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 63) == 63 && std::chrono::steady_clock::now() > deadline)
            fail(ANH_ERR_DEVICE, std::string(what) + ": the device did not finish within " + std::to_string(replica_timeout_seconds()) +
                                     " s (ANH_REPLICA_TIMEOUT_S) — a collective between the replicas of this handle did not complete");
        e = query();
    }
    HIP_CHECK(e);
}
}  // namespace
void wait_event(hipEvent_t ev, bool bounded) {
    if (!bounded) { HIP_CHECK(hipEventSynchronize(ev)); return; }
    poll_until_done([&] { return hipEventQuery(ev); }, "wait for an event");
}
void wait_stream(hipStream_t st, bool bounded) {
    if (!bounded) { HIP_CHECK(hipStreamSynchronize(st)); return; }
    poll_until_done([&] { return hipStreamQuery(st); }, "wait for a stream");
}
}  // namespace anh

// ANH_REPLICA_WORKERS=0: round 3's form (threads created and joined on every call) for the before / after figure of DESIGN.md §6
static bool persistent_workers() { static const bool on = !(getenv("ANH_REPLICA_WORKERS") && atoi(getenv("ANH_REPLICA_WORKERS")) == 0); return on; }

struct anh_runtime {
    std::unique_ptr<Engine> eng;          // replica 0 (the only one unless anh_set_devices named several devices)
    // anh_set_devices: one process drives several GPUs.  Replica r lives on devices[r]; every replica holds the same weights;
    // annonet_infer() shards its tile list over them (infer_multi below).  Empty list = the creating thread's current device.
    std::vector<int> devices;
    struct Replica { std::unique_ptr<Engine> eng; };
    std::vector<Replica> extra;           // replicas 1 .. R-1
    std::unique_ptr<Collective> coll;
    std::unique_ptr<ReplicaWorkers> workers;   // persistent host threads, one per replica beyond the first (multidev.h)
    template <class F> void each_replica(F&& fn) { if (workers && persistent_workers()) workers->run(fn); else for_each_replica(replicas(), fn); }
    struct Exchange { DevBuf packed, rects, offsets; };
    std::vector<Exchange> exchange;       // per replica: scratch of the overlap exchange
    size_t replicas() const { return 1 + extra.size(); }
    int device_of(size_t r) const { return devices.empty() ? -1 : devices[r]; }
    Engine& replica(size_t r) { return r == 0 ? *eng : *extra[r - 1].eng; }
    // `only_device` >= 0: ONE replica on that device whatever anh_set_devices named (a trainer's snapshot, below)
    void build(const anh_net_config& cfg, int only_device = -1) {
        devices = selected_devices();
        if (only_device >= 0) devices.assign(1, only_device);
        { DeviceScope scope(device_of(0)); eng = std::make_unique<Engine>(cfg, false); }
        for (size_t r = 1; r < devices.size(); ++r) { DeviceScope scope(devices[r]); extra.push_back(Replica{std::make_unique<Engine>(cfg, false)}); }
        if (devices.size() > 1) { coll = std::make_unique<Collective>(devices); workers = std::make_unique<ReplicaWorkers>(devices); }
        for (size_t r = 0; r < replicas(); ++r) replica(r).bounded_waits = replicas() > 1;   // host waits with a deadline (common.h)
        exchange.resize(replicas());
    }
    void set_params_all(const float* params, const float* running) {
        for (size_t r = 0; r < replicas(); ++r) { DeviceScope scope(device_of(r)); replica(r).set_params(params, running); }
    }
    // host-buffer annonet_infer(): image strips go up and label strips come down through small pinned rings while the tiles
    // compute, so that neither transfer is exposed (anh_infer below)
    static constexpr int kRing = 3;
    struct Pinned { void* p = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; bool busy = false; };
    Pinned up[kRing], down[kRing];
    hipStream_t copy_up = nullptr, copy_down = nullptr;
    std::vector<hipEvent_t> strip_events;
    ~anh_runtime() {
        for (auto* ring : {up, down})
            for (int i = 0; i < kRing; ++i) {
                if (ring[i].p) (void)hipHostFree(ring[i].p);
                if (ring[i].done) (void)hipEventDestroy(ring[i].done);
            }
        for (auto ev : strip_events) (void)hipEventDestroy(ev);
        if (copy_up) (void)hipStreamDestroy(copy_up);
        if (copy_down) (void)hipStreamDestroy(copy_down);
        for (size_t r = extra.size(); r-- > 0;) { DeviceScope scope(device_of(r + 1)); extra[r].eng.reset(); exchange[r + 1] = Exchange{}; }
    }
};

struct anh_trainer {
    anh_net_config cfg{2, 3, 3, 1.0, 1, ANH_BF16};
    uint64_t seed = 0;
    std::unique_ptr<Engine> eng;
    LrSchedule sched;
    double weight_decay = 0.0005, momentum = 0.9;
    unsigned long bn_window = 100;
    std::string sync_path;
    double sync_seconds = 0;
    std::chrono::steady_clock::time_point last_sync = std::chrono::steady_clock::now();
    bool verbose = false;
    unsigned long steps = 0;
    double last_loss = 0;
    // Losses travel back asynchronously (pinned slots + events) and enter the learning-rate schedule at a FIXED lag: the
    // update of step k first records the loss of step k - kLossLag (waiting for it if it has not arrived).  Which loss drives
    // which lr decision therefore does not depend on GPU / host timing: runs are reproducible and the ranks of a data-parallel
    // job (who all see the same all-reduced loss) shrink their rate at the same step.
    static constexpr unsigned long kLossLag = 4;
    // pinned ring of posted losses: the update kernel of step k stores (tag(k) << 32 | float bits) into slot k mod 256 (SgdArgs::loss_post)
    unsigned long long* loss_ring = nullptr;      // host address
    unsigned long long* loss_ring_dev = nullptr;  // the same words as the device sees them
    static unsigned int loss_tag(unsigned long step) { return (unsigned int)(step & 0x7fffffffu) | 0x80000000u; }   // never the 0 of a fresh slot
    struct PendingLoss { int slot; unsigned long step; bool arrived; double value; };
    std::deque<PendingLoss> pending;   // step order; not yet recorded into the schedule
    int next_slot = 0;
    bool resume_pending = false;       // SetSynchronizationFile named a file: resume from it once the net structure is final
    // Host-buffer steps (StartTraining): two staging sets, each a pinned host block + a device block holding
    // [images | labels u16 | weights f32].  Step k packs into set k&1 while the GPU still runs step k-1, uploads it on a copy
    // stream and chains the compute behind the upload with events; the call returns once the inputs are packed
    // (annonet_train_main.cpp:585-586 refills samples/labels right after StartTraining returns).
    struct StageSet {
        void* pinned = nullptr; size_t pinned_bytes = 0;
        DevBuf dev;
        hipEvent_t uploaded = nullptr, consumed = nullptr;   // H2D of this set done / the step that read it has finished
        bool in_flight = false;
        double wait_us = 0;   // how long the last call that packed into this set waited for the GPU to release it
    };
    StageSet stage[2];
    hipStream_t copy_stream = nullptr;
    unsigned long host_steps = 0;
    // anh_set_devices: data-parallel training from ONE process (annonet_train_main.cpp:583-614 stays as it is).  Replica r lives on
    // devices[r] with its own staging sets; StartTraining splits the mini-batch along N, one all-reduce (RCCL) sums the flat
    // gradient buckets, every replica applies the identical update.  Batch-norm statistics are per replica.
    std::vector<int> devices;
    struct Replica { std::unique_ptr<Engine> eng; StageSet stage[2]; hipStream_t copy_stream = nullptr; };
    std::vector<std::unique_ptr<Replica>> extra;   // replicas 1 .. R-1
    std::unique_ptr<Collective> coll;
    std::unique_ptr<ReplicaWorkers> workers;       // persistent host threads, one per replica beyond the first (multidev.h)
    template <class F> void each_replica(F&& fn) { if (workers && persistent_workers()) workers->run(fn); else for_each_replica(replicas(), fn); }
    // what one StartTraining costs the host, and the exchange step on the device (anh_trainer_exchange_stats): host time of every call;
    // on every `xs_every`-th step an event pair around each part of the all-reduce on replica 0's streams (ANH_EXCHANGE_SAMPLE, default 8, 0 = never)
    struct ExchangeStats {
        int64_t calls = 0; double host_us_sum = 0, host_us_last = 0, wait_us_sum = 0;
        uint64_t worker_calls_base = 0;
        hipEvent_t tail0 = nullptr, tail1 = nullptr, head0 = nullptr, head1 = nullptr;
        bool pending = false, split = false;
        int64_t samples = 0; double tail_us_sum = 0, head_us_sum = 0, tail_us_last = 0, head_us_last = 0;
    } xs;
    void collect_exchange_sample() {   // the pair of a sampled step, once that step has finished (blocks until then)
        if (!xs.pending) return;
        DeviceScope scope(device_of(0));
        float ms = 0;
        HIP_CHECK(hipEventSynchronize(xs.head1));
        HIP_CHECK(hipEventElapsedTime(&ms, xs.head0, xs.head1));
        xs.head_us_last = 1e3 * ms; xs.head_us_sum += xs.head_us_last;
        xs.tail_us_last = 0;
        if (xs.split) { HIP_CHECK(hipEventSynchronize(xs.tail1)); HIP_CHECK(hipEventElapsedTime(&ms, xs.tail0, xs.tail1)); xs.tail_us_last = 1e3 * ms; xs.tail_us_sum += xs.tail_us_last; }
        ++xs.samples;
        xs.pending = false;
    }
    size_t replicas() const { return devices.size() > 1 ? devices.size() : 1; }
    int device_of(size_t r) const { return devices.empty() ? -1 : devices[r]; }
    Engine& replica(size_t r) { return r == 0 ? *eng : *extra[r - 1]->eng; }
    StageSet* stage_of(size_t r) { return r == 0 ? stage : extra[r - 1]->stage; }
    hipStream_t& copy_stream_of(size_t r) { return r == 0 ? copy_stream : extra[r - 1]->copy_stream; }

    bool initialized = false, dirty = false;
    // The reference configures AFTER Initialize() (annonet_train_main.cpp:400-410: Initialize, SetNetWidth, ..., SetClassCount);
    // dlib allocates lazily, so the net is (re)built here on first use after a structural setting changed.
    Engine& engine() {
        if (!initialized) fail(ANH_ERR_INVALID, "TrainingNet::Initialize has not been called");
        if (!eng || dirty) {
            if (steps > 0) fail(ANH_ERR_INVALID, "the net structure cannot change once training has started");
            { DeviceScope scope(device_of(0)); eng = std::make_unique<Engine>(cfg, true); eng->random_init(seed); }
            extra.clear();
            for (size_t r = 1; r < replicas(); ++r) {   // same seed: every replica starts from the same weights
                DeviceScope scope(device_of(r));
                auto rep = std::make_unique<Replica>();
                rep->eng = std::make_unique<Engine>(cfg, true);
                rep->eng->random_init(seed);
                extra.push_back(std::move(rep));
            }
            if (replicas() > 1 && !coll) coll = std::make_unique<Collective>(devices);
            if (replicas() > 1 && !workers) workers = std::make_unique<ReplicaWorkers>(devices);
            for (size_t r = 0; r < replicas(); ++r) replica(r).bounded_waits = replicas() > 1;   // host waits with a deadline (common.h)
            dirty = false;
        }
        if (resume_pending) {   // the reference names the file BEFORE SetClassCount (annonet_train_main.cpp:400-405): resume on first use
            std::ifstream probe(sync_path, std::ios::binary);
            // A file that exists but cannot be loaded (older format, another net, truncated) is an ERROR on every use of the net until the
            // caller removes it or names another: the flag is cleared only once the resume has succeeded, so no step can run — and the
            // periodic save can never overwrite a state the user still has — after a failed resume.
            if (probe.good()) { probe.close(); resume(); }
            resume_pending = false;
        }
        return *eng;
    }
    void resume();
    void structural_change() { if (steps > 0) fail(ANH_ERR_INVALID, "the net structure cannot change once training has started"); dirty = true; }
    void arrive(PendingLoss& p) {   // blocks until the loss of that step is on the host
        if (p.arrived) return;
        const volatile unsigned long long* word = loss_ring + p.slot;
        const unsigned int want = loss_tag(p.step);
        unsigned long long w = *word;
        if ((unsigned int)(w >> 32) != want) {
            // not there yet: the stream is at most kLossLag steps behind.  Drain it (this also surfaces a device fault) and look again.
            { DeviceScope scope(device_of(0)); eng->synchronize(); }
            w = *word;
            if ((unsigned int)(w >> 32) != want) fail(ANH_ERR_INTERNAL, "the loss of a finished step was not posted");
        }
        const unsigned int bits = (unsigned int)(w & 0xffffffffu);
        float v;
        std::memcpy(&v, &bits, sizeof v);
        p.value = (double)v;
        p.arrived = true;
    }
    // records the losses of steps <= upto into the schedule, oldest first
    void record_until(unsigned long upto) {
        while (!pending.empty() && pending.front().step <= upto) {
            arrive(pending.front());
            sched.record(pending.front().value);
            pending.pop_front();
        }
    }
    // waits for every loss in flight (host queries: GetLastLoss, state file) WITHOUT recording any of them early
    void fetch_all() {
        for (auto& p : pending) arrive(p);
        if (!pending.empty()) last_loss = pending.back().value;
    }
    ~anh_trainer() {
        workers.reset();   // the worker threads first: nothing of theirs outlives the replicas
        for (hipEvent_t ev : {xs.tail0, xs.tail1, xs.head0, xs.head1}) if (ev) (void)hipEventDestroy(ev);
        if (loss_ring) (void)hipHostFree(loss_ring);
        for (auto& st : stage) {
            if (st.pinned) (void)hipHostFree(st.pinned);
            if (st.uploaded) (void)hipEventDestroy(st.uploaded);
            if (st.consumed) (void)hipEventDestroy(st.consumed);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (size_t r = extra.size(); r-- > 0;) {
            DeviceScope scope(device_of(r + 1));
            Replica& rep = *extra[r];
            rep.eng.reset();
            for (auto& st : rep.stage) {
                st.dev.release();
                if (st.pinned) (void)hipHostFree(st.pinned);
                if (st.uploaded) (void)hipEventDestroy(st.uploaded);
                if (st.consumed) (void)hipEventDestroy(st.consumed);
            }
            if (rep.copy_stream) (void)hipStreamDestroy(rep.copy_stream);
        }
    }
};

// Full images of a dataset, resident in HBM, plus the scratch of the device crop path.
struct anh_dataset {
    int channels = 3;
    struct Item { DevBuf image, labels; int height = 0, width = 0; };
    std::vector<Item> items;          // a removed image leaves an empty slot (height 0) that a later add reuses
    std::vector<int> free_slots;
    uint64_t resident_bytes = 0;
    hipStream_t stream = nullptr;
    DevBuf d_specs, d_hist, d_first, d_table, d_bad;
    void* pinned = nullptr; size_t pinned_bytes = 0;
    DevBuf out;   // outputs of the host-array form
    ~anh_dataset() {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (pinned) (void)hipHostFree(pinned);
    }
    // Cuts n crops into device arrays on `stream`.  One host round trip inside: the per-crop label histograms come back,
    // set_weights' table is computed with the host's own arithmetic (bit-identical to set_weights on the crop) and goes up.
    void crop_batch(const anh_crop_spec* specs, int n, int dim, int classes, double cw, double iw, uint8_t* d_images, uint16_t* d_labels, float* d_weights) {
        ANH_REQUIRE(specs && n >= 1 && dim >= 1 && dim <= 32768, "crop batch: bad argument");
        ANH_REQUIRE(classes >= 1 && classes <= kCropMaxClasses, "crop batch: class count must be 1..64");
        const size_t spec_bytes = (size_t)n * sizeof(CropSource), tab = (size_t)n * kCropMaxClasses * 4;
        const size_t need = spec_bytes + 3 * tab + 64;
        if (need > pinned_bytes) {
            HIP_CHECK(hipStreamSynchronize(stream));
            if (pinned) HIP_CHECK(hipHostFree(pinned));
            pinned = nullptr; pinned_bytes = 0;
            HIP_CHECK(hipHostMalloc(&pinned, need, hipHostMallocDefault));
            pinned_bytes = need;
        }
        d_specs.reserve(spec_bytes); d_hist.reserve(tab); d_first.reserve(tab); d_table.reserve(tab); d_bad.reserve(4);
        char* pin = static_cast<char*>(pinned);
        CropSource* hs = reinterpret_cast<CropSource*>(pin);
        unsigned* h_hist = reinterpret_cast<unsigned*>(pin + spec_bytes);
        unsigned* h_first = reinterpret_cast<unsigned*>(pin + spec_bytes + tab);
        float* h_table = reinterpret_cast<float*>(pin + spec_bytes + 2 * tab);
        int* h_bad = reinterpret_cast<int*>(pin + spec_bytes + 3 * tab);
        for (int i = 0; i < n; ++i) {
            const anh_crop_spec& c = specs[i];
            ANH_REQUIRE(c.image >= 0 && (size_t)c.image < items.size() && items[(size_t)c.image].height > 0, "crop batch: image index out of range (or removed)");
            ANH_REQUIRE(c.left > -(1L << 30) && c.left < (1L << 30) && c.top > -(1L << 30) && c.top < (1L << 30), "crop batch: rectangle out of range");
            ANH_REQUIRE(c.brightness_change >= 0, "crop batch: negative brightness change");
            const Item& it = items[(size_t)c.image];
            ANH_REQUIRE(c.further_downscaling_factor == 0.0 || (c.further_downscaling_factor >= 1.0 && c.further_downscaling_factor <= 64.0), "crop batch: further downscaling factor must be >= 1");
            ANH_REQUIRE(c.noise_level >= 0 && c.noise_level <= 255, "crop batch: noise level must be 0..255");
            const double f = c.further_downscaling_factor == 0.0 ? 1.0 : c.further_downscaling_factor;
            const int src_dim = (int)std::round(dim * f);   // dim_before_downscaling (annonet_train_main.cpp:127)
            hs[i] = CropSource{it.image.as<uint8_t>(), it.labels.as<uint16_t>(), it.height, it.width, (int)c.left, (int)c.top,
                               c.flip_left_right ? 1 : 0, c.flip_upside_down ? 1 : 0, c.brightness_change, src_dim, c.noise_level,
                               (unsigned long long)c.noise_seed, {c.color_offset[0], c.color_offset[1], c.color_offset[2]}};
        }
        HIP_CHECK(hipMemcpyAsync(d_specs.p, hs, spec_bytes, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemsetAsync(d_bad.p, 0, 4, stream));
        launch_crop_pixels(d_specs.as<CropSource>(), n, dim, channels, d_images, d_labels, d_hist.as<unsigned>(), d_first.as<unsigned>(), d_bad.as<int>(), classes, stream);
        HIP_CHECK(hipMemcpyAsync(h_hist, d_hist.p, tab, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(h_first, d_first.p, tab, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(h_bad, d_bad.p, 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        ANH_REQUIRE(*h_bad == 0, "label value exceeds the class count");
        for (int i = 0; i < n; ++i)
            set_weights_table(h_hist + (size_t)i * kCropMaxClasses, h_first + (size_t)i * kCropMaxClasses, kCropMaxClasses, (long long)dim * dim, cw, iw,
                              h_table + (size_t)i * kCropMaxClasses);
        HIP_CHECK(hipMemcpyAsync(d_table.p, h_table, tab, hipMemcpyHostToDevice, stream));
        launch_crop_weights(d_labels, d_table.as<float>(), n, dim, d_weights, stream);
    }
};

extern "C" {

const char* anh_last_error(void) { return g_error.c_str(); }
void anh_free(void* p) { std::free(p); }

int anh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int anh_set_device(int device) {
    return guarded([&] { HIP_CHECK(hipSetDevice(device)); });
}
int anh_set_devices(const int* devices, int n) {
    return guarded([&] { select_devices(devices, n); if (n >= 1) HIP_CHECK(hipSetDevice(devices[0])); });
}
int anh_handle_replicas(void* handle, int is_trainer) {
    if (!handle) return 0;
    return is_trainer ? (int)((anh_trainer*)handle)->replicas() : (int)((anh_runtime*)handle)->replicas();
}
int anh_host_register(void* p, size_t bytes) {
    return guarded([&] { ANH_REQUIRE(p && bytes, "host register: null block"); HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault)); });
}
int anh_host_unregister(void* p) {
    return guarded([&] { ANH_REQUIRE(p, "host unregister: null block"); HIP_CHECK(hipHostUnregister(p)); });
}
int anh_labels_rect_to_host(const uint16_t* d_labels, uint16_t* h_labels, int width, int height, int left, int top, int right, int bottom, void* stream) {
    return guarded([&] {
        ANH_REQUIRE(d_labels && h_labels && width >= 1 && height >= 1, "labels to host: null map");
        ANH_REQUIRE(left >= 0 && top >= 0 && right < width && bottom < height && left <= right && top <= bottom, "labels to host: rectangle outside the map");
        const size_t off = (size_t)top * width + left;
        if (left == 0 && right == width - 1)   // whole rows: one contiguous block
            HIP_CHECK(hipMemcpyAsync(h_labels + off, d_labels + off, (size_t)(bottom - top + 1) * width * 2, hipMemcpyDeviceToHost, (hipStream_t)stream));
        else
            HIP_CHECK(hipMemcpy2DAsync(h_labels + off, (size_t)width * 2, d_labels + off, (size_t)width * 2, (size_t)(right - left + 1) * 2, (size_t)(bottom - top + 1),
                                       hipMemcpyDeviceToHost, (hipStream_t)stream));
    });
}
int anh_shard_range(int64_t n, int world, int rank, int64_t* lo, int64_t* hi) {
    return guarded([&] { ANH_REQUIRE(lo && hi && world >= 1 && rank >= 0 && rank < world && n >= 0, "shard range: bad argument"); shard_range(n, world, rank, *lo, *hi); });
}
int anh_cross_replica_overlaps(const anh_tile* tiles, size_t n_tiles, int world, int width, int height, anh_rect** rects, size_t* count) {
    return guarded([&] {
        ANH_REQUIRE((tiles || n_tiles == 0) && rects && count && world >= 1, "null argument");
        const std::vector<anh_rect> r = cross_replica_overlaps(std::vector<anh_tile>(tiles, tiles + n_tiles), world, width, height);
        anh_rect* out = (anh_rect*)std::malloc(std::max<size_t>(1, r.size()) * sizeof(anh_rect));
        if (!out) fail(ANH_ERR_OOM, "host allocation failed");
        std::copy(r.begin(), r.end(), out);
        *rects = out; *count = r.size();
    });
}

// ---- dimension maths / spec ----
int anh_required_input_dim(const anh_net_config* cfg) {
    int r = -1;
    guarded([&] { ANH_REQUIRE(cfg, "null config"); r = Spec::build(*cfg).required_input_dim(); });
    return r;
}
int anh_recommended_input_dim(int levels, int n) {
    if (levels < 0 || levels > 3) { g_error = "level count must be 0..3"; return -1; }
    return Spec::recommended_input_dim(levels, n);
}
int anh_net_layer_count(const anh_net_config* cfg) {
    int r = -1;
    guarded([&] { ANH_REQUIRE(cfg, "null config"); r = (int)Spec::build(*cfg).layers.size(); });
    return r;
}
int anh_net_layer(const anh_net_config* cfg, int index, anh_layer_desc* out) {
    return guarded([&] {
        ANH_REQUIRE(cfg && out, "null argument");
        Spec s = Spec::build(*cfg);
        ANH_REQUIRE(index >= 0 && index < (int)s.layers.size(), "layer index out of range");
        *out = s.layers[index];
    });
}
int64_t anh_net_param_count(const anh_net_config* cfg) {
    int64_t r = -1;
    guarded([&] { ANH_REQUIRE(cfg, "null config"); r = Spec::build(*cfg).n_params; });
    return r;
}
int64_t anh_net_running_count(const anh_net_config* cfg) {
    int64_t r = -1;
    guarded([&] { ANH_REQUIRE(cfg, "null config"); r = Spec::build(*cfg).n_running; });
    return r;
}

// ---- RuntimeNet ----
int anh_runtime_create(const anh_net_config* cfg, anh_runtime** out) {
    return guarded([&] {
        ANH_REQUIRE(cfg && out, "null argument");
        auto h = std::make_unique<anh_runtime>();
        h->build(*cfg);
        *out = h.release();
    });
}
void anh_runtime_destroy(anh_runtime* h) { delete h; }
int anh_runtime_config(const anh_runtime* h, anh_net_config* out) {
    return guarded([&] { ANH_REQUIRE(h && out, "null argument"); *out = h->eng->spec.cfg; });
}
int anh_runtime_set_params(anh_runtime* h, const float* params, int64_t n_params, const float* running, int64_t n_running) {
    return guarded([&] {
        ANH_REQUIRE(h && params && running, "null argument");
        ANH_REQUIRE(n_params == h->eng->spec.n_params && n_running == h->eng->spec.n_running, "parameter blob size mismatch");
        h->set_params_all(params, running);
    });
}
int anh_runtime_get_params(const anh_runtime* h, float* params, int64_t n_params, float* running, int64_t n_running) {
    return guarded([&] {
        ANH_REQUIRE(h, "null handle");
        ANH_REQUIRE((!params || n_params == h->eng->spec.n_params) && (!running || n_running == h->eng->spec.n_running), "parameter blob size mismatch");
        DeviceScope scope(h->device_of(0));
        h->eng->get_params(params, running);
    });
}
int anh_runtime_serialize(const anh_runtime* h, void** blob, size_t* size) {
    return guarded([&] {
        ANH_REQUIRE(h && blob && size, "null argument");
        DeviceScope scope(h->device_of(0));
        const Spec& s = h->eng->spec;
        const size_t bytes = sizeof(BlobHeader) + (size_t)(s.n_params + s.n_running) * 4;
        char* p = (char*)std::malloc(bytes);
        if (!p) fail(ANH_ERR_OOM, "host allocation failed");
        BlobHeader hd{};
        std::memcpy(hd.magic, kRuntimeMagic, 8);
        hd.levels = s.cfg.levels; hd.in_channels = s.cfg.in_channels; hd.classes = s.cfg.classes; hd.min_filters = s.cfg.min_filters;
        hd.width_scaler = s.cfg.width_scaler; hd.n_params = s.n_params; hd.n_running = s.n_running;
        std::memcpy(p, &hd, sizeof hd);
        try { h->eng->get_params((float*)(p + sizeof hd), (float*)(p + sizeof hd) + s.n_params); }
        catch (...) { std::free(p); throw; }
        *blob = p; *size = bytes;
    });
}
int anh_runtime_deserialize(const void* blob, size_t size, int precision, anh_runtime** out) {
    return guarded([&] {
        ANH_REQUIRE(blob && out, "null argument");
        if (size < sizeof(BlobHeader)) fail(ANH_ERR_IO, "runtime blob truncated");
        BlobHeader hd;
        std::memcpy(&hd, blob, sizeof hd);
        if (std::memcmp(hd.magic, kRuntimeMagic, 8) != 0) fail(ANH_ERR_IO, "not an annonet_hip runtime blob");
        anh_net_config cfg{hd.levels, hd.in_channels, hd.classes, hd.width_scaler, hd.min_filters, precision};
        Spec s = Spec::build(cfg);
        if (s.n_params != hd.n_params || s.n_running != hd.n_running || size != sizeof hd + (size_t)(hd.n_params + hd.n_running) * 4)
            fail(ANH_ERR_IO, "runtime blob does not match its header");
        auto h = std::make_unique<anh_runtime>();
        h->build(cfg);
        const float* f = (const float*)((const char*)blob + sizeof hd);
        h->set_params_all(f, f + hd.n_params);
        *out = h.release();
    });
}

int anh_runtime_forward_device(anh_runtime* h, const uint8_t* d_image, int n, int height, int width, float* d_out_nchw) {
    return guarded([&] {
        ANH_REQUIRE(h && d_image && d_out_nchw, "null argument");
        DeviceScope scope(h->device_of(0));
        Src img;
        img.kind = SRC_IMAGE; img.img = d_image; img.img_h = height; img.img_w = width;
        img.img_sample_stride = (int64_t)height * width * h->eng->spec.cfg.in_channels;
        h->eng->forward_inference(img, n, height, width, d_out_nchw);
    });
}

int anh_runtime_forward(anh_runtime* h, const uint8_t* image, int n, int height, int width, const float** out, int* k, int* nr, int* nc) {
    return guarded([&] {
        ANH_REQUIRE(h && image && out, "null argument");
        ANH_REQUIRE(n >= 1 && height >= 1 && width >= 1, "empty input");
        DeviceScope scope(h->device_of(0));
        Engine& e = *h->eng;
        const int K = e.spec.cfg.classes, C = e.spec.cfg.in_channels;
        const size_t in_bytes = (size_t)n * height * width * C, out_elems = (size_t)n * K * height * width;
        e.stage_image.reserve(in_bytes);
        e.stage_out.reserve(out_elems * 4);
        HIP_CHECK(hipMemcpyAsync(e.stage_image.p, image, in_bytes, hipMemcpyHostToDevice, e.stream));
        Src img;
        img.kind = SRC_IMAGE; img.img = e.stage_image.as<uint8_t>(); img.img_h = height; img.img_w = width;
        img.img_sample_stride = (int64_t)height * width * C;
        e.forward_inference(img, n, height, width, e.stage_out.as<float>());
        e.host_out.resize(out_elems);
        HIP_CHECK(hipMemcpyAsync(e.host_out.data(), e.stage_out.p, out_elems * 4, hipMemcpyDeviceToHost, e.stream));
        e.synchronize();
        *out = e.host_out.data();
        if (k) *k = K;
        if (nr) *nr = height;
        if (nc) *nc = width;
    });
}

static std::vector<anh_tile> tiles_for(const anh_tiling_params* tiling, int width, int height) {
    if (tiling) return make_tiles(width, height, *tiling);
    anh_tile t;
    t.full_rect = {0, 0, width - 1, height - 1};
    t.unique_rect = t.full_rect;
    return {t};
}

int anh_infer_device(anh_runtime* h, const uint8_t* d_image, int height, int width, const double* gains, const anh_tiling_params* tiling,
                     const anh_tile* tiles, size_t n_tiles, uint16_t* d_result, float* d_blended) {
    return guarded([&] {
        ANH_REQUIRE(h && d_image && d_blended, "null argument");
        DeviceScope scope(h->device_of(0));
        std::vector<anh_tile> list = tiles ? std::vector<anh_tile>(tiles, tiles + n_tiles) : tiles_for(tiling, width, height);
        bool whole = tiles == nullptr;
        if (tiles && tiling) {   // a one-rank job hands over its "share": the complete tiling
            const std::vector<anh_tile> all = tiles_for(tiling, width, height);
            whole = all.size() == list.size() && std::memcmp(all.data(), list.data(), all.size() * sizeof(anh_tile)) == 0;
        }
        h->eng->infer_device(d_image, height, width, gains, list, d_result, d_blended, whole);
    });
}

int anh_argmax_device(anh_runtime* h, const float* d_blended, int height, int width, int row0, int row1, const double* gains, uint16_t* d_result) {
    return guarded([&] {
        ANH_REQUIRE(h && d_blended && d_result, "null argument");
        ANH_REQUIRE(height >= 1 && width >= 1 && row0 >= 0 && row0 <= row1 && row1 <= height, "argmax: bad row range");
        DeviceScope scope(h->device_of(0));
        h->eng->argmax_rows(d_blended, height, width, row0, row1, gains, d_result);
    });
}

namespace {
void ring_reserve(anh_runtime::Pinned& b, size_t bytes) {
    if (!b.done) HIP_CHECK(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
    if (bytes <= b.bytes) return;
    if (b.p) HIP_CHECK(hipHostFree(b.p));
    b.p = nullptr; b.bytes = 0;
    HIP_CHECK(hipHostMalloc(&b.p, bytes, hipHostMallocDefault));
    b.bytes = bytes;
}

// The streamed form of annonet_infer() for host buffers: the image is uploaded in row strips (host memcpy into a pinned ring
// -> async DMA on a copy stream), a tile starts as soon as the rows of its window are resident, and the label rows that no
// later tile can touch are arg-maxed and sent back (pinned ring -> host memcpy) while the next tile row computes.
void infer_streamed(anh_runtime* h, const uint8_t* image, int H, int W, const double* gains, const std::vector<anh_tile>& tiles, uint16_t* result) {
    Engine& e = *h->eng;
    const int K = e.spec.cfg.classes, C = e.spec.cfg.in_channels;
    const size_t plane = (size_t)H * W, row_bytes = (size_t)W * C;
    if (!h->copy_up) {
        HIP_CHECK(hipStreamCreateWithFlags(&h->copy_up, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&h->copy_down, hipStreamNonBlocking));
    }
    const int strip_rows = std::max(1, (int)std::min<size_t>((size_t)H, ((size_t)4 << 20) / std::max<size_t>(row_bytes, 1)));   // ~4 MiB strips
    const int n_strips = (H + strip_rows - 1) / strip_rows;
    while ((int)h->strip_events.size() < n_strips) {
        hipEvent_t ev;
        HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        h->strip_events.push_back(ev);
    }
    for (int i = 0; i < anh_runtime::kRing; ++i) { h->up[i].busy = false; h->down[i].busy = false; }
    uint8_t* d_image = e.stage_image.as<uint8_t>();
    float* d_blended = e.stage_blended.as<float>();
    uint16_t* d_labels = e.stage_result.as<uint16_t>();
    launch_fill_zero(d_blended, plane * K * 4, e.stream);
    const double* d_gains = e.upload_gains(gains);

    int uploaded = 0, next_strip = 0;
    auto upload_until = [&](int rows) {   // rows [0, rows) resident (as far as the copy stream is concerned)
        rows = std::min(rows, H);
        while (uploaded < rows) {
            anh_runtime::Pinned& b = h->up[next_strip % anh_runtime::kRing];
            const int r0 = uploaded, r1 = std::min(H, r0 + strip_rows);
            const size_t bytes = (size_t)(r1 - r0) * row_bytes;
            if (b.busy) HIP_CHECK(hipEventSynchronize(b.done));
            ring_reserve(b, (size_t)strip_rows * row_bytes);
            std::memcpy(b.p, image + (size_t)r0 * row_bytes, bytes);
            HIP_CHECK(hipMemcpyAsync(d_image + (size_t)r0 * row_bytes, b.p, bytes, hipMemcpyHostToDevice, h->copy_up));
            HIP_CHECK(hipEventRecord(b.done, h->copy_up));
            HIP_CHECK(hipEventRecord(h->strip_events[next_strip], h->copy_up));
            b.busy = true;
            uploaded = r1; ++next_strip;
        }
    };
    struct Down { int slot, r0, r1; };
    std::vector<Down> pending;
    auto drain = [&](size_t keep) {       // copy finished label strips out of the pinned ring
        while (pending.size() > keep) {
            const Down d = pending.front();
            pending.erase(pending.begin());
            anh_runtime::Pinned& b = h->down[d.slot];
            HIP_CHECK(hipEventSynchronize(b.done));
            std::memcpy(result + (size_t)d.r0 * W, b.p, (size_t)(d.r1 - d.r0) * W * 2);
            b.busy = false;
        }
    };
    hipEvent_t labels_ready;
    HIP_CHECK(hipEventCreateWithFlags(&labels_ready, hipEventDisableTiming));
    int labels_done = 0, down_slot = 0;
    const size_t max_down_rows = std::min<size_t>((size_t)H, std::max<size_t>(1, ((size_t)16 << 20) / ((size_t)W * 2)));   // label strips of <= 16 MiB
    auto send_labels = [&](int until) {   // rows [labels_done, until) are final on the compute stream
        while (labels_done < until) {
            const int r0 = labels_done, r1 = (int)std::min<size_t>((size_t)until, (size_t)r0 + max_down_rows);
            launch_argmax_range(d_blended, K, (int64_t)plane, (int64_t)r0 * W, (int64_t)r1 * W, d_gains, d_labels, e.stream);
            HIP_CHECK(hipEventRecord(labels_ready, e.stream));
            drain(anh_runtime::kRing - 1);
            anh_runtime::Pinned& b = h->down[down_slot];   // free: at most kRing - 1 strips are pending
            ring_reserve(b, max_down_rows * W * 2);
            HIP_CHECK(hipStreamWaitEvent(h->copy_down, labels_ready, 0));
            HIP_CHECK(hipMemcpyAsync(b.p, d_labels + (size_t)r0 * W, (size_t)(r1 - r0) * W * 2, hipMemcpyDeviceToHost, h->copy_down));
            HIP_CHECK(hipEventRecord(b.done, h->copy_down));
            b.busy = true;
            pending.push_back(Down{down_slot, r0, r1});
            down_slot = (down_slot + 1) % anh_runtime::kRing;
            labels_done = r1;
        }
    };
    // rows below final_after[i] receive nothing from the tiles after i
    std::vector<int> final_after(tiles.size());
    int lowest_top = H;
    for (size_t i = tiles.size(); i-- > 0;) { final_after[i] = std::min(lowest_top, H); lowest_top = std::min(lowest_top, (int)std::max(0l, (long)tiles[i].full_rect.top)); }
    auto rows_needed = [&](const TileWindow& w) { return std::min(H, std::max(1, w.top + w.height)); };   // clamp-to-edge reads reach row 0 / row H-1 at most
    for (size_t i = 0; i < tiles.size();) {
        // the tiles of one tile row (equal windows, same image rows) run as batches (Engine::infer_tiles)
        const TileWindow win = tile_window(tiles[i], e.spec.cfg.levels);
        const int need = rows_needed(win), limit = e.tile_batch(win.height, win.width);
        size_t j = i + 1;
        while (j < tiles.size() && j - i < (size_t)limit) {
            const TileWindow wj = tile_window(tiles[j], e.spec.cfg.levels);
            if (wj.height != win.height || wj.width != win.width || rows_needed(wj) != need) break;
            ++j;
        }
        upload_until(need);
        HIP_CHECK(hipStreamWaitEvent(e.stream, h->strip_events[(need - 1) / strip_rows], 0));
        e.infer_tiles(&tiles[i], (int)(j - i), d_image, H, W, d_blended);
        upload_until(uploaded + strip_rows * (int)(j - i));   // stay a few strips ahead of the tiles: one more strip per tile enqueued
        if (final_after[j - 1] > labels_done) send_labels(final_after[j - 1]);
        i = j;
    }
    upload_until(H);        // an image larger than its tiles' windows cannot occur, but keep the invariant
    send_labels(H);
    drain(0);
    HIP_CHECK(hipEventDestroy(labels_ready));
    e.synchronize();
}
}  // namespace

namespace {
// annonet_infer() on a handle with several replicas (anh_set_devices): the tile list is split into contiguous row-major chunks
// (tiles are independent, annonet_infer.cpp:46-165, except for the additive blended_output), every replica blends its chunk
// into planes of its own, ONE all-reduce sums the planes inside the rectangles where tiles of different replicas overlap
// (strips one receptive field wide), every replica labels the rows its tiles cover, and the host label map is assembled tile
// by tile from the owner of each tile (in an overlap both owners hold the same sums, hence the same labels).
void infer_multi(anh_runtime* h, const uint8_t* image, int H, int W, const double* gains, const std::vector<anh_tile>& tiles, uint16_t* result, float* blended_out) {
    const size_t R = h->replicas();
    const int K = h->eng->spec.cfg.classes, C = h->eng->spec.cfg.in_channels;
    const size_t plane = (size_t)H * W;
    const RectTable table = make_rect_table(cross_replica_overlaps(tiles, (int)R, W, H));
    const int64_t total = table.total();
    std::vector<float*> packed(R, nullptr);
    std::vector<hipStream_t> streams(R);
    std::vector<int64_t> lo(R), hi(R);
    h->each_replica([&](size_t r) {
        DeviceScope scope(h->device_of(r));
        Engine& e = h->replica(r);
        shard_range((int64_t)tiles.size(), (int)R, (int)r, lo[r], hi[r]);
        e.stage_image.reserve(plane * C);
        e.stage_blended.reserve(plane * K * 4);
        e.stage_result.reserve(plane * 2);
        HIP_CHECK(hipMemcpyAsync(e.stage_image.p, image, plane * C, hipMemcpyHostToDevice, e.stream));
        const std::vector<anh_tile> mine(tiles.begin() + lo[r], tiles.begin() + hi[r]);
        e.infer_device(e.stage_image.as<uint8_t>(), H, W, gains, mine, nullptr, e.stage_blended.as<float>());
        streams[r] = e.stream;
        if (total > 0) {
            anh_runtime::Exchange& x = h->exchange[r];
            x.packed.reserve((size_t)K * total * 4);
            x.rects.reserve(table.rects.size() * sizeof(anh_rect));
            x.offsets.reserve(table.offset.size() * sizeof(int64_t));
            HIP_CHECK(hipMemcpyAsync(x.rects.p, table.rects.data(), table.rects.size() * sizeof(anh_rect), hipMemcpyHostToDevice, e.stream));
            HIP_CHECK(hipMemcpyAsync(x.offsets.p, table.offset.data(), table.offset.size() * sizeof(int64_t), hipMemcpyHostToDevice, e.stream));
            launch_pack_rects(e.stage_blended.as<float>(), K, H, W, x.rects.as<anh_rect>(), x.offsets.as<int64_t>(), (int)table.rects.size(), total, x.packed.as<float>(), e.stream);
            packed[r] = x.packed.as<float>();
        }
    });
    // the table's host vectors must outlive the async uploads: every replica's stream passes the uploads before the collective returns control below
    if (total > 0) h->coll->all_reduce_sum(packed, (size_t)K * total, streams);   // the ONE exchange step of the path
    h->each_replica([&](size_t r) {
        DeviceScope scope(h->device_of(r));
        Engine& e = h->replica(r);
        if (hi[r] == lo[r]) return;
        if (total > 0) {
            anh_runtime::Exchange& x = h->exchange[r];
            launch_unpack_rects(e.stage_blended.as<float>(), K, H, W, x.rects.as<anh_rect>(), x.offsets.as<int64_t>(), (int)table.rects.size(), total, x.packed.as<float>(), e.stream);
        }
        long top = H, bottom = -1;
        for (int64_t i = lo[r]; i < hi[r]; ++i) { top = std::min(top, tiles[(size_t)i].full_rect.top); bottom = std::max(bottom, tiles[(size_t)i].full_rect.bottom); }
        const int row0 = (int)std::max(0L, top), row1 = (int)std::min((long)H, bottom + 1);
        const double* d_gains = e.upload_gains(gains);
        launch_argmax_range(e.stage_blended.as<float>(), K, (int64_t)plane, (int64_t)row0 * W, (int64_t)row1 * W, d_gains, e.stage_result.as<uint16_t>(), e.stream);
        for (int64_t i = lo[r]; i < hi[r]; ++i) {   // this replica's share of the host label map (and planes): its tiles' full rectangles
            const anh_rect& f = tiles[(size_t)i].full_rect;
            const long l = std::max(0L, f.left), t = std::max(0L, f.top), rt = std::min((long)W - 1, f.right), b = std::min((long)H - 1, f.bottom);
            if (l > rt || t > b) continue;
            const size_t off = (size_t)t * W + l;
            HIP_CHECK(hipMemcpy2DAsync(result + off, (size_t)W * 2, e.stage_result.as<uint16_t>() + off, (size_t)W * 2, (size_t)(rt - l + 1) * 2, (size_t)(b - t + 1),
                                       hipMemcpyDeviceToHost, e.stream));
            if (blended_out)
                for (int k = 0; k < K; ++k)
                    HIP_CHECK(hipMemcpy2DAsync(blended_out + k * plane + off, (size_t)W * 4, e.stage_blended.as<float>() + k * plane + off, (size_t)W * 4,
                                               (size_t)(rt - l + 1) * 4, (size_t)(b - t + 1), hipMemcpyDeviceToHost, e.stream));
        }
    });
    for (size_t r = 0; r < R; ++r) { DeviceScope scope(h->device_of(r)); h->replica(r).synchronize(); }
}
}  // namespace

int anh_infer(anh_runtime* h, const uint8_t* image, int height, int width, const double* gains, const double* detection_levels,
              const anh_tiling_params* tiling, uint16_t* result, float* blended_out) {
    return guarded([&] {
        ANH_REQUIRE(h && image && result, "null argument");
        ANH_REQUIRE(height >= 1 && width >= 1, "empty image");
        DeviceScope scope(h->device_of(0));
        Engine& e = *h->eng;
        const int K = e.spec.cfg.classes, C = e.spec.cfg.in_channels;
        const size_t plane = (size_t)height * width;
        std::vector<anh_tile> tiles = tiles_for(tiling, width, height);
        e.stage_image.reserve(plane * C);
        e.stage_blended.reserve(plane * K * 4);
        e.stage_result.reserve(plane * 2);
        bool use_det = false;
        if (detection_levels) for (int k = 0; k < K; ++k) { ANH_REQUIRE(detection_levels[k] >= 0.0, "detection levels must be >= 0"); if (detection_levels[k] > 0.0) use_det = true; }
        // several replicas: shard the tile list (the detection-level filter walks connected blobs of the WHOLE label map: it runs on
        // replica 0 alone, as does an image with fewer tiles than replicas would gain nothing)
        if (h->replicas() > 1 && !use_det && tiles.size() >= 2) { infer_multi(h, image, height, width, gains, tiles, result, blended_out); return; }
        static const bool streamed = !(getenv("ANH_INFER_STREAMED") && atoi(getenv("ANH_INFER_STREAMED")) == 0);
        if (streamed && !use_det && !blended_out) { infer_streamed(h, image, height, width, gains, tiles, result); return; }
        HIP_CHECK(hipMemcpyAsync(e.stage_image.p, image, plane * C, hipMemcpyHostToDevice, e.stream));
        e.infer_device(e.stage_image.as<uint8_t>(), height, width, gains, tiles, e.stage_result.as<uint16_t>(), e.stage_blended.as<float>(), /*whole_image=*/true);
        if (use_det) {
            // detection-level filter (annonet_infer.cpp:187-239), on the resident planes and label map.  Seeds are looked up at
            // (row, col): the reference stores (r, c) but reads (point.y(), point.x()) — transposed (:210 vs :222); see DESIGN.md.
            e.stage_out.reserve(plane + (size_t)K * sizeof(double) + 64);
            uint8_t* flags = e.stage_out.as<uint8_t>();
            double* d_det = reinterpret_cast<double*>(flags + ((plane + 15) / 16) * 16);
            int* d_changed = reinterpret_cast<int*>(d_det + K);
            HIP_CHECK(hipMemcpyAsync(d_det, detection_levels, (size_t)K * sizeof(double), hipMemcpyHostToDevice, e.stream));
            run_detection_filter(e.stage_blended.as<float>(), e.stage_result.as<uint16_t>(), K, height, width, d_det, flags, d_changed, e.stream);
        }
        HIP_CHECK(hipMemcpyAsync(result, e.stage_result.p, plane * 2, hipMemcpyDeviceToHost, e.stream));
        if (blended_out) HIP_CHECK(hipMemcpyAsync(blended_out, e.stage_blended.p, plane * K * 4, hipMemcpyDeviceToHost, e.stream));
        e.synchronize();
    });
}

int anh_runtime_set_stream(anh_runtime* h, void* s) { return guarded([&] { ANH_REQUIRE(h, "null handle"); ANH_REQUIRE(h->replicas() == 1, "a handle that drives several devices keeps its own streams"); h->eng->set_stream((hipStream_t)s); }); }
int anh_runtime_get_stream(anh_runtime* h, void** s) { return guarded([&] { ANH_REQUIRE(h && s, "null argument"); *s = (void*)h->eng->stream; }); }
int anh_runtime_stores_activations(anh_runtime* h, int* yes) { return guarded([&] { ANH_REQUIRE(h && yes, "null argument"); *yes = h->eng->infer_post ? 1 : 0; }); }
int anh_runtime_synchronize(anh_runtime* h) {
    return guarded([&] { ANH_REQUIRE(h, "null handle"); for (size_t r = 0; r < h->replicas(); ++r) { DeviceScope scope(h->device_of(r)); h->replica(r).synchronize(); } });
}

// ---- TrainingNet ----
int anh_trainer_create(anh_trainer** out) {
    return guarded([&] { ANH_REQUIRE(out, "null argument"); auto h = std::make_unique<anh_trainer>(); h->devices = selected_devices(); *out = h.release(); });
}
void anh_trainer_destroy(anh_trainer* h) { delete h; }

#define TRAINER_SETTER(name, body) \
    return guarded([&] { ANH_REQUIRE(h, "null handle"); body; })

int anh_trainer_set_net_width(anh_trainer* h, double scaler, int min_filters) {
    TRAINER_SETTER(net_width, { ANH_REQUIRE(scaler > 0 && min_filters >= 1, "bad net width");
        if (scaler != h->cfg.width_scaler || min_filters != h->cfg.min_filters) h->structural_change();
        h->cfg.width_scaler = scaler; h->cfg.min_filters = min_filters; });
}
int anh_trainer_set_class_count(anh_trainer* h, size_t classes) {
    TRAINER_SETTER(class_count, { ANH_REQUIRE(classes >= 1 && classes <= 64, "class count must be 1..64");
        if ((int)classes != h->cfg.classes) h->structural_change();
        h->cfg.classes = (int)classes; });
}
int anh_trainer_set_levels(anh_trainer* h, int levels) { TRAINER_SETTER(levels, { ANH_REQUIRE(levels >= 0 && levels <= 3, "level count must be 0..3"); if (levels != h->cfg.levels) h->structural_change(); h->cfg.levels = levels; }); }
int anh_trainer_set_input_channels(anh_trainer* h, int c) { TRAINER_SETTER(channels, { ANH_REQUIRE(c == 1 || c == 3, "input channels must be 1 or 3"); if (c != h->cfg.in_channels) h->structural_change(); h->cfg.in_channels = c; }); }
int anh_trainer_set_precision(anh_trainer* h, int p) { TRAINER_SETTER(precision, { ANH_REQUIRE(p == ANH_FP32 || p == ANH_BF16, "unknown precision"); if (p != h->cfg.precision) h->structural_change(); h->cfg.precision = p; }); }
int anh_trainer_set_seed(anh_trainer* h, uint64_t seed) { TRAINER_SETTER(seed, { if (seed != h->seed) h->structural_change(); h->seed = seed; }); }

int anh_trainer_initialize(anh_trainer* h) {
    return guarded([&] {
        ANH_REQUIRE(h, "null handle");
        if (anh_device_count() == 0) fail(ANH_ERR_DEVICE, "no MI355X / HIP device visible");
        (void)Spec::build(h->cfg);  // validates the configuration
        h->initialized = true;
        h->dirty = true;
        if (!h->loss_ring) {
            // portable + mapped: the update kernel of replica 0 writes it, whichever device that replica lives on (anh_set_devices may come later)
            HIP_CHECK(hipHostMalloc((void**)&h->loss_ring, 256 * sizeof(unsigned long long), hipHostMallocPortable | hipHostMallocMapped));
            std::memset(h->loss_ring, 0, 256 * sizeof(unsigned long long));
            HIP_CHECK(hipHostGetDevicePointer((void**)&h->loss_ring_dev, h->loss_ring, 0));
        }
    });
}
int anh_trainer_set_learning_rate(anh_trainer* h, double lr) { TRAINER_SETTER(lr, { ANH_REQUIRE(lr > 0, "learning rate must be positive"); h->sched.lr = lr; }); }
int anh_trainer_set_learning_rate_shrink_factor(anh_trainer* h, double f) { TRAINER_SETTER(shrink, { ANH_REQUIRE(f > 0 && f <= 1, "shrink factor must be in (0,1]"); h->sched.shrink = f; }); }
int anh_trainer_set_iterations_without_progress_threshold(anh_trainer* h, unsigned long n) { TRAINER_SETTER(thresh, { h->sched.threshold = n; }); }
int anh_trainer_set_previous_loss_values_dump_amount(anh_trainer* h, unsigned long n) { TRAINER_SETTER(dump, { h->sched.dump_amount = n; }); }
int anh_trainer_set_all_bn_running_stats_window_sizes(anh_trainer* h, unsigned long n) { TRAINER_SETTER(window, { ANH_REQUIRE(n >= 1, "window must be >= 1"); h->bn_window = n; }); }
int anh_trainer_set_sgd(anh_trainer* h, double wd, double mom) { TRAINER_SETTER(sgd, { ANH_REQUIRE(wd >= 0 && mom >= 0 && mom < 1, "bad sgd parameters"); h->weight_decay = wd; h->momentum = mom; }); }
int anh_trainer_be_verbose(anh_trainer* h) { TRAINER_SETTER(verbose, { h->verbose = true; }); }
int anh_trainer_set_synchronization_file(anh_trainer* h, const char* path, double seconds) {
    return guarded([&] {
        ANH_REQUIRE(h && path, "null argument");
        h->sync_path = path; h->sync_seconds = seconds;
        h->last_sync = std::chrono::steady_clock::now();
        // dlib's trainer resumes from an existing synchronization file.  The reference calls this BEFORE SetClassCount
        // (annonet_train_main.cpp:400-405), so the file is read on the first use of the net, once every structural setter ran.
        h->resume_pending = h->steps == 0;
    });
}
double anh_trainer_get_learning_rate(const anh_trainer* h) {
    if (!h) return 0;
    // a pending resume may change the rate: the host polls this in its loop condition before the first step (annonet_train_main.cpp:583).
    // This getter has no status to return: a resume that fails here leaves resume_pending set and the message in anh_last_error(), and
    // the next call with a status (StartTraining) fails with the same error.
    if (h->resume_pending && h->initialized) (void)guarded([&] { (void)const_cast<anh_trainer*>(h)->engine(); });
    return h->sched.lr;   // changes only inside apply_update, at the fixed loss lag: no timing-dependent polling
}
double anh_trainer_get_last_loss(anh_trainer* h) {
    double v = NAN;
    guarded([&] { ANH_REQUIRE(h, "null handle"); h->engine().synchronize(); h->fetch_all(); v = h->steps ? h->last_loss : h->engine().read_loss(); });
    return v;
}
unsigned long anh_trainer_get_step_count(const anh_trainer* h) { return h ? h->steps : 0; }
int anh_trainer_config(const anh_trainer* h, anh_net_config* out) { return guarded([&] { ANH_REQUIRE(h && out, "null argument"); *out = h->cfg; }); }

int anh_trainer_forward_backward_device(anh_trainer* h, const uint8_t* d_images, const uint16_t* d_labels, const float* d_weights,
                                        int n, int height, int width, double loss_scale_n) {
    return guarded([&] {
        ANH_REQUIRE(h && d_images && d_labels && d_weights, "null argument");
        ANH_REQUIRE(loss_scale_n > 0, "loss scale batch must be positive");
        ANH_REQUIRE(h->replicas() == 1, "device-resident steps drive ONE device: a handle over several devices takes host mini-batches (anh_trainer_step)");
        DeviceScope scope(h->device_of(0));
        Engine& e = h->engine();
        Src img;
        img.kind = SRC_IMAGE; img.img = d_images; img.img_h = height; img.img_w = width;
        img.img_sample_stride = (int64_t)height * width * e.spec.cfg.in_channels;
        e.bn_window = h->bn_window;
        e.forward_training(img, n, height, width);
        e.backward(d_labels, d_weights, loss_scale_n);
    });
}

int anh_trainer_apply_update(anh_trainer* h, double grad_scale) {
    return guarded([&] {
        ANH_REQUIRE(h, "null handle");
        Engine& e = h->engine();
        // this is step h->steps + 1: its rate is decided by the losses of steps <= h->steps + 1 - kLossLag (deterministic lag)
        if (h->steps + 1 > anh_trainer::kLossLag) h->record_until(h->steps + 1 - anh_trainer::kLossLag);
        for (size_t r = 1; r < h->replicas(); ++r) { DeviceScope scope(h->device_of(r)); h->replica(r).apply_update(h->sched.lr, h->weight_decay, h->momentum, grad_scale, h->bn_window); }
        DeviceScope scope0(h->device_of(0));
        // the update kernel also ships this step's loss (gradient bucket's trailing slot: already all-reduced under data parallelism)
        const int slot = h->next_slot;
        h->next_slot = (h->next_slot + 1) % 256;
        e.apply_update(h->sched.lr, h->weight_decay, h->momentum, grad_scale, h->bn_window, h->loss_ring_dev + slot, anh_trainer::loss_tag(h->steps + 1));
        ++h->steps;
        h->pending.push_back({slot, h->steps, false, 0.0});
        if (h->verbose && h->steps % 100 == 0) {
            if (!h->pending.empty()) { h->arrive(h->pending.front()); h->last_loss = h->pending.front().value; }   // the oldest loss in flight
            std::printf("step#: %lu  learning rate: %g  loss: %g  steps without apparent progress: %lu\n", h->steps, h->sched.lr, h->last_loss,
                        h->sched.steps_without_progress);
            std::fflush(stdout);
        }
        if (!h->sync_path.empty() && h->sync_seconds > 0) {
            const auto now = std::chrono::steady_clock::now();
            if (std::chrono::duration<double>(now - h->last_sync).count() >= h->sync_seconds) {
                h->last_sync = now;
                const int rc = anh_trainer_save_state(h, h->sync_path.c_str());
                if (rc != ANH_OK) fail(rc, g_error);
            }
        }
    });
}

extern "C++" {
namespace {
// Packs samples of a host mini-batch into one of a replica's two pinned staging sets, uploads them on the replica's copy stream
// and enqueues forward + backward behind the upload.  Returns once the inputs are packed (the host refills its vectors right
// after StartTraining returns, annonet_train_main.cpp:585-586); packing step k+1 overlaps the GPU work of step k.
anh_trainer::StageSet& stage_and_run(anh_trainer* h, Engine& e, anh_trainer::StageSet* sets, hipStream_t& copy_stream, const uint8_t* const* images,
                                     const anh_wlabel* const* labels, int n, int height, int width, double loss_scale_n) {
    const int C = e.spec.cfg.in_channels, K = e.spec.cfg.classes;
    const size_t plane = (size_t)height * width;
    const size_t img_bytes = (size_t)n * plane * C, lab_off = (img_bytes + 255) / 256 * 256, lab_bytes = (size_t)n * plane * 2;
    const size_t w_off = (lab_off + lab_bytes + 255) / 256 * 256, total = w_off + (size_t)n * plane * 4;
    anh_trainer::StageSet& st = sets[h->host_steps & 1];
    if (!copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    if (!st.uploaded) {
        HIP_CHECK(hipEventCreateWithFlags(&st.uploaded, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&st.consumed, hipEventDisableTiming));
    }
    if (st.in_flight) {   // the pinned block is free again once the upload of step k-2 is done: the one place a call blocks on the GPU
        const auto w0 = std::chrono::steady_clock::now();
        wait_event(st.uploaded, h->replicas() > 1);
        st.wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
    } else st.wait_us = 0;
    if (total > st.pinned_bytes) {
        if (st.in_flight) wait_event(st.consumed, h->replicas() > 1);
        if (st.pinned) HIP_CHECK(hipHostFree(st.pinned));
        st.pinned = nullptr; st.pinned_bytes = 0;
        HIP_CHECK(hipHostMalloc(&st.pinned, total, hipHostMallocDefault));
        st.pinned_bytes = total;
        st.dev.reserve(total);
    }
    // ---- pack: images as they are, weighted labels split into a u16 and an f32 plane; labels validated on the way ----
    uint8_t* base = static_cast<uint8_t*>(st.pinned);
    uint16_t* plab = reinterpret_cast<uint16_t*>(base + lab_off);
    float* pw = reinterpret_cast<float*>(base + w_off);
    bool bad_label = false;
    for (int i = 0; i < n; ++i) {
        std::memcpy(base + (size_t)i * plane * C, images[i], plane * C);
        const anh_wlabel* src = labels[i];
        uint16_t* dl = plab + (size_t)i * plane;
        float* dw = pw + (size_t)i * plane;
        unsigned worst = 0;
        for (size_t p = 0; p < plane; ++p) {
            const uint16_t l = src[p].label;
            dl[p] = l; dw[p] = src[p].weight;
            worst |= (unsigned)(l != ANH_LABEL_IGNORE && l >= K);
        }
        bad_label = bad_label || worst != 0;
    }
    ANH_REQUIRE(!bad_label, "label value exceeds the class count");
    // ---- upload behind the step that last read this device block, compute behind the upload ----
    if (st.in_flight) HIP_CHECK(hipStreamWaitEvent(copy_stream, st.consumed, 0));
    HIP_CHECK(hipMemcpyAsync(st.dev.p, st.pinned, total, hipMemcpyHostToDevice, copy_stream));
    HIP_CHECK(hipEventRecord(st.uploaded, copy_stream));
    HIP_CHECK(hipStreamWaitEvent(e.stream, st.uploaded, 0));
    uint8_t* dbase = st.dev.as<uint8_t>();
    Src img;
    img.kind = SRC_IMAGE; img.img = dbase; img.img_h = height; img.img_w = width;
    img.img_sample_stride = (int64_t)height * width * C;
    e.bn_window = h->bn_window;
    e.forward_training(img, n, height, width);
    e.backward(reinterpret_cast<uint16_t*>(dbase + lab_off), reinterpret_cast<float*>(dbase + w_off), loss_scale_n);
    return st;
}
}  // namespace
}  // extern "C++"

int anh_trainer_step(anh_trainer* h, const uint8_t* const* images, const anh_wlabel* const* labels, int n, int height, int width) {
    return guarded([&] {
        ANH_REQUIRE(h && images && labels, "null argument");
        ANH_REQUIRE(n >= 1 && height >= 1 && width >= 1, "empty mini-batch");
        for (int i = 0; i < n; ++i) ANH_REQUIRE(images[i] && labels[i], "null sample");
        (void)h->engine();   // builds every replica on first use
        const size_t R = h->replicas();
        ANH_REQUIRE((size_t)n >= R, "mini-batch smaller than the number of devices");
        // data parallel (annonet_train_main.cpp:583-614): replica r takes samples [n r / R, n (r+1) / R); the loss
        // scale 1/(N nr nc) uses the WHOLE batch, so the exchange step is a plain sum of the gradient buckets
        std::vector<anh_trainer::StageSet*> used(R, nullptr);
        const auto host_t0 = std::chrono::steady_clock::now();
        h->each_replica([&](size_t r) {   // packing, upload and the step's launches of every replica in parallel (persistent worker threads)
            DeviceScope scope(h->device_of(r));
            int64_t lo, hi;
            shard_range(n, (int)R, (int)r, lo, hi);
            used[r] = &stage_and_run(h, h->replica(r), h->stage_of(r), h->copy_stream_of(r), images + lo, labels + lo, (int)(hi - lo), height, width, (double)n);
        });
        if (R > 1) {
            std::vector<float*> buckets(R);
            std::vector<hipStream_t> streams(R);
            for (size_t r = 0; r < R; ++r) { buckets[r] = h->replica(r).grad_bucket(); streams[r] = h->replica(r).stream; }
            const int64_t count = (int64_t)h->eng->spec.n_params + 1, first = h->eng->early_grad_first();   // the trailing slot carries the loss
            static const bool early_on = !(getenv("ANH_EARLY_REDUCE") && atoi(getenv("ANH_EARLY_REDUCE")) == 0);
            static const int xs_every = getenv("ANH_EXCHANGE_SAMPLE") ? atoi(getenv("ANH_EXCHANGE_SAMPLE")) : 8;
            anh_trainer::ExchangeStats& xs = h->xs;
            const bool split = early_on && first > 0 && first < count - 1;
            bool sample = false;
            if (xs_every > 0 && h->host_steps % (unsigned long)xs_every == (unsigned long)xs_every - 1) {
                h->collect_exchange_sample();   // (the previous sampled step ended long ago)
                DeviceScope scope(h->device_of(0));
                if (!xs.head0) for (hipEvent_t* ev : {&xs.tail0, &xs.tail1, &xs.head0, &xs.head1}) HIP_CHECK(hipEventCreate(ev));
                sample = true;
                xs.split = split;
            }
            if (split) {
                // the tail of the bucket (every layer but the first two, head, loss) is final while backward still runs
                // (Engine::ev_early_grads): reduce it on side streams now, the short head on the replicas' own streams afterwards
                std::vector<float*> tails(R);
                std::vector<hipStream_t> side(R);
                for (size_t r = 0; r < R; ++r) {
                    DeviceScope scope(h->device_of(r));
                    Engine& e = h->replica(r);
                    side[r] = e.early_reduce_stream();
                    HIP_CHECK(hipStreamWaitEvent(side[r], e.ev_early_grads, 0));
                    tails[r] = buckets[r] + first;
                }
                if (sample) { DeviceScope scope(h->device_of(0)); HIP_CHECK(hipEventRecord(xs.tail0, side[0])); }
                h->coll->all_reduce_sum(tails, (size_t)(count - first), side);
                if (sample) { DeviceScope scope(h->device_of(0)); HIP_CHECK(hipEventRecord(xs.tail1, side[0])); HIP_CHECK(hipEventRecord(xs.head0, streams[0])); }
                h->coll->all_reduce_sum(buckets, (size_t)first, streams);
                if (sample) { DeviceScope scope(h->device_of(0)); HIP_CHECK(hipEventRecord(xs.head1, streams[0])); xs.pending = true; }
                for (size_t r = 0; r < R; ++r) {   // the update reads the whole bucket
                    DeviceScope scope(h->device_of(r));
                    Engine& e = h->replica(r);
                    HIP_CHECK(hipEventRecord(e.ev_early_reduced, side[r]));
                    HIP_CHECK(hipStreamWaitEvent(e.stream, e.ev_early_reduced, 0));
                }
            } else {
                if (sample) { DeviceScope scope(h->device_of(0)); HIP_CHECK(hipEventRecord(xs.head0, streams[0])); }
                h->coll->all_reduce_sum(buckets, (size_t)count, streams);
                if (sample) { DeviceScope scope(h->device_of(0)); HIP_CHECK(hipEventRecord(xs.head1, streams[0])); xs.pending = true; }
            }
        }
        const int rc = anh_trainer_apply_update(h, 1.0);
        if (rc != ANH_OK) fail(rc, g_error);
        for (size_t r = 0; r < R; ++r) {
            DeviceScope scope(h->device_of(r));
            HIP_CHECK(hipEventRecord(used[r]->consumed, h->replica(r).stream));
            used[r]->in_flight = true;
        }
        ++h->host_steps;
        h->xs.host_us_last = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_t0).count();
        h->xs.host_us_sum += h->xs.host_us_last;
        h->xs.wait_us_sum += used[0]->wait_us;   // replica 0 packs on the calling thread: its wait is wall time of the call
        ++h->xs.calls;
    });
}

int anh_trainer_exchange_stats(anh_trainer* h, anh_exchange_stats* out) {
    return guarded([&] {
        ANH_REQUIRE(h && out, "null argument");
        h->collect_exchange_sample();
        const anh_trainer::ExchangeStats& xs = h->xs;
        *out = anh_exchange_stats{};
        out->replicas = (int)h->replicas();
        out->steps = xs.calls;
        out->host_us_mean = xs.calls ? xs.host_us_sum / (double)xs.calls : 0.0;
        out->host_us_last = xs.host_us_last;
        out->host_wait_us_mean = xs.calls ? xs.wait_us_sum / (double)xs.calls : 0.0;
        out->samples = xs.samples;
        out->allreduce_tail_us_mean = xs.samples ? xs.tail_us_sum / (double)xs.samples : 0.0;
        out->allreduce_head_us_mean = xs.samples ? xs.head_us_sum / (double)xs.samples : 0.0;
        out->allreduce_tail_us_last = xs.tail_us_last;
        out->allreduce_head_us_last = xs.head_us_last;
        out->early_reduce = xs.split ? 1 : 0;
        out->uses_rccl = h->coll ? h->coll->transport() : 0;
        out->rccl_version = Collective::rccl_version();
        out->worker_calls = h->workers ? (int64_t)(h->workers->calls() - xs.worker_calls_base) : 0;
        out->bucket_bytes = h->eng ? ((int64_t)h->eng->spec.n_params + 1) * 4 : 0;
    });
}
void anh_trainer_reset_exchange_stats(anh_trainer* h) {
    if (!h) return;
    try { h->collect_exchange_sample(); } catch (...) {}
    anh_trainer::ExchangeStats& xs = h->xs;
    xs.calls = 0; xs.host_us_sum = 0; xs.wait_us_sum = 0; xs.samples = 0; xs.tail_us_sum = 0; xs.head_us_sum = 0;
    xs.worker_calls_base = h->workers ? h->workers->calls() : 0;
}

int anh_trainer_grad_buffer(anh_trainer* h, void** d_ptr, int64_t* count) {
    return guarded([&] { ANH_REQUIRE(h && d_ptr && count, "null argument"); *d_ptr = h->engine().grad_bucket(); *count = h->engine().spec.n_params + 1; });
}
int anh_trainer_get_params(anh_trainer* h, float* params, int64_t n_params, float* running, int64_t n_running) {
    return guarded([&] {
        ANH_REQUIRE(h, "null handle");
        Engine& e = h->engine();
        ANH_REQUIRE((!params || n_params == e.spec.n_params) && (!running || n_running == e.spec.n_running), "parameter blob size mismatch");
        e.get_params(params, running);
    });
}
int anh_trainer_set_params(anh_trainer* h, const float* params, int64_t n_params, const float* running, int64_t n_running) {
    return guarded([&] {
        ANH_REQUIRE(h && params && running, "null argument");
        Engine& e = h->engine();
        ANH_REQUIRE(n_params == e.spec.n_params && n_running == e.spec.n_running, "parameter blob size mismatch");
        for (size_t r = 0; r < h->replicas(); ++r) { DeviceScope scope(h->device_of(r)); h->replica(r).synchronize(); h->replica(r).set_params(params, running); }
    });
}
int anh_trainer_replica_params(anh_trainer* h, int replica, float* params, int64_t n_params) {
    return guarded([&] {
        ANH_REQUIRE(h && params, "null argument");
        (void)h->engine();
        ANH_REQUIRE(replica >= 0 && (size_t)replica < h->replicas(), "replica index out of range");
        ANH_REQUIRE(n_params == h->eng->spec.n_params, "size mismatch");
        DeviceScope scope(h->device_of((size_t)replica));
        h->replica((size_t)replica).get_params(params, nullptr);
    });
}
int anh_trainer_get_grads(anh_trainer* h, float* grads, int64_t n_params) {
    return guarded([&] { ANH_REQUIRE(h && grads, "null argument"); ANH_REQUIRE(n_params == h->engine().spec.n_params, "size mismatch"); h->engine().get_grads_canonical(grads); });
}
int anh_trainer_get_momentum(anh_trainer* h, float* m, int64_t n_params) {
    return guarded([&] { ANH_REQUIRE(h && m, "null argument"); ANH_REQUIRE(n_params == h->engine().spec.n_params, "size mismatch"); h->engine().get_momentum(m); });
}
int anh_trainer_set_momentum(anh_trainer* h, const float* m, int64_t n_params) {
    return guarded([&] {
        ANH_REQUIRE(h && m, "null argument");
        ANH_REQUIRE(n_params == h->engine().spec.n_params, "size mismatch");
        for (size_t r = 0; r < h->replicas(); ++r) { DeviceScope scope(h->device_of(r)); h->replica(r).set_momentum(m); }
    });
}

int anh_trainer_snapshot_runtime(anh_trainer* h, int precision, anh_runtime** out) {
    return guarded([&] {
        ANH_REQUIRE(h && out, "null argument");
        Engine& e = h->engine();
        std::vector<float> p((size_t)e.spec.n_params), r((size_t)e.spec.n_running);
        e.get_params(p.data(), r.data());  // synchronises: a step in flight finishes first (annonet_train_main.cpp:558)
        anh_net_config cfg = e.spec.cfg;
        cfg.precision = precision;
        // A snapshot is ONE replica on the trainer's first device: the host serializes it (annonet_train_main.cpp:557-565, at step 0,
        // every save interval and at the end) while the trainer's own communicator may have collectives in flight — no second set of
        // engines, no second ncclCommInitAll.  Handles the caller creates for inference (anh_runtime_create / _deserialize) span the
        // selected devices.
        auto rt = std::make_unique<anh_runtime>();
        rt->build(cfg, h->replicas() > 1 ? h->device_of(0) : -1);
        rt->set_params_all(p.data(), r.data());
        *out = rt.release();
    });
}

int anh_trainer_save_state(anh_trainer* h, const char* path) {
    return guarded([&] {
        ANH_REQUIRE(h && path, "null argument");
        Engine& e = h->engine();
        e.synchronize();
        h->fetch_all();   // the losses still ahead of the schedule's lag are stored as such: a resumed run records them when an uninterrupted one would
        const Spec& s = e.spec;
        std::vector<float> p((size_t)s.n_params), m((size_t)s.n_params), r((size_t)s.n_running);
        e.get_params(p.data(), r.data());
        e.get_momentum(m.data());
        const std::vector<double> updates = e.get_running_updates();
        const std::string tmp = std::string(path) + ".tmp";
        {
            std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
            if (!f) fail(ANH_ERR_IO, "cannot write " + tmp);
            BlobHeader hd{};
            std::memcpy(hd.magic, kStateMagic, 8);
            hd.levels = s.cfg.levels; hd.in_channels = s.cfg.in_channels; hd.classes = s.cfg.classes; hd.min_filters = s.cfg.min_filters;
            hd.width_scaler = s.cfg.width_scaler; hd.n_params = s.n_params; hd.n_running = s.n_running;
            f.write((const char*)&hd, sizeof hd);
            const uint64_t steps = h->steps, nh = h->sched.history.size(), budget = h->sched.check_budget, swp = h->sched.steps_without_progress;
            const uint64_t n_unrec = h->pending.size(), n_bn = updates.size();
            f.write((const char*)&steps, 8); f.write((const char*)&h->sched.lr, 8); f.write((const char*)&budget, 8); f.write((const char*)&swp, 8);
            f.write((const char*)&h->last_loss, 8);
            f.write((const char*)&nh, 8);
            for (double v : h->sched.history) f.write((const char*)&v, 8);
            f.write((const char*)&n_unrec, 8);
            for (const auto& pl : h->pending) { const uint64_t st = pl.step; f.write((const char*)&st, 8); f.write((const char*)&pl.value, 8); }
            f.write((const char*)&n_bn, 8);
            for (double v : updates) f.write((const char*)&v, 8);
            f.write((const char*)p.data(), p.size() * 4); f.write((const char*)m.data(), m.size() * 4); f.write((const char*)r.data(), r.size() * 4);
            if (!f) fail(ANH_ERR_IO, "short write to " + tmp);
        }
        if (std::rename(tmp.c_str(), path) != 0) fail(ANH_ERR_IO, std::string("cannot move state file into place: ") + path);
    });
}

namespace {
void load_state_into(anh_trainer* h, Engine& e, const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail(ANH_ERR_IO, std::string("cannot read ") + path);
    BlobHeader hd;
    f.read((char*)&hd, sizeof hd);
    if (!f || std::memcmp(hd.magic, kStateMagic, 8) != 0) fail(ANH_ERR_IO, "not an annonet_hip trainer state file (or one of an older format)");
    const Spec& s = e.spec;
    if (hd.levels != s.cfg.levels || hd.in_channels != s.cfg.in_channels || hd.classes != s.cfg.classes || hd.n_params != s.n_params || hd.n_running != s.n_running)
        fail(ANH_ERR_IO, "trainer state file was written for a different net");
    uint64_t steps = 0, nh = 0, budget = 0, swp = 0, n_unrec = 0, n_bn = 0;
    double lr = 0, last_loss = 0;
    f.read((char*)&steps, 8); f.read((char*)&lr, 8); f.read((char*)&budget, 8); f.read((char*)&swp, 8); f.read((char*)&last_loss, 8); f.read((char*)&nh, 8);
    if (!f || nh > (1u << 26)) fail(ANH_ERR_IO, "trainer state file is corrupt");
    std::deque<double> hist;
    for (uint64_t i = 0; i < nh; ++i) { double v; f.read((char*)&v, 8); hist.push_back(v); }
    f.read((char*)&n_unrec, 8);
    if (!f || n_unrec > 4096) fail(ANH_ERR_IO, "trainer state file is corrupt");
    std::deque<anh_trainer::PendingLoss> unrec;
    for (uint64_t i = 0; i < n_unrec; ++i) { uint64_t st; double v; f.read((char*)&st, 8); f.read((char*)&v, 8); unrec.push_back({0, (unsigned long)st, true, v}); }
    f.read((char*)&n_bn, 8);
    if (!f || n_bn > 4096) fail(ANH_ERR_IO, "trainer state file is corrupt");
    std::vector<double> updates((size_t)n_bn);
    for (auto& v : updates) f.read((char*)&v, 8);
    std::vector<float> p((size_t)s.n_params), m((size_t)s.n_params), r((size_t)s.n_running);
    f.read((char*)p.data(), p.size() * 4); f.read((char*)m.data(), m.size() * 4); f.read((char*)r.data(), r.size() * 4);
    if (!f) fail(ANH_ERR_IO, "trainer state file is truncated");
    for (size_t rr = 0; rr < h->replicas(); ++rr) {   // every replica resumes from the same state
        DeviceScope scope(h->device_of(rr));
        Engine& er = h->replica(rr);
        er.synchronize();
        er.set_params(p.data(), r.data());
        er.set_momentum(m.data());
        er.set_running_updates(updates);
    }
    h->pending = unrec;
    h->steps = (unsigned long)steps; h->last_loss = last_loss;
    h->sched.lr = lr; h->sched.check_budget = (unsigned long)budget; h->sched.steps_without_progress = (unsigned long)swp; h->sched.history = hist;
}
}  // namespace

void anh_trainer::resume() { load_state_into(this, *eng, sync_path.c_str()); }

int anh_trainer_load_state(anh_trainer* h, const char* path) {
    return guarded([&] {
        ANH_REQUIRE(h && path, "null argument");
        h->resume_pending = false;   // an explicit load wins over the synchronization file
        load_state_into(h, h->engine(), path);
    });
}

int anh_trainer_set_stream(anh_trainer* h, void* s) {
    return guarded([&] { ANH_REQUIRE(h, "null handle"); ANH_REQUIRE(h->replicas() == 1, "a handle that drives several devices keeps its own streams"); h->engine().set_stream((hipStream_t)s); });
}
int anh_trainer_get_stream(anh_trainer* h, void** s) { return guarded([&] { ANH_REQUIRE(h && s, "null argument"); *s = (void*)h->engine().stream; }); }
int anh_trainer_early_grads(anh_trainer* h, int64_t* first) {
    return guarded([&] {
        ANH_REQUIRE(h && first, "null argument");
        Engine& e = h->engine();
        *first = h->replicas() > 1 ? (int64_t)e.spec.n_params + 1 : e.early_grad_first();   // (several replicas: the library reduces the buckets itself)
    });
}
int anh_trainer_step_graph_stats(anh_trainer* h, int64_t* captures, int64_t* launches) {
    return guarded([&] {
        ANH_REQUIRE(h && captures && launches, "null argument");
        Engine& e = h->engine();
        *captures = (int64_t)e.step_graph_captures; *launches = (int64_t)e.step_graph_launches;
    });
}
int anh_trainer_wait_early_grads(anh_trainer* h, void* hip_stream) {
    return guarded([&] {
        ANH_REQUIRE(h && hip_stream, "null argument");
        Engine& e = h->engine();
        ANH_REQUIRE(e.early_grad_first() <= e.spec.n_params, "this net has no early gradient part");
        HIP_CHECK(hipStreamWaitEvent((hipStream_t)hip_stream, e.ev_early_grads, 0));
    });
}
int anh_trainer_synchronize(anh_trainer* h) {
    return guarded([&] {
        ANH_REQUIRE(h, "null handle");
        (void)h->engine();
        bool bad = false;
        for (size_t r = 0; r < h->replicas(); ++r) {
            DeviceScope scope(h->device_of(r));
            h->replica(r).synchronize();
            bad = h->replica(r).read_error_flag_and_clear() || bad;
        }
        if (bad) fail(ANH_ERR_INVALID, "a label value exceeds the class count");
    });
}
int anh_trainer_layer_tensor(anh_trainer* h, int layer, int which, float* out, int64_t capacity, int dims4[4]) {
    return guarded([&] { ANH_REQUIRE(h && dims4, "null argument"); h->engine().layer_tensor(layer, which, out, capacity, dims4); });
}

// ---- profiling ----
static Engine* engine_of(void* handle, int is_trainer) {
    ANH_REQUIRE(handle, "null handle");
    return is_trainer ? &((anh_trainer*)handle)->engine() : ((anh_runtime*)handle)->eng.get();
}
int anh_profile_enable(void* handle, int is_trainer, int enable) {
    return guarded([&] { Engine* e = engine_of(handle, is_trainer); e->synchronize(); e->prof.enabled = enable != 0; });
}
int anh_profile_set_filter(void* handle, int is_trainer, const char* substring) {
    return guarded([&] { Engine* e = engine_of(handle, is_trainer); e->synchronize(); e->prof.filter = substring ? substring : ""; });
}
int anh_profile_set_sampling(void* handle, int is_trainer, int every) {
    return guarded([&] { ANH_REQUIRE(every >= 1, "sampling interval must be >= 1"); Engine* e = engine_of(handle, is_trainer); e->synchronize(); e->prof.sample_every = every; e->prof.pass_index = 0; });
}
int anh_profile_reset(void* handle, int is_trainer) {
    return guarded([&] { Engine* e = engine_of(handle, is_trainer); e->synchronize(); e->prof.reset(); });
}
int anh_profile_count(void* handle, int is_trainer) {
    int n = -1;
    guarded([&] { Engine* e = engine_of(handle, is_trainer); e->synchronize(); n = (int)e->prof.entries.size(); });
    return n;
}
int anh_profile_entry(void* handle, int is_trainer, int index, char* name, size_t name_cap, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    return guarded([&] {
        Engine* e = engine_of(handle, is_trainer);
        ANH_REQUIRE(index >= 0 && index < (int)e->prof.entries.size(), "profile index out of range");
        const Profiler::Entry& en = e->prof.entries[index];
        if (name && name_cap) { std::strncpy(name, en.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
        if (total_ms) *total_ms = en.total_ms;
        if (launches) *launches = en.launches;
        if (flops) *flops = en.flops;
        if (bytes) *bytes = en.bytes;
    });
}
int anh_profile_launch_order(void* handle, int is_trainer, char* buf, size_t cap, size_t* needed) {
    return guarded([&] {
        Engine* e = engine_of(handle, is_trainer);
        std::string joined;
        for (const std::string& n : e->prof.order) { joined += n; joined += '\n'; }
        if (needed) *needed = joined.size() + 1;
        if (buf && cap) { std::strncpy(buf, joined.c_str(), cap - 1); buf[cap - 1] = 0; }
    });
}

// ---- host logic ----
int anh_get_tiles(int width, int height, const anh_tiling_params* params, anh_tile** tiles, size_t* count) {
    return guarded([&] {
        ANH_REQUIRE(tiles && count, "null argument");
        std::vector<anh_tile> t = tiles_for(params, width, height);
        anh_tile* out = (anh_tile*)std::malloc(std::max<size_t>(1, t.size()) * sizeof(anh_tile));
        if (!out) fail(ANH_ERR_OOM, "host allocation failed");
        std::memcpy(out, t.data(), t.size() * sizeof(anh_tile));
        *tiles = out; *count = t.size();
    });
}
int anh_set_weights(const uint16_t* labels, int nr, int nc, double cw, double iw, anh_wlabel* out) {
    return guarded([&] { ANH_REQUIRE(labels && out, "null argument"); set_weights(labels, nr, nc, cw, iw, out); });
}
int anh_random_rect_containing_point(uint32_t dx, uint32_t dy, long px, long py, long w, long hgt, anh_rect* out) {
    return guarded([&] { ANH_REQUIRE(out, "null argument"); *out = random_rect_containing_point(dx, dy, px, py, w, hgt); });
}
int anh_outpaint(uint8_t* image, int nr, int nc, int channels, const anh_rect* inside) {
    return guarded([&] { ANH_REQUIRE(image && inside, "null argument"); outpaint(image, nr, nc, channels, *inside); });
}
int anh_dataset_create(int channels, anh_dataset** out) {
    return guarded([&] {
        ANH_REQUIRE(out, "null argument");
        ANH_REQUIRE(channels == 1 || channels == 3, "input channels must be 1 or 3");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); fail(ANH_ERR_DEVICE, "no MI355X / HIP device visible"); }
        auto d = std::make_unique<anh_dataset>();
        d->channels = channels;
        HIP_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        *out = d.release();
    });
}
void anh_dataset_destroy(anh_dataset* d) { delete d; }
int anh_dataset_add(anh_dataset* d, const uint8_t* image_hwc, const uint16_t* labels, int height, int width, int* index) {
    return guarded([&] {
        ANH_REQUIRE(d && image_hwc && labels, "null argument");
        ANH_REQUIRE(height >= 1 && width >= 1 && (int64_t)height * width < ((int64_t)1 << 31), "bad image size");
        anh_dataset::Item it;
        const size_t px = (size_t)height * width;
        it.image.reserve(px * d->channels); it.labels.reserve(px * 2);
        it.height = height; it.width = width;
        HIP_CHECK(hipMemcpy(it.image.p, image_hwc, px * d->channels, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(it.labels.p, labels, px * 2, hipMemcpyHostToDevice));
        d->resident_bytes += px * d->channels + px * 2;
        int slot;
        if (!d->free_slots.empty()) { slot = d->free_slots.back(); d->free_slots.pop_back(); d->items[(size_t)slot] = std::move(it); }
        else { d->items.push_back(std::move(it)); slot = (int)d->items.size() - 1; }
        if (index) *index = slot;
    });
}
int anh_dataset_remove(anh_dataset* d, int index) {
    return guarded([&] {
        ANH_REQUIRE(d, "null dataset");
        ANH_REQUIRE(index >= 0 && (size_t)index < d->items.size() && d->items[(size_t)index].height > 0, "dataset remove: no such image");
        HIP_CHECK(hipStreamSynchronize(d->stream));   // the crop kernels that read the image have finished
        anh_dataset::Item& it = d->items[(size_t)index];
        d->resident_bytes -= (uint64_t)it.height * it.width * (d->channels + 2);
        it.image.release(); it.labels.release(); it.height = it.width = 0;
        d->free_slots.push_back(index);
    });
}
int anh_dataset_resident_bytes(const anh_dataset* d, uint64_t* bytes) {
    return guarded([&] { ANH_REQUIRE(d && bytes, "null argument"); *bytes = d->resident_bytes; });
}
int anh_dataset_crop_batch(anh_dataset* d, const anh_crop_spec* specs, int n, int dim, int classes, double class_weight, double image_weight,
                           uint8_t* images, anh_wlabel* labels) {
    return guarded([&] {
        ANH_REQUIRE(d && specs && images && labels && n >= 1 && dim >= 1, "crop batch: bad argument");
        const size_t plane = (size_t)dim * dim, img = (size_t)n * plane * d->channels;
        const size_t lab_off = (img + 255) / 256 * 256, w_off = (lab_off + (size_t)n * plane * 2 + 255) / 256 * 256;
        d->out.reserve(w_off + (size_t)n * plane * 4);
        uint8_t* base = d->out.as<uint8_t>();
        d->crop_batch(specs, n, dim, classes, class_weight, image_weight, base, reinterpret_cast<uint16_t*>(base + lab_off), reinterpret_cast<float*>(base + w_off));
        std::vector<uint16_t> hl((size_t)n * plane);
        std::vector<float> hw((size_t)n * plane);
        HIP_CHECK(hipMemcpyAsync(images, base, img, hipMemcpyDeviceToHost, d->stream));
        HIP_CHECK(hipMemcpyAsync(hl.data(), base + lab_off, hl.size() * 2, hipMemcpyDeviceToHost, d->stream));
        HIP_CHECK(hipMemcpyAsync(hw.data(), base + w_off, hw.size() * 4, hipMemcpyDeviceToHost, d->stream));
        HIP_CHECK(hipStreamSynchronize(d->stream));
        for (size_t i = 0; i < hl.size(); ++i) { labels[i].label = hl[i]; labels[i].weight = hw[i]; }
    });
}
int anh_trainer_step_crops(anh_trainer* h, anh_dataset* d, const anh_crop_spec* specs, int n, int dim, double class_weight, double image_weight) {
    return guarded([&] {
        ANH_REQUIRE(h && d && specs, "null argument");
        ANH_REQUIRE(n >= 1 && dim >= 1, "empty mini-batch");
        ANH_REQUIRE(h->replicas() == 1, "device-cut crops live on ONE device: a handle over several devices takes host mini-batches (anh_trainer_step)");
        Engine& e = h->engine();
        const int C = e.spec.cfg.in_channels, K = e.spec.cfg.classes;
        ANH_REQUIRE(C == d->channels, "the dataset's channel count differs from the net's");
        const size_t plane = (size_t)dim * dim, img_bytes = (size_t)n * plane * C, lab_off = (img_bytes + 255) / 256 * 256;
        const size_t w_off = (lab_off + (size_t)n * plane * 2 + 255) / 256 * 256, total = w_off + (size_t)n * plane * 4;
        anh_trainer::StageSet& st = h->stage[h->host_steps & 1];
        if (!st.uploaded) {
            HIP_CHECK(hipEventCreateWithFlags(&st.uploaded, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&st.consumed, hipEventDisableTiming));
        }
        if (total > st.dev.bytes) {
            if (st.in_flight) wait_event(st.consumed, h->replicas() > 1);
            st.dev.reserve(total);
        }
        // the crops of step k are cut on the dataset's stream while the trainer's stream still runs step k-1
        if (st.in_flight) HIP_CHECK(hipStreamWaitEvent(d->stream, st.consumed, 0));
        uint8_t* dbase = st.dev.as<uint8_t>();
        d->crop_batch(specs, n, dim, K, class_weight, image_weight, dbase, reinterpret_cast<uint16_t*>(dbase + lab_off), reinterpret_cast<float*>(dbase + w_off));
        HIP_CHECK(hipEventRecord(st.uploaded, d->stream));
        HIP_CHECK(hipStreamWaitEvent(e.stream, st.uploaded, 0));
        int rc = anh_trainer_forward_backward_device(h, dbase, reinterpret_cast<uint16_t*>(dbase + lab_off), reinterpret_cast<float*>(dbase + w_off), n, dim, dim, (double)n);
        if (rc != ANH_OK) fail(rc, g_error);
        rc = anh_trainer_apply_update(h, 1.0);
        if (rc != ANH_OK) fail(rc, g_error);
        HIP_CHECK(hipEventRecord(st.consumed, e.stream));
        st.in_flight = true;
        ++h->host_steps;
    });
}
int anh_ignore_large_nonzero_regions(uint16_t* labels, int nr, int nc, double by_area, double by_width, double by_height, int rf, int64_t* ignored) {
    return guarded([&] {
        ANH_REQUIRE(labels && nr > 0 && nc > 0 && rf > 0, "ignore_large_nonzero_regions: bad argument");
        ANH_REQUIRE((int64_t)nr * nc < (int64_t)1 << 31, "ignore_large_nonzero_regions: image too large");
        ANH_REQUIRE(!(by_area < 0) && !(by_width < 0) && !(by_height < 0), "ignore_large_nonzero_regions: negative threshold");
        const int64_t n = ignore_large_nonzero_regions(labels, nr, nc, by_area, by_width, by_height, rf);
        if (ignored) *ignored = n;
    });
}
static void* host_copy(const std::string& s) {
    void* p = std::malloc(s.size() ? s.size() : 1);
    if (!p) fail(ANH_ERR_OOM, "host allocation failed");
    std::memcpy(p, s.data(), s.size());
    return p;
}
int anh_dnn_envelope_pack(const char* classes_json, size_t json_size, double downscaling_factor, const void* net_blob, size_t net_size,
                          void** file, size_t* file_size) {
    return guarded([&] {
        ANH_REQUIRE(file && file_size && (classes_json || json_size == 0) && (net_blob || net_size == 0), "null argument");
        const std::string out = dnn_envelope_pack(std::string(classes_json ? classes_json : "", json_size), downscaling_factor,
                                                  std::string(net_blob ? (const char*)net_blob : "", net_size));
        *file = host_copy(out); *file_size = out.size();
    });
}
int anh_dnn_envelope_unpack(const void* file, size_t file_size, char** classes_json, size_t* json_size, double* downscaling_factor,
                            void** net_blob, size_t* net_size) {
    return guarded([&] {
        ANH_REQUIRE(file && classes_json && json_size && downscaling_factor && net_blob && net_size, "null argument");
        std::string json, net;
        double factor = 0;
        dnn_envelope_unpack(std::string((const char*)file, file_size), json, factor, net);
        void* j = host_copy(json);
        void* n = nullptr;
        try { n = host_copy(net); } catch (...) { std::free(j); throw; }
        *classes_json = (char*)j; *json_size = json.size(); *downscaling_factor = factor; *net_blob = n; *net_size = net.size();
    });
}
int64_t anh_count_steps_without_decrease(const double* values, int64_t n, double p) {
    if (!values && n > 0) return -1;
    return count_steps_without_decrease(values, n, p);
}

}  // extern "C"
