// w2x::Img2Img - the MI355X engine behind the reference's trt::Img2Img interface
// (/root/reference/src/tensorrt/img2img.h:14-50).  build(): ONNX -> plan file; load(): plan -> HBM; render(): frame ->
// tiles -> fused HIP kernels -> blended frame.  Everything on the device is HIP; there is no CPU fallback.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

#include "../../include/w2x/img2img.h"
#include "common.h"
#include "fragorder.h"
#include "kernels.h"
#include "lower.h"
#include "plan.h"
#include "sha256.h"
#include "switches.h"
#include "tiles.h"

#define W2X_LOG(sev, message) impl->log(sev, message, __FUNCTION__, __LINE__)
#define W2X_LOG_AS(who, sev, message) impl->log(sev, message, who, __LINE__)

namespace w2x {

namespace {

// helper.h:13-17 (cudaAssert): a failed runtime call becomes an exception carrying the runtime's message
inline void hipAssert(hipError_t e) {
    if (e != hipSuccess) throw std::runtime_error(hipGetErrorString(e));
}

// helper.h:27-46
std::string hipGetDeviceName(int deviceId) {
    hipDeviceProp_t prop{};
    hipAssert(hipGetDeviceProperties(&prop, deviceId));
    // The marketing name depends on which libdrm data file the process finds (empty or generic on some boxes, and different
    // between the ROCm runtime and the copy bundled with PyTorch); a plan is tied to the ISA and the CU count, so that is the key.
    std::string arch = prop.gcnArchName;
    return arch + " x" + std::to_string(prop.multiProcessorCount);
}

// Logical -> physical device ordinal.  W2X_DEVICE_MAP="0,0,1" makes logical devices 0 and 1 share physical device 0 (test hook:
// the multi-device code paths - one engine and one host thread per logical device - can then run on a single-GPU box).
int physical_device(int logical) {   // called by build() and load() only
    std::vector<int> map;
    if (const char* e = getenv("W2X_DEVICE_MAP")) { std::stringstream ss(e); std::string tok; while (std::getline(ss, tok, ',')) if (!tok.empty()) map.push_back(atoi(tok.c_str())); }
    return logical >= 0 && logical < (int)map.size() ? map[logical] : logical;
}

// One instance = one device (img2img_load.cpp:129 cudaSetDevice in load only).  The reference is driven from one thread; here
// several engines may live in one process, each on its own host thread, so every public entry that allocates, copies or launches
// makes the engine's device current for its own duration and restores the caller's on the way out.
struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(int d) : dev(d) {
        if (d < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != d) hipAssert(hipSetDevice(d));
    }
    ~DeviceGuard() { if (dev >= 0 && prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// "" when `physical` is a device of this process, else the sentence build() / load() log (a raw "invalid device ordinal" names neither the
// count nor the map that produced the ordinal)
std::string device_ordinal_problem(int logical, int physical) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
    if (count <= 0) return "no HIP device is visible to this process (hipGetDeviceCount = 0): this library has no CPU path";
    if (physical >= 0 && physical < count) return "";
    std::string m = "device id " + std::to_string(logical);
    if (physical != logical) m += " (W2X_DEVICE_MAP -> " + std::to_string(physical) + ")";
    return m + " does not exist: this process sees " + std::to_string(count) + " HIP device" + (count == 1 ? "" : "s") + " (valid ids 0.." + std::to_string(count - 1) + ")";
}

}  // namespace
bool device_pci_bus_id(int deviceId, char* buf, size_t cap) {
    const int dev = physical_device(deviceId);
    if (!device_ordinal_problem(deviceId, dev).empty()) return false;
    const bool ok = hipDeviceGetPCIBusId(buf, (int)cap, dev) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}
namespace {

std::string precision_name(Precision p) { return p == Precision::FP16 ? "FP16" : p == Precision::FP32 ? "FP32" : "TF32"; }

// img2img_build.cpp:8-27
std::string getConfigHash(const BuildConfig& c, std::string deviceName) {
    deviceName.erase(std::remove_if(deviceName.begin(), deviceName.end(), ::isspace), deviceName.end());
    std::ostringstream oss;
    oss << deviceName << "." << precision_name(c.precision) << "."
        << c.minBatchSize << "." << c.optBatchSize << "." << c.maxBatchSize << "."
        << c.minChannels << "." << c.optChannels << "." << c.maxChannels << "."
        << c.minWidth << "." << c.optWidth << "." << c.maxWidth << "."
        << c.minHeight << "." << c.optHeight << "." << c.maxHeight;
    std::string s = oss.str();
    return sha256_hex(s.data(), s.size());
}

// img2img_build.cpp:29-50 - same keys, same order, 4-space indent
void serializeConfig(const std::string& path, const BuildConfig& c, const std::string& deviceName) {
    std::ofstream f(path);
    if (!f.is_open()) throw std::runtime_error("could not open config \"" + path + "\"");
    auto esc = [](const std::string& s) { std::string o; for (char ch : s) { if (ch == '"' || ch == '\\') o.push_back('\\'); o.push_back(ch); } return o; };
    f << "{\n"
      << "    \"deviceName\": \"" << esc(deviceName) << "\",\n"
      << "    \"precision\": \"" << precision_name(c.precision) << "\",\n"
      << "    \"minBatchSize\": " << c.minBatchSize << ",\n    \"optBatchSize\": " << c.optBatchSize << ",\n    \"maxBatchSize\": " << c.maxBatchSize << ",\n"
      << "    \"minChannels\": " << c.minChannels << ",\n    \"optChannels\": " << c.optChannels << ",\n    \"maxChannels\": " << c.maxChannels << ",\n"
      << "    \"minWidth\": " << c.minWidth << ",\n    \"optWidth\": " << c.optWidth << ",\n    \"maxWidth\": " << c.maxWidth << ",\n"
      << "    \"minHeight\": " << c.minHeight << ",\n    \"optHeight\": " << c.optHeight << ",\n    \"maxHeight\": " << c.maxHeight << "\n}";
}

// img2img_load.cpp:54-77 (flat object of strings and integers)
void deserializeConfig(const std::string& path, BuildConfig& c, std::string& deviceName) {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("could not open config \"" + path + "\"");
    std::stringstream ss; ss << f.rdbuf();
    const std::string j = ss.str();
    auto find_value = [&](const std::string& key) -> size_t {
        size_t k = j.find("\"" + key + "\"");
        if (k == std::string::npos) throw std::runtime_error("config key \"" + key + "\" missing");
        size_t c2 = j.find(':', k);
        if (c2 == std::string::npos) throw std::runtime_error("config malformed");
        ++c2; while (c2 < j.size() && isspace((unsigned char)j[c2])) ++c2;
        return c2;
    };
    auto get_str = [&](const std::string& key) {
        size_t p = find_value(key);
        if (j[p] != '"') throw std::runtime_error("config value of \"" + key + "\" is not a string");
        std::string o; ++p;
        while (p < j.size() && j[p] != '"') { if (j[p] == '\\' && p + 1 < j.size()) ++p; o.push_back(j[p++]); }
        return o;
    };
    auto get_int = [&](const std::string& key) { return (int)std::strtol(j.c_str() + find_value(key), nullptr, 10); };
    deviceName = get_str("deviceName");
    const std::string prec = get_str("precision");
    c.precision = prec == "FP16" ? Precision::FP16 : prec == "FP32" ? Precision::FP32 : Precision::TF32;
    c.minBatchSize = get_int("minBatchSize"); c.optBatchSize = get_int("optBatchSize"); c.maxBatchSize = get_int("maxBatchSize");
    c.minChannels = get_int("minChannels"); c.optChannels = get_int("optChannels"); c.maxChannels = get_int("maxChannels");
    c.minWidth = get_int("minWidth"); c.optWidth = get_int("optWidth"); c.maxWidth = get_int("maxWidth");
    c.minHeight = get_int("minHeight"); c.optHeight = get_int("optHeight"); c.maxHeight = get_int("maxHeight");
}

// img2img_load.cpp:9-20; quirk Q12 fixed: the device is compared by name, not by the index of the first device with that name
bool isCompatible(const RenderConfig& r, const BuildConfig& b, const std::string& builtOn, const std::string& runningOn) {
    return builtOn == runningOn && r.precision == b.precision &&
           r.batchSize >= b.minBatchSize && r.batchSize <= b.maxBatchSize &&
           r.channels >= b.minChannels && r.channels <= b.maxChannels &&
           r.width >= b.minWidth && r.width <= b.maxWidth && r.height >= b.minHeight && r.height <= b.maxHeight;
}
// img2img_load.cpp:22-27
bool isOptimized(const RenderConfig& r, const BuildConfig& b) {
    return r.batchSize == b.optBatchSize && r.channels == b.optChannels && r.width == b.optWidth && r.height == b.optHeight;
}

constexpr const char* kEngineExt = ".w2x";

// Lowering of a model for one input shape, shared by build() (which writes the result to disk) and by load() when the render
// configuration lies inside an engine's [min, max] range but is not the shape that engine was specialised for.
// Super-batching: tiles are independent, so one network pass may carry several reference batches (S x batchSize tiles); results
// are bit-identical, launches per frame drop S-fold and the low-resolution stages fill all 256 CUs.  W2X_SUPERBATCH (switches.h) overrides;
// the default targets the pixel count of 48 tiles of 256x256 per pass (one 1080p frame at config 3), capped at 64 tiles.  Small
// tiles gain the most: at tile 64 / batch 1 a pass of one tile is ~40 launches of a few microseconds of work each (a 1080p frame:
// 1100 passes, 650 ms, hipGraph replay or not); 64 tiles per pass make it 18 passes.
Plan lower_for_shape(const std::string& onnxModelPath, int batch, int channels, int height, int width, bool fp32) {
    int S = 1;
    if (switches().superbatch > 0) S = switches().superbatch;
    else {
        const double want = 48.0 * 256 * 256 / ((double)batch * height * width);
        S = std::max(1, (int)std::lround(want));
        while (S > 1 && S * batch > 64) --S;
    }
    Plan plan;
    try {
        plan = build_plan(onnxModelPath, batch * S, channels, height, width, fp32);
    } catch (const std::exception&) {
        if (S == 1) throw;
        plan = build_plan(onnxModelPath, batch, channels, height, width, fp32);   // e.g. a graph with a static batch dimension
    }
    plan.userB = batch;
    return plan;
}

}  // namespace

struct Img2Img::Impl {
    MessageCallback messageCallback{};
    ProgressCallback progressCallback{};

    // logger.cpp:14-22
    void log(Severity s, const std::string& m) { if (messageCallback) messageCallback(s, m); }
    void log(Severity s, const std::string& m, const std::string& fn, int line) { if (messageCallback) messageCallback(s, "[" + fn + "@" + std::to_string(line) + "] " + m); }
    void log(int current, int total, double speed) { if (progressCallback) progressCallback(current, total, speed); }

    bool loaded = false;
    int device = -1;                   // physical HIP ordinal this engine lives on (set by load)
    // copies of the operational switches (switches.h), taken once per load(), never read from the environment on the launch path
    bool poison = false;               // W2X_POISON: stale activations become fp16 NaNs before every frame (tests)
    bool check_general = false;        // W2X_CHECK_GENERAL: every shape-specialised launch is compared with the general kernel
    bool use_graphs = true;            // W2X_NO_GRAPH switches the hipGraph replay of network passes off
    Plan plan;
    RenderConfig cfg;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // W2X_ROCTX=1 (read at load): roctxRangePushA / roctxRangePop around the launches of every plan op, resolved from libroctx64.so at run time (no link dependency)
    int (*roctx_push)(const char*) = nullptr; int (*roctx_pop)() = nullptr;
    hipEvent_t ev_shard = nullptr;       // renderSharded(): this engine's tiles are in its slab
    static constexpr int kMaxRenderParts = 4;
    hipEvent_t ev_part[kMaxRenderParts] = {nullptr, nullptr, nullptr, nullptr};
    int pipeline_parts = 3;              // W2X_RENDER_PARTS (read at load): parts a render() call runs a frame in (1 = one part; at most kMaxRenderParts)
    size_t shard_halo_slots = 0;         // renderSharded(): slab slots in front of this engine's own tiles (the bands copied from the preceding parts)
    int shard_rows = 0, shard_cols = 0;  // shardCompute(): the frame shardFinish() completes
    hipStream_t gstream[3] = {nullptr, nullptr, nullptr};   // further tile groups of a pass (run_frame)
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    int groups = 2;                      // W2X_GROUPS (1..4; W2X_NO_SPLIT = 1): tile groups a pass is cut into
    std::vector<void*> tensors, blobs;   // tensors point into one arena
    std::vector<void*> frag_blobs;       // per blob id: fragment-major copy of a weight matrix (or null)
    std::vector<void*> perm_blobs;       // per blob id: frag_conv3b copy of a 3x3 convolution's weights (or null)
    std::vector<int> pool_blocks;        // per tensor id: pooling partials per image written by the last producer (0: plan default)
    void* arena_base = nullptr; size_t arena_bytes = 0;
    std::vector<GemmParams> gemm;      // per op (kind == OP_GEMM)
    std::vector<int> pool_tensors;
    int final_op = -1;
    std::vector<int> tensor_last;      // last op that touches each tensor (upload_plan)
    std::vector<char> fuse_up;         // op i is a pixel-shuffle projection (cunet's ConvTranspose) whose map only feeds the 64 -> 64 3x3 convolution that follows: op i + 1's launch computes it in its halo stage (k_conv3.hip conv3_kernel UP), op i is skipped
    std::vector<char> fuse_stem;       // op i is a stem convolution whose 48- / 32-channel map only feeds the 3x3 convolution that follows: op i + 1's launch computes it in its halo stage (k_conv48.hip; k_conv3.hip for cunet), op i is skipped
    std::vector<char> fuse_head;       // op i is a C = 96 MLP whose rows only feed the image head that follows: one launch (k_mlp96q.hip), op i + 1 is skipped
    // fp32 plans: op i (LayerNorm + fc1 + GELU) and op i + 1 (fc2 + residual) are one mlp32_kernel launch when the engine runs Precision::TF32 (k_f32.hip); the hidden
    // map between them is neither written nor read.  mlp32_w[i] = the bf16 hi / lo planes of both matrices in fragment-major order (device memory, freed by release())
    std::vector<char> fuse_mlp32;
    std::vector<std::array<void*, 4>> mlp32_w;
    // ... and op i (LayerNorm + window gather + qkv), op i + 1 (attention core), op i + 2 (proj + window scatter + residual) are one swinattn32_kernel launch;
    // attn32_w[i] = the planes of Wqkv and Wproj
    std::vector<char> fuse_attn32;
    std::vector<std::array<void*, 4>> attn32_w;

    // frame-level buffers (grown on demand, reused across frames like the reference's input/output GpuMats, img2img.h:37-38)
    bool deep = false;                   // the frame in d_frame / d_out has 16-bit samples (Image::depth == 16)
    uint8_t* d_frame = nullptr; size_t frame_cap = 0;
    uint8_t* d_out = nullptr; size_t out_cap = 0;
    void* d_slab = nullptr; size_t slab_cap = 0;
    void* d_slab2 = nullptr; size_t slab2_cap = 0;      // second tile slab of a rolling sequence (run_rolling_frame)
    hipEvent_t ev_g0[2] = {nullptr, nullptr}, ev_cmp[2] = {nullptr, nullptr};   // rolling sequence: first group's passes of a frame issued / its compose done
    bool rolling = false;                               // inside run_rolling_frame: split passes do not join their streams
    bool rolling_ok = true;                             // W2X_NO_ROLLING switches the frame-to-frame pipeline of benchResident / renderSequence off
    TileSlot* d_slots = nullptr; size_t slots_cap = 0;
    float *d_rampx = nullptr, *d_rampy = nullptr;
    int ovx = 0, ovy = 0;
    float* d_blob_in = nullptr; float* d_blob_out = nullptr;
    // renderSequence(): second frame/output buffer and the copy streams
    uint8_t* d_frame2 = nullptr; size_t frame2_cap = 0;
    uint8_t* d_out2 = nullptr; size_t out2_cap = 0;
    hipStream_t s_up = nullptr, s_dn = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_dn[2] = {nullptr, nullptr};
    std::vector<void*> pinned;
    std::vector<void*> host_allocs;     // allocHost(): page-locked buffers handed to the caller
    std::vector<TileSlot> h_slots;

    // per-launch HIP-event profiling (profileFrame): events recorded on the compute stream around every launch
    bool profiling = false;
    struct Stamp { int kind; hipEvent_t a, b; double flops; int op; };
    std::vector<double> op_ms;   // per plan op, summed over the batches of the profiled frame
    int cur_op = -1;
    std::vector<Stamp> stamps;
    void stamp_begin(int kind, double flops) {
        if (!profiling) return;
        Stamp st{kind, nullptr, nullptr, flops, cur_op};
        hipAssert(hipEventCreate(&st.a)); hipAssert(hipEventCreate(&st.b));
        hipAssert(hipEventRecord(st.a, stream));
        stamps.push_back(st);
    }
    void stamp_end() { if (profiling) hipAssert(hipEventRecord(stamps.back().b, stream)); }

    // The copy streams of renderSequence() (upload, download) and of a two-part render() (download): ONE pair, created in one place and in one
    // order - the runtime spreads streams over its hardware queues in creation order, and a further stream in between moved the copy streams
    // onto the compute streams' queues (host-to-host 7.6 -> 9.8 ms per frame, profiles/r4_kernels/render_parts_ab.txt, first run).
    void ensure_copy_streams() {
        if (s_up) return;
        hipAssert(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
        hipAssert(hipStreamCreateWithFlags(&s_dn, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) { hipAssert(hipEventCreateWithFlags(&ev_up[b], hipEventDisableTiming)); hipAssert(hipEventCreateWithFlags(&ev_comp[b], hipEventDisableTiming)); hipAssert(hipEventCreateWithFlags(&ev_dn[b], hipEventDisableTiming)); }
    }

    // last frame (for benchResident / profileFrame, which replay it as ONE part)
    bool one_part_stale = false;          // the last render() ran in parts: d_slots holds the parts' slot tables
    void one_part_slots() {
        if (!one_part_stale) return;
        const int steps = cfg.tta ? 8 : 1, B = plan.B, S = plan.B / plan.userB;
        const int batchCount = (int)std::lround(std::ceil((double)(last_strip.tile_count * steps) / plan.userB));
        const size_t stepCount = (size_t)((batchCount + S - 1) / S) * B;
        h_slots.resize(stepCount);
        for (size_t st = 0; st < stepCount; ++st) {
            const int ti = (int)(st / steps), aug = (int)(st % steps);
            TileSlot sl{0, 0, aug, 0};
            if (ti < last_strip.tile_count) { sl.x = last_grid.in[last_strip.first_tile + ti].x; sl.y = last_grid.in[last_strip.first_tile + ti].y; sl.valid = 1; }
            h_slots[st] = sl;
        }
        ensure(d_slots, slots_cap, stepCount * sizeof(TileSlot));
        hipAssert(hipMemcpy(d_slots, h_slots.data(), stepCount * sizeof(TileSlot), hipMemcpyHostToDevice));
        ensure(d_slab, slab_cap, stepCount * plan.Tout * plan.Tout * 4 * plan.elt);      // (already that large: renderPart sizes the slab for both layouts)
        one_part_stale = false;
    }
    int last_rows = 0, last_cols = 0, last_batches = 0;
    TileGrid last_grid;
    StripPlan last_strip;
    float last_ms = 0.f;

    // One network pass (gather + every plan op) captured as a hipGraph: the reference's whole network is a single
    // enqueueV3 (img2img_infer.cpp:80); here a pass is ~40 launches, which the host cannot issue fast enough for small tiles.
    // A pass is captured the second time it is met (the first run stays eager so that one-time attribute calls are out of the
    // way) and replayed from then on.  The key holds everything the captured launches bake in.
    using GraphKey = std::tuple<const void*, const void*, const void*, const void*, int, int, int, int>;   // frame, slots, slab out, arena, rows, cols, live, 16-bit samples
    // A pass that runs as NG tile groups is NG graphs, one per group, each a straight line of launches replayed on that group's OWN stream (fork / join
    // events between the streams are issued around the replays): a single captured graph with NG branches runs its side branches on streams the runtime
    // creates at instantiation, which land on whichever hardware queue has the fewest users at that moment - sometimes the copy streams' queue, where the
    // next download or upload then waits behind a whole pass (render() in three parts 10.1 ms, in two or four 9.0 / 8.6, same work).
    struct PassGraphs { hipGraphExec_t g[4] = {nullptr, nullptr, nullptr, nullptr}; int n = 0; };
    std::map<GraphKey, PassGraphs> graphs;
    std::map<GraphKey, int> graph_seen;
    long graph_replays = 0, eager_passes = 0;
    void drop_graphs() {
        for (auto& kv : graphs) for (int k = 0; k < kv.second.n; ++k) if (kv.second.g[k]) (void)hipGraphExecDestroy(kv.second.g[k]);
        graphs.clear(); graph_seen.clear();
    }

    ~Impl() {
        release();
        for (void* h : host_allocs) if (hipHostFree(h) != hipSuccess) (void)hipGetLastError();
    }

    void release() {
        std::unique_ptr<DeviceGuard> guard;
        if (device >= 0) { try { guard.reset(new DeviceGuard(device)); } catch (...) {} }
        drop_graphs();
        // img2img_base.cpp:6-10 frees the IO buffers; here everything the engine owns
        if (arena_base) { (void)hipFree(arena_base); arena_base = nullptr; }
        for (void* p : blobs) if (p) (void)hipFree(p);
        for (void* p : frag_blobs) if (p) (void)hipFree(p);
        frag_blobs.clear();
        for (void* p : perm_blobs) if (p) (void)hipFree(p);
        perm_blobs.clear();
        for (auto& w : mlp32_w) for (void* p : w) if (p) (void)hipFree(p);
        mlp32_w.clear(); fuse_mlp32.clear();
        for (auto& w : attn32_w) for (void* p : w) if (p) (void)hipFree(p);
        attn32_w.clear(); fuse_attn32.clear();
        tensors.clear(); blobs.clear(); gemm.clear(); pool_tensors.clear();
        for (void* h : pinned) if (hipHostUnregister(h) != hipSuccess) (void)hipGetLastError();
        pinned.clear();
        // (allocHost() buffers belong to the caller's frames and outlive a re-load: they go in the destructor)
        for (hipEvent_t* e : {&ev_up[0], &ev_up[1], &ev_comp[0], &ev_comp[1], &ev_dn[0], &ev_dn[1]}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
        if (s_up) { (void)hipStreamDestroy(s_up); s_up = nullptr; }
        if (s_dn) { (void)hipStreamDestroy(s_dn); s_dn = nullptr; }
        frame2_cap = out2_cap = 0;
        for (void** p : {(void**)&d_frame, (void**)&d_out, (void**)&d_frame2, (void**)&d_out2, &d_slab, &d_slab2, (void**)&d_slots, (void**)&d_rampx, (void**)&d_rampy, (void**)&d_blob_in, (void**)&d_blob_out})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        frame_cap = out_cap = slab_cap = slab2_cap = slots_cap = 0;
        shard_rows = shard_cols = 0; shard_halo_slots = 0; rolling = false; one_part_stale = false;
        last_rows = last_cols = last_batches = 0;      // (benchResident / profileFrame replay the last frame of THIS load)
        if (ev0) { (void)hipEventDestroy(ev0); ev0 = nullptr; }
        if (ev1) { (void)hipEventDestroy(ev1); ev1 = nullptr; }
        if (ev_shard) { (void)hipEventDestroy(ev_shard); ev_shard = nullptr; }
        for (hipEvent_t& e : ev_part) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (hipEvent_t& e : ev_g0) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (hipEvent_t& e : ev_cmp) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (ev_fork) { (void)hipEventDestroy(ev_fork); ev_fork = nullptr; }
        for (hipEvent_t& e : ev_join) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (hipStream_t& st : gstream) if (st) { (void)hipStreamDestroy(st); st = nullptr; }
        if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
        loaded = false;
    }

    TView tview(const View& v) const {
        TView t;
        if (v.t < 0) return t;
        const TensorDesc& d = plan.tensors[v.t];
        t.p = tensors[v.t]; t.Hs = d.H; t.Ws = d.W; t.Cs = d.C; t.y0 = v.y0; t.x0 = v.x0;
        return t;
    }

    // fp32 plans, Precision::TF32: which runs of un-fused ops one fused launch of k_f32.hip serves.  Facts of the plan alone (lower.cpp's fuse_mlp() / fuse_attn() state
    // the same patterns for the fp16 plan); the plan itself stays un-fused - one lowering serves TF32 and FP32, and Precision::FP32 runs every launch.
    bool whole_view(const View& v) const { if (v.t < 0) return false; const TensorDesc& t = plan.tensors[v.t]; return v.y0 == 0 && v.x0 == 0 && v.H == t.H && v.W == t.W; }
    int readers_of(int t) const {
        int n = 0;
        for (const Op& o : plan.ops) {
            if (o.kind == OP_GEMM && (o.g.a.t == t || o.g.res.t == t || o.g.res2.t == t)) ++n;
            if (o.kind == OP_ATTN && o.at.qkv == t) ++n;
        }
        return n;
    }
    bool mlp32_pair(size_t i) const {
        if (plan.elt != 4 || switches().no_fuse || i + 1 >= plan.ops.size() || plan.ops[i].kind != OP_GEMM || plan.ops[i + 1].kind != OP_GEMM) return false;
        const GemmOp& g1 = plan.ops[i].g; const GemmOp& g2 = plan.ops[i + 1].g;
        const int Cm = g1.K;
        return mlp32_supported(Cm) && g1.amode == A_ROWS && g1.ln && g1.stats_in >= 0 && g1.act == ACT_GELU && g1.omode == O_ROWS && g1.res.t < 0 && g1.res2.t < 0 && !g1.has_clip &&
               g1.N == 2 * Cm && g1.stats_out < 0 && g1.pool_out < 0 && g1.se_scale < 0 && whole_view(g1.a) && whole_view(g1.out) && plan.tensors[g1.a.t].C == Cm && plan.tensors[g1.out.t].C == 2 * Cm &&
               g1.Mrows == plan.tensors[g1.a.t].H * plan.tensors[g1.a.t].W &&
               g2.amode == A_ROWS && !g2.ln && g2.act == ACT_NONE && g2.omode == O_ROWS && g2.a.t == g1.out.t && g2.K == 2 * Cm && g2.N == Cm && g2.res.t == g1.a.t && g2.res2.t < 0 &&
               !g2.has_clip && g2.pool_out < 0 && g2.se_scale < 0 && g2.res_scale < 0 && whole_view(g2.a) && whole_view(g2.res) && whole_view(g2.out) && plan.tensors[g2.out.t].C == Cm && g2.Mrows == g1.Mrows &&
               g1.out.t != plan.out_tensor && readers_of(g1.out.t) == 1 &&
               plan.blobs[g1.w].data.size() == (size_t)2 * Cm * Cm * 4 && plan.blobs[g2.w].data.size() == (size_t)2 * Cm * Cm * 4;      // fp32 [2C][C] and [C][2C], rows unpadded
    }
    bool attn32_triple(size_t i) const {
        if (plan.elt != 4 || switches().no_fuse || switches().no_fuse_attn || i + 2 >= plan.ops.size() || plan.ops[i].kind != OP_GEMM || plan.ops[i + 1].kind != OP_ATTN || plan.ops[i + 2].kind != OP_GEMM) return false;
        const GemmOp& g1 = plan.ops[i].g; const AttnOp& a = plan.ops[i + 1].at; const GemmOp& g2 = plan.ops[i + 2].g;
        const int Cm = g1.K;
        return swinattn32_supported(Cm, a.heads, a.hd, a.ws * a.ws) && a.heads * a.hd == Cm &&
               g1.amode == A_WIN && g1.win_table >= 0 && g1.ln && g1.stats_in >= 0 && g1.act == ACT_NONE && g1.omode == O_ROWS && g1.res.t < 0 && g1.res2.t < 0 && !g1.has_clip && g1.N == 3 * Cm &&
               g1.stats_out < 0 && g1.pool_out < 0 && g1.se_scale < 0 && whole_view(g1.a) && whole_view(g1.out) && plan.tensors[g1.a.t].C == Cm && plan.tensors[g1.out.t].C == 3 * Cm &&
               g1.Mrows == a.nwin * a.ws * a.ws && g1.Mrows == plan.tensors[g1.a.t].H * plan.tensors[g1.a.t].W &&
               a.qkv == g1.out.t && a.bias >= 0 && a.maskid >= 0 &&
               g2.amode == A_ROWS && g2.a.t == a.out && !g2.ln && g2.act == ACT_NONE && g2.omode == O_WIN && g2.win_table >= 0 && g2.K == Cm && g2.N == Cm && g2.res.t == g1.a.t && g2.res2.t < 0 &&
               !g2.has_clip && g2.pool_out < 0 && g2.se_scale < 0 && g2.res_scale < 0 && whole_view(g2.a) && whole_view(g2.res) && whole_view(g2.out) && plan.tensors[g2.out.t].C == Cm &&
               plan.tensors[g2.out.t].H * plan.tensors[g2.out.t].W == g1.Mrows && g2.Mrows == g1.Mrows &&
               g1.out.t != plan.out_tensor && a.out != plan.out_tensor && readers_of(g1.out.t) == 1 && readers_of(a.out) == 1 &&
               plan.blobs[g1.w].data.size() == (size_t)3 * Cm * Cm * 4 && plan.blobs[g2.w].data.size() == (size_t)Cm * Cm * 4;
    }
    // the bf16 hi / lo planes of an fp32 matrix [N][K] in fragment-major order (split4 of k_f32.hip on the host: hi = bf16(x), lo = bf16(x - hi), round to nearest even)
    void upload_planes(const std::vector<uint8_t>& w, int N, int K, void*& dh, void*& dl) {
        std::vector<uint16_t> hi((size_t)N * K), lo((size_t)N * K);
        const float* f = (const float*)w.data();
        auto bf = [](float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); };
        for (size_t k = 0; k < hi.size(); ++k) {
            const uint16_t h = bf(f[k]);
            const uint32_t hu = (uint32_t)h << 16; float hf; memcpy(&hf, &hu, 4);
            hi[k] = h; lo[k] = bf(f[k] - hf);
        }
        for (int pl = 0; pl < 2; ++pl) {
            const std::vector<uint16_t> fr = frag_major(pl ? lo.data() : hi.data(), N, K);
            void*& d = pl ? dl : dh;
            hipAssert(hipMalloc(&d, fr.size() * 2 + 256));
            hipAssert(hipMemcpy(d, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
        }
    }

    void upload_plan() {
        // Activation arena: tensors whose lifetimes (first writer .. last reader, in op order) do not overlap share
        // memory.  An op's outputs are placed before its inputs are released, so no op reads and writes one address.
        {
            const int nt = (int)plan.tensors.size(), nops = (int)plan.ops.size();
            std::vector<int> first(nt, nops), last(nt, -1);
            auto touch = [&](int t, int op) { if (t < 0) return; first[t] = std::min(first[t], op); last[t] = std::max(last[t], op); };
            for (int i = 0; i < nops; ++i) {
                const Op& op = plan.ops[i];
                switch (op.kind) {
                    case OP_GEMM: touch(op.g.a.t, i); touch(op.g.res.t, i); touch(op.g.res2.t, i); touch(op.g.stats_in, i); touch(op.g.se_scale, i); touch(op.g.res_scale, i);
                                  touch(op.g.out.t, i); touch(op.g.stats_out, i); touch(op.g.pool_out, i); break;
                    case OP_ATTN: touch(op.at.qkv, i); touch(op.at.out, i); break;
                    case OP_SE: touch(op.se.pool, i); touch(op.se.scale, i); break;
                    case OP_SCALE_ADD: touch(op.se.pool, i); touch(op.se.scale, i); break;
                    case OP_MLP: touch(op.m.x, i); touch(op.m.y, i); touch(op.m.stats_out, i); break;
                    case OP_SWINATTN: touch(op.sa.x, i); touch(op.sa.y, i); touch(op.sa.stats_out, i); break;
                    default: break;
                }
            }
            first[plan.in_tensor] = -1;                       // written by the gather kernel before op 0
            last[plan.out_tensor] = nops;                     // read after the last op (infer) / replaced by the frame slab
            // An image head that rides on the MLP launch in front of it (fuse_head, decided below) writes its output WHILE that MLP still reads its input:
            // the output must be alive from the MLP on, or it would be given the memory of the MLP's input, which dies at the MLP in the un-fused order.
            // fp32 plans: a fused launch (mlp32_kernel / swinattn32_kernel) writes the LAST op's outputs while it still reads the first op's inputs - the row statistics among
            // them, which die at the first op in the un-fused order: the outputs are alive from the first op on
            fuse_mlp32.assign(nops, 0); fuse_attn32.assign(nops, 0);
            for (int i = 0; i < nops; ++i) {
                const int span = attn32_triple((size_t)i) ? 2 : mlp32_pair((size_t)i) ? 1 : 0;
                if (!span) continue;
                (span == 2 ? fuse_attn32 : fuse_mlp32)[i] = 1;
                const GemmOp& gl = plan.ops[i + span].g;
                first[gl.out.t] = std::min(first[gl.out.t], i);
                if (gl.stats_out >= 0) first[gl.stats_out] = std::min(first[gl.stats_out], i);
                i += span;
            }
            fuse_head.assign(nops, 0);
            if (plan.elt == 2 && !switches().no_fuse_head && mlp_frag32(96))      // (the head rides on k_mlp96q.hip's launch only)
                for (int i = 0; i + 1 < nops; ++i) {
                    const Op& a = plan.ops[i]; const Op& b = plan.ops[i + 1];
                    if (a.kind == OP_MLP && b.kind == OP_GEMM && a.m.C == 96 && a.m.stats_out < 0 && b.g.a.t == a.m.y && last[a.m.y] == i + 1 && b.g.amode == A_ROWS && b.g.K == 96 && b.g.N == 64 &&
                        b.g.r == 4 && b.g.omode == O_PIXSHUF && b.g.res.t < 0 && b.g.res2.t < 0 && b.g.act == ACT_NONE && !b.g.ln && b.g.se_scale < 0 && b.g.res_scale < 0 && b.g.stats_out < 0 && b.g.pool_out < 0) {
                        fuse_head[i] = 1;
                        first[b.g.out.t] = std::min(first[b.g.out.t], i);
                    }
                }
            // The stem folded into the patch convolution behind it (fuse_stem, confirmed below with the prepared parameters): that launch reads the stem's INPUT,
            // which must therefore outlive the stem by one op (its memory would otherwise go to the convolution's output).
            fuse_stem.assign(nops, 0);
            if (plan.elt == 2)
                for (int i = 0; i + 1 < nops; ++i) {
                    const Op& a = plan.ops[i]; const Op& b = plan.ops[i + 1];
                    if (a.kind == OP_GEMM && b.kind == OP_GEMM && a.g.a.t >= 0 && plan.tensors[a.g.a.t].C == 4 && ((a.g.N == 48 && b.g.K == 9 * 48) || (a.g.N == 32 && b.g.K == 9 * 32 && b.g.N == 64 && b.g.pool_out < 0)) && b.g.a.t == a.g.out.t && first[a.g.out.t] == i && last[a.g.out.t] == i + 1 &&
                        a.g.out.t != plan.out_tensor && a.g.stats_out < 0 && a.g.pool_out < 0) {
                        fuse_stem[i] = 1;
                        last[a.g.a.t] = std::max(last[a.g.a.t], i + 1);
                    }
                }
            // cunet's transposed convolution folded into the 3x3 convolution behind it (fuse_up, confirmed below with the prepared parameters): that launch reads the
            // projection's rows, its skip map and its gate, which must therefore outlive the projection by one op.
            fuse_up.assign(nops, 0);
            if (plan.elt == 2)
                for (int i = 0; i + 1 < nops; ++i) {
                    const Op& a = plan.ops[i]; const Op& b = plan.ops[i + 1];
                    if (a.kind == OP_GEMM && b.kind == OP_GEMM && a.g.omode == O_PIXSHUF && a.g.r == 2 && a.g.K == 64 && a.g.N == 256 && a.g.res.t >= 0 && a.g.res2.t < 0 && a.g.res_scale < 0 && !a.g.ln &&
                        a.g.stats_out < 0 && a.g.pool_out < 0 && b.g.amode == A_CONV && b.g.kh == 3 && b.g.kw == 3 && b.g.stride == 1 && b.g.K == 9 * 64 && b.g.N == 64 && b.g.pool_out < 0 && b.g.a.t == a.g.out.t &&
                        first[a.g.out.t] == i && last[a.g.out.t] == i + 1 && a.g.out.t != plan.out_tensor) {
                        fuse_up[i] = 1;
                        for (int t : {a.g.a.t, a.g.res.t, a.g.se_scale})
                            if (t >= 0) last[t] = std::max(last[t], i + 1);
                    }
                }
            tensor_last = last;
            struct Block { size_t off, size; };
            std::vector<Block> free_list;
            std::vector<size_t> off(nt, 0);
            size_t arena = 0;
            // 256 bytes x 12: offsets stay 256-byte aligned when the arena is cut into 2, 3 or 4 group parts (run_network).
            constexpr size_t unit = 3072;
            auto align = [](size_t v) { return (v + unit - 1) / unit * unit; };
            auto alloc = [&](size_t bytes) -> size_t {
                bytes = align(bytes);
                int best = -1;
                for (int k = 0; k < (int)free_list.size(); ++k) if (free_list[k].size >= bytes && (best < 0 || free_list[k].size < free_list[best].size)) best = k;
                if (best >= 0) { size_t o = free_list[best].off; free_list[best].off += bytes; free_list[best].size -= bytes; if (!free_list[best].size) free_list.erase(free_list.begin() + best); return o; }
                if (!free_list.empty()) {   // grow the last free block if it touches the end of the arena
                    for (int k = 0; k < (int)free_list.size(); ++k) if (free_list[k].off + free_list[k].size == arena) { size_t o = free_list[k].off; arena = o + bytes; free_list.erase(free_list.begin() + k); return o; }
                }
                size_t o = arena; arena += bytes; return o;
            };
            auto release = [&](size_t o, size_t bytes) {
                bytes = align(bytes);
                free_list.push_back({o, bytes});
                std::sort(free_list.begin(), free_list.end(), [](const Block& a, const Block& b) { return a.off < b.off; });
                for (size_t k = 0; k + 1 < free_list.size();) { if (free_list[k].off + free_list[k].size == free_list[k + 1].off) { free_list[k].size += free_list[k + 1].size; free_list.erase(free_list.begin() + k + 1); } else ++k; }
            };
            std::vector<char> placed(nt, 0);
            for (int step = -1; step <= nops; ++step) {
                for (int t = 0; t < nt; ++t) if (!placed[t] && first[t] == step && last[t] >= 0) { off[t] = alloc((size_t)plan.tensors[t].bytes()); placed[t] = 1; }
                for (int t = 0; t < nt; ++t) if (placed[t] == 1 && last[t] == step) { release(off[t], (size_t)plan.tensors[t].bytes()); placed[t] = 2; }
            }
            // (tensors no op touches - the q / k / v, score and hidden maps inside the fused attention and MLP ops - get no memory: at config 3 they
            //  were 19 of the arena's 20.7 GiB until round 3)
            hipAssert(hipMalloc(&arena_base, arena + 1024));   // slack: vector reads past a table's last row; the group parts rounded up to 256 bytes
            hipAssert(hipMemsetAsync(arena_base, 0, arena + 1024, stream));
            arena_bytes = arena;
            pool_blocks.assign(nt, 0);
        tensors.assign(nt, nullptr);
            for (int t = 0; t < nt; ++t) tensors[t] = placed[t] ? (uint8_t*)arena_base + off[t] : nullptr;
        }
        blobs.assign(plan.blobs.size(), nullptr);
        for (size_t i = 0; i < plan.blobs.size(); ++i) {
            const auto& d = plan.blobs[i].data;
            hipAssert(hipMalloc(&blobs[i], d.size() + 256));   // slack: kernels may read a vector past a table's last row
            hipAssert(hipMemcpy(blobs[i], d.data(), d.size(), hipMemcpyHostToDevice));
        }
        // fragment-major copies (fragorder.h) of the weights that kernels read straight from L2
        frag_blobs.assign(plan.blobs.size(), nullptr);
        auto upload_frag = [&](int blob, const std::vector<uint16_t>& f) {
            hipAssert(hipMalloc(&frag_blobs[blob], f.size() * 2 + 256));
            hipAssert(hipMemcpy(frag_blobs[blob], f.data(), f.size() * 2, hipMemcpyHostToDevice));
        };
        auto frag_major_blob = [&](int blob, int N, int K) {
            if (frag_blobs[blob]) return;
            const auto& d = plan.blobs[blob].data;
            if (d.size() != (size_t)N * K * 2) throw std::runtime_error("plan: weight shape");
            upload_frag(blob, frag_major((const uint16_t*)d.data(), N, K));
        };
        auto frag_w2_blob = [&](int blob, int Cc) {
            if (frag_blobs[blob]) return;
            const auto& d = plan.blobs[blob].data;
            if (d.size() != (size_t)Cc * Cc * 4) throw std::runtime_error("plan: mlp weight shape");
            upload_frag(blob, frag_w2((const uint16_t*)d.data(), Cc));
        };
        for (const Op& op : plan.ops)   // pixel-shuffle projections served by k_pixgemm.hip
            if (plan.elt == 2 && op.kind == OP_GEMM && ((op.g.omode == 2 && (op.g.amode == 0 || (op.g.amode == 2 && op.g.kh == 1 && op.g.kw == 1))) || (op.g.amode == 2 && op.g.kh == 2 && op.g.kw == 2 && op.g.stride == 2)) && op.g.K % 32 == 0 && op.g.N % 16 == 0 &&
                plan.blobs[op.g.w].data.size() == (size_t)op.g.N * op.g.K * 2) frag_major_blob(op.g.w, op.g.N, op.g.K);
        perm_blobs.assign(plan.blobs.size(), nullptr);
        for (const Op& op : plan.ops)   // plain 3x3 convolutions onto 64 / 128 / 256 channels (k_conv3.hip)
            if (plan.elt == 2 && op.kind == OP_GEMM && op.g.amode == 2 && op.g.kh == 3 && op.g.kw == 3 && op.g.stride == 1 && op.g.omode == 0 && op.g.K % 32 == 0 && op.g.N % 64 == 0 &&
                plan.blobs[op.g.w].data.size() == (size_t)op.g.N * op.g.K * 2 && !perm_blobs[op.g.w]) {
                const std::vector<uint16_t> f = frag_conv3b((const uint16_t*)plan.blobs[op.g.w].data.data(), op.g.N, op.g.K);
                hipAssert(hipMalloc(&perm_blobs[op.g.w], f.size() * 2 + 256));
                hipAssert(hipMemcpy(perm_blobs[op.g.w], f.data(), f.size() * 2, hipMemcpyHostToDevice));
            }
        for (const Op& op : plan.ops)   // 48 -> 96 channel 3x3 convolution (k_conv48.hip): K = 432 padded with zero columns to 14 k-steps of 32
            if (plan.elt == 2 && op.kind == OP_GEMM && op.g.amode == 2 && op.g.kh == 3 && op.g.kw == 3 && op.g.stride == 1 && op.g.omode == 0 && op.g.K == 432 && op.g.N == 96 &&
                plan.tensors[op.g.a.t].C == 48 && !frag_blobs[op.g.w]) {
                const auto& d = plan.blobs[op.g.w].data;
                const int Kw = round_up(op.g.K, 8), Kp = 448;
                if (d.size() != (size_t)op.g.N * Kw * 2) throw std::runtime_error("plan: weight shape");
                std::vector<uint16_t> padded((size_t)op.g.N * Kp, 0);
                for (int n = 0; n < op.g.N; ++n) memcpy(&padded[(size_t)n * Kp], d.data() + (size_t)n * Kw * 2, (size_t)op.g.K * 2);
                upload_frag(op.g.w, frag_major(padded.data(), op.g.N, Kp));
            }
        for (const Op& op : plan.ops)
            if (plan.elt == 2 && op.kind == OP_MLP && mlp_supported(op.m.C)) {
                const int Cm = op.m.C;
                const auto& w1 = plan.blobs[op.m.w1].data; const auto& w2 = plan.blobs[op.m.w2].data;
                if (w1.size() != (size_t)2 * Cm * Cm * 2 || w2.size() != w1.size()) throw std::runtime_error("plan: MLP weight size");
                // C = 96: 32x32x16 fragments (k_mlp96q.hip).  C = 192: 16x16x32 fragments (mlp2_kernel<192,2,4> in k_mlp2.hip) since round 6 - the 32x32x16 kernel of
                // rounds 3-5 (mlp2q_kernel, -DW2X_MLP192_TILE32 here) takes the same time per launch but more energy per product, and the frame runs at the
                // board's power cap: 7.272 / 7.274 against 7.306 / 7.312 / 7.307 ms per frame in alternating pairs (profiles/r6_kernels/lib_mlp192_tile16_frame_level.txt)
                if (!mlp_frag32(Cm)) {
                    if (!frag_blobs[op.m.w1]) upload_frag(op.m.w1, frag_major((const uint16_t*)w1.data(), 2 * Cm, Cm));
                    if (!frag_blobs[op.m.w2]) upload_frag(op.m.w2, frag_w2((const uint16_t*)w2.data(), Cm));
                    continue;
                }
                if (!frag_blobs[op.m.w1]) upload_frag(op.m.w1, frag32_major((const uint16_t*)w1.data(), 2 * Cm, Cm));
                if (!frag_blobs[op.m.w2]) upload_frag(op.m.w2, frag32_w2((const uint16_t*)w2.data(), Cm));
            }
        for (const Op& op : plan.ops)
            if (plan.elt == 2 && op.kind == OP_SWINATTN) { frag_major_blob(op.sa.wqkv, 3 * op.sa.C, op.sa.C); frag_major_blob(op.sa.wproj, op.sa.C, op.sa.C); }
        gemm.assign(plan.ops.size(), GemmParams{});
        for (size_t i = 0; i < plan.ops.size(); ++i) {
            const Op& op = plan.ops[i];
            if (op.kind != OP_GEMM) continue;
            const GemmOp& g = op.g;
            GemmParams& p = gemm[i];
            p.a = tview(g.a); p.amode = g.amode; p.kh = g.kh; p.kw = g.kw; p.stride = g.stride;
            p.B = plan.B; p.Mrows = g.Mrows; p.aW = g.aW;
            p.win_table = g.win_table >= 0 ? (const int*)blobs[g.win_table] : nullptr;
            p.K = g.K; p.N = g.N; p.Kw = round_up(g.K, 8);
            p.wt = blobs[g.w]; p.wt_frag = frag_blobs[g.w]; p.wt_perm = perm_blobs[g.w]; p.bias = (const float*)blobs[g.bias];
            p.ln = g.ln; p.csum = g.csum >= 0 ? (const float*)blobs[g.csum] : nullptr;
            p.stats_in = g.stats_in >= 0 ? (const float*)tensors[g.stats_in] : nullptr;
            p.act = g.act; p.alpha = g.alpha; p.has_clip = g.has_clip; p.clip_lo = g.clip_lo; p.clip_hi = g.clip_hi;
            p.res = tview(g.res); p.res2 = tview(g.res2); p.out = tview(g.out);
            p.omode = g.omode; p.r = g.r; p.Cout = g.Cout;
            p.stats_out = g.stats_out >= 0 ? (float*)tensors[g.stats_out] : nullptr; p.ln_eps = g.ln_eps;
            p.pool_out = g.pool_out >= 0 ? (float*)tensors[g.pool_out] : nullptr;
            p.a_scale = g.se_scale >= 0 ? (const float*)tensors[g.se_scale] : nullptr;
            p.res_scale = g.res_scale >= 0 ? (const float*)tensors[g.res_scale] : nullptr;
            if ((p.a_scale || p.res_scale) && plan.elt != 2) throw std::runtime_error("plan: folded gates in an fp32 plan");
            if (g.pool_out >= 0) pool_tensors.push_back(g.pool_out);
            if (g.out.t == plan.out_tensor) final_op = (int)i;
            // shape checks the kernels rely on (a wrong shape would fault on the device)
            if (p.ln && (!p.stats_in || !p.csum)) throw std::runtime_error("plan: LayerNorm op without statistics");
            if (p.stats_out && p.Cout != p.out.Cs) throw std::runtime_error("plan: statistics over padded rows");
            if (p.pool_out && p.omode != O_ROWS) throw std::runtime_error("plan: pooling on a scattered output");
            if (p.omode == O_PIXSHUF && p.N != p.r * p.r * p.out.Cs) throw std::runtime_error("plan: pixel-shuffle width mismatch");
            if (p.omode != O_PIXSHUF && p.N != p.out.Cs) throw std::runtime_error("plan: output width mismatch");
            if ((p.amode == A_WIN || p.omode == O_WIN) && !p.win_table) throw std::runtime_error("plan: missing window table");
            if (p.a.Cs != 4 && (p.a.Cs % 8)) throw std::runtime_error("plan: unaligned input channels");
        }
        if (final_op < 0) throw std::runtime_error("plan: the output tensor is not produced by a fused op");
        // fp32 plans: the weights of the fused launches decided above (used only while cfg.precision == TF32)
        mlp32_w.assign(plan.ops.size(), std::array<void*, 4>{{nullptr, nullptr, nullptr, nullptr}});
        attn32_w.assign(plan.ops.size(), std::array<void*, 4>{{nullptr, nullptr, nullptr, nullptr}});
        for (size_t i = 0; i < plan.ops.size(); ++i) {
            if (fuse_mlp32[i]) {
                const GemmOp& g1 = plan.ops[i].g; const GemmOp& g2 = plan.ops[i + 1].g;
                upload_planes(plan.blobs[g1.w].data, 2 * g1.K, g1.K, mlp32_w[i][0], mlp32_w[i][1]);
                upload_planes(plan.blobs[g2.w].data, g1.K, 2 * g1.K, mlp32_w[i][2], mlp32_w[i][3]);
            }
            if (fuse_attn32[i]) {
                const GemmOp& g1 = plan.ops[i].g; const GemmOp& g2 = plan.ops[i + 2].g;
                upload_planes(plan.blobs[g1.w].data, 3 * g1.K, g1.K, attn32_w[i][0], attn32_w[i][1]);
                upload_planes(plan.blobs[g2.w].data, g1.K, g1.K, attn32_w[i][2], attn32_w[i][3]);
            }
        }
        // The image head (Linear 96 -> 4x4 sub-pixels x 4 channels, Clip) behind the last MLP: its input rows have no other reader, so the MLP launch
        // runs the head on every tile it produces and neither stores nor re-reads the 96-channel map (switches.h no_fuse_head keeps the two launches).
        for (size_t i = 0; i + 1 < plan.ops.size(); ++i)       // the candidates of the arena pass above, now with the prepared launch parameters
            if (fuse_head[i]) {
                const GemmParams& g = gemm[i + 1];
                if (!(g.wt_frag && g.out.Cs == 4 && g.a.y0 == 0 && g.a.x0 == 0 && g.a.Ws == g.aW && (long)g.a.Hs * g.a.Ws == g.Mrows && g.Mrows % 32 == 0 && g.aW >= 32 && pixgemm_supported(g) &&
                      (!g.has_clip || (f16_to_f32(f32_to_f16(g.clip_lo)) == g.clip_lo && f16_to_f32(f32_to_f16(g.clip_hi)) == g.clip_hi)))) fuse_head[i] = 0;   // (k_mlp96q.hip: a 32-row tile inside one image, at most two token rows; clip bounds that fp16 holds exactly)
            }
        // The stem (3x3, 4 -> 48 channels) in front of the patch convolution (3x3, 48 -> 96): its output has no other reader, so the convolution computes the halo tile
        // it needs from the input tile and the 48-channel map is neither stored nor read (k_conv48.hip conv48_kernel<true>; switches.h no_fuse_stem keeps the two launches).
        // Round 6: the same for cunet's two U-Nets, whose stems (4 -> 32) feed a 32 -> 64 convolution each (k_conv3.hip conv3_kernel<false, true>).
        for (size_t i = 0; i + 1 < plan.ops.size(); ++i)
            if (fuse_stem[i] && !conv48_stem_supported(gemm[i + 1], gemm[i]) && !conv3_stem_supported(gemm[i + 1], gemm[i])) fuse_stem[i] = 0;
        // Round 6: cunet's ConvTranspose 2x2 stride 2 (a pixel-shuffle projection with LeakyReLU, a gate on its rows and a skip add) in front of a 64 -> 64 convolution
        // that alone reads it: the convolution assembles its halo tile from the skip map and the projection's input rows (k_conv3.hip conv3_kernel UP; switches.h no_fuse_up).
        for (size_t i = 0; i + 1 < plan.ops.size(); ++i)
            if (fuse_up[i] && !conv3_up_supported(gemm[i + 1], gemm[i])) fuse_up[i] = 0;
        hipAssert(hipStreamSynchronize(stream));
    }

    // one pass of the network over the tiles currently in plan.in_tensor; the last op writes to `out_override`
    // `live` = tiles of this pass that carry image data; the zero-pad slots the reference appends to fill its last batch
    // (img2img_render.cpp:281) are never read back (:298-299), so they are not computed at all.
    // `grp` / `s`: the pass may be cut into NG tile groups that run side by side on NG streams (run_frame).  Tensors with disjoint
    // lifetimes share arena memory, which only holds while the ops run one after the other - so each group gets its own PART of the
    // arena, laid out like the whole one at 1/NG of the size (a tensor of B tiles at offset o becomes B/NG tiles at o/NG; offsets are
    // multiples of 3072 bytes): group grp of up to B/NG tiles runs the plan on arena_base + grp * arena_part.  out_override is the
    // slab address of the group's first tile.
    int ng_now = 1;                      // NG of the pass being issued
    size_t arena_part() const { return ((arena_bytes / ng_now) + 255) / 256 * 256; }
    uint8_t* group_ptr(const void* full, int grp) const {
        return full ? (uint8_t*)arena_base + (size_t)grp * arena_part() + (size_t)((const uint8_t*)full - (const uint8_t*)arena_base) / ng_now : nullptr;
    }
    void run_network(void* out_override, int live = -1, int grp = -1, hipStream_t s = nullptr) {
        if (!s) s = stream;
        const int cap = grp < 0 ? plan.B : plan.B / ng_now;
        if (live < 0 || live > cap) live = cap;
        auto tp = [&](int t) -> uint8_t* { return t < 0 ? nullptr : grp < 0 ? (uint8_t*)tensors[t] : group_ptr(tensors[t], grp); };
        auto shift = [&](const void* ptr, int) -> void* { return group_ptr(ptr, grp); };
        const int b0 = grp < 0 ? 0 : 1;   // (non-zero: re-address the prepared parameters)
        bool skip_next = false;           // the op was folded into the previous launch (fuse_head, mlp32)
        int skip_attn32 = 0;              // ops still to skip behind a swinattn32 launch
        GemmParams stem_p; bool stem_held = false, held_up = false; double stem_flops = 0;   // the op is folded into the NEXT launch (fuse_stem): its re-addressed parameters wait here
        for (size_t i = 0; i < plan.ops.size(); ++i) try {
            const Op& op = plan.ops[i];
            cur_op = (int)i;
            struct Range { Impl* e; bool on; ~Range() { if (on) e->roctx_pop(); } } range{this, roctx_push != nullptr};   // W2X_ROCTX=1: a roctx range per plan op (rocprofv3 --marker-trace)
            if (range.on) roctx_push((std::to_string(i) + " " + op.name).c_str());
            switch (op.kind) {
                case OP_GEMM: {
                    if (skip_next) { skip_next = false; break; }
                    if (skip_attn32) { --skip_attn32; break; }
                    GemmParams p = gemm[i];
                    p.B = live;
                    if (b0) {
                        const GemmOp& g = op.g;
                        p.a.p = shift(p.a.p, g.a.t); p.res.p = shift(p.res.p, g.res.t); p.res2.p = shift(p.res2.p, g.res2.t); p.out.p = shift(p.out.p, g.out.t);
                        p.stats_in = (const float*)shift(p.stats_in, g.stats_in); p.stats_out = (float*)shift(p.stats_out, g.stats_out); p.pool_out = (float*)shift(p.pool_out, g.pool_out);
                        p.a_scale = (const float*)shift(p.a_scale, g.se_scale); p.res_scale = (const float*)shift(p.res_scale, g.res_scale);
                    }
                    if ((int)i == final_op && out_override) p.out.p = out_override;
                    if (fuse_attn32[i] && cfg.precision == Precision::TF32) {     // LayerNorm + window gather + qkv, attention core, proj + scatter + residual in one launch
                        const AttnOp& a = plan.ops[i + 1].at;
                        GemmParams q = gemm[i + 2];
                        if (b0) { const GemmOp& g2 = plan.ops[i + 2].g; q.out.p = shift(q.out.p, g2.out.t); q.res.p = shift(q.res.p, g2.res.t); q.stats_out = (float*)shift(q.stats_out, g2.stats_out); }
                        SwinAttn32Params m;
                        m.x = (const float*)p.a.p; m.y = (float*)q.out.p; m.res = (const float*)q.res.p; m.B = live; m.nwin = a.nwin; m.C = p.K; m.pix_per_item = p.Mrows;
                        m.table_in = p.win_table; m.table_out = q.win_table; m.stats_in = p.stats_in;
                        m.wqkv_h = attn32_w[i][0]; m.wqkv_l = attn32_w[i][1]; m.wproj_h = attn32_w[i][2]; m.wproj_l = attn32_w[i][3];
                        m.bqkv = p.bias; m.bproj = q.bias; m.scale = a.scale; m.bias = (const float*)blobs[a.bias]; m.maskid = (const int*)blobs[a.maskid];
                        m.stats_out = q.stats_out; m.eps_out = q.ln_eps;
                        stamp_begin(1, op.flops + plan.ops[i + 1].flops + plan.ops[i + 2].flops);
                        hipAssert(launch_swinattn32(m, s));
                        stamp_end();
                        skip_attn32 = 1;          // (the attention op skips itself below, then the proj op here)
                        break;
                    }
                    if (fuse_mlp32[i] && cfg.precision == Precision::TF32) {      // fc1 + GELU + fc2 + residual on fp32 rows in one launch; op i + 1 is skipped
                        GemmParams q = gemm[i + 1];
                        if (b0) { const GemmOp& g2 = plan.ops[i + 1].g; q.out.p = shift(q.out.p, g2.out.t); q.stats_out = (float*)shift(q.stats_out, g2.stats_out); }
                        Mlp32Params m;
                        m.x = (const float*)p.a.p; m.y = (float*)q.out.p; m.M = (long)live * p.Mrows; m.C = p.K; m.stats_in = p.stats_in;
                        m.w1h = mlp32_w[i][0]; m.w1l = mlp32_w[i][1]; m.w2h = mlp32_w[i][2]; m.w2l = mlp32_w[i][3];
                        m.b1 = p.bias; m.b2 = q.bias; m.stats_out = q.stats_out; m.eps_out = q.ln_eps;
                        stamp_begin(0, op.flops + plan.ops[i + 1].flops);
                        hipAssert(launch_mlp32(m, s));
                        stamp_end();
                        skip_next = true;
                        break;
                    }
                    if ((fuse_stem[i] || fuse_up[i]) && !check_general) { stem_p = p; stem_held = true; held_up = fuse_up[i] != 0; stem_flops = op.flops; break; }    // computed by the next op's launch
                    if (stem_held) {
                        stem_held = false;
                        stamp_begin(0, op.flops + stem_flops);
                        hipAssert(held_up ? launch_conv3_up(p, stem_p, s) : conv48_stem_supported(p, stem_p) ? launch_conv48_stem(p, stem_p, s) : launch_conv3_stem(p, stem_p, s));
                        stamp_end();
                        break;
                    }
                    stamp_begin(0, op.flops);
                    if (op.g.pool_out >= 0) pool_blocks[op.g.pool_out] = plan.elt == 2 && conv3_supported(p) ? conv3_tiles(p) : 0;   // partial sums per image written by this launch (0: plan default)
                    hipAssert(plan.elt == 4 ? launch_gemm_f32(p, s, cfg.precision == Precision::FP32) : pixgemm_supported(p) ? launch_pixgemm(p, s) : conv3_supported(p) ? launch_conv3(p, s) : conv3h_supported(p) ? launch_conv3h(p, s) : conv48_supported(p) ? launch_conv48(p, s) : stem_supported(p) ? launch_stem(p, s) : launch_gemm(p, s));
                    stamp_end();
                    if (check_general && plan.elt == 2 && (pixgemm_supported(p) || conv3_supported(p) || conv3h_supported(p) || conv48_supported(p) || stem_supported(p))) {   // diagnostic: the general kernel must agree
                        const TensorDesc& od = plan.tensors[op.g.out.t];
                        const size_t n = (size_t)live * od.H * od.W * od.C;
                        std::vector<uint16_t> a(n), b(n);
                        void* tmp = nullptr;
                        hipAssert(hipMalloc(&tmp, n * 2));
                        hipAssert(hipStreamSynchronize(stream));
                        hipAssert(hipMemcpy(a.data(), p.out.p, n * 2, hipMemcpyDeviceToHost));
                        hipAssert(hipMemcpy(tmp, p.out.p, n * 2, hipMemcpyDeviceToDevice));   // pixels neither kernel writes compare equal
                        GemmParams q = p; q.out.p = tmp; q.pool_out = nullptr;   // the pooling partials of the real launch stay
                        hipAssert(launch_gemm(q, stream));
                        hipAssert(hipStreamSynchronize(stream));
                        hipAssert(hipMemcpy(b.data(), tmp, n * 2, hipMemcpyDeviceToHost));
                        hipAssert(hipFree(tmp));
                        double md = 0; size_t at = 0, bad = 0;
                        for (size_t k = 0; k < n; ++k) { const double d = std::fabs(f16_to_f32(a[k]) - f16_to_f32(b[k])); if (d > 0.01) ++bad; if (d > md) { md = d; at = k; } }
                        log(Severity::warn, "pixgemm check op " + std::to_string(i) + " [" + op.name + "]: max|d|=" + std::to_string(md) + " at pixel " + std::to_string(at / od.C) +
                            " ch " + std::to_string(at % od.C) + " (x=" + std::to_string(at / od.C % od.W) + ", y=" + std::to_string(at / od.C / od.W % od.H) + "), " + std::to_string(bad) + " of " + std::to_string(n) + " off by > 0.01");
                    }
                    break;
                }
                case OP_ATTN: {
                    if (i > 0 && fuse_attn32[i - 1] && cfg.precision == Precision::TF32) break;      // computed by the launch of the op in front
                    const AttnOp& a = op.at;
                    AttnParams p;
                    p.qkv = tp(a.qkv); p.out = tp(a.out); p.B = live; p.nwin = a.nwin; p.heads = a.heads; p.hd = a.hd;
                    p.ntok = a.ws * a.ws; p.scale = a.scale; p.bias = blobs[a.bias]; p.maskid = (const int*)blobs[a.maskid];
                    stamp_begin(1, op.flops);
                    hipAssert(plan.elt == 4 ? launch_attn_f32(p, s) : launch_attn(p, s));
                    stamp_end();
                    break;
                }
                case OP_SWINATTN: {
                    const SwinAttnOp& a = op.sa;
                    const TensorDesc& d = plan.tensors[a.x];
                    SwinAttnParams p;
                    p.x = tp(a.x); p.y = tp(a.y); p.table = (const int*)blobs[a.table]; p.H = a.H; p.W = a.W; p.ry = a.ry; p.rx = a.rx; p.B = live; p.nwin = a.nwin; p.C = a.C; p.hd = a.hd;
                    p.wqkv = blobs[a.wqkv]; p.bqkv = (const float*)blobs[a.bqkv]; p.scale = a.scale; p.bias32 = (const float*)blobs[a.bias]; p.maskid = (const int*)blobs[a.maskid];
                    p.wproj = blobs[a.wproj]; p.bproj = (const float*)blobs[a.bproj]; p.eps = a.eps;
                    p.wqkv_frag = frag_blobs[a.wqkv]; p.wproj_frag = frag_blobs[a.wproj];
                    p.stats_out = (float*)tp(a.stats_out); p.eps_out = a.eps_out;
                    if (d.C != a.C || plan.tensors[a.y].C != a.C || d.H * d.W != a.nwin * a.ws * a.ws) throw std::runtime_error("plan: attention geometry mismatch");
                    stamp_begin(1, op.flops);
                    hipAssert(launch_swin_attn(p, s));
                    stamp_end();
                    break;
                }
                case OP_MLP: {
                    const MlpOp& m = op.m;
                    const TensorDesc& d = plan.tensors[m.x];
                    MlpParams p;
                    p.x = tp(m.x); p.y = tp(m.y); p.M = (long)live * d.H * d.W; p.C = m.C;
                    p.w1 = blobs[m.w1]; p.b1 = (const float*)blobs[m.b1]; p.w2 = blobs[m.w2]; p.b2 = (const float*)blobs[m.b2];
                    p.w1_frag = frag_blobs[m.w1]; p.w2_frag = frag_blobs[m.w2]; p.frag32 = mlp_frag32(m.C);
                    p.eps = m.eps; p.stats_out = (float*)tp(m.stats_out); p.eps_out = m.eps_out;
                    if (d.C != m.C || plan.tensors[m.y].C != m.C) throw std::runtime_error("plan: MLP width mismatch");
                    double flops = op.flops;
                    if (fuse_head[i] && !check_general) {      // the image head rides on this launch
                        const GemmParams& g = gemm[i + 1];
                        void* out = b0 ? shift(g.out.p, plan.ops[i + 1].g.out.t) : g.out.p;
                        if ((int)i + 1 == final_op && out_override) out = out_override;
                        p.ti_w = g.wt_frag; p.ti_b = g.bias; p.ti_out = out; p.ti_Hs = g.out.Hs; p.ti_Ws = g.out.Ws; p.ti_Mrows = g.Mrows; p.ti_aW = g.aW;
                        p.ti_clip = g.has_clip; p.ti_lo = g.clip_lo; p.ti_hi = g.clip_hi;
                        flops += plan.ops[i + 1].flops;
                        skip_next = true;
                    }
                    stamp_begin(5, flops);
                    hipAssert(launch_mlp(p, s));
                    stamp_end();
                    break;
                }
                case OP_SE: {
                    const SeOp& se = op.se;
                    SeParams p;
                    p.pool = (const float*)tp(se.pool); p.scale = (float*)tp(se.scale); p.B = live; p.C = se.C;
                    p.Cs = plan.tensors[se.pool].C; p.Cmid = se.Cmid; p.inv_count = se.inv_count; p.nblocks = pool_blocks[se.pool] > 0 ? pool_blocks[se.pool] : se.nblocks; p.Mrows = se.Mrows;
                    p.w1 = (const float*)blobs[se.w1]; p.b1 = (const float*)blobs[se.b1]; p.w2 = (const float*)blobs[se.w2]; p.b2 = (const float*)blobs[se.b2];
                    stamp_begin(2, 0);
                    hipAssert(launch_se(p, s));
                    stamp_end();
                    break;
                }
                case OP_SCALE_ADD: {
                    const TensorDesc& d = plan.tensors[op.se.pool];
                    stamp_begin(2, 0);
                    hipAssert(launch_scale(tp(op.se.pool), (const float*)tp(op.se.scale), live, d.H * d.W, d.C, plan.elt == 4, s));
                    stamp_end();
                    break;
                }
                default: throw std::runtime_error("plan: unknown op kind");
            }
        } catch (const std::exception& e) {     // name the op: "invalid argument" alone says nothing about a 60-op plan
            throw std::runtime_error("op " + std::to_string(i) + " [" + plan.ops[i].name + "]: " + e.what());
        }
        cur_op = -1;
    }

    template <class T> void ensure(T*& p, size_t& cap, size_t bytes) {
        if (bytes <= cap) return;
        drop_graphs();                        // captured passes hold the old addresses
        if (p) hipAssert(hipFree(p));
        p = nullptr; cap = 0;
        hipAssert(hipMalloc((void**)&p, bytes));
        cap = bytes;
    }

    // device part of one frame: gather -> network per batch -> compose.  Frame must already be in d_frame.
    void run_frame(int rows, int cols, const TileGrid& grid, bool report, const StripPlan& sp) {
        run_passes(rows, cols, sp.tile_count, 0, report, 0, true);
        compose_rect(rows, cols, grid, sp.x0, sp.x1, 0, 0, sp.first_tile);
    }

    // One frame of a ROLLING sequence (benchResident, renderSequence): frames of one size, one after the other, whose passes all run as two tile groups.
    // A lone frame ends with a join (the first stream waits for the second group), then the compose launch runs alone, then the next frame forks again:
    // a tenth of a millisecond of compose plus the last tiles of the longer group with half the device idle, per frame.  Here nothing joins: the first
    // stream carries its group from pass to pass and from frame to frame; the second stream carries the other group and, behind it, the frame's compose
    // launch (which waits for the first stream's last pass of the frame by event) - so frame f is composed while frame f + 1's first group is already
    // running.  What that takes: a second tile slab (frame f + 1's tiles must not land on the tiles compose(f) is reading; `which` alternates), and
    // events instead of stream order where a buffer comes round again (ev_cmp).  Same launches on the same data: the frames are the bytes of render().
    // Returns false when the frame cannot roll (a pass that does not split, profiling, one group): the caller then runs run_frame().
    bool can_roll(int tile_count) const {
        const int steps = cfg.tta ? 8 : 1, B = plan.B;
        const int last_live = tile_count * steps - (tile_count * steps - 1) / B * B;                    // live slots of the frame's last pass
        return rolling_ok && groups == 2 && use_graphs && !profiling && !check_general && !poison && gstream[0] && last_live >= 8 && B % 2 == 0;
    }
    void run_rolling_frame(int rows, int cols, const TileGrid& grid, const StripPlan& sp, int which, hipEvent_t out_free) {
        for (int k = 0; k < 2; ++k) {
            if (!ev_g0[k]) hipAssert(hipEventCreateWithFlags(&ev_g0[k], hipEventDisableTiming));
            if (!ev_cmp[k]) hipAssert(hipEventCreateWithFlags(&ev_cmp[k], hipEventDisableTiming));
        }
        struct Swap { Impl* e; void* slab; size_t cap; bool on; ~Swap() { if (on) { std::swap(e->d_slab, e->d_slab2); std::swap(e->slab_cap, e->slab2_cap); } e->rolling = false; } } sw{this, d_slab, slab_cap, which == 1};
        if (which == 1) { std::swap(d_slab, d_slab2); std::swap(slab_cap, slab2_cap); }
        rolling = true;
        hipAssert(hipStreamWaitEvent(stream, ev_cmp[which], 0));          // the compose launch that last read this slab (two frames ago) is done (a never-recorded event does not wait)
        run_passes(rows, cols, sp.tile_count, 0, false, 0, true);
        hipAssert(hipEventRecord(ev_g0[which], stream));
        hipStream_t s2 = gstream[0];
        hipAssert(hipStreamWaitEvent(s2, ev_g0[which], 0));
        if (out_free) hipAssert(hipStreamWaitEvent(s2, out_free, 0));     // the frame that last left through this output buffer has been downloaded
        compose_rect(rows, cols, grid, sp.x0, sp.x1, 0, 0, sp.first_tile, s2);
        hipAssert(hipEventRecord(ev_cmp[which], s2));
    }
    // after the last frame of a rolling sequence: the first stream waits for the second, so that whatever follows on it sees the sequence done
    void end_rolling() { hipAssert(hipEventRecord(ev_join[0], gstream[0])); hipAssert(hipStreamWaitEvent(stream, ev_join[0], 0)); }
    // The error exits of renderPart() and renderSequence(): every stream that may still touch the caller's buffers, the slabs or d_out drains before the
    // function returns false (errors here are ignored, the first one is what gets reported); the next call starts outside a rolling sequence.
    void drain_after_error() {
        rolling = false;
        for (hipStream_t st : {s_up, s_dn, gstream[0], stream}) if (st && hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError();
    }

    // the network passes of `tile_count` tiles (slots d_slots[slots_off ..]); their outputs go to slab slots slab_slot0, slab_slot0 + 1, ...
    // fresh: the first passes of a frame (W2X_POISON wipes the arena and the slab here, not between the parts of a pipelined frame)
    void run_passes(int rows, int cols, int tile_count, size_t slab_slot0, bool report, size_t slots_off, bool fresh, int batch0 = 0, int batch_total = 0) {
        const int B = plan.B, T = plan.T, To = plan.Tout;
        const int steps = cfg.tta ? 8 : 1;
        const int userB = plan.userB, S = B / userB;
        const int batchCount = (int)std::lround(std::ceil((double)(tile_count * steps) / userB));   // img2img_render.cpp:249
        const int passCount = (batchCount + S - 1) / S;
        const size_t slot_bytes = (size_t)To * To * 4 * plan.elt;
        if (poison && fresh) {
            hipAssert(hipMemsetAsync(arena_base, 0x7E, arena_bytes, stream));
            hipAssert(hipMemsetAsync(d_slab, 0x7E, slab_cap, stream));
        }
        // (W2X_POISON fills the whole arena before the frame and so runs the default path: tile groups in their arena parts, replayed graphs)
        const bool graphable = use_graphs && !profiling && !check_general;
        for (int bi = 0; bi < passCount; ++bi) {
            const auto t0 = std::chrono::steady_clock::now();
            const int live = std::max(0, std::min(B, tile_count * steps - bi * B));
            void* const slab_out = (uint8_t*)d_slab + (slab_slot0 + (size_t)bi * B) * slot_bytes;
            GatherParams gp0;
            gp0.frame = d_frame; gp0.rows = rows; gp0.cols = cols; gp0.step = (size_t)cols * 3 * (deep ? 2 : 1); gp0.deep = deep ? 1 : 0;
            gp0.out = tensors[plan.in_tensor]; gp0.slots = d_slots + slots_off + (size_t)bi * B; gp0.B = B; gp0.T = T; gp0.fp32 = plan.elt == 4;
            const int NG = groups;
            struct NgReset { int& r; ~NgReset() { r = 1; } } ng_reset{ng_now};   // also when a launch throws mid-pass
            ng_now = NG;                                                          // (arena_part() is the part of an NG-group pass)
            const bool split = NG > 1 && !profiling && !check_general && live >= 4 * NG && B % NG == 0 && (size_t)NG * arena_part() <= arena_bytes + 1024;
            ng_now = 1;
            // NG tile groups side by side: tiles never exchange data, so the groups run the same launches on their own streams in
            // their own parts of the arena.  Each kernel then has 1/NG of the workgroups, but a kernel's ramp and tail (and the
            // gaps between launches) fill with the other groups' work.  Bit-identical by construction.
            int first[5] = {0, 0, 0, 0, 0};
            for (int grp = 0; grp < NG; ++grp) first[grp + 1] = first[grp] + live / NG + (grp < live % NG ? 1 : 0);
            if (split) {
                if (!ev_fork) hipAssert(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
                for (int k = 0; k + 1 < NG; ++k) if (!gstream[k]) {
                    hipAssert(hipStreamCreateWithFlags(&gstream[k], hipStreamNonBlocking));
                    hipAssert(hipEventCreateWithFlags(&ev_join[k], hipEventDisableTiming));
                }
            }
            auto group_stream = [&](int grp) { return grp ? gstream[grp - 1] : stream; };
            auto gather_group = [&](int grp, hipStream_t gs) {       // the group's tiles at the start of its part of the arena
                GatherParams gp = gp0;
                gp.out = group_ptr(tensors[plan.in_tensor], grp); gp.slots = gp0.slots + (size_t)first[grp]; gp.B = first[grp + 1] - first[grp];
                hipAssert(launch_gather(gp, gs));
            };
            auto network_group = [&](int grp, hipStream_t gs) { run_network((uint8_t*)slab_out + (size_t)first[grp] * slot_bytes, first[grp + 1] - first[grp], grp, gs); };
            auto fork = [&] { hipAssert(hipEventRecord(ev_fork, stream)); for (int grp = 1; grp < NG; ++grp) hipAssert(hipStreamWaitEvent(gstream[grp - 1], ev_fork, 0)); };
            auto join = [&] { for (int grp = 1; grp < NG; ++grp) { hipAssert(hipEventRecord(ev_join[grp - 1], gstream[grp - 1])); hipAssert(hipStreamWaitEvent(stream, ev_join[grp - 1], 0)); } };
            // the whole pass as launches issued from `stream` (the groups forked off it by events: capturable as ONE graph with NG branches)
            auto run_pass = [&] {
                if (!split) {
                    stamp_begin(3, 0);
                    hipAssert(launch_gather(gp0, stream));
                    stamp_end();
                    run_network(slab_out, live);
                    return;
                }
                ng_now = NG;
                for (int grp = 0; grp < NG; ++grp) gather_group(grp, stream);
                fork();
                for (int grp = 0; grp < NG; ++grp) network_group(grp, group_stream(grp));
                join();
                ng_now = 1;
            };
            // one group of a split pass, gather included, as launches on ONE stream (capturable as a straight-line graph)
            auto run_group = [&](int grp) { ng_now = NG; gather_group(grp, group_stream(grp)); network_group(grp, group_stream(grp)); ng_now = 1; };
            const bool per_group = split;
            // (a rolling sequence leaves the groups un-joined: each stream carries its group from pass to pass and from frame to frame, run_rolling_frame)
            const bool no_join = rolling && split;
            auto run_eager = [&] { if (per_group) { fork(); for (int grp = 0; grp < NG; ++grp) run_group(grp); if (!no_join) join(); } else { if (rolling && gstream[0]) { join(); } run_pass(); } };
            if (!graphable) run_eager();
            else {
                const GraphKey key{d_frame, d_slots + slots_off + (size_t)bi * B, slab_out, arena_base, rows, cols, live, deep ? 1 : 0};
                auto replay = [&](const PassGraphs& pg) {
                    if (pg.n == 1) { if (rolling && gstream[0]) join(); hipAssert(hipGraphLaunch(pg.g[0], stream)); return; }   // (a whole-arena pass inside a rolling sequence: the other stream's group first)
                    fork();
                    for (int grp = 0; grp < pg.n; ++grp) hipAssert(hipGraphLaunch(pg.g[grp], group_stream(grp)));
                    if (!no_join) join();
                };
                auto it = graphs.find(key);
                if (it != graphs.end()) { replay(it->second); ++graph_replays; }
                else if (graph_seen.size() >= 4096 && !graph_seen.count(key)) { graph_seen.clear(); run_eager(); ++eager_passes; }   // sizes that keep changing: bounded bookkeeping
                else if (graph_seen[key]++ == 0 && !(rolling && !graphs.empty())) { run_eager(); ++eager_passes; }   // (a rolling frame on the second slab repeats launches that have run: captured at first sight)
                else {
                    if (graphs.size() >= 1024) drop_graphs();      // frames of ever-changing sizes: start over rather than grow without bound
                    // Capture -> instantiate -> launch.  Nothing runs while a stream captures, so whatever fails on the way (begin,
                    // a capture invalidated by a runtime call inside a launcher, end, instantiate) the pass is still to be done:
                    // graphs are switched off for good (otherwise every later frame would retry and fail again) and it runs on plain launches.
                    PassGraphs pg;
                    std::string why;
                    const int ncap = per_group ? NG : 1;
                    for (int c = 0; c < ncap && why.empty(); ++c) {
                        hipStream_t cs = per_group ? group_stream(c) : stream;
                        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
                        hipError_t ge = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
                        if (ge != hipSuccess) { why = std::string("begin capture: ") + hipGetErrorString(ge); break; }
                        try { if (per_group) run_group(c); else run_pass(); } catch (const std::exception& e) { why = std::string("capture: ") + e.what(); }
                        ge = hipStreamEndCapture(cs, &graph);
                        if (why.empty() && ge != hipSuccess) why = std::string("end capture: ") + hipGetErrorString(ge);
                        if (why.empty()) { ge = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0); if (ge != hipSuccess) why = std::string("instantiation: ") + hipGetErrorString(ge); }
                        if (graph) (void)hipGraphDestroy(graph);
                        if (why.empty()) pg.g[pg.n++] = exec;
                    }
                    if (!why.empty()) {
                        (void)hipGetLastError();
                        for (int c = 0; c < pg.n; ++c) (void)hipGraphExecDestroy(pg.g[c]);
                        use_graphs = false;
                        log(Severity::warn, "hipGraph " + why + " - passes stay on plain launches");
                        run_eager(); ++eager_passes;          // a real launch error surfaces here, outside the capture
                    } else {
                        graphs[key] = pg;
                        replay(pg); ++graph_replays;
                    }
                }
            }
            if (report) {
                const auto t1 = std::chrono::steady_clock::now();
                const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
                for (int k = bi * S; k < std::min((bi + 1) * S, batchCount); ++k)
                    log(batch0 + k + 1, batch_total > 0 ? batch_total : batchCount, 1000.0 * S / std::max(ms, 1e-6));                     // :336-338
            }
        }
        last_batches = batch0 + batchCount;
    }

    // compose: output columns [x0, x1) (x1 = 0: to the right edge) and rows [y0, y1) (y1 = 0: to the bottom) from the slab, whose slot 0 holds
    // global tile `first_tile`
    void compose_rect(int rows, int cols, const TileGrid& grid, int x0, int x1, int y0, int y1, long first_tile, hipStream_t on = nullptr) {
        const int To = plan.Tout;
        ComposeParams cp;
        cp.tiles = d_slab; cp.fp32 = plan.elt == 4; cp.dst = d_out; cp.dst_step = (size_t)cols * cfg.scaling * 3 * (deep ? 2 : 1); cp.deep = deep ? 1 : 0;
        cp.outW = cols * cfg.scaling; cp.outH = rows * cfg.scaling; cp.To = To;
        cp.nx = grid.nx; cp.ny = grid.ny; cp.stride_x = To - grid.outOvX; cp.stride_y = To - grid.outOvY;
        const bool overlapping = cfg.overlapX != 0 || cfg.overlapY != 0;                      // :244
        cp.ovx = overlapping ? ovx : 0; cp.ovy = overlapping ? ovy : 0;
        cp.ramp_x = d_rampx; cp.ramp_y = d_rampy; cp.tta = cfg.tta ? 1 : 0; cp.tta_bug_compat = cfg.ttaBugCompat ? 1 : 0;
        cp.x0 = x0; cp.x1 = x1; cp.y0 = y0; cp.y1 = y1; cp.first_tile = first_tile;
        stamp_begin(4, 0);
        hipAssert(launch_compose(cp, on ? on : stream));
        stamp_end();
    }
};

Img2Img::Img2Img() : impl(new Impl) {}
Img2Img::~Img2Img() = default;

void Img2Img::setMessageCallback(MessageCallback callback) { impl->messageCallback = std::move(callback); }
void Img2Img::setProgressCallback(ProgressCallback callback) { impl->progressCallback = std::move(callback); }

bool Img2Img::build(const std::string& onnxModelPath, const BuildConfig& config) try {
    // img2img_build.cpp:56-64
    const int dev = physical_device(config.deviceId);
    if (const std::string why = device_ordinal_problem(config.deviceId, dev); !why.empty()) {
        W2X_LOG(error, "Failed to set hip device: " + why + ".");
        return false;
    }
    std::unique_ptr<DeviceGuard> guard;
    try {
        guard.reset(new DeviceGuard(dev));
    } catch (const std::exception& e) {
        W2X_LOG(error, "Failed to set hip device to device id " + std::to_string(config.deviceId) + ": " + std::string(e.what()) + ".");
        return false;
    }
    // :123-135 - precision: FP16 = the fused fp16 kernels; gfx950 has no TF32 matrix instruction, a TF32 request gets the fp32-storage
    // engine (k_f32.hip: fp32 maps and accumulation, un-fused operator set) with split-bf16 products, FP32 the same engine with exact
    // fp32 products (include/w2x/config.h)
    switches_from_env();                                   // (W2X_SUPERBATCH is read by the lowering: switches.h)
    const bool fp32 = config.precision != Precision::FP16;
    if (config.precision == Precision::TF32) W2X_LOG(info, "Precision TF32: this platform has no TF32 matrix instructions, the engine keeps fp32 maps and multiplies split-bf16 operands (16 significant bits each).");
    // :81-88 parse ; :102-116 one profile: the plan on disk is specialised for the opt shape (channels come from the model);
    // any other shape inside [min, max] is specialised by load() from the same ONNX file
    if (config.minBatchSize > config.optBatchSize || config.optBatchSize > config.maxBatchSize || config.minWidth > config.optWidth || config.optWidth > config.maxWidth ||
        config.minHeight > config.optHeight || config.optHeight > config.maxHeight || config.minChannels > config.optChannels || config.optChannels > config.maxChannels) {
        W2X_LOG(error, "Failed to build engine: optimization profile is not min <= opt <= max.");
        return false;
    }
    Plan plan;
    try {
        W2X_LOG(info, "ONNX graph \"" + onnxModelPath + "\": " + onnx_op_histogram(onnxModelPath) + ".");
        plan = lower_for_shape(onnxModelPath, config.optBatchSize, config.optChannels, config.optHeight, config.optWidth, fp32);
    } catch (const std::exception& e) {
        W2X_LOG(error, "Failed to parse ONNX model: " + std::string(e.what()) + ".");
        return false;
    }
    W2X_LOG(info, "Lowered \"" + onnxModelPath + "\": " + std::to_string(plan.ops.size()) + " fused ops, " +
                      std::to_string((long long)plan.flops) + " algorithmic FLOP per batch, output tile " + std::to_string(plan.Tout) + ".");
    // :151-161
    const std::string deviceName = hipGetDeviceName(dev);
    const auto basePath = std::filesystem::path(onnxModelPath).replace_extension("").string() + "_" + getConfigHash(config, deviceName).substr(0, 16);
    serializeConfig(basePath + ".json", config, deviceName);
    try {
        const auto bytes = plan.serialize();
        std::ofstream engineFile(basePath + kEngineExt, std::ios::binary);
        if (!engineFile.is_open()) throw std::runtime_error("could not open \"" + basePath + kEngineExt + "\"");
        engineFile.write((const char*)bytes.data(), (std::streamsize)bytes.size());
    } catch (const std::exception& e) {
        W2X_LOG(error, "Failed to serialize network to disk: " + std::string(e.what()) + ".");
        return false;
    }
    return true;
} catch (const std::exception& e) {
    W2X_LOG(error, "Engine build failed unexpectedly: " + std::string(e.what()) + ".");
    return false;
}

bool Img2Img::load(const std::string& modelPath, const RenderConfig& config) try {
    namespace fs = std::filesystem;
    // img2img_load.cpp:127-135
    const int dev = physical_device(config.deviceId);
    if (const std::string why = device_ordinal_problem(config.deviceId, dev); !why.empty()) {
        W2X_LOG(error, "Failed to set hip device: " + why + ".");
        return false;
    }
    std::unique_ptr<DeviceGuard> guard;
    try {
        guard.reset(new DeviceGuard(dev));
    } catch (const std::exception& e) {
        W2X_LOG(error, "Failed to set hip device to device id " + std::to_string(config.deviceId) + ": " + std::string(e.what()) + ".");
        return false;
    }
    // :79-114 getEnginePath
    std::string enginePath;
    bool optimized = false;
    try {
        if (!fs::exists(modelPath)) throw std::runtime_error("model file does not exist");
        const std::string runningOn = hipGetDeviceName(dev);
        const std::string engineName = fs::path(modelPath).stem().string();
        fs::path dir = fs::path(modelPath).parent_path();
        if (dir.empty()) dir = ".";
        std::vector<fs::path> entries;
        for (const auto& entry : fs::directory_iterator(dir)) if (entry.is_regular_file()) entries.push_back(entry.path());
        std::sort(entries.begin(), entries.end());
        for (const auto& path : entries) {
            if (path.filename().string().rfind(engineName, 0) != 0 || path.extension().string() != kEngineExt) continue;
            const std::string configPath = fs::path(path).replace_extension("").string() + ".json";
            if (!fs::exists(configPath)) continue;
            BuildConfig bc; std::string builtOn;
            deserializeConfig(configPath, bc, builtOn);
            // img2img_load.cpp:100-107: the first optimized engine, else the first compatible one
            if (!isCompatible(config, bc, builtOn, runningOn)) continue;
            if (isOptimized(config, bc)) { enginePath = path.string(); optimized = true; break; }
            if (enginePath.empty()) enginePath = path.string();
        }
        if (enginePath.empty()) throw std::runtime_error("could not satisfy render configuration");
    } catch (const std::exception& e) {
        W2X_LOG(error, "Failed to find engine file for model \"" + modelPath + "\": " + std::string(e.what()) + ".");
        return false;
    }
    // :137-147
    std::ifstream file(enginePath, std::ios::binary | std::ios::ate);
    if (!file.is_open()) { W2X_LOG(error, "Failed to open engine file \"" + enginePath + "\"."); return false; }
    std::streamsize fileSize = file.tellg();
    std::vector<char> engineBuffer((size_t)fileSize);
    file.seekg(0, std::ios::beg);
    file.read(engineBuffer.data(), fileSize);

    impl->release();   // :149-154, :209-222
    impl->device = dev;
    switches_from_env();                                   // every switch of the library: switches.h
    const Switches& sw = switches();
    impl->poison = sw.poison;
    impl->check_general = sw.check_general;
    impl->use_graphs = !sw.no_graph;
    impl->rolling_ok = !sw.no_rolling;
    impl->pipeline_parts = std::min(Impl::kMaxRenderParts, std::max(1, sw.render_parts));
    if (const std::string nd = switches_nondefault(); !nd.empty()) W2X_LOG(warn, "Switches off their defaults: " + nd + ".");
    if (sw.roctx && !impl->roctx_push) {
        if (void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL)) {
            impl->roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            impl->roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
            if (!impl->roctx_push || !impl->roctx_pop) impl->roctx_push = nullptr;
        }
        if (!impl->roctx_push) W2X_LOG(warn, "W2X_ROCTX is set but libroctx64.so could not be loaded: no marker ranges.");
    }
    try {
        impl->plan = Plan::deserialize((const uint8_t*)engineBuffer.data(), engineBuffer.size());
    } catch (const std::exception& e) {
        W2X_LOG(error, "Failed to deserialize engine from buffer: " + std::string(e.what()) + ".");
        return false;
    }
    if (!optimized) {
        // A TensorRT engine runs any shape of its optimization profile (img2img_build.cpp:102-116); a plan is specialised for one
        // shape, so for a compatible-but-not-optimized engine the same ONNX file is lowered again for the requested shape (the
        // engine file only vouches for the device, the precision and the range).
        W2X_LOG(warn, "Engine \"" + enginePath + "\" is compatible with but not optimized for the render configuration; specialising the plan for batch " +
                          std::to_string(config.batchSize) + ", tile " + std::to_string(config.width) + "x" + std::to_string(config.height) + ".");
        try {
            impl->plan = lower_for_shape(modelPath, config.batchSize, config.channels, config.height, config.width, config.precision != Precision::FP16);
        } catch (const std::exception& e) {
            W2X_LOG(error, "Failed to set input tensor shape: " + std::string(e.what()) + ".");
            return false;
        }
    }
    const Plan& plan = impl->plan;
    if ((plan.elt == 4) != (config.precision != Precision::FP16)) {   // the JSON sidecar and the engine file disagree
        W2X_LOG(error, "Failed to deserialize engine from buffer: precision of the plan differs from the engine's configuration.");
        return false;
    }
    // :197-203 - the input shape must be the one the plan was specialised for; T' is read from the plan
    if (plan.userB != config.batchSize || plan.B % std::max(plan.userB, 1) || plan.Cin != config.channels || plan.T != config.height || plan.T != config.width) {
        W2X_LOG(error, "Failed to set input tensor shape.");
        return false;
    }
    hipAssert(hipStreamCreateWithFlags(&impl->stream, hipStreamNonBlocking));   // :206
    hipAssert(hipEventCreate(&impl->ev0));
    hipAssert(hipEventCreate(&impl->ev1));
    impl->groups = sw.groups;
    // Every stream the engine will use is created HERE, before anything touches the null stream (upload_plan()'s synchronous copies do): the runtime hands its
    // hardware queues (four by default) to streams in creation order and lets later streams share, so created now the compute stream, the two copy streams and the
    // second tile group's stream have a queue each and the null stream shares one.  Created lazily, in whatever order the first calls needed them, the second
    // group's kernels shared a queue with a copy stream on some orders: renderSequence() 7.69 (this order) / 7.85 (lazy) / 10.0 ms per frame (group stream
    // first), profiles/r4_kernels/stream_order.txt.  Further group streams (W2X_GROUPS > 2) are created by the first pass that splits.
    impl->ensure_copy_streams();
    if (impl->groups > 1) { hipAssert(hipStreamCreateWithFlags(&impl->gstream[0], hipStreamNonBlocking)); hipAssert(hipEventCreateWithFlags(&impl->ev_join[0], hipEventDisableTiming)); }
    try {
        impl->upload_plan();                                                      // :225-248
    } catch (const std::exception& e) {
        W2X_LOG(error, "Failed to allocate resources: " + std::string(e.what()) + ".");
        impl->release();
        return false;
    }
    impl->cfg = config;
    W2X_LOG(info, "Loaded \"" + enginePath + "\": " + std::to_string(plan.ops.size()) + " ops, " + std::to_string(plan.B) + " tiles per pass, activation arena " +
                      std::to_string(impl->arena_bytes >> 20) + " MiB" +
                      (std::count(impl->fuse_stem.begin(), impl->fuse_stem.end(), (char)1) ? ", " + std::to_string(std::count(impl->fuse_stem.begin(), impl->fuse_stem.end(), (char)1)) + " stem folded into the launch of the convolution behind it" : "") +
                      (std::count(impl->fuse_up.begin(), impl->fuse_up.end(), (char)1) ? ", " + std::to_string(std::count(impl->fuse_up.begin(), impl->fuse_up.end(), (char)1)) + " transposed convolution folded into the launch of the convolution behind it" : "") +
                      (std::count(impl->fuse_head.begin(), impl->fuse_head.end(), (char)1) ? ", image head folded into the last MLP launch" : "") +
                      (config.precision == Precision::TF32 && std::count(impl->fuse_attn32.begin(), impl->fuse_attn32.end(), (char)1) + std::count(impl->fuse_mlp32.begin(), impl->fuse_mlp32.end(), (char)1) > 0
                           ? ", " + std::to_string(std::count(impl->fuse_attn32.begin(), impl->fuse_attn32.end(), (char)1)) + " attention and " +
                                 std::to_string(std::count(impl->fuse_mlp32.begin(), impl->fuse_mlp32.end(), (char)1)) + " MLP branches as fused fp32-row launches." : "."));
    // :262-269 blend ramps
    impl->ovx = (int)std::lround(plan.T * config.scaling * config.overlapX);
    impl->ovy = (int)std::lround(plan.T * config.scaling * config.overlapY);
    if (config.overlapX != 0 || config.overlapY != 0) {
        auto rx = blend_ramp(impl->ovx), ry = blend_ramp(impl->ovy);
        hipAssert(hipMalloc((void**)&impl->d_rampx, std::max<size_t>(rx.size(), 1) * sizeof(float)));
        hipAssert(hipMalloc((void**)&impl->d_rampy, std::max<size_t>(ry.size(), 1) * sizeof(float)));
        hipAssert(hipMemcpy(impl->d_rampx, rx.data(), rx.size() * sizeof(float), hipMemcpyHostToDevice));
        hipAssert(hipMemcpy(impl->d_rampy, ry.data(), ry.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    impl->loaded = true;
    return true;
} catch (const std::exception& e) {
    W2X_LOG(error, "Engine load failed unexpectedly: " + std::string(e.what()) + ".");
    impl->release();
    return false;
}

bool Img2Img::render(const Image& src, Image& dst) { return renderPart(src, dst, 0, 1, "render"); }

// One GPU's share of a frame when a single image is spread over several devices: renders and writes only the
// output columns strip_plan() assigns to `part`; the other columns of dst are left untouched.  part 0 of 1 = render().
bool Img2Img::renderStrip(const Image& src, Image& dst, int part, int parts) { return renderPart(src, dst, part, parts, "renderStrip"); }

bool Img2Img::renderPart(const Image& src, Image& dst, int part, int parts, const char* who) try {
    if (!impl->loaded) { W2X_LOG_AS(who, error, "Render called before a successful load."); return false; }
    DeviceGuard guard(impl->device);
    if (parts <= 0 || part < 0 || part >= parts) { W2X_LOG_AS(who, error, "Invalid strip index."); return false; }
    const RenderConfig& cfg = impl->cfg;
    const Plan& plan = impl->plan;
    const int rows = src.rows, cols = src.cols, s = cfg.scaling;
    if ((src.depth != 8 && src.depth != 16) || dst.depth != src.depth) { W2X_LOG_AS(who, error, "Input and output images must both be 8-bit or both 16-bit."); return false; }
    const size_t bps = src.depth / 8;                          // bytes per sample
    if (!src.data || rows <= 0 || cols <= 0 || src.step < (size_t)cols * 3 * bps) { W2X_LOG_AS(who, error, "Input image is empty or has an invalid step."); return false; }
    if (!dst.data || dst.rows != rows * s || dst.cols != cols * s || dst.step < (size_t)dst.cols * 3 * bps) {
        W2X_LOG_AS(who, error, "Output image has invalid size: expected " + std::to_string(cols * s) + "x" + std::to_string(rows * s) + ".");
        return false;
    }
    hipStream_t stream = impl->stream;
    // img2img_render.cpp:226 upload
    impl->ensure(impl->d_frame, impl->frame_cap, (size_t)rows * cols * 3 * bps);
    impl->ensure(impl->d_out, impl->out_cap, (size_t)rows * s * cols * s * 3 * bps);
    impl->deep = bps == 2;
    // img2img_render.cpp:226 upload: below, once the parts are known (a frame that runs in parts uploads the first part's columns first)
    // :232-240
    TileGrid grid = calculate_tiles(cols, rows, cols * s, rows * s, plan.T, plan.T, plan.Tout, plan.Tout, s, cfg.overlapX, cfg.overlapY);
    if (grid.count <= 0) { W2X_LOG_AS(who, error, "Tile grid is empty."); return false; }
    for (const Rect& r : grid.out) if (r.w <= 0 || r.h <= 0) { W2X_LOG_AS(who, error, "Tile grid does not fit the output (scaling does not match the model)."); return false; }
    const StripPlan sp = strip_plan(grid, cols * s, plan.Tout, part, parts);
    if (sp.tile_count == 0) return true;                       // more devices than tile columns: this one has no share
    // :246-267 step schedule: slot = step index, tile = step / stepsPerTile, aug = step % stepsPerTile, zero pad slots at the end
    const int steps = cfg.tta ? 8 : 1, B = plan.B, S = plan.B / plan.userB;
    // A whole frame (render()) runs as a few PARTS one after the other: a part's tiles, its canvas cells composed and handed to the download
    // stream, then the next part's while those cells travel to the host - the synchronous contract of img2img_render.cpp:226-344 with most of
    // the 100 MB download of a 4K frame off the critical path: only the last part's cells travel after the last kernel (config 3: 9.9 ms per
    // call as one part, 8.4 as three, profiles/r4_kernels/render_parts_*.txt).  The parts are those of shard_plan() (contiguous tile ranges, canvas cells of their
    // tiles; a part reads the tiles in front of its range from the same slab), cut at multiples of the batch size so that batches, their order
    // and the progress schedule (:246-250, 336-338) are the reference's.
    constexpr int kMaxParts = Impl::kMaxRenderParts;
    int npart = 1;
    int first_of[kMaxParts + 1] = {0, sp.tile_count, 0, 0, 0};   // part k = tiles [first_of[k], first_of[k + 1]) of the strip
    if (parts == 1 && impl->pipeline_parts > 1 && sp.tile_count >= 16 && grid.outOvX < plan.Tout - grid.outOvX && grid.outOvY < plan.Tout - grid.outOvY) {
        int q = 1; while ((q * steps) % plan.userB) ++q;        // part boundaries: whole reference batches
        const int want = std::min({impl->pipeline_parts, kMaxParts, sp.tile_count / 8});
        // the LAST part is the small one - its cells are the only ones that travel after the last kernel (a fifth of the tiles, at least 8: smaller passes no longer
        // fill the device) - and the tiles in front of it split evenly: config 3's 45 tiles run as 16 + 20 + 9 (8.4 ms per call; as 15 + 15 + 15: 8.8, 12 + 12 + 12 + 9: 8.45)
        int n = 0;
        const int body = want >= 2 ? (sp.tile_count - std::max(8, sp.tile_count / 5)) / q * q : 0;
        for (int k = 1; k < want; ++k) {
            const int t = k + 1 == want ? body : (int)((long)k * body / (want - 1)) / q * q;
            if (t > first_of[n] && t < sp.tile_count) first_of[++n] = t;
        }
        first_of[++n] = sp.tile_count;
        npart = n;
    }
    int tiles_of[kMaxParts] = {}, batches_of[kMaxParts] = {}, batches_before[kMaxParts] = {};
    size_t slots_off[kMaxParts] = {}; size_t stepTotal = 0; int batch_total = 0;
    for (int k = 0; k < npart; ++k) {
        tiles_of[k] = first_of[k + 1] - first_of[k];
        batches_of[k] = (int)std::lround(std::ceil((double)(tiles_of[k] * steps) / plan.userB));
        batches_before[k] = batch_total; batch_total += batches_of[k];
        slots_off[k] = stepTotal;
        stepTotal += (size_t)((batches_of[k] + S - 1) / S) * B;   // reference batches rounded up to whole network passes
    }
    // :226 upload.  In parts: the frame columns the first part's tiles read go first, on the compute stream; the rest follows on the upload stream while that
    // part computes (tiles are ordered by column, :43-44, so a part reads a column range; the second part waits for the event)
    int x_split = cols;
    if (npart > 1) {
        int xm = 0;
        for (int t = 0; t < first_of[1]; ++t) xm = std::max(xm, grid.in[sp.first_tile + t].x + plan.T);
        if (xm > 0 && xm < cols - 64) x_split = xm;
    }
    hipAssert(hipMemcpy2DAsync(impl->d_frame, (size_t)cols * 3 * bps, src.data, src.step, (size_t)x_split * 3 * bps, rows, hipMemcpyHostToDevice, stream));
    impl->h_slots.resize(stepTotal);
    for (int k = 0; k < npart; ++k) {
        const int t0 = first_of[k];
        const size_t n = (k + 1 < npart ? slots_off[k + 1] : stepTotal) - slots_off[k];
        for (size_t st = 0; st < n; ++st) {
            const int ti = (int)(st / steps), aug = (int)(st % steps);
            TileSlot sl{0, 0, aug, 0};
            if (ti < tiles_of[k]) { sl.x = grid.in[sp.first_tile + t0 + ti].x; sl.y = grid.in[sp.first_tile + t0 + ti].y; sl.valid = 1; }
            impl->h_slots[slots_off[k] + st] = sl;
        }
    }
    impl->ensure(impl->d_slots, impl->slots_cap, stepTotal * sizeof(TileSlot));
    hipAssert(hipMemcpyAsync(impl->d_slots, impl->h_slots.data(), stepTotal * sizeof(TileSlot), hipMemcpyHostToDevice, stream));
    // the slab holds the frame's tiles in tile order: the last part's passes end at most a pass beyond them (and the frame as ONE part - what
    // benchResident / profileFrame replay - ends at most a pass beyond its tiles: sized for both now, an allocation later would drop the captured passes)
    const size_t one_part_steps = (size_t)(((int)std::lround(std::ceil((double)(sp.tile_count * steps) / plan.userB)) + S - 1) / S) * B;
    impl->ensure(impl->d_slab, impl->slab_cap, std::max((size_t)first_of[npart - 1] * steps + (stepTotal - slots_off[npart - 1]), one_part_steps) * plan.Tout * plan.Tout * 4 * plan.elt);

    hipAssert(hipEventRecord(impl->ev0, stream));
    if (npart == 1) {
        impl->run_frame(rows, cols, grid, true, sp);
        hipAssert(hipEventRecord(impl->ev1, stream));
        // :344 download ; the reference leaves the sync commented out (:345, quirk Q10) - we wait before handing dst back
        hipAssert(hipMemcpy2DAsync(dst.data + (size_t)sp.x0 * 3 * bps, dst.step, impl->d_out + (size_t)sp.x0 * 3 * bps, (size_t)dst.cols * 3 * bps, (size_t)(sp.x1 - sp.x0) * 3 * bps, dst.rows, hipMemcpyDeviceToHost, stream));
        hipAssert(hipStreamSynchronize(stream));
    } else {
        impl->ensure_copy_streams();
        for (int k = 0; k < npart; ++k) if (!impl->ev_part[k]) hipAssert(hipEventCreateWithFlags(&impl->ev_part[k], hipEventDisableTiming));
        ShardPlan part_plan[kMaxParts];
        for (int k = 0; k < npart; ++k) {
            // the cells of the part's tiles: shard_plan()'s rectangles for these (batch-aligned) boundaries
            ShardPlan q; q.first_tile = first_of[k]; q.tile_count = tiles_of[k];
            const int sx = plan.Tout - grid.outOvX, sy = plan.Tout - grid.outOvY;
            auto x_of = [&](int i) { return i >= grid.nx ? cols * s : i * sx; };
            auto y_of = [&](int j) { return j >= grid.ny ? rows * s : j * sy; };
            auto add = [&](int i0, int i1, int j0, int j1) { if (i0 < i1 && j0 < j1) q.rect[q.nrect++] = Rect{x_of(i0), y_of(j0), x_of(i1) - x_of(i0), y_of(j1) - y_of(j0)}; };
            const int t0 = q.first_tile, t1 = t0 + q.tile_count;
            const int c0 = t0 / grid.ny, r0 = t0 % grid.ny, c1 = t1 / grid.ny, r1 = t1 % grid.ny;
            if (c0 == c1) add(c0, c0 + 1, r0, r1);
            else { int full0 = c0; if (r0) { add(c0, c0 + 1, r0, grid.ny); full0 = c0 + 1; } add(full0, c1, 0, grid.ny); if (r1) add(c1, c1 + 1, 0, r1); }
            part_plan[k] = q;
        }
        hipStream_t dn = impl->s_up;      // (which of the two copy streams: below)
        auto download_part = [&](int k) {
            hipAssert(hipStreamWaitEvent(dn, impl->ev_part[k], 0));
            for (int r = 0; r < part_plan[k].nrect; ++r) {
                const Rect& rc = part_plan[k].rect[r];
                hipAssert(hipMemcpy2DAsync(dst.data + (size_t)rc.y * dst.step + (size_t)rc.x * 3 * bps, dst.step, impl->d_out + ((size_t)rc.y * dst.cols + rc.x) * 3 * bps, (size_t)dst.cols * 3 * bps,
                                           (size_t)rc.w * 3 * bps, rc.h, hipMemcpyDeviceToHost, dn));
            }
        };
        // the parts ROLL like the frames of a sequence (run_rolling_frame) when all their passes split: no join at the end of a part - the second group's stream
        // composes the part's cells behind its own tiles while the first stream starts the next part's (one slab: the parts' tiles lie side by side in it)
        bool roll = true;
        for (int k = 0; k < npart; ++k) roll = roll && impl->can_roll(tiles_of[k]);
        struct Unroll { Impl* e; ~Unroll() { e->rolling = false; } } unroll{impl.get()};
        if (roll) for (int k = 0; k < 2; ++k) if (!impl->ev_g0[k]) hipAssert(hipEventCreateWithFlags(&impl->ev_g0[k], hipEventDisableTiming));
        for (int k = 0; k < npart; ++k) {
            if (k == 1 && x_split < cols) hipAssert(hipStreamWaitEvent(stream, impl->ev_up[0], 0));      // the rest of the frame has arrived
            impl->rolling = roll;
            impl->run_passes(rows, cols, tiles_of[k], (size_t)first_of[k] * steps, true, slots_off[k], k == 0, batches_before[k], batch_total);
            impl->rolling = false;
            if (k == 0 && x_split < cols) {      // issued behind the first part's launches: from pageable memory the call returns when the copy is done
                hipAssert(hipMemcpy2DAsync(impl->d_frame + (size_t)x_split * 3 * bps, (size_t)cols * 3 * bps, src.data + (size_t)x_split * 3 * bps, src.step, (size_t)(cols - x_split) * 3 * bps, rows,
                                           hipMemcpyHostToDevice, impl->s_dn));
                hipAssert(hipEventRecord(impl->ev_up[0], impl->s_dn));
            }
            hipStream_t cs = stream;
            if (roll) {
                cs = impl->gstream[0];
                hipAssert(hipEventRecord(impl->ev_g0[k & 1], stream));
                hipAssert(hipStreamWaitEvent(cs, impl->ev_g0[k & 1], 0));
            }
            for (int r = 0; r < part_plan[k].nrect; ++r) {
                const Rect& rc = part_plan[k].rect[r];
                impl->compose_rect(rows, cols, grid, rc.x, rc.x + rc.w, rc.y, rc.y + rc.h, 0, cs);
            }
            hipAssert(hipEventRecord(impl->ev_part[k], cs));
            if (k + 1 == npart) { if (roll) impl->end_rolling(); hipAssert(hipEventRecord(impl->ev1, stream)); }
        }
        // the downloads, in order, on a copy stream: a part's cells travel while the next part computes.  Which of the two copy streams: the one created
        // FIRST (renderSequence()'s upload stream, idle here) - the runtime spreads streams over its hardware queues in creation order, and from
        // pageable memory the download only ran beside the kernels on that one (9.1 against 9.9 ms per call, profiles/r4_kernels/render_parts_ab*.txt)
        for (int k = 0; k < npart; ++k) download_part(k);
        hipAssert(hipStreamSynchronize(dn));
        hipAssert(hipStreamSynchronize(stream));
    }
    hipAssert(hipEventElapsedTime(&impl->last_ms, impl->ev0, impl->ev1));
    impl->last_rows = rows; impl->last_cols = cols; impl->last_grid = grid; impl->last_strip = sp;
    impl->one_part_stale = npart > 1;     // (benchResident / profileFrame replay the frame as ONE part and rebuild the slot table when they are called: one_part_slots)
    // the second slab of a rolling sequence now (an allocation drops the captured passes: better here, on the frame size's first sight, than inside benchResident / renderSequence)
    if (parts == 1 && impl->slab2_cap < impl->slab_cap && impl->can_roll(sp.tile_count)) impl->ensure(impl->d_slab2, impl->slab2_cap, impl->slab_cap);
    return true;
} catch (const std::exception& e) {
    // a frame in parts copies to and from the caller's buffers on the side streams and may have left the second group's stream un-joined: let everything
    // drain before the caller gets its buffers back (errors here are ignored, the first one is what gets reported)
    impl->drain_after_error();
    W2X_LOG_AS(who, error, "Render failed unexpectedly: " + std::string(e.what()) + ".");
    return false;
}

// ONE frame over several engines, every tile computed once (img2img.h; SURVEY 8e second option; the reference is single-device,
// main.cpp:70-74).  Engine k takes part k of tiles.cpp shard_plan(): a contiguous range of the reference's column-major tile order
// (img2img_render.cpp:43-44) and the canvas cells of those tiles.  Phase 1, every engine on its own stream: upload, gather + network
// passes of its own tiles into its slab, an event.  Phase 2, per engine: behind the events of the engines that computed the ny + 1
// tiles in front of its range, copy their blend bands into the slab slots in front of its own (the seam exchange: two bands per tile,
// all four per augmented tile under TTA - a dihedral map sends the right / bottom band to any of the four), compose its cells (the
// same kernel and the same ascending-tile order as a whole-frame render, img2img_render.cpp:329-330: identical bytes), download them.
// The calling thread drives all engines; nothing here is a collective.
// what a part needs to know about the part(s) in front of it: where their slab is and how it is laid out
struct ShardPeer { const void* slab = nullptr; size_t halo_slots = 0; int device = -1; hipEvent_t ev = nullptr; int first_tile = 0, tile_count = 0; };

// phase 1 of a sharded frame on this engine: upload, slots of its own tiles, their network passes into the slab behind the slots reserved for the
// preceding parts' bands; records ev0 and ev_shard on the compute stream (no host synchronisation)
static void shard_phase1(Img2Img::Impl& e, const Image& src, const TileGrid& grid, const ShardPlan& sp, bool report) {
    const int rows = src.rows, cols = src.cols, s = e.cfg.scaling, steps = e.cfg.tta ? 8 : 1;
    const size_t slot_bytes = (size_t)e.plan.Tout * e.plan.Tout * 4 * e.plan.elt;
    e.deep = false;
    e.ensure(e.d_frame, e.frame_cap, (size_t)rows * cols * 3);
    e.ensure(e.d_out, e.out_cap, (size_t)rows * s * cols * s * 3);
    hipAssert(hipMemcpy2DAsync(e.d_frame, (size_t)cols * 3, src.data, src.step, (size_t)cols * 3, rows, hipMemcpyHostToDevice, e.stream));
    const int B = e.plan.B, S = e.plan.B / e.plan.userB;
    const int batchCount = (int)std::lround(std::ceil((double)(sp.tile_count * steps) / e.plan.userB));
    const int stepCount = ((batchCount + S - 1) / S) * B;
    e.h_slots.resize(stepCount);
    for (int st = 0; st < stepCount; ++st) {
        const int ti = st / steps, aug = st % steps;
        TileSlot sl{0, 0, aug, 0};
        if (ti < sp.tile_count) { sl.x = grid.in[sp.first_tile + ti].x; sl.y = grid.in[sp.first_tile + ti].y; sl.valid = 1; }
        e.h_slots[st] = sl;
    }
    e.ensure(e.d_slots, e.slots_cap, (size_t)stepCount * sizeof(TileSlot));
    hipAssert(hipMemcpyAsync(e.d_slots, e.h_slots.data(), (size_t)stepCount * sizeof(TileSlot), hipMemcpyHostToDevice, e.stream));
    e.shard_halo_slots = (size_t)(sp.first_tile - sp.halo_first) * steps;
    e.ensure(e.d_slab, e.slab_cap, (e.shard_halo_slots + (size_t)stepCount) * slot_bytes);
    hipAssert(hipEventRecord(e.ev0, e.stream));
    e.run_passes(rows, cols, sp.tile_count, e.shard_halo_slots, report, 0, true);
    if (!e.ev_shard) hipAssert(hipEventCreateWithFlags(&e.ev_shard, hipEventDisableTiming));
    hipAssert(hipEventRecord(e.ev_shard, e.stream));
}

// phase 2 on this engine (part k): the seam exchange - behind the owners' events (where there are events: engines of this process; slabs of other
// processes are complete when their handles arrive) copy the blend bands of the tiles [halo_first, first_tile) out of the owners' slabs - then compose and
// download this part's rectangles (asynchronously: the caller synchronises the stream)
static void shard_phase2(Img2Img::Impl& e, int k, const std::vector<ShardPlan>& sp, const std::vector<ShardPeer>& peers, const TileGrid& grid, int rows, int cols, Image& dst) {
    const int steps = e.cfg.tta ? 8 : 1, To = e.plan.Tout;
    const size_t px = (size_t)4 * e.plan.elt, slot_bytes = (size_t)To * To * px;
    const bool overlapping = e.cfg.overlapX != 0 || e.cfg.overlapY != 0;
    int waited = -1;
    bool peer2d = true;                                                    // (of the owner last waited for)
    for (int g = sp[k].halo_first; g < sp[k].first_tile && overlapping; ++g) {
        int q = k - 1;
        while (q >= 0 && !(sp[q].tile_count > 0 && g >= sp[q].first_tile && g < sp[q].first_tile + sp[q].tile_count)) --q;
        if (q < 0 || !peers[q].slab) throw std::runtime_error("shard plan: tile " + std::to_string(g) + " has no owner");
        const ShardPeer& o = peers[q];
        if (q != waited) {                                                 // once per owner: its event, and how its bands can travel
            if (o.ev) hipAssert(hipStreamWaitEvent(e.stream, o.ev, 0));
            waited = q;
            peer2d = o.device == e.device;
            if (!peer2d) {      // strided band copies between devices need peer access; without it whole slots travel by hipMemcpyPeerAsync
                int can = 0;
                if (o.device >= 0 && hipDeviceCanAccessPeer(&can, e.device, o.device) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(o.device, 0);
                    peer2d = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
                }
                (void)hipGetLastError();
            }
        }
        for (int a = 0; a < steps; ++a) {
            const uint8_t* from = (const uint8_t*)o.slab + (o.halo_slots + (size_t)(g - sp[q].first_tile) * steps + a) * slot_bytes;
            uint8_t* to = (uint8_t*)e.d_slab + ((size_t)(g - sp[k].halo_first) * steps + a) * slot_bytes;
            if (!peer2d) {
                if (o.device >= 0) hipAssert(hipMemcpyPeerAsync(to, e.device, from, o.device, slot_bytes, e.stream));
                else hipAssert(hipMemcpyAsync(to, from, slot_bytes, hipMemcpyDefault, e.stream));
                continue;
            }
            int bx = std::min(To, grid.outOvX), by = std::min(To, grid.outOvY);
            if (steps > 1) bx = by = std::max(bx, by);                     // a rotation turns a column band into a row band of the same width
            const size_t pitch = (size_t)To * px;
            auto cols_band = [&](int x0) { if (bx > 0) hipAssert(hipMemcpy2DAsync(to + (size_t)x0 * px, pitch, from + (size_t)x0 * px, pitch, (size_t)bx * px, To, hipMemcpyDeviceToDevice, e.stream)); };
            auto rows_band = [&](int y0) { if (by > 0) hipAssert(hipMemcpyAsync(to + (size_t)y0 * pitch, from + (size_t)y0 * pitch, (size_t)by * pitch, hipMemcpyDeviceToDevice, e.stream)); };
            cols_band(To - bx); rows_band(To - by);                        // what the cells to the right of / below the tile read
            if (steps > 1) { cols_band(0); rows_band(0); }                // an augmented tile is read through its inverse dihedral map
        }
    }
    for (int r = 0; r < sp[k].nrect; ++r) {
        const Rect& rc = sp[k].rect[r];
        e.compose_rect(rows, cols, grid, rc.x, rc.x + rc.w, rc.y, rc.y + rc.h, sp[k].halo_first);
    }
    hipAssert(hipEventRecord(e.ev1, e.stream));
    for (int r = 0; r < sp[k].nrect; ++r) {
        const Rect& rc = sp[k].rect[r];
        hipAssert(hipMemcpy2DAsync(dst.data + (size_t)rc.y * dst.step + (size_t)rc.x * 3, dst.step, e.d_out + ((size_t)rc.y * dst.cols + rc.x) * 3, (size_t)dst.cols * 3,
                                   (size_t)rc.w * 3, rc.h, hipMemcpyDeviceToHost, e.stream));
    }
}

// frame / output checks shared by the sharded entries; "" when fine
static std::string shard_frame_problem(const Img2Img::Impl& e, const Image& src, const Image* dst) {
    const int s = e.cfg.scaling;
    if (src.depth != 8 || (dst && dst->depth != 8)) return "sharded rendering takes 8-bit frames.";
    if (!src.data || src.rows <= 0 || src.cols <= 0 || src.step < (size_t)src.cols * 3) return "Input image is empty or has an invalid step.";
    if (dst && (!dst->data || dst->rows != src.rows * s || dst->cols != src.cols * s || dst->step < (size_t)dst->cols * 3))
        return "Output image has invalid size: expected " + std::to_string(src.cols * s) + "x" + std::to_string(src.rows * s) + ".";
    return "";
}

bool Img2Img::renderSharded(Img2Img* const* engines, int count, const Image& src, Image& dst) {
    if (!engines || count <= 0 || !engines[0]) return false;
    Img2Img* const first = engines[0];
    Impl* const impl = first->impl.get();                      // (the log macros name `impl`)
    try {
        for (int k = 0; k < count; ++k) {
            if (!engines[k] || !engines[k]->impl->loaded) { W2X_LOG(error, "Render called before a successful load (engine " + std::to_string(k) + ")."); return false; }
            const Impl& a = *engines[k]->impl;
            const RenderConfig &c = a.cfg, &c0 = impl->cfg;
            if (a.plan.T != impl->plan.T || a.plan.Tout != impl->plan.Tout || a.plan.elt != impl->plan.elt || c.scaling != c0.scaling || c.overlapX != c0.overlapX ||
                c.overlapY != c0.overlapY || c.tta != c0.tta || c.ttaBugCompat != c0.ttaBugCompat || c.batchSize != c0.batchSize) {
                W2X_LOG(error, "renderSharded: engine " + std::to_string(k) + " was loaded with another model or configuration."); return false;
            }
            for (int j = 0; j < k; ++j) if (engines[j] == engines[k]) { W2X_LOG(error, "renderSharded: the same engine twice."); return false; }
        }
        const RenderConfig& cfg = impl->cfg;
        const Plan& plan = impl->plan;
        const int rows = src.rows, cols = src.cols, s = cfg.scaling;
        if (const std::string why = shard_frame_problem(*impl, src, &dst); !why.empty()) { W2X_LOG(error, why); return false; }
        const TileGrid grid = calculate_tiles(cols, rows, cols * s, rows * s, plan.T, plan.T, plan.Tout, plan.Tout, s, cfg.overlapX, cfg.overlapY);
        if (grid.count <= 0) { W2X_LOG(error, "Tile grid is empty."); return false; }
        for (const Rect& r : grid.out) if (r.w <= 0 || r.h <= 0) { W2X_LOG(error, "Tile grid does not fit the output (scaling does not match the model)."); return false; }
        std::vector<ShardPlan> sp(count);
        for (int k = 0; k < count; ++k) sp[k] = shard_plan(grid, cols * s, rows * s, plan.Tout, plan.Tout, k, count);
        {   // every tile must have an owner: shard_plan() hands out nothing when the blend bands are as wide as the tile stride
            int owned = 0;
            for (int k = 0; k < count; ++k) owned += sp[k].tile_count;
            if (owned != grid.count) { W2X_LOG(error, "renderSharded: the blend bands are wider than the tile stride; use render or renderStrip."); return false; }
        }
        // ---- phase 1: every engine computes its own tiles
        for (int k = 0; k < count; ++k) {
            Impl& e = *engines[k]->impl;
            if (sp[k].tile_count == 0) continue;
            DeviceGuard guard(e.device);
            shard_phase1(e, src, grid, sp[k], k == 0);
        }
        // ---- phase 2: seam exchange, compose, download
        std::vector<ShardPeer> peers(count);
        for (int k = 0; k < count; ++k) {
            const Impl& e = *engines[k]->impl;
            if (sp[k].tile_count) peers[k] = ShardPeer{e.d_slab, e.shard_halo_slots, e.device, e.ev_shard, sp[k].first_tile, sp[k].tile_count};
        }
        for (int k = 0; k < count; ++k) {
            Impl& e = *engines[k]->impl;
            if (sp[k].tile_count == 0) continue;
            DeviceGuard guard(e.device);
            shard_phase2(e, k, sp, peers, grid, rows, cols, dst);
        }
        for (int k = 0; k < count; ++k) {
            Impl& e = *engines[k]->impl;
            if (sp[k].tile_count == 0) continue;
            DeviceGuard guard(e.device);
            hipAssert(hipStreamSynchronize(e.stream));
            hipAssert(hipEventElapsedTime(&e.last_ms, e.ev0, e.ev1));
        }
        return true;
    } catch (const std::exception& ex) {
        for (int k = 0; k < count; ++k) if (engines[k] && engines[k]->impl->stream) { try { DeviceGuard guard(engines[k]->impl->device); (void)hipStreamSynchronize(engines[k]->impl->stream); } catch (...) {} }
        W2X_LOG(error, "Render failed unexpectedly: " + std::string(ex.what()) + ".");
        return false;
    }
}

// The same two phases for ONE PROCESS PER GPU (the launch contract of bench.py / torch.distributed.run): rank r calls shardCompute(src, r, N), publishes
// its slab (shardSlab: a device pointer the caller exports with w2x_ipc_export and its peers open with w2x_ipc_open), and after every rank has done so
// (a host-side barrier - the handle exchange itself) shardFinish(dst, r, N, slabs) with the opened pointers of the parts in front of it.  No collective on
// the data path: the seam bands are device-to-device copies out of the neighbours' slabs.
bool Img2Img::shardCompute(const Image& src, int part, int parts) try {
    if (!impl->loaded) { W2X_LOG(error, "Render called before a successful load."); return false; }
    if (parts <= 0 || part < 0 || part >= parts) { W2X_LOG(error, "Invalid part index."); return false; }
    DeviceGuard guard(impl->device);
    if (const std::string why = shard_frame_problem(*impl, src, nullptr); !why.empty()) { W2X_LOG(error, why); return false; }
    const Plan& plan = impl->plan;
    const int rows = src.rows, cols = src.cols, s = impl->cfg.scaling;
    const TileGrid grid = calculate_tiles(cols, rows, cols * s, rows * s, plan.T, plan.T, plan.Tout, plan.Tout, s, impl->cfg.overlapX, impl->cfg.overlapY);
    if (grid.count <= 0) { W2X_LOG(error, "Tile grid is empty."); return false; }
    const ShardPlan sp = shard_plan(grid, cols * s, rows * s, plan.Tout, plan.Tout, part, parts);
    impl->shard_rows = rows; impl->shard_cols = cols;
    {   // every tile must have an owner (shard_plan() hands out nothing when the blend bands are as wide as the tile stride)
        int owned = 0;
        for (int k = 0; k < parts; ++k) owned += shard_plan(grid, cols * s, rows * s, plan.Tout, plan.Tout, k, parts).tile_count;
        if (owned != grid.count) { W2X_LOG(error, "the blend bands are wider than the tile stride; use render or renderStrip."); return false; }
    }
    if (sp.tile_count == 0) return true;                       // more parts than tiles: this one has no share
    shard_phase1(*impl, src, grid, sp, true);
    hipAssert(hipStreamSynchronize(impl->stream));           // the slab is complete when this returns: what the peers copy from after the barrier
    return true;
} catch (const std::exception& e) {
    W2X_LOG(error, "Render failed unexpectedly: " + std::string(e.what()) + ".");
    return false;
}

const void* Img2Img::shardSlab(size_t* bytes) const { if (bytes) *bytes = impl->slab_cap; return impl->d_slab; }

bool Img2Img::shardFinish(Image& dst, int part, int parts, const void* const* slabs, const int* devices) try {
    if (!impl->loaded || impl->shard_rows <= 0) { W2X_LOG(error, "shardFinish without a shardCompute."); return false; }
    if (parts <= 0 || part < 0 || part >= parts || !slabs) { W2X_LOG(error, "Invalid part index."); return false; }
    DeviceGuard guard(impl->device);
    const Plan& plan = impl->plan;
    const int rows = impl->shard_rows, cols = impl->shard_cols, s = impl->cfg.scaling, steps = impl->cfg.tta ? 8 : 1;
    Image probe; probe.data = (uint8_t*)1; probe.rows = rows; probe.cols = cols; probe.step = (size_t)cols * 3;
    if (const std::string why = shard_frame_problem(*impl, probe, &dst); !why.empty()) { W2X_LOG(error, why); return false; }
    const TileGrid grid = calculate_tiles(cols, rows, cols * s, rows * s, plan.T, plan.T, plan.Tout, plan.Tout, s, impl->cfg.overlapX, impl->cfg.overlapY);
    std::vector<ShardPlan> sp(parts);
    std::vector<ShardPeer> peers(parts);
    for (int k = 0; k < parts; ++k) {
        sp[k] = shard_plan(grid, cols * s, rows * s, plan.Tout, plan.Tout, k, parts);
        if (sp[k].tile_count) peers[k] = ShardPeer{k == part ? impl->d_slab : slabs[k], (size_t)(sp[k].first_tile - sp[k].halo_first) * steps, k == part ? impl->device : devices ? physical_device(devices[k]) : -1,
                                                   nullptr, sp[k].first_tile, sp[k].tile_count};
    }
    if (sp[part].tile_count == 0) return true;
    shard_phase2(*impl, part, sp, peers, grid, rows, cols, dst);
    hipAssert(hipStreamSynchronize(impl->stream));
    hipAssert(hipEventElapsedTime(&impl->last_ms, impl->ev0, impl->ev1));
    return true;
} catch (const std::exception& e) {
    if (impl->stream) (void)hipStreamSynchronize(impl->stream);
    W2X_LOG(error, "Render failed unexpectedly: " + std::string(e.what()) + ".");
    return false;
}

// device memory of another process: export / open / close (hipIpc*); the opened pointer is valid in this process on the current device
bool ipc_export(const void* device_ptr, uint8_t out[64]) {
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "handle size");
    hipIpcMemHandle_t h;
    if (!device_ptr || hipIpcGetMemHandle(&h, const_cast<void*>(device_ptr)) != hipSuccess) { (void)hipGetLastError(); return false; }
    memset(out, 0, 64); memcpy(out, &h, sizeof(h));
    return true;
}
void* ipc_open(const uint8_t handle[64], int deviceId) {
    const int dev = physical_device(deviceId);
    if (!device_ordinal_problem(deviceId, dev).empty()) return nullptr;
    DeviceGuard guard(dev);
    hipIpcMemHandle_t h; memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void ipc_close(void* p) { if (p && hipIpcCloseMemHandle(p) != hipSuccess) (void)hipGetLastError(); }

// A sequence of equally sized frames (a video, main.cpp:263-269) with the PCIe copies taken off the critical path: frame i+1 is
// uploaded and frame i-1 downloaded on two copy streams while frame i runs on the compute stream (two device frame / output
// buffers, events between the three streams).  Each frame is the same gather -> network -> compose as render(), so outputs are
// bit-identical; the progress callback is not raised per batch here.  Copies only overlap when the host buffers are page-locked
// (pinHost); with pageable memory the call is still correct, the runtime just serialises the copies.
bool Img2Img::renderSequence(const Image* srcs, Image* dsts, int count) try {
    if (!impl->loaded) { W2X_LOG(error, "Render called before a successful load."); return false; }
    if (count <= 0) return true;
    DeviceGuard guard(impl->device);
    const RenderConfig& cfg = impl->cfg;
    const Plan& plan = impl->plan;
    const int rows = srcs[0].rows, cols = srcs[0].cols, s = cfg.scaling;
    for (int i = 0; i < count; ++i) if (srcs[i].depth != 8 || dsts[i].depth != 8) { W2X_LOG(error, "renderSequence takes 8-bit frames (16-bit images go through render())."); return false; }
    impl->deep = false;
    for (int i = 0; i < count; ++i) {
        if (!srcs[i].data || srcs[i].rows != rows || srcs[i].cols != cols || srcs[i].step < (size_t)cols * 3 || rows <= 0 || cols <= 0) { W2X_LOG(error, "Input images must be non-empty and of one size."); return false; }
        if (!dsts[i].data || dsts[i].rows != rows * s || dsts[i].cols != cols * s || dsts[i].step < (size_t)cols * s * 3) { W2X_LOG(error, "Output image has invalid size: expected " + std::to_string(cols * s) + "x" + std::to_string(rows * s) + "."); return false; }
    }
    hipStream_t stream = impl->stream;
    impl->ensure_copy_streams();
    const size_t in_bytes = (size_t)rows * cols * 3, out_bytes = in_bytes * s * s;
    impl->ensure(impl->d_frame, impl->frame_cap, in_bytes);   impl->ensure(impl->d_frame2, impl->frame2_cap, in_bytes);
    impl->ensure(impl->d_out, impl->out_cap, out_bytes);      impl->ensure(impl->d_out2, impl->out2_cap, out_bytes);
    TileGrid grid = calculate_tiles(cols, rows, cols * s, rows * s, plan.T, plan.T, plan.Tout, plan.Tout, s, cfg.overlapX, cfg.overlapY);
    if (grid.count <= 0) { W2X_LOG(error, "Tile grid is empty."); return false; }
    for (const Rect& r : grid.out) if (r.w <= 0 || r.h <= 0) { W2X_LOG(error, "Tile grid does not fit the output (scaling does not match the model)."); return false; }
    const StripPlan sp = strip_plan(grid, cols * s, plan.Tout, 0, 1);
    const int steps = cfg.tta ? 8 : 1, B = plan.B, S = plan.B / plan.userB;
    const int batchCount = (int)std::lround(std::ceil((double)(sp.tile_count * steps) / plan.userB));
    const int stepCount = ((batchCount + S - 1) / S) * B;
    impl->h_slots.resize(stepCount);
    for (int st = 0; st < stepCount; ++st) {
        int ti = st / steps, aug = st % steps;
        TileSlot sl{0, 0, aug, 0};
        if (ti < sp.tile_count) { sl.x = grid.in[ti].x; sl.y = grid.in[ti].y; sl.valid = 1; }
        impl->h_slots[st] = sl;
    }
    impl->ensure(impl->d_slots, impl->slots_cap, (size_t)stepCount * sizeof(TileSlot));
    hipAssert(hipMemcpyAsync(impl->d_slots, impl->h_slots.data(), (size_t)stepCount * sizeof(TileSlot), hipMemcpyHostToDevice, stream));
    impl->ensure(impl->d_slab, impl->slab_cap, (size_t)stepCount * plan.Tout * plan.Tout * 4 * plan.elt);
    hipAssert(hipStreamSynchronize(stream));

    uint8_t* const frames[2] = {impl->d_frame, impl->d_frame2};
    uint8_t* const outs[2] = {impl->d_out, impl->d_out2};
    struct Restore { Impl* im; uint8_t* f; uint8_t* o; ~Restore() { im->d_frame = f; im->d_out = o; } } restore{impl.get(), frames[0], outs[0]};   // also on exceptions
    // frame f is composed on the second group's stream while frame f + 1's first group already runs (run_rolling_frame) when every pass of a frame splits
    const bool roll = count > 1 && impl->can_roll(sp.tile_count);
    if (roll) impl->ensure(impl->d_slab2, impl->slab2_cap, impl->slab_cap);
    hipAssert(hipStreamSynchronize(stream));
    hipAssert(hipEventRecord(impl->ev0, stream));
    for (int i = 0; i < count; ++i) {
        const int b = i & 1;
        if (i >= 2) hipAssert(hipStreamWaitEvent(impl->s_up, impl->ev_comp[b], 0));       // frame i-2 has been gathered out of this buffer
        hipAssert(hipMemcpy2DAsync(frames[b], (size_t)cols * 3, srcs[i].data, srcs[i].step, (size_t)cols * 3, rows, hipMemcpyHostToDevice, impl->s_up));
        hipAssert(hipEventRecord(impl->ev_up[b], impl->s_up));
        hipAssert(hipStreamWaitEvent(stream, impl->ev_up[b], 0));
        impl->d_frame = frames[b]; impl->d_out = outs[b];
        if (roll) {
            impl->run_rolling_frame(rows, cols, grid, sp, b, i >= 2 ? impl->ev_dn[b] : nullptr);
            hipAssert(hipEventRecord(impl->ev_comp[b], impl->gstream[0]));              // (behind the compose launch: both groups have gathered, the output is whole)
        } else {
            if (i >= 2) hipAssert(hipStreamWaitEvent(stream, impl->ev_dn[b], 0));         // frame i-2 has left this output buffer
            impl->run_frame(rows, cols, grid, false, sp);
            hipAssert(hipEventRecord(impl->ev_comp[b], stream));
        }
        hipAssert(hipStreamWaitEvent(impl->s_dn, impl->ev_comp[b], 0));
        hipAssert(hipMemcpy2DAsync(dsts[i].data, dsts[i].step, outs[b], (size_t)cols * s * 3, (size_t)cols * s * 3, rows * s, hipMemcpyDeviceToHost, impl->s_dn));
        hipAssert(hipEventRecord(impl->ev_dn[b], impl->s_dn));
    }
    if (roll) impl->end_rolling();
    hipAssert(hipEventRecord(impl->ev1, stream));
    hipAssert(hipStreamSynchronize(impl->s_dn));
    hipAssert(hipStreamSynchronize(stream));
    hipAssert(hipStreamSynchronize(impl->s_up));
    float total_ms = 0.f;
    hipAssert(hipEventElapsedTime(&total_ms, impl->ev0, impl->ev1));
    impl->last_ms = total_ms / count;        // lastRenderMs(): compute-stream time per frame of the sequence
    impl->last_rows = rows; impl->last_cols = cols; impl->last_grid = grid; impl->last_strip = sp;
    return true;
} catch (const std::exception& e) {
    // copies on the side streams may still be reading or writing the caller's buffers, and a rolling sequence composes on the second group's stream
    // (end_rolling() has not run when an exception fires): let all four drain before the caller gets its buffers back
    impl->drain_after_error();
    W2X_LOG(error, "Render failed unexpectedly: " + std::string(e.what()) + ".");
    return false;
}

// Page-locked frame buffers for renderSequence(): owned by the engine, freed by freeHost() or with the engine.
void* Img2Img::allocHost(size_t bytes) try {
    if (!bytes) return nullptr;
    std::unique_ptr<DeviceGuard> guard;
    if (impl->device >= 0) guard.reset(new DeviceGuard(impl->device));
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    impl->host_allocs.push_back(p);
    return p;
} catch (const std::exception&) {
    return nullptr;
}
void Img2Img::freeHost(void* data) try {
    auto it = std::find(impl->host_allocs.begin(), impl->host_allocs.end(), data);
    if (it == impl->host_allocs.end()) return;
    std::unique_ptr<DeviceGuard> guard;
    if (impl->device >= 0) guard.reset(new DeviceGuard(impl->device));
    if (hipHostFree(data) != hipSuccess) (void)hipGetLastError();
    impl->host_allocs.erase(it);
} catch (const std::exception&) {
}

// Page-lock a caller-owned frame buffer in place so that renderSequence() can copy it by DMA while kernels run (whole pages only).
bool Img2Img::pinHost(void* data, size_t bytes) try {
    if (!data || !bytes) return false;
    if (((size_t)data | bytes) & 4095) { W2X_LOG(warn, "pinHost: only whole pages can be page-locked in place (use allocHost)."); return false; }
    DeviceGuard guard(impl->device);
    if (hipHostRegister(data, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
    impl->pinned.push_back(data);
    return true;
} catch (const std::exception&) {
    return false;
}
void Img2Img::unpinHost(void* data) try {
    DeviceGuard guard(impl->device);
    auto it = std::find(impl->pinned.begin(), impl->pinned.end(), data);
    if (it == impl->pinned.end()) return;
    if (hipHostUnregister(data) != hipSuccess) (void)hipGetLastError();
    impl->pinned.erase(it);
} catch (const std::exception&) {
}

bool Img2Img::infer(const float* input, float* output) try {
    if (!impl->loaded) { W2X_LOG(error, "Infer called before a successful load."); return false; }
    DeviceGuard guard(impl->device);
    const Plan& plan = impl->plan;
    const size_t in_elems = (size_t)plan.B * 3 * plan.T * plan.T, out_elems = (size_t)plan.B * 3 * plan.Tout * plan.Tout;
    const size_t user_in = (size_t)plan.userB * 3 * plan.T * plan.T, user_out = (size_t)plan.userB * 3 * plan.Tout * plan.Tout;
    hipStream_t stream = impl->stream;
    if (!impl->d_blob_in) {
        hipAssert(hipMalloc((void**)&impl->d_blob_in, in_elems * sizeof(float)));      // img2img_load.cpp:228-232: f32 IO buffers
        hipAssert(hipMalloc((void**)&impl->d_blob_out, out_elems * sizeof(float)));
        hipAssert(hipMemsetAsync(impl->d_blob_in, 0, in_elems * sizeof(float), stream));   // slots beyond the caller's batch stay zero
    }
    hipAssert(hipMemcpyAsync(impl->d_blob_in, input, user_in * sizeof(float), hipMemcpyHostToDevice, stream));
    hipAssert(launch_blob_to_nhwc(impl->d_blob_in, impl->tensors[plan.in_tensor], plan.B, plan.T, plan.elt == 4, stream));
    impl->run_network(nullptr, plan.userB);                                            // img2img_infer.cpp:80 (the slots beyond the caller's batch are not computed)
    hipAssert(launch_nhwc_to_blob(impl->tensors[plan.out_tensor], impl->d_blob_out, plan.B, plan.Tout, plan.elt == 4, stream));
    hipAssert(hipMemcpyAsync(output, impl->d_blob_out, user_out * sizeof(float), hipMemcpyDeviceToHost, stream));
    hipAssert(hipStreamSynchronize(stream));
    return true;
} catch (const std::exception& e) {
    W2X_LOG(error, "Engine inference failed unexpectedly: " + std::string(e.what()) + ".");
    return false;
}

int Img2Img::outputTileSize() const { return impl->loaded ? impl->plan.Tout : 0; }
int Img2Img::opTimes(double* out, int cap) const { int n = (int)impl->op_ms.size(); for (int i = 0; i < n && i < cap; ++i) out[i] = impl->op_ms[i]; return n; }
int Img2Img::scaling() const { return impl->loaded ? impl->cfg.scaling : 0; }
int Img2Img::passTiles() const { return impl->loaded ? impl->plan.B : 0; }
double Img2Img::planFlops() const { return impl->loaded ? impl->plan.flops : 0.0; }
float Img2Img::lastRenderMs() const { return impl->last_ms; }

float Img2Img::benchResident(int iters) try {
    if (!impl->loaded || impl->last_rows == 0 || iters <= 0) return -1.f;
    DeviceGuard guard(impl->device);
    hipStream_t stream = impl->stream;
    impl->one_part_slots();
    // frames of one size back to back: a rolling sequence where every pass splits (run_rolling_frame), else frame after frame
    const bool roll = iters > 1 && impl->can_roll(impl->last_strip.tile_count);
    if (roll) impl->ensure(impl->d_slab2, impl->slab2_cap, impl->slab_cap);
    hipAssert(hipEventRecord(impl->ev0, stream));
    for (int i = 0; i < iters; ++i) {
        if (roll) impl->run_rolling_frame(impl->last_rows, impl->last_cols, impl->last_grid, impl->last_strip, i & 1, nullptr);
        else impl->run_frame(impl->last_rows, impl->last_cols, impl->last_grid, false, impl->last_strip);
    }
    if (roll) impl->end_rolling();
    hipAssert(hipEventRecord(impl->ev1, stream));
    hipAssert(hipStreamSynchronize(stream));
    float ms = 0.f;
    hipAssert(hipEventElapsedTime(&ms, impl->ev0, impl->ev1));
    return ms / iters;
} catch (const std::exception& e) {
    W2X_LOG(error, "Bench failed unexpectedly: " + std::string(e.what()) + ".");
    return -1.f;
}

// Per-kernel-family device time of one resident frame, measured with HIP events on the compute stream.
// out[5*k + {0,1,2}] = {milliseconds, launches, algorithmic FLOP} for k = 0 gemm, 1 attention, 2 se/scale, 3 gather,
// 4 compose, 5 fused mlp; out[30] = wall ms of the whole frame (first launch start to last launch end).
bool Img2Img::profileFrame(double* out, int cap) try {
    if (!impl->loaded || impl->last_rows == 0 || cap < 31) return false;
    DeviceGuard guard(impl->device);
    for (int i = 0; i < 31; ++i) out[i] = 0;
    impl->one_part_slots();
    impl->profiling = true; impl->stamps.clear();
    impl->run_frame(impl->last_rows, impl->last_cols, impl->last_grid, false, impl->last_strip);
    impl->profiling = false;
    hipAssert(hipStreamSynchronize(impl->stream));
    impl->op_ms.assign(impl->plan.ops.size(), 0.0);
    for (auto& st : impl->stamps) {
        float ms = 0.f;
        hipAssert(hipEventElapsedTime(&ms, st.a, st.b));
        if (st.op >= 0) impl->op_ms[st.op] += ms;
        out[5 * st.kind] += ms; out[5 * st.kind + 1] += 1; out[5 * st.kind + 2] += st.flops;
    }
    if (!impl->stamps.empty()) { float ms = 0.f; hipAssert(hipEventElapsedTime(&ms, impl->stamps.front().a, impl->stamps.back().b)); out[30] = ms; }
    for (auto& st : impl->stamps) { (void)hipEventDestroy(st.a); (void)hipEventDestroy(st.b); }
    impl->stamps.clear();
    return true;
} catch (const std::exception& e) {
    impl->profiling = false;
    W2X_LOG(error, "Profile failed unexpectedly: " + std::string(e.what()) + ".");
    return false;
}

}  // namespace w2x
