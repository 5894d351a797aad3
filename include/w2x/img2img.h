// The engine interface: drop-in for trt::Img2Img (/root/reference/src/tensorrt/img2img.h:14-50) with the same five
// public methods, the same bool + callback error convention (function-try-blocks that log "[func@line] msg" at
// severity `error` and return false: logger.h:8, logger.cpp:19-22), one instance = one GPU = one compute stream.
// cv::Mat is replaced by a (pointer, rows, cols, step) view of interleaved 8-bit BGR pixels.
#ifndef W2X_IMG2IMG_H
#define W2X_IMG2IMG_H

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>

#include "config.h"

namespace w2x {

enum Severity { critical, error, warn, info, debug, trace };             // logger.h:11-18
using MessageCallback = std::function<void(Severity, const std::string&)>;  // logger.h:20
using ProgressCallback = std::function<void(int, int, double)>;             // logger.h:21  (current, total, it/s)

struct Image {          // stand-in for cv::Mat of type CV_8UC3
    uint8_t* data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;    // bytes per row
    // Extension (alpha / 16-bit images are a TODO upstream, README.md:88): depth 16 = CV_16UC3, data points at uint16_t samples (BGR), step stays
    // in bytes.  render() / renderStrip() take 16-bit frames when src and dst agree: u16 -> x * float(1/65535) in, sat(rint(x * 65535)) out.
    int depth = 8;
};

class Img2Img {
public:
    Img2Img();
    virtual ~Img2Img();
    // img2img.h:18 - ONNX path in; writes <stem>_<sha256(cfg)[:16]>.{w2x,json} next to it (img2img_build.cpp:151-161)
    bool build(const std::string& path, const BuildConfig& config);
    // img2img.h:19 - the ONNX path again; the engine is found by scanning its directory (img2img_load.cpp:79-114)
    bool load(const std::string& path, const RenderConfig& config);
    // img2img.h:20 - dst must be rows*scaling x cols*scaling (caller pre-sizes it, main.cpp:234-235)
    bool render(const Image& src, Image& dst);
    // One device's share of a single frame spread over `parts` devices (tile-column strips): composes and writes only
    // the output columns of strip `part` (w2x_strip_plan); identical bytes to render() there.  render() == renderStrip(.., 0, 1).
    bool renderStrip(const Image& src, Image& dst, int part, int parts);
    // ONE frame over `count` engines with every tile computed exactly once (seam exchange): engine k runs the k-th of `count`
    // contiguous ranges of the reference's tile order (w2x_shard_plan), the blend bands of the ny + 1 tiles in front of a range are copied
    // from the engine(s) that computed them (device-to-device on one GPU, hipMemcpyPeerAsync / peer 2-D copies across GPUs, no collective),
    // and every engine composes and downloads the canvas cells of its own tiles.  All engines must be loaded with the same model and
    // RenderConfig, live in this process, and are driven from the calling thread; the bytes are render()'s.  engines[0] reports progress.
    static bool renderSharded(Img2Img* const* engines, int count, const Image& src, Image& dst);
    // The same with ONE PROCESS PER GPU (bench.py / torch.distributed.run ranks): rank r runs shardCompute(src, r, N) - its own tiles into its slab,
    // complete on return -, exports shardSlab() with ipc_export(), and once every rank has published its handle (the caller's barrier: the exchange
    // itself) shardFinish(dst, r, N, slabs, devices) with the ipc_open()ed slabs of the parts in front of it (entries of later / own parts are ignored):
    // seam bands copied device to device, its canvas cells composed and written to dst.
    bool shardCompute(const Image& src, int part, int parts);
    const void* shardSlab(size_t* bytes = nullptr) const;
    bool shardFinish(Image& dst, int part, int parts, const void* const* slabs, const int* devices);
    // A sequence of equally sized frames (the per-frame loop of main.cpp:263-269) with upload, compute and download overlapped on
    // three HIP streams; outputs are the bytes render() gives.  The copies only overlap for page-locked host memory: take the frame
    // buffers from allocHost() (owned by the engine, released by freeHost(), release at destruction at the latest).
    bool renderSequence(const Image* srcs, Image* dsts, int count);
    void* allocHost(size_t bytes);
    void freeHost(void* data);
    // Page-locks caller-owned memory in place.  Only whole pages are accepted (data and bytes multiples of 4096): a registration
    // covers whole pages, and buffers that share a page with other live data (heap blocks, neighbouring registrations) left stale
    // registrations behind on this runtime - a later copy from recycled addresses then aborted the process.
    bool pinHost(void* data, size_t bytes);
    void unpinHost(void* data);
    void setMessageCallback(MessageCallback callback);   // img2img.h:21
    void setProgressCallback(ProgressCallback callback); // img2img.h:22

    // Test hook mirroring the private trt::Img2Img::infer (img2img.h:25, img2img_infer.cpp:41-93):
    // host NCHW f32 blob [B,3,T,T] in [0,1] -> [B,3,T',T'] f32.
    bool infer(const float* input, float* output);
    int outputTileSize() const;
    int scaling() const;   // RenderConfig::scaling of the loaded configuration (0 before load)
    double planFlops() const;
    int passTiles() const;   // tiles carried by one network pass (batchSize x super-batch factor)
    // Steady-state device timing of the last render() (ms), HIP events on the compute stream.
    float lastRenderMs() const;
    // Re-run the device part of the last render() (gather, network, compose; no H2D/D2H) `iters` times and return the
    // average milliseconds per frame - inputs already resident in HBM (bench.py's timed region).
    float benchResident(int iters);
    // Per-kernel-family HIP-event timing of one resident frame (layout documented at the definition).
    bool profileFrame(double* out, int cap);
    int opTimes(double* out, int cap) const;   // ms per plan op of the last profileFrame(); returns the op count

    struct Impl;
private:
    bool renderPart(const Image& src, Image& dst, int part, int parts, const char* who);
    std::unique_ptr<Impl> impl;
};

// Device memory across processes (hipIpc*): export a device pointer of this process as a 64-byte handle; open another process's handle on logical device
// `deviceId` (nullptr on failure); close it again.
bool ipc_export(const void* device_ptr, uint8_t out[64]);
void* ipc_open(const uint8_t handle[64], int deviceId);
void ipc_close(void* p);
// PCI bus id of HIP device `deviceId` (the logical id RenderConfig::deviceId takes, W2X_DEVICE_MAP applied); false if there is no such device
bool device_pci_bus_id(int deviceId, char* buf, size_t cap);

}  // namespace w2x

#endif
