/* C ABI of libw2x.so - the FFI boundary for the hot path.
 *
 * Each entry point names the reference interface it replaces (paths relative to /root/reference/src/tensorrt/):
 *   w2x_create / w2x_destroy      trt::Img2Img::Img2Img / ~Img2Img            img2img.h:16-17, img2img_base.cpp:4-10
 *   w2x_set_message_callback      trt::Img2Img::setMessageCallback            img2img.h:21, logger.h:20
 *   w2x_set_progress_callback     trt::Img2Img::setProgressCallback           img2img.h:22, logger.h:21
 *   w2x_build                     trt::Img2Img::build(path, BuildConfig)      img2img.h:18, img2img_build.cpp:54-173
 *   w2x_load                      trt::Img2Img::load(path, RenderConfig)      img2img.h:19, img2img_load.cpp:117-291
 *   w2x_render                    trt::Img2Img::render(cv::Mat, cv::Mat&)     img2img.h:20, img2img_render.cpp:224-352
 *   w2x_infer                     trt::Img2Img::infer (private)               img2img.h:25, img2img_infer.cpp:41-93
 *   w2x_render_sharded            (extension) one frame over N engines         img2img_render.cpp:43-44, 329-330
 *   w2x_calculate_tiles           calculateTiles (file-static)                img2img_render.cpp:7-66
 *   w2x_tile_weights              createTileWeights (file-static)             img2img_load.cpp:29-52
 * Return convention: 1 = true, 0 = false (after the message callback received the error text), like the reference's
 * bool returns.  Plain pointers and sizes only.
 */
#ifndef W2X_C_API_H
#define W2X_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct w2x_engine w2x_engine;

enum { W2X_PRECISION_TF32 = 0, W2X_PRECISION_FP16 = 1, W2X_PRECISION_FP32 = 2 };  /* config.h:7-10; CLI map main.cpp:76-84; FP32: an addition (include/w2x/config.h) */

typedef struct w2x_build_config {   /* trt::BuildConfig, config.h:12-31 */
    int deviceId, precision;
    int minBatchSize, optBatchSize, maxBatchSize;
    int minChannels, optChannels, maxChannels;
    int minWidth, optWidth, maxWidth;
    int minHeight, optHeight, maxHeight;
} w2x_build_config;

typedef struct w2x_render_config {  /* trt::RenderConfig, config.h:33-43 */
    int deviceId, precision, batchSize, channels, height, width, scaling;
    double overlapX, overlapY;
    int tta;
    int ttaBugCompat;               /* extension, see include/w2x/config.h */
} w2x_render_config;

typedef void (*w2x_message_fn)(int severity, const char* message, void* user);        /* logger.h:20 */
typedef void (*w2x_progress_fn)(int current, int total, double speed, void* user);    /* logger.h:21 */

w2x_engine* w2x_create(void);
void w2x_destroy(w2x_engine* e);
void w2x_set_message_callback(w2x_engine* e, w2x_message_fn fn, void* user);
void w2x_set_progress_callback(w2x_engine* e, w2x_progress_fn fn, void* user);
int w2x_build(w2x_engine* e, const char* onnx_path, const w2x_build_config* cfg);
int w2x_load(w2x_engine* e, const char* onnx_path, const w2x_render_config* cfg);
/* src/dst: interleaved 8-bit BGR, `step` bytes per row; dst must be rows*scaling x cols*scaling. */
int w2x_render(w2x_engine* e, const uint8_t* src, int rows, int cols, size_t src_step, uint8_t* dst, size_t dst_step);
/* The same on 16-bit samples (extension; the reference reads and writes 8-bit frames only, capture.cpp:96-99, and lists deeper images as a
 * TODO, README.md:88): interleaved BGR uint16, steps in BYTES; x = u16 * float(1/65535) into the network, sat(rint(x * 65535)) out. */
int w2x_render16(w2x_engine* e, const uint16_t* src, int rows, int cols, size_t src_step, uint16_t* dst, size_t dst_step);
/* Multi-GPU split of ONE frame (no reference counterpart: main.cpp:70-74 is single-device): strip `part` of
 * `parts` = a contiguous range of the reference's column-major tile order (img2img_render.cpp:43-44) plus the output columns
 * it alone composes.  w2x_strip_plan is pure host logic: out[0..3] = first_tile, tile_count, x0, x1 (x in output pixels). */
int w2x_render_strip(w2x_engine* e, const uint8_t* src, int rows, int cols, size_t src_step, uint8_t* dst, size_t dst_step, int part, int parts);
/* ONE frame over `count` engines of this process with every tile computed once (no reference counterpart; SURVEY.md 8e, seam exchange):
 * engine k renders part k of w2x_shard_plan - a contiguous range of the reference's column-major tile order (img2img_render.cpp:43-44) -
 * the blend bands of the ny + 1 tiles in front of a range are copied device-to-device from the engine(s) that computed them, and each
 * engine composes and downloads the canvas cells of its own tiles.  Same bytes as w2x_render.  w2x_shard_plan is pure host logic:
 * out[0..3] = first_tile, tile_count, halo_first, nrect; out[4 + 4r ..] = x, y, w, h of rectangle r (r < 3) in output pixels. */
int w2x_render_sharded(w2x_engine* const* engines, int count, const uint8_t* src, int rows, int cols, size_t src_step, uint8_t* dst, size_t dst_step);
int w2x_shard_plan(int in_w, int in_h, int out_w, int out_h, int tile_in, int tile_out, int scaling, double overlap_x, double overlap_y,
                   int part, int parts, int* out16);
/* The same split with ONE PROCESS PER GPU: rank r calls w2x_shard_compute (its own tiles into its slab, complete on return), exports w2x_shard_slab with
 * w2x_ipc_export (a 64-byte handle, hipIpcGetMemHandle) and hands it to its peers by whatever channel the caller has (bench.py: gloo all_gather - the
 * exchange is also the barrier); w2x_shard_finish takes the w2x_ipc_open'ed slabs of the parts in front of `part` (slabs[q] for q < part that own tiles;
 * other entries are ignored; devices[q] = the logical device the slab lives on, or NULL), copies the seam bands device to device, composes and writes this
 * part's canvas cells of dst (dst rows x cols = the OUTPUT size).  No collective on the data path. */
int w2x_shard_compute(w2x_engine* e, const uint8_t* src, int rows, int cols, size_t src_step, int part, int parts);
const void* w2x_shard_slab(w2x_engine* e, size_t* bytes);
int w2x_shard_finish(w2x_engine* e, uint8_t* dst, int out_rows, int out_cols, size_t dst_step, int part, int parts, const void* const* slabs, const int* devices);
int w2x_ipc_export(const void* device_ptr, uint8_t* out64);
void* w2x_ipc_open(const uint8_t* handle64, int device);
void w2x_ipc_close(void* p);
/* Frame sequence with the PCIe copies overlapped (no reference counterpart: main.cpp:263-269 renders frame by frame): srcs/dsts are
 * arrays of `count` frame pointers of one size.  The copies run by DMA beside the kernels only for page-locked memory: take the
 * frame buffers from w2x_alloc_host (engine-owned, w2x_free_host or w2x_destroy releases them).  w2x_pin_host page-locks caller
 * memory in place and accepts whole pages only (data and bytes multiples of 4096). */
int w2x_render_sequence(w2x_engine* e, const uint8_t* const* srcs, int rows, int cols, size_t src_step, uint8_t* const* dsts, size_t dst_step, int count);
void* w2x_alloc_host(w2x_engine* e, size_t bytes);
void w2x_free_host(w2x_engine* e, void* data);
int w2x_pin_host(w2x_engine* e, void* data, size_t bytes);
void w2x_unpin_host(w2x_engine* e, void* data);
int w2x_strip_plan(int in_w, int in_h, int out_w, int out_h, int tile_in, int tile_out, int scaling, double overlap_x, double overlap_y,
                   int part, int parts, int* out4);
int w2x_infer(w2x_engine* e, const float* input_nchw, float* output_nchw);
int w2x_output_tile_size(w2x_engine* e);
double w2x_plan_flops(w2x_engine* e);   /* algorithmic FLOP of one network pass */
int w2x_pass_tiles(w2x_engine* e);      /* tiles per network pass (batchSize x super-batch factor) */
float w2x_last_render_ms(w2x_engine* e);
float w2x_bench_resident(w2x_engine* e, int iters);
/* out[5*k+{0,1,2}] = {ms, launches, algorithmic FLOP} for k = 0 gemm, 1 attention, 2 se/scale, 3 gather, 4 compose, 5 fused mlp;
   out[30] = frame ms; cap >= 31 */
int w2x_profile_frame(w2x_engine* e, double* out, int cap);
/* ms per plan op (same order as w2x_describe_plan) of the last w2x_profile_frame; returns the op count */
int w2x_op_times(w2x_engine* e, double* out, int cap);

/* Host-only helpers (no GPU needed). */
/* rects: 4 ints (x,y,w,h) per tile, column-major tile order; returns tile count (or -1 if cap is too small). */
int w2x_calculate_tiles(int in_w, int in_h, int out_w, int out_h, int tile_in, int tile_out, int scaling,
                        double overlap_x, double overlap_y, int* in_rects, int* out_rects, int cap);
/* which: 0 top, 1 right, 2 bottom, 3 left (weights[] index, img2img_load.cpp:30-51); out: size*size floats. */
int w2x_tile_weights(int which, int overlap_x, int overlap_y, int size, float* out);
/* Lower an ONNX file at [batch,3,tile,tile] and write a textual description of the plan (ops, FLOPs) into buf. */
int w2x_describe_plan(const char* onnx_path, int batch, int tile, char* buf, size_t cap);
/* the same for any precision (W2X_PRECISION_FP16 / _TF32 / _FP32: the plan build() would write for that BuildConfig::precision; TF32 and FP32 share one) */
int w2x_describe_plan_precision(const char* onnx_path, int batch, int tile, int precision, char* buf, size_t cap);
/* Host-only halves of build() / load() for tools and tests: lower an ONNX file and write the plan (no .json side file, no
 * device); read a plan file back and run the consistency checks load() runs (img2img_load.cpp:149-154 "Failed to deserialize
 * engine"), writing "ok" or the reason into buf. */
int w2x_write_engine_file(const char* onnx_path, int batch, int tile, const char* out_path);
int w2x_validate_engine_file(const char* path, char* buf, size_t cap);
/* PCI bus id ("0000:c1:00.0") of HIP device `device` of this process (after W2X_DEVICE_MAP), for callers that place their host threads and
 * page-locked buffers on the GPU's NUMA node (/sys/bus/pci/devices/<id>/local_cpulist); 1 on success. */
int w2x_device_pci_bus_id(int device, char* buf, size_t cap);
/* sha256 hex digest (names engine files; utilities/sha256.h:39-94). out: 65 bytes. */
void w2x_sha256_hex(const void* data, size_t len, char* out);
const char* w2x_version(void);
/* Test hook, process-wide: the reference paths the A/B tests compare the shipped kernels and plans with - un-fused lowering ("no_fuse", "no_fuse_attn",
 * "no_se_fold"; read by build), separate launches ("no_fuse_head", "no_fuse_stem", "no_fuse_up"; read by load), the general kernel instead of a shape-specialised one
 * ("no_pixgemm", "no_conv3", "no_conv3h", "no_conv48", "no_stem", "attn_valu"; read per launch).  They have no environment names.  The operational
 * switches (W2X_GROUPS, W2X_NO_GRAPH, ... - csrc/switches.h) can be set here too, by field name, but are re-read from the environment by
 * every build / load.  1 = set, 0 = no such switch. */
int w2x_debug_set(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif
