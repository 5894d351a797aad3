// Every switch of the library in ONE place.  Host-only (lower.cpp and the sanitizer build include it).
//
// Two kinds:
//  * operational - environment variables, read by switches_from_env() when an engine is built or loaded and never on the launch
//    path.  Each one changes HOW a frame is scheduled (tile groups, graphs, parts, passes), none changes an output byte; the GPU
//    suite asserts that for every one of them.  The non-default ones are logged at `warn` by load().
//  * reference paths - the un-fused plans and the general kernel behind every shape-specialised one.  They have no environment
//    name: only w2x_debug_set() (include/w2x/c_api.h) reaches them, and only the A/B tests call it.
#pragma once
#include <string>

namespace w2x {

struct Switches {
    // operational (environment)
    int superbatch = 0;          // W2X_SUPERBATCH=S    reference batches per network pass (0: sized for ~48 tiles of 256 x 256)
    int groups = 2;              // W2X_GROUPS=1..4     tile groups a pass is cut into, each on its own stream (1: passes stay whole)
    int render_parts = 3;        // W2X_RENDER_PARTS    parts a render() call pipelines a frame in (1: one part)
    bool no_graph = false;       // W2X_NO_GRAPH        plain launches instead of hipGraph replay
    bool no_rolling = false;     // W2X_NO_ROLLING      frames of a sequence join their streams after every frame
    bool poison = false;         // W2X_POISON          stale activations become fp16 NaNs before every frame
    bool check_general = false;  // W2X_CHECK_GENERAL   every shape-specialised launch is compared with the general kernel (slow; logs at warn)
    bool roctx = false;          // W2X_ROCTX           a roctx range per plan op
    // reference paths (w2x_debug_set only)
    bool no_fuse = false;        // lowering: keep LayerNorm / attention / MLP as separate ops
    bool no_fuse_attn = false;   // lowering: keep the attention branch un-fused (the MLP still fuses)
    bool no_se_fold = false;     // lowering: squeeze-excite gates as an in-place pass instead of folded into the consumers
    bool no_fuse_head = false;   // engine: the image head as its own launch behind the last MLP
    bool no_fuse_stem = false;   // engine: the stem as its own launch in front of the patch convolution
    bool no_fuse_up = false;     // engine: cunet's transposed convolutions as their own launches in front of the 3x3 convolutions that read them
    bool no_conv3h_walk = false; // k_conv3h.hip: the tile kernel instead of the column walk for the 64-channel image heads
    bool no_pixgemm = false, no_conv3 = false, no_conv3h = false, no_conv48 = false, no_stem = false;   // launchers: the general kernel instead
    bool attn_valu = false;      // launchers: the lane-per-query attention core instead of the matrix-pipe one
};

Switches& switches();                            // process-wide; (W2X_DEVICE_MAP, a test hook of its own, is read where devices are resolved)
void switches_from_env();                        // re-reads the operational ones (build() and load() call it)
bool set_switch(const char* name, long value);   // a field by its name above; false: no such switch
std::string switches_nondefault();               // "groups=1 no_graph" - what load() logs

}  // namespace w2x
