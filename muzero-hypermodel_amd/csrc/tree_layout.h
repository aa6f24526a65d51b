// tree_layout.h -- device data layout of the batched search trees (gfx950 / MI355X).
//
// E independent trees are searched in lock step.  Simulation s expands exactly one node per tree,
// so "expanded node k" (k = 0 for the root, k = s+1 for the node expanded by simulation s) is a
// dense index shared by all trees, and every pool is laid out [k][e][...]:
//
//   child blocks   [(S+1)][E] lines of `line_stride` bytes.  A node's block is found through its parent's link
//                  (ChildLinks::block_loc = 2 * slab + half): normally line (k, e) of its own expansion index k,
//                  half 0; with 64-byte blocks (A <= 2) the FIRST child a node expands moves into the free second
//                  half of that node's line instead (when the node owns its line), so that the most frequent hop of
//                  a descent (65 % on the CartPole traces) needs no new 128-byte line.  Block (k,e) holds the A children of
//                  expanded node k of tree e as A 32-byte child records (HbmChild)
//                      HbmChild    { ChildStats { f64 value_sum; f64 prior };
//                                    ChildLinks { f32 reward; i32 visits; i32 child_node; i32 block_loc } }
//                  one lane per child reads its record with two dwordx4; the backup's read-modify-write of a path node
//                  (value_sum, reward, visits) stays inside ONE 32-byte sector, the unit the memory system fetches and
//                  writes back (round 2: as two member arrays a node update cost two sectors each way).  32*A bytes,
//                  padded to a multiple of 64 B (CartPole A=2: exactly one 64-B block per descent step).  Trees that
//                  live in LDS (fused kernels) keep the two members as separate 16-byte-wide arrays (conflict-free
//                  ds_read_b128 across the lanes) and publish records.
//   hidden pool    f32 [(S+1)][E][H]: the state of expanded node k.  The network writes slab s+1
//                  as one contiguous [E,H] matrix; select gathers rows (k_e, e) into a batch.
//   path           i32 [S][E]: level d of the current descent, packed (parent k << 16 | slot).
//   per-tree [E]   min-max stats, root value sum / reward, counters, RNG position.
//   RNG            u32 [E][624] MT19937 keys (numpy legacy clone), see np_legacy_rng.h.
//
// to_play is not stored: with players == range(P) the reference's virtual_to_play rotation
// (self_play.py:332-335) makes it (root_to_play + tree_depth) mod P.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace mz {

struct alignas(16) ChildStats {
    double value_sum;  // Node.value_sum (self_play.py:439)
    double prior;      // Node.prior
};

struct alignas(16) ChildLinks {
    float reward;        // Node.reward: always an fp32 value (.item() of an fp32 tensor, self_play.py:345)
    int32_t visits;      // Node.visit_count
    int32_t child_node;  // expanded-node index of this child, -1 while it is a leaf (not expanded)
    int32_t block_loc;   // where the child's own block lives: 2 * (slab index) + (half of the line), see line_stride
};

// A child as it lives in HBM: one 32-byte sector holds everything a descent reads and a backup rewrites of it.
struct alignas(32) HbmChild {
    ChildStats stats;
    ChildLinks links;
};
constexpr uint32_t kChildRecordBytes = 32;
static_assert(sizeof(HbmChild) == kChildRecordBytes, "a child record is one 32-byte sector");

struct alignas(16) MinMax {
    double minimum, maximum;  // MinMaxStats (self_play.py:551-568)
};

struct TreeParams {
    int32_t E, A, S, P, F, support, H;
    int32_t chunks;          // ceil(A / 64) when A > 64, else 1
    int32_t group;           // lanes per tree: pow2 >= min(A, 64), one wavefront holds 64/group trees
    uint32_t block_stride;   // bytes per child block
    uint32_t links_offset;   // 16 * A: where the links array of an LDS-resident block starts (HBM blocks hold records)
    uint32_t line_stride;    // HBM pool: bytes between the lines of consecutive trees in a slab.  = block_stride, or
                             // 128 when a 64-byte block shares its line with the block of its node's first-expanded
                             // child (A <= 2): the memory system fetches 128-byte lines whatever a request asks for
                             // (profiles/r02_record_size_ceiling.jsonl), so the most likely next hop rides along
    int32_t* leaf_loc;       // [E] block location select chose for the node the coming expand_backup creates
    double discount;
    double noise_frac;
    double noise_alpha;      // config.root_dirichlet_alpha (device-drawn exploration noise, root_noise_kernel)
    double* noise_rows;      // [E][A] Dirichlet rows of the current search when the device draws them
    uint32_t* noise_words;   // [E] RNG words that draw consumed (the host mirror steps over them at readout)
    // pools
    uint8_t* blocks;
    float* hidden;
    int32_t* path;
    int32_t* path_ties;      // optional [S][E] tie-list sizes (debug / parity tests), may be null
    int32_t* path_len;       // [E] depth of the current descent
    int32_t* leaf_parent;    // [E] expanded-node index of the leaf's parent (hidden-state slab)
    MinMax* min_max;         // [E]
    double* root_value_sum;  // [E]
    double* root_reward;     // [E]
    float* root_predicted;   // [E]
    int32_t* root_children;  // [E] number of root children; 0 = tree inactive this search
    int32_t* root_to_play;   // [E]
    int32_t* root_action;    // [E][A] slot -> action at the root
    int32_t* max_depth;      // [E]
    int64_t* depth_sum;      // [E]
    uint32_t* tie_words;     // [E]
    uint32_t* mt_key;        // [E][624]
    int32_t* mt_pos;         // [E]
    const double* pbc_log;   // [S+1] log((N + base + 1)/base) + init   (host libm, self_play.py:385-390)
    const double* pbc_sqrt;  // [S+1] sqrt(N)
    int32_t* error_flag;     // sticky device-side error word
};

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// Optional pair of HIP events a launcher binds to the kernel dispatch (profiling mode).
struct LaunchTiming {
    hipEvent_t start = nullptr;
    hipEvent_t stop = nullptr;
};

}  // namespace mz
