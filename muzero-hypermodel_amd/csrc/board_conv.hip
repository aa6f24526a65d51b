// board_conv.hip -- the 3x3 convolution of the board residual networks on the matrix cores (gfx950), with what
// follows it in the reference fused into the epilogue:
//
//     out = act( conv3x3(x, w; padding 1, stride 1, no bias) * scale[c] + shift[c]  (+ residual) )
//
// = Conv2d -> BatchNorm2d (eval: running statistics folded into scale / shift) -> (+ x) -> ReLU of
// reference models.py:213-229 (conv3x3, ResidualBlock.forward), 318-330 (RepresentationNetwork), 399-420
// (DynamicsNetwork), one launch instead of MIOpen's fp32 Winograd kernel + an element-wise launch.
//
// Implicit GEMM, exact fp32:  D[m][n] = sum_k A[m][k] * B[k][n]
//     m = (sample, y, x) output position      n = output channel      k = (tap, input channel)
// on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit for bit a k-ordered fmaf chain, one rounding per product --
// the same arithmetic as a scalar fp32 loop over taps and channels).  A workgroup (4 waves) owns SB samples:
//   * their input goes to LDS once, channel-fastest, as padded planes with a zero border (rows share one zero column), so
//     a tap is an address offset and the padding needs no predicate; channels are padded to whole groups of 16;
//   * wave w (8 per workgroup, two per SIMD) owns the 16 output channels of column tile (w % NT) and the row tiles
//     w / NT, w / NT + 8 / NT, ...; per group of 16 input channels of one tap (4 k-steps) it reads ONE 16-byte value per
//     row tile from LDS and ONE 16-byte value from the packed weights (global, L2-resident, a group ahead), and issues
//     4 MFMAs per row tile, each tile into its own accumulator;
//   * epilogue: accumulators -> LDS tile -> folded batch norm, residual, ReLU -> coalesced NCHW stores.
// The weights arrive packed by mzmcts_board_conv_pack, refreshed in place whenever the parameters change.
//
// Roofline: MFMA-bound by construction (fp32 matrix peak 157.3 TFLOP/s = 64 FLOP/clk/SIMD): Connect4 [1024, 64, 6, 7]
// -> 3.17 GFLOP per call = 20.2 us at peak; per group a wave issues 24 MFMAs (768 cycles) against 6 LDS reads and one
// global load.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "../../include/mzmcts.h"

namespace mz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kConvWaves = 8;   // two per SIMD: one wave's LDS reads and weight loads hide under the other's MFMAs
constexpr int kConvGroup = 16;  // input channels per group = 4 k-steps; a lane fetches its 4 channels with one 16-byte read

__host__ __device__ inline int conv_groups(int cin) { return (cin + kConvGroup - 1) / kConvGroup; }
__host__ __device__ inline int conv_packed_floats(int cin, int cout) {  // [9 * groups + 2 spare][4 kk][cout][4 g]
    return (9 * conv_groups(cin) + 2) * 4 * cout * 4;
}

// Packed weights: wt[(((tap * NG + grp) * 4 + kk) * cout + n) * 4 + g] = w[n][grp * 16 + 4 * kk + g][tap]
// (0 for channels >= cin and for the spare group): lane (n, kk) of a wave reads the four weights of its four k-steps of
// a group with ONE 16-byte load, 16 lanes x 16 B contiguous per kk.
__global__ __launch_bounds__(256) void board_conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int cin,
                                                              int cout) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= conv_packed_floats(cin, cout)) return;
    const int g = i & 3;
    const int n = (i >> 2) % cout;
    const int kk = ((i >> 2) / cout) & 3;
    const int tg = (i >> 2) / cout / 4;                 // tap * NG + grp
    const int ng = conv_groups(cin);
    const int tap = tg / ng, grp = tg % ng;
    const int ci = grp * kConvGroup + 4 * kk + g;
    wt[i] = (tap < 9 && ci < cin) ? w[(static_cast<size_t>(n) * cin + ci) * 9 + tap] : 0.f;
}

template <int NT, int H, int W, int SB, bool RESIDUAL, bool RELU>
__global__ __launch_bounds__(64 * kConvWaves) void board_conv3x3_kernel(const float* __restrict__ x,
                                                                         const float* __restrict__ wt,
                                                                         const float* __restrict__ scale,
                                                                         const float* __restrict__ shift,
                                                                         const float* __restrict__ residual,
                                                                         float* __restrict__ out, int batch, int cin, uint32_t cin_magic) {
    constexpr int P = H * W;
    constexpr int PW = W + 1;                           // one zero column between rows: right border of row y = left of y+1
    constexpr int PP = (H + 2) * PW + 1;                // positions of a padded plane
    constexpr int ROWS = SB * P;                        // output rows of the workgroup
    constexpr int MT = (ROWS + 15) / 16;                // row tiles of the workgroup
    constexpr int RG = kConvWaves / NT;                 // waves sharing a column tile
    constexpr int MTW = (MT + RG - 1) / RG;             // row tiles per wave
    constexpr int COUT = 16 * NT;
    constexpr int LDM = ROWS + 1;                       // row stride of the output staging tile [COUT][LDM]
    constexpr int THREADS = 64 * kConvWaves;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // planes [SB][PP][CP], later the staging tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b0 = blockIdx.x * SB;
    const int n_samples = min(SB, batch - b0);
    const int ng = conv_groups(cin);
    const int CP = ng * kConvGroup + 4;                 // channel stride (+4 words: rows of a tile fall into different banks)

    // ---- stage the input: zero the LDS tile (borders, padding channels, missing samples), then the boards, channel-
    //      fastest; global loads 16 bytes per lane, four in flight per thread ------------------------------------------
    const int count = n_samples * cin * P;              // contiguous in global memory (NCHW)
    const float* src = x + static_cast<size_t>(b0) * cin * P;
    {
        const int words = SB * PP * CP;                 // (CP is a multiple of 4)
        float4* z = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < words / 4; i += THREADS) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    auto place = [&](int i, float v) {
        const int p = i % P;
        const int sc = i / P;                           // s * cin + ci
        const int s = static_cast<int>(__umulhi(static_cast<uint32_t>(sc), cin_magic));   // sc / cin (sc < 2^16: exact)
        const int ci = sc - s * cin;
        lds[(s * PP + (p / W + 1) * PW + (p % W) + 1) * CP + ci] = v;
    };
    for (int i0 = tid * 4; i0 < count; i0 += 4 * THREADS * 4) {
        float4 staged[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * THREADS * 4;
            staged[k] = (i + 3 < count) ? *reinterpret_cast<const float4*>(src + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * THREADS * 4;
            if (i + 3 < count) {
                place(i, staged[k].x), place(i + 1, staged[k].y), place(i + 2, staged[k].z), place(i + 3, staged[k].w);
            } else {
                for (int q = i; q < count; ++q) place(q, src[q]);   // ragged tail (at most 3 floats, one thread)
            }
        }
    }
    __syncthreads();

    // ---- per-lane row addresses ---------------------------------------------------------------------------------
    const int col_tile = wave % NT;
    const int row_group = wave / NT;
    const int i_row = lane & 15;
    const int kk = lane >> 4;                           // this lane feeds channels 4 kk .. 4 kk + 3 of a group
    int base[MTW];
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        const int tile = row_group + t * RG;
        int m = tile * 16 + i_row;
        if (tile >= MT || m >= ROWS) m = 0;             // rows past the end read a valid address; never stored
        const int s = m / P, p = m % P;
        base[t] = (s * PP + (p / W + 1) * PW + (p % W) + 1) * CP + 4 * kk;
    }
    f32x4 acc[MTW];
#pragma unroll
    for (int t = 0; t < MTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop: 9 taps x ng channel groups; per group 4 k-steps (fewer in a short last group) ----------------------
    // k-step g of a group multiplies channels {4 kk + g : kk = 0..3}: lane (row i, kk) holds A[i][4 kk + g] in element g of
    // ONE 16-byte LDS read, lane (column n, kk) holds B[4 kk + g][n] in element g of ONE 16-byte global load.  Software
    // pipeline: the next group's A rows are read while this group's MFMAs issue, the next group's weights are in flight
    // from L2 meanwhile.
    const int n_col = col_tile * 16 + i_row;
    const f32x4* wlane = reinterpret_cast<const f32x4*>(wt) + kk * COUT + n_col;   // + (tap * ng + grp) * 4 * COUT
    const int last_steps = min(4, cin - (ng - 1) * kConvGroup);                    // k-steps of a tap's last group
    const int iterations = 9 * ng;

    int grp = 0, tap = 0;                               // the group whose A rows are being fetched
    auto lds_offset = [&]() { return ((tap / 3 - 1) * PW + (tap % 3 - 1)) * CP + grp * kConvGroup; };
    auto advance = [&]() {
        if (++grp == ng) {
            grp = 0;
            tap = tap < 8 ? tap + 1 : 8;                // (the read issued in the last iteration is not used)
        }
    };
    f32x4 a[MTW];
    f32x4 b = wlane[0];
    f32x4 b1 = wlane[4 * COUT];                         // weights stay two groups ahead of the MFMAs (L2 round trip)
    {
        const int off = lds_offset();
#pragma unroll
        for (int t = 0; t < MTW; ++t) a[t] = *reinterpret_cast<const f32x4*>(lds + base[t] + off);
        advance();
    }
    int grp_now = 0;
    for (int it = 0; it < iterations; ++it) {
        const f32x4 bn = wlane[static_cast<size_t>(it + 2) * 4 * COUT];   // (two spare zero groups follow the last one)
        const int off = lds_offset();
        f32x4 an[MTW];
#pragma unroll
        for (int t = 0; t < MTW; ++t) an[t] = *reinterpret_cast<const f32x4*>(lds + base[t] + off);
        advance();
        const int steps = (grp_now == ng - 1) ? last_steps : 4;   // uniform
        grp_now = (grp_now + 1 == ng) ? 0 : grp_now + 1;
        if (steps == 4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int t = 0; t < MTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][g], b[g], acc[t], 0, 0, 0);
            }
        } else {                                         // short last group (e.g. the dynamics input's action plane)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                if (g < steps) {
#pragma unroll
                    for (int t = 0; t < MTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][g], b[g], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < MTW; ++t) a[t] = an[t];
        b = b1;
        b1 = bn;
    }

    // ---- epilogue: accumulators -> LDS tile [column][row] -> folded batch norm, residual, ReLU, coalesced NCHW stores ----
    const int out_count = n_samples * COUT * P;         // contiguous in global memory; a multiple of 4 (COUT is)
    const size_t g0 = static_cast<size_t>(b0) * COUT * P;
    __syncthreads();                                    // every wave has finished reading the planes
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        const int tile = row_group + t * RG;
        if (tile >= MT) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                   // D[row = 4 * (lane >> 4) + r][col = lane & 15]
            const int m = tile * 16 + 4 * kk + r;
            if (m < ROWS) lds[n_col * LDM + m] = acc[t][r];
        }
    }
    float* affine = lds + COUT * LDM;                   // [scale | shift] behind the tile
    if (tid < COUT) affine[tid] = scale[tid];
    else if (tid < 2 * COUT) affine[tid] = shift[tid - COUT];
    __syncthreads();
    auto finish = [&](int i, float skip) {
        const int p = i % P;
        const int sn = i / P;                           // s * COUT + n
        const int n = sn % COUT;
        const int s = sn / COUT;
        float v = lds[n * LDM + s * P + p] * affine[n] + affine[COUT + n];
        if (RESIDUAL) v = v + skip;
        if (RELU) v = v < 0.f ? 0.f : v;                // (a NaN stays a NaN, as torch.relu keeps it)
        return v;
    };
    for (int i0 = tid * 4; i0 < out_count; i0 += 3 * THREADS * 4) {
        float4 res[3];
        if (RESIDUAL) {                                 // three 16-byte loads of the skip connection in flight per thread
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = i0 + k * THREADS * 4;
                res[k] = i < out_count ? *reinterpret_cast<const float4*>(residual + g0 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = i0 + k * THREADS * 4;
            if (i < out_count) {
                const float4 r4 = RESIDUAL ? res[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(out + g0 + i) =
                    make_float4(finish(i, r4.x), finish(i + 1, r4.y), finish(i + 2, r4.z), finish(i + 3, r4.w));
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------
// A whole tower of such convolutions in ONE launch: the activations of the workgroup's SB samples stay in LDS from layer
// to layer (two padded-plane buffers, a layer reads one and writes the other), so only the tower's input is read from
// HBM and only the tensors a later launch needs are written back:
//     dynamics   conv(C+1 -> C) -> N x ResidualBlock -> [raw state -> reward head] -> min-max rescale -> next state
//     prediction N x ResidualBlock -> [features -> value / policy heads]
// (reference models.py:399-420 DynamicsNetwork.forward, 586-602 the rescale in MuZeroResidualNetwork.dynamics, 500-522
// PredictionNetwork.forward).  A residual block is two layers: conv1 reads the block's input x from buffer U and writes
// V; conv2 reads V and writes relu(conv * scale + shift + x) over x in U (every element is read and written by the one
// lane that owns it).  Per layer the same exact-fp32 MFMA main loop as board_conv3x3_kernel.
// -------------------------------------------------------------------------------------------------------------------
// Optional input of a tower straight from the search's pools (include/mzmcts.h mzmcts_tower_gather): sample b reads the
// hidden state of its leaf's parent, pool[(parent[b] * envs + b) * hidden + c * P + p] for its first C channels, and
// action[b] / action_space as its last (constant) plane -- reference models.py:553-568 without the [E, C + 1, H, W]
// tensor in between.  pool == nullptr: the tower reads x.
struct TowerGather {
    const float* pool;
    const int32_t* parent;
    const int64_t* action;
    long long envs;
    int hidden;
    float action_space;
};

struct TowerLayer {
    const float* wt;        // packed weights (mzmcts_board_conv_pack)
    const float* scale;     // folded batch norm
    const float* shift;
    float* export_raw;      // NCHW [batch, C, H, W] copy of this layer's output, or null
    float* export_unit;     // the same after the per-plane min-max rescale (which then also replaces it in LDS), or null
    int32_t cin;
    int32_t relu;
    int32_t skip;           // add the destination buffer's previous content (the residual block's input) before the ReLU
    int32_t pad;
};
constexpr int kMaxTowerLayers = 16;
struct TowerArgs {
    TowerLayer layer[kMaxTowerLayers];
    int32_t n_layers;
    // not null: a workgroup runs only if a gate entry covering one of its samples is != 0 (entry i = samples
    // i * gate_samples .. + gate_samples - 1: the workgroups of the split-precision launch that flagged them as
    // overflowed) -- the re-run of those samples, queued behind that launch with no host in between
    const int32_t* gate;
    int32_t gate_samples;
};

// A reward / value / policy head computed inside a tower launch from the activations of `layer` while they are in LDS
// (board_tower_cols_kernel): 1x1 convolution -> Linear + ELU -> Linear, reference models.py:467-480, 500-522.
struct TowerHead {
    mzmcts_head_desc d;
    float* out;        // [batch][outputs]
    int32_t layer;     // reads this layer's output (before an export_unit rescale replaces it)
    int32_t pad;
};
struct TowerHeads {       // (fixed roles instead of an indexed array: the kernel addresses its argument segment statically)
    TowerHead mid;        // one head on an export_unit layer (`mid.layer`): the reward head on the raw dynamics output
    TowerHead last0, last1;   // up to two heads on the last layer: value and policy
    int32_t has_mid, n_last;
};

// Diagnostic build only (-DMZ_TOWER_STAMPS, tools/stamp_tower.py): cycles of wave 0 of every workgroup per phase of the
// tower kernel, summed into a debug buffer nothing else reads.  The production library never defines it.
#ifdef MZ_TOWER_STAMPS
static __device__ unsigned long long g_tower_stamps[16];
#define MZ_TSTAMP_DECL unsigned long long tstamp_prev = __builtin_readcyclecounter(), tstamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MZ_TSTAMP(slot)                                                    \
    do {                                                                   \
        const unsigned long long now__ = __builtin_readcyclecounter();     \
        tstamp_acc[slot] += now__ - tstamp_prev;                           \
        tstamp_prev = now__;                                               \
    } while (0)
#define MZ_TSTAMP_FLUSH                                                                              \
    do {                                                                                             \
        if (threadIdx.x == 0)                                                                        \
            for (int s__ = 0; s__ < 8; ++s__) atomicAdd(&g_tower_stamps[s__], tstamp_acc[s__]);      \
    } while (0)
#else
#define MZ_TSTAMP_DECL
#define MZ_TSTAMP(slot)
#define MZ_TSTAMP_FLUSH
#endif

template <int NT, int H, int W, int SB>
__device__ __forceinline__ void board_tower_block(const float* __restrict__ x, int batch, int cin0, uint32_t cin0_magic, int cp0,
                                                  int cp1, const TowerArgs& args, const TowerGather& gather, int block_index) {
    constexpr int P = H * W;
    constexpr int PW = W + 1;
    constexpr int PP = (H + 2) * PW + 1;
    constexpr int ROWS = SB * P;
    constexpr int MT = (ROWS + 15) / 16;
    constexpr int RG = kConvWaves / NT;
    constexpr int MTW = (MT + RG - 1) / RG;
    constexpr int COUT = 16 * NT;
    constexpr int THREADS = 64 * kConvWaves;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // buffer 0: [SB][PP][cp0] | buffer 1: [SB][PP][cp1]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b0 = block_index * SB;
    const int n_samples = min(SB, batch - b0);
    const int buf1_at = SB * PP * cp0;                  // (offsets into lds, so that every access stays an LDS access)
    MZ_TSTAMP_DECL

    // ---- zero both buffers (borders, padding channels, missing samples), then the tower's input into buffer 0 -------
    {
        const int words = SB * PP * (cp0 + cp1);
        float4* z = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < words / 4; i += THREADS) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // gathered input: where sample q's hidden state starts in the pool, and its action plane's value (behind the buffers)
    const float** in_row = reinterpret_cast<const float**>(lds + SB * PP * (cp0 + cp1));
    float* act_plane = reinterpret_cast<float*>(in_row + SB);
    if (gather.pool && tid < SB) {
        const long long b = b0 + tid;
        const bool present = tid < n_samples;
        in_row[tid] = present ? gather.pool + (static_cast<size_t>(gather.parent[b]) * gather.envs + b) * gather.hidden : nullptr;
        act_plane[tid] = present ? static_cast<float>(gather.action[b]) / gather.action_space : 0.f;
    }
    __syncthreads();
    MZ_TSTAMP(0);
    if (gather.pool) {
        // input gathered from the hidden-state pool (+ the action plane).  A thread keeps ONE board position and walks over
        // the (sample, channel) planes, TPP planes per pass: consecutive threads read consecutive addresses of a row, the
        // position's arithmetic is done once, the row's address comes from LDS instead of a parent-index load per element.
        constexpr int TPP = THREADS / P;
        if (tid < TPP * P) {
            const int p = tid % P;
            const int at_p = ((p / W + 1) * PW + (p % W) + 1) * cp0;
            const int planes = n_samples * cin0;         // plane sc = sample * cin0 + channel
            for (int sc0 = tid / P; sc0 < planes; sc0 += 4 * TPP) {
                float v[4];
                int at[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int sc = sc0 + k * TPP;
                    at[k] = -1;
                    v[k] = 0.f;
                    if (sc < planes) {
                        const int s = static_cast<int>(__umulhi(static_cast<uint32_t>(sc), cin0_magic));
                        const int ci = sc - s * cin0;
                        v[k] = (ci * P < gather.hidden) ? in_row[s][ci * P + p] : act_plane[s];
                        at[k] = s * PP * cp0 + at_p + ci;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (at[k] >= 0) lds[at[k]] = v[k];
            }
        }
    } else {
        const int count = n_samples * cin0 * P;
        const float* src = x + static_cast<size_t>(b0) * cin0 * P;
        auto place = [&](int i, float v) {
            const int p = i % P;
            const int sc = i / P;
            const int s = static_cast<int>(__umulhi(static_cast<uint32_t>(sc), cin0_magic));
            const int ci = sc - s * cin0;
            lds[(s * PP + (p / W + 1) * PW + (p % W) + 1) * cp0 + ci] = v;
        };
        for (int i0 = tid * 4; i0 < count; i0 += 4 * THREADS * 4) {
            float4 staged[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * THREADS * 4;
                staged[k] = (i + 3 < count) ? *reinterpret_cast<const float4*>(src + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * THREADS * 4;
                if (i + 3 < count) {
                    place(i, staged[k].x), place(i + 1, staged[k].y), place(i + 2, staged[k].z), place(i + 3, staged[k].w);
                } else {
                    for (int q = i; q < count; ++q) place(q, src[q]);
                }
            }
        }
    }
    __syncthreads();

    MZ_TSTAMP(1);
    const int col_tile = wave % NT;
    const int row_group = wave / NT;
    const int i_row = lane & 15;
    const int kk = lane >> 4;
    const int n_col = col_tile * 16 + i_row;
    // plane position (word offset / channel stride) of this lane's A rows, and of the D column it stores (-1 = not a
    // position of a present sample).  The products are issued as W x X^T (weights as the MFMA's first operand; the same
    // k-ordered sums): a lane's four D values are FOUR CONSECUTIVE OUTPUT CHANNELS (4 kk + r of its column tile) of
    // position i_row of its row tile, so the epilogue is one 16-byte LDS read and write per tile with no index arithmetic.
    int pos_a[MTW], pos_d[MTW];
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        const int tile = row_group + t * RG;
        int m = tile * 16 + i_row;
        const bool present = tile < MT && m < ROWS && m / P < n_samples;
        if (tile >= MT || m >= ROWS) m = 0;
        const int sidx = m / P, p = m % P;
        pos_a[t] = sidx * PP + (p / W + 1) * PW + (p % W) + 1;
        pos_d[t] = present ? pos_a[t] : -1;
    }

    for (int l = 0; l < args.n_layers; ++l) {
        const TowerLayer& L = args.layer[l];
        const int in = (l & 1) ? buf1_at : 0;
        const int dst = (l & 1) ? 0 : buf1_at;
        const int CPI = (l & 1) ? cp1 : cp0, CPO = (l & 1) ? cp0 : cp1;
        const int cin = L.cin;
        const int ng = conv_groups(cin);

        f32x4 acc[MTW];
#pragma unroll
        for (int t = 0; t < MTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4* wlane = reinterpret_cast<const f32x4*>(L.wt) + kk * COUT + n_col;
        const int last_steps = min(4, cin - (ng - 1) * kConvGroup);
        const int iterations = 9 * ng;
        int grp = 0, tap = 0;
        auto lds_offset = [&]() { return ((tap / 3 - 1) * PW + (tap % 3 - 1)) * CPI + grp * kConvGroup + 4 * kk; };
        auto advance = [&]() {
            if (++grp == ng) {
                grp = 0;
                tap = tap < 8 ? tap + 1 : 8;
            }
        };
        f32x4 a[MTW];
        f32x4 b = wlane[0];
        f32x4 b1 = wlane[4 * COUT];
        {
            const int off = lds_offset();
#pragma unroll
            for (int t = 0; t < MTW; ++t) a[t] = reinterpret_cast<const f32x4*>(lds)[(in + pos_a[t] * CPI + off) >> 2];   // (16-byte aligned)
            advance();
        }
        int grp_now = 0;
        for (int it = 0; it < iterations; ++it) {
#ifdef MZ_TOWER_NO_WLOAD
            const f32x4 bn = b;
#else
            const f32x4 bn = wlane[static_cast<size_t>(it + 2) * 4 * COUT];
#endif
            const int off = lds_offset();
            f32x4 an[MTW];
#ifdef MZ_TOWER_NO_ALOAD
#pragma unroll
            for (int t = 0; t < MTW; ++t) an[t] = a[t];
#else
#pragma unroll
            for (int t = 0; t < MTW; ++t) an[t] = reinterpret_cast<const f32x4*>(lds)[(in + pos_a[t] * CPI + off) >> 2];
#endif
            advance();
            const int steps = (grp_now == ng - 1) ? last_steps : 4;
            grp_now = (grp_now + 1 == ng) ? 0 : grp_now + 1;
            if (steps == 4) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int t = 0; t < MTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[g], a[t][g], acc[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (g < steps) {
#pragma unroll
                        for (int t = 0; t < MTW; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[g], a[t][g], acc[t], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < MTW; ++t) a[t] = an[t];
            b = b1;
            b1 = bn;
        }

        MZ_TSTAMP(2);
        // ---- layer epilogue into the destination planes: D[channel = 4 kk + r][position = lane & 15] ------------------
        {
            const float4 s4 = *reinterpret_cast<const float4*>(L.scale + col_tile * 16 + 4 * kk);
            const float4 h4v = *reinterpret_cast<const float4*>(L.shift + col_tile * 16 + 4 * kk);
            const f32x4 sc = {s4.x, s4.y, s4.z, s4.w}, sh = {h4v.x, h4v.y, h4v.z, h4v.w};
            const int skip = L.skip, relu = L.relu;
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
                const int pos = pos_d[t];
                if (pos < 0) continue;                  // (missing samples stay zero)
                f32x4* cell = reinterpret_cast<f32x4*>(lds) + ((dst + pos * CPO + col_tile * 16 + 4 * kk) >> 2);   // (16-byte aligned)
                f32x4 v = acc[t] * sc + sh;
                if (skip) v = v + *cell;
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
                }
                *cell = v;
            }
        }
        MZ_TSTAMP(3);
        __syncthreads();                                // the layer's output is complete; its input may be overwritten
        MZ_TSTAMP(4);

        if (L.export_raw || L.export_unit) {
            // (a thread keeps one board position and walks over the (sample, channel) planes, TPP planes per pass:
            // consecutive threads write consecutive addresses, the position's arithmetic is done once)
            constexpr int TPP = THREADS / P;
            const size_t g0 = static_cast<size_t>(b0) * COUT * P;
            const int walker_p = tid % P;
            const int walker_at = (walker_p / W + 1) * PW + (walker_p % W) + 1;
            auto export_planes = [&](float* out) {
                if (tid < TPP * P)
                    for (int sn = tid / P; sn < n_samples * COUT; sn += TPP)
                        out[g0 + sn * P + walker_p] = lds[dst + ((sn / COUT) * PP + walker_at) * CPO + sn % COUT];
            };
            if (L.export_raw) export_planes(L.export_raw);
            if (L.export_unit) {
                // per (sample, channel) plane: (x - min) / span, span = max - min (+ 1e-5 below 1e-5) -- models.py:525-549,
                // the operations of unit_rescale_kernel, so the same bits
                __syncthreads();
                for (int q = tid; q < n_samples * COUT; q += THREADS) {
                    const int n = q % COUT, sidx = q / COUT;
                    float* plane = &lds[dst + (sidx * PP) * CPO + n];
                    float lo = plane[(PW + 1) * CPO], hi = lo;
                    for (int p = 1; p < P; ++p) {
                        const float v = plane[((p / W + 1) * PW + (p % W) + 1) * CPO];
                        lo = (v < lo || v != v) ? v : lo;
                        hi = (v > hi || v != v) ? v : hi;
                    }
                    float span = hi - lo;
                    if (span < 1e-5f) span = span + 1e-5f;
                    for (int p = 0; p < P; ++p) {
                        float* c = plane + ((p / W + 1) * PW + (p % W) + 1) * CPO;
                        *c = (*c - lo) / span;
                    }
                }
                __syncthreads();
                export_planes(L.export_unit);
            }
            __syncthreads();
            MZ_TSTAMP(5);
        }
    }
    MZ_TSTAMP_FLUSH;
}

// One workgroup per block of SB samples -- or, as the re-run behind a split-precision launch (args.gate), a grid of about
// one workgroup per CU whose workgroups walk over the blocks and run the flagged ones: the usual case, nothing flagged,
// then costs a few microseconds instead of the dispatch of batch / SB workgroups of 150 KB of LDS each (33 us per
// launch at 8192 Connect4 boards, profiles/r03_bench_connect4_kernel_stats.csv before this).
template <int NT, int H, int W, int SB>
__global__ __launch_bounds__(64 * kConvWaves) void board_tower_kernel(const float* __restrict__ x, int batch, int cin0,
                                                                       uint32_t cin0_magic, int cp0, int cp1,
                                                                       TowerArgs args, TowerGather gather) {
    if (!args.gate) {
        board_tower_block<NT, H, W, SB>(x, batch, cin0, cin0_magic, cp0, cp1, args, gather, blockIdx.x);
        return;
    }
    const int n_blocks = (batch + SB - 1) / SB;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int b0 = blk * SB, last = min(batch, b0 + SB) - 1;
        int flagged = 0;                                   // (uniform over the workgroup)
        for (int e = b0 / args.gate_samples; e <= last / args.gate_samples; ++e) flagged |= args.gate[e];
        if (!flagged) continue;
        board_tower_block<NT, H, W, SB>(x, batch, cin0, cin0_magic, cp0, cp1, args, gather, blk);
        __syncthreads();                                   // (the next block zeroes the buffers this one still reads)
    }
}

// -------------------------------------------------------------------------------------------------------------------
// 16-channel towers on SMALL boards (TicTacToe's 3 x 3), boards as the matrix tiles' COLUMNS (round 3).
//
// board_tower_kernel maps (sample, position) pairs to MFMA rows: its 8 wavefronts share the rows of a workgroup's boards,
// so every layer ends in a workgroup barrier, an LDS round trip of the activations with padded planes, and a single
// accumulator chain per wavefront; a third of its time is MFMA (profiles/r02_tower_phase_stamps.jsonl).  On a 3 x 3 board
// all of that can go.  Here a WAVEFRONT owns 16 whole boards -- the 16 columns of every MFMA -- and an output position
// p is a tile: D[channel][board] (16 x 16) = sum over the taps that stay inside the board of W_tap[channel][c] x
// X[board][p + tap][c].  Consequences:
//   * a layer's whole input for the wavefront's boards is NINE 16-byte LDS reads per lane (x[p'] = four channels of
//     position p' of the lane's board), its weights NINE more (one 16 x 16 block per tap, shared by all tiles): every
//     operand of the layer's 196 MFMAs (49 (position, tap) pairs inside the board x 4 channel steps) is in registers
//     before the first one issues -- taps that fall outside the board are not multiplied by padding zeros, they are
//     skipped (40 % of the products of the padded-plane form), and nine independent accumulator chains keep the matrix
//     pipe full from one wavefront;
//   * no wavefront ever needs another wavefront's activations: NO barrier between layers, the layer's output overwrites
//     its input in LDS in place (the input is in registers), the skip connection of a residual block waits in registers;
//   * the weights of ALL layers (9 KB each) are staged into LDS once per workgroup of 128 boards.
// Arithmetic: per output the same k-ordered chain of fused multiply-adds as board_tower_kernel -- taps ascending, inside a
// tap the channel steps {g, 4 + g, 8 + g, 12 + g}, g = 0..3, then the 17th input channel (the dynamics input's action
// plane) -- minus terms that are exact zeros there (w x padding zero), and the same epilogue operations: BIT-IDENTICAL
// outputs (tests/test_gpu_board_conv.py compares the two kernels).  Reads the weights as mzmcts_board_conv_pack lays them
// out (its [tap][group][kk][cout][g] order is this kernel's operand order).
// -------------------------------------------------------------------------------------------------------------------
constexpr int kColWaves = 4;                  // wavefronts per workgroup (independent of each other): 64 boards
constexpr int kColBoardStride = 148;          // floats between two boards' [9][16] activations (+4: bank spread of 16-byte reads)
constexpr int kColActStride = 12;             // floats between two boards' 17th-channel values [9]
constexpr int kColWaveFloats = 16 * (kColBoardStride + kColActStride) + 32;   // + the boards' input rows (16 pointers)

// Where value idx = lane + 64 it of a wavefront's NCHW boards ([board][16 channels][P positions]) lives in the
// [board][position][channel] LDS image (boards STRIDE floats apart).  64 * 9 = 576 is a whole number of boards both for
// P = 9 (four) and for P = 36 (one): nine iterations later the same (channel, position) of a board further on -- so a lane
// computes nine offsets once and every later access adds a compile-time constant (an instruction offset) instead of
// two divisions.
template <int P, int STRIDE>
struct PlaneWalk {
    static constexpr int ROW = 16 * P, BOARDS_PER_9 = 576 / ROW;
    static_assert(576 % ROW == 0, "nine iterations = whole boards");
    int base[9];    // LDS float offset of iterations 0..8
    __device__ __forceinline__ int at(int it) const { return base[it % 9] + (it / 9) * BOARDS_PER_9 * STRIDE; }
};
template <int P, int STRIDE>
__device__ __forceinline__ PlaneWalk<P, STRIDE> plane_walk(int lane) {
    PlaneWalk<P, STRIDE> w;
#pragma unroll
    for (int it = 0; it < 9; ++it) {
        const int idx = lane + 64 * it;
        const int q = idx / (16 * P), cp = idx - q * (16 * P);
        const int c = cp / P, p0 = cp - c * P;
        w.base[it] = q * STRIDE + p0 * 16 + c;
    }
    return w;
}
using ColsPlaneWalk = PlaneWalk<9, kColBoardStride>;
__device__ __forceinline__ ColsPlaneWalk cols_plane_walk(int lane) { return plane_walk<9, kColBoardStride>(lane); }

// Orders this wavefront's LDS writes before its following LDS reads of other lanes' data; no other wavefront is involved.
__device__ __forceinline__ void cols_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The three layers of a head for the wavefront's 16 boards (= the 16 samples of a tile), on the matrix cores, with the
// arithmetic of conv_head_mfma_kernel (net_kernels.hip mfma_head_tiles) operation for operation -- the same k-ordered
// chains (1x1 convolution: channel steps 4 ks + kk; Linear + ELU: four partial chains over the steps s = q (mod 4), added
// in order; Linear: one chain) -- so a fused launch returns that kernel's logits bit for bit.
//   conv_y     the 1x1 convolution of the activations in `xw` ([board][position][16 channels]): D[reduced channel][board]
//              per position, kept in registers
//   finish     y -> LDS as [board][r P + p] (over the activations, which the caller no longer needs), the two Linear
//              layers, logits to global
// a head's weights as the lane's MFMA operands, asked for from global memory (L2-resident: every wavefront reads the same
// few KB) BEFORE the arithmetic that needs them: the round trips run under the tower's last products
struct ColsHeadConv {
    float conv[4];          // 1x1 convolution: W[r = lane & 15][4 ks + kk]
    float conv_bias[4];     // bias[4 kk + r']
};
template <int P>
struct ColsHeadFc {
    const float* fc1;       // Linear 1 weights in LDS: [unit][R P + 1], staged once per workgroup (cols_stage_head)
    float fc1_bias;
    float fc2[2][4];        // Linear 2, column tile n, k-step ks: W2[16 n + (lane & 15)][4 ks + kk]   (hidden <= 16)
    float fc2_bias[2];
};

__device__ __forceinline__ void cols_head_load_conv(const mzmcts_head_desc& d, int lane, ColsHeadConv& w) {
    const int i = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w.conv[ks] = i < d.reduced ? d.conv_w[i * 16 + 4 * ks + kk] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) w.conv_bias[rr] = 4 * kk + rr < d.reduced ? d.conv_b[4 * kk + rr] : 0.f;
}

constexpr int kColHeadW1Floats = 16 * (16 * 9 + 1);          // LDS floats of one head's Linear-1 weights, worst case

// Linear-1 weights of a head -> LDS [unit][R P + 1] by the whole workgroup (rows beyond `hidden` are never read)
__device__ __forceinline__ void cols_stage_head(const mzmcts_head_desc& d, float* dst, int P, int tid, int nthreads) {
    const int RP = d.reduced * P, n = d.hidden * RP;
    constexpr int kInFlight = 8;                                    // (the staging is latency: independent loads per thread)
    for (int base = tid; base < n; base += nthreads * kInFlight) {
        float v[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const int idx = base + u * nthreads;
            v[u] = idx < n ? d.fc1_w[idx] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const int idx = base + u * nthreads;
            if (idx < n) {
                const int row = idx / RP;
                dst[row * (RP + 1) + idx - row * RP] = v[u];
            }
        }
    }
}

template <int P>
__device__ __forceinline__ void cols_head_load_fc(const mzmcts_head_desc& d, const float* w1_lds, int lane, ColsHeadFc<P>& w) {
    const int i = lane & 15, kk = lane >> 4;
    const int Hd = d.hidden, O = d.outputs;
    w.fc1 = w1_lds;
    w.fc1_bias = i < Hd ? d.fc1_b[i] : 0.f;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 4 * ks + kk;
            w.fc2[n][ks] = (k < Hd && 16 * n + i < O) ? d.fc2_w[(16 * n + i) * Hd + k] : 0.f;
        }
        w.fc2_bias[n] = 16 * n + i < O ? d.fc2_b[16 * n + i] : 0.f;
    }
}

template <int P>
__device__ __forceinline__ void cols_head_conv(const ColsHeadConv& w, const float* xw, int lane, f32x4 (&acc)[P]) {
    const int i = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int p0 = 0; p0 < P; ++p0) acc[p0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int p0 = 0; p0 < P; ++p0)
            acc[p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.conv[ks], xw[i * kColBoardStride + p0 * 16 + 4 * ks + kk], acc[p0], 0, 0, 0);
}

template <int P>
__device__ __forceinline__ void cols_head_finish(const TowerHead& hd, const ColsHeadConv& cw, const ColsHeadFc<P>& w,
                                                 const f32x4 (&y)[P], float* ys, float* hs, int b0, int n_boards, int lane) {
    const mzmcts_head_desc& d = hd.d;
    const int i = lane & 15, kk = lane >> 4;
    const int R = d.reduced, RP = R * P, Hd = d.hidden, O = d.outputs;
    const int ys_stride = RP + 1, hs_stride = Hd + 1;
    // y[board = i][r P + p] = D[r = 4 kk + r'][board] + bias[r]
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        if (4 * kk + rr < R) {
#pragma unroll
            for (int p0 = 0; p0 < P; ++p0) ys[i * ys_stride + (4 * kk + rr) * P + p0] = y[p0][rr] + cw.conv_bias[rr];
        }
    }
    cols_wave_sync();
    {   // Linear + ELU: h[sample][unit] = elu(sum_k y[sample][k] W1[unit][k] + b1[unit]); hidden <= 16: one column tile,
        // four partial chains over the k-steps s = q (mod 4), added in order (mfma_head_tiles, kParts = 4)
        f32x4 part[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) part[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int steps = (RP + 3) / 4;
        for (int ks0 = 0; ks0 < steps; ks0 += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * (ks0 + q) + kk;
                const bool in = ks0 + q < steps && k < RP;
                const float yv = in ? ys[i * ys_stride + k] : 0.f;
                const float wv = (in && i < Hd) ? w.fc1[i * ys_stride + k] : 0.f;
                part[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv, wv, part[q], 0, 0, 0);
            }
        }
        f32x4 acc = part[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) acc = acc + part[q];
        if (i < Hd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[r] + w.fc1_bias;
                hs[(4 * kk + r) * hs_stride + i] = v > 0.f ? v : expf(v) - 1.f;
            }
        }
    }
    cols_wave_sync();
    {   // Linear: logits[sample][o], outputs <= 32: one or two column tiles
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const int steps = (Hd + 3) / 4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < steps) {
                const int k = 4 * ks + kk;
                const float a = k < Hd ? hs[i * hs_stride + k] : 0.f;
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    if (16 * n < O) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.fc2[n][ks], acc[n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int o = 16 * n + i;
            if (o < O) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 4 * kk + r;
                    if (q < n_boards) hd.out[static_cast<size_t>(b0 + q) * O + o] = acc[n][r] + w.fc2_bias[n];
                }
            }
        }
    }
    cols_wave_sync();   // (the caller rewrites ys / hs)
}

// HEADS = false is the production instantiation; HEADS = true also computes heads in the launch (opt-in, MZ_TOWER_HEADS=on:
// bit-identical logits, but measured SLOWER than tower + conv_head_mfma_kernel -- 65536 TicTacToe boards: 262 us against 130 +
// 73 us -- because the extra live state spills registers at two wavefronts per SIMD and a head is a serial chain of three
// small GEMMs; a separate instantiation so that the plain tower keeps its registers).
template <int H, int W, bool HEADS>
__global__ __launch_bounds__(64 * kColWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) void board_tower_cols_kernel(const float* __restrict__ x, int batch, int cin0,
                                                                          TowerArgs args, TowerGather gather, TowerHeads heads) {
    constexpr int P = H * W;
    static_assert(P == 9, "written for 3 x 3 boards: 9 positions = 9 tiles, 36 accumulator registers");
    constexpr int ROW = 16 * P;                                     // floats of a board's 16 planes (NCHW)
    constexpr int FILL = 16 * ROW / 64;                             // = 36 values per lane: the wavefront's 16 boards
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int board = lane & 15, kk = lane >> 4;
    const int b0 = (blockIdx.x * kColWaves + wave) * 16;          // first board of this wavefront
    const int n_boards = min(16, batch - b0);
    // heads inside the launch: their Linear-1 weights (the only operand too big for registers) go to LDS once per workgroup
    float* w1_lds = lds + kColWaves * kColWaveFloats;               // [mid | last0 | last1][16][R P + 1]
    if (HEADS && (heads.has_mid | heads.n_last)) {
        if (heads.has_mid) cols_stage_head(heads.mid.d, w1_lds, P, tid, 64 * kColWaves);
        if (heads.n_last > 0) cols_stage_head(heads.last0.d, w1_lds + kColHeadW1Floats, P, tid, 64 * kColWaves);
        if (heads.n_last > 1) cols_stage_head(heads.last1.d, w1_lds + 2 * kColHeadW1Floats, P, tid, 64 * kColWaves);
        __syncthreads();                                             // (the only barrier; every wavefront is still here)
    }
    if (n_boards <= 0) return;                                      // (wavefronts are independent from here on)
    float* xw = lds + wave * kColWaveFloats;                        // [16 boards][9][16] (+ pad)
    float* xact = xw + 16 * kColBoardStride;                        // [16 boards][9]: the 17th input channel
    const float** in_row = reinterpret_cast<const float**>(xact + 16 * kColActStride);

    // this lane's slice of a layer's weights, straight from the packed global buffer (L2 / L1 resident: every wavefront
    // of the launch reads the same 9 KB per layer): one 16 x 4 block per tap, + the 17th channel's column
    struct LayerWeights {
        f32x4 w[9];
    };
    auto load_weights = [&](int l, LayerWeights& out) {
        const int ng = conv_groups(args.layer[l].cin);
        const f32x4* wq = reinterpret_cast<const f32x4*>(args.layer[l].wt);
#pragma unroll
        for (int t = 0; t < 9; ++t) out.w[t] = wq[((t * ng) * 4 + kk) * 16 + board];   // (lane & 15 = the output channel here)
    };
    LayerWeights wcur;

    // ---- the tower's input: [board][position][channel] (+ the 17th channel apart); all loads issued, then stored --------
    if (lane < 16) {
        const long long b = b0 + lane;
        const float* row = nullptr;
        if (lane < n_boards)
            row = gather.pool ? gather.pool + (static_cast<size_t>(gather.parent[b]) * gather.envs + b) * gather.hidden
                              : x + static_cast<size_t>(b) * cin0 * P;
        in_row[lane] = row;
    }
    static_assert(P == 9 && ROW == 144, "ColsPlaneWalk is written for 3 x 3 boards");
    const ColsPlaneWalk walk = cols_plane_walk(lane);
    {
        float v[FILL];
        int q9[9], cp9[9];                                          // (board, offset inside the board) of iterations 0..8
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            const int idx = lane + 64 * it;
            q9[it] = idx / ROW;
            cp9[it] = idx - q9[it] * ROW;
        }
#pragma unroll
        for (int it = 0; it < FILL; ++it) {                         // idx = (board q, channel c, position p), NCHW order
            const float* row = in_row[q9[it % 9] + 4 * (it / 9)];   // (64 * 9 = 4 * ROW: see ColsPlaneWalk)
            v[it] = row ? row[cp9[it % 9]] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < FILL; ++it) xw[walk.at(it)] = v[it];
#pragma unroll
        for (int it = 0; it < (16 * P + 63) / 64; ++it) {
            const int idx = lane + 64 * it;
            if (idx < 16 * P) {
                const int q = idx / P, p0 = idx - q * P;
                const float* row = in_row[q];
                float a = 0.f;
                if (row && cin0 == 17)
                    a = gather.pool ? static_cast<float>(gather.action[b0 + q]) / gather.action_space : row[16 * P + p0];
                xact[q * kColActStride + p0] = a;
            }
        }
    }

    load_weights(0, wcur);
    float wact[9];                                                  // layer 0 only (tower_cols_applies: later layers read 16 channels)
#pragma unroll
    for (int t = 0; t < 9; ++t)
        wact[t] = cin0 == 17 ? reinterpret_cast<const f32x4*>(args.layer[0].wt)[((t * 2 + 1) * 4 + kk) * 16 + board][0] : 0.f;

    auto export_planes = [&](float* out) {                          // NCHW [batch][16][P]: consecutive lanes, consecutive addresses
        float v[FILL];
#pragma unroll
        for (int it = 0; it < FILL; ++it) v[it] = xw[walk.at(it)];
#pragma unroll
        for (int it = 0; it < FILL; ++it) {
            const int idx = lane + 64 * it;
            if (idx < n_boards * ROW) out[static_cast<size_t>(b0) * ROW + idx] = v[it];
        }
    };

    f32x4 xr[P], prev[P];
#pragma unroll
    for (int p0 = 0; p0 < P; ++p0) prev[p0] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < args.n_layers; ++l) {
        const TowerLayer& L = args.layer[l];
        const bool has_act = l == 0 && cin0 == 17;                  // 17 input channels: the dynamics input's action plane
        // operands: this lane's four channels of every position of its board (the weights were asked for a layer ago)
#pragma unroll
        for (int p0 = 0; p0 < P; ++p0) xr[p0] = *reinterpret_cast<const f32x4*>(xw + board * kColBoardStride + p0 * 16 + 4 * kk);
        float xav[P];
#pragma unroll
        for (int p0 = 0; p0 < P; ++p0) xav[p0] = 0.f;
        if (has_act) {      // channel 16 = step g = 0 of the second channel group; lanes kk > 0 carry that group's zero padding
#pragma unroll
            for (int p0 = 0; p0 < P; ++p0) xav[p0] = kk == 0 ? xact[board * kColActStride + p0] : 0.f;
        }
        const float4 s4 = *reinterpret_cast<const float4*>(L.scale + 4 * kk);
        const float4 h4v = *reinterpret_cast<const float4*>(L.shift + 4 * kk);
        const f32x4 sc = {s4.x, s4.y, s4.z, s4.w}, sh = {h4v.x, h4v.y, h4v.z, h4v.w};
        const int skip = L.skip, relu = L.relu;

        f32x4 acc[P];
#pragma unroll
        for (int p0 = 0; p0 < P; ++p0) acc[p0] = f32x4{0.f, 0.f, 0.f, 0.f};
        // taps outermost: consecutive MFMAs belong to different tiles (independent chains); per tile the order is taps
        // ascending, channel steps ascending, then the 17th channel -- board_tower_kernel's chain without its zero terms
        auto products = [&](auto with_act) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int p0 = 0; p0 < P; ++p0) {
                        const int y = p0 / W + dy, xx = p0 % W + dx;
                        if (y < 0 || y >= H || xx < 0 || xx >= W) continue;      // (compile-time: the loops are unrolled)
                        acc[p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur.w[t][g], xr[y * W + xx][g], acc[p0], 0, 0, 0);
                    }
                }
                if constexpr (decltype(with_act)::value) {
#pragma unroll
                    for (int p0 = 0; p0 < P; ++p0) {
                        const int y = p0 / W + dy, xx = p0 % W + dx;
                        if (y < 0 || y >= H || xx < 0 || xx >= W) continue;
                        acc[p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wact[t], xav[y * W + xx], acc[p0], 0, 0, 0);
                    }
                }
            }
        };
        if (has_act) products(std::true_type{}); else products(std::false_type{});
        // ---- epilogue, in place: D[channel = 4 kk + r][board] of tile p -> [board][p][4 kk .. 4 kk + 3] ----------------
        f32x4 v[P];
#pragma unroll
        for (int p0 = 0; p0 < P; ++p0) {
            v[p0] = acc[p0] * sc + sh;
            if (skip) v[p0] = v[p0] + prev[p0];                      // the residual block's input: last layer's operand
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[p0][r] = v[p0][r] < 0.f ? 0.f : v[p0][r];
            }
            prev[p0] = xr[p0];
        }
        auto store_planes = [&]() {
            if (board < n_boards) {
#pragma unroll
                for (int p0 = 0; p0 < P; ++p0) *reinterpret_cast<f32x4*>(xw + board * kColBoardStride + p0 * 16 + 4 * kk) = v[p0];
            }
        };
        store_planes();
        if (L.export_raw) export_planes(L.export_raw);
        // heads reading this layer's output (host-checked: only where the activations are rewritten afterwards -- an
        // export_unit layer -- or no longer needed -- the last layer --, because y and h take their place in LDS)
        if constexpr (HEADS) {
            float* ys = xw;                                          // [16][R P + 1] <= 16 x 145 floats
            float* hs = xw + 16 * (16 * P + 1);                      // [16][hidden + 1] <= 272 floats: the rest of the region
            if (heads.has_mid && heads.mid.layer == l) {
                ColsHeadConv c0;
                ColsHeadFc<P> f0;
                cols_head_load_conv(heads.mid.d, lane, c0);
                cols_head_load_fc<P>(heads.mid.d, w1_lds, lane, f0);         // (asked for before the arithmetic that needs them)
                cols_wave_sync();                                    // (the planes just stored, read across lanes)
                f32x4 y0[P];
                cols_head_conv<P>(c0, xw, lane, y0);
                cols_wave_sync();                                    // (every lane has read the planes: y may overwrite them)
                cols_head_finish<P>(heads.mid, c0, f0, y0, ys, hs, b0, n_boards, lane);
            }
            if (heads.n_last > 0 && l == args.n_layers - 1) {
                ColsHeadConv c0, c1;
                ColsHeadFc<P> f0, f1;
                cols_head_load_conv(heads.last0.d, lane, c0);
                if (heads.n_last > 1) cols_head_load_conv(heads.last1.d, lane, c1);
                cols_head_load_fc<P>(heads.last0.d, w1_lds + kColHeadW1Floats, lane, f0);
                cols_wave_sync();
                f32x4 y0[P], y1[P];
                cols_head_conv<P>(c0, xw, lane, y0);
                if (heads.n_last > 1) cols_head_conv<P>(c1, xw, lane, y1);
                cols_wave_sync();
                if (heads.n_last > 1) cols_head_load_fc<P>(heads.last1.d, w1_lds + 2 * kColHeadW1Floats, lane, f1);   // in flight under the first head's layers
                cols_head_finish<P>(heads.last0, c0, f0, y0, ys, hs, b0, n_boards, lane);
                if (heads.n_last > 1) cols_head_finish<P>(heads.last1, c1, f1, y1, ys, hs, b0, n_boards, lane);
            }
        }
        if (L.export_unit) {
            // per (board, channel) plane: (x - min) / span -- models.py:525-549, the operations of unit_rescale_kernel in
            // the same order; a plane's nine values are element r of this lane's nine registers
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float lo = v[0][r], hi = lo;
#pragma unroll
                for (int p0 = 1; p0 < P; ++p0) {
                    const float u = v[p0][r];
                    lo = (u < lo || u != u) ? u : lo;
                    hi = (u > hi || u != u) ? u : hi;
                }
                float span = hi - lo;
                if (span < 1e-5f) span = span + 1e-5f;
#pragma unroll
                for (int p0 = 0; p0 < P; ++p0) v[p0][r] = (v[p0][r] - lo) / span;
            }
            store_planes();
            export_planes(L.export_unit);
        }
        if (l + 1 < args.n_layers) load_weights(l + 1, wcur);        // in flight under the next layer's operand reads
    }
}

// -------------------------------------------------------------------------------------------------------------------
// The same idea for boards made of several 3 x 3 PATCHES (6 x 6: the 84 x 84 configuration's hidden state): an MFMA
// column is a (board, patch) pair -- 16 columns = 4 boards x 4 patches per wavefront --, a tile one of the patch's nine
// positions, and a lane's operands are the 5 x 5 halo around its patch (cells outside the board are zeros, cells of the
// neighbouring patches come from the same wavefront's LDS: all of a board's patches live in one wavefront, so there is
// still no barrier and the layer still overwrites its input in place once every lane has read its halo).  No product is
// skipped here (which halo cells are outside depends on the lane's patch), so the arithmetic is board_tower_kernel's chain
// term for term: bit-identical outputs.  The residual block's input waits in a second LDS buffer (the registers are taken
// by the 25 halo operands).
// -------------------------------------------------------------------------------------------------------------------
template <int H, int W>
struct PatchGeometry {
    static constexpr int P = H * W;
    static constexpr int NPX = W / 3, NPY = H / 3, NP = NPX * NPY;    // patches per board
    static constexpr int BPW = 16 / NP;                                // boards per wavefront
    static constexpr int BS = P * 16 + 4;                              // floats between two boards' [P][16] activations
    static constexpr int WAVE_FLOATS = 2 * BPW * BS + BPW * P + 32;    // activations | skip | 17th channel | input rows
    static_assert(H % 3 == 0 && W % 3 == 0 && 16 % NP == 0 && NP > 1, "boards of 2, 4, 8 or 16 patches of 3 x 3");
    static_assert((BPW * 16 * P) % 64 == 0, "the fill walks whole wavefronts");
};

template <int H, int W>
__global__ __launch_bounds__(64 * kColWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) void board_tower_patch_kernel(
    const float* __restrict__ x, int batch, int cin0, TowerArgs args, TowerGather gather) {
    using G = PatchGeometry<H, W>;
    constexpr int P = G::P, NP = G::NP, NPX = G::NPX, BPW = G::BPW, BS = G::BS;
    constexpr int ROW = 16 * P;
    constexpr int FILL = BPW * ROW / 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kk = lane >> 4;
    const int bl = col / NP, patch = col % NP;                       // this lane's board (in the wavefront) and patch
    const int py = (patch / NPX) * 3, px = (patch % NPX) * 3;
    const int b0 = (blockIdx.x * kColWaves + wave) * BPW;
    const int n_boards = min(BPW, batch - b0);
    if (n_boards <= 0) return;                                      // (no barrier anywhere: wavefronts are independent)
    float* xw = lds + wave * G::WAVE_FLOATS;                         // [BPW][P][16]
    float* xskip = xw + BPW * BS;                                    // the residual block's input, same layout
    float* xact = xskip + BPW * BS;                                  // [BPW][P]: the 17th input channel
    const float** in_row = reinterpret_cast<const float**>(xact + BPW * P);

    if (lane < BPW) {
        const long long b = b0 + lane;
        const float* row = nullptr;
        if (lane < n_boards)
            row = gather.pool ? gather.pool + (static_cast<size_t>(gather.parent[b]) * gather.envs + b) * gather.hidden
                              : x + static_cast<size_t>(b) * cin0 * P;
        in_row[lane] = row;
    }
    static_assert(ROW == 576, "nine iterations of 64 lanes = one board (PlaneWalk)");
    const PlaneWalk<P, BS> walk = plane_walk<P, BS>(lane);
    {
        float v[FILL];
#pragma unroll
        for (int it = 0; it < FILL; ++it) {                         // idx = (board q, channel c, position p), NCHW order
            const float* row = in_row[it / 9];                      // (board it / 9, offset lane + 64 (it % 9) inside it)
            v[it] = row ? row[lane + 64 * (it % 9)] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < FILL; ++it) {
            xw[walk.at(it)] = v[it];
        }
        for (int idx = lane; idx < BPW * P; idx += 64) {
            const int q = idx / P, p0 = idx - q * P;
            const float* row = in_row[q];
            float a = 0.f;
            if (row && cin0 == 17)
                a = gather.pool ? static_cast<float>(gather.action[b0 + q]) / gather.action_space : row[16 * P + p0];
            xact[idx] = a;
        }
    }
    f32x4 wcur[9];
    auto load_weights = [&](int l) {
        const int ng = conv_groups(args.layer[l].cin);
        const f32x4* wq = reinterpret_cast<const f32x4*>(args.layer[l].wt);
#pragma unroll
        for (int t = 0; t < 9; ++t) wcur[t] = wq[((t * ng) * 4 + kk) * 16 + col];      // (lane & 15 = the output channel here)
    };
    load_weights(0);
    float wact[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
        wact[t] = cin0 == 17 ? reinterpret_cast<const f32x4*>(args.layer[0].wt)[((t * 2 + 1) * 4 + kk) * 16 + col][0] : 0.f;

    auto export_planes = [&](float* out) {
        float v[FILL];
#pragma unroll
        for (int it = 0; it < FILL; ++it) v[it] = xw[walk.at(it)];
#pragma unroll
        for (int it = 0; it < FILL; ++it) {
            const int idx = lane + 64 * it;
            if (idx < n_boards * ROW) out[static_cast<size_t>(b0) * ROW + idx] = v[it];
        }
    };
    // halo cell (hy, hx) in 0..4 of this lane's patch: board position, or outside the board
    auto cell = [&](int hy, int hx, bool& inside) {
        const int y = py + hy - 1, xx = px + hx - 1;
        inside = y >= 0 && y < H && xx >= 0 && xx < W;
        return inside ? y * W + xx : 0;
    };

    for (int l = 0; l < args.n_layers; ++l) {
        const TowerLayer& L = args.layer[l];
        const bool has_act = l == 0 && cin0 == 17;
        const bool next_skips = l + 1 < args.n_layers && args.layer[l + 1].skip;
        f32x4 xr[25];
#pragma unroll
        for (int h = 0; h < 25; ++h) {
            bool inside;
            const int pos = cell(h / 5, h % 5, inside);
            const f32x4 v = *reinterpret_cast<const f32x4*>(xw + bl * BS + pos * 16 + 4 * kk);
            xr[h] = inside ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (next_skips) {                                            // this layer's input is the next layer's skip connection
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int pos = (py + t / 3) * W + px + t % 3;
                *reinterpret_cast<f32x4*>(xskip + bl * BS + pos * 16 + 4 * kk) = xr[(t / 3 + 1) * 5 + t % 3 + 1];
            }
        }
        const float4 s4 = *reinterpret_cast<const float4*>(L.scale + 4 * kk);
        const float4 h4v = *reinterpret_cast<const float4*>(L.shift + 4 * kk);
        const f32x4 sc = {s4.x, s4.y, s4.z, s4.w}, sh = {h4v.x, h4v.y, h4v.z, h4v.w};
        const int skip = L.skip, relu = L.relu;

        f32x4 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto products = [&](auto with_act) {
            float xav[25];
            if constexpr (decltype(with_act)::value) {
#pragma unroll
                for (int h = 0; h < 25; ++h) {
                    bool inside;
                    const int pos = cell(h / 5, h % 5, inside);
                    const float a = xact[bl * P + pos];
                    xav[h] = (inside && kk == 0) ? a : 0.f;
                }
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[tap][g], xr[(t / 3 + tap / 3) * 5 + t % 3 + tap % 3][g], acc[t], 0, 0, 0);
                if constexpr (decltype(with_act)::value) {
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wact[tap], xav[(t / 3 + tap / 3) * 5 + t % 3 + tap % 3], acc[t], 0, 0, 0);
                }
            }
        };
        if (has_act) products(std::true_type{}); else products(std::false_type{});

#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pos = (py + t / 3) * W + px + t % 3;
            f32x4 v = acc[t] * sc + sh;
            if (skip) v = v + *reinterpret_cast<const f32x4*>(xskip + bl * BS + pos * 16 + 4 * kk);
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
            }
            if (bl < n_boards) *reinterpret_cast<f32x4*>(xw + bl * BS + pos * 16 + 4 * kk) = v;
        }
        if (l + 1 < args.n_layers) load_weights(l + 1);
        if (L.export_raw || L.export_unit) {
            cols_wave_sync();                                        // (planes written by other lanes of the wavefront)
            if (L.export_raw) export_planes(L.export_raw);
            if (L.export_unit) {
                // per (board, channel) plane, sequentially over its positions: models.py:525-549, unit_rescale_kernel's operations
                for (int q = lane; q < n_boards * 16; q += 64) {
                    float* plane = xw + (q >> 4) * BS + (q & 15);
                    float lo = plane[0], hi = lo;
                    for (int p0 = 1; p0 < P; ++p0) {
                        const float u = plane[p0 * 16];
                        lo = (u < lo || u != u) ? u : lo;
                        hi = (u > hi || u != u) ? u : hi;
                    }
                    float span = hi - lo;
                    if (span < 1e-5f) span = span + 1e-5f;
                    for (int p0 = 0; p0 < P; ++p0) plane[p0 * 16] = (plane[p0 * 16] - lo) / span;
                }
                cols_wave_sync();
                export_planes(L.export_unit);
            }
        }
        cols_wave_sync();                                            // (the next layer's halo reads cross lanes)
    }
}

static int launch_board_tower_patch66(const float* x, int batch, int cin0, const TowerArgs& args, hipStream_t stream,
                                      const TowerGather& gather) {
    using G = PatchGeometry<6, 6>;
    for (int l = 0; l < args.n_layers; ++l)
        if (reinterpret_cast<uintptr_t>(args.layer[l].wt) & 15u) return MZMCTS_ERR_INVALID;   // (read as float4)
    const size_t lds = sizeof(float) * static_cast<size_t>(kColWaves) * G::WAVE_FLOATS;
    auto kernel = board_tower_patch_kernel<6, 6>;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(lds)) != hipSuccess)
        return MZMCTS_ERR_HIP;
    const int per_group = G::BPW * kColWaves;
    kernel<<<dim3(static_cast<unsigned>((batch + per_group - 1) / per_group)), dim3(64 * kColWaves), lds, stream>>>(x, batch, cin0,
                                                                                                                  args, gather);
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

// -------------------------------------------------------------------------------------------------------------------
// The heads of a 16-channel 3 x 3 network as a launch of their own in the board-column shape (round 3): a wavefront takes
// 16 boards, brings their NCHW planes into LDS as [board][position][channel] with coalesced loads, and runs cols_head_* --
// conv_head_mfma_kernel's arithmetic operation for operation, so the logits are that kernel's bit for bit
// (tests/test_gpu_net.py) -- with the Linear-1 weights staged once per workgroup.  One head reading tensor x0 (the reward
// head on the raw dynamics output) and up to two reading x1 (value and policy on the prediction features): the pattern
// of recurrent_inference (models.py:467-480, 500-522).
// -------------------------------------------------------------------------------------------------------------------
struct HeadsColsArgs {
    const float* x[3];
    TowerHead head[3];
    int32_t n;
};

// one head per workgroup (blockIdx.y): three times the wavefronts of a launch that ran a board's heads one after the
// other, each with a third of the dependent chain (the heads are latency, not arithmetic: ~80 MFMAs per 16 boards)
__global__ __launch_bounds__(64 * kColWaves) void board_heads_cols_kernel(HeadsColsArgs a, int batch) {
    constexpr int P = 9, ROW = 16 * P, FILL = 16 * ROW / 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (static selection of the head: the argument segment is addressed with constant offsets)
    const TowerHead hd = blockIdx.y == 0 ? a.head[0] : (blockIdx.y == 1 ? a.head[1] : a.head[2]);
    const float* x = blockIdx.y == 0 ? a.x[0] : (blockIdx.y == 1 ? a.x[1] : a.x[2]);
    float* w1_lds = lds + kColWaves * kColWaveFloats;
    cols_stage_head(hd.d, w1_lds, P, tid, 64 * kColWaves);
    __syncthreads();
    const int b0 = (blockIdx.x * kColWaves + wave) * 16;
    const int n_boards = min(16, batch - b0);
    if (n_boards <= 0) return;
    float* xw = lds + wave * kColWaveFloats;
    float* ys = xw;
    float* hs = xw + 16 * (16 * P + 1);
    ColsHeadConv c0;
    ColsHeadFc<P> f0;
    cols_head_load_conv(hd.d, lane, c0);
    cols_head_load_fc<P>(hd.d, w1_lds, lane, f0);
    {                                                               // 16 boards' planes, contiguous in global memory
        float v[FILL];
#pragma unroll
        for (int it = 0; it < FILL; ++it) {
            const int idx = lane + 64 * it;
            v[it] = idx < n_boards * ROW ? x[static_cast<size_t>(b0) * ROW + idx] : 0.f;
        }
        const ColsPlaneWalk walk = cols_plane_walk(lane);
#pragma unroll
        for (int it = 0; it < FILL; ++it) xw[walk.at(it)] = v[it];
    }
    cols_wave_sync();
    f32x4 y0[P];
    cols_head_conv<P>(c0, xw, lane, y0);
    cols_wave_sync();
    cols_head_finish<P>(hd, c0, f0, y0, ys, hs, b0, n_boards, lane);
}

// mzmcts_conv_heads_multi's fast path (net_kernels.hip calls it first): MZMCTS_ERR_INVALID = this launch does not cover
// the shapes, the caller takes conv_head_mfma_kernel.
int launch_board_heads_cols(const float* const* xs, const mzmcts_head_desc* heads, int n_heads, float* const* outs,
                            int64_t batch, hipStream_t stream) {
    const char* env = std::getenv("MZ_HEADS_COLS");
    if (env && std::string(env) == "off") return MZMCTS_ERR_INVALID;
    if (n_heads < 1 || n_heads > 3 || batch > 0x3fffffff) return MZMCTS_ERR_INVALID;
    HeadsColsArgs a{};
    a.n = n_heads;
    for (int h = 0; h < n_heads; ++h) {
        const mzmcts_head_desc& d = heads[h];
        if (!xs[h] || !outs[h] || !d.conv_w || !d.conv_b || !d.fc1_w || !d.fc1_b || !d.fc2_w || !d.fc2_b || d.channels != 16 ||
            d.plane != 9 || d.reduced < 1 || d.reduced > 16 || d.hidden < 1 || d.hidden > 16 || d.outputs < 1 || d.outputs > 32)
            return MZMCTS_ERR_INVALID;
        a.x[h] = xs[h];
        a.head[h] = TowerHead{d, outs[h], 0, 0};
    }
    if (batch == 0) return MZMCTS_OK;
    const size_t lds = sizeof(float) * (static_cast<size_t>(kColWaves) * kColWaveFloats + kColHeadW1Floats);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(board_heads_cols_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
        return MZMCTS_ERR_HIP;
    const int per_group = 16 * kColWaves;
    board_heads_cols_kernel<<<dim3(static_cast<unsigned>((batch + per_group - 1) / per_group), static_cast<unsigned>(n_heads)),
                              dim3(64 * kColWaves), lds, stream>>>(a, static_cast<int>(batch));
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

static bool tower_cols_applies(int channels, int height, int width, int cin0, const TowerArgs& args) {
    const char* env = std::getenv("MZ_TOWER_COLS");              // "off": the row-tile kernel (A/B runs, the equality test)
    const bool off = env && std::string(env) == "off";
    const bool board = (height == 3 && width == 3) || (height == 6 && width == 6);
    if (off || channels != 16 || !board || (cin0 != 16 && cin0 != 17)) return false;
    for (int l = 1; l < args.n_layers; ++l)
        if (args.layer[l].cin != 16) return false;
    return true;
}

static int launch_board_tower_cols(const float* x, int batch, int cin0, const TowerArgs& args, hipStream_t stream,
                                   const TowerGather& gather, const TowerHeads& heads = TowerHeads{}) {
    static_assert(16 * (16 * 9 + 1) + 16 * 17 <= kColWaveFloats, "a head's y and h take the place of the activations");
    auto head_ok = [&](const TowerHead& hd) {
        const mzmcts_head_desc& d = hd.d;
        return hd.out && d.conv_w && d.conv_b && d.fc1_w && d.fc1_b && d.fc2_w && d.fc2_b && d.channels == 16 && d.plane == 9 &&
               d.reduced >= 1 && d.reduced <= 16 && d.hidden >= 1 && d.hidden <= 16 && d.outputs >= 1 && d.outputs <= 32;
    };
    if (heads.has_mid && !(head_ok(heads.mid) && heads.mid.layer >= 0 && heads.mid.layer < args.n_layers - 1 &&
                           args.layer[heads.mid.layer].export_unit))
        return MZMCTS_ERR_INVALID;
    if (heads.n_last < 0 || heads.n_last > 2 || (heads.n_last > 0 && !head_ok(heads.last0)) || (heads.n_last > 1 && !head_ok(heads.last1)))
        return MZMCTS_ERR_INVALID;
    const bool with_heads = heads.has_mid || heads.n_last > 0;
    const size_t lds = sizeof(float) * (static_cast<size_t>(kColWaves) * kColWaveFloats + (with_heads ? 3 * kColHeadW1Floats : 0));
    const int per_group = 16 * kColWaves;
    const dim3 grid(static_cast<unsigned>((batch + per_group - 1) / per_group)), block(64 * kColWaves);
    if (with_heads) {
        auto kernel = board_tower_cols_kernel<3, 3, true>;
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   static_cast<int>(lds)) != hipSuccess)
            return MZMCTS_ERR_HIP;
        kernel<<<grid, block, lds, stream>>>(x, batch, cin0, args, gather, heads);
    } else {
        board_tower_cols_kernel<3, 3, false><<<grid, block, lds, stream>>>(x, batch, cin0, args, gather, heads);
    }
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

template <int NT, int H, int W, int SB>
static int launch_board_tower(const float* x, int batch, int cin0, const TowerArgs& args, hipStream_t stream,
                              const TowerGather& gather = TowerGather{}) {
    constexpr int PP = (H + 2) * (W + 1) + 1;
    int cp0 = conv_groups(cin0) * kConvGroup + 4;
    int cp1 = 4;
    for (int l = 0; l < args.n_layers; ++l) {             // layer l reads buffer l & 1
        int& cp = (l & 1) ? cp1 : cp0;
        cp = std::max(cp, conv_groups(args.layer[l].cin) * kConvGroup + 4);
    }
    cp0 = std::max(cp0, 16 * NT + 4);                     // outputs (16 NT channels) land in either buffer
    cp1 = std::max(cp1, 16 * NT + 4);
    const size_t lds = sizeof(float) * static_cast<size_t>(SB) * PP * (cp0 + cp1) + (sizeof(float*) + sizeof(float)) * SB;
    if (lds > 160 * 1024) return MZMCTS_ERR_INVALID;
    auto kernel = board_tower_kernel<NT, H, W, SB>;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(lds)) != hipSuccess)
        return MZMCTS_ERR_HIP;
    const int blocks = (batch + SB - 1) / SB;
    const dim3 grid(static_cast<unsigned>(args.gate ? std::min(blocks, 256) : blocks)), block(64 * kConvWaves);
    kernel<<<grid, block, lds, stream>>>(x, batch, cin0, 0xFFFFFFFFu / static_cast<uint32_t>(cin0) + 1u, cp0, cp1, args, gather);
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

// -------------------------------------------------------------------------------------------------------------------
// The same tower on the 16-bit matrix path at fp32 accuracy ("split" precision): every fp32 operand is carried as TWO
// fp16 numbers, x * 2^s = h0 + h1 (h0 = fp16(x 2^s), h1 = fp16(x 2^s - h0): 22 significant bits), and a product as
//     a b  ~  (a0 b0) + (a0 b1 + a1 b0)          (three v_mfma_f32_16x16x32_f16, fp32 accumulation; a1 b1 ~ 2^-22 dropped)
// with the three partial sums kept in two accumulators (large / small terms).  The representation error of a length-576
// dot product is below that of an fp32 fmaf chain (measured: rms 0.7e-8 vs 3.3e-8 of sum |a b|), at 3 x 16 cycles per 32
// input channels instead of 8 x 32: 5.3 x fewer matrix-pipe cycles than the fp32 form, which on this chip runs into
// its power limit (the fp32 tower sustains ~1.1 GHz-equivalent).  Powers of two (activations x 8, weights x 64) keep
// the low halves out of the fp16 subnormal range; they are removed exactly in the epilogue.  |activation| must stay
// below 8188 (fp16 range / 8): larger values become inf / NaN, never a silently wrong number.
// Activations live in LDS as [position][2 halves][channel] fp16; the skip connection and the exports read h0 + h1.
// The dynamics input's action plane (one constant per sample, models.py:553-568) is not convolved: its contribution
// a * sum over the taps inside the board of w[n][C][tap] comes from a table made at pack time (fp32).
// -------------------------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr float kActScale = 8.f, kWtScale = 64.f;
constexpr int kSplitGroup = 32;   // input channels per MFMA (K of v_mfma_f32_16x16x32_f16)

__host__ __device__ inline int split_groups(int cin) { return (cin + kSplitGroup - 1) / kSplitGroup; }
__host__ __device__ inline int64_t split_packed_halfs(int cin, int cout) {   // [9 NG + 2 spare][2 q][4 kk][cout][8 j]
    return static_cast<int64_t>(9 * split_groups(cin) + 2) * 2 * 4 * cout * 8;
}

// wh[((((tap * NG + grp) * 2 + q) * 4 + kk) * cout + n) * 8 + j] = half q of 64 * w[n][grp * 32 + 8 kk + j][tap]
// over the first `cin_conv` input channels of a [cout, cin_total, 3, 3] weight (0 beyond, and in the spare groups).
__global__ __launch_bounds__(256) void board_conv_pack_split_kernel(const float* __restrict__ w, _Float16* __restrict__ wh,
                                                                    int cin_total, int cin_conv, int cout) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= split_packed_halfs(cin_conv, cout)) return;
    const int j = static_cast<int>(i & 7);
    const int n = static_cast<int>((i >> 3) % cout);
    const int kk = static_cast<int>(((i >> 3) / cout) & 3);
    const int q = static_cast<int>(((i >> 3) / cout / 4) & 1);
    const int tg = static_cast<int>((i >> 3) / cout / 8);
    const int ng = split_groups(cin_conv);
    const int tap = tg / ng, grp = tg % ng;
    const int ci = grp * kSplitGroup + 8 * kk + j;
    float v = (tap < 9 && ci < cin_conv) ? w[(static_cast<size_t>(n) * cin_total + ci) * 9 + tap] * kWtScale : 0.f;
    const _Float16 h0 = static_cast<_Float16>(v);
    const _Float16 h1 = static_cast<_Float16>(v - static_cast<float>(h0));
    wh[i] = q ? h1 : h0;
}

// table[n * P + p] = sum over the taps whose source position lies inside the H x W board of w[n][ci][tap]
__global__ __launch_bounds__(256) void board_conv_const_plane_kernel(const float* __restrict__ w, float* __restrict__ table,
                                                                     int cin_total, int ci, int cout, int H, int W) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cout * H * W) return;
    const int p = i % (H * W), n = i / (H * W);
    const int y = p / W, xx = p % W;
    float acc = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xs = xx + tap % 3 - 1;
        if (yy >= 0 && yy < H && xs >= 0 && xs < W) acc += w[(static_cast<size_t>(n) * cin_total + ci) * 9 + tap];
    }
    table[i] = acc;
}

struct SplitLayer {
    const _Float16* wh;        // packed split weights
    const float* scale;
    const float* shift;
    const float* const_table;  // layer 0 only: the constant plane's contribution per (channel, position), or null
    float* export_raw;
    float* export_unit;
    int32_t cin;               // channels convolved (without the constant plane)
    int32_t relu, skip, pad;
};
struct SplitArgs {
    SplitLayer layer[kMaxTowerLayers];
    int32_t n_layers;
    // not null: gate[i] = 1 if a value of workgroup i's samples left the fp16 range (or was not finite), else 0, written
    // by every launch; gate[number of workgroups] counts the flagged workgroups since it was last cleared
    int32_t* gate;
    // != 0: the wavefront in the SIMD's even slot runs at a raised issue priority (see the kernel)
    int32_t slot_priority;
};

template <int H, int W, int SB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void board_tower_split_kernel(const float* __restrict__ x, int batch, int cin0,
                                                                             int const_plane, uint32_t cin_load_magic,
                                                                             int cph0, int cph1, SplitArgs args,
                                                                             TowerGather gather) {
    constexpr int P = H * W;
    constexpr int PW = W + 1;
    constexpr int PP = (H + 2) * PW + 1;
    constexpr int ROWS = SB * P;
    constexpr int MT = (ROWS + 15) / 16;
    constexpr int RG = WAVES / 2;                       // row groups; wave = (column pair, row group)
    constexpr int MTW = (MT + RG - 1) / RG;
    constexpr int COUT = 64;
    constexpr int THREADS = 64 * WAVES;
    static_assert(WAVES >= 2 && WAVES % 2 == 0 && THREADS >= P && THREADS >= SB, "workgroup shape");
    extern __shared__ __attribute__((aligned(16))) _Float16 hl[];   // buffer 0 [SB*PP][2][cph0] | buffer 1 [SB*PP][2][cph1] | consts

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b0 = blockIdx.x * SB;
    const int n_samples = min(SB, batch - b0);
    // Two workgroups share a CU (one wavefront of each per SIMD).  Left alone they fall into step -- whoever is ahead
    // meets the other in the MFMA loop, where the one behind then runs alone at twice the rate and catches up -- and the
    // matrix pipe idles through both epilogues.  A fixed priority by wavefront slot (HW_ID bits 3:0) breaks the symmetry:
    // the even slot's workgroup never waits, the odd slot's multiplies while the even one is in its epilogue / barrier /
    // export phases.
    if (WAVES < 8 && args.slot_priority) {
        if ((__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) == 0) __builtin_amdgcn_s_setprio(3);
    }
    MZ_TSTAMP_DECL
    const int buf1_at = SB * PP * 2 * cph0;             // (offsets into hl, so that every access stays an LDS access)
    float* aconst = reinterpret_cast<float*>(hl + SB * PP * 2 * (cph0 + cph1));   // [SB] the constant plane's value
    const float** in_row = reinterpret_cast<const float**>(aconst + SB);            // [SB] where sample q's input planes start
    // A thread of the plane-wise passes (input fill, exports) keeps ONE board position and walks over the (sample, channel)
    // planes, TPP planes per pass: consecutive threads touch consecutive addresses of global memory, and the position's
    // arithmetic is done once.
    constexpr int TPP = THREADS / P;
    constexpr int WALKERS = TPP * P;

    // |value| * 8 must stay inside the fp16 range: anything else (inf / NaN included) flags the workgroup's samples, and
    // the exact-fp32 tower re-runs them behind this launch (SplitArgs::gate)
    constexpr float kHalfMax = 65504.f;
    int overflow = 0;
    auto plane_pos = [&](int p) { return (p / W + 1) * PW + (p % W) + 1; };
    auto store_val = [&](int b, int cph, int pos, int n, float v) {       // b: the buffer's offset in hl
        const float vs = v * kActScale;
        const _Float16 h0 = static_cast<_Float16>(vs);
        const _Float16 h1 = static_cast<_Float16>(vs - static_cast<float>(h0));
        hl[b + pos * 2 * cph + n] = h0;
        hl[b + pos * 2 * cph + cph + n] = h1;
    };
    auto load_val = [&](int b, int cph, int pos, int n) {
        return (static_cast<float>(hl[b + pos * 2 * cph + n]) + static_cast<float>(hl[b + pos * 2 * cph + cph + n])) *
               (1.0f / kActScale);
    };

    // ---- zero both buffers, then the tower's input (without the constant plane) into buffer 0 ---------------------
    const int cin_load = cin0 - const_plane;
    {
        const int bytes = (SB * PP * 2 * (cph0 + cph1)) * 2;          // a multiple of 16
        float4* z = reinterpret_cast<float4*>(hl);
        for (int i = tid; i < bytes / 16; i += THREADS) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < SB) {
            // (gathered input: rows of the hidden-state pool, the constant plane is action / action_space)
            const long long b = b0 + tid;
            const float* row = nullptr;
            float plane_value = 0.f;
            if (tid < n_samples) {
                row = gather.pool ? gather.pool + (static_cast<size_t>(gather.parent[b]) * gather.envs + b) * gather.hidden
                                  : x + static_cast<size_t>(b) * cin0 * P;
                if (const_plane)
                    plane_value = gather.pool ? static_cast<float>(gather.action[b]) / gather.action_space
                                              : row[static_cast<size_t>(cin_load) * P];
            }
            in_row[tid] = row;
            aconst[tid] = plane_value;
        }
    }
    __syncthreads();
    MZ_TSTAMP(0);
    if (tid < WALKERS) {
        const int p = tid % P;
        const int at_p = plane_pos(p) * 2 * cph0;
        const int planes = n_samples * cin_load;         // plane sc = sample * cin_load + channel
        for (int sc0 = tid / P; sc0 < planes; sc0 += 4 * TPP) {
            float v[4];
            int at[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sc = sc0 + k * TPP;
                at[k] = -1;
                v[k] = 0.f;
                if (sc < planes) {
                    const int sidx = static_cast<int>(__umulhi(static_cast<uint32_t>(sc), cin_load_magic));
                    const int ci = sc - sidx * cin_load;
                    v[k] = in_row[sidx][ci * P + p];
                    at[k] = sidx * PP * 2 * cph0 + at_p + ci;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (at[k] >= 0) {
                    const float vs = v[k] * kActScale;
                    overflow |= !(__builtin_fabsf(vs) < kHalfMax);
                    const _Float16 h0 = static_cast<_Float16>(vs);
                    hl[at[k]] = h0;
                    hl[at[k] + cph0] = static_cast<_Float16>(vs - static_cast<float>(h0));
                }
            }
        }
    }
    __syncthreads();
    MZ_TSTAMP(1);

    const int col_pair = wave & 1;                      // output channels col_pair * 32 .. + 31
    const int row_group = wave >> 1;
    const int i_row = lane & 15;
    const int kk = lane >> 4;
    int pos_a[MTW];
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        const int tile = row_group + t * RG;
        int m = tile * 16 + i_row;
        if (tile >= MT || m >= ROWS) m = 0;
        pos_a[t] = (m / P) * PP + plane_pos(m % P);
    }

    // The products are issued as W x X^T (weights as the MFMA's first operand): a lane's four D values are FOUR
    // CONSECUTIVE OUTPUT CHANNELS (4 kk + r of its column tile) of ONE position (i_row of its row tile), so the epilogue
    // reads the skip connection and writes the result with 8-byte LDS accesses.
    // plane position of the D column this lane stores, -1 = not a position of a present sample
    int pos_d[MTW];
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        const int tile = row_group + t * RG;
        const int m = tile * 16 + i_row;
        const bool ok = tile < MT && m < ROWS && m / P < n_samples;
        pos_d[t] = ok ? (m / P) * PP + plane_pos(m % P) : -1;
    }
    // the first two weight groups of a layer are fetched before the previous layer's epilogue (L2 / HBM latency)
    h8 bq[2][2];
    auto wload_of = [&](const SplitLayer& layer, int it, int c, int q) {
        const h8* w = reinterpret_cast<const h8*>(layer.wh) + kk * COUT + col_pair * 32 + i_row;
        return w[(static_cast<size_t>(it) * 2 + q) * 4 * COUT + 16 * c];
    };
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bq[c][q] = wload_of(args.layer[0], 0, c, q);
        }

    for (int l = 0; l < args.n_layers; ++l) {
        const SplitLayer& L = args.layer[l];
        const int in = (l & 1) ? buf1_at : 0;            // offsets of the layer's input / output buffers in hl
        const int dst = (l & 1) ? 0 : buf1_at;
        const int CPI = (l & 1) ? cph1 : cph0, CPO = (l & 1) ? cph0 : cph1;
        const int ng = split_groups(L.cin);
        const int iterations = 9 * ng;
        const int skip = L.skip, relu = L.relu;          // (read now: in registers long before the epilogue asks)
        const float* ctab = L.const_table;

        f32x4 hi[MTW][2], lo[MTW][2];
#pragma unroll
        for (int t = 0; t < MTW; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) hi[t][c] = lo[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

        // Two register sets (operands of an even / an odd iteration), each refilled for iteration + 2 right after its MFMAs
        // have issued: no register copies in the loop, LDS reads and weight loads one iteration ahead.
        const h8* wlane = reinterpret_cast<const h8*>(L.wh) + kk * COUT + col_pair * 32 + i_row;
        auto wload = [&](int it, int c, int q) { return wlane[(static_cast<size_t>(it) * 2 + q) * 4 * COUT + 16 * c]; };
        int grp = 0, tap = 0;                            // the (tap, group) of the NEXT A fetch
        auto lds_offset = [&]() { return ((tap / 3 - 1) * PW + (tap % 3 - 1)) * 2 * CPI + grp * kSplitGroup + 8 * kk; };
        auto advance = [&]() {
            if (++grp == ng) {
                grp = 0;
                tap = tap < 8 ? tap + 1 : 8;             // (fetches past the last iteration re-read the last tap: unused)
            }
        };
        // (Measured at 8192 Connect4 boards: with the weight fetches switched off -- MZ_TOWER_NO_WLOAD -- the launch is 10 %
        // shorter, 1295 -> 1167 us with heads; a THIRD weight set fetched three iterations ahead changes nothing, 1301 us:
        // the fetches cost L2 -> CU bandwidth, 1.9 MB of weights per workgroup of 4 boards, not exposed latency.)
        h8 a_even[MTW][2], a_odd[MTW][2], w_even[2][2], w_odd[2][2];
        auto fetch_a = [&](h8 (&a)[MTW][2]) {
            const int off = lds_offset();
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int q = 0; q < 2; ++q) a[t][q] = reinterpret_cast<const h8*>(hl)[(in + pos_a[t] * 2 * CPI + q * CPI + off) >> 3];
            advance();
        };
        auto fetch_b = [&](h8 (&b)[2][2], int it) {      // (two spare zero groups follow the last iteration's weights)
#ifndef MZ_TOWER_NO_WLOAD                                 // (diagnostic builds: what the weight fetches cost)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) b[c][q] = wload(it, c, q);
#endif
        };
        auto multiply = [&](const h8 (&a)[MTW][2], const h8 (&b)[2][2]) {
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) hi[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[c][0], a[t][0], hi[t][c], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) lo[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[c][1], a[t][0], lo[t][c], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) lo[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[c][0], a[t][1], lo[t][c], 0, 0, 0);
        };
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) w_even[c][q] = bq[c][q];
        fetch_b(w_odd, 1);
        float4 sc4[2], sh4[2];                           // folded batch norm of this lane's 2 x 4 channels (in flight early)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            sc4[c] = *reinterpret_cast<const float4*>(L.scale + col_pair * 32 + c * 16 + 4 * kk);
            sh4[c] = *reinterpret_cast<const float4*>(L.shift + col_pair * 32 + c * 16 + 4 * kk);
        }
        fetch_a(a_even);
        fetch_a(a_odd);
        for (int it = 0; it < iterations; it += 2) {     // (an odd count runs one more iteration on the spare zero weights)
            multiply(a_even, w_even);
            fetch_a(a_even);
            fetch_b(w_even, min(it + 2, iterations + 1));
            multiply(a_odd, w_odd);
            fetch_a(a_odd);
            fetch_b(w_odd, min(it + 3, iterations + 1));
        }
        MZ_TSTAMP(2);

        if (l + 1 < args.n_layers) {                     // next layer's first weights: in flight under this epilogue
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    bq[c][q] = wload_of(args.layer[l + 1], 0, c, q);
                }
        }
        // ---- layer epilogue: D[channel = 4 kk + r][position = lane & 15] of column tiles 2 col_pair, 2 col_pair + 1 ----
        // (four channels at a time as vectors: two-wide fp32 instructions where the chip has them; plain IEEE operations)
        constexpr float kDescale = 1.0f / (kActScale * kWtScale);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n0 = col_pair * 32 + c * 16 + 4 * kk;
            const f32x4 scl = {sc4[c].x, sc4[c].y, sc4[c].z, sc4[c].w};
            const f32x4 sft = {sh4[c].x, sh4[c].y, sh4[c].z, sh4[c].w};
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
                const int pos = pos_d[t];
                if (pos < 0) continue;
                // (indexed in units of four halves: CPO and n0 are multiples of 4, and the compiler may use 8-byte accesses)
                h4* cell0 = reinterpret_cast<h4*>(hl) + ((dst + pos * 2 * CPO + n0) >> 2);          // halves 0
                h4* cell1 = reinterpret_cast<h4*>(hl) + ((dst + pos * 2 * CPO + CPO + n0) >> 2);    // halves 1
                f32x4 conv = (hi[t][c] + lo[t][c]) * kDescale;
                if (ctab) {                              // (layer 0 only) position inside the board from the plane position
                    const int pp = pos % PP;
                    const float a = aconst[pos / PP];
                    const float* tab = ctab + n0 * P + (pp / PW - 1) * W + pp % PW - 1;
                    conv = conv + a * f32x4{tab[0], tab[P], tab[2 * P], tab[3 * P]};
                }
                f32x4 v = conv * scl + sft;
                if (skip) {
                    const h4 s0 = *cell0, s1 = *cell1;
                    v = v + (__builtin_convertvector(s0, f32x4) + __builtin_convertvector(s1, f32x4)) * (1.0f / kActScale);
                }
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
                }
                const f32x4 vs = v * kActScale;
                const h4 o0 = __builtin_convertvector(vs, h4);
                {   // inf or NaN among the four high halves: an fp16 exponent field of all ones (0x7c00 + 0x0400 carries into bit 15)
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 bits = __builtin_bit_cast(u32x2, o0);
                    overflow |= (((bits[0] & 0x7fff7fffu) + 0x04000400u) | ((bits[1] & 0x7fff7fffu) + 0x04000400u)) & 0x80008000u;
                }
                const h4 o1 = __builtin_convertvector(vs - __builtin_convertvector(o0, f32x4), h4);
                *cell0 = o0;
                *cell1 = o1;
            }
        }
        MZ_TSTAMP(3);
        __syncthreads();
        MZ_TSTAMP(4);

        if (L.export_raw || L.export_unit) {
            const int planes = n_samples * COUT;         // plane sn = sample * COUT + channel
            const size_t g0 = static_cast<size_t>(b0) * COUT * P;
            const int walker_p = tid % P;
            const int walker_at = plane_pos(walker_p);
            auto export_planes = [&](float* out) {
                if (tid < WALKERS)
                    for (int sn = tid / P; sn < planes; sn += TPP)
                        out[g0 + sn * P + walker_p] = load_val(dst, CPO, (sn / COUT) * PP + walker_at, sn % COUT);
            };
            if (L.export_raw) export_planes(L.export_raw);
            if (L.export_unit) {
                __syncthreads();
                // per plane: models.py:525-549.  Two threads per plane (first / second half of its positions, all loads
                // issued before the first compare); the halves' minima and maxima are combined in position order, so the
                // result is the sequential scan's, signs of zero included.
                constexpr int HALF = (P + 1) / 2;
                for (int q = tid >> 1; q < n_samples * COUT; q += THREADS / 2) {
                    const int n = q % COUT, sidx = q / COUT;
                    const int second = tid & 1;
                    float v[HALF];
                    int at[HALF];
#pragma unroll
                    for (int k = 0; k < HALF; ++k) {
                        const int p1 = HALF + k < P ? HALF + k : P - 1;          // (past the plane: masked below)
                        at[k] = sidx * PP + (second ? plane_pos(p1) : plane_pos(k));
                        v[k] = load_val(dst, CPO, at[k], n);
                    }
                    float lo_v = v[0], hi_v = v[0];
#pragma unroll
                    for (int k = 1; k < HALF; ++k) {
                        if (!second || HALF + k < P) {
                            lo_v = (v[k] < lo_v || v[k] != v[k]) ? v[k] : lo_v;
                            hi_v = (v[k] > hi_v || v[k] != v[k]) ? v[k] : hi_v;
                        }
                    }
                    const float lo_o = __shfl_xor(lo_v, 1), hi_o = __shfl_xor(hi_v, 1);
                    const float lo_a = second ? lo_o : lo_v, lo_b = second ? lo_v : lo_o;   // first half's, second half's
                    const float hi_a = second ? hi_o : hi_v, hi_b = second ? hi_v : hi_o;
                    lo_v = (lo_b < lo_a || lo_b != lo_b) ? lo_b : lo_a;
                    hi_v = (hi_b > hi_a || hi_b != hi_b) ? hi_b : hi_a;
                    float span = hi_v - lo_v;
                    if (span < 1e-5f) span = span + 1e-5f;
#pragma unroll
                    for (int k = 0; k < HALF; ++k)
                        if (!second || HALF + k < P) store_val(dst, CPO, at[k], n, (v[k] - lo_v) / span);
                }
                __syncthreads();
                export_planes(L.export_unit);
            }
            __syncthreads();
            MZ_TSTAMP(5);
        }
    }
    if (args.gate) {
        const int any = __syncthreads_or(overflow);
        if (tid == 0) {
            args.gate[blockIdx.x] = any ? 1 : 0;
            if (any) atomicAdd(&args.gate[gridDim.x], 1);
        }
    }
    MZ_TSTAMP_FLUSH;
}

template <int H, int W, int SB, int WAVES = kConvWaves>
static int launch_board_tower_split(const float* x, int batch, int cin0, int const_plane, const SplitArgs& args,
                                    hipStream_t stream, const TowerGather& gather = TowerGather{}) {
    constexpr int PP = (H + 2) * (W + 1) + 1;
    int cph0 = 64 + 8, cph1 = 64 + 8;                     // outputs are 64 channels in either buffer
    for (int l = 0; l < args.n_layers; ++l) {
        int& cp = (l & 1) ? cph1 : cph0;
        cp = std::max(cp, split_groups(args.layer[l].cin) * kSplitGroup + 8);
    }
    const size_t lds = sizeof(_Float16) * static_cast<size_t>(SB) * PP * 2 * (cph0 + cph1) + (sizeof(float) + sizeof(float*)) * SB;
    if (lds > 160 * 1024) return MZMCTS_ERR_INVALID;
    auto kernel = board_tower_split_kernel<H, W, SB, WAVES>;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(lds)) != hipSuccess)
        return MZMCTS_ERR_HIP;
    const dim3 grid(static_cast<unsigned>((batch + SB - 1) / SB)), block(64 * WAVES);
    const int cin_load = cin0 - const_plane;
    kernel<<<grid, block, lds, stream>>>(x, batch, cin0, const_plane, 0xFFFFFFFFu / static_cast<uint32_t>(cin_load) + 1u, cph0,
                                         cph1, args, gather);
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

template <int NT, int H, int W, int SB>
static int launch_board_conv(const float* x, const float* wt, const float* scale, const float* shift, const float* residual,
                             float* out, int batch, int cin, int relu, hipStream_t stream) {
    constexpr int PP = (H + 2) * (W + 1) + 1;
    const int cp = conv_groups(cin) * kConvGroup + 4;
    const size_t planes = static_cast<size_t>(SB) * PP * cp;
    const size_t stage = static_cast<size_t>(16 * NT) * (SB * H * W + 1) + 2 * 16 * NT;
    const size_t lds = sizeof(float) * std::max(planes, stage);
    if (lds > 160 * 1024) return MZMCTS_ERR_INVALID;
    const dim3 grid(static_cast<unsigned>((batch + SB - 1) / SB)), block(64 * kConvWaves);
#define MZ_CONV_LAUNCH(RES, ACT)                                                                                        \
    do {                                                                                                                \
        auto kernel = board_conv3x3_kernel<NT, H, W, SB, RES, ACT>;                                                     \
        if (lds > 64 * 1024 &&                                                                                          \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                static_cast<int>(lds)) != hipSuccess)                                                   \
            return MZMCTS_ERR_HIP;                                                                                      \
        kernel<<<grid, block, lds, stream>>>(x, wt, scale, shift, residual, out, batch, cin, 0xFFFFFFFFu / static_cast<uint32_t>(cin) + 1u);                           \
    } while (0)
    if (residual) {
        if (relu) MZ_CONV_LAUNCH(true, true); else MZ_CONV_LAUNCH(true, false);
    } else {
        if (relu) MZ_CONV_LAUNCH(false, true); else MZ_CONV_LAUNCH(false, false);
    }
#undef MZ_CONV_LAUNCH
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

}  // namespace mz

extern "C" int64_t mzmcts_board_conv_packed_floats(int32_t cin, int32_t cout) {
    if (cin <= 0 || cout <= 0) return -1;
    return mz::conv_packed_floats(cin, cout);
}

extern "C" int mzmcts_board_conv_pack(const float* weight, float* packed, int32_t cin, int32_t cout, void* stream_) {
    if (!weight || !packed || cin <= 0 || cout <= 0) return MZMCTS_ERR_INVALID;
    const int total = mz::conv_packed_floats(cin, cout);
    mz::board_conv_pack_kernel<<<dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_)>>>(weight, packed,
                                                                                                             cin, cout);
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

extern "C" int mzmcts_board_conv_supported(int32_t cin, int32_t cout, int32_t height, int32_t width) {
    const bool shape = (height == 6 && width == 7) || (height == 6 && width == 6) || (height == 3 && width == 3);
    return shape && (cout == 64 || cout == 16) && cin >= 1 && cin <= 80 ? 1 : 0;
}

extern "C" int mzmcts_board_conv3x3(const float* x, const float* packed, const float* scale, const float* shift,
                                    const float* residual, float* out, int64_t batch, int32_t cin, int32_t cout,
                                    int32_t height, int32_t width, int32_t relu, void* stream_) {
    if (!x || !packed || !scale || !shift || !out || batch < 0 || batch > 0x3fffffff ||
        !mzmcts_board_conv_supported(cin, cout, height, width) || out == x)
        return MZMCTS_ERR_INVALID;
    if (batch == 0) return MZMCTS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int b = static_cast<int>(batch);
    // samples per workgroup: 8 waves x <= 6 row tiles each, planes within LDS
    if (height == 6 && width == 7) {
        if (cout == 64) return mz::launch_board_conv<4, 6, 7, 4>(x, packed, scale, shift, residual, out, b, cin, relu, stream);
        return mz::launch_board_conv<1, 6, 7, 8>(x, packed, scale, shift, residual, out, b, cin, relu, stream);
    }
    if (height == 6 && width == 6) {
        if (cout == 64) return mz::launch_board_conv<4, 6, 6, 4>(x, packed, scale, shift, residual, out, b, cin, relu, stream);
        return mz::launch_board_conv<1, 6, 6, 16>(x, packed, scale, shift, residual, out, b, cin, relu, stream);
    }
    if (cout == 64) return mz::launch_board_conv<4, 3, 3, 16>(x, packed, scale, shift, residual, out, b, cin, relu, stream);
    return mz::launch_board_conv<1, 3, 3, 32>(x, packed, scale, shift, residual, out, b, cin, relu, stream);
}

// Boards per workgroup of the split-precision 64-channel tower.  6 x 7: TWO boards on FOUR wavefronts (84 rows = 6 tiles,
// three per wavefront, as with 4 boards on 8), because half the LDS lets two workgroups share a CU: one's fill / epilogue
// / barrier / export phases (55 % of a workgroup's life, profiles/r02_tower_phase_stamps.jsonl) run under the other's MFMAs.
// MZ_SPLIT_BOARDS=4|2|1 selects the shape (A/B measurements).
static int split_block_samples(int32_t height, int32_t width) {
    if (height == 6 && width == 7) {
        static const int chosen = [] {
            const char* env = std::getenv("MZ_SPLIT_BOARDS");
            const int v = env ? std::atoi(env) : 2;
            return (v == 4 || v == 2 || v == 1) ? v : 2;
        }();
        return chosen;
    }
    return height == 3 ? 16 : 4;
}

static int board_tower_impl(const float* x, const mz::TowerGather& gather, int64_t batch, int32_t cin0, int32_t channels,
                            int32_t height, int32_t width, const mzmcts_tower_layer* layers, int32_t n_layers, void* stream_,
                            const mzmcts_tower_head* heads = nullptr, int32_t n_heads = 0) {
    if ((!x && !gather.pool) || !layers || batch < 0 || batch > 0x3fffffff || n_layers < 1 || n_layers > mz::kMaxTowerLayers ||
        !mzmcts_board_conv_supported(cin0, channels, height, width) || n_heads < 0 || n_heads > 3 || (n_heads > 0 && !heads))
        return MZMCTS_ERR_INVALID;
    mz::TowerHeads th{};
    for (int q = 0; q < n_heads; ++q) {
        const mz::TowerHead hd{heads[q].head, heads[q].out, heads[q].layer, 0};
        if (hd.layer == n_layers - 1) {
            if (th.n_last >= 2) return MZMCTS_ERR_INVALID;
            (th.n_last == 0 ? th.last0 : th.last1) = hd;
            ++th.n_last;
        } else {
            if (th.has_mid) return MZMCTS_ERR_INVALID;
            th.mid = hd;
            th.has_mid = 1;
        }
    }
    mz::TowerArgs args{};
    args.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        const mzmcts_tower_layer& d = layers[l];
        if (!d.packed || !d.scale || !d.shift || d.cin != (l == 0 ? cin0 : channels)) return MZMCTS_ERR_INVALID;
        if ((reinterpret_cast<uintptr_t>(d.scale) | reinterpret_cast<uintptr_t>(d.shift)) & 15u)
            return MZMCTS_ERR_INVALID;                   // (read four channels at a time)
        args.layer[l] = mz::TowerLayer{static_cast<const float*>(d.packed), d.scale, d.shift, d.export_raw, d.export_unit, d.cin, d.relu, d.skip, 0};
    }
    args.gate = layers[0].gate;
    if (args.gate && channels != 64) return MZMCTS_ERR_INVALID;   // (the hand-over exists between the two 64-channel forms)
    args.gate_samples = split_block_samples(height, width);
    // heads inside the launch: the 3 x 3 board-column kernel only -- decided before anything is launched
    if (n_heads > 0 && !(height == 3 && width == 3 && mz::tower_cols_applies(channels, height, width, cin0, args)))
        return MZMCTS_ERR_INVALID;
    if (batch == 0) return MZMCTS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int b = static_cast<int>(batch);
    // Samples per workgroup.  A 16-channel tower has ONE column tile, so its 8 wavefronts split the SB x H x W output
    // rows into 16-row tiles and every wavefront runs ceil(tiles / 8) of them per k-step: 16 x 9 = 144 rows = 9 tiles
    // cost two rounds for little more than one round's work.  With few boards that choice stands (more workgroups than
    // CUs matters most); with many, SB is the count whose rows fit ONE round and whose LDS lets two or more workgroups
    // share a CU, so that one's fill / epilogue / export phases run under another's MFMAs: 14 x 9 = 126 rows = 8 tiles,
    // a tile for every wavefront (measured at 65536 TicTacToe boards: SB 16 / 28 / 12 / 8 = 318 / 313 / 300 / 353 us per
    // launch, and 12 -> 14: 374 -> 338 us with heads -- 12 x 9 = 108 rows are 7 tiles and leave the eighth wavefront
    // idle; 6x6, 16384 boards: SB 4 / 7 / 3 = 513 / 513 / 428 us).
    const bool many = b >= 16384;
    if (height == 6 && width == 7) {
        if (channels == 64) return mz::launch_board_tower<4, 6, 7, 4>(x, b, cin0, args, stream, gather);
        if (many && (cin0 * 42) % 2 == 0) return mz::launch_board_tower<1, 6, 7, 6>(x, b, cin0, args, stream, gather);
        return mz::launch_board_tower<1, 6, 7, 4>(x, b, cin0, args, stream, gather);
    }
    if (height == 6 && width == 6) {
        if (channels == 64) return mz::launch_board_tower<4, 6, 6, 4>(x, b, cin0, args, stream, gather);
        if (mz::tower_cols_applies(channels, height, width, cin0, args)) {
            const int rc = mz::launch_board_tower_patch66(x, b, cin0, args, stream, gather);
            if (rc != MZMCTS_ERR_INVALID) return rc;
        }
        if (many) return mz::launch_board_tower<1, 6, 6, 3>(x, b, cin0, args, stream, gather);
        return mz::launch_board_tower<1, 6, 6, 4>(x, b, cin0, args, stream, gather);
    }
    if (channels == 64) return mz::launch_board_tower<4, 3, 3, 16>(x, b, cin0, args, stream, gather);
    if (mz::tower_cols_applies(channels, height, width, cin0, args)) {
        const int rc = mz::launch_board_tower_cols(x, b, cin0, args, stream, gather, th);
        if (rc != MZMCTS_ERR_INVALID || n_heads > 0) return rc;
    }
    if (n_heads > 0) return MZMCTS_ERR_INVALID;                      // (heads inside the launch: the board-column kernel only)
    if (many) return mz::launch_board_tower<1, 3, 3, 14>(x, b, cin0, args, stream, gather);
    return mz::launch_board_tower<1, 3, 3, 16>(x, b, cin0, args, stream, gather);
}

// samples per workgroup of the tower launches of a given shape (the table board_tower_impl / board_tower_split_impl use;
// for 64 channels: of the SPLIT launch, whose workgroups are the units of the overflow hand-over)
static int tower_block_samples(int64_t batch, int32_t channels, int32_t height, int32_t width) {
    const bool many = batch >= 16384;
    if (height == 6 && width == 7) return channels == 64 ? split_block_samples(height, width) : 0;   // (16 channels: depends on the input's parity too)
    if (height == 6 && width == 6) return channels == 64 ? 4 : (many ? 3 : 4);
    return channels == 64 ? 16 : (many ? 14 : 16);
}

extern "C" int64_t mzmcts_board_tower_blocks(int64_t batch, int32_t channels, int32_t height, int32_t width) {
    const int sb = tower_block_samples(batch, channels, height, width);
    if (batch < 0 || sb <= 0 || !mzmcts_board_conv_supported(channels, channels, height, width)) return -1;
    return (batch + sb - 1) / sb;
}

extern "C" int mzmcts_board_tower(const float* x, int64_t batch, int32_t cin0, int32_t channels, int32_t height, int32_t width,
                                  const mzmcts_tower_layer* layers, int32_t n_layers, void* stream) {
    if (!x) return MZMCTS_ERR_INVALID;
    return board_tower_impl(x, mz::TowerGather{}, batch, cin0, channels, height, width, layers, n_layers, stream);
}

extern "C" int mzmcts_board_tower_heads(const float* x, const mzmcts_tower_gather* g, int64_t batch, int32_t cin0,
                                        int32_t channels, int32_t height, int32_t width, const mzmcts_tower_layer* layers,
                                        int32_t n_layers, const mzmcts_tower_head* heads, int32_t n_heads, void* stream) {
    if ((!x) == (!g) || n_heads < 1) return MZMCTS_ERR_INVALID;     // exactly one input form
    mz::TowerGather gather{};
    if (g) {
        if (!g->pool || !g->parent || !g->action || g->envs < batch || !(g->action_space > 0.f) ||
            g->hidden_floats != channels * height * width || cin0 != channels + 1)
            return MZMCTS_ERR_INVALID;
        gather = mz::TowerGather{g->pool, g->parent, g->action, static_cast<long long>(g->envs), g->hidden_floats, g->action_space};
    }
    return board_tower_impl(x, gather, batch, cin0, channels, height, width, layers, n_layers, stream, heads, n_heads);
}

extern "C" int64_t mzmcts_board_conv_split_halfs(int32_t cin_conv, int32_t cout) {
    if (cin_conv <= 0 || cout <= 0) return -1;
    return mz::split_packed_halfs(cin_conv, cout);
}

extern "C" int mzmcts_board_conv_pack_split(const float* weight, void* packed, float* const_table, int32_t cin, int32_t cout,
                                            int32_t const_plane, int32_t height, int32_t width, void* stream_) {
    if (!weight || !packed || cin <= 0 || cout <= 0 || (const_plane && (!const_table || cin < 2 || height <= 0 || width <= 0)))
        return MZMCTS_ERR_INVALID;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int cin_conv = cin - (const_plane ? 1 : 0);
    const int64_t total = mz::split_packed_halfs(cin_conv, cout);
    mz::board_conv_pack_split_kernel<<<dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, stream>>>(
        weight, static_cast<_Float16*>(packed), cin, cin_conv, cout);
    if (const_plane)
        mz::board_conv_const_plane_kernel<<<dim3((cout * height * width + 255) / 256), dim3(256), 0, stream>>>(
            weight, const_table, cin, cin - 1, cout, height, width);
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

static int board_tower_split_impl(const float* x, const mz::TowerGather& gather, int64_t batch, int32_t cin0,
                                  int32_t const_plane, int32_t channels, int32_t height, int32_t width,
                                  const mzmcts_tower_layer* layers, int32_t n_layers, void* stream_) {
    if ((!x && !gather.pool) || !layers || batch < 0 || batch > 0x3fffffff || n_layers < 1 || n_layers > mz::kMaxTowerLayers || channels != 64 ||
        !mzmcts_board_conv_supported(cin0, channels, height, width) || (const_plane && cin0 < 2))
        return MZMCTS_ERR_INVALID;
    mz::SplitArgs args{};
    args.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        const mzmcts_tower_layer& d = layers[l];
        const int cin_conv = l == 0 ? cin0 - (const_plane ? 1 : 0) : channels;
        if (!d.packed || !d.scale || !d.shift || d.cin != (l == 0 ? cin0 : channels) ||
            (l == 0 && const_plane && !d.const_table))
            return MZMCTS_ERR_INVALID;
        if ((reinterpret_cast<uintptr_t>(d.scale) | reinterpret_cast<uintptr_t>(d.shift)) & 15u)
            return MZMCTS_ERR_INVALID;                   // (read four channels at a time)
        args.layer[l] = mz::SplitLayer{static_cast<const _Float16*>(d.packed), d.scale, d.shift,
                                       (l == 0 && const_plane) ? d.const_table : nullptr, d.export_raw, d.export_unit,
                                       cin_conv, d.relu, d.skip, 0};
    }
    args.gate = layers[0].gate;
    {
        static const int slot_priority = [] {
            const char* env = std::getenv("MZ_SPLIT_PRIORITY");
            return (env && std::string(env) == "off") ? 0 : 1;
        }();
        args.slot_priority = slot_priority;
    }
    if (batch == 0) return MZMCTS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int b = static_cast<int>(batch);
    // (4 boards per workgroup: 3 fill the MFMA rounds better -- 126 rows = 8 tiles -- and 2 let two workgroups share a
    // CU, but both measured slower at 4096 Connect4 boards: 700 / 744 / 900 us per launch for 4 / 3 / 2; again after the
    // packed epilogue, 8192 boards with heads: 1289 us for 4, 1366 us for 3)
    if (height == 6 && width == 7) {
        const int sb = split_block_samples(height, width);
        if (sb == 2) return mz::launch_board_tower_split<6, 7, 2, 4>(x, b, cin0, const_plane, args, stream, gather);
        if (sb == 1) return mz::launch_board_tower_split<6, 7, 1, 2>(x, b, cin0, const_plane, args, stream, gather);
        return mz::launch_board_tower_split<6, 7, 4>(x, b, cin0, const_plane, args, stream, gather);
    }
    if (height == 6 && width == 6) return mz::launch_board_tower_split<6, 6, 4>(x, b, cin0, const_plane, args, stream, gather);
    return mz::launch_board_tower_split<3, 3, 16>(x, b, cin0, const_plane, args, stream, gather);
}

extern "C" int mzmcts_board_tower_split(const float* x, int64_t batch, int32_t cin0, int32_t const_plane, int32_t channels,
                                        int32_t height, int32_t width, const mzmcts_tower_layer* layers, int32_t n_layers,
                                        void* stream) {
    if (!x) return MZMCTS_ERR_INVALID;
    return board_tower_split_impl(x, mz::TowerGather{}, batch, cin0, const_plane, channels, height, width, layers, n_layers,
                                  stream);
}

extern "C" int mzmcts_board_tower_gathered(const mzmcts_tower_gather* g, int64_t batch, int32_t cin0, int32_t split,
                                           int32_t channels, int32_t height, int32_t width, const mzmcts_tower_layer* layers,
                                           int32_t n_layers, void* stream) {
    if (!g || !g->pool || !g->parent || !g->action || g->envs < batch || !(g->action_space > 0.f) ||
        g->hidden_floats != channels * height * width || cin0 != channels + 1)
        return MZMCTS_ERR_INVALID;
    const mz::TowerGather gather{g->pool, g->parent, g->action, static_cast<long long>(g->envs), g->hidden_floats,
                                 g->action_space};
    if (split)
        return board_tower_split_impl(nullptr, gather, batch, cin0, 1, channels, height, width, layers, n_layers, stream);
    return board_tower_impl(nullptr, gather, batch, cin0, channels, height, width, layers, n_layers, stream);
}

#ifdef MZ_TOWER_STAMPS
extern "C" int mzmcts_tower_stamps(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mz::g_tower_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -2;
    if (reset) {
        unsigned long long zeros[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mz::g_tower_stamps), zeros, sizeof(zeros)) != hipSuccess) return -2;
    }
    return 0;
}
#endif
