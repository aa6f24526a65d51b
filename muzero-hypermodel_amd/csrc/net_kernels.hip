// net_kernels.hip -- element-wise epilogues of the residual networks' convolutions in inference mode.
//
// The convolutions themselves stay with PyTorch-ROCm / MIOpen (fp32 Winograd on these boards); what follows
// each of them in the reference -- BatchNorm2d in eval(), the residual add, ReLU (models.py:215-237, 249-275,
// 318-335) -- is three element-wise launches per convolution through torch and one through this kernel:
//     out = act(x * scale[c] + shift[c] (+ residual))        NCHW, c = (index / plane) % channels
// Same operations in the same order as the torch expression it replaces (mul, add, add, max; the library is
// built with -ffp-contract=off), so the results are bit-identical to it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "../../include/mzmcts.h"

namespace mz {

template <bool RESIDUAL, bool RELU>
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ residual, float* __restrict__ out,
                                                         uint32_t count, uint32_t channels, uint32_t plane) {
    const uint32_t first = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (first >= count) return;
    uint32_t q = first / plane;        // (sample, channel) row of the first element
    uint32_t r = first - q * plane;    // position inside the plane
    uint32_t c = q % channels;
    float v[4], res[4] = {0.f, 0.f, 0.f, 0.f};
    const bool whole = first + 4u <= count;
    if (whole) {
        const float4 xv = *reinterpret_cast<const float4*>(x + first);
        v[0] = xv.x, v[1] = xv.y, v[2] = xv.z, v[3] = xv.w;
        if (RESIDUAL) {
            const float4 rv = *reinterpret_cast<const float4*>(residual + first);
            res[0] = rv.x, res[1] = rv.y, res[2] = rv.z, res[3] = rv.w;
        }
    } else {
        for (uint32_t i = 0; i < 4u; ++i) {
            v[i] = first + i < count ? x[first + i] : 0.f;
            if (RESIDUAL) res[i] = first + i < count ? residual[first + i] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float t = v[i] * scale[c] + shift[c];
        if (RESIDUAL) t = t + res[i];
        if (RELU) t = t < 0.f ? 0.f : t;  // (a NaN stays a NaN, as torch.relu keeps it)
        v[i] = t;
        if (++r == plane) {
            r = 0;
            if (++c == channels) c = 0;
        }
    }
    if (whole) {
        *reinterpret_cast<float4*>(out + first) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (uint32_t i = 0; i < 4u && first + i < count; ++i) out[first + i] = v[i];
    }
}

// Min-max rescale of every row to [0, 1] (reference models.py:525-549: per (sample, channel) over the board):
//     out = (x - min(row)) / span,   span = max(row) - min(row), + 1e-5 if it is below 1e-5
// torch spells it amin, amax, sub, lt, add, where, div (seven launches).  A workgroup moves a tile of 256 rows
// through LDS with coalesced accesses on both sides; one thread owns a row in between.  Same fp32 operations as the
// torch expression (IEEE subtraction and division), so the result is bit-identical to it; NaNs propagate as
// torch.amin / amax propagate them.  x may be the same tensor as out (the tile is read whole before it is written).
constexpr int kRescaleRows = 256;

__global__ __launch_bounds__(kRescaleRows) void unit_rescale_kernel(const float* x, float* out, uint32_t rows,
                                                                    uint32_t row_len) {
    extern __shared__ float tile[];  // [kRescaleRows][row_len]
    const uint32_t first_row = blockIdx.x * kRescaleRows;
    const uint32_t n_rows = min(static_cast<uint32_t>(kRescaleRows), rows - first_row);
    const size_t base = static_cast<size_t>(first_row) * row_len;
    const uint32_t n = n_rows * row_len;
    for (uint32_t i = threadIdx.x; i < n; i += kRescaleRows) tile[i] = x[base + i];
    __syncthreads();
    if (threadIdx.x < n_rows) {
        float* row = tile + threadIdx.x * row_len;
        float lo = row[0], hi = row[0];
        for (uint32_t p = 1; p < row_len; ++p) {
            const float v = row[p];
            lo = (v < lo || v != v) ? v : lo;
            hi = (v > hi || v != v) ? v : hi;
        }
        float span = hi - lo;
        if (span < 1e-5f) span = span + 1e-5f;
        for (uint32_t p = 0; p < row_len; ++p) row[p] = (row[p] - lo) / span;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kRescaleRows) out[base + i] = tile[i];
}

// One prediction / reward head of the residual networks (reference models.py:467-480 reward, 500-522 value and
// policy): 1x1 convolution (+ bias) over the board, flatten, Linear, ELU, Linear -- six launches through torch (two
// of them layout copies around the 1x1 GEMM), one here.  A wave owns a sample: its board goes to LDS once, the three
// layers run out of LDS, only the logits are written.  The weights are staged per workgroup as plain copies (many
// loads in flight per thread: the staging is latency, not bandwidth); a lane that owns output row j of a Linear
// layer starts its dot product at column j and wraps around, so that the 64 rows' reads fall into 64 different
// banks whatever the row length.  A hidden layer narrower than the wave splits its dot products over the idle
// lanes.  fp32 throughout; hipBLASLt's
// summation order is its own, so the two agree to fp32 rounding, not bit for bit (tests/test_gpu_net.py: 1e-5).
constexpr int kHeadWaves = 4;

struct HeadShape {
    int C, P, R, Hd, O;  // channels, board positions, reduced channels, hidden units, outputs
    int split;           // lanes sharing one hidden unit's dot product: 64 / pow2(Hd), at least 1
    __host__ __device__ int RP() const { return R * P; }
    __host__ __device__ int conv_w() const { return 0; }
    __host__ __device__ int conv_b() const { return conv_w() + R * C; }
    __host__ __device__ int fc1_w() const { return conv_b() + R; }
    __host__ __device__ int fc1_b() const { return fc1_w() + Hd * RP(); }
    __host__ __device__ int fc2_w() const { return fc1_b() + Hd; }
    __host__ __device__ int fc2_b() const { return fc2_w() + O * Hd; }
    __host__ __device__ int per_wave() const { return (fc2_b() + O + 3) & ~3; }  // 16-byte aligned boards
    // per wave: [x C*P | y R*P | partial sums split*Hd | h Hd], padded to a multiple of 4 words
    __host__ __device__ int wave_floats() const { return (C * P + RP() + split * Hd + Hd + 3) & ~3; }
    __host__ __device__ int total() const { return per_wave() + kHeadWaves * wave_floats(); }
};

// Orders this wave's LDS writes before its following LDS reads (other lanes' data), for the compiler and the
// memory counters; no other wave is involved.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// dst[0..n) = src[0..n) by the whole workgroup, eight independent loads per thread in flight
__device__ __forceinline__ void stage_copy(float* dst, const float* __restrict__ src, int n, int tid) {
    constexpr int kInFlight = 8, kThreads = 64 * kHeadWaves;
    for (int base = tid; base < n; base += kThreads * kInFlight) {
        float v[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const int i = base + u * kThreads;
            v[u] = i < n ? src[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const int i = base + u * kThreads;
            if (i < n) dst[i] = v[u];
        }
    }
}

// Up to two heads that read the same board (value and policy) share a launch: blockIdx.y picks the head.
constexpr int kMaxHeads = 3;
struct HeadSet {
    const float* x[kMaxHeads];   // the tensor each head reads (value / policy share one, the reward head has its own)
    mzmcts_head_desc desc[kMaxHeads];
    HeadShape shape[kMaxHeads];
    float* out[kMaxHeads];
};

__global__ __launch_bounds__(64 * kHeadWaves) void conv_head_kernel(HeadSet set, int batch) {
    const float* __restrict__ x = set.x[blockIdx.y];
    const mzmcts_head_desc d = set.desc[blockIdx.y];
    const HeadShape s = set.shape[blockIdx.y];
    float* __restrict__ out = set.out[blockIdx.y];
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int RP = s.RP();
    stage_copy(lds + s.conv_w(), d.conv_w, s.R * s.C, tid);
    stage_copy(lds + s.conv_b(), d.conv_b, s.R, tid);
    stage_copy(lds + s.fc1_w(), d.fc1_w, s.Hd * RP, tid);
    stage_copy(lds + s.fc1_b(), d.fc1_b, s.Hd, tid);
    stage_copy(lds + s.fc2_w(), d.fc2_w, s.O * s.Hd, tid);
    stage_copy(lds + s.fc2_b(), d.fc2_b, s.O, tid);
    const int CP = s.C * s.P;
    float* xs = lds + s.per_wave() + wave * s.wave_floats();
    float* ys = xs + CP;
    float* part = ys + RP;
    float* hs = part + s.split * s.Hd;
    // The sample buffers belong to the wave: between its steps it needs its own LDS writes to have landed (LDS serves
    // a wave's accesses in order), not a workgroup barrier.  The one barrier is for the staged weights.
    auto load_board = [&](int b) {
        const float* src = x + static_cast<size_t>(b) * CP;
        if ((CP & 3) == 0) {
#pragma unroll 4
            for (int i = lane; i < CP / 4; i += 64) reinterpret_cast<float4*>(xs)[i] = reinterpret_cast<const float4*>(src)[i];
        } else {
            for (int i = lane; i < CP; i += 64) xs[i] = src[i];
        }
    };
    int b = blockIdx.x * kHeadWaves + wave;
    if (b < batch) load_board(b);  // in flight together with the weights
    __syncthreads();               // the staged weights, for every wave (also the ones without a sample)
    for (bool first = true; b < batch; b += gridDim.x * kHeadWaves, first = false) {
        if (!first) load_board(b);
        wave_sync();
        {
            for (int idx = lane; idx < RP; idx += 64) {  // 1x1 convolution: y[r][p] = sum_c w[r][c] x[c][p] + b[r]
                const int r = idx / s.P, p = idx - r * s.P;
                const float* w = lds + s.conv_w() + r * s.C;
                float acc = 0.f;
#pragma unroll 8
                for (int c = 0; c < s.C; ++c) acc += w[c] * xs[c * s.P + p];
                ys[idx] = acc + lds[s.conv_b() + r];
            }
        }
        wave_sync();
        {  // Linear: hidden unit j; its columns g, g + split, ... per lane, or all of them from column j on
            const int g = lane / s.Hd, j = lane - g * s.Hd;
            if (g < s.split) {
                const float* w = lds + s.fc1_w() + j * RP;
                float acc = 0.f;
                if (s.split > 1) {
#pragma unroll 8
                    for (int k = g; k < RP; k += s.split) acc += w[k] * ys[k];
                } else {
                    int k = j % RP;
#pragma unroll 8
                    for (int t = 0; t < RP; ++t) {
                        acc += w[k] * ys[k];
                        k = (k + 1 == RP) ? 0 : k + 1;
                    }
                }
                part[g * s.Hd + j] = acc;
            }
            for (int jj = lane + 64; jj < s.Hd; jj += 64) {  // hidden layers wider than the wave (split == 1)
                const float* w = lds + s.fc1_w() + jj * RP;
                float acc = 0.f;
                int k = jj % RP;
                for (int t = 0; t < RP; ++t) {
                    acc += w[k] * ys[k];
                    k = (k + 1 == RP) ? 0 : k + 1;
                }
                part[jj] = acc;
            }
        }
        wave_sync();
        {
            for (int j = lane; j < s.Hd; j += 64) {  // partial sums in order, bias, ELU
                float acc = part[j];
                for (int g = 1; g < s.split; ++g) acc += part[g * s.Hd + j];
                acc += lds[s.fc1_b() + j];
                hs[j] = acc > 0.f ? acc : expf(acc) - 1.f;
            }
        }
        wave_sync();
        {
            for (int o = lane; o < s.O; o += 64) {  // Linear
                const float* w = lds + s.fc2_w() + o * s.Hd;
                float acc = 0.f;
                int j = o % s.Hd;
#pragma unroll 8
                for (int t = 0; t < s.Hd; ++t) {
                    acc += w[j] * hs[j];
                    j = (j + 1 == s.Hd) ? 0 : j + 1;
                }
                out[static_cast<size_t>(b) * s.O + o] = acc + lds[s.fc2_b() + o];
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------
// The same heads on the matrix cores, 16 samples per wavefront and step (exact fp32, v_mfma_f32_16x16x4_f32).  The
// wave-per-sample kernel above is bound by instruction issue -- a multiply-add costs two LDS reads and an index update
// -- which at 65536 TicTacToe boards made the heads a third of a simulation step.  Here the three layers are three small
// GEMMs per 16-sample tile:
//     1x1 convolution  rows = (sample, position) pairs (16 P rows = P tiles), K = channels, N = reduced channels (<= 16);
//                      the A operand comes straight from the NCHW board (16 lanes read 16 consecutive positions of a
//                      channel), the weights wait in registers; y goes to LDS as [sample][r P + p], the Linear's input
//     Linear + ELU     rows = samples, K = R P, N = hidden units (<= 64), weights in LDS; h to LDS as [sample][unit]
//     Linear           rows = samples, K = hidden, N = outputs (<= 32); logits to global
// Every output is a k-ordered chain of fused multiply-adds in the matrix pipe (the order differs from the kernel above
// and from hipBLASLt: equal to fp32 rounding, tests/test_gpu_net.py).
// -------------------------------------------------------------------------------------------------------------------
typedef float head_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTileSamples = 16;
constexpr int kMaxConvSteps = 16;   // channels / 4 <= 16: the 1x1 convolution's weights stay in registers (4 or 16 k-steps)

struct MfmaHeadShape {
    int C, P, R, Hd, O;
    __host__ __device__ int RP() const { return R * P; }
    __host__ __device__ int ys_stride() const { return RP() + 1; }      // (+1: the 16 sample rows fall into different banks)
    __host__ __device__ int hs_stride() const { return Hd + 1; }
    __host__ __device__ int w1_stride() const { return RP() + 1; }
    __host__ __device__ int w2_stride() const { return Hd + 1; }
    __host__ __device__ int nt1() const { return (Hd + 15) / 16; }
    __host__ __device__ int nt2() const { return (O + 15) / 16; }
    // LDS (floats): W1 [16 nt1][w1_stride] | b1 [16 nt1] | W2 [16 nt2][w2_stride] | b2 [16 nt2] | per wave { ys, hs }
    __host__ __device__ int off_b1() const { return 16 * nt1() * w1_stride(); }
    __host__ __device__ int off_w2() const { return off_b1() + 16 * nt1(); }
    __host__ __device__ int off_b2() const { return off_w2() + 16 * nt2() * w2_stride(); }
    __host__ __device__ int off_waves() const { return off_b2() + 16 * nt2(); }
    __host__ __device__ int wave_floats() const { return kTileSamples * (ys_stride() + hs_stride()); }
    __host__ __device__ int total() const { return off_waves() + kHeadWaves * wave_floats(); }
};

struct MfmaHeadSet {
    const float* x[kMaxHeads];
    mzmcts_head_desc desc[kMaxHeads];
    MfmaHeadShape shape[kMaxHeads];
    float* out[kMaxHeads];
};

// KS = k-steps of the 1x1 convolution (channels / 4, rounded up to 4 or 16), G = row tiles whose board values a lane
// keeps in registers at once: the values of the NEXT chunk of G tiles -- of this 16-sample tile or of the wave's next
// one -- are requested before the current chunk goes through the matrix pipe, so the L2 round trip of a chunk hides
// under the chunk before it.
template <int NT1, int NT2, int KS, int G>
__device__ __forceinline__ void mfma_head_tiles(const MfmaHeadShape& s, const mzmcts_head_desc& d, const float* __restrict__ x,
                                                float* __restrict__ out, float* lds, int batch, int lane, int wave) {
    const int i = lane & 15, kk = lane >> 4;
    const int RP = s.RP(), CP = s.C * s.P;
    const float* w1 = lds;
    const float* b1 = lds + s.off_b1();
    const float* w2 = lds + s.off_w2();
    const float* b2 = lds + s.off_b2();
    float* ys = lds + s.off_waves() + wave * s.wave_floats();
    float* hs = ys + kTileSamples * s.ys_stride();
    // 1x1 convolution weights of this lane: B[k = 4 ks + kk][n = i] = conv_w[i][4 ks + kk]
    float wc[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = 4 * ks + kk;
        wc[ks] = (i < s.R && c < s.C) ? d.conv_w[i * s.C + c] : 0.f;
    }
    const float conv_bias = i < s.R ? d.conv_b[i] : 0.f;

    const int tiles = (batch + kTileSamples - 1) / kTileSamples;
    const int stride = gridDim.x * kHeadWaves;
    // m / P for the row indices m <= 16 P + 15 of a tile, without an integer division per use (exact while m P < 2^20)
    const uint32_t p_magic = ((1u << 20) + static_cast<uint32_t>(s.P) - 1u) / static_cast<uint32_t>(s.P);
    auto div_p = [&](int m) { return static_cast<int>((static_cast<uint32_t>(m) * p_magic) >> 20); };
    // rows of row tile t: m = 16 t + (0..15), m = sample * P + position
    auto load_chunk = [&](int tile, int c0, float (&dst)[G][KS]) {
        const int b0 = tile * kTileSamples;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int t = c0 + g;
            const int m = 16 * t + i;                       // this lane's A row
            const int sa = div_p(m), pa = m - sa * s.P;
            const bool live = tile < tiles && t < s.P && b0 + sa < batch;
            const float* row = x + static_cast<size_t>(b0 + sa) * CP + pa;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c = 4 * ks + kk;
                dst[g][ks] = (live && c < s.C) ? row[c * s.P] : 0.f;
            }
        }
    };
    int tile = blockIdx.x * kHeadWaves + wave;
    int c0 = 0;
    float cur[G][KS], nxt[G][KS];
    load_chunk(tile, 0, cur);
    while (tile < tiles) {
        const int b0 = tile * kTileSamples;
        const bool last_chunk = c0 + G >= s.P;
        const int ntile = last_chunk ? tile + stride : tile;
        const int nc0 = last_chunk ? 0 : c0 + G;
        load_chunk(ntile, nc0, nxt);
        // ---- 1x1 convolution of this chunk's row tiles (k-step outermost: consecutive MFMAs belong to different tiles,
        //      a dependent accumulate would wait out the matrix pipe's latency) ----------------------------------------
        {
            head_f32x4 acc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = head_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[g][ks], wc[ks], acc[g], 0, 0, 0);
            if (i < s.R) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int t = c0 + g;
                    if (t < s.P) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {       // D[row = 4 kk + r][col = i]
                            const int md = 16 * t + 4 * kk + r;
                            const int sd = div_p(md), pd = md - sd * s.P;
                            ys[sd * s.ys_stride() + i * s.P + pd] = acc[g][r] + conv_bias;
                        }
                    }
                }
            }
        }
        if (last_chunk) {
            wave_sync();
            // ---- Linear + ELU: h[sample][unit] = elu(sum_k y[sample][k] W1[unit][k] + b1[unit]) ----------------------
            {
                // (kParts accumulators per column tile take the k-steps in turn and are added at the end, so that
                // consecutive MFMAs are independent also when there is a single column tile)
                constexpr int kParts = NT1 >= 4 ? 1 : 4;
                head_f32x4 part[NT1][kParts];
#pragma unroll
                for (int n = 0; n < NT1; ++n)
#pragma unroll
                    for (int q = 0; q < kParts; ++q) part[n][q] = head_f32x4{0.f, 0.f, 0.f, 0.f};
                const float* yrow = ys + i * s.ys_stride();
                const int steps = (RP + 3) / 4;
                for (int ks0 = 0; ks0 < steps; ks0 += kParts) {
#pragma unroll
                    for (int q = 0; q < kParts; ++q) {
                        const int k = 4 * (ks0 + q) + kk;
                        const bool in = ks0 + q < steps && k < RP;
                        const float a = in ? yrow[k] : 0.f;
#pragma unroll
                        for (int n = 0; n < NT1; ++n) {
                            const float b = in ? w1[(16 * n + i) * s.w1_stride() + k] : 0.f;
                            part[n][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, part[n][q], 0, 0, 0);
                        }
                    }
                }
                head_f32x4 acc[NT1];
#pragma unroll
                for (int n = 0; n < NT1; ++n) {
                    acc[n] = part[n][0];
#pragma unroll
                    for (int q = 1; q < kParts; ++q) acc[n] = acc[n] + part[n][q];
                }
#pragma unroll
                for (int n = 0; n < NT1; ++n) {
                    const int unit = 16 * n + i;
                    if (unit < s.Hd) {
                        const float bias = b1[unit];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[n][r] + bias;
                            hs[(4 * kk + r) * s.hs_stride() + unit] = v > 0.f ? v : expf(v) - 1.f;
                        }
                    }
                }
            }
            wave_sync();
            // ---- Linear: logits[sample][o] --------------------------------------------------------------------------------
            {
                head_f32x4 acc[NT2];
#pragma unroll
                for (int n = 0; n < NT2; ++n) acc[n] = head_f32x4{0.f, 0.f, 0.f, 0.f};
                const float* hrow = hs + i * s.hs_stride();
                const int steps = (s.Hd + 3) / 4;
                for (int ks = 0; ks < steps; ++ks) {
                    const int k = 4 * ks + kk;
                    const float a = k < s.Hd ? hrow[k] : 0.f;
#pragma unroll
                    for (int n = 0; n < NT2; ++n) {
                        const float b = k < s.Hd ? w2[(16 * n + i) * s.w2_stride() + k] : 0.f;
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int n = 0; n < NT2; ++n) {
                    const int o = 16 * n + i;
                    if (o < s.O) {
                        const float bias = b2[o];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int sample = b0 + 4 * kk + r;
                            if (sample < batch) out[static_cast<size_t>(sample) * s.O + o] = acc[n][r] + bias;
                        }
                    }
                }
            }
            wave_sync();   // (the next tile rewrites ys / hs)
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) cur[g][ks] = nxt[g][ks];
        tile = ntile;
        c0 = nc0;
    }
}

template <int KS, int G>
__device__ __forceinline__ void mfma_head_dispatch(const MfmaHeadShape& s, const mzmcts_head_desc& d, const float* __restrict__ x,
                                                   float* __restrict__ out, float* lds, int batch, int lane, int wave) {
    const int nt1 = s.nt1(), nt2 = s.nt2();
    if (nt1 == 1 && nt2 == 1) mfma_head_tiles<1, 1, KS, G>(s, d, x, out, lds, batch, lane, wave);
    else if (nt1 == 1 && nt2 == 2) mfma_head_tiles<1, 2, KS, G>(s, d, x, out, lds, batch, lane, wave);
    else if (nt2 == 1) mfma_head_tiles<4, 1, KS, G>(s, d, x, out, lds, batch, lane, wave);
    else mfma_head_tiles<4, 2, KS, G>(s, d, x, out, lds, batch, lane, wave);
}

__global__ __launch_bounds__(64 * kHeadWaves) void conv_head_mfma_kernel(MfmaHeadSet set, int batch) {
    const float* __restrict__ x = set.x[blockIdx.y];
    const mzmcts_head_desc d = set.desc[blockIdx.y];
    const MfmaHeadShape s = set.shape[blockIdx.y];
    float* __restrict__ out = set.out[blockIdx.y];
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kThreads = 64 * kHeadWaves;
    // Linear weights, rows padded to whole 16-row tiles with zeros (so the B operand needs no row guard)
    const int RP = s.RP();
    // zeros first (padding rows / columns), then the weights with eight independent loads per thread in flight: the
    // staging is latency, not bandwidth
    for (int idx = tid; idx < s.off_waves(); idx += kThreads) lds[idx] = 0.f;
    __syncthreads();
    auto stage_rows = [&](float* dst, const float* __restrict__ src, int rows, int cols, int dst_stride) {
        constexpr int kInFlight = 8;
        const int n = rows * cols;
        const uint32_t magic = ((1u << 20) + static_cast<uint32_t>(cols) - 1u) / static_cast<uint32_t>(cols);
        const bool fast = static_cast<int64_t>(n) * cols < (1 << 20);       // (exactness bound of the magic division)
        for (int base = tid; base < n; base += kThreads * kInFlight) {
            float v[kInFlight];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {
                const int idx = base + u * kThreads;
                v[u] = idx < n ? src[idx] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {
                const int idx = base + u * kThreads;
                if (idx < n) {
                    const int row = fast ? static_cast<int>((static_cast<uint32_t>(idx) * magic) >> 20) : idx / cols;
                    dst[row * dst_stride + (idx - row * cols)] = v[u];
                }
            }
        }
    };
    stage_rows(lds, d.fc1_w, s.Hd, RP, s.w1_stride());
    stage_rows(lds + s.off_b1(), d.fc1_b, 1, s.Hd, s.Hd);
    stage_rows(lds + s.off_w2(), d.fc2_w, s.O, s.Hd, s.w2_stride());
    stage_rows(lds + s.off_b2(), d.fc2_b, 1, s.O, s.O);
    __syncthreads();
    if (s.C <= 16) mfma_head_dispatch<4, 6>(s, d, x, out, lds, batch, lane, wave);
    else mfma_head_dispatch<16, 2>(s, d, x, out, lds, batch, lane, wave);
}

// shapes the matrix-core heads take: reduced channels <= 16 (one column tile), channels <= 64, hidden <= 64, outputs <= 32
static bool mfma_head_ok(const mzmcts_head_desc& d) {
    return d.reduced <= 16 && d.channels <= 4 * kMaxConvSteps && d.hidden <= 64 && d.outputs <= 32;
}

// The dynamics network's input (reference models.py:553-568): the hidden state's planes followed by one plane
// holding action / action_space_size -- through torch a cast, a division, an expand and a concatenation.
__global__ __launch_bounds__(256) void state_action_planes_kernel(const float* __restrict__ state,
                                                                  const int64_t* __restrict__ action,
                                                                  float* __restrict__ out, uint32_t state_floats,
                                                                  uint32_t plane, float action_space) {
    const uint32_t b = blockIdx.y;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= state_floats + plane) return;
    const size_t o = static_cast<size_t>(b) * (state_floats + plane) + i;
    if (i < state_floats)
        out[o] = state[static_cast<size_t>(b) * state_floats + i];
    else
        out[o] = static_cast<float>(action[b]) / action_space;
}

}  // namespace mz

extern "C" int mzmcts_state_action_planes(const float* state, const int64_t* action, float* out, int64_t batch,
                                          int32_t channels, int32_t plane, int32_t action_space, void* stream_) {
    if (!state || !action || !out || batch < 0 || batch > 65535 || channels <= 0 || plane <= 0 || action_space <= 0 ||
        static_cast<int64_t>(channels + 1) * plane > 0x7fffffff)
        return MZMCTS_ERR_INVALID;
    if (batch == 0) return MZMCTS_OK;
    const uint32_t state_floats = static_cast<uint32_t>(channels) * static_cast<uint32_t>(plane);
    const dim3 grid((state_floats + static_cast<uint32_t>(plane) + 255u) / 256u, static_cast<unsigned>(batch));
    mz::state_action_planes_kernel<<<grid, dim3(256), 0, static_cast<hipStream_t>(stream_)>>>(
        state, action, out, state_floats, static_cast<uint32_t>(plane), static_cast<float>(action_space));
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

extern "C" int mzmcts_unit_rescale(const float* x, float* out, int64_t rows, int32_t row_len, void* stream_) {
    if (!x || !out || rows < 0 || rows > 0x7fffffff || row_len <= 0 || row_len > 128) return MZMCTS_ERR_INVALID;
    if (rows == 0) return MZMCTS_OK;
    const size_t lds = sizeof(float) * mz::kRescaleRows * static_cast<size_t>(row_len);  // <= 128 KB
    const dim3 grid(static_cast<unsigned>((rows + mz::kRescaleRows - 1) / mz::kRescaleRows));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (lds > 64 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(mz::unit_rescale_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (err != hipSuccess) return MZMCTS_ERR_HIP;
    }
    mz::unit_rescale_kernel<<<grid, dim3(mz::kRescaleRows), lds, stream>>>(x, out, static_cast<uint32_t>(rows),
                                                                          static_cast<uint32_t>(row_len));
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

namespace mz {
// board_conv.hip: the heads of 16-channel 3 x 3 networks in the board-column shape (same bits as conv_head_mfma_kernel)
int launch_board_heads_cols(const float* const* xs, const mzmcts_head_desc* heads, int n_heads, float* const* outs,
                            int64_t batch, hipStream_t stream);
}

extern "C" int mzmcts_conv_heads_multi(const float* const* xs, const mzmcts_head_desc* heads, int32_t n_heads,
                                       float* const* outs, int64_t batch, void* stream_) {
    if (!xs || !heads || !outs || n_heads < 1 || n_heads > mz::kMaxHeads || batch < 0 || batch > 0x7fffffff)
        return MZMCTS_ERR_INVALID;
    for (int h = 0; h < n_heads; ++h)
        if (!xs[h] || (reinterpret_cast<uintptr_t>(xs[h]) & 15u)) return MZMCTS_ERR_INVALID;
    {
        const int rc = mz::launch_board_heads_cols(xs, heads, n_heads, outs, batch, static_cast<hipStream_t>(stream_));
        if (rc != MZMCTS_ERR_INVALID) return rc;
    }
    static const bool use_mfma = std::getenv("MZ_HEADS_WAVE_PER_SAMPLE") == nullptr;
    bool mfma = use_mfma;      // (any batch: a sample's logits do not depend on how many samples share its launch)
    for (int h = 0; h < n_heads && mfma; ++h) mfma = mz::mfma_head_ok(heads[h]);
    if (mfma) {
        mz::MfmaHeadSet mset{};
        size_t mlds = 0;
        for (int h = 0; h < n_heads; ++h) {
            const mzmcts_head_desc* d = heads + h;
            if (!outs[h] || !d->conv_w || !d->conv_b || !d->fc1_w || !d->fc1_b || !d->fc2_w || !d->fc2_b || d->channels <= 0 ||
                d->plane <= 0 || d->reduced <= 0 || d->hidden <= 0 || d->outputs <= 0 || d->channels != heads[0].channels ||
                d->plane != heads[0].plane)
                return MZMCTS_ERR_INVALID;
            mset.x[h] = xs[h];
            mset.desc[h] = *d;
            mset.shape[h] = mz::MfmaHeadShape{d->channels, d->plane, d->reduced, d->hidden, d->outputs};
            mset.out[h] = outs[h];
            mlds = std::max(mlds, sizeof(float) * static_cast<size_t>(mset.shape[h].total()));
        }
        if (mlds <= 160 * 1024) {
            if (batch == 0) return MZMCTS_OK;
            if (mlds > 64 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(mz::conv_head_mfma_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(mlds)) != hipSuccess)
                return MZMCTS_ERR_HIP;
            const int64_t tiles = (batch + mz::kTileSamples - 1) / mz::kTileSamples;
            const int64_t rounds = (tiles + mz::kHeadWaves - 1) / mz::kHeadWaves;
            const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / mlds)));
            const dim3 grid(static_cast<unsigned>(std::min<int64_t>(rounds, 256 * per_cu)), static_cast<unsigned>(n_heads));
            mz::conv_head_mfma_kernel<<<grid, dim3(64 * mz::kHeadWaves), mlds, static_cast<hipStream_t>(stream_)>>>(
                mset, static_cast<int>(batch));
            return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
        }
    }
    mz::HeadSet set{};
    size_t lds = 0;
    for (int h = 0; h < n_heads; ++h) {
        const mzmcts_head_desc* d = heads + h;
        if (!outs[h] || !d->conv_w || !d->conv_b || !d->fc1_w || !d->fc1_b || !d->fc2_w || !d->fc2_b || d->channels <= 0 ||
            d->plane <= 0 || d->reduced <= 0 || d->hidden <= 0 || d->outputs <= 0 || d->channels != heads[0].channels ||
            d->plane != heads[0].plane)
            return MZMCTS_ERR_INVALID;
        int split = 1;
        while (split * 2 * d->hidden <= 64) split *= 2;  // lanes per hidden unit (a power of two; 1 from 33 units on)
        set.x[h] = xs[h];
        set.desc[h] = *d;
        set.shape[h] = mz::HeadShape{d->channels, d->plane, d->reduced, d->hidden, d->outputs, split};
        set.out[h] = outs[h];
        lds = std::max(lds, sizeof(float) * static_cast<size_t>(set.shape[h].total()));
    }
    if (lds > 160 * 1024) return MZMCTS_ERR_INVALID;  // the caller keeps the torch modules
    if (batch == 0) return MZMCTS_OK;
    if (lds > 64 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(mz::conv_head_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (err != hipSuccess) return MZMCTS_ERR_HIP;
    }
    const int64_t rounds = (batch + mz::kHeadWaves - 1) / mz::kHeadWaves;
    const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds)));
    const dim3 grid(static_cast<unsigned>(std::min<int64_t>(rounds, 256 * per_cu)), static_cast<unsigned>(n_heads));
    mz::conv_head_kernel<<<grid, dim3(64 * mz::kHeadWaves), lds, static_cast<hipStream_t>(stream_)>>>(set,
                                                                                                    static_cast<int>(batch));
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}

extern "C" int mzmcts_conv_heads(const float* x, const mzmcts_head_desc* heads, int32_t n_heads, float* const* outs,
                                 int64_t batch, void* stream) {
    if (n_heads < 1 || n_heads > mz::kMaxHeads) return MZMCTS_ERR_INVALID;
    const float* xs[mz::kMaxHeads] = {x, x, x};
    return mzmcts_conv_heads_multi(xs, heads, n_heads, outs, batch, stream);
}

extern "C" int mzmcts_conv_head(const float* x, const mzmcts_head_desc* d, float* out, int64_t batch, void* stream) {
    float* outs[1] = {out};
    return mzmcts_conv_heads(x, d, 1, outs, batch, stream);
}

extern "C" int mzmcts_affine_act(const float* x, const float* scale, const float* shift, const float* residual, float* out,
                                 int64_t count, int32_t channels, int32_t plane, int32_t relu, void* stream_) {
    if (!x || !scale || !shift || !out || count < 0 || count > 0x7fffffff || channels <= 0 || plane <= 0 ||
        count % (static_cast<int64_t>(channels) * plane) != 0)
        return MZMCTS_ERR_INVALID;
    // 16-byte accesses: torch allocations are 256-byte aligned, views into them need not be
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(residual)) & 15u)
        return MZMCTS_ERR_INVALID;
    if (count == 0) return MZMCTS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t n = static_cast<uint32_t>(count);
    const dim3 grid((n / 4u + 255u + (n % 4u ? 1u : 0u)) / 256u), block(256);
    const uint32_t C = static_cast<uint32_t>(channels), P = static_cast<uint32_t>(plane);
    if (residual) {
        if (relu)
            mz::affine_act_kernel<true, true><<<grid, block, 0, stream>>>(x, scale, shift, residual, out, n, C, P);
        else
            mz::affine_act_kernel<true, false><<<grid, block, 0, stream>>>(x, scale, shift, residual, out, n, C, P);
    } else {
        if (relu)
            mz::affine_act_kernel<false, true><<<grid, block, 0, stream>>>(x, scale, shift, nullptr, out, n, C, P);
        else
            mz::affine_act_kernel<false, false><<<grid, block, 0, stream>>>(x, scale, shift, nullptr, out, n, C, P);
    }
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}
