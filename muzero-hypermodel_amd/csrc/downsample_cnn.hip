// downsample_cnn.hip -- the CNN down-sampler of the representation network (reference models.py:278-297, 318-327) in
// inference mode, one launch for its seven layers:
//     Conv2d(C, mid, 2 h, stride 4, padding 2) -> ReLU -> MaxPool2d(3, 2) -> Conv2d(mid, out, 5, padding 2) -> ReLU
//     -> MaxPool2d(3, 2) -> AdaptiveAvgPool2d(h, w)
// for BASELINE config #5's shape (4 x 84 x 84 frames, h = w = 6: a 12 x 12 stride-4 convolution to 20 x 20, pooled to
// 9 x 9, a 5 x 5 convolution, pooled to 4 x 4, spread to 6 x 6).  It is the root inference of every move of that config;
// through MIOpen the first convolution alone was 4.3 ms per 32768 frames plus 2.5 ms of layout transposes, and its
// immediate mode falls back to a per-image im2col + GEMM loop for this shape (two launches per frame).
//
// One workgroup (eight wavefronts, two per SIMD) per frame at a time, persistent over the batch:
//   * the frame lives in LDS, zero-padded ([C][88][88] floats: 121 KB); the NEXT frame's 16-byte global loads are issued
//     as soon as the first convolution of the current one is done (nothing after it reads the frame), land in registers
//     under the pooling and the second convolution, and are written over the frame at the end of the iteration;
//   * convolution 1 is an implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products and
//     sums): D[channel][position] over K = C * 12 * 12 = 576.  The K order is free as long as both operands use the
//     same one: lane quarter kk takes input channel kk (C = 4), and in "super-step" s = (ky, kx group g) a lane reads
//     ONE 16-byte LDS word -- four consecutive kx of its channel's row ky, at a compile-time offset from the lane's base
//     -- and feeds component m to product m of the super-step, whose weight operand is component m of the matching 16
//     bytes of the untouched [mid][C][12][12] weight tensor.  All 36 x 16 bytes of a lane's weights stay in registers
//     across the frames of the launch; the two wavefronts of a SIMD interleave their dependent product chains;
//   * pooling, the second convolution (matrix cores again, K = mid * 25, weights staged once in LDS k-major, the im2col
//     index arithmetic per lane) and the final pooling stay in LDS; 576 floats per frame go back to memory.
// fp32 throughout; the sums run in a different order than MIOpen's, so results agree with torch to rounding (tests).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mzmcts.h"

namespace mz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// torch.relu and torch's max pooling keep a NaN
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float max_keep_nan(float m, float v) { return (v > m || v != v) ? v : m; }

template <int C, int H, int W>
struct DownsampleShape {
    static constexpr int K1 = 12, S1 = 4, P1 = 2;                 // Conv2d(C, mid, 2 * ceil(H / 16), stride 4, padding 2)
    static constexpr int HP = H + 2 * P1, WP = W + 2 * P1;        // padded frame in LDS
    static constexpr int O1 = (H + 2 * P1 - K1) / S1 + 1;          // 20
    static constexpr int N1 = O1 * O1;                            // positions of convolution 1
    static constexpr int Q1 = (O1 - 3) / 2 + 1;                   // 9: after MaxPool2d(3, 2)
    static constexpr int QP = Q1 + 4;                             // padded by 2 for the 5 x 5 convolution
    static constexpr int N2 = Q1 * Q1;                            // positions of convolution 2
    static constexpr int Q2 = (Q1 - 3) / 2 + 1;                   // 4
    static constexpr int SUPER = K1 * (K1 / 4);                   // super-steps: (ky, kx / 4) of the lane quarter's channel
    static constexpr int TILES1 = (N1 + 15) / 16, TILES2 = (N2 + 15) / 16;
    static constexpr int kFrameFloats = C * HP * WP;
    static constexpr int kFrameLoads = C * H * (W / 4);           // 16-byte loads of one frame
    static_assert(C == 4, "one input channel per lane quarter");
    static_assert(W % 4 == 0 && WP % 4 == 0 && K1 % 4 == 0, "16-byte groups");
    // LDS floats: frame | y1 [mid][N1] (later y2 [16][N2]) | p1 [mid][QP][QP] | w2 k-major [4 steps2][16]
    // (every region a whole number of 16-byte words: the one-time clearing runs in such words)
    static constexpr int y1_floats(int mid) { return (mid * N1 + 3) / 4 * 4; }
    static constexpr int p1_floats(int mid) { return (mid * QP * QP + 3) / 4 * 4; }
    static constexpr int steps2(int mid) { return ((mid * 25 + 3) / 4 + 3) / 4 * 4; }   // K steps of convolution 2, in fours
    static constexpr size_t lds_floats(int mid) {
        return static_cast<size_t>(kFrameFloats) + y1_floats(mid) + p1_floats(mid) + static_cast<size_t>(steps2(mid)) * 4 * 16;
    }
};

constexpr int kDownWaves = 8;      // two wavefronts per SIMD: one multiplies while the other waits for its LDS word
constexpr int kDownChains = 1;     // tiles (independent product chains) a wavefront runs side by side

template <int C, int H, int W>
__global__ __launch_bounds__(64 * kDownWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) void downsample_cnn_kernel(
    const float* __restrict__ x, int batch, const float* __restrict__ w1, const float* __restrict__ b1, int mid,
    const float* __restrict__ w2, const float* __restrict__ b2, int cout, int out_h, int out_w, float* __restrict__ out) {
    using S = DownsampleShape<C, H, W>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K2 = mid * 25, steps2 = S::steps2(mid);
    float* frame = lds;                          // [C][HP][WP], borders zero
    float* y1 = frame + S::kFrameFloats;         // [mid][N1] convolution 1 + bias + ReLU; later y2 [16][N2]
    float* p1 = y1 + S::y1_floats(mid);          // [mid][QP][QP] pooled, borders zero
    float* w2s = p1 + S::p1_floats(mid);         // [4 steps2][16]: w2[channel][k] k-major, zero beyond K2 / cout
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
    constexpr int THREADS = 64 * kDownWaves;
    constexpr int LOADS = (S::kFrameLoads + THREADS - 1) / THREADS;   // per thread and frame

    // borders (and everything else) to zero once: the frames and pooled maps only ever rewrite the interiors
    for (int i = tid; i < static_cast<int>(S::lds_floats(mid)) / 4; i += THREADS)
        reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < 4 * steps2 * 16; i += THREADS) {
        const int k = i >> 4, n = i & 15;
        if (k < K2 && n < cout) w2s[i] = w2[static_cast<size_t>(n) * K2 + k];
    }

    // this lane's weights of convolution 1: channel kk of the input, super-step s = (ky, g): the 16 bytes
    // w1[output channel i16][kk][ky][4 g .. 4 g + 3]
    f32x4 wa[S::SUPER];
#pragma unroll
    for (int s = 0; s < S::SUPER; ++s) {
        const int ky = s / (S::K1 / 4), g = s - ky * (S::K1 / 4);
        wa[s] = i16 < mid ? *reinterpret_cast<const f32x4*>(w1 + ((static_cast<size_t>(i16) * C + kk) * S::K1 + ky) * S::K1 + 4 * g)
                          : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float bias1[4], bias2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bias1[r] = 4 * kk + r < mid ? b1[4 * kk + r] : 0.f;
        bias2[r] = 4 * kk + r < cout ? b2[4 * kk + r] : 0.f;
    }

    // a frame's 16-byte loads of this thread, and their places in the padded LDS frame (8-byte aligned: + P1 columns)
    float4 next[LOADS];
    auto fetch = [&](int b) {
        const float4* src = reinterpret_cast<const float4*>(x + static_cast<size_t>(b) * C * H * W);
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int i = tid + u * THREADS;
            next[u] = i < S::kFrameLoads ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto deposit = [&]() {
        constexpr int PER_ROW = W / 4;
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int i = tid + u * THREADS;
            if (i < S::kFrameLoads) {
                const int row = i / PER_ROW, q = i - row * PER_ROW;      // row = c * H + y
                const int c = row / H, y = row - c * H;
                float2* dst = reinterpret_cast<float2*>(frame + (c * S::HP + y + S::P1) * S::WP + S::P1 + 4 * q);
                dst[0] = make_float2(next[u].x, next[u].y);
                dst[1] = make_float2(next[u].z, next[u].w);
            }
        }
    };
    if (static_cast<int>(blockIdx.x) < batch) {
        fetch(blockIdx.x);
        deposit();
    }
    // (every load so far -- weights, biases, the first frame -- has landed: inside the loop only the next frame's loads
    //  are in flight, and nothing before the deposit waits for them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int b = blockIdx.x; b < batch; b += gridDim.x) {
        const int b_next = b + gridDim.x;

        // ---- convolution 1 + bias + ReLU: tiles of 16 positions, D[channel 4 kk + r][position i16]; a wavefront takes
        //      kDownChains tiles at a time (independent product chains on one set of weight operands) ----
        for (int tile = wave; tile < S::TILES1; tile += kDownChains * kDownWaves) {
            const float* base[kDownChains];
            int pos[kDownChains];
            f32x4 acc[kDownChains];
#pragma unroll
            for (int t = 0; t < kDownChains; ++t) {
                const int p = (tile + t * kDownWaves) * 16 + i16;
                pos[t] = (tile + t * kDownWaves < S::TILES1 && p < S::N1) ? p : -1;
                const int pc = pos[t] >= 0 ? p : 0;
                const int oy = pc / S::O1, ox = pc - oy * S::O1;
                base[t] = frame + (kk * S::HP + oy * S::S1) * S::WP + ox * S::S1;
                acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // (the LDS word of super-step s + 1 is asked for before the products of super-step s)
            constexpr int G = S::K1 / 4;
            f32x4 v[kDownChains], ahead[kDownChains];
#pragma unroll
            for (int t = 0; t < kDownChains; ++t) v[t] = *reinterpret_cast<const f32x4*>(base[t]);
#pragma unroll
            for (int s = 0; s < S::SUPER; ++s) {
                if (s + 1 < S::SUPER) {
                    const int off = ((s + 1) / G) * S::WP + 4 * ((s + 1) % G);
#pragma unroll
                    for (int t = 0; t < kDownChains; ++t) ahead[t] = *reinterpret_cast<const f32x4*>(base[t] + off);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int t = 0; t < kDownChains; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][m], v[t][m], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < kDownChains; ++t) v[t] = ahead[t];
            }
#pragma unroll
            for (int t = 0; t < kDownChains; ++t) {
                if (pos[t] >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * kk + r < mid) y1[(4 * kk + r) * S::N1 + pos[t]] = relu_keep_nan(acc[t][r] + bias1[r]);
                }
            }
        }
        __syncthreads();
        // the next frame's loads: in flight under the pooling and the second convolution (nothing below reads the frame,
        // and the registers of the first convolution's operands are free now)
        // (unconditional, so that the registers are plainly dead during the first convolution: the last iteration fetches
        //  its own frame again)
        fetch(b_next < batch ? b_next : b);

        // ---- MaxPool2d(3, 2) into the padded map ----
        for (int i = tid; i < mid * S::N2; i += THREADS) {
            const int c = i / S::N2, q = i - c * S::N2;
            const int py = q / S::Q1, px = q - py * S::Q1;
            const float* src = y1 + c * S::N1 + (2 * py) * S::O1 + 2 * px;
            float m = src[0];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) m = max_keep_nan(m, src[dy * S::O1 + dx]);
            p1[(c * S::QP + py + 2) * S::QP + px + 2] = m;
        }
        __syncthreads();

        // ---- convolution 2 + bias + ReLU (y2 over y1's place): K index k = (c, ky, kx) = 25 c + 5 ky + kx ----
        float* y2 = y1;
        for (int tile = wave; tile < S::TILES2; tile += kDownWaves) {
            const int p = tile * 16 + i16;
            const int pc = p < S::N2 ? p : 0;
            const int py = pc / S::Q1, px = pc - py * S::Q1;
            const float* base = p1 + py * S::QP + px;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int ks0 = 0; ks0 < steps2; ks0 += 4) {            // four steps' operands asked for before their products
                float a[4], v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 4 * (ks0 + u) + kk;
                    const int kc = k < K2 ? k : 0;                 // (w2s is zero beyond K2: the product vanishes)
                    const int c = kc / 25, r = kc - 25 * c;
                    const int ky = r / 5, kx = r - 5 * ky;
                    a[u] = w2s[k * 16 + i16];
                    v[u] = base[(c * S::QP + ky) * S::QP + kx];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], v[u], acc, 0, 0, 0);
            }
            if (p < S::N2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y2[(4 * kk + r) * S::N2 + p] = relu_keep_nan(acc[r] + bias2[r]);
            }
        }
        __syncthreads();

        // ---- MaxPool2d(3, 2) to Q2 x Q2, then AdaptiveAvgPool2d: window [floor(i Q2 / h), ceil((i + 1) Q2 / h)) ----
        float* out_b = out + static_cast<size_t>(b) * cout * out_h * out_w;
        for (int i = tid; i < cout * out_h * out_w; i += THREADS) {
            const int c = i / (out_h * out_w), q = i - c * (out_h * out_w);
            const int oi = q / out_w, oj = q - oi * out_w;
            const int y0 = (oi * S::Q2) / out_h, y1e = ((oi + 1) * S::Q2 + out_h - 1) / out_h;
            const int x0 = (oj * S::Q2) / out_w, x1e = ((oj + 1) * S::Q2 + out_w - 1) / out_w;
            float sum = 0.f;
            for (int yy = y0; yy < y1e; ++yy)
                for (int xx = x0; xx < x1e; ++xx) {
                    const float* src = y2 + c * S::N2 + (2 * yy) * S::Q1 + 2 * xx;
                    float m = src[0];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) m = max_keep_nan(m, src[dy * S::Q1 + dx]);
                    sum += m;
                }
            out_b[i] = sum / static_cast<float>((y1e - y0) * (x1e - x0));
        }
        deposit();
        __syncthreads();   // (the next frame's convolution 1 reads the frame and rewrites y1 / y2)
    }
}

}  // namespace mz

extern "C" int mzmcts_downsample_cnn(const float* x, int64_t batch, int32_t channels, int32_t height, int32_t width,
                                     const float* w1, const float* b1, int32_t mid, int32_t kernel1, const float* w2,
                                     const float* b2, int32_t cout, int32_t out_h, int32_t out_w, float* out, void* stream_) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !out || batch < 0 || batch > 0x7fffffff) return MZMCTS_ERR_INVALID;
    // the one shape family this launch covers (config #5); anything else stays with the caller's convolution library
    if (channels != 4 || height != 84 || width != 84 || kernel1 != 12 || mid < 4 || mid > 16 || cout < 1 || cout > 16 ||
        out_h < 1 || out_h > 8 || out_w < 1 || out_w > 8)
        return MZMCTS_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w1)) & 15u) return MZMCTS_ERR_INVALID;
    using S = mz::DownsampleShape<4, 84, 84>;
    const size_t lds_bytes = sizeof(float) * S::lds_floats(mid);
    if (lds_bytes > 160 * 1024) return MZMCTS_ERR_INVALID;         // (mid > 10 with this frame size: does not fit a CU's LDS)
    if (batch == 0) return MZMCTS_OK;
    auto kernel = mz::downsample_cnn_kernel<4, 84, 84>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds_bytes)) != hipSuccess)
        return MZMCTS_ERR_HIP;
    int device = 0, cus = 256;
    if (hipGetDevice(&device) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const int grid = static_cast<int>(batch < cus ? batch : cus);   // one resident workgroup per CU (LDS), persistent
    kernel<<<dim3(grid), dim3(64 * mz::kDownWaves), lds_bytes, static_cast<hipStream_t>(stream_)>>>(
        x, static_cast<int>(batch), w1, b1, mid, w2, b2, cout, out_h, out_w, out);
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}
