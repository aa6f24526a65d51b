// net_kernels.hip -- element-wise epilogues of the residual networks' convolutions in inference mode.
//
// The convolutions themselves stay with PyTorch-ROCm / MIOpen (fp32 Winograd on these boards); what follows
// each of them in the reference -- BatchNorm2d in eval(), the residual add, ReLU (models.py:215-237, 249-275,
// 318-335) -- is three element-wise launches per convolution through torch and one through this kernel:
//     out = act(x * scale[c] + shift[c] (+ residual))        NCHW, c = (index / plane) % channels
// Same operations in the same order as the torch expression it replaces (mul, add, add, max; the library is
// built with -ffp-contract=off), so the results are bit-identical to it.
#include <hip/hip_runtime.h>

#include <cstdint>

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

}  // namespace mz

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
