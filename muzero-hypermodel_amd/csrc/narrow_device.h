// narrow_device.h -- the "narrow" fully-connected fast path: MuZeroFullyConnectedNetwork (reference
// models.py:84-195) for networks whose every layer fits one 16-lane DPP row, plus the tree walk and
// the backup written for a tree that lives in LDS and is owned by exactly that row.
//
// Why a second implementation next to fc_net_device.h / tree_device.h: at the headline configuration
// (4096 envs on 256 CUs) every SIMD runs ONE wavefront, so a move costs exactly the length of one
// tree's dependent instruction chain.  The generic code pays an LDS round trip (~130 cycles, nobody to
// hide it) per layer boundary, per backup hand-over and per cross-lane publish; here
//   * a layer is 16 `v_fmac_f32` with a DPP row-rotate on the activation operand: lane j holds neuron
//     j's activation, step r multiplies the activation of lane src_r(j) with W[j][src_r(j)], whose
//     copy in LDS is laid out in rotation order -- activations never leave the registers;
//   * the value recursion of the backup runs down the lanes with `row_shl:1` (lane = path level);
//   * the descent keeps per child the term r + discount * (+-Q) the backup already computes for the
//     min-max statistics (bit-identical to recomputing it in ucb_score) and reads
//     (log(..)+init) * (sqrt(N) / (n+1)) from a table over (N, n) built with the same IEEE operations,
//     leaving one fp64 division (the normalisation) on the chain.
// The fp64 tree arithmetic performs the reference's operations on the reference's operands, so the
// bit-exactness contract of tree_device.h holds unchanged; the fp32 network accumulates each neuron in
// rotation order with FMAs, which moves logits by ~1e-7 (inside the 1e-5 bar of BASELINE.json).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fc_net_device.h"
#include "tree_device.h"

namespace mz {

constexpr int kRow = 16;  // lanes per tree == one DPP row

// units of 16x16 weights, in rotation order; slots that a network does not use stay zero
enum NarrowUnit : int {
    kURepr0 = 0,  // representation: first Linear (obs -> hidden, or obs -> enc when there is no hidden layer)
    kURepr1,      // representation: hidden -> enc (only with a hidden layer)
    kUDyn1,       // dynamics: [state | one-hot action] -> hidden
    kUDyn2,       // dynamics: hidden -> enc
    kURew1,
    kUVal1,
    kUPol1,
    kURew2a,      // reward support logits 0..15
    kURew2b,      // reward support logits 16..31
    kUVal2a,
    kUVal2b,
    kUPol2,
    kNarrowUnits
};

struct NarrowLayout {
    uint32_t off_pbc;      // double[2][S+1]
    uint32_t off_pbc2;     // double[(S+1)(S+2)/2] (pbc2_mode 1), double[S+1][64] (mode 2) or 0xffffffff (mode 0)
    uint32_t off_units;    // float4[kNarrowUnits][4][16]
    uint32_t off_bias;     // float[kNarrowUnits][16]
    uint32_t off_trees;
    uint32_t tree_bytes;
    uint32_t off_side;     // within a tree region: SideStats[(S+1)][SPAN] (SPAN = 2 for two actions, else pow2 >= max(A, 4))
    uint32_t off_path;     // int32[S]
    uint32_t off_hidden;   // float[(S+1)][enc]
    uint32_t off_misc;     // int32[16] root actions | float[16] root policy logits
    uint32_t off_desc;     // uint8[(S+1)][16] descendant tables (two-action trees: descend_window), else unused
    uint32_t total_bytes;
    int32_t waves;         // wavefronts per workgroup
    int32_t rows;          // trees (16-lane rows) a wavefront carries: 4, or fewer to cut the wait for its deepest tree
    int32_t pbc2_mode;     // exploration table over (N, n): 0 none, 1 triangular, 2 rows of 64 (index = N << 6 | n; S <= 63)
    int32_t exact_division; // != 0: every quotient by the division itself (MZMCTS_NARROW_EXACT_DIV=1; the tests' A/B of the
                           // reciprocal-prepared forms, see Normalizer)
};

template <int R>
__device__ __forceinline__ int row_ror_bits(int v) {
    if constexpr (R == 0) return v;
    else return __builtin_amdgcn_update_dpp(0, v, 0x120 + R, 0xF, 0xF, true);  // row_ror:R
}
template <int R>
__device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, row_ror_bits<R>(__builtin_bit_cast(int, v)));
}
// lane i of a row receives lane i+1's value; lane 15 receives 0
__device__ __forceinline__ int row_shl1_bits(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true); }
// lane i of a row receives lane i-1's value; lane 0 receives 0
__device__ __forceinline__ int row_shr1_bits(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); }
__device__ __forceinline__ double row_shl1(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = row_shl1_bits(static_cast<int>(bits & 0xffffffffll));
    const int hi = row_shl1_bits(static_cast<int>(bits >> 32));
    return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

// fmax / fmin on values that are never signalling NaNs, without the operand canonicalisation the library calls
// carry (three instructions per call instead of one)
__device__ __forceinline__ double dmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double dmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// OR over the row, every lane receives the result
__device__ __forceinline__ int row_or(int v) {
    v |= partner_bits<1>(v);
    v |= partner_bits<2>(v);
    v |= partner_bits<4>(v);
    v |= partner_bits<8>(v);
    return v;
}

// ---- weights in rotation order -----------------------------------------------------------------------
struct NarrowUnitSource {
    FcLayer layer;
    int out_base;  // neuron of lane 0
    bool active;
};

__device__ __forceinline__ NarrowUnitSource narrow_unit_source(const FcNet& net, int u) {
    NarrowUnitSource s{};
    s.active = true;
    s.out_base = 0;
    const bool repr_hidden = net.repr.n_layers == 2;
    switch (u) {
        case kURepr0: s.layer = net.repr.layer[0]; break;
        case kURepr1: s.layer = net.repr.layer[1]; s.active = repr_hidden; break;
        case kUDyn1: s.layer = net.dyn.layer[0]; break;
        case kUDyn2: s.layer = net.dyn.layer[1]; break;
        case kURew1: s.layer = net.reward.layer[0]; break;
        case kUVal1: s.layer = net.value.layer[0]; break;
        case kUPol1: s.layer = net.policy.layer[0]; break;
        case kURew2a: s.layer = net.reward.layer[1]; break;
        case kURew2b: s.layer = net.reward.layer[1]; s.out_base = kRow; s.active = net.F > kRow; break;
        case kUVal2a: s.layer = net.value.layer[1]; break;
        case kUVal2b: s.layer = net.value.layer[1]; s.out_base = kRow; s.active = net.F > kRow; break;
        default: s.layer = net.policy.layer[1]; break;
    }
    return s;
}

// All threads of the workgroup; the caller synchronises afterwards.  Thread t stages for lane j = t % 16
// the 16 weights of its neuron in the order its rotate steps will meet the activations: the source lane of
// step r is measured by rotating the lane index itself, so the table is right by construction.
__device__ __forceinline__ void stage_narrow_units(const FcNet& net, const float* __restrict__ flat, float4* units,
                                                   float* bias, int tid, int nthreads) {
    const int j = tid & (kRow - 1);
    int src[kRow];
#define MZ_SRC(R) src[R] = row_ror_bits<R>(j);
    MZ_SRC(0) MZ_SRC(1) MZ_SRC(2) MZ_SRC(3) MZ_SRC(4) MZ_SRC(5) MZ_SRC(6) MZ_SRC(7)
    MZ_SRC(8) MZ_SRC(9) MZ_SRC(10) MZ_SRC(11) MZ_SRC(12) MZ_SRC(13) MZ_SRC(14) MZ_SRC(15)
#undef MZ_SRC
    for (int u = tid / kRow; u < kNarrowUnits; u += nthreads / kRow) {
        const NarrowUnitSource s = narrow_unit_source(net, u);
        const int n = s.out_base + j;
        const bool row_ok = s.active && n < s.layer.out;
        float w[kRow];
#pragma unroll
        for (int r = 0; r < kRow; ++r) {
            const int k = src[r];
            w[r] = (row_ok && k < s.layer.in) ? flat[s.layer.w_off + n * s.layer.in + k] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) units[(u * 4 + c) * kRow + j] = float4{w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]};
        bias[u * kRow + j] = row_ok ? flat[s.layer.b_off + n] : 0.f;
    }
}

// bias + W x for this lane's neuron of unit U; x holds the layer input, element k in lane k (0 beyond it).
// Sixteen v_fmac_f32 whose activation operand carries the DPP row rotate, written out because the compiler
// will not fold a DPP move into the tied-accumulator form (it emits mov_dpp + fmac + hazard nops: 2.4x the
// issue slots).  Two accumulators halve the dependent chain.  Hazards the assembler cannot see inside the
// block: a VALU write of x (or of EXEC) immediately before the first DPP read -- covered by the leading nop.
struct UnitWeights {
    float4 w0, w1, w2, w3;
    float bias;
};

// the 16 weights (rotation order) and the bias of this lane's neuron of unit U: four 16-byte LDS reads
template <int U>
__device__ __forceinline__ UnitWeights load_unit(const float4* units, const float* bias, int j) {
    const float4* w = units + (U * 4) * kRow + j;
    return UnitWeights{w[0], w[kRow], w[2 * kRow], w[3 * kRow], bias[U * kRow + j]};
}

__device__ __forceinline__ float apply_unit(const UnitWeights& u, float x) {
    float a = u.bias;
    float b = 0.f;
    asm volatile(
        "s_nop 4\n\t"
        "v_fmac_f32_e32 %0, %2, %3\n\t"
        "v_fmac_f32_dpp %1, %2, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %5 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %6 row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %7 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %8 row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %9 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %10 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %12 row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %13 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %14 row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %15 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %16 row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %17 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %18 row_ror:15 row_mask:0xf bank_mask:0xf"
        : "+v"(a), "+v"(b)
        : "v"(x), "v"(u.w0.x), "v"(u.w0.y), "v"(u.w0.z), "v"(u.w0.w), "v"(u.w1.x), "v"(u.w1.y), "v"(u.w1.z), "v"(u.w1.w),
          "v"(u.w2.x), "v"(u.w2.y), "v"(u.w2.z), "v"(u.w2.w), "v"(u.w3.x), "v"(u.w3.y), "v"(u.w3.z), "v"(u.w3.w));
    return a + b;
}

// bias + W x for this lane's neuron of unit U; x holds the layer input, element k in lane k (0 beyond it).
// Sixteen v_fmac_f32 whose activation operand carries the DPP row rotate, written out because the compiler
// will not fold a DPP move into the tied-accumulator form (it emits mov_dpp + fmac + hazard nops: 2.4x the
// issue slots).  Two accumulators halve the dependent chain.  Hazards the assembler cannot see inside the
// block: a VALU write of x (or of EXEC) immediately before the first DPP read -- covered by the leading nop.
template <int U>
__device__ __forceinline__ float narrow_unit(const float4* units, const float* bias, float x, int j) {
    return apply_unit(load_unit<U>(units, bias, j), x);
}

// expf for arguments that are never positive (ELU's negative side, soft-max terms x - max): the library's sequence --
// x log2(e) split into a rounded part and a remainder, v_exp_f32, v_ldexp_f32, zero below -103.97 -- without its final
// select for arguments above 88.7 (the compiled code of the two differs in exactly that select and its compare).
__device__ __forceinline__ float exp_nonpositive(float x) {
    const float p = 0x1.715476p+0f * x;
    const float p_error = __builtin_fmaf(x, 0x1.715476p+0f, -p);
    const float n = __builtin_rintf(p);
    const float low = __builtin_fmaf(x, 0x1.4ae0bep-26f, p_error);
    const float f = (p - n) + low;
    const float r = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), static_cast<int>(n));
    return x < -0x1.9d1da0p+6f ? 0.f : r;
}

// 1.0f / d for a sum of soft-max terms (1 <= d <= 32: the term of the maximum is 1): the division sequence without
// v_div_scale / v_div_fixup, which are the identity on such operands -- the same seven operations, the same bits
__device__ __forceinline__ float reciprocal_of_sum(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    const float e2 = __builtin_fmaf(-d, r1, 1.0f);
    const float q1 = __builtin_fmaf(e2, r1, r1);
    const float e3 = __builtin_fmaf(-d, q1, 1.0f);
    return __builtin_fmaf(e3, r1, q1);
}

// (the exponentials are computed for every lane and selected afterwards -- the empty asm keeps the compiler from wrapping
//  them in a branch on the selecting condition, which costs more than the few lanes it would spare)
__device__ __forceinline__ float narrow_elu(float v) {
    float e = exp_nonpositive(fminf(v, 0.f)) - 1.0f;
    asm volatile("" : "+v"(e));
    return v > 0.f ? v : e;
}

// models.py:137-145 / 161-168: min-max rescale of the first `enc` lanes to [0, 1]; lanes beyond read 0
__device__ __forceinline__ float narrow_rescale(float raw, int enc, int j) {
    const bool in = j < enc;
    float mn = in ? raw : INFINITY, mx = in ? raw : -INFINITY;
    const int span = enc <= 8 ? 8 : kRow;   // steps 1, 2, 4 pair lanes inside the lower half row: enough for 8 elements
    MZ_BUTTERFLY(kRow, span, (mn = fminf(mn, partner<M>(mn)), mx = fmaxf(mx, partner<M>(mx))));
    float scale = mx - mn;
    if (scale < 1e-5f) scale += 1e-5f;
    return in ? (raw - mn) / scale : 0.f;
}

struct NarrowHeads {
    float norm;             // next hidden state (lanes < enc)
    float reward_a, reward_b, value_a, value_b;  // support logits: element j and 16 + j
    float policy;           // logits, lanes < A
};

// models.py:147-170, 192-195 recurrent_inference; x0 = [hidden | one-hot(action)] across the lanes
// the weights a lane needs for the first phases of every recurrent inference: kept in registers across the
// simulations of a move (the kernel runs one wave per SIMD: the register file is otherwise idle)
struct ResidentWeights {
    UnitWeights dyn1, dyn2, rew1, val1, pol1;
};

__device__ __forceinline__ ResidentWeights load_resident_weights(const float4* units, const float* bias, int j) {
    return ResidentWeights{load_unit<kUDyn1>(units, bias, j), load_unit<kUDyn2>(units, bias, j),
                           load_unit<kURew1>(units, bias, j), load_unit<kUVal1>(units, bias, j),
                           load_unit<kUPol1>(units, bias, j)};
}

// the remaining weights are asked for a phase ahead of their use, so that their LDS round trips run under the
// previous phase's arithmetic (the addresses never change; only the activations are on the dependent chain)
// after_first / after_second: work the caller wants issued behind the first / second layer of the dynamics function
// (the fused kernel's backup asks for its tree records there, so their LDS round trips run under the network)
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
template <typename Hook1 = NoHook, typename Hook2 = NoHook>
__device__ __forceinline__ NarrowHeads narrow_recurrent(const float4* units, const float* bias, int enc, bool wide_support,
                                                        float x0, int j, const ResidentWeights& resident,
                                                        Hook1 after_first = Hook1{}, Hook2 after_second = Hook2{}) {
    NarrowHeads h{};
    const UnitWeights &wd1 = resident.dyn1, &wd2 = resident.dyn2, &wr1 = resident.rew1, &wv1 = resident.val1,
                      &wp1 = resident.pol1;
    const float d1 = narrow_elu(apply_unit(wd1, x0));
    after_first();
    const float raw = apply_unit(wd2, d1);
    after_second();
    const UnitWeights wr2 = load_unit<kURew2a>(units, bias, j), wv2 = load_unit<kUVal2a>(units, bias, j),
                      wp2 = load_unit<kUPol2>(units, bias, j);
    UnitWeights wr2b{}, wv2b{};
    if (wide_support) {
        wr2b = load_unit<kURew2b>(units, bias, j);
        wv2b = load_unit<kUVal2b>(units, bias, j);
    }
    h.norm = narrow_rescale(raw, enc, j);
    // the reward head reads the UN-normalised next state (models.py:157-159)
    const float r1 = narrow_elu(apply_unit(wr1, raw));
    const float v1 = narrow_elu(apply_unit(wv1, h.norm));
    const float p1 = narrow_elu(apply_unit(wp1, h.norm));
    h.reward_a = apply_unit(wr2, r1);
    h.value_a = apply_unit(wv2, v1);
    h.policy = apply_unit(wp2, p1);
    h.reward_b = 0.f;
    h.value_b = 0.f;
    if (wide_support) {
        h.reward_b = apply_unit(wr2b, r1);
        h.value_b = apply_unit(wv2b, v1);
    }
    return h;
}

// models.py:172-190 initial_inference (reward = log(one_hot(centre)), decodes to exactly 0)
__device__ __forceinline__ NarrowHeads narrow_initial(const float4* units, const float* bias, int enc, bool wide_support,
                                                      bool repr_hidden, float obs, int j) {
    NarrowHeads h{};
    float raw = narrow_unit<kURepr0>(units, bias, obs, j);
    if (repr_hidden) raw = narrow_unit<kURepr1>(units, bias, narrow_elu(raw), j);
    h.norm = narrow_rescale(raw, enc, j);
    const float v1 = narrow_elu(narrow_unit<kUVal1>(units, bias, h.norm, j));
    const float p1 = narrow_elu(narrow_unit<kUPol1>(units, bias, h.norm, j));
    h.value_a = narrow_unit<kUVal2a>(units, bias, v1, j);
    h.policy = narrow_unit<kUPol2>(units, bias, p1, j);
    h.value_b = wide_support ? narrow_unit<kUVal2b>(units, bias, v1, j) : 0.f;
    h.reward_a = 0.f;
    h.reward_b = 0.f;
    return h;
}

// models.py:656-661 (invert the value scaling) with the two library calls of tree_device.h's inverse_value_transform cut
// down to what its operands need -- the same instructions in the same order, so the same bits:
//   * sqrtf(1 + w), 1 <= 1 + w: the library's v_sqrt_f32 + one-ulp-down / one-ulp-up residual test, without the scaling
//     it wraps around operands below 2^-96 and the zero / infinity pass-through;
//   * r / 0.002f, 0.002 <= r: the division sequence's refined reciprocal of the CONSTANT denominator is computed once
//     (inverse_transform_reciprocal, from an operand the compiler cannot fold) and a quotient pays the sequence's last
//     five operations; v_div_scale / v_div_fixup are the identity on these operands.
__device__ __forceinline__ float inverse_transform_reciprocal() {
    float c = 0.002f;
    asm volatile("" : "+v"(c));                  // (the hardware's v_rcp_f32 of it, not a folded constant)
    const float r0 = __builtin_amdgcn_rcpf(c);
    const float e = __builtin_fmaf(-c, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float inverse_value_transform_narrow(float x, float reciprocal) {
    const float u = (fabsf(x) + 1.0f) + 0.001f;
    const float w = 0.004f * u;
    const float a = 1.0f + w;
    float root;
    {
        const float s = __builtin_amdgcn_sqrtf(a);
        const float down = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
        const float up = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
        const float r_down = __builtin_fmaf(-down, s, a);
        const float r_up = __builtin_fmaf(-up, s, a);
        root = 0.f >= r_down ? down : s;
        root = 0.f < r_up ? up : root;
    }
    const float r = root - 1.0f;
    float q;
    {
        const float c = 0.002f;
        const float q0 = r * reciprocal;
        const float e2 = __builtin_fmaf(-c, q0, r);
        const float q1 = __builtin_fmaf(e2, reciprocal, q0);
        const float e3 = __builtin_fmaf(-c, q1, r);
        q = __builtin_fmaf(e3, reciprocal, q1);
    }
    const float y = q * q - 1.0f;
    const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    return sgn * y;
}

// models.py:641-662 support_to_scalar for two logit vectors held in registers (element j in `a`, 16 + j in
// `b`); the two reductions are interleaved.  Every lane of the row receives both results.
// CUT: the decode ends in inverse_value_transform_narrow (`reciprocal` from inverse_transform_reciprocal)
template <bool CUT = false>
__device__ __forceinline__ void narrow_support_pair(float va, float vb, float ra, float rb, int F, int support, int j,
                                                    float& value, float& reward, float reciprocal = 0.f) {
    const bool in_a = j < F, in_b = kRow + j < F;
    const float xva = in_a ? va : -INFINITY, xvb = in_b ? vb : -INFINITY;
    const float xra = in_a ? ra : -INFINITY, xrb = in_b ? rb : -INFINITY;
    float mv = fmaxf(xva, xvb), mr = fmaxf(xra, xrb);
    MZ_BUTTERFLY(kRow, kRow, (mv = fmaxf(mv, partner<M>(mv)), mr = fmaxf(mr, partner<M>(mr))));
    float tva = exp_nonpositive(xva - mv), tvb = exp_nonpositive(xvb - mv);
    float tra = exp_nonpositive(xra - mr), trb = exp_nonpositive(xrb - mr);
    asm volatile("" : "+v"(tva), "+v"(tvb), "+v"(tra), "+v"(trb));
    const float eva = in_a ? tva : 0.f, evb = in_b ? tvb : 0.f;
    const float era = in_a ? tra : 0.f, erb = in_b ? trb : 0.f;
    float sv = eva + evb, sr = era + erb;
    MZ_BUTTERFLY(kRow, kRow, (sv = sv + partner<M>(sv), sr = sr + partner<M>(sr)));
    const float iv = CUT ? reciprocal_of_sum(sv) : 1.0f / sv, ir = CUT ? reciprocal_of_sum(sr) : 1.0f / sr;
    const float fa = static_cast<float>(j - support), fb = static_cast<float>(kRow + j - support);
    float av = fa * (eva * iv) + fb * (evb * iv);
    float ar = fa * (era * ir) + fb * (erb * ir);
    MZ_BUTTERFLY(kRow, kRow, (av = av + partner<M>(av), ar = ar + partner<M>(ar)));
    if constexpr (CUT) {
        value = inverse_value_transform_narrow(av, reciprocal);
        reward = inverse_value_transform_narrow(ar, reciprocal);
    } else {
        value = inverse_value_transform(av);
        reward = inverse_value_transform(ar);
    }
}

// fp32 softmax over the lanes with valid == true (Node.expand, self_play.py:461-463), widened like .tolist()
// SPAN: lanes 0 .. SPAN-1 hold every valid entry, and only they receive a meaningful result
// CUT: 1 / sum by reciprocal_of_sum (at least one valid entry among the SPAN lanes)
template <int SPAN = kRow, bool CUT = false>
__device__ __forceinline__ double narrow_softmax(float logit, bool valid) {
    float m = valid ? logit : -INFINITY;
    MZ_BUTTERFLY(SPAN, SPAN, m = fmaxf(m, partner<M>(m)));
    float t = exp_nonpositive(logit - m);
    asm volatile("" : "+v"(t));
    const float e = valid ? t : 0.f;
    float s = e;
    MZ_BUTTERFLY(SPAN, SPAN, s = s + partner<M>(s));
    return static_cast<double>(e * (CUT ? reciprocal_of_sum(s) : 1.0f / s));
}

// ---- tree in LDS with the per-child value term ---------------------------------------------------------
// A block keeps, per child c, the pair { vterm, prior } (16 bytes at c * 16: everything a score needs, one
// ds_read_b128) and ChildLinks (16 bytes at 16 * SPAN + c * 16); SPAN = pow2 >= A is the kernel's template
// argument, so strides are shifts and the links' distance is an instruction offset.  vterm = r + discount*(+-Q)
// is what the backup computes for the min-max statistics; it sits where the HBM record has value_sum, and the
// value sums (which only the backup touches) live in a side array -- the publish step puts them back.
// With exactly two actions (SPAN == 2, the pair-wise descent) the second member is not the prior but the whole
// prior_score = table[N][n] * prior of ucb_score, refreshed by the backup for both children of every node on the
// path (N or n changed for exactly those) -- lane-parallel over the levels there, instead of a dependent LDS
// round trip per level of the descent; the prior itself then sits in the side array next to the value sum.
using lds_u8 = __attribute__((address_space(3))) uint8_t;
using f64x2 = double __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const lds_u8*)p));
}
template <typename T>
__device__ __forceinline__ T lds_load(uint32_t address) {
    return *(const __attribute__((address_space(3))) T*)(static_cast<uintptr_t>(address));
}

struct alignas(16) SideStats {
    double value_sum;  // Node.value_sum
    double prior;      // Node.prior (SPAN == 2 only: elsewhere the block itself holds it)
};

constexpr int narrow_log2(int span) { return span == 2 ? 1 : span == 4 ? 2 : span == 8 ? 3 : 4; }

template <int SPAN>
struct LdsTreeV {
    static constexpr bool kInLds = true;
    static constexpr int kShift = narrow_log2(SPAN);
    static constexpr uint32_t kBlockStride = 32u * SPAN;
    static constexpr uint32_t kLinksOffset = 16u * SPAN;
    uint8_t* blocks;
    SideStats* side_base;  // [(S+1)][SPAN]
    int32_t* path;
    // (ChildStats::value_sum of an LDS block holds the child's vterm; ::prior the prior_score when SPAN == 2)
    __device__ __forceinline__ ChildStats* stats(int k) const {
        return reinterpret_cast<ChildStats*>(blocks + (static_cast<uint32_t>(k) << (5 + kShift)));
    }
    __device__ __forceinline__ ChildLinks* links(int k) const {
        return reinterpret_cast<ChildLinks*>(blocks + (static_cast<uint32_t>(k) << (5 + kShift)) + kLinksOffset);
    }
    __device__ __forceinline__ SideStats* side(int k) const { return side_base + (k << kShift); }
    __device__ __forceinline__ void path_store(int level, int packed) const { path[level] = packed; }
    __device__ __forceinline__ int path_load(int level) const { return path[level]; }
};

// normalize() of MinMaxStats (self_play.py:562-566) for the children of one descent.  The quotient
// (v - min) / (max - min) is IEEE division; the compiler's sequence for it is
//     y = rcp(d) refined by two Newton steps;  q0 = n * y;  r = fma(-d, q0, n);  q = fma(r, y, q0)
// wrapped in v_div_scale / v_div_fixup, which only act on operands near the ends of the exponent range.  d is the
// same for every level of a descent, so y is computed once per simulation and a level pays the last three
// operations.  The backups watch every value that can become v, min or max (`exotic`): while all of them are zero
// or within 2^-400 .. 2^400, n and d are zero or within 2^-453 .. 2^401 and n <= d, where scale and fixup are the
// identity (n == 0 gives 0 either way) -- the same bits as the division.  Otherwise the division itself runs.
struct Normalizer {
    double minimum, range, y;
    bool fast;
};

// While max <= min (MinMaxStats.normalize returns the value itself, self_play.py:563) the normalizer is (v - 0) / 1:
// x = v, q0 = v * 1, r = fma(-1, v, v) = 0, q = fma(0, 1, v) = v -- exact, so the descent needs no second case.
__device__ __forceinline__ Normalizer make_normalizer(const MinMax& mm, unsigned long long exotic) {
    Normalizer n;
    const bool has_range = mm.maximum > mm.minimum;
    n.minimum = has_range ? mm.minimum : 0.0;
    n.range = has_range ? mm.maximum - mm.minimum : 1.0;
    n.fast = exotic == 0ull;
    const double y0 = __builtin_amdgcn_rcp(n.range);
    const double e0 = __builtin_fma(-n.range, y0, 1.0);
    const double y1 = __builtin_fma(y0, e0, y0);
    const double e1 = __builtin_fma(-n.range, y1, 1.0);
    n.y = __builtin_fma(y1, e1, y1);
    return n;
}

// (the result is discarded by the caller unless the child was visited)
__device__ __forceinline__ double normalized_value(const Normalizer& n, double v) {
    const double x = v - n.minimum;
    double q;
    if (n.fast) {
        const double q0 = x * n.y;
        const double r = __builtin_fma(-n.range, q0, x);
        q = __builtin_fma(r, n.y, q0);
    } else {
        q = x / n.range;
    }
    // keep the loads and this arithmetic out of a `visits > 0` branch: they must not wait for the visit count
    asm volatile("" : "+v"(q));
    return q;
}

// two values at once, the two chains side by side (one decision between the short form and the division)
__device__ __forceinline__ void normalized_pair(const Normalizer& n, double v0, double v1, double& q0_out, double& q1_out) {
    const double x0 = v0 - n.minimum, x1 = v1 - n.minimum;
    double a, b;
    if (n.fast) {
        const double a0 = x0 * n.y, b0 = x1 * n.y;
        const double ra = __builtin_fma(-n.range, a0, x0), rb = __builtin_fma(-n.range, b0, x1);
        a = __builtin_fma(ra, n.y, a0);
        b = __builtin_fma(rb, n.y, b0);
    } else {
        a = x0 / n.range;
        b = x1 / n.range;
    }
    asm volatile("" : "+v"(a), "+v"(b));
    q0_out = a;
    q1_out = b;
}

// numerator / denominator with the denominator's reciprocal prepared ahead (refined_reciprocal): the last three
// operations of the IEEE division sequence -- the same bits while the numerator is zero or within 2^-400 .. 2^400 and
// the denominator a small positive integer (see Normalizer); the caller checks the numerator (leaves_plain_range)
__device__ __forceinline__ double refined_reciprocal(double d) {
    const double y0 = __builtin_amdgcn_rcp(d);
    const double e0 = __builtin_fma(-d, y0, 1.0);
    const double y1 = __builtin_fma(y0, e0, y0);
    const double e1 = __builtin_fma(-d, y1, 1.0);
    return __builtin_fma(y1, e1, y1);
}
__device__ __forceinline__ double quotient_with(double n, double d, double y) {
    const double q0 = n * y;
    const double r = __builtin_fma(-d, q0, n);
    return __builtin_fma(r, y, q0);
}

// true when a value handed to the min-max statistics leaves the range normalized_value's short form is exact for
__device__ __forceinline__ bool leaves_plain_range(double seen) {
    const uint32_t hi = static_cast<uint32_t>(__double2hiint(seen)) & 0x7fffffffu;
    return (hi - 0x26F00000u) > 0x32000000u && seen != 0.0;  // exponent outside 1023 - 400 .. 1023 + 400
}

// The `while node.expanded()` loop (self_play.py:321-335) with select_child / ucb_score
// (self_play.py:364-405) for A <= SPAN <= 16 children, child c in lane c of the tree's row.
//   prior_score = ((log(..)+init) * (sqrt(N)/(n+1))) * prior      table entry [N][n] (same two IEEE ops)
//   value_score = normalize(vterm)                                  vterm = r + discount*(+-Q), stored by the backup
// Scores, tie lists and RNG draws are those of tree_device.h's descend(), bit for bit.
// Diagnostic build (-DMZ_STAMPS): the descent's inner phases go to slots 8..11 of the caller's stamp
// accumulators; outstanding LDS reads are made to land first so a phase is charged with its own waiting.
#ifdef MZ_STAMPS
#define MZ_DSTAMP_PARAMS , unsigned long long* dstamp_acc, unsigned long long& dstamp_prev
#define MZ_DSTAMP_ARGS , stamp_acc, stamp_prev
#define MZ_DSTAMP(slot)                                                \
    do {                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
        const unsigned long long now__ = __builtin_readcyclecounter(); \
        dstamp_acc[slot] += now__ - dstamp_prev;                       \
        dstamp_prev = now__;                                           \
    } while (0)
#else
#define MZ_DSTAMP_PARAMS
#define MZ_DSTAMP_ARGS
#define MZ_DSTAMP(slot)
#endif

// exploration factor of a child with n visits under a node with N: table modes 2 (rows of 64), 1 (triangular), 0 (none)
template <int MODE>
__device__ __forceinline__ double exploration_factor(const double* pbc, const double* pbc2, int S, int N, int n) {
    if constexpr (MODE == 2) {
        return pbc2[(N << 6) + n];
    } else if constexpr (MODE == 1) {
        return pbc2[(__mul24(N, N + 1) >> 1) + n];
    } else {
        const double pb = pbc[N];
        return pb * (pbc[S + 1 + N] / static_cast<double>(n + 1));
    }
}

template <int SPAN, int MODE>
__device__ __forceinline__ Descent descend_row(const LdsTreeV<SPAN>& acc, const double* pbc, const double* pbc2, int S, int A,
                                               int sim, int n_root_children, const MinMax& mm, unsigned long long exotic,
                                               uint32_t* mt_key, int32_t& mt_pos, uint32_t& words, int j, int group_base,
                                               int32_t* error_flag MZ_DSTAMP_PARAMS) {
    const Normalizer norm = make_normalizer(mm, exotic);
    int n_children = n_root_children;
    int k = 0, N = sim, depth = 0, slot = 0;
    for (;;) {
        const bool valid = j < n_children;
        const int c = valid ? j : 0;
        const ChildStats vp = acc.stats(k)[c];  // { vterm, prior }
        const ChildLinks* lk = acc.links(k) + c;
        const double prior = vp.prior;
        const int visits = lk->visits;
        const int child = lk->child_node;
        const double vt = vp.value_sum;
        MZ_DSTAMP(8);
        const double pb = exploration_factor<MODE>(pbc, pbc2, S, N, visits);
        const double prior_score = pb * prior;
        const double normalized = normalized_value(norm, vt);
        const double value_score = visits > 0 ? normalized : 0.0;
        const double score = valid ? prior_score + value_score : -INFINITY;
        MZ_DSTAMP(9);
        double best = score;
        MZ_BUTTERFLY(SPAN, SPAN, best = fmax(best, partner<M>(best)));

        const bool is_max = valid && (score == best);
        const unsigned long long ballot = __ballot(is_max);
        unsigned int mask = static_cast<unsigned int>(ballot >> group_base) & ((1u << SPAN) - 1u);
        const int n_ties = __popc(mask);
        if (n_ties != 1) {
            if (n_ties > 1) {  // numpy.random.choice over the tie list (self_play.py:372-378)
                int r = 0;
                if (j == 0) r = static_cast<int>(mt_below(mt_key, &mt_pos, static_cast<uint32_t>(n_ties), &words));
                r = row_or(r);
                for (int i = 0; i < r; ++i) mask &= mask - 1u;
            } else {  // NaN scores: the reference would raise; flag and take slot 0
                if (j == 0) atomicOr(error_flag, 1);
                mask = 1u;
            }
        }
        slot = __ffs(static_cast<int>(mask)) - 1;
        MZ_DSTAMP(10);
        const int packed = row_or((j == slot) ? (visits | ((child + 1) << 16)) : 0);
        const int sel_visits = packed & 0xffff;
        const int sel_child = (packed >> 16) - 1;
        if (j == 0) acc.path_store(depth, (k << 16) | slot);
        ++depth;
        if (sel_child < 0) break;
        if (depth > sim) {  // cannot happen on a consistent tree; guarantees every wave leaves the loop
            if (j == 0) atomicOr(error_flag, 2);
            break;
        }
        k = sel_child;
        N = sel_visits;
        n_children = A;
        MZ_DSTAMP(11);
    }
    return Descent{depth, k, slot};
}

// The same loop for exactly two actions (A == 2; the root may have one legal child): a WINDOW of four levels per
// step instead of one level per step.  With one wavefront per SIMD a level of the plain loop costs its whole
// dependent chain (load, normalize, add, compare, select: ~500 cycles); here lane h of the tree's row (h = 1..15,
// heap order: children of h are 2h and 2h+1) looks at the node that sits at heap position h under the window's
// root -- every node keeps a 16-byte table of the block indices of its descendants down to three levels
// (`desc`, maintained when a node is expanded: link_new_node) -- loads that node's two child records, scores both
// and picks the winner by itself: fifteen nodes' chains run side by side.  Two ballots (winner bit, stop bit) then
// tell every lane the path through the window: h1 = 1, h2 = 2 h1 + W[h1], ... until a stop bit (the winner is not
// expanded: the descent ends; or the node could not decide: a tie, which needs the tree's RNG stream, or NaN scores).
// The lanes that turn out to sit on the path store their own path entries; the last node's block index and child
// links reach the row through one OR-reduction.  Scores, tie lists, RNG draws and path entries are those of the
// level-by-level loop, bit for bit: a node's decision depends on its own block and the min-max statistics only.
// A record carries the finished prior_score (see LdsTreeV), so no table look-up sits behind the visit counts.
// Addresses are plain LDS byte addresses: block k of the tree is at (k << 6): records at +0 / +16, links at +32 / +48.
// (a desc entry of 0 = no such descendant: the root is nobody's descendant; such a lane scores the root's block and is
// never on the path)

__device__ __forceinline__ uint32_t row_bits(unsigned long long ballot, bool upper_half, int shift) {
    const uint32_t word = upper_half ? static_cast<uint32_t>(ballot >> 32) : static_cast<uint32_t>(ballot);
    return word >> shift;   // bit h = the ballot bit of lane h of this lane's row (bits >= 16: the next row's, never read)
}

// this lane's entry of the ROOT's descendant table: asked for by the caller as soon as the previous simulation has
// linked its new node, a round trip ahead of the descent that needs it
__device__ __forceinline__ int window_root_entry(const uint8_t* desc, int j) {
    return lds_load<uint8_t>(lds_address(desc) + static_cast<uint32_t>(j > 0 ? j : 1));
}

__device__ __forceinline__ Descent descend_window(const LdsTreeV<2>& acc, const uint8_t* desc, int root_entry, int sim,
                                                  int n_root_children, const MinMax& mm, unsigned long long exotic,
                                                  uint32_t* mt_key, int32_t& mt_pos, uint32_t& words, int j, int group_base,
                                                  int32_t* error_flag) {
    const Normalizer norm = make_normalizer(mm, exotic);
    const int h = j > 0 ? j : 1;   // heap position inside the window (lane 0 doubles lane 1)
    const int my_level = h >= 8 ? 3 : h >= 4 ? 2 : h >= 2 ? 1 : 0;
    const bool upper_half = (group_base & 32) != 0;
    const int row_shift = group_base & 16;
    const uint32_t blocks_address = lds_address(acc.blocks);
    const uint32_t desc_lane = lds_address(desc) + static_cast<uint32_t>(h);
    const bool masked_root = n_root_children < 2;   // the root's second child is not a legal action
    int depth = 0, parent = 0, slot = 0;
    int entry = root_entry;
    bool at_root = true;
    for (;;) {
        // the node this lane looks at, its two child records and their links
        const int node = entry;
        const uint32_t block = blocks_address + (static_cast<uint32_t>(node) << 6);
        const f64x2 rec0 = lds_load<f64x2>(block), rec1 = lds_load<f64x2>(block + 16u);
        const int visits0 = lds_load<int>(block + 36u), child0 = lds_load<int>(block + 40u);
        const int visits1 = lds_load<int>(block + 52u), child1 = lds_load<int>(block + 56u);
        // ucb_score of both children (self_play.py:381-405); select_child over two (self_play.py:364-379)
        double n0, n1;
        normalized_pair(norm, rec0[0], rec1[0], n0, n1);
        const double score0 = rec0[1] + (visits0 > 0 ? n0 : 0.0);
        double score1 = rec1[1] + (visits1 > 0 ? n1 : 0.0);
        if (at_root && masked_root && h == 1) score1 = -INFINITY;
        const bool second = score1 > score0;
        const bool undecided = !second && !(score0 > score1);   // a tie, or NaN scores
        const bool tie = score0 == score1;
        const uint32_t winners = row_bits(__ballot(second), upper_half, row_shift);
        const uint32_t stoppers = row_bits(__ballot(undecided) | __ballot((second ? child1 : child0) < 0), upper_half, row_shift);
        // the path through the window (the same in every lane of the row): h_k = h3 >> (3 - k)
        const uint32_t h1 = 2u | __builtin_amdgcn_ubfe(winners, 1u, 1u);
        const uint32_t h2 = (h1 << 1) | __builtin_amdgcn_ubfe(winners, h1, 1u);
        const uint32_t h3 = (h2 << 1) | __builtin_amdgcn_ubfe(winners, h2, 1u);
        const uint32_t stop_levels = __builtin_amdgcn_ubfe(stoppers, 1u, 1u) | (__builtin_amdgcn_ubfe(stoppers, h1, 1u) << 1) |
                                     (__builtin_amdgcn_ubfe(stoppers, h2, 1u) << 2) | 8u;
        const int last = __builtin_ctz(stop_levels);   // window level of the last node of this step
        const uint32_t h_last = h3 >> (3 - last);
        const bool on_path = static_cast<uint32_t>(h) == (h3 >> (3 - my_level)) && my_level <= last;
        const bool is_last = static_cast<uint32_t>(h) == h_last;
        // every node on the path but an undecided last one knows its own entry
        if (on_path && j > 0 && !(is_last && undecided)) acc.path_store(depth + my_level, (node << 16) | (second ? 1 : 0));
        // the last node's block index, child links and what stopped it: to every lane of the row
        const int packed = row_or(is_last ? (node | ((child0 + 1) << 8) | ((child1 + 1) << 16) | (second ? 1 << 24 : 0) |
                                             (undecided ? 1 << 25 : 0) | (tie ? 1 << 26 : 0))
                                          : 0);
        parent = packed & 0xff;
        slot = (packed >> 24) & 1;
        if (packed & (1 << 25)) {          // (the same for every lane of the row)
            if (packed & (1 << 26)) {      // tie: numpy.random.choice over [0, 1]
                int r = 0;
                if (j == 0) r = static_cast<int>(mt_below(mt_key, &mt_pos, 2u, &words));
                slot = row_or(r);
            } else {                       // NaN scores: the reference would raise; flag and take slot 0
                if (j == 0) atomicOr(error_flag, 1);
                slot = 0;
            }
            if (j == 0) acc.path_store(depth + last, (parent << 16) | slot);
        }
        const int next = ((packed >> (slot ? 16 : 8)) & 0xff) - 1;
        depth += last + 1;
        if (next < 0) break;
        if (depth > sim) {  // cannot happen on a consistent tree; guarantees every wave leaves the loop
            if (j == 0) atomicOr(error_flag, 2);
            break;
        }
        entry = lds_load<uint8_t>(desc_lane + (static_cast<uint32_t>(next) << 4));
        at_root = false;
    }
    return Descent{depth, parent, slot};
}

// The descendant tables after node k_new was expanded as the child the descent ended at: the three nodes above
// it on the path (lane t: path level depth-1-t) enter it at its heap position under them -- a leading one, then the
// child slots from that node down to the new one.  Its own table starts with itself at position 1.
__device__ __forceinline__ int link_fetch_path(const LdsTreeV<2>& acc, int depth, int j) {
    const int level = depth - 1 - j;
    return acc.path_load(j < 3 && level >= 0 ? level : 0);
}
__device__ __forceinline__ void link_new_node(uint8_t* desc, int entry, int depth, int k_new, int j) {
    const bool mine = j < 3 && depth - 1 - j >= 0;
    const int own = mine ? (entry & 1) << j : 0;                       // this level's slot at its place in the position
    const int below = row_shr1_bits(own);
    const int below2 = row_shr1_bits(below);
    const int position = (2 << j) | own | below | below2;
    if (mine) desc[((entry >> 16) << 4) + position] = static_cast<uint8_t>(k_new);
    if (j == 0)
        *reinterpret_cast<uint4*>(desc + (k_new << 4)) =
            uint4{static_cast<uint32_t>(k_new) << 8, 0u, 0u, 0u};
}

// children of a node of the two-action tree: { vterm (unused until visited), prior_score }, the prior aside
__device__ __forceinline__ void write_pair_children(const LdsTreeV<2>& acc, int k, int A, double prior, double factor, int j) {
    if (j < A) {
        acc.stats(k)[j] = ChildStats{0.0, factor * prior};
        acc.side(k)[j] = SideStats{0.0, prior};
        acc.links(k)[j] = ChildLinks{0.f, 0, -1, 0};
    }
}

// backpropagate (self_play.py:407-431) for a tree in LDS, lane = path level (16 levels per round, leaf
// side first).  Every lane fetches its node; the value recursion `value = (+-r) + discount * value` runs
// down the lanes through row_shl:1 (each step recomputes all lanes; a lane is final one step after its
// upper neighbour, and recomputing a final lane reproduces it); then every lane finishes its own node
// (the division for the node's mean runs in parallel over the levels) and stores the child's value term
// for the next descents.  min-max statistics are reduced over the row (max / min are exact under any
// association) into every lane.  Same operations on the same operands as the sequential walk.
// `exotic` collects (per wavefront) whether any value handed to the statistics left normalized_value's plain range.
// Two-action trees (SPAN == 2): every node on the path got one more visit, so the prior_score of BOTH its children
// changes (table row N) -- each lane refreshes the two children of its level's parent block.
//
// Nothing a round reads depends on the value being backed up, and with one wavefront per SIMD a wait for LDS is
// idle time: the fused kernel asks for the leaf-side round's operands in three steps placed between the layers of
// the network (path entry -> records -> exploration factors), each step's answers landing under the next layer.
template <int SPAN>
struct BackupRound {
    int base, cnt;
    bool mine, leaf;
    int packed;                        // path entry of this lane's level
    SideStats own;                     // { value_sum, prior } of the child the path took
    float reward;
    int visits;
    int sibling_visits, above_visits;  // two actions only
    double sibling_prior;
    int visits_new, parent_visits;
    double factor_own, factor_sibling; // table[parent_visits][visits_new / sibling_visits]
    double visits_new_f, visits_new_reciprocal;
};

template <int SPAN>
__device__ __forceinline__ BackupRound<SPAN> backup_fetch_path(const LdsTreeV<SPAN>& acc, int depth, int base, int j) {
    BackupRound<SPAN> r{};
    r.base = base;
    r.cnt = (depth - base < kRow) ? depth - base : kRow;  // levels base .. base + cnt - 1
    r.mine = j < r.cnt;
    r.leaf = base + j == depth - 1;
    r.packed = acc.path_load(r.mine ? base + j : base);
    if constexpr (SPAN == 2) {
        // (lane 0 of a round that is not the last: the node above this round's top level, for its visit count)
        if (base > 0 && j == 0) r.above_visits = acc.path_load(base - 1);
    }
    return r;
}

template <int SPAN>
__device__ __forceinline__ void backup_fetch_records(const LdsTreeV<SPAN>& acc, BackupRound<SPAN>& r, int j) {
    const int slot = r.packed & 0xffff;
    const int kk = r.packed >> 16;
    const ChildLinks* lk = acc.links(kk) + slot;
    r.own = acc.side(kk)[slot];
    r.reward = lk->reward;
    r.visits = lk->visits;
    if constexpr (SPAN == 2) {
        r.sibling_visits = acc.links(kk)[1 - slot].visits;
        r.sibling_prior = acc.side(kk)[1 - slot].prior;
        if (r.base > 0 && j == 0) {
            const int up = r.above_visits;
            r.above_visits = acc.links(up >> 16)[up & 0xffff].visits;
        }
    }
}

template <int SPAN, int MODE>
__device__ __forceinline__ void backup_fetch_factors(BackupRound<SPAN>& r, const double* pbc, const double* pbc2, int S,
                                                     int sim, int j) {
    if (r.leaf) r.visits = 0;   // first visit of the new leaf
    r.visits_new = r.visits + 1;
    r.visits_new_f = static_cast<double>(r.visits_new);
    r.visits_new_reciprocal = refined_reciprocal(r.visits_new_f);
    if constexpr (SPAN == 2) {
        // visit count, after this backup, of the node whose block this lane's level sits in: the level above
        // (the lane to the left), the root for level 0
        int parent_visits = row_shr1_bits(r.visits_new);
        if (j == 0) parent_visits = r.base > 0 ? r.above_visits + 1 : sim + 1;
        r.parent_visits = parent_visits;
        if (r.mine) {
            r.factor_own = exploration_factor<MODE>(pbc, pbc2, S, parent_visits, r.visits_new);
            r.factor_sibling = exploration_factor<MODE>(pbc, pbc2, S, parent_visits, r.sibling_visits);
        }
    }
}

// `first`: the leaf-side round, already fetched (backup_fetch_*); deeper paths fetch their further rounds in place
template <int SPAN, int MODE>
__device__ __forceinline__ void backup_row(const LdsTreeV<SPAN>& acc, const BackupRound<SPAN>& first, int depth, int sim,
                                           double value, float reward_f, bool two_player, double discount, MinMax& mm,
                                           double& root_value_sum, double root_reward, unsigned long long& exotic,
                                           const double* pbc, const double* pbc2, int S, int j) {
    const int k_new = sim + 1;
    // (the root's division, prepared before the value arrives)
    const double root_visits = static_cast<double>(sim + 1);
    const double root_visits_reciprocal = refined_reciprocal(root_visits);
    double carry = value;  // value arriving at the deepest node not yet processed (uniform over the row)
    double seen_max = -INFINITY, seen_min = INFINITY;
    double into_root = 0.0;
    bool odd = false;
    BackupRound<SPAN> r = first;
    for (;;) {
        const int cnt = r.cnt;
        const int level = r.base + j;
        const bool mine = r.mine;
        const bool leaf = r.leaf;
        const int slot = r.packed & 0xffff;
        const int kk = r.packed >> 16;
        ChildStats* st = acc.stats(kk) + slot;
        ChildLinks* lk = acc.links(kk) + slot;
        SideStats* sd = acc.side(kk) + slot;
        double vs = r.own.value_sum;
        float r_f = r.reward;
        if (leaf) {  // first visit of the new leaf: it carries the reward just predicted
            vs = 0.0;
            r_f = reward_f;
        }
        const double rw = static_cast<double>(r_f);
        const bool same = ((depth - (level + 1)) & 1) == 0;  // node.to_play == to_play of the leaf
        const double r_signed = two_player ? (same ? -rw : rw) : rw;
        double val = carry;
        const bool is_top = j >= cnt - 1;
        for (int s = cnt - 1; s > 0; --s) {
            const double passed = row_shl1(r_signed + discount * val);
            val = is_top ? val : passed;
        }
        const double leaving = r_signed + discount * val;  // lane 0: the value arriving one level up
        if (mine) {
            const double vs_new = vs + ((two_player && !same) ? -val : val);
            double q = quotient_with(vs_new, r.visits_new_f, r.visits_new_reciprocal);
            if ((exotic | __ballot(leaves_plain_range(vs_new))) != 0ull) q = vs_new / r.visits_new_f;   // (per wavefront; never seen)
            const double seen = rw + discount * (two_player ? -q : q);
            sd->value_sum = vs_new;
            if (leaf)
                *lk = ChildLinks{reward_f, 1, k_new, 2 * k_new};   // (published trees keep block k at slab k, half 0)
            else
                lk->visits = r.visits_new;
            st->value_sum = seen;  // the vterm slot
            if constexpr (SPAN == 2) {
                st->prior = r.factor_own * r.own.prior;
                acc.stats(kk)[1 - slot].prior = r.factor_sibling * r.sibling_prior;
            }
            seen_max = dmax(seen_max, seen);
            seen_min = dmin(seen_min, seen);
            odd = odd || leaves_plain_range(seen);
        }
        if (r.base == 0) {
            into_root = leaving;
            break;
        }
        carry = __shfl(leaving, 0, kRow);
        r = backup_fetch_path(acc, depth, r.base - kRow, j);
        backup_fetch_records(acc, r, j);
        backup_fetch_factors<SPAN, MODE>(r, pbc, pbc2, S, sim, j);
    }
    // root (tree depth 0)
    if (j == 0) {
        double seen;
        if (!two_player) {
            root_value_sum += into_root;
        } else {
            const bool same = (depth & 1) == 0;
            root_value_sum += same ? into_root : -into_root;
        }
        double mean = quotient_with(root_value_sum, root_visits, root_visits_reciprocal);
        if (exotic != 0ull || leaves_plain_range(root_value_sum)) mean = root_value_sum / root_visits;
        seen = root_reward + discount * (two_player ? -mean : mean);
        seen_max = dmax(seen_max, seen);
        seen_min = dmin(seen_min, seen);
        odd = odd || leaves_plain_range(seen);
    }
    exotic |= __ballot(odd);
    MZ_BUTTERFLY(kRow, kRow, (seen_max = dmax(seen_max, partner<M>(seen_max)), seen_min = dmin(seen_min, partner<M>(seen_min))));
    mm.maximum = dmax(mm.maximum, seen_max);
    mm.minimum = dmin(mm.minimum, seen_min);
}

}  // namespace mz
