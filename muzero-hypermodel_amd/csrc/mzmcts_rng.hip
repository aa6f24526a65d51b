// mzmcts_rng.hip -- stand-alone numpy-compatible host streams (include/mzmcts.h mzmcts_rng_*): the legacy
// RandomState pieces the path uses (np_legacy_rng.h) behind the C ABI, for host code that samples in the reference's
// order (replay-buffer sampling, opponents, SelfPlay.select_action on a given stream).
#include "engine_host.h"

namespace {

// glibc's log / pow on the device (glibc_libm.h): out_log[i] = log(x[i]), out_pow[i] = pow(x[i], y[i])
__global__ void device_libm_kernel(const double* __restrict__ x, const double* __restrict__ y, int64_t n,
                                   double* __restrict__ out_log, double* __restrict__ out_pow) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_log[i] = mz::libm::glibc_log(x[i]);
    out_pow[i] = mz::libm::glibc_pow(x[i], y[i]);
}

// stream s = numpy.random.seed(seeds[s]); then `draws` x numpy.random.dirichlet([alpha] * k), all on the device
__global__ void device_dirichlet_kernel(const uint32_t* __restrict__ seeds, int n_streams, double alpha, int k, int draws,
                                        uint32_t* __restrict__ keys, double* __restrict__ out,
                                        uint32_t* __restrict__ words) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    uint32_t* key = keys + static_cast<size_t>(s) * mz::kMtN;
    int32_t pos;
    mz::mt_seed(key, &pos, seeds[s]);
    mz::DeviceStream stream{key, pos, 0u, nullptr, 0, 0};
    for (int d = 0; d < draws; ++d) stream.dirichlet(alpha, k, out + (static_cast<size_t>(s) * draws + d) * k);
    words[s] = stream.words;
}

}  // namespace

extern "C" {

#define MZ_RNG_HIP(call)                         \
    do {                                         \
        if ((call) != hipSuccess) return MZMCTS_ERR_HIP; \
    } while (0)

int mzmcts_device_libm(const double* x, const double* y, int64_t n, double* log_out, double* pow_out) {
    if (!x || !y || !log_out || !pow_out || n <= 0) return MZMCTS_ERR_INVALID;
    double *d_x = nullptr, *d_y = nullptr, *d_l = nullptr, *d_p = nullptr;
    const size_t bytes = sizeof(double) * static_cast<size_t>(n);
    MZ_RNG_HIP(hipMalloc(&d_x, bytes));
    MZ_RNG_HIP(hipMalloc(&d_y, bytes));
    MZ_RNG_HIP(hipMalloc(&d_l, bytes));
    MZ_RNG_HIP(hipMalloc(&d_p, bytes));
    MZ_RNG_HIP(hipMemcpy(d_x, x, bytes, hipMemcpyHostToDevice));
    MZ_RNG_HIP(hipMemcpy(d_y, y, bytes, hipMemcpyHostToDevice));
    device_libm_kernel<<<dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256)>>>(d_x, d_y, n, d_l, d_p);
    MZ_RNG_HIP(hipGetLastError());
    MZ_RNG_HIP(hipMemcpy(log_out, d_l, bytes, hipMemcpyDeviceToHost));
    MZ_RNG_HIP(hipMemcpy(pow_out, d_p, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(d_x);
    (void)hipFree(d_y);
    (void)hipFree(d_l);
    (void)hipFree(d_p);
    return MZMCTS_OK;
}

int mzmcts_device_dirichlet(const uint32_t* seeds, int32_t n_streams, double alpha, int32_t k, int32_t draws, double* out,
                            uint32_t* words_out) {
    if (!seeds || !out || !words_out || n_streams <= 0 || k <= 0 || draws <= 0 || !(alpha > 0.0) || alpha > 1.0)
        return MZMCTS_ERR_INVALID;
    uint32_t *d_seeds = nullptr, *d_keys = nullptr, *d_words = nullptr;
    double* d_out = nullptr;
    const size_t out_bytes = sizeof(double) * static_cast<size_t>(n_streams) * draws * k;
    MZ_RNG_HIP(hipMalloc(&d_seeds, sizeof(uint32_t) * n_streams));
    MZ_RNG_HIP(hipMalloc(&d_keys, sizeof(uint32_t) * static_cast<size_t>(n_streams) * mz::kMtN));
    MZ_RNG_HIP(hipMalloc(&d_words, sizeof(uint32_t) * n_streams));
    MZ_RNG_HIP(hipMalloc(&d_out, out_bytes));
    MZ_RNG_HIP(hipMemcpy(d_seeds, seeds, sizeof(uint32_t) * n_streams, hipMemcpyHostToDevice));
    device_dirichlet_kernel<<<dim3((n_streams + 63) / 64), dim3(64)>>>(d_seeds, n_streams, alpha, k, draws, d_keys, d_out,
                                                                       d_words);
    MZ_RNG_HIP(hipGetLastError());
    MZ_RNG_HIP(hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost));
    MZ_RNG_HIP(hipMemcpy(words_out, d_words, sizeof(uint32_t) * n_streams, hipMemcpyDeviceToHost));
    (void)hipFree(d_seeds);
    (void)hipFree(d_keys);
    (void)hipFree(d_words);
    (void)hipFree(d_out);
    return MZMCTS_OK;
}

// ---- stand-alone host streams ----------------------------------------------------------------------
struct mzmcts_rng {
    mz::HostStream s;
};

mzmcts_rng* mzmcts_rng_create(uint32_t seed) {
    auto* r = new mzmcts_rng();
    r->s.seed(seed);
    return r;
}
void mzmcts_rng_destroy(mzmcts_rng* r) { delete r; }
void mzmcts_rng_reseed(mzmcts_rng* r, uint32_t seed) { r->s.seed(seed); }
uint32_t mzmcts_rng_next_u32(mzmcts_rng* r) { return r->s.u32(); }
double mzmcts_rng_random_sample(mzmcts_rng* r) { return r->s.uniform(); }
uint32_t mzmcts_rng_choice(mzmcts_rng* r, uint32_t n) { return r->s.below(n); }
int32_t mzmcts_rng_choice_p(mzmcts_rng* r, const double* p, int32_t n) { return r->s.choice_p(p, n); }
void mzmcts_rng_choice_p_many(mzmcts_rng* r, const double* p, int32_t n, int32_t count, int32_t* out) {
    for (int32_t i = 0; i < count; ++i) out[i] = r->s.choice_p(p, n);
}
int32_t mzmcts_rng_choice_priorities(mzmcts_rng* r, const float* priorities, int32_t n, float* prob_out) {
    // position_probs = priorities / sum(priorities): a left-to-right float32 sum (Python's sum over float32
    // scalars), a float32 division per entry, then RandomState.choice(n, p=position_probs)
    float total = 0.f;
    for (int32_t i = 0; i < n; ++i) total = total + priorities[i];
    std::vector<double> p(static_cast<size_t>(n));
    for (int32_t i = 0; i < n; ++i) p[i] = static_cast<double>(priorities[i] / total);
    const int32_t idx = r->s.choice_p(p.data(), n);
    if (prob_out && idx >= 0 && idx < n) *prob_out = priorities[idx] / total;
    return idx;
}
void mzmcts_rng_dirichlet(mzmcts_rng* r, double alpha, int32_t k, double* out) { r->s.dirichlet(alpha, k, out); }
void mzmcts_rng_export(const mzmcts_rng* r, uint32_t* key, int32_t* pos, int32_t* has_gauss, double* cached) {
    std::memcpy(key, r->s.key, sizeof(r->s.key));
    *pos = r->s.pos;
    *has_gauss = r->s.has_gauss;
    *cached = r->s.gauss;
}
void mzmcts_rng_import(mzmcts_rng* r, const uint32_t* key, int32_t pos, int32_t has_gauss, double cached) {
    std::memcpy(r->s.key, key, sizeof(r->s.key));
    r->s.pos = pos;
    r->s.has_gauss = has_gauss;
    r->s.gauss = cached;
}
int32_t mzmcts_rng_select_action(mzmcts_rng* r, const int32_t* visits, int32_t n, double temperature) {
    return r->s.select_action(visits, n, temperature);
}

}  // extern "C"
