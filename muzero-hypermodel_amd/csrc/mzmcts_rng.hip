// mzmcts_rng.hip -- stand-alone numpy-compatible host streams (include/mzmcts.h mzmcts_rng_*): the legacy
// RandomState pieces the path uses (np_legacy_rng.h) behind the C ABI, for host code that samples in the reference's
// order (replay-buffer sampling, opponents, SelfPlay.select_action on a given stream).
#include "engine_host.h"

extern "C" {

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
