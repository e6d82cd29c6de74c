// MEASUREMENT ONLY: per-thread TSC timers around functions of the SAM phase, compiled in with -DMEME_DROPIN_PROF (oracle/_ref/bwa-meme_dropin_prof).
// In the product build PROF_SCOPE() is nothing.  The counters, the interposed timers of the reference's own functions and the report live in
// meme_dropin_prof.cpp.
#ifndef MEME_DROPIN_PROF_H
#define MEME_DROPIN_PROF_H
#ifdef MEME_DROPIN_PROF
#include <stdint.h>
#include <x86intrin.h>
namespace dropin_prof {
enum { P_POST, P_PRE, P_MATESW_POST, P_MATESW_POST_MS, P_MARK_PRIMARY, P_PAIR, P_GEN_ALT, P_REG2ALN, P_REG2SAM, P_SORT_DEDUP, P_SORT_DEDUP_MS, P_APPROX_MAPQ,
       P_GEN_CIGAR2_HOOK, P_ALN2SAM, P_N };
void flush(const uint64_t* t, const uint64_t* n);
struct Local {
    uint64_t t[P_N] = {}, n[P_N] = {};
    ~Local() { flush(t, n); }
};
inline Local& local() { static thread_local Local l; return l; }
struct Scope { int id; uint64_t t0; explicit Scope(int i) : id(i), t0(__rdtsc()) {} ~Scope() { Local& l = local(); l.t[id] += __rdtsc() - t0; ++l.n[id]; } };
}  // namespace dropin_prof
#define PROF_SCOPE(id_) dropin_prof::Scope prof_scope_(dropin_prof::id_)
#else
#define PROF_SCOPE(id_) do { } while (0)
#endif
#endif
