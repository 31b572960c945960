// prof.hip — optional hipEvent bracketing of the library's kernels (measurement only).
#include <mutex>
#include <vector>

#include "gg_common.h"

namespace {
struct Pair {
    int id;
    hipEvent_t a, b;
    bool open;
};
std::mutex g_mu;
std::vector<Pair> g_pairs;                 // pairs recorded since the last reset
std::vector<hipEvent_t> g_pool;            // events created once and reused: no hipEventCreate in a
bool g_on = false;                         // measured region after its first step

hipEvent_t take_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void gg_prof_begin(int id, hipStream_t s) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Pair p{id, take_event(), take_event(), true};
    if (!p.a || !p.b) {
        if (p.a) g_pool.push_back(p.a);
        if (p.b) g_pool.push_back(p.b);
        return;
    }
    (void)hipEventRecord(p.a, s);
    g_pairs.push_back(p);
}
void gg_prof_end(int id, hipStream_t s) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_pairs.size(); i-- > 0;)
        if (g_pairs[i].id == id && g_pairs[i].open) {
            (void)hipEventRecord(g_pairs[i].b, s);
            g_pairs[i].open = false;
            return;
        }
}
extern "C" int gg_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    int prev = g_on ? 1 : 0;
    g_on = on != 0;
    return prev;
}
extern "C" int gg_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &p : g_pairs) {              // back to the pool (an event may be re-recorded)
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_pairs.clear();
    return GG_OK;
}
extern "C" int gg_prof_get(int id, int *launches, double *total_ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    double tot = 0.0;
    for (auto &p : g_pairs) {
        if (p.id != id || p.open) continue;
        if (hipEventSynchronize(p.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            tot += ms;
            ++n;
        }
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = tot;
    return GG_OK;
}
extern "C" const char *gg_prof_name(int id) {
    static const char *w[] = {"1", "3", "4", "8", "16", "32"};
    static thread_local char buf[48];
    switch (id) {
        case GG_K_PROJECT_FWD: return "project_fwd_kernel";
        case GG_K_PROJECT_BWD: return "project_bwd_kernel";
        case GG_K_SH_FWD: return "sh_fwd_kernel";
        case GG_K_SH_BWD: return "sh_bwd_kernel";
        case GG_K_BIN_SORT: return "gg_bin_sort(all launches)";
        case GG_K_BLEND_PREP: return "blend_prep_kernel";
        case GG_K_QUAT_FWD: return "quat_to_rotmat_fwd_kernel";
        case GG_K_QUAT_BWD: return "quat_to_rotmat_bwd_kernel";
        case GG_K_MLP_FWD: return "mlp_fwd_kernel";
        case GG_K_MLP_BWD: return "mlp_bwd_kernel";
        case GG_K_BLEND_FWD_PAIR: return "blend_fwd_pair_kernel<40>";
        case GG_K_BLEND_BWD_PAIR: return "blend_bwd_pair_kernel<40>";
        case GG_K_COMPACT: return "compact_rows_kernel";
        case GG_K_DENSIFY: return "densify_rows_kernel";
        case GG_K_ADAM: return "adam_kernel";
        case GG_K_VIEW_BWD: return "view_bwd_kernel";
        case GG_K_VIEW_FWD: return "view_fwd_kernel";
        case GG_K_ACTIVATE_FWD: return "activate_fwd_kernel";
        case GG_K_ACTIVATE_BWD: return "activate_bwd_kernel";
        case GG_K_COUNT: return "count_kernel";
        case GG_K_TAIL_SPLIT: return "tail_split_kernel";
        default: break;
    }
    if (id >= GG_K_BLEND_FWD && id < GG_K_BLEND_FWD + 6) {
        snprintf(buf, sizeof(buf), "blend_fwd_kernel<%s>", w[id - GG_K_BLEND_FWD]);
        return buf;
    }
    if (id >= GG_K_BLEND_BWD && id < GG_K_BLEND_BWD + 6) {
        snprintf(buf, sizeof(buf), "blend_bwd_kernel<%s>", w[id - GG_K_BLEND_BWD]);
        return buf;
    }
    return "";
}
