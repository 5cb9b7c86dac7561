// tree_static.cpp -- O1/O2: static (EAGLE-1 / LANTERN++) tree buffers, host side.
//
// Built once per tree shape and cached by the caller, so this is plain C++ (no GPU): a trie
// keyed by choice paths replaces the reference's O(N^2) list.index() scans.
//
// Reference: models/ea_model_lumina_mgpt.py:140-277 (target side: mask, tree_indices,
// position ids, retrieve_indices, p_indices, b_indices); models/drafters/utils_c.py:35-179
// (drafter side: per-depth masks / tree_indices / repeat_nums over non-leaf nodes).
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/lantern_hip.h"

namespace lantern {
void set_error(const char *fmt, ...);

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

static thread_local void *g_ev_start = nullptr, *g_ev_stop = nullptr;
void take_launch_events(void **start, void **stop) {
    *start = g_ev_start;
    *stop = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
}
void arm_launch_events(void *start, void *stop) {
    g_ev_start = start;
    g_ev_stop = stop;
}

// ---------------------------------------------------------------------------------------------- tuning values
// (common.h `enum Tuning` names the same slots; this file does not include the HIP headers)
namespace {
struct TuningSlot { const char *name; int dflt; };
const TuningSlot kTuning[] = {
    {"epw_tp", 5}, {"epw_tp4", 1}, {"epw_tp_raw", 256}, {"epw_spec", 2}, {"epw_occ2", -1}, {"o7_nt", 0}, {"prep_nt", 0}, {"kv_u", 0}, {"kv_ks", 4},
    {"kv_variant", 0}, {"gemm_tiled_from", 129}, {"sk_groups", 0}, {"sk_whole_mb", 40}, {"sk_nt_min_mb", 80}, {"ta_splits", 0}, {"ta_min_tiles", 2}, {"epw_tp_lg", 1}, {"epw_fused_helpers", 1}};
constexpr int kTuningCount = sizeof(kTuning) / sizeof(kTuning[0]);
std::atomic<int> g_tuning[kTuningCount];
std::atomic<bool> g_tuning_init{false};
void tuning_init() {
    if (g_tuning_init.load(std::memory_order_acquire)) return;
    static std::atomic_flag once = ATOMIC_FLAG_INIT;
    if (!once.test_and_set()) {
        for (int i = 0; i < kTuningCount; ++i) g_tuning[i].store(kTuning[i].dflt, std::memory_order_relaxed);
        g_tuning_init.store(true, std::memory_order_release);
    } else {
        while (!g_tuning_init.load(std::memory_order_acquire)) {}
    }
}
int tuning_index(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < kTuningCount; ++i)
        if (!strcmp(name, kTuning[i].name)) return i;
    return -1;
}
}  // namespace
int tuning(int t) {
    tuning_init();
    return (t >= 0 && t < kTuningCount) ? g_tuning[t].load(std::memory_order_relaxed) : 0;
}
int tuning_count() { return kTuningCount; }

namespace {

using Path = std::vector<int32_t>;

struct Tree {
    std::vector<Path> nodes;           // sorted by (len, lexicographic); node id = index + 1
    std::map<Path, int> id;            // path -> node id (root = 0 for the empty path)
    std::vector<int> parent;           // by node id
    std::vector<int> depth;            // by node id
    std::vector<std::vector<int>> kids;  // by node id, in sorted order
    int max_depth = 0;
    bool ok = true;

    Tree(const int32_t *choices, const int32_t *off, int n) {
        nodes.reserve(n);
        for (int i = 0; i < n; ++i) nodes.emplace_back(choices + off[i], choices + off[i + 1]);
        std::stable_sort(nodes.begin(), nodes.end(), [](const Path &a, const Path &b) {
            if (a.size() != b.size()) return a.size() < b.size();
            return a < b;
        });
        id[Path()] = 0;
        for (int i = 0; i < n; ++i) id[nodes[i]] = i + 1;
        parent.assign(n + 1, 0);
        depth.assign(n + 1, 0);
        kids.assign(n + 1, {});
        for (int i = 0; i < n; ++i) {
            Path p(nodes[i].begin(), nodes[i].end() - 1);
            auto it = id.find(p);
            if (nodes[i].empty() || it == id.end()) {
                ok = false;  // the reference raises ValueError from list.index
                continue;
            }
            parent[i + 1] = it->second;
            depth[i + 1] = (int)nodes[i].size();
            kids[it->second].push_back(i + 1);
            max_depth = std::max(max_depth, depth[i + 1]);
        }
    }
    int n() const { return (int)nodes.size(); }
    bool is_leaf(int node) const { return kids[node].empty(); }

    // leaf rows: root..leaf node ids, -1 padded to D, sorted with -1 ranked above every id
    std::vector<std::vector<int64_t>> retrieve_rows() const {
        const int D = max_depth + 1;
        std::vector<std::vector<int64_t>> rows;
        for (int node = 1; node <= n(); ++node) {
            if (!is_leaf(node)) continue;
            std::vector<int64_t> r(D, -1);
            for (int cur = node; cur > 0; cur = parent[cur]) r[depth[cur]] = cur;
            r[0] = 0;
            rows.push_back(r);
        }
        const int64_t big = (int64_t)n() + 5;
        std::stable_sort(rows.begin(), rows.end(), [big](const std::vector<int64_t> &a, const std::vector<int64_t> &b) {
            for (size_t i = 0; i < a.size(); ++i) {
                const int64_t x = a[i] < 0 ? big : a[i], y = b[i] < 0 ? big : b[i];
                if (x != y) return x < y;
            }
            return false;
        });
        return rows;
    }
    // earlier siblings (same parent, smaller id)
    std::vector<int> earlier_siblings(int node) const {
        std::vector<int> out;
        for (int k : kids[parent[node]])
            if (k < node) out.push_back(k);
        return out;
    }
};

}  // namespace
}  // namespace lantern

using namespace lantern;

extern "C" int lantern_version(void) { return LANTERN_VERSION; }
extern "C" const char *lantern_last_error(void) { return lantern::last_error(); }
namespace lantern {
void arm_launch_events(void *start, void *stop);
}
extern "C" int lantern_profile_next_launch(void *start_event, void *stop_event) {
    lantern::arm_launch_events(start_event, stop_event);
    return LANTERN_OK;
}

extern "C" int lantern_tree_static_sizes(const int32_t *choices, const int32_t *choice_off, int n_choices, int *N, int *P,
                                         int *D, int *b_total) {
    if (!choices || !choice_off || n_choices <= 0 || !N || !P || !D || !b_total) {
        set_error("tree_static_sizes: bad arguments");
        return LANTERN_E_INVALID;
    }
    Tree t(choices, choice_off, n_choices);
    if (!t.ok) {
        set_error("tree_static_sizes: a choice's parent path is missing from tree_choices");
        return LANTERN_E_INVALID;
    }
    auto rows = t.retrieve_rows();
    int bt = 0;
    for (auto &r : rows)
        for (int64_t v : r)
            if (v > 0) bt += (int)t.earlier_siblings((int)v).size();
    *N = t.n() + 1;
    *P = (int)rows.size();
    *D = t.max_depth + 1;
    *b_total = bt;
    return LANTERN_OK;
}

extern "C" int lantern_tree_static_build(const int32_t *choices, const int32_t *choice_off, int n_choices, int top_k, float *mask,
                                         int64_t *tree_indices, int64_t *pos_ids, int64_t *retrieve, int32_t *p_idx,
                                         int32_t *b_off, int32_t *b_idx) {
    if (!choices || !choice_off || n_choices <= 0 || !mask || !tree_indices || !pos_ids || !retrieve || !p_idx || !b_off || !b_idx) {
        set_error("tree_static_build: bad arguments");
        return LANTERN_E_INVALID;
    }
    Tree t(choices, choice_off, n_choices);
    if (!t.ok) {
        set_error("tree_static_build: a choice's parent path is missing from tree_choices");
        return LANTERN_E_INVALID;
    }
    const int N = t.n() + 1, D = t.max_depth + 1;
    std::fill(mask, mask + (size_t)N * N, 0.0f);
    for (int node = 0; node < N; ++node) {
        mask[(size_t)node * N] = 1.0f;
        for (int cur = node; cur > 0; cur = t.parent[cur]) mask[(size_t)node * N + cur] = 1.0f;
        mask[(size_t)node * N + node] = 1.0f;
        pos_ids[node] = t.depth[node];
    }
    // tree_indices[node] = slot + top_k * (drafter row) + 1 where the drafter row of a sibling
    // group is its rank among all parents in sorted order (root's children = row 0); the
    // in-layer ordinal of that parent is p_node.
    std::vector<int> p_node(N, -1);
    tree_indices[0] = 0;
    int row = -1, prev_parent = -1, prev_depth = 0, inlayer = 0;
    for (int node = 1; node < N; ++node) {
        if (t.depth[node] != prev_depth) {
            inlayer = 0;
            ++row;
        } else if (t.parent[node] != prev_parent) {
            ++inlayer;
            ++row;
        }
        prev_depth = t.depth[node];
        prev_parent = t.parent[node];
        tree_indices[node] = t.nodes[node - 1].back() + (int64_t)top_k * row + 1;
        p_node[node] = inlayer;
    }
    auto rows = t.retrieve_rows();
    const int P = (int)rows.size();
    int bt = 0;
    for (int r = 0; r < P; ++r)
        for (int c = 0; c < D; ++c) {
            const int64_t v = rows[r][c];
            retrieve[(size_t)r * D + c] = v;
            p_idx[(size_t)r * D + c] = p_node[v >= 0 ? v : N - 1];  // torch index -1 wraps to the last node
            b_off[(size_t)r * D + c] = bt;
            if (v > 0)
                for (int s : t.earlier_siblings((int)v)) b_idx[bt++] = s;
        }
    b_off[(size_t)P * D] = bt;
    return LANTERN_OK;
}

extern "C" int lantern_tree_drafter_sizes(const int32_t *choices, const int32_t *choice_off, int n_choices, int *n_levels,
                                          int *level_counts) {
    if (!choices || !choice_off || n_choices <= 0 || !n_levels || !level_counts) {
        set_error("tree_drafter_sizes: bad arguments");
        return LANTERN_E_INVALID;
    }
    Tree t(choices, choice_off, n_choices);
    if (!t.ok) {
        set_error("tree_drafter_sizes: malformed tree_choices");
        return LANTERN_E_INVALID;
    }
    const int L = t.max_depth - 1;
    for (int l = 0; l < L; ++l) level_counts[l] = 0;
    for (int node = 1; node <= t.n(); ++node)
        if (!t.is_leaf(node)) level_counts[t.depth[node] - 1] += 1;
    *n_levels = L;
    return LANTERN_OK;
}

extern "C" int lantern_tree_drafter_build(const int32_t *choices, const int32_t *choice_off, int n_choices, int top_k,
                                          float *masks_concat, int64_t *tree_indices_concat, int32_t *repeat_nums_concat,
                                          int32_t *repeat_off) {
    if (!choices || !choice_off || n_choices <= 0 || !masks_concat || !tree_indices_concat || !repeat_nums_concat || !repeat_off) {
        set_error("tree_drafter_build: bad arguments");
        return LANTERN_E_INVALID;
    }
    Tree t(choices, choice_off, n_choices);
    if (!t.ok) {
        set_error("tree_drafter_build: malformed tree_choices");
        return LANTERN_E_INVALID;
    }
    // non-leaf nodes in sorted order get consecutive indices
    std::vector<int> wc;            // node ids
    std::vector<int> wc_index(t.n() + 1, -1);
    for (int node = 1; node <= t.n(); ++node)
        if (!t.is_leaf(node)) {
            wc_index[node] = (int)wc.size();
            wc.push_back(node);
        }
    const int L = t.max_depth - 1;
    size_t mo = 0, to = 0;
    int rp = 0, start = 0, cum = 0;
    for (int l = 0; l < L; ++l) {
        int cnt = 0;
        while (start + cnt < (int)wc.size() && t.depth[wc[start + cnt]] == l + 1) ++cnt;
        cum += cnt;
        for (int a = 0; a < cnt; ++a) {
            const int node = wc[start + a];
            for (int c = 0; c < cum; ++c) masks_concat[mo + c] = 0.0f;
            for (int cur = node; cur > 0; cur = t.parent[cur]) masks_concat[mo + wc_index[cur]] = 1.0f;
            mo += cum;
        }
        repeat_off[l] = rp;
        int bias = 0, run = 0;
        for (int j = 0; j < cnt; ++j) {
            const int node = wc[start + j];
            if (j != 0 && t.parent[node] != t.parent[wc[start + j - 1]]) {
                ++bias;
                repeat_nums_concat[rp++] = run;
                run = 0;
            }
            ++run;
            tree_indices_concat[to++] = t.nodes[node - 1].back() + (int64_t)top_k * bias;
        }
        repeat_nums_concat[rp++] = run;
        start += cnt;
    }
    repeat_off[L] = rp;
    return LANTERN_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Node view of a verify tree, for the node-parallel evaluate_posterior (node_kernels.hip).
//
// The reference walks the tree level by level over PATHS (ea_model_lumina_mgpt.py:628-713): at level i it keeps the paths whose
// first i tokens equal the accepted prefix, and tries their distinct tokens at depth i in path order.  While sibling tokens
// are distinct (sampling without replacement / top-k: the only trees the drafters build) "paths sharing the accepted prefix" =
// "paths through the accepted node", so the candidates of a level are the CHILDREN of the accepted node, tried in the order of
// their first path.  Everything a child's test consumes -- the parent's row, the drafter row, the earlier siblings, the
// cart_candidates_prob cell and even the position of its uniform in the random.random() stream (one draw per tried
// candidate: root..node) -- is then a function of the node alone, which is what lets every node's chain run in its own
// workgroup.  Layout of `out` (int32):
//   hdr[8]        = {N, n_internal, n_children, max_children, D, P, prefix_siblings, 0}
//   internal[n_internal][16] = {node, c0, nch, depth, uoff, qrow, first_path, 0, node of child 0..3, cell of child 0..3}, longest child lists first
//   child[n_children][4]    = {node, cell (= path*D + depth of its first cell), b0 (into the caller's b_idx), nsib}, per parent in try order
//   node[N][4]              = {first_path, depth, internal rank or -1, parent}
//   order[N]                = node ids in launch order: the internal nodes (longest child lists first), then the leaves
extern "C" int lantern_tree_node_tables_size(int N, int P, int D) {
    (void)P;
    (void)D;
    return 8 + 16 * N + 4 * N + 4 * N + N;
}

extern "C" int lantern_tree_node_tables(const int64_t *retrieve, const int32_t *p_idx, const int32_t *b_off, const int32_t *b_idx,
                                        const int32_t *op_off, int N, int P, int D, int32_t *out, int out_ints) {
    if (!retrieve || !out || N <= 0 || P <= 0 || D <= 0 || out_ints < lantern_tree_node_tables_size(N, P, D)) {
        set_error("tree_node_tables: bad arguments");
        return LANTERN_E_INVALID;
    }
    const bool is_static = p_idx && b_off && op_off;
    std::vector<int> parent(N, -1), depth(N, -1), cell(N, -1);
    std::vector<std::vector<int>> kids(N);
    for (int j = 0; j < P; ++j) {
        if (retrieve[(size_t)j * D] != 0) {
            set_error("tree_node_tables: path %d does not start at the root", j);
            return LANTERN_E_INVALID;
        }
        for (int i = 1; i < D; ++i) {
            const int64_t c = retrieve[(size_t)j * D + i];
            if (c < 0) break;                 // -1 pad: the path ended
            const int64_t p = retrieve[(size_t)j * D + i - 1];
            if (c == 0 || c >= N) {
                set_error("tree_node_tables: retrieve[%d,%d]=%lld outside (0,%d)", j, i, (long long)c, N);
                return LANTERN_E_INVALID;
            }
            if (depth[c] < 0) {
                depth[c] = i;
                parent[c] = (int)p;
                cell[c] = j * D + i;
                kids[p].push_back((int)c);
            } else if (depth[c] != i || parent[c] != (int)p) {
                set_error("tree_node_tables: node %lld has two parents / depths: not a tree", (long long)c);
                return LANTERN_E_INVALID;
            }
        }
    }
    depth[0] = 0;
    cell[0] = 0;
    std::vector<int> uoff(N, 0), internal;
    // uniforms consumed before a node's own level: one per candidate tried on the way (children in try order, all distinct)
    std::vector<int> order;
    order.push_back(0);
    for (size_t h = 0; h < order.size(); ++h) {
        const int n = order[h];
        for (size_t t = 0; t < kids[n].size(); ++t) {
            uoff[kids[n][t]] = uoff[n] + (int)t + 1;
            order.push_back(kids[n][t]);
        }
        if (!kids[n].empty()) internal.push_back(n);
    }
    std::stable_sort(internal.begin(), internal.end(), [&](int a, int b) { return kids[a].size() > kids[b].size(); });
    std::vector<int> rank(N, -1);
    for (size_t r = 0; r < internal.size(); ++r) rank[internal[r]] = (int)r;
    const int n_int = (int)internal.size();
    int32_t *hdr = out, *it = out + 8, *ch = it + 16 * n_int;
    bool prefix_sibs = true;
    int n_child = 0, max_ch = 0;
    for (int r = 0; r < n_int; ++r) {
        const int n = internal[r];
        const int nch = (int)kids[n].size();
        int qrow = 0;
        if (is_static) {
            qrow = op_off[depth[n]] + p_idx[cell[kids[n][0]]];
            for (int c : kids[n])
                if (op_off[depth[n]] + p_idx[cell[c]] != qrow) {
                    set_error("tree_node_tables: children of node %d disagree on their drafter row", n);
                    return LANTERN_E_INVALID;
                }
        }
        int32_t *e = it + 16 * r;
        e[0] = n; e[1] = n_child; e[2] = nch; e[3] = depth[n]; e[4] = uoff[n]; e[5] = qrow; e[6] = cell[n] / D; e[7] = 0;
        for (int t = 0; t < 4; ++t) {
            e[8 + t] = t < nch ? kids[n][t] : 0;
            e[12 + t] = t < nch ? cell[kids[n][t]] : 0;
        }
        int rank_in_parent = 0;
        for (int c : kids[n]) {
            int32_t *q = ch + 4 * n_child++;
            q[0] = c; q[1] = cell[c];
            q[2] = is_static ? b_off[cell[c]] : 0;
            q[3] = is_static ? b_off[cell[c] + 1] - b_off[cell[c]] : 0;
            // the node kernel takes "earlier siblings" = "the children tried before this one"
            if (is_static && q[3] != rank_in_parent) prefix_sibs = false;
            if (is_static && b_idx)
                for (int u = 0; u < q[3] && u < rank_in_parent; ++u)
                    if (b_idx[q[2] + u] != kids[n][u]) prefix_sibs = false;
            ++rank_in_parent;
        }
        max_ch = std::max(max_ch, nch);
    }
    int32_t *nd = ch + 4 * n_child;
    for (int n = 0; n < N; ++n) {
        nd[4 * n] = cell[n] >= 0 ? cell[n] / D : -1;
        nd[4 * n + 1] = depth[n];
        nd[4 * n + 2] = rank[n];
        nd[4 * n + 3] = parent[n];
    }
    int32_t *ord = nd + 4 * N;
    int no = 0;
    for (int r = 0; r < n_int; ++r) ord[no++] = internal[r];
    for (int n = 0; n < N; ++n)
        if (rank[n] < 0) ord[no++] = n;
    hdr[0] = N; hdr[1] = n_int; hdr[2] = n_child; hdr[3] = max_ch; hdr[4] = D; hdr[5] = P; hdr[6] = prefix_sibs ? 1 : 0; hdr[7] = 0;
    return LANTERN_OK;
}

// ---------------------------------------------------------------------------------------------- lantern_tuning_set / get / name
extern "C" int lantern_tuning_set(const char *name, int value) {
    const int i = lantern::tuning_index(name);
    if (i < 0) {
        lantern::set_error("tuning_set: unknown name '%s'", name ? name : "(null)");
        return LANTERN_E_INVALID;
    }
    lantern::tuning_init();
    lantern::g_tuning[i].store(value, std::memory_order_relaxed);
    return LANTERN_OK;
}
extern "C" int lantern_tuning_get(const char *name, int *value) {
    const int i = lantern::tuning_index(name);
    if (i < 0 || !value) {
        lantern::set_error("tuning_get: unknown name '%s'", name ? name : "(null)");
        return LANTERN_E_INVALID;
    }
    *value = lantern::tuning(i);
    return LANTERN_OK;
}
extern "C" const char *lantern_tuning_name(int index) { return (index >= 0 && index < lantern::kTuningCount) ? lantern::kTuning[index].name : nullptr; }
extern "C" int lantern_tuning_reset(void) {
    lantern::tuning_init();
    for (int i = 0; i < lantern::kTuningCount; ++i) lantern::g_tuning[i].store(lantern::kTuning[i].dflt, std::memory_order_relaxed);
    return LANTERN_OK;
}
