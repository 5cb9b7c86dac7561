// tree_static.cpp -- O1/O2: static (EAGLE-1 / LANTERN++) tree buffers, host side.
//
// Built once per tree shape and cached by the caller, so this is plain C++ (no GPU): a trie
// keyed by choice paths replaces the reference's O(N^2) list.index() scans.
//
// Reference: models/ea_model_lumina_mgpt.py:140-277 (target side: mask, tree_indices,
// position ids, retrieve_indices, p_indices, b_indices); models/drafters/utils_c.py:35-179
// (drafter side: per-depth masks / tree_indices / repeat_nums over non-leaf nodes).
#include <algorithm>
#include <cstdarg>
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
