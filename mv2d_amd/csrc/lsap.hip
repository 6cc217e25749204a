// Host side of the training targets: the linear sum assignment of HungarianAssigner3D
// (mmdet3d_plugin/core/bbox/assigners/hungarian_assigner_3d.py:137: scipy.optimize.linear_sum_assignment on the [queries, boxes] cost of a
// layer).  The reference calls SciPy once per decoder layer from Python; SciPy holds the interpreter lock while it solves, so the six
// layers of a step run one after the other (0.75 ms per step for 300 x 40).  Here: the same algorithm -- the shortest augmenting path
// method for rectangular problems of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE Trans. Aerospace and
// Electronic Systems 52(4), 2016, which SciPy (>= 1.4) implements -- restated in C++, one thread per layer.  The scan orders and the tie
// rule (columns visited from the last to the first, an unassigned column wins a tie) are the published ones, so that equal-cost
// assignments are resolved the same way; tests/test_lsap_cpu.py compares with SciPy on random, integer (tie-rich) and degenerate costs.
// No device code in this file: plain host C++ behind the C-ABI.
#include "common.h"

#include <cmath>
#include <thread>
#include <vector>

namespace {

// shortest augmenting path from row i; returns the sink column or -1 (infeasible)
static int augmenting_path(int nc, const double* cost, const std::vector<double>& u, const std::vector<double>& v, std::vector<int>& path,
                           const std::vector<int>& row4col, std::vector<double>& spc, int i, std::vector<char>& SR, std::vector<char>& SC,
                           std::vector<int>& remaining, double* p_min) {
    double min_val = 0.0;
    int num_remaining = nc;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;           // reverse order: a constant matrix is solved by the identity
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(spc.begin(), spc.end(), INFINITY);
    int sink = -1;
    while (sink == -1) {
        int index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int it = 0; it < num_remaining; ++it) {
            const int j = remaining[it];
            const double r = min_val + cost[(long long)i * nc + j] - u[i] - v[j];
            if (r < spc[j]) { path[j] = i; spc[j] = r; }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }   // a tie goes to a new sink
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        const int j = remaining[index];
        if (row4col[j] == -1) sink = j;
        else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

// cost [nr, nc] (row-major fp32) -> match [nr]: the column assigned to every row, -1 for the rows left out (nr > nc).  0 ok, 1 infeasible / invalid
static int solve_one(const float* cost_f, int nr0, int nc0, int* match) {
    for (int r = 0; r < nr0; ++r) match[r] = -1;
    if (nr0 == 0 || nc0 == 0) return 0;
    const bool transpose = nc0 < nr0;                                       // a tall matrix is solved transposed
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    std::vector<double> cost((size_t)nr * nc);
    for (int i = 0; i < nr0; ++i)
        for (int j = 0; j < nc0; ++j) {
            const double c = (double)cost_f[(long long)i * nc0 + j];
            if (std::isnan(c) || c == -INFINITY) return 1;
            cost[transpose ? (size_t)j * nc + i : (size_t)i * nc + j] = c;
        }
    std::vector<double> u(nr, 0.0), v(nc, 0.0), spc(nc);
    std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (int cur = 0; cur < nr; ++cur) {
        double min_val;
        const int sink = augmenting_path(nc, cost.data(), u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
        if (sink < 0) return 1;
        u[cur] += min_val;
        for (int i = 0; i < nr; ++i)
            if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        int j = sink;
        while (true) {
            const int i = path[j];
            row4col[j] = i;
            std::swap(col4row[i], j);
            if (i == cur) break;
        }
    }
    if (transpose) { for (int i = 0; i < nr; ++i) match[col4row[i]] = i; }   // (rows of the transposed problem are the boxes)
    else { for (int i = 0; i < nr; ++i) match[i] = col4row[i]; }
    return 0;
}

}  // namespace

// cost [L, R, G] fp32 on the HOST -> match [L, R] int32 on the host: the box assigned to every query of every layer or -1.  The layers are
// solved on `threads` host threads (<= 0: one per layer).  Returns MV2D_OK, or an error when a layer has a NaN / -inf entry or no feasible
// assignment (SciPy raises ValueError there).
extern "C" int mv2d_lsap_layers(const float* cost, int L, int R, int G, int* match, int threads) {
    MV2D_CHECK_ARG(match && L >= 0 && R >= 0 && G >= 0 && (cost || (long long)L * R * G == 0), "mv2d_lsap_layers: bad args");
    if (L == 0) return MV2D_OK;
    std::vector<int> rc(L, 0);
    auto work = [&](int l) { rc[l] = solve_one(cost + (long long)l * R * G, R, G, match + (long long)l * R); };
    const int nt = threads <= 0 ? L : (threads < L ? threads : L);
    if (nt <= 1 || L == 1) {
        for (int l = 0; l < L; ++l) work(l);
    } else {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back([&, t] { for (int l = t; l < L; l += nt) work(l); });
        for (int l = 0; l < L; l += nt) work(l);
        for (auto& x : th) x.join();
    }
    for (int l = 0; l < L; ++l) MV2D_CHECK_ARG(rc[l] == 0, "mv2d_lsap_layers: cost matrix is infeasible or has NaN / -inf entries");
    return MV2D_OK;
}
