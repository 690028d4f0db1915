/*
 * subsetsum_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
 * CPU restatement of the reference's two-sided subset-sum DP (scallop/subsetsum.cc:20-206).
 * PINNED: checked against the reference's own known-answer test (subsetsum.cc:263-282:
 * source {10,20,39} target {29,54} -> S=(3 1), T=(2), error 0.05) in tests/test_oracle_pins.py and,
 * on random instances, against oracle/_ref/ref_subsetsum (the reference's subsetsum.cc + equation.cc
 * compiled unmodified from /root/reference by oracle/Makefile).
 */
#pragma once
#include <vector>
#include <algorithm>
#include <climits>

namespace ora {

struct SubsetSum {
    typedef std::pair<int, int> PI;            // (value, label)
    std::vector<PI> source, target;
    int ubound1 = 0, ubound2 = 0;
    std::vector<std::vector<int>> table1, table2;
    std::vector<int> eqn_s, eqn_t; double e = 0;

    SubsetSum(const std::vector<PI> &s, const std::vector<PI> &t) : source(s), target(t) {}

    int solve() {                               // subsetsum.cc:20-29
        if(source.empty() || target.empty()) return -1;
        rescale();
        init(source, table1, ubound1); fill(source, table1, ubound1);
        init(target, table2, ubound2); fill(target, table2, ubound2);
        return optimize();
    }
    void rescale() {                            // subsetsum.cc:31-71
        int s1 = 0, s2 = 0;
        for(auto &p : source) s1 += p.first;
        for(auto &p : target) s2 += p.first;
        int ubound = (s1 > s2) ? s1 : s2;
        if(ubound > 1000) ubound = 1000;
        double r1 = ubound * 1.0 / s1, r2 = ubound * 1.0 / s2;
        for(auto &p : source) { p.first = (int)(p.first * r1); if(p.first <= 0) p.first = 1; }
        for(auto &p : target) { p.first = (int)(p.first * r2); if(p.first <= 0) p.first = 1; }
        s1 = 0; s2 = 0;
        for(auto &p : source) s1 += p.first;
        for(auto &p : target) s2 += p.first;
        ubound1 = s1 - 1; ubound2 = s2 - 1;
        std::sort(source.begin(), source.end());
        std::sort(target.begin(), target.end());
    }
    static void init(const std::vector<PI> &vv, std::vector<std::vector<int>> &table, int ubound) {   // subsetsum.cc:73-91
        table.assign(vv.size() + 1, std::vector<int>(ubound + 1, -1));
        for(size_t i = 0; i <= vv.size(); i++) table[i][0] = 0;
    }
    static void fill(const std::vector<PI> &vv, std::vector<std::vector<int>> &table, int ubound) {   // subsetsum.cc:93-112
        for(int j = 1; j <= ubound; j++) for(int i = 1; i <= (int)vv.size(); i++) {
            int s = vv[i - 1].first;
            if(j >= s && table[i - 1][j - s] >= 0) table[i][j] = i;
            if(table[i - 1][j] >= 0) table[i][j] = table[i - 1][j];
        }
    }
    static int backtrace(int t, const std::vector<PI> &vv, const std::vector<std::vector<int>> &table, std::vector<int> &ss) {  // subsetsum.cc:114-135
        ss.clear();
        if(table.empty()) return -1;
        if(t <= 0 || t > (int)table[0].size()) return -1;
        int n = (int)vv.size();
        if(table[n][t] == -1) return -1;
        int x = t, s = table[n][t];
        while(x >= 1 && s >= 1) { ss.push_back(vv[s - 1].second); x -= vv[s - 1].first; s = table[s - 1][x]; }
        return 0;
    }
    int optimize() {                            // subsetsum.cc:137-206
        std::vector<PI> v;
        int n1 = (int)source.size(), n2 = (int)target.size();
        for(int i = 1; i <= ubound1; i++) { if(table1[n1][i] < 0) continue; v.push_back(PI(i, 1)); }
        for(int i = 1; i <= ubound2; i++) { if(table2[n2][i] < 0) continue; v.push_back(PI(i, 2)); }
        std::sort(v.begin(), v.end());
        int d = INT_MAX, k = -1;
        for(int i = 0; i + 1 < (int)v.size(); i++) {
            if(v[i].second == v[i + 1].second) continue;
            if(v[i + 1].first - v[i].first >= d) continue;
            d = v[i + 1].first - v[i].first; k = i;
        }
        if(k == -1) return -1;                  // reference: assert(k != -1)
        if(v[k].second == 1) backtrace(v[k].first, source, table1, eqn_s); else backtrace(v[k].first, target, table2, eqn_t);
        if(v[k + 1].second == 1) backtrace(v[k + 1].first, source, table1, eqn_s); else backtrace(v[k + 1].first, target, table2, eqn_t);
        int s = 0;
        for(auto &p : source) s += p.first;
        for(auto &p : target) s += p.first;
        s = s / 2.0;
        e = d * 1.0 / s;
        return 0;
    }
};

} // namespace ora
