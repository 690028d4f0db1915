// subsetsum_kernel.hip -- two-sided subset-sum DP (reference scallop/subsetsum.cc:20-206), one wavefront per instance.
//
// The reference rescales both integer multisets to a common sum <= 1000, fills a reachability table with
// last-item back-pointers per side (table[i][j]), merges the achievable sums of both sides, picks the closest
// cross pair and back-traces it.  Here the DP rows are filled 64 sums per step (lane = sum j), the achievable
// sums of both sides are compacted into one ordered list with ballot + popcount prefix sums, and the closest
// cross pair is a wave arg-min.  Tables are u8 back-pointers in LDS (2 sides x 33 rows x 1000 sums = 66 KB).
//
// This component is dead in the reference's live path and is wired here exactly as a separately
// KAT-checked kernel (tests/test_gpu_parity.py::test_subsetsum_kernel_matches_reference_golden, tests/test_oracle_pins.py: reference KAT subsetsum.cc:263-282 + oracle/_ref/ref_subsetsum).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/aletsch_decomp.h"

namespace {

constexpr int SS_MAXN = 32;       // items per side
constexpr int SS_MAXS = 1040;     // sums: the common bound is 1000, plus up to one unit per item bumped from 0 to 1 (subsetsum.cc:48-55)

struct SSLds {
    uint8_t t[2][SS_MAXN + 1][SS_MAXS];     // back-pointer + 1 (0 = unreachable, k+1 = table value k)
    int32_t val[2][SS_MAXN], lab[2][SS_MAXN];
    int32_t lsum[2 * SS_MAXS]; uint8_t ltag[2 * SS_MAXS];
    int32_t n[2], ub[2], nlist, total;
};

__device__ __forceinline__ int tget(const SSLds &S, int side, int i, int j) { return (int)S.t[side][i][j] - 1; }

__global__ void __launch_bounds__(64) subsetsum_kernel(int32_t n_inst, const int32_t *ns, const int32_t *nt, const int64_t *off_s, const int64_t *off_t,
                                                        const int32_t *src_val, const int32_t *src_lab, const int32_t *tgt_val, const int32_t *tgt_lab,
                                                        double *err, int32_t *out_ns, int32_t *out_nt, int32_t *out_s, int32_t *out_t, int32_t *status)
{
    __shared__ SSLds S;
    const int lane = threadIdx.x;
    for(int inst = blockIdx.x; inst < n_inst; inst += gridDim.x) {
        int n1 = ns[inst], n2 = nt[inst];
        if(n1 < 1 || n2 < 1 || n1 > SS_MAXN || n2 > SS_MAXN) { if(lane == 0) { status[inst] = ALD_ERR_INVALID; out_ns[inst] = 0; out_nt[inst] = 0; err[inst] = 0; } continue; }
        // ---- rescale (subsetsum.cc:31-71), scalar ----
        if(lane == 0) {
            int s1 = 0, s2 = 0;
            for(int i = 0; i < n1; i++) { S.val[0][i] = src_val[off_s[inst] + i]; S.lab[0][i] = src_lab[off_s[inst] + i]; s1 += S.val[0][i]; }
            for(int i = 0; i < n2; i++) { S.val[1][i] = tgt_val[off_t[inst] + i]; S.lab[1][i] = tgt_lab[off_t[inst] + i]; s2 += S.val[1][i]; }
            int ubound = (s1 > s2) ? s1 : s2;
            if(ubound > 1000) ubound = 1000;
            double r1 = ubound * 1.0 / s1, r2 = ubound * 1.0 / s2;
            for(int i = 0; i < n1; i++) { int v = (int)(S.val[0][i] * r1); if(v <= 0) v = 1; S.val[0][i] = v; }
            for(int i = 0; i < n2; i++) { int v = (int)(S.val[1][i] * r2); if(v <= 0) v = 1; S.val[1][i] = v; }
            s1 = 0; s2 = 0;
            for(int i = 0; i < n1; i++) s1 += S.val[0][i];
            for(int i = 0; i < n2; i++) s2 += S.val[1][i];
            S.ub[0] = s1 - 1; S.ub[1] = s2 - 1; S.n[0] = n1; S.n[1] = n2; S.total = s1 + s2;
            for(int side = 0; side < 2; side++) {          // sort (value, label) pairs ascending
                int n = S.n[side];
                for(int i = 1; i < n; i++) { int v = S.val[side][i], l = S.lab[side][i], j = i - 1;
                    while(j >= 0 && (S.val[side][j] > v || (S.val[side][j] == v && S.lab[side][j] > l))) { S.val[side][j + 1] = S.val[side][j]; S.lab[side][j + 1] = S.lab[side][j]; j--; }
                    S.val[side][j + 1] = v; S.lab[side][j + 1] = l; }
            }
        }
        __syncthreads();
        bool bad = (S.ub[0] >= SS_MAXS || S.ub[1] >= SS_MAXS || S.ub[0] < 0 || S.ub[1] < 0);
        if(bad) { if(lane == 0) { status[inst] = ALD_ERR_INVALID; out_ns[inst] = 0; out_nt[inst] = 0; err[inst] = 0; } __syncthreads(); continue; }
        // ---- init + fill (subsetsum.cc:73-112): row i depends on row i-1 only, so a row is filled 64 sums at a time ----
        for(int side = 0; side < 2; side++) {
            int n = S.n[side], ub = S.ub[side];
            for(int j = lane; j <= ub; j += 64) S.t[side][0][j] = (j == 0) ? 1 : 0;
            __syncthreads();
            for(int i = 1; i <= n; i++) {
                int s = S.val[side][i - 1];
                for(int j = lane; j <= ub; j += 64) {
                    int v = -1;
                    if(j == 0) v = 0;
                    else {
                        if(j >= s && tget(S, side, i - 1, j - s) >= 0) v = i;
                        int up = tget(S, side, i - 1, j);
                        if(up >= 0) v = up;
                    }
                    S.t[side][i][j] = (uint8_t)(v + 1);
                }
                __syncthreads();
            }
        }
        // ---- optimize (subsetsum.cc:137-206): ordered list of achievable sums of both sides, ballot + popcount prefix ----
        int base = 0;
        int ubmax = S.ub[0] > S.ub[1] ? S.ub[0] : S.ub[1];
        for(int j0 = 1; j0 <= ubmax; j0 += 64) {
            int j = j0 + lane;
            bool a1 = (j <= S.ub[0]) && tget(S, 0, S.n[0], j) >= 0;
            bool a2 = (j <= S.ub[1]) && tget(S, 1, S.n[1], j) >= 0;
            unsigned long long b1 = __ballot(a1), b2 = __ballot(a2);
            unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            int pos = base + __popcll(b1 & lt) + __popcll(b2 & lt);
            if(a1) { S.lsum[pos] = j; S.ltag[pos] = 1; pos++; }
            if(a2) { S.lsum[pos] = j; S.ltag[pos] = 2; }
            base += __popcll(b1) + __popcll(b2);
        }
        __syncthreads();
        int nl = base;
        // closest pair of adjacent entries with different tags; the FIRST minimum wins (`>= d` -> continue)
        int bd = INT_MAX, bk = -1;
        for(int i = lane; i + 1 < nl; i += 64) {
            if(S.ltag[i] == S.ltag[i + 1]) continue;
            int d = S.lsum[i + 1] - S.lsum[i];
            if(d < bd) { bd = d; bk = i; }
        }
        for(int off = 32; off >= 1; off >>= 1) {
            int d2 = __shfl_xor(bd, off, 64), k2 = __shfl_xor(bk, off, 64);
            if(k2 >= 0 && (bk < 0 || d2 < bd || (d2 == bd && k2 < bk))) { bd = d2; bk = k2; }
        }
        if(lane == 0) {
            if(bk < 0) { status[inst] = ALD_ERR_INVALID; out_ns[inst] = 0; out_nt[inst] = 0; err[inst] = 0; }      // reference: assert(k != -1)
            else {
                int cs = 0, ct = 0;
                for(int q = 0; q < 2; q++) {                 // backtrace (subsetsum.cc:114-135) for v[k] then v[k+1]
                    int idx = bk + q; int side = S.ltag[idx] - 1; int x = S.lsum[idx];
                    int n = S.n[side];
                    int32_t *dst = side == 0 ? out_s + 64ll * inst : out_t + 64ll * inst;
                    int cnt = 0;
                    int s = tget(S, side, n, x);
                    while(x >= 1 && s >= 1) { dst[cnt++] = S.lab[side][s - 1]; x -= S.val[side][s - 1]; s = tget(S, side, s - 1, x); }
                    if(side == 0) cs = cnt; else ct = cnt;
                }
                out_ns[inst] = cs; out_nt[inst] = ct;
                int half = (int)(S.total / 2.0);
                err[inst] = bd * 1.0 / half;
                status[inst] = ALD_OK;
            }
        }
        __syncthreads();
    }
}

} // namespace

extern "C" void ald_internal_set_error(const char *);
#define SSCHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { ald_internal_set_error((std::string(#x) + ": " + hipGetErrorString(e_)).c_str()); ok = false; } } while(0)

extern "C" int ald_subsetsum_batch(int device, int32_t n, const int32_t *ns, const int32_t *nt,
                                   const int32_t *src_val, const int32_t *src_lab, const int32_t *tgt_val, const int32_t *tgt_lab,
                                   double *err, int32_t *out_ns, int32_t *out_nt, int32_t *out_s, int32_t *out_t)
{
    if(n < 0 || (n > 0 && (!ns || !nt || !src_val || !src_lab || !tgt_val || !tgt_lab || !err || !out_ns || !out_nt || !out_s || !out_t))) return ALD_ERR_INVALID;
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ALD_ERR_NO_DEVICE;
    if(device < 0 || device >= ndev) return ALD_ERR_INVALID;
    if(n == 0) return ALD_OK;
    if(hipSetDevice(device) != hipSuccess) return ALD_ERR_HIP;
    int64_t *h_off = new int64_t[2 * (size_t)n + 2];
    int64_t *os = h_off, *ot = h_off + n + 1; os[0] = 0; ot[0] = 0;
    for(int i = 0; i < n; i++) { os[i + 1] = os[i] + (ns[i] > 0 ? ns[i] : 0); ot[i + 1] = ot[i] + (nt[i] > 0 ? nt[i] : 0); }
    size_t S = (size_t)os[n], T = (size_t)ot[n];
    void *d = nullptr;
    size_t b_ns = 4 * (size_t)n, b_off = 8 * ((size_t)n + 1), b_s = 4 * (S ? S : 1), b_t = 4 * (T ? T : 1), b_err = 8 * (size_t)n, b_o64 = 4 * 64 * (size_t)n;
    size_t tot = 0; auto take = [&](size_t b) { size_t r = tot; tot = (tot + b + 255) / 256 * 256; return r; };
    size_t o_ns = take(b_ns), o_nt = take(b_ns), o_os = take(b_off), o_ot = take(b_off), o_sv = take(b_s), o_sl = take(b_s), o_tv = take(b_t), o_tl = take(b_t),
           o_err = take(b_err), o_ons = take(b_ns), o_ont = take(b_ns), o_outs = take(b_o64), o_outt = take(b_o64), o_st = take(b_ns);
    int rc = ALD_OK;
    if(hipMalloc(&d, tot) != hipSuccess) { delete[] h_off; return ALD_ERR_NOMEM; }
    uint8_t *D = (uint8_t*)d;
    bool ok = hipMemcpy(D + o_ns, ns, b_ns, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(D + o_nt, nt, b_ns, hipMemcpyHostToDevice) == hipSuccess
           && hipMemcpy(D + o_os, os, b_off, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(D + o_ot, ot, b_off, hipMemcpyHostToDevice) == hipSuccess
           && (S == 0 || (hipMemcpy(D + o_sv, src_val, 4 * S, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(D + o_sl, src_lab, 4 * S, hipMemcpyHostToDevice) == hipSuccess))
           && (T == 0 || (hipMemcpy(D + o_tv, tgt_val, 4 * T, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(D + o_tl, tgt_lab, 4 * T, hipMemcpyHostToDevice) == hipSuccess));
    if(ok) {
        int blocks = n < 2048 ? n : 2048;
        (void)hipGetLastError();
        hipLaunchKernelGGL(subsetsum_kernel, dim3(blocks), dim3(64), 0, 0, n, (const int32_t*)(D + o_ns), (const int32_t*)(D + o_nt), (const int64_t*)(D + o_os), (const int64_t*)(D + o_ot),
                           (const int32_t*)(D + o_sv), (const int32_t*)(D + o_sl), (const int32_t*)(D + o_tv), (const int32_t*)(D + o_tl),
                           (double*)(D + o_err), (int32_t*)(D + o_ons), (int32_t*)(D + o_ont), (int32_t*)(D + o_outs), (int32_t*)(D + o_outt), (int32_t*)(D + o_st));
        SSCHK(hipGetLastError());
        if(ok) SSCHK(hipDeviceSynchronize());
    }
    if(ok) {
        int32_t *st = new int32_t[n];
        ok = hipMemcpy(err, D + o_err, b_err, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(out_ns, D + o_ons, b_ns, hipMemcpyDeviceToHost) == hipSuccess
          && hipMemcpy(out_nt, D + o_ont, b_ns, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(out_s, D + o_outs, b_o64, hipMemcpyDeviceToHost) == hipSuccess
          && hipMemcpy(out_t, D + o_outt, b_o64, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(st, D + o_st, b_ns, hipMemcpyDeviceToHost) == hipSuccess;
        if(ok) for(int i = 0; i < n; i++) if(st[i] != ALD_OK) { out_ns[i] = -1; out_nt[i] = -1; }     // instance the reference would assert on / out of range
        delete[] st;
    }
    if(!ok) rc = ALD_ERR_HIP;
    hipFree(d); delete[] h_off;
    return rc;
}
