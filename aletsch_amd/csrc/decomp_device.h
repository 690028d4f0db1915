// decomp_device.h -- the per-graph decomposition engine executed by ONE 64-lane wavefront.
//
// One wavefront owns one splice graph.  The graph's hot state (sorted adjacency lists, endpoints,
// creation ids, FP64 weights, degrees) is a file-scope LDS struct (g_H); cold per-edge / per-vertex
// state (coverage bookkeeping, sample support, phasing lists, path bitmasks) lives in the wave's
// private HBM slab, laid out at compile time.  Rule sweeps (reference scallop.cc:844-945,
// 1180-1234) are evaluated one vertex per lane and reduced with ballots / shuffles; graph surgery
// (scallop.cc:2198-2484, 1675-1986) and the router (router.cc) are scalar routines run by lane 0,
// compiled as real functions (not inlined) to keep the register footprint small.
//
// Every function cites the reference file:line whose behaviour it reproduces.  Ordering rules are
// the canonical ones of SURVEY.md Appendix A: edge "pointer order" == creation id.
//
// Build modes (one translation unit per size class, -DALD_CLASS_ID=k):
//   hipcc --offload-arch=gfx950   : the product (ALD_WAVE == 64).
//   g++ -DALD_EMU                 : single-lane emulation, compiled ONLY by tests/kernel_emu to debug
//                                   the algorithm on a CPU-only box.  Never part of the product library.
#pragma once
#include "decomp_common.h"

#ifndef ALD_CLASS_ID
#error "compile with -DALD_CLASS_ID=<0..4> (one translation unit per size class)"
#endif
#define ALD_CAT2(a, b) a##b
#define ALD_CAT(a, b) ALD_CAT2(a, b)
#define ALD_CLASS_NS ALD_CAT(ald_c, ALD_CLASS_ID)

namespace ALD_CLASS_NS {
using namespace ald;

enum { MAXV = ClassDims<ALD_CLASS_ID>::MAXV, MAXE = ClassDims<ALD_CLASS_ID>::MAXE, NW = ClassDims<ALD_CLASS_ID>::NW };
typedef uint16_t IDX;
static constexpr IDX NIL = (IDX)0xFFFF;
typedef ColdLayoutT<MAXV, MAXE, NW> CL;

// ---------------------------------------------------------------------------------------------
// hot state: ONE instance per workgroup (= per wavefront)
// ---------------------------------------------------------------------------------------------
struct Hot {
    double   ew[MAXE];                          // splice_graph::ewrt
    uint32_t eid[MAXE];                         // creation id == scallop edge index
    IDX      es[MAXE], et[MAXE];                // endpoints; es == NIL  <=> slot dead
    IDX      inx[MAXE], onx[MAXE];              // next edge in target's in-list / source's out-list (sorted)
    IDX      in_head[MAXV], out_head[MAXV], in_deg[MAXV], out_deg[MAXV];
    IDX      uidx[MAXE];                        // scratch: edge slot -> router / decomposition local index
    uint8_t  nz[MAXV];                          // scallop::nonzeroset membership
    uint8_t  hflag[MAXE];                       // HF_* (phasing occupancy / extend flags / protect)
    // wave-uniform context
    ALD_GLOBAL uint8_t *cold;                   // this wave's HBM slab
    ALD_GLOBAL const KernelArgs *args;
    double   ro_ratio;                          // router result
    int32_t  ro_type, ro_degree, ro_npairs, tmp0;
    int32_t  g, V0, gstrand;
    int32_t  nv, next_id, slot_hw, free_head, free_cnt, pend_head, status, any_strand, hs_dirty, n_paths, n_iters, n_trace;
    uint32_t sp_used, hl_used; int32_t hl_n;
    int32_t  s_next;
};

#ifdef ALD_EMU
static thread_local Hot g_H;
#else
__shared__ Hot g_H;
#endif
#define H g_H

// cold state: typed views at compile-time offsets of the slab
struct Cold {
    ALD_GLOBAL double *vw; ALD_GLOBAL int32_t *lpos, *rpos, *vtype, *v2v;
    ALD_GLOBAL double *med, *eabd, *econf; ALD_GLOBAL int32_t *mei, *ecount; ALD_GLOBAL uint8_t *estrand; ALD_GLOBAL uint32_t *sp_off, *sp_len;
    ALD_GLOBAL uint64_t *mask;                  // [MAXE*NW] bitmask over ORIGINAL vertices (scallop::mev as a set)
    ALD_GLOBAL int32_t *sp_id; ALD_GLOBAL double *sp_abd;
    ALD_GLOBAL int32_t *hl, *hl_off, *hl_len, *hl_capk, *hl_cnt;   // phasing lists (hyper_set::edges / ecnts); elements are edge SLOTS or -1
    ALD_GLOBAL int32_t *wi; ALD_GLOBAL double *wd;                 // scalar work arrays
    static constexpr uint32_t sp_cap = CL::SP_CAP, hl_cap = CL::HL_CAP;
    static constexpr int32_t hl_maxlists = CL::HL_MAXLISTS, w_cap = CL::W_CAP;
};
ALD_INL Cold cold_view()
{
    ALD_GLOBAL uint8_t *b = H.cold; Cold C;
    C.vw = (ALD_GLOBAL double*)(b + CL::o_vw); C.lpos = (ALD_GLOBAL int32_t*)(b + CL::o_lpos); C.rpos = (ALD_GLOBAL int32_t*)(b + CL::o_rpos);
    C.vtype = (ALD_GLOBAL int32_t*)(b + CL::o_vtype); C.v2v = (ALD_GLOBAL int32_t*)(b + CL::o_v2v);
    C.med = (ALD_GLOBAL double*)(b + CL::o_med); C.eabd = (ALD_GLOBAL double*)(b + CL::o_eabd); C.econf = (ALD_GLOBAL double*)(b + CL::o_econf);
    C.mei = (ALD_GLOBAL int32_t*)(b + CL::o_mei); C.ecount = (ALD_GLOBAL int32_t*)(b + CL::o_ecount); C.estrand = (ALD_GLOBAL uint8_t*)(b + CL::o_estrand);
    C.sp_off = (ALD_GLOBAL uint32_t*)(b + CL::o_spoff); C.sp_len = (ALD_GLOBAL uint32_t*)(b + CL::o_splen); C.mask = (ALD_GLOBAL uint64_t*)(b + CL::o_mask);
    C.sp_id = (ALD_GLOBAL int32_t*)(b + CL::o_spid); C.sp_abd = (ALD_GLOBAL double*)(b + CL::o_spabd);
    C.hl = (ALD_GLOBAL int32_t*)(b + CL::o_hl); C.hl_off = (ALD_GLOBAL int32_t*)(b + CL::o_hloff); C.hl_len = (ALD_GLOBAL int32_t*)(b + CL::o_hllen);
    C.hl_capk = (ALD_GLOBAL int32_t*)(b + CL::o_hlcapk); C.hl_cnt = (ALD_GLOBAL int32_t*)(b + CL::o_hlcnt);
    C.wi = (ALD_GLOBAL int32_t*)(b + CL::o_wi); C.wd = (ALD_GLOBAL double*)(b + CL::o_wd);
    return C;
}
#define COLD const Cold C = cold_view()
#define PRM (H.args->prm)

// ---------------------------------------------------------------- small helpers
ALD_INL void fail(int st) { if(H.status == 0) H.status = st; }
ALD_FN void trace(int code, int a, int b, double v)
{
    H.n_iters++;
    ALD_GLOBAL const KernelArgs *A = H.args;
    int cap = A->out.trace_cap;
    if(cap <= 0) return;
    int k = H.n_trace++;
    if(k < cap) { int64_t o = (int64_t)H.g * cap + k; A->out.trace_codes[3 * o] = code; A->out.trace_codes[3 * o + 1] = a; A->out.trace_codes[3 * o + 2] = b; A->out.trace_vals[o] = v; }
}
ALD_INL bool tracing() { return H.args->out.trace_cap > 0; }
ALD_INL int first_in(int v) { return H.in_head[v] == NIL ? -1 : (int)H.in_head[v]; }
ALD_INL int first_out(int v) { return H.out_head[v] == NIL ? -1 : (int)H.out_head[v]; }
ALD_INL int next_in(int e) { return H.inx[e] == NIL ? -1 : (int)H.inx[e]; }
ALD_INL int next_out(int e) { return H.onx[e] == NIL ? -1 : (int)H.onx[e]; }
ALD_INL double in_weights(int v) { double w = 0; for(int e = first_in(v); e >= 0; e = next_in(e)) w += H.ew[e]; return w; }    // splice_graph.cc:187-198
ALD_INL double out_weights(int v) { double w = 0; for(int e = first_out(v); e >= 0; e = next_out(e)) w += H.ew[e]; return w; } // splice_graph.cc:174-185

// ---------------------------------------------------------------- sorted adjacency lists (scalar code)
// in-list of v ordered by (source, id); out-list ordered by (target, id): graph/edge_base.h:35-45
ALD_FN void link_in(int v, int e)
{
    uint32_t ks = H.es[e], kid = H.eid[e];
    int prev = -1, cur = first_in(v);
    while(cur >= 0) { uint32_t cs = H.es[cur]; if(cs > ks || (cs == ks && H.eid[cur] > kid)) break; prev = cur; cur = next_in(cur); }
    H.inx[e] = cur < 0 ? NIL : (IDX)cur;
    if(prev < 0) H.in_head[v] = (IDX)e; else H.inx[prev] = (IDX)e;
    H.in_deg[v]++;
}
ALD_FN void link_out(int v, int e)
{
    uint32_t kt = H.et[e], kid = H.eid[e];
    int prev = -1, cur = first_out(v);
    while(cur >= 0) { uint32_t ct = H.et[cur]; if(ct > kt || (ct == kt && H.eid[cur] > kid)) break; prev = cur; cur = next_out(cur); }
    H.onx[e] = cur < 0 ? NIL : (IDX)cur;
    if(prev < 0) H.out_head[v] = (IDX)e; else H.onx[prev] = (IDX)e;
    H.out_deg[v]++;
}
ALD_FN void unlink_in(int v, int e)
{
    int prev = -1, cur = first_in(v), guard = MAXE;
    while(cur >= 0 && cur != e && guard-- > 0) { prev = cur; cur = next_in(cur); }
    if(cur != e) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }       // cannot happen on a consistent state; never walk off a list
    if(prev < 0) H.in_head[v] = H.inx[e]; else H.inx[prev] = H.inx[e];
    H.in_deg[v]--;
}
ALD_FN void unlink_out(int v, int e)
{
    int prev = -1, cur = first_out(v), guard = MAXE;
    while(cur >= 0 && cur != e && guard-- > 0) { prev = cur; cur = next_out(cur); }
    if(cur != e) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
    if(prev < 0) H.out_head[v] = H.onx[e]; else H.onx[prev] = H.onx[e];
    H.out_deg[v]--;
}
ALD_INL int free_slots() { return H.free_cnt + (MAXE - H.slot_hw); }
// directed_graph::add_edge (directed_graph.cc:38-48) + i2e.push_back: the new id is the largest
ALD_FN int add_edge(int s, int t)
{
    int e;
    if(H.free_head >= 0) { e = H.free_head; H.free_head = H.onx[e] == NIL ? -1 : (int)H.onx[e]; H.free_cnt--; }
    else if(H.slot_hw < MAXE) e = H.slot_hw++;
    else { fail(ALD_ST_CAPACITY); return -1; }
    H.es[e] = (IDX)s; H.et[e] = (IDX)t; H.eid[e] = (uint32_t)H.next_id++; H.hflag[e] = 0; H.ew[e] = 0;
    link_out(s, e); link_in(t, e);
    return e;
}
// scallop::remove_edge (scallop.cc:2380-2392).  A slot that phasing lists may still name (HF_PROT) is parked
// until the compound operation has called hs_remove on it; all others are recycled at once.
ALD_FN void kill_edge(int e)
{
    unlink_out(H.es[e], e); unlink_in(H.et[e], e);
    H.es[e] = NIL;
    if(H.hflag[e] & HF_PROT) { H.onx[e] = H.pend_head < 0 ? NIL : (IDX)H.pend_head; H.pend_head = e; }
    else { H.onx[e] = H.free_head < 0 ? NIL : (IDX)H.free_head; H.free_head = e; H.free_cnt++; }
}
ALD_FN void flush_pending()
{
    int guard = MAXE;
    while(H.pend_head >= 0 && guard-- > 0) { int e = H.pend_head; H.pend_head = H.onx[e] == NIL ? -1 : (int)H.onx[e]; H.hflag[e] = 0; H.onx[e] = H.free_head < 0 ? NIL : (IDX)H.free_head; H.free_head = e; H.free_cnt++; }
}
ALD_FN void move_edge(int e, int x, int y)      // directed_graph.cc:180-194
{
    unlink_out(H.es[e], e); unlink_in(H.et[e], e);
    H.es[e] = (IDX)x; H.et[e] = (IDX)y;
    link_out(x, e); link_in(y, e);
}

// splice_graph::get_strand_degree / mixed_strand_vertex (splice_graph.cc:1375-1406)
ALD_INL void strand_degree(int v, int vs[6])
{
    COLD;
    for(int k = 0; k < 6; k++) vs[k] = 0;
    for(int e = first_in(v); e >= 0; e = next_in(e)) vs[C.estrand[e]]++;
    for(int e = first_out(v); e >= 0; e = next_out(e)) vs[C.estrand[e] + 3]++;
}
ALD_INL bool mixed_strand_vertex(int v)
{
    if(!H.any_strand) return false;
    int vs[6]; strand_degree(v, vs);
    return (vs[1] + vs[4] >= 1) && (vs[2] + vs[5] >= 1);
}
ALD_INL void borrow_edge_strand(const Cold &C, int e1, int e2) { int s2 = C.estrand[e2]; if(s2 == 0) return; C.estrand[e1] = (uint8_t)s2; }   // scallop.cc:1997-2007

// ---------------------------------------------------------------- sample support (edge_info.samples / spAbd)
// intersection with per-sample min, abd = sum in ascending sample order (scallop.cc:2300-2318, 1915-1933)
ALD_FN bool intersect_samples(int e1, int e2, int z)
{
    COLD;
    uint32_t o1 = C.sp_off[e1], n1 = C.sp_len[e1], o2 = C.sp_off[e2], n2 = C.sp_len[e2];
    uint32_t need = n1 < n2 ? n1 : n2;
    uint32_t o = H.sp_used;
    if(o + need > C.sp_cap) { fail(ALD_ST_CAPACITY); return false; }
    uint32_t i = 0, j = 0, k = 0; double abd = 0;
    while(i < n1 && j < n2) {
        int a = C.sp_id[o1 + i], b = C.sp_id[o2 + j];
        if(a < b) i++; else if(b < a) j++;
        else { double x = C.sp_abd[o1 + i], y = C.sp_abd[o2 + j]; double c = (y < x) ? y : x;     // std::min(x, y)
               C.sp_id[o + k] = a; C.sp_abd[o + k] = c; abd += c; k++; i++; j++; }
    }
    H.sp_used = o + k;
    C.sp_off[z] = o; C.sp_len[z] = k; C.ecount[z] = (int32_t)k; C.eabd[z] = abd;
    return true;
}
// router.cc:1035-1038: sum over common samples of 0.99*min + 0.01*max
ALD_FN double common_abd(int e1, int e2)
{
    COLD;
    uint32_t o1 = C.sp_off[e1], n1 = C.sp_len[e1], o2 = C.sp_off[e2], n2 = C.sp_len[e2];
    uint32_t i = 0, j = 0; double c = 0;
    while(i < n1 && j < n2) {
        int a = C.sp_id[o1 + i], b = C.sp_id[o2 + j];
        if(a < b) i++; else if(b < a) j++;
        else { double x = C.sp_abd[o1 + i], y = C.sp_abd[o2 + j]; double mn = (y < x) ? y : x, mx = (x < y) ? y : x; c += 0.99 * mn + 0.01 * mx; i++; j++; }
    }
    return c;
}

// ---------------------------------------------------------------- phasing lists: hyper_set (scalar code)
// Lists hold edge SLOTS; every query scans the lists, which is equivalent to the reference's e2s index because
// e2s[e] is always a superset of the lists that contain e (hyper_set.cc:626-675,787-818,865-902).
ALD_FN void hs_refresh_flags()                  // per-slot OCC / LEXT / REXT: hyper_set.cc:949-983 left/right_extend
{
    if(!H.hs_dirty) return;
    COLD;
    for(int e = 0; e < H.slot_hw; e++) H.hflag[e] &= (uint8_t)HF_PROT;
    int nl = H.hl_n;
    for(int k = 0; k < nl; k++) {
        ALD_GLOBAL int32_t *v = C.hl + C.hl_off[k]; int n = C.hl_len[k];
        for(int i = 0; i < n; i++) {
            int e = v[i]; if(e < 0) continue;
            uint8_t f = HF_OCC;
            if(i >= 1 && v[i - 1] != -1) f |= HF_LEXT;
            if(i + 1 < n && v[i + 1] != -1) f |= HF_REXT;
            H.hflag[e] |= f;
        }
    }
    H.hs_dirty = 0;
}
ALD_FN void hs_remove(int e)                    // hyper_set.cc:787-818
{
    int nl = H.hl_n; if(nl == 0) return;
    COLD;
    for(int k = 0; k < nl; k++) { ALD_GLOBAL int32_t *v = C.hl + C.hl_off[k]; int n = C.hl_len[k]; for(int i = 0; i < n; i++) if(v[i] == e) { v[i] = -1; H.hs_dirty = 1; } }
}
ALD_FN void hs_replace1(int x, int e)           // hyper_set.cc:609-615 -> 626-675 with |v| == 1
{
    int nl = H.hl_n; if(nl == 0) return;
    COLD;
    for(int k = 0; k < nl; k++) { ALD_GLOBAL int32_t *v = C.hl + C.hl_off[k]; int n = C.hl_len[k]; for(int i = 0; i < n; i++) if(v[i] == x) { v[i] = e; H.hs_dirty = 1; } }
}
ALD_FN void hs_replace2(int x, int y, int e)    // hyper_set.cc:617-624 -> 626-675 with |v| == 2
{
    int nl = H.hl_n; if(nl == 0) return;
    COLD;
    for(int k = 0; k < nl; k++) {
        ALD_GLOBAL int32_t *v = C.hl + C.hl_off[k]; int n = C.hl_len[k]; int w = 0;
        // matches of a 2-pattern with x != y cannot overlap; replace (x,y) by e left to right
        for(int i = 0; i < n; i++) {
            if(i + 1 < n && v[i] == x && v[i + 1] == y) { v[w++] = e; i++; H.hs_dirty = 1; }
            else v[w++] = v[i];
        }
        C.hl_len[k] = w;
    }
}
ALD_FN void hs_insert_between(int x, int y, int e)   // hyper_set.cc:865-902
{
    int nl = H.hl_n; if(nl == 0) return;
    COLD;
    for(int k = 0; k < nl; k++) {
        int n = C.hl_len[k]; ALD_GLOBAL int32_t *v = C.hl + C.hl_off[k];
        int cnt = 0;
        for(int i = 0; i + 1 < n; i++) if(v[i] == x && v[i + 1] == y) cnt++;
        if(cnt == 0) continue;
        if(n + cnt > C.hl_capk[k]) {             // relocate the list to the end of the pool with slack
            uint32_t ncap = (uint32_t)(n + cnt) * 2u + 4u, o = H.hl_used;
            if(o + ncap > C.hl_cap) { fail(ALD_ST_CAPACITY); return; }
            for(int i = 0; i < n; i++) C.hl[o + i] = v[i];
            H.hl_used = o + ncap; C.hl_off[k] = (int32_t)o; C.hl_capk[k] = (int32_t)ncap; v = C.hl + o;
        }
        // the reference scans left to right and inserts e after every x that is followed by y
        int w = n + cnt - 1;
        for(int i = n - 1; i >= 0; i--) {
            v[w--] = v[i];
            if(i >= 1 && v[i - 1] == x && v[i] == y) v[w--] = e;
        }
        C.hl_len[k] = n + cnt; H.hs_dirty = 1;
    }
}
// hyper_set.cc:1003-1042 (side == 2, left_dominate) and 1044-1082 (side == 1, right_dominate)
ALD_FN bool hs_dominate(int e, int side)
{
    COLD;
    int nl = H.hl_n;
    ALD_GLOBAL int32_t *x1 = C.wi, *x2 = C.wi + C.w_cap / 4; int n1 = 0, n2 = 0; const int cap = C.w_cap / 8;
    for(int k = 0; k < nl; k++) {
        ALD_GLOBAL int32_t *v = C.hl + C.hl_off[k]; int n = C.hl_len[k];
        if(side == 2) {
            for(int i = 0; i + 1 < n; i++) {
                if(v[i] != e) continue;
                if(v[i + 1] == -1) continue;
                int a = v[i + 1], b = (i + 2 < n) ? v[i + 2] : -1;
                if(i == 0 || v[i - 1] == -1) { if(n1 < cap) { x1[2 * n1] = a; x1[2 * n1 + 1] = b; n1++; } else fail(ALD_ST_CAPACITY); }
                else { if(n2 + 2 <= cap) { x2[2 * n2] = a; x2[2 * n2 + 1] = -1; n2++; if(i + 2 < n) { x2[2 * n2] = a; x2[2 * n2 + 1] = b; n2++; } } else fail(ALD_ST_CAPACITY); }
            }
        } else {
            for(int i = 1; i < n; i++) {
                if(v[i] != e) continue;
                if(v[i - 1] == -1) continue;
                int a = v[i - 1], b = (i - 2 >= 0) ? v[i - 2] : -1;
                if(i == n - 1 || v[i + 1] == -1) { if(n1 < cap) { x1[2 * n1] = a; x1[2 * n1 + 1] = b; n1++; } else fail(ALD_ST_CAPACITY); }
                else { if(n2 + 2 <= cap) { x2[2 * n2] = a; x2[2 * n2 + 1] = -1; n2++; if(i - 2 >= 0) { x2[2 * n2] = a; x2[2 * n2 + 1] = b; n2++; } } else fail(ALD_ST_CAPACITY); }
            }
        }
    }
    for(int i = 0; i < n1; i++) { bool f = false; for(int j = 0; j < n2 && !f; j++) f = (x2[2 * j] == x1[2 * i] && x2[2 * j + 1] == x1[2 * i + 1]); if(!f) return false; }
    return true;
}

// ---------------------------------------------------------------- edge surgery (scalar code)
// scallop::split_edge (scallop.cc:2433-2484)
ALD_FN int split_edge(int ei, double w)
{
    if(!(w >= PRM.min_w - kSMIN)) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return -1; }
    double ww = H.ew[ei];
    if(fabs(ww - w) <= kSMIN) return ei;
    int s = H.es[ei], t = H.et[ei];
    int p2 = add_edge(s, t);
    if(p2 < 0) return -1;
    COLD;
    double www = ww - w;
    double mw = PRM.min_w;
    if(www <= mw) www = mw;
    H.ew[ei] = www; H.ew[p2] = w;
    C.estrand[p2] = C.estrand[ei]; C.ecount[p2] = C.ecount[ei]; C.eabd[p2] = C.eabd[ei]; C.econf[p2] = C.econf[ei];
    C.sp_off[p2] = C.sp_off[ei]; C.sp_len[p2] = C.sp_len[ei];            // immutable support lists are shared
    for(int k = 0; k < NW; k++) C.mask[(int64_t)p2 * NW + k] = C.mask[(int64_t)ei * NW + k];
    C.mei[p2] = C.mei[ei]; C.med[p2] = C.med[ei] * w / ww;
    return p2;
}
// scallop::merge_adjacent_equal_edges (scallop.cc:2242-2378)
ALD_FN int merge_adjacent_equal_edges(int x, int y)
{
    if(x < 0 || y < 0) return -1;
    int xs = H.es[x], xt = H.et[x], ys = H.es[y], yt = H.et[y];
    if(xt != ys && yt != xs) return -1;
    if(yt == xs) { int t = x; x = y; y = t; xs = H.es[x]; xt = H.et[x]; ys = H.es[y]; yt = H.et[y]; }
    int n = add_edge(xs, yt);
    if(n < 0) return -1;
    COLD;
    double wx0 = H.ew[x], wy0 = H.ew[y];
    if(!(fabs(wx0 - wy0) <= kSMIN)) { fail(ALD_ST_INVARIANT + ALD_INV_MERGE_EQUAL); return -1; }
    H.ew[n] = wx0 * 0.5 + wy0 * 0.5;
    if(!(C.ecount[x] > 0 && C.ecount[y] > 0)) { fail(ALD_ST_INVARIANT + ALD_INV_COUNT); return -1; }
    if(!intersect_samples(x, y, n)) return -1;
    C.econf[n] = C.econf[x] + C.econf[y];
    C.estrand[n] = 0; borrow_edge_strand(C, n, x); borrow_edge_strand(C, n, y);
    int ov = C.v2v[xt];
    for(int k = 0; k < NW; k++) C.mask[(int64_t)n * NW + k] = C.mask[(int64_t)x * NW + k] | C.mask[(int64_t)y * NW + k];
    if(ov >= 0) C.mask[(int64_t)n * NW + (ov >> 6)] |= (1ull << (ov & 63));
    double sum1 = in_weights(xt), sum2 = out_weights(xt);
    double sum = (sum1 + sum2) * 0.5;
    double r1 = C.vw[xt] * (wx0 + wy0) * 0.5 / sum;
    double r2 = C.vw[xt] - r1;
    C.vw[xt] = r2;
    int mi = C.rpos[xt] - C.lpos[xt] + C.mei[x] + C.mei[y];
    double md = mi * r1 + C.med[x] + C.med[y];
    C.med[n] = md; C.mei[n] = mi;
    kill_edge(x); kill_edge(y);
    if(H.in_deg[xt] == 0 && H.out_deg[xt] == 0) H.nz[xt] = 0;
    return n;
}
// scallop::merge_adjacent_edges(x, y, ww) (scallop.cc:2394-2416)
ALD_FN int merge_adjacent_edges(int x, int y, double ww)
{
    if(!(ww >= PRM.min_w - kSMIN)) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return -1; }
    if(H.et[x] != H.es[y]) { int t = x; x = y; y = t; }
    int x1 = split_edge(x, ww); if(x1 < 0) return -1;
    int y1 = split_edge(y, ww); if(y1 < 0) return -1;
    return merge_adjacent_equal_edges(x1, y1);
}
// scallop::balance_vertex (scallop.cc:2486-2576)
ALD_FN void balance_vertex(int v)
{
    if(H.in_deg[v] == 0 || H.out_deg[v] == 0) return;
    const double mw = PRM.min_w;
    double w1 = 0, w2 = 0;
    for(int e = first_in(v); e >= 0; e = next_in(e)) { double w = H.ew[e]; if(!(w >= mw - kSMIN)) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; } w1 += w; }
    for(int e = first_out(v); e >= 0; e = next_out(e)) { double w = H.ew[e]; if(!(w >= mw - kSMIN)) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; } w2 += w; }
    double ww = sqrt(w1 * w2);
    double r1 = ww / w1, r2 = ww / w2;
    double m1 = 0, m2 = 0;
    for(int e = first_in(v); e >= 0; e = next_in(e)) { double wy = H.ew[e] * r1; if(wy < mw) { m1 += mw - wy; wy = mw; } H.ew[e] = wy; }
    for(int e = first_out(v); e >= 0; e = next_out(e)) { double wy = H.ew[e] * r2; if(wy < mw) { m2 += mw - wy; wy = mw; } H.ew[e] = wy; }
    if(m1 > m2) { int e = first_out(v); H.ew[e] = H.ew[e] + m1 - m2; }
    else if(m1 < m2) { int e = first_in(v); H.ew[e] = H.ew[e] + m2 - m1; }
}

// pe2w as a sorted array in the work area: keys (id1,id2) ascending == std::map<PI,double> order (router.h:23)
// pair arrays live in the upper halves of wi / wd
#define PW_E1(C) ((C).wi + Cold::w_cap / 2)
#define PW_E2(C) ((C).wi + Cold::w_cap / 2 + Cold::w_cap / 4)
#define PW_W(C)  ((C).wd + Cold::w_cap / 2)
static constexpr int PW_CAP = Cold::w_cap / 4;
ALD_INL bool pair_less(int a1, int a2, int b1, int b2)
{
    uint32_t x1 = H.eid[a1], y1 = H.eid[b1];
    if(x1 != y1) return x1 < y1;
    return H.eid[a2] < H.eid[b2];
}
ALD_FN void sort_pairs(int n)                   // insertion sort by (id(e1), id(e2)); keys are unique
{
    COLD;
    ALD_GLOBAL int32_t *a = PW_E1(C), *b = PW_E2(C); ALD_GLOBAL double *w = PW_W(C);
    for(int i = 1; i < n; i++) {
        int x = a[i], y = b[i]; double z = w[i]; int j = i - 1;
        while(j >= 0 && pair_less(x, y, a[j], b[j])) { a[j + 1] = a[j]; b[j + 1] = b[j]; w[j + 1] = w[j]; j--; }
        a[j + 1] = x; b[j + 1] = y; w[j + 1] = z;
    }
}

// scallop::decompose_vertex_replace (scallop.cc:2009-2142), pe2w = n sorted pairs in the work area
ALD_FN void decompose_vertex_replace(int root, int n)
{
    COLD;
    ALD_GLOBAL int32_t *a = PW_E1(C), *b = PW_E2(C); ALD_GLOBAL double *w = PW_W(C);
    ALD_GLOBAL double *md = C.wd;                 // per-edge sum of its pe2w entries, in pe2w order
    int nloc = 0; ALD_GLOBAL int32_t *loc_e = C.wi;
    for(int e = first_in(root); e >= 0; e = next_in(e)) { H.uidx[e] = (IDX)nloc; loc_e[nloc] = e; md[nloc] = 0; nloc++; }
    for(int e = first_out(root); e >= 0; e = next_out(e)) { H.uidx[e] = (IDX)nloc; loc_e[nloc] = e; md[nloc] = 0; nloc++; }
    ALD_GLOBAL int32_t *mdeg = C.wi + C.w_cap / 4;     // [nloc] pe2w degree m[e]
    for(int i = 0; i < nloc; i++) mdeg[i] = 0;
    const double mw = PRM.min_w;
    for(int i = 0; i < n; i++) {
        if(!(w[i] >= mw - kSMIN)) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; }
        int u1 = H.uidx[a[i]], u2 = H.uidx[b[i]];
        if(mdeg[u1] == 0) md[u1] = w[i]; else md[u1] += w[i];
        if(mdeg[u2] == 0) md[u2] = w[i]; else md[u2] += w[i];
        mdeg[u1]++; mdeg[u2]++;
    }
    for(int i = 0; i < nloc; i++) { if(mdeg[i] == 0) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; } H.ew[loc_e[i]] = md[i]; H.hflag[loc_e[i]] |= HF_PROT; }
    for(int i = 0; i < n; i++) {
        int e1 = a[i], e2 = b[i];
        int m1 = mdeg[H.uidx[e1]], m2 = mdeg[H.uidx[e2]];
        if(free_slots() < 3) { fail(ALD_ST_CAPACITY); return; }
        int e = merge_adjacent_edges(e1, e2, w[i]);
        if(e < 0 || H.status) { if(!H.status) fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
        hs_replace2(e1, e2, e);
        if(m1 == 1) hs_replace1(e1, e);
        if(m2 == 1) hs_replace1(e2, e);
    }
    for(int i = 0; i < nloc; i++) hs_remove(loc_e[i]);
    flush_pending();
    if(H.in_deg[root] != 0 || H.out_deg[root] != 0) { fail(ALD_ST_INVARIANT + ALD_INV_DEGREE); return; }
    H.nz[root] = 0;
}
// scallop::decompose_trivial_vertex (scallop.cc:2144-2167)
ALD_FN void decompose_trivial_vertex(int x)
{
    balance_vertex(x);
    if(H.status) return;
    COLD;
    ALD_GLOBAL int32_t *a = PW_E1(C), *b = PW_E2(C); ALD_GLOBAL double *w = PW_W(C); int n = 0;
    if((int)H.in_deg[x] * (int)H.out_deg[x] > PW_CAP || (int)H.in_deg[x] + (int)H.out_deg[x] > C.w_cap / 4) { fail(ALD_ST_CAPACITY); return; }
    for(int e1 = first_in(x); e1 >= 0; e1 = next_in(e1)) { double w1 = H.ew[e1];
        for(int e2 = first_out(x); e2 >= 0; e2 = next_out(e2)) { double w2 = H.ew[e2]; a[n] = e1; b[n] = e2; w[n] = w1 <= w2 ? w1 : w2; n++; } }
    sort_pairs(n);
    decompose_vertex_replace(x, n);
}

ALD_FN bool resolve_single_trivial_vertex(int i, double jump_ratio);

// scallop::decompose_vertex_extend (scallop.cc:1675-1986); pe2w = n sorted pairs in the work area
ALD_FN void decompose_vertex_extend(int root, int n)
{
    COLD;
    ALD_GLOBAL int32_t *a = PW_E1(C), *b = PW_E2(C); ALD_GLOBAL double *w = PW_W(C);
    int nloc = 0; ALD_GLOBAL int32_t *loc_e = C.wi;
    if((int)H.in_deg[root] + (int)H.out_deg[root] > C.w_cap / 8) { fail(ALD_ST_CAPACITY); return; }
    for(int e = first_in(root); e >= 0; e = next_in(e)) { H.uidx[e] = (IDX)nloc; loc_e[nloc++] = e; }
    int nin = nloc;
    for(int e = first_out(root); e >= 0; e = next_out(e)) { H.uidx[e] = (IDX)nloc; loc_e[nloc++] = e; }
    ALD_GLOBAL int32_t *mdeg = C.wi + C.w_cap / 8;   // [nloc]
    ALD_GLOBAL int32_t *evx = C.wi + C.w_cap / 4;    // [nloc] new vertex of the edge (ev1 / ev2), or -1
    ALD_GLOBAL double *mweight = C.wd;               // [nloc]
    for(int i = 0; i < nloc; i++) { mdeg[i] = 0; evx[i] = -1; mweight[i] = 0; }
    const double mw = PRM.min_w;
    double total_weight = 0;
    for(int i = 0; i < n; i++) {
        if(!(w[i] >= mw - kSMIN)) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; }
        int u1 = H.uidx[a[i]], u2 = H.uidx[b[i]];
        total_weight += w[i];
        if(mdeg[u1] == 0) mweight[u1] = w[i]; else mweight[u1] += w[i];
        if(mdeg[u2] == 0) mweight[u2] = w[i]; else mweight[u2] += w[i];
        mdeg[u1]++; mdeg[u2]++;
    }
    int rlen = C.rpos[root] - C.lpos[root];
    double vertex_weight = C.vw[root] * rlen;
    for(int i = 0; i < nloc; i++) mweight[i] = mweight[i] / total_weight * vertex_weight;
    int m = H.nv - 1, nn = m;
    for(int i = 0; i < nloc; i++) { if(mdeg[i] == 0) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; } if(mdeg[i] >= 2) evx[i] = nn++; }
    int newedges = 0;
    for(int i = 0; i < n; i++) { int u1 = H.uidx[a[i]], u2 = H.uidx[b[i]]; if(mdeg[u1] == 1 && mdeg[u2] == 1) evx[u1] = nn++; else if(mdeg[u1] >= 2 && mdeg[u2] >= 2) newedges++; }
    if(nn + 1 > MAXV || free_slots() < newedges) { fail(ALD_ST_CAPACITY); return; }
    // add vertices and exchange sink (scallop.cc:1793-1806, 2198-2215)
    for(int i = m + 1; i <= nn; i++) { H.in_head[i] = NIL; H.out_head[i] = NIL; H.in_deg[i] = 0; H.out_deg[i] = 0; H.nz[i] = 0; C.vw[i] = 0; C.lpos[i] = 0; C.rpos[i] = 0; C.vtype[i] = -1; C.v2v[i] = -1; }
    for(int i = m; i < nn; i++) H.nz[i] = 1;
    H.nv = nn + 1;
    if(m != nn) {
        C.v2v[nn] = C.v2v[m]; C.lpos[nn] = C.lpos[m]; C.rpos[nn] = C.rpos[m]; C.vtype[nn] = C.vtype[m];
        int guard = MAXE;
        while(first_in(m) >= 0 && guard-- > 0) { int e = first_in(m); move_edge(e, H.es[e], nn); }
        for(int i = m; i < nn; i++) C.v2v[i] = -1;
    }
    for(int i = 0; i < nin; i++) {               // ev1: detach in-edges onto their new vertex
        int k = evx[i]; if(k < 0) continue; int e = loc_e[i];
        int p = C.rpos[H.es[e]];
        move_edge(e, H.es[e], k); C.lpos[k] = p; C.rpos[k] = p; C.vtype[k] = -1; C.vw[k] = 0; C.v2v[k] = -2;
    }
    for(int i = nin; i < nloc; i++) {            // ev2
        int k = evx[i]; if(k < 0) continue; int e = loc_e[i];
        int p = C.lpos[H.et[e]];
        move_edge(e, k, H.et[e]); C.lpos[k] = p; C.rpos[k] = p; C.vtype[k] = -1; C.vw[k] = 0; C.v2v[k] = -2;
    }
    int rv = C.v2v[root];
    for(int i = 0; i < n; i++) {
        int e1 = a[i], e2 = b[i]; int u1 = H.uidx[e1], u2 = H.uidx[e2]; double ww = w[i];
        if(mdeg[u1] == 1 && mdeg[u2] >= 2) {
            borrow_edge_strand(C, e1, e2);
            move_edge(e1, H.es[e1], evx[u2]);
            if(rv >= 0) C.mask[(int64_t)e1 * NW + (rv >> 6)] |= (1ull << (rv & 63));
            C.med[e1] += mweight[u1]; C.mei[e1] += rlen;
        } else if(mdeg[u2] == 1) {
            if(evx[u1] < 0) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
            borrow_edge_strand(C, e2, e1);
            move_edge(e2, evx[u1], H.et[e2]);
            if(rv >= 0) C.mask[(int64_t)e2 * NW + (rv >> 6)] |= (1ull << (rv & 63));
            C.med[e2] += mweight[u2]; C.mei[e2] += rlen;
        } else {
            int z = add_edge(evx[u1], evx[u2]);
            if(z < 0) return;
            H.ew[z] = ww;
            if(!(C.ecount[e1] > 0 && C.ecount[e2] > 0)) { fail(ALD_ST_INVARIANT + ALD_INV_COUNT); return; }
            if(!intersect_samples(e1, e2, z)) return;
            if(C.ecount[z] <= 0) { fail(ALD_ST_INVARIANT + ALD_INV_COUNT); return; }
            C.econf[z] = 0; C.estrand[z] = 0;
            for(int k = 0; k < NW; k++) C.mask[(int64_t)z * NW + k] = 0;
            if(rv >= 0) C.mask[(int64_t)z * NW + (rv >> 6)] |= (1ull << (rv & 63));
            C.med[z] = ww / total_weight * vertex_weight; C.mei[z] = rlen;
            borrow_edge_strand(C, z, e1); borrow_edge_strand(C, z, e2);
            hs_insert_between(e1, e2, z);
            if(H.status) return;
        }
    }
    if(H.in_deg[root] != 0 || H.out_deg[root] != 0) { fail(ALD_ST_INVARIANT + ALD_INV_DEGREE); return; }
    H.nz[root] = 0;
    // scallop.cc:1976-1985 resolve_single_trivial_vertex(k, jump_ratio) on the new vertices: a no-op unless jump_ratio > 1
    double jump = PRM.max_ratio[7];
    if(jump > 1.0) {
        // the reference walks ev1 then ev2, each a std::map keyed by edge id.  The nested decompositions reuse the work area, so
        // the visiting order is parked first in [3/8, 1/2) of wi, which no routine touches.
        ALD_GLOBAL int32_t *order = C.wi + 3 * (C.w_cap / 8); int no = 0;
        for(int part = 0; part < 2; part++) {
            int lo = part == 0 ? 0 : nin, hi = part == 0 ? nin : nloc; int first = no;
            for(int i = lo; i < hi; i++) if(evx[i] >= 0) {
                int k = evx[i]; uint32_t id = H.eid[loc_e[i]]; int j = no - 1;
                while(j >= first && H.eid[mdeg[j]] > id) { order[j + 1] = order[j]; mdeg[j + 1] = mdeg[j]; j--; }      // mdeg is free now: reuse it for the sort keys' edges
                order[j + 1] = k; mdeg[j + 1] = loc_e[i]; no++;
            }
        }
        for(int q = 0; q < no; q++) { resolve_single_trivial_vertex(order[q], jump); if(H.status) return; }
    }
}

// ---------------------------------------------------------------- per-vertex rule evaluation (one vertex per lane)
// scallop::classify_trivial_vertex (scallop.cc:2169-2196); -2 = needs a dominate query on need_e
ALD_INL int classify_trivial_fastpath(int x, bool fast)
{
    int d1 = H.in_deg[x], d2 = H.out_deg[x];
    if(d1 != 1 && d2 != 1) return -1;
    int e1 = first_in(x), e2 = first_out(x);
    if(d1 == 1) { int s = H.es[e1]; if(H.out_deg[s] == 1) return 1; if(fast) { if(!(H.hflag[e1] & HF_OCC)) return 1; return -2; } }
    if(d2 == 1) { int t = H.et[e2]; if(H.in_deg[t] == 1) return 1; if(fast) { if(!(H.hflag[e2] & HF_OCC)) return 1; return -2; } }
    return 2;
}
ALD_FN int classify_trivial_vertex(int x, bool fast)     // scalar version with the dominate queries
{
    int d1 = H.in_deg[x], d2 = H.out_deg[x];
    if(d1 != 1 && d2 != 1) return -1;
    int e1 = first_in(x), e2 = first_out(x);
    if(d1 == 1) { int s = H.es[e1]; if(H.out_deg[s] == 1) return 1; if(fast && hs_dominate(e1, 1)) return 1; }
    if(d2 == 1) { int t = H.et[e2]; if(H.in_deg[t] == 1) return 1; if(fast && hs_dominate(e2, 2)) return 1; }
    return 2;
}
ALD_INL double compute_balance_ratio(int v, bool &ok)    // scallop.cc:2578-2602
{
    double w1 = in_weights(v), w2 = out_weights(v);
    ok = (w1 >= kSMIN) && (w2 >= kSMIN);
    if(w1 >= w2) return w1 / w2; else return w2 / w1;
}
// scallop::resolve_single_trivial_vertex (scallop.cc:1236-1254), scalar
ALD_FN bool resolve_single_trivial_vertex(int i, double jump_ratio)
{
    if(H.in_deg[i] == 0 || H.out_deg[i] == 0) return false;
    if(H.in_deg[i] >= 2 && H.out_deg[i] >= 2) return false;
    if(mixed_strand_vertex(i)) return false;
    if(classify_trivial_vertex(i, false) != 1) return false;
    bool ok; double r = compute_balance_ratio(i, ok);
    if(!ok) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return false; }
    if(r >= jump_ratio) return false;
    trace(OP_TRIVIAL_FAST, i, 0, r);
    decompose_trivial_vertex(i);
    return true;
}
// scallop::compute_smallest_edge + the guards of resolve_smallest_edges (scallop.cc:858-896, 2967-3030)
ALD_INL int eval_smallest(int i, double &r)
{
    if(!H.nz[i]) return -1;
    if(H.in_deg[i] <= 1 || H.out_deg[i] <= 1) return -1;
    int e1 = -1, e2 = -1; double sum1 = 0, sum2 = 0, min1 = DBL_MAX, min2 = DBL_MAX;
    for(int e = first_in(i); e >= 0; e = next_in(e)) { double w = H.ew[e]; sum1 += w; if(w > min1) continue; min1 = w; e1 = e; }
    for(int e = first_out(i); e >= 0; e = next_out(e)) { double w = H.ew[e]; sum2 += w; if(w > min2) continue; min2 = w; e2 = e; }
    if(e1 < 0 || e2 < 0) return -1;
    if(!(sum1 >= kSMIN) || !(sum2 >= kSMIN)) return -3;          // reference assert(sum1 >= SMIN)
    double r1 = min1 / sum1, r2 = min2 / sum2;
    int e; if(r1 < r2) { r = r1; e = e1; } else { r = r2; e = e2; }
    int s = H.es[e], t = H.et[e];
    if(H.out_deg[s] <= 1) return -1;
    if(H.in_deg[t] <= 1) return -1;
    uint8_t f = H.hflag[e];
    if((f & HF_REXT) && (f & HF_LEXT)) return -1;
    if(t == i && (f & HF_REXT)) return -1;
    if(s == i && (f & HF_LEXT)) return -1;
    if(H.any_strand) {
        COLD;
        int z = C.estrand[e];
        if(z >= 1) { int vs[6]; strand_degree(i, vs); if(s == i && vs[0] + vs[z] <= 1) return -1; if(t == i && vs[3] + vs[z + 3] <= 1) return -1; }
    }
    return e;
}

// ---------------------------------------------------------------- wave-level sweeps
// scallop::resolve_broken_vertex (scallop.cc:190-236)
ALD_FN bool resolve_broken_vertex()
{
    const int lane = lane_id();
    int vend = H.nv - 1; int x = -1;
    for(int base = 0; base < vend && x < 0; base += ALD_WAVE) {
        int i = base + lane;
        bool p = (i >= 1 && i < vend) && H.nz[i] && !(H.in_deg[i] >= 1 && H.out_deg[i] >= 1);
        uint64_t m = wballot(p);
        if(m) x = base + ffs64(m);
    }
    if(x < 0) return false;
    if(lane == 0) {
        if(H.in_deg[x] + H.out_deg[x] == 0) fail(ALD_ST_INVARIANT + ALD_INV_OTHER);      // assert(ve.size() >= 1)
        else {
            trace(OP_BROKEN, x, H.in_deg[x] + H.out_deg[x], 0);
            int guard = MAXE;
            while(first_in(x) >= 0 && guard-- > 0) { int e = first_in(x); kill_edge(e); hs_remove(e); }
            while(first_out(x) >= 0 && guard-- > 0) { int e = first_out(x); kill_edge(e); hs_remove(e); }
            H.nz[x] = 0;
        }
    }
    wsync();
    return true;
}

// generic trivial-vertex sweep: mode 0 = resolve_trivial_vertex_fast (scallop.cc:1256-1270: fast=false, type 1, r < jump)
//                               mode 1 = resolve_trivial_vertex(type, fast=true, jump)  (scallop.cc:1180-1234)
ALD_FN bool sweep_trivial(int mode, int type, double jump_ratio)
{
    const int lane = lane_id();
    const bool fast = (mode == 1);
    const double now_thr = (mode == 1) ? 1.02 : jump_ratio;
    int vend = H.nv - 1;                       // snapshot of nonzeroset: vertices created later are not visited
    bool flag = false;
    double best_r = DBL_MAX; int best_v = -1;  // running (ratio, root) of the sequential loop
    bool stopped = false;
    int start = 1;
    if(lane == 0) hs_refresh_flags();
    wsync();
    while(start < vend) {
        int hit = -1; double hit_r = 0;
        for(int base = (start / ALD_WAVE) * ALD_WAVE; base < vend && hit < 0; base += ALD_WAVE) {
            int i = base + lane;
            int cls = -9; double r = 0; bool bad = false;
            if(i >= start && i < vend && H.nz[i] && H.in_deg[i] >= 1 && H.out_deg[i] >= 1 && !(H.in_deg[i] >= 2 && H.out_deg[i] >= 2) && !mixed_strand_vertex(i)) {
                cls = classify_trivial_fastpath(i, fast);
            }
            // lanes that need a dominate query are served one at a time by lane 0 (rare: only edges on phasing paths)
            uint64_t need = wballot(cls == -2);
            while(need) {
                int l = ffs64(need); need &= need - 1;
                int res = 0;
                if(lane == 0) res = classify_trivial_vertex(base + l, fast);
                wsync();
                res = wshfl(res, 0);
                if(lane == l) cls = res;
            }
            bool cand = (cls == type);
            if(cand) { bool ok; r = compute_balance_ratio(i, ok); if(!ok) bad = true; }
            if(wballot(bad)) { if(lane == 0) fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); wsync(); return true; }
            uint64_t now = wballot(cand && r < now_thr);
            // scallop.cc:1222 `if(ratio < jump_ratio) break;`: the first candidate with 1.02 <= r < jump_ratio becomes the root and ends the sweep
            uint64_t stp = (mode == 1) ? wballot(cand && !(r < now_thr) && r < jump_ratio) : 0ull;
            if(stp && (!now || ffs64(stp) < ffs64(now))) {
                int l = ffs64(stp); best_v = base + l; best_r = wshfl(r, l); stopped = true; break;
            }
            // sequential semantics: candidates before the first "now" vertex update (ratio, root); ties go to the LATER vertex
            int lim = now ? ffs64(now) : ALD_WAVE;
            if(mode == 1) {
                bool mine = cand && lane < lim;
                double rr = mine ? r : DBL_MAX; int vv = mine ? i : -1;
                for(int off = ALD_WAVE / 2; off >= 1; off >>= 1) {
                    double r2 = wshfl(rr, lane ^ off); int v2 = wshfl(vv, lane ^ off);
                    bool take = (v2 >= 0) && (vv < 0 || r2 < rr || (r2 == rr && v2 > vv));
                    if(take) { rr = r2; vv = v2; }
                }
                rr = wshfl(rr, 0); vv = wshfl(vv, 0);
                if(vv >= 0 && !(best_r < rr)) { best_r = rr; best_v = vv; }      // if(ratio < r) continue;
            }
            if(now) { hit = base + ffs64(now); hit_r = wshfl(r, ffs64(now)); }
        }
        if(hit < 0 || stopped) break;
        if(lane == 0) {
            trace(mode == 1 ? OP_TRIVIAL_NOW : OP_TRIVIAL_FAST, hit, mode == 1 ? type : 0, hit_r);
            decompose_trivial_vertex(hit);
            hs_refresh_flags();
        }
        wsync();
        flag = true;
        if(H.status) return true;
        start = hit + 1;
    }
    if(flag) return true;
    if(mode == 0) return false;
    if(best_v < 0) return false;
    if(lane == 0) {
        trace(OP_TRIVIAL_BEST, best_v, type, best_r);
        decompose_trivial_vertex(best_v);
    }
    wsync();
    return true;
}

// scallop::resolve_smallest_edges (scallop.cc:844-945)
ALD_FN bool sweep_smallest(double max_ratio)
{
    const int lane = lane_id();
    int vend = H.nv - 1;
    bool flag = false;
    double best_r = max_ratio; int best_e = -1, best_v = -1;
    int start = 1;
    if(lane == 0) hs_refresh_flags();
    wsync();
    while(start < vend) {
        int hit = -1, hit_e = -1; double hit_r = 0;
        for(int base = (start / ALD_WAVE) * ALD_WAVE; base < vend && hit < 0; base += ALD_WAVE) {
            int i = base + lane; double r = 0; int e = -1;
            if(i >= start && i < vend) e = eval_smallest(i, r);
            if(wballot(e == -3)) { if(lane == 0) fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); wsync(); return true; }
            bool cand = e >= 0;
            uint64_t now = wballot(cand && r < 0.01);
            int lim = now ? ffs64(now) : ALD_WAVE;
            bool mine = cand && lane < lim;
            double rr = mine ? r : DBL_MAX; int vv = mine ? i : -1; int ee = mine ? e : -1;
            for(int off = ALD_WAVE / 2; off >= 1; off >>= 1) {
                double r2 = wshfl(rr, lane ^ off); int v2 = wshfl(vv, lane ^ off); int e2 = wshfl(ee, lane ^ off);
                bool take = (v2 >= 0) && (vv < 0 || r2 < rr || (r2 == rr && v2 > vv));
                if(take) { rr = r2; vv = v2; ee = e2; }
            }
            rr = wshfl(rr, 0); vv = wshfl(vv, 0); ee = wshfl(ee, 0);
            if(vv >= 0 && !(best_r < rr)) { best_r = rr; best_v = vv; best_e = ee; }   // if(ratio < r) continue;
            if(now) { int l = ffs64(now); hit = base + l; hit_e = wshfl(e, l); hit_r = wshfl(r, l); }
        }
        if(hit < 0) break;
        if(lane == 0) {
            trace(OP_SMALL_NOW, (int)H.eid[hit_e], hit, hit_r);
            kill_edge(hit_e); hs_remove(hit_e); hs_refresh_flags();
        }
        wsync();
        flag = true;
        start = hit + 1;
    }
    if(flag) return true;
    if(best_e < 0) return false;
    if(lane == 0) {
        trace(OP_SMALLEST, (int)H.eid[best_e], best_v, best_r);
        kill_edge(best_e); hs_remove(best_e);
    }
    wsync();
    return true;
}

// ---------------------------------------------------------------- router (scallop/router.cc), scalar on lane 0
// Results in H.ro_type / H.ro_degree / H.ro_ratio / H.ro_npairs; pe2w pairs (sorted, clamped) in the pair area.
ALD_FN bool router_run(int root, int want_type, int max_degree)
{
    COLD;
    // ---- build_indices (router.cc:225-248)
    int nin = H.in_deg[root], nout = H.out_deg[root], n = nin + nout;
    const int cap = C.w_cap / 2;
    if(n > cap / 8) { fail(ALD_ST_CAPACITY); return false; }
    ALD_GLOBAL int32_t *u2e = C.wi;
    { int k = 0; for(int e = first_in(root); e >= 0; e = next_in(e)) { H.uidx[e] = (IDX)k; u2e[k++] = e; }
      for(int e = first_out(root); e >= 0; e = next_out(e)) { H.uidx[e] = (IDX)k; u2e[k++] = e; } }
    if(mixed_strand_vertex(root)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }     // router.cc:71-76
    // ---- routes from the phasing lists (hyper_set::get_routes, hyper_set.cc:553-571), gathered in the pair area
    const int half = PW_CAP / 2;               // the upper half of the pair area may hold a parked candidate (save_pairs)
    int nr = 0;
    ALD_GLOBAL int32_t *ra = PW_E1(C), *rb = PW_E2(C); ALD_GLOBAL double *rc = PW_W(C);
    {
        int nl = H.hl_n;
        for(int k = 0; k < nl; k++) {
            ALD_GLOBAL int32_t *v = C.hl + C.hl_off[k]; int len = C.hl_len[k]; int c = C.hl_cnt[k];
            for(int i = 0; i + 1 < len; i++) {
                int x = v[i], y = v[i + 1];
                if(x < 0 || y < 0) continue;
                if(H.es[x] == NIL || (int)H.et[x] != root) continue;
                int f = -1;
                for(int j = 0; j < nr; j++) if(ra[j] == x && rb[j] == y) { f = j; break; }
                if(f >= 0) rc[f] += c;
                else { if(nr >= half) { fail(ALD_ST_CAPACITY); return false; } ra[nr] = x; rb[nr] = y; rc[nr] = c; nr++; }
            }
        }
        sort_pairs(nr);                        // MPII order == (id(e1), id(e2)) == creation order of the ug edges
    }
    // ---- arena (sized now that the number of routes is known)
    int maxue = nr + n;                        // + one edge per isolated node
    int o = n;
    ALD_GLOBAL int32_t *udeg = C.wi + o; o += n;
    ALD_GLOBAL int32_t *comp = C.wi + o; o += n;
    ALD_GLOBAL int32_t *queue = C.wi + o; o += n;
    ALD_GLOBAL int32_t *iso = C.wi + o; o += n;          // isolated flag -> econf pending
    ALD_GLOBAL int32_t *us = C.wi + o; o += maxue;
    ALD_GLOBAL int32_t *ut = C.wi + o; o += maxue;
    ALD_GLOBAL int32_t *ualive = C.wi + o; o += maxue;
    if(o > cap || 2 * n + maxue > cap) { fail(ALD_ST_CAPACITY); return false; }
    ALD_GLOBAL double *vw = C.wd, *uw = C.wd + n, *econf = C.wd + n + maxue;
    // ---- build_bipartite_graph (router.cc:250-325)
    int nue = 0;
    for(int i = 0; i < n; i++) { udeg[i] = 0; iso[i] = 0; }
    for(int j = 0; j < nr; j++) {
        int y = rb[j];
        if(H.es[y] == NIL || (int)H.es[y] != root) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }   // assert(e2u.find(e2) != end)
        int s = H.uidx[ra[j]], t = H.uidx[y];
        us[nue] = s; ut[nue] = t; uw[nue] = rc[j]; ualive[nue] = 1; udeg[s]++; udeg[t]++; nue++;
    }
    // isolated vertices attach to the best partner by shared sample abundance (router.cc:1010-1129)
    for(int v = 0; v < nin; v++) {
        if(C.ecount[u2e[v]] == 0) continue;       // "Warning!(count = 0)": not in `left`
        if(udeg[v] != 0) continue;
        int partner = -1; double max_abd = 0.0, sum_abd = 0.0;
        for(int r = nin; r < n; r++) { if(C.ecount[u2e[r]] == 0) continue; double c = common_abd(u2e[v], u2e[r]); sum_abd += c; if(c > max_abd) { max_abd = c; partner = r; } }
        if(partner < 0) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
        us[nue] = v; ut[nue] = partner; uw[nue] = max_abd; ualive[nue] = 1; udeg[v]++; udeg[partner]++; nue++;
        iso[v] = 1; econf[v] = log(max_abd / sum_abd);
    }
    for(int v = nin; v < n; v++) {
        if(C.ecount[u2e[v]] == 0) continue;
        if(udeg[v] != 0) continue;
        int partner = -1; double max_abd = 0.0, sum_abd = 0.0;
        for(int l = 0; l < nin; l++) { if(C.ecount[u2e[l]] == 0) continue; double c = common_abd(u2e[l], u2e[v]); sum_abd += c; if(c > max_abd) { max_abd = c; partner = l; } }
        if(partner < 0) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
        us[nue] = partner; ut[nue] = v; uw[nue] = max_abd; ualive[nue] = 1; udeg[v]++; udeg[partner]++; nue++;
        iso[v] = 1; econf[v] = log(max_abd / sum_abd);
    }
    // ---- classify_plain_vertex (router.cc:116-171)
    H.ro_npairs = 0; H.ro_ratio = 0;
    if(nin == 1 || nout == 1) { H.ro_type = T_TRIVIAL; H.ro_degree = n; return true; }
    for(int i = 0; i < n; i++) if(udeg[i] < 1) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
    int ncomp = 0;
    for(int i = 0; i < n; i++) comp[i] = -1;
    for(int i = 0; i < n; i++) {
        if(comp[i] >= 0) continue;
        int qh = 0, qt = 0; queue[qt++] = i; comp[i] = ncomp;
        while(qh < qt) { int x = queue[qh++]; for(int k = 0; k < nue; k++) { int y = (us[k] == x) ? ut[k] : ((ut[k] == x) ? us[k] : -1); if(y < 0 || comp[y] >= 0) continue; comp[y] = ncomp; queue[qt++] = y; } }
        ncomp++;
    }
    int rtype, rdeg;
    if(ncomp == 1) { rtype = T_UNSPLITTABLE_SINGLE; rdeg = nue - n + 2; }
    else {
        bool b1 = true, b2 = true;             // one_side_connected (router.cc:173-191) -> assert(false)
        for(int i = 1; i < nin; i++) if(comp[i] != comp[0]) b1 = false;
        for(int i = nin + 1; i < n; i++) if(comp[i] != comp[nin]) b2 = false;
        if(b1 || b2) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
        int a = 0, b = 0;
        for(int c = 0; c < ncomp; c++) { int sz = 0; for(int i = 0; i < n; i++) if(comp[i] == c) sz++; if(sz == 1) a++; if(sz >= 2) b++; }
        if(b < 1) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
        rtype = T_SPLITTABLE_PURE; rdeg = b - 1 + (a + 1) / 2;
    }
    H.ro_type = rtype; H.ro_degree = rdeg;
    if(rtype != want_type) return true;
    if(rdeg > max_degree) return true;
    // ---- build() -> thread() (router.cc:193-223, 738-857)
    // compute_balanced_weights_components (router.cc:1248-1275): components by smallest member, members ascending
    for(int i = 0; i < n; i++) vw[i] = 0;
    for(int c = 0; c < ncomp; c++) {
        double sum1 = 0, sum2 = 0;
        for(int i = 0; i < n; i++) { if(comp[i] != c) continue; double wgt = H.ew[u2e[i]]; if(i < nin) sum1 += wgt; else sum2 += wgt; vw[i] = wgt; }
        double r1 = sqrt(sum2 / sum1), r2 = sqrt(sum1 / sum2);
        for(int i = 0; i < n; i++) { if(comp[i] != c) continue; if(i < nin) vw[i] *= r1; else vw[i] *= r2; }
    }
    double weight_sum = 0;
    for(int i = 0; i < n; i++) weight_sum += vw[i];
    ALD_GLOBAL int32_t *pa = PW_E1(C), *pb = PW_E2(C); ALD_GLOBAL double *pwt = PW_W(C); int np = 0;
    int live = nue;
    int guard = 4 * (nue + n) + 8;
    while(guard-- > 0) {
        // thread_leaf (router.cc:859-897): edges in creation order
        bool b = false;
        for(int k = 0; k < nue && !b; k++) {
            if(!ualive[k]) continue;
            int s = us[k], t = ut[k];
            if(s >= t) { int q = s; s = t; t = q; }
            if(vw[s] < -0.5) continue;
            if(vw[t] < -0.5) continue;
            int x = -1, y = -1;
            if(udeg[s] == 1 && vw[s] <= vw[t]) { x = s; y = t; }
            else if(udeg[t] == 1 && vw[t] <= vw[s]) { x = t; y = s; }
            if(x < 0) continue;
            if(np >= half) { fail(ALD_ST_CAPACITY); return false; }
            pa[np] = u2e[s]; pb[np] = u2e[t]; pwt[np] = vw[x]; np++;
            for(int q = 0; q < nue; q++) if(ualive[q] && (us[q] == x || ut[q] == x)) { ualive[q] = 0; udeg[us[q]]--; udeg[ut[q]]--; live--; }   // clear_vertex
            vw[y] -= vw[x]; vw[x] = -1; b = true;
        }
        if(b) continue;
        // thread_turn (router.cc:899-936)
        int x = -1;
        for(int k = 0; k < n; k++) { if(vw[k] < -0.5) continue; if(udeg[k] <= 1) continue; if(x != -1 && vw[k] > vw[x]) continue; x = k; }
        if(x == -1) break;
        double sum = 0;
        // out_edges(x) order: by the other endpoint, then creation (pairs are unique)
        for(int t = 0; t < n; t++) for(int k = 0; k < nue; k++) { if(!ualive[k]) continue; int y = (us[k] == x) ? ut[k] : ((ut[k] == x) ? us[k] : -1); if(y != t) continue; sum += uw[k]; if(!(vw[t] >= vw[x])) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; } }
        for(int t = 0; t < n; t++) for(int k = 0; k < nue; k++) {
            if(!ualive[k]) continue; int y = (us[k] == x) ? ut[k] : ((ut[k] == x) ? us[k] : -1); if(y != t) continue;
            double wgt = vw[x] * uw[k] / sum;
            if(np >= half) { fail(ALD_ST_CAPACITY); return false; }
            if(x < t) { pa[np] = u2e[x]; pb[np] = u2e[t]; } else { pa[np] = u2e[t]; pb[np] = u2e[x]; }
            pwt[np] = wgt; np++;
            vw[t] -= wgt;
        }
        vw[x] = -1;
        for(int q = 0; q < nue; q++) if(ualive[q] && (us[q] == x || ut[q] == x)) { ualive[q] = 0; udeg[us[q]]--; udeg[ut[q]]--; live--; }
    }
    if(live != 0) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
    double weight_remain = 0;
    for(int i = 0; i < n; i++) { if(vw[i] <= 0) continue; weight_remain += vw[i]; }
    H.ro_ratio = weight_remain / weight_sum;
    for(int i = 0; i < n; i++) if(iso[i]) C.econf[u2e[i]] += econf[i];     // router.cc:849-855: side effect of every build()
    sort_pairs(np);
    const double mw = PRM.min_w;
    for(int i = 0; i < np; i++) if(pwt[i] < mw) pwt[i] = mw;                // router.cc:217-220
    H.ro_npairs = np;
    return true;
}
// the pair area holds PW_CAP pairs; the upper half parks the best candidate while a sweep goes on
ALD_FN void save_pairs(int n)
{
    COLD;
    const int h = PW_CAP / 2;
    if(n > h) { fail(ALD_ST_CAPACITY); return; }
    ALD_GLOBAL int32_t *a = PW_E1(C), *b = PW_E2(C); ALD_GLOBAL double *w = PW_W(C);
    for(int i = 0; i < n; i++) { a[h + i] = a[i]; b[h + i] = b[i]; w[h + i] = w[i]; }
}
ALD_FN void restore_pairs(int n)
{
    COLD;
    const int h = PW_CAP / 2;
    ALD_GLOBAL int32_t *a = PW_E1(C), *b = PW_E2(C); ALD_GLOBAL double *w = PW_W(C);
    for(int i = 0; i < n; i++) { a[i] = a[h + i]; b[i] = b[h + i]; w[i] = w[h + i]; }
}
// scallop::resolve_unsplittable_vertex (scallop.cc:1004-1060)
ALD_FN bool sweep_unsplittable(int type, int degree, double max_ratio)
{
    const int lane = lane_id();
    int vend = H.nv - 1;
    bool flag = false;
    int root = -1; double ratio = max_ratio; int best_np = 0;      // meaningful on lane 0 only
    // The sweep is sequential in the reference: a decomposition (and, for jump_ratio > 1, the trivial decompositions nested in
    // it) can make a LATER vertex newly eligible, so the next candidate is looked up again after every action.
    int cur = 1;
    while(cur < vend) {
        int i = -1;
        for(int base = (cur / ALD_WAVE) * ALD_WAVE; base < vend && i < 0; base += ALD_WAVE) {
            int i0 = base + lane;
            uint64_t m = wballot(i0 >= cur && i0 < vend && H.nz[i0] && H.in_deg[i0] >= 2 && H.out_deg[i0] >= 2);
            if(m) i = base + ffs64(m);
        }
        if(i < 0) break;
        int act = 0;
        if(lane == 0) {
            if(router_run(i, type, degree) && H.ro_type == type && H.ro_degree <= degree) {
                double rr = H.ro_ratio;
                if(rr < 0.01) {
                    trace(OP_UNSPLIT_NOW, i, type, rr);
                    decompose_vertex_extend(i, H.ro_npairs);
                    act = 1;
                } else if(!(rr > ratio)) {
                    root = i; ratio = rr; best_np = H.ro_npairs;
                    save_pairs(best_np);
                }
            }
        }
        wsync();
        act = wshfl(act, 0);
        if(H.status) return true;
        if(act) flag = true;
        cur = i + 1;
    }
    if(flag) return true;
    root = wshfl(root, 0);
    if(root < 0) return false;
    if(lane == 0) {
        restore_pairs(best_np);
        trace(OP_UNSPLIT_BEST, root, type, ratio);
        decompose_vertex_extend(root, best_np);
    }
    wsync();
    return true;
}

// ---------------------------------------------------------------- paths out
// scallop::collect_path (scallop.cc:2766-2834), scalar on lane 0
ALD_FN void collect_path(int e)
{
    COLD;
    ALD_GLOBAL const KernelArgs *A = H.args;
    int n = C.v2v[H.nv - 1];
    int cnt = 0, mi = 0; bool empty = false;
    for(int k = 0; k < NW; k++) { uint64_t mk = C.mask[(int64_t)e * NW + k]; while(mk) { int b = ffs64(mk); mk &= mk - 1; int x = k * 64 + b; cnt++; mi += C.rpos[x] - C.lpos[x]; if(C.vtype[x] == K_EMPTY_VERTEX) empty = true; } }
    if(C.mei[e] != mi || cnt == 0) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
    if(C.vtype[0] == K_EMPTY_VERTEX || C.vtype[n] == K_EMPTY_VERTEX) empty = true;
    if(!empty) {
        int nvp = cnt + 2;
        unsigned long long words = (unsigned long long)(REC_HDR_WORDS + nvp + ((REC_HDR_WORDS + nvp) & 1));
        unsigned long long o = atomic_add_u64(A->out.pool_used, words);
        if(o + words > A->out.pool_cap) { fail(ALD_ST_CAPACITY); return; }
        ALD_GLOBAL uint32_t *r = A->out.pool + o;
        int st = '.';
        if(C.estrand[e] == 1) st = '+';
        if(C.estrand[e] == 2) st = '-';
        if(st == '.') st = H.gstrand;
        r[0] = (uint32_t)H.g; r[1] = (uint32_t)H.n_paths; r[2] = (uint32_t)nvp; r[3] = (uint32_t)mi; r[4] = (uint32_t)C.ecount[e]; r[5] = (uint32_t)st | ((uint32_t)(A->attempt & 0xFF) << 8);
        ALD_GLOBAL double *d = (ALD_GLOBAL double*)(r + 6);
        d[0] = H.ew[e]; d[1] = C.eabd[e]; d[2] = exp(C.econf[e]); d[3] = C.med[e];
        ALD_GLOBAL uint32_t *pv = r + REC_HDR_WORDS; int w = 0;
        pv[w++] = 0;
        for(int k = 0; k < NW; k++) { uint64_t mk = C.mask[(int64_t)e * NW + k]; while(mk) { int b = ffs64(mk); mk &= mk - 1; pv[w++] = (uint32_t)(k * 64 + b); } }
        pv[w++] = (uint32_t)n;
        if((REC_HDR_WORDS + nvp) & 1) pv[w] = 0;
        if(tracing()) { int save = H.n_iters; trace(OP_COLLECT, (int)H.eid[e], nvp, H.ew[e]); H.n_iters = save; }
        H.n_paths++;
    }
    H.hflag[e] = 0;
    kill_edge(e);
}
// scallop::collect_existing_st_paths (scallop.cc:2742-2752): ascending edge index == ascending creation id
ALD_FN void collect_existing_st_paths()
{
    int sink = H.nv - 1;
    // the source's out-list is ordered by (target, id): the edges to the sink are its tail, already ascending in id
    int e = first_out(0); int guard = MAXE;
    while(e >= 0 && guard-- > 0) { int nx = next_out(e); if((int)H.et[e] == sink) { collect_path(e); if(H.status) return; } e = nx; }
}
// splice_graph::compute_maximum_path_w (splice_graph.cc:819-885) + directed_graph::topological_sort (directed_graph.cc:420-451)
// path edges -> PW_E1 area, length -> H.tmp0
ALD_FN double compute_maximum_path()
{
    COLD;
    int n = H.nv;
    ALD_GLOBAL int32_t *vd = C.wi, *q = C.wi + n, *back = C.wi + 2 * n; ALD_GLOBAL double *table = C.wd;
    ALD_GLOBAL int32_t *path = PW_E1(C);
    int qt = 0;
    for(int i = 0; i < n; i++) { int d = H.in_deg[i]; vd[i] = d; if(d == 0) q[qt++] = i; table[i] = -1; back[i] = -1; }
    int k = 0;
    while(k < qt) { int x = q[k++]; for(int e = first_out(x); e >= 0; e = next_out(e)) { int t = H.et[e]; if(--vd[t] == 0) q[qt++] = t; } }
    H.tmp0 = 0;
    if(qt != n) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return -1; }
    int ssi = -1, tti = -1;
    for(int i = 0; i < n; i++) { if(q[i] == 0) ssi = i; if(q[i] == n - 1) tti = i; }
    table[0] = DBL_MAX;
    for(int ii = ssi + 1; ii <= tti; ii++) {
        int i = q[ii];
        if(H.in_deg[i] + H.out_deg[i] == 0) continue;
        double max_abd = 0; int max_edge = -1;
        for(int e = first_in(i); e >= 0; e = next_in(e)) {
            int s = H.es[e];
            double ts = table[s];
            if(ts <= -1) continue;
            double xw = H.ew[e];
            double ww = xw < ts ? xw : ts;
            if(ww >= max_abd) { max_abd = ww; max_edge = e; }
        }
        if(max_edge < 0) continue;
        back[i] = max_edge; table[i] = max_abd;
    }
    int plen = 0;
    int x = n - 1;
    while(plen < n) { int e = back[x]; if(e < 0) break; path[plen++] = e; x = H.es[e]; }
    for(int i = 0; i < plen / 2; i++) { int t = path[i]; path[i] = path[plen - 1 - i]; path[plen - 1 - i] = t; }
    H.tmp0 = plen;
    return table[n - 1];
}
// scallop::greedy_decompose (scallop.cc:2874-2897) + split_merge_path (scallop.cc:2230-2240)
ALD_FN void greedy_decompose()
{
    COLD;
    bool any = false;
    for(int i = 0; i < H.nv && !any; i++) if(H.out_deg[i]) any = true;
    if(!any) return;
    for(int rep = 0; rep < 2; rep++) for(int i = 1; i < H.nv - 1; i++) { balance_vertex(i); if(H.status) return; }
    if(3 * H.nv > C.w_cap / 2 || H.nv > PW_CAP) { fail(ALD_ST_CAPACITY); return; }
    ALD_GLOBAL int32_t *path = PW_E1(C);
    const double min_cov = PRM.min_cov;
    int guard = 4 * MAXE;
    while(guard-- > 0) {
        double w = compute_maximum_path();
        int plen = H.tmp0;
        if(H.status) return;
        if(w < 0) break;
        if(w <= min_cov) break;
        if(tracing()) { int save = H.n_iters; trace(OP_GREEDY, plen, 0, w); H.n_iters = save; }
        if(plen == 0) break;
        if(free_slots() < 3) { fail(ALD_ST_CAPACITY); return; }
        int ee = split_edge(path[0], w);
        for(int i = 1; i < plen && ee >= 0 && !H.status; i++) {
            if(free_slots() < 3) { fail(ALD_ST_CAPACITY); return; }
            int x = split_edge(path[i], w);
            if(x < 0) { ee = -1; break; }
            ee = merge_adjacent_equal_edges(ee, x);
        }
        if(H.status) return;
        if(ee < 0) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
        collect_path(ee);
        if(H.status) return;
    }
}

// ---------------------------------------------------------------- load: packed wire arrays -> working state (wave-parallel)
ALD_FN bool load_graph()
{
    COLD;
    const int lane = lane_id();
    ALD_GLOBAL const KernelArgs *A = H.args;
    const int g = H.g;
    int V = A->in.g_nv[g], E = A->in.g_ne[g], NP = A->in.g_np[g];
    int64_t ov = A->in.off_v[g], ovo = ov + g, oe = A->in.off_e[g], oeo = oe + g, os = A->in.off_s[g], op = A->in.off_p[g], opo = op + g, opv = A->in.off_pv[g];
    if(lane == 0) {
        H.V0 = V; H.gstrand = (int)(unsigned char)A->in.graph_strand[g];
        H.nv = V; H.next_id = E; H.slot_hw = E; H.free_head = -1; H.free_cnt = 0; H.pend_head = -1; H.status = 0; H.any_strand = 0; H.hs_dirty = 1;
        H.n_paths = 0; H.n_iters = 0; H.n_trace = 0; H.sp_used = 0; H.hl_used = 0; H.hl_n = 0;
    }
    wsync();
    if(V + 1 > MAXV || E > MAXE || V > NW * 64 || V < 2) { if(lane == 0) H.status = ALD_ST_CAPACITY; wsync(); return false; }
    int64_t ns = A->in.edge_sample_offset[oeo + E];
    if(ns > (int64_t)C.sp_cap) { if(lane == 0) H.status = ALD_ST_CAPACITY; wsync(); return false; }
    ALD_GLOBAL const int32_t *vo = A->in.vertex_offset + ovo, *io = A->in.in_offset + ovo, *ie = A->in.in_edge + oe;
    for(int i = lane; i < V; i += ALD_WAVE) {
        int o0 = vo[i], o1 = vo[i + 1], i0 = io[i], i1 = io[i + 1];
        H.out_head[i] = o1 > o0 ? (IDX)o0 : NIL; H.out_deg[i] = (IDX)(o1 - o0);
        H.in_head[i] = i1 > i0 ? (IDX)ie[i0] : NIL; H.in_deg[i] = (IDX)(i1 - i0);
        H.nz[i] = (i >= 1 && i < V - 1 && (o1 - o0) + (i1 - i0) > 0) ? 1 : 0;
        for(int k = o0; k < o1; k++) { H.es[k] = (IDX)i; H.onx[k] = (k + 1 < o1) ? (IDX)(k + 1) : NIL; }
        for(int k = i0; k < i1; k++) { H.inx[ie[k]] = (k + 1 < i1) ? (IDX)ie[k + 1] : NIL; }
        C.vw[i] = A->in.vertex_weight[ov + i]; C.lpos[i] = A->in.vertex_lpos[ov + i]; C.rpos[i] = A->in.vertex_rpos[ov + i];
        C.vtype[i] = A->in.vertex_type[ov + i]; C.v2v[i] = i;
    }
    bool strand = false;
    ALD_GLOBAL const int32_t *so = A->in.edge_sample_offset + oeo;
    for(int k = lane; k < E; k += ALD_WAVE) {
        H.et[k] = (IDX)A->in.edge_target[oe + k]; H.ew[k] = A->in.edge_weight[oe + k]; H.eid[k] = (uint32_t)k; H.hflag[k] = 0;
        uint8_t st = A->in.edge_strand[oe + k]; C.estrand[k] = st; if(st) strand = true;
        C.med[k] = 0; C.mei[k] = 0; C.econf[k] = 0; C.eabd[k] = A->in.edge_abd[oe + k];
        C.sp_off[k] = (uint32_t)so[k]; C.sp_len[k] = (uint32_t)(so[k + 1] - so[k]); C.ecount[k] = so[k + 1] - so[k];
        for(int q = 0; q < NW; q++) C.mask[(int64_t)k * NW + q] = 0;
    }
    for(int64_t k = lane; k < ns; k += ALD_WAVE) { C.sp_id[k] = A->in.sample_id[os + k]; C.sp_abd[k] = A->in.sample_abd[os + k]; }
    uint64_t sb = wballot(strand);
    if(sb && lane == 0) H.any_strand = 1;
    wsync();
    if(lane == 0) {
        H.sp_used = (uint32_t)ns;
        // hyper_set::build_edges (hyper_set.cc:323-354): keep lists with count >= 2, >= 2 edges, every consecutive pair an edge;
        // directed_graph::edge(s,t) returns the NEWEST parallel edge (directed_graph.cc:60-76)
        ALD_GLOBAL const int32_t *po = A->in.phasing_offset + opo; int nl = 0; uint32_t used = 0;
        for(int p = 0; p < NP && H.status == 0; p++) {
            int c = A->in.phasing_count[op + p]; int a = po[p], b = po[p + 1]; int len = b - a;
            if(c <= 1 || len <= 1) continue;
            uint32_t capk = (uint32_t)(len - 1) * 2u + 4u;
            if(nl >= C.hl_maxlists || used + capk > C.hl_cap) { H.status = ALD_ST_CAPACITY; break; }
            bool ok = true;
            for(int k = 0; k + 1 < len && ok; k++) {
                int s = A->in.phasing_vertex[opv + a + k], t = A->in.phasing_vertex[opv + a + k + 1];
                if(!(s < t) || s < 0 || t >= V) { H.status = ALD_ST_INVARIANT + ALD_INV_OTHER; ok = false; break; }
                int best = -1;
                for(int e = first_out(s); e >= 0; e = next_out(e)) { int tt = H.et[e]; if(tt == t) best = e; else if(tt > t) break; }
                if(best < 0) ok = false; else C.hl[used + k] = best;
            }
            if(!ok || len - 1 < 2) continue;
            C.hl_off[nl] = (int32_t)used; C.hl_len[nl] = len - 1; C.hl_capk[nl] = (int32_t)capk; C.hl_cnt[nl] = c; used += capk; nl++;
        }
        H.hl_used = used; H.hl_n = nl;
    }
    wsync();
    return H.status == 0;
}

ALD_FN void finish_graph()
{
    if(lane_id() == 0) {
        ALD_GLOBAL const KernelArgs *A = H.args; const int g = H.g;
        A->out.status[g] = H.status; A->out.n_paths[g] = (H.status == 0 || H.status == ALD_ST_SKIPPED_LARGE) ? H.n_paths : 0; A->out.n_iters[g] = H.n_iters;
        if(A->out.trace_cap > 0) A->out.trace_n[g] = H.n_trace;
    }
    wsync();
}

// ---------------------------------------------------------------- scallop::assemble (scallop.cc:38-188)
ALD_FN void run_graph()
{
    if(!load_graph()) { finish_graph(); return; }
    bool skipped = false;
    const double r_triv = PRM.max_ratio[7], r_small = PRM.max_ratio[0], r_single = PRM.max_ratio[5], r_pure = PRM.max_ratio[4];
    const int max_exons = PRM.max_num_exons;
    int guard = 64 * MAXE;                     // every successful rule consumes an edge or a vertex; far above any real count
    while(guard-- > 0) {
        if(H.nv > max_exons) { skipped = true; break; }
        if(H.status) break;
        if(resolve_broken_vertex()) continue;
        if(r_triv > 1.0) { if(sweep_trivial(0, 1, r_triv)) continue; }     // resolve_trivial_vertex_fast: a no-op for jump_ratio <= 1 (r >= 1 always)
        if(sweep_trivial(1, 1, r_triv)) continue;
        if(sweep_smallest(r_small)) continue;
        if(sweep_unsplittable(T_UNSPLITTABLE_SINGLE, 1, 0.01)) continue;
        if(sweep_unsplittable(T_SPLITTABLE_PURE, 1, 0.01)) continue;
        if(sweep_unsplittable(T_UNSPLITTABLE_SINGLE, INT_MAX, r_single)) continue;
        if(sweep_unsplittable(T_SPLITTABLE_PURE, INT_MAX, r_pure)) continue;
        if(sweep_unsplittable(T_UNSPLITTABLE_SINGLE, INT_MAX, DBL_MAX)) continue;
        if(sweep_unsplittable(T_SPLITTABLE_PURE, INT_MAX, DBL_MAX)) continue;
        if(sweep_trivial(1, 2, r_triv)) continue;
        break;
    }
    if(lane_id() == 0 && H.status == 0) {
        if(guard <= 0) fail(ALD_ST_INVARIANT + ALD_INV_OTHER);
        else {
            collect_existing_st_paths();
            if(H.status == 0) greedy_decompose();
            if(H.status == 0 && skipped) H.status = ALD_ST_SKIPPED_LARGE;
        }
    }
    wsync();
    finish_graph();
}

// one wave's whole life: pull graphs of this size class from the shared counter until the class is drained
ALD_INL void wave_main(ALD_GLOBAL const KernelArgs *A, int block)
{
    if(lane_id() == 0) { H.args = A; H.cold = A->slabs + (uint64_t)block * A->slab_stride; }
    wsync();
    while(true) {
        if(lane_id() == 0) H.s_next = atomic_add_i32(A->counter, 1);
        wsync();
        int k = H.s_next;
        wsync();
        if(k >= A->n_work) break;
        if(lane_id() == 0) H.g = A->work[k];
        wsync();
        run_graph();
    }
}

#undef H
#undef COLD
#undef PRM
} // namespace ALD_CLASS_NS
