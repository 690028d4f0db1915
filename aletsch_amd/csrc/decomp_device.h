// decomp_device.h -- the per-graph decomposition engine executed by ONE 64-lane wavefront.
//
// One wavefront owns one splice graph.  The graph's hot state (sorted adjacency lists, endpoints, creation ids, FP64 weights,
// degrees: struct Hot) and the wave's context + exchange scratch (struct HotCtx) live in LDS; cold per-edge / per-vertex state
// (coverage bookkeeping, sample support, phasing lists, path bitmasks) lives in the wave's private HBM slab, laid out at compile
// time.  The slab-resident classes (the catch-all and the twins) keep Hot in the slab as well, HotCtx stays in LDS.
//
// Who runs what:
//   * rule sweeps (scallop.cc:844-945, 1180-1234): one vertex per lane, ballots, one DPP arg-min per sweep;
//   * the trivial decomposition (scallop.cc:2144-2576): the whole wave, one fan edge per lane (star_wave_body);
//   * edge removal of the smallest-edge rule: two lanes, one per adjacency list (kill_edge_wave);
//   * collecting finished source->sink paths: one path per lane (collect_existing_st_paths);
//   * the router's inputs: gathered by the wave (router_prepare), the router itself (router.cc), the extend decomposition
//     (scallop.cc:1675-1986), phasing-list edits and the greedy tail: scalar on lane 0, values marked wave-uniform (uni()) so that
//     integer work runs on the SALU.  The wave-level drivers are inlined into the kernel root (one call site each); only cold
//     paths are real calls.
// Lanes hand values to each other through LDS (wsync() = wavefront-scope fence, no s_barrier: a workgroup is one wave), so the
// single-lane emulation below runs every phase as a loop.
//
// Every function cites the reference file:line whose behaviour it reproduces.  Ordering rules are the canonical ones of SURVEY.md
// Appendix A: edge "pointer order" == creation id (the rank handed across the ABI, else CSR position).
//
// Build modes (one translation unit per size class, -DALD_CLASS_ID=k):
//   hipcc --offload-arch=gfx950   : the product (ALD_WAVE == 64).
//   g++ -DALD_EMU                 : single-lane emulation, compiled ONLY by tests/kernel_emu to check the algorithm against the
//                                   oracle on a CPU-only box.  Never part of the product library.
#pragma once
#include "decomp_common.h"
#ifdef ALD_EMU
#include <cstdio>
#include <cstdlib>
#endif

#ifndef ALD_CLASS_ID
#error "compile with -DALD_CLASS_ID=<0..13> (one translation unit per size class)"
#endif
#if defined(__HIP__) || defined(__clang__)
  #define ALD_UNROLL _Pragma("unroll")
#else
  #define ALD_UNROLL _Pragma("GCC unroll 16")
#endif
#define ALD_CAT2(a, b) a##b
#define ALD_CAT(a, b) ALD_CAT2(a, b)
// Two builds of every class: the PLAIN kernel (staged graphs only: no call to the device pre-steps anywhere in it, the rule cascade
// inlined into the kernel root) and, with -DALD_RAW_VARIANT, the kernel that also takes RAW graphs (pre_assemble_device reached through
// load_graph, run_graph a call).  The host sends a class's raw graphs to the second one; a batch without raw graphs never launches it.
#ifdef ALD_RAW_VARIANT
#define ALD_CLASS_NS ALD_CAT(ald_r, ALD_CLASS_ID)
#else
#define ALD_CLASS_NS ALD_CAT(ald_c, ALD_CLASS_ID)
#endif

namespace ALD_CLASS_NS {
using namespace ald;
#ifdef ALD_EMU_COUNT
static long g_cnt_router = 0, g_cnt_unsweep = 0, g_cnt_star[34] = {0}, g_cnt_rdeg[34] = {0}, g_cnt_build = 0, g_cnt_pre[3] = {0}, g_cnt_fix[8] = {0}, g_cnt_fix_declined = 0;
struct CntPrinter { ~CntPrinter() { if(g_cnt_unsweep) { fprintf(stderr, "[emu-count] class %d: unsplittable sweeps %ld router runs %ld (of which reach build(): %ld; prepared at level 0/1/2: %ld %ld %ld)\n", ALD_CLASS_ID, g_cnt_unsweep, g_cnt_router, g_cnt_build, g_cnt_pre[0], g_cnt_pre[1], g_cnt_pre[2]);
    fprintf(stderr, "[emu-count]   stars by fan size:"); for(int i = 0; i < 34; i++) if(g_cnt_star[i]) fprintf(stderr, " %d:%ld", i, g_cnt_star[i]); fprintf(stderr, "\n");
    fprintf(stderr, "[emu-count]   stars through the fixed-size form:"); for(int i = 0; i < 8; i++) if(g_cnt_fix[i]) fprintf(stderr, " %d:%ld", i, g_cnt_fix[i]); fprintf(stderr, " declined:%ld\n", g_cnt_fix_declined);
    fprintf(stderr, "[emu-count]   router runs by degree:"); for(int i = 0; i < 34; i++) if(g_cnt_rdeg[i]) fprintf(stderr, " %d:%ld", i, g_cnt_rdeg[i]); fprintf(stderr, "\n"); } } }; static CntPrinter g_cnt_printer;
#endif

enum { MAXV = ClassDims<ALD_CLASS_ID>::MAXV, MAXE = ClassDims<ALD_CLASS_ID>::MAXE, NW = ClassDims<ALD_CLASS_ID>::NW };
typedef uint16_t IDX;
static constexpr IDX NIL = (IDX)0xFFFF;
#if ALD_CLASS_ID == ALD_HUGE_CLASS
typedef uint32_t EID;                           // creation ids: a graph of the largest class makes more than 65 535 edges in its life
enum { EID_LIMIT = 0x7FFFFFF0 };
#else
typedef uint16_t EID;
enum { EID_LIMIT = 0xFFF0 };
#endif
typedef ColdLayoutT<MAXV, MAXE, NW, 0> CL;
#ifndef ALD_KEEP
  /* the per-vertex records of the two evaluation sweeps kept in the slab between sweeps (sweep_smallest, scan_trivial): for the classes whose
     graph arrays live in the slab -- there a list step is a round trip to L2 and evaluating a vertex costs tens of them: the twins run the
     385..512-vertex band of cfg3 in 66.6 ms instead of 79.5 (profiles/r04/r_*).  The LDS classes LOSE with it (a chunk evaluated out of
     LDS costs less than the marks, the dense pass and the trip to the slab: 16.1 -> 17.4, 40.6 -> 42.5, 86.3 -> 89.7 ms for the three
     upper bands), classes 0 / 1 evaluate their one or two chunks at every sweep anyway. */
  #define ALD_KEEP (ALD_CLASS_ID >= ALD_FIRST_GLOBAL_CLASS ? 1 : 0)
#endif
#ifndef ALD_KEEP_TRIV
  #define ALD_KEEP_TRIV ALD_KEEP               /* the trivial-scan records as well (experiments: -DALD_KEEP=1 -DALD_KEEP_TRIV=0 keeps the smallest-edge evaluations only) */
#endif         // (no row pool in this form: decomp_device_rows.h is the build that has one)
enum { LP = 16, ARENA_I = 96, ARENA_D = 48, SCR_I = 4 * LP + ARENA_I, SCR_D = 2 * LP + ARENA_D };   // LDS scratch geometry (ints / doubles)

// ---------------------------------------------------------------------------------------------
// hot state: ONE instance per workgroup (= per wavefront)
// ---------------------------------------------------------------------------------------------
struct Hot {
    struct alignas(8) Link { IDX es, et, inx, onx; };      // endpoints (es == NIL <=> slot dead) + next edge in the target's in-list / the source's
                                                // out-list (both sorted); one 8-byte word so that a list step is ONE LDS round trip
    struct alignas(16) EdgeHot { double w; Link lk; };     // splice_graph::ewrt + the links: 16 bytes, so a walk that needs the weight too
    EdgeHot  ed[MAXE];                          // (sums, balance, smallest-edge evaluation) still makes one LDS access per step
    EID      eid[MAXE];                         // creation id == scallop edge index (16 bits; ids beyond -> the graph moves up a class; 32 bits in the largest class)
    // the vertex record: list heads and degrees in ONE 8-byte word.  An LDS instruction occupies the CU's LDS pipeline for ~4.5 cycles
    // whether it moves two bytes for one lane or eight for sixty-four (profiles/r03/zf_lds_issue_microbench.txt), and twenty resident
    // waves keep that pipeline more than half busy: the sweeps read degrees and heads together, as one access instead of up to four.
    struct alignas(8) VertexHot { IDX in_head, out_head, in_deg, out_deg; };
    VertexHot vx[MAXV];
    uint8_t  nz[MAXV];                          // bit 0: scallop::nonzeroset membership; bits 1..5: router class of the vertex on the CURRENT graph (NZ_MEMO_*)
    uint8_t  hflag[MAXE];                       // HF_* (phasing occupancy / extend flags / protect)
};
// Everything else a wave keeps about the graph at hand: scalars, the sweep state, the scratch the lanes exchange values through.
// ALWAYS in LDS -- also for the classes whose graph arrays live in the wave's HBM slab (the twins, the catch-all class): these ~1.7 KB
// are what every phase boundary and every branch of the rule cascade reads, and a round trip to L2 / the Infinity Cache for each of
// them was a large part of what made a graph 3-4 times slower there than in the LDS form.
struct HotCtx {
    // wave-uniform context
    ALD_GLOBAL uint8_t *cold;                   // this wave's HBM slab
    ALD_GLOBAL const KernelArgs *args;
    double   ro_ratio;                          // router result
    int32_t  ro_type, ro_degree, ro_npairs, tmp0;
    // Classes 2 and up (ALD_KEEP) keep the smallest-edge evaluation of every vertex (ratio, edge) in the wave's slab BETWEEN the sweeps;
    // what changed since it was taken is recorded here by the routines that change it: a bit per vertex whose lists / weights / membership
    // changed -- or whose evaluation looks at a neighbour's degree that crossed the 1 | 2 line (the guards of resolve_smallest_edges) --,
    // and "everything" for the phasing flags and for new vertices.  (Classes 0 / 1 evaluate their one or two chunks at every sweep.)
    int32_t  ev_all;
#if ALD_KEEP
    uint32_t ev_dirty[(MAXV + 31) / 32];
    // the same for the trivial-vertex scan (class + balance ratio of every vertex, scan_trivial): it is brought up to date at other moments
    // than the smallest-edge evaluations, so it has marks of its own -- set by the same calls
  #if ALD_KEEP_TRIV
    int32_t  tv_all; uint32_t tv_dirty[(MAXV + 31) / 32];
  #endif
#endif
    int32_t  g, V0, gstrand;
    int32_t  nv, next_id, slot_hw, free_head, free_cnt, status, any_strand, hs_dirty, n_paths, n_iters, n_trace;
    uint32_t sp_used, hl_used; int32_t hl_n;
    int32_t  s_next;
    // parameters cached once per wave (saves a dependent HBM/L2 round per use)
    double   p_min_w, p_min_cov, p_ratio[8]; int32_t p_max_exons, p_trace_cap;
    // small-case scratch: the pe2w pair area (+ a parked copy) and the router / decomposition arenas live here whenever the
    // vertex at hand is small (the common case); larger cases use the slab's work arrays
    int32_t  pw_lds, park_lds;
    // the sink keeps ONE physical index (V0-1) for the whole run: the reference re-numbers it to stay last whenever
    // decompose_vertex_extend appends vertices (exchange_sink, scallop.cc:2198-2215); here order comparisons map it to +inf
    // instead.  Physical order of all other vertices == the reference's index order.
    int32_t  sinkp, special_linked;
    int32_t  maybe_triv;                        // 0 => no type-1 trivial vertex exists (set when a degree drops to <= 1, phasing flags change, vertices appear)
    int32_t  maybe_broken;                      // 0 => no vertex can be broken (a degree only reaches 0 in unlink_*; new vertices appear in extend)
    // state of the trivial-vertex sweep in flight (scan_trivial <-> sweep_trivial)
    double   sw_best_r, sw_hit_r; int32_t sw_best_v, sw_hit, sw_vend, sw_dom_base; uint32_t sw_need_lo, sw_need_hi;
    int32_t  scr_i[SCR_I]; double scr_d[SCR_D];
#ifdef ALD_PROF
    unsigned long long prof[32];
#endif
};

#if defined(ALD_EMU)
static thread_local Hot g_H;
static thread_local HotCtx g_HC;
#define H g_H
#define HC g_HC
#elif ALD_CLASS_ID >= ALD_FIRST_GLOBAL_CLASS
// catch-all class and the twins: the graph arrays do not fit LDS (or leave it to other classes); Hot is the first part of the wave's HBM slab
__shared__ ALD_GLOBAL Hot *g_Hp;
__shared__ HotCtx g_HC;
#define H (*g_Hp)
#define HC g_HC
#define ALD_HOT_IN_SLAB 1
#else
__shared__ Hot g_H;
__shared__ HotCtx g_HC;
#define H g_H
#define HC g_HC
#endif

// cold state: typed views at compile-time offsets of the slab
struct ColdVertex { double vw; int32_t lpos, rpos, vtype, v2v, memo, pad_; };     // splice_graph::vwrt / vertex_info + scallop::v2v; memo: router class of this vertex (epoch << 16 | type << 13 | degree)
struct alignas(64) ColdEdge {
    double   med, eabd, econf, s0abd;            // scallop::med, edge_info.abd / confidence, abundance of the first supporting sample
    uint64_t mask[NW];                           // scallop::mev as a bitmask over ORIGINAL vertices
    int32_t  mei, ecount, s0id;                  // scallop::mei, edge_info.count, id of the first (smallest-id) supporting sample
    uint32_t sp_off, sp_len;                     // support list in the pool (edge_info.samples / spAbd)
    uint8_t  estrand;                            // edge_info.strand
};
static_assert(sizeof(ColdVertex) == CL::VX_BYTES && sizeof(ColdEdge) == CL::ED_BYTES, "slab layout out of sync");
struct Cold {
    ALD_GLOBAL ColdVertex *vx;
    ALD_GLOBAL ColdEdge *ed;
    ALD_GLOBAL int32_t *sp_id; ALD_GLOBAL double *sp_abd;
    ALD_GLOBAL int32_t *hl, *hl_off, *hl_len, *hl_capk, *hl_cnt;   // phasing lists (hyper_set::edges / ecnts); elements are edge SLOTS or -1
    ALD_GLOBAL int32_t *wi; ALD_GLOBAL double *wd;                 // scalar work arrays
    ALD_GLOBAL unsigned long long *po;                             // pool offsets of this graph's records, in path order
    static constexpr int32_t po_cap = CL::PO_CAP;
    static constexpr uint32_t sp_cap = CL::SP_CAP, hl_cap = CL::HL_CAP;
    static constexpr int32_t hl_maxlists = CL::HL_MAXLISTS, w_cap = CL::W_CAP;
};
ALD_INL Cold cold_view()
{
    ALD_GLOBAL uint8_t *b = HC.cold; Cold C;
    C.vx = (ALD_GLOBAL ColdVertex*)(b + CL::o_vx); C.ed = (ALD_GLOBAL ColdEdge*)(b + CL::o_ed);
    C.sp_id = (ALD_GLOBAL int32_t*)(b + CL::o_spid); C.sp_abd = (ALD_GLOBAL double*)(b + CL::o_spabd);
    C.hl = (ALD_GLOBAL int32_t*)(b + CL::o_hl); C.hl_off = (ALD_GLOBAL int32_t*)(b + CL::o_hloff); C.hl_len = (ALD_GLOBAL int32_t*)(b + CL::o_hllen);
    C.hl_capk = (ALD_GLOBAL int32_t*)(b + CL::o_hlcapk); C.hl_cnt = (ALD_GLOBAL int32_t*)(b + CL::o_hlcnt);
    C.wi = (ALD_GLOBAL int32_t*)(b + CL::o_wi); C.wd = (ALD_GLOBAL double*)(b + CL::o_wd);
    C.po = (ALD_GLOBAL unsigned long long*)(b + CL::o_po);
    return C;
}
#define COLD const Cold C = cold_view()

// Diagnostic build only (-DALD_PROF): per-phase cycle sums, emitted as trace events 100+k at the end of each graph.
// The product build compiles none of this (no stamp executes in the measured kernel).
#if defined(ALD_PROF) && !defined(ALD_EMU)
  #define PROF_DECL unsigned long long prof_t_ = __builtin_readcyclecounter()
  #define PROF_ADD(k) do { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[k] += t1_ - prof_t_; prof_t_ = t1_; } while(0)
  #define PROF_RESET() (prof_t_ = __builtin_readcyclecounter())
#else
  #define PROF_DECL do {} while(0)
  #define PROF_ADD(k) do {} while(0)
  #define PROF_RESET() do {} while(0)
#endif
enum { PF_LOAD = 0, PF_BROKEN, PF_TRIV_EVAL, PF_TRIV_MUT, PF_SMALL_EVAL, PF_SMALL_MUT, PF_UNSPLIT, PF_COLLECT0, PF_G_BALANCE, PF_G_DP, PF_G_SPLITMERGE, PF_G_COLLECT, PF_FINISH,
       PF_T_BALANCE, PF_T_PAIRS, PF_T_SETUP, PF_T_MERGE_LOAD, PF_T_MERGE_ADD, PF_T_MERGE_ISECT, PF_T_MERGE_MASK, PF_T_MERGE_SUMS, PF_T_MERGE_KILL, PF_T_HS, PF_T_TAIL,
       PF_S5_DUP, PF_S5_BODY, PF_S5_RELINK, PF_S6_WALK, PF_S6_LINK, PF_S7, PF_SM_KILL, PF_SM_REEVAL, PF_COUNT };      // finer stamps inside the wave star (parts of M_mask / M_add) and the smallest-edge removal (parts of small_mut)

// ---------------------------------------------------------------- small helpers
ALD_INL void fail_(int st, int line)
{
#ifdef ALD_EMU
    if(HC.status == 0 && getenv("ALD_EMU_VERBOSE")) fprintf(stderr, "[emu] graph %d: status %d raised at decomp_device.h:%d\n", HC.g, st, line);
#endif
    if(HC.status == 0) HC.status = st;
}
#define fail(st) fail_((st), __LINE__)
#define ALD_UNLIKELY(x) __builtin_expect(!!(x), 0)     // the checks that mirror the reference's asserts: the common path falls through
ALD_FN void trace_emit(int code, int a, int b, double v)
{
    int cap = HC.p_trace_cap;
    ALD_GLOBAL const KernelArgs *A = HC.args;
    int k = HC.n_trace++;
    if(k < cap) { int64_t o = (int64_t)HC.g * cap + k; A->out.trace_codes[3 * o] = code; A->out.trace_codes[3 * o + 1] = a; A->out.trace_codes[3 * o + 2] = b; A->out.trace_vals[o] = v; }
}
#ifdef ALD_PROF
ALD_INL bool tracing_u() { return false; }
ALD_INL bool tracing() { return false; }        // the profiling build reports its cycle sums through the trace buffer, but takes no op trace: the phases are timed as the product runs them
#else
ALD_INL bool tracing_u() { return uni(HC.p_trace_cap) > 0; }
ALD_INL bool tracing() { return HC.p_trace_cap > 0; }
#endif
ALD_INL void trace(int code, int a, int b, double v) { HC.n_iters++; if(tracing()) trace_emit(code, a, b, v); }
// u_*: the same accessors for the scalar (lane-0) routines, with the result marked wave-uniform (see uni() in decomp_common.h)
// an edge slot or NIL -> slot or -1.  Slots of every class but the largest stay below 2^15, so NIL (0xFFFF) read as a SIGNED 16-bit value
// is already the -1 the walks test for: the load itself sign-extends (ds_read_i16) and the compare + select per list step goes away.
#ifndef ALD_NO_SEXT_LINKS
ALD_INL int slot_or_neg(IDX h) { return MAXE < 32768 ? (int)(int16_t)h : (h == NIL ? -1 : (int)h); }
#else
ALD_INL int slot_or_neg(IDX h) { return h == NIL ? -1 : (int)h; }
#endif
ALD_INL int u_first_in(int v) { return uni(slot_or_neg(H.vx[v].in_head)); }
ALD_INL int u_first_out(int v) { return uni(slot_or_neg(H.vx[v].out_head)); }
ALD_INL int u_next_in(int e) { return uni(slot_or_neg(H.ed[e].lk.inx)); }
ALD_INL int u_next_out(int e) { return uni(slot_or_neg(H.ed[e].lk.onx)); }
ALD_INL int first_in(int v) { return slot_or_neg(H.vx[v].in_head); }
ALD_INL int first_out(int v) { return slot_or_neg(H.vx[v].out_head); }
ALD_INL int next_in(int e) { return slot_or_neg(H.ed[e].lk.inx); }
ALD_INL int next_out(int e) { return slot_or_neg(H.ed[e].lk.onx); }
ALD_INL double in_weights(int v) { double w = 0; for(int e = first_in(v); e >= 0; e = next_in(e)) w += H.ed[e].w; return w; }    // splice_graph.cc:187-198
ALD_INL double out_weights(int v) { double w = 0; for(int e = first_out(v); e >= 0; e = next_out(e)) w += H.ed[e].w; return w; } // splice_graph.cc:174-185

// ---------------------------------------------------------------- sorted adjacency lists (scalar code)
// in-list of v ordered by (source, id); out-list ordered by (target, id): graph/edge_base.h:35-45
// out(source 0) and in(sink) grow to dozens of entries and are never iterated by the rule cascade: until the final collect /
// greedy phase (materialize_special) edges are only counted there, not linked.
ALD_INL uint32_t tkey(uint32_t p) { return (int)p == HC.sinkp ? 0xFFFFu : p; }      // the sink sorts after every other vertex
ALD_INL int vlog(int p) { return p < HC.V0 - 1 ? p : (p == HC.sinkp ? HC.nv - 1 : p - 1); }   // physical -> reference index (traces)
ALD_INL uint64_t lkw(int e) { return uni(*(const uint64_t*)&H.ed[e].lk); }            // es | et << 16 | inx << 32 | onx << 48
ALD_INL int lk_next(uint32_t f) { return slot_or_neg((IDX)f); }
// ---- what the kept smallest-edge evaluations must forget (HotCtx::ev_dirty / ev_all).  Classes 0 / 1 keep none: nothing here costs them anything.
ALD_INL void ev_mark(int v)                     // v's lists, the weights in them or its membership changed
{
#if ALD_KEEP
  #ifdef ALD_EMU
    HC.ev_dirty[v >> 5] |= 1u << (v & 31);
    #if ALD_KEEP_TRIV
    HC.tv_dirty[v >> 5] |= 1u << (v & 31);
    #endif
  #else
    __hip_atomic_fetch_or(&HC.ev_dirty[v >> 5], 1u << (v & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);     // (lanes of the star mark different vertices of one word)
    #if ALD_KEEP_TRIV
    __hip_atomic_fetch_or(&HC.tv_dirty[v >> 5], 1u << (v & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    #endif
  #endif
#else
    (void)v;
#endif
}
// The hand-over in front of a star's lane-0 tail.  Where the sweep records are kept (ALD_KEEP: the slab-resident classes, whose lists live in
// GLOBAL memory) the tail walks far's list at once (ev_degree) -- the links the other lanes have just written: the stores are drained first.
ALD_INL void star_tail_sync() { if(ALD_KEEP) wsync_mem(); else wsync(); }
ALD_INL void ev_mark_all()
{
#if ALD_KEEP
    HC.ev_all = 1;
    #if ALD_KEEP_TRIV
    HC.tv_all = 1;
    #endif
#endif
}
// All marks off (ALL lanes call it).  Written WITHOUT a divergent region -- every lane stores, lanes beyond the last word store to the last word
// again --: as `if(lane == 0) ...; for(k = lane; k < words; ...)` these were two-instruction regions of narrowed EXEC at the point where the
// sweep's register pressure peaks, and in one build of round 4 the register allocator put four VGPR spills INSIDE one of them: the spill saved
// the active lanes only, the reload after the sweep ran under full EXEC, and the other lanes went on with what the scratch slot held before
// (the memory fault of the -DALD_STARFIX_MAX=1 test build on the twins; tools/isa_spill_audit.py finds the pattern in the assembly,
// tests/test_abi_cpu.py runs it over every kernel of the product build).
ALD_INL void ev_clear_marks(int vend)
{
#if ALD_KEEP
    const int words = (vend + 31) / 32;
    HC.ev_all = 0;
    for(int base = 0; base < words; base += ALD_WAVE) { const int k = base + lane_id(); HC.ev_dirty[k < words ? k : words - 1] = 0; }     // (wave-uniform trip count, every lane stores)
#else
    (void)vend;
#endif
}
ALD_INL void tv_clear_marks(int nv_now)
{
#if ALD_KEEP && ALD_KEEP_TRIV
    const int words = (nv_now + 31) / 32;
    HC.tv_all = 0;
    for(int base = 0; base < words; base += ALD_WAVE) { const int k = base + lane_id(); HC.tv_dirty[k < words ? k : words - 1] = 0; }
#else
    (void)nv_now;
#endif
}
// v's in- (out = false) or out-degree went from `before` to `after`.  v itself is marked; and when the degree crossed the 1 | 2 line, every
// vertex at the far end of an edge of that list: its evaluation tests exactly this degree (out_deg(s) > 1 for an in-edge s -> j, in_deg(t)
// > 1 for an out-edge j -> t: scallop.cc:858-896).  out(source) / in(sink) are not linked before the final phase: "everything" then.
// Per-lane code (kill_edge_wave calls it from two lanes at once).
ALD_INL void ev_degree(int v, int before, int after, bool out)
{
#if ALD_KEEP
    ev_mark(v);
    if((before <= 1) == (after <= 1)) return;
    if(!HC.special_linked && (out ? v == 0 : v == HC.sinkp)) { ev_mark_all(); return; }
    int guard = MAXE;
    for(int e = out ? slot_or_neg(H.vx[v].out_head) : slot_or_neg(H.vx[v].in_head); e >= 0 && guard-- > 0; e = out ? slot_or_neg(H.ed[e].lk.onx) : slot_or_neg(H.ed[e].lk.inx))
        ev_mark(out ? (int)H.ed[e].lk.et : (int)H.ed[e].lk.es);
#else
    (void)v; (void)before; (void)after; (void)out;
#endif
}
ALD_INL void link_in(int v, int e)
{
    v = uni(v); e = uni(e);
    if(v == uni(HC.sinkp) && !uni(HC.special_linked)) { const int dg = uni((int)H.vx[v].in_deg); H.vx[v].in_deg = (IDX)(dg + 1); ev_degree(v, dg, dg + 1, false); return; }
    const uint32_t ks = uni(H.ed[e].lk.es), kid = uni(H.eid[e]);
    IDX *pp = &H.vx[v].in_head; IDX cur = *pp;
    for(int guard = MAXE; uni(cur != NIL) && guard > 0; guard--) {      // (the guard only matters on a corrupted list: never spin)
        const uint64_t w = *(const uint64_t*)&H.ed[cur].lk; const uint32_t cs = (uint32_t)(w & 0xFFFF);
        bool stop = cs > ks; if(uni(cs == ks)) stop = H.eid[cur] > kid;
        if(uni(stop)) break;
        pp = &H.ed[cur].lk.inx; cur = (IDX)((w >> 32) & 0xFFFF);
    }
    H.ed[e].lk.inx = cur; *pp = (IDX)e;
    { const int dg = uni((int)H.vx[v].in_deg); H.vx[v].in_deg = (IDX)(dg + 1); ev_degree(v, dg, dg + 1, false); }
}
ALD_INL void link_out(int v, int e)
{
    v = uni(v); e = uni(e);
    if(v == 0 && !uni(HC.special_linked)) { const int dg = uni((int)H.vx[v].out_deg); H.vx[v].out_deg = (IDX)(dg + 1); ev_degree(v, dg, dg + 1, true); return; }
    const uint32_t sk = (uint32_t)uni(HC.sinkp);
    uint32_t kt = uni(H.ed[e].lk.et); const uint32_t kid = uni(H.eid[e]);
    if(kt == sk) kt = 0xFFFFu;
    IDX *pp = &H.vx[v].out_head; IDX cur = *pp;
    for(int guard = MAXE; uni(cur != NIL) && guard > 0; guard--) {      // (the guard only matters on a corrupted list: never spin)
        const uint64_t w = *(const uint64_t*)&H.ed[cur].lk; uint32_t ct = (uint32_t)((w >> 16) & 0xFFFF); if(ct == sk) ct = 0xFFFFu;
        bool stop = ct > kt; if(uni(ct == kt)) stop = H.eid[cur] > kid;
        if(uni(stop)) break;
        pp = &H.ed[cur].lk.onx; cur = (IDX)(w >> 48);
    }
    H.ed[e].lk.onx = cur; *pp = (IDX)e;
    { const int dg = uni((int)H.vx[v].out_deg); H.vx[v].out_deg = (IDX)(dg + 1); ev_degree(v, dg, dg + 1, true); }
}
// link_out with a starting point: `hint` is an edge of v's out-list known to sort before e (its target key is smaller)
ALD_INL void link_out_after(int v, int e, int hint)
{
    v = uni(v); e = uni(e); hint = uni(hint);
    if(v == 0 && !uni(HC.special_linked)) { const int dg = uni((int)H.vx[v].out_deg); H.vx[v].out_deg = (IDX)(dg + 1); ev_degree(v, dg, dg + 1, true); return; }
    const uint32_t sk = (uint32_t)uni(HC.sinkp);
    uint32_t kt = uni(H.ed[e].lk.et); const uint32_t kid = uni(H.eid[e]);
    if(kt == sk) kt = 0xFFFFu;
    IDX *pp = &H.ed[hint].lk.onx; IDX cur = *pp;
    for(int guard = MAXE; uni(cur != NIL) && guard > 0; guard--) {      // (the guard only matters on a corrupted list: never spin)
        const uint64_t w = *(const uint64_t*)&H.ed[cur].lk; uint32_t ct = (uint32_t)((w >> 16) & 0xFFFF); if(ct == sk) ct = 0xFFFFu;
        bool stop = ct > kt; if(uni(ct == kt)) stop = H.eid[cur] > kid;
        if(uni(stop)) break;
        pp = &H.ed[cur].lk.onx; cur = (IDX)(w >> 48);
    }
    H.ed[e].lk.onx = cur; *pp = (IDX)e;
    { const int dg = uni((int)H.vx[v].out_deg); H.vx[v].out_deg = (IDX)(dg + 1); ev_degree(v, dg, dg + 1, true); }
}
// The walks below keep the cursor in a vector register (an LDS address has to be in one anyway) and follow the ADDRESS of the link
// that points at the current edge; only the loop condition is made wave-uniform.
ALD_INL void unlink_in(int v, int e)
{
    v = uni(v); e = uni(e);
    if(v == uni(HC.sinkp) && !uni(HC.special_linked)) { const int dg = uni((int)H.vx[v].in_deg); H.vx[v].in_deg = (IDX)(dg - 1); ev_degree(v, dg, dg - 1, false); return; }
    IDX *pp = &H.vx[v].in_head; IDX cur = *pp; int guard = MAXE;
    while(uni((int)cur != e && cur != NIL) && guard-- > 0) { pp = &H.ed[cur].lk.inx; cur = *pp; }
    if(ALD_UNLIKELY(uni((int)cur != e))) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }       // cannot happen on a consistent state; never walk off a list
    *pp = H.ed[e].lk.inx;
    { int dg = (int)uni(H.vx[v].in_deg) - 1; H.vx[v].in_deg = (IDX)dg; ev_degree(v, dg + 1, dg, false); if(dg <= 1) { HC.maybe_triv = 1; if(dg == 0) HC.maybe_broken = 1; } }
}
ALD_INL void unlink_out(int v, int e)
{
    v = uni(v); e = uni(e);
    if(v == 0 && !uni(HC.special_linked)) { const int dg = uni((int)H.vx[v].out_deg); H.vx[v].out_deg = (IDX)(dg - 1); ev_degree(v, dg, dg - 1, true); return; }
    IDX *pp = &H.vx[v].out_head; IDX cur = *pp; int guard = MAXE;
    while(uni((int)cur != e && cur != NIL) && guard-- > 0) { pp = &H.ed[cur].lk.onx; cur = *pp; }
    if(ALD_UNLIKELY(uni((int)cur != e))) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
    *pp = H.ed[e].lk.onx;
    { int dg = (int)uni(H.vx[v].out_deg) - 1; H.vx[v].out_deg = (IDX)dg; ev_degree(v, dg + 1, dg, true); if(dg <= 1) { HC.maybe_triv = 1; if(dg == 0) HC.maybe_broken = 1; } }
}
// e stays in v's in-list but its key becomes (ks, newest id): one walk finds its predecessor and its new place
ALD_INL void relink_in(int v, int e, uint32_t ks)
{
    v = uni(v); e = uni(e);
    ev_mark(v);
    if(v == uni(HC.sinkp) && !uni(HC.special_linked)) return;
    int last = -1, pe = -1, ip = -1; bool seen = false, placed = false;
    int guard = MAXE;
    for(int cur = u_first_in(v); cur >= 0 && guard-- > 0; ) {
        uint64_t w = lkw(cur); int nx = lk_next((uint32_t)((w >> 32) & 0xFFFF));
        if(cur == e) { pe = last; seen = true; if(placed) break; }
        else { if(!placed && (uint32_t)(w & 0xFFFF) > ks) { ip = last; placed = true; if(seen) break; } last = cur; }
        cur = nx;
    }
    if(ALD_UNLIKELY(!seen)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
    if(!placed) ip = last;
    if(ip == pe) return;                                   // same place
    IDX nxe = uni(H.ed[e].lk.inx);
    if(pe < 0) H.vx[v].in_head = nxe; else H.ed[pe].lk.inx = nxe;
    if(ip < 0) { H.ed[e].lk.inx = uni(H.vx[v].in_head); H.vx[v].in_head = (IDX)e; } else { H.ed[e].lk.inx = uni(H.ed[ip].lk.inx); H.ed[ip].lk.inx = (IDX)e; }
}
ALD_INL void relink_out(int v, int e, uint32_t kt)         // kt already mapped by tkey()
{
    v = uni(v); e = uni(e);
    ev_mark(v);
    if(v == 0 && !uni(HC.special_linked)) return;
    const uint32_t sk = (uint32_t)uni(HC.sinkp);
    int last = -1, pe = -1, ip = -1; bool seen = false, placed = false;
    int guard = MAXE;
    for(int cur = u_first_out(v); cur >= 0 && guard-- > 0; ) {
        uint64_t w = lkw(cur); int nx = lk_next((uint32_t)(w >> 48));
        if(cur == e) { pe = last; seen = true; if(placed) break; }
        else { uint32_t ct = (uint32_t)((w >> 16) & 0xFFFF); if(ct == sk) ct = 0xFFFFu; if(!placed && ct > kt) { ip = last; placed = true; if(seen) break; } last = cur; }
        cur = nx;
    }
    if(ALD_UNLIKELY(!seen)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
    if(!placed) ip = last;
    if(ip == pe) return;
    IDX nxe = uni(H.ed[e].lk.onx);
    if(pe < 0) H.vx[v].out_head = nxe; else H.ed[pe].lk.onx = nxe;
    if(ip < 0) { H.ed[e].lk.onx = uni(H.vx[v].out_head); H.vx[v].out_head = (IDX)e; } else { H.ed[e].lk.onx = uni(H.ed[ip].lk.onx); H.ed[ip].lk.onx = (IDX)e; }
}
// The same two moves for the lane-parallel star: every lane works on ITS OWN vertex / edge (distinct vertices -> disjoint lists), so
// nothing here may be routed through the scalar unit.
ALD_INL void relink_in_lane(int v, int e, uint32_t ks)
{
    ev_mark(v);
    if(v == HC.sinkp && !HC.special_linked) return;
    int last = -1, pe = -1, ip = -1; bool seen = false, placed = false;
    int guard = MAXE;
    for(int cur = first_in(v); cur >= 0 && guard-- > 0; ) {
        const uint64_t w = *(const uint64_t*)&H.ed[cur].lk; const int nx = lk_next((uint32_t)((w >> 32) & 0xFFFF));
        if(cur == e) { pe = last; seen = true; if(placed) break; }
        else { if(!placed && (uint32_t)(w & 0xFFFF) > ks) { ip = last; placed = true; if(seen) break; } last = cur; }
        cur = nx;
    }
    if(ALD_UNLIKELY(!seen)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
    if(!placed) ip = last;
    if(ip == pe) return;
    const IDX nxe = H.ed[e].lk.inx;
    if(pe < 0) H.vx[v].in_head = nxe; else H.ed[pe].lk.inx = nxe;
    if(ip < 0) { H.ed[e].lk.inx = H.vx[v].in_head; H.vx[v].in_head = (IDX)e; } else { H.ed[e].lk.inx = H.ed[ip].lk.inx; H.ed[ip].lk.inx = (IDX)e; }
}
ALD_INL void relink_out_lane(int v, int e, uint32_t kt)    // kt already mapped by tkey()
{
    ev_mark(v);
    if(v == 0 && !HC.special_linked) return;
    const uint32_t sk = (uint32_t)HC.sinkp;
    int last = -1, pe = -1, ip = -1; bool seen = false, placed = false;
    int guard = MAXE;
    for(int cur = first_out(v); cur >= 0 && guard-- > 0; ) {
        const uint64_t w = *(const uint64_t*)&H.ed[cur].lk; const int nx = lk_next((uint32_t)(w >> 48));
        if(cur == e) { pe = last; seen = true; if(placed) break; }
        else { uint32_t ct = (uint32_t)((w >> 16) & 0xFFFF); if(ct == sk) ct = 0xFFFFu; if(!placed && ct > kt) { ip = last; placed = true; if(seen) break; } last = cur; }
        cur = nx;
    }
    if(ALD_UNLIKELY(!seen)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
    if(!placed) ip = last;
    if(ip == pe) return;
    const IDX nxe = H.ed[e].lk.onx;
    if(pe < 0) H.vx[v].out_head = nxe; else H.ed[pe].lk.onx = nxe;
    if(ip < 0) { H.ed[e].lk.onx = H.vx[v].out_head; H.vx[v].out_head = (IDX)e; } else { H.ed[e].lk.onx = H.ed[ip].lk.onx; H.ed[ip].lk.onx = (IDX)e; }
}
// x is left without edges.  One 8-byte store; the constant is made HERE (the compiler kept the one copy it had made at kernel entry in a
// spilled register pair and fetched it back from scratch memory -- a round trip to L2 -- at every use)
ALD_INL void clear_vertex(int x)
{
    ev_mark(x);
#if defined(ALD_EMU)
    H.vx[x].in_head = NIL; H.vx[x].out_head = NIL; H.vx[x].in_deg = 0; H.vx[x].out_deg = 0; H.nz[x] = 0;
#else
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    asm volatile("" : "+v"(lo), "+v"(hi));
    static_assert(sizeof(Hot::VertexHot) == 8 && NIL == 0xFFFF, "vertex record layout");
    union { uint64_t u; Hot::VertexHot v; } z; z.u = ((uint64_t)hi << 32) | lo;
    H.vx[x] = z.v;                                         // in_head = out_head = NIL, in_deg = out_deg = 0
    H.nz[x] = 0;
#endif
}
ALD_INL int free_slots() { return uni(HC.free_cnt) + (MAXE - uni(HC.slot_hw)); }
// directed_graph::add_edge (directed_graph.cc:38-48) + i2e.push_back: the new id is the largest
ALD_INL int add_edge_i(int s, int t)
{
    s = uni(s); t = uni(t);
    int e; int fh = uni(HC.free_head), hw = uni(HC.slot_hw);
    if(fh >= 0) { e = fh; IDX nx = uni(H.ed[e].lk.onx); HC.free_head = nx == NIL ? -1 : (int)nx; HC.free_cnt--; }
    else if(hw < MAXE) { e = hw; HC.slot_hw = hw + 1; }
    else { fail(ALD_ST_CAPACITY); return -1; }
    int id = uni(HC.next_id); HC.next_id = id + 1;
    if(ALD_UNLIKELY(id >= EID_LIMIT)) { fail(ALD_ST_CAPACITY); return -1; }
    H.ed[e].lk.es = (IDX)s; H.ed[e].lk.et = (IDX)t; H.eid[e] = (EID)id; H.hflag[e] = 0; H.ed[e].w = 0;
    link_out(s, e); link_in(t, e);
    return e;
}
// scallop::remove_edge (scallop.cc:2380-2392); the slot is recycled at once (every caller drops the edge from the phasing lists
// -- hs_remove / hs_replace -- before it creates another edge)
ALD_INL void kill_edge_i(int e)
{
    e = uni(e);
    unlink_out(uni(H.ed[e].lk.es), e); unlink_in(uni(H.ed[e].lk.et), e);
    H.ed[e].lk.es = NIL;
    { int fh = uni(HC.free_head); H.ed[e].lk.onx = fh < 0 ? NIL : (IDX)fh; HC.free_head = e; HC.free_cnt = uni(HC.free_cnt) + 1; }
}
// remove_edge by the wave (ALL lanes call, e wave-uniform): the edge leaves its source's out-list and its target's in-list AT THE SAME
// TIME -- lane 0 walks one list, lane 1 the other, same instruction stream, different links -- where kill_edge_i walks them one after
// the other.  The two lists share no link field (onx / inx), no head and no degree.
ALD_INL void kill_edge_wave(int e)
{
    e = uni(e);
    const uint64_t w = *(const uint64_t*)&H.ed[e].lk;            // es | et << 16 | inx << 32 | onx << 48, read before anything moves
    const bool special = HC.special_linked != 0; const int sinkp = HC.sinkp;
    for(int side = lane_id(); side < 2; side += ALD_WAVE) {
        const bool out = (side == 0);
        const int v = out ? (int)(w & 0xFFFF) : (int)((w >> 16) & 0xFFFF);
        const IDX nxe = out ? (IDX)(w >> 48) : (IDX)((w >> 32) & 0xFFFF);
        const bool counted = !special && (out ? v == 0 : v == sinkp);       // out(source) / in(sink): only counted until the final phase
        IDX *deg = out ? &H.vx[v].out_deg : &H.vx[v].in_deg;
        if(!counted) {
            IDX *pp = out ? &H.vx[v].out_head : &H.vx[v].in_head; IDX cur = *pp; int guard = MAXE;
            while((int)cur != e && cur != NIL && guard-- > 0) { pp = out ? &H.ed[cur].lk.onx : &H.ed[cur].lk.inx; cur = *pp; }
            if(ALD_UNLIKELY((int)cur != e)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); continue; }      // cannot happen on a consistent state
            *pp = nxe;
        }
        const int dg = (int)*deg - 1; *deg = (IDX)dg;
        ev_degree(v, dg + 1, dg, out);           // (the sweep that called evaluates the two vertices again itself -- unless it goes back to the cascade first)
        if(!counted && dg <= 1) { HC.maybe_triv = 1; if(dg == 0) HC.maybe_broken = 1; }
    }
    wsync();
    if(lane_id() == 0) {
        H.ed[e].lk.es = NIL;
        const int fh = uni(HC.free_head); H.ed[e].lk.onx = fh < 0 ? NIL : (IDX)fh; HC.free_head = e; HC.free_cnt = uni(HC.free_cnt) + 1;
    }
}
ALD_FN int add_edge(int s, int t) { return add_edge_i(uni(s), uni(t)); }
ALD_FN void kill_edge(int e) { kill_edge_i(uni(e)); }
ALD_FN void move_edge(int e, int x, int y)      // directed_graph.cc:180-194
{
    e = uni(e); x = uni(x); y = uni(y);
    unlink_out(uni(H.ed[e].lk.es), e); unlink_in(uni(H.ed[e].lk.et), e);
    H.ed[e].lk.es = (IDX)x; H.ed[e].lk.et = (IDX)y;
    link_out(x, e); link_in(y, e);
}

// splice_graph::get_strand_degree / mixed_strand_vertex (splice_graph.cc:1375-1406)
ALD_INL void strand_degree(int v, int vs[6])
{
    COLD;
    for(int k = 0; k < 6; k++) vs[k] = 0;
    for(int e = first_in(v); e >= 0; e = next_in(e)) vs[C.ed[e].estrand]++;
    for(int e = first_out(v); e >= 0; e = next_out(e)) vs[C.ed[e].estrand + 3]++;
}
ALD_INL bool mixed_strand_vertex(int v)
{
    if(!HC.any_strand) return false;
    int vs[6]; strand_degree(v, vs);
    return (vs[1] + vs[4] >= 1) && (vs[2] + vs[5] >= 1);
}
ALD_INL void borrow_edge_strand(const Cold &C, int e1, int e2) { int s2 = C.ed[e2].estrand; if(s2 == 0) return; C.ed[e1].estrand = (uint8_t)s2; }   // scallop.cc:1997-2007

// ---------------------------------------------------------------- sample support (edge_info.samples / spAbd)
// intersection with per-sample min, abd = sum in ascending sample order (scallop.cc:2300-2318, 1915-1933)
// Support lists: a list of ONE sample lives inline in the edge record (s0id / s0abd) and never touches the pool; longer lists
// live in the pool at [sp_off, sp_off + sp_len).
ALD_FN bool intersect_samples(int e1, int e2, int z)
{
    e1 = uni(e1); e2 = uni(e2); z = uni(z);
    COLD;
    const uint32_t n1 = uni(C.ed[e1].sp_len), n2 = uni(C.ed[e2].sp_len);
    if(n1 == 1 && n2 == 1) {                     // single-sample edges: everything is inline, no pool traffic at all
        int a = uni(C.ed[e1].s0id), b = uni(C.ed[e2].s0id); double x = uni(C.ed[e1].s0abd), y = uni(C.ed[e2].s0abd);
        if(a == b) { double c = (y < x) ? y : x; C.ed[z].sp_off = 0; C.ed[z].sp_len = 1; C.ed[z].ecount = 1; C.ed[z].eabd = 0.0 + c; C.ed[z].s0id = a; C.ed[z].s0abd = c; }
        else { C.ed[z].sp_off = 0; C.ed[z].sp_len = 0; C.ed[z].ecount = 0; C.ed[z].eabd = 0; C.ed[z].s0id = 0; C.ed[z].s0abd = 0; }
        return true;
    }
    const uint32_t o1 = uni(C.ed[e1].sp_off), o2 = uni(C.ed[e2].sp_off);
    const int i1 = uni(C.ed[e1].s0id), i2 = uni(C.ed[e2].s0id); const double a1 = uni(C.ed[e1].s0abd), a2 = uni(C.ed[e2].s0abd);
    const uint32_t need = n1 < n2 ? n1 : n2;
    const uint32_t o = HC.sp_used;
    if(ALD_UNLIKELY(o + need > C.sp_cap)) { fail(ALD_ST_CAPACITY); return false; }
    uint32_t i = 0, j = 0, k = 0; double abd = 0; int first_id = 0; double first_abd = 0;
    while(i < n1 && j < n2) {
        int a = n1 == 1 ? i1 : uni(C.sp_id[o1 + i]), b = n2 == 1 ? i2 : uni(C.sp_id[o2 + j]);
        if(a < b) i++; else if(b < a) j++;
        else { double x = n1 == 1 ? a1 : uni(C.sp_abd[o1 + i]), y = n2 == 1 ? a2 : uni(C.sp_abd[o2 + j]); double c = (y < x) ? y : x;     // std::min(x, y)
               C.sp_id[o + k] = a; C.sp_abd[o + k] = c; abd += c; if(k == 0) { first_id = a; first_abd = c; } k++; i++; j++; }
    }
    C.ed[z].sp_len = k; C.ed[z].ecount = (int32_t)k; C.ed[z].eabd = abd; C.ed[z].s0id = first_id; C.ed[z].s0abd = first_abd;
    if(k >= 2) { C.ed[z].sp_off = o; HC.sp_used = o + k; } else C.ed[z].sp_off = 0;      // a single survivor stays inline
    return true;
}
// router.cc:1035-1038: sum over common samples of 0.99*min + 0.01*max
ALD_FN double common_abd(int e1, int e2)
{
    e1 = uni(e1); e2 = uni(e2);
    COLD;
    const uint32_t o1 = uni(C.ed[e1].sp_off), n1 = uni(C.ed[e1].sp_len), o2 = uni(C.ed[e2].sp_off), n2 = uni(C.ed[e2].sp_len);
    const int i1 = uni(C.ed[e1].s0id), i2 = uni(C.ed[e2].s0id); const double a1 = uni(C.ed[e1].s0abd), a2 = uni(C.ed[e2].s0abd);
    uint32_t i = 0, j = 0; double c = 0;
    while(i < n1 && j < n2) {
        int a = n1 == 1 ? i1 : uni(C.sp_id[o1 + i]), b = n2 == 1 ? i2 : uni(C.sp_id[o2 + j]);
        if(a < b) i++; else if(b < a) j++;
        else { double x = n1 == 1 ? a1 : uni(C.sp_abd[o1 + i]), y = n2 == 1 ? a2 : uni(C.sp_abd[o2 + j]); double mn = (y < x) ? y : x, mx = (x < y) ? y : x; c += 0.99 * mn + 0.01 * mx; i++; j++; }
    }
    return c;
}

// ---------------------------------------------------------------- phasing lists: hyper_set (scalar code)
// Lists hold edge SLOTS; every query scans the lists, which is equivalent to the reference's e2s index because
// e2s[e] is always a superset of the lists that contain e (hyper_set.cc:626-675,787-818,865-902).
ALD_FN void hs_refresh_flags()                  // per-slot OCC / LEXT / REXT: hyper_set.cc:949-983 left/right_extend
{
    if(!HC.hs_dirty) return;
    HC.maybe_triv = 1; ev_mark_all();
    COLD;
    for(int e = 0; e < HC.slot_hw; e++) H.hflag[e] = 0;
    int nl = HC.hl_n;
    for(int k = 0; k < nl; k++) {
        ALD_GLOBAL int32_t *v = C.hl + uni(C.hl_off[k]); int n = uni(C.hl_len[k]);
        for(int i = 0; i < n; i++) {
            int e = v[i]; if(e < 0) continue;
            uint8_t f = HF_OCC;
            if(i >= 1 && v[i - 1] != -1) f |= HF_LEXT;
            if(i + 1 < n && v[i + 1] != -1) f |= HF_REXT;
            H.hflag[e] |= f;
        }
    }
    HC.hs_dirty = 0;
}
ALD_FN void hs_remove_lists(int e)              // hyper_set.cc:787-818
{
    e = uni(e);
    int nl = HC.hl_n;
    COLD;
    for(int k = 0; k < nl; k++) { ALD_GLOBAL int32_t *v = C.hl + uni(C.hl_off[k]); int n = uni(C.hl_len[k]); for(int i = 0; i < n; i++) if(v[i] == e) { v[i] = -1; HC.hs_dirty = 1; } }
}
ALD_FN void hs_replace1_lists(int x, int e)     // hyper_set.cc:609-615 -> 626-675 with |v| == 1
{
    x = uni(x); e = uni(e);
    int nl = HC.hl_n;
    COLD;
    for(int k = 0; k < nl; k++) { ALD_GLOBAL int32_t *v = C.hl + uni(C.hl_off[k]); int n = uni(C.hl_len[k]); for(int i = 0; i < n; i++) if(v[i] == x) { v[i] = e; HC.hs_dirty = 1; } }
}
ALD_FN void hs_replace2_lists(int x, int y, int e)   // hyper_set.cc:617-624 -> 626-675 with |v| == 2
{
    x = uni(x); y = uni(y); e = uni(e);
    int nl = HC.hl_n;
    COLD;
    for(int k = 0; k < nl; k++) {
        ALD_GLOBAL int32_t *v = C.hl + uni(C.hl_off[k]); int n = uni(C.hl_len[k]); int w = 0;
        // matches of a 2-pattern with x != y cannot overlap; replace (x,y) by e left to right
        for(int i = 0; i < n; i++) {
            if(i + 1 < n && v[i] == x && v[i + 1] == y) { v[w++] = e; i++; HC.hs_dirty = 1; }
            else v[w++] = v[i];
        }
        C.hl_len[k] = w;
    }
}
ALD_FN void hs_insert_between_lists(int x, int y, int e)   // hyper_set.cc:865-902
{
    x = uni(x); y = uni(y); e = uni(e);
    int nl = HC.hl_n;
    COLD;
    for(int k = 0; k < nl; k++) {
        int n = uni(C.hl_len[k]); ALD_GLOBAL int32_t *v = C.hl + uni(C.hl_off[k]);
        int cnt = 0;
        for(int i = 0; i + 1 < n; i++) if(v[i] == x && v[i + 1] == y) cnt++;
        if(cnt == 0) continue;
        if(n + cnt > uni(C.hl_capk[k])) {             // relocate the list to the end of the pool with slack
            uint32_t ncap = (uint32_t)(n + cnt) * 2u + 4u, o = HC.hl_used;
            if(ALD_UNLIKELY(o + ncap > C.hl_cap)) { fail(ALD_ST_CAPACITY); return; }
            for(int i = 0; i < n; i++) C.hl[o + i] = v[i];
            HC.hl_used = o + ncap; C.hl_off[k] = (int32_t)o; C.hl_capk[k] = (int32_t)ncap; v = C.hl + o;
        }
        // the reference scans left to right and inserts e after every x that is followed by y
        int w = n + cnt - 1;
        for(int i = n - 1; i >= 0; i--) {
            v[w--] = v[i];
            if(i >= 1 && v[i - 1] == x && v[i] == y) v[w--] = e;
        }
        C.hl_len[k] = n + cnt; HC.hs_dirty = 1;
    }
}
// a graph without phasing lists (hl_n == 0) pays one LDS read per edit, not a call
ALD_INL void hs_remove(int e) { if(uni(HC.hl_n) != 0) hs_remove_lists(e); }
ALD_INL void hs_replace1(int x, int e) { if(uni(HC.hl_n) != 0) hs_replace1_lists(x, e); }
ALD_INL void hs_replace2(int x, int y, int e) { if(uni(HC.hl_n) != 0) hs_replace2_lists(x, y, e); }
ALD_INL void hs_insert_between(int x, int y, int e) { if(uni(HC.hl_n) != 0) hs_insert_between_lists(x, y, e); }
// hyper_set.cc:1003-1042 (side == 2, left_dominate) and 1044-1082 (side == 1, right_dominate)
ALD_FN bool hs_dominate(int e, int side)
{
    e = uni(e); side = uni(side);
    COLD;
    int nl = HC.hl_n;
    ALD_GLOBAL int32_t *x1 = C.wi, *x2 = C.wi + C.w_cap / 4; int n1 = 0, n2 = 0; const int cap = C.w_cap / 8;
    for(int k = 0; k < nl; k++) {
        ALD_GLOBAL int32_t *v = C.hl + uni(C.hl_off[k]); int n = uni(C.hl_len[k]);
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
    ei = uni(ei); w = uni(w);
    if(ALD_UNLIKELY(!(w >= HC.p_min_w - kSMIN))) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return -1; }
    double ww = uni(H.ed[ei].w);
    if(fabs(ww - w) <= kSMIN) return ei;
    int s = uni(H.ed[ei].lk.es), t = uni(H.ed[ei].lk.et);
    int p2 = add_edge(s, t);
    if(p2 < 0) return -1;
    COLD;
    double www = ww - w;
    double mw = HC.p_min_w;
    if(www <= mw) www = mw;
    H.ed[ei].w = www; H.ed[p2].w = w;
    C.ed[p2].estrand = uni(C.ed[ei].estrand); C.ed[p2].ecount = uni(C.ed[ei].ecount); C.ed[p2].eabd = uni(C.ed[ei].eabd); C.ed[p2].econf = uni(C.ed[ei].econf);
    C.ed[p2].sp_off = uni(C.ed[ei].sp_off); C.ed[p2].sp_len = uni(C.ed[ei].sp_len); C.ed[p2].s0id = uni(C.ed[ei].s0id); C.ed[p2].s0abd = uni(C.ed[ei].s0abd);   // immutable support lists are shared
    for(int k = 0; k < NW; k++) C.ed[p2].mask[k] = uni(C.ed[ei].mask[k]);
    C.ed[p2].mei = uni(C.ed[ei].mei); C.ed[p2].med = uni(C.ed[ei].med) * w / ww;
    return p2;
}
// scallop::merge_adjacent_edges(x, y, ww) (scallop.cc:2394-2416) = split_edge(x, ww) + split_edge(y, ww) (scallop.cc:2433-2484)
// + merge_adjacent_equal_edges (scallop.cc:2242-2378), fused: the two split pieces live only between the split and the merge in
// the reference, so they are never materialised here -- their creation ids are consumed, their weights take part in the vertex
// sums at the position their (endpoint, id) keys would have had, and everything else is computed from the originals' state.
ALD_INL int merge_adjacent_edges_i(int x, int y, double ww)
{
    const double mw = HC.p_min_w;
    if(ALD_UNLIKELY(!(ww >= mw - kSMIN))) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return -1; }
    if(x < 0 || y < 0) return -1;
    if(H.ed[x].lk.et != uni(H.ed[y].lk.es)) { int t = x; x = y; y = t; }
    const int xs = uni(H.ed[x].lk.es), xt = uni(H.ed[x].lk.et), yt = uni(H.ed[y].lk.et);
    if((int)uni(H.ed[y].lk.es) != xt) return -1;
    PROF_DECL;
    COLD;
    const double wx = uni(H.ed[x].w), wy = uni(H.ed[y].w);
    const bool sx = !(fabs(wx - ww) <= kSMIN), sy = !(fabs(wy - ww) <= kSMIN);     // does split_edge cut a piece off?
    // cold state of the originals (one round of independent loads)
    const double medx = uni(C.ed[x].med), medy = uni(C.ed[y].med), cx = uni(C.ed[x].econf), cy = uni(C.ed[y].econf), vwt = uni(C.vx[xt].vw);
    const int meix = uni(C.ed[x].mei), meiy = uni(C.ed[y].mei), cntx = uni(C.ed[x].ecount), cnty = uni(C.ed[y].ecount), lt = uni(C.vx[xt].lpos), rt = uni(C.vx[xt].rpos), ov = uni(C.vx[xt].v2v);
    const int stx = uni(C.ed[x].estrand), sty = uni(C.ed[y].estrand);
    // split_edge(x, ww), split_edge(y, ww): a piece of weight ww gets the next id, the original keeps max(w - ww, min_w)
    if(ALD_UNLIKELY(uni(HC.next_id) >= EID_LIMIT)) { fail(ALD_ST_CAPACITY); return -1; }
    if(sx) { HC.next_id++; double r = wx - ww; if(r <= mw) r = mw; H.ed[x].w = r; }
    if(sy) { HC.next_id++; double r = wy - ww; if(r <= mw) r = mw; H.ed[y].w = r; }
    const double wx0 = sx ? ww : wx, wy0 = sy ? ww : wy;                 // weights of the two pieces being merged
    const double medx1 = sx ? medx * ww / wx : medx, medy1 = sy ? medy * ww / wy : medy;
    // merge_adjacent_equal_edges(piece x, piece y)
    PROF_ADD(PF_T_MERGE_LOAD);
    int n = add_edge_i(xs, yt);
    PROF_ADD(PF_T_MERGE_ADD);
    if(n < 0) return -1;
    if(ALD_UNLIKELY(!(fabs(wx0 - wy0) <= kSMIN))) { fail(ALD_ST_INVARIANT + ALD_INV_MERGE_EQUAL); return -1; }
    H.ed[n].w = wx0 * 0.5 + wy0 * 0.5;
    if(ALD_UNLIKELY(!(cntx > 0 && cnty > 0))) { fail(ALD_ST_INVARIANT + ALD_INV_COUNT); return -1; }
    if(!intersect_samples(x, y, n)) return -1;
    PROF_ADD(PF_T_MERGE_ISECT);
    C.ed[n].econf = cx + cy;
    C.ed[n].estrand = (uint8_t)(sty != 0 ? sty : stx);                     // borrow_edge_strand(n, x) then (n, y): a non-zero strand of y wins
    for(int k = 0; k < NW; k++) C.ed[n].mask[k] = uni(C.ed[x].mask[k]) | uni(C.ed[y].mask[k]);
    if(ov >= 0) C.ed[n].mask[(ov >> 6)] |= (1ull << (ov & 63));
    // get_in_weights(xt) / get_out_weights(xt) while both pieces are still attached: a piece sorts behind every edge with the
    // same far endpoint (its id is the newest), before the first edge with a larger one
    PROF_ADD(PF_T_MERGE_MASK);
    double sum1 = 0, sum2 = 0;
    { bool ins = !sx; for(int e = u_first_in(xt); e >= 0; e = u_next_in(e)) { if(!ins && (int)uni(H.ed[e].lk.es) > xs) { sum1 += ww; ins = true; } sum1 += uni(H.ed[e].w); } if(!ins) sum1 += ww; }
    { bool ins = !sy; for(int e = u_first_out(xt); e >= 0; e = u_next_out(e)) { if(!ins && tkey(H.ed[e].lk.et) > tkey(yt)) { sum2 += ww; ins = true; } sum2 += uni(H.ed[e].w); } if(!ins) sum2 += ww; }
    const double sum = (sum1 + sum2) * 0.5;
    const double r1 = vwt * (wx0 + wy0) * 0.5 / sum;
    C.vx[xt].vw = vwt - r1;
    const int mi = rt - lt + meix + meiy;
    C.ed[n].med = mi * r1 + medx1 + medy1; C.ed[n].mei = mi;
    // the pieces disappear; an edge that was not cut IS the piece
    PROF_ADD(PF_T_MERGE_SUMS);
    if(!sx) kill_edge_i(x);
    if(!sy) kill_edge_i(y);
    if(H.vx[xt].in_deg == 0 && uni(H.vx[xt].out_deg) == 0) H.nz[xt] = 0;
    PROF_ADD(PF_T_MERGE_KILL);
    return n;
}
ALD_FN int merge_adjacent_edges(int x, int y, double ww) { return merge_adjacent_edges_i(uni(x), uni(y), uni(ww)); }     // out-of-line copy for the greedy phase
// scallop::balance_vertex (scallop.cc:2486-2576)
ALD_INL void balance_vertex_i(int v)
{
    if(H.vx[v].in_deg == 0 || uni(H.vx[v].out_deg) == 0) return;
    const double mw = HC.p_min_w;
    double w1 = 0, w2 = 0;
    for(int e = u_first_in(v); e >= 0; e = u_next_in(e)) { double w = uni(H.ed[e].w); if(ALD_UNLIKELY(!(w >= mw - kSMIN))) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; } w1 += w; }
    for(int e = u_first_out(v); e >= 0; e = u_next_out(e)) { double w = uni(H.ed[e].w); if(ALD_UNLIKELY(!(w >= mw - kSMIN))) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; } w2 += w; }
    double ww = sqrt(w1 * w2);
    double r1 = ww / w1, r2 = ww / w2;
    double m1 = 0, m2 = 0;
    for(int e = u_first_in(v); e >= 0; e = u_next_in(e)) { double wy = uni(H.ed[e].w) * r1; if(wy < mw) { m1 += mw - wy; wy = mw; } H.ed[e].w = wy; }
    for(int e = u_first_out(v); e >= 0; e = u_next_out(e)) { double wy = uni(H.ed[e].w) * r2; if(wy < mw) { m2 += mw - wy; wy = mw; } H.ed[e].w = wy; }
    if(m1 > m2) { int e = u_first_out(v); H.ed[e].w = uni(H.ed[e].w) + m1 - m2; }
    else if(m1 < m2) { int e = u_first_in(v); H.ed[e].w = uni(H.ed[e].w) + m2 - m1; }
}

ALD_FN void balance_vertex(int v) { balance_vertex_i(uni(v)); }
// pe2w as a sorted array: keys (id1,id2) ascending == std::map<PI,double> order (router.h:23).  The pairs of a small vertex
// (<= LP pairs) live in the LDS scratch, larger sets in the upper halves of the slab's work arrays.  A second area parks the best
// candidate of an unsplittable sweep while the sweep goes on.
// A pair entry packs (edge slot | local index << 16): the local index is the edge's position among the root's in-edges followed
// by its out-edges (adjacency order), which every consumer needs and which would otherwise cost a per-slot lookup table in LDS.
struct Pairs { int32_t *a, *b; double *w; int cap; };
#define PSLOT(x) ((int)((x) & 0xFFFF))
#define PLOC(x)  ((int)(((uint32_t)(x)) >> 16))
#define PMAKE(slot, loc) ((int32_t)((uint32_t)(slot) | ((uint32_t)(loc) << 16)))
static constexpr int PW_CAP = Cold::w_cap / 8;       // pairs per area in the slab (current / parked)
ALD_INL Pairs pairs_at(bool lds, bool parked)
{
    Pairs p;
    if(lds) { int o = parked ? 2 * LP : 0; p.a = (int32_t*)HC.scr_i + o; p.b = p.a + LP; p.w = (double*)HC.scr_d + (parked ? LP : 0); p.cap = LP; }
    else { COLD; int o = parked ? PW_CAP : 0; p.a = (int32_t*)(C.wi + Cold::w_cap / 2) + o; p.b = (int32_t*)(C.wi + Cold::w_cap / 2 + Cold::w_cap / 4) + o; p.w = (double*)(C.wd + Cold::w_cap / 2) + o; p.cap = PW_CAP; }
    return p;
}
ALD_INL Pairs pairs_cur() { return pairs_at(HC.pw_lds != 0, false); }
// scalar arenas for the router / the decompositions: LDS when the vertex is small, else the lower half of the slab's work arrays
struct Arena { int32_t *i; double *d; int cap_i, cap_d; };
ALD_INL Arena arena_at(bool lds)
{
    Arena a;
    if(lds) { a.i = (int32_t*)HC.scr_i + 4 * LP; a.d = (double*)HC.scr_d + 2 * LP; a.cap_i = ARENA_I; a.cap_d = ARENA_D; }
    else { COLD; a.i = (int32_t*)C.wi; a.d = (double*)C.wd; a.cap_i = Cold::w_cap / 2; a.cap_d = Cold::w_cap / 2; }
    return a;
}
ALD_INL bool pair_less(int a1, int a2, int b1, int b2)
{
    uint32_t x1 = H.eid[a1], y1 = H.eid[b1];
    if(x1 != y1) return x1 < y1;
    return H.eid[a2] < H.eid[b2];
}
ALD_INL void sort_pairs(const Pairs &P, int n)   // insertion sort by (id(e1), id(e2)); keys are unique
{
    for(int i = 1; i < n; i++) {
        int x = P.a[i], y = P.b[i]; double z = P.w[i]; int j = i - 1;
        while(j >= 0 && pair_less(PSLOT(x), PSLOT(y), PSLOT(P.a[j]), PSLOT(P.b[j]))) { P.a[j + 1] = P.a[j]; P.b[j + 1] = P.b[j]; P.w[j + 1] = P.w[j]; j--; }
        P.a[j + 1] = x; P.b[j + 1] = y; P.w[j + 1] = z;
    }
}

// scallop::decompose_trivial_vertex (scallop.cc:2144-2167) = balance_vertex + pe2w (in x out) + decompose_vertex_replace
// (scallop.cc:2009-2142), written for the shape every trivial vertex has -- ONE edge c on one side (A: the in-edge, else the
// out-edge) and a fan of d edges on the other -- without the pair array: pe2w is {(c, f_j)} with weight min(w(c), w(f_j)), visited in
// creation-id order of f_j (the sorted order of map<PI,double>, router.h:23).  What merge_adjacent_edges (scallop.cc:2394-2431)
// does to such a pair is known in advance: the fan edge is never cut (its weight IS the pair weight), c is cut until the last
// pair consumes it, and the merged edge differs from f_j only in one endpoint.  So the merged edge takes over f_j's slot (new id,
// new endpoint, re-sorted into the two lists it belongs to), c's record and the vertex are read once, and the weight sums
// around x are taken from the gathered fan in list order -- the same additions, in the same order, as the general form.
enum { STAR_MAX = 32 };
template<bool A, bool SMALL> ALD_INL void decompose_trivial_star(int x)
{
    COLD;
    PROF_DECL;
    const double mw = HC.p_min_w;
    const int c = A ? u_first_in(x) : u_first_out(x);
    // [n] fan edges in adjacency-list order, -1 once merged / [n] positions in fe, ascending creation id / [n] pe2w weight of (c, fan
    // edge): a fan of at most STAR_MAX edges uses the arena part of the LDS scratch (the parked pair area stays intact), a larger
    // one the first quarter of the slab's work arrays (the region [3/8, 1/2) of wi belongs to decompose_vertex_extend)
    int32_t *fe = SMALL ? (int32_t*)HC.scr_i + 4 * LP : (int32_t*)C.wi;
    int32_t *ord = SMALL ? fe + STAR_MAX : (int32_t*)(C.wi + Cold::w_cap / 8);
    double *fw = SMALL ? (double*)HC.scr_d + 2 * LP : (double*)C.wd;
    // balance_vertex(x) (scallop.cc:2486-2576) on the gathered weights: the same sums, ratios, clamps and remainder fix-up in the
    // same order as the list-walking form (balance_vertex_i), but the fan is walked once and nothing is written back until the
    // pe2w sums below are known
    int n = 0; double wcen = H.ed[c].w;          // weights stay in vector registers: they only feed FP arithmetic and LDS stores
    if(ALD_UNLIKELY(uni(!(wcen >= mw - kSMIN)))) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; }
    double sfan0 = 0;
    for(int e = A ? u_first_out(x) : u_first_in(x); e >= 0; e = A ? u_next_out(e) : u_next_in(e)) {
        double w2 = H.ed[e].w; if(ALD_UNLIKELY(uni(!(w2 >= mw - kSMIN)))) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; }
        fe[n] = e; fw[n] = w2; sfan0 += w2; n++;
    }
    {
        double scen0 = 0; scen0 += wcen;
        const double w_in = A ? scen0 : sfan0, w_out = A ? sfan0 : scen0;
        const double bw = sqrt(w_in * w_out);
        const double r_in = bw / w_in, r_out = bw / w_out;
        double m_cen = 0, m_fan = 0;
        { double wy = wcen * (A ? r_in : r_out); if(wy < mw) { m_cen += mw - wy; wy = mw; } wcen = wy; }
        if(A) { /* in side first (c), then the fan: the order only matters for which sum a clamp goes to */ }
        for(int j = 0; j < n; j++) { double wy = fw[j] * (A ? r_out : r_in); if(wy < mw) { m_fan += mw - wy; wy = mw; } fw[j] = wy; }
        const double m1 = A ? m_cen : m_fan, m2 = A ? m_fan : m_cen;            // m1: in side, m2: out side
        if(m1 > m2) { if(A) fw[0] = fw[0] + m1 - m2; else wcen = wcen + m1 - m2; }          // first out-edge
        else if(m1 < m2) { if(A) wcen = wcen + m2 - m1; else fw[0] = fw[0] + m2 - m1; }     // first in-edge
    }
    PROF_ADD(PF_T_BALANCE);
    const double wc = wcen;
    for(int j = 0; j < n; j++) { double w2 = fw[j]; fw[j] = A ? (wc <= w2 ? wc : w2) : (w2 <= wc ? w2 : wc); }
    double *pfx = SMALL ? nullptr : (double*)(C.wd + Cold::w_cap / 8);     // [n] large form only: prefix sums of fw over the live fan edges (list order)
    if(!SMALL) { double run = 0; for(int k = 0; k < n; k++) { run += fw[k]; pfx[k] = run; } }
    int32_t *aux = SMALL ? nullptr : (int32_t*)(C.wi + 2 * (Cold::w_cap / 8));        // [n] second buffer of the large form (merge sort, deferred inserts)
    if(SMALL) { for(int i = 0; i < n; i++) { int k = i; uint32_t id = uni(H.eid[fe[i]]); while(k > 0 && (uint32_t)uni(H.eid[fe[ord[k - 1]]]) > id) { ord[k] = ord[k - 1]; k--; } ord[k] = i; } }
    else {      // a hub's fan can hold hundreds of edges: bottom-up merge sort of the positions by creation id
        for(int i = 0; i < n; i++) ord[i] = i;
        int32_t *src = ord, *dst = aux;
        for(int wdt = 1; wdt < n; wdt *= 2) {
            for(int lo = 0; lo < n; lo += 2 * wdt) {
                int mid = lo + wdt < n ? lo + wdt : n, hi = lo + 2 * wdt < n ? lo + 2 * wdt : n, i = lo, j = mid, k = lo;
                while(i < mid && j < hi) { if((uint32_t)H.eid[fe[src[j]]] < (uint32_t)H.eid[fe[src[i]]]) dst[k++] = src[j++]; else dst[k++] = src[i++]; }
                while(i < mid) dst[k++] = src[i++];
                while(j < hi) dst[k++] = src[j++];
            }
            int32_t *t2 = src; src = dst; dst = t2;
        }
        if(src != ord) for(int i = 0; i < n; i++) ord[i] = src[i];
    }
    double mdc = 0;
    for(int q = 0; q < n; q++) { double w = fw[ord[q]]; if(ALD_UNLIKELY(!(w >= mw - kSMIN))) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; } mdc = (q == 0) ? w : mdc + w; }
    H.ed[c].w = mdc;
    for(int j = 0; j < n; j++) H.ed[fe[j]].w = fw[j];
    PROF_ADD(PF_T_SETUP);
    const int far = A ? (int)uni(H.ed[c].lk.es) : (int)uni(H.ed[c].lk.et);
    const double medc = C.ed[c].med, cc = C.ed[c].econf;
    const int meic = uni(C.ed[c].mei), cntc = uni(C.ed[c].ecount), stc = uni(C.ed[c].estrand);
    const uint32_t nsc = uni(C.ed[c].sp_len); const int idc = uni(C.ed[c].s0id); const double abc = C.ed[c].s0abd;     // c's support never changes
    double vwt = C.vx[x].vw; const int lt = uni(C.vx[x].lpos), rt = uni(C.vx[x].rpos), ov = uni(C.vx[x].v2v);
    bool consumed = false;
    for(int q = 0; q < n; q++) {
        if(consumed) { C.vx[x].vw = vwt; fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }     // the general form would be handed a dead edge here
        const int j = ord[q], f = fe[j]; const double ww = fw[j];
        const double wcur = H.ed[c].w;                                   // what is left of c
        const bool sc = uni(!(fabs(wcur - ww) <= kSMIN));                        // split_edge(c, ww) cuts a piece off (scallop.cc:2433-2484)
        int nid = uni(HC.next_id);
        if(nid >= EID_LIMIT) { C.vx[x].vw = vwt; fail(ALD_ST_CAPACITY); return; }
        double rem = wcur;
        if(sc) { nid++; rem = wcur - ww; if(rem <= mw) rem = mw; H.ed[c].w = rem; }    // the piece takes an id and disappears in the merge
        HC.next_id = nid + 1;                                                // id of the merged edge
        const double wc0 = sc ? ww : wcur;
        const double medc1 = sc ? medc * ww / wcur : medc;
        // everything the step needs from f's record, in one round of independent loads (one 64-byte line for NW == 1)
        const double medf = C.ed[f].med, cf = C.ed[f].econf;
        const int meif = uni(C.ed[f].mei), cntf = uni(C.ed[f].ecount), stf = uni(C.ed[f].estrand);
        const uint32_t nsf = uni(C.ed[f].sp_len); const int idf = uni(C.ed[f].s0id); const double abf = C.ed[f].s0abd;
        const uint64_t mk0 = C.ed[f].mask[0] | C.ed[c].mask[0];
        PROF_ADD(PF_T_MERGE_LOAD);
        if(!(cntc > 0 && cntf > 0)) { C.vx[x].vw = vwt; fail(ALD_ST_INVARIANT + ALD_INV_COUNT); return; }
        if(nsc == 1 && nsf == 1) {      // one supporting sample on both sides: intersect_samples' inline case, c's half already in registers
            if(idf == idc) { const double xa = A ? abc : abf, ya = A ? abf : abc; const double mn = (ya < xa) ? ya : xa; C.ed[f].sp_off = 0; C.ed[f].ecount = 1; C.ed[f].eabd = 0.0 + mn; C.ed[f].s0abd = mn; }
            else { C.ed[f].sp_off = 0; C.ed[f].sp_len = 0; C.ed[f].ecount = 0; C.ed[f].eabd = 0; C.ed[f].s0id = 0; C.ed[f].s0abd = 0; }
        }
        else if(!(A ? intersect_samples(c, f, f) : intersect_samples(f, c, f))) { C.vx[x].vw = vwt; return; }
        PROF_ADD(PF_T_MERGE_ISECT);
        C.ed[f].econf = A ? cc + cf : cf + cc;
        { const int sty = A ? stf : stc, stx = A ? stc : stf; C.ed[f].estrand = (uint8_t)(sty != 0 ? sty : stx); }
        C.ed[f].mask[0] = (ov >= 0 && ov < 64) ? (mk0 | (1ull << ov)) : mk0;
        for(int k = 1; k < NW; k++) { uint64_t mk = C.ed[c].mask[k] | C.ed[f].mask[k]; if(ov >= 0 && (ov >> 6) == k) mk |= (1ull << (ov & 63)); C.ed[f].mask[k] = mk; }
        PROF_ADD(PF_T_MERGE_MASK);
        // get_in_weights(x) / get_out_weights(x) with both pieces attached: c's side is (rest of c) + piece, the fan side is
        // whatever has not been merged yet, in list order
        double sfan = 0;
        if(SMALL) { for(int k = 0; k < n; k++) if(fe[k] >= 0) sfan += fw[k]; }
        else sfan = pfx[n - 1];                      // large form: running left-to-right sums over the live fan edges, kept up to date below
        double sc_side = 0; sc_side += sc ? rem : wcur; if(sc) sc_side += ww;
        const double sum = A ? (sc_side + sfan) * 0.5 : (sfan + sc_side) * 0.5;
        const double r1 = A ? vwt * (wc0 + ww) * 0.5 / sum : vwt * (ww + wc0) * 0.5 / sum;
        vwt = vwt - r1;
        const int mi = A ? rt - lt + meic + meif : rt - lt + meif + meic;
        C.ed[f].med = A ? mi * r1 + medc1 + medf : mi * r1 + medf + medc1; C.ed[f].mei = mi;
        PROF_ADD(PF_T_MERGE_SUMS);
        // f becomes the merged edge: newest id, far endpoint of c, weight of the two equal pieces
        const int other = A ? (int)uni(H.ed[f].lk.et) : (int)uni(H.ed[f].lk.es);
        H.eid[f] = (EID)nid; H.ed[f].w = A ? wc0 * 0.5 + ww * 0.5 : ww * 0.5 + wc0 * 0.5;
        // the new edge far -> other sorts behind c = far -> x whenever other's key is above x's (always, except for vertices added by
        // decompose_vertex_extend): the walk starts at c
        if(A) { H.ed[f].lk.es = (IDX)far; relink_in(other, f, (uint32_t)far); if(SMALL) { if(tkey((uint32_t)other) > tkey((uint32_t)x)) link_out_after(far, f, c); else link_out(far, f); } }
        else { H.ed[f].lk.et = (IDX)far; relink_out(other, f, tkey((uint32_t)far)); if(SMALL) link_in(far, f); }
        if(!SMALL) aux[q] = f;            // large form: the far vertex's list takes all new edges in ONE merge after the loop (nothing reads it meanwhile)
        fe[j] = -1;
        if(!SMALL) { double run = j > 0 ? pfx[j - 1] : 0.0; for(int k = j; k < n; k++) { if(fe[k] >= 0) run += fw[k]; pfx[k] = run; } }      // same additions, same order, from j on
        PROF_ADD(PF_T_MERGE_ADD);
        if(A) hs_replace2(c, f, f); else hs_replace2(f, c, f);
        if(n == 1) hs_replace1(c, f);
        if(!sc) consumed = true;
        PROF_ADD(PF_T_HS);
    }
    C.vx[x].vw = vwt;
    if(!SMALL) {
        // aux[0..n) = the merged edges in creation order; stable merge sort by the list's primary key (target for an out-list, source
        // for an in-list) gives (key, id) order, then one walk of far's list places them all
        int32_t *src = aux, *dst = ord;
        for(int wdt = 1; wdt < n; wdt *= 2) {
            for(int lo = 0; lo < n; lo += 2 * wdt) {
                int mid = lo + wdt < n ? lo + wdt : n, hi = lo + 2 * wdt < n ? lo + 2 * wdt : n, i = lo, j = mid, k = lo;
                while(i < mid && j < hi) {
                    const uint32_t kj = A ? tkey(H.ed[src[j]].lk.et) : (uint32_t)H.ed[src[j]].lk.es, ki = A ? tkey(H.ed[src[i]].lk.et) : (uint32_t)H.ed[src[i]].lk.es;
                    if(kj < ki) dst[k++] = src[j++]; else dst[k++] = src[i++];
                }
                while(i < mid) dst[k++] = src[i++];
                while(j < hi) dst[k++] = src[j++];
            }
            int32_t *t2 = src; src = dst; dst = t2;
        }
        const bool counted = A ? (far == 0 && !uni(HC.special_linked)) : (far == (int)uni(HC.sinkp) && !uni(HC.special_linked));
        if(!counted) {
            IDX *pp = A ? &H.vx[far].out_head : &H.vx[far].in_head; IDX cur = *pp; int guard = MAXE + n;
            for(int q = 0; q < n; q++) {
                const int e = src[q]; const uint32_t ke = A ? tkey(H.ed[e].lk.et) : (uint32_t)H.ed[e].lk.es;
                while(uni(cur != NIL) && guard-- > 0) {       // existing edges with key <= ke stay in front (their ids are older)
                    const uint32_t kc = A ? tkey(H.ed[cur].lk.et) : (uint32_t)H.ed[cur].lk.es;
                    if(uni(kc > ke)) break;
                    pp = A ? &H.ed[cur].lk.onx : &H.ed[cur].lk.inx; cur = *pp;
                }
                if(A) H.ed[e].lk.onx = cur; else H.ed[e].lk.inx = cur;
                *pp = (IDX)e; pp = A ? &H.ed[e].lk.onx : &H.ed[e].lk.inx;
            }
        }
        { const int dg = A ? (int)uni(H.vx[far].out_deg) : (int)uni(H.vx[far].in_deg); if(A) H.vx[far].out_deg = (IDX)(dg + n); else H.vx[far].in_deg = (IDX)(dg + n); ev_degree(far, dg, dg + n, A); }
    }
    if(n >= 2) hs_remove(c);
    if(ALD_UNLIKELY(!consumed)) { fail(ALD_ST_INVARIANT + ALD_INV_DEGREE); return; }      // c kept a remainder: the reference asserts on the degree of x
    // remove_edge(c); x is left without edges
    if(A) unlink_out(far, c); else unlink_in(far, c);
    H.ed[c].lk.es = NIL; H.hflag[c] = 0;
    { int fh = uni(HC.free_head); H.ed[c].lk.onx = fh < 0 ? NIL : (IDX)fh; HC.free_head = c; HC.free_cnt = uni(HC.free_cnt) + 1; }
    clear_vertex(x);
    PROF_ADD(PF_T_TAIL);
}
// fans above STAR_MAX are rare (hubs next to the source / sink late in the run): out of line, so the common path stays compact
ALD_FN void decompose_trivial_star_large(int x, int in_side) { x = uni(x); if(uni(in_side)) decompose_trivial_star<true, false>(x); else decompose_trivial_star<false, false>(x); }
ALD_FN void decompose_trivial_vertex(int x)
{
    x = uni(x);
    const int nin = uni(H.vx[x].in_deg), nout = uni(H.vx[x].out_deg);
    if(nin == 1 && nout >= 1 && nout <= STAR_MAX) decompose_trivial_star<true, true>(x);
    else if(nout == 1 && nin >= 1 && nin <= STAR_MAX) decompose_trivial_star<false, true>(x);
    else if(nin == 1 && nout >= 1 && nout <= Cold::w_cap / 8) decompose_trivial_star_large(x, 1);
    else if(nout == 1 && nin >= 1 && nin <= Cold::w_cap / 8) decompose_trivial_star_large(x, 0);
    else fail(ALD_ST_INVARIANT + ALD_INV_OTHER);        // not a trivial vertex (a fan cannot exceed MAXE edges)
}

ALD_FN bool resolve_single_trivial_vertex(int i, double jump_ratio);

// ---------------------------------------------------------------------------------------------------------------------------------
// The same decomposition (scallop.cc:2144-2167 -> 2486-2576, 2009-2142, 2394-2484, 2242-2378) executed by the WHOLE WAVE, one fan
// edge per lane.  What the merges of one star depend on each other through is small and numeric:
//   * what is left of c after every cut             -- a recurrence over the pair weights in creation-id order
//   * the ids handed out                            -- a running count of the cuts
//   * the weight of vertex x, reduced merge by merge -- a chain of multiply / divide / subtract
//   * the sum of the not yet merged fan weights      -- per merge, over the fan in list order
// Every lane replays the first two up to its own merge (a few arithmetic steps each), computes its own fan sum, ONE lane runs the
// vertex-weight chain over the per-merge sums left in LDS, and then every lane finishes its own merged edge: the cold record (one
// round of global loads and stores for the whole fan instead of one per merge), the new id / weight / endpoint, and its place in
// the far list of ITS other endpoint (distinct vertices -> disjoint lists; a fan with two edges to the same vertex takes the
// sequential walk).  Only the insertions into the ONE list all merged edges share (far's), the phasing-list edits and multi-sample
// support intersections (pool allocation) stay sequential, in creation order.  Every floating-point operation is the one the
// sequential form performs, on the same operands in the same order.  Called by ALL lanes; cross-lane values travel through the LDS
// scratch (so the single-lane emulation runs the phases as loops).
enum { SW_N = 128, SW_C, SW_FAR, SW_FAIL, SW_FAILQ, SW_SERIAL, SW_SMALL };          // context words in scr_i behind fe / ord / inv / oth (32 each)
enum { STAR_SMALL = 4 };                  // fans up to this size (79 % of them on the bench workload): lane 0 also ranks and sums in phase 0 -- two hand-overs fewer
template<bool A> ALD_INL void star_wave_body(int x)
{
    COLD;
    PROF_DECL;
#if defined(ALD_PROF) && defined(ALD_PROF_STAR_BY_SIZE)
    const unsigned long long prof_star0_ = __builtin_readcyclecounter();
#endif
    const int lane = lane_id();
    const double mw = HC.p_min_w;
    int32_t *fe = (int32_t*)HC.scr_i, *ord = fe + STAR_MAX, *inv = fe + 2 * STAR_MAX, *oth = fe + 3 * STAR_MAX, *ctx = (int32_t*)HC.scr_i;
    double *fw = (double*)HC.scr_d, *sq = fw + STAR_MAX;                  // pair weights (list order) / per merge: (sum, later r1), in merge order
    double *dctx = fw + 2 * STAR_MAX;                                      // [0] = weight c starts with
    // what the merges need of c's record and of vertex x is asked for NOW, by every lane (one broadcast request each): the round
    // trip to L2 runs under phases 0..2 instead of in front of phase 3
    const int c_early = A ? first_in(x) : first_out(x);
    const double medc = C.ed[c_early].med, cc = C.ed[c_early].econf, abc = C.ed[c_early].s0abd, vw_early = C.vx[x].vw;
    const int meic_v = C.ed[c_early].mei, cntc_v = C.ed[c_early].ecount, stc_v = C.ed[c_early].estrand, idc_v = C.ed[c_early].s0id;
    const uint32_t nsc_v = C.ed[c_early].sp_len;
    const int lt_v = C.vx[x].lpos, rt_v = C.vx[x].rpos, ov_v = C.vx[x].v2v;
    uint64_t cmask_pf[NW <= 2 ? NW : 1];                                   // c's vertex set too when it is one or two words (larger classes read it in phase 5)
    if(NW <= 2) for(int k = 0; k < NW; k++) cmask_pf[k] = C.ed[c_early].mask[k];
    // ---- phase 0 (lane 0): gather the fan, balance_vertex(x) on the gathered weights, pair weights -- as in the sequential form
    if(lane == 0) {
        const int c = A ? u_first_in(x) : u_first_out(x);
        int n = 0; double wcen = H.ed[c].w; int bad = 0;
        if(ALD_UNLIKELY(uni(!(wcen >= mw - kSMIN)))) bad = ALD_ST_INVARIANT + ALD_INV_WEIGHT;
        double sfan0 = 0;
        for(int e = A ? u_first_out(x) : u_first_in(x); e >= 0 && n < STAR_MAX; e = A ? u_next_out(e) : u_next_in(e)) {
            double w2 = H.ed[e].w; if(ALD_UNLIKELY(uni(!(w2 >= mw - kSMIN)))) bad = ALD_ST_INVARIANT + ALD_INV_WEIGHT;
            fe[n] = e; fw[n] = w2; sfan0 += w2; n++;
        }
        if(!bad) {
            double scen0 = 0; scen0 += wcen;
            const double w_in = A ? scen0 : sfan0, w_out = A ? sfan0 : scen0;
            const double bw = sqrt(w_in * w_out);
            const double r_in = bw / w_in, r_out = bw / w_out;
            double m_cen = 0, m_fan = 0;
            { double wy = wcen * (A ? r_in : r_out); if(wy < mw) { m_cen += mw - wy; wy = mw; } wcen = wy; }
            for(int j = 0; j < n; j++) { double wy = fw[j] * (A ? r_out : r_in); if(wy < mw) { m_fan += mw - wy; wy = mw; } fw[j] = wy; }
            const double m1 = A ? m_cen : m_fan, m2 = A ? m_fan : m_cen;
            if(m1 > m2) { if(A) fw[0] = fw[0] + m1 - m2; else wcen = wcen + m1 - m2; }
            else if(m1 < m2) { if(A) wcen = wcen + m2 - m1; else fw[0] = fw[0] + m2 - m1; }
            const double wc = wcen;
            for(int j = 0; j < n; j++) { double w2 = fw[j]; fw[j] = A ? (wc <= w2 ? wc : w2) : (w2 <= wc ? w2 : wc); }
        }
        int small = 0;
#ifndef ALD_STAR_NO_SMALL
        if(!bad && n <= STAR_SMALL) {           // phases 1 and 2 right here: merge order by creation id (insertion sort), centre weight, the fan edges' new weights
            small = 1;
            for(int i = 0; i < n; i++) { int k = i; const uint32_t id = uni(H.eid[fe[i]]); while(k > 0 && (uint32_t)uni(H.eid[fe[ord[k - 1]]]) > id) { ord[k] = ord[k - 1]; k--; } ord[k] = i; }
            double mdc = 0;
            for(int q = 0; q < n; q++) { inv[ord[q]] = q; const double w = fw[ord[q]]; if(ALD_UNLIKELY(!(w >= mw - kSMIN))) bad = ALD_ST_INVARIANT + ALD_INV_WEIGHT; mdc = (q == 0) ? w : mdc + w; }
            dctx[0] = mdc;
            for(int j = 0; j < n; j++) H.ed[fe[j]].w = fw[j];
        }
#endif
        ctx[SW_N] = n; ctx[SW_C] = c; ctx[SW_FAR] = A ? (int)uni(H.ed[c].lk.es) : (int)uni(H.ed[c].lk.et); ctx[SW_FAIL] = bad; ctx[SW_FAILQ] = -1; ctx[SW_SERIAL] = 0; ctx[SW_SMALL] = small;
    }
    wsync();
    const int n = uni(ctx[SW_N]), c = uni(ctx[SW_C]), far = uni(ctx[SW_FAR]);
#ifdef ALD_EMU_COUNT
    g_cnt_star[n < 33 ? n : 33]++;
#endif
    if(uni(ctx[SW_FAIL])) { if(lane == 0) fail(ctx[SW_FAIL]); wsync(); return; }
    PROF_ADD(PF_T_BALANCE);
    if(!uni(ctx[SW_SMALL])) {
    // ---- phase 1 (lane j): rank of fan edge j by creation id -> ord (merge order) and its inverse
    for(int j = lane; j < n; j += ALD_WAVE) {
        const uint32_t id = H.eid[fe[j]]; int r = 0;
        for(int k = 0; k < n; k++) r += ((uint32_t)H.eid[fe[k]] < id) ? 1 : 0;
        ord[r] = j; inv[j] = r;
    }
    wsync();
    // ---- phase 2 (lane 0): weight of c = sum of its pair weights in merge order; the fan edges take their pair weight
    if(lane == 0) {
        double mdc = 0; int bad = 0;
        for(int q = 0; q < n; q++) { double w = fw[ord[q]]; if(ALD_UNLIKELY(!(w >= mw - kSMIN))) bad = ALD_ST_INVARIANT + ALD_INV_WEIGHT; mdc = (q == 0) ? w : mdc + w; }
        dctx[0] = mdc;
        if(bad) ctx[SW_FAIL] = bad;
    }
    for(int j = lane; j < n; j += ALD_WAVE) H.ed[fe[j]].w = fw[j];
    wsync();
    if(uni(ctx[SW_FAIL])) { if(lane == 0) fail(ctx[SW_FAIL]); wsync(); return; }
    }
    PROF_ADD(PF_T_SETUP);
    const int id0 = uni(HC.next_id);
    const int meic = uni(meic_v), cntc = uni(cntc_v), stc = uni(stc_v), idc = uni(idc_v);
    const uint32_t nsc = uni(nsc_v);
    const int lt = uni(lt_v), rt = uni(rt_v), ov = uni(ov_v);
    // what is left of c before merge q, whether merge q cuts a piece off, the id of its merged edge: replayed by lane q
    #define SW_REPLAY(q_, wcur_, sc_, rem_, nid_, dead_) \
        double wcur_ = dctx[0], rem_ = 0; bool sc_ = false, dead_ = false; int nid_ = id0; \
        for(int qq = 0; qq <= (q_); qq++) { \
            if(qq > 0) { if(!sc_) dead_ = true; wcur_ = rem_; nid_ += 1; } \
            const double w2_ = fw[ord[qq]]; \
            sc_ = !(fabs(wcur_ - w2_) <= kSMIN); \
            rem_ = wcur_; if(sc_) { nid_ += 1; rem_ = wcur_ - w2_; if(rem_ <= mw) rem_ = mw; } \
        }
    // ---- phase 3 (lane q): the two weight sums around x at merge q, left in LDS for the vertex-weight chain
    // The fan edge's own record is asked for here and used in phase 5: a lane keeps ONE merge (n <= STAR_MAX <= the wave), so the
    // values stay in its registers across phase 4 and the round trip to L2 runs under it.  (The single-lane emulation walks all merges
    // in every phase and reads the record where it is used.)
    double pf_med = 0, pf_conf = 0, pf_abd = 0; int pf_mei = 0, pf_st = 0, pf_cnt = 0, pf_id = 0; uint32_t pf_ns = 0; uint64_t pf_mask0 = 0;
    double rp_wcur = 0; bool rp_sc = false, rp_dead = false; int rp_nid = 0;      // the lane's replay, kept for phase 5 as well (one merge per lane)
    for(int q = lane; q < n; q += ALD_WAVE) {
        SW_REPLAY(q, wcur, sc, rem, nid, dead);
        const int j = ord[q]; const double ww = fw[j];
        rp_wcur = wcur; rp_sc = sc; rp_dead = dead; rp_nid = nid;
#ifndef ALD_EMU
        { const int f = fe[j]; pf_med = C.ed[f].med; pf_conf = C.ed[f].econf; pf_abd = C.ed[f].s0abd; pf_mei = C.ed[f].mei; pf_st = C.ed[f].estrand; pf_cnt = C.ed[f].ecount;
          pf_id = C.ed[f].s0id; pf_ns = C.ed[f].sp_len; pf_mask0 = C.ed[f].mask[0]; }
#endif
        double sfan = 0;
        for(int k = 0; k < n; k++) if(inv[k] >= q) sfan += fw[k];             // not merged yet, list order
        double sc_side = 0; sc_side += sc ? rem : wcur; if(sc) sc_side += ww;
        sq[q] = A ? (sc_side + sfan) * 0.5 : (sfan + sc_side) * 0.5;
        oth[q] = A ? (int)H.ed[fe[j]].lk.et : (int)H.ed[fe[j]].lk.es;
    }
    wsync();
    PROF_ADD(PF_T_MERGE_LOAD);
    // ---- phase 4 (lane 0): the weight of x, merge by merge; sq[q] becomes r1 of merge q
    if(lane == 0) {
        double vwt = vw_early, wcur = dctx[0]; bool sc = false;          // (lane 0 asked for the vertex weight itself, at the top)
        for(int q = 0; q < n; q++) {
            const double ww = fw[ord[q]];
            sc = !(fabs(wcur - ww) <= kSMIN);                                     // split_edge(c, ww) cuts a piece off (scallop.cc:2433-2484)
            const double wc0 = sc ? ww : wcur;
            const double r1 = A ? vwt * (wc0 + ww) * 0.5 / sq[q] : vwt * (ww + wc0) * 0.5 / sq[q];
            vwt = vwt - r1; sq[q] = r1;
            if(sc) { double rem = wcur - ww; if(rem <= mw) rem = mw; wcur = rem; }
        }
        C.vx[x].vw = vwt;
        ctx[SW_SERIAL] = sc ? 0 : 1;                                              // the last merge consumed c
    }
    wsync();
    PROF_ADD(PF_T_MERGE_SUMS);
    // ---- phase 5 (lane q): the merged edge takes over the fan edge's slot -- cold record, id, weight, endpoint, place in the far
    // list of its other endpoint
    bool dup = false;
    for(int q = lane; q < n; q += ALD_WAVE) { const int o = oth[q]; for(int k = 0; k < q; k++) if(oth[k] == o) dup = true; }
    const bool any_dup = wballot(dup) != 0;
    bool multi = false, broken = false;
#ifdef ALD_PROF
    { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[PF_S5_DUP] += t1_ - prof_t_; }
    unsigned long long prof_s5_ = __builtin_readcyclecounter();
#endif
    for(int q = lane; q < n; q += ALD_WAVE) {
#ifdef ALD_EMU
        SW_REPLAY(q, wcur, sc, rem, nid, dead);
        (void)rem;
#else
        const double wcur = rp_wcur; const bool sc = rp_sc, dead = rp_dead; const int nid = rp_nid;
#endif
        const int j = ord[q], f = fe[j]; const double ww = fw[j];
        const double wc0 = sc ? ww : wcur;
        const double medc1 = sc ? medc * ww / wcur : medc;
        const double r1 = sq[q];
#ifdef ALD_EMU
        pf_med = C.ed[f].med; pf_conf = C.ed[f].econf; pf_abd = C.ed[f].s0abd; pf_mei = C.ed[f].mei; pf_st = C.ed[f].estrand; pf_cnt = C.ed[f].ecount;
        pf_id = C.ed[f].s0id; pf_ns = C.ed[f].sp_len; pf_mask0 = C.ed[f].mask[0];
#endif
        const double medf = pf_med, cf = pf_conf;
        const int meif = pf_mei, stf = pf_st, cntf = pf_cnt;
        const uint32_t nsf = pf_ns; const int idf = pf_id; const double abf = pf_abd;
        {   // what the sequential form checks when it reaches merge q, in its order: a consumed c, the id counter, the two counts
            int code = 0;
            if(dead) code = ALD_ST_INVARIANT + ALD_INV_OTHER;
            else if(nid - (sc ? 1 : 0) >= EID_LIMIT) code = ALD_ST_CAPACITY;
            else if(!(cntc > 0 && cntf > 0)) code = ALD_ST_INVARIANT + ALD_INV_COUNT;
            inv[q] = code; if(ALD_UNLIKELY(code)) broken = true;
        }
        if(nsc == 1 && nsf == 1) {
            if(idf == idc) { const double xa = A ? abc : abf, ya = A ? abf : abc; const double mn = (ya < xa) ? ya : xa; C.ed[f].sp_off = 0; C.ed[f].ecount = 1; C.ed[f].eabd = 0.0 + mn; C.ed[f].s0abd = mn; }
            else { C.ed[f].sp_off = 0; C.ed[f].sp_len = 0; C.ed[f].ecount = 0; C.ed[f].eabd = 0; C.ed[f].s0id = 0; C.ed[f].s0abd = 0; }
        } else multi = true;                                                  // pool allocation: sequential, below
        C.ed[f].econf = A ? cc + cf : cf + cc;
        { const int sty = A ? stf : stc, stx = A ? stc : stf; C.ed[f].estrand = (uint8_t)(sty != 0 ? sty : stx); }
        for(int k = 0; k < NW; k++) { uint64_t mk = (NW <= 2 ? cmask_pf[NW <= 2 ? k : 0] : C.ed[c].mask[k]) | (k == 0 ? pf_mask0 : C.ed[f].mask[k]); if(ov >= 0 && (ov >> 6) == k) mk |= (1ull << (ov & 63)); C.ed[f].mask[k] = mk; }
        const int mi = A ? rt - lt + meic + meif : rt - lt + meif + meic;
        C.ed[f].med = A ? mi * r1 + medc1 + medf : mi * r1 + medf + medc1; C.ed[f].mei = mi;
        H.eid[f] = (EID)nid; H.ed[f].w = A ? wc0 * 0.5 + ww * 0.5 : ww * 0.5 + wc0 * 0.5;
        if(A) H.ed[f].lk.es = (IDX)far; else H.ed[f].lk.et = (IDX)far;
#ifdef ALD_PROF
        { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[PF_S5_BODY] += t1_ - prof_s5_; prof_s5_ = t1_; }
#endif
        if(!any_dup) { if(A) relink_in_lane(oth[q], f, (uint32_t)far); else relink_out_lane(oth[q], f, tkey((uint32_t)far)); }
#ifdef ALD_PROF
        { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[PF_S5_RELINK] += t1_ - prof_s5_; prof_s5_ = t1_; }
#endif
    }
    const bool any_multi = wballot(multi) != 0;
    if(ALD_UNLIKELY(wballot(broken) != 0)) {                                    // the failure the sequence meets first: the smallest q that has one
        wsync();
        if(lane == 0) for(int q = 0; q < n; q++) if(inv[q]) { fail(inv[q]); break; }
        wsync();
        return;
    }
    // far's list: c leaves it now (lane 0, while nothing else touches a list of that kind), the merged edges enter it below
    const bool counted = A ? (far == 0 && !uni(HC.special_linked)) : (far == (int)uni(HC.sinkp) && !uni(HC.special_linked));     // out(source) / in(sink) are only counted
    const bool consumed = uni(ctx[SW_SERIAL]) != 0;
    if(lane == 0 && consumed) { if(A) unlink_out(far, c); else unlink_in(far, c); }
    wsync();
    PROF_ADD(PF_T_MERGE_MASK);
    if(ALD_UNLIKELY(!consumed)) { if(lane == 0) fail(ALD_ST_INVARIANT + ALD_INV_DEGREE); wsync(); return; }     // c kept a remainder: the reference asserts on the degree of x
    // ---- phase 6 (lane q): place of merged edge q in far's list = behind the last old entry whose key does not exceed its own
    // (old entries carry older ids), and among the new ones by (key, creation order).  Read-only walk; the links are written in 6c.
    int32_t *pred = inv, *succ = (int32_t*)sq, *srt = (int32_t*)sq + STAR_MAX;
    if(!counted) {
        for(int q = lane; q < n; q += ALD_WAVE) {
            const uint32_t key = A ? tkey((uint32_t)oth[q]) : (uint32_t)oth[q];
            int last = -1, cur = A ? first_out(far) : first_in(far), guard = MAXE;
            while(cur >= 0 && guard-- > 0) {
                const uint64_t w = *(const uint64_t*)&H.ed[cur].lk;
                const uint32_t kc = A ? tkey((uint32_t)((w >> 16) & 0xFFFF)) : (uint32_t)(w & 0xFFFF);
                if(kc > key) break;
                last = cur; cur = A ? lk_next((uint32_t)(w >> 48)) : lk_next((uint32_t)((w >> 32) & 0xFFFF));
            }
            int r = 0;
            for(int k = 0; k < n; k++) { const uint32_t kk = A ? tkey((uint32_t)oth[k]) : (uint32_t)oth[k]; r += (kk < key || (kk == key && k < q)) ? 1 : 0; }
            pred[q] = last; succ[q] = cur; srt[r] = q;
        }
        wsync();
#ifdef ALD_PROF
        { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[PF_S6_WALK] += t1_ - prof_t_; }
#endif
        for(int r = lane; r < n; r += ALD_WAVE) {
            const int q = srt[r], f = fe[ord[q]];
            const bool first_of_gap = (r == 0) || pred[srt[r - 1]] != pred[q], last_of_gap = (r + 1 >= n) || pred[srt[r + 1]] != pred[q];
            const IDX nx = last_of_gap ? (succ[q] < 0 ? NIL : (IDX)succ[q]) : (IDX)fe[ord[srt[r + 1]]];
            if(A) H.ed[f].lk.onx = nx; else H.ed[f].lk.inx = nx;
            if(first_of_gap) { if(pred[q] < 0) { if(A) H.vx[far].out_head = (IDX)f; else H.vx[far].in_head = (IDX)f; } else { if(A) H.ed[pred[q]].lk.onx = (IDX)f; else H.ed[pred[q]].lk.inx = (IDX)f; } }
        }
    }
    star_tail_sync();
#ifdef ALD_PROF
    unsigned long long prof_s7_ = __builtin_readcyclecounter();
#endif
    // ---- phase 7 (lane 0): what is left and inherently ordered -- the support pool, the phasing lists, the counters
    if(lane == 0) {
        { const int dg = A ? (int)uni(H.vx[far].out_deg) : (int)uni(H.vx[far].in_deg); if(A) H.vx[far].out_deg = (IDX)(dg + n); else H.vx[far].in_deg = (IDX)(dg + n); ev_degree(far, dg, dg + n, A); }
        if(any_dup || any_multi || uni(HC.hl_n) != 0) for(int q = 0; q < n; q++) {
            const int f = fe[ord[q]];
            if(any_dup) { if(A) relink_in(oth[q], f, (uint32_t)far); else relink_out(oth[q], f, tkey((uint32_t)far)); }
            if(any_multi) { const uint32_t nsf = uni(C.ed[f].sp_len); if(!(nsc == 1 && nsf == 1)) { if(!(A ? intersect_samples(c, f, f) : intersect_samples(f, c, f))) break; } }
            if(A) hs_replace2(c, f, f); else hs_replace2(f, c, f);
            if(n == 1) hs_replace1(c, f);
        }
        HC.next_id = (int)uni(H.eid[fe[ord[n - 1]]]) + 1;
        if(n >= 2) hs_remove(c);
        // remove_edge(c) (already out of far's list); x is left without edges
        H.ed[c].lk.es = NIL; H.hflag[c] = 0;
        { int fh = uni(HC.free_head); H.ed[c].lk.onx = fh < 0 ? NIL : (IDX)fh; HC.free_head = c; HC.free_cnt = uni(HC.free_cnt) + 1; }
        clear_vertex(x);
    }
    wsync();
#ifdef ALD_PROF
    { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[PF_S7] += t1_ - prof_s7_; }
#endif
    PROF_ADD(PF_T_MERGE_ADD);
#if defined(ALD_PROF) && defined(ALD_PROF_STAR_BY_SIZE)
    { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[n == 1 ? PF_T_PAIRS : n == 2 ? PF_T_MERGE_ISECT : n == 3 ? PF_T_MERGE_KILL : n == 4 ? PF_S6_LINK : PF_FINISH] += t1_ - prof_star0_; }
#endif
    #undef SW_REPLAY
}
ALD_INL void star_wave_in(int x) { star_wave_body<true>(uni(x)); }       // inlined into the kernel entry, once (sweep_trivial has ONE decomposition site,
ALD_INL void star_wave_out(int x) { star_wave_body<false>(uni(x)); }     // run_graph ONE sweep_trivial): no prologue that parks callee-saved registers in scratch
#ifndef ALD_EMU
// ---------------------------------------------------------------------------------------------------------------------------------
// The decomposition by the whole wave ONCE MORE (round 4), with the fan in REGISTERS: lane k holds fan edge k -- slot, weight, creation id,
// other endpoint -- for the whole routine, and what one lane needs of another's travels through v_readlane (a few cycles, no memory),
// not through the LDS scratch.  star_wave_body hands every intermediate array from phase to phase through LDS: eight hand-overs, each a
// wsync() in front of a loop of DEPENDENT LDS round trips (fw[ord[q]]: two per step), 150-190 cycles a trip at this occupancy -- 20 000
// cycles for a fan of three, 36 000 for the average fan above four (profiles/r03/zl_phase_star_by_size.txt), 22 % of the kernel.  Here:
//   * the fan list is walked ONCE, in lock step, lane k keeping the k-th edge; every edge's record is then read in one round;
//   * sums that the reference takes in list order or in merge order (the balance step's side sums, the weight c starts with) are
//     uniform loops over readlane'd values -- the same additions in the same order;
//   * merge order: rank of the lane's creation id among the fan's, by comparing against every lane's id in turn;
//   * "what is left of c before merge q", the ids, the two sums around x and the vertex-weight chain -- the recurrences that tie the
//     merges together -- run ONCE, as one uniform loop over the merges (every lane computes the same values in lock step, the lane that
//     owns merge q keeps them); star_wave_body has every lane replay the recurrence up to its own merge out of LDS and a single lane run
//     the chain behind another hand-over;
//   * every lane then finishes ITS merged edge exactly as phase 5 does.
// Only the places in far's list still go through the scratch (two small hand-overs): the links are written by rank, which is a permutation.
// Every floating-point operation is the one the sequential form performs (scallop.cc:2144-2167 -> 2486-2576, 2009-2142, 2394-2484,
// 2242-2378), on the same operands in the same order.  The single-lane emulation cannot run this form (it has no lanes to read from): it
// runs star_wave_body, the same arithmetic through arrays; the GPU tier checks this one against the oracle.
template<bool A> ALD_INL void star_reg(int x)
{
    COLD;
    PROF_DECL;
    const int lane = lane_id();
    const double mw = HC.p_min_w;
#if defined(ALD_PROF) && defined(ALD_PROF_STAR_BY_SIZE)
    const unsigned long long prof_star0_ = __builtin_readcyclecounter();
#endif
    const Hot::VertexHot vrx = H.vx[x];
    const int c = uni(slot_or_neg(A ? vrx.in_head : vrx.out_head));
    const int n = uni((int)(A ? vrx.out_deg : vrx.in_deg));                  // 1 .. STAR_MAX: the caller chose the form from the degrees
    // ---- the fan: ONE walk in lock step, lane k keeps fan edge k; then every edge's record in one round of loads
    int f = c;
    { int e = uni(slot_or_neg(A ? vrx.out_head : vrx.in_head));
      for(int k = 0; k < n && e >= 0; k++) { if(lane == k) f = e; e = A ? u_next_out(e) : u_next_in(e); } }
    const bool act = lane < n;
    const double w0 = H.ed[f].w; const uint64_t lw = *(const uint64_t*)&H.ed[f].lk; const int id = (int)H.eid[f];
    const int oth = A ? (int)((lw >> 16) & 0xFFFF) : (int)(lw & 0xFFFF);
    const uint64_t lwc = lkw(c);
    const int far = A ? (int)(lwc & 0xFFFF) : (int)((lwc >> 16) & 0xFFFF);
    double wcen = H.ed[c].w;
    bool badw = !(wcen >= mw - kSMIN) || wballot(act && !(w0 >= mw - kSMIN)) != 0;
    // ---- balance_vertex(x) (scallop.cc:2486-2576): the side sums in list order, the scale factors, the clamps, the remainder fix-up
    double sfan0 = 0;
    for(int k = 0; k < n; k++) sfan0 += wread(w0, k);
    double fw;
    {
        double scen0 = 0; scen0 += wcen;
        const double w_in = A ? scen0 : sfan0, w_out = A ? sfan0 : scen0;
        const double bw = sqrt(w_in * w_out);
        const double r_in = bw / w_in, r_out = bw / w_out;
        double m_cen = 0, m_fan = 0;
        { double wy = wcen * (A ? r_in : r_out); if(wy < mw) { m_cen += mw - wy; wy = mw; } wcen = wy; }
        fw = w0 * (A ? r_out : r_in);
        const bool clamped = act && fw < mw; const double deficit = mw - fw;
        if(clamped) fw = mw;
        { uint64_t m = wballot(clamped); while(m) { const int l = ffs64(m); m &= m - 1; m_fan += wread(deficit, l); } }      // (list order; nearly always empty)
        const double m1 = A ? m_cen : m_fan, m2 = A ? m_fan : m_cen;
        if(m1 > m2) { if(A) { if(lane == 0) fw = fw + m1 - m2; } else wcen = wcen + m1 - m2; }
        else if(m1 < m2) { if(A) wcen = wcen + m2 - m1; else { if(lane == 0) fw = fw + m2 - m1; } }
        const double wc = wcen;
        fw = A ? (wc <= fw ? wc : fw) : (fw <= wc ? fw : wc);                   // pe2w of (c, fan edge): the lane's pair weight
    }
    PROF_ADD(PF_T_BALANCE);
    // ---- merge order = ascending creation id; the weight c starts with = the pair weights summed in that order
    int inv = 0;
    for(int k = 0; k < n; k++) { const int idk = wread(id, k); inv += ((uint32_t)idk < (uint32_t)id) ? 1 : 0; }
    double mdc = 0;
    for(int q = 0; q < n; q++) {
        const int src = ffs64(wballot(act && inv == q));
        const double pw = wread(fw, src < 0 ? 0 : src);
        if(!(pw >= mw - kSMIN)) badw = true;
        mdc = (q == 0) ? pw : mdc + pw;
    }
    if(ALD_UNLIKELY(uni(badw))) { if(lane == 0) fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); wsync(); return; }      // nothing has been written
    PROF_ADD(PF_T_SETUP);
    // c's record and the vertex (one broadcast request each) and the lane's own fan edge: asked for here, used behind the merge loop
    const double medc = C.ed[c].med, cc = C.ed[c].econf, abc = C.ed[c].s0abd, vw0 = C.vx[x].vw;
    const int meic_v = C.ed[c].mei, cntc_v = C.ed[c].ecount, stc_v = C.ed[c].estrand, idc_v = C.ed[c].s0id;
    const uint32_t nsc_v = C.ed[c].sp_len;
    const int lt_v = C.vx[x].lpos, rt_v = C.vx[x].rpos, ov_v = C.vx[x].v2v;
    uint64_t cmask_pf[NW <= 2 ? NW : 1];
    if(NW <= 2) for(int k = 0; k < NW; k++) cmask_pf[k] = C.ed[c].mask[k];
    const double pf_med = C.ed[f].med, pf_conf = C.ed[f].econf, pf_abd = C.ed[f].s0abd;
    const int pf_mei = C.ed[f].mei, pf_st = C.ed[f].estrand, pf_cnt = C.ed[f].ecount, pf_id = C.ed[f].s0id;
    const uint32_t pf_ns = C.ed[f].sp_len; const uint64_t pf_mask0 = C.ed[f].mask[0];
    // ---- the sum of the fan weights not merged before the lane's own merge, list order (get_in/out_weights(x) at that merge)
    double mysfan = 0;
    for(int k = 0; k < n; k++) { const double fwk = wread(fw, k); const int invk = wread(inv, k); if(invk >= inv) mysfan += fwk; }
    // ---- ONE pass over the merges, in merge order: what is left of c, whether the merge cuts a piece off, the ids, the sums around x, the
    // weight x loses to the merged edge (scallop.cc:2242-2378).  Uniform work; the lane that owns merge q keeps q's values.
    const int id0 = uni(HC.next_id);
    double vwt = vw0, wcur = mdc, rem = 0; bool sc = false; int nid = id0;
    int failq = -1, failcode = 0;
    double my_wcur = 0, my_r1 = 0; bool my_sc = false; int my_nid = 0;
    for(int q = 0; q < n; q++) {
        const int src0 = ffs64(wballot(act && inv == q)), src = src0 < 0 ? 0 : src0;
        const double ww = wread(fw, src);
        if(q > 0) { if(!sc && failq < 0) { failq = q; failcode = ALD_ST_INVARIANT + ALD_INV_OTHER; } wcur = rem; nid += 1; }      // (the general form would be handed a dead edge here)
        sc = uni(!(fabs(wcur - ww) <= kSMIN));                                // split_edge(c, ww) cuts a piece off (scallop.cc:2433-2484)
        rem = wcur; if(sc) { nid += 1; rem = wcur - ww; if(rem <= mw) rem = mw; }
        if(failq < 0 && nid - (sc ? 1 : 0) >= EID_LIMIT) { failq = q; failcode = ALD_ST_CAPACITY; }
        const double sfan = wread(mysfan, src);
        double sc_side = 0; sc_side += sc ? rem : wcur; if(sc) sc_side += ww;
        const double sum = A ? (sc_side + sfan) * 0.5 : (sfan + sc_side) * 0.5;
        const double wc0 = sc ? ww : wcur;
        const double r1 = A ? vwt * (wc0 + ww) * 0.5 / sum : vwt * (ww + wc0) * 0.5 / sum;
        vwt = vwt - r1;
        if(lane == src) { my_wcur = wcur; my_sc = sc; my_nid = nid; my_r1 = r1; }
    }
    const bool consumed = !sc;                                                 // the last merge used c up
    const int nid_last = nid;
    PROF_ADD(PF_T_MERGE_SUMS);
    const int meic = uni(meic_v), cntc = uni(cntc_v), stc = uni(stc_v), idc = uni(idc_v);
    const uint32_t nsc = uni(nsc_v);
    const int lt = uni(lt_v), rt = uni(rt_v), ov = uni(ov_v);
    {   // what the sequential form checks when it reaches merge q, in its order: a consumed c, the id counter, the two counts -- the
        // failure it meets first is the one of the smallest q
        int cq = -1;
        if(!(cntc > 0)) cq = 0;
        else { uint64_t m = wballot(act && !(pf_cnt > 0)); while(m) { const int l = ffs64(m); m &= m - 1; const int ql = wread(inv, l); if(cq < 0 || ql < cq) cq = ql; } }
        int code = 0;
        if(failq >= 0 && (cq < 0 || failq <= cq)) code = failcode; else if(cq >= 0) code = ALD_ST_INVARIANT + ALD_INV_COUNT;
        if(ALD_UNLIKELY(code != 0)) { if(lane == 0) { C.vx[x].vw = vwt; fail(code); } wsync(); return; }
    }
    if(lane == 0) C.vx[x].vw = vwt;
    if(ALD_UNLIKELY(!consumed)) { if(lane == 0) fail(ALD_ST_INVARIANT + ALD_INV_DEGREE); wsync(); return; }     // c kept a remainder: the reference asserts on the degree of x
    // two fan edges to one vertex: their relinks share a list -> sequential, in merge order, at the end
    bool dup = false;
    for(int k = 0; k < n; k++) { const int othk = wread(oth, k); if(act && lane != k && othk == oth) dup = true; }
    const bool any_dup = wballot(dup) != 0;
    // ---- lane k: fan edge k becomes merged edge inv -- cold record, id, weight, endpoint, place in the list of its other endpoint
    int32_t *fem = (int32_t*)HC.scr_i, *othm = fem + STAR_MAX, *pred = fem + 2 * STAR_MAX, *succ = fem + 3 * STAR_MAX, *srt = (int32_t*)HC.scr_d;
    bool multi = false;
    if(act) {
        const double ww = fw, wc0 = my_sc ? ww : my_wcur;
        const double medc1 = my_sc ? medc * ww / my_wcur : medc;
        if(nsc == 1 && pf_ns == 1) {
            if(pf_id == idc) { const double xa = A ? abc : pf_abd, ya = A ? pf_abd : abc; const double mn = (ya < xa) ? ya : xa; C.ed[f].sp_off = 0; C.ed[f].ecount = 1; C.ed[f].eabd = 0.0 + mn; C.ed[f].s0abd = mn; }
            else { C.ed[f].sp_off = 0; C.ed[f].sp_len = 0; C.ed[f].ecount = 0; C.ed[f].eabd = 0; C.ed[f].s0id = 0; C.ed[f].s0abd = 0; }
        } else multi = true;                                                  // pool allocation: sequential, below
        C.ed[f].econf = A ? cc + pf_conf : pf_conf + cc;
        { const int sty = A ? pf_st : stc, stx = A ? stc : pf_st; C.ed[f].estrand = (uint8_t)(sty != 0 ? sty : stx); }
        for(int k = 0; k < NW; k++) { uint64_t mk = (NW <= 2 ? cmask_pf[NW <= 2 ? k : 0] : C.ed[c].mask[k]) | (k == 0 ? pf_mask0 : C.ed[f].mask[k]); if(ov >= 0 && (ov >> 6) == k) mk |= (1ull << (ov & 63)); C.ed[f].mask[k] = mk; }
        const int mi = A ? rt - lt + meic + pf_mei : rt - lt + pf_mei + meic;
        C.ed[f].med = A ? mi * my_r1 + medc1 + pf_med : mi * my_r1 + pf_med + medc1; C.ed[f].mei = mi;
        H.eid[f] = (EID)my_nid; H.ed[f].w = A ? wc0 * 0.5 + ww * 0.5 : ww * 0.5 + wc0 * 0.5;
        if(A) H.ed[f].lk.es = (IDX)far; else H.ed[f].lk.et = (IDX)far;
        if(!any_dup) { if(A) relink_in_lane(oth, f, (uint32_t)far); else relink_out_lane(oth, f, tkey((uint32_t)far)); }
        fem[inv] = f; othm[inv] = oth;                                        // (merge order: what the tail and the link pass read)
    }
    const bool any_multi = wballot(multi) != 0;
    PROF_ADD(PF_T_MERGE_LOAD);
    // far's list: c leaves it now (lane 0, while nothing else touches a list of that kind), the merged edges enter it below
    const bool counted = A ? (far == 0 && !uni(HC.special_linked)) : (far == (int)uni(HC.sinkp) && !uni(HC.special_linked));     // out(source) / in(sink) are only counted
    if(lane == 0) { if(A) unlink_out(far, c); else unlink_in(far, c); }
    wsync();
    PROF_ADD(PF_T_MERGE_MASK);
    // ---- the place of merged edge q in far's list = behind the last old entry whose key does not exceed its own (old entries carry older
    // ids), and among the new ones by (key, creation order).  Read-only walk; the links are written by rank after one hand-over.
    if(!counted) {
        const uint32_t key = A ? tkey((uint32_t)oth) : (uint32_t)oth;
        int r = 0;
        for(int k = 0; k < n; k++) { const uint32_t kk = (uint32_t)wread((int)key, k); const int invk = wread(inv, k); r += (kk < key || (kk == key && invk < inv)) ? 1 : 0; }
        if(act) {
            int last = -1, cur = A ? first_out(far) : first_in(far), guard = MAXE;
            while(cur >= 0 && guard-- > 0) {
                const uint64_t w = *(const uint64_t*)&H.ed[cur].lk;
                const uint32_t kc = A ? tkey((uint32_t)((w >> 16) & 0xFFFF)) : (uint32_t)(w & 0xFFFF);
                if(kc > key) break;
                last = cur; cur = A ? lk_next((uint32_t)(w >> 48)) : lk_next((uint32_t)((w >> 32) & 0xFFFF));
            }
            pred[inv] = last; succ[inv] = cur; srt[r] = inv;
        }
        wsync();
#if defined(ALD_PROF) && !defined(ALD_PROF_STAR_BY_SIZE)
        PROF_ADD(PF_S6_WALK);                    // (profiling build: the walk over far's list, then the links, then -- M_add -- the lane-0 tail)
#endif
        for(int r2 = lane; r2 < n; r2 += ALD_WAVE) {
            const int q = srt[r2], fq = fem[q];
            const bool first_of_gap = (r2 == 0) || pred[srt[r2 - 1]] != pred[q], last_of_gap = (r2 + 1 >= n) || pred[srt[r2 + 1]] != pred[q];
            const IDX nx = last_of_gap ? (succ[q] < 0 ? NIL : (IDX)succ[q]) : (IDX)fem[srt[r2 + 1]];
            if(A) H.ed[fq].lk.onx = nx; else H.ed[fq].lk.inx = nx;
            if(first_of_gap) { if(pred[q] < 0) { if(A) H.vx[far].out_head = (IDX)fq; else H.vx[far].in_head = (IDX)fq; } else { if(A) H.ed[pred[q]].lk.onx = (IDX)fq; else H.ed[pred[q]].lk.inx = (IDX)fq; } }
        }
    }
    star_tail_sync();
#if defined(ALD_PROF) && !defined(ALD_PROF_STAR_BY_SIZE)
    PROF_ADD(PF_S6_LINK);
#endif
    // ---- lane 0: what is left and inherently ordered -- the support pool, the phasing lists, the counters
    if(lane == 0) {
        { const int dg = A ? (int)uni(H.vx[far].out_deg) : (int)uni(H.vx[far].in_deg); if(A) H.vx[far].out_deg = (IDX)(dg + n); else H.vx[far].in_deg = (IDX)(dg + n); ev_degree(far, dg, dg + n, A); }
        if(any_dup || any_multi || uni(HC.hl_n) != 0) for(int q = 0; q < n; q++) {
            const int fq = fem[q];
            if(any_dup) { if(A) relink_in(othm[q], fq, (uint32_t)far); else relink_out(othm[q], fq, tkey((uint32_t)far)); }
            if(any_multi) { const uint32_t nsf = uni(C.ed[fq].sp_len); if(!(nsc == 1 && nsf == 1)) { if(!(A ? intersect_samples(c, fq, fq) : intersect_samples(fq, c, fq))) break; } }
            if(A) hs_replace2(c, fq, fq); else hs_replace2(fq, c, fq);
            if(n == 1) hs_replace1(c, fq);
        }
        HC.next_id = nid_last + 1;
        if(n >= 2) hs_remove(c);
        // remove_edge(c) (already out of far's list); x is left without edges
        H.ed[c].lk.es = NIL; H.hflag[c] = 0;
        { int fh = uni(HC.free_head); H.ed[c].lk.onx = fh < 0 ? NIL : (IDX)fh; HC.free_head = c; HC.free_cnt = uni(HC.free_cnt) + 1; }
        clear_vertex(x);
    }
    wsync();
    PROF_ADD(PF_T_MERGE_ADD);
#if defined(ALD_PROF) && defined(ALD_PROF_STAR_BY_SIZE)
    { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[n == 1 ? PF_T_PAIRS : n == 2 ? PF_T_MERGE_ISECT : n == 3 ? PF_T_MERGE_KILL : n == 4 ? PF_S6_LINK : PF_FINISH] += t1_ - prof_star0_; }
#endif
}
#endif
// ---------------------------------------------------------------------------------------------------------------------------------
// The same decomposition once more, written for a fan of EXACTLY N = 2..4 edges (two thirds of the stars of the bench workload), with
// nothing handed from lane to lane: every lane gathers the whole fan and runs the whole numeric part -- balance_vertex, pair
// weights, merge order, what is left of c after every cut, the ids, the sums around x, the vertex-weight chain -- on named values
// in registers (N is a template parameter, every loop over the fan unrolls); the lanes execute in lock step, so computing it in 64
// lanes costs what computing it in one does.  Lane j then finishes fan edge j: its cold record (asked for at the top, the round trip
// runs under the arithmetic), id / weight / endpoint, its place in the list of its other endpoint, and its link into far's list.  The
// places in far's list come from ONE walk every lane makes for all N keys at once; the same walk finds c's predecessor, so c leaves the
// list without a walk of its own.  No scratch, no hand-over until the final one.
// Anything out of the ordinary -- a weight below min_w, two fan edges to the same vertex, an exhausted id space, a zero count, a c that
// keeps a remainder, an inconsistent list -- is detected BEFORE the first write and makes the form decline (false): the caller then runs
// star_wave_body, which treats or reports the case exactly as before.  Every floating-point operation is the one the sequential
// form performs (scallop.cc:2144-2167 -> 2486-2576, 2009-2142, 2394-2484, 2242-2378), on the same operands in the same order.
#ifndef ALD_STARFIX_MAX
  // classes 2..9 run one to three waves per SIMD (their LDS footprint caps the occupancy) and have 132..272 VGPRs each: fans of up to four
  // edges go through this form there without a spill; the twins of classes 7 / 8 (168 VGPRs) take it as well (mixed batch 108.7 -> 107.0
  // -> 106.5 ms, profiles/r03/zk_*).  Class 1 (96 VGPRs at five waves) and class 0 (80 at six) keep it for two edges: three / four inline
  // cost the kernel root 15 / 55 spilled VGPRs (42.4 / 43.4 against 40.65 ms, profiles/r03/za_*, zh_*).
  #if ((ALD_CLASS_ID >= 2 && ALD_CLASS_ID <= 9) || ALD_CLASS_ID == 11 || ALD_CLASS_ID == 12) && !defined(ALD_STARFIX_ROOMY_OFF)
    #define ALD_STARFIX_MAX 4
  #else
    #define ALD_STARFIX_MAX 2
  #endif
#endif
// a[i] for a run-time i as a chain of selects over constant indices.  Written by template recursion, not as a loop: the optimiser turns the loop
// "for k: if(i == k) v = a[k]" back into an indexed load, and an array indexed at run time lives in scratch memory (what made the fans of
// three and four edges slow in this form: 48 scratch loads and 104 scratch stores in a kernel without a single spilled register).
template<int K, int N, class T> struct PickFrom { ALD_INL T get(const T (&a)[N], int i, T v) { return PickFrom<K + 1, N, T>::get(a, i, i == K ? a[K] : v); } };
template<int N, class T> struct PickFrom<N, N, T> { ALD_INL T get(const T (&)[N], int, T v) { return v; } };
template<int N, class T> ALD_INL T pick(const T (&a)[N], int i) { return PickFrom<1, N, T>::get(a, i, a[0]); }
#ifdef ALD_STARFIX_CALL
template<bool A, int N> ALD_FN bool star_fixed(int x)
#else
template<bool A, int N> ALD_INL bool star_fixed(int x)
#endif
{
    COLD;
    PROF_DECL;
    const int lane = lane_id();
    const double mw = HC.p_min_w;
    const Hot::VertexHot vrx = H.vx[x];           // both list heads of x in one LDS read
    const int c = uni(slot_or_neg(A ? vrx.in_head : vrx.out_head));
    int bad = (c < 0) ? 1 : 0;
    const int cs = c >= 0 ? c : 0;
    // the fan in list order
    int fe[N], oth[N]; double fw[N]; uint32_t id[N];
    {
        int e = slot_or_neg(A ? vrx.out_head : vrx.in_head);
        ALD_UNROLL for(int k = 0; k < N; k++) {
            if(e < 0) bad = 1;
            const int es = e >= 0 ? e : 0;
            const double w = H.ed[es].w; const uint64_t lw = *(const uint64_t*)&H.ed[es].lk;
            fe[k] = es; fw[k] = w; id[k] = H.eid[es];
            oth[k] = A ? (int)((lw >> 16) & 0xFFFF) : (int)(lw & 0xFFFF);
            e = A ? lk_next((uint32_t)(lw >> 48)) : lk_next((uint32_t)((lw >> 32) & 0xFFFF));
        }
        if(e >= 0) bad = 1;                       // (the caller chose N from the degree)
    }
    const uint64_t lwc = *(const uint64_t*)&H.ed[cs].lk;
    const int far = uni(A ? (int)(lwc & 0xFFFF) : (int)((lwc >> 16) & 0xFFFF));
    const int c_next = A ? lk_next((uint32_t)(lwc >> 48)) : lk_next((uint32_t)((lwc >> 32) & 0xFFFF));     // c's successor in far's list
    bool dupf = false;                            // two fan edges to one vertex: their relinks share a list -> sequential, in merge order, at the end
    ALD_UNROLL for(int k = 0; k < N; k++) { ALD_UNROLL for(int k2 = 0; k2 < N; k2++) if(k2 < k && oth[k] == oth[k2]) dupf = true; }      // (every loop over the fan must unroll: an array indexed at run time lives in scratch memory)
    dupf = uni(dupf);
    // ---- balance_vertex(x) on the gathered weights (scallop.cc:2486-2576), pair weights
    double wcen = H.ed[cs].w;
    if(!(wcen >= mw - kSMIN)) bad = 1;
    double sfan0 = 0;
    ALD_UNROLL for(int k = 0; k < N; k++) { if(!(fw[k] >= mw - kSMIN)) bad = 1; sfan0 += fw[k]; }
    {
        double scen0 = 0; scen0 += wcen;
        const double w_in = A ? scen0 : sfan0, w_out = A ? sfan0 : scen0;
        const double bw = sqrt(w_in * w_out);
        const double r_in = bw / w_in, r_out = bw / w_out;
        double m_cen = 0, m_fan = 0;
        { double wy = wcen * (A ? r_in : r_out); if(wy < mw) { m_cen += mw - wy; wy = mw; } wcen = wy; }
        ALD_UNROLL for(int k = 0; k < N; k++) { double wy = fw[k] * (A ? r_out : r_in); if(wy < mw) { m_fan += mw - wy; wy = mw; } fw[k] = wy; }
        const double m1 = A ? m_cen : m_fan, m2 = A ? m_fan : m_cen;
        if(m1 > m2) { if(A) fw[0] = fw[0] + m1 - m2; else wcen = wcen + m1 - m2; }
        else if(m1 < m2) { if(A) wcen = wcen + m2 - m1; else fw[0] = fw[0] + m2 - m1; }
        const double wc = wcen;
        ALD_UNROLL for(int k = 0; k < N; k++) { const double w2 = fw[k]; fw[k] = A ? (wc <= w2 ? wc : w2) : (w2 <= wc ? w2 : wc); }
    }
    // c's record, the vertex (one broadcast request each) and, lane j, the record of fan edge j: asked for HERE -- behind the square root
    // and the divisions of the balance step, whose temporaries would otherwise share the register file with two dozen pending values
    // (the kernel root spilled eleven more VGPRs with the requests at the top) -- and used in the write phase; the round trip runs under
    // the replay, the sums, the vertex-weight chain and the walk over far's list.
    const double medc = C.ed[cs].med, cc = C.ed[cs].econf, abc = C.ed[cs].s0abd, vw0 = C.vx[x].vw;
    const int meic_v = C.ed[cs].mei, cntc_v = C.ed[cs].ecount, stc_v = C.ed[cs].estrand, idc_v = C.ed[cs].s0id;
    const uint32_t nsc_v = C.ed[cs].sp_len;
    const int lt_v = C.vx[x].lpos, rt_v = C.vx[x].rpos, ov_v = C.vx[x].v2v;
    uint64_t cmask_pf[NW <= 2 ? NW : 1];
    if(NW <= 2) { ALD_UNROLL for(int k = 0; k < (NW <= 2 ? NW : 1); k++) cmask_pf[k] = C.ed[cs].mask[k]; }
#ifndef ALD_EMU
    double pf_med, pf_conf, pf_abd; int pf_mei, pf_st, pf_cnt, pf_id; uint32_t pf_ns; uint64_t pf_mask0;
    { const int f = pick<N>(fe, lane < N ? lane : 0); pf_med = C.ed[f].med; pf_conf = C.ed[f].econf; pf_abd = C.ed[f].s0abd; pf_mei = C.ed[f].mei; pf_st = C.ed[f].estrand;
      pf_cnt = C.ed[f].ecount; pf_id = C.ed[f].s0id; pf_ns = C.ed[f].sp_len; pf_mask0 = C.ed[f].mask[0]; }
#endif
    // ---- merge order = ascending creation id; pair weights in that order; the weight c starts with
    int inv[N];                                   // list position -> merge index
    ALD_UNROLL for(int k = 0; k < N; k++) { int r = 0; ALD_UNROLL for(int k2 = 0; k2 < N; k2++) r += (id[k2] < id[k]) ? 1 : 0; inv[k] = r; }
    double pw[N]; int fq[N], oq[N];               // merge index -> pair weight, fan edge, its other endpoint
    ALD_UNROLL for(int q = 0; q < N; q++) { double w = fw[0]; int f = fe[0], o = oth[0]; ALD_UNROLL for(int k = 1; k < N; k++) if(inv[k] == q) { w = fw[k]; f = fe[k]; o = oth[k]; } pw[q] = w; fq[q] = f; oq[q] = o; }
    double mdc = 0;
    ALD_UNROLL for(int q = 0; q < N; q++) { if(!(pw[q] >= mw - kSMIN)) bad = 1; mdc = (q == 0) ? pw[q] : mdc + pw[q]; }
    // ---- what is left of c before merge q, whether merge q cuts a piece off, the id of its merged edge
    const int id0 = uni(HC.next_id);
    double wcur_q[N], rem_q[N]; int sc_q[N], nid_q[N];
    {
        double wcur = mdc, rem = 0; bool sc = false; int nid = id0;
        ALD_UNROLL for(int q = 0; q < N; q++) {
            if(q > 0) { if(!sc) bad = 1; wcur = rem; nid += 1; }                // (a consumed c in front of the last merge)
            sc = !(fabs(wcur - pw[q]) <= kSMIN);
            rem = wcur; if(sc) { nid += 1; rem = wcur - pw[q]; if(rem <= mw) rem = mw; }
            if(nid - (sc ? 1 : 0) >= EID_LIMIT) bad = 1;
            wcur_q[q] = wcur; rem_q[q] = rem; sc_q[q] = sc ? 1 : 0; nid_q[q] = nid;
        }
        if(sc) bad = 1;                                                            // c keeps a remainder: the reference asserts on the degree of x
    }
    // ---- the sums around x at merge q and the vertex-weight chain (scallop.cc:2242-2378)
    double r1_q[N]; double vwt = vw0;
    ALD_UNROLL for(int q = 0; q < N; q++) {
        double sfan = 0;
        ALD_UNROLL for(int k = 0; k < N; k++) if(inv[k] >= q) sfan += fw[k];     // not merged yet, list order
        const bool sc = sc_q[q] != 0; const double ww = pw[q], wcur = wcur_q[q];
        double sc_side = 0; sc_side += sc ? rem_q[q] : wcur; if(sc) sc_side += ww;
        const double sum = A ? (sc_side + sfan) * 0.5 : (sfan + sc_side) * 0.5;
        const double wc0 = sc ? ww : wcur;
        const double r1 = A ? vwt * (wc0 + ww) * 0.5 / sum : vwt * (ww + wc0) * 0.5 / sum;
        vwt = vwt - r1; r1_q[q] = r1;
    }
    // ---- far's list (read only): the place of every merged edge -- behind the last entry whose key does not exceed its own (the old
    // entries carry older ids; new ones with the same key follow each other in merge order) -- and c's predecessor, in one walk
    const bool counted = uni(A ? (far == 0 && !HC.special_linked) : (far == HC.sinkp && !HC.special_linked));      // out(source) / in(sink) are only counted
    uint32_t key[N]; int pred[N], succ[N]; int pc = -1;
    ALD_UNROLL for(int k = 0; k < N; k++) { key[k] = A ? tkey((uint32_t)oth[k]) : (uint32_t)oth[k]; pred[k] = -1; succ[k] = -1; }
    if(!counted) {
        int placed[N]; ALD_UNROLL for(int k = 0; k < N; k++) placed[k] = 0;
        bool seen_c = false; int last = -1, guard = MAXE;
        int cur = A ? first_out(far) : first_in(far);
        while(uni(cur >= 0) && guard-- > 0) {
            const uint64_t w = *(const uint64_t*)&H.ed[cur].lk;
            const int nx = A ? lk_next((uint32_t)(w >> 48)) : lk_next((uint32_t)((w >> 32) & 0xFFFF));
            if(cur == cs) { pc = last; seen_c = true; }
            else {
                const uint32_t kc = A ? tkey((uint32_t)((w >> 16) & 0xFFFF)) : (uint32_t)(w & 0xFFFF);
                ALD_UNROLL for(int k = 0; k < N; k++) if(!placed[k] && kc > key[k]) { placed[k] = 1; pred[k] = last; succ[k] = cur; }
                last = cur;
            }
            bool all = seen_c; ALD_UNROLL for(int k = 0; k < N; k++) all = all && placed[k];
            if(uni(all)) break;
            cur = nx;
        }
        ALD_UNROLL for(int k = 0; k < N; k++) if(!placed[k]) pred[k] = last;       // behind everything
        if(!seen_c || guard <= 0) bad = 1;
    }
    // (wave-uniform values of c's record and of the vertex go to scalar registers now: the loads have had their time)
    const int meic = uni(meic_v), cntc = uni(cntc_v), stc = uni(stc_v), idc = uni(idc_v), lt = uni(lt_v), rt = uni(rt_v), ov = uni(ov_v);
    const uint32_t nsc = uni(nsc_v);
    if(!(cntc > 0)) bad = 1;
    // ---- the decision: nothing has been written so far
    {
        bool cnt_bad = false;
#ifdef ALD_EMU
        for(int k = 0; k < N; k++) if(!(C.ed[fe[k]].ecount > 0)) cnt_bad = true;
#else
        cnt_bad = lane < N && !(pf_cnt > 0);
#endif
#ifdef ALD_EMU_COUNT
        if(cnt_bad || bad) g_cnt_fix_declined++; else { g_cnt_star[N]++; g_cnt_fix[N]++; }
#endif
        if(wballot(cnt_bad) != 0 || uni(bad) != 0) return false;
    }
    // ---- lane j: fan edge j becomes merged edge inv[j]
    bool multi = false;
    for(int j = lane; j < N; j += ALD_WAVE) {
        const int q = pick<N>(inv, j), f = pick<N>(fe, j), o = pick<N>(oth, j);
        const double ww = pick<N>(fw, j), wcur = pick<N>(wcur_q, q), r1 = pick<N>(r1_q, q);
        const bool sc = pick<N>(sc_q, q) != 0; const int nid = pick<N>(nid_q, q);
        const double wc0 = sc ? ww : wcur;
        const double medc1 = sc ? medc * ww / wcur : medc;
#ifdef ALD_EMU
        const double pf_med = C.ed[f].med, pf_conf = C.ed[f].econf, pf_abd = C.ed[f].s0abd; const int pf_mei = C.ed[f].mei, pf_st = C.ed[f].estrand, pf_id = C.ed[f].s0id;
        const uint32_t pf_ns = C.ed[f].sp_len; const uint64_t pf_mask0 = C.ed[f].mask[0];
#endif
        if(nsc == 1 && pf_ns == 1) {          // one supporting sample on both sides: intersect_samples' inline case
            if(pf_id == idc) { const double xa = A ? abc : pf_abd, ya = A ? pf_abd : abc; const double mn = (ya < xa) ? ya : xa; C.ed[f].sp_off = 0; C.ed[f].ecount = 1; C.ed[f].eabd = 0.0 + mn; C.ed[f].s0abd = mn; }
            else { C.ed[f].sp_off = 0; C.ed[f].sp_len = 0; C.ed[f].ecount = 0; C.ed[f].eabd = 0; C.ed[f].s0id = 0; C.ed[f].s0abd = 0; }
        } else multi = true;                                                  // pool allocation: sequential, below
        C.ed[f].econf = A ? cc + pf_conf : pf_conf + cc;
        { const int sty = A ? pf_st : stc, stx = A ? stc : pf_st; C.ed[f].estrand = (uint8_t)(sty != 0 ? sty : stx); }
        for(int k = 0; k < NW; k++) { uint64_t mk = (NW <= 2 ? cmask_pf[NW <= 2 ? k : 0] : C.ed[cs].mask[k]) | (k == 0 ? pf_mask0 : C.ed[f].mask[k]); if(ov >= 0 && (ov >> 6) == k) mk |= (1ull << (ov & 63)); C.ed[f].mask[k] = mk; }
        const int mi = A ? rt - lt + meic + pf_mei : rt - lt + pf_mei + meic;
        C.ed[f].med = A ? mi * r1 + medc1 + pf_med : mi * r1 + pf_med + medc1; C.ed[f].mei = mi;
        H.eid[f] = (EID)nid; H.ed[f].w = A ? wc0 * 0.5 + ww * 0.5 : ww * 0.5 + wc0 * 0.5;
        if(A) H.ed[f].lk.es = (IDX)far; else H.ed[f].lk.et = (IDX)far;
        if(!dupf) { if(A) relink_in_lane(o, f, (uint32_t)far); else relink_out_lane(o, f, tkey((uint32_t)far)); }
        if(!counted) {
            // neighbours in (key) order among the new edges; a run of new edges behind the same old entry is chained
            const uint32_t kj = pick<N>(key, j); const int pj = pick<N>(pred, j), sj = pick<N>(succ, j);
            int rj = 0; ALD_UNROLL for(int k = 0; k < N; k++) rj += (key[k] < kj || (key[k] == kj && inv[k] < q)) ? 1 : 0;
            int prv_pred = -2, nxt_pred = -2, nxt_f = -1;                    // (-2: no such neighbour; a pred is >= -1)
            ALD_UNROLL for(int k = 0; k < N; k++) { int rk = 0; ALD_UNROLL for(int k2 = 0; k2 < N; k2++) rk += (key[k2] < key[k] || (key[k2] == key[k] && inv[k2] < inv[k])) ? 1 : 0;
                if(rk == rj - 1) prv_pred = pred[k]; if(rk == rj + 1) { nxt_pred = pred[k]; nxt_f = fe[k]; } }
            const bool first_of_gap = prv_pred != pj, last_of_gap = nxt_pred != pj;
            const IDX nx = last_of_gap ? (sj < 0 ? NIL : (IDX)sj) : (IDX)nxt_f;
            if(A) H.ed[f].lk.onx = nx; else H.ed[f].lk.inx = nx;
            if(first_of_gap) { if(pj < 0) { if(A) H.vx[far].out_head = (IDX)f; else H.vx[far].in_head = (IDX)f; } else { if(A) H.ed[pj].lk.onx = (IDX)f; else H.ed[pj].lk.inx = (IDX)f; } }
        }
    }
    const bool any_multi = wballot(multi) != 0;
    star_tail_sync();
    // ---- lane 0: c leaves far's list (unless a merged edge took its predecessor's link), the counters, what is inherently ordered
    if(lane == 0) {
        IDX *deg = A ? &H.vx[far].out_deg : &H.vx[far].in_deg;
        const int dg = (int)*deg - 1;
        if(!counted) {
            bool taken = false; ALD_UNROLL for(int k = 0; k < N; k++) if(pred[k] == pc) taken = true;
            if(!taken) { const IDX nx = c_next < 0 ? NIL : (IDX)c_next; if(pc < 0) { if(A) H.vx[far].out_head = nx; else H.vx[far].in_head = nx; } else { if(A) H.ed[pc].lk.onx = nx; else H.ed[pc].lk.inx = nx; } }
            if(dg <= 1) { HC.maybe_triv = 1; if(dg == 0) HC.maybe_broken = 1; }     // as unlink_in / unlink_out
        }
        *deg = (IDX)(dg + N); ev_degree(far, dg + 1, dg + N, A);
        C.vx[x].vw = vwt;
        if(dupf || any_multi || uni(HC.hl_n) != 0) {
            ALD_UNROLL for(int q = 0; q < N; q++) {
                const int f = uni(fq[q]);
                if(dupf) { if(A) relink_in(uni(oq[q]), f, (uint32_t)far); else relink_out(uni(oq[q]), f, tkey((uint32_t)far)); }
                if(any_multi) { const uint32_t nsf = uni(C.ed[f].sp_len); if(!(uni(nsc) == 1 && nsf == 1)) { if(!(A ? intersect_samples(cs, f, f) : intersect_samples(f, cs, f))) break; } }
                if(A) hs_replace2(cs, f, f); else hs_replace2(f, cs, f);
            }
        }
        HC.next_id = nid_q[N - 1] + 1;
        hs_remove(cs);
        // remove_edge(c); x is left without edges
        H.ed[cs].lk.es = NIL; H.hflag[cs] = 0;
        { int fh = uni(HC.free_head); H.ed[cs].lk.onx = fh < 0 ? NIL : (IDX)fh; HC.free_head = cs; HC.free_cnt = uni(HC.free_cnt) + 1; }
        clear_vertex(x);
    }
    wsync();
    PROF_ADD(PF_T_MERGE_ADD);
    return true;
}
template<bool A> ALD_INL bool star_fixed_any(int x, int n)
{
#if ALD_STARFIX_MAX >= 2
    if(n == 2) return uni(star_fixed<A, 2>(x));
#endif
#if ALD_STARFIX_MAX >= 3
    if(n == 3) return uni(star_fixed<A, 3>(x));
#endif
#if ALD_STARFIX_MAX >= 4
    if(n == 4) return uni(star_fixed<A, 4>(x));
#endif
    return false;
}
// wave-level entry (ALL lanes): fans of up to STAR_MAX edges go lane-parallel, anything else through the sequential form on lane 0
ALD_INL void decompose_trivial_vertex_wave(int x)
{
    x = uni(x);
    const int nin = uni(H.vx[x].in_deg), nout = uni(H.vx[x].out_deg);
#if ALD_STARFIX_MAX >= 2
    if(nin == 1 && nout >= 2 && nout <= ALD_STARFIX_MAX) { if(star_fixed_any<true>(x, nout)) return; }
    else if(nout == 1 && nin >= 2 && nin <= ALD_STARFIX_MAX) { if(star_fixed_any<false>(x, nin)) return; }
#endif
#ifdef ALD_STAR_SEQ_MAX
    // experiment: the smallest fans through the sequential form on lane 0 (no hand-overs at all)
    if((nin == 1 && nout >= 1 && nout <= ALD_STAR_SEQ_MAX) || (nout == 1 && nin >= 1 && nin <= ALD_STAR_SEQ_MAX)) { if(lane_id() == 0) decompose_trivial_vertex(x); wsync(); return; }
#endif
#if !defined(ALD_EMU) && !defined(ALD_STAR_SCRATCH_FORM)
    if(nin == 1 && nout >= 1 && nout <= STAR_MAX) star_reg<true>(x);         // the fan in registers (the emulation and -DALD_STAR_SCRATCH_FORM: the same
    else if(nout == 1 && nin >= 1 && nin <= STAR_MAX) star_reg<false>(x);    // arithmetic through arrays in LDS, star_wave_body)
#else
    if(nin == 1 && nout >= 1 && nout <= STAR_MAX) star_wave_in(x);
    else if(nout == 1 && nin >= 1 && nin <= STAR_MAX) star_wave_out(x);
#endif
    else { if(lane_id() == 0) decompose_trivial_vertex(x); wsync(); }
}

// scallop::decompose_vertex_extend (scallop.cc:1675-1986); pe2w = n sorted pairs in the work area
// SMALL: pe2w and the work arrays are in the LDS scratch; otherwise BOTH are in the slab (decompose_vertex_extend_any moves pe2w there
// first when the router left it in LDS).  Either way every pointer of an instance has ONE address space the compiler can see: ds_* or
// global_* accesses, never FLAT ones.  (Round 4: with the place chosen at run time this routine and save / restore_pairs were the only
// code of the engine that compiled to flat_load / flat_store -- 69 of them.  A FLAT access that ends in LDS travels through the vector
// memory path and is not ordered against the wave's DS instructions to the same LDS words -- the scratch every other routine uses.)
template<bool SMALL> ALD_INL void decompose_vertex_extend_body(int root, int n)
{
    COLD;
    const Pairs P = pairs_at(SMALL, false);
    int32_t *a = P.a, *b = P.b; double *w = P.w;
    const int deg = (int)uni(H.vx[root].in_deg) + (int)uni(H.vx[root].out_deg);
    // the visiting order of the nested decompositions (jump_ratio > 1 only) must survive them: it always lives in the slab
    const Arena AR = arena_at(SMALL);
    if(ALD_UNLIKELY(4 * deg > AR.cap_i || deg > AR.cap_d || deg > C.w_cap / 16)) { fail(ALD_ST_CAPACITY); return; }
    int nloc = 0; int32_t *loc_e = AR.i;
    // (Tried: asking for the cold records of the root's edges and of the vertices at their far ends right here, with loads nobody waits
    // for, so that the ~12 serial round trips further down overlap -- 42.2 against 41.95 ms, profiles/r03/ze_kernel_ab_extend_touch.txt:
    // the routine is bound by its scalar list work, not by those round trips.)
    for(int e = u_first_in(root); e >= 0; e = u_next_in(e)) { loc_e[nloc++] = e; }
    int nin = nloc;
    for(int e = u_first_out(root); e >= 0; e = u_next_out(e)) { loc_e[nloc++] = e; }
    int32_t *mdeg = AR.i + deg;                   // [nloc]
    int32_t *evx = AR.i + 2 * deg;                // [nloc] new vertex of the edge (ev1 / ev2), or -1
    double *mweight = AR.d;                       // [nloc]
    for(int i = 0; i < nloc; i++) { mdeg[i] = 0; evx[i] = -1; mweight[i] = 0; }
    const double mw = HC.p_min_w;
    double total_weight = 0;
    for(int i = 0; i < n; i++) {
        if(ALD_UNLIKELY(!(w[i] >= mw - kSMIN))) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return; }
        int u1 = PLOC(a[i]), u2 = PLOC(b[i]);
        total_weight += w[i];
        if(mdeg[u1] == 0) mweight[u1] = w[i]; else mweight[u1] += w[i];
        if(mdeg[u2] == 0) mweight[u2] = w[i]; else mweight[u2] += w[i];
        mdeg[u1]++; mdeg[u2]++;
    }
    int rlen = uni(C.vx[root].rpos) - uni(C.vx[root].lpos);
    double vertex_weight = uni(C.vx[root].vw) * rlen;
    for(int i = 0; i < nloc; i++) mweight[i] = mweight[i] / total_weight * vertex_weight;
    // new vertices (scallop.cc:1753-1806) are appended at the end of the physical index space; the reference gives them the
    // indices m.. and moves the sink behind them -- same relative order, no edge has to move here
    int m = HC.nv, nn = m;
    for(int i = 0; i < nloc; i++) { if(ALD_UNLIKELY(mdeg[i] == 0)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; } if(mdeg[i] >= 2) evx[i] = nn++; }
    int newedges = 0;
    for(int i = 0; i < n; i++) { int u1 = PLOC(a[i]), u2 = PLOC(b[i]); if(mdeg[u1] == 1 && mdeg[u2] == 1) evx[u1] = nn++; else if(mdeg[u1] >= 2 && mdeg[u2] >= 2) newedges++; }
    if(ALD_UNLIKELY(nn > MAXV || free_slots() < newedges)) { fail(ALD_ST_CAPACITY); return; }
    HC.maybe_broken = 1; HC.maybe_triv = 1; ev_mark_all();
    for(int i = m; i < nn; i++) { H.vx[i].in_head = NIL; H.vx[i].out_head = NIL; H.vx[i].in_deg = 0; H.vx[i].out_deg = 0; H.nz[i] = 1; C.vx[i].vw = 0; C.vx[i].lpos = 0; C.vx[i].rpos = 0; C.vx[i].vtype = -1; C.vx[i].v2v = -1; }
    HC.nv = nn;
    for(int i = 0; i < nin; i++) {               // ev1: detach in-edges onto their new vertex
        int k = evx[i]; if(k < 0) continue; int e = loc_e[i];
        int p = uni(C.vx[uni(H.ed[e].lk.es)].rpos);
        move_edge(e, uni(H.ed[e].lk.es), k); C.vx[k].lpos = p; C.vx[k].rpos = p; C.vx[k].vtype = -1; C.vx[k].vw = 0; C.vx[k].v2v = -2;
    }
    for(int i = nin; i < nloc; i++) {            // ev2
        int k = evx[i]; if(k < 0) continue; int e = loc_e[i];
        int p = uni(C.vx[uni(H.ed[e].lk.et)].lpos);
        move_edge(e, k, uni(H.ed[e].lk.et)); C.vx[k].lpos = p; C.vx[k].rpos = p; C.vx[k].vtype = -1; C.vx[k].vw = 0; C.vx[k].v2v = -2;
    }
    int rv = uni(C.vx[root].v2v);
    for(int i = 0; i < n; i++) {
        int e1 = PSLOT(a[i]), e2 = PSLOT(b[i]); int u1 = PLOC(a[i]), u2 = PLOC(b[i]); double ww = w[i];
        if(mdeg[u1] == 1 && mdeg[u2] >= 2) {
            borrow_edge_strand(C, e1, e2);
            move_edge(e1, uni(H.ed[e1].lk.es), evx[u2]);
            if(rv >= 0) C.ed[e1].mask[(rv >> 6)] |= (1ull << (rv & 63));
            C.ed[e1].med += mweight[u1]; C.ed[e1].mei += rlen;
        } else if(mdeg[u2] == 1) {
            if(ALD_UNLIKELY(evx[u1] < 0)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
            borrow_edge_strand(C, e2, e1);
            move_edge(e2, evx[u1], uni(H.ed[e2].lk.et));
            if(rv >= 0) C.ed[e2].mask[(rv >> 6)] |= (1ull << (rv & 63));
            C.ed[e2].med += mweight[u2]; C.ed[e2].mei += rlen;
        } else {
            int z = add_edge(evx[u1], evx[u2]);
            if(z < 0) return;
            H.ed[z].w = ww;
            if(ALD_UNLIKELY(!(C.ed[e1].ecount > 0 && uni(C.ed[e2].ecount) > 0))) { fail(ALD_ST_INVARIANT + ALD_INV_COUNT); return; }
            if(!intersect_samples(e1, e2, z)) return;
            if(ALD_UNLIKELY(C.ed[z].ecount <= 0)) { fail(ALD_ST_INVARIANT + ALD_INV_COUNT); return; }
            C.ed[z].econf = 0; C.ed[z].estrand = 0;
            for(int k = 0; k < NW; k++) C.ed[z].mask[k] = 0;
            if(rv >= 0) C.ed[z].mask[(rv >> 6)] |= (1ull << (rv & 63));
            C.ed[z].med = ww / total_weight * vertex_weight; C.ed[z].mei = rlen;
            borrow_edge_strand(C, z, e1); borrow_edge_strand(C, z, e2);
            hs_insert_between(e1, e2, z);
            if(HC.status) return;
        }
    }
    if(ALD_UNLIKELY(H.vx[root].in_deg != 0 || uni(H.vx[root].out_deg) != 0)) { fail(ALD_ST_INVARIANT + ALD_INV_DEGREE); return; }
    H.nz[root] = 0;
    // scallop.cc:1976-1985 resolve_single_trivial_vertex(k, jump_ratio) on the new vertices: a no-op unless jump_ratio > 1
    double jump = HC.p_ratio[7];
    if(jump > 1.0) {
        // the reference walks ev1 then ev2, each a std::map keyed by edge id.  The nested decompositions reuse the work area, so
        // the visiting order is parked first in [3/8, 1/2) of wi, which no routine touches.
        int32_t *order = (int32_t*)(C.wi + 3 * (C.w_cap / 8)); int no = 0;     // slab region [3/8, 1/2) of wi: untouched by every routine
        for(int part = 0; part < 2; part++) {
            int lo = part == 0 ? 0 : nin, hi = part == 0 ? nin : nloc; int first = no;
            for(int i = lo; i < hi; i++) if(evx[i] >= 0) {
                int k = evx[i]; uint32_t id = uni(H.eid[loc_e[i]]); int j = no - 1;
                while(j >= first && uni(H.eid[mdeg[j]]) > id) { order[j + 1] = order[j]; mdeg[j + 1] = mdeg[j]; j--; }      // mdeg is free now: reuse it for the sort keys' edges
                order[j + 1] = k; mdeg[j + 1] = loc_e[i]; no++;
            }
        }
        for(int q = 0; q < no; q++) { resolve_single_trivial_vertex(order[q], jump); if(HC.status) return; }
    }
}

ALD_FN void decompose_vertex_extend_small(int root, int n) { decompose_vertex_extend_body<true>(uni(root), uni(n)); }
template<bool LDS> ALD_INL void copy_pairs(bool from_parked, bool to_parked, int n)
{
    const Pairs S = pairs_at(LDS, from_parked), D = pairs_at(LDS, to_parked);
    for(int i = 0; i < n; i++) { D.a[i] = S.a[i]; D.b[i] = S.b[i]; D.w[i] = S.w[i]; }
}
ALD_FN void decompose_vertex_extend_any(int root, int n)
{
    root = uni(root); n = uni(n);
    if(uni(HC.pw_lds) != 0) {                        // pe2w from the LDS scratch to the slab's area: the general form addresses the slab only
        const Pairs S = pairs_at(true, false), D = pairs_at(false, false);
        if(ALD_UNLIKELY(n > D.cap)) { fail(ALD_ST_CAPACITY); return; }
        for(int i = 0; i < n; i++) { D.a[i] = S.a[i]; D.b[i] = S.b[i]; D.w[i] = S.w[i]; }
        HC.pw_lds = 0;
    }
    decompose_vertex_extend_body<false>(root, n);
}
ALD_INL void decompose_vertex_extend(int root, int n)
{
    root = uni(root); n = uni(n);
    const int deg = (int)uni(H.vx[root].in_deg) + (int)uni(H.vx[root].out_deg);
    if(uni(HC.pw_lds) != 0 && 4 * deg <= ARENA_I && deg <= ARENA_D && !(uni(HC.p_ratio[7]) > 1.0)) decompose_vertex_extend_small(root, n);
    else decompose_vertex_extend_any(root, n);
}

// ---------------------------------------------------------------- per-vertex rule evaluation (one vertex per lane)
// scallop::classify_trivial_vertex (scallop.cc:2169-2196); -2 = needs a dominate query on need_e
ALD_INL int classify_trivial_fastpath(int x, bool fast)
{
    int d1 = H.vx[x].in_deg, d2 = H.vx[x].out_deg;
    if(d1 != 1 && d2 != 1) return -1;
    int e1 = first_in(x), e2 = first_out(x);
    if(d1 == 1) { int s = H.ed[e1].lk.es; if(H.vx[s].out_deg == 1) return 1; if(fast) { if(!(H.hflag[e1] & HF_OCC)) return 1; return -2; } }
    if(d2 == 1) { int t = H.ed[e2].lk.et; if(H.vx[t].in_deg == 1) return 1; if(fast) { if(!(H.hflag[e2] & HF_OCC)) return 1; return -2; } }
    return 2;
}
ALD_FN int classify_trivial_vertex(int x, bool fast)     // scalar version with the dominate queries
{
    x = uni(x); fast = uni(fast);
    int d1 = uni(H.vx[x].in_deg), d2 = uni(H.vx[x].out_deg);
    if(d1 != 1 && d2 != 1) return -1;
    int e1 = u_first_in(x), e2 = u_first_out(x);
    if(d1 == 1) { int s = uni(H.ed[e1].lk.es); if(H.vx[s].out_deg == 1) return 1; if(fast && hs_dominate(e1, 1)) return 1; }
    if(d2 == 1) { int t = uni(H.ed[e2].lk.et); if(H.vx[t].in_deg == 1) return 1; if(fast && hs_dominate(e2, 2)) return 1; }
    return 2;
}
ALD_INL double compute_balance_ratio(int v, bool &ok)    // scallop.cc:2578-2602
{
#ifndef ALD_HOT_IN_SLAB
    double w1 = in_weights(v), w2 = out_weights(v);
#else
    double w1 = 0, w2 = 0;
    {   // in_weights(v) and out_weights(v), their list steps taken together (see eval_smallest)
        int a = first_in(v), b = first_out(v);
        while((a >= 0) | (b >= 0)) {
            const int ea = a >= 0 ? a : 0, eb = b >= 0 ? b : 0;
            const double wa = H.ed[ea].w, wb = H.ed[eb].w; const IDX na = H.ed[ea].lk.inx, nb = H.ed[eb].lk.onx;
            if(a >= 0) { w1 += wa; a = na == NIL ? -1 : (int)na; }
            if(b >= 0) { w2 += wb; b = nb == NIL ? -1 : (int)nb; }
        }
    }
#endif
    ok = (w1 >= kSMIN) && (w2 >= kSMIN);
    if(w1 >= w2) return w1 / w2; else return w2 / w1;
}
// scallop::resolve_single_trivial_vertex (scallop.cc:1236-1254), scalar
ALD_FN bool resolve_single_trivial_vertex(int i, double jump_ratio)
{
    i = uni(i); jump_ratio = uni(jump_ratio);
    if(H.vx[i].in_deg == 0 || H.vx[i].out_deg == 0) return false;
    if(H.vx[i].in_deg >= 2 && H.vx[i].out_deg >= 2) return false;
    if(mixed_strand_vertex(i)) return false;
    if(classify_trivial_vertex(i, false) != 1) return false;
    bool ok; double r = compute_balance_ratio(i, ok);
    if(ALD_UNLIKELY(!ok)) { fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); return false; }
    if(r >= jump_ratio) return false;
    trace(OP_TRIVIAL_FAST, vlog(i), 0, r);
    decompose_trivial_vertex(i);
    return true;
}
// scallop::compute_smallest_edge + the guards of resolve_smallest_edges (scallop.cc:858-896, 2967-3030)
ALD_INL int eval_smallest(int i, double &r)
{
    int hin, hout;
    {   // the vertex record in one round of loads (the short-circuit form waits for each field before it asks for the next)
        const Hot::VertexHot vr = H.vx[i];       // heads and degrees: one 8-byte LDS read
        const int nzv = H.nz[i] & NZ_MEMBER, d1 = vr.in_deg, d2 = vr.out_deg;
        if((nzv == 0) | (d1 <= 1) | (d2 <= 1)) return -1;
        hin = slot_or_neg(vr.in_head); hout = slot_or_neg(vr.out_head);
    }
    int e1 = -1, e2 = -1; double sum1 = 0, sum2 = 0, min1 = DBL_MAX, min2 = DBL_MAX;
#ifndef ALD_HOT_IN_SLAB
    for(int e = hin; e >= 0; e = next_in(e)) { double w = H.ed[e].w; sum1 += w; if(w > min1) continue; min1 = w; e1 = e; }
    for(int e = hout; e >= 0; e = next_out(e)) { double w = H.ed[e].w; sum2 += w; if(w > min2) continue; min2 = w; e2 = e; }
#else
    {   // hot state in the slab (the catch-all class and the twins): every list step is a round trip to L2 -- the two walks advance
        // together, their steps being independent of each other (in LDS the merged loop costs more than the latency it hides)
        int a = hin, b = hout;
        while((a >= 0) | (b >= 0)) {
            const int ea = a >= 0 ? a : 0, eb = b >= 0 ? b : 0;
            const double wa = H.ed[ea].w, wb = H.ed[eb].w; const IDX na = H.ed[ea].lk.inx, nb = H.ed[eb].lk.onx;
            if(a >= 0) { sum1 += wa; if(!(wa > min1)) { min1 = wa; e1 = a; } a = na == NIL ? -1 : (int)na; }
            if(b >= 0) { sum2 += wb; if(!(wb > min2)) { min2 = wb; e2 = b; } b = nb == NIL ? -1 : (int)nb; }
        }
    }
#endif
    if(e1 < 0 || e2 < 0) return -1;
    if(!(sum1 >= kSMIN) || !(sum2 >= kSMIN)) return -3;          // reference assert(sum1 >= SMIN)
    double r1 = min1 / sum1, r2 = min2 / sum2;
    int e; if(r1 < r2) { r = r1; e = e1; } else { r = r2; e = e2; }
    int s = H.ed[e].lk.es, t = H.ed[e].lk.et;
    uint8_t f = H.hflag[e];
    { const int ods = H.vx[s].out_deg, idt = H.vx[t].in_deg; if((ods <= 1) | (idt <= 1)) return -1; }
    if((f & HF_REXT) && (f & HF_LEXT)) return -1;
    if(t == i && (f & HF_REXT)) return -1;
    if(s == i && (f & HF_LEXT)) return -1;
    if(HC.any_strand) {
        COLD;
        int z = C.ed[e].estrand;
        if(z >= 1) { int vs[6]; strand_degree(i, vs); if(s == i && vs[0] + vs[z] <= 1) return -1; if(t == i && vs[3] + vs[z + 3] <= 1) return -1; }
    }
    return e;
}

// ---------------------------------------------------------------- wave-level sweeps
// scallop::resolve_broken_vertex (scallop.cc:190-236)
ALD_INL bool resolve_broken_vertex()
{
    if(!uni(HC.maybe_broken)) return false;
    const int lane = lane_id();
    int vend = HC.nv; int x = -1;
    for(int base = 0; base < vend && x < 0; base += ALD_WAVE) {
        int i = base + lane;
        const bool inr = (i >= 1) & (i < vend) & (i != HC.sinkp); const int ii = inr ? i : 0;
        const Hot::VertexHot vr = H.vx[ii];
        const int nzv = H.nz[ii] & NZ_MEMBER, d1 = vr.in_deg, d2 = vr.out_deg;
        bool p = inr & (nzv != 0) & !((d1 >= 1) & (d2 >= 1));
        uint64_t m = wballot(p);
        if(m) x = base + ffs64(m);
    }
    if(x < 0) { wsync(); if(lane == 0) HC.maybe_broken = 0; wsync(); return false; }
    if(lane == 0) {
        if(H.vx[x].in_deg + H.vx[x].out_deg == 0) fail(ALD_ST_INVARIANT + ALD_INV_OTHER);      // assert(ve.size() >= 1)
        else {
            HC.maybe_triv = 1;
            trace(OP_BROKEN, vlog(x), H.vx[x].in_deg + H.vx[x].out_deg, 0);
            int guard = MAXE;
            while(first_in(x) >= 0 && guard-- > 0) { int e = first_in(x); kill_edge(e); hs_remove(e); }
            while(first_out(x) >= 0 && guard-- > 0) { int e = first_out(x); kill_edge(e); hs_remove(e); }
            H.nz[x] = 0; ev_mark(x);
        }
    }
    wsync();
    return true;
}

// generic trivial-vertex sweep: mode 0 = resolve_trivial_vertex_fast (scallop.cc:1256-1270: fast=false, type 1, r < jump)
//                               mode 1 = resolve_trivial_vertex(type, fast=true, jump)  (scallop.cc:1180-1234)
// The wave-parallel evaluation is a LEAF (scan_trivial: no calls, so no callee-saved registers go to scratch on the ~500 scans
// a graph takes); what it carries from one scan of a sweep to the next -- the running (ratio, root) of the reference's
// sequential loop -- lives in the LDS context, and the thin driver below only keeps `start` across the decomposition call.
// the r-th set bit of m (r < popcount(m)): six halving steps on the popcount of the low part
ALD_INL int nth_set_bit(uint64_t m, int r)
{
    int pos = 0;
    ALD_UNROLL for(int w = 32; w >= 1; w >>= 1) {
        const uint64_t low = m & ((1ull << w) - 1ull); const int c = (int)__builtin_popcountll(low);
        if(r >= c) { r -= c; m >>= w; pos += w; } else m = low;
    }
    return pos;
}
// classify_trivial_fastpath (fast = true: resolve_trivial_vertex's mode) + compute_balance_ratio for ONE vertex, every lane its own: -> class
// (-9: not a trivial vertex at all, 1 / 2, -2: needs a dominate query), the ratio (classes 1, 2, -2) and whether its weights are in order.
// Every load that does not depend on another is issued together: three LDS round trips (vertex record / first edges / their far ends)
// instead of one per condition of the short-circuit form.
ALD_INL int classify_with_ratio(int i, bool inr, double &r, bool &bad)
{
    int cls = -9; r = 0; bad = false;
    const int ii = inr ? i : 0;
    const Hot::VertexHot vr = H.vx[ii];       // heads and degrees: one 8-byte LDS read
    const int nzv = H.nz[ii] & NZ_MEMBER, d1 = vr.in_deg, d2 = vr.out_deg; const IDX h1 = vr.in_head, h2 = vr.out_head;
    bool elig = inr & (nzv != 0) & (d1 >= 1) & (d2 >= 1) & !((d1 >= 2) & (d2 >= 2));
    if(HC.any_strand) elig = elig && !mixed_strand_vertex(i);
    const int e1 = (elig & (h1 != NIL)) ? (int)h1 : 0, e2 = (elig & (h2 != NIL)) ? (int)h2 : 0;
    const IDX sv = H.ed[e1].lk.es, tv = H.ed[e2].lk.et; const uint8_t f1 = H.hflag[e1], f2 = H.hflag[e2];
    const int s_ = (elig & (sv != NIL)) ? (int)sv : 0, t_ = (elig & (tv != NIL)) ? (int)tv : 0;
    const int ods = H.vx[s_].out_deg, idt = H.vx[t_].in_deg;
    if(elig) {
        if(d1 == 1 && ods == 1) cls = 1;
        else if(d1 == 1) cls = (f1 & HF_OCC) ? -2 : 1;
        else if(d2 == 1 && idt == 1) cls = 1;
        else if(d2 == 1) cls = (f2 & HF_OCC) ? -2 : 1;
        else cls = 2;
        bool ok; r = compute_balance_ratio(i, ok); bad = !ok;
    }
    return cls;
}
enum { SC_NONE = 0, SC_HIT = 1, SC_STOP = 2, SC_NEED = 3, SC_BAD = 4 };
enum { EV_NC_ = MAXV / ALD_WAVE };
// A call.  Inlined into its one call site (-DALD_SCAN_INLINE) class 1 gains 0.6 % (38.2 against 38.45 ms) for 12 more spilled SGPRs in the root, class 0
// loses 3 %, the mixed batch moves neither way (profiles/r04/zg_kernel_ab_scan_inlined.txt): not adopted
#ifdef ALD_SCAN_INLINE
ALD_INL int scan_trivial(int start, int mode, int type, double jump_ratio)
#else
ALD_FN int scan_trivial(int start, int mode, int type, double jump_ratio)
#endif
{
    start = uni(start); mode = uni(mode); type = uni(type); jump_ratio = uni(jump_ratio);
    const int lane = lane_id();
    const bool fast = (mode == 1);
    const double now_thr = (mode == 1) ? 1.02 : jump_ratio;
    const int vend = HC.sw_vend;                // snapshot of nonzeroset: vertices created later are not visited (the sink is never in it: nz == 0)
    double best_r = HC.sw_best_r; int best_v = HC.sw_best_v;
    const int dom_base = HC.sw_dom_base;        // chunk whose dominate queries the driver has answered (scr_i[lane]), or -1
    int code = SC_NONE, hit = -1; double hit_r = 0;
    double frr = DBL_MAX; int fvv = -1;          // this lane's best candidate over the chunks scanned (later vertex wins ties)
#if ALD_KEEP_TRIV
    // The slab-resident classes keep (class, balance ratio) of every vertex in the wave's slab between the scans, as the smallest-edge sweep keeps its
    // evaluations: first the vertices marked since the last scan -- in dense lanes --, or all of them; then every chunk of the scan is one
    // coalesced read.  (resolve_trivial_vertex's mode only: the classification of resolve_trivial_vertex_fast, fast = false, is another one.)
    ALD_GLOBAL double *kr = (ALD_GLOBAL double*)(HC.cold + CL::o_tvr); ALD_GLOBAL int32_t *kc = (ALD_GLOBAL int32_t*)(HC.cold + CL::o_tvc);
    const bool kept = (mode == 1);
    if(kept) {
        const int nv_now = uni(HC.nv);
  #if defined(ALD_EMU) && defined(ALD_EMU_CHECK)
        if(HC.tv_all == 0) for(int v = 1; v < nv_now; v++) {
            if((HC.tv_dirty[v >> 5] >> (v & 31)) & 1u) continue;
            double r2; bool b2; const int c2 = classify_with_ratio(v, true, r2, b2); const int kc_ = kc[v]; const double kr_ = kr[v];
            if((c2 & 0xFF) != (kc_ & 0xFF) || (int)b2 != ((kc_ >> 8) & 1) || (c2 != -9 && memcmp(&r2, &kr_, 8) != 0)) { fprintf(stderr, "[check] graph %d: stale trivial-scan record of vertex %d: kept (%d, %.17g), fresh (%d, %.17g)\n", HC.g, v, kc_, kr_, c2, r2); abort(); }
        }
  #endif
        if(uni(HC.tv_all) != 0) {
            for(int b0 = 0; b0 < nv_now; b0 += ALD_WAVE) { const int v = b0 + lane; const bool in = v >= 1 && v < nv_now; double r2; bool b2; const int c2 = classify_with_ratio(v, in, r2, b2); if(v < nv_now) { kr[v] = r2; kc[v] = (c2 & 0xFF) | (b2 ? 0x100 : 0); } }
        } else {
  #ifdef ALD_EMU
            for(int v = 1; v < nv_now; v++) if((HC.tv_dirty[v >> 5] >> (v & 31)) & 1u) { double r2; bool b2; const int c2 = classify_with_ratio(v, true, r2, b2); kr[v] = r2; kc[v] = (c2 & 0xFF) | (b2 ? 0x100 : 0); }
  #else
            int total = 0;
            for(int w = 0; w < (nv_now + 31) / 32; w++) total += (int)__builtin_popcount(uni(HC.tv_dirty[w]));
            for(int b0 = 0; b0 < total; b0 += ALD_WAVE) {
                const int k = b0 + lane; int run = 0, vtx = -1;
                for(int w = 0; w < (nv_now + 31) / 32; w++) {
                    const uint32_t m = uni(HC.tv_dirty[w]); const int pc = (int)__builtin_popcount(m);
                    if(k >= run && k < run + pc) vtx = 32 * w + nth_set_bit((uint64_t)m, k - run);
                    run += pc;
                }
                const bool in = vtx >= 1 && vtx < nv_now;
                double r2; bool b2; const int c2 = classify_with_ratio(vtx, in, r2, b2);
                if(in) { kr[vtx] = r2; kc[vtx] = (c2 & 0xFF) | (b2 ? 0x100 : 0); }
            }
  #endif
        }
        wsync_mem();                              // (as in sweep_smallest: the records travel through the slab)
        tv_clear_marks(nv_now);
        wsync();
    }
#else
    const bool kept = false;
#endif
#if ALD_KEEP_TRIV
    // every chunk the scan may visit is asked for NOW, all reads in flight together (the scan leaves at its first hit, but a chunk asked for
    // when the loop reaches it would cost a round trip to L2 each, one after the other); classes of more than sixteen chunks read as they go
    constexpr int KNC = (EV_NC_ <= 16) ? EV_NC_ : 1;
    double pr_[KNC]; int pc_[KNC];
    if(kept && EV_NC_ <= 16) {
        ALD_UNROLL for(int c = 0; c < KNC; c++) {
            const int i = c * ALD_WAVE + lane; const bool in = (i >= start) & (i < vend);
            pr_[c] = in ? kr[i] : 0.0; pc_[c] = in ? kc[i] : (-9 & 0xFF);
        }
    }
#endif
    for(int base = (start / ALD_WAVE) * ALD_WAVE; base < vend; base += ALD_WAVE) {
        int i = base + lane;
        int cls = -9; double r = 0; bool bad = false; bool kbad = false;
#if ALD_KEEP_TRIV
        if(kept) {
            const bool inr = (i >= start) & (i < vend);
            int kc_ = (-9 & 0xFF); double kr_ = 0;
            if(EV_NC_ <= 16) { const int cq = base / ALD_WAVE; ALD_UNROLL for(int c = 0; c < KNC; c++) if(c == cq) { kc_ = pc_[c]; kr_ = pr_[c]; } }
            else if(inr) { kc_ = kc[i]; kr_ = kr[i]; }
            if(inr) { r = kr_; cls = (int)(int8_t)(kc_ & 0xFF); kbad = (kc_ & 0x100) != 0; }
        } else
#endif
        {
            // classify_trivial_fastpath with every load that does not depend on another issued together: three LDS round trips
            // (vertex record / first edges / their far ends) instead of one per condition of the short-circuit form
            const bool inr = (i >= start) & (i < vend); const int ii = inr ? i : 0;
            const Hot::VertexHot vr = H.vx[ii];       // heads and degrees: one 8-byte LDS read
            const int nzv = H.nz[ii] & NZ_MEMBER, d1 = vr.in_deg, d2 = vr.out_deg; const IDX h1 = vr.in_head, h2 = vr.out_head;
            bool elig = inr & (nzv != 0) & (d1 >= 1) & (d2 >= 1) & !((d1 >= 2) & (d2 >= 2));
            if(HC.any_strand) elig = elig && !mixed_strand_vertex(i);
            const int e1 = (elig & (h1 != NIL)) ? (int)h1 : 0, e2 = (elig & (h2 != NIL)) ? (int)h2 : 0;
            const IDX sv = H.ed[e1].lk.es, tv = H.ed[e2].lk.et; const uint8_t f1 = H.hflag[e1], f2 = H.hflag[e2];
            const int s_ = (elig & (sv != NIL)) ? (int)sv : 0, t_ = (elig & (tv != NIL)) ? (int)tv : 0;
            const int ods = H.vx[s_].out_deg, idt = H.vx[t_].in_deg;
            if(elig) {
                if(d1 == 1 && ods == 1) cls = 1;
                else if(d1 == 1 && fast) cls = (f1 & HF_OCC) ? -2 : 1;
                else if(d2 == 1 && idt == 1) cls = 1;
                else if(d2 == 1 && fast) cls = (f2 & HF_OCC) ? -2 : 1;
                else cls = 2;
            }
        }
        // lanes that need a dominate query (rare: only edges on phasing paths) are answered by the driver, one at a time on lane 0
        if(cls == -2 && base == dom_base) cls = HC.scr_i[lane];
        uint64_t need = wballot(cls == -2);
        if(need) { if(lane == 0) { HC.sw_dom_base = base; HC.sw_need_lo = (uint32_t)need; HC.sw_need_hi = (uint32_t)(need >> 32); } code = SC_NEED; break; }
        bool cand = (cls == type);
        if(cand) { if(kept) bad = kbad; else { bool ok; r = compute_balance_ratio(i, ok); if(!ok) bad = true; } }
        if(wballot(bad)) { code = SC_BAD; break; }
        uint64_t now = wballot(cand && r < now_thr);
        // scallop.cc:1222 `if(ratio < jump_ratio) break;`: the first candidate with 1.02 <= r < jump_ratio becomes the root and ends the sweep
        uint64_t stp = (mode == 1) ? wballot(cand && !(r < now_thr) && r < jump_ratio) : 0ull;
        if(stp && (!now || ffs64(stp) < ffs64(now))) {
            int l = ffs64(stp); best_v = base + l; best_r = wread(r, l); code = SC_STOP; break;
        }
        // sequential semantics: candidates before the first "now" vertex update (ratio, root); ties go to the LATER vertex
        int lim = now ? ffs64(now) : ALD_WAVE;
        if(mode == 1 && cand && lane < lim && (fvv < 0 || !(frr < r))) { frr = r; fvv = i; }     // folded per lane; ONE reduction after the loop
        if(now) { hit = base + ffs64(now); hit_r = wread(r, ffs64(now)); code = SC_HIT; break; }
    }
    if(mode == 1 && code != SC_STOP && code != SC_BAD) {
        double rr = frr; int vv = fvv;
        wave_argmin(rr, vv);                                             // (all 64 lanes are here: the loop above leaves through wave-uniform breaks only)
        if(vv >= 0 && !(best_r < rr)) { best_r = rr; best_v = vv; }      // if(ratio < r) continue;
    }
    if(lane == 0) { HC.sw_best_r = best_r; HC.sw_best_v = best_v; HC.sw_hit = hit; HC.sw_hit_r = hit_r; }
    wsync();
    return code;
}
// The count of rule firings (HotCtx::n_iters, reported per graph) is kept in a register while a sweep runs and added once when it ends, and
// the arguments of an op-trace event are only formed when a trace is being taken: trace() itself costs a read-modify-write of the counter
// and a read of the trace capacity in LDS per firing, and its arguments (creation id, reference index of the vertex) four more reads --
// seven LDS instructions per firing, each one ~4.5 cycles of the CU's LDS pipeline, for nothing in an untraced run.
ALD_INL bool sweep_trivial_body(int mode, int type, double jump_ratio, int &fired, const bool tr);
ALD_INL bool sweep_trivial(int mode, int type, double jump_ratio)
{
    int fired = 0; const bool tr = tracing_u();
    const bool rv = sweep_trivial_body(mode, type, jump_ratio, fired, tr);
    if(fired && lane_id() == 0) HC.n_iters += fired;
    return rv;
}
ALD_INL bool sweep_trivial_body(int mode, int type, double jump_ratio, int &fired, const bool tr)
{
    mode = uni(mode); type = uni(type); jump_ratio = uni(jump_ratio);
    const int lane = lane_id();
    PROF_DECL;
    // R3 (mode 1, type 1) acts whenever a type-1 vertex exists.  After a sweep that found none, one can only appear when a degree
    // drops to <= 1, the phasing flags change or vertices are created -- all of which raise maybe_triv; until then the scan is
    // skipped.  (Stranded graphs also gain candidates when a removal un-mixes a vertex: they always scan.)
    const int hsd_q = HC.hs_dirty, strand_q = HC.any_strand, triv_q = HC.maybe_triv, nv_q = HC.nv;      // one round of LDS reads for the tests below
    const bool skippable = (mode == 1 && type == 1 && !uni(strand_q));
    if(skippable && !uni(hsd_q) && !uni(triv_q)) return false;                // (a pending flag refresh raises maybe_triv itself: not skipped then)
    if(lane == 0) { if(uni(hsd_q)) hs_refresh_flags(); HC.sw_vend = nv_q; HC.sw_best_r = DBL_MAX; HC.sw_best_v = -1; HC.sw_dom_base = -1; }
    wsync();
    if(skippable && !uni(HC.maybe_triv)) return false;
    bool flag = false;
    int start = 1;
    // ONE place decomposes -- the vertex a scan hits (the sweep then goes on behind it) or, when the sweep ends without a hit, its best
    // candidate: the wave-wide decomposition is inlined here, so it must not be instantiated twice
    for(;;) {
        int target = -1; bool last = false;
        while(start < uni(HC.sw_vend)) {
            int code = uni(scan_trivial(start, mode, type, jump_ratio));     // chunks before `start` hold no unvisited vertex
            PROF_ADD(PF_TRIV_EVAL);
            if(code == SC_NEED) {
                if(lane == 0) {
                    uint64_t need = (uint64_t)HC.sw_need_lo | ((uint64_t)HC.sw_need_hi << 32); const int base = HC.sw_dom_base;
                    while(need) { int l = ffs64(need); need &= need - 1; HC.scr_i[l] = classify_trivial_vertex(base + l, mode == 1); }
                }
                wsync();
                if(uni(HC.sw_dom_base) > start) start = uni(HC.sw_dom_base);       // earlier chunks are done (no hit; their candidates are in sw_best_*)
                continue;
            }
            if(code == SC_BAD) { if(lane == 0) fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); wsync(); return true; }
            if(code == SC_HIT) target = uni(HC.sw_hit);
            break;
        }
        if(target < 0) {                             // the sweep is over
            if(flag) return true;
            if(mode == 0) return false;
            if(uni(HC.sw_best_v) < 0) { if(skippable) { if(lane == 0) HC.maybe_triv = 0; wsync(); } return false; }
            target = uni(HC.sw_best_v); last = true;
        }
        fired++;
        if(tr && lane == 0) { if(last) trace_emit(OP_TRIVIAL_BEST, vlog(target), type, HC.sw_best_r); else trace_emit(mode == 1 ? OP_TRIVIAL_NOW : OP_TRIVIAL_FAST, vlog(target), mode == 1 ? type : 0, HC.sw_hit_r); }
        decompose_trivial_vertex_wave(target);
        if(last) { wsync(); PROF_ADD(PF_TRIV_MUT); return true; }
        if(lane == 0) {
            if(uni(HC.hs_dirty)) hs_refresh_flags();
            HC.sw_dom_base = -1;
        }
        wsync();
        PROF_ADD(PF_TRIV_MUT);
        flag = true;
        if(HC.status) return true;
        start = target + 1;
    }
}

// scallop::resolve_smallest_edges (scallop.cc:844-945), with the per-vertex result (ratio, edge) kept per lane: removing
// edge s->t changes the lists of s and t only (degree guards elsewhere cannot flip while no degree drops to <= 1, phasing flags
// only change with hs_dirty), so after a removal just those two lanes evaluate again.  And while nothing else can fire -- no
// broken vertex, no type-1 trivial vertex, phasing flags untouched: exactly what R1..R3 would find out -- the next sweep of the
// reference's outer loop (scallop.cc:38-188) starts right here instead of going back through the cascade.
// "anything R1..R3 would react to": four context words asked for TOGETHER and combined -- as a chain of `||` each word was read, waited for and
// branched on before the next one was asked for: four dependent LDS round trips at the end of every iteration of the smallest-edge loop
ALD_INL bool back_to_cascade()
{
    const int a = HC.status, b = HC.maybe_broken, c = HC.maybe_triv, d = HC.hs_dirty;
    return uni(a | b | c | d) != 0;
}
// Classes 2 and up (ALD_KEEP, round 4): the per-vertex result (ratio, edge) of compute_smallest_edge + guards STAYS in the wave's slab between
// the sweeps.  A sweep that comes back after a trivial decomposition or an unsplittable vertex first evaluates the vertices the rules in
// between have marked (HotCtx::ev_dirty) -- gathered into DENSE lanes, whatever chunk they belong to: lanes run in lock step, so a chunk
// of 64 costs the same for one marked vertex as for 64 -- writes them to the slab, then loads every chunk with one coalesced read.  Up to
// round 3 every sweep evaluated every vertex: 15 % of a 500-vertex graph's time (profiles/r04/n_phase_big.txt).
enum { EV_NC = MAXV / ALD_WAVE };                 // chunks of the class: registers for the small classes, private memory beyond
ALD_INL bool sweep_smallest_body(double max_ratio, int &fired, const bool tr);
ALD_INL bool sweep_smallest(double max_ratio)
{
    int fired = 0; const bool tr = tracing_u();
    const bool rv = sweep_smallest_body(max_ratio, fired, tr);
    if(fired && lane_id() == 0) HC.n_iters += fired;
    return rv;
}
ALD_INL bool sweep_smallest_body(double max_ratio, int &fired, const bool tr)
{
    max_ratio = uni(max_ratio);
    const int lane = lane_id();
    const int vend = uni(HC.nv);
    constexpr int NC = EV_NC;
    if(vend > NC * ALD_WAVE) { if(lane == 0) fail(ALD_ST_CAPACITY); wsync(); return true; }     // cannot happen: nv <= MAXV
    PROF_DECL;
    if(lane == 0 && uni(HC.hs_dirty)) hs_refresh_flags();
    wsync();
    double cr[NC]; int ce[NC];                    // this lane's vertex of every chunk: (ratio, edge)
    const int nch0 = (vend + ALD_WAVE - 1) / ALD_WAVE;
#if ALD_KEEP
    ALD_GLOBAL double *mr = (ALD_GLOBAL double*)(HC.cold + CL::o_evr); ALD_GLOBAL int32_t *me = (ALD_GLOBAL int32_t*)(HC.cold + CL::o_eve);
    const bool stale_all = uni(HC.ev_all) != 0;
  #if defined(ALD_EMU) && defined(ALD_EMU_CHECK)
    { static long n_entry = 0, n_all = 0, n_marked = 0, n_live = 0; static struct P { long *a, *b, *c, *d; ~P() { if(*a && getenv("ALD_EMU_VERBOSE")) fprintf(stderr, "[emu-ev] class %d: %ld sweep entries, %.1f %% with everything stale, else %.1f marked of %.1f vertices\n", ALD_CLASS_ID, *a, 100.0 * *b / *a, (double)*c / ((*a - *b) ? (*a - *b) : 1), (double)*d / *a); } } pr{&n_entry, &n_all, &n_marked, &n_live};
      n_entry++; n_live += vend; if(stale_all) n_all++; else for(int i = 1; i < vend; i++) if((HC.ev_dirty[i >> 5] >> (i & 31)) & 1u) n_marked++; }
    // test build of the emulation: a kept evaluation that is neither marked nor covered by "everything" must equal a fresh one
    if(!stale_all) for(int i = 1; i < vend; i++) {
        if((HC.ev_dirty[i >> 5] >> (i & 31)) & 1u) continue;
        double r = 0; const int e = eval_smallest(i, r); const double kr_ = mr[i]; const int ke_ = me[i];
        if(e != ke_ || (e >= 0 && memcmp(&r, &kr_, 8) != 0)) { fprintf(stderr, "[check] graph %d: stale smallest-edge evaluation of vertex %d: kept (%d, %.17g), fresh (%d, %.17g)\n", HC.g, i, ke_, kr_, e, r); abort(); }
    }
  #endif
    if(!stale_all) {
        // the marked vertices, dense: lane l of a pass takes the (base + l)-th set bit of the mask -- found by walking the chunks' popcounts
        // (wave-uniform words) and a bit select --, evaluates that vertex and leaves the result in the slab
  #ifdef ALD_EMU
        for(int i = 1; i < vend; i++) if((HC.ev_dirty[i >> 5] >> (i & 31)) & 1u) { double r = 0; const int e = eval_smallest(i, r); mr[i] = r; me[i] = e; }
  #else
        int total = 0;
        for(int w = 0; w < (vend + 31) / 32; w++) total += (int)__builtin_popcount(uni(HC.ev_dirty[w]));
        for(int base = 0; base < total; base += ALD_WAVE) {
            const int k = base + lane; int run = 0, vtx = -1;
            for(int w = 0; w < (vend + 31) / 32; w++) {
                const uint32_t m = uni(HC.ev_dirty[w]); const int pc = (int)__builtin_popcount(m);
                if(k >= run && k < run + pc) vtx = 32 * w + nth_set_bit((uint64_t)m, k - run);
                run += pc;
            }
            if(vtx >= 1 && vtx < vend) { double r = 0; const int e = eval_smallest(vtx, r); mr[vtx] = r; me[vtx] = e; }
        }
  #endif
        wsync_mem();                              // (the records travel through the slab: the stores have to have left the wave before the reads are asked for)
        if(NC <= 16) { ALD_UNROLL for(int c = 0; c < (NC <= 16 ? NC : 1); c++) { const bool in = c < nch0; cr[c] = in ? mr[c * ALD_WAVE + lane] : 0.0; ce[c] = in ? me[c * ALD_WAVE + lane] : -1; } }
        else for(int c = 0; c < nch0; c++) { cr[c] = mr[c * ALD_WAVE + lane]; ce[c] = me[c * ALD_WAVE + lane]; }
    }
    else if(NC <= 16) { ALD_UNROLL for(int c = 0; c < (NC <= 16 ? NC : 1); c++) { cr[c] = 0; ce[c] = -1; } }
    auto keep = [&]() {                           // the evaluations stay behind for the next sweep
        if(NC <= 16) { ALD_UNROLL for(int c = 0; c < (NC <= 16 ? NC : 1); c++) if(c < nch0) { mr[c * ALD_WAVE + lane] = cr[c]; me[c * ALD_WAVE + lane] = ce[c]; } }
        else for(int c = 0; c < nch0; c++) { mr[c * ALD_WAVE + lane] = cr[c]; me[c * ALD_WAVE + lane] = ce[c]; }
    };
#else
    const bool stale_all = true;
    if(NC <= 16) { ALD_UNROLL for(int c = 0; c < (NC <= 16 ? NC : 1); c++) { cr[c] = 0; ce[c] = -1; } }
    auto keep = [&]() {};
#endif
    // Classes of up to 16 chunks keep the arrays in registers: every loop over the chunks is fully unrolled (constant indices) and the few
    // accesses with a run-time chunk number go through a select chain; with a run-time index they would live in scratch memory and
    // every sweep would wait for it chunk by chunk.  Larger classes use private memory.
    constexpr bool REG = (NC <= 16);
    auto cput = [&](int c, double r, int e) { if(!REG) { cr[c] = r; ce[c] = e; } else { ALD_UNROLL for(int k = 0; k < NC; k++) if(k == c) { cr[k] = r; ce[k] = e; } } };
    auto cget_e = [&](int c) -> int { if(!REG) return ce[c]; int v = -1; ALD_UNROLL for(int k = 0; k < NC; k++) if(k == c) v = ce[k]; return v; };
    auto eval_chunk = [&](int c, bool every, int ds, int dt) {        // (re-)evaluate this lane's vertex of chunk c
        int i = c * ALD_WAVE + lane; double r = 0; int e = -1; bool mine = i >= 1 && i < vend && (every || i == ds || i == dt);
        if(mine) e = eval_smallest(i, r);
        if(!REG) { if(mine) { cr[c] = r; ce[c] = e; } }
        else { ALD_UNROLL for(int k = 0; k < NC; k++) if(k == c && mine) { cr[k] = r; ce[k] = e; } }
    };
    if(stale_all) { for(int c = 0; c < nch0; c++) eval_chunk(c, true, -1, -1); }
#if ALD_KEEP
    wsync();
    ev_clear_marks(vend);
    wsync();
#endif
    PROF_ADD(PF_T_MERGE_KILL);                   // (profiling build: bringing the kept evaluations up to date at the sweep's entry)
    const bool may_chain = !(uni(HC.p_ratio[7]) > 1.0) && !uni(HC.any_strand);
    auto sweeps = [&]() -> bool {
    bool any = false;
    int guard = MAXE + 8;
    while(guard-- > 0) {                          // one sweep of the reference per iteration
        bool flag = false;
        double best_r = max_ratio; int best_e = -1, best_v = -1;
        int start = 1;
        while(start < vend) {
            int hit = -1, hit_e = -1; double hit_r = 0;
            {
            // (1) chunk by chunk, cheap: an invalid evaluation anywhere in the chunk, else the first "now" vertex (ratio < 0.01)
            const int c0 = start / ALD_WAVE; bool badw = false;
            ALD_UNROLL for(int c = 0; c < NC; c++) {
                if(c < c0 || c * ALD_WAVE >= vend || hit >= 0 || badw) continue;          // (wave-uniform)
                const int i = c * ALD_WAVE + lane;
                const int e = (i >= start && i < vend) ? ce[c] : -1;
                if(wballot(e == -3)) { badw = true; continue; }
                const uint64_t now = wballot(e >= 0 && cr[c] < 0.01);
                if(now) { const int l = ffs64(now); hit = c * ALD_WAVE + l; hit_e = wread(e, l); hit_r = wread(cr[c], l); }
            }
            if(badw) { if(lane == 0) fail(ALD_ST_INVARIANT + ALD_INV_WEIGHT); wsync(); return true; }
            // (2) the candidates before it: each lane folds its own chunks (later vertex wins ties), then ONE reduction across lanes
            {
                const int lim = hit >= 0 ? hit : vend;
                double rr = DBL_MAX; int vv = -1;
                ALD_UNROLL for(int c = 0; c < NC; c++) {
                    if(c < c0 || c * ALD_WAVE >= lim) continue;
                    const int i = c * ALD_WAVE + lane;
                    if(i >= start && i < lim && ce[c] >= 0 && (vv < 0 || !(rr < cr[c]))) { rr = cr[c]; vv = i; }
                }
                wave_argmin(rr, vv);                                         // (all 64 lanes are here: every branch around it is wave-uniform)
                if(vv >= 0 && !(best_r < rr)) { best_r = rr; best_v = vv; best_e = wread(cget_e(vv / ALD_WAVE), vv % ALD_WAVE); }   // if(ratio < r) continue;
            }
            }
            PROF_ADD(PF_SMALL_EVAL);
            if(hit < 0) break;
            const int ds = H.ed[hit_e].lk.es, dt = H.ed[hit_e].lk.et;         // the two vertices whose lists change
            fired++;
            if(tr && lane == 0) trace_emit(OP_SMALL_NOW, (int)H.eid[hit_e], vlog(hit), hit_r);
            kill_edge_wave(hit_e);
            if(lane == 0) hs_remove(hit_e);
            wsync();
            // other vertices only look at ds / dt through the guards out_deg[ds] > 1 and in_deg[dt] > 1 (both held for the edge just
            // removed); if one of them stops holding, or the phasing flags moved, every lane evaluates again
            const bool all = uni(HC.hs_dirty) != 0 || (int)uni(H.vx[ds].out_deg) <= 1 || (int)uni(H.vx[dt].in_deg) <= 1;
            if(uni(HC.hs_dirty)) { if(lane == 0) hs_refresh_flags(); wsync(); }
#if ALD_KEEP
            if(all) { wsync(); ev_clear_marks(vend); wsync(); }      // every vertex is evaluated again right here
#endif
            if(NC <= 2) { for(int c = 0; c < NC; c++) { int i = c * ALD_WAVE + lane; if(i >= 1 && i < vend && (all || i == ds || i == dt)) { cr[c] = 0; ce[c] = eval_smallest(i, cr[c]); } } }
            else if(all) { const int nch = (vend + ALD_WAVE - 1) / ALD_WAVE; for(int c = 0; c < nch; c++) eval_chunk(c, true, -1, -1); }
            else {                                                          // many chunks: go straight to the (one or two) chunks of ds and dt
                const int c1 = uni(ds) / ALD_WAVE, c2 = uni(dt) / ALD_WAVE;
                eval_chunk(c1, false, ds, dt);
                if(c2 != c1) eval_chunk(c2, false, ds, dt);
            }
            PROF_ADD(PF_SMALL_MUT);
            flag = true;
            start = hit + 1;
        }
        if(!flag) {
            if(best_e < 0) return any;
            const int ds = H.ed[best_e].lk.es, dt = H.ed[best_e].lk.et;
            fired++;
            if(tr && lane == 0) trace_emit(OP_SMALLEST, (int)H.eid[best_e], vlog(best_v), best_r);
            kill_edge_wave(best_e);
            if(lane == 0) hs_remove(best_e);
            wsync();
#ifdef ALD_PROF
            { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[PF_SM_KILL] += t1_ - prof_t_; }
            const unsigned long long prof_re_ = __builtin_readcyclecounter();
#endif
            any = true;
            // back to the cascade unless R1..R3 provably have nothing to do
            if(!may_chain || back_to_cascade()) { PROF_ADD(PF_SMALL_MUT); return true; }
            if(NC <= 2) { for(int c = 0; c < NC; c++) { int i = c * ALD_WAVE + lane; if(i >= 1 && i < vend && (i == ds || i == dt)) { cr[c] = 0; ce[c] = eval_smallest(i, cr[c]); } } }
            else {
                const int c1 = uni(ds) / ALD_WAVE, c2 = uni(dt) / ALD_WAVE;
                eval_chunk(c1, false, ds, dt);
                if(c2 != c1) eval_chunk(c2, false, ds, dt);
            }
#ifdef ALD_PROF
            { unsigned long long t1_ = __builtin_readcyclecounter(); if(lane_id() == 0) HC.prof[PF_SM_REEVAL] += t1_ - prof_re_; }
#endif
            PROF_ADD(PF_SMALL_MUT);
        } else {
            any = true;
            if(!may_chain || back_to_cascade()) return true;
        }
    }
    return true;
    };
    const bool rv = sweeps();
    keep();
    return rv;
}

// ---------------------------------------------------------------- router (scallop/router.cc), scalar on lane 0
// Results in HC.ro_type / HC.ro_degree / HC.ro_ratio / HC.ro_npairs; pe2w pairs (sorted, clamped) in the pair area.
// SMALL: every router array lives in the LDS scratch (the common case; the compiler then knows the address space and emits ds_*
// instead of flat_* accesses); otherwise in the slab's work arrays
template<bool SMALL> ALD_INL bool router_body(int root, int want_type, int max_degree, int pre)
{
#ifdef ALD_EMU_COUNT
    g_cnt_pre[pre < 0 ? 0 : pre > 2 ? 2 : pre]++;
#endif
    COLD;
    // ---- build_indices (router.cc:225-248)
    int nin = uni(H.vx[root].in_deg), nout = uni(H.vx[root].out_deg), n = nin + nout;
    const int route_bound = (HC.hl_n == 0) ? 0 : nin * nout;          // routes only come from phasing lists
    HC.pw_lds = SMALL ? 1 : 0;
    const Arena AR = arena_at(SMALL);
    const Pairs PW = pairs_at(SMALL, false);
    const int cap = AR.cap_i;
    if(ALD_UNLIKELY(5 * n > cap)) { fail(ALD_ST_CAPACITY); return false; }
    int32_t *u2e = AR.i;
    if(!pre) { int k = 0; for(int e = u_first_in(root); e >= 0; e = u_next_in(e)) { u2e[k++] = e; }
      for(int e = u_first_out(root); e >= 0; e = u_next_out(e)) { u2e[k++] = e; } }
    if(ALD_UNLIKELY(mixed_strand_vertex(root))) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }     // router.cc:71-76
    // ---- routes from the phasing lists (hyper_set::get_routes, hyper_set.cc:553-571), gathered in the pair area
    const int half = PW.cap;
    int nr = 0;
    int32_t *ra = PW.a, *rb = PW.b; double *rc = PW.w;
    {
        int nl = HC.hl_n;
        for(int k = 0; k < nl; k++) {
            ALD_GLOBAL int32_t *v = C.hl + uni(C.hl_off[k]); int len = uni(C.hl_len[k]); int c = uni(C.hl_cnt[k]);
            for(int i = 0; i + 1 < len; i++) {
                int x = v[i], y = v[i + 1];
                if(x < 0 || y < 0) continue;
                if(H.ed[x].lk.es == NIL || (int)uni(H.ed[x].lk.et) != root) continue;
                int f = -1;
                for(int j = 0; j < nr; j++) if(ra[j] == x && rb[j] == y) { f = j; break; }
                if(f >= 0) rc[f] += c;
                else { if(ALD_UNLIKELY(nr >= half)) { fail(ALD_ST_CAPACITY); return false; } ra[nr] = x; rb[nr] = y; rc[nr] = c; nr++; }
            }
        }
        sort_pairs(PW, nr);                    // MPII order == (id(e1), id(e2)) == creation order of the ug edges
    }
    // ---- arena (sized now that the number of routes is known)
    int maxue = nr + n;                        // + one edge per isolated node
    int o = n;
    int32_t *udeg = AR.i + o; o += n;
    int32_t *comp = AR.i + o; o += n;
    int32_t *queue = AR.i + o; o += n;
    int32_t *iso = AR.i + o; o += n;          // isolated flag -> econf pending
    int32_t *us = AR.i + o; o += maxue;
    int32_t *ut = AR.i + o; o += maxue;
    int32_t *ualive = AR.i + o; o += maxue;
    if(ALD_UNLIKELY(o > cap || 3 * n + maxue > AR.cap_d)) { fail(ALD_ST_CAPACITY); return false; }
    double *vw = AR.d, *uw = AR.d + n, *econf = AR.d + n + maxue;
    // ---- build_bipartite_graph (router.cc:250-325)
    int nue = 0;
    for(int i = 0; i < n; i++) udeg[i] = 0;
    for(int j = 0; j < nr; j++) {
        int y = rb[j];
        if(ALD_UNLIKELY(H.ed[y].lk.es == NIL || (int)uni(H.ed[y].lk.es) != root)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }   // assert(e2u.find(e2) != end)
        int s = -1, t = -1;                       // local indices by search (routes exist only with phasing paths)
        for(int q = 0; q < nin; q++) if(u2e[q] == ra[j]) { s = q; break; }
        for(int q = nin; q < n; q++) if(u2e[q] == y) { t = q; break; }
        if(ALD_UNLIKELY(s < 0 || t < 0)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
        us[nue] = s; ut[nue] = t; uw[nue] = rc[j]; ualive[nue] = 1; udeg[s]++; udeg[t]++; nue++;
    }
    // isolated vertices attach to the best partner by shared sample abundance (router.cc:1010-1129).
    // Per-node support is fetched once (one round of independent loads); single-sample pairs are then pure arithmetic.
    int32_t *ncnt = comp, *nsid = queue;         // comp / queue are not needed before classify: reuse them as (count, first sample id)
    double *nabd = AR.d;                         // [n] abundance of the first sample (the place of vw, which build() fills much later)
    if(!pre) {
        for(int v = 0; v < n; v++) { int e = u2e[v]; ncnt[v] = (int32_t)uni(C.ed[e].sp_len); nsid[v] = uni(C.ed[e].s0id); nabd[v] = uni(C.ed[e].s0abd); }
        for(int v = 0; v < n; v++) iso[v] = (C.ed[u2e[v]].ecount == 0) ? 2 : 0;      // 2 = "Warning!(count = 0)": not in left / right
    }
#define ALD_COMMON(l, r) ((ncnt[l] == 1 && ncnt[r] == 1) ? ((nsid[l] == nsid[r]) ? (0.0 + (0.99 * ((nabd[r] < nabd[l]) ? nabd[r] : nabd[l]) + 0.01 * ((nabd[l] < nabd[r]) ? nabd[r] : nabd[l]))) : 0.0) : common_abd(u2e[l], u2e[r]))
    if(pre == 2) {
        // the wave left every node's (partner, shared abundance, log of its share of the total) in comp / nabd / econf (router_prepare): what stays
        // sequential is which nodes are still isolated when their turn comes
        for(int v = 0; v < n; v++) {
            if(iso[v] == 2) continue;
            if(udeg[v] != 0) continue;
            const int partner = comp[v];
            if(ALD_UNLIKELY(partner < 0)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
            us[nue] = v < nin ? v : partner; ut[nue] = v < nin ? partner : v; uw[nue] = nabd[v]; ualive[nue] = 1; udeg[v]++; udeg[partner]++; nue++;
            iso[v] = 1;
        }
    } else {
    for(int v = 0; v < nin; v++) {
        if(iso[v] == 2) continue;
        if(udeg[v] != 0) continue;
        int partner = -1; double max_abd = 0.0, sum_abd = 0.0;
        for(int r = nin; r < n; r++) { if(iso[r] == 2) continue; double c = ALD_COMMON(v, r); sum_abd += c; if(c > max_abd) { max_abd = c; partner = r; } }
        if(ALD_UNLIKELY(partner < 0)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
        us[nue] = v; ut[nue] = partner; uw[nue] = max_abd; ualive[nue] = 1; udeg[v]++; udeg[partner]++; nue++;
        iso[v] = 1; econf[v] = max_abd / sum_abd;        // the log is taken where it is used (end of build())
    }
    for(int v = nin; v < n; v++) {
        if(iso[v] == 2) continue;
        if(udeg[v] != 0) continue;
        int partner = -1; double max_abd = 0.0, sum_abd = 0.0;
        for(int l = 0; l < nin; l++) { if(iso[l] == 2) continue; double c = ALD_COMMON(l, v); sum_abd += c; if(c > max_abd) { max_abd = c; partner = l; } }
        if(ALD_UNLIKELY(partner < 0)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
        us[nue] = partner; ut[nue] = v; uw[nue] = max_abd; ualive[nue] = 1; udeg[v]++; udeg[partner]++; nue++;
        iso[v] = 1; econf[v] = max_abd / sum_abd;        // the log is taken where it is used (end of build())
    }
    }
    // ---- classify_plain_vertex (router.cc:116-171)
    HC.ro_npairs = 0; HC.ro_ratio = 0;
    if(nin == 1 || nout == 1) { HC.ro_type = T_TRIVIAL; HC.ro_degree = n; return true; }
    for(int i = 0; i < n; i++) if(ALD_UNLIKELY(udeg[i] < 1)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
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
        if(ALD_UNLIKELY(b1 || b2)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
        int a = 0, b = 0;
        for(int c = 0; c < ncomp; c++) { int sz = 0; for(int i = 0; i < n; i++) if(comp[i] == c) sz++; if(sz == 1) a++; if(sz >= 2) b++; }
        if(ALD_UNLIKELY(b < 1)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
        rtype = T_SPLITTABLE_PURE; rdeg = b - 1 + (a + 1) / 2;
    }
    HC.ro_type = rtype; HC.ro_degree = rdeg;
    if(rtype != want_type) return true;
    if(rdeg > max_degree) return true;
#ifdef ALD_EMU_COUNT
    g_cnt_build++;
#endif
    // ---- build() -> thread() (router.cc:193-223, 738-857)
    // compute_balanced_weights_components (router.cc:1248-1275): components by smallest member, members ascending
    for(int i = 0; i < n; i++) vw[i] = 0;
    for(int c = 0; c < ncomp; c++) {
        double sum1 = 0, sum2 = 0;
        for(int i = 0; i < n; i++) { if(comp[i] != c) continue; double wgt = uni(H.ed[u2e[i]].w); if(i < nin) sum1 += wgt; else sum2 += wgt; vw[i] = wgt; }
        double r1 = sqrt(sum2 / sum1), r2 = sqrt(sum1 / sum2);
        for(int i = 0; i < n; i++) { if(comp[i] != c) continue; if(i < nin) vw[i] *= r1; else vw[i] *= r2; }
    }
    double weight_sum = 0;
    for(int i = 0; i < n; i++) weight_sum += vw[i];
    int32_t *pa = PW.a, *pb = PW.b; double *pwt = PW.w; int np = 0;
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
            if(ALD_UNLIKELY(np >= half)) { fail(ALD_ST_CAPACITY); return false; }
            pa[np] = PMAKE(u2e[s], s); pb[np] = PMAKE(u2e[t], t); pwt[np] = vw[x]; np++;
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
        for(int t = 0; t < n; t++) for(int k = 0; k < nue; k++) { if(!ualive[k]) continue; int y = (us[k] == x) ? ut[k] : ((ut[k] == x) ? us[k] : -1); if(y != t) continue; sum += uw[k]; if(ALD_UNLIKELY(!(vw[t] >= vw[x]))) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; } }
        for(int t = 0; t < n; t++) for(int k = 0; k < nue; k++) {
            if(!ualive[k]) continue; int y = (us[k] == x) ? ut[k] : ((ut[k] == x) ? us[k] : -1); if(y != t) continue;
            double wgt = vw[x] * uw[k] / sum;
            if(ALD_UNLIKELY(np >= half)) { fail(ALD_ST_CAPACITY); return false; }
            if(x < t) { pa[np] = PMAKE(u2e[x], x); pb[np] = PMAKE(u2e[t], t); } else { pa[np] = PMAKE(u2e[t], t); pb[np] = PMAKE(u2e[x], x); }
            pwt[np] = wgt; np++;
            vw[t] -= wgt;
        }
        vw[x] = -1;
        for(int q = 0; q < nue; q++) if(ualive[q] && (us[q] == x || ut[q] == x)) { ualive[q] = 0; udeg[us[q]]--; udeg[ut[q]]--; live--; }
    }
    if(ALD_UNLIKELY(live != 0)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return false; }
    double weight_remain = 0;
    for(int i = 0; i < n; i++) { if(vw[i] <= 0) continue; weight_remain += vw[i]; }
    HC.ro_ratio = weight_remain / weight_sum;
    // router.cc:849-855: side effect of every build().  (pre: the wave fetched the confidences with the rest -- router_prepare --, so
    // the updates are stores only instead of one global round trip per attached node)
    if(pre == 2) { const double *ecf = AR.d + ARENA_D - n; for(int i = 0; i < n; i++) if(iso[i] == 1) C.ed[u2e[i]].econf = ecf[i] + econf[i]; }     // econf[] already holds the logarithms
    else if(pre) { const double *ecf = AR.d + ARENA_D - n; for(int i = 0; i < n; i++) if(iso[i] == 1) C.ed[u2e[i]].econf = ecf[i] + log(econf[i]); }
    else for(int i = 0; i < n; i++) if(iso[i] == 1) C.ed[u2e[i]].econf += log(econf[i]);
    sort_pairs(PW, np);
    const double mw = HC.p_min_w;
    for(int i = 0; i < np; i++) if(pwt[i] < mw) pwt[i] = mw;                // router.cc:217-220
    HC.ro_npairs = np;
    return true;
}
// ---------------------------------------------------------------------------------------------------------------------------------
// The router for the vertex it nearly always sees (93 % of its runs on the bench workload): 2 in-edges + 2 out-edges, no phasing
// routes, one supporting sample per edge -- i.e. router_prepare returned level 2 and left, for the four nodes (0, 1 = the in-edges,
// 2, 3 = the out-edges, adjacency order), the partner each would attach to, the shared abundance and the logarithm of its share.
// Same statements as router_body (router.cc:250-325 with every node isolated, 116-171, 738-936, 1010-1129, 1248-1275), but on
// NAMED scalars instead of arrays in LDS: four node weights, four degrees, at most three edges of the bipartite graph
//   in-nodes attach first, in index order:   k0 = (0, p0), k1 = (1, p1)
//   an out-node nobody attached to then attaches itself:   k2 = (partner, that out-node)     (only when p0 == p1)
// so the graph is either two components of one edge each (SPLITTABLE_PURE, degree 1) or one path of three edges
// (UNSPLITTABLE_SINGLE, degree 1).  thread() runs as in the general form -- leaf edges in creation order, else one turn at the
// lightest node of degree >= 2, ties to the later node -- with run-time node indices resolved by select chains over the four
// scalars (a handful of v_cndmask each) instead of dependent LDS round trips.  Every floating-point operation is the one the general
// form performs, on the same operands in the same order.  Returns -1 when the vertex is not of this shape (the general form runs),
// else what router_body returns.
#define R4_GET(a0, a1, a2, a3, i) ((i) == 0 ? (a0) : ((i) == 1 ? (a1) : ((i) == 2 ? (a2) : (a3))))
#define R4_SET(a0, a1, a2, a3, i, val) do { if((i) == 0) a0 = (val); else if((i) == 1) a1 = (val); else if((i) == 2) a2 = (val); else a3 = (val); } while(0)
ALD_FN int router_22(int root, int want_type, int max_degree)
{
    root = uni(root); want_type = uni(want_type); max_degree = uni(max_degree);
    COLD;
    const Arena AR = arena_at(true);
    const Pairs PW = pairs_at(true, false);
    const int n = 4;
    const int32_t *u2e = AR.i, *part = AR.i + 2 * n, *iso = AR.i + 4 * n;
    const double *nabd = AR.d, *share = AR.d + 2 * n, *ecf = AR.d + ARENA_D - n;
    const int i0 = uni(iso[0]), i1 = uni(iso[1]), i2 = uni(iso[2]), i3 = uni(iso[3]);
    if((i0 | i1 | i2 | i3) != 0) return -1;                                   // an edge with count == 0 ("Warning!"): general form
    const int p0 = uni(part[0]), p1 = uni(part[1]);
    if(p0 < 2 || p1 < 2) return -1;                                           // no partner (the general form reports it)
    HC.pw_lds = 1;
    if(ALD_UNLIKELY(mixed_strand_vertex(root))) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return 0; }     // router.cc:71-76
    const int e0 = uni(u2e[0]), e1 = uni(u2e[1]), e2 = uni(u2e[2]), e3 = uni(u2e[3]);
    // ---- build_bipartite_graph + the attachments (router.cc:250-325, 1010-1129)
    int nue = 2; int ks0 = 0, kt0 = p0, ks1 = 1, kt1 = p1, ks2 = 0, kt2 = 0;                   // edge k: (in-node, out-node)
    double uw0 = uni(nabd[0]), uw1 = uni(nabd[1]), uw2 = 0;
    int d0 = 1, d1 = 1, d2 = (p0 == 2) + (p1 == 2), d3 = (p0 == 3) + (p1 == 3);
    int q = -1;                                                                // the out-node that attaches itself (iso[q] becomes 1)
    if(p0 == p1) {
        q = 5 - p0;
        const int r = uni(part[q]);
        if(r < 0 || r > 1) return -1;
        ks2 = r; kt2 = q; uw2 = uni(nabd[q]); nue = 3;
        if(r == 0) d0++; else d1++;
        if(q == 2) d2++; else d3++;
    }
    // ---- classify_plain_vertex (router.cc:116-171): one path of three edges, or two components of one edge each
    HC.ro_npairs = 0; HC.ro_ratio = 0;
    const int rtype = (nue == 3) ? T_UNSPLITTABLE_SINGLE : T_SPLITTABLE_PURE, rdeg = 1;    // nue - n + 2 = 1 / b - 1 + (a + 1) / 2 = 1 with a = 0, b = 2
    HC.ro_type = rtype; HC.ro_degree = rdeg;
    if(rtype != want_type) return 1;
    if(rdeg > max_degree) return 1;
#ifdef ALD_EMU_COUNT
    g_cnt_build++;
#endif
    // ---- compute_balanced_weights_components (router.cc:1248-1275)
    const double w0 = uni(H.ed[e0].w), w1 = uni(H.ed[e1].w), w2 = uni(H.ed[e2].w), w3 = uni(H.ed[e3].w);
    double v0, v1, v2, v3;
    if(nue == 3) {
        double sum1 = 0, sum2 = 0; sum1 += w0; sum1 += w1; sum2 += w2; sum2 += w3;
        const double r1 = sqrt(sum2 / sum1), r2 = sqrt(sum1 / sum2);
        v0 = w0 * r1; v1 = w1 * r1; v2 = w2 * r2; v3 = w3 * r2;
    } else {
        // component 0 = {0, p0}, component 1 = {1, p1}
        const double wp0 = p0 == 2 ? w2 : w3, wp1 = p1 == 2 ? w2 : w3;
        double s1 = 0, s2 = 0; s1 += w0; s2 += wp0;
        const double a1 = sqrt(s2 / s1), a2 = sqrt(s1 / s2);
        double t1 = 0, t2 = 0; t1 += w1; t2 += wp1;
        const double b1 = sqrt(t2 / t1), b2 = sqrt(t1 / t2);
        v0 = w0 * a1; v1 = w1 * b1;
        v2 = (p0 == 2) ? w2 * a2 : w2 * b2; v3 = (p0 == 3) ? w3 * a2 : w3 * b2;
    }
    double weight_sum = 0; weight_sum += v0; weight_sum += v1; weight_sum += v2; weight_sum += v3;
    // ---- thread() (router.cc:738-936)
    int32_t *pa = PW.a, *pb = PW.b; double *pwt = PW.w; int np = 0;
    bool al0 = true, al1 = true, al2 = (nue == 3); int live = nue;
    #define R4_VW(i) R4_GET(v0, v1, v2, v3, i)
    #define R4_DEG(i) R4_GET(d0, d1, d2, d3, i)
    #define R4_U2E(i) R4_GET(e0, e1, e2, e3, i)
    #define R4_CLEAR(x_) do { \
        if(al0 && (ks0 == (x_) || kt0 == (x_))) { al0 = false; live--; R4_SET(d0, d1, d2, d3, ks0, R4_DEG(ks0) - 1); R4_SET(d0, d1, d2, d3, kt0, R4_DEG(kt0) - 1); } \
        if(al1 && (ks1 == (x_) || kt1 == (x_))) { al1 = false; live--; R4_SET(d0, d1, d2, d3, ks1, R4_DEG(ks1) - 1); R4_SET(d0, d1, d2, d3, kt1, R4_DEG(kt1) - 1); } \
        if(al2 && (ks2 == (x_) || kt2 == (x_))) { al2 = false; live--; R4_SET(d0, d1, d2, d3, ks2, R4_DEG(ks2) - 1); R4_SET(d0, d1, d2, d3, kt2, R4_DEG(kt2) - 1); } } while(0)
    for(int guard = 0; guard < 8; guard++) {
        // thread_leaf (router.cc:859-897): edges in creation order
        int fx = -1, fy = -1, fs = 0, ft = 0;
        #define R4_LEAF(al, ks, kt) if(fx < 0 && (al)) { const int s_ = (ks), t_ = (kt); const double vs_ = R4_VW(s_), vt_ = R4_VW(t_); \
            if(!(vs_ < -0.5) && !(vt_ < -0.5)) { if(R4_DEG(s_) == 1 && vs_ <= vt_) { fx = s_; fy = t_; fs = s_; ft = t_; } else if(R4_DEG(t_) == 1 && vt_ <= vs_) { fx = t_; fy = s_; fs = s_; ft = t_; } } }
        R4_LEAF(al0, ks0, kt0) R4_LEAF(al1, ks1, kt1) R4_LEAF(al2, ks2, kt2)
        #undef R4_LEAF
        if(fx >= 0) {
            const double vx = R4_VW(fx);
            pa[np] = PMAKE(R4_U2E(fs), fs); pb[np] = PMAKE(R4_U2E(ft), ft); pwt[np] = vx; np++;
            R4_CLEAR(fx);
            const double vy = R4_VW(fy) - vx;
            R4_SET(v0, v1, v2, v3, fy, vy); R4_SET(v0, v1, v2, v3, fx, -1.0);
            continue;
        }
        // thread_turn (router.cc:899-936): the lightest node of degree >= 2, ties to the later one
        int x = -1; double vx = 0;
        #define R4_TURN(k_, vk_, dk_) if(!((vk_) < -0.5) && (dk_) > 1 && !(x != -1 && (vk_) > vx)) { x = (k_); vx = (vk_); }
        R4_TURN(0, v0, d0) R4_TURN(1, v1, d1) R4_TURN(2, v2, d2) R4_TURN(3, v3, d3)
        #undef R4_TURN
        if(x == -1) break;
        // out_edges(x): by the other endpoint (ascending node index), then creation; sum first, then the shares
        double sum = 0; bool bad = false;
        for(int t = 0; t < 4; t++) {
            #define R4_SUM(al, ks, kt, uw) if(al) { const int y_ = ((ks) == x) ? (kt) : (((kt) == x) ? (ks) : -1); if(y_ == t) { sum += (uw); if(!(R4_VW(t) >= vx)) bad = true; } }
            R4_SUM(al0, ks0, kt0, uw0) R4_SUM(al1, ks1, kt1, uw1) R4_SUM(al2, ks2, kt2, uw2)
            #undef R4_SUM
        }
        if(ALD_UNLIKELY(bad)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return 0; }
        for(int t = 0; t < 4; t++) {
            #define R4_SHARE(al, ks, kt, uw) if(al) { const int y_ = ((ks) == x) ? (kt) : (((kt) == x) ? (ks) : -1); if(y_ == t) { \
                const double wgt = vx * (uw) / sum; const int a_ = x < t ? x : t, b_ = x < t ? t : x; \
                pa[np] = PMAKE(R4_U2E(a_), a_); pb[np] = PMAKE(R4_U2E(b_), b_); pwt[np] = wgt; np++; \
                const double vt_ = R4_VW(t) - wgt; R4_SET(v0, v1, v2, v3, t, vt_); } }
            R4_SHARE(al0, ks0, kt0, uw0) R4_SHARE(al1, ks1, kt1, uw1) R4_SHARE(al2, ks2, kt2, uw2)
            #undef R4_SHARE
        }
        R4_SET(v0, v1, v2, v3, x, -1.0);
        R4_CLEAR(x);
    }
    #undef R4_VW
    #undef R4_DEG
    #undef R4_U2E
    #undef R4_CLEAR
    if(ALD_UNLIKELY(live != 0)) { fail(ALD_ST_INVARIANT + ALD_INV_ROUTER); return 0; }
    double weight_remain = 0;
    if(!(v0 <= 0)) weight_remain += v0;
    if(!(v1 <= 0)) weight_remain += v1;
    if(!(v2 <= 0)) weight_remain += v2;
    if(!(v3 <= 0)) weight_remain += v3;
    HC.ro_ratio = weight_remain / weight_sum;
    // router.cc:849-855, the side effect of every build(): the attached nodes' confidences (logarithms taken by router_prepare)
    C.ed[e0].econf = uni(ecf[0]) + uni(share[0]);
    C.ed[e1].econf = uni(ecf[1]) + uni(share[1]);
    if(q == 2) C.ed[e2].econf = uni(ecf[2]) + uni(share[2]);
    if(q == 3) C.ed[e3].econf = uni(ecf[3]) + uni(share[3]);
    sort_pairs(PW, np);
    const double mw = HC.p_min_w;
    for(int i = 0; i < np; i++) if(pwt[i] < mw) pwt[i] = mw;                // router.cc:217-220
    HC.ro_npairs = np;
    return 1;
}
#undef R4_GET
#undef R4_SET
ALD_FN bool router_small(int root, int want_type, int max_degree, int pre) { return router_body<true>(uni(root), uni(want_type), uni(max_degree), uni(pre)); }
ALD_FN bool router_large(int root, int want_type, int max_degree) { return router_body<false>(uni(root), uni(want_type), uni(max_degree), 0); }
// What the router needs of the root's edges, fetched by the WHOLE WAVE before lane 0 runs it (ALL lanes call; returns the `pre` level
// for router_run).  Lane l walks to the l-th local edge (in-edges, then out-edges) and asks for ITS support record -- one round of
// global loads for the vertex instead of one round trip per edge on lane 0.  And where the attachment of isolated nodes is a pure
// table (no phasing routes, every edge supported by ONE sample: router.cc:1010-1129 reduces to arithmetic on (sample id, abundance)
// pairs) lane l also finds the partner l would attach to -- best shared abundance, first of equals, with the sum over all candidates
// in their order -- and leaves (partner, that abundance, the logarithm of its share) where the sequential attachment picks them up.
ALD_FN int router_prepare(int root)
{
    COLD;
    root = uni(root);
    const int lane = lane_id();
    const int nin = uni(H.vx[root].in_deg), nout = uni(H.vx[root].out_deg), n = nin + nout;
    const int route_bound = (uni(HC.hl_n) == 0) ? 0 : nin * nout;
    const bool small = (route_bound + n <= LP) && (5 * n + 3 * (route_bound + n) <= ARENA_I) && (3 * n + route_bound + n <= ARENA_D);
    if(!small) return 0;                                                     // (n <= LP <= the wave)
    const Arena AR = arena_at(true);
    int32_t *u2e = AR.i, *ncnt = AR.i + 2 * n, *nsid = AR.i + 3 * n, *iso = AR.i + 4 * n; double *nabd = AR.d, *econf = AR.d + 2 * n;
    double *ecf = AR.d + ARENA_D - n;             // the edges' confidences, for the end of build(): the last n slots (`small` keeps 4 n + routes <= ARENA_D)
    bool multi = false;
    for(int l = lane; l < n; l += ALD_WAVE) {
        int e;
        if(l < nin) { e = first_in(root); for(int k = 0; k < l && e >= 0; k++) e = next_in(e); }
        else { e = first_out(root); for(int k = nin; k < l && e >= 0; k++) e = next_out(e); }
        if(e < 0) e = 0;                                                       // cannot happen: the degrees count the lists
        const int cnt = (int)C.ed[e].sp_len, ec = C.ed[e].ecount;
        u2e[l] = e; ncnt[l] = cnt; nsid[l] = C.ed[e].s0id; nabd[l] = C.ed[e].s0abd; iso[l] = (ec == 0) ? 2 : 0; ecf[l] = C.ed[e].econf;
        if(ec != 0 && cnt != 1) multi = true;
    }
    wsync();
    if(route_bound != 0 || wballot(multi) != 0) return 1;
    // attachment table (maxue == n here, so econf sits at AR.d + 2 n)
    int partner[(LP + ALD_WAVE - 1) / ALD_WAVE]; double mabd[(LP + ALD_WAVE - 1) / ALD_WAVE], share[(LP + ALD_WAVE - 1) / ALD_WAVE];
    { int q = 0; for(int v = lane; v < n; v += ALD_WAVE, q++) {
        int pt = -1; double max_abd = 0.0, sum_abd = 0.0;
        const int lo = v < nin ? nin : 0, hi = v < nin ? n : nin;
        const int sv = nsid[v]; const double av = nabd[v];
        if(iso[v] != 2) for(int r = lo; r < hi; r++) {
            if(iso[r] == 2) continue;
            const double ar = nabd[r];
            // common abundance of (left, right) = 0.99 * min + 0.01 * max of the two abundances when the samples agree; the left node is the in-edge
            const double al = v < nin ? av : ar, arr = v < nin ? ar : av;
            const double c = (nsid[r] == sv) ? (0.0 + (0.99 * ((arr < al) ? arr : al) + 0.01 * ((al < arr) ? arr : al))) : 0.0;
            sum_abd += c; if(c > max_abd) { max_abd = c; pt = r; }
        }
        partner[q] = pt; mabd[q] = max_abd; share[q] = log(max_abd / sum_abd);      // the logarithm build() adds to the edge's confidence (router.cc:849-855): one per lane, at once
    } }
    wsync();                                                                   // every lane has read ncnt / nsid / nabd: their places take the results
    { int q = 0; for(int v = lane; v < n; v += ALD_WAVE, q++) { ncnt[v] = partner[q]; nabd[v] = mabd[q]; econf[v] = share[q]; } }
    wsync();
    return 2;
}
ALD_INL bool router_run(int root, int want_type, int max_degree, int pre = 0)
{
#ifdef ALD_EMU_COUNT
    g_cnt_router++; { int dg_ = (int)H.vx[root].in_deg + (int)H.vx[root].out_deg; g_cnt_rdeg[dg_ < 33 ? dg_ : 33]++; }
#endif
    root = uni(root);
    const int nin = uni(H.vx[root].in_deg), nout = uni(H.vx[root].out_deg), n = nin + nout;
    const int route_bound = (uni(HC.hl_n) == 0) ? 0 : nin * nout;
    const bool small = (route_bound + n <= LP) && (5 * n + 3 * (route_bound + n) <= ARENA_I) && (3 * n + route_bound + n <= ARENA_D);
#ifndef ALD_NO_ROUTER22
    if(pre == 2 && nin == 2 && nout == 2) { const int r22 = uni(router_22(root, want_type, max_degree)); if(r22 >= 0) return r22 != 0; }
#endif
    return small ? uni(router_small(root, want_type, max_degree, pre)) : uni(router_large(root, want_type, max_degree));
}
// park / un-park the best candidate's pe2w while an unsplittable sweep goes on
ALD_FN void save_pairs(int n)
{
    n = uni(n);
    const bool lds = uni(HC.pw_lds) != 0;
    if(ALD_UNLIKELY(n > (lds ? (int)LP : (int)PW_CAP))) { fail(ALD_ST_CAPACITY); return; }
    if(lds) copy_pairs<true>(false, true, n); else copy_pairs<false>(false, true, n);      // (one address space per copy: no FLAT accesses)
    HC.park_lds = lds ? 1 : 0;
}
ALD_FN void restore_pairs(int n)
{
    n = uni(n);
    const bool lds = uni(HC.park_lds) != 0;
    HC.pw_lds = lds ? 1 : 0;
    if(lds) copy_pairs<true>(true, false, n); else copy_pairs<false>(true, false, n);
}
// forget every vertex's remembered router class (the graph changed); called by ALL lanes
ALD_INL void memo_clear()
{
    const int nv = uni(HC.nv);
    for(int i = lane_id(); i < nv; i += ALD_WAVE) H.nz[i] &= NZ_MEMBER;
    wsync();
}
// scallop::resolve_unsplittable_vertex (scallop.cc:1004-1060)
#ifdef ALD_UNSPLIT_CALL
ALD_FN bool sweep_unsplittable(int type, int degree, double max_ratio)
#else
ALD_INL bool sweep_unsplittable(int type, int degree, double max_ratio)
#endif
{
#ifdef ALD_EMU_COUNT
    g_cnt_unsweep++;
#endif
    type = uni(type); degree = uni(degree); max_ratio = uni(max_ratio);
    COLD;
    const int lane = lane_id();
    int vend = HC.nv;
    bool flag = false;
    int root = -1; double ratio = max_ratio; int best_np = 0;      // meaningful on lane 0 only
    // The sweep is sequential in the reference: a decomposition (and, for jump_ratio > 1, the trivial decompositions nested in
    // it) can make a LATER vertex newly eligible, so the next candidate is looked up again after every action.
    int cur = 1;
    while(cur < vend) {
        int i = -1;
        for(int base = (cur / ALD_WAVE) * ALD_WAVE; base < vend && i < 0; base += ALD_WAVE) {
            int i0 = base + lane;
            const bool inr = (i0 >= cur) & (i0 < vend); const int ii = inr ? i0 : 0;
            const Hot::VertexHot vr = H.vx[ii];
            const int nzv = H.nz[ii] & NZ_MEMBER, d1 = vr.in_deg, d2 = vr.out_deg;
            uint64_t m = wballot(inr & (nzv != 0) & (d1 >= 2) & (d2 >= 2));
            if(m) i = base + ffs64(m);
        }
        if(i < 0) break;
        int act = 0;
        // (the memo test below, on every lane: the same LDS byte) -- when the router is going to run, the wave prepares its inputs
        int pre = 0;
        { const int mm = uni(H.nz[i]);
          const bool skip = (mm & NZ_MEMO_VALID) && (((mm >> NZ_MEMO_TYPE_SHIFT) & 7) != type || ((mm & NZ_MEMO_DEG_GT1) && degree <= 1));
          if(!skip) pre = router_prepare(i); }
        if(lane == 0) {
            PROF_DECL;
            // classify() comes first in the reference and build() -- with its side effect on the edge confidences -- only runs when
            // type and degree fit (router.cc:61-81, scallop.cc:1004-1030).  The six passes of one iteration classify the same vertices
            // on the same graph: the class is kept per vertex -- in the spare bits of its nonzeroset byte in LDS, wiped whenever the
            // graph changes (memo_clear) -- and a pass that cannot use the vertex does not run the router again.  (The sweeps only ever
            // ask for degree <= 1 or any degree: one bit says which side of that line the vertex is on.)
            const int mm = uni(H.nz[i]);
            bool rok;
            if((mm & NZ_MEMO_VALID) && (((mm >> NZ_MEMO_TYPE_SHIFT) & 7) != type || ((mm & NZ_MEMO_DEG_GT1) && degree <= 1))) rok = false;
            else {
                rok = router_run(i, type, degree, pre);
                if(rok) H.nz[i] = (uint8_t)((mm & NZ_MEMBER) | NZ_MEMO_VALID | ((uni(HC.ro_type) & 7) << NZ_MEMO_TYPE_SHIFT) | (uni(HC.ro_degree) > 1 ? NZ_MEMO_DEG_GT1 : 0));
            }
            PROF_ADD(PF_G_BALANCE);
            if(rok && HC.ro_type == type && HC.ro_degree <= degree) {
                double rr = HC.ro_ratio;
                if(rr < 0.01) {
                    trace(OP_UNSPLIT_NOW, vlog(i), type, rr);
                    decompose_vertex_extend(i, HC.ro_npairs);
                    PROF_ADD(PF_G_DP);
                    act = 1;
                } else if(!(rr > ratio)) {
                    root = i; ratio = rr; best_np = HC.ro_npairs;
                    save_pairs(best_np);
                }
            }
        }
        wsync();
        act = wread(act, 0);
        if(HC.status) return true;
        if(act) { flag = true; memo_clear(); }
        cur = i + 1;
    }
    if(flag) return true;
    root = wread(root, 0);
    if(root < 0) return false;
    if(lane == 0) {
        PROF_DECL;
        restore_pairs(best_np);
        trace(OP_UNSPLIT_BEST, vlog(root), type, ratio);
        decompose_vertex_extend(root, best_np);
        PROF_ADD(PF_G_DP);
    }
    wsync();
    return true;
}

// ---------------------------------------------------------------- paths out
// scallop::collect_path (scallop.cc:2766-2834), scalar on lane 0
ALD_FN void collect_path(int e)
{
    e = uni(e);
    COLD;
    ALD_GLOBAL const KernelArgs *A = HC.args;
    int n = HC.V0 - 1;                           // v2v[sink]: the sink's original index
    int cnt = 0, mi = 0, nexw = 0, last_r = INT_MIN; bool empty = false;      // nexw: exon words of the transcript (build_transcript, essential.cc:719-748)
    for(int k = 0; k < NW; k++) { uint64_t mk = uni(C.ed[e].mask[k]); while(mk) { int b = ffs64(mk); mk &= mk - 1; int x = k * 64 + b; cnt++; const int l = uni(C.vx[x].lpos), r = uni(C.vx[x].rpos); mi += r - l; if(C.vx[x].vtype == K_EMPTY_VERTEX) empty = true;
        if(l < r) { if(nexw == 0 || last_r != l) nexw += 2; last_r = r; } } }
    if(ALD_UNLIKELY(C.ed[e].mei != mi || cnt == 0)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
    if(C.vx[0].vtype == K_EMPTY_VERTEX || uni(C.vx[n].vtype) == K_EMPTY_VERTEX) empty = true;
    if(!empty) {
        int nvp = cnt + 2;
        if(ALD_UNLIKELY(HC.n_paths >= C.po_cap)) { fail(ALD_ST_CAPACITY); return; }
        unsigned long long words = rec_words((unsigned)nvp, (unsigned)nexw);
        unsigned long long o = atomic_add_u64(A->out.pool_used, words);
        if(ALD_UNLIKELY(o + words > A->out.pool_cap)) { fail(ALD_ST_POOL_FULL); return; }     // not this graph's class that is too small: the host grows the pool
        C.po[HC.n_paths] = o;
        ALD_GLOBAL uint32_t *r = A->out.pool + o;
        int st = '.';
        if(C.ed[e].estrand == 1) st = '+';
        if(C.ed[e].estrand == 2) st = '-';
        if(st == '.') st = HC.gstrand;
        r[0] = (uint32_t)HC.g; r[1] = (uint32_t)HC.n_paths; r[2] = (uint32_t)nvp; r[3] = (uint32_t)mi; r[4] = (uint32_t)uni(C.ed[e].ecount); r[5] = (uint32_t)st | ((uint32_t)(A->attempt & 0xFF) << 8);
        ALD_GLOBAL double *d = (ALD_GLOBAL double*)(r + 6);
        d[0] = uni(H.ed[e].w); d[1] = uni(C.ed[e].eabd); d[2] = exp(C.ed[e].econf); d[3] = uni(C.ed[e].med);
        ALD_GLOBAL uint32_t *pv = r + REC_HDR_WORDS; int w = 0;
        pv[w++] = 0;
        for(int k = 0; k < NW; k++) { uint64_t mk = uni(C.ed[e].mask[k]); while(mk) { int b = ffs64(mk); mk &= mk - 1; pv[w++] = (uint32_t)(k * 64 + b); } }
        pv[w++] = (uint32_t)n;
        r[REC_NEXW] = (uint32_t)nexw; r[REC_NEXW + 1] = 0;
        { int q = 0; for(int k = 0; k < NW; k++) { uint64_t mk = uni(C.ed[e].mask[k]); while(mk) { int b = ffs64(mk); mk &= mk - 1; int x = k * 64 + b; const int l = uni(C.vx[x].lpos), rr = uni(C.vx[x].rpos);
            if(l >= rr) continue; if(q > 0 && (int)pv[w + q - 1] == l) pv[w + q - 1] = (uint32_t)rr; else { pv[w + q] = (uint32_t)l; pv[w + q + 1] = (uint32_t)rr; q += 2; } } } w += q; }
        if((REC_HDR_WORDS + nvp + nexw) & 1) pv[w] = 0;
        if(tracing()) { int save = HC.n_iters; trace(OP_COLLECT, (int)uni(H.eid[e]), nvp, uni(H.ed[e].w)); HC.n_iters = save; }
        HC.n_paths++;
    }
    H.hflag[e] = 0;
    kill_edge(e);
}
// Before the final phase walks out(0) / in(sink): link every live edge of those two lists (sorted insertion).
ALD_FN void materialize_special()
{
    if(HC.special_linked) return;
    HC.special_linked = 1;
    const int sinkp = HC.sinkp;
    H.vx[0].out_head = NIL; H.vx[sinkp].in_head = NIL; H.vx[0].out_deg = 0; H.vx[sinkp].in_deg = 0;
    for(int e = 0; e < HC.slot_hw; e++) {
        if(H.ed[e].lk.es == NIL) continue;
        if((int)uni(H.ed[e].lk.es) == 0) link_out(0, e);
        if((int)uni(H.ed[e].lk.et) == sinkp) link_in(sinkp, e);
    }
}
// (The lanes of this routine, of finish_graph and of the device pre-steps exchange values through the wave's SLAB -- global memory, work
// arrays whose lines a lane may hold in the vector L1 from an earlier use --: their hand-overs are wsync_mem(), which drains the stores.)
// scallop::collect_existing_st_paths (scallop.cc:2742-2752): ascending edge index == ascending creation id.
// out(0) / in(sink) are not linked at this point (see link_out): the source->sink edges are picked out of the slot array by the
// whole wave and ordered by id on lane 0; collect_path's remove_edge then only counts.  Called by ALL lanes.
ALD_FN void collect_existing_st_paths()
{
    COLD;
    PROF_DECL;
    const int lane = lane_id();
    const int sink = uni(HC.sinkp), hw = uni(HC.slot_hw);
    ALD_GLOBAL int32_t *lst = C.wi; int n = 0;           // the source -> sink edges (work array of the slab: up to MAXE entries), by every lane
    for(int base = 0; base < hw; base += ALD_WAVE) {
        int e = base + lane;
        bool p = e < hw && H.ed[e].lk.es == 0 && (int)H.ed[e].lk.et == sink;
        uint64_t m = wballot(p);
#ifdef ALD_EMU
        if(p) lst[n] = e;
#else
        if(p) lst[n + __builtin_popcountll(m & ((1ull << lane) - 1))] = e;
#endif
        n += __builtin_popcountll(m);
    }
    wsync_mem();
    PROF_ADD(PF_S5_DUP);
    if(n == 0) return;
    if(tracing() || 5 * n > Cold::w_cap || uni(HC.n_paths) + n > Cold::po_cap) {        // the op trace lists the paths in order: one at a time (which also reports a full offset table)
        if(lane == 0) {
            for(int i = 1; i < n; i++) { int x = lst[i]; uint32_t id = uni(H.eid[x]); int j = i - 1; while(j >= 0 && (uint32_t)uni(H.eid[lst[j]]) > id) { lst[j + 1] = lst[j]; j--; } lst[j + 1] = x; }
            for(int i = 0; i < n && !HC.status; i++) collect_path(lst[i]);
        }
        wsync_mem();
        return;
    }
    // One finished path per lane (scallop::collect_path, scallop.cc:2766-2834, for all of them at once): the vertex set and its length
    // check, the EMPTY_VERTEX filter, the place among the kept paths in creation-id order (= the path index), the record.
    ALD_GLOBAL const KernelArgs *A = HC.args;
    ALD_GLOBAL int32_t *ids = C.wi + n, *keep = C.wi + 2 * n;          // [n] creation id / 1 = becomes a path, 0 = filtered, -1 = inconsistent
    ALD_GLOBAL int32_t *nvs = C.wi + 3 * n, *nxs = C.wi + 4 * n;       // [n] vertices on the path / exon words of its transcript (build_transcript, essential.cc:719-748)
    const int nlast = HC.V0 - 1;
    const bool ends_empty = C.vx[0].vtype == K_EMPTY_VERTEX || C.vx[nlast].vtype == K_EMPTY_VERTEX;
    for(int j = lane; j < n; j += ALD_WAVE) {
        const int e = lst[j]; int cnt = 0, mi = 0, nexw = 0, last_r = INT_MIN; bool empty = ends_empty;
        for(int k = 0; k < NW; k++) { uint64_t mk = C.ed[e].mask[k]; while(mk) { int b = ffs64(mk); mk &= mk - 1; int x = k * 64 + b; cnt++; const int l = C.vx[x].lpos, rr = C.vx[x].rpos; mi += rr - l; if(C.vx[x].vtype == K_EMPTY_VERTEX) empty = true;
            if(l < rr) { if(nexw == 0 || last_r != l) nexw += 2; last_r = rr; } } }
        ids[j] = (int32_t)H.eid[e]; nvs[j] = cnt; nxs[j] = nexw;
        keep[j] = (C.ed[e].mei != mi || cnt == 0) ? -1 : (empty ? 0 : 1);
    }
    wsync_mem();
    PROF_ADD(PF_S5_BODY);
    bool bad = false, full = false; int kept_total = 0;
    for(int j = lane; j < n; j += ALD_WAVE) {
        const int e = lst[j]; const int32_t id = ids[j]; int rank = 0; bool later_bad = false;
        for(int k = 0; k < n; k++) { const int kk = keep[k]; if(kk < 0) later_bad = true; if(kk == 1 && ids[k] < id) rank++; }
        if(later_bad) bad = true;
        if(keep[j] != 1 || later_bad) continue;
        const int cnt = nvs[j], nexw = nxs[j];
        const int nvp = cnt + 2;
        const unsigned long long words = rec_words((unsigned)nvp, (unsigned)nexw);
        const unsigned long long o = atomic_add_u64(A->out.pool_used, words);
        if(ALD_UNLIKELY(o + words > A->out.pool_cap)) { full = true; continue; }
        C.po[HC.n_paths + rank] = o;
        ALD_GLOBAL uint32_t *r = A->out.pool + o;
        int st = '.';
        if(C.ed[e].estrand == 1) st = '+';
        if(C.ed[e].estrand == 2) st = '-';
        if(st == '.') st = HC.gstrand;
        r[0] = (uint32_t)HC.g; r[1] = (uint32_t)(HC.n_paths + rank); r[2] = (uint32_t)nvp; r[3] = (uint32_t)C.ed[e].mei; r[4] = (uint32_t)C.ed[e].ecount; r[5] = (uint32_t)st | ((uint32_t)(A->attempt & 0xFF) << 8);
        ALD_GLOBAL double *d = (ALD_GLOBAL double*)(r + 6);
        d[0] = H.ed[e].w; d[1] = C.ed[e].eabd; d[2] = exp(C.ed[e].econf); d[3] = C.ed[e].med;
        ALD_GLOBAL uint32_t *pv = r + REC_HDR_WORDS; int w = 0;
        pv[w++] = 0;
        for(int k = 0; k < NW; k++) { uint64_t mk = C.ed[e].mask[k]; while(mk) { int b = ffs64(mk); mk &= mk - 1; pv[w++] = (uint32_t)(k * 64 + b); } }
        pv[w++] = (uint32_t)nlast;
        r[REC_NEXW] = (uint32_t)nexw; r[REC_NEXW + 1] = 0;
        { int q = 0; for(int k = 0; k < NW; k++) { uint64_t mk = C.ed[e].mask[k]; while(mk) { int b = ffs64(mk); mk &= mk - 1; int x = k * 64 + b; const int l = C.vx[x].lpos, rr = C.vx[x].rpos;
            if(l >= rr) continue; if(q > 0 && (int)pv[w + q - 1] == l) pv[w + q - 1] = (uint32_t)rr; else { pv[w + q] = (uint32_t)l; pv[w + q + 1] = (uint32_t)rr; q += 2; } } } w += q; }
        if((REC_HDR_WORDS + nvp + nexw) & 1) pv[w] = 0;
    }
    const bool any_bad = wballot(bad) != 0, any_full = wballot(full) != 0;
    wsync_mem();
    PROF_ADD(PF_S5_RELINK);
    if(lane == 0) {
        for(int k = 0; k < n; k++) if(keep[k] == 1) kept_total++;
        if(any_bad) fail(ALD_ST_INVARIANT + ALD_INV_OTHER);                  // assert(mei[e] == mi) / an empty vertex list (scallop.cc:2783)
        else if(any_full) fail(ALD_ST_POOL_FULL);
        else HC.n_paths += kept_total;
        // remove_edge for all of them: out(source) / in(sink) are only counted at this point, the slots go back to the free list
        for(int k = 0; k < n; k++) { const int e = lst[k]; H.hflag[e] = 0; kill_edge_i(e); }
    }
    wsync_mem();
    PROF_ADD(PF_S6_WALK);
}
// splice_graph::compute_maximum_path_w (splice_graph.cc:819-885) + directed_graph::topological_sort (directed_graph.cc:420-451)
// path edges -> upper half of wi, length -> HC.tmp0
ALD_FN double compute_maximum_path()
{
    COLD;
    int n = HC.nv;
    ALD_GLOBAL int32_t *vd = C.wi, *q = C.wi + n, *back = C.wi + 2 * n; ALD_GLOBAL double *table = C.wd;
    ALD_GLOBAL int32_t *path = C.wi + Cold::w_cap / 2;
    int qt = 0;
    const int sinkp = HC.sinkp;
    // queue seeded in the reference's index order: physical order with the sink last
    for(int i = 0; i < n; i++) { int d = uni(H.vx[i].in_deg); vd[i] = d; if(d == 0 && i != sinkp) q[qt++] = i; table[i] = -1; back[i] = -1; }
    if(vd[sinkp] == 0) q[qt++] = sinkp;
    int k = 0;
    while(k < qt) { int x = q[k++]; for(int e = u_first_out(x); e >= 0; e = u_next_out(e)) { int t = uni(H.ed[e].lk.et); if(--vd[t] == 0) q[qt++] = t; } }
    HC.tmp0 = 0;
    if(ALD_UNLIKELY(qt != n)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return -1; }
    int ssi = -1, tti = -1;
    for(int i = 0; i < n; i++) { if(q[i] == 0) ssi = i; if(q[i] == sinkp) tti = i; }
    table[0] = DBL_MAX;
    for(int ii = ssi + 1; ii <= tti; ii++) {
        int i = q[ii];
        if(H.vx[i].in_deg + uni(H.vx[i].out_deg) == 0) continue;
        double max_abd = 0; int max_edge = -1;
        for(int e = u_first_in(i); e >= 0; e = u_next_in(e)) {
            int s = uni(H.ed[e].lk.es);
            double ts = table[s];
            if(ts <= -1) continue;
            double xw = uni(H.ed[e].w);
            double ww = xw < ts ? xw : ts;
            if(ww >= max_abd) { max_abd = ww; max_edge = e; }
        }
        if(max_edge < 0) continue;
        back[i] = max_edge; table[i] = max_abd;
    }
    int plen = 0;
    int x = sinkp;
    while(plen < n) { int e = back[x]; if(e < 0) break; path[plen++] = e; x = uni(H.ed[e].lk.es); }
    for(int i = 0; i < plen / 2; i++) { int t = path[i]; path[i] = path[plen - 1 - i]; path[plen - 1 - i] = t; }
    HC.tmp0 = plen;
    return table[sinkp];
}
// scallop::greedy_decompose (scallop.cc:2874-2897) + split_merge_path (scallop.cc:2230-2240)
ALD_FN void greedy_decompose()
{
    COLD;
    bool any = false;
    for(int i = 0; i < HC.nv && !any; i++) if(H.vx[i].out_deg) any = true;
    if(!any) return;
    materialize_special();                         // the DP and the path surgery walk out(0) / in(sink)
    PROF_DECL;
    // (a vertex without in- or out-edges is left alone by balance_vertex, scallop.cc:2488-2489: most of them by now -- no call for those)
    for(int rep = 0; rep < 2; rep++) for(int i = 1; i < HC.nv; i++) { if(i == HC.sinkp) continue; if(H.vx[i].in_deg == 0 || uni(H.vx[i].out_deg) == 0) continue; balance_vertex(i); if(HC.status) return; }
    PROF_ADD(PF_G_BALANCE);
    if(ALD_UNLIKELY(3 * HC.nv > C.w_cap / 2)) { fail(ALD_ST_CAPACITY); return; }
    ALD_GLOBAL int32_t *path = C.wi + Cold::w_cap / 2;
    const double min_cov = HC.p_min_cov;
    int guard = 4 * MAXE;
    while(guard-- > 0) {
        double w = compute_maximum_path();
        int plen = HC.tmp0;
        PROF_ADD(PF_G_DP);
        if(HC.status) return;
        if(w < 0) break;
        if(w <= min_cov) break;
        if(tracing()) { int save = HC.n_iters; trace(OP_GREEDY, plen, 0, w); HC.n_iters = save; }
        if(plen == 0) break;
        if(ALD_UNLIKELY(free_slots() < 2)) { fail(ALD_ST_CAPACITY); return; }
        int ee = split_edge(path[0], w);
        for(int i = 1; i < plen && ee >= 0 && !HC.status; i++) {
            if(ALD_UNLIKELY(free_slots() < 2)) { fail(ALD_ST_CAPACITY); return; }
            ee = merge_adjacent_edges(ee, path[i], w);     // split(path[i]) + merge with the (already equal) running edge
        }
        if(HC.status) return;
        if(ALD_UNLIKELY(ee < 0)) { fail(ALD_ST_INVARIANT + ALD_INV_OTHER); return; }
        PROF_ADD(PF_G_SPLITMERGE);
        collect_path(ee);
        PROF_ADD(PF_G_COLLECT);
        if(HC.status) return;
    }
}


// hyper_set::build_edges (hyper_set.cc:323-354) on lane 0: keep lists with count >= 2, >= 2 edges, every consecutive pair an edge;
// directed_graph::edge(s,t) returns the NEWEST parallel edge (directed_graph.cc:60-76).  po / pv / pc: NP vertex lists (offsets,
// vertices, counts) -- the staged phasing lists of the wire buffer, or the lists a raw graph's phases were turned into.
ALD_FN void build_phasing_lists(ALD_GLOBAL const int32_t *vo, int V, int NP, ALD_GLOBAL const int32_t *po, ALD_GLOBAL const int32_t *pvx, ALD_GLOBAL const int32_t *pcn)
{
    COLD;
    int nl = 0; uint32_t used = 0;
    for(int p = 0; p < NP && HC.status == 0; p++) {
        int c = pcn[p]; int a = po[p], b = po[p + 1]; int len = b - a;
        if(c <= 1 || len <= 1) continue;
        uint32_t capk = (uint32_t)(len - 1) * 2u + 4u;
        if(nl >= C.hl_maxlists || used + capk > C.hl_cap) { HC.status = ALD_ST_CAPACITY; break; }
        bool ok = true;
        for(int k = 0; k + 1 < len && ok; k++) {
            int s = pvx[a + k], t = pvx[a + k + 1];
            if(!(s < t) || s < 0 || t >= V) { HC.status = ALD_ST_INVARIANT + ALD_INV_OTHER; ok = false; break; }
            int best = -1;
            if(s == 0) { for(int e = vo[0]; e < vo[1]; e++) { if(H.ed[e].lk.es == NIL) continue; int tt = H.ed[e].lk.et; if(tt == t) best = e; else if(tt > t) break; } }      // row 0 is not linked: scan the CSR row
            else for(int e = first_out(s); e >= 0; e = next_out(e)) { int tt = H.ed[e].lk.et; if(tt == t) best = e; else if(tt > t) break; }
            if(best < 0) ok = false; else C.hl[used + k] = best;
        }
        if(!ok || len - 1 < 2) continue;
        C.hl_off[nl] = (int32_t)used; C.hl_len[nl] = len - 1; C.hl_capk[nl] = (int32_t)capk; C.hl_cnt[nl] = c; used += capk; nl++;
    }
    HC.hl_used = used; HC.hl_n = nl;
}

#ifdef ALD_RAW_VARIANT
// ---------------------------------------------------------------- raw graphs: what assembler::assemble(gx, px, sid)
// does to a graph and its phase set BEFORE it builds the scallop object (meta/assembler.cc:1075-1086), in the wave that just loaded it:
//   gx.extend_strands()                            rnacore/splice_graph.cc:1338-1373      one candidate edge per lane
//   group_start_boundaries / group_end_boundaries  rnacore/graph_reviser.cc:916-1066      a left-to-right fold over the source's /
//                                                                                          sink's few edges: lane 0
//   px.project_boundaries(smap, tmap)              rnacore/phase_set.cc:50-67             one phase per lane
//   hyper_set hx(gx, px) + hx.filter_nodes(gx)     scallop/hyper_set.cc:17-29, 356-371    one phase per lane (coordinate -> vertex by
//        (build_path_from_exon_coordinates, check_valid_path: essential.cc:321-366, 448-459)      binary search), equal lists fold
// The graph is still exactly the input at this point, so adjacency is read from the wire's CSR rows (sorted by (target, creation))
// and in-CSR; an edge the grouping removes is flagged first and leaves the lists at the end, the creation ids are then compacted as
// the host path's re-staging does.  Work arrays of the slab: [0, 2E) of wi for the compaction, smap / tmap and the phase table behind
// them, the phases' vertex lists in wd.  Called by ALL lanes; false (status set) where the reference would have asserted.
enum { RAW_DEAD = 0x80, RAW_SMAP = 2 * MAXE, RAW_TMAP = 2 * MAXE + 2 * MAXV, RAW_TV = 6 * MAXE, RAW_TV_CAP = (2 * MAXE - 4) / 3, RAW_VTX_CAP = 2 * Cold::w_cap };
static_assert(4 * MAXV <= 4 * MAXE, "boundary maps do not fit their slab region");
ALD_INL int raw_edge(ALD_GLOBAL const int32_t *vo, int s, int t)      // directed_graph::edge(s, t) on the input rows: the newest live parallel edge, or -1
{
    int best = -1;
    for(int k = vo[s]; k < vo[s + 1]; k++) { if(H.hflag[k] & RAW_DEAD) continue; const int tt = H.ed[k].lk.et; if(tt == t) best = k; else if(tt > t) break; }
    return best;
}
ALD_INL bool raw_continuous(const Cold &C, ALD_GLOBAL const int32_t *vo, int x, int y)      // check_continuous_vertices (essential.cc:436-446)
{
    if(x >= y) return true;
    for(int i = x; i < y; i++) { if(raw_edge(vo, i, i + 1) < 0) return false; if(C.vx[i].rpos != C.vx[i + 1].lpos) return false; }
    return true;
}
ALD_FN bool pre_assemble_device(int dist, int V, int E, ALD_GLOBAL const int32_t *vo, ALD_GLOBAL const int32_t *io, ALD_GLOBAL const int32_t *ie,
                                int NPH, ALD_GLOBAL const int32_t *pho, ALD_GLOBAL const int32_t *phc, ALD_GLOBAL const int32_t *phn)
{
    COLD;
    const int lane = lane_id();
    const int n = V - 1;
    // ---- splice_graph::extend_strands: a junction s -> s+2 that jumps exactly over vertex s+1 and outweighs it lends its strand to the
    // edges s -> s+1 and s+1 -> s+2 (the newest parallel ones) if they have none; junctions are visited in creation order, so an edge
    // takes the strand of the OLDEST qualifying junction that has one.  Here every edge a -> a+1 looks for its junctions itself (they
    // are never targets themselves, nothing a lane reads is written by another).
    if(uni(HC.any_strand)) {
        for(int k = lane; k < E; k += ALD_WAVE) {
            const int a = H.ed[k].lk.es, t = H.ed[k].lk.et;
            if(t != a + 1 || C.ed[k].estrand != 0) continue;
            if(k + 1 < vo[a + 1] && (int)H.ed[k + 1].lk.et == t) continue;               // not the newest a -> a+1
            uint32_t best_id = 0xFFFFFFFFu; int best_st = 0;
            for(int leg = 0; leg < 2; leg++) {
                const int s = leg == 0 ? a : a - 1;                                       // the junction s -> s+2; this edge is its first / second leg
                if(s < 0 || s + 2 >= V) continue;
                const int32_t p1 = C.vx[s].rpos, p2 = C.vx[s + 2].lpos;
                if(p1 >= p2 || C.vx[s + 1].lpos != p1 || C.vx[s + 1].rpos != p2) continue;
                const double vmid = C.vx[s + 1].vw;
                for(int j = vo[s]; j < vo[s + 1]; j++) {
                    if((int)H.ed[j].lk.et != s + 2) continue;
                    if(H.ed[j].w <= vmid) continue;
                    const int st = C.ed[j].estrand; const uint32_t id = H.eid[j];
                    if(st != 0 && id < best_id) { best_id = id; best_st = st; }
                }
            }
            if(best_st) C.ed[k].estrand = (uint8_t)best_st;
        }
        wsync_mem();
    }
    // ---- group_start_boundaries / group_end_boundaries (lane 0)
    if(lane == 0) {
        int ns = 0, nt = 0, bad = 0, removed = 0;
        ALD_GLOBAL int32_t *smap = C.wi + RAW_SMAP, *tmap = C.wi + RAW_TMAP;
        {   // start boundaries that reach the same run of touching vertices within `dist` fold into the leftmost one
            const int r0 = vo[0], r1 = vo[1];                                           // the source's row: targets ascending (parallel ones were refused at staging)
            if(r1 - r0 > 1) {
                int v0 = H.ed[r0].lk.et; int32_t p1 = C.vx[v0].lpos, p2 = p1; int k1 = v0, k2 = v0, pa = r0;
                for(int q = r0 + 1; q < r1 && !bad; q++) {
                    const int vi = H.ed[q].lk.et, pb = q; const int32_t p = C.vx[vi].lpos;
                    const double wb = H.ed[pb].w; const int cb = C.ed[pb].ecount;
                    bool b = raw_continuous(C, vo, k2, vi);
                    if(p < p2) { bad = 1; break; }                                        // assert(p >= p2)
                    if(p - p2 > dist) b = false;
                    if(!b) { p1 = p; p2 = p; k1 = vi; k2 = vi; pa = pb; continue; }
                    smap[2 * ns] = p; smap[2 * ns + 1] = p1; ns++;
                    for(int j = k1; j < vi; j++) {
                        const int pc = raw_edge(vo, j, j + 1);
                        if(pc < 0) { bad = 1; break; }                                    // assert(pc.second == true)
                        C.vx[j].vw = C.vx[j].vw + wb; C.ed[pc].ecount += cb; H.ed[pc].w = H.ed[pc].w + wb;
                    }
                    if(bad) break;
                    H.ed[pa].w += wb; C.ed[pa].ecount += cb;
                    H.hflag[pb] |= RAW_DEAD; removed++;
                    k2 = vi; p2 = p;
                }
            }
        }
        if(!bad) {   // the mirror image from the right -- with the reference's own asymmetries (vertex takes edge weight + wb; no count moves)
            const int i0 = io[n], i1 = io[n + 1];                                       // in-edges of the sink: sources ascending
            int q = i1 - 1; while(q >= i0 && (H.hflag[ie[q]] & RAW_DEAD)) q--;
            if(q >= i0) {
                int pa = ie[q]; const int v0 = H.ed[pa].lk.es; int32_t p1 = C.vx[v0].rpos, p2 = p1; int k1 = v0, k2 = v0; int live = 0;
                for(int z = i0; z < i1; z++) if(!(H.hflag[ie[z]] & RAW_DEAD)) live++;
                if(live > 1) for(q--; q >= i0 && !bad; q--) {
                    const int pb = ie[q]; if(H.hflag[pb] & RAW_DEAD) continue;
                    const int vi = H.ed[pb].lk.es; const int32_t p = C.vx[vi].rpos; const double wb = H.ed[pb].w;
                    bool b = raw_continuous(C, vo, vi, k2);
                    if(p > p2) { bad = 1; break; }                                        // assert(p <= p2)
                    if(p2 - p > dist) b = false;
                    if(!b) { p1 = p; p2 = p; k1 = vi; k2 = vi; pa = pb; continue; }
                    tmap[2 * nt] = p; tmap[2 * nt + 1] = p1; nt++;
                    for(int j = vi; j < k1; j++) {
                        const int pc = raw_edge(vo, j, j + 1);
                        if(pc < 0) { bad = 1; break; }
                        const double wc = H.ed[pc].w; H.ed[pc].w = wc + wb; C.vx[j + 1].vw = wc + wb;
                    }
                    if(bad) break;
                    H.ed[pa].w += wb;
                    H.hflag[pb] |= RAW_DEAD; removed++;
                    k2 = vi; p2 = p;
                }
            }
        }
        if(bad) fail(ALD_ST_INVARIANT + ALD_INV_OTHER);
        HC.scr_i[0] = ns; HC.scr_i[1] = nt; HC.scr_i[2] = removed;
    }
    wsync_mem();
    if(uni(HC.status)) return false;
    const int ns = uni(HC.scr_i[0]), nt = uni(HC.scr_i[1]), removed = uni(HC.scr_i[2]);
    // ---- the folded boundary edges leave the graph; the creation ids of the others close ranks (the order is what matters: every id
    // handed out later is larger)
    if(removed) {
        ALD_GLOBAL int32_t *flag = C.wi, *nrank = C.wi + E;
        for(int k = lane; k < E; k += ALD_WAVE) flag[H.eid[k]] = (H.hflag[k] & RAW_DEAD) ? 0 : 1;
        wsync_mem();
        int base = 0;
        for(int r0 = 0; r0 < E; r0 += ALD_WAVE) {
            const int r = r0 + lane; const bool f = r < E && flag[r] != 0;
            const uint64_t m = wballot(f);
#ifdef ALD_EMU
            if(r < E) nrank[r] = base;
#else
            if(r < E) nrank[r] = base + __builtin_popcountll(m & ((1ull << lane) - 1));
#endif
            base += __builtin_popcountll(m);
        }
        wsync_mem();
        for(int k = lane; k < E; k += ALD_WAVE) if(!(H.hflag[k] & RAW_DEAD)) H.eid[k] = (EID)nrank[H.eid[k]];
        wsync_mem();
        if(lane == 0) {
            for(int k = 0; k < E; k++) if(H.hflag[k] & RAW_DEAD) { H.hflag[k] = 0; kill_edge_i(k); }
            HC.next_id = base;
            for(int i = 1; i < n; i++) if(H.vx[i].in_deg == 0 && H.vx[i].out_deg == 0) H.nz[i] = 0;      // (nonzeroset is taken after the pre-steps: scallop.cc:1664-1673)
        }
        wsync_mem();
    }
    // ---- phases: exon coordinates -> vertex lists (one phase per lane), equal lists fold their counts
    ALD_GLOBAL int32_t *tv = C.wi + RAW_TV;                                   // [0] = number of lists, [1 ..] offsets (NPH + 1), then counts (NPH)
    ALD_GLOBAL int32_t *tv_off = tv + 1, *tv_cnt = tv + 1 + RAW_TV_CAP + 1;
    ALD_GLOBAL int32_t *vtx = (ALD_GLOBAL int32_t*)C.wd;
    if(NPH > RAW_TV_CAP) { if(lane == 0) fail(ALD_ST_CAPACITY); wsync_mem(); return false; }
    if(NPH == 0) { if(lane == 0) { tv[0] = 0; tv_off[0] = 0; } wsync_mem(); return true; }
    // are the vertices' left / right ends in ascending order (they are, for a splice graph)?  Then a coordinate is found by bisection
    bool unsorted = false;
    for(int i = 1 + lane; i < n; i += ALD_WAVE) { if(C.vx[i].lpos > C.vx[i + 1].lpos) unsorted = true; if(i + 1 < n && C.vx[i].rpos > C.vx[i + 1].rpos) unsorted = true; }
    if(lane == 0 && n >= 2 && C.vx[0].rpos > C.vx[1].rpos) unsorted = true;
    const bool sorted = wballot(unsorted) == 0;
    ALD_GLOBAL const int32_t *smap = C.wi + RAW_SMAP, *tmap = C.wi + RAW_TMAP;
    // build_vertex_index (splice_graph.cc:1087-1099): lindex over vertices 1..n, rindex over 0..n-1, the FIRST vertex with a coordinate wins
    auto lfind = [&](int32_t p) -> int {
        if(sorted) { int lo = 1, hi = n + 1; while(lo < hi) { const int mid = (lo + hi) >> 1; if(C.vx[mid].lpos < p) lo = mid + 1; else hi = mid; } return (lo <= n && C.vx[lo].lpos == p) ? lo : -1; }
        for(int i = 1; i <= n; i++) if(C.vx[i].lpos == p) return i;
        return -1; };
    auto rfind = [&](int32_t q) -> int {
        if(sorted) { int lo = 0, hi = n; while(lo < hi) { const int mid = (lo + hi) >> 1; if(C.vx[mid].rpos < q) lo = mid + 1; else hi = mid; } return (lo < n && C.vx[lo].rpos == q) ? lo : -1; }
        for(int i = 0; i < n; i++) if(C.vx[i].rpos == q) return i;
        return -1; };
    // pass 1: is the phase a path of the graph, and how long is its vertex list?  (-1: dropped)
    int total = 0; bool assert_hit = false;
    for(int p0 = 0; p0 < NPH; p0 += ALD_WAVE) {
        const int p = p0 + lane; int len = -1;
        if(p < NPH) {
            const int a0 = pho[p], m = pho[p + 1] - a0, ne = m / 2; len = 0; int prev = -1; bool backwards = false;
            for(int k = 0; k < ne && len >= 0; k++) {
                int32_t x = phc[a0 + 2 * k], y = phc[a0 + 2 * k + 1];
                if(k == 0) for(int z = 0; z < ns; z++) if(smap[2 * z] == x) { x = smap[2 * z + 1]; break; }            // project_boundaries: first / last coordinate
                if(k == ne - 1) for(int z = 0; z < nt; z++) if(tmap[2 * z] == y) { y = tmap[2 * z + 1]; break; }
                if(x < 0 || y < 0 || x >= y) { len = -1; break; }
                const int a = lfind(x), b = rfind(y);
                if(a < 0 || b < 0 || a > b || !raw_continuous(C, vo, a, b)) { len = -1; break; }
                if(a <= prev) backwards = true;
                prev = b; len += b - a + 1;
            }
            if(len >= 0 && backwards) assert_hit = true;                                      // a valid list that does not ascend: assert(vv[i] < vv[i + 1]) (essential.cc:364)
        }
        // offsets of the chunk's lists: a serial prefix over the wave's lengths through the LDS scratch
        if(p < NPH) HC.scr_i[8 + lane] = len;
        wsync_mem();
        if(lane == 0) { const int cnt = (NPH - p0) < ALD_WAVE ? (NPH - p0) : ALD_WAVE; int run = total; for(int l = 0; l < cnt; l++) { tv_off[p0 + l] = run; const int x = HC.scr_i[8 + l]; tv_cnt[p0 + l] = x < 0 ? -1 : 0; if(x > 0) run += x; } HC.scr_i[7] = run; }
        wsync_mem();
        total = uni(HC.scr_i[7]);
    }
    if(wballot(assert_hit)) { if(lane == 0) fail(ALD_ST_INVARIANT + ALD_INV_OTHER); wsync_mem(); return false; }
    if(total > RAW_VTX_CAP) { if(lane == 0) fail(ALD_ST_CAPACITY); wsync_mem(); return false; }
    if(lane == 0) tv_off[NPH] = total;
    wsync_mem();
    // pass 2: the vertex lists themselves
    for(int p = lane; p < NPH; p += ALD_WAVE) {
        if(tv_cnt[p] < 0) continue;
        const int a0 = pho[p], m = pho[p + 1] - a0, ne = m / 2; int w = tv_off[p];
        for(int k = 0; k < ne; k++) {
            int32_t x = phc[a0 + 2 * k], y = phc[a0 + 2 * k + 1];
            if(k == 0) for(int z = 0; z < ns; z++) if(smap[2 * z] == x) { x = smap[2 * z + 1]; break; }
            if(k == ne - 1) for(int z = 0; z < nt; z++) if(tmap[2 * z] == y) { y = tmap[2 * z + 1]; break; }
            const int a = lfind(x), b = rfind(y);
            for(int j = a; j <= b; j++) vtx[w++] = j;
        }
    }
    wsync_mem();
    // equal lists are ONE node of hyper_set::nodes (a std::map keyed by the list): the first of them takes the counts of all
    for(int p = lane; p < NPH; p += ALD_WAVE) {
        if(tv_cnt[p] < 0) continue;
        const int o = tv_off[p], len = tv_off[p + 1] - o; int rep = p;
        for(int q = 0; q < p && rep == p; q++) {
            if(tv_cnt[q] < 0 || tv_off[q + 1] - tv_off[q] != len) continue;
            bool eq = true; for(int k = 0; k < len && eq; k++) eq = vtx[tv_off[q] + k] == vtx[o + k];
            if(eq) rep = q;
        }
        if(rep != p) tv_cnt[p] = -2 - rep;                                                    // folded into list `rep`
    }
    wsync_mem();
    for(int p = lane; p < NPH; p += ALD_WAVE) {
        if(tv_cnt[p] != 0) continue;                                                          // a representative: its own count + the counts of its copies
        int c = phn[p];
        for(int q = p + 1; q < NPH; q++) if(tv_cnt[q] == -2 - p) c += phn[q];
        tv_cnt[p] = c;
    }
    wsync_mem();
    for(int p = lane; p < NPH; p += ALD_WAVE) if(tv_cnt[p] < 0) tv_cnt[p] = 0;                // dropped or folded: build_edges skips a count <= 1
    if(lane == 0) tv[0] = NPH;
    wsync_mem();
    return true;
}
#else
enum { RAW_TV = 0, RAW_TV_CAP = 0 };          // (named by load_graph's raw branch, which the plain build never takes)
#endif

// ---------------------------------------------------------------- load: packed wire arrays -> working state (wave-parallel)
ALD_FN bool load_graph()
{
    COLD;
    const int lane = lane_id();
    ALD_GLOBAL const KernelArgs *A = HC.args;
    const int g = HC.g;
    int V = A->in.g_nv[g], E = A->in.g_ne[g], NP = A->in.g_np[g];
    int64_t ov = A->in.off_v[g], ovo = ov + g, oe = A->in.off_e[g], oeo = oe + g, os = A->in.off_s[g], op = A->in.off_p[g], opo = op + g, opv = A->in.off_pv[g];
    if(lane == 0) {
        HC.V0 = V; HC.gstrand = (int)(unsigned char)A->in.graph_strand[g];
        HC.sinkp = V - 1; HC.special_linked = 0; HC.maybe_broken = 1; HC.maybe_triv = 1; ev_mark_all();
        HC.nv = V; HC.next_id = E; HC.slot_hw = E; HC.free_head = -1; HC.free_cnt = 0; HC.status = 0; HC.any_strand = 0; HC.hs_dirty = 1;
        HC.n_paths = 0; HC.n_iters = 0; HC.n_trace = 0; HC.sp_used = 0; HC.hl_used = 0; HC.hl_n = 0;
    }
    wsync();
    if(V + 1 > MAXV || E > MAXE || V > NW * 64 || V < 2) { if(lane == 0) HC.status = ALD_ST_CAPACITY; wsync(); return false; }
    int64_t ns = A->in.edge_sample_offset[oeo + E];
    if(ns > (int64_t)C.sp_cap) { if(lane == 0) HC.status = ALD_ST_CAPACITY; wsync(); return false; }
    ALD_GLOBAL const int32_t *vo = A->in.vertex_offset + ovo, *io = A->in.in_offset + ovo, *ie = A->in.in_edge + oe;
    for(int i = lane; i < V; i += ALD_WAVE) {
        int o0 = vo[i], o1 = vo[i + 1], i0 = io[i], i1 = io[i + 1];
        // out(0) and in(sink) are counted, not linked (see link_in / link_out)
        H.vx[i].out_head = (o1 > o0 && i != 0) ? (IDX)o0 : NIL; H.vx[i].out_deg = (IDX)(o1 - o0);
        H.vx[i].in_head = (i1 > i0 && i != V - 1) ? (IDX)ie[i0] : NIL; H.vx[i].in_deg = (IDX)(i1 - i0);
        H.nz[i] = (i >= 1 && i < V - 1 && (o1 - o0) + (i1 - i0) > 0) ? 1 : 0;
        for(int k = o0; k < o1; k++) { H.ed[k].lk.es = (IDX)i; H.ed[k].lk.onx = (k + 1 < o1) ? (IDX)(k + 1) : NIL; }
        for(int k = i0; k < i1; k++) { H.ed[ie[k]].lk.inx = (k + 1 < i1) ? (IDX)ie[k + 1] : NIL; }
        C.vx[i].vw = A->in.vertex_weight[ov + i]; C.vx[i].lpos = A->in.vertex_lpos[ov + i]; C.vx[i].rpos = A->in.vertex_rpos[ov + i];
        C.vx[i].vtype = A->in.vertex_type[ov + i]; C.vx[i].v2v = i;
    }
    bool strand = false, listed = false;        // listed: some edge has two or more supporting samples (its list lives in the pool)
    ALD_GLOBAL const int32_t *so = A->in.edge_sample_offset + oeo;
    ALD_GLOBAL const int32_t *rank = A->in.edge_rank;      // scallop::scallop -> get_edge_indices (scallop.cc:24, graph_base.cc:139-153): e2i of the input edges
    for(int k = lane; k < E; k += ALD_WAVE) {
        H.ed[k].lk.et = (IDX)A->in.edge_target[oe + k]; H.ed[k].w = A->in.edge_weight[oe + k]; H.eid[k] = rank ? (EID)rank[oe + k] : (EID)k; H.hflag[k] = 0;
        uint8_t st = A->in.edge_strand[oe + k]; C.ed[k].estrand = st; if(st) strand = true;
        C.ed[k].med = 0; C.ed[k].mei = 0; C.ed[k].econf = 0; C.ed[k].eabd = A->in.edge_abd[oe + k];
        C.ed[k].sp_off = (uint32_t)so[k]; C.ed[k].sp_len = (uint32_t)(so[k + 1] - so[k]); C.ed[k].ecount = A->in.edge_count[oe + k];
        if(so[k + 1] - so[k] >= 2) listed = true;
        if(so[k + 1] > so[k]) { C.ed[k].s0id = A->in.sample_id[os + so[k]]; C.ed[k].s0abd = A->in.sample_abd[os + so[k]]; } else { C.ed[k].s0id = 0; C.ed[k].s0abd = 0; }
        for(int q = 0; q < NW; q++) C.ed[k].mask[q] = 0;
    }
    // a list of ONE sample is carried inline (s0id / s0abd) and never read from the pool: a batch of single-sample graphs copies nothing
    if(wballot(listed)) for(int64_t k = lane; k < ns; k += ALD_WAVE) { C.sp_id[k] = A->in.sample_id[os + k]; C.sp_abd[k] = A->in.sample_abd[os + k]; }
    uint64_t sb = wballot(strand);
    if(sb && lane == 0) HC.any_strand = 1;
    if(lane == 0) HC.sp_used = (uint32_t)ns;
    wsync();
    // a raw graph: the pre-steps of assembler::assemble(gx, px, sid) first
    const int rawdist = A->in.g_rawdist ? uni(A->in.g_rawdist[g]) : -1;
    if(rawdist >= 0) {
#ifdef ALD_RAW_VARIANT
        const int64_t orp = A->in.off_rp[g], orc = A->in.off_rc[g];
        if(!uni(pre_assemble_device(rawdist, V, E, vo, io, ie, (int)(A->in.off_rp[g + 1] - orp), A->in.rphase_offset + orp + g, A->in.rphase_coord + orc, A->in.rphase_count + orp))) { wsync(); return false; }
#else
        if(lane == 0) HC.status = ALD_ST_INVARIANT + ALD_INV_OTHER;          // cannot happen: the host sends raw graphs to the raw build of the class
        wsync(); return false;
#endif
    }
    if(lane == 0) {
        if(rawdist < 0) build_phasing_lists(vo, V, NP, A->in.phasing_offset + opo, A->in.phasing_vertex + opv, A->in.phasing_count + op);
        else build_phasing_lists(vo, V, uni(C.wi[RAW_TV + 0]), C.wi + RAW_TV + 1, (ALD_GLOBAL const int32_t*)C.wd, C.wi + RAW_TV + 1 + RAW_TV_CAP + 1);
    }
    wsync();
    return HC.status == 0;
}

ALD_FN void finish_graph()
{
    // the graph's part of the result index: a graph that ended well reserves n_paths entries (one atomic) and the wave copies the pool
    // offsets of its records there, in path order; a graph that did not end well publishes nothing, so its records are unreachable
    {
        COLD;
        ALD_GLOBAL const KernelArgs *A = HC.args; const int g = uni(HC.g);
        const int st = uni(HC.status), np = (st == 0 || st == ALD_ST_SKIPPED_LARGE) ? uni(HC.n_paths) : 0;
        if(lane_id() == 0) {
            long long base = -1;
            if(np > 0) { const unsigned long long b0 = atomic_add_u64(A->out.index_used, (unsigned long long)np); if(ALD_UNLIKELY(b0 + (unsigned long long)np > A->out.index_cap)) HC.status = ALD_ST_POOL_FULL; else base = (long long)b0; }     // (the host grows pool and index together)
            A->out.graph_first[g] = base;
            HC.scr_i[0] = (int32_t)(uint32_t)((unsigned long long)base & 0xFFFFFFFFull); HC.scr_i[1] = (int32_t)(uint32_t)((unsigned long long)base >> 32);
        }
        wsync_mem();
        const long long base = (long long)(((unsigned long long)(uint32_t)uni(HC.scr_i[1]) << 32) | (unsigned long long)(uint32_t)uni(HC.scr_i[0]));
        if(base >= 0) for(int i = lane_id(); i < np; i += ALD_WAVE) A->out.index[base + i] = C.po[i];
        wsync_mem();
    }
    if(lane_id() == 0) {
        ALD_GLOBAL const KernelArgs *A = HC.args; const int g = HC.g;
        A->out.status[g] = HC.status; A->out.n_paths[g] = (HC.status == 0 || HC.status == ALD_ST_SKIPPED_LARGE) ? HC.n_paths : 0; A->out.n_iters[g] = HC.n_iters;
#ifdef ALD_PROF
        if(HC.p_trace_cap > 0) for(int k = 0; k < PF_COUNT; k++) { int q = HC.n_trace++; if(q < A->out.trace_cap) { int64_t o = (int64_t)g * A->out.trace_cap + q; A->out.trace_codes[3 * o] = 100 + k; A->out.trace_codes[3 * o + 1] = 0; A->out.trace_codes[3 * o + 2] = 0; A->out.trace_vals[o] = (double)HC.prof[k]; } }
#endif
        if(A->out.trace_cap > 0) A->out.trace_n[g] = HC.n_trace;
    }
    wsync_mem();
}

// ---------------------------------------------------------------- scallop::assemble (scallop.cc:38-188)
// Plain build: inlined into the kernel root.  Raw build: a real call from the root (one per graph) -- with the device pre-steps, a large
// callee reached through load_graph, in the same function as the cascade, the register allocator lost the caller-saved registers the
// cascade keeps its long-lived values in (43 VGPR spills spread over every sweep: 42.7 -> 56.2 ms for the bench batch); as a callee the
// cascade is allocated on its own (42.9 ms) at the price of saving 32 callee-saved VGPRs per graph (8 KB of scratch traffic each way).
#ifndef ALD_RAW_VARIANT
ALD_INL void run_graph()
#else
ALD_FN void run_graph()
#endif
{
    PROF_DECL;
#ifdef ALD_PROF
    if(lane_id() == 0) for(int k = 0; k < 32; k++) HC.prof[k] = 0;
    wsync();
#endif
    if(!uni(load_graph())) { finish_graph(); return; }
    PROF_ADD(PF_LOAD);
    bool skipped = false;
#ifndef ALD_SGPR_DIET
    const double r_triv = uni(HC.p_ratio[7]), r_small = uni(HC.p_ratio[0]), r_single = uni(HC.p_ratio[5]), r_pure = uni(HC.p_ratio[4]);
    const int max_exons = uni(HC.p_max_exons);
#else
    // experiment (profiles/r04/e_*): the cascade's long-lived wave-uniform values -- four ratios, the vertex limit: nine SGPRs for a graph's
    // whole life, part of what the kernel root spills to VGPR lanes -- re-read from the LDS context where they are used
    #define r_triv uni(HC.p_ratio[7])
    #define r_small uni(HC.p_ratio[0])
    #define r_single uni(HC.p_ratio[5])
    #define r_pure uni(HC.p_ratio[4])
    #define max_exons uni(HC.p_max_exons)
#endif
    int guard = 64 * MAXE;                     // every successful rule consumes an edge or a vertex; far above any real count
    while(guard-- > 0) {
        bool brk = false;
        { const int nvq = HC.nv, stq = HC.status, mbq = HC.maybe_broken;      // one round of LDS reads for the three tests
          if(uni(nvq) > max_exons) { skipped = true; break; }
          if(uni(stq)) break;
          PROF_RESET();
          brk = uni(mbq) != 0 && uni(resolve_broken_vertex()); }
        PROF_ADD(PF_BROKEN);
        if(brk) continue;
        // the rest of the cascade as a stage loop, so that every rule is instantiated ONCE (the trivial-vertex sweep serves three stages:
        // resolve_trivial_vertex_fast -- a no-op for jump_ratio <= 1, r >= 1 always -- then type 1, and type 2 at the very end)
        bool fired = false;
        for(int stage = (r_triv > 1.0) ? 0 : 1; stage <= 4 && !fired; stage++) {
            if(stage == 0 || stage == 1 || stage == 4) fired = uni(sweep_trivial(stage == 0 ? 0 : 1, stage == 4 ? 2 : 1, r_triv));
            else if(stage == 2) fired = uni(sweep_smallest(r_small));
            else {
                PROF_RESET();
                memo_clear();                                          // anything may have changed since the last cascade
                for(int pass = 0; pass < 6 && !fired; pass++)
                    fired = uni(sweep_unsplittable((pass & 1) ? T_SPLITTABLE_PURE : T_UNSPLITTABLE_SINGLE, pass < 2 ? 1 : INT_MAX,
                                                   pass < 2 ? 0.01 : (pass == 2 ? r_single : (pass == 3 ? r_pure : DBL_MAX))));
                PROF_ADD(PF_UNSPLIT);
            }
        }
        if(fired) continue;
        break;
    }
    if(lane_id() == 0 && HC.status == 0 && guard <= 0) fail(ALD_ST_INVARIANT + ALD_INV_OTHER);
    wsync();
    if(uni(HC.status) == 0) {
        PROF_RESET();
        collect_existing_st_paths();
        PROF_ADD(PF_COLLECT0);
        if(lane_id() == 0) {
            if(HC.status == 0) greedy_decompose();
            if(HC.status == 0 && skipped) HC.status = ALD_ST_SKIPPED_LARGE;
        }
        PROF_ADD(PF_S7);
    }
    wsync();
    finish_graph();
#ifdef ALD_SGPR_DIET
    #undef r_triv
    #undef r_small
    #undef r_single
    #undef r_pure
    #undef max_exons
#endif
}

#ifdef ALD_ISA_PROBE
// diagnostic only (tools/isa_stats.py with ISA_EXTRA=-DALD_ISA_PROBE): hot inlined pieces as functions of their own, so that their instruction mix can be read
__attribute__((used)) ALD_FN void probe_kill_edge_wave(int e) { kill_edge_wave(e); }
__attribute__((used)) ALD_FN int probe_eval_smallest(int i, double *r) { double rr = 0; int e = eval_smallest(i, rr); *r = rr; return e; }
__attribute__((used)) ALD_FN void probe_wave_argmin(double *r, int *v) { double rr = *r; int vv = *v; wave_argmin(rr, vv); *r = rr; *v = vv; }
__attribute__((used)) ALD_FN bool probe_sweep_smallest(double r) { return sweep_smallest(r); }
#endif
// one wave's whole life: pull graphs of this size class from the shared counter until the class is drained
ALD_INL void wave_main(ALD_GLOBAL const KernelArgs *A, int block)
{
#ifdef ALD_HOT_IN_SLAB
    if(lane_id() == 0) g_Hp = (ALD_GLOBAL Hot*)(A->slabs + (uint64_t)block * A->slab_stride);
    wsync();
#endif
    if(lane_id() == 0) {
        HC.args = A;
#ifdef ALD_HOT_IN_SLAB
        HC.cold = A->slabs + (uint64_t)block * A->slab_stride + ((sizeof(Hot) + 255) / 256 * 256);
#else
        HC.cold = A->slabs + (uint64_t)block * A->slab_stride;
#endif
        for(int k = 0; k < 8; k++) HC.p_ratio[k] = A->prm.max_ratio[k];
        HC.p_min_w = A->prm.min_w; HC.p_min_cov = A->prm.min_cov; HC.p_max_exons = A->prm.max_num_exons; HC.p_trace_cap = A->out.trace_cap;
        HC.pw_lds = 0; HC.park_lds = 0;
    }
    wsync();
    while(true) {
#ifdef ALD_SGPR_DIET
        A = (ALD_GLOBAL const KernelArgs*)uni((unsigned long long)HC.args);       // (experiment: the argument block's address from LDS instead of two SGPRs held across the graph)
#endif
        if(lane_id() == 0) HC.s_next = atomic_add_i32(A->counter, 1);
        wsync();
        int k = HC.s_next;
        wsync();
        if(k >= A->n_work) break;
        if(lane_id() == 0) HC.g = A->work[k];
        wsync();
        run_graph();
    }
}

#undef H
#undef COLD
} // namespace ALD_CLASS_NS
