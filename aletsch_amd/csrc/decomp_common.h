// decomp_common.h -- types shared by the device engine (decomp_device.h), the kernels and the host ABI.
#pragma once
#ifndef ALD_FIRST_GLOBAL_CLASS
#define ALD_FIRST_GLOBAL_CLASS 10     /* classes from here on keep the Hot struct in the wave's HBM slab (overridable for experiments) */
#endif
#include <cstdlib>
#include <stdint.h>
#include <math.h>
#include <float.h>
#include <limits.h>
#include "../../include/aletsch_decomp.h"

#ifdef ALD_EMU
  // single-lane emulation of the engine, compiled ONLY by tests/kernel_emu (never part of the product library)
  #include <string.h>
  #define ALD_FN static
  #define ALD_INL static inline
  #define ALD_GLOBAL
  #define ALD_WAVE 1
  namespace ald {
  static inline int      lane_id() { return 0; }
  static inline uint64_t wballot(bool p) { return p ? 1ull : 0ull; }
  template<class T> static inline T wshfl(T v, int) { return v; }
  template<class T> static inline T wread(T v, int) { return v; }
  static inline void     wave_argmin(double &, int &) {}
  static inline void     wsync() {}
  static inline void     wsync_mem() {}
  static inline unsigned long long atomic_add_u64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
  static inline int      atomic_add_i32(int *p, int v) { int o = *p; *p += v; return o; }
  static inline int      ffs64(uint64_t m) { return __builtin_ffsll((long long)m) - 1; }
  template<class T> static inline T uni(T v) { return v; }
  }
#elif !defined(__HIP__)
  // plain host C++ translation unit (ald_abi.cpp, synth.cpp): only the shared structs are needed
  #define ALD_GLOBAL
#else
  #include <hip/hip_runtime.h>
  #define ALD_FN  static __device__ __attribute__((noinline))
  #define ALD_INL static __device__ __forceinline__
  #if defined(__HIP_DEVICE_COMPILE__)
    #define ALD_GLOBAL __attribute__((address_space(1)))   /* device pass: global_* instead of flat_* memory ops */
  #else
    #define ALD_GLOBAL                                     /* host pass of the same source: plain pointers, same layout */
  #endif
  #define ALD_WAVE 64
  namespace ald {
  __device__ __forceinline__ int      lane_id() { return (int)threadIdx.x; }
  __device__ __forceinline__ uint64_t wballot(bool p) { return __ballot(p); }
  template<class T> __device__ __forceinline__ T wshfl(T v, int src) { return __shfl(v, src, 64); }
  // value of lane `src`, src WAVE-UNIFORM: v_readlane_b32 (a few cycles) instead of the ds_bpermute_b32 a general shuffle takes
  __device__ __forceinline__ int      wread(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
  __device__ __forceinline__ double   wread(double v, int src)
  {
      const unsigned long long b = (unsigned long long)__double_as_longlong(v);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
      return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  }
  // unsigned 32-bit minimum over the wave, result wave-uniform: six v_min_u32 with a DPP source operand (quad, quad, half row, row, row
  // broadcast x 2 -- the CDNA reduction idiom), lane 63 then holds the minimum.  Written as ONE asm block: the compiler does not fold
  // a v_mov_b32_dpp into the v_min_u32 that consumes it (four instructions per step instead of one); the block carries the wait
  // states the hardware asks for itself -- two between a VALU write of a VGPR and a DPP read of it, one before the v_readlane -- and
  // opens with FIVE (s_nop 4): the compiler's hazard recognizer cannot see into the block, and a VALU write of EXEC (v_cmpx) scheduled
  // directly in front of it would need five wait states before the first DPP step.
  __device__ __forceinline__ unsigned wave_umin32(unsigned k)
  {
      asm volatile("s_nop 4\n\t"
                   "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                   "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                   "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                   "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                   "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                   "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
                   : "+v"(k));
      return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
  }
  // wave_argmin: over the lanes with vv >= 0, the smallest rr; among equal ones the LARGEST vv (the sequential sweeps keep the later
  // vertex on `!(ratio < r)`); every lane receives the winner, vv = -1 when no lane has a candidate.  The ratios are non-negative
  // doubles, whose order is the order of their bit patterns as unsigned integers: the minimum is a 64-bit unsigned min over the wave --
  // two 32-bit DPP reductions (wave_umin32), high word first --, the winner then comes from one ballot.  A butterfly of shuffles costs
  // eighteen ds_bpermute_b32 round trips for the same result.  Any candidate that is negative or NaN (never produced by the rules; bit
  // pattern above +inf) sends the wave through the butterfly.
  // PRECONDITION: every lane of the wave is active (EXEC all ones) -- a DPP step reads the registers of its source lanes whatever their
  // EXEC bit, and the final ballot / readlane assume 64 candidates.  Both call sites (scan_trivial, sweep_smallest) sit in code every
  // lane runs; the -DALD_WSYNC_BARRIER test build traps if that is ever not so.
  __device__ __forceinline__ void     wave_argmin(double &rr, int &vv)
  {
  #ifdef ALD_WSYNC_BARRIER
      if(__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
  #endif
      const unsigned long long bits = (unsigned long long)__double_as_longlong(rr);
      if(__builtin_expect(__ballot(vv >= 0 && bits > 0x7FF0000000000000ull) != 0, 0)) {
          const int lane = (int)threadIdx.x;
          for(int off = 32; off >= 1; off >>= 1) {
              const double r2 = __shfl(rr, lane ^ off, 64); const int v2 = __shfl(vv, lane ^ off, 64);
              if((v2 >= 0) && (vv < 0 || r2 < rr || (r2 == rr && v2 > vv))) { rr = r2; vv = v2; }
          }
          rr = __shfl(rr, 0, 64); vv = __shfl(vv, 0, 64);
          return;
      }
      unsigned long long k = vv >= 0 ? bits : ~0ull;
      const unsigned long long mine = k;
      // 64-bit minimum = minimum of the high words, then of the low words of the lanes that hold it
      const unsigned khi = (unsigned)(k >> 32), klo = (unsigned)k;
      const unsigned hi = wave_umin32(khi);
      const unsigned lo = wave_umin32(khi == hi ? klo : 0xFFFFFFFFu);
      const unsigned long long kmin = ((unsigned long long)hi << 32) | lo;
      unsigned long long tied = __ballot(vv >= 0 && mine == kmin);
      int best = -1;
      while(tied) { const int l = __ffsll(tied) - 1; tied &= tied - 1; const int v = __builtin_amdgcn_readlane(vv, l); best = v > best ? v : best; }
      rr = __longlong_as_double((long long)kmin); vv = best;
  }
  // workgroup == ONE wavefront: lanes of a wave execute in lock step and the hardware performs a wave's LDS instructions in issue order and its
  // vector-memory instructions in issue order, so handing data from one lane to another -- through LDS or through the wave's slab in global
  // memory -- needs no s_barrier and no wait on the memory counters, only that the compiler keeps the accesses on their side of the hand-over
  // (LLVM AMDGPU memory model: a fence at "wavefront" scope emits no instruction).  __syncthreads() here cost an s_waitcnt + s_barrier at every
  // one of the few thousand hand-overs a graph takes.  tools/microbench/lane_raw.hip checks the global-memory half of that sentence on gfx950:
  // a 16- or 32-bit store by one lane, the fence, a 64-bit load of the record by all 64 lanes, the line in L1 or not, up to 3 072 waves:
  // no stale read in 4 x 10^9 (profiles/r04/zb_slab_handover_drain.txt).
  //
  // Round 4 chased a memory fault of a test build on the slab-resident twins through these hand-overs -- explicit waits at every one of them
  // cured it -- before finding the cause elsewhere: VGPR spills placed by the register allocator inside a two-instruction region of narrowed
  // EXEC (decomp_device.h: ev_clear_marks); the waits had only moved the spills.  Two things learnt on the way stay: a fence at "workgroup"
  // scope and __syncthreads() emit NOTHING in these kernels either (workgroup == one wave: the compiler lowers the scope), so a hand-over
  // that is meant to wait says so with an explicit s_waitcnt (wsync_mem()); and -DALD_SLAB_DRAIN makes every hand-over of the slab-resident
  // classes wait (twins' band 66.5 -> 67.8 ms), for A/B.
  #define ALD_WSYNC_WAIT_() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while(0)
  #if defined(ALD_WSYNC_BARRIER)
  __device__ __forceinline__ void     wsync() { __syncthreads(); __builtin_amdgcn_s_waitcnt(0); }
  #elif defined(ALD_SLAB_DRAIN) && defined(ALD_CLASS_ID) && ALD_CLASS_ID >= ALD_FIRST_GLOBAL_CLASS
  __device__ __forceinline__ void     wsync() { ALD_WSYNC_WAIT_(); }
  #else
  __device__ __forceinline__ void     wsync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
  #endif
  // wsync_mem(): the hand-over that waits for the wave's memory counters (s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)) in every class: where the
  // slab-resident classes hand the kept sweep records from the lane that evaluated a vertex to the lane that owns it, and in the routines
  // whose lanes exchange through the slab whatever the class (collecting finished paths, finish_graph, the device pre-steps).  Not needed for
  // ordering (see above) and not measurable in time; it bounds how far a wave runs ahead of its own stores at these few places.
  #ifdef ALD_WSYNC_BARRIER
  __device__ __forceinline__ void     wsync_mem() { __syncthreads(); __builtin_amdgcn_s_waitcnt(0); }
  #else
  __device__ __forceinline__ void     wsync_mem() { ALD_WSYNC_WAIT_(); }
  #endif
  __device__ __forceinline__ unsigned long long atomic_add_u64(ALD_GLOBAL unsigned long long *p, unsigned long long v) { return atomicAdd((unsigned long long*)p, v); }
  __device__ __forceinline__ int      atomic_add_i32(ALD_GLOBAL int *p, int v) { return atomicAdd((int*)p, v); }
  __device__ __forceinline__ int      ffs64(uint64_t m) { return __ffsll((unsigned long long)m) - 1; }
  // uni(): "this value is the same in every active lane".  The scalar routines run with only lane 0 active, but the compiler
  // cannot know that a loaded value is wave-uniform; routing it through v_readfirstlane tells it, so loop control, address math
  // and branches on it use SGPRs / scalar branches instead of VALU compares and exec-mask save/restore sequences.
  __device__ __forceinline__ int      uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
  __device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
  __device__ __forceinline__ unsigned short uni(unsigned short v) { return (unsigned short)__builtin_amdgcn_readfirstlane((int)v); }
  __device__ __forceinline__ unsigned char uni(unsigned char v) { return (unsigned char)__builtin_amdgcn_readfirstlane((int)v); }
  __device__ __forceinline__ bool     uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
  __device__ __forceinline__ unsigned long long uni(unsigned long long v) { unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)); return ((unsigned long long)hi << 32) | lo; }
  __device__ __forceinline__ unsigned long uni(unsigned long v) { return (unsigned long)uni((unsigned long long)v); }
  __device__ __forceinline__ double   uni(double v) { return __longlong_as_double((long long)uni((unsigned long long)__double_as_longlong(v))); }
  }
#endif

namespace ald {

static const double kSMIN = 0.00001;            // rnacore/splice_graph.h:18
enum { T_TRIVIAL = 0, T_SPLITTABLE_PURE = 4, T_UNSPLITTABLE_SINGLE = 5 };   // util/constants.h:30-40
enum { K_EMPTY_VERTEX = -9 };                   // util/constants.h:53
enum { OP_BROKEN = 1, OP_TRIVIAL_FAST = 2, OP_TRIVIAL_NOW = 3, OP_TRIVIAL_BEST = 4, OP_SMALL_NOW = 5, OP_SMALLEST = 6,
       OP_UNSPLIT_NOW = 7, OP_UNSPLIT_BEST = 8, OP_GREEDY = 9, OP_COLLECT = 10 };
enum { HF_OCC = 1, HF_LEXT = 2, HF_REXT = 4 };
enum { NZ_MEMBER = 1, NZ_MEMO_VALID = 2, NZ_MEMO_TYPE_SHIFT = 2 /* 3 bits */, NZ_MEMO_DEG_GT1 = 32 };      // bits of Hot::nz
// path record (4-byte words): [0]=graph [1]=path index [2]=#vertices [3]=length [4]=count [5]=strand char | attempt << 8
//                             [6..13] = weight, abd, conf, reads (f64)   [14] = #exon words (2 per exon)   [15] = 0
//                             [16..16+nv) vertices, then the exon words (l, r)* of the transcript the path becomes -- touching
//                             vertex intervals joined, empty ones dropped (essential.cc:719-748) --, padded to an even word count
enum { REC_HDR_WORDS = 16, REC_NEXW = 14 };
#if defined(__HIP__)
  #define ALD_HD __host__ __device__
#else
  #define ALD_HD
#endif
ALD_HD static inline unsigned long long rec_words(unsigned nv, unsigned nexw) { unsigned long long w = (unsigned long long)REC_HDR_WORDS + nv + nexw; return w + (w & 1); }

// ---- wire format as the kernel sees it: device pointers into ONE coalesced HBM buffer ----
struct BatchIn {
    int32_t n_graphs;
    ALD_GLOBAL const int32_t *g_nv, *g_ne, *g_np;
    ALD_GLOBAL const int64_t *off_v, *off_e, *off_s, *off_p, *off_pv;   // [n+1] prefix sums; local CSR arrays start at off_x[g] + g
    ALD_GLOBAL const int32_t *vertex_offset, *edge_target;
    ALD_GLOBAL const double  *edge_weight; ALD_GLOBAL const uint8_t *edge_strand; ALD_GLOBAL const double *edge_abd;
    ALD_GLOBAL const int32_t *edge_sample_offset, *sample_id; ALD_GLOBAL const double *sample_abd;
    ALD_GLOBAL const double  *vertex_weight; ALD_GLOBAL const int32_t *vertex_lpos, *vertex_rpos, *vertex_type;
    ALD_GLOBAL const int32_t *in_offset, *in_edge;                      // host-built in-CSR: ids sorted by (target, source, id)
    ALD_GLOBAL const int32_t *phasing_offset, *phasing_vertex, *phasing_count;
    ALD_GLOBAL const char    *graph_strand;
    ALD_GLOBAL const int32_t *edge_count;                               // edge_info.count at hand-over (not always |samples|)
    ALD_GLOBAL const int32_t *edge_rank;                                // creation rank (scallop edge index) of every input edge, or null: CSR position
    // raw graphs (the pre-steps of assembler::assemble run in the kernel's load phase): null when the batch holds none
    ALD_GLOBAL const int32_t *g_rawdist;                                // [n] -1: staged graph, else max_group_boundary_distance of a raw one
    ALD_GLOBAL const int64_t *off_rp, *off_rc;                          // [n+1] prefix of raw phases / of their coordinates
    ALD_GLOBAL const int32_t *rphase_offset, *rphase_coord, *rphase_count;   // local CSR (np + 1 entries at off_rp[g] + g), exon coordinates, counts
};
struct BatchOut {
    ALD_GLOBAL int32_t *status, *n_paths, *n_iters;        // [n]
    ALD_GLOBAL unsigned long long *pool_used;              // words used (atomic bump)
    ALD_GLOBAL uint32_t *pool; unsigned long long pool_cap;// path-record pool (4-byte words)
    // result index, written by the kernel: a graph that ENDS WELL reserves n_paths consecutive entries (one atomic per graph) and
    // leaves the pool offsets of its records there in path order -- index[graph_first[g] + p] = word offset of record (g, p).
    // Records of abandoned attempts are referenced by nothing.
    ALD_GLOBAL unsigned long long *index_used; ALD_GLOBAL unsigned long long *index; unsigned long long index_cap;
    ALD_GLOBAL long long *graph_first;                      // [n]
    ALD_GLOBAL int32_t *trace_n, *trace_codes; ALD_GLOBAL double *trace_vals; int32_t trace_cap;   // optional op trace
};
struct Params { double max_ratio[8]; double min_w; double min_cov; int32_t max_num_exons; int32_t pad; };
struct KernelArgs {                               // lives in device memory; every wave keeps a pointer to it in LDS
    BatchIn in; BatchOut out; Params prm;
    ALD_GLOBAL const int32_t *work; int32_t n_work; int32_t attempt;   // attempt tags the records of a retry pass
    ALD_GLOBAL int32_t *counter;
    ALD_GLOBAL uint8_t *slabs; uint64_t slab_stride;
};

// ---- size classes ----
// Geometric-ish ladder: a graph pays for the LDS of its class for as long as it runs, and a mixed batch is bound by exactly that
// product (LDS bytes x time), so the steps are fine where the time is spent (257..512 vertices in cfg3).
// Classes 0..10 are what pick_class() chooses from.  Classes 11 and 12 are TWINS of 7 and 8 (same capacities) that keep the hot state in
// the wave's HBM slab instead of LDS: a graph runs 2-3 times longer there, but twelve workgroups fit a CU instead of three and none
// of them takes LDS away from the other classes -- the host moves a class's graphs to its twin when there are more of them than the
// LDS form can hold at once (ald_abi.cpp: ald_batch_upload).
// Class 13 takes what is beyond the catch-all: graphs of up to 10 240 vertices (the reference runs its rule loop up to max_num_exons =
// 10 000 vertices, util/parameters.cc; beyond that it only runs the greedy phase) and 58 752 edges, hot state in the slab like class 10,
// creation ids in 32 bits (a graph of that size makes more than 65 535 edges in its life).  ~105 MB of slab per wave, at most 32 waves.
#define ALD_NUM_PICK_CLASSES 11
#define ALD_NUM_CLASSES 14
#define ALD_CATCH_ALL_CLASS 10
#define ALD_HUGE_CLASS 13
#define ALD_HUGE_CLASS_WAVES 32
#define ALD_FOR_EACH_PICK_CLASS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#define ALD_FOR_EACH_CLASS(X) ALD_FOR_EACH_PICK_CLASS(X) X(11) X(12) X(13)
template<int ID> struct ClassDims;
template<> struct ClassDims<0>  { enum { MAXV = 64,   MAXE = 160,  NW = 1 }; };     // V <= 32
template<> struct ClassDims<1>  { enum { MAXV = 128,  MAXE = 288,  NW = 1 }; };     // V <= 64   (the bench workload: 20 workgroups per CU)
template<> struct ClassDims<2>  { enum { MAXV = 256,  MAXE = 640,  NW = 2 }; };     // V <= 128
template<> struct ClassDims<3>  { enum { MAXV = 384,  MAXE = 960,  NW = 3 }; };     // V <= 192
template<> struct ClassDims<4>  { enum { MAXV = 512,  MAXE = 1280, NW = 4 }; };     // V <= 256
template<> struct ClassDims<5>  { enum { MAXV = 640,  MAXE = 1440, NW = 5 }; };     // V <= 320
template<> struct ClassDims<6>  { enum { MAXV = 768,  MAXE = 1696, NW = 6 }; };     // V <= 384  (1696 edges: four workgroups per CU)
template<> struct ClassDims<7>  { enum { MAXV = 896,  MAXE = 2000, NW = 7 }; };     // V <= 448
template<> struct ClassDims<8>  { enum { MAXV = 1024, MAXE = 2280, NW = 8 }; };     // V <= 512  (2280 edges: three workgroups per CU; holds E <= 2052)
template<> struct ClassDims<9>  { enum { MAXV = 2048, MAXE = 6600, NW = 32 }; };    // one workgroup per CU: 145 KB of its 160 KB LDS; V <= 1024, E <= 5940
template<> struct ClassDims<10> { enum { MAXV = 4096, MAXE = 16384, NW = 64 }; };   // catch-all: hot state in the wave's HBM slab, not LDS
template<> struct ClassDims<11> { enum { MAXV = 896,  MAXE = 2000, NW = 7 }; };     // twin of class 7, hot state in the slab
template<> struct ClassDims<12> { enum { MAXV = 1024, MAXE = 2280, NW = 8 }; };     // twin of class 8, hot state in the slab
template<> struct ClassDims<13> { enum { MAXV = 20480, MAXE = 65280, NW = 160 }; }; // V <= 10240, E <= 58752: hot state in the slab, 32-bit creation ids
// Row pool of a class (the ALD_ROWS build, decomp_device_rows.h: Hot::adj): the adjacency ROWS of all vertices, 16-bit edge slots, in segments of whole
// 4-entry chunks.  Every live edge has two entries (its source's out-row, its target's in-row), a segment is rounded up to a chunk
// and a row that outgrows its segment moves to a larger one: three entries per edge slot, compacted when the pool runs out.  Class 9
// fills its CU's LDS (one workgroup per CU) and gets what is left of the 160 KB.
template<int ID> struct ClassAdj { enum { ADJ = (3 * ClassDims<ID>::MAXE + 3) / 4 * 4 }; };
template<> struct ClassAdj<9> { enum { ADJ = 18400 }; };                            // 2.79 x 6600
static inline int class_twin(int c) { return c == 7 ? 11 : c == 8 ? 12 : -1; }       // slab-resident twin of an LDS class (-1: none)
static inline int class_retry_up(int c)                                              // where a graph goes when its working set overflowed class c
{
    if(c == 11) return 12;
    if(c == 12) return 9;
    if(c == ALD_CATCH_ALL_CLASS) return ALD_HUGE_CLASS;
    return c + 1 < ALD_NUM_PICK_CLASSES ? c + 1 : -1;
}

// per-wave HBM slab, laid out at compile time (so that cold pointers cost no registers)
template<int MAXV, int MAXE, int NW, int ADJ>
struct ColdLayoutT {
    static constexpr uint64_t al(uint64_t x) { return (x + 15) / 16 * 16; }
    static constexpr uint32_t SP_CAP = 8u * MAXE;        // sample-support pool entries (input + intersections)
    static constexpr uint32_t HL_CAP = 8u * MAXE;        // phasing-list pool (ints)
    static constexpr int32_t  HL_MAXLISTS = MAXE;
    static constexpr int32_t  W_CAP = 8 * MAXE;          // scalar work arrays (ints / doubles)
    // per-vertex and per-edge-slot cold state are arrays of records (ColdVertex / ColdEdge in decomp_device.h): one merge
    // touches the two or three records involved, i.e. two or three cache lines, instead of one line per field
    static constexpr uint64_t VX_BYTES = 32;                                        // vw f64, lpos, rpos, vtype, v2v, memo i32 (+ pad)
    static constexpr uint64_t ED_BYTES = (56 + 8ull * NW + 63) / 64 * 64;          // 4 f64 + mask[NW] + 5 i32 + strand, padded to whole lines
    static constexpr uint64_t o_vx = 0;
    static constexpr uint64_t o_ed = (o_vx + VX_BYTES * MAXV + 63) / 64 * 64;
    static constexpr uint64_t o_spid = al(o_ed + ED_BYTES * MAXE);
    static constexpr uint64_t o_spabd = al(o_spid + 4ull * SP_CAP);
    static constexpr uint64_t o_hl = al(o_spabd + 8ull * SP_CAP);
    static constexpr uint64_t o_hloff = al(o_hl + 4ull * HL_CAP);
    static constexpr uint64_t o_hllen = al(o_hloff + 4ull * HL_MAXLISTS);
    static constexpr uint64_t o_hlcapk = al(o_hllen + 4ull * HL_MAXLISTS);
    static constexpr uint64_t o_hlcnt = al(o_hlcapk + 4ull * HL_MAXLISTS);
    static constexpr uint64_t o_wi = al(o_hlcnt + 4ull * HL_MAXLISTS);
    static constexpr uint64_t o_wd = al(o_wi + 4ull * W_CAP);
    static constexpr int32_t  PO_CAP = MAXE;             // record offsets of the graph at hand, in path order (published by finish_graph)
    static constexpr uint64_t o_po = al(o_wd + 8ull * W_CAP);
    static constexpr uint64_t o_adjtmp = al(o_po + 8ull * PO_CAP);        // compaction of the row pool: the rows in their new places (ADJ x u16)
    static constexpr uint64_t o_evr = al(o_adjtmp + 2ull * ADJ);          // the smallest-edge evaluation of every vertex, kept between the sweeps
    static constexpr uint64_t o_eve = al(o_evr + 8ull * MAXV);            // (decomp_device.h: ALD_KEEP, sweep_smallest): ratio f64 / edge i32 per vertex
    static constexpr uint64_t o_tvr = al(o_eve + 4ull * MAXV);            // the trivial-vertex scan's view of every vertex, kept likewise (scan_trivial): balance
    static constexpr uint64_t o_tvc = al(o_tvr + 8ull * MAXV);            // ratio f64 / class i32 per vertex
    static constexpr uint64_t total = (o_tvc + 4ull * MAXV + 255) / 256 * 256;
};

struct ClassInfo { int maxv, maxe, nw; uint32_t sp_cap, hl_cap; uint64_t slab_bytes; };
static inline ClassInfo class_info(int c)
{
    switch(c) {
#ifdef ALD_ROWS
#define ALD_SLAB_ADJ(ID) ClassAdj<ID>::ADJ
#else
#define ALD_SLAB_ADJ(ID) 0
#endif
#define ALD_CI(ID) case ID: { typedef ColdLayoutT<ClassDims<ID>::MAXV, ClassDims<ID>::MAXE, ClassDims<ID>::NW, ALD_SLAB_ADJ(ID)> L; \
        return ClassInfo{ClassDims<ID>::MAXV, ClassDims<ID>::MAXE, ClassDims<ID>::NW, L::SP_CAP, L::HL_CAP, L::total}; }
    ALD_FOR_EACH_CLASS(ALD_CI)
#undef ALD_CI
    }
    return ClassInfo{0, 0, 0, 0, 0, 0};
}
// smallest class whose working set can hold the graph: vertices may double (decompose_vertex_extend adds pseudo vertices,
// scallop.cc:1793-1806); edges need transient head-room (split_edge before merge); the sample pool must hold the input
// plus the intersections created by merges.
static inline int pick_class(int V, int E, int64_t n_samples, int64_t n_phasing_vertices, int first = 0)
{
    auto fits = [&](int c) { ClassInfo k = class_info(c);
        return V <= k.nw * 64 && 2 * V <= k.maxv && E + k.maxe / 10 <= k.maxe && 2 * n_samples <= (int64_t)k.sp_cap && 4 * n_phasing_vertices <= (int64_t)k.hl_cap; };
    for(int c = first < 0 ? 0 : first; c < ALD_NUM_PICK_CLASSES; c++) if(fits(c)) return c;
    if(fits(ALD_HUGE_CLASS)) return ALD_HUGE_CLASS;
    return -1;                                    // beyond every class: status ALD_ST_TOO_LARGE
}

#if !defined(__HIP_DEVICE_COMPILE__)
// test knob: ALD_DEBUG_UNDERCLASS=k starts every graph k classes below the one pick_class chose, so that the capacity-retry path
// (status ALD_ST_CAPACITY -> re-queued one class up, stale records dropped) runs on ordinary inputs
static inline int debug_underclass(int c)
{
    const char *e = getenv("ALD_DEBUG_UNDERCLASS");
    const int k = e ? atoi(e) : 0;
    if(c < 0 || k <= 0) return c;
    return c - k < 0 ? 0 : c - k;
}
#endif

} // namespace ald
