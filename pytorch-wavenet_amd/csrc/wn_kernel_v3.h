// wn_kernel_v3.h -- multi-stream generation chain with WAVE-SPECIALISED layer workgroups (gfx950, device only).
//
// The chain, its hand-off granules, the HBM buffers and the per-lane weight images are those of DESIGN.md sections 2 and 7 (shapes and
// images: wn_chain_regs.h, WnV2Shape); this file decides WHO inside a layer workgroup does what.  The form it replaced (rounds 1-2: one
// 256-thread workgroup per layer slice with a single instruction stream, profiles/HISTORY.md) measured on cfg3 (profiles/archive/r02_sweep_one_chain.txt):
// a pipeline item keeps such a workgroup busy ~1.15 us, of which only ~0.47 us (stage x, filter/gate, gate, residual,
// publish x') is on the token's critical path; the other ~0.7 us (skip 1x1 + the running skip lane, queue push, queue tap, the
// next step's tap-0 half of the dilated conv) is work nobody downstream is waiting for -- but it sits in the same instruction
// stream, so with 64 tokens in flight the chain saturates at 64 x 1.87 us per timestep while its 53 stages could turn a token
// around in 58 us (16 tokens: 58.2 us per timestep).  Each of these instruction streams is latency bound (~6 cycles per
// instruction with one wave per SIMD), so the cure is more waves, each with its own part of the item.  A layer workgroup has
// 768 threads = three wave groups, one wave of each per SIMD:
//     waves 0-3  C "critical": poll x' partials (two request sets in flight, hand-scheduled: wn_ap_poll4) -> stage x -> [A]
//                              -> filter/gate dot (tap 1) + parked tap 0 -> tanh*sigmoid -> z -> [B] -> residual partial
//                              -> publish x' -> request the next item's inputs
//     waves 4-7  S "skip":     [A] -> request the upstream skip lane (its round trip runs next to the critical dot) -> [B]
//                              -> skip 1x1 partial -> add -> publish                                                -> [A]
//     waves 8-11 Q "queue":    d = 1 / few streams and small d:  [A] -> queue push of x[t], stage the tap -> [B] -> tap-0 half of the
//                              dilated conv for the NEXT timestep of this stream -> park it -> [A];
//                              every other layer ("late"): nothing between [A] and [B]; after [B]: push x[t] (the waves that load
//                              no taps), tap-0 dot on the tap staged an item ago, request the tap of item i+6, stage the tap of
//                              item i+1 (hand-scheduled register FIFO: wn_q_issue / wn_q_take) -> [A]
// [A] and [B] are the two LDS-only workgroup barriers of an item (all 12 waves).  Each group holds only the weights of its
// part (C: tap 1 + residual slice, S: skip slice, Q: tap 0) and has a memory stream of its own (C polls x' partials, S the
// skip lane, Q reads/writes the queue): a slow queue read can never sit in front of a poll.  ONE chain serves all streams
// (no second copy of the weights, no second set of hand-off buffers).  Head workgroups request the next item's skip lanes as soon
// as they have staged the current ones; samplers perform layer 0's start_conv.
//
// Throughput form (template parameter G = 2, chosen by the host from 56 streams up together with two replicas of the head
// workgroups, wn_v3_mode in wn_runtime.hip): a pipeline item of a layer workgroup carries TWO streams, s and s+1.  Hand-offs stay per
// stream (the granules of consecutive streams are R / S apart), so head and sampler workgroups and the protocol are unchanged; the
// request round trip, both barriers and every LDS / DPP / transcendental latency of an item are paid once per two streams and
// every weight operand is used twice.  All G*R lanes poll / stage / push (lane t: element t % R of stream s + t / R); x, z and the
// taps are [G][...] in LDS; the tap FIFO is unchanged (every queue wave still issues one tap load per item).  The two streams'
// dependency chains of a window are written as ONE basic block (unconditional LDS reads, selects, stores after both chains):
// a lane-predicated store between them makes the compiler emit the second stream's whole chain after the first one's.  In this
// form a critical lane computes the filter AND the gate row of a channel on half an x slice (WN_V3_PAIR_ROWS).  A trip through a
// stage is longer (0.10 + 0.37 + 0.26 us against 0.08 + 0.26 + 0.17), which is why few streams keep G = 1.
//
// LDS hazards (i = item index; x and the late layers' tap are double buffered, everything else single; W = written in, R = read in):
//   xs[buf(i)]   W: C before A(i).  R: C in A(i)..A(i+1), Q in A(i)..A(i+1) (late layers push after B(i)).  Next W (item i+2) after B(i+1).
//   zs           W: C in A(i)..B(i).  R: C and S in B(i)..A(i+1).
//   xo           not late: W: Q in A(i)..B(i).  R: Q in B(i)..A(i+1).   late: xo[i & 1]: W: Q in B(i-1)..A(i).  R: Q in B(i)..A(i+1).
//   pre[s]       W: Q in B(i)..A(i+1) for item (e, s).  R: C in A(j)..B(j) of item j = (e+1, s) = i + n_streams >= i + 1.
// Queue (HBM) hazard: the tap of item j is requested WN_V3_TAP_AHEAD items ahead, after barrier B of item j-WN_V3_TAP_AHEAD; it was pushed
// (d >= 2) at item j - n_streams*(d-1).  Where that is no more than WN_V3_TAP_AHEAD items back (few streams, small d) the queue group takes
// the row from its own registers instead.
#ifndef WN_KERNEL_V3_H
#define WN_KERNEL_V3_H

#include "wn_chain_regs.h"

#ifndef WN_THREADS_V3
#define WN_THREADS_V3 768
#endif
#define WN_V3_MIN_STREAMS 1
#define WN_V3_TAP_AHEAD 6
#ifndef WN_SAMPLER_PREFETCH
#define WN_SAMPLER_PREFETCH 2  // the sampler asks for the item's uniform BEFORE it waits for the logits (2; 1: also temperature / given sample -- the
                               // build whose scalar pressure spilled the input poll's base pointers, see WN_AP_SGPR_HAZARD; 0: everything behind the logits)
#endif
#ifndef WN_V3_WIDE_SAMPLER
#define WN_V3_WIDE_SAMPLER 0  // experiment: head slices from which the samplers of variant 3 collect the logits with four waves (0 = never)
#endif
#define WN_V3_COMPILER_VGPRS 152  // v152-v167: request sets of the input poll / the queue group's tap FIFO (see wn_ap_*, wn_q_*)
// ---- experiment switches.  A product build (build.py) leaves every one of them at its default; setting one requires -DWN_EXPERIMENT,
// which build.py never passes (tools/ builds the A/B variants): a library with wrong-on-purpose timing ablations cannot ship by accident.
#if !defined(WN_EXPERIMENT) && (defined(WN_V3_SKIP_DEFER) || defined(WN_V3_QUEUE_DEFER) || defined(WN_V3_SKIP_SLEEP) || defined(WN_V3_ABL) || defined(WN_V3_PAIR_ROWS) || defined(WN_V3_PRIO) || defined(WN_V3_LAST_SKIP_PRIO) || defined(WN_V3_SKIP_CHAINS) || defined(WN_V3_FG_CHAINS) || defined(WN_V3_SKIP_SLOTS) || defined(WN_V3_TAP_AT_A) || defined(WN_V3_FAST_GATE) || defined(WN_V3_KPACK))
#error "WN_V3_* experiment switches need -DWN_EXPERIMENT"
#endif
#ifndef WN_V3_SKIP_DEFER
#define WN_V3_SKIP_DEFER 0   // two-streams-per-item form: s_sleep count (64 clocks each) at the head of the skip group's chunk behind barrier B (not the last
                             // layer's: its lanes are the head's input) -- the chunk then runs while the critical waves wait for the next token instead of next
                             // to their residual dot, the piece of the window that is on the token's path
#endif
#ifndef WN_V3_QUEUE_DEFER
#define WN_V3_QUEUE_DEFER 0  // ... and the same in front of the queue group's chunk behind barrier B (push, tap-0 dot of the next timestep)
#endif
#ifndef WN_V3_SKIP_SLEEP
#define WN_V3_SKIP_SLEEP 0  // s_sleep between the skip group's poll retries (the skip lane is not latency critical; fewer polls on the fabric)
#endif
#ifndef WN_V3_ABL
#define WN_V3_ABL 0  // timing ablations (results are WRONG when != 0): 1 the skip group only polls the input and passes its barriers, 2 the queue group, 3 both;
                     // 4: the skip group publishes its own dot WITHOUT the upstream lane (no chain through the layers), 8: the upstream lane is taken as it
                     // comes back from the request at barrier A, fresh or not (no polling)
#endif
#ifndef WN_V3_PAIR_ROWS
#define WN_V3_PAIR_ROWS 2  // a critical lane computes the filter AND the gate row of one channel on a half-width slice of x (see wn_v3_layer):
                           // 0 never, 1 always, 2 in the two-streams-per-item form only (64 streams: 961 -> 974 k samples/s, 128: 1.464 -> 1.478 M;
                           // one stream: 18.87 -> 18.74 k -- the longer lane reduction is on the single token's path; profiles/archive/r02_v3_forms_final.txt)
#endif
#ifndef WN_V3_LAST_SKIP_PRIO
#define WN_V3_LAST_SKIP_PRIO 3  // wave priority of the LAST layer's skip group in the two-streams-per-item form (64 streams: 998.6 -> 1004.5 k)
#endif
#ifndef WN_V3_SKIP_SLOTS
#define WN_V3_SKIP_SLOTS 0   // > 0: hand-off slots of a skip lane that stays inside one XCD are re-used per in-flight ITEM instead of one per stream (0: per
                             // stream), in EVERY kernel (experiment switch).  Correct, cuts the job's L2 <-> fabric traffic, and LOSES 2-6 % where the ring is
                             // latency bound (cfg3 up to 80 streams; round 4: profiles/r04_skip_lane_slot_reuse_experiment.txt) -- and GAINS 1-6 % where it is
                             // throughput bound (cfg3 from 96 streams: 128 streams 1.50 -> 1.58 M): the product has it as a FORM of the cfg3 kernel (template
                             // parameter SK = 4 of wn_generate_kernel_v3m, chosen by the host from 96 streams up: wn_v3_slots_for).  Other shapes lose with it
                             // at every stream count (cfg2 x 128 -21 %, the train_script shape x 64 -25 %) and have no such form.
#endif
#ifndef WN_V3_SKIP_CHAINS
#define WN_V3_SKIP_CHAINS 1  // independent FMA chains per row pair of the skip group's dot (1: one chain of DC packed FMAs, the arithmetic of rounds 2-3)
#endif
#ifndef WN_V3_FG_CHAINS
#define WN_V3_FG_CHAINS 4    // ... per stream of the critical group's filter/gate dot in the pair-rows form (2: rounds 2-3; 4: 64 streams 1000 -> 1010 k, profiles/r04_fma_chain_experiments.txt)
#endif
#ifndef WN_V3_TAP_AT_A
#define WN_V3_TAP_AT_A 2     // when a late layer's queue waves request the tap of item i + 6: 0 behind their tap-0 dot after barrier B(i) (rounds 2-4), 1 right behind
                             // barrier A(i), 2 behind A in the two-streams-per-item form only.  The rings of the layers with d >= 64 do not fit the L2 at 64
                             // streams: such a tap load is a ~1 us miss, and whatever this CU requests behind it -- the critical group's input polls -- is
                             // answered behind it.  Requested after the dot the miss was in flight when the next token arrived; behind barrier A it has the
                             // item's whole service time and the wait for the next token to itself (round 5: hop into a layer with d >= 64 0.50 -> 0.35 us,
                             // 64 streams 1.002 -> 1.069 M samples/s; one stream per item loses 1-3 %: profiles/r05_tap_request_behind_barrier_a.txt)
#endif
#ifndef WN_V3_FAST_GATE
#define WN_V3_FAST_GATE 1    // e^-v of the gated unit as exp2(v * -log2 e) -- three instructions (select, multiply by a per-lane constant, v_exp_f32) instead of
                             // the eleven of the library-accurate wn_exp; the single rounding of the product moves sigma by <= 1.4e-8 (the bound of variant 4's
                             // gate, wn_kernel_v4.h).  The filter/gate window of a layer is ISSUE bound (round 5: ~125 instructions of one wave per SIMD in 0.39 us):
                             // every instruction taken out of it is ~2.5 ns per layer, 0.12 us per timestep of the 64-stream ring.  0: wn_exp as in rounds 2-4
#endif
#ifndef WN_V3_KPACK
#define WN_V3_KPACK 1        // pair-rows form: the packed FMAs pair CONSECUTIVE x elements {x[k], x[k+1]} against {w[k], w[k+1]} of the filter row and of the gate
                             // row (no broadcast of one x element into both halves: the compiler spent a v_mov per fourth element on those), and the parked tap-0
                             // sums enter through one FMA with a per-lane 0 / 1 factor behind the dot instead of two selects in front of it.  0: {filter, gate} pairs
#endif
#ifndef WN_V3_PRIO
#define WN_V3_PRIO 1  // 1: critical waves at a higher static wave priority (the queue and skip waves share their SIMDs: x64 911 -> 919 k, profiles/archive/r02_v3_tap_fifo.txt)
#endif

// ---- 16-byte skip-lane hand-offs.  A lane of the skip group owns rows t and t + 256 of the running skip sum.  Written as two
// 8-byte granules that is two write-through stores per lane and item -- and the fabric retires write-through stores per LANE, not
// per byte (MI355X guide: dwordx2 stores cost 2.7x the dwordx4 time per byte): at 64 streams the chain sat on a ceiling of ~130 G
// lane-stores/s, 1.0-1.06 TB/s of granules whatever the stream count (profiles/archive/r02_v3_lazy_clock.txt).  The two granules of a lane
// are therefore adjacent in memory, {value(t), tag, value(t+256), tag}, written by ONE 16-byte store and read by ONE 16-byte
// load; each 8-byte half still carries its own tag, so nothing depends on the 16 bytes arriving together.
typedef int wn_v4i __attribute__((ext_vector_type(4)));
typedef int wn_v2i __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t wn_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);  // raw buffer, 2 GB window
}
static __device__ __forceinline__ wn_v4i wn_ld_pair(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);  // aux 16 = sc1: served by L2, never by this CU's L1
}
static __device__ __forceinline__ void wn_st_pair(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, uint32_t tag, float v0, float v1, bool local) {
    const wn_v4i d = {__float_as_int(v0), (int)tag, __float_as_int(v1), (int)tag};
    if (local) __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 0);   // every consumer sits on this XCD: the line may stay in its L2
    else __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 16);        // write-through
}
// spins until both halves of the pair carry `tag` (bounded like wn_poll_fixed)
static __device__ __forceinline__ wn_v4i wn_poll_pair(WnCtx& cx, __amdgpu_buffer_rsrc_t rs, unsigned byte_off, uint32_t tag, int where, long long e, int s, int sleep = 0) {
    wn_v4i v = {0, 0, 0, 0};
    if (cx.fail) return v;
    unsigned spins = 0;
    for (;;) {
        v = wn_ld_pair(rs, byte_off);
        if ((uint32_t)v.y == tag && (uint32_t)v.w == tag) return v;
        if (sleep > 0) __builtin_amdgcn_s_sleep(WN_V3_SKIP_SLEEP > 0 ? WN_V3_SKIP_SLEEP : 1);
        if ((++spins & 127u) == 0u) {
            if (__hip_atomic_load(cx.p->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; return v; }
            const long long now = (long long)wall_clock64();
            if (spins == 128u) cx.t_start = now;
            else if (now - cx.t_start > cx.r->timeout_ticks) { wn_give_up(cx, where, e, s); return v; }
        }
    }
}

// ---- Input polling of the critical group with TWO request sets in flight, hand-scheduled.
// What the stamps of a 64-stream run show (profiles/archive/r02_v3_first_check.txt, r02_load_flavour_probe.txt): a load takes ~0.1 us
// on an idle chip and 0.25-0.3 us in the running chain, a store becomes visible ~0.25 us after its issue idle and ~0.45 us in the
// running chain.  The request issued at the end of an item therefore comes back STALE in every item, and the token is caught by
// the next poll a whole round trip later: the stage's cycle is quantised, service + 2 round trips (1.19 us).  With two sets in
// flight half a round trip apart the token is caught within half a round trip of becoming visible.
// The compiler cannot be made to schedule this: it derives every s_waitcnt vmcnt(N) from the order of the memory operations it
// sees and falls back to waiting for (nearly) everything wherever paths with different sequences meet (loop entry vs back edge,
// early exits, a store under a lane predicate) -- every variant written in C++ ended up waiting for the YOUNGEST set at every
// check, and registers of a set that is still in flight when the wave moves on are handed to the next temporary behind a
// full wait (profiles/archive/r02_v3_request_experiments.txt).  So the sets live in the sixteen HIGHEST registers of the
// 168 a 768-thread workgroup leaves each lane (A = v[152:159], B = v[160:167]; the compiler's own allocation stays below them --
// tests/test_abi.py disassembles the library and checks that no instruction outside these blocks touches them; accumulation
// registers would make the allocator split the register file in halves), and the loop is written out: issue, s_waitcnt vmcnt(4) = "the older set
// is complete", check, reissue.  The wave leaves as soon as all its lanes have their input, with one set still in flight; it
// lands in registers nobody else uses and is overwritten (in order) by the next request.
// (the clobbers make the kernel's register count cover v167; the compiler cannot allocate them anyway: amdgpu_num_vgpr)
#define WN_AP_CLOBBERS "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "vcc", "scc", "memory"
// The blocks take their base pointers in SGPR pairs ("s" operands).  Under register pressure the compiler keeps such a pointer in a lane of
// its SGPR-spill VGPR and brings it back with v_readlane_b32 right in front of the block -- a VALU write of an SGPR that a VMEM instruction
// reads as its address needs FIVE wait states in between on gfx9 (the backend's hazard recognizer inserts them for its own instructions,
// not in front of inline assembly): without them the load goes out with whatever the SGPR pair held before -- round 4 met it as a memory
// access fault on address nil / 0xffffa4db9000 in the two-slice form (tests/test_gpu_parity.py two_way_split, tools/stress_split.py),
// the first build whose scalar pressure spilled those pointers.  Every block whose first VMEM instruction reads an "s" operand starts with
// WN_AP_SGPR_HAZARD: five wait states by default.  They sit on the token's path (cfg3 single stream: 49.5 -> 51.3 us per timestep), and a
// build in which no such pointer is reloaded in front of a block does not need them: build.py compiles with -DWN_AP_SGPR_HAZARD="" FIRST,
// DISASSEMBLES the result and checks every hand-scheduled load for the hazard (check_hand_scheduled_registers, rule 4), and only falls back
// to this default when the check finds one.  Any other way of compiling this file gets the safe form.
#ifndef WN_AP_SGPR_HAZARD
#define WN_AP_SGPR_HAZARD "s_nop 4\n\t"
#endif
#ifndef WN_AP_LOOP_ALIGN
#define WN_AP_LOOP_ALIGN ""   // e.g. ".p2align 6\n\t": the head of the polling loops on an instruction-cache line (filled with s_nop, executed once per entry)
#endif
// (addresses: a wave-uniform base per partial in an SGPR pair + ONE 32-bit byte offset per lane -- four 64-bit lane pointers were
//  eight registers of the polling waves' budget)
static __device__ __forceinline__ void wn_ap_issue_a4(unsigned off, const wn_u64* b0, const wn_u64* b1, const wn_u64* b2, const wn_u64* b3) {
    asm volatile(
        WN_AP_SGPR_HAZARD
        "global_load_dwordx2 v[152:153], %0, %1 sc1\n\t"
        "global_load_dwordx2 v[154:155], %0, %2 sc1\n\t"
        "global_load_dwordx2 v[156:157], %0, %3 sc1\n\t"
        "global_load_dwordx2 v[158:159], %0, %4 sc1"
        ::"v"(off), "s"(b0), "s"(b1), "s"(b2), "s"(b3) : WN_AP_CLOBBERS);
}
static __device__ __forceinline__ void wn_ap_issue_a2(unsigned off, const wn_u64* b0, const wn_u64* b1) {
    asm volatile(
        WN_AP_SGPR_HAZARD
        "global_load_dwordx2 v[152:153], %0, %1 sc1\n\t"
        "global_load_dwordx2 v[154:155], %0, %2 sc1"
        ::"v"(off), "s"(b0), "s"(b1) : WN_AP_CLOBBERS);
}
static __device__ __forceinline__ void wn_ap_issue_a1(unsigned off, const wn_u64* b0) {
    asm volatile(WN_AP_SGPR_HAZARD "global_load_dwordx2 v[152:153], %0, %1 sc1" ::"v"(off), "s"(b0) : WN_AP_CLOBBERS);
}
// one check of a set: lanes that are not ok yet and see fresh tags take their sum (fixed order ((0+x0)+x1)+x2)+x3, as wn_poll_fixed)
#define WN_AP_MERGE                                      \
    "v_cmp_eq_u32_e64 %[m], 0, %[ok]\n\t"                \
    "s_and_b64 %[m], %[m], vcc\n\t"                      \
    "v_cndmask_b32_e64 %[sum], %[sum], %[t0], %[m]\n\t"  \
    "v_cndmask_b32_e64 %[ok], %[ok], 1, %[m]\n\t"        \
    "v_cmp_eq_u32_e32 vcc, 0, %[ok]\n\t"                 \
    "s_nop 4\n\t"
#define WN_AP_CHECK4(B0, B1, B2, B3, B4, B5, B6, B7)    \
    "v_cmp_eq_u32_e32 vcc, %[tag], v" #B1 "\n\t"         \
    "v_cmp_eq_u32_e64 %[m], %[tag], v" #B3 "\n\t"        \
    "s_and_b64 vcc, vcc, %[m]\n\t"                       \
    "v_cmp_eq_u32_e64 %[m], %[tag], v" #B5 "\n\t"        \
    "s_and_b64 vcc, vcc, %[m]\n\t"                       \
    "v_cmp_eq_u32_e64 %[m], %[tag], v" #B7 "\n\t"        \
    "s_and_b64 vcc, vcc, %[m]\n\t"                       \
    "v_add_f32_e32 %[t0], 0, v" #B0 "\n\t"               \
    "v_add_f32_e32 %[t0], %[t0], v" #B2 "\n\t"           \
    "v_add_f32_e32 %[t0], %[t0], v" #B4 "\n\t"           \
    "v_add_f32_e32 %[t0], %[t0], v" #B6 "\n\t"           \
    WN_AP_MERGE
#define WN_AP_CHECK1(B0, B1)                             \
    "v_cmp_eq_u32_e32 vcc, %[tag], v" #B1 "\n\t"         \
    "v_add_f32_e32 %[t0], 0, v" #B0 "\n\t"               \
    WN_AP_MERGE
// Polls the granules p0..p3 until every active lane has seen four tags == tag, at most `rounds` double rounds (set A was issued by
// wn_ap_issue_a4 during the previous item; set B is issued here).  ok == 0: the lane ran out of rounds (caller: bounded wait).
// BETWEEN = vector-memory operations this wave issued between set A and this call (the publication store of the previous item,
// when set A is requested in front of it): the first wait lets them stay in flight next to set B.
#define WN_AP_ISSUE4(R0, R1, R2, R3, R4, R5, R6, R7)                             \
    "global_load_dwordx2 v[" #R0 ":" #R1 "], %[off], %[p0] sc1\n\t"               \
    "global_load_dwordx2 v[" #R2 ":" #R3 "], %[off], %[p1] sc1\n\t"               \
    "global_load_dwordx2 v[" #R4 ":" #R5 "], %[off], %[p2] sc1\n\t"               \
    "global_load_dwordx2 v[" #R6 ":" #R7 "], %[off], %[p3] sc1\n\t"
// The poll comes in two pieces, so that the caller can put its own (compiler-scheduled) publication stores between the first look and
// the spin: wn_ap_look4 waits for everything this wave has in flight (set A, requested a window ago, is the youngest load) and
// merges set A; wn_ap_spin4 -- only entered when a lane is still without its input -- issues set B, then set A again, and keeps two
// sets in flight half a round trip apart until every lane is served or `rounds` double rounds have passed.  Set B is only ever
// issued after a stale first look: with tokens queued in front of the stage the first look succeeds and no further poll is spent.
#define WN_AP_LOOK4_BODY                                                          \
        "v_mov_b32_e32 %[ok], 0\n\t"                                              \
        "v_mov_b32_e32 %[sum], 0\n\t"                                             \
        "s_waitcnt vmcnt(0)\n\t"                                                  \
        WN_AP_CHECK4(152, 153, 154, 155, 156, 157, 158, 159)
#define WN_AP_SPIN4_BODY                                                          \
        WN_AP_SGPR_HAZARD                                                         \
        "s_mov_b32 %[cnt], %[rounds]\n\t"                                         \
        WN_AP_ISSUE4(160, 161, 162, 163, 164, 165, 166, 167)                      \
        "s_sleep 2\n\t"                                                           \
        WN_AP_ISSUE4(152, 153, 154, 155, 156, 157, 158, 159)                      \
        WN_AP_LOOP_ALIGN                                                          \
        "1:\n\t"                                                                  \
        "s_waitcnt vmcnt(4)\n\t"                                                  \
        WN_AP_CHECK4(160, 161, 162, 163, 164, 165, 166, 167)                      \
        "s_cbranch_vccz 2f\n\t"                                                   \
        WN_AP_ISSUE4(160, 161, 162, 163, 164, 165, 166, 167)                      \
        "s_waitcnt vmcnt(4)\n\t"                                                  \
        WN_AP_CHECK4(152, 153, 154, 155, 156, 157, 158, 159)                      \
        "s_cbranch_vccz 2f\n\t"                                                   \
        WN_AP_ISSUE4(152, 153, 154, 155, 156, 157, 158, 159)                      \
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                         \
        "s_cmp_lg_u32 %[cnt], 0\n\t"                                              \
        "s_cbranch_scc1 1b\n"                                                     \
        "2:"
static __device__ __forceinline__ void wn_ap_look4(uint32_t tag, float& sum, int& ok) {
    float t0;
    long long m;
    asm volatile(WN_AP_LOOK4_BODY
                 : [sum] "=&v"(sum), [ok] "=&v"(ok), [t0] "=&v"(t0), [m] "=&s"(m)
                 : [tag] "s"(tag)
                 : WN_AP_CLOBBERS);
}
// (returns the double rounds that were left when the last lane was served: rounds - that = how long the wave polled, in round trips)
static __device__ __forceinline__ int wn_ap_spin4(unsigned off, const wn_u64* p0, const wn_u64* p1, const wn_u64* p2, const wn_u64* p3, uint32_t tag, int rounds,
                                                   float& sum, int& ok) {
    float t0;
    long long m;
    int cnt;
    asm volatile(WN_AP_SPIN4_BODY
                 : [sum] "+v"(sum), [ok] "+v"(ok), [t0] "=&v"(t0), [m] "=&s"(m), [cnt] "=&s"(cnt)
                 : [off] "v"(off), [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2), [p3] "s"(p3), [tag] "s"(tag), [rounds] "s"(rounds)
                 : WN_AP_CLOBBERS);
    return cnt;
}
// the two-partial form (layer split 2): A = v[152:155], B = v[160:163]; fixed order (0 + x0) + x1
#define WN_AP_CHECK2(B0, B1, B2, B3)                     \
    "v_cmp_eq_u32_e32 vcc, %[tag], v" #B1 "\n\t"         \
    "v_cmp_eq_u32_e64 %[m], %[tag], v" #B3 "\n\t"        \
    "s_and_b64 vcc, vcc, %[m]\n\t"                       \
    "v_add_f32_e32 %[t0], 0, v" #B0 "\n\t"               \
    "v_add_f32_e32 %[t0], %[t0], v" #B2 "\n\t"           \
    WN_AP_MERGE
#define WN_AP_ISSUE2(R0, R1, R2, R3)                                             \
    "global_load_dwordx2 v[" #R0 ":" #R1 "], %[off], %[p0] sc1\n\t"               \
    "global_load_dwordx2 v[" #R2 ":" #R3 "], %[off], %[p1] sc1\n\t"
#define WN_AP_LOOK2_BODY                                                          \
        "v_mov_b32_e32 %[ok], 0\n\t"                                              \
        "v_mov_b32_e32 %[sum], 0\n\t"                                             \
        "s_waitcnt vmcnt(0)\n\t"                                                  \
        WN_AP_CHECK2(152, 153, 154, 155)
#define WN_AP_SPIN2_BODY                                                          \
        WN_AP_SGPR_HAZARD                                                         \
        "s_mov_b32 %[cnt], %[rounds]\n\t"                                         \
        WN_AP_ISSUE2(160, 161, 162, 163)                                          \
        "s_sleep 2\n\t"                                                           \
        WN_AP_ISSUE2(152, 153, 154, 155)                                          \
        WN_AP_LOOP_ALIGN                                                          \
        "1:\n\t"                                                                  \
        "s_waitcnt vmcnt(2)\n\t"                                                  \
        WN_AP_CHECK2(160, 161, 162, 163)                                          \
        "s_cbranch_vccz 2f\n\t"                                                   \
        WN_AP_ISSUE2(160, 161, 162, 163)                                          \
        "s_waitcnt vmcnt(2)\n\t"                                                  \
        WN_AP_CHECK2(152, 153, 154, 155)                                          \
        "s_cbranch_vccz 2f\n\t"                                                   \
        WN_AP_ISSUE2(152, 153, 154, 155)                                          \
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                         \
        "s_cmp_lg_u32 %[cnt], 0\n\t"                                              \
        "s_cbranch_scc1 1b\n"                                                     \
        "2:"
static __device__ __forceinline__ void wn_ap_look2(uint32_t tag, float& sum, int& ok) {
    float t0;
    long long m;
    asm volatile(WN_AP_LOOK2_BODY
                 : [sum] "=&v"(sum), [ok] "=&v"(ok), [t0] "=&v"(t0), [m] "=&s"(m)
                 : [tag] "s"(tag)
                 : WN_AP_CLOBBERS);
}
static __device__ __forceinline__ int wn_ap_spin2(unsigned off, const wn_u64* p0, const wn_u64* p1, uint32_t tag, int rounds, float& sum, int& ok) {
    float t0;
    long long m;
    int cnt;
    asm volatile(WN_AP_SPIN2_BODY
                 : [sum] "+v"(sum), [ok] "+v"(ok), [t0] "=&v"(t0), [m] "=&s"(m), [cnt] "=&s"(cnt)
                 : [off] "v"(off), [p0] "s"(p0), [p1] "s"(p1), [tag] "s"(tag), [rounds] "s"(rounds)
                 : WN_AP_CLOBBERS);
    return cnt;
}
// the single-granule form (layer 0: one complete row per stream from the sampler; layers of an unsplit stack): A = v[152:153], B = v[160:161]
#define WN_AP_LOOK1_BODY                                                          \
        "v_mov_b32_e32 %[ok], 0\n\t"                                              \
        "v_mov_b32_e32 %[sum], 0\n\t"                                             \
        "s_waitcnt vmcnt(0)\n\t"                                                  \
        WN_AP_CHECK1(152, 153)
#define WN_AP_SPIN1_BODY                                                          \
        WN_AP_SGPR_HAZARD                                                         \
        "s_mov_b32 %[cnt], %[rounds]\n\t"                                         \
        "global_load_dwordx2 v[160:161], %[off], %[p0] sc1\n\t"                      \
        "s_sleep 2\n\t"                                                           \
        "global_load_dwordx2 v[152:153], %[off], %[p0] sc1\n\t"                      \
        WN_AP_LOOP_ALIGN                                                          \
        "1:\n\t"                                                                  \
        "s_waitcnt vmcnt(1)\n\t"                                                  \
        WN_AP_CHECK1(160, 161)                                                    \
        "s_cbranch_vccz 2f\n\t"                                                   \
        "global_load_dwordx2 v[160:161], %[off], %[p0] sc1\n\t"                      \
        "s_waitcnt vmcnt(1)\n\t"                                                  \
        WN_AP_CHECK1(152, 153)                                                    \
        "s_cbranch_vccz 2f\n\t"                                                   \
        "global_load_dwordx2 v[152:153], %[off], %[p0] sc1\n\t"                      \
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                         \
        "s_cmp_lg_u32 %[cnt], 0\n\t"                                              \
        "s_cbranch_scc1 1b\n"                                                     \
        "2:"
static __device__ __forceinline__ void wn_ap_look1(uint32_t tag, float& sum, int& ok) {
    float t0;
    long long m;
    asm volatile(WN_AP_LOOK1_BODY
                 : [sum] "=&v"(sum), [ok] "=&v"(ok), [t0] "=&v"(t0), [m] "=&s"(m)
                 : [tag] "s"(tag)
                 : WN_AP_CLOBBERS);
}
static __device__ __forceinline__ int wn_ap_spin1(unsigned off, const wn_u64* p0, uint32_t tag, int rounds, float& sum, int& ok) {
    float t0;
    long long m;
    int cnt;
    asm volatile(WN_AP_SPIN1_BODY
                 : [sum] "+v"(sum), [ok] "+v"(ok), [t0] "=&v"(t0), [m] "=&s"(m), [cnt] "=&s"(cnt)
                 : [off] "v"(off), [p0] "s"(p0), [tag] "s"(tag), [rounds] "s"(rounds)
                 : WN_AP_CLOBBERS);
    return cnt;
}

// ---- The queue group's tap FIFO, hand-scheduled for the same reason.  Queue taps x[t+1-d] are requested WN_V3_TAP_AHEAD items ahead
// (rows of large-d layers miss the L2).  Held in a C++ array the FIFO is loop carried, and the compiler's wait in front of the
// OLDEST entry is a wait for the YOUNGEST one, requested one item ago: layers with d >= 128 ran 0.06-0.12 us per item slower than
// the others and set the pace of the 64-stream chain (tools/dilation_probe.py: all dilations <= 64: 930 k samples/s, cfg3: 837 k;
// three entries hand-scheduled: cfg3 881 k, d <= 128 as fast as d = 1; profiles/archive/r02_v3_tap_fifo.txt).  The loads carry sc1 (served by the
// L2, never by this CU's L1): the row was pushed by ANOTHER wave of the workgroup with a plain store as little as one item (two LDS-only
// barriers) earlier, and the L2 is where that store is ordered with this load.  The entries live in
// v152-v157 of the queue waves (the critical waves' request sets are other waves' registers of the same numbers).
// A wave's vector-memory operations per item, in program order: push store (window A), [take], tap load (window B); so the entry
// taken at item i has AHEAD-1 younger loads and AHEAD younger stores: s_waitcnt vmcnt(2 AHEAD - 1); in the first AHEAD-1 items
// fewer were issued (AHEAD + i of them), and vmcnt(AHEAD) is the safe count.
#define WN_Q_STR2(x) #x
#define WN_Q_STR(x) WN_Q_STR2(x)
#define WN_Q_ISSUE_CASE(SL, REG) \
    if constexpr (SLOT == SL) asm volatile("global_load_dword v" #REG ", %0, off sc1" ::"v"(ptr) : "v" #REG, "memory")
#define WN_Q_TAKE_CASE(SL, REG, CNT) \
    if constexpr (SLOT == SL) asm volatile("s_waitcnt vmcnt(%1)\n\tv_mov_b32_e32 %0, v" #REG : "=v"(v) : "n"(CNT) : "memory")
template <int SLOT>
static __device__ __forceinline__ void wn_q_issue(const float* ptr) {
    static_assert(WN_V3_TAP_AHEAD == 6 && SLOT >= 0 && SLOT < 6, "tap FIFO registers v152-v157");
    WN_Q_ISSUE_CASE(0, 152); WN_Q_ISSUE_CASE(1, 153); WN_Q_ISSUE_CASE(2, 154); WN_Q_ISSUE_CASE(3, 155); WN_Q_ISSUE_CASE(4, 156); WN_Q_ISSUE_CASE(5, 157);
}
// CNT = vector-memory operations of this wave that may stay in flight while the entry is read (everything younger than its load)
template <int SLOT, int CNT>
static __device__ __forceinline__ float wn_q_take() {
    float v;
    WN_Q_TAKE_CASE(0, 152, CNT); WN_Q_TAKE_CASE(1, 153, CNT); WN_Q_TAKE_CASE(2, 154, CNT);
    WN_Q_TAKE_CASE(3, 155, CNT); WN_Q_TAKE_CASE(4, 156, CNT); WN_Q_TAKE_CASE(5, 157, CNT);
    return v;
}
// the entry is a run-time (wave-uniform) value at the call sites
static __device__ __forceinline__ void wn_q_issue_slot(int slot, const float* q) {
    switch (slot) {
        case 0: wn_q_issue<0>(q); break; case 1: wn_q_issue<1>(q); break; case 2: wn_q_issue<2>(q); break;
        case 3: wn_q_issue<3>(q); break; case 4: wn_q_issue<4>(q); break; default: wn_q_issue<5>(q); break;
    }
}
template <int CNT>
static __device__ __forceinline__ float wn_q_take_slot(int slot) {
    switch (slot) {
        case 0: return wn_q_take<0, CNT>(); case 1: return wn_q_take<1, CNT>(); case 2: return wn_q_take<2, CNT>();
        case 3: return wn_q_take<3, CNT>(); case 4: return wn_q_take<4, CNT>(); default: return wn_q_take<5, CNT>();
    }
}

// Barrier whose "did any wave give up?" word is read but not yet looked at: the LDS read is issued with the reads that follow the
// barrier and the caller branches on it at the END of the window -- wn_barrier_failed branches at once, and that puts an LDS round
// trip (~0.03 us) between every barrier and the first instruction of the critical path, twice per item and hop.
static __device__ __forceinline__ int wn_barrier_flag(WnCtx& cx, int* flag) {
    if (cx.fail) *flag = 1;
    wn_lds_barrier();
    return *flag;
}

template <class SH, int G = 1>
struct WnV3Lds {
    using M = WnV2LdsM<SH, G>;  // G streams per pipeline item: x, z, taps, the head's staging areas are [G][...]
    static constexpr int XR = M::XR, SKP = M::SKP, DCP = M::DCP;
    static constexpr int xs = M::xs, zs = M::zs, xo = M::xo, sk = M::sk, ev = M::ev, smp = M::smp, park = M::park;
    static constexpr int pre = M::pre;        // [n_streams][256]
    static __host__ __device__ int floats(int n_streams) { return pre + n_streams * 256; }
};

// Shapes whose roles fit the 168 VGPRs a 768-thread workgroup leaves each lane: the skip group holds RS*DC floats, the critical
// group K1 + K2, a head workgroup its end_conv_1 slice + end_conv_2 rows (K3 + EC); the rest is working set.
template <class SH, int P>
static constexpr bool wn_v3_fits() {
    return (P == 1 || P == 2 || P == 4) && SH::RS * SH::DC <= 100 && SH::K1 + SH::K2 <= 80 && (SH::K3 <= 100 || (SH::K3 <= 128 && SH::EC <= 32)) && SH::K3 % 4 == 0 && SH::EC % 4 == 0 &&
           SH::DC % 4 == 0 && SH::R % 4 == 0;
}
// ... and the two-streams-per-item form: a critical lane takes the filter AND the gate row of a channel on half an x slice, read as float4
template <class SH>
static constexpr bool wn_v3_g2_fits() { return SH::K1 % 8 == 0 && 2 * SH::T1 <= 16 && 2 * SH::R <= 256; }

// dot of one register weight vector with the G vectors x + g * xstride in LDS (the G streams of an item share every weight operand;
// per stream the arithmetic is wn_dot_lds's: two packed chains, same summation order -- bit-identical whatever G)
template <int K, int G>
static __device__ __forceinline__ void wn_dot_lds_gp(const float (&w)[K], const float* x, int xstride, const float (&init)[G], float (&out)[G]) {
    if constexpr (K % 4 == 0) {
        float4 v[G][K / 4];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < K / 4; ++k) v[g][k] = reinterpret_cast<const float4*>(x + g * xstride)[k];
        wn_f2 a01[G], a23[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { a01[g] = wn_f2{init[g], 0.f}; a23[g] = wn_f2{0.f, 0.f}; }
#pragma unroll
        for (int k = 0; k < K / 4; ++k)
#pragma unroll
            for (int g = 0; g < G; ++g) {  // (the streams' chains interleaved in program order)
                a01[g] = __builtin_elementwise_fma(wn_f2{w[4 * k], w[4 * k + 1]}, wn_f2{v[g][k].x, v[g][k].y}, a01[g]);
                a23[g] = __builtin_elementwise_fma(wn_f2{w[4 * k + 2], w[4 * k + 3]}, wn_f2{v[g][k].z, v[g][k].w}, a23[g]);
            }
#pragma unroll
        for (int g = 0; g < G; ++g) out[g] = (a01[g].x + a01[g].y) + (a23[g].x + a23[g].y);
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) out[g] = wn_dot_lds<K>(w, x + g * xstride, init[g]);
    }
}

template <class SH, int P, int G, int SK = 0>
static __device__ void wn_v3_layer(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds, int l, int c) {
    constexpr int R = SH::R, DC = SH::DC, S = SH::S, T1 = SH::T1, K1 = SH::K1, T2 = SH::T2, K2 = SH::K2, RS = SH::RS;
    using L = WnV3Lds<SH, G>;
    static_assert(G >= 1 && G * R <= 256, "the G streams of an item are polled / pushed by G*R lanes");
    const int tid = threadIdx.x, t = tid & 255;
    const int group = tid >> 8;  // wave-uniform: 0 critical, 1 skip, 2 queue
    const int ns = p.n_streams, NL = p.NL;
    const int nI = ns / G;  // pipeline items per evaluation: item (e, s) carries the streams s .. s+G-1 (s a multiple of G)
    // lanes t < G*R handle element tr of stream s + tg of the item (hand-off granules and queue rows of consecutive streams are
    // R apart: their index is simply + t)
    const int tg = t < G * R ? t / R : 0, tr = t < G * R ? t - tg * R : 0;
    const float* img = p.blobs + (size_t)cx.w * (SH::NWL * 256) + t;  // image rows: w1[K1] w0[K1] w2[K2] w3[RS][DC] bfg bres bskip[RS]
    const int kq1 = t % T1, grp = t / T1, ch = grp >> 1, is_gate = grp & 1;
    const int kq2 = t % T2, row2 = t / T2;
    const int d = p.dil[l];
    const int ML = d + 1;
    float* xs = lds + L::xs;
    float* zs = lds + L::zs;
    float* xol = lds + L::xo;
    float* pre = lds + L::pre;
    float* smp = lds + L::smp;
    int* failflag = reinterpret_cast<int*>(smp + 48);
    int* locflags = reinterpret_cast<int*>(smp + 52);
    if (tid == 0) {
        *failflag = 0;
        const int mine = wn_xcc_id();
        __hip_atomic_store(p.xcc_tab + cx.w, (unsigned)(mine + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int lx = 0, lsk = 0, lup = 0, lcons = 0;
        if (p.allow_plain) {
            if (l < NL - 1) {
                lx = wn_same_xcd(cx, mine, (l + 1) * P, P);
                lsk = wn_same_xcd(cx, mine, (l + 1) * P + c, 1);
                if (l < NL - 2) lcons = lsk && wn_same_xcd(cx, mine, (l + 2) * P + c, 1);   // ... and the consumer's consumer (slot re-use, skip group)
            } else {
                lsk = wn_same_xcd(cx, mine, NL * P, p.PA * p.HR);
            }
            if (l > 0) lup = wn_same_xcd(cx, mine, (l - 1) * P + c, 1);
        }
        locflags[0] = lx; locflags[1] = lsk; locflags[2] = lup; locflags[3] = lcons;
    }
    __syncthreads();
    const bool local_x = locflags[0] != 0, local_s = locflags[1] != 0;
    const int n_prime = (int)(r.n_given - 1);

    // ---- fetching the layer's input (lanes t < G*R of the critical group): request (set A), first look, spin with two sets in flight, stage.
    // (Round 3 also tried the SKIP group as the fetcher -- request after barrier B, look after its chunk, the critical group pure compute --
    //  and the skip group's chunk deferred into the next item's filter/gate window so that its polling waves are free between B and the
    //  next A: profiles/archive/r03_skip_group_polls.txt.  Slower wherever tokens queue: the request just misses the token, the next look comes a
    //  chunk later; and ANY work next to the critical group's filter/gate window doubles that window -- the LDS pipe is what both wait for.)
    // (Round 6 tried barrier A as an LDS counter only its consumers wait for -- the skip and queue groups free to run their chunk anywhere between
    //  B(i) and B(i + 1): one stream per item 64 streams 0.910 -> 0.904 M, 32 streams 0.657 -> 0.573 M; two streams per item 1.092 -> 0.872 M: the
    //  stage's cycle is NOT the critical window plus the longest chunk, and four waves spinning on an LDS word cost more than the hardware barrier
    //  they replace: profiles/r06_soft_barrier_a_experiment.txt, the patch next to it.)
    // the layer's input granules of stream s2: layer 0 ONE complete row per stream (g0), layers > 0 the P partials of the upstream slices
    static_assert(P == 1 || P == 2 || P == 4, "the hand-scheduled input poll is written for one, two and four partials");
    const bool poller = t < G * R;  // (the polling blocks run under this lane mask: a partly filled wave leaves when ITS active lanes are served)
    const bool one = l == 0 || P == 1;  // ONE granule per lane: the sampler's complete row (layer 0), the only slice of an unsplit stack
    const wn_u64* xb0 = l == 0 ? p.g0 : p.gx + ((size_t)(l - 1) * P) * ns * R;  // (wave-uniform) partial j: + j * xstep_j
    const size_t xstep_j = (size_t)ns * R;
    const unsigned xlane = (poller ? (unsigned)t : 0u) * 8u;  // this lane's byte offset inside the granules of the item's first stream
    long long* park4 = reinterpret_cast<long long*>(smp + 56);
    auto request = [&](int s2) {  // set A for the item whose first stream is s2
        const unsigned off = xlane + (unsigned)s2 * (unsigned)(R * 8);
        if (one) wn_ap_issue_a1(off, xb0);
        else if constexpr (P == 2) wn_ap_issue_a2(off, xb0, xb0 + xstep_j);
        else wn_ap_issue_a4(off, xb0, xb0 + xstep_j, xb0 + 2 * xstep_j, xb0 + 3 * xstep_j);
    };
    auto look = [&](uint32_t tag2, float& sum, int& ok) {
        if (one) wn_ap_look1(tag2, sum, ok);
        else if constexpr (P == 2) wn_ap_look2(tag2, sum, ok);
        else wn_ap_look4(tag2, sum, ok);
    };
    // (Round 3 tried a pre-poll sleep here -- a polling wave sleeping through a fraction of its predicted wait before it requests again, to
    //  take the chain's own polls off the L2s: 3/8 of the wait changed nothing (64 streams 994 k against 991 k), 4/8 and more collapse --
    //  a token that waits for a sleeping stage lengthens every later wait, which lengthens the sleep: profiles/archive/r03_presleep.txt.)
    // spins until every lane of this wave has its input (bounded like wn_poll_fixed), then stages it
    auto finish_input = [&](long long e2, int s2, float* xb2, float sum, int ok, long long item2) {
        const uint32_t tag2 = (uint32_t)(e2 + 1);
        const unsigned off = xlane + (unsigned)s2 * (unsigned)(R * 8);
        unsigned spins = 0;
        while (!cx.fail && __builtin_amdgcn_ballot_w64(ok == 0) != 0) {  // the wave leaves together (its lanes share the barrier that follows)
            if (one) wn_ap_spin1(off, xb0, tag2, 64, sum, ok);
            else if constexpr (P == 2) wn_ap_spin2(off, xb0, xb0 + xstep_j, tag2, 64, sum, ok);
            else wn_ap_spin4(off, xb0, xb0 + xstep_j, xb0 + 2 * xstep_j, xb0 + 3 * xstep_j, tag2, 64, sum, ok);
            if (__builtin_amdgcn_ballot_w64(ok == 0) == 0) break;
            // slow path: ~64 double rounds between looks at the abort word and the wall clock
            if (__hip_atomic_load(cx.p->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; break; }
            const long long now = (long long)wall_clock64();
            if (spins++ == 0u) cx.t_start = now;
            else if (now - cx.t_start > cx.r->timeout_ticks) { wn_give_up(cx, l == 0 ? WN_W_LOGITS : WN_W_X, e2, s2); break; }
            look(tag2, sum, ok);  // (set A was re-requested at the end of the spin)
        }
        xb2[tg * L::XR + SH::xpad(tr)] = sum;
        if (r.prof && item2 < r.prof_items && (tid & 255) == 0) park4[item2 & 1] = (long long)wall_clock64();
    };
    // the first evaluation's input of layer 0 is a given sample: start_conv row gather (wavenet_model.py:127, 256-257)
    auto given_input = [&](int s2, float* xb2) {
        const int idx = r.first[(size_t)(s2 + tg) * r.n_given];
        xb2[tg * L::XR + SH::xpad(tr)] = p.start_t[(size_t)idx * R + tr] + (p.start_b ? p.start_b[tr] : 0.f);
    };
    if (group == 0) {
        // ================================================================== critical group
#if WN_V3_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        // Filter/gate lanes.  The register images give lane (row, kq1) the K1 tap-1 weights of ONE filter or gate row on x slice kq1;
        // the 2 DC / T1-lane groups of a wave then read the same x slices from LDS, 8 x 16 bytes per lane and stream -- and the LDS
        // pipe (128 bytes a clock for the whole CU), not the FMA issue, is what the filter/gate window waits for.  A critical lane
        // therefore takes BOTH rows of a channel on HALF a slice: lane (ch8, kq8), 2 T1 lanes per channel, K1 / 2 x values per lane
        // (half the LDS reads per row pair), each feeding one packed FMA {filter, gate}; its weights are those of the images' lanes
        // (2 ch8, kq8 / 2) and (2 ch8 + 1, kq8 / 2), rows (kq8 & 1) K1/2 ..., gathered here once.
        constexpr int T8 = 2 * T1, K8 = K1 / 2;
        constexpr bool PAIR = WN_V3_PAIR_ROWS == 1 || (WN_V3_PAIR_ROWS == 2 && G >= 2);
        static_assert(!PAIR || (K1 % 8 == 0 && T8 <= 16), "half slices are read as float4");
        const int ch8 = t / T8, kq8 = t % T8, half8 = kq8 & 1;
        const int t_f = (2 * ch8) * T1 + kq8 / 2;  // the image lane of the channel's filter row (gate row: + T1)
        float w1[PAIR ? 1 : K1], w2[K2];
        wn_f2 wfg[PAIR ? K8 : 1];
        constexpr bool KPACK = PAIR && WN_V3_KPACK != 0;
        if constexpr (KPACK) {   // wfg[2 j] = {w_f[2 j], w_f[2 j + 1]}, wfg[2 j + 1] = {w_g[2 j], w_g[2 j + 1]} of this lane's half slice
            const float* imw = p.blobs + (size_t)cx.w * (SH::NWL * 256);
#pragma unroll
            for (int j = 0; j < K8 / 2; ++j) {
                wfg[2 * j] = wn_f2{imw[(size_t)(half8 * K8 + 2 * j) * 256 + t_f], imw[(size_t)(half8 * K8 + 2 * j + 1) * 256 + t_f]};
                wfg[2 * j + 1] = wn_f2{imw[(size_t)(half8 * K8 + 2 * j) * 256 + t_f + T1], imw[(size_t)(half8 * K8 + 2 * j + 1) * 256 + t_f + T1]};
            }
        } else if constexpr (PAIR) {
            const float* imw = p.blobs + (size_t)cx.w * (SH::NWL * 256);
#pragma unroll
            for (int k = 0; k < K8; ++k) wfg[k] = wn_f2{imw[(size_t)(half8 * K8 + k) * 256 + t_f], imw[(size_t)(half8 * K8 + k) * 256 + t_f + T1]};
        } else {
#pragma unroll
            for (int k = 0; k < K1; ++k) w1[k] = img[(size_t)k * 256];
        }
#pragma unroll
        for (int k = 0; k < K2; ++k) w2[k] = img[(size_t)(2 * K1 + k) * 256];
        const float bres = img[(size_t)(2 * K1 + K2 + RS * DC + 1) * 256];
        // per-lane constants of the gate: the lane with the first half of a slice evaluates the filter factor 2 s(2 f) - 1, its neighbour the gate factor s(g)
        const float gate_c = (PAIR ? half8 : is_gate) ? -1.44269504088896341f : -2.88539008177792681f;   // -log2 e, -2 log2 e
        const float pre_m = half8 ? 0.f : 1.f;   // (the lane that takes the parked tap-0 sums of its slice)
        const float res_m = (c == 0 && kq2 == 0) ? 1.f : 0.f;   // (the lane that adds x[t] to its row of the residual partial: wavenet_model.py:165)
        long long* park = reinterpret_cast<long long*>(lds + L::park);
        const __amdgpu_buffer_rsrc_t rs_gx = wn_rsrc(p.gx);
        if (poller) request(0);
        int buf = 0;
        for (long long e = 0; e < r.n_eval; ++e) {
            const uint32_t tag = (uint32_t)(e + 1);
            for (int s = 0; s < ns; s += G, buf ^= 1) {
                float* xb = xs + buf * (G * L::XR);  // [G][XR]
                const long long item = e * nI + s / G;
                wn_stamp(r, park, item, 0);
                // ---- 1. layer input x[t] of the item's G streams
                if (l == 0 && e == 0) { if (poller) given_input(s, xb); }
                else if (poller) {  // (set A was requested at the end of the previous item)
                    float sum;
                    int ok;
                    look(tag, sum, ok);
                    finish_input(e, s, xb, sum, ok, item);
                }
                const int fail_a = wn_barrier_flag(cx, failflag);  // ---- A(i): x staged
                wn_stamp(r, park, item, 1);
                // ---- 2. filter/gate: tap 1 on x[t] + parked tap 0, tanh * sigmoid   (wavenet_model.py:147-151)
                // (the G streams' chains are kept in ONE basic block -- unconditional LDS reads, selects instead of lane-predicated
                //  branches, the z stores after both chains -- so that the scheduler can interleave them: a predicated store between
                //  them made the second stream's whole chain wait for the first one's)
                float xres[G];   // x[t] of this lane's row: the residual add (slice 0's lane kq2 == 0 adds it: res_m), one FMA behind the residual dot
#pragma unroll
                for (int g = 0; g < G; ++g) xres[g] = xb[g * L::XR + SH::xpad(row2)];
                if constexpr (KPACK) {
                    // lane (ch8, kq8): filter row AND gate row of channel ch8 on half a slice of x, K8 elements: per float4 of x two packed FMAs per row
                    // ({x0, x1} and {x2, x3} against the rows' consecutive weights) -- four chains per stream, the streams interleaved
                    wn_f2 f01[G], f23[G], g01[G], g23[G];
                    float pf[G], pg[G];
                    const float* prs = pre + s * 256 + t_f;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        pf[g] = prs[g * 256]; pg[g] = prs[g * 256 + T1];   // parked per image lane (row, kq1): taken by the lane with the first half of slice kq1
                        f01[g] = f23[g] = g01[g] = g23[g] = wn_f2{0.f, 0.f};
                    }
                    float4 v[G][K8 / 4];
                    const float* xsl = xb + (kq8 / 2) * (K1 + 4) + half8 * K8;
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int k = 0; k < K8 / 4; ++k) v[g][k] = reinterpret_cast<const float4*>(xsl + g * L::XR)[k];
#pragma unroll
                    for (int k = 0; k < K8 / 4; ++k)
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            f01[g] = __builtin_elementwise_fma(wfg[4 * k], wn_f2{v[g][k].x, v[g][k].y}, f01[g]);
                            g01[g] = __builtin_elementwise_fma(wfg[4 * k + 1], wn_f2{v[g][k].x, v[g][k].y}, g01[g]);
                            f23[g] = __builtin_elementwise_fma(wfg[4 * k + 2], wn_f2{v[g][k].z, v[g][k].w}, f23[g]);
                            g23[g] = __builtin_elementwise_fma(wfg[4 * k + 3], wn_f2{v[g][k].z, v[g][k].w}, g23[g]);
                        }
                    float z[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const wn_f2 fs = f01[g] + f23[g], gs = g01[g] + g23[g];
                        const float f = wn_reduce<T8>(fmaf(pf[g], pre_m, fs.x + fs.y)), gt = wn_reduce<T8>(fmaf(pg[g], pre_m, gs.x + gs.y));
                        const float rc = __builtin_amdgcn_rcpf(1.0f + (WN_V3_FAST_GATE ? __builtin_amdgcn_exp2f((half8 ? gt : f) * gate_c) : wn_exp(half8 ? -gt : -2.0f * f)));
                        const float fac = half8 ? rc : fmaf(2.0f, rc, -1.0f);
                        z[g] = fac * wn_partner<1>(fac);  // (the DPP move outside any lane-dependent branch)
                    }
                    if (kq8 == 0) {
#pragma unroll
                        for (int g = 0; g < G; ++g) zs[g * L::DCP + ch8] = z[g];
                    }
                } else if constexpr (PAIR) {
                    // the parked tap-0 sums are per image lane (row, kq1): the lane with the first half of slice kq1 takes both rows' sums
                    wn_f2 a0[G], a1[G], a2[G], a3x[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float pf = pre[(s + g) * 256 + t_f], pg = pre[(s + g) * 256 + t_f + T1];
                        a0[g] = half8 ? wn_f2{0.f, 0.f} : wn_f2{pf, pg};
                        a1[g] = a2[g] = a3x[g] = wn_f2{0.f, 0.f};
                    }
                    float4 v[G][K8 / 4];
                    const float* xsl = xb + (kq8 / 2) * (K1 + 4) + half8 * K8;
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int k = 0; k < K8 / 4; ++k) v[g][k] = reinterpret_cast<const float4*>(xsl + g * L::XR)[k];
#pragma unroll
                    for (int k = 0; k < K8 / 4; ++k)
#pragma unroll
                        for (int g = 0; g < G; ++g) {  // two (or four) chains of packed {filter, gate} FMAs per stream, the streams interleaved
                            a0[g] = __builtin_elementwise_fma(wfg[4 * k], wn_f2{v[g][k].x, v[g][k].x}, a0[g]);
                            a1[g] = __builtin_elementwise_fma(wfg[4 * k + 1], wn_f2{v[g][k].y, v[g][k].y}, a1[g]);
                            if constexpr (WN_V3_FG_CHAINS == 4) {
                                a2[g] = __builtin_elementwise_fma(wfg[4 * k + 2], wn_f2{v[g][k].z, v[g][k].z}, a2[g]);
                                a3x[g] = __builtin_elementwise_fma(wfg[4 * k + 3], wn_f2{v[g][k].w, v[g][k].w}, a3x[g]);
                            } else {
                                a0[g] = __builtin_elementwise_fma(wfg[4 * k + 2], wn_f2{v[g][k].z, v[g][k].z}, a0[g]);
                                a1[g] = __builtin_elementwise_fma(wfg[4 * k + 3], wn_f2{v[g][k].w, v[g][k].w}, a1[g]);
                            }
                        }
                    if constexpr (WN_V3_FG_CHAINS == 4) {
#pragma unroll
                        for (int g = 0; g < G; ++g) { a0[g] = a0[g] + a2[g]; a1[g] = a1[g] + a3x[g]; }
                    }
                    // tanh(f) * sigmoid(g): even lanes of the channel's group evaluate the filter factor 2 sigmoid(2f) - 1, odd lanes the gate
                    // factor sigmoid(g) (one exp and one reciprocal per lane), and each takes the other from its neighbour
                    float z[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float f = wn_reduce<T8>(a0[g].x + a1[g].x), gt = wn_reduce<T8>(a0[g].y + a1[g].y);
                        const float rc = __builtin_amdgcn_rcpf(1.0f + (WN_V3_FAST_GATE ? __builtin_amdgcn_exp2f((half8 ? gt : f) * gate_c) : wn_exp(half8 ? -gt : -2.0f * f)));
                        const float fac = half8 ? rc : fmaf(2.0f, rc, -1.0f);
                        z[g] = fac * wn_partner<1>(fac);  // (the DPP move outside any lane-dependent branch)
                    }
                    if (kq8 == 0) {
#pragma unroll
                        for (int g = 0; g < G; ++g) zs[g * L::DCP + ch8] = z[g];
                    }
                } else {
                    float pin[G], acc[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) pin[g] = pre[(s + g) * 256 + t];
                    wn_dot_lds_gp<K1, G>(w1, xb + kq1 * (K1 + 4), L::XR, pin, acc);
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g] = wn_reduce<T1>(acc[g]);
                    // tanh(f) * sigmoid(g) as in wn_gate, but each lane evaluates only ITS factor (filter lanes 2 sigmoid(2f) - 1, gate
                    // lanes sigmoid(g): one exp and one reciprocal instead of two each) and takes the other from its partner row; the
                    // product is the same two numbers multiplied: bit-identical
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float rc = __builtin_amdgcn_rcpf(1.0f + (WN_V3_FAST_GATE ? __builtin_amdgcn_exp2f(acc[g] * gate_c) : wn_exp(is_gate ? -acc[g] : -2.0f * acc[g])));
                        const float fac = is_gate ? rc : fmaf(2.0f, rc, -1.0f);
                        const float z = fac * wn_partner<T1>(fac);  // (the DPP move outside any lane-dependent branch)
                        if (!is_gate && kq1 == 0) zs[g * L::DCP + ch] = z;
                    }
                }
                if (fail_a) return;
                const int fail_b = wn_barrier_flag(cx, failflag);  // ---- B(i): z staged
                wn_stamp(r, park, item, 5);
                // ---- 3. residual 1x1 partial, published at once                      (wavenet_model.py:164-165)
                if (l < NL - 1) {
                    float a2[G], zero[G], xn[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) zero[g] = 0.f;
                    wn_dot_lds_gp<K2, G>(w2, zs + kq2 * K2, L::DCP, zero, a2);
#pragma unroll
                    for (int g = 0; g < G; ++g) xn[g] = fmaf(xres[g], res_m, wn_reduce<T2>(a2[g]) + bres);  // valid on the kq2 == 0 lane of every row
                    if constexpr (T2 == 2) {
                        // rows 2j and 2j+1 sit on lanes 4j and 4j+2: one 16-byte store {x'(2j), tag, x'(2j+1), tag} by lane 4j instead of
                        // two 8-byte stores (write-through stores are retired per lane; consumers keep reading their own 8-byte half)
                        float xn1[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) xn1[g] = wn_dpp<0x4E>(xn[g]);  // quad_perm [2,3,0,1]
                        if ((t & 3) == 0) {
#pragma unroll
                            for (int g = 0; g < G; ++g) wn_st_pair(rs_gx, (unsigned)((((size_t)cx.w * ns + s + g) * R + row2) * 8), tag, xn[g], xn1[g], local_x);
                        }
                    } else {
                        if (kq2 == 0) {
#pragma unroll
                            for (int g = 0; g < G; ++g) wn_publish_at(p.gx + ((size_t)cx.w * ns + s + g) * R + row2, tag, xn[g], local_x);
                        }
                    }
                }
                wn_stamp(r, park, item, 2);
                wn_stamp(r, park, item, 3);
                if (r.prof && item < r.prof_items && tid == 0) {  // slots 0-5 (6 and 7 belong to the skip and queue groups)
                    long long* dst = r.prof + ((size_t)cx.w * r.prof_items + item) * WN_STAMPS;
#pragma unroll
                    for (int k = 0; k < 6; ++k) dst[k] = k == 4 ? park4[item & 1] : park[k];
                }
                if (poller) request(s + G < ns ? s + G : 0);  // set A of the coming item
                if (fail_b) return;
            }
        }
        if (wn_barrier_failed(cx, failflag)) return;  // A(N), B(N): the tail group's last chunk 2 runs between them
        (void)wn_barrier_failed(cx, failflag);
        return;
    }

    if (group == 1) {
        // ================================================================== skip group
#if WN_V3_LAST_SKIP_PRIO
        // the LAST layer's skip lanes are the head's input -- the one place where this group is on the token's critical path (and the
        // critical group of that layer has nothing to do after barrier B: it publishes no x')
        if (G >= 2 && l == NL - 1) __builtin_amdgcn_s_setprio(WN_V3_LAST_SKIP_PRIO);
#endif
        // A lane owns the rows t + 256 q (q < RS) of this slice's lane of the running skip sum.  Rows 2h and 2h+1 of a lane sit side by
        // side: one packed FMA (v_pk_fma_f32) per z element and row pair, one 16-byte hand-off {v(t + 512 h), tag, v(t + 512 h + 256), tag} at
        // byte 4096 h + 16 t of the lane.  An odd RS (S = 256: one row per lane) leaves a last row without a partner in its lane: its dot runs
        // as two half-length chains (even / odd channels, packed), and the last 256 rows of the lane are kept in natural order (row
        // 256 (RS-1) + t at byte 2048 (RS-1) + 8 t) and handed over as the pairs of NEIGHBOURING lanes -- the even lane stores both, every
        // lane loads the pair that holds its row.
        constexpr int NPL = RS / 2, ODD = RS % 2, NPR = NPL + ODD;
        static_assert(DC % 4 == 0 && L::DCP % 4 == 0, "z is read as float4");
        wn_f2 w3p[NPL > 0 ? NPL : 1][DC];
        wn_f2 w3o[ODD ? DC / 2 : 1];  // the unpaired row: {w[2k], w[2k+1]}
        float bskip[RS];
#pragma unroll
        for (int h2 = 0; h2 < NPL; ++h2)
#pragma unroll
            for (int k = 0; k < DC; ++k)
                w3p[h2][k] = wn_f2{img[(size_t)(2 * K1 + K2 + (2 * h2) * DC + k) * 256], img[(size_t)(2 * K1 + K2 + (2 * h2 + 1) * DC + k) * 256]};
        if constexpr (ODD) {
#pragma unroll
            for (int k = 0; k < DC / 2; ++k)
                w3o[k] = wn_f2{img[(size_t)(2 * K1 + K2 + (RS - 1) * DC + 2 * k) * 256], img[(size_t)(2 * K1 + K2 + (RS - 1) * DC + 2 * k + 1) * 256]};
        }
#pragma unroll
        for (int q = 0; q < RS; ++q) bskip[q] = img[(size_t)(2 * K1 + K2 + RS * DC + 2 + q) * 256];
        const __amdgpu_buffer_rsrc_t rs_gs = wn_rsrc(p.gs);
        const size_t up_wg = (size_t)(l > 0 ? l - 1 : 0) * P + c;  // the upstream slice (l > 0)
        constexpr unsigned SB = (unsigned)S * 8;                               // bytes between the lanes of consecutive streams
        constexpr unsigned ODD_BASE = 2048u * (unsigned)(RS - 1);              // byte offset of the natural-order tail inside a lane
        const unsigned lane16 = (unsigned)t * 16, odd_ld = ODD_BASE + (unsigned)(t & ~1) * 8, odd_st = ODD_BASE + (unsigned)t * 8;
        // ---- Slot re-use.  A lane is S granules per STREAM: 50 layers x 4 slices x 64 streams x 4 KB = 52 MB that is written once and read once per
        // timestep -- it never fits the 4 MB of an XCD's L2, so every line goes out to the fabric once (rocprofv3 PMC: WRITE_SIZE 142 GB per 2000
        // timesteps, 2/3 of the job's traffic, 3.3x its algorithmic bytes) although producer and consumer sit on the same XCD.  Where they do
        // (same XCD: the store is L2-resident anyway), the lane of pipeline ITEM i goes into slot i mod NSLOT instead: NSLOT x G x 4 KB per
        // workgroup stay dirty in the L2 and are overwritten there.  A slot may only be overwritten once its reader is done with it; the reader
        // (the next layer's skip group, same slice) publishes ITS item j after it has read ours, so the producer of item i looks at one pair
        // of the reader's own lane of item i - NSLOT before it stores (requested a window ahead, next to the upstream request: normally fresh).
        // Tags of re-used slots count items (item + 1), the others evaluations (e + 1) as before.  The last layer's lanes (read by the head
        // replicas, which publish no lane) and every lane that crosses an XCD boundary keep one slot per stream.
#ifndef WN_V3_SLOT_DEBUG
#define WN_V3_SLOT_DEBUG 0   // timing experiments (-DWN_EXPERIMENT): 1 = re-used slots WITHOUT the back-pressure look (unsafe), 2 = the look without re-use
#endif
        constexpr int NSLOT_C = SK > 0 ? SK : WN_V3_SKIP_SLOTS;   // (SK: the form the host picks for cfg3 from 96 streams up, wn_v3_slots_for)
        const int NSLOT = NSLOT_C > 0 && NSLOT_C * G <= ns ? NSLOT_C : 0;
        const bool slot_look = NSLOT > 0 && l < NL - 1 && local_s && WN_V3_SLOT_DEBUG != 1;
        const bool slot_out = NSLOT > 0 && l < NL - 1 && local_s && WN_V3_SLOT_DEBUG != 2;                 // my lane: re-used slots
        const bool slot_in = NSLOT > 0 && l > 0 && locflags[2] != 0 && WN_V3_SLOT_DEBUG != 2;              // the upstream's lane (its slot_out: the same relation seen from the other side)
        const bool slot_cons = NSLOT > 0 && l < NL - 2 && locflags[3] != 0;       // the reader's OWN lane (where its progress is visible)
        const size_t cons_wg = (size_t)(l < NL - 1 ? l + 1 : l) * P + c;
        int slot = 0;   // item mod NSLOT
        if (wn_barrier_failed(cx, failflag)) return;  // A(0)
        // The upstream skip lane of the coming item is requested right after barrier A: its producer published it a little after
        // the x' this workgroup has just consumed, so the load returns it, and its round trip runs next to the critical group's
        // filter/gate dot instead of inside this group's chunk after barrier B (requested at B the chunk took 0.57 us at 64 streams,
        // nearly as long as the critical group needs from B to the next A).  One item ahead it would come back stale in the
        // latency-bound regime.
        wn_v4i sk_req[G][NPR];
        wn_v4i ck_req = {0, 0, 0, 0};   // one pair of the reader's lane of the item whose slot the coming item re-uses
        // where the reader's lane of item (ej, sj) lives: its slot (the same index as ours: both count items) or its stream
        auto check_off = [&](int slot2, int sj) -> unsigned {
            return (unsigned)(((cons_wg * ns + (slot_cons ? slot2 * G : sj)) * (size_t)S) * 8) + (NPL > 0 ? lane16 : odd_ld);
        };
        auto request_up = [&](int s2, int slot2) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const unsigned base = (unsigned)(((up_wg * ns + (slot_in ? slot2 * G : s2) + g) * (size_t)S) * 8);
#pragma unroll
                for (int h2 = 0; h2 < NPL; ++h2) sk_req[g][h2] = __builtin_amdgcn_raw_buffer_load_b128(rs_gs, lane16, base + h2 * 4096, 16);
                if constexpr (ODD) sk_req[g][NPL] = __builtin_amdgcn_raw_buffer_load_b128(rs_gs, odd_ld, base, 16);
            }
            if (slot_look) {
                const int sj = s2 - NSLOT * G < 0 ? s2 - NSLOT * G + ns : s2 - NSLOT * G;
                ck_req = wn_ld_pair(rs_gs, check_off(slot2, sj));
            }
        };
        request_up(0, 0);
        long long item = 0;
        for (long long e = 0; e < r.n_eval; ++e) {
            const bool prime = e < n_prime;
            const uint32_t tag = (uint32_t)(e + 1);
            for (int s = 0; s < ns; s += G, ++item) {
                if (wn_barrier_failed(cx, failflag)) return;  // ---- B(i): z of this item staged
                if (WN_V3_SKIP_DEFER > 0 && G >= 2 && l < NL - 1) __builtin_amdgcn_s_sleep(WN_V3_SKIP_DEFER);
                const bool stamp = r.prof && item < r.prof_items && tid == 256;
                const long long t0 = stamp ? (long long)wall_clock64() : 0;
                const int s2 = s + G < ns ? s + G : 0;  // the coming item's first stream
                const int slot2 = NSLOT > 0 && slot + 1 < NSLOT ? slot + 1 : 0;
                const uint32_t tag_in = slot_in ? (uint32_t)(item + 1) : tag, tag_out = slot_out ? (uint32_t)(item + 1) : tag;
                const unsigned base_up = (unsigned)(((up_wg * ns + (slot_in ? slot * G : s)) * (size_t)S) * 8);    // upstream slice's lane, first stream of the item
                const unsigned base_me = (unsigned)((((size_t)cx.w * ns + (slot_out ? slot * G : s)) * (size_t)S) * 8);
                // ---- skip 1x1 partial on this lane of the running skip sum          (wavenet_model.py:154-162)
                float a3[G][RS];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int q = 0; q < RS; ++q) a3[g][q] = 0.f;
                const bool work = !prime && !(WN_V3_ABL & 1);
                if (work) {
                    // per row: bias, then + w[k] z[k] for k = 0..DC-1 in order (one fused multiply-add each), as before the packing
                    constexpr int SC = WN_V3_SKIP_CHAINS;
                    static_assert(SC == 1 || SC == 2 || SC == 4, "chains per row pair");
                    wn_f2 a3c[G][NPL > 0 ? NPL : 1][SC], a3o[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
#pragma unroll
                        for (int h2 = 0; h2 < NPL; ++h2) {
                            a3c[g][h2][0] = wn_f2{bskip[2 * h2], bskip[2 * h2 + 1]};
#pragma unroll
                            for (int c2 = 1; c2 < SC; ++c2) a3c[g][h2][c2] = wn_f2{0.f, 0.f};
                        }
                        a3o[g] = wn_f2{ODD ? bskip[RS - 1] : 0.f, 0.f};
                    }
                    float4 z4[G][DC / 4];
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int k = 0; k < DC / 4; ++k) z4[g][k] = reinterpret_cast<const float4*>(zs + g * L::DCP)[k];
#pragma unroll
                    for (int k = 0; k < DC / 4; ++k) {
#pragma unroll
                        for (int g = 0; g < G; ++g) {  // (the streams' chains interleaved in program order)
#pragma unroll
                            for (int h2 = 0; h2 < NPL; ++h2) {   // (chain j takes the z elements 4 k + j', j' = j mod SC)
                                a3c[g][h2][0] = __builtin_elementwise_fma(w3p[h2][4 * k], wn_f2{z4[g][k].x, z4[g][k].x}, a3c[g][h2][0]);
                                a3c[g][h2][1 % SC] = __builtin_elementwise_fma(w3p[h2][4 * k + 1], wn_f2{z4[g][k].y, z4[g][k].y}, a3c[g][h2][1 % SC]);
                                a3c[g][h2][2 % SC] = __builtin_elementwise_fma(w3p[h2][4 * k + 2], wn_f2{z4[g][k].z, z4[g][k].z}, a3c[g][h2][2 % SC]);
                                a3c[g][h2][3 % SC] = __builtin_elementwise_fma(w3p[h2][4 * k + 3], wn_f2{z4[g][k].w, z4[g][k].w}, a3c[g][h2][3 % SC]);
                            }
                            if constexpr (ODD) {
                                a3o[g] = __builtin_elementwise_fma(w3o[2 * k], wn_f2{z4[g][k].x, z4[g][k].y}, a3o[g]);
                                a3o[g] = __builtin_elementwise_fma(w3o[2 * k + 1], wn_f2{z4[g][k].z, z4[g][k].w}, a3o[g]);
                            }
                        }
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
#pragma unroll
                        for (int h2 = 0; h2 < NPL; ++h2) {
                            wn_f2 v = a3c[g][h2][0];
                            if constexpr (SC == 2) v = v + a3c[g][h2][1];
                            if constexpr (SC == 4) v = (v + a3c[g][h2][1]) + (a3c[g][h2][2] + a3c[g][h2][3]);
                            a3[g][2 * h2] = v.x; a3[g][2 * h2 + 1] = v.y;
                        }
                        if constexpr (ODD) a3[g][RS - 1] = a3o[g].x + a3o[g].y;
                    }
                    if (l > 0 && !(WN_V3_ABL & 4)) {
#pragma unroll
                        for (int g = 0; g < G; ++g) {
#pragma unroll
                            for (int h2 = 0; h2 < NPL; ++h2) {
                                wn_v4i v = sk_req[g][h2];   // (only ever this item's streams: re-requested after every barrier A)
                                if (!(WN_V3_ABL & 8) && ((uint32_t)v.y != tag_in || (uint32_t)v.w != tag_in)) v = wn_poll_pair(cx, rs_gs, base_up + g * SB + h2 * 4096 + lane16, tag_in, WN_W_SKIN, e, s + g, WN_V3_SKIP_SLEEP);
                                a3[g][2 * h2] += __int_as_float(v.x);
                                a3[g][2 * h2 + 1] += __int_as_float(v.z);
                            }
                            if constexpr (ODD) {  // (both halves of the pair come from ONE store of the upstream's even lane)
                                wn_v4i v = sk_req[g][NPL];
                                if ((uint32_t)v.y != tag_in || (uint32_t)v.w != tag_in) v = wn_poll_pair(cx, rs_gs, base_up + g * SB + odd_ld, tag_in, WN_W_SKIN, e, s + g, WN_V3_SKIP_SLEEP);
                                a3[g][RS - 1] += __int_as_float((t & 1) ? v.z : v.x);
                            }
                        }
                    }
                }
                if (work || l == NL - 1) {  // (priming: only the head's lanes are kept moving, with zeros)
                    if (slot_look) {
                        // back-pressure: the slot still holds item - NSLOT until its reader has read it -- visible as the reader's own publication of that
                        // item (tags only grow: anything at or beyond it will do).  Items of the priming phase were never published, nor read.
                        const int sj = s - NSLOT * G < 0 ? s - NSLOT * G + ns : s - NSLOT * G;
                        const long long ej = s - NSLOT * G < 0 ? e - 1 : e;
                        if (ej >= n_prime && ej >= 0) {
                            const uint32_t want = WN_V3_SLOT_DEBUG == 2 ? 0u : slot_cons ? (uint32_t)(item - NSLOT + 1) : (uint32_t)(ej + 1);
                            wn_v4i v = ck_req;
                            unsigned spins = 0;
                            while ((int32_t)((uint32_t)v.y - want) < 0 && !cx.fail) {   // (bounded like wn_poll_pair)
                                v = wn_ld_pair(rs_gs, check_off(slot, sj));
                                if ((++spins & 127u) == 0u) {
                                    if (__hip_atomic_load(cx.p->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; break; }
                                    const long long now = (long long)wall_clock64();
                                    if (spins == 128u) cx.t_start = now;
                                    else if (now - cx.t_start > cx.r->timeout_ticks) { wn_give_up(cx, WN_W_SKIN, e, s); break; }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
#pragma unroll
                        for (int h2 = 0; h2 < NPL; ++h2) wn_st_pair(rs_gs, base_me + g * SB + h2 * 4096 + lane16, tag_out, a3[g][2 * h2], a3[g][2 * h2 + 1], local_s);
                        if constexpr (ODD) {
                            const float nb = wn_dpp<0xB1>(a3[g][RS - 1]);  // quad_perm [1,0,3,2]: the odd neighbour's row
                            if ((t & 1) == 0) wn_st_pair(rs_gs, base_me + g * SB + odd_st, tag_out, a3[g][RS - 1], nb, local_s);
                        }
                    }
                }
                if (stamp)  // slot 6: the skip group's B(i) | its chunk length << 40 (10 ns ticks)
                    r.prof[((size_t)cx.w * r.prof_items + item) * WN_STAMPS + 6] = (t0 & 0xffffffffffll) | (((long long)wall_clock64() - t0) << 40);
                if (wn_barrier_failed(cx, failflag)) return;  // ---- A(i+1)
                request_up(s2, slot2);  // the upstream lane of the coming item (and the reader's progress on the slot it re-uses)
                slot = slot2;
            }
        }
        (void)wn_barrier_failed(cx, failflag);  // B(N)
        return;
    }

    // ====================================================================== queue group
    // In the two-streams-per-item form a lane takes the filter AND the gate row of a channel on HALF a tap slice, like the critical
    // group's lanes (QPAIR): half the LDS reads per row pair -- the LDS pipe is what the groups of a workgroup share, and this group's
    // dot is its longest piece (128 streams: its 0.76 us after barrier B set the pace).  The lane pair of a slice adds its halves (one
    // DPP move) and the even lane parks both rows' sums where the image lanes' sums go: pre[] keeps its layout.
    constexpr int QT8 = 2 * T1, QK8 = K1 / 2;
    constexpr bool QPAIR = (WN_V3_PAIR_ROWS == 1 || (WN_V3_PAIR_ROWS == 2 && G >= 2)) && K1 % 8 == 0 && QT8 <= 16;
    const int qch8 = t / QT8, qkq8 = t % QT8, qhalf8 = qkq8 & 1;
    const int qt_f = (2 * qch8) * T1 + qkq8 / 2;  // the image lane of the channel's filter row (gate row: + T1)
    float w0[QPAIR ? 1 : K1];
    wn_f2 wq0[QPAIR ? QK8 : 1];
    const float* imwq = p.blobs + (size_t)cx.w * (SH::NWL * 256);
    if constexpr (QPAIR) {
#pragma unroll
        for (int k = 0; k < QK8; ++k)
            wq0[k] = wn_f2{imwq[(size_t)(K1 + qhalf8 * QK8 + k) * 256 + qt_f], imwq[(size_t)(K1 + qhalf8 * QK8 + k) * 256 + qt_f + T1]};
    } else {
#pragma unroll
        for (int k = 0; k < K1; ++k) w0[k] = img[(size_t)(K1 + k) * 256];
    }
    const float bfg = img[(size_t)(2 * K1 + K2 + RS * DC) * 256];
    const float bfg0 = kq1 == 0 ? bfg : 0.f;
    // (QPAIR: the biases of the two rows this lane parks, on the lanes of slice 0)
    const float qbf = (QPAIR && qkq8 == 0) ? imwq[(size_t)(2 * K1 + K2 + RS * DC) * 256 + qt_f] : 0.f;
    const float qbg = (QPAIR && qkq8 == 0) ? imwq[(size_t)(2 * K1 + K2 + RS * DC) * 256 + qt_f + T1] : 0.f;
    // tap-0 sums of the G streams of an item from the staged taps xo, parked in pre[] (image-lane layout)
    auto tap0_dot = [&](const float* xo_c, int s) {
        if constexpr (QPAIR) {
            wn_f2 a0[G], a1[G];
            float4 v[G][QK8 / 4];
            const float* xsl = xo_c + (qkq8 / 2) * (K1 + 4) + qhalf8 * QK8;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                a0[g] = wn_f2{qbf, qbg};
                a1[g] = wn_f2{0.f, 0.f};
#pragma unroll
                for (int k = 0; k < QK8 / 4; ++k) v[g][k] = reinterpret_cast<const float4*>(xsl + g * L::XR)[k];
            }
#pragma unroll
            for (int k = 0; k < QK8 / 4; ++k)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    a0[g] = __builtin_elementwise_fma(wq0[4 * k], wn_f2{v[g][k].x, v[g][k].x}, a0[g]);
                    a1[g] = __builtin_elementwise_fma(wq0[4 * k + 1], wn_f2{v[g][k].y, v[g][k].y}, a1[g]);
                    a0[g] = __builtin_elementwise_fma(wq0[4 * k + 2], wn_f2{v[g][k].z, v[g][k].z}, a0[g]);
                    a1[g] = __builtin_elementwise_fma(wq0[4 * k + 3], wn_f2{v[g][k].w, v[g][k].w}, a1[g]);
                }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float f = a0[g].x + a1[g].x, gt = a0[g].y + a1[g].y;
                f += wn_partner<1>(f);    // the other half of the slice (the DPP moves outside any lane-dependent branch)
                gt += wn_partner<1>(gt);
                if (qhalf8 == 0) {
                    pre[(s + g) * 256 + qt_f] = f;
                    pre[(s + g) * 256 + qt_f + T1] = gt;
                }
            }
        } else {
            float acc[G], bin[G];
#pragma unroll
            for (int g = 0; g < G; ++g) bin[g] = bfg0;
            wn_dot_lds_gp<K1, G>(w0, xo_c + kq1 * (K1 + 4), L::XR, bin, acc);
#pragma unroll
            for (int g = 0; g < G; ++g) pre[(s + g) * 256 + t] = acc[g];
        }
    };
#pragma clang loop vectorize(disable) interleave(disable)
    for (int s = 0; s < ns; ++s) {  // tap 0 of the first evaluation of every stream: x[t_base - d] from the queue (zeros after reset)
        const float* ring = p.rings + p.ring_off[l] + ((size_t)c * ns + s) * (size_t)ML * R;
        long long pos = (r.t_base - d) % ML;
        if (pos < 0) pos += ML;
        float acc = bfg0;
#pragma unroll
        for (int k = 0; k < K1; ++k) acc += img[(size_t)(K1 + k) * 256] * ring[(size_t)pos * R + kq1 * K1 + k];  // (once per job: the weights straight from the image)
        pre[s * 256 + t] = acc;
    }
    float* rings_l = p.rings + p.ring_off[l] + (size_t)c * ns * (size_t)ML * R;  // stream s: + s * ML * R
    int tmod = (int)(r.t_base % ML);  // queue slot of x[t] of the current item, kept incrementally
    // ---- Few streams: the layer's dilation queues live in LDS (round 3; wavenet_modules.py:42-77).  One stream leaves the workgroup's
    // LDS nearly empty (6 KB of activations), and a ring of (d + 1) rows of R floats fits it up to d = 256 (148 KB in the padded row
    // layout of x); n streams fit up to d + 1 <= 256 / n rows each, the other layers keep their rings in HBM (below).  The rings are
    // loaded when the job starts and written back when it ends (queues outlive a job: continuation, wn_export_queue, DilatedQueue
    // objects of the facade); in between push and tap are LDS accesses and the tap-0 dot reads its row in place -- no staging copy,
    // no prefetch FIFO, no HBM traffic at all.
    if (G == 1 && (long long)L::floats(ns) + (long long)ns * ML * L::XR <= (long long)p.lds_floats && !(WN_V3_ABL & 2)) {
        float* lring = lds + L::floats(ns);  // [n_streams][ML][XR]
        for (int i = t; i < ns * ML * R; i += 256) lring[(i / R) * L::XR + SH::xpad(i % R)] = rings_l[i];
        const int xq1 = SH::xpad(tr);
        int bufq = 0;
        long long itemq = 0;
        for (long long e = 0; e < r.n_eval; ++e, tmod = (tmod + 1 == ML) ? 0 : tmod + 1) {
            // tap-0 half of the dilated conv for the NEXT timestep: x[t+1-d] sits in slot (t + 2) mod (d + 1) (d = 1: the row just pushed)
            const int tap = tmod + 2 >= ML ? tmod + 2 - ML : tmod + 2;
            for (int s = 0; s < ns; ++s, bufq ^= 1, ++itemq) {
                float* ring_s = lring + (size_t)s * ML * L::XR;
                if (wn_barrier_failed(cx, failflag)) return;  // ---- A(i): x of this item staged
                const bool stamp = r.prof && itemq < r.prof_items && tid == 512;
                const long long t0 = stamp ? (long long)wall_clock64() : 0;
                if (t < R) ring_s[tmod * L::XR + xq1] = xs[bufq * L::XR + xq1];  // the push (wavenet_modules.py:55-57)
                if (wn_barrier_failed(cx, failflag)) return;  // ---- B(i)
                const long long t1 = stamp ? (long long)wall_clock64() : 0;
                tap0_dot(ring_s + tap * L::XR, s);
                if (stamp)
                    r.prof[((size_t)cx.w * r.prof_items + itemq) * WN_STAMPS + 7] =
                        (t0 & 0xffffffffffll) | (((t1 - t0) & 0xfff) << 40) | ((((long long)wall_clock64() - t1) & 0xfff) << 52);
            }
        }
        if (wn_barrier_failed(cx, failflag)) return;  // A(N)
        for (int i = t; i < ns * ML * R; i += 256) rings_l[i] = lring[(i / R) * L::XR + SH::xpad(i % R)];  // the queues' state back to HBM
        (void)wn_barrier_failed(cx, failflag);        // B(N)
        return;
    }
    // Queue taps x[t+1-d] are read WN_V3_TAP_AHEAD items ahead (rows of large-d layers are an HBM miss): a register FIFO of three
    // entries, hand-scheduled (wn_q_issue / wn_q_take); entry = item mod 3.
    // lanes t < G*R: element tr of stream s + tg of the item -- every wave issues ONE tap load per item, whatever G
    const bool qlane = t < G * R;
    const bool fifo = d != 1 && qlane && !(WN_V3_ABL & 2);  // (wave-uniform: R is a multiple of 64 in every instantiated shape)
    int s_a = 0, tmod_a = tmod;  // coordinates of the item whose tap is requested next
    auto next_tap_ptr = [&]() -> const float* {
        const int tap = tmod_a + 2 >= ML ? tmod_a + 2 - ML : tmod_a + 2;  // slot of x[t+1-d]: (t+1-d) mod (d+1) = (t+2) mod (d+1)
        const float* q = rings_l + ((size_t)(s_a + tg) * ML + tap) * R + tr;
        s_a += G;
        if (s_a == ns) { s_a = 0; tmod_a = (tmod_a + 1 == ML) ? 0 : tmod_a + 1; }
        return q;
    };
    if (fifo) {
#pragma unroll
        for (int j = 0; j < WN_V3_TAP_AHEAD; ++j) wn_q_issue_slot(j, next_tap_ptr());
    }
    int slot = 0;  // item mod WN_V3_TAP_AHEAD
    // With few streams and a small dilation the tap row of an item was pushed fewer than WN_V3_TAP_AHEAD items before it (it is
    // x of item i - n_streams*(d-1)): the lane keeps its own last x values instead of reading the queue ahead of the push.
    const long long back = (long long)nI * (d - 1);  // in items
    const bool near = d != 1 && back <= WN_V3_TAP_AHEAD;
    float hx[WN_V3_TAP_AHEAD];  // x of the last WN_V3_TAP_AHEAD items, newest first
#pragma unroll
    for (int j = 0; j < WN_V3_TAP_AHEAD; ++j) hx[j] = 0.f;
    // Layers whose tap does not depend on recent x (d != 1, not `near`) do NOTHING between barriers A and B: the tap of item i is
    // staged (double buffered) at the end of item i-1's window after barrier B, and x[t] is pushed after barrier B as well (x stays
    // in LDS until item i+1's B).  With the packed filter/gate dot of the critical group the queue group's push + staging (0.29 us)
    // had become as long as the critical group's own A -> B.
    const bool late_wg = d != 1 && !near && !(WN_V3_ABL & 2);  // workgroup-uniform
    const bool late = late_wg && fifo;                          // ... and this wave loads taps
    float* xol1 = lds + L::sk;  // second tap buffer (the head's staging area is free in a layer workgroup)
    const long long n_items = r.n_eval * nI;
    constexpr int D = WN_V3_TAP_AHEAD;
    const int xq = tg * L::XR + SH::xpad(tr);  // this lane's element in a [G][XR] staging area
    if (late && n_items > 0) xol[xq] = wn_q_take_slot<D - 1>(0);  // item 0's tap (D - 1 younger loads of the initial fill)
    constexpr bool TAP_A = WN_V3_TAP_AT_A == 1 || (WN_V3_TAP_AT_A == 2 && G >= 2);   // the tap request behind barrier A (see WN_V3_TAP_AT_A)
    const float* q_next = (TAP_A && late) ? next_tap_ptr() : nullptr;   // address of the tap of item D (requested behind barrier A of item 0)
    // ... and the push of a late layer is done by the waves that load no taps (lanes R..2R-1, when there are that many): the tap
    // waves then issue nothing but tap loads, and an entry always has exactly D - 1 younger operations
    constexpr bool PUSH_HI = 2 * G * R <= 256;
    constexpr bool PUSH_SEP = PUSH_HI && (G * R) % 64 == 0;  // ... in waves of their own: only then does a tap wave issue nothing but tap loads
    const bool pusher = late_wg && (PUSH_HI ? (t >= G * R && t < 2 * G * R) : qlane);
    const int pg = PUSH_HI ? (t - G * R) / R : tg, prow = PUSH_HI ? (t - G * R) % R : tr;  // (stream of the item, element) this lane pushes
    int buf = 0;
    long long item = 0;
    for (long long e = 0; e < r.n_eval; ++e, tmod = (tmod + 1 == ML) ? 0 : tmod + 1) {
        for (int s = 0; s < ns; s += G, buf ^= 1, ++item) {
            float* xo_cur = (late_wg && (item & 1)) ? xol1 : xol;
            float* xo_nxt = (item & 1) ? xol : xol1;
            if (wn_barrier_failed(cx, failflag)) return;  // ---- A(i): x of this item staged
            const bool stamp = r.prof && item < r.prof_items && tid == 512;
            const long long t0 = stamp ? (long long)wall_clock64() : 0;
            if (TAP_A && late) wn_q_issue_slot(slot, q_next);  // the tap of item i + D goes into the entry item i gave up at the end of item i - 1
            // ---- queue push (wavenet_modules.py:55-57); stage the tap x[t+1-d] (d = 1: it is x[t] itself)
            if (!late_wg && qlane && !(WN_V3_ABL & 2)) {
                const float xv = xs[buf * (G * L::XR) + xq];
                rings_l[((size_t)(s + tg) * ML + tmod) * R + tr] = xv;
                // the tap: d = 1: x[t] itself; a row pushed fewer items ago than the prefetch distance (few streams, small d):
                // this lane's own copy of it (the queue read ahead of the push would be stale); else the prefetched queue row
                // (program order here: push, take, [B], load: an entry has D - 1 younger loads and D younger stores)
                float tap = 0.f;
                if (fifo) tap = item < D - 1 ? wn_q_take_slot<D>(slot) : wn_q_take_slot<2 * D - 1>(slot);
                if (d == 1) tap = xv;
                else if (near && item >= back) {
#pragma unroll
                    for (int j = 0; j < D; ++j) tap = back == j + 1 ? hx[j] : tap;  // (back is workgroup-uniform)
                }
                xo_cur[xq] = tap;
#pragma unroll
                for (int j = D - 1; j > 0; --j) hx[j] = hx[j - 1];
                hx[0] = xv;
            }
            if (wn_barrier_failed(cx, failflag)) return;  // ---- B(i): the tap is staged
            if (WN_V3_QUEUE_DEFER > 0 && G >= 2) __builtin_amdgcn_s_sleep(WN_V3_QUEUE_DEFER);
            const long long t1 = stamp ? (long long)wall_clock64() : 0;
            // ---- tap-0 half of the dilated conv for the NEXT timestep of this stream, parked for the critical group
            if (!(WN_V3_ABL & 2)) {
                if (pusher) rings_l[((size_t)(s + pg) * ML + tmod) * R + prow] = xs[buf * (G * L::XR) + pg * L::XR + SH::xpad(prow)];  // the push, off the A -> B window
                tap0_dot(xo_cur, s);
                if constexpr (TAP_A) {
                    if (fifo && !late) wn_q_issue_slot(slot, next_tap_ptr());
                    if (late) q_next = next_tap_ptr();   // (its address arithmetic here, the load behind the coming barrier A)
                    if (late && item + 1 < n_items) {
                        // program order of a tap wave per item: tap load (behind A), [push store], take (here).  The entry of item i + 1 was requested in
                        // item i + 1 - D: D - 1 younger loads and, in a wave that also pushes, D younger stores (the initial fill: D + i operations in all)
                        const int ns1 = slot == D - 1 ? 0 : slot + 1;
                        xo_nxt[xq] = (PUSH_SEP || item < D - 1) ? wn_q_take_slot<D - 1>(ns1) : wn_q_take_slot<2 * D - 1>(ns1);
                    }
                } else {
                    if (fifo) wn_q_issue_slot(slot, next_tap_ptr());  // the tap of item i + D goes into the entry item i has just given up
                    if (late && item + 1 < n_items) {
                        // the NEXT item's tap into the other buffer: D - 1 younger loads; the stores of a wave that also pushes (R > 128:
                        // program order push, load, take) only add to what may stay in flight from item D - 2 on
                        const int ns1 = slot == D - 1 ? 0 : slot + 1;
                        xo_nxt[xq] = (PUSH_SEP || item < D - 2) ? wn_q_take_slot<D - 1>(ns1) : wn_q_take_slot<2 * D - 2>(ns1);
                    }
                }
            }
            slot = slot == D - 1 ? 0 : slot + 1;
            if (stamp)  // slot 7: the queue group's A(i) | push+stage length << 40 | dot length << 52
                r.prof[((size_t)cx.w * r.prof_items + item) * WN_STAMPS + 7] =
                    (t0 & 0xffffffffffll) | (((t1 - t0) & 0xfff) << 40) | ((((long long)wall_clock64() - t1) & 0xfff) << 52);
        }
    }
    if (wn_barrier_failed(cx, failflag)) return;  // A(N)
    (void)wn_barrier_failed(cx, failflag);        // B(N)
}

// dot(w[0..K), x[0..K)) with x in LDS, the float4 reads issued CH at a time (wn_dot_lds issues all K/4 up front: 64 VGPRs for the
// head's K = 64, which a 768-thread workgroup cannot afford)
template <int K, int CH>
static __device__ __forceinline__ float wn_dot_lds_chunked(const float (&w)[K], const float* x, float init) {
    static_assert(K % (4 * CH) == 0, "chunking");
    const float4* x4 = reinterpret_cast<const float4*>(x);
    wn_f2 a01 = {init, 0.f}, a23 = {0.f, 0.f};  // two packed chains (v_pk_fma_f32), as wn_dot_lds
#pragma unroll
    for (int c0 = 0; c0 < K / 4; c0 += CH) {
        float4 v[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) v[k] = x4[c0 + k];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            a01 = __builtin_elementwise_fma(wn_f2{w[4 * (c0 + k)], w[4 * (c0 + k) + 1]}, wn_f2{v[k].x, v[k].y}, a01);
            a23 = __builtin_elementwise_fma(wn_f2{w[4 * (c0 + k) + 2], w[4 * (c0 + k) + 3]}, wn_f2{v[k].z, v[k].w}, a23);
        }
    }
    return (a01.x + a01.y) + (a23.x + a23.y);
}

// Head role (threads 0-255 of the workgroup; same arithmetic, granules and LDS offsets as wn_v2_head_multi with G = 1): relu(skip)
// -> end_conv_1 slice (+b, relu) -> its K-slice of end_conv_2 -> partial logits   (wavenet_model.py:167-169)
// p.HR replicas of the PA head workgroups share the streams (replica j serves the streams s = j mod HR): a head workgroup's cycle
// per stream (compute 0.53 us + the round trip of its publication) is the slowest stage once the layer workgroups process two
// streams per item -- and the chain leaves 44 of the 256 CUs unused.
template <class SH, int P>
static __device__ void wn_v3_head(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds, int hw) {
    constexpr int S = SH::S, EC = SH::EC, T3 = SH::T3, K3 = SH::K3, QS = S / 256;
    constexpr int CH3 = (K3 / 4) % 4 == 0 ? 4 : (K3 / 4) % 2 == 0 ? 2 : 1;  // (float4 reads of the long dot in flight together)
    static_assert(K3 % 4 == 0 && EC % 4 == 0, "head slices are read as float4");
    using L = WnV3Lds<SH>;
    const int tid = threadIdx.x, ns = p.n_streams;
    const int HR = p.HR, h = hw % p.PA, rep = hw / p.PA;  // slice of end_conv_1 / end_conv_2, replica
    const int n_mine = rep < ns ? (ns - rep + HR - 1) / HR : 0;  // streams this replica serves
    // end_conv_1's slice stays in registers and the lane's end_conv_2 row (EC floats) lives in LDS as float4 [EC / 4][256 lanes] -- a head
    // workgroup has the LDS to itself, and with both in registers the role did not fit the 152 registers the compiler may use.  A slice
    // too long for the registers (K3 > 100: the train_script.py shape, 1024 skip channels over 8 lanes) swaps places with the row: W4LDS.
    constexpr bool W4LDS = K3 > 100;
    static_assert(!W4LDS || EC <= 32, "one of the head's two weight vectors has to fit the registers");
    static_assert(L::pre % 4 == 0, "the head's LDS-resident weights are read as float4");
    float w4[W4LDS ? 1 : K3], w5[W4LDS ? EC : 1];
    const float* img = p.blobs + p.head_blob_off + (size_t)h * (SH::NWH * 256) + tid;
    float4* wl = reinterpret_cast<float4*>(lds + L::pre) + tid;  // [k4 * 256]: the LDS-resident vector of this lane
    if constexpr (W4LDS) {
#pragma unroll 4
        for (int k4 = 0; k4 < K3 / 4; ++k4)
            wl[k4 * 256] = float4{img[(size_t)(4 * k4) * 256], img[(size_t)(4 * k4 + 1) * 256], img[(size_t)(4 * k4 + 2) * 256], img[(size_t)(4 * k4 + 3) * 256]};
#pragma unroll
        for (int k = 0; k < EC; ++k) w5[k] = img[(size_t)(K3 + k) * 256];
    } else {
#pragma unroll
        for (int k = 0; k < K3; ++k) w4[k] = img[(size_t)k * 256];
#pragma unroll
        for (int k4 = 0; k4 < EC / 4; ++k4)
            wl[k4 * 256] = float4{img[(size_t)(K3 + 4 * k4) * 256], img[(size_t)(K3 + 4 * k4 + 1) * 256], img[(size_t)(K3 + 4 * k4 + 2) * 256], img[(size_t)(K3 + 4 * k4 + 3) * 256]};
    }
    float4* w5l = wl;
    const float b1 = img[(size_t)(K3 + EC) * 256], b2 = img[(size_t)(K3 + EC + 1) * 256];
    const int kq3 = tid % T3, row3 = tid / T3;
    float* sk = lds + L::sk;
    float* ev = lds + L::ev;
    int* failflag = reinterpret_cast<int*>(lds + L::smp + 48);
    int* locflags = reinterpret_cast<int*>(lds + L::smp + 52);
    long long* park = reinterpret_cast<long long*>(lds + L::park);
    if (tid == 0) {
        *failflag = 0;
        const int mine = wn_xcc_id();
        __hip_atomic_store(p.xcc_tab + cx.w, (unsigned)(mine + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        locflags[0] = p.allow_plain ? (int)wn_same_xcd(cx, mine, p.n_lw + p.PA * p.HR, p.n_smp) : 0;  // logits feed the samplers
    }
    wn_lds_barrier();
    const bool local_l = locflags[0] != 0;
    const __amdgpu_buffer_rsrc_t rs_gs = wn_rsrc(p.gs);
    // The P lanes of the running skip sum (published by the last layer's skip groups; layout: see the skip group) of the NEXT item are
    // requested early (after this item's long dot): with tokens queued in front of it the head's cycle would otherwise be compute PLUS a
    // request round trip -- the slowest stage of the 64-stream chain, whatever the layer stages did (profiles/archive/r02_v3_head_request.txt).
    // Nothing queued (latency-bound runs): the early request comes back stale and the lanes are polled when due.
    constexpr int NPL = QS / 2, ODD = QS % 2, NPR = NPL + ODD;
    constexpr unsigned ODD_BASE = 2048u * (unsigned)(QS - 1);
    wn_v4i nv[NPR][P];
    // (offsets: the lane's part in ONE register, the wave-uniform part of every lane in the scalar offset of the buffer load)
    const unsigned lane16 = (unsigned)tid * 16, lane8 = (unsigned)tid * 8, odd_ld = ODD_BASE + (unsigned)(tid & ~1) * 8;
    const __amdgpu_buffer_rsrc_t rs_gl = wn_rsrc(p.gl);
    auto request = [&](int s2) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const unsigned base = (unsigned)(((((size_t)(p.n_lw - P) + j) * ns + s2) * (size_t)S) * 8);   // the last layer-role workgroups' lanes
#pragma unroll
            for (int h2 = 0; h2 < NPL; ++h2) nv[h2][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_gs, lane16, base + h2 * 4096, 16);
            if constexpr (ODD) nv[NPL][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_gs, odd_ld, base, 16);
        }
    };
    if (n_mine > 0) request(rep);
    for (long long e = 0; e < r.n_eval; ++e) {
        const bool prime = e < r.n_given - 1;
        const uint32_t tag = (uint32_t)(e + 1);
        for (int s = rep; s < ns; s += HR) {
            const long long item = e * n_mine + (s - rep) / HR;
            wn_stamp(r, park, item, 0);
            // (this item's stream: requested for it one item ago.)  Stale lanes are re-requested TOGETHER until all carry the tag: polled
            // one by one, every late lane cost a round trip of its own -- with nothing queued in front of the head (latency-bound runs)
            // that was 4 x ~0.5 us between the last layer's publication and the head's first instruction (profiles/archive/r03_ring_tail.txt)
            {
                unsigned spins = 0;
                while (!cx.fail) {
                    bool fresh = true;
#pragma unroll
                    for (int h2 = 0; h2 < NPR; ++h2)
#pragma unroll
                        for (int j = 0; j < P; ++j) fresh = fresh && (uint32_t)nv[h2][j].y == tag && (uint32_t)nv[h2][j].w == tag;
                    if (fresh) break;
                    if ((++spins & 127u) == 0u) {  // bounded wait, as wn_poll_pair
                        if (__hip_atomic_load(cx.p->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; break; }
                        const long long now = (long long)wall_clock64();
                        if (spins == 128u) cx.t_start = now;
                        else if (now - cx.t_start > cx.r->timeout_ticks) { wn_give_up(cx, WN_W_HEAD, e, s); break; }
                    }
                    request(s);
                }
            }
#pragma unroll
            for (int h2 = 0; h2 < NPL; ++h2) {
                float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
                for (int j = 0; j < P; ++j) {  // fixed order j = 0..P-1
                    sum0 += __int_as_float(nv[h2][j].x);
                    sum1 += __int_as_float(nv[h2][j].z);
                }
                sk[SH::skpad(tid + 512 * h2)] = sum0 > 0.f ? sum0 : 0.f;        // relu(skip), rows tid + 512 h2 and tid + 512 h2 + 256
                sk[SH::skpad(tid + 512 * h2 + 256)] = sum1 > 0.f ? sum1 : 0.f;
            }
            if constexpr (ODD) {  // the natural-order tail: row 256 (QS-1) + tid, the half of the pair that is this lane's
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < P; ++j) sum += __int_as_float((tid & 1) ? nv[NPL][j].z : nv[NPL][j].x);
                sk[SH::skpad(256 * (QS - 1) + tid)] = sum > 0.f ? sum : 0.f;
            }
            if (wn_barrier_failed(cx, failflag)) return;
            wn_stamp(r, park, item, 1);
            const unsigned gl_off = (unsigned)((((size_t)h * ns + s) * 256) * 8);  // (wave-uniform: the scalar offset of the store)
            auto publish_logit = [&](float v) {
                const wn_v2i d = {__float_as_int(v), (int)tag};
                if (local_l) __builtin_amdgcn_raw_buffer_store_b64(d, rs_gl, lane8, gl_off, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(d, rs_gl, lane8, gl_off, 16);  // write-through
            };
            if (!prime) {
                float a;
                if constexpr (W4LDS) {  // the arithmetic of wn_dot_lds_chunked (two packed chains, same order), the weights from LDS
                    wn_f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
                    const float4* x4 = reinterpret_cast<const float4*>(sk + kq3 * (K3 + 4));
#pragma unroll
                    for (int c0 = 0; c0 < K3 / 4; c0 += CH3) {
                        float4 wv[CH3], xv[CH3];
#pragma unroll
                        for (int k = 0; k < CH3; ++k) { wv[k] = wl[(c0 + k) * 256]; xv[k] = x4[c0 + k]; }
#pragma unroll
                        for (int k = 0; k < CH3; ++k) {
                            a01 = __builtin_elementwise_fma(wn_f2{wv[k].x, wv[k].y}, wn_f2{xv[k].x, xv[k].y}, a01);
                            a23 = __builtin_elementwise_fma(wn_f2{wv[k].z, wv[k].w}, wn_f2{xv[k].z, xv[k].w}, a23);
                        }
                    }
                    a = (a01.x + a01.y) + (a23.x + a23.y);
                } else {
                    a = wn_dot_lds_chunked<K3, CH3>(w4, sk + kq3 * (K3 + 4), 0.f);
                }
                a = wn_reduce<T3>(a) + b1;
                if (kq3 == 0) ev[row3] = a > 0.f ? a : 0.f;  // relu(end_conv_1)
                request(s + HR < ns ? s + HR : rep);  // (after the long dot: its sixteen registers are not live next to that dot's operands)
                // the lane's end_conv_2 row: fetched from LDS while the other waves finish end_conv_1 where it is short enough to sit in
                // registers next to end_conv_1's slice (EC <= 32), read chunk by chunk inside the dot otherwise
                constexpr bool W5PRE = !W4LDS && EC <= 32;
                float4 w5r[W5PRE ? EC / 4 : 1];
                if constexpr (W5PRE) {
#pragma unroll
                    for (int k4 = 0; k4 < EC / 4; ++k4) w5r[k4] = w5l[k4 * 256];
                }
                wn_lds_barrier();
                {   // partial end_conv_2: the arithmetic of wn_dot_lds_chunked (two packed chains, same order)
                    wn_f2 a01 = {b2, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
                    for (int k4 = 0; k4 < EC / 4; ++k4) {
                        float4 w;
                        if constexpr (W4LDS) w = float4{w5[W4LDS ? 4 * k4 : 0], w5[W4LDS ? 4 * k4 + 1 : 0], w5[W4LDS ? 4 * k4 + 2 : 0], w5[W4LDS ? 4 * k4 + 3 : 0]};
                        else w = W5PRE ? w5r[W5PRE ? k4 : 0] : w5l[k4 * 256];
                        const float4 v = reinterpret_cast<const float4*>(ev)[k4];
                        a01 = __builtin_elementwise_fma(wn_f2{w.x, w.y}, wn_f2{v.x, v.y}, a01);
                        a23 = __builtin_elementwise_fma(wn_f2{w.z, w.w}, wn_f2{v.z, v.w}, a23);
                    }
                    publish_logit((a01.x + a01.y) + (a23.x + a23.y));
                }
            } else {
                request(s + HR < ns ? s + HR : rep);
                publish_logit(0.f);
            }
            wn_stamp(r, park, item, 2);
            wn_lds_barrier();
            wn_stamp(r, park, item, 3);
            wn_stamp_flush(r, park, cx.w, item);
        }
    }
}

// ---- one-wave sampling: C = 256 classes, lane i holds the four CONSECUTIVE classes 4i .. 4i+3 (wavenet_model.py:280-294).
// The four-wave sampler (wn_sample_v2: one class per lane) spends 1.65 us per token, most of it in six workgroup barriers and LDS
// round trips between its waves (profiles/archive/r03_ablations_two_streams_form_and_ring_tail.txt); a first one-wave version that kept its
// arithmetic register by register (lane i: classes i, i+64, ...) was issue bound -- four of every wave-level reduction and scan in one
// instruction stream: 2.0 us.  Here a lane reduces its own four classes first, so there is ONE wave-level max, sum and float64 scan
// (DPP inside the 16-lane rows, row_bcast across them) and no LDS at all.  The probabilities are the same floats up to the order of
// the float32 sum behind 1 / sum (1 ulp), the CDF the same float64 sums up to their order: an index can only differ from the
// four-wave sampler's where the uniform sits within ~1e-7 of a CDF boundary -- an order of magnitude inside what the logits' own
// rounding (2e-6 against the reference) moves those boundaries.
template <int CTRL, int ROWS>  // value of the DPP source lane in the rows of ROWS (a row mask), 0.0 elsewhere
static __device__ __forceinline__ double wn_dpp_rows_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ int wn_sample_1w(const WnRun& r, const float (&logit)[4], int lane, double u, bool greedy, float temperature) {
    float x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = logit[k];
        if (r.reg) x[k] -= r.reg[4 * lane + k];   // (rare option, loaded behind the logits: four more registers live across the poll cost the kernel a stack slot)
        if (!greedy && temperature != 1.0f) x[k] = x[k] / temperature;   // (x / 1 == x exactly: the four divisions are ~40 instructions of the ring)
    }
    const float gm = wn_wave_max(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
    if (greedy) {  // first index of the maximum (torch.max semantics)
        const int mine = x[0] == gm ? 4 * lane : x[1] == gm ? 4 * lane + 1 : x[2] == gm ? 4 * lane + 2 : x[3] == gm ? 4 * lane + 3 : 0x7fffffff;
        return wn_wave_min_i(mine);
    }
    float pk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pk[k] = expf(x[k] - gm);
    const float tot = wn_wave_sum(((pk[0] + pk[1]) + pk[2]) + pk[3]);
    const float inv = 1.0f / tot;
    // inclusive float64 scan (np.cumsum): inside the lane, then over the lanes' totals
    double c[4];
    c[0] = (double)(pk[0] * inv);
    c[1] = c[0] + (double)(pk[1] * inv);
    c[2] = c[1] + (double)(pk[2] * inv);
    c[3] = c[2] + (double)(pk[3] * inv);
    double run = c[3];
    run += wn_row_shr_f64<1>(run);
    run += wn_row_shr_f64<2>(run);
    run += wn_row_shr_f64<4>(run);
    run += wn_row_shr_f64<8>(run);
    run += wn_dpp_rows_f64<0x142, 0xA>(run);  // row_bcast:15 into rows 1 and 3
    run += wn_dpp_rows_f64<0x143, 0xC>(run);  // row_bcast:31 into rows 2 and 3
    const double total = wn_lane_d(run, 63);
    const double before = __hiloint2double(wn_dpp_i<0x138>(__double2hiint(run)), wn_dpp_i<0x138>(__double2loint(run)));  // wave_shr:1 -- the classes before this lane's (lane 0: 0.0)
    // searchsorted(cdf / cdf[-1], u, side='right') = the number of classes with cdf / total <= u.  The quotient only has to be formed
    // where the comparison is within rounding of the boundary: cdf <= u total (1 - 2^-50) implies cdf / total < u, cdf > u total (1 + 2^-50)
    // implies cdf / total > u + ulp -- the float64 division (four per lane, ~25 instructions each) is taken on the rare wave that holds a
    // lane in between, and then by all its lanes: the count is the one the division gives, always.
    const double ut = u * total;
    const double lo = ut * (1. - 0x1p-50), hi = ut * (1. + 0x1p-50);
    double cdf[4];
    bool le[4], unsure = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cdf[k] = k == 3 ? run : before + c[k];
        le[k] = cdf[k] <= lo;
        unsure = unsure || (cdf[k] > lo && cdf[k] <= hi);
    }
    if (__builtin_amdgcn_ballot_w64(unsure) != 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) le[k] = cdf[k] / total <= u;
    }
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) idx += __popcll(__builtin_amdgcn_ballot_w64(le[k]));
    return idx > 255 ? 255 : idx;
}

// the partial logits of N head slices (h0 ..) for this lane's four consecutive classes (32 contiguous bytes per slice: two 16-byte
// loads), re-requested together until all carry the tag; added to logit[] in the order h0, h0 + 1, ... (bounded like wn_poll_fixed)
template <int N>
static __device__ __forceinline__ void wn_poll_logits(WnCtx& cx, __amdgpu_buffer_rsrc_t rs_gl, unsigned lane32, int h0, int s, int ns, uint32_t tag, long long e,
                                                      float (&logit)[4]) {
    if (cx.fail) return;
    unsigned spins = 0;
    for (;;) {
        wn_v4i v[N][2];
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                v[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_gl, lane32, (unsigned)((((size_t)(h0 + j) * ns + s) * 256) * 8 + 16 * q), 16);  // sc1
        unsigned stale = 0;  // tags only ever grow towards `tag`: any difference is a stale granule
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) stale |= ((uint32_t)v[j][q].y ^ tag) | ((uint32_t)v[j][q].w ^ tag);
        if (__builtin_amdgcn_ballot_w64(stale != 0u) == 0) {
#pragma unroll
            for (int j = 0; j < N; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    logit[2 * q] += __int_as_float(v[j][q].x);
                    logit[2 * q + 1] += __int_as_float(v[j][q].z);
                }
            return;
        }
        if ((++spins & 127u) == 0u) {
            if (__hip_atomic_load(cx.p->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; return; }
            const long long now = (long long)wall_clock64();
            if (spins == 128u) cx.t_start = now;
            else if (now - cx.t_start > cx.r->timeout_ticks) { wn_give_up(cx, WN_W_LOGITS, e, s); return; }
        }
    }
}

// Wide form of the sampler's wait (WIDE: all 256 threads of the workgroup, thread c = class c): the N partial logits of class c from
// the head slices h0 .. h0 + N - 1, one 8-byte granule each, re-requested together until all carry the tag; added in the order h0, h0 + 1, ...
template <int N>
static __device__ __forceinline__ void wn_poll_class(WnCtx& cx, __amdgpu_buffer_rsrc_t rs_gl, unsigned c8, int h0, int s, int ns, uint32_t tag, long long e, float& logit) {
    if (cx.fail) return;
    unsigned spins = 0;
    for (;;) {
        wn_v2i v[N];
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b64(rs_gl, c8, (unsigned)((((size_t)(h0 + j) * ns + s) * 256) * 8), 16);  // sc1
        unsigned stale = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) stale |= (uint32_t)v[j].y ^ tag;
        if (__builtin_amdgcn_ballot_w64(stale != 0u) == 0) {
#pragma unroll
            for (int j = 0; j < N; ++j) logit += __int_as_float(v[j].x);
            return;
        }
        if ((++spins & 127u) == 0u) {
            if (__hip_atomic_load(cx.p->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; return; }
            const long long now = (long long)wall_clock64();
            if (spins == 128u) cx.t_start = now;
            else if (now - cx.t_start > cx.r->timeout_ticks) { wn_give_up(cx, WN_W_LOGITS, e, s); return; }
        }
    }
}

// Sampler role (ONE wave: threads 0-63 of the workgroup; workgroup j of n_smp serves the streams s = j mod n_smp): turns the head's
// partial logits of evaluation e-1 into the class index that enters evaluation e (teacher forced while priming) -- and then does
// layer 0's start_conv itself: the row of start_conv^T for that class (+ bias) goes out as layer 0's input granules g0[s][R]
// (tag e+1, 16-byte pairs), so that layer 0 consumes a ready vector like every other layer (wavenet_model.py:127, 300-302).
// start_conv^T ([C][R] floats) is copied into the workgroup's LDS at start when it fits (p.start_in_lds: 128 KB at R = 128): the row
// gather after the draw is an LDS read instead of a dependent L2 / HBM load.
// WIDE (many head slices, a single stream's ring: 16 slices are 32 KB of granules per token -- one wave needs ~0.6 us per look at them):
// the caller lets the workgroup's first FOUR waves in, thread c collects class c from every slice (8-byte loads, 16 in flight), the
// sums meet in LDS (lds_lg, 256 floats) and wave 0 draws as in the one-wave form -- same sums in the same order, two LDS barriers more.
template <class SH, bool WIDE = false>
static __device__ void wn_v3_sampler(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds_smp, float* lds_tab, int j, float* lds_lg = nullptr) {
    constexpr int R = SH::R, NP = (R / 2 + 63) / 64;  // pairs of row elements per lane
    constexpr int NT = WIDE ? 256 : 64;
    static_assert(R <= 256 && R % 4 == 0, "one start_conv row per sampler wave, copied as float4");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ns = p.n_streams;  // (one-wave form: the caller lets only wave 0 in)
    int* failflag = reinterpret_cast<int*>(lds_smp + 48);
    if (WIDE && threadIdx.x == 0) *failflag = 0;
    const int mine = wn_xcc_id();
    if (lane == 0) __hip_atomic_store(p.xcc_tab + cx.w, (unsigned)(mine + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool local_i = p.allow_plain && wn_same_xcd(cx, mine, 0, p.P);  // the row feeds every slice of layer 0  (wave-uniform)
    const __amdgpu_buffer_rsrc_t rs_g0 = wn_rsrc(p.g0), rs_gl = wn_rsrc(p.gl);
    const float* tab = p.start_t;
    if (p.start_in_lds) {
        const float4* src = reinterpret_cast<const float4*>(p.start_t);
        float4* dst = reinterpret_cast<float4*>(lds_tab);
        for (int i = threadIdx.x; i < 256 * R / 4; i += NT) dst[i] = src[i];   // (WIDE: the per-item barriers order the copy before wave 0's first read)
        tab = lds_tab;
    }
    float bias0[NP][2];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
        const int pi = lane + 64 * m;
        bias0[m][0] = (p.start_b && 2 * pi < R) ? p.start_b[2 * pi] : 0.f;
        bias0[m][1] = (p.start_b && 2 * pi < R) ? p.start_b[2 * pi + 1] : 0.f;
    }
    long long* park = reinterpret_cast<long long*>(lds_smp + 64);
    const unsigned lane32 = (unsigned)lane * 32;  // byte offset of the lane's four granules inside a slice's 256
    if constexpr (WIDE) wn_lds_barrier();  // (the fail flag is initialised)
    const int lane0 = (int)threadIdx.x >> 8;
    for (long long e = 1; e <= r.n_eval; ++e) {
        for (int s = j; s < ns; s += p.n_smp) {
            const long long item = (e - 1) * ns + s;  // stamps (diagnostics): 0 start of the wait, 1 logits complete, 2 row published
            wn_stamp(r, park, item, 0);
            // The item's uniform (streamed from HBM, eight bytes per item) is requested BEFORE the wait for the logits: a global load behind
            // their arrival sat on the ring (a single stream's timestep is 8 - 20 us, an HBM miss is half a microsecond of it).  The
            // temperature and the given sample (kernel argument / L2 hits) stay behind the wait -- asking for them early as well raised
            // the kernels' scalar pressure to where the input poll's base pointers were spilled (WN_AP_SGPR_HAZARD).
            // (+ lane0: 0 for every lane of this wave, unknown to the compiler -- vector loads into VGPRs; as scalar loads the values sit in
            //  SGPRs across the poll, and the kernel's scalar registers are already spilling into VGPR lanes)
            const long long g = e - r.n_given;
#if WN_SAMPLER_PREFETCH == 1
            const float temp = r.stream_temps ? r.stream_temps[s + lane0] : r.temperature;
            const bool greedy = r.greedy != 0 || !(temp > 0.f);
            const double u = (g >= 0 && !greedy) ? r.uniforms[(size_t)s * r.num_samples + g + lane0] : 0.;
            const int given = g < 0 ? r.first[(size_t)s * r.n_given + e + lane0] : 0;
#elif WN_SAMPLER_PREFETCH == 2   // only the uniform (the one that streams from HBM), taken whatever the temperature says: r.uniforms != NULL is all it needs
            const double u = (g >= 0 && r.uniforms) ? r.uniforms[(size_t)s * r.num_samples + g + lane0] : 0.;
#endif
            float logit[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (WIDE) {   // class threadIdx.x: the PA partial logits summed in the order h = 0 .. PA-1, then four classes per lane of wave 0
                float mine1 = 0.f;
                const unsigned c8 = threadIdx.x * 8;
                int h = 0;
                for (; p.PA - h >= 16; h += 16) wn_poll_class<16>(cx, rs_gl, c8, h, s, ns, (uint32_t)e, e, mine1);
                if (p.PA - h >= 8) { wn_poll_class<8>(cx, rs_gl, c8, h, s, ns, (uint32_t)e, e, mine1); h += 8; }
                if (p.PA - h >= 4) { wn_poll_class<4>(cx, rs_gl, c8, h, s, ns, (uint32_t)e, e, mine1); h += 4; }
                if (p.PA - h >= 2) { wn_poll_class<2>(cx, rs_gl, c8, h, s, ns, (uint32_t)e, e, mine1); h += 2; }
                if (p.PA - h >= 1) wn_poll_class<1>(cx, rs_gl, c8, h, s, ns, (uint32_t)e, e, mine1);
                lds_lg[threadIdx.x] = mine1;
                if (wn_barrier_failed(cx, failflag)) return;
                if (wave == 0) {
                    const float4 v = reinterpret_cast<const float4*>(lds_lg)[lane];
                    logit[0] = v.x; logit[1] = v.y; logit[2] = v.z; logit[3] = v.w;
                }
                wn_lds_barrier();   // (lds_lg is free for the next item)
                if (wave != 0) continue;
            } else {   // the PA partial logits of the stream, summed in the order h = 0 .. PA-1
                int h = 0;
                for (; p.PA - h >= 8; h += 8) wn_poll_logits<8>(cx, rs_gl, lane32, h, s, ns, (uint32_t)e, e, logit);
                if (p.PA - h >= 4) { wn_poll_logits<4>(cx, rs_gl, lane32, h, s, ns, (uint32_t)e, e, logit); h += 4; }
                if (p.PA - h >= 2) { wn_poll_logits<2>(cx, rs_gl, lane32, h, s, ns, (uint32_t)e, e, logit); h += 2; }
                if (p.PA - h >= 1) wn_poll_logits<1>(cx, rs_gl, lane32, h, s, ns, (uint32_t)e, e, logit);
                if (cx.fail) return;
            }
            wn_stamp(r, park, item, 1);
#if WN_SAMPLER_PREFETCH == 0   // (A/B switch: the loads behind the logits' arrival, as before round 4)
            const float temp = r.stream_temps ? r.stream_temps[s + lane0] : r.temperature;
            const bool greedy = r.greedy != 0 || !(temp > 0.f);
            const double u = (g >= 0 && !greedy) ? r.uniforms[(size_t)s * r.num_samples + g + lane0] : 0.;
            const int given = g < 0 ? r.first[(size_t)s * r.n_given + e + lane0] : 0;
#elif WN_SAMPLER_PREFETCH == 2
            const float temp = r.stream_temps ? r.stream_temps[s + lane0] : r.temperature;
            const bool greedy = r.greedy != 0 || !(temp > 0.f);
            const int given = g < 0 ? r.first[(size_t)s * r.n_given + e + lane0] : 0;
#endif
            int idx;
            if (g < 0) {
                idx = given;
            } else {
                if (r.dbg_logits) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) r.dbg_logits[((size_t)s * r.num_samples + g) * 256 + 4 * lane + k] = logit[k];
                }
                idx = wn_sample_1w(r, logit, lane, u, greedy, temp);
                if (lane == 0) r.out_idx[(size_t)s * r.num_samples + g] = idx;
            }
            if (e < r.n_eval) {
#pragma unroll
                for (int m = 0; m < NP; ++m) {
                    const int pi = lane + 64 * m;
                    if (2 * pi < R) {
                        const float2 v = *reinterpret_cast<const float2*>(tab + (size_t)idx * R + 2 * pi);
                        wn_st_pair(rs_g0, (unsigned)(((size_t)s * R + 2 * pi) * 8), (uint32_t)(e + 1), v.x + bias0[m][0], v.y + bias0[m][1], local_i);
                    }
                }
            }
            wn_stamp(r, park, item, 2);
            wn_stamp_flush(r, park, cx.w, item);
        }
    }
}

// G = streams per pipeline item of the layer workgroups (hand-offs are per stream: head and sampler workgroups see the same granules
// different G talk to each other unchanged; the stream count must be a multiple of both)
// amdgpu_num_vgpr(WN_V3_COMPILER_VGPRS): the compiler's own allocation ends below the registers the hand-scheduled blocks keep
// loads in flight into (v152-v167 are RESERVED registers for it: it can neither allocate nor spill into them -- the reservation holds
// by construction, not by the luck of the allocator; the launch bound still gives every lane the 168 the blocks address).  On gfx90a and
// later the backend DOUBLES the attribute's value (unified VGPR + AGPR file) before it compares it with the occupancy limit -- a value above
// 84 is silently dropped at 3 waves per SIMD -- hence WN_V3_COMPILER_VGPRS / 2; build.py disassembles the library and checks the result.
template <int R, int DC, int S, int EC, int P, int G = 1, int SK = 0>
__global__ __launch_bounds__(WN_THREADS_V3) __attribute__((amdgpu_num_vgpr(WN_V3_COMPILER_VGPRS / 2)))
void wn_generate_kernel_v3m(WnPlan p, WnRun r) {
    using SH = WnV2Shape<R, DC, S, EC>;
    extern __shared__ __attribute__((aligned(16))) float wn_lds3m[];
    const int w = p.wg_map[blockIdx.x];
    if (w < 0) return;
    WnCtx cx;
    cx.p = &p; cx.r = &r; cx.lds = wn_lds3m; cx.w = w; cx.fail = 0;
    cx.t_start = (long long)wall_clock64();
    if (wn_not_resident(cx, wn_lds3m)) return;   // (every workgroup of the job is resident from here on)
    const int n_layer_wg = p.NL * p.P;
    if (w < n_layer_wg) {
        wn_v3_layer<SH, P, G, SK>(p, r, cx, wn_lds3m, w / P, w % P);
        return;
    }
    if (threadIdx.x >= WN_THREADS) return;  // the head role is a 256-thread role, the sampler role a one-wave role
    if (w < n_layer_wg + p.PA * p.HR) wn_v3_head<SH, P>(p, r, cx, wn_lds3m, w - n_layer_wg);
#if WN_V3_WIDE_SAMPLER
    else if (p.PA >= WN_V3_WIDE_SAMPLER)
        wn_v3_sampler<SH, true>(p, r, cx, wn_lds3m + WnV3Lds<SH, 1>::smp, wn_lds3m + WnV3Lds<SH, 1>::pre, w - n_layer_wg - p.PA * p.HR, wn_lds3m + WnV3Lds<SH, 1>::sk);
#endif
    else if (threadIdx.x < 64) wn_v3_sampler<SH>(p, r, cx, wn_lds3m + WnV3Lds<SH, 1>::smp, wn_lds3m + WnV3Lds<SH, 1>::pre, w - n_layer_wg - p.PA * p.HR);
}

#endif  // WN_KERNEL_V3_H
