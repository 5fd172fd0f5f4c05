// fast_loop_asm.h -- the FAST stepping loop of trace_ray<true> spelled in gfx950 assembly (one extended-asm statement).
//
// The arithmetic is rk4_planar_position + rk4_planar_velocity of trace_device.h, instruction for instruction what hipcc makes of them
// (66 f64 VALU per step, 4 of them v_rsq_f64; with BS_FL_SERIES, below, stages 1 and 3 get their r^-5 from a neighbouring evaluation instead:
// 65 + 2); what is written by hand is everything AROUND the arithmetic, which the compiler cannot be talked into
// (profiles/EXPERIMENTS.md 1.2, 6.6):
//   * the state is updated IN PLACE, y and r^2 ping-pong between two register pairs across the two copies of the step: no v_mov at all
//     (the compiled loop carries one v_mov_b64 per step);
//   * the wavefront's scalar bookkeeping is 7 SALU and two never-taken branches per step (compiled: 13 SALU, three branches that fall
//     through and one that is taken per step), and the ONLY taken branch is the back edge, once per two steps;
//   * the rare events -- some lane's guard fired, some lane crossed the disk plane -- LEAVE the statement: the C++ around it (trace_ray)
//     snapshots / records and re-enters.  Nothing rare sits between the steps, so where the loop's second step lands is no longer a matter of
//     what the rare blocks happen to weigh (a 1.5 % lottery per build, profiles/r06_code_alignment_ab.txt).
//
// Operands (all 64-bit pairs except it / maxs / ev):
//   VGPR state   x, y, wx, wy, r2      "+v"   the planar state in the ray's own units; on exit: the state the NEXT step starts from
//   VGPR scratch yb, r2b, t0..t7       "=&v"  on a crossing exit t0 = y and t1 = r^2 BEFORE the step that crossed
//   VGPR consts  c25, lo, hi, thr      "v"    2.5; the per-lane guard thresholds; the crossing threshold (0 or -inf)
//   SGPR consts  c4375, m23, maxs, amask
//   BS_FL_SERIES q4, iq4, c4 "+v" (stage 4's squared radius, its reciprocal, its r^-5: carried from step to step), c6 "v" (-105/16), thr15 "s" (2^-15);
//                t8 "=&v", c8 "s" (3465/384), thr12 "s" (2^-12) for stage 3
//   SGPR state   ok "+s" (guards of the state about to be stepped), it "+s"
//   SGPR out     go (the lanes that go on), crossed (the crossing ballot), ev (0: a guard fired BEFORE a step, nothing stepped; 1: a step crossed)
// Clobbers vcc, scc.  exec is not touched: the steps run unmasked (finished lanes free-run, trace_device.h "per-lane LDS scratch").
//
// Hazards (gfx940 family, what LLVM's GCNHazardRecognizer would have checked): a v_rsq_f64 result is first read two instructions later
// (trans -> VALU forwarding needs one); SALU reads of VALU-written SGPRs and s_cbranch_vccnz after a v_cmp are interlocked in hardware.
#pragma once

#ifndef BS_FL_SERIES
#define BS_FL_SERIES 2  // r^-5 by a series from a neighbouring evaluation where the two squared radii agree closely enough: 1 = stage 1 from the previous
                        // step's stage 4 (95 % of all steps), 2 = also stage 3 from stage 2 (94 %); 0 = every stage its own v_rsq_f64
#endif

// go = (it < maxs) ? amask & ok : 0;  SCC = (go != amask)
#define BS_FL_HEAD                              \
    "s_and_b64 %[go], %[amask], %[ok]\n\t"      \
    "s_cmp_lt_i32 %[it], %[maxs]\n\t"           \
    "s_cselect_b64 %[go], %[go], 0\n\t"         \
    "s_cmp_lg_u64 %[go], %[amask]\n\t"

// the guards of the new state (v_cmp into ok / go) become the next step's ok; it counts the step; then the next step's head
#define BS_FL_SCALAR                            \
    "s_and_b64 %[ok], %[ok], %[go]\n\t"         \
    "s_add_i32 %[it], %[it], 1\n\t"             \
    BS_FL_HEAD

// stages 1-3, the new position (x in place, y -> YN, r^2 -> R2N), the three compares, stage 4 up to the sums T = S + R
#define BS_FL_PART1(Y, R2, YN, R2N)                                  \
    "v_rsq_f64 " YN ", " R2 "\n\t"                                   \
    "v_fma_f64 " R2N ", 0.5, %[wx], %[x]\n\t"                        \
    "v_fma_f64 %[t0], 0.5, %[wy], " Y "\n\t"                         \
    "v_mul_f64 %[t1], " YN ", " YN "\n\t"                            \
    "v_fma_f64 %[t2], -" R2 ", %[t1], 1.0\n\t"                       \
    "v_mul_f64 %[t1], %[t1], %[t1]\n\t"                              \
    "v_mul_f64 " YN ", " YN ", %[t1]\n\t"                            \
    "v_mul_f64 %[t1], " R2N ", " R2N "\n\t"                          \
    "v_fmac_f64 %[t1], %[t0], %[t0]\n\t"                             \
    "v_rsq_f64 %[t3], %[t1]\n\t"                                     \
    "v_fma_f64 %[t4], %[c4375], %[t2], %[c25]\n\t"                   \
    "v_mul_f64 %[t2], %[t2], " YN "\n\t"                             \
    "v_fmac_f64 " YN ", %[t2], %[t4]\n\t"                            \
    "v_fma_f64 %[t4], -" YN ", %[x], " R2N "\n\t"                    \
    "v_fma_f64 %[t5], -" YN ", " Y ", %[t0]\n\t"                     \
    "v_mul_f64 %[t6], %[t4], %[t4]\n\t"                              \
    "v_mul_f64 %[t2], %[t3], %[t3]\n\t"                              \
    "v_fmac_f64 %[t6], %[t5], %[t5]\n\t"                             \
    "v_fma_f64 %[t1], -%[t1], %[t2], 1.0\n\t"                        \
    "v_mul_f64 %[t2], %[t2], %[t2]\n\t"                              \
    "v_rsq_f64 %[t7], %[t6]\n\t"                                     \
    "v_mul_f64 %[t2], %[t3], %[t2]\n\t"                              \
    "v_fma_f64 %[t3], %[c4375], %[t1], %[c25]\n\t"                   \
    "v_mul_f64 %[t1], %[t1], %[t2]\n\t"                              \
    "v_fmac_f64 %[t2], %[t1], %[t3]\n\t"                             \
    "v_mul_f64 %[t1], " R2N ", %[t2]\n\t"                            \
    "v_mul_f64 " R2N ", %[t7], %[t7]\n\t"                            \
    "v_mul_f64 %[t0], %[t0], %[t2]\n\t"                              \
    "v_fma_f64 %[t2], -%[t6], " R2N ", 1.0\n\t"                      \
    "v_mul_f64 " R2N ", " R2N ", " R2N "\n\t"                        \
    "v_mul_f64 " R2N ", %[t7], " R2N "\n\t"                          \
    "v_fma_f64 %[t3], %[c4375], %[t2], %[c25]\n\t"                   \
    "v_mul_f64 %[t2], %[t2], " R2N "\n\t"                            \
    "v_fmac_f64 " R2N ", %[t2], %[t3]\n\t"                           \
    "v_add_f64 %[t2], %[wx], %[x]\n\t"                               \
    "v_add_f64 %[t3], %[wy], " Y "\n\t"                              \
    "v_fma_f64 %[t6], -2.0, %[t1], %[t2]\n\t"                        \
    "v_fmac_f64 %[t1], " R2N ", %[t4]\n\t"                           \
    "v_fma_f64 %[t7], -2.0, %[t0], %[t3]\n\t"                        \
    "v_fmac_f64 %[t0], " R2N ", %[t5]\n\t"                           \
    "v_fma_f64 %[t4], " YN ", %[x], %[t1]\n\t"                       \
    "v_fma_f64 %[x], %[m23], %[t4], %[t2]\n\t"                       \
    "v_fma_f64 %[t5], " YN ", " Y ", %[t0]\n\t"                      \
    "v_mul_f64 " R2N ", %[x], %[x]\n\t"                              \
    "v_fma_f64 " YN ", %[m23], %[t5], %[t3]\n\t"                     \
    "v_fmac_f64 " R2N ", " YN ", " YN "\n\t"                         \
    "v_mul_f64 %[t2], " Y ", " YN "\n\t"                             \
    "v_cmp_nlt_f64 %[ok], " R2N ", %[lo]\n\t"                        \
    "v_cmp_ngt_f64 %[go], " R2N ", %[hi]\n\t"                        \
    "v_cmp_le_f64 vcc, %[t2], %[thr]\n\t"                            \
    "v_mul_f64 %[t2], %[t6], %[t6]\n\t"                              \
    "v_fmac_f64 %[t2], %[t7], %[t7]\n\t"                             \
    "v_rsq_f64 %[t3], %[t2]\n\t"                                     \
    "v_add_f64 %[t1], %[t1], %[t4]\n\t"                              \
    "v_add_f64 %[t0], %[t0], %[t5]\n\t"

// the rest of stage 4 and the new displacement per step
#define BS_FL_PART2                                                  \
    "v_mul_f64 %[t4], %[t3], %[t3]\n\t"                              \
    "v_fma_f64 %[t2], -%[t2], %[t4], 1.0\n\t"                        \
    "v_mul_f64 %[t4], %[t4], %[t4]\n\t"                              \
    "v_mul_f64 %[t3], %[t3], %[t4]\n\t"                              \
    "v_fma_f64 %[t4], %[c4375], %[t2], %[c25]\n\t"                   \
    "v_mul_f64 %[t2], %[t2], %[t3]\n\t"                              \
    "v_fmac_f64 %[t3], %[t2], %[t4]\n\t"                             \
    "v_fmac_f64 %[t1], %[t3], %[t6]\n\t"                             \
    "v_fmac_f64 %[t0], %[t3], %[t7]\n\t"                             \
    "v_fmac_f64 %[wx], %[m23], %[t1]\n\t"                            \
    "v_fmac_f64 %[wy], %[m23], %[t0]\n\t"

// ---- BS_FL_SERIES: three transcendentals per step instead of four ---------------------------------------------------------------------
// Stage 4 of a step evaluates r^-5 at the PREDICTED end point p4, stage 1 of the next step at the end point itself: the two squared radii
// differ by delta = q1 / q4 - 1, 2e-8 in the median of a frame's steps and below 2^-15 in 95 % of them (all but the few steps next to the
// hole).  There c1 = c4 (1 + delta)^(-5/2) = c4 (1 - 5/2 d + 35/8 d^2 - 105/16 d^3), truncation < 9.1 d^4 < 8e-18: seven full-rate
// instructions (the difference, its quotient by q4 -- 1 / q4 = y0^2 (1 + e) comes out of stage 4 for one FMA --, the compare, three FMAs,
// the product) instead of v_rsq_f64 (a quarter of the rate) + 7.  c4 is stage 4's own value, fresh from its v_rsq_f64 every step: nothing
// accumulates.  If ANY live lane's delta is larger (wave-uniform branch; finished lanes and their NaNs are masked out with amask) the
// wavefront visits the out-of-line block, which redoes THOSE lanes with the old sequence.  The first step of a ray has no stage 4 behind it: q4 = 1, 1 / q4 = inf
// make its delta infinite.  62 + 4 -> 63 + 3 per step: 312 -> 300 issue cycles.
#define BS_FL_S_HEAD(Y, R2, YN, R2N, TAG)                              \
    "v_add_f64 %[t2], " R2 ", -%[q4]\n\t"                            \
    "v_fma_f64 " R2N ", 0.5, %[wx], %[x]\n\t"                        \
    "v_fma_f64 %[t0], 0.5, %[wy], " Y "\n\t"                         \
    "v_mul_f64 %[t2], %[t2], %[iq4]\n\t"                             \
    "v_mul_f64 %[t1], " R2N ", " R2N "\n\t"                          \
    "v_fmac_f64 %[t1], %[t0], %[t0]\n\t"                             \
    "v_cmp_lt_f64 vcc, |%[t2]|, %[thr15]\n\t"                        \
    "v_rsq_f64 %[t3], %[t1]\n\t"                                     \
    "s_andn2_b64 %[crossed], %[amask], vcc\n\t"                      \
    "v_fma_f64 %[t4], %[c6], %[t2], %[c4375]\n\t"                    \
    "v_fma_f64 %[t4], %[t4], %[t2], -%[c25]\n\t"                     \
    "v_fma_f64 %[t4], %[t4], %[t2], 1.0\n\t"                         \
    "v_mul_f64 " YN ", %[c4], %[t4]\n\t"                             \
    "s_cbranch_scc1 .Lbs_slow_" TAG "%=\n"                            \
    ".Lbs_join_" TAG "%=:\n\t"

// the rest of stage 2 and stage 3 with its own v_rsq_f64
#define BS_FL_S_MID(Y, R2, YN, R2N, TAG)                             \
    "v_fma_f64 %[t4], -" YN ", %[x], " R2N "\n\t"                    \
    "v_fma_f64 %[t5], -" YN ", " Y ", %[t0]\n\t"                     \
    "v_mul_f64 %[t6], %[t4], %[t4]\n\t"                              \
    "v_mul_f64 %[t2], %[t3], %[t3]\n\t"                              \
    "v_fmac_f64 %[t6], %[t5], %[t5]\n\t"                             \
    "v_fma_f64 %[t1], -%[t1], %[t2], 1.0\n\t"                        \
    "v_mul_f64 %[t2], %[t2], %[t2]\n\t"                              \
    "v_rsq_f64 %[t7], %[t6]\n\t"                                     \
    "v_mul_f64 %[t2], %[t3], %[t2]\n\t"                              \
    "v_fma_f64 %[t3], %[c4375], %[t1], %[c25]\n\t"                   \
    "v_mul_f64 %[t1], %[t1], %[t2]\n\t"                              \
    "v_fmac_f64 %[t2], %[t1], %[t3]\n\t"                             \
    "v_mul_f64 %[t1], " R2N ", %[t2]\n\t"                            \
    "v_mul_f64 " R2N ", %[t7], %[t7]\n\t"                            \
    "v_mul_f64 %[t0], %[t0], %[t2]\n\t"                              \
    "v_fma_f64 %[t2], -%[t6], " R2N ", 1.0\n\t"                      \
    "v_mul_f64 " R2N ", " R2N ", " R2N "\n\t"                        \
    "v_mul_f64 " R2N ", %[t7], " R2N "\n\t"                          \
    "v_fma_f64 %[t3], %[c4375], %[t2], %[c25]\n\t"                   \
    "v_mul_f64 %[t2], %[t2], " R2N "\n\t"                            \
    "v_fmac_f64 " R2N ", %[t2], %[t3]\n\t"

// BS_FL_SERIES 2: stage 3's r^-5 from stage 2's as well.  p3 = p2 - c1 p, so q3 / q2 - 1 is about -2 r^-5 in the ray's units: below 2^-12 in 94 % of
// a frame's steps; there c3 = c2 (1 - 5/2 d + 35/8 d^2 - 105/16 d^3 + 3465/384 d^4), truncation < 12.4 d^5 < 1.1e-17.  Same arrangement as stage 1:
// the lanes whose delta is larger redo the old sequence out of line under exec &= ~vcc.  63 + 3 -> 65 + 2 per step: 300 -> 292 issue cycles.
#define BS_FL_S_MID2(Y, R2, YN, R2N, TAG)                            \
    "v_fma_f64 %[t4], -" YN ", %[x], " R2N "\n\t"                    \
    "v_fma_f64 %[t5], -" YN ", " Y ", %[t0]\n\t"                     \
    "v_mul_f64 %[t6], %[t4], %[t4]\n\t"                              \
    "v_mul_f64 %[t2], %[t3], %[t3]\n\t"                              \
    "v_fmac_f64 %[t6], %[t5], %[t5]\n\t"                             \
    "v_add_f64 %[t7], %[t6], -%[t1]\n\t"                             \
    "v_fma_f64 %[t1], -%[t1], %[t2], 1.0\n\t"                        \
    "v_fma_f64 %[t8], %[t2], %[t1], %[t2]\n\t"                       \
    "v_mul_f64 %[t2], %[t2], %[t2]\n\t"                              \
    "v_mul_f64 %[t7], %[t7], %[t8]\n\t"                              \
    "v_mul_f64 %[t2], %[t3], %[t2]\n\t"                              \
    "v_fma_f64 %[t3], %[c4375], %[t1], %[c25]\n\t"                   \
    "v_cmp_lt_f64 vcc, |%[t7]|, %[thr12]\n\t"                        \
    "v_mul_f64 %[t1], %[t1], %[t2]\n\t"                              \
    "s_andn2_b64 %[crossed], %[amask], vcc\n\t"                      \
    "v_fmac_f64 %[t2], %[t1], %[t3]\n\t"                             \
    "v_fma_f64 %[t8], %[c8], %[t7], %[c6]\n\t"                       \
    "v_fma_f64 %[t8], %[t8], %[t7], %[c4375]\n\t"                    \
    "v_fma_f64 %[t8], %[t8], %[t7], -%[c25]\n\t"                     \
    "v_fma_f64 %[t8], %[t8], %[t7], 1.0\n\t"                         \
    "v_mul_f64 %[t1], " R2N ", %[t2]\n\t"                            \
    "v_mul_f64 %[t0], %[t0], %[t2]\n\t"                              \
    "v_mul_f64 " R2N ", %[t2], %[t8]\n\t"                            \
    "s_cbranch_scc1 .Lbs_slow3_" TAG "%=\n"                           \
    ".Lbs_join3_" TAG "%=:\n\t"

// stage 3 out of line for the lanes that need it (vcc = the lanes that are fine): the old sequence on q3 = t6, its r^-5 into R2N
#define BS_FL_SLOW3(R2N, TAG)                                        \
    ".Lbs_slow3_" TAG "%=:\n\t"                                      \
    "s_mov_b64 %[crossed], exec\n\t"                                 \
    "s_andn2_b64 exec, exec, vcc\n\t"                                \
    "v_rsq_f64 %[t7], %[t6]\n\t"                                     \
    "s_nop 0\n\t"                                                    \
    "v_mul_f64 " R2N ", %[t7], %[t7]\n\t"                            \
    "v_fma_f64 %[t2], -%[t6], " R2N ", 1.0\n\t"                      \
    "v_mul_f64 " R2N ", " R2N ", " R2N "\n\t"                        \
    "v_mul_f64 " R2N ", %[t7], " R2N "\n\t"                          \
    "v_fma_f64 %[t3], %[c4375], %[t2], %[c25]\n\t"                   \
    "v_mul_f64 %[t2], %[t2], " R2N "\n\t"                            \
    "v_fmac_f64 " R2N ", %[t2], %[t3]\n\t"                           \
    "s_mov_b64 exec, %[crossed]\n\t"                                 \
    "s_branch .Lbs_join3_" TAG "%=\n"

// from p + w on: p4, the sums, the new position, the compares, stage 4, the new displacement
#define BS_FL_S_TAIL(Y, R2, YN, R2N, TAG)                            \
    "v_add_f64 %[t2], %[wx], %[x]\n\t"                               \
    "v_add_f64 %[t3], %[wy], " Y "\n\t"                              \
    "v_fma_f64 %[t6], -2.0, %[t1], %[t2]\n\t"                        \
    "v_fmac_f64 %[t1], " R2N ", %[t4]\n\t"                           \
    "v_fma_f64 %[t7], -2.0, %[t0], %[t3]\n\t"                        \
    "v_fmac_f64 %[t0], " R2N ", %[t5]\n\t"                           \
    "v_fma_f64 %[t4], " YN ", %[x], %[t1]\n\t"                       \
    "v_fma_f64 %[x], %[m23], %[t4], %[t2]\n\t"                       \
    "v_fma_f64 %[t5], " YN ", " Y ", %[t0]\n\t"                      \
    "v_mul_f64 " R2N ", %[x], %[x]\n\t"                              \
    "v_fma_f64 " YN ", %[m23], %[t5], %[t3]\n\t"                     \
    "v_fmac_f64 " R2N ", " YN ", " YN "\n\t"                         \
    "v_mul_f64 %[t2], " Y ", " YN "\n\t"                             \
    "v_cmp_nlt_f64 %[ok], " R2N ", %[lo]\n\t"                        \
    "v_cmp_ngt_f64 %[go], " R2N ", %[hi]\n\t"                        \
    "v_cmp_le_f64 vcc, %[t2], %[thr]\n\t"                            \
    "v_mul_f64 %[q4], %[t6], %[t6]\n\t"                              \
    "v_fmac_f64 %[q4], %[t7], %[t7]\n\t"                             \
    "v_rsq_f64 %[t3], %[q4]\n\t"                                     \
    "v_add_f64 %[t1], %[t1], %[t4]\n\t"                              \
    "v_add_f64 %[t0], %[t0], %[t5]\n\t"                              \
    "v_mul_f64 %[t4], %[t3], %[t3]\n\t"                              \
    "v_fma_f64 %[t2], -%[q4], %[t4], 1.0\n\t"                        \
    "v_fma_f64 %[iq4], %[t4], %[t2], %[t4]\n\t"                      \
    "v_mul_f64 %[t4], %[t4], %[t4]\n\t"                              \
    "v_mul_f64 %[t3], %[t3], %[t4]\n\t"                              \
    "v_fma_f64 %[t4], %[c4375], %[t2], %[c25]\n\t"                   \
    "v_mul_f64 %[t2], %[t2], %[t3]\n\t"                              \
    "v_fma_f64 %[c4], %[t2], %[t4], %[t3]\n\t"                       \
    "v_fmac_f64 %[t1], %[c4], %[t6]\n\t"                             \
    "v_fmac_f64 %[t0], %[c4], %[t7]\n\t"                             \
    "v_fmac_f64 %[wx], %[m23], %[t1]\n\t"                            \
    "v_fmac_f64 %[wy], %[m23], %[t0]\n\t"

// The out-of-line stage 1 for the LANES whose radius has moved too far from the previous stage 4's (vcc = the lanes that are fine): the old
// sequence, same bits as without BS_FL_SERIES, under exec &= ~vcc -- the other lanes keep their series value.  Which of the two a ray gets is
// decided by the ray's own delta alone: its result does not depend on the rays that share its wavefront (row bands, split frames and
// whole frames must stay bit-identical).
#define BS_FL_SLOW(R2, YN, TAG)                                      \
    ".Lbs_slow_" TAG "%=:\n\t"                                       \
    "s_mov_b64 %[crossed], exec\n\t"                                 \
    "s_andn2_b64 exec, exec, vcc\n\t"                                \
    "v_rsq_f64 " YN ", " R2 "\n\t"                                   \
    "s_nop 0\n\t"                                                    \
    "v_mul_f64 %[t4], " YN ", " YN "\n\t"                            \
    "v_fma_f64 %[t2], -" R2 ", %[t4], 1.0\n\t"                       \
    "v_mul_f64 %[t4], %[t4], %[t4]\n\t"                              \
    "v_mul_f64 " YN ", " YN ", %[t4]\n\t"                            \
    "v_fma_f64 %[t4], %[c4375], %[t2], %[c25]\n\t"                   \
    "v_mul_f64 %[t2], %[t2], " YN "\n\t"                             \
    "v_fmac_f64 " YN ", %[t2], %[t4]\n\t"                            \
    "s_mov_b64 exec, %[crossed]\n\t"                                 \
    "s_branch .Lbs_join_" TAG "%=\n"

#if BS_FL_SERIES == 2
#define BS_FL_STEP(Y, R2, YN, R2N, TAG) BS_FL_S_HEAD(Y, R2, YN, R2N, TAG) BS_FL_S_MID2(Y, R2, YN, R2N, TAG) BS_FL_S_TAIL(Y, R2, YN, R2N, TAG) BS_FL_SCALAR
#elif BS_FL_SERIES
#define BS_FL_STEP(Y, R2, YN, R2N, TAG) BS_FL_S_HEAD(Y, R2, YN, R2N, TAG) BS_FL_S_MID(Y, R2, YN, R2N, TAG) BS_FL_S_TAIL(Y, R2, YN, R2N, TAG) BS_FL_SCALAR
#else
#define BS_FL_STEP(Y, R2, YN, R2N, TAG) BS_FL_PART1(Y, R2, YN, R2N) BS_FL_PART2 BS_FL_SCALAR
#endif

// step A: (y, r2) -> (yb, r2b); step B: back.  After each: leave if the step crossed, leave (before stepping) if the next step's guards fire.
#define BS_FL_BODY                                                   \
    BS_FL_STEP("%[y]", "%[r2]", "%[yb]", "%[r2b]", "a")              \
    "s_cbranch_vccnz .Lbs_cross_a%=\n\t"                             \
    "s_cbranch_scc1 .Lbs_guard_b%=\n\t"                              \
    BS_FL_STEP("%[yb]", "%[r2b]", "%[y]", "%[r2]", "b")              \
    "s_cbranch_vccnz .Lbs_cross_b%=\n\t"                             \
    "s_cbranch_scc1 .Lbs_guard_a%=\n\t"

#if BS_FL_SERIES == 2
#define BS_FL_SLOW_BLOCKS BS_FL_SLOW("%[r2]", "%[yb]", "a") BS_FL_SLOW("%[r2b]", "%[y]", "b") BS_FL_SLOW3("%[r2b]", "a") BS_FL_SLOW3("%[r2]", "b")
#elif BS_FL_SERIES
#define BS_FL_SLOW_BLOCKS BS_FL_SLOW("%[r2]", "%[yb]", "a") BS_FL_SLOW("%[r2b]", "%[y]", "b")
#else
#define BS_FL_SLOW_BLOCKS
#endif

#define BS_FAST_LOOP_ASM                                             \
    BS_FL_HEAD                                                       \
    "s_cbranch_scc1 .Lbs_guard_a%=\n"                                \
    ".Lbs_loop%=:\n\t"                                               \
    BS_FL_BODY                                                       \
    "s_branch .Lbs_loop%=\n"                                         \
    BS_FL_SLOW_BLOCKS                                                \
    ".Lbs_cross_a%=:\n\t"                                            \
    "v_mov_b64 %[t0], %[y]\n\t"                                      \
    "v_mov_b64 %[t1], %[r2]\n\t"                                     \
    "v_mov_b64 %[y], %[yb]\n\t"                                      \
    "v_mov_b64 %[r2], %[r2b]\n\t"                                    \
    "s_branch .Lbs_cross%=\n"                                        \
    ".Lbs_cross_b%=:\n\t"                                            \
    "v_mov_b64 %[t0], %[yb]\n\t"                                     \
    "v_mov_b64 %[t1], %[r2b]\n"                                      \
    ".Lbs_cross%=:\n\t"                                              \
    "s_mov_b64 %[crossed], vcc\n\t"                                  \
    "s_mov_b32 %[ev], 1\n\t"                                         \
    "s_branch .Lbs_done%=\n"                                         \
    ".Lbs_guard_b%=:\n\t"                                            \
    "v_mov_b64 %[y], %[yb]\n\t"                                      \
    "v_mov_b64 %[r2], %[r2b]\n"                                      \
    ".Lbs_guard_a%=:\n\t"                                            \
    "s_mov_b32 %[ev], 0\n"                                           \
    ".Lbs_done%=:\n"
