// classfit_asm.hpp — the decider's PLAIN step of the class-ordered best fit (classfit_walk.hpp) as hand-placed gfx950 instructions.
//
// A step's critical path is one instruction after the other of ONE wave (a CU issues to a wave every fourth cycle at best), so its length in
// instructions is its time; the compiler's form of the step ran to ~400 instructions (exec-mask regions around every condition, 0 / 1 round trips
// for ballots, copies of the lane state at every join).  This is the same step for the common case, ~130 instructions: a job of hosts without gpus
// and without constraints whose candidates are all on the board and acceptable, no second offer inside the guard band, and not the placement that
// ends an epoch.  Anything else leaves with status 1 BEFORE any state has changed, and the C++ step does it.
//
// Registers (physical, bound by the asm constraints in classfit_walk.hpp):
//   v[64:79]  the lane: v64 valid, v65 offer, v[66:67] 0.5 / Tc, v[68:69] 0.5 / Tm, v70 class, v71 free cpus, v72 free mem, v73 removals of "its" class wave
//             so far (lanes 58..63), v74 / v75 the last two removed positions, v76 the batch's results (lane = batch slot)
//   v[80:87]  v80 / v81 / v82 the batch's jobs: cpus, mem, meta (lane = batch slot), v83 the lane's column of the board (bytes), v84 the lane number
//   s[36:43]  s36 matched so far, s37 least free cpus of any offer, s[38:39] jobs of the batch some offer lacks room for, s40 least free mem,
//             s41 OUT status: 0 nobody takes the job, 1 not a plain step (nothing changed), 2 placed (one log entry written)
//   s[44:51]  s44 batch lane of the job, s45 the tag a candidate must carry (job << 12 | generation << 8), s46 LDS address of the board's row, s47 the
//             head word, s48 LDS address of the log entry, s49 / s50 least cpus / mem any job asks for, s51 LDS address of the fixed records (CfFixed)
//   s[52:55]  s[52:53] the candidate lanes whose class wave holds hosts without gpus, s[54:55] the overlay's lanes
//   clobbered: v[88:119], s[56:79], vcc, scc
// Offsets into CfFixed (static_asserts in classfit_walk.hpp): ctrl 13728, class table 9472 (56 bytes a class, 0.5 / Tc at 32), the arrays' start 15280.
#pragma once
#define CF_ASM_MAX_STEP(ctrl) "v_max_f32_dpp v113, v113, v113 " ctrl "\n\ts_nop 1\n\t"
#define CF_ASM_DECIDER_STEP                                                                                                             \
  "v_readlane_b32 s56, v80, s44\n\t"                                                                                                    \
  "v_readlane_b32 s57, v81, s44\n\t"                                                                                                    \
  "v_readlane_b32 s58, v82, s44\n\t"                                                                                                    \
  "s_mov_b32 s41, 1\n\t"                                                                                                                \
  "s_and_b32 s59, s58, 0xfffff0ff\n\t"                                                                                                  \
  "s_cmp_lg_u32 s59, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  /* the head word; the candidates' entries */                                                                                         \
  "v_mov_b32_e32 v88, s47\n\t"                                                                                                          \
  "v_mov_b32_e32 v89, s51\n\t"                                                                                                          \
  "v_add_u32_e32 v100, s46, v83\n\t"                                                                                                    \
  "s_mov_b64 exec, 1\n\t"                                                                                                               \
  "ds_write_b32 v89, v88 offset:13732\n\t"                                                                                              \
  "s_mov_b64 exec, -1\n\t"                                                                                                              \
  "ds_read_b32 v92, v100\n\t"                                                                                                           \
  "ds_read_b64 v[94:95], v100 offset:8\n\t"                                                                                             \
  "ds_read_b128 v[96:99], v100 offset:16\n\t"                                                                                           \
  "ds_read_b32 v93, v100\n\t"                                                                                                           \
  /* the overlay's fitness while the loads fly: 1 - ((fc - c) * 0.5 / Tc + (fm - m) * 0.5 / Tm) */                                      \
  "v_subrev_u32_e32 v101, s56, v71\n\t"                                                                                                 \
  "v_subrev_u32_e32 v102, s57, v72\n\t"                                                                                                 \
  "v_cvt_f64_u32_e32 v[104:105], v101\n\t"                                                                                              \
  "v_cvt_f64_u32_e32 v[106:107], v102\n\t"                                                                                              \
  "v_mul_f64 v[104:105], v[104:105], v[66:67]\n\t"                                                                                      \
  "v_mul_f64 v[106:107], v[106:107], v[68:69]\n\t"                                                                                      \
  "v_add_f64 v[104:105], v[104:105], v[106:107]\n\t"                                                                                    \
  "v_add_f64 v[104:105], -v[104:105], 1.0\n\t"                                                                                          \
  "v_cmp_le_u32_e64 s[60:61], s56, v71\n\t"                                                                                             \
  "v_cmp_le_u32_e64 s[62:63], s57, v72\n\t"                                                                                             \
  "v_cmp_ne_u32_e64 s[64:65], 0, v64\n\t"                                                                                               \
  "s_and_b64 s[60:61], s[60:61], s[62:63]\n\t"                                                                                          \
  "s_and_b64 s[60:61], s[60:61], s[64:65]\n\t"                                                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                            \
  /* are the answers there, and acceptable?  tag = want | removals known; up to two unknown removals, neither of the last two positions */ \
  "v_and_b32_e32 v108, 0xffffff00, v92\n\t"                                                                                             \
  "v_cmp_eq_u32_e64 s[66:67], s45, v108\n\t"                                                                                            \
  "v_cmp_eq_u32_e64 s[68:69], v92, v93\n\t"                                                                                             \
  "v_sub_u32_e32 v108, v73, v92\n\t"                                                                                                    \
  "v_and_b32_e32 v108, 0xff, v108\n\t"                                                                                                  \
  "v_cmp_gt_u32_e64 s[70:71], 3, v108\n\t"                                                                                              \
  "v_and_b32_e32 v109, 0xffff, v94\n\t"                                                                                                 \
  "v_cmp_ne_u32_e64 s[72:73], v109, v74\n\t"                                                                                            \
  "v_cmp_ne_u32_e64 s[74:75], v109, v75\n\t"                                                                                            \
  "v_cmp_gt_i32_e64 s[76:77], 0, v95\n\t"                                                                                               \
  "s_and_b64 s[70:71], s[70:71], s[72:73]\n\t"                                                                                          \
  "s_and_b64 s[70:71], s[70:71], s[74:75]\n\t"                                                                                          \
  "s_or_b64 s[70:71], s[70:71], s[76:77]\n\t"                                                                                           \
  "s_and_b64 s[66:67], s[66:67], s[68:69]\n\t"                                                                                          \
  "s_and_b64 s[66:67], s[66:67], s[70:71]\n\t"                                                                                          \
  "s_andn2_b64 s[68:69], s[52:53], s[66:67]\n\t"                                                                                        \
  "s_cmp_lg_u64 s[68:69], 0\n\t"                                                                                                        \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  "s_andn2_b64 s[70:71], s[52:53], s[76:77]\n\t" /* candidates */                                                                       \
  "s_or_b64 s[60:61], s[60:61], s[70:71]\n\t"    /* lanes that can take the job */                                                      \
  "v_cndmask_b32_e64 v110, v98, v104, s[54:55]\n\t"                                                                                     \
  "v_cndmask_b32_e64 v111, v99, v105, s[54:55]\n\t"                                                                                     \
  "v_cndmask_b32_e64 v110, 0, v110, s[60:61]\n\t"                                                                                       \
  "v_cndmask_b32_e64 v111, 0, v111, s[60:61]\n\t"                                                                                       \
  "v_cvt_f32_f64_e32 v112, v[110:111]\n\t"                                                                                              \
  "s_nop 0\n\t"                                                                                                                         \
  "v_mov_b32_e32 v113, v112\n\t"                                                                                                        \
  "s_nop 1\n\t"                                                                                                                         \
  CF_ASM_MAX_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")                                                                     \
  CF_ASM_MAX_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")                                                                     \
  CF_ASM_MAX_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")                                                                         \
  CF_ASM_MAX_STEP("row_mirror row_mask:0xf bank_mask:0xf")                                                                              \
  CF_ASM_MAX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")                                                                            \
  CF_ASM_MAX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")                                                                            \
  "v_readlane_b32 s59, v113, 63\n\t"                                                                                                    \
  "s_mov_b32 s41, 0\n\t"                                                                                                                \
  "s_cmp_eq_u32 s59, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 9f\n\t" /* nobody takes it */                                                                                         \
  "s_mov_b32 s41, 1\n\t"                                                                                                                \
  "v_cmp_eq_f32_e64 s[74:75], s59, v112\n\t"                                                                                            \
  "v_add_f32_e32 v108, 0x35000000, v112\n\t" /* + 2^-21 */                                                                              \
  "v_cmp_le_f32_e64 s[72:73], s59, v108\n\t"                                                                                            \
  "v_and_b32_e32 v109, 0x40000000, v95\n\t"                                                                                             \
  "v_cmp_ne_u32_e64 s[66:67], 0, v109\n\t"                                                                                              \
  "s_and_b64 s[72:73], s[72:73], s[60:61]\n\t" /* lanes whose fitness may round to the greatest */                                     \
  "s_bcnt1_i32_b64 s58, s[72:73]\n\t"                                                                                                   \
  "s_cmp_gt_u32 s58, 1\n\t"                                                                                                             \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  "s_and_b64 s[66:67], s[66:67], s[72:73]\n\t"                                                                                          \
  "s_and_b64 s[66:67], s[66:67], s[70:71]\n\t" /* the winner is a candidate whose class wave saw a possible tie */                      \
  "s_cmp_lg_u64 s[66:67], 0\n\t"                                                                                                        \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  "s_ff1_i32_b64 s78, s[74:75]\n\t"                                                                                                     \
  "s_cmp_lt_u32 s78, 58\n\t"                                                                                                            \
  "s_cbranch_scc0 5f\n\t"                                                                                                               \
  /* ---- an overlay lane wins */                                                                                                       \
  "v_readlane_b32 s60, v71, s78\n\t"                                                                                                    \
  "v_readlane_b32 s61, v72, s78\n\t"                                                                                                    \
  "v_readlane_b32 s62, v65, s78\n\t"                                                                                                    \
  "s_sub_u32 s63, s60, s56\n\t"                                                                                                         \
  "s_sub_u32 s66, s61, s57\n\t"                                                                                                         \
  "v_cmp_eq_u32_e64 vcc, s78, v84\n\t"                                                                                                  \
  "v_mov_b32_e32 v88, s63\n\t"                                                                                                          \
  "v_mov_b32_e32 v89, s66\n\t"                                                                                                          \
  "s_cmp_lt_u32 s63, s49\n\t"                                                                                                           \
  "s_cselect_b32 s67, 1, 0\n\t"                                                                                                         \
  "s_cmp_lt_u32 s66, s50\n\t"                                                                                                           \
  "s_cselect_b32 s58, 1, 0\n\t"                                                                                                         \
  "s_or_b32 s67, s67, s58\n\t"                                                                                                          \
  "v_cndmask_b32_e32 v71, v71, v88, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v72, v72, v89, vcc\n\t"                                                                                            \
  "s_cmp_lg_u32 s67, 0\n\t"                                                                                                             \
  "s_cselect_b64 s[58:59], vcc, 0\n\t" /* a lane that cannot take the smallest job any more is free again */                            \
  "s_mov_b32 s68, s44\n\t"                                                                                                              \
  "s_mov_b32 s69, 0\n\t"                                                                                                                \
  "v_cndmask_b32_e64 v64, v64, 0, s[58:59]\n\t"                                                                                         \
  "s_branch 7f\n\t"                                                                                                                     \
  /* ---- a class wave's candidate wins: the member leaves its arrays (zeroed, the wave's count moves on), an overlay lane opens */      \
  "5:\n\t"                                                                                                                              \
  "v_readlane_b32 s60, v96, s78\n\t"                                                                                                    \
  "v_readlane_b32 s61, v97, s78\n\t"                                                                                                    \
  "v_readlane_b32 s72, v95, s78\n\t"                                                                                                    \
  "v_readlane_b32 s73, v94, s78\n\t"                                                                                                    \
  "v_readlane_b32 s74, v73, s78\n\t"                                                                                                    \
  "s_sub_u32 s63, s60, s56\n\t"                                                                                                         \
  "s_sub_u32 s66, s61, s57\n\t"                                                                                                         \
  "s_and_b32 s62, s72, 0x3fff\n\t"                                                                                                      \
  "s_bfe_u32 s75, s72, 0x80010\n\t"                                                                                                     \
  "s_and_b32 s69, s73, 0xffff\n\t"                                                                                                      \
  "s_lshr_b32 s77, s73, 16\n\t"                                                                                                         \
  "s_sub_u32 s76, s78, 57\n\t"                                                                                                          \
  "s_cmp_lt_u32 s63, s49\n\t"                                                                                                           \
  "s_cselect_b32 s67, 1, 0\n\t"                                                                                                         \
  "s_cmp_lt_u32 s66, s50\n\t"                                                                                                           \
  "s_cselect_b32 s58, 1, 0\n\t"                                                                                                         \
  "s_or_b32 s67, s67, s58\n\t"                                                                                                          \
  "s_bcnt1_i32_b64 s58, s[64:65]\n\t"                                                                                                   \
  "s_cmp_lg_u32 s67, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 6f\n\t"                                                                                                               \
  "s_cmp_ge_u32 s58, " CF_ASM_EPOCH_LIVE "\n\t"                                                                                          \
  "s_cbranch_scc1 9f\n\t" /* the placement that fills the overlay: the epoch's end is the C++ step's */                                 \
  "6:\n\t"                                                                                                                              \
  "s_lshl_b32 s58, s69, 3\n\t"                                                                                                          \
  "s_add_u32 s58, s58, s51\n\t"                                                                                                         \
  "s_lshl_b32 s59, s76, 2\n\t"                                                                                                          \
  "s_add_u32 s59, s59, s51\n\t"                                                                                                         \
  "s_add_u32 s74, s74, 1\n\t"                                                                                                           \
  "v_mov_b32_e32 v88, 0\n\t"                                                                                                            \
  "v_mov_b32_e32 v89, 0\n\t"                                                                                                            \
  "v_mov_b32_e32 v90, s58\n\t"                                                                                                          \
  "v_mov_b32_e32 v91, s59\n\t"                                                                                                          \
  "v_mov_b32_e32 v100, s74\n\t"                                                                                                         \
  "s_mov_b64 exec, 1\n\t"                                                                                                               \
  "ds_write_b64 v90, v[88:89] offset:15280\n\t"                                                                                         \
  "ds_write_b32 v91, v100 offset:13732\n\t"                                                                                             \
  "s_mov_b64 exec, -1\n\t"                                                                                                              \
  "v_cmp_eq_u32_e64 vcc, s78, v84\n\t"                                                                                                  \
  "v_mov_b32_e32 v101, s69\n\t"                                                                                                         \
  "s_nop 0\n\t"                                                                                                                         \
  "v_cndmask_b32_e32 v75, v75, v74, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v74, v74, v101, vcc\n\t"                                                                                           \
  "v_cndmask_b32_e32 v73, v73, v100, vcc\n\t"                                                                                           \
  "s_lshl_b32 s58, s76, 8\n\t"                                                                                                          \
  "s_lshl_b32 s59, s77, 12\n\t"                                                                                                         \
  "s_or_b32 s68, s44, s58\n\t"                                                                                                          \
  "s_or_b32 s68, s68, s59\n\t"                                                                                                          \
  "s_cmp_lg_u32 s67, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 7f\n\t" /* dead at once: nothing opens */                                                                             \
  "s_mul_i32 s58, s75, 56\n\t"                                                                                                          \
  "s_add_u32 s58, s58, s51\n\t"                                                                                                         \
  "v_mov_b32_e32 v90, s58\n\t"                                                                                                          \
  "ds_read_b128 v[116:119], v90 offset:9504\n\t"                                                                                        \
  "s_andn2_b64 s[58:59], s[54:55], s[64:65]\n\t"                                                                                        \
  "s_ff1_i32_b64 s79, s[58:59]\n\t"                                                                                                     \
  "v_cmp_eq_u32_e64 vcc, s79, v84\n\t"                                                                                                  \
  "v_mov_b32_e32 v88, s62\n\t"                                                                                                          \
  "v_mov_b32_e32 v89, s75\n\t"                                                                                                          \
  "v_mov_b32_e32 v91, s63\n\t"                                                                                                          \
  "v_mov_b32_e32 v101, s66\n\t"                                                                                                         \
  "v_cndmask_b32_e64 v64, v64, 1, vcc\n\t"                                                                                              \
  "v_cndmask_b32_e32 v65, v65, v88, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v70, v70, v89, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v71, v71, v91, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v72, v72, v101, vcc\n\t"                                                                                           \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                            \
  "v_cndmask_b32_e32 v66, v66, v116, vcc\n\t"                                                                                           \
  "v_cndmask_b32_e32 v67, v67, v117, vcc\n\t"                                                                                           \
  "v_cndmask_b32_e32 v68, v68, v118, vcc\n\t"                                                                                           \
  "v_cndmask_b32_e32 v69, v69, v119, vcc\n\t"                                                                                           \
  /* ---- the books of a placement: the result, the jobs behind it some offer has no room for any more, the least free values, the log */ \
  "7:\n\t"                                                                                                                              \
  "v_cmp_eq_u32_e64 vcc, s44, v84\n\t"                                                                                                  \
  "v_mov_b32_e32 v88, s62\n\t"                                                                                                          \
  "v_cmp_lt_u32_e64 s[58:59], s63, v80\n\t"                                                                                             \
  "v_cmp_lt_u32_e64 s[72:73], s66, v81\n\t"                                                                                             \
  "v_cndmask_b32_e32 v76, v76, v88, vcc\n\t"                                                                                            \
  "s_or_b64 s[58:59], s[58:59], s[72:73]\n\t"                                                                                           \
  "s_lshl_b64 s[72:73], -2, s44\n\t"                                                                                                    \
  "s_and_b64 s[58:59], s[58:59], s[72:73]\n\t"                                                                                          \
  "s_or_b64 s[38:39], s[38:39], s[58:59]\n\t"                                                                                           \
  "s_min_u32 s37, s37, s63\n\t"                                                                                                         \
  "s_min_u32 s40, s40, s66\n\t"                                                                                                         \
  "s_add_u32 s36, s36, 1\n\t"                                                                                                           \
  "v_mov_b32_e32 v88, s68\n\t"                                                                                                          \
  "v_mov_b32_e32 v89, s69\n\t"                                                                                                          \
  "v_mov_b32_e32 v90, s60\n\t"                                                                                                          \
  "v_mov_b32_e32 v91, s61\n\t"                                                                                                          \
  "v_mov_b32_e32 v100, s48\n\t"                                                                                                         \
  "s_mov_b64 exec, 1\n\t"                                                                                                               \
  "ds_write_b128 v100, v[88:91]\n\t"                                                                                                    \
  "s_mov_b64 exec, -1\n\t"                                                                                                              \
  "s_mov_b32 s41, 2\n\t"                                                                                                                \
  "9:\n\t"
