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
//   s[52:59]  s[52:53] the candidate lanes whose class wave holds hosts without gpus, s[54:55] the overlay's lanes, s56 LDS address of the offers' attribute bytes
//             (v85 / v86: the batch's jobs' EQUALS constraints, two per word)
//   clobbered: v[88:119], s[60:83], vcc, scc
// Offsets into CfFixed (static_asserts in classfit_walk.hpp): ctrl 13728, class table 9472 (56 bytes a class, 0.5 / Tc at 32), the arrays' start 15280.
#pragma once
// the board's rows (CF_SLOTS_N, classfit_walk.hpp) move everything behind the board in CfFixed: the offsets as strings, per row count
#if CF_SLOTS_N == 12
#define CF_O_CTRL "13728"
#define CF_O_CTRL4 "13732"
#define CF_O_ARR "15280"
#define CF_O_CLS32 "9504"
#define CF_SLOT_SHIFT "9"
#define CF_SLOT_N "12"
#elif CF_SLOTS_N == 24
#define CF_O_CTRL "16800"
#define CF_O_CTRL4 "16804"
#define CF_O_ARR "18352"
#define CF_O_CLS32 "12576"
#define CF_SLOT_SHIFT "10"
#define CF_SLOT_N "24"
#else
#error "CF_SLOTS_N: 12 or 24"
#endif
#ifndef CF_ASM_WAIT_READS
#define CF_ASM_WAIT_READS "30"  // (measured on a C4 pool: 44.64 / 44.05 / 43.86 ms at 4 / 12 / 30)
#endif
#define CF_ASM_MAX_STEP(ctrl) "v_max_f32_dpp v113, v113, v113 " ctrl "\n\ts_nop 1\n\t"
#define CF_ASM_DECIDER_STEP                                                                                                             \
  "v_readlane_b32 s60, v80, s44\n\t"                                                                                                    \
  "v_readlane_b32 s61, v81, s44\n\t"                                                                                                    \
  "v_readlane_b32 s62, v82, s44\n\t"                                                                                                    \
  "s_mov_b32 s41, 1\n\t"                                                                                                                \
  "s_and_b32 s63, s62, 0xffff00ff\n\t" /* a gpu kind, novel hosts, a group: not a plain step (EQUALS constraints are) */                                                                                                  \
  "s_cmp_lg_u32 s63, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  /* the head word; the candidates' entries */                                                                                         \
  "v_mov_b32_e32 v88, s47\n\t"                                                                                                          \
  "v_mov_b32_e32 v89, s51\n\t"                                                                                                          \
  "v_add_u32_e32 v100, s46, v83\n\t"                                                                                                    \
  "s_mov_b64 exec, 1\n\t"                                                                                                               \
  "ds_write_b32 v89, v88 offset:" CF_O_CTRL4 "\n\t"                                                                                              \
  "s_mov_b64 exec, -1\n\t"                                                                                                              \
  "s_mov_b32 s82, " CF_ASM_WAIT_READS "\n\t" /* reads of the entries before a missing answer makes this the C++ step's job */             \
  "1:\n\t"                                                                                                                              \
  "ds_read_b32 v92, v100\n\t"                                                                                                           \
  "ds_read_b64 v[94:95], v100 offset:8\n\t"                                                                                             \
  "ds_read_b128 v[96:99], v100 offset:16\n\t"                                                                                           \
  "ds_read_b32 v93, v100\n\t"                                                                                                           \
  "v_lshlrev_b32_e32 v114, 3, v65\n\t" /* the lane's offer's attribute bytes (read whatever the job: two instructions and an LDS slot) */ \
  "v_add_u32_e32 v114, s56, v114\n\t"                                                                                                   \
  "ds_read_b64 v[114:115], v114\n\t"                                                                                                    \
  /* the overlay's fitness while the loads fly: 1 - ((fc - c) * 0.5 / Tc + (fm - m) * 0.5 / Tm) */                                      \
  "v_subrev_u32_e32 v101, s60, v71\n\t"                                                                                                 \
  "v_subrev_u32_e32 v102, s61, v72\n\t"                                                                                                 \
  "v_cvt_f64_u32_e32 v[104:105], v101\n\t"                                                                                              \
  "v_cvt_f64_u32_e32 v[106:107], v102\n\t"                                                                                              \
  "v_mul_f64 v[104:105], v[104:105], v[66:67]\n\t"                                                                                      \
  "v_mul_f64 v[106:107], v[106:107], v[68:69]\n\t"                                                                                      \
  "v_add_f64 v[104:105], v[104:105], v[106:107]\n\t"                                                                                    \
  "v_add_f64 v[104:105], -v[104:105], 1.0\n\t"                                                                                          \
  "v_cmp_le_u32_e64 s[64:65], s60, v71\n\t"                                                                                             \
  "v_cmp_le_u32_e64 s[66:67], s61, v72\n\t"                                                                                             \
  "v_cmp_ne_u32_e64 s[68:69], 0, v64\n\t"                                                                                               \
  "s_and_b64 s[64:65], s[64:65], s[66:67]\n\t"                                                                                          \
  "s_and_b64 s[64:65], s[64:65], s[68:69]\n\t"                                                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                            \
  /* user-defined EQUALS constraints (constraints.clj:356-377) on the overlay's lanes: attribute byte `key` of the offer == value */      \
  "s_bfe_u32 s63, s62, 0x4000c\n\t"                                                                                                     \
  "s_cmp_eq_u32 s63, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 3f\n\t"                                                                                                               \
  "v_readlane_b32 s66, v85, s44\n\t"                                                                                                    \
  "v_readlane_b32 s67, v86, s44\n\t"                                                                                                    \
  "s_cmp_lt_u32 s63, 1\n\t"                                                                                                           \
  "s_cbranch_scc1 3f\n\t"                                                                                                               \
  "s_bfe_u32 s72, s66, 0x100000\n\t" /* key << 8 | value */                                                                     \
  "s_lshr_b32 s73, s72, 8\n\t"                                                                                                          \
  "s_lshl_b32 s73, s73, 3\n\t"                                                                                                          \
  "s_and_b32 s72, s72, 0xff\n\t"                                                                                                        \
  "v_lshrrev_b64 v[108:109], s73, v[114:115]\n\t"                                                                                       \
  "v_and_b32_e32 v108, 0xff, v108\n\t"                                                                                                  \
  "v_cmp_eq_u32_e64 s[74:75], s72, v108\n\t"                                                                                            \
  "s_and_b64 s[64:65], s[64:65], s[74:75]\n\t"                                                                                          \
  "s_cmp_lt_u32 s63, 2\n\t"                                                                                                           \
  "s_cbranch_scc1 3f\n\t"                                                                                                               \
  "s_bfe_u32 s72, s66, 0x100010\n\t" /* key << 8 | value */                                                                     \
  "s_lshr_b32 s73, s72, 8\n\t"                                                                                                          \
  "s_lshl_b32 s73, s73, 3\n\t"                                                                                                          \
  "s_and_b32 s72, s72, 0xff\n\t"                                                                                                        \
  "v_lshrrev_b64 v[108:109], s73, v[114:115]\n\t"                                                                                       \
  "v_and_b32_e32 v108, 0xff, v108\n\t"                                                                                                  \
  "v_cmp_eq_u32_e64 s[74:75], s72, v108\n\t"                                                                                            \
  "s_and_b64 s[64:65], s[64:65], s[74:75]\n\t"                                                                                          \
  "s_cmp_lt_u32 s63, 3\n\t"                                                                                                           \
  "s_cbranch_scc1 3f\n\t"                                                                                                               \
  "s_bfe_u32 s72, s67, 0x100000\n\t" /* key << 8 | value */                                                                     \
  "s_lshr_b32 s73, s72, 8\n\t"                                                                                                          \
  "s_lshl_b32 s73, s73, 3\n\t"                                                                                                          \
  "s_and_b32 s72, s72, 0xff\n\t"                                                                                                        \
  "v_lshrrev_b64 v[108:109], s73, v[114:115]\n\t"                                                                                       \
  "v_and_b32_e32 v108, 0xff, v108\n\t"                                                                                                  \
  "v_cmp_eq_u32_e64 s[74:75], s72, v108\n\t"                                                                                            \
  "s_and_b64 s[64:65], s[64:65], s[74:75]\n\t"                                                                                          \
  "s_cmp_lt_u32 s63, 4\n\t"                                                                                                           \
  "s_cbranch_scc1 3f\n\t"                                                                                                               \
  "s_bfe_u32 s72, s67, 0x100010\n\t" /* key << 8 | value */                                                                     \
  "s_lshr_b32 s73, s72, 8\n\t"                                                                                                          \
  "s_lshl_b32 s73, s73, 3\n\t"                                                                                                          \
  "s_and_b32 s72, s72, 0xff\n\t"                                                                                                        \
  "v_lshrrev_b64 v[108:109], s73, v[114:115]\n\t"                                                                                       \
  "v_and_b32_e32 v108, 0xff, v108\n\t"                                                                                                  \
  "v_cmp_eq_u32_e64 s[74:75], s72, v108\n\t"                                                                                            \
  "s_and_b64 s[64:65], s[64:65], s[74:75]\n\t"                                                                                          \
  "3:\n\t"                                                                                                                              \
  /* are the answers there, and acceptable?  tag = want | removals known; up to two unknown removals, neither of the last two positions */ \
  "v_and_b32_e32 v108, 0xffffff00, v92\n\t"                                                                                             \
  "v_cmp_eq_u32_e64 s[70:71], s45, v108\n\t"                                                                                            \
  "v_cmp_eq_u32_e64 s[72:73], v92, v93\n\t"                                                                                             \
  "v_sub_u32_e32 v108, v73, v92\n\t"                                                                                                    \
  "v_and_b32_e32 v108, 0xff, v108\n\t"                                                                                                  \
  "v_cmp_gt_u32_e64 s[74:75], 3, v108\n\t"                                                                                              \
  "v_and_b32_e32 v109, 0xffff, v94\n\t"                                                                                                 \
  "v_cmp_ne_u32_e64 s[76:77], v109, v74\n\t"                                                                                            \
  "v_cmp_ne_u32_e64 s[78:79], v109, v75\n\t"                                                                                            \
  "v_cmp_gt_i32_e64 s[80:81], 0, v95\n\t"                                                                                               \
  "s_and_b64 s[74:75], s[74:75], s[76:77]\n\t"                                                                                          \
  "s_and_b64 s[74:75], s[74:75], s[78:79]\n\t"                                                                                          \
  "s_or_b64 s[74:75], s[74:75], s[80:81]\n\t"                                                                                           \
  "s_and_b64 s[72:73], s[70:71], s[72:73]\n\t" /* the entry is this job's (and was read whole) */                                         \
  "s_and_b64 s[70:71], s[72:73], s[74:75]\n\t"                                                                                            \
  "s_andn2_b64 s[76:77], s[52:53], s[70:71]\n\t" /* answers missing */                                                                    \
  "s_cmp_eq_u64 s[76:77], 0\n\t"                                                                                                          \
  "s_cbranch_scc1 2f\n\t"                                                                                                                 \
  /* an answer for this job that names a member taken since: the offer is in the overlay now, fuller than the answer knew it; if its lane can take */ \
  /* the job it beats whatever the wave would answer today, and the step does not wait: the wave has no candidate (classfit_walk.hpp, the C++ step) */ \
  "s_and_b64 s[78:79], s[76:77], s[72:73]\n\t"                                                                                            \
  "s_cmp_eq_u64 s[78:79], 0\n\t"                                                                                                          \
  "s_cbranch_scc1 4f\n\t"                                                                                                                 \
  "v_and_b32_e32 v109, 0x3fff, v95\n\t"                                                                                                   \
  "s_and_b64 s[72:73], s[64:65], s[54:55]\n\t" /* overlay lanes that can take the job */                                                  \
  "10:\n\t"                                                                                                                               \
  "s_ff1_i32_b64 s83, s[78:79]\n\t"                                                                                                       \
  "v_readlane_b32 s63, v109, s83\n\t"                                                                                                     \
  "s_bitset0_b64 s[78:79], s83\n\t"                                                                                                       \
  "v_cmp_eq_u32_e64 vcc, s63, v65\n\t"                                                                                                    \
  "s_and_b64 vcc, vcc, s[72:73]\n\t"                                                                                                      \
  "s_cmp_eq_u64 vcc, 0\n\t"                                                                                                               \
  "s_cbranch_scc1 11f\n\t"                                                                                                                \
  "s_bitset1_b64 s[80:81], s83\n\t"                                                                                                       \
  "s_bitset0_b64 s[76:77], s83\n\t"                                                                                                       \
  "11:\n\t"                                                                                                                               \
  "s_cmp_lg_u64 s[78:79], 0\n\t"                                                                                                          \
  "s_cbranch_scc1 10b\n\t"                                                                                                                \
  "s_cmp_eq_u64 s[76:77], 0\n\t"                                                                                                          \
  "s_cbranch_scc1 2f\n\t"                                                                                                                 \
  "4:\n\t"                                                                                                                                \
  "s_sub_u32 s82, s82, 1\n\t" /* an answer is missing: look again a few times (a class wave is about to publish it), then give up */     \
  "s_cmp_eq_u32 s82, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  "s_sleep 1\n\t"                                                                                                                       \
  "s_branch 1b\n\t"                                                                                                                     \
  "2:\n\t"                                                                                                                              \
  "s_andn2_b64 s[74:75], s[52:53], s[80:81]\n\t" /* candidates */                                                                       \
  "s_or_b64 s[64:65], s[64:65], s[74:75]\n\t"    /* lanes that can take the job */                                                      \
  "v_cndmask_b32_e64 v110, v98, v104, s[54:55]\n\t"                                                                                     \
  "v_cndmask_b32_e64 v111, v99, v105, s[54:55]\n\t"                                                                                     \
  "v_cndmask_b32_e64 v110, 0, v110, s[64:65]\n\t"                                                                                       \
  "v_cndmask_b32_e64 v111, 0, v111, s[64:65]\n\t"                                                                                       \
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
  "v_readlane_b32 s63, v113, 63\n\t"                                                                                                    \
  "s_mov_b32 s41, 0\n\t"                                                                                                                \
  "s_cmp_eq_u32 s63, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 9f\n\t" /* nobody takes it */                                                                                         \
  "s_mov_b32 s41, 1\n\t"                                                                                                                \
  "v_cmp_eq_f32_e64 s[78:79], s63, v112\n\t"                                                                                            \
  "v_add_f32_e32 v108, 0x35000000, v112\n\t" /* + 2^-21 */                                                                              \
  "v_cmp_le_f32_e64 s[76:77], s63, v108\n\t"                                                                                            \
  "v_and_b32_e32 v109, 0x40000000, v95\n\t"                                                                                             \
  "v_cmp_ne_u32_e64 s[70:71], 0, v109\n\t"                                                                                              \
  "s_and_b64 s[76:77], s[76:77], s[64:65]\n\t" /* lanes whose fitness may round to the greatest */                                     \
  "s_bcnt1_i32_b64 s62, s[76:77]\n\t"                                                                                                   \
  "s_cmp_gt_u32 s62, 1\n\t"                                                                                                             \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  "s_and_b64 s[70:71], s[70:71], s[76:77]\n\t"                                                                                          \
  "s_and_b64 s[70:71], s[70:71], s[74:75]\n\t" /* the winner is a candidate whose class wave saw a possible tie */                      \
  "s_cmp_lg_u64 s[70:71], 0\n\t"                                                                                                        \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  "s_ff1_i32_b64 s82, s[78:79]\n\t"                                                                                                     \
  "s_cmp_lt_u32 s82, 58\n\t"                                                                                                            \
  "s_cbranch_scc0 5f\n\t"                                                                                                               \
  /* ---- an overlay lane wins */                                                                                                       \
  "v_readlane_b32 s64, v71, s82\n\t"                                                                                                    \
  "v_readlane_b32 s65, v72, s82\n\t"                                                                                                    \
  "v_readlane_b32 s66, v65, s82\n\t"                                                                                                    \
  "s_sub_u32 s67, s64, s60\n\t"                                                                                                         \
  "s_sub_u32 s70, s65, s61\n\t"                                                                                                         \
  "v_cmp_eq_u32_e64 vcc, s82, v84\n\t"                                                                                                  \
  "v_mov_b32_e32 v88, s67\n\t"                                                                                                          \
  "v_mov_b32_e32 v89, s70\n\t"                                                                                                          \
  "s_cmp_lt_u32 s67, s49\n\t"                                                                                                           \
  "s_cselect_b32 s71, 1, 0\n\t"                                                                                                         \
  "s_cmp_lt_u32 s70, s50\n\t"                                                                                                           \
  "s_cselect_b32 s62, 1, 0\n\t"                                                                                                         \
  "s_or_b32 s71, s71, s62\n\t"                                                                                                          \
  "v_cndmask_b32_e32 v71, v71, v88, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v72, v72, v89, vcc\n\t"                                                                                            \
  "s_cmp_lg_u32 s71, 0\n\t"                                                                                                             \
  "s_cselect_b64 s[62:63], vcc, 0\n\t" /* a lane that cannot take the smallest job any more is free again */                            \
  "s_mov_b32 s72, s44\n\t"                                                                                                              \
  "s_mov_b32 s73, 0\n\t"                                                                                                                \
  "v_cndmask_b32_e64 v64, v64, 0, s[62:63]\n\t"                                                                                         \
  "s_branch 7f\n\t"                                                                                                                     \
  /* ---- a class wave's candidate wins: the member leaves its arrays (zeroed, the wave's count moves on), an overlay lane opens */      \
  "5:\n\t"                                                                                                                              \
  "v_readlane_b32 s64, v96, s82\n\t"                                                                                                    \
  "v_readlane_b32 s65, v97, s82\n\t"                                                                                                    \
  "v_readlane_b32 s76, v95, s82\n\t"                                                                                                    \
  "v_readlane_b32 s77, v94, s82\n\t"                                                                                                    \
  "v_readlane_b32 s78, v73, s82\n\t"                                                                                                    \
  "s_sub_u32 s67, s64, s60\n\t"                                                                                                         \
  "s_sub_u32 s70, s65, s61\n\t"                                                                                                         \
  "s_and_b32 s66, s76, 0x3fff\n\t"                                                                                                      \
  "s_bfe_u32 s79, s76, 0x80010\n\t"                                                                                                     \
  "s_and_b32 s73, s77, 0xffff\n\t"                                                                                                      \
  "s_lshr_b32 s81, s77, 16\n\t"                                                                                                         \
  "s_sub_u32 s80, s82, 57\n\t"                                                                                                          \
  "s_cmp_lt_u32 s67, s49\n\t"                                                                                                           \
  "s_cselect_b32 s71, 1, 0\n\t"                                                                                                         \
  "s_cmp_lt_u32 s70, s50\n\t"                                                                                                           \
  "s_cselect_b32 s62, 1, 0\n\t"                                                                                                         \
  "s_or_b32 s71, s71, s62\n\t"                                                                                                          \
  "s_bcnt1_i32_b64 s62, s[68:69]\n\t"                                                                                                   \
  "s_cmp_lg_u32 s71, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 6f\n\t"                                                                                                               \
  "s_cmp_ge_u32 s62, " CF_ASM_EPOCH_LIVE "\n\t"                                                                                          \
  "s_cbranch_scc1 9f\n\t" /* the placement that fills the overlay: the epoch's end is the C++ step's */                                 \
  "6:\n\t"                                                                                                                              \
  "s_lshl_b32 s62, s73, 3\n\t"                                                                                                          \
  "s_add_u32 s62, s62, s51\n\t"                                                                                                         \
  "s_lshl_b32 s63, s80, 2\n\t"                                                                                                          \
  "s_add_u32 s63, s63, s51\n\t"                                                                                                         \
  "s_add_u32 s78, s78, 1\n\t"                                                                                                           \
  "v_mov_b32_e32 v88, 0\n\t"                                                                                                            \
  "v_mov_b32_e32 v89, 0\n\t"                                                                                                            \
  "v_mov_b32_e32 v90, s62\n\t"                                                                                                          \
  "v_mov_b32_e32 v91, s63\n\t"                                                                                                          \
  "v_mov_b32_e32 v100, s78\n\t"                                                                                                         \
  "s_mov_b64 exec, 1\n\t"                                                                                                               \
  "ds_write_b64 v90, v[88:89] offset:" CF_O_ARR "\n\t"                                                                                         \
  "ds_write_b32 v91, v100 offset:" CF_O_CTRL4 "\n\t"                                                                                             \
  "s_mov_b64 exec, -1\n\t"                                                                                                              \
  "v_cmp_eq_u32_e64 vcc, s82, v84\n\t"                                                                                                  \
  "v_mov_b32_e32 v101, s73\n\t"                                                                                                         \
  "s_nop 0\n\t"                                                                                                                         \
  "v_cndmask_b32_e32 v75, v75, v74, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v74, v74, v101, vcc\n\t"                                                                                           \
  "v_cndmask_b32_e32 v73, v73, v100, vcc\n\t"                                                                                           \
  "s_lshl_b32 s62, s80, 8\n\t"                                                                                                          \
  "s_lshl_b32 s63, s81, 12\n\t"                                                                                                         \
  "s_or_b32 s72, s44, s62\n\t"                                                                                                          \
  "s_or_b32 s72, s72, s63\n\t"                                                                                                          \
  "s_cmp_lg_u32 s71, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 7f\n\t" /* dead at once: nothing opens */                                                                             \
  "s_mul_i32 s62, s79, 56\n\t"                                                                                                          \
  "s_add_u32 s62, s62, s51\n\t"                                                                                                         \
  "v_mov_b32_e32 v90, s62\n\t"                                                                                                          \
  "ds_read_b128 v[116:119], v90 offset:" CF_O_CLS32 "\n\t"                                                                                        \
  "s_andn2_b64 s[62:63], s[54:55], s[68:69]\n\t"                                                                                        \
  "s_ff1_i32_b64 s83, s[62:63]\n\t"                                                                                                     \
  "v_cmp_eq_u32_e64 vcc, s83, v84\n\t"                                                                                                  \
  "v_mov_b32_e32 v88, s66\n\t"                                                                                                          \
  "v_mov_b32_e32 v89, s79\n\t"                                                                                                          \
  "v_mov_b32_e32 v91, s67\n\t"                                                                                                          \
  "v_mov_b32_e32 v101, s70\n\t"                                                                                                         \
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
  "v_mov_b32_e32 v88, s66\n\t"                                                                                                          \
  "v_cmp_lt_u32_e64 s[62:63], s67, v80\n\t"                                                                                             \
  "v_cmp_lt_u32_e64 s[76:77], s70, v81\n\t"                                                                                             \
  "v_cndmask_b32_e32 v76, v76, v88, vcc\n\t"                                                                                            \
  "s_or_b64 s[62:63], s[62:63], s[76:77]\n\t"                                                                                           \
  "s_lshl_b64 s[76:77], -2, s44\n\t"                                                                                                    \
  "s_and_b64 s[62:63], s[62:63], s[76:77]\n\t"                                                                                          \
  "s_or_b64 s[38:39], s[38:39], s[62:63]\n\t"                                                                                           \
  "s_min_u32 s37, s37, s67\n\t"                                                                                                         \
  "s_min_u32 s40, s40, s70\n\t"                                                                                                         \
  "s_add_u32 s36, s36, 1\n\t"                                                                                                           \
  "v_mov_b32_e32 v88, s72\n\t"                                                                                                          \
  "v_mov_b32_e32 v89, s73\n\t"                                                                                                          \
  "v_mov_b32_e32 v90, s64\n\t"                                                                                                          \
  "v_mov_b32_e32 v91, s65\n\t"                                                                                                          \
  "v_mov_b32_e32 v100, s48\n\t"                                                                                                         \
  "s_mov_b64 exec, 1\n\t"                                                                                                               \
  "ds_write_b128 v100, v[88:91]\n\t"                                                                                                    \
  "s_mov_b64 exec, -1\n\t"                                                                                                              \
  "s_mov_b32 s41, 2\n\t"                                                                                                                \
  "9:\n\t"

// ---- a class wave's PLAIN answer ------------------------------------------------------------------------------------------------------------------------------
// One pass of a class wave that holds ONE class: the three words the decider writes, the jobs of the set's other waves skipped, the run-ahead limit, the job,
// and — for a job of the class's kind without constraints — the chunk its level summary promises, its first member with room, that member's entry on the board.
// Anything else leaves BEFORE a change with status 1 (the C++ path answers the job), 2 (nothing to do now: idle) or 3 (a collective turn, or members of the
// wave have left: the C++ loop's business).
//   v[64:71]  the lane's (= chunk's) eight level summaries      v72 / v73 / v74 the batch's jobs: cpus, mem, meta      v75 the lane number
//   s[36:43]  s[36:37] jobs not answered yet, s38 walked ordinal of the first of them, s39 that ordinal modulo the set's waves, s40 removals known,
//             s41 OUT status, s42 OUT the head word, s43 OUT the removal count read
//   s[44:51]  s44 LDS address of the fixed records, s45 the set (board column), s46 first job of the batch, s47 generation << 8, s48 / s49 the class's
//             offset / size in the arrays, s50 LDS address of the offer-id array, s51 class << 16
//   s[52:59]  s[52:53] 0.5 / Tc, s[54:55] 0.5 / Tm, s56 this wave's turn among the set's waves, s57 their number, s58 the class's kind, s59 run-ahead limit
//   clobbered: v[80:99], s[60:83], vcc, scc
#define CF_ASM_CLASS_STEP                                                                                                               \
  "v_mov_b32_e32 v80, s44\n\t"                                                                                                          \
  "s_lshl_b32 s60, s45, 2\n\t"                                                                                                          \
  "s_add_u32 s60, s60, s44\n\t"                                                                                                         \
  "v_mov_b32_e32 v81, s60\n\t"                                                                                                          \
  "ds_read_b64 v[82:83], v80 offset:" CF_O_CTRL "\n\t" /* mode, head word */                                                                    \
  "ds_read_b32 v84, v81 offset:" CF_O_CTRL4 "\n\t"      /* removals from this set */                                                             \
  "s_mov_b32 s41, 3\n\t"                                                                                                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                            \
  "v_readfirstlane_b32 s61, v82\n\t"                                                                                                    \
  "v_readfirstlane_b32 s42, v83\n\t"                                                                                                    \
  "v_readfirstlane_b32 s43, v84\n\t"                                                                                                    \
  "s_cmp_lg_u32 s61, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  "s_cmp_lg_u32 s43, s40\n\t"                                                                                                           \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  "s_mov_b32 s41, 2\n\t"                                                                                                                \
  /* the jobs of the set's other waves */                                                                                               \
  "1:\n\t"                                                                                                                              \
  "s_cmp_eq_u64 s[36:37], 0\n\t"                                                                                                        \
  "s_cbranch_scc1 9f\n\t"                                                                                                               \
  "s_cmp_eq_u32 s39, s56\n\t"                                                                                                           \
  "s_cbranch_scc1 2f\n\t"                                                                                                               \
  "s_add_u32 s62, s36, -1\n\t"                                                                                                          \
  "s_addc_u32 s63, s37, -1\n\t"                                                                                                         \
  "s_and_b64 s[36:37], s[36:37], s[62:63]\n\t"                                                                                          \
  "s_add_u32 s38, s38, 1\n\t"                                                                                                           \
  "s_add_u32 s39, s39, 1\n\t"                                                                                                           \
  "s_cmp_eq_u32 s39, s57\n\t"                                                                                                           \
  "s_cselect_b32 s39, 0, s39\n\t"                                                                                                       \
  "s_branch 1b\n\t"                                                                                                                     \
  "2:\n\t"                                                                                                                                \
  "s_lshr_b32 s60, s42, 8\n\t"                                                                                                            \
  "s_cmp_lt_u32 s38, s60\n\t" /* the decider has gone past this wave: the C++ loop moves it on */                                         \
  "s_cbranch_scc0 12f\n\t"                                                                                                                \
  "s_mov_b32 s41, 3\n\t"                                                                                                                  \
  "s_branch 9f\n\t"                                                                                                                       \
  "12:\n\t"                                                                                                                               \
  "s_add_u32 s60, s60, s59\n\t"                                                                                                         \
  "s_cmp_ge_u32 s38, s60\n\t"                                                                                                           \
  "s_cbranch_scc1 9f\n\t" /* too far ahead of the decider */                                                                            \
  "s_ff1_i32_b64 s64, s[36:37]\n\t"                                                                                                     \
  "v_readlane_b32 s65, v72, s64\n\t"                                                                                                    \
  "v_readlane_b32 s66, v73, s64\n\t"                                                                                                    \
  "v_readlane_b32 s67, v74, s64\n\t"                                                                                                    \
  "s_mov_b32 s41, 1\n\t"                                                                                                                \
  "s_lshr_b32 s60, s67, 12\n\t"                                                                                                         \
  "s_cmp_lg_u32 s60, 0\n\t"                                                                                                             \
  "s_cbranch_scc1 9f\n\t" /* constraints: the C++ path */                                                                               \
  "s_and_b32 s60, s67, 0xff\n\t"                                                                                                        \
  "s_cmp_lg_u32 s60, s58\n\t"                                                                                                           \
  "s_cbranch_scc1 8f\n\t" /* not our kind: nobody asks */                                                                               \
  /* the level summary of the job's cpus level: a tree of selects on the level's three bits */                                         \
  "s_bfe_u32 s60, s67, 0x30008\n\t"                                                                                                     \
  "s_bitcmp1_b32 s60, 0\n\t"                                                                                                            \
  "s_cselect_b64 vcc, -1, 0\n\t"                                                                                                        \
  "v_cndmask_b32_e32 v80, v64, v65, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v81, v66, v67, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v82, v68, v69, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v83, v70, v71, vcc\n\t"                                                                                            \
  "s_bitcmp1_b32 s60, 1\n\t"                                                                                                            \
  "s_cselect_b64 vcc, -1, 0\n\t"                                                                                                        \
  "v_cndmask_b32_e32 v80, v80, v81, vcc\n\t"                                                                                            \
  "v_cndmask_b32_e32 v82, v82, v83, vcc\n\t"                                                                                            \
  "s_bitcmp1_b32 s60, 2\n\t"                                                                                                            \
  "s_cselect_b64 vcc, -1, 0\n\t"                                                                                                        \
  "v_cndmask_b32_e32 v80, v80, v82, vcc\n\t"                                                                                            \
  /* the entry's address: column s45 of row (ordinal mod 12); the tag */                                                               \
  "s_mul_i32 s68, s38, 43\n\t"                                                                                                          \
  "s_lshr_b32 s68, s68, " CF_SLOT_SHIFT "\n\t"                                                                                                          \
  "s_mul_i32 s68, s68, " CF_SLOT_N "\n\t"                                                                                                          \
  "s_sub_u32 s68, s38, s68\n\t"                                                                                                         \
  "s_lshl_b32 s68, s68, 8\n\t"                                                                                                          \
  "s_lshl_b32 s69, s45, 5\n\t"                                                                                                          \
  "s_add_u32 s68, s68, s69\n\t"                                                                                                         \
  "s_add_u32 s68, s68, s44\n\t"                                                                                                         \
  "s_add_u32 s69, s46, s64\n\t"                                                                                                         \
  "s_lshl_b32 s69, s69, 12\n\t"                                                                                                         \
  "s_or_b32 s69, s69, s47\n\t"                                                                                                          \
  "s_and_b32 s70, s40, 0xff\n\t"                                                                                                        \
  "s_or_b32 s69, s69, s70\n\t"                                                                                                          \
  "v_mov_b32_e32 v94, s68\n\t"                                                                                                          \
  "v_mov_b32_e32 v95, s69\n\t"                                                                                                          \
  "v_mov_b32_e32 v96, -1\n\t"                                                                                                           \
  "v_cmp_lt_u32_e64 s[70:71], s66, v80\n\t" /* chunks that promise a member with room */                                                \
  "s_cmp_eq_u64 s[70:71], 0\n\t"                                                                                                        \
  "s_cbranch_scc1 6f\n\t"                                                                                                               \
  "s_ff1_i32_b64 s72, s[70:71]\n\t"                                                                                                     \
  "s_lshl_b32 s73, s72, 6\n\t"                                                                                                          \
  "s_add_u32 s74, s73, s48\n\t" /* first position of the chunk */                                                                       \
  "s_sub_u32 s75, s49, s73\n\t" /* members from there on */                                                                             \
  "v_add_u32_e32 v97, s74, v75\n\t"                                                                                                     \
  "v_lshlrev_b32_e32 v98, 3, v97\n\t"                                                                                                   \
  "v_add_u32_e32 v98, s44, v98\n\t"                                                                                                     \
  "v_lshlrev_b32_e32 v99, 1, v97\n\t"                                                                                                   \
  "v_add_u32_e32 v99, s50, v99\n\t"                                                                                                     \
  "ds_read_b64 v[84:85], v98 offset:" CF_O_ARR "\n\t"                                                                                          \
  "ds_read_u16 v90, v99\n\t"                                                                                                            \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                                            \
  "v_cmp_le_u32_e64 s[76:77], s65, v84\n\t"                                                                                             \
  "v_cmp_le_u32_e64 s[78:79], s66, v85\n\t"                                                                                             \
  "v_cmp_gt_u32_e64 s[80:81], s75, v75\n\t"                                                                                             \
  "v_and_b32_e32 v91, 0x8000, v90\n\t"                                                                                                  \
  "v_cmp_eq_u32_e64 s[82:83], 0, v91\n\t"                                                                                               \
  "s_and_b64 s[76:77], s[76:77], s[78:79]\n\t"                                                                                          \
  "s_and_b64 s[76:77], s[76:77], s[80:81]\n\t"                                                                                          \
  "s_and_b64 s[76:77], s[76:77], s[82:83]\n\t"                                                                                          \
  "s_cmp_eq_u64 s[76:77], 0\n\t"                                                                                                        \
  "s_cbranch_scc1 9f\n\t" /* the summary promised too much (members have left): the C++ path looks on and recomputes it */              \
  /* every lane's fitness and entry words; the first lane with room stores its own */                                                  \
  "v_subrev_u32_e32 v91, s65, v84\n\t"                                                                                                  \
  "v_subrev_u32_e32 v92, s66, v85\n\t"                                                                                                  \
  "v_cvt_f64_u32_e32 v[86:87], v91\n\t"                                                                                                 \
  "v_cvt_f64_u32_e32 v[88:89], v92\n\t"                                                                                                 \
  "v_mul_f64 v[86:87], v[86:87], s[52:53]\n\t"                                                                                          \
  "v_mul_f64 v[88:89], v[88:89], s[54:55]\n\t"                                                                                          \
  "v_add_f64 v[86:87], v[86:87], v[88:89]\n\t"                                                                                          \
  "v_add_f64 v[86:87], -v[86:87], 1.0\n\t"                                                                                              \
  "v_mov_b32_e32 v91, s72\n\t"                                                                                                          \
  "v_lshl_or_b32 v92, v91, 16, v97\n\t" /* position | chunk lane << 16 */                                                               \
  "v_and_b32_e32 v93, 0x3fff, v90\n\t"                                                                                                  \
  "v_or_b32_e32 v93, s51, v93\n\t"                                                                                                      \
  "v_lshlrev_b32_e32 v91, 16, v90\n\t"                                                                                                  \
  "v_and_b32_e32 v91, 0x40000000, v91\n\t" /* the member's "next one may round to the same fitness" flag */                             \
  "v_or_b32_e32 v93, v93, v91\n\t"                                                                                                      \
  "s_ff1_i32_b64 s78, s[76:77]\n\t"                                                                                                     \
  "s_lshl_b64 s[78:79], 1, s78\n\t"                                                                                                     \
  "s_mov_b64 exec, s[78:79]\n\t"                                                                                                        \
  "ds_write_b32 v94, v96\n\t"                                                                                                           \
  "ds_write_b128 v94, v[84:87] offset:16\n\t"                                                                                           \
  "ds_write_b64 v94, v[92:93] offset:8\n\t"                                                                                             \
  "ds_write_b32 v94, v95\n\t"                                                                                                           \
  "s_mov_b64 exec, -1\n\t"                                                                                                              \
  "s_branch 8f\n\t"                                                                                                                     \
  /* no chunk promises room: "none" */                                                                                                  \
  "6:\n\t"                                                                                                                              \
  "v_mov_b32_e32 v92, 0\n\t"                                                                                                            \
  "v_mov_b32_e32 v93, 0x80000000\n\t"                                                                                                   \
  "s_mov_b64 exec, 1\n\t"                                                                                                               \
  "ds_write_b32 v94, v96\n\t"                                                                                                           \
  "ds_write_b64 v94, v[92:93] offset:8\n\t"                                                                                             \
  "ds_write_b32 v94, v95\n\t"                                                                                                           \
  "s_mov_b64 exec, -1\n\t"                                                                                                              \
  /* the job is done with */                                                                                                            \
  "8:\n\t"                                                                                                                              \
  "s_add_u32 s62, s36, -1\n\t"                                                                                                          \
  "s_addc_u32 s63, s37, -1\n\t"                                                                                                         \
  "s_and_b64 s[36:37], s[36:37], s[62:63]\n\t"                                                                                          \
  "s_add_u32 s38, s38, 1\n\t"                                                                                                           \
  "s_add_u32 s39, s39, 1\n\t"                                                                                                           \
  "s_cmp_eq_u32 s39, s57\n\t"                                                                                                           \
  "s_cselect_b32 s39, 0, s39\n\t"                                                                                                       \
  "s_mov_b32 s41, 0\n\t"                                                                                                                \
  "9:\n\t"
