/* wf_experiments.h -- the laboratory of the wavefront kernels (wavefront.hip includes this ONLY under -DNORI_LAB, which
 * tools/build_variant.sh / build_variant_fast.sh pass; the product library never sees it: its hooks are empty).
 *
 * What is here answered "what bounds this kernel" by perturbing one resource at a time (DESIGN_HISTORY.md; tools/ab.sh alternates
 * the variant libraries on one box):
 *   -DNORI_EXP_SENS=n    the hand-written BVH2 node step of wf_extend: 1 = one more global_load_dwordx4, 2 = the LDS reads over
 *                        again, 3 = sixteen v_mov, 4 = sixteen s_mov, 5 = 64 idle cycles
 *   -DNORI_WF_PROFILE=1  a wave's cycles by phase (refill / node loop / triangle step), summed into the stats slots and printed
 *   -DNORI_EXP_SHADE=n   wf_shade: 1 = one more dense 16-B load per path, 2 = one more 16-B store per survivor, 3 = 64 more VALU
 *                        instructions per path, 4 = one more dependent 16-B gather from the shading records, 5 / 6 = LDS padding that
 *                        leaves 3 / 2 workgroups per CU instead of 4
 *   -DNORI_EXP_WIDE_SENS=n  the WIDE node step (rt_trace.h): 1 = one more load, 3 = 32 more VALU instructions, 5 = 64 idle cycles
 * Every hook is a macro used at exactly one place of wavefront.hip; the names of the variables they touch are the kernels'. */
#pragma once

#define NORI_X16(s) s s s s s s s s s s s s s s s s

/* ---- the node step of bvh2q_node_loop_asm */
#if NORI_EXP_SENS == 1
#define NORI_EXP_Q_GLOBAL "\n\tglobal_load_dwordx4 v[36:39], v40, %[nodes] offset:16\n"
#else
#define NORI_EXP_Q_GLOBAL
#endif
#if NORI_EXP_SENS == 2
#define NORI_EXP_Q_LDS "ds_read_b128 v[32:35], v40\n\tds_read_b128 v[36:39], v40 offset:16\n\t"
#else
#define NORI_EXP_Q_LDS
#endif
#if NORI_EXP_SENS == 3
#define NORI_EXP_Q_ALU NORI_X16("v_mov_b32 v41, v40\n\t")
#elif NORI_EXP_SENS == 4
#define NORI_EXP_Q_ALU NORI_X16("s_mov_b64 %[u], %[t]\n\t")
#elif NORI_EXP_SENS == 5
#define NORI_EXP_Q_ALU "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#else
#define NORI_EXP_Q_ALU
#endif

/* ---- where the waves of wf_extend spend their cycles: refill (results out, new rays in), node loop, triangle step -- summed over
   waves into stats slots 6, S_NODES, S_TRIS and 7 (total) */
#if NORI_WF_PROFILE
#define NORI_PROF_DECL unsigned long long prof_t = __builtin_amdgcn_s_memrealtime(), prof_refill = 0ull, prof_node = 0ull, prof_leaf = 0ull; \
    const unsigned long long prof_t0 = prof_t;
#define NORI_PROF_MARK(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); acc += now_ - prof_t; prof_t = now_; }
#define NORI_PROF_STORE atomicAdd(&b.stats[6], prof_refill); atomicAdd(&b.stats[S_NODES], prof_node); atomicAdd(&b.stats[S_TRIS], prof_leaf); \
    atomicAdd(&b.stats[7], __builtin_amdgcn_s_memrealtime() - prof_t0);
#define NORI_PROF_REPORT(h, k) { const double tot = (double) h[k * S_COUNT + 7]; \
    if (tot > 0.0) fprintf(stderr, "[wf_extend profile] wave cycles: refill %.3f, node loop %.3f, triangle step %.3f of %.4g total\n", h[k * S_COUNT + 6] / tot, \
                           h[k * S_COUNT + S_NODES] / tot, h[k * S_COUNT + S_TRIS] / tot, tot); }
#else
#define NORI_PROF_DECL
#define NORI_PROF_MARK(acc)
#define NORI_PROF_STORE
#define NORI_PROF_REPORT(h, k)
#endif

/* ---- the WIDE node step (rt_trace.h, trav_wide_step): -DNORI_EXP_WIDE_SENS=1 one more load of the node's record, 3 = 32 more VALU instructions */
#if defined(NORI_EXP_WIDE_SENS) && NORI_EXP_WIDE_SENS == 3
#define NORI_LAB_WIDE_STEP { float x_ = q0.x; asm volatile(NORI_X16("v_mov_b32 %0, %0\n\t") NORI_X16("v_mov_b32 %0, %0\n\t") : "+v"(x_)); q0.x = x_; }
#elif defined(NORI_EXP_WIDE_SENS) && NORI_EXP_WIDE_SENS == 5
#define NORI_LAB_WIDE_STEP asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#elif defined(NORI_EXP_WIDE_SENS)
#define NORI_LAB_WIDE_STEP if (!(tv.node & kTopBit)) { const f4 x_ = sc.nodes[(size_t) tv.node * kNodeQuads + 3]; asm volatile("" :: "v"(x_.x), "v"(x_.y), "v"(x_.z), "v"(x_.w)); }
#endif

/* ---- wf_shade */
#if NORI_EXP_SHADE == 5 || NORI_EXP_SHADE == 6
#define NORI_LAB_SHADE_PAD __shared__ volatile char s_pad[NORI_EXP_SHADE == 5 ? 36 * 1024 : 50 * 1024]; s_pad[threadIdx.x * 64] = 0;
#else
#define NORI_LAB_SHADE_PAD
#endif
#if NORI_EXP_SHADE == 1
#define NORI_LAB_SHADE_VERTEX if (!FRESH) { const f4 x = ld_f4<2>(&S.dB[i]); asm volatile("" :: "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w)); }
#elif NORI_EXP_SHADE == 3
#define NORI_LAB_SHADE_VERTEX { float x = h.x; asm volatile(NORI_X16("v_mov_b32 %0, %0\n\t") NORI_X16("v_mov_b32 %0, %0\n\t") NORI_X16("v_mov_b32 %0, %0\n\t") NORI_X16("v_mov_b32 %0, %0\n\t") : "+v"(x)); }
#elif NORI_EXP_SHADE == 4
#define NORI_LAB_SHADE_VERTEX if ((hw & kMissA) != kMissA) { const f4 x = sc.shade_tris[(size_t) (hw & kMissA) * kShadeQuads + 5]; asm volatile("" :: "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w)); }
#else
#define NORI_LAB_SHADE_VERTEX
#endif
#if NORI_EXP_SHADE == 2
#define NORI_LAB_SHADE_STORE st_f4<2>(&S.dB[j], n_o);      /* (nobody reads S.dB in this kernel) */
#else
#define NORI_LAB_SHADE_STORE
#endif
