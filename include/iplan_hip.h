/* iplan_hip.h -- C ABI of libiplan_hip.so, the MI355X (gfx950) implementation of the iPLAN
 * multi-agent forward/backward hot path.
 *
 * The reference (wuxiyang1996/iPLAN) is pure Python on PyTorch: it has no FFI of its own, so the
 * "binding a maintainer would add" is a ctypes stub under the reference's nn.Module / policy
 * classes.  Each entry point below names the reference code it replaces
 * (file:line relative to the reference root).
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer to contiguous fp32 unless
 *     stated; int64 strides are in ELEMENTS.
 *   - the caller owns all memory (PyTorch caching allocator in the shipped host code); the
 *     library allocates nothing, keeps no global mutable state, and enqueues every kernel on the
 *     stream handed in (no implicit synchronisation) -> thread-safe and stream-ordered.
 *   - return value: 0 on success, a negative IPLAN_E* code otherwise; iplan_last_error() returns
 *     a thread-local human-readable description of the last failure on the calling thread.
 *   - "nets": the reference keeps one private network set per learning agent and loops over
 *     agents in Python (controllers/dcntrl_controller.py:34, nova/prediction_policy.py:101,
 *     nova/stable_behavior_policy.py:101).  Here the n_nets parameter sets are stacked in one
 *     parameter arena (net stride `params_s_net`, per-tensor element offsets `off[]` in
 *     state_dict order) and ONE launch covers every (net, env) pair.
 */
#ifndef IPLAN_HIP_H
#define IPLAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* iplan_stream_t; /* hipStream_t */

enum {
    IPLAN_OK = 0,
    IPLAN_EINVAL = -1,      /* unsupported dimension / null pointer            */
    IPLAN_EALIGN = -2,      /* pointer not 16-byte aligned                     */
    IPLAN_EHIP = -3         /* HIP runtime error at launch                     */
};

const char* iplan_last_error(void);
int iplan_version(void);

/* Limits of this build (compile-time tile sizes). */
#define IPLAN_MAX_ENTITIES 64      /* N  : entities per (env, agent) scene             */
#define IPLAN_GAT_HIDDEN 32        /* H == A == 32 (config/default.yaml:89-90)         */

/* ------------------------------------------------------------------------------------------
 * GAT_Net.forward  (nova/GAT_Net.py:41-142)  --  K1..K7 of SURVEY.md §2b, all nets, one launch.
 * Parameter tensors in state_dict order:
 */
enum {
    IPLAN_GAT_ENC_W = 0,    /* encoding.weight                [H, D]   */
    IPLAN_GAT_ENC_B,        /* encoding.bias                  [H]      */
    IPLAN_GAT_F_WIH,        /* hard_bi_GRU.weight_ih_l0       [3H, 2H] */
    IPLAN_GAT_F_WHH,        /* hard_bi_GRU.weight_hh_l0       [3H, H]  */
    IPLAN_GAT_F_BIH,        /* hard_bi_GRU.bias_ih_l0         [3H]     */
    IPLAN_GAT_F_BHH,        /* hard_bi_GRU.bias_hh_l0         [3H]     */
    IPLAN_GAT_R_WIH,        /* ..._reverse                             */
    IPLAN_GAT_R_WHH,
    IPLAN_GAT_R_BIH,
    IPLAN_GAT_R_BHH,
    IPLAN_GAT_HARD_W,       /* hard_encoding.weight           [2, 2H]  */
    IPLAN_GAT_HARD_B,       /* hard_encoding.bias             [2]      */
    IPLAN_GAT_Q_W,          /* q.weight                       [A, H]   */
    IPLAN_GAT_K_W,          /* k.weight                       [A, H]   */
    IPLAN_GAT_V_W,          /* v.weight                       [A, H]   */
    IPLAN_GAT_V_B,          /* v.bias                         [A]      */
    IPLAN_GAT_C_WIH,        /* rnn.weight_ih (GRUCell)        [3A, A]  */
    IPLAN_GAT_C_WHH,        /* rnn.weight_hh                  [3A, A]  */
    IPLAN_GAT_C_BIH,        /* rnn.bias_ih                    [3A]     */
    IPLAN_GAT_C_BHH,        /* rnn.bias_hh                    [3A]     */
    IPLAN_GAT_NPARAM
};

/* Activations kept for the backward pass (all [n_nets, B, ...] contiguous; NULL = inference). */
typedef struct {
    float* h_enc;   /* [n_nets,B,N,H]            ReLU(encoding(obs))                         */
    float* gru;     /* [n_nets,B,2,N,N-1,5,H]    per pair-step and direction: h, r, z, n, hn */
    float* qkv;     /* [n_nets,B,3,N,A]          q, k, v                                     */
    float* soft;    /* [n_nets,B,N,N-1]          soft attention weights                      */
    float* hard;    /* [n_nets,B,N,N-1]          gumbel-softmax class-1 weights              */
    float* x;       /* [n_nets,B,N,A]            aggregated neighbour feature                */
    float* cell;    /* [n_nets,B,N,4,A]          GRUCell r, z, n, hn                         */
} IplanGatSaved;

typedef struct {
    int32_t n_nets, B, N;     /* N <= IPLAN_MAX_ENTITIES                                   */
    int32_t d0, d1;           /* GAT input = [src0 (d0 floats) || src1 (d1 floats)] per entity */
    const float* src0;        /* element (net,b,i,c) at src0 + net*src0_s_net + b*src0_s_b + i*d0 + c */
    int64_t src0_s_net, src0_s_b;
    const float* src1;        /* may be NULL iff d1 == 0 (GAT_use_behavior False)           */
    int64_t src1_s_net, src1_s_b;
    const float* h_prev;      /* previous attention latent, rows of A floats: (net,b,i)     */
    int64_t h_s_net, h_s_b;
    float* out;               /* new attention latent, same addressing                      */
    int64_t out_s_net, out_s_b;
    const float* noise;       /* gumbel samples [n_nets,B,N,N-1,2] contiguous               */
    const float* params;      /* parameter arena                                            */
    int64_t params_s_net;
    int64_t off[IPLAN_GAT_NPARAM];
    float tau;                /* 0.01 (nova/GAT_Net.py:93)                                  */
    IplanGatSaved saved;      /* all-NULL for inference                                     */
} IplanGatFwdArgs;

int iplan_gat_fwd(const IplanGatFwdArgs* args, iplan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IPLAN_HIP_H */
