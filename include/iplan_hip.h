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
 *     library allocates nothing (one documented exception: iplan_p2p_alloc), keeps no global mutable state, and
 *     enqueues every kernel on the stream handed in (no implicit synchronisation) -> thread-safe and stream-ordered.
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
/* sizeof() of an argument struct by name ("IplanGatFwdArgs", ...), 0 if unknown: lets a foreign-language binding check
 * its own struct mirror against the library it loaded. */
size_t iplan_sizeof(const char* struct_name);

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

/* Activations kept for the backward pass (contiguous; NULL = inference).  Rows are (b, i) = b*N + i. */
typedef struct {
    float* h_enc;   /* [n_nets, B*N, H]              ReLU(encoding(obs))                          */
    float* gru;     /* [n_nets, 2, B, ceil(N/16), N-1, 8, 16, 16]  per direction, scene, 16-ego tile and pair step: eight 16-column
                       groups (h h r r z z n n) of [16 egos][16 columns] -- 1 KiB blocks, one per wave store / load; the
                       backward recomputes gh_n = W_hn h_prev + b_hn                                                    */
    float* qkv;     /* [n_nets, B*N, 3A]             q | k | v                                    */
    float* soft;    /* [n_nets, B*N, N-1]            soft attention weights                       */
    float* hard;    /* [n_nets, B*N, N-1]            gumbel-softmax class-1 weights               */
    float* x;       /* [n_nets, B*N, A]              aggregated neighbour feature                 */
    float* cell;    /* [n_nets, B*N, 4A]             GRUCell r, z, n, hn                          */
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
    int64_t* phase_clocks;    /* optional profiling aid: [n_nets*B, 5] wall_clock64 at kernel entry and after each
                                 of the 4 phases (written by thread 0 of each workgroup); NULL = off          */
} IplanGatFwdArgs;

int iplan_gat_fwd(const IplanGatFwdArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * EncoderRNN.forward + soft latent update (nova/behavior_net.py:17-22,
 * nova/stable_behavior_policy.py:83-123): Linear(d->R)+ReLU -> GRU(R) over the L-step window from
 * the carried hidden state -> Linear(R->Z) -> softmax -> new = (1-c)*prev + c*latent.
 * Rows are (net, b, i) with i < N; R == 32, d <= 16, Z <= 16 in this build.
 */
enum {
    IPLAN_ENC_LIN_W = 0,    /* linear.weight      [R, d]  */
    IPLAN_ENC_LIN_B,        /* linear.bias        [R]     */
    IPLAN_ENC_WIH,          /* rnn.weight_ih_l0   [3R, R] */
    IPLAN_ENC_WHH,          /* rnn.weight_hh_l0   [3R, R] */
    IPLAN_ENC_BIH,          /* rnn.bias_ih_l0     [3R]    */
    IPLAN_ENC_BHH,          /* rnn.bias_hh_l0     [3R]    */
    IPLAN_ENC_OUT_W,        /* out.weight         [Z, R]  */
    IPLAN_ENC_OUT_B,        /* out.bias           [Z]     */
    IPLAN_ENC_NPARAM
};

typedef struct {
    int32_t n_nets, B, N, L, d, Z;
    const float* x;            /* window element (net,b,i,t,c) at x + net*x_s_net + b*x_s_b + i*x_s_i + t*x_s_t + c */
    int64_t x_s_net, x_s_b;
    const float* h0;           /* carried hidden, rows of 32 floats */
    int64_t h0_s_net, h0_s_b;
    float* hL;                 /* new hidden */
    int64_t hL_s_net, hL_s_b;
    const float* prev_latent;  /* rows of Z floats; NULL -> latent_out = softmax only */
    int64_t pl_s_net, pl_s_b;
    float* latent_out;         /* rows of Z floats */
    int64_t lo_s_net, lo_s_b;
    float one_minus_c, c;      /* soft_update_coef (config/default.yaml:75) */
    const float* params;
    int64_t params_s_net;
    int64_t off[IPLAN_ENC_NPARAM];
    int64_t x_s_i, x_s_t;      /* 0, 0 = the contiguous default (L*d, d); lets a sliding window over a time-major
                                  observation log be read in place                                          */
} IplanEncFwdArgs;

int iplan_enc_fwd(const IplanEncFwdArgs* args, iplan_stream_t stream);
/* iplan_gat_fwd and iplan_enc_fwd of one rollout vector step (ippo_parallel_runner.py:223-230: both read the previous
 * latents) as ONE launch: the GAT scenes' workgroups first, the encoder's behind them on the remaining CUs. */
int iplan_gat_enc_fwd(const IplanGatFwdArgs* gat, const IplanEncFwdArgs* enc, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * R_Actor / R_Critic forward (modules/agents/ippo_actor.py:43-102, modules/critics/ippo_critic.py:47-65,
 * utils/mappo_utils/{mlp,rnn,act,distributions,popart}.py) fused with the feature assembly of
 * DcntrlMAC._build_inputs / _build_inputs_ippo (controllers/dcntrl_controller.py:187-213, 87-115):
 *   x = [ per entity i<N: src0[i] || src1[i] || src2[i] ] || onehot(last_action) || onehot(agent)
 *   LN(F) -> Linear(F->M)+ReLU -> LN -> Linear(M->M)+ReLU -> LN -> GRU(M) one step -> LN -> head
 * for every agent (grid.y) and for actor and critic (grid.z) in one launch.  M == 64.
 * The [rows, F] input matrix the reference materialises (231 MB per agent in PPO) never exists.
 */
enum {
    IPLAN_AC_FN_W = 0,      /* base.feature_norm.weight   [F]     */
    IPLAN_AC_FN_B,          /* base.feature_norm.bias     [F]     */
    IPLAN_AC_FC1_W,         /* base.mlp.fc1.0.weight      [M, F]  */
    IPLAN_AC_FC1_B,         /* base.mlp.fc1.0.bias        [M]     */
    IPLAN_AC_LN1_W,         /* base.mlp.fc1.2.weight      [M]     */
    IPLAN_AC_LN1_B,         /* base.mlp.fc1.2.bias        [M]     */
    IPLAN_AC_FC2_W,         /* base.mlp.fc2.0.0.weight    [M, M]  */
    IPLAN_AC_FC2_B,         /* base.mlp.fc2.0.0.bias      [M]     */
    IPLAN_AC_LN2_W,         /* base.mlp.fc2.0.2.weight    [M]     */
    IPLAN_AC_LN2_B,         /* base.mlp.fc2.0.2.bias      [M]     */
    IPLAN_AC_WIH,           /* rnn.rnn.weight_ih_l0       [3M, M] */
    IPLAN_AC_WHH,           /* rnn.rnn.weight_hh_l0       [3M, M] */
    IPLAN_AC_BIH,           /* rnn.rnn.bias_ih_l0         [3M]    */
    IPLAN_AC_BHH,           /* rnn.rnn.bias_hh_l0         [3M]    */
    IPLAN_AC_LN3_W,         /* rnn.norm.weight            [M]     */
    IPLAN_AC_LN3_B,         /* rnn.norm.bias              [M]     */
    IPLAN_AC_HEAD_W,        /* act.action_out.linear.weight [n_act, M]  |  v_out.weight [1, M] */
    IPLAN_AC_HEAD_B,        /* act.action_out.linear.bias   [n_act]     |  v_out.bias   [1]    */
    IPLAN_AC_NPARAM
};
#define IPLAN_AC_HIDDEN 64
#define IPLAN_AC_SAVE_FLOATS (10 * IPLAN_AC_HIDDEN + 8)
#define IPLAN_AC_KS_SLOT_FLOATS (16 * IPLAN_AC_HIDDEN + 32)   /* a workgroup's partial fc1 sums of one row tile + row sums, row sums of squares */

typedef struct {
    const float* params;        /* arena of the actors (or critics) */
    int64_t params_s_net;
    int64_t off[IPLAN_AC_NPARAM];
    int32_t n_out;              /* n_actions for actors, 1 for critics */
} IplanAcNet;

/* Row r (< rows) of agent `net` lives at physical row  pr = (r / T) * T_phys + r % T  of each
 * source; entity i of source k at  src[k] + net*s_net[k] + pr*s_row[k] + i*w[k]. */
typedef struct {
    int32_t N;
    int32_t w[3];               /* widths per entity (0 = source absent) */
    const float* src[3];
    int64_t s_net[3], s_row[3];
    int32_t n_actions;          /* width of the last-action one-hot (0 = obs_last_action False) */
    const int32_t* last_action; /* hot index per row or -1 (all zeros): last_action[net*la_s_net + pr*la_s_row] */
    int64_t la_s_net, la_s_row;
    int32_t n_id;               /* width of the agent-id one-hot (0 = obs_agent_id False); hot index = net */
    int32_t T, T_phys;          /* logical / physical steps per episode (equal when rows are contiguous) */
    const int64_t* last_action64; /* alternative to last_action: int64 indices read in place (e.g. EpisodeBatch "actions"), */
    int64_t la64_s_net, la64_s_row; /* last_action64[net*la64_s_net + pr*la64_s_row]; used when last_action == NULL      */
} IplanAcFeatures;

typedef struct {
    int32_t n_agents, rows;
    int32_t which;              /* 0 = actors only, 1 = critics only, 2 = both */
    int32_t ksplit;             /* 1 or 8: waves cooperating on the F-contraction of one 16-row tile */
    IplanAcFeatures feat;
    IplanAcNet actor, critic;
    const float* h_actor;       /* GRU state rows of M floats: h + net*hs_net + pr*hs_row */
    const float* h_critic;
    int64_t hs_net, hs_row;
    float* h_actor_out;         /* [n_agents, rows, M] contiguous (NULL = discard) */
    float* h_critic_out;
    /* actor head */
    const int32_t* avail;       /* [.., n_actions] int32, avail + net*av_s_net + pr*av_s_row; NULL = all available */
    int64_t av_s_net, av_s_row;
    int32_t mode;               /* 0 = argmax(probs), 1 = sample argmax(probs / q), 2 = evaluate given actions */
    const float* q_noise;       /* mode 1: Exp(1) samples [n_agents, rows, n_actions] (torch.multinomial's trick) */
    const int64_t* actions_in;  /* mode 2: actions_in[net*act_s_net + pr*act_s_row] */
    int64_t act_s_net, act_s_row;
    int64_t* actions_out;       /* [n_agents, rows] (modes 0,1) */
    float* logp;                /* [n_agents, rows] log-prob of the chosen / given action */
    float* entropy;             /* [n_agents, rows] per-row entropy (NULL = skip) */
    float* probs;               /* [n_agents, rows, n_actions] (NULL = skip) */
    /* critic head */
    float* values;              /* [n_agents, rows] */
    /* activations for the backward pass, [2, n_agents, rows, IPLAN_AC_SAVE_FLOATS] (NULL = inference) */
    float* saved;
    /* optional strided destinations (all-zero strides = the contiguous defaults above) so that a rollout step
     * writes straight into the episode buffer: h_*_out rows at net*ho_s_net + r*ho_s_row, actions_out at
     * net*ao_s_net + r*ao_s_row, and the action's one-hot (n_actions floats) at onehot_out + net*oh_s_net + r*oh_s_row */
    int64_t ho_s_net, ho_s_row;
    int64_t ao_s_net, ao_s_row;
    float* onehot_out;
    int64_t oh_s_net, oh_s_row;
    /* LayerNorm(F) statistics (mean, rstd) per (net, physical row): the features do not change across PPO
     * epochs, so they are computed once (mode 1 = compute + store) and re-read (mode 2); mode 0 = private.   */
    float* ln_stats;
    int64_t ln_stats_s_net;
    int32_t ln_stats_mode;
    int64_t* phase_clocks;      /* optional profiling aid: wall_clock64 of workgroup (0,0,0) at entry, after the
                                   LayerNorm statistics, after the fc1 contraction, at the end of the tail; NULL = off */
    /* optional pre-packed fc1 operands (iplan_ac_pack_fc1): fc1.weight and feature_norm.{weight,bias} of every net in the
     * kernels' own K order and MFMA fragment order, so that a wave's fragment load is ONE contiguous 1 KiB block instead of
     * 64 scattered 64-byte pieces of a row-major [64, F] matrix (the rollout contraction was bound by exactly that:
     * profiles/r02b_notes.md).  NULL = read the arena in place.  The caller repacks after every weight update.          */
    const float* packed_actor;
    const float* packed_critic;
    int64_t packed_s_net;
    /* optional (which = 2, ksplit = 1, ln_stats_mode = 2): fc1's pre-activation  W LN_F(x)  (WITHOUT fc1.bias) of every row, as
     * iplan_ac_fc1_split_fwd leaves it, [2, n_agents, rows, 64]; the launch then skips the F-wide contraction and runs the
     * 64-wide tail only.  NULL = contract here.                                                                            */
    const float* fc1_pre;
    /* optional, rollout shape only (ksplit = 8, ln_stats_mode = 0, packed operands, saved = NULL): the F-wide contraction of a
     * (row tile, net) unit is split over ksplit_wg (2 .. 8) workgroups; their partial sums meet in ks_scratch
     * [units, ksplit_wg, IPLAN_AC_KS_SLOT_FLOATS] and the last arrival (ticket in ks_count [units], zero before the first
     * launch, left zero by every launch) adds them in workgroup order and runs the tail.  units = ceil(rows / 16) * n_agents *
     * (which == 2 ? 2 : 1).  0 / 1 = off.                                                                                   */
    int32_t ksplit_wg;
    float* ks_scratch;
    int32_t* ks_count;
    int32_t act_tanh;           /* MLPBase activation (mlp.py:10, args.use_ReLU): 0 = ReLU (shipped), 1 = tanh -- forward and, through
                                   IplanAcBwdArgs.fwd, the backward tail (round 4: the second activation of this kernel family) */
    int32_t fc1_pre_parts;      /* 0 / 1: fc1_pre is one array; > 1: that many arrays [2, n_agents, rows, 64] back to back
                                   (iplan_ac_fc1_split_fwd with kparts), added in order                                       */
} IplanAcFwdArgs;

int iplan_ac_fwd(const IplanAcFwdArgs* args, iplan_stream_t stream);
/* One rollout vector step as ONE launch (runners/ippo_parallel_runner.py:166-268, controllers/dcntrl_controller.py:27-58): the
 * latent updates of step t (iplan_gat_enc_fwd) and, behind them in the same grid, select_actions_ippo of step t + 1, which reads
 * exactly what they write (the environment steps after the action selection, not between the two).  `ac` must be a rollout-shaped
 * iplan_ac_fwd argument set (ksplit 8, nothing saved).  `sync`: 3 int32 counters owned by the caller, zero before the first use,
 * private to one stream; [0] / [1] are back at zero when the launch ends, [2] != 0 reports a wait that gave up (poll limit). */
int iplan_gat_enc_ac_fwd(const IplanGatFwdArgs* gat, const IplanEncFwdArgs* enc, const IplanAcFwdArgs* ac, int32_t* sync, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Gumbel(0, 1) samples  g = -log(-log(u)),  u uniform on (0, 1): the noise F.gumbel_softmax draws inside
 * GAT_Net.forward (nova/GAT_Net.py:93, `-empty_like(logits).exponential_().log()`), for a whole rollout in one launch
 * (one pass over `out` instead of the three of the torch expression).  Counter based: out[i] depends on (seed, i) only, so
 * any partition of the range gives the same samples.  n % 4 == 0, out 16-byte aligned.
 */
int iplan_gumbel_noise(float* out, int64_t n, uint64_t seed, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * clip_grad_norm_ + torch.optim.Adam on flat arenas (learners/ippo_learner.py:204-221,
 * nova/prediction_policy.py:231-241, nova/stable_behavior_policy.py:252-262), all nets per launch.
 */
#define IPLAN_MAX_NETS 16

/* out[net*out_stride + slot] = sum of squares of grad[net*stride + off .. + n) (fixed order). */
int iplan_grad_sqnorm(const float* grad, int64_t stride, int64_t off, int64_t n, int32_t n_nets,
                      float* out, int32_t out_stride, int32_t slot, iplan_stream_t stream);

typedef struct {
    float* param;               /* arena base; net k slice at + k*stride + off, n elements        */
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t stride, off, n;
    int32_t n_nets;
    const float* sqnorm;        /* squared grad norms (iplan_grad_sqnorm output); NULL = no clip  */
    int32_t sqnorm_stride, sqnorm_slot;
    float max_norm;             /* grads scaled by min(1, max_norm / (norm + 1e-6))               */
    int32_t write_clipped;      /* store the clipped gradient back (clip_grad_norm_ is in place)  */
    float lr, beta1, beta2, eps;
    float bc1[IPLAN_MAX_NETS];      /* 1 - beta1^step   per net                                   */
    float bc2_sqrt[IPLAN_MAX_NETS]; /* sqrt(1 - beta2^step)                                       */
    float weight_decay;         /* torch.optim.Adam's L2 form: g <- g + weight_decay * p, after the clip (0 = off) */
} IplanAdamArgs;

int iplan_adam_step(const IplanAdamArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Weight-gradient contraction (the autograd of every nn.Linear / nn.GRU / nn.GRUCell weight on the
 * path: torch's addmm/GRU backward under loss.backward() at nova/prediction_policy.py:228,
 * nova/stable_behavior_policy.py:249, learners/ippo_learner.py:202,216):
 *     dW[o][k] = scale * sum_rows dY[row][ocol(o)] * X[row'][x_col0 + k],   db[o] = scale * sum_rows dY[row][ocol(o)]
 * rows = (outer, inner), outer < n_outer, inner < n_inner;  row' = (outer, inner + x_shift); when
 * inner + x_shift falls outside [0, n_inner) the row x0[outer] is used (zeros if x0 == NULL).
 * ocol(o) = o < seg_split ? seg_c0 + o : seg_c1 + (o - seg_split)  selects the columns of dY.
 * Results are written (beta = 0) or accumulated (beta = 1) into the gradient arena of each net at
 * dw_off + o*dw_ld + dw_col0 + k  /  db_off + o  (offset -1 = not wanted).  Fixed summation order.
 */
#define IPLAN_WGRAD_MAX 16
#define IPLAN_WGRAD_MAX_CHUNKS 128

typedef struct {
    const float* dy;
    int64_t dy_s_net, dy_s_outer, dy_s_inner;
    const float* x;
    int64_t x_s_net, x_s_outer, x_s_inner;
    const float* x0;
    int64_t x0_s_net, x0_s_outer;
    int64_t dw_off, db_off;
    int64_t ws_off;             /* filled by the library */
    int32_t O, K;
    int32_t seg_split, seg_c0, seg_c1;
    int32_t x_col0, x_shift;
    int32_t n_outer, n_inner;
    int32_t dw_ld, dw_col0;
    float beta, scale;
    /* Column-grouped operands (0 = plain rows): column c of dY / X lives at element (c >> 4) * cg_stride + (c & 15) of its row
     * -- the behaviour decoder's records are stored [chain tile][16-column group][step][chain][16] so that a row's 16-column
     * group is contiguous over (step, chain) (rows = (tile, step * 16 + chain), row stride 16).  dY columns are then addressed
     * through seg_c0 / seg_c1 and X columns through x_col0 (not by offsetting the pointers).                                */
    int32_t dy_cg_stride, x_cg_stride;
    /* x_shift < 0 only: the |x_shift| rows in front of every outer index's first row exist in memory (the previous steps of a
     * window range that does not start at step 0) and are read in place instead of x0 / zeros.                              */
    int32_t x_pre_valid;
} IplanWgradProblem;

typedef struct {
    int32_t n_problems, n_nets;
    float* grad;                /* gradient arena base */
    int64_t grad_s_net;
    float* workspace;           /* >= iplan_wgrad_workspace_floats() floats */
    int64_t workspace_floats;
    IplanWgradProblem p[IPLAN_WGRAD_MAX];
} IplanWgradArgs;

size_t iplan_wgrad_workspace_floats(const IplanWgradArgs* args);
int iplan_wgrad(IplanWgradArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the recurrent actor / critic (autograd of R_Actor.evaluate_actions /
 * R_Critic.forward under learners/ippo_learner.py:202,216).  Launch order:
 *   iplan_ac_fwd(mode 2, saved != NULL) -> loss gradients per row -> iplan_ac_bwd_tail ->
 *   iplan_wgrad problems for the 64-wide layers (host assembles them from `dsave` / `saved`) and
 *   iplan_ac_bwd_fc1 -> iplan_ac_bwd_fc1_finalize (needs fc1.bias' gradient from iplan_wgrad).
 */
#define IPLAN_AC_DSAVE_FLOATS (6 * IPLAN_AC_HIDDEN + 16) /* dz1 | dz2 | dr dz dn_i dn_h | dhead(16) */
#define IPLAN_AC_LNPART_FLOATS (6 * IPLAN_AC_HIDDEN)     /* [rnn.norm w,b | fc2.0.2 w,b | fc1.2 w,b] */

typedef struct {
    IplanAcFwdArgs fwd;         /* the descriptor of the forward launch (mode 2, saved != NULL)        */
    const float* g_logp;        /* [n_agents, rows] dLoss/dlogp                                          */
    const float* g_entropy;     /* [n_agents, rows] dLoss/d(entropy of the row); NULL -> g_entropy_const */
    float g_entropy_const;
    const float* g_values;      /* [n_agents, rows] dLoss/dvalue                                         */
    float* dsave;               /* [2, n_agents, rows, IPLAN_AC_DSAVE_FLOATS]                            */
    float* ln_part;             /* [2, n_agents, ceil(rows/16), IPLAN_AC_LNPART_FLOATS]                  */
    float* g_part;              /* [2, n_agents, fc1_chunks, 64, Kpad] partial G tiles, Kpad = iplan_ac_kpad()  */
    int32_t fc1_chunk_rows;     /* rows per chunk, multiple of 16                                        */
    int32_t fc1_chunks;
    float* actor_grad;          /* gradient arenas (fc1.weight, feature_norm.* are written by finalize)  */
    float* critic_grad;
    int64_t actor_grad_s_net, critic_grad_s_net;
    const float* xb;            /* iplan_ac_bwd_fc1_split only: the row-major-K fragments of iplan_ac_xhat_pack  */
} IplanAcBwdArgs;

/* iplan_ac_pack_fc1: packed[net] = { Wp [KT][4 o-tiles][64 lanes][4] | gamma_p [KT][16] | beta_p [KT][16] | W gamma [64] | W beta [64] },
 * KT = iplan_ac_kpad / 16 (the last two: fc1.weight x feature_norm.{weight, bias}, operands of the folded LayerNorm(F)):
 * Wp[T][oo][lane (n, g)][q] = fc1.weight[16 oo + n][column of k-order position 16 T + 4 g + q] (0 past a block's end).      */
typedef struct {
    int32_t n_nets;
    IplanAcFeatures feat;        /* only N, w[], n_actions, n_id are read                             */
    const float* params;
    int64_t params_s_net, off_w1, off_fn_w, off_fn_b;
    float* packed;
    int64_t packed_s_net;        /* >= iplan_ac_packed_floats(feat)                                   */
    int32_t parts;               /* 0 = everything; 1 = Wp | gamma_p | beta_p only (all the streaming forward of a PPO
                                    epoch reads); 2 = W gamma | W beta only (operands of the rollout's folded form)    */
} IplanAcPackArgs;
int64_t iplan_ac_packed_floats(const IplanAcFeatures* feat);
int iplan_ac_pack_fc1(const IplanAcPackArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * fc1 of the PPO epochs on the bf16 matrix cores (learners/ippo_learner.py:190-221: the 15 epochs evaluate the same stored
 * rows, only the weights move).  The normalised feature rows  xhat = (x - mean) * rstd  (LayerNorm(F) without its affine
 * part, which is folded into the weights:  fc1(LN(x)) = (W o gamma) xhat + W beta + b)  are gathered ONCE per train() into
 * two fp32 fragment-major copies -- `xf` for the forward (K = features), `xb` for the weight gradient (K = rows) -- and every
 * epoch runs
 *    iplan_ac_fc1_split_fwd    z1 = (W o gamma) xhat + W beta     for the actor and the critic of every agent, one pass over xf
 *    iplan_ac_fwd(fc1_pre=z1)  the 64-wide tail
 *    iplan_ac_bwd_fc1_split    G = dz1^T xhat  for both nets, one pass over xb  (then iplan_ac_bwd_fc1_finalize as before)
 * with every fp32 operand split into three bf16 pieces in registers (x = p0 + p1 + p2 exactly) and the six largest piece
 * products accumulated in fp32 on v_mfma_f32_16x16x32_bf16: the result agrees with the fp32 contraction to fp32 round-off.
 * Fragment layouts (KT = iplan_ac_kpad / 16 k-tiles of the source-major feature order, KS = ceil(KT / 2) k-steps of 32):
 *    xf [n_agents, 2 * RB, KS, 64 lanes, 8]   lane (n, g) of row tile t, slot j: xhat[16 t + n][k-tile 2 ks + (j >> 2)][4 g + (j & 3)]
 *    xb [n_agents, RB, KT, 64 lanes, 8]       lane (f, g) of 32-row block b, slot j: xhat[32 b + 8 g + j][k-tile T][f]
 * RB = ceil(rows / 32); rows past `rows` and features past F are zeros.                                                    */
typedef struct {
    int32_t n_agents, rows;
    IplanAcFeatures feat;
    const float* ln_stats;      /* (mean, rstd) per (net, physical row), as iplan_ac_fwd(ln_stats_mode = 1) stored them */
    int64_t ln_stats_s_net;
    float* xf;
    float* xb;
} IplanAcXhatArgs;
int64_t iplan_ac_xhat_floats(const IplanAcFeatures* feat, int32_t rows, int32_t which);   /* per agent: which 0 = xf, 1 = xb */
int iplan_ac_xhat_pack(const IplanAcXhatArgs* args, iplan_stream_t stream);

typedef struct {
    int32_t n_agents, rows;
    IplanAcFeatures feat;        /* only N, w[], n_actions, n_id are read                                               */
    IplanAcNet actor, critic;
    const float* xf;
    void* wsplit;                /* workspace, n_agents * KS * 24576 bytes: the bf16 pieces of W o gamma in fragment order */
    float* wbeta;                /* workspace [2, n_agents, 64]: W beta                                                  */
    float* z1;                   /* out [kparts][2, n_agents, rows, 64]: part p holds the contribution of K-step range p (part 0 + W beta) */
    int32_t kparts;              /* 0 / 1: one part (the whole K loop in one workgroup).  > 1 -- small batches only (<= iplan_ac_fc1_split_parts):
                                    a data-parallel rank's 2 880 rows are 113 workgroups of one 79-step K chain each; the chain is dealt
                                    to `kparts` workgroups and iplan_ac_fwd(fc1_pre, fc1_pre_parts) adds the parts in order            */
} IplanAcFc1SplitArgs;
/* the number of parts the host layer should ask for at this size (1 = the full-buffer shape) */
int iplan_ac_fc1_split_parts(int32_t n_agents, int32_t rows);
int iplan_ac_fc1_split_fwd(const IplanAcFc1SplitArgs* args, iplan_stream_t stream);
/* args->xb set, fwd.which = 2, fc1_chunk_rows a multiple of 32; g_part as for iplan_ac_bwd_fc1.  iplan_ac_fc1_split_chunks:
 * the row chunking that fills the chip once (returns fc1_chunks, writes fc1_chunk_rows). */
int iplan_ac_fc1_split_chunks(const IplanAcFeatures* feat, int32_t n_agents, int32_t rows, int32_t* chunk_rows);
int iplan_ac_bwd_fc1_split(const IplanAcBwdArgs* args, iplan_stream_t stream);

int iplan_ac_kpad(const IplanAcFeatures* feat);   /* padded length of the kernels' source-major feature order */
int iplan_ac_fc1_groups(const IplanAcFeatures* feat); /* wave jobs along the feature axis of iplan_ac_bwd_fc1 (the caller
                                                          sizes fc1_chunks so that groups x chunks x nets fill the chip:
                                                          2048 waves, two per SIMD) */
int iplan_ac_bwd_tail(const IplanAcBwdArgs* args, iplan_stream_t stream);
int iplan_ac_bwd_fc1(const IplanAcBwdArgs* args, iplan_stream_t stream);
int iplan_ac_bwd_fc1_finalize(const IplanAcBwdArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PPO bookkeeping of IPPOLearner.train (learners/ippo_learner.py).
 * iplan_ppo_prepare: compute_returns (:344-365, GAE) + advantage normalisation (:273-279):
 *   mask_t = 1 - terminated_t;  delta_t = r_t + gamma V_{t+1} mask_{t+1} - V_t;
 *   gae_t = delta_t + gamma lam mask_{t+1} gae_{t+1};  ret_t = gae_t + V_t;
 *   adv = (ret - V) zeroed where mask == 0, then (adv - mean) / (unbiased std + 1e-5) over all bs*T.
 */
typedef struct {
    int32_t n_agents, bs, T;
    const float* reward;         /* reward[net*rw_s_net + b*rw_s_ep + t*rw_s_t], t < T              */
    int64_t rw_s_net, rw_s_ep, rw_s_t;
    const uint8_t* terminated;   /* same addressing, t <= T                                          */
    int64_t tm_s_net, tm_s_ep, tm_s_t;
    const float* values;         /* [n_agents, bs, T+1] critic values of every stored step           */
    float gamma, lam;
    float* returns;              /* [n_agents, bs, T]                                                */
    float* adv;                  /* [n_agents, bs, T] normalised advantages                          */
    float* mask;                 /* [n_agents, bs, T] 1 - terminated                                 */
    float* value_preds;          /* [n_agents, bs, T] = values[:, :, :T]                             */
    int32_t skip_norm;           /* 1 = leave adv un-normalised (data-parallel callers normalise with the
                                    mean / std over ALL ranks' rows between this call and iplan_ppo_loss)      */
    int32_t no_gae;              /* 1 = use_gae False (:360-362): ret_T = V_T, ret_t = ret_{t+1} gamma mask_{t+1} + r_t */
} IplanPpoPrepareArgs;

int iplan_ppo_prepare(const IplanPpoPrepareArgs* args, iplan_stream_t stream);

/* iplan_ppo_adv_norm: the advantage normalisation of ippo_learner.py:276-279 over the rows of ALL data-parallel ranks, in
 * three launches around two sum all-reduces (the caller only moves bytes between them; every arithmetic step is here):
 *   phase 0:  sum[net]   = sum_i adv[net, i]                              (this rank's rows)       -> all-reduce sum
 *   phase 1:  sqdev[net] = sum_i (adv[net, i] - sum[net] / count)^2        (sum = global)           -> all-reduce sqdev
 *   phase 2:  adv[net, i] = (adv[net, i] - mean) / (sqrt(sqdev[net] / (count - 1)) + 1e-5)
 * count = rows of all ranks.  With one rank this reproduces iplan_ppo_prepare's own normalisation.                    */
typedef struct {
    int32_t n_agents, n;         /* n = this rank's rows per agent                                   */
    int64_t row_stride;          /* per-agent stride of adv                                          */
    float* adv;                  /* [n_agents, row_stride]                                           */
    float* sum;                  /* [n_agents]                                                       */
    float* sqdev;                /* [n_agents]                                                       */
    float count;                 /* rows per agent over all ranks                                    */
    int32_t phase;
} IplanAdvNormArgs;

int iplan_ppo_adv_norm(const IplanAdvNormArgs* args, iplan_stream_t stream);

/* iplan_ppo_loss: ppo_update's losses (:185-197) and cal_value_loss (:128-159) over the first
 * `rows` rows of every agent, plus dLoss/dlogp and d(value_loss_coef * value_loss)/dvalue per row.
 * stats[net][0..4] = policy_loss, value_loss, mean ratio, mean entropy, sum(mask).              */
typedef struct {
    int32_t n_agents, rows;
    int64_t row_stride;          /* per-agent stride of the [n_agents, bs*T] inputs below            */
    const float* logp;           /* [n_agents, rows] current log-probs (iplan_ac_fwd output)         */
    const float* entropy;        /* [n_agents, rows]                                                 */
    const float* values;         /* [n_agents, rows] current values                                  */
    const float* old_logp;       /* [n_agents, row_stride]                                           */
    const float* adv;
    const float* value_preds;
    const float* returns;
    const float* mask;
    float clip, huber_delta, value_loss_coef;
    float* g_logp;               /* [n_agents, rows]                                                 */
    float* g_values;             /* [n_agents, rows]                                                 */
    float* stats;                /* [n_agents, 8]                                                    */
    const float* mask_sum;       /* optional [n_agents]: sum(mask) over the rows of ALL data-parallel ranks -- the
                                    losses' denominator (NULL = this launch's own rows)                        */
    int32_t flags;               /* IPLAN_PPO_* bits below; 0 = config/algs/ippo.yaml as shipped                 */
    float row_count;             /* rows of ALL data-parallel ranks, the denominator of the *_MEAN forms (0 = rows) */
    int32_t n_parts;             /* 0 / 1: one workgroup per agent (stats as above).  P > 1: the rows of an agent are dealt to P
                                    workgroups (contiguous ranges); REQUIRES mask_sum (every workgroup needs the denominator up
                                    front); stats is [n_agents, P, 8] and holds each range's SHARE of the five values -- the caller
                                    adds the P shares (fixed order: reproducible).  22 950 rows: 46.5 -> ~10 us per launch.     */
} IplanPpoLossArgs;
#define IPLAN_PPO_MSE 1          /* use_huber_loss False: e^2 / 2 (:145-147)                                     */
#define IPLAN_PPO_NO_VCLIP 2     /* use_clipped_value_loss False: the unclipped value loss alone (:149-152)      */
#define IPLAN_PPO_VALUE_MEAN 4   /* use_value_active_masks False: plain mean over the rows (:154-157)            */
#define IPLAN_PPO_POLICY_MEAN 8  /* use_policy_active_masks False: plain mean over the rows (:190-196)           */

int iplan_ppo_loss(const IplanPpoLossArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward of GAT_Net.forward (autograd under loss.backward() at nova/prediction_policy.py:228) for
 * every (net, env) scene in one launch: GRUCell', gated soft attention', gumbel gate', BPTT through
 * the bidirectional pair GRU (its recurrent weight / bias gradients accumulated in-kernel), node
 * projections'.  Emits node-level pre-activation gradients; their weight gradients are iplan_wgrad
 * contractions over them (host assembles the problem list).
 * node_dy row layout (floats): ENC 0 (H) | DA_fwd 32 | DB_fwd 128 | DA_rev 224 | DB_rev 320 (3H each)
 *                              | DQ 416 | DK 448 | DV 480 (A each) | CELL 512 (dr dz dn_i dn_h, 4A).
 * hard_part row (per scene): [dir][tile<4][H] partial sums of dDelta * h, then sum(dDelta) at 8H.
 */
#define IPLAN_GAT_NODE_DY 640
#define IPLAN_GAT_HARD_PART (8 * IPLAN_GAT_HIDDEN + 16)
#define IPLAN_GAT_WHH_PART (3 * IPLAN_GAT_HIDDEN * IPLAN_GAT_HIDDEN + 3 * IPLAN_GAT_HIDDEN)   /* dW_hh [3H, H] | db_hh [3H] */

typedef struct {
    IplanGatFwdArgs fwd;        /* descriptor of the forward launch (saved.* all non-NULL)           */
    const float* g_out;         /* dLoss/d out, rows of A floats: (net,b,i)                          */
    int64_t g_s_net, g_s_b;
    float* dgru;                /* scratch [n_nets, 2, B, ceil(N/16), N-1, 2, 3H] (kernel-private)   */
    float* node_dy;             /* [n_nets, B*N, IPLAN_GAT_NODE_DY]                                  */
    float* hard_part;           /* [n_nets, B, IPLAN_GAT_HARD_PART]                                  */
    float* whh_part;            /* scratch [n_nets, B, 2, 4, IPLAN_GAT_WHH_PART]                     */
    float* grad;                /* gradient arena: hard_bi_GRU.weight_hh_l0{,_reverse} and bias_hh_l0{,_reverse} are
                                   WRITTEN here (at fwd.off[...]); all other GAT gradients are iplan_wgrad contractions
                                   over node_dy / hard_part assembled by the host                                    */
    int64_t grad_s_net;
} IplanGatBwdArgs;

int iplan_gat_bwd(const IplanGatBwdArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Prediction_Decoder.forward (nova/prediction_net.py:40-63; DecoderRNN :6-26) + the masked-L1 loss
 * of Prediction_policy.learn (nova/prediction_policy.py:223-225), and their backward.
 * Rows are (sample s, entity i) = s*N + i; hidden size == 32; d <= 16.
 * Parameter tensors of DecoderRNN in state_dict order (also used by the behaviour decoder):
 */
enum {
    IPLAN_DEC_LIN_W = 0,    /* decoder.linear.weight      [Hd, In]   */
    IPLAN_DEC_LIN_B,        /* decoder.linear.bias        [Hd]       */
    IPLAN_DEC_WIH,          /* decoder.rnn.weight_ih_l0   [3Hd, Hd]  */
    IPLAN_DEC_WHH,          /* decoder.rnn.weight_hh_l0   [3Hd, Hd]  */
    IPLAN_DEC_BIH,          /* decoder.rnn.bias_ih_l0     [3Hd]      */
    IPLAN_DEC_BHH,          /* decoder.rnn.bias_hh_l0     [3Hd]      */
    IPLAN_DEC_OUT_W,        /* decoder.out.weight         [d, Hd]    */
    IPLAN_DEC_OUT_B,        /* decoder.out.bias           [d]        */
    IPLAN_DEC_NPARAM
};
#define IPLAN_PDEC_SAVE 256   /* per (row, step): xin 0 (16) | u 16 | r 48 | z 80 | n 112 | hn 144 | h 176 | a 208 (32 each) | y 240 (16) */
#define IPLAN_PDEC_DSAVE 176  /* per (row, step): dy 0 (16) | du 16 | dr 48 | dz 80 | dn_i 112 | dn_h 144 (32 each)                      */

typedef struct {
    int32_t n_nets, rows, N, P, d;
    const float* x0;            /* [n_nets, rows, d]     current state (decoder input of step 0)        */
    const float* h0;            /* [n_nets, rows, 32]    GAT output = initial hidden state              */
    const float* target;        /* [n_nets, rows, P, d]  actual next states                             */
    const float* mask;          /* [n_nets, rows / N]    per-sample mask                                */
    const float* keep;          /* [n_nets, P, rows, 32] dropout keep flags {0,1}; NULL = no dropout    */
    float drop_p;
    const int32_t* teacher;     /* [n_nets, P] 1 = feed the actual state to the next step; NULL = never */
    const float* params;
    int64_t params_s_net;
    int64_t off[IPLAN_DEC_NPARAM];
    float* pred;                /* [n_nets, rows, P, d]                                                 */
    float* saved;               /* [n_nets, rows, P, IPLAN_PDEC_SAVE]                                   */
    float* loss_part;           /* [n_nets, ceil(rows/16)]                                              */
    float* loss;                /* [n_nets]  sum|target-pred|*m / (sum m + 1e-10) * d * P               */
    float* dsave;               /* backward: [n_nets, rows, P, IPLAN_PDEC_DSAVE]                        */
    float* g_h0;                /* backward: [n_nets, rows, 32] dLoss/d h0                              */
    const float* mask_sum;      /* optional [n_nets]: sum(mask) over the samples of ALL data-parallel ranks (the
                                   loss normaliser); NULL = this launch's own samples                       */
} IplanPdecArgs;

int iplan_pdec_fwd(const IplanPdecArgs* args, iplan_stream_t stream);
int iplan_pdec_bwd(const IplanPdecArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Behavior_policy.learn, soft update (nova/stable_behavior_policy.py:161-279): the whole episode of
 * every (env, entity) chain of every agent-net in one forward and one backward launch.
 * Encoder parameters: IPLAN_ENC_* (EncoderRNN); decoder parameters: IPLAN_DEC_* of
 * behavior_decoder[i].decoder (DecoderRNN, input d + Z, hidden 64).  rows = E * N, J = T - 1 - L.
 * Per-step records (floats):
 *   saved_dec  [x_t || latent] 0 (d+Z <= 16 cols) | 16 (16, unused) | u 32 | r 96 | z 160 | n 224 | hn 288 | h 352 | a 416 (64 each) | y 480 (16)
 *   saved_enc  u 0 | r 32 | z 64 | n 96 | hn 128 | h 160 (32 each)
 *   dsave_dec  dy 0 (16) | du 16 | dr 80 | dz 144 | dn_i 208 | dn_h 272 (64 each)
 * The three per-step records are stored column-grouped: [n_nets, ceil(rows / 16) chain tiles, columns / 16 groups, J * L steps,
 * 16 chains, 16 floats] -- column c of (chain, step) at group c >> 4, float c & 15.  A wave's access to one 16-column group of
 * its 16 chains is then one contiguous 1 KiB block, and a group's rows are contiguous over (step, chain): the weight-gradient
 * contraction reads them as rows (tile, step * 16 + chain) with IplanWgradProblem.dy_cg_stride / x_cg_stride = J * L * 256.
 * The chain slots of a ragged last tile are never written: the caller zeroes them in the decoder records (nothing reads them in
 * saved_enc, which is stored the same way: 12 groups).
 */
#define IPLAN_BEH_SAVE_DEC 496
#define IPLAN_BEH_SAVE_ENC 192
#define IPLAN_BEH_SAVE_LAT 32        /* per (row, window): softmax output 0 (16) | latent used by the decoder 16 (16) */
#define IPLAN_BEH_DSAVE_DEC 336
#define IPLAN_BEH_DSAVE_LAT 16
#define IPLAN_BEH_DEC_THIN_PART 2144 /* per workgroup of the BPTT's second form: out.weight [16][64] | out.bias [16] | linear.weight [64][16] | linear.bias [64] */
#define IPLAN_BEH_ENC_PART 7408       /* per-wave encoder weight-gradient partial: W_ih 0 | W_hh 3072 | b_ih 6144 | b_hh 6240
                                         | lin.W [32][16] 6336 | lin.b 6848 | out.W [16][32] 6880 | out.b 7392          */

typedef struct {
    int32_t n_nets, E, N, T, L, d, Z;  /* T = stored steps used (= episode_limit)                      */
    const float* hist;          /* x(net,e,t,i,c) = hist[net*h_s_net + e*h_s_e + t*h_s_t + i*d + c]     */
    int64_t h_s_net, h_s_e, h_s_t;
    const float* mask;          /* [n_nets, E, T] loss mask (env-dependent polarity applied by caller)  */
    const uint8_t* keep;        /* [n_nets, J, rows, L, 64] dropout keep flags; NULL = drawn in-kernel  */
    uint64_t seed;              /* seed of the in-kernel counter-based Bernoulli draw                   */
    float drop_p, coef, thres;  /* decoder_dropout, soft_update_coef, thres_small_variation             */
    const float* enc_params;
    int64_t enc_s_net;
    int64_t enc_off[IPLAN_ENC_NPARAM];
    const float* dec_params;
    int64_t dec_s_net;
    int64_t dec_off[IPLAN_DEC_NPARAM];
    float* saved_dec;           /* column-grouped (above): n_nets * ceil(rows/16)*16 * J * L * IPLAN_BEH_SAVE_DEC floats */
    float* saved_enc;           /* column-grouped: n_nets * ceil(rows/16)*16 * J * L * IPLAN_BEH_SAVE_ENC floats       */
    float* saved_lat;           /* [n_nets, rows, J, IPLAN_BEH_SAVE_LAT]  softmax output of window j    */
    float* loss_part;           /* [n_nets, ceil(rows/16), 2]                                           */
    float* loss;                /* [n_nets, 2]  behaviour error, stability error                        */
    float* dsave_dec;           /* backward: column-grouped like saved_dec, IPLAN_BEH_DSAVE_DEC columns               */
    float* dsave_lat;           /* backward: [n_nets, rows, J, IPLAN_BEH_DSAVE_LAT] d(loss)/d(latent_j) through
                                   the decoder inputs of window j (decoder BPTT -> encoder BPTT hand-off) */
    /* single-window decoder mode = Behavior_Latent_Decoder.forward (nova/behavior_net.py:55-69): set T = L + 2 (one
     * window) and pass the window, latent and hidden state explicitly; the encoder and the loss are skipped.     */
    const float* win;           /* [n_nets, rows, L, d] or NULL                                          */
    const float* lat_in;        /* [n_nets, rows, Z]                                                     */
    const float* hd_in;         /* [n_nets, rows, 64]                                                    */
    float* pred_out;            /* [n_nets, rows, L, d]                                                  */
    float* hd_out;              /* [n_nets, rows, 64]                                                    */
    /* hard = 1: the iPLAN-Hard ablation (nova/behavior_policy.py:119-215): non-overlapping windows
     * (window j = steps j*L .. j*L+L-1, J = T/L - 1), the latent is REPLACED by the encoder output, one global
     * normaliser sum(mask[:, :J*L]) for the whole loss and the mask taken at the window's own steps.            */
    int32_t hard;
    float* enc_part;            /* backward: [n_nets, ceil(rows/16), IPLAN_BEH_ENC_PART] per-wave partials */
    float* enc_grad;            /* backward: encoder gradient arena (written at enc_off[])               */
    int64_t enc_grad_s_net;
    int32_t bwd_phase;          /* iplan_beh_bwd: 0 = decoder then encoder, 1 = decoder BPTT only, 2 = encoder BPTT only
                                   (lets the host run the encoder's BPTT on a second stream beside the decoder's
                                   weight-gradient contractions; phase 2 needs phase 1's dsave_lat)               */
    /* BPTT in pieces (bwd_phase 1 or 2): windows [bwd_j_lo, bwd_j_hi) only, processed from the top down;
     * bwd_j_hi == 0 means all windows.  A piece that does not end at window 0 leaves d(loss)/d(h) of its last
     * step in dec_carry (decoder) / enc_carry (encoder, plus d(loss)/d(latent)), the next piece (bwd_j_hi = the
     * previous bwd_j_lo) picks it up -- so the weight-gradient contraction of the rows a decoder piece produced and
     * the encoder piece of the same windows can run beside the next decoder piece.  Encoder pieces accumulate
     * their weight-gradient partials in enc_part; the piece with bwd_j_lo == 0 reduces them into enc_grad.      */
    int32_t bwd_j_lo, bwd_j_hi;
    float* dec_carry;           /* [n_nets, ceil(rows/16), 2, 512]; needed when the decoder runs in pieces */
    /* The forward in pieces, same idea (the encoder of the next windows runs beside the decoder of the current
     * ones): fwd_phase 0 = encoder, decoder and loss reduction; 1 = encoder only; 2 = decoder only; 3 = loss
     * reduction only.  Windows [fwd_j_lo, fwd_j_hi), bottom up; fwd_j_hi == 0 means all.  Hidden states / latent
     * cross the pieces in enc_carry [n_nets, ceil(rows/16), 768] and dec_carry; loss partials accumulate.      */
    int32_t fwd_phase, fwd_j_lo, fwd_j_hi;
    float* enc_carry;
    const float* win_norm;      /* optional [n_nets, J]: per window, sum of the mask over the window's target steps and
                                   the envs of ALL data-parallel ranks (hard update: the one global sum, repeated);
                                   NULL = summed in-kernel over this launch's envs                              */
    float enc_grad_beta;        /* backward: enc_grad = enc_grad_beta * enc_grad + sum of the wave partials (1 = accumulate
                                   over several launches on disjoint env chunks, 0 = overwrite)                  */
    float penalty;              /* backward: behavior_variation_penalty -- weight of the stability term
                                   mean_j sum_{chain, t} max(||x_t - y_t||_2 - thres, 0) / (E_norm L) in the differentiated loss
                                   (nova/stable_behavior_policy.py:238-246); 0 = the shipped configuration       */
    int32_t E_norm;             /* envs the stability term is averaged over (all chunks / ranks); 0 = E           */
    /* Round 4 (decoder BPTT, second form): the THIN decoder weight gradients -- out.weight / out.bias (d x 64) and
     * linear.weight / linear.bias (64 x (d + Z)) -- accumulated IN the BPTT kernel (its waves hold dy, the output-layer input
     * and the Linear's gradient in registers) instead of from row gradients by iplan_wgrad: dec_thin_part != NULL switches it
     * on, the kernel then does NOT store the dy / du columns of dsave_dec, window-range pieces accumulate in the partials, and
     * the piece with bwd_j_lo == 0 reduces them (workgroup order: reproducible) into dec_grad at dec_off[] as
     * dec_grad_beta * old + sum.  Only the second form supports it (iplan_beh_bwd fails if it cannot run that form).   */
    float* dec_thin_part;       /* [n_nets, ceil(ceil(rows/16) / 3), IPLAN_BEH_DEC_THIN_PART] or NULL              */
    float* dec_grad;            /* decoder gradient arena                                                          */
    int64_t dec_grad_s_net;
    float dec_grad_beta;
    int32_t fwd_skip_act;       /* forward (second form): do not store the output layer's input (record columns 416..479) --
                                   only the out.weight contraction of iplan_wgrad reads it, and with dec_thin_part the BPTT
                                   re-forms it from the hidden state it holds anyway: 64 of the 480 floats a chain-step stores */
} IplanBehArgs;

int iplan_beh_fwd(const IplanBehArgs* args, iplan_stream_t stream);
int iplan_beh_bwd(const IplanBehArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Three-layer perceptron of the FC behaviour ablation (nova/behavior_FC_net.py:6-37, Encoder_3FC / Decoder_3FC):
 *   out = [softmax] (W3 tanh(W2 tanh(W1 x + b1) + b2) + b3)   for n_nets stacked nets, rows per net.
 * off[0..5] = linear_1.weight [H,K0], linear_1.bias, linear_2.weight [H,H], linear_2.bias, out.weight [O,H], out.bias.
 * Forward: writes out, saved = [tanh1 (H) | tanh2 (H)] per row and, when `target` is given, the per-wave sums of
 * |target - out| (L1 loss numerator) in loss_part [n_nets, 4 * ceil(rows/64)].
 * Backward: d(out) = g_out, or -sign(target - out) * g_scale when g_out is NULL; writes dsave = [d pre1 (H) | d pre2 (H) |
 * d pre3 (16 * ceil(O/16))] per row (operands of iplan_wgrad) and dx [n_nets, rows, K0] if not NULL.
 */
typedef struct {
    int32_t n_nets, K0, H, O, softmax;
    int64_t rows;
    const float* x;             /* [n_nets, rows, K0]                                                   */
    const float* params;
    int64_t params_s_net;
    int64_t off[6];
    float* out;                 /* [n_nets, rows, O]                                                    */
    float* saved;               /* [n_nets, rows, 2H] (forward: optional)                               */
    const float* target;        /* [n_nets, rows, O] or NULL                                            */
    float* loss_part;
    const float* g_out;         /* backward: [n_nets, rows, O] or NULL                                  */
    float g_scale;
    float* dsave;               /* backward: [n_nets, rows, 2H + 16*ceil(O/16)]                         */
    float* dx;                  /* backward: [n_nets, rows, K0] or NULL                                 */
} IplanMlp3Args;

int iplan_mlp3_fwd(const IplanMlp3Args* args, iplan_stream_t stream);
int iplan_mlp3_bwd(const IplanMlp3Args* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * nova/Seq2Seq.py:41-70 -- Seq2Seq.forward (encoder GRU over the input sequence from a zero state, then pred_length
 * autoregressive decoder steps  y_t = Linear(Dropout(tanh(GRU(y_{t-1}, h))))  from last_location, optionally teacher forced).
 * nn.GRU semantics, batch_first, `layers` stacked layers (1..4) of width H in {32, 64}; rows = N*V.  Inference only.
 * enc_off / dec_off: per layer l  [4l + 0..3] = weight_ih_l<l> [3H, In_l], weight_hh_l<l> [3H, H], bias_ih_l<l>, bias_hh_l<l>
 * (In_0 = input_size for the encoder, output_size for the decoder; In_l = H above); lin_off = decoder.linear.weight [O, H], .bias.
 */
#define IPLAN_S2S_MAX_LAYERS 4
typedef struct {
    int32_t rows, T_in, In, H, layers, P, O;
    const float* x;             /* [rows, T_in, In]                                                     */
    const float* last;          /* [rows, O]   last_location                                            */
    const float* teacher;       /* [rows, P, O] or NULL                                                 */
    const int32_t* coins;       /* [P] 1 = feed teacher[:, t] into step t + 1 (teacher != NULL), or NULL */
    const float* keep;          /* [P, rows, H] dropout keep flags of the decoder output, or NULL (no dropout) */
    float drop_p;
    const float* params;
    int64_t enc_off[4 * IPLAN_S2S_MAX_LAYERS], dec_off[4 * IPLAN_S2S_MAX_LAYERS], lin_off[2];
    float* out;                 /* [rows, P, O]                                                         */
    float* hidden_out;          /* optional [layers, rows, H]: the decoder's final hidden state         */
    float* save;                /* optional (training): the record iplan_seq2seq_bwd reads,
                                   [T_in + P, layers, rows, 6H] = h_prev | r | z | n | gh_n | h_new per GRU step (encoder steps first),
                                   followed by [P, rows, 16 + H] = the decoder step's input (O <= 16 columns) | the output layer's input */
} IplanSeq2SeqArgs;

int iplan_seq2seq_fwd(const IplanSeq2SeqArgs* args, iplan_stream_t stream);

/* iplan_seq2seq_bwd: the autograd of the above (nova/Seq2Seq.py:52-70 under loss.backward()) for a loss on `out` (BPTT through the decoder's feedback -- a step whose
 * input was the previous prediction passes its input gradient on to it -- and through both GRU stacks).  It walks the
 * data gradients and writes the row-level pre-activation gradients; the weight gradients are dY^T X contractions over them
 * (iplan_wgrad; iplan_amd/nova/Seq2Seq.py lists the problems):
 *   dsave [T_in + P, layers, rows, 4H] = dr | dz | dn_i | dn_h per GRU step, followed by [P, rows, 16] = d out (first O columns).
 * No gradient is returned for in_data / last_location / teacher_location (data).                                          */
typedef struct {
    IplanSeq2SeqArgs fwd;       /* the forward launch's arguments (its `save` filled)                   */
    const float* g_out;         /* [rows, P, O] dLoss/d out                                             */
    float* dsave;
} IplanSeq2SeqBwdArgs;

int iplan_seq2seq_bwd(const IplanSeq2SeqBwdArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One-shot peer-to-peer sum all-reduce of a gradient arena over xGMI (the reference has no
 * multi-GPU path -- this replaces the torch.distributed all-reduce behind parallel.DataParallel.all_reduce_grads when
 * IPLAN_P2P_ALLREDUCE=1).  Arenas are 0.3-4 MB: a ring's 2 (n - 1) latency-bound steps cost more than every rank reading
 * its peers' buffers directly.  Protocol, per collective `seq` (same on all ranks, +1 per call, >= 1):
 *   publish: copy `data` into this rank's staging half (seq & 1), then store `seq` into flags[peer][rank] of EVERY rank
 *            (system-scope release);
 *   reduce : wait until flags[rank][p] >= seq for all p (system-scope acquire), then data[i] = sum over p = 0 .. world-1, in
 *            that order on every rank (bit-identical replicas), of stage[p][half][i].
 * Double-buffered staging needs no second barrier: a rank can only publish seq + 2 (the same half) after it finished
 * seq + 1, which needed every peer's seq + 1, published only after that peer finished reading seq.
 * The ONLY entry points that allocate: staging / flag buffers must be shareable between processes, so they come from
 * hipMalloc directly (a caching allocator's sub-allocations cannot be exported); iplan_p2p_free releases them.
 */
#define IPLAN_P2P_MAX_RANKS 8
typedef struct {
    unsigned char bytes[64];                    /* hipIpcMemHandle_t */
} IplanIpcHandle;
int iplan_p2p_alloc(size_t bytes, void** dev_ptr);               /* zero-filled device memory of the current device */
int iplan_p2p_free(void* dev_ptr);
int iplan_p2p_export(void* dev_ptr, IplanIpcHandle* out);        /* hipIpcGetMemHandle */
int iplan_p2p_open(const IplanIpcHandle* handle, void** dev_ptr); /* hipIpcOpenMemHandle (another process' allocation) */
int iplan_p2p_close(void* dev_ptr);
typedef struct {
    int32_t world, rank;
    int64_t count;                              /* floats to reduce, <= capacity, multiple of 4                         */
    float* data;                                /* in / out, 16-byte aligned                                            */
    float* stage[IPLAN_P2P_MAX_RANKS];          /* staging buffers of all ranks (own: local, peers: opened), 2 * capacity floats each */
    uint32_t* flags[IPLAN_P2P_MAX_RANKS];       /* flag arrays of all ranks, IPLAN_P2P_MAX_RANKS words each              */
    int64_t capacity;                           /* floats per staging half, multiple of 4                               */
    uint32_t seq;
    int32_t* error;                             /* device int, set to 1 when the wait gives up after spin_limit polls    */
    int64_t spin_limit;                         /* 0 = wait for ever                                                     */
} IplanP2pArgs;
int iplan_p2p_publish(const IplanP2pArgs* args, iplan_stream_t stream);
int iplan_p2p_reduce(const IplanP2pArgs* args, iplan_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * iplan_obs_history_step: one environment step of the id -> slot history wrapper on the device
 * (observation_wrapper.py:68-141: obs_history_create + obs_history_output + obs_single_history_output).
 * State (caller-owned device memory, zeroed / -1-filled at the head of an episode -- agent_obs_profile_init,
 * observation_wrapper.py:26-46, stays on the host: it runs once per episode):
 *   slot_id[k][i][s] = vehicle id of slot s of (thread k, registered agent i), -1 = free; n_slots[k][i];
 *   win[k][i][s] = the slot's last L entries, right-aligned (the deque views the reference rebuilds every step).
 * A step appends this step's observed rows to the slots of their ids (new ids take the next free slot in row order) and a
 * zero entry to every known slot the agent did not observe; `win` IS obs_history_output(), `single` receives
 * obs_single_history_output() (any strides: it is written straight into the episode container's history field).
 * Values are copied bit for bit.  err: 0 ok; |1 an ego id that was never registered (the reference raises ValueError);
 * |2 more than N vehicles observed by one agent (IndexError).  One wave per (thread, registered agent).            */
typedef struct {
    int32_t K, nA, N, L, d, obs_num;
    const float* obs;            /* [K, nA, obs_num, 1 + d]: id column first; an all-zero row is padding            */
    const int32_t* agent_ids;    /* [K, nA] ego ids in order of first appearance (unused entries: INT32_MIN)         */
    int32_t* slot_id;            /* [K, nA, N]                                                                       */
    int32_t* n_slots;            /* [K, nA]                                                                          */
    float* win;                  /* [K, nA, N, L, d]                                                                 */
    float* single;               /* optional: element (k, i, s, c) at k * single_s_k + i * single_s_a + s * d + c    */
    int64_t single_s_k, single_s_a;
    int32_t* err;                /* [1] device int, OR-ed                                                            */
} IplanObsHistArgs;
int iplan_obs_history_step(const IplanObsHistArgs* args, iplan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IPLAN_HIP_H */
