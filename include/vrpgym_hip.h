/*
 * vrpgym_hip.h — C ABI of libvrpgym_hip.so (gfx950 / MI355X).
 *
 * The reference (kevin-schumann/VRP-GYM) has no FFI: its hot path is Python
 * (numpy/networkx environments + torch modules).  This header is the boundary a
 * maintainer would bind with ctypes from the reference's own classes; every entry
 * point names the reference code it replaces (paths relative to the reference
 * repository root).  The binding shipped here is vrp-gym_amd/vrpgym_hip/_lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - no ownership is transferred, nothing is allocated or freed by the library;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never
 *     synchronises, and is safe to capture into a hipGraph;
 *   - return value 0 = ok, non-zero = error; vrp_last_error() returns a
 *     thread-local description of the last failure;
 *   - process-wide state: none that results depend on (per device, under a lock: the measured
 *     residency of the persistent kernel, the stream of its last launch, a pinned counter of
 *     episodes that fell back, and the mapping of the cross-process lease word);
 *   - kind: 0 = TSP, 1 = VRP, 2 = IRP  (gym_vrp/envs/{tsp,vrp,irp}.py).
 */
#ifndef VRPGYM_HIP_H
#define VRPGYM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this header's struct layouts and entry points: vrp_abi_version() of a matching
 * library returns it; the shipped binding (vrpgym_hip/_lib.py: ABI_VERSION) refuses any other. */
#define VRP_ABI_VERSION 8

#define VRP_KIND_TSP 0
#define VRP_KIND_VRP 1
#define VRP_KIND_IRP 2

#define VRP_EMB 128      /* embedding width the kernels are built for            */
#define VRP_HEADS 8      /* decoder heads (agents/graph_tsp_agent.py:53-55)      */
#define VRP_MAX_NODES 128
#define VRP_MAX_LAYERS 16 /* encoder attention layers (graph_encoder.py:30-39 loops num_attention_layers freely) */

/* ---- environment state (replaces the numpy attributes of TSPEnv/IRPEnv) ------ */
#define VRP_ENV_RESET_ON_ROLLOUT 1 /* vrp_env.flags: vrp_rollout starts the episode itself --
                                      visited := 0, current_location := depots, load := 1, the
                                      state part of TSPEnv.reset (tsp.py:150-160,172-174) -- inside
                                      its set-up kernel instead of expecting a vrp_env_reset launch */
typedef struct vrp_env {
  int32_t kind, B, N, flags;
  const double *pos;     /* (B,N,2) fp64 coordinates   vrp_graph.py:28-31          */
  const double *demand;  /* (B,N)   fp64, IRP only     vrp_graph.py:41-43          */
  const int32_t *depot;  /* (B)                        vrp_graph.py:34             */
  uint8_t *visited;      /* (B,N)  TSPEnv.visited      tsp.py:86,131-148           */
  uint8_t *mask;         /* (2,B,N) ping-pong copies of the state's mask column;   */
                         /*   buffer `p` is read by decode step with parity p and  */
                         /*   buffer p^1 written by the env step that follows it   */
  int32_t *cur;          /* (B)    TSPEnv.current_location                         */
  double *load;          /* (B)    IRPEnv.load (fp64), unused otherwise            */
} vrp_env;

/* E1  The state part of TSPEnv.reset / generate_graphs (tsp.py:150-160,172-174; irp.py:47,184)
 * for instances already in place: visited := 0, mask buffers := 0, current_location :=
 * depots, load := 1 (when env->load is set). */
int vrp_env_reset(const vrp_env *env, void *stream);

/* E7/E8  TSPEnv.generate_mask tsp.py:131-148 / vrp.py:13-37 / irp.py:126-155.
 * Applies the depot fix-ups to `visited` in place and writes mask buffer `parity`. */
int vrp_env_mask(const vrp_env *env, int parity, void *stream);

/* E4-E8  TSPEnv.step tsp.py:60-101 / IRPEnv.step irp.py:49-99 for host-driven callers
 * (RandomAgent, user code).  actions (B) int64.  Writes reward_f64 (B) = -distance,
 * *notdone := 1 if any graph's visited row is not all ones (evaluated before the
 * fix-ups, tsp.py:95); the caller zeroes *notdone first; done <=> *notdone == 0.
 * Writes mask buffer `parity_out`. */
int vrp_env_step(const vrp_env *env, const int64_t *actions, int parity_out,
                 double *reward_f64, int32_t *notdone, void *stream);

/* E3  TSPEnv.get_state tsp.py:106-129 / IRPEnv.get_state irp.py:101-124 as fp32
 * network inputs: x (B,N,3) = [x, y, demand-or-0], is_depot (B,N) u8. */
int vrp_env_features(const vrp_env *env, float *x, uint8_t *is_depot, void *stream);

/* ---- policy weights (views of the torch parameters; same names as state_dict) -- */
typedef struct vrp_encoder_layer {
  const float *in_proj_weight, *in_proj_bias;    /* (384,128) (384) */
  const float *out_proj_weight, *out_proj_bias;  /* (128,128) (128) */
  const float *bn1_weight, *bn1_bias;
  float *bn1_running_mean, *bn1_running_var;
  int64_t *bn1_num_batches_tracked;
  const float *ff0_weight, *ff0_bias;            /* (512,128) (512) */
  const float *ff2_weight, *ff2_bias;            /* (128,512) (128) */
  const float *bn2_weight, *bn2_bias;
  float *bn2_running_mean, *bn2_running_var;
  int64_t *bn2_num_batches_tracked;
} vrp_encoder_layer;

typedef struct vrp_encoder_weights {
  int32_t node_dim, depot_dim, hidden, num_layers;     /* 2|3, 2|0, 512, <= VRP_MAX_LAYERS */
  int32_t heads, reserved_;  /* encoder heads: 8 (0 = 8; every fused kernel), or 4 / 16 (head width
                              * 32 / 8: plain GEMM + per-(graph, head) VALU attention kernels, forward
                              * and backward; graph_encoder.py:170-172) */
  const float *node_embed_weight, *node_embed_bias;    /* (128,node_dim) */
  const float *depot_embed_weight, *depot_embed_bias;  /* (128,2) or NULL */
  vrp_encoder_layer layer[VRP_MAX_LAYERS];
  const void *split;   /* vrp_encoder_prepare's output for THESE weights, or NULL.  With it the
                        * eval-mode kernels run their dense products on the bf16 matrix cores (each
                        * fp32 operand as three bf16 planes, six MFMAs per product: fp32 accuracy at
                        * 2.7x the fp32 MFMA rate, csrc/encoder_x3.h); without it, on the fp32 MFMA. */
} vrp_encoder_weights;

typedef struct vrp_decoder_weights {
  const float *first_node, *last_node;            /* (128) placeholders       */
  const float *q_proj_weight;                     /* (384,384)                */
  const float *k_proj_weight, *v_proj_weight;     /* (384,128)                */
  const float *in_proj_bias;                      /* (1152) = bq|bk|bv        */
  const float *out_proj_weight, *out_proj_bias;   /* (384,384) (384)          */
  const float *kp_weight;                         /* (128,128)                */
  const float *att_output_weight;                 /* (128,384)                */
  const float *context_proj_weight;               /* (384,257) IRP only       */
} vrp_decoder_weights;

/* Scratch sizes (bytes) the caller must provide; pure functions of the shape. */
int64_t vrp_encoder_workspace_bytes(int B, int N, int hidden);
int64_t vrp_decoder_workspace_bytes(int kind, int B, int N);
int64_t vrp_decoder_derived_bytes(void);

/* Splits the dense weights of every layer (in_proj, out_proj, ff.0, ff.2) into three bf16 planes
 * in MFMA fragment order for vrp_encoder_weights.split (vrp_encoder_split_bytes bytes; hidden a
 * multiple of 128).  Re-run whenever those parameters change (the shipped binding compares the
 * tensors' version counters before every launch); a stale buffer means stale weights. */
int64_t vrp_encoder_split_bytes(int hidden, int num_layers);
int vrp_encoder_prepare(const vrp_encoder_weights *w, void *split, void *stream);

/* N1-N3  GraphEncoder.forward agents/graph_encoder.py:41-58, GraphDemandEncoder
 * .forward :95-138, MultiHeadAttentionLayer.forward :183-198, BatchNorm :141-154.
 * x (B,N,3) fp32 [x,y,demand], depot_mask (B,N) u8 or NULL (TSP), emb (B,N,128).
 * train != 0: batch statistics + running-stat update (momentum 0.1). */
int vrp_encoder_forward(const vrp_encoder_weights *w, int train, int B, int N,
                        const float *x, const uint8_t *depot_mask, float *emb,
                        void *workspace, void *stream);

/* D1  Folds the decoder parameters into the matrices the step kernel consumes
 * (agents/graph_decoder.py:29-44 parameters; algebra in DESIGN.md).  Must be
 * re-run whenever the parameters change. */
int vrp_decoder_prepare(int kind, const vrp_decoder_weights *w, void *derived,
                        void *stream);

/* Per-episode decoder state + outputs. */
typedef struct vrp_rollout_io {
  float *acc_loss;        /* (B) fp32 sum of -distance, step order  graph_tsp_agent.py:85 */
  float *acc_logp;        /* (B) fp32 sum of log-prob               graph_tsp_agent.py:86 */
  int32_t *notdone;       /* (max_steps+1) per-step flag "some graph is unfinished";     */
                          /*   notdone[t]==0 <=> env.step t returned done (tsp.py:95)    */
  int64_t *actions;       /* (max_steps,B) chosen nodes, or NULL                         */
  const int64_t *forced;  /* (max_steps,B) teacher-forced actions, or NULL               */
  const float *noise;     /* (max_steps,B,N) Exp(1) noise for sampling (parity runs: drawn   */
                          /*   on the host like the reference), or NULL                      */
  float *logits;          /* (max_steps,B,N) masked logits u, or NULL (debug/tests)      */
  float *step_logp;       /* (max_steps,B) per-step log-prob, or NULL                    */
  uint8_t *mask_trace;    /* (max_steps,B,N) the mask column step t decoded with, or NULL */
  float *load_trace;      /* (max_steps,B) IRP vehicle load fed to step t, or NULL        */
  uint64_t noise_seed;    /* sampling with noise == NULL: the Exp(1) noise is drawn inside    */
                          /*   the step kernels from a Philox stream keyed by this seed       */
                          /*   (counter = graph, node, step): throughput mode, no parity      */
  float logit_clip;       /* C of GraphDecoder.forward(..., C=10) (graph_decoder.py:56,97):   */
                          /*   u = C tanh(.); 0 = the reference's default 10.  The backward   */
                          /*   pass (vrp_decoder_backward) is built for 10 only               */
} vrp_rollout_io;

/* D2  Per-episode constants of GraphDecoder.forward (agents/graph_decoder.py:75-83):
 * graph embedding, hoisted K/V/_kp projections, the step-0 score row, the pointer-logit
 * table and (IRP) the last-node score table. */
int vrp_decode_prologue(int kind, const void *derived, int B, int N, const float *emb,
                        void *workspace, void *stream);

/* D2-D6 + E4-E8 fused: one GraphDecoder.forward call (agents/graph_decoder.py:51-115)
 * followed by env.step on its actions (tsp.py:60-101 / irp.py:49-99), i.e. one
 * iteration of the loop in TSPModel.forward agents/graph_tsp_agent.py:78-88.
 * `t` is the step index; the kernel is a no-op once notdone[t-1]==0.
 * flags: VRP_STEP_SAMPLE = sample (io->noise, or in-kernel noise from io->noise_seed when
 *        io->noise is NULL) instead of argmax;
 *        VRP_STEP_DECODE_ONLY = GraphDecoder.forward alone: no env.step, no
 *        accumulation (only env->{B,N,mask,load} are read; results go to
 *        io->actions / io->step_logp / io->logits);
 *        VRP_STEP_TILE_KERNEL = use the raw-embedding-tile kernel instead of the
 *        table-driven one;
 *        VRP_STEP_NO_FIRST_ROW = do not append vrp_decode_first_row to step 0;
 *        VRP_STEP_THROUGHPUT_KERNEL = large-batch variant of the table kernel at any B.
 * The kernel-selection flags (TILE / TABLE / THROUGHPUT) must be the same for every step of an
 * episode: which kernel takes step t is a schedule fixed by (kind, B, N, t, flags), and a step
 * kernel leaves behind what the NEXT step's kernel of that schedule reads (the latency-mode
 * score row). */
#define VRP_STEP_SAMPLE 1
#define VRP_STEP_DECODE_ONLY 2
#define VRP_STEP_TILE_KERNEL 4 /* use the raw-embedding-tile kernel (N <= 104) */
#define VRP_STEP_NO_FIRST_ROW 8 /* t == 0: the caller runs vrp_decode_first_row itself */
#define VRP_STEP_THROUGHPUT_KERNEL 16 /* B <= 2048: use the large-batch variant (4 graphs per
                                         workgroup, score row read by pointer chase) anyway */
#define VRP_STEP_TABLE_KERNEL 64 /* table-driven kernel at every step, also where the dispatch
                                    would give an episode's first steps to the raw-tile kernel
                                    (64 < N <= 104; 32 < N <= 40 at B > 2048) */
#define VRP_STEP_NO_PERSISTENT 32 /* vrp_rollout / vrp_rollout_steps[_range]: one launch per step
                                     even where the persistent multi-step kernel applies (B <= 2048,
                                     3 <= N <= 63, no logits trace, no teacher forcing) */
int vrp_decode_step(int kind, const void *derived, const vrp_decoder_weights *w,
                    const vrp_env *env, const float *emb, void *workspace,
                    const vrp_rollout_io *io, int t, int max_steps, int flags,
                    void *stream);

/* Single-wave workgroups of the persistent multi-step kernel the current device keeps resident
 * at once, MEASURED: the compute units a census kernel finds usable (a CU mask or a partition
 * mode shrinks them) x the largest per-CU count for which a census launch of the kernel itself
 * had every workgroup see all the others (starting from the occupancy query's answer).  vrp_rollout* use the
 * persistent kernel only for B <= this AND max_steps + 1 <= 2N (the hand-off words hold 2N step
 * rows), decided before the episode starts (otherwise one launch per step).  The first call on a
 * device synchronises the device and a private stream (call it once outside any stream capture,
 * with nothing in flight; vrpgym_hip.require_gpu does when it first sees the GPU).  A census that
 * comes out more than one workgroup per CU under the occupancy query was disturbed by other work
 * and is not cached: the next call measures again. */
int vrp_persistent_capacity(void);

/* The persistent launch is an optimisation that cannot fail an episode: a wave waits at most
 * 20 ms for another graph's mask word (VRP_PERSISTENT_SPIN_MS overrides, tests); if one gives up
 * -- the grid was not resident, e.g. another process' kernels held the compute units -- the
 * whole grid leaves within that time and the one-workgroup kernel launched behind it reruns the
 * steps from the saved start state with the per-step kernel's arithmetic: same actions, same
 * accumulators, bit for bit, no host round trip.  This call returns how many episodes of the
 * current device took that route in this process (after one, the library stays off the persistent
 * path for 100 ms, doubling up to 10 s while they keep coming).  Processes that share a device
 * additionally take turns through a lease word in /dev/shm (500 ms; VRP_PERSISTENT_LEASE=0
 * disables): a process that does not hold it runs one launch per step. */
int vrp_persistent_failures(void);

/* Name of the kernel vrp_decode_step launches for this shape and these flags (what a
 * rocprofv3 kernel trace will show). */
const char *vrp_step_kernel_name(int kind, int B, int N, int flags);

/* D2  GraphDecoder.forward's `first_` update (agents/graph_decoder.py:111-113) for
 * TSP/VRP: after step 0 the first chosen node is fixed; its part of the glimpse query is
 * folded, together with the graph-embedding part, into the last-node score table that every
 * later step reads one row of (the table is built here).  Appended to step 0 by
 * vrp_decode_step unless VRP_STEP_NO_FIRST_ROW is set.  No-op for IRP, whose context has no
 * first-node term (vrp_decode_prologue builds its table).  (B <= 1024, N <= 63: one launch,
 * first_base_kernel, from the glimpse keys the prologue kept; vrp_rollout's persistent decode
 * grid computes the same rows itself and skips this call.) */
int vrp_decode_first_row(int kind, const void *derived, int B, int N, const float *emb,
                         void *workspace, void *stream);

/* R1  TSPModel/VRPModel/IRPModel.forward (agents/graph_tsp_agent.py:61-92,
 * graph_vrp_agent.py:52-83, graph_irp_agent.py:54-105): mask init, features,
 * encoder, prologue and max_steps decode+env steps, all on `stream`.
 * emb (B,N,128) receives the node embeddings.  max_steps >= 2(N-1) (N-1 for TSP).
 * `sample` carries the step flags VRP_STEP_SAMPLE | VRP_STEP_TILE_KERNEL |
 * VRP_STEP_THROUGHPUT_KERNEL | VRP_STEP_TABLE_KERNEL | VRP_STEP_NO_PERSISTENT. */
int vrp_rollout(int kind, const vrp_encoder_weights *ew, const vrp_decoder_weights *dw,
                void *derived, const vrp_env *env, int train, int sample,
                float *emb, void *enc_workspace, void *dec_workspace,
                const vrp_rollout_io *io, int max_steps, void *stream);

/* Only the encoder phase of vrp_rollout: mask init, features, embedding + encoder layers (for
 * small eval-mode batches also the decoder's per-graph constants), zeroed accumulators --
 * exported so a benchmark can bracket exactly these launches with one event pair. */
int vrp_rollout_encode(int kind, const vrp_encoder_weights *ew, void *derived,
                       const vrp_env *env, int train, float *emb, void *enc_workspace,
                       void *dec_workspace, const vrp_rollout_io *io, int max_steps, void *stream);

/* Name of the kernel(s) the encoder phase launches for this shape (profiles, bench line). */
const char *vrp_encoder_kernel_name(const vrp_encoder_weights *w, int train, int B, int N);

/* Only the T-step decode+env loop of vrp_rollout (emb and prologue already done).
 * Host-side behaviour of the loop (vrp_rollout, vrp_rollout_steps, and a _range call that runs
 * to max_steps): where it issues one launch per step, a VRP / IRP call paces itself by the
 * device's done flag from step N - 1 on -- a chunk of eight launches, then hipEventSynchronize on
 * the chunk before the previous one, whose last step's flag came back through a pinned word --
 * and stops at the first finished chunk instead of queueing all 2 (N - 1) launches (a launch
 * behind the end leaves at its first instruction, but still costs its 3-4 us).  The call
 * therefore returns once the batch has (nearly) finished.  Not while the stream is being
 * captured into a hipGraph, not with VRP_NO_THROTTLE set (read per call), never for TSP
 * (always N - 1 steps) and never where the steps run as one persistent launch. */
int vrp_rollout_steps(int kind, const void *derived, const vrp_decoder_weights *dw,
                      const vrp_env *env, const float *emb, void *dec_workspace,
                      const vrp_rollout_io *io, int max_steps, int flags, void *stream);

/* Steps t_begin..t_end-1 of that loop, issued back to back from C (the loop vrp_rollout and
 * vrp_rollout_steps run; exported so a benchmark can bracket exactly these launches with
 * one event pair).  Accumulators are NOT reset.  flags as for vrp_decode_step. */
int vrp_rollout_steps_range(int kind, const void *derived, const vrp_decoder_weights *dw,
                            const vrp_env *env, const float *emb, void *dec_workspace,
                            const vrp_rollout_io *io, int t_begin, int t_end, int max_steps,
                            int flags, void *stream);

/* E1  Host-side instance sampler, bit-exact with numpy's legacy global stream
 * (VRPGraph.__init__ gym_vrp/graph/vrp_graph.py:28-43 called B times,
 * vrp_network.py:41-42): rand(N,2) -> choice(N,1,replace=False) -> uniform(1,10)/C per
 * graph.  key_host (624 words) / *pos_host are numpy's MT19937 state from
 * np.random.get_state(), advanced in place; all pointers are HOST pointers. */
int vrp_draw_instances_host(uint32_t *key_host, int32_t *pos_host, int B, int N,
                            double *pos_out_host, int64_t *depots_host, double *demands_host);

/* The same stream advanced over all B graphs, storing only graphs [first, first+count) (outputs
 * sized for `count` graphs): what a rank of a sharded env (rank r owns rows
 * [rB/R, (r+1)B/R) of the seed-ordered stream) needs -- the rest is drawn and discarded natively. */
int vrp_draw_instances_host_range(uint32_t *key_host, int32_t *pos_host, int B, int N, int first,
                                  int count, double *pos_out_host, int64_t *depots_host,
                                  double *demands_host);

/* E1' Device-side instance sampler for throughput runs: the
 * reference's distributions (vrp_graph.py:28-43) from a counter-based Philox4x32-10
 * stream -- NOT the reference's numpy stream, so no instance-level parity.  Counter =
 * (first_graph + b, node, episode, draw): a rank's shard equals the same rows of the
 * unsharded batch.  pos (B,N,2) f64, depot (B) i32, demand (B,N) f64 (depot's = 0). */
int vrp_draw_instances_device(uint64_t seed, uint64_t episode, int first_graph, int B, int N,
                              double *pos, int32_t *depot, double *demand, void *stream);

/* R3' RandomAgent.forward (agents/random_agent.py:15-41) on the device for throughput runs:
 * max_steps x (uniform draw among the unmasked nodes from the Philox stream + env.step),
 * acc_loss (B) fp32 = sum of -distance, notdone (max_steps+1) per-step flags as in
 * vrp_rollout_io, actions (max_steps,B) or NULL.  Same distribution as the reference's
 * RandomAgent, not its numpy stream. */
int vrp_random_rollout(const vrp_env *env, uint64_t seed, uint64_t episode, int first_graph,
                       int max_steps, float *acc_loss, int32_t *notdone, int64_t *actions,
                       void *stream);

/* ---- backward pass (K4): building blocks, each unit-tested against torch autograd ---- */
/* C (N1,N2) (+)= X^T Y over R rows; deterministic split-K; slab_ws from *_workspace_bytes. */
int64_t vrp_gemm_tn_workspace_bytes(int R, int N1, int N2);
int vrp_gemm_tn(const float *X, int ldx, const float *Y, int ldy, float *C, int R, int N1, int N2,
                int accumulate, void *slab_ws, void *stream);
/* out (N) (+)= column sums of Y (R,N); deterministic two-stage reduction. */
int64_t vrp_colsum_workspace_bytes(int R, int N);
int vrp_colsum(const float *Y, int ldy, int R, int N, float *out, int accumulate, void *ws,
               void *stream);
/* BatchNorm1d(128) backward with batch statistics (agents/graph_encoder.py:141-154 in
 * train mode): stats = [mean | invstd], z = pre-normalisation input. */
int64_t vrp_bn_bwd_workspace_bytes(void);
int vrp_bn_bwd(const float *dy, const float *z, const float *stats, const float *gamma, int R,
               float *dz, float *dgamma, float *dbeta, int accumulate, void *ws, void *stream);
/* Encoder self-attention backward per (graph, head) (graph_encoder.py:170-172,196):
 * qkv (B*N,384) forward projections, dO (B*N,128) -> dqkv (B*N,384). */
int vrp_attention_bwd(const float *qkv, const float *dO, float *dqkv, int B, int N, void *stream);

/* Parameter-gradient views (same names as the state_dict; every buffer is overwritten). */
typedef struct vrp_encoder_layer_grads {
  float *in_proj_weight, *in_proj_bias, *out_proj_weight, *out_proj_bias;
  float *bn1_weight, *bn1_bias, *ff0_weight, *ff0_bias, *ff2_weight, *ff2_bias;
  float *bn2_weight, *bn2_bias;
} vrp_encoder_layer_grads;
typedef struct vrp_encoder_grads {
  float *node_embed_weight, *node_embed_bias, *depot_embed_weight, *depot_embed_bias;
  vrp_encoder_layer_grads layer[VRP_MAX_LAYERS];
} vrp_encoder_grads;

/* N1-N3 in train mode with every intermediate kept on a tape (for the backward pass):
 * same arithmetic as vrp_encoder_forward(train=1); update_running = 0 leaves the
 * BatchNorm running statistics untouched (a recompute of an already counted pass). */
int64_t vrp_encoder_tape_bytes(int B, int N, int hidden, int num_layers);
int vrp_encoder_forward_tape(const vrp_encoder_weights *w, int B, int N, const float *x,
                             const uint8_t *depot_mask, float *emb, void *tape,
                             int update_running, void *stream);
/* Backward of the encoder: d_emb (B,N,128) -> gradients of every encoder parameter. */
int64_t vrp_encoder_backward_workspace_bytes(int B, int N, int hidden);
int vrp_encoder_backward(const vrp_encoder_weights *w, const vrp_encoder_grads *g, int B, int N,
                         const float *x, const uint8_t *depot_mask, const void *tape,
                         const float *d_emb, void *workspace, void *stream);

/* Decoder parameter gradients (same names as vrp_decoder_weights; all overwritten). */
typedef struct vrp_decoder_grads {
  float *first_node, *last_node;
  float *q_proj_weight, *k_proj_weight, *v_proj_weight, *in_proj_bias;
  float *out_proj_weight, *out_proj_bias, *kp_weight, *att_output_weight;
  float *context_proj_weight;                     /* (384,257) IRP only, else NULL */
} vrp_decoder_grads;

/* Backward of T recorded decoder steps (GraphDecoder.forward agents/graph_decoder.py:51-115
 * applied T times by TSPModel.forward graph_tsp_agent.py:78-88): gradient of
 * sum_b d_logp[b] * sum_t log p(actions[t,b]) w.r.t. the decoder parameters and the node
 * embeddings.  actions (T,B), masks (T,B,N) = io.mask_trace, loads (T,B) = io.load_trace
 * (IRP, else NULL).  d_emb (B,N,128) is overwritten; step_logp (T,B) optionally receives
 * the recomputed per-step log-probabilities. */
int64_t vrp_decoder_backward_workspace_bytes(int kind, int B, int N, int T);
int vrp_decoder_backward(int kind, const vrp_decoder_weights *w, const vrp_decoder_grads *g,
                         int B, int N, int T, const float *emb, const int64_t *actions,
                         const uint8_t *masks, const float *loads, const float *d_logp,
                         float *d_emb, float *step_logp, void *workspace, void *stream);

/* Building blocks exported for tests and profiling. */
int vrp_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                const float *residual, int ldr, float *C, int ldc, int M, int N, int K,
                int relu, void *stream);

/* C = (A W^T + residual) with entries zeroed where gate <= 0 (ReLU backward). */
int vrp_gemm_nt_gated(const float *A, int lda, const float *W, int ldw, const float *residual,
                      int ldr, const float *gate, float *C, int ldc, int M, int N, int K,
                      void *stream);

/* Test hook: the map from 32 random bits to the Exp(1) noise of the in-kernel sampler
 * (Categorical.sample's q, agents/graph_decoder.py:104-107, throughput mode): out[i] in (0, inf)
 * for every input word. */
int vrp_debug_exp1_from_bits(const uint32_t *bits, float *out, int n, void *stream);

/* First 16 hex digits of the sha256 over the kernel sources (every .hip and .h under csrc/, this header) the
 * library was built from: ties a measurement kept under profiles/ to the build it was taken on. */
const char *vrp_source_hash(void);

const char *vrp_last_error(void);
int vrp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VRPGYM_HIP_H */
