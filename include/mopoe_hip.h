/*
 * mopoe_hip.h -- C ABI of libmopoe_hip.so: the MI355X (gfx950) MoPoE-VAE
 * training hot path.
 *
 * The reference (neurospin-projects/2022_cambroise_interpret_multivae) is pure
 * Python and has no FFI of its own; the entry points below are what a ctypes
 * binding for its hot path binds (INTEGRATION.md shows the stub).  Each one
 * names the reference interface it replaces (paths relative to the
 * reference's experiments/ directory).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float32 / int32 unless it is the
 *     descriptor struct itself (host memory, read during the call only);
 *   - all matrices are row-major; weights are (out, in) as in torch.nn.Linear;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); the
 *     library never synchronises, allocates or frees device memory, so every
 *     call is legal inside hipStreamBeginCapture / torch.cuda.graph;
 *   - return 0 on success, MOPOE_ERR_ARG (-1) for a rejected descriptor,
 *     MOPOE_ERR_HIP (-2) for a HIP runtime error; mopoe_last_error() returns
 *     a thread-local message for the last non-zero return.
 */
#ifndef MOPOE_HIP_H
#define MOPOE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOPOE_ABI_VERSION 11
#define MOPOE_MAX_MODS 5      /* modalities                                   */
#define MOPOE_MAX_SUBSETS 31  /* 2^MAX_MODS - 1 non-empty subsets             */
#define MOPOE_MAX_JOBS 10     /* decoder passes: 1 joint + 1 unimodal per mod */
#define MOPOE_HIDDEN 256      /* networks/networks.py:14,50 (hard-coded)      */
#define MOPOE_MAX_RANKS 8     /* GPUs of one node (xGMI full mesh)            */
#define MOPOE_IPC_HANDLE_BYTES 80 /* hipIpcMemHandle_t (64) + the device's UUID (16): ABI 11 */
#define MOPOE_RCCL_ID_BYTES 128 /* ncclUniqueId                                  */
#define MOPOE_ROWS 16         /* batch rows per MFMA tile; a row group has <= 16 */
#define MOPOE_MAX_LAYERS 4    /* hidden layers of an encoder / a decoder (general topology) */

#define MOPOE_ERR_ARG (-1)
#define MOPOE_ERR_HIP (-2)

/* sub_kind: how a subset's (mu, logvar) is formed
 * (utils/BaseMMVae.py:43-61,96-122) */
#define MOPOE_SUB_POE 0        /* product of experts, no prior expert          */
#define MOPOE_SUB_POE_PRIOR 1  /* product of experts + N(0,1) prior expert     */
#define MOPOE_SUB_SLICES 2     /* moe_fusion: contiguous row slices of members */

/* joint_mode: BaseMMVae.inference (utils/BaseMMVae.py:226-231) */
#define MOPOE_OPERANDS_F32 0
#define MOPOE_OPERANDS_BF16 1
#define MOPOE_LIK_NORMAL 0     /* torch.distributions.Normal(loc, scale)       */
#define MOPOE_LIK_LAPLACE 1    /* torch.distributions.Laplace(loc, scale)      */
#define MOPOE_JOINT_MIXTURE 0  /* sample=True: mixture_component_selection     */
#define MOPOE_JOINT_MEAN 1     /* sample=False: mean of mus and of logvars     */
#define MOPOE_JOINT_EXPERT 2   /* use_expert=<subset key>                      */

/* indices into the stats buffer written by every step (float32) */
#define MOPOE_STAT_TOTAL_LOSS 0
#define MOPOE_STAT_JOINT_DIV 1
#define MOPOE_STAT_KLD_SUBSET 2                        /* + subset index       */
#define MOPOE_STAT_KLD_STYLE (2 + MOPOE_MAX_SUBSETS)   /* + modality index     */
#define MOPOE_STAT_NLL (MOPOE_STAT_KLD_STYLE + MOPOE_MAX_MODS) /* + job index  */
/* utils/TBLogger.py:26-37 (write_latent_distr): mean over the batch and the latent
 * dimensions of every encoder output, + 4 * modality + {0 style mu, 1 style logvar,
 * 2 content mu, 3 content logvar} */
#define MOPOE_STAT_LATENT_MEAN (MOPOE_STAT_NLL + MOPOE_MAX_JOBS)
#define MOPOE_NUM_STATS (MOPOE_STAT_LATENT_MEAN + 4 * MOPOE_MAX_MODS)

/* the int32 `counters` buffer (device), zeroed once by the caller */
#define MOPOE_NUM_COUNTERS 64
#define MOPOE_CTR_STEPS_BEGUN 0   /* training steps begun                            */
#define MOPOE_CTR_STEPS_DONE 1    /* training steps whose last kernel has finished   */
#define MOPOE_CTR_INVALID 2       /* != 0: a step could not be completed (a hand-off
                                     inside the fused launch or a wait of the gradient
                                     exchange timed out, or the ranks' batches held
                                     different modalities).  STICKY: while it is
                                     non-zero no Adam update is applied; the caller
                                     reads it (or its mirror in status_host), raises,
                                     and re-zeroes `partials` and this word to go on */
#define MOPOE_CTR_FIRST_INVALID 3 /* training step (value of MOPOE_CTR_STEPS_BEGUN) at which
                                     MOPOE_CTR_INVALID went up, 0 while it is down: the
                                     steps from this one on were NOT applied -- the caller
                                     re-arms (zeroes both words and `partials`) and runs
                                     those batches again (run_epochs.py:180-182 applies a
                                     step or nothing; ABI 9)                            */
#define MOPOE_CTR_ADAM_STEPS 4    /* + modality: Adam updates applied to that
                                     modality's parameters (torch.optim.Adam keeps
                                     state['step'] per parameter and skips parameters
                                     whose .grad is None: an encoder / decoder whose
                                     modality sat out k batches is k steps behind)   */
#define MOPOE_CTR_TICKET 9        /* last-block detection of the Adam kernel (0)     */
#define MOPOE_CTR_BIAS 16         /* 2 slots (step parity) x MAX_MODS records of 4
                                     words: the bias corrections of the step          */

/* ---------------------------------------------------------------------------
 * Model description: replaces the flags the reference's Encoder / Decoder /
 * BaseMMVae constructors read (multimodal_cohort/networks/networks.py:9-28,
 * 44-64; utils/BaseMMVae.py:17-34).  One hidden encoder layer of 256 units
 * with ReLU, no hidden decoder layer (workflow.py:41-49 defaults).
 *
 * Flat parameter buffer layout (float32, every segment 64-float aligned), per
 * modality m:
 *   w1  (256, d_m)      encoders.<m>.shared_encoder.0.weight
 *   b1  (256)           encoders.<m>.shared_encoder.0.bias
 *   wh  (nh_m, 256)     rows: style_mu | style_logvar | class_mu | class_logvar
 *   bh  (nh_m)          same order;  nh_m = 2*style_dim[m] + 2*class_dim
 *   wd  (d_m, zd_m)     decoders.<m>.out_mu.weight, zd_m = style_dim[m]+class_dim
 *   bd  (d_m)           decoders.<m>.out_mu.bias
 *   lvo (d_m)           decoders.<m>.logvar  (1, d_m)
 * followed by 64 control words (off_ctrl) that are no parameters.
 * ------------------------------------------------------------------------- */
typedef struct mopoe_model {
    int32_t num_mods;
    int32_t class_dim;
    int32_t input_dim[MOPOE_MAX_MODS];
    int32_t style_dim[MOPOE_MAX_MODS]; /* 0 = no style branch                  */
    int32_t learn_output_scale;        /* decoders.<m>.logvar trainable        */
    /* filled by mopoe_model_layout(): offsets in floats into the flat buffer  */
    int32_t off_w1[MOPOE_MAX_MODS];
    int32_t off_b1[MOPOE_MAX_MODS];
    int32_t off_wh[MOPOE_MAX_MODS];
    int32_t off_bh[MOPOE_MAX_MODS];
    int32_t off_wd[MOPOE_MAX_MODS];
    int32_t off_bd[MOPOE_MAX_MODS];
    int32_t off_lvo[MOPOE_MAX_MODS];
    int32_t off_ctrl;                  /* 64 control words behind the last segment: in
                                          buf->grads, word m = 1.0 if modality m was in
                                          the step's batch, word MOPOE_MAX_MODS = 1.0 if
                                          this rank's backward could not be completed
                                          (MOPOE_CTR_INVALID) -- summed by the ranks'
                                          all-reduce, checked by the Adam kernel: no rank
                                          applies a step that one rank could not finish  */
    int32_t num_floats;                /* length of the flat buffer            */
} mopoe_model;

/* Fills the off_* fields and num_floats from the dims. */
int mopoe_model_layout(mopoe_model* model);

/* ---------------------------------------------------------------------------
 * Step description: what BaseMMVae.inference / forward and
 * run_epochs.basic_routine_epoch derive from the batch and the flags
 * (utils/BaseMMVae.py:137-239, run_epochs.py:73-135, utils/utils.py:58-112).
 * Built on the host by the Python side (plan.py); all float32 host arithmetic
 * the reference performs (weights 1/K, floor(N*w) slice sizes) is done there
 * and handed over as numbers.
 * ------------------------------------------------------------------------- */
typedef struct mopoe_step {
    int32_t n;              /* batch rows                                      */
    int32_t present_mask;   /* bit m: modality m is in the batch               */
    int32_t sample;         /* sample_latents                                  */
    int32_t joint_mode;     /* MOPOE_JOINT_*                                   */
    int32_t expert_subset;  /* subset index for MOPOE_JOINT_EXPERT             */
    int32_t backward;       /* also compute gradients (training step)          */
    int32_t rows_per_group; /* 0 = the library picks: 16 batch rows per workgroup of the
                               fused per-sample kernel when its tiles fit the LDS,
                               else 8, 4, 2 or 1 (wide inputs, five modalities);
                               1, 2, 4, 8 or 16 pins it                          */
    int32_t group_rows;     /* 0, or rows per logical batch: the n rows are
                               n/group_rows independent batches folded into the
                               batch axis (the repeated forwards of
                               workflow.py:388-419 as one launch); the
                               row-position rules (mixture slices, utils.py:63-85;
                               moe slices) apply to row mod group_rows, and
                               comp_f / sub_f are sized for group_rows          */

    /* non-empty subsets in BaseExperiment.set_subsets order
     * (utils/BaseExperiment.py:58-79) */
    int32_t num_subsets;
    int32_t sub_mask[MOPOE_MAX_SUBSETS];   /* member bitmask                   */
    int32_t sub_avail[MOPOE_MAX_SUBSETS];  /* all members present              */
    int32_t sub_kind[MOPOE_MAX_SUBSETS];   /* MOPOE_SUB_*                      */
    int32_t sub_members[MOPOE_MAX_SUBSETS][MOPOE_MAX_MODS]; /* sorted-name order */
    int32_t sub_f[MOPOE_MAX_SUBSETS];      /* SLICES: rows per member slice    */
    float sub_kl_coef[MOPOE_MAX_SUBSETS];  /* d loss / d KL(subset)            */

    /* mixture components = subsets passing fusion_condition
     * (utils/BaseMMVae.py:125-134,213-227); comp_f = int(floor(N*w_0)) */
    int32_t num_comp;
    int32_t comp_sub[MOPOE_MAX_SUBSETS];
    int32_t comp_f;
    float comp_w[MOPOE_MAX_SUBSETS];       /* reweighted weights (float32)     */

    float style_kl_coef[MOPOE_MAX_MODS];   /* d loss / d KL(style_m)           */

    /* decoder jobs: job 0.. = joint pass (one per present modality); method
     * poe adds one unimodal job per present modality (run_epochs.py:104-128) */
    int32_t num_jobs;
    int32_t job_mod[MOPOE_MAX_JOBS];
    int32_t job_slot[MOPOE_MAX_JOBS];      /* row block inside z/loc/g_xhat    */
    int32_t job_src[MOPOE_MAX_JOBS];        /* -1: joint latent, else subset    */
    int32_t job_stream[MOPOE_MAX_JOBS];    /* pass id, non-decreasing; jobs of
                                              a pass share the content eps    */
    float job_nll_coef[MOPOE_MAX_JOBS];    /* d loss / d nll(job)              */
    int32_t likelihood;                    /* MOPOE_LIK_*: what the decoders' (loc, scale)
                                              parameterises (modalities/modality.py:18-30,
                                              42-45); ABI 8 -- the word was padding before   */

    /* noise: injected eps (parity runs) or NULL -> on-device Philox4x32-10   */
    const float* job_eps_content[MOPOE_MAX_JOBS]; /* (n, class_dim)            */
    const float* job_eps_style[MOPOE_MAX_JOBS];   /* (n, style_dim[m])         */
    uint64_t seed;
    int32_t gemm_operands;  /* MOPOE_OPERANDS_*: what the encoder layer's GEMM of a LARGE batch
                               (from 2,048 rows: k_linear_big) multiplies -- the float32 values
                               (0, default: the reference's arithmetic, exact-f32 MFMA) or their
                               bfloat16 roundings with float32 accumulation (1, opt-in: BASELINE
                               configs[1] names "bf16 compute / fp32 accumulate"; what it costs in
                               accuracy is measured in tests/test_oracle_precision.py).  ABI 10  */
    int32_t pad_;
} mopoe_step;

/* ---------------------------------------------------------------------------
 * Buffers.  Caller-owned (PyTorch caching allocator).  Row-major float32.
 *   R_m = (number of jobs of modality m) * n   rows for the per-job tensors.
 * ------------------------------------------------------------------------- */
typedef struct mopoe_buffers {
    float* params;                       /* (num_floats)                       */
    float* grads;                        /* (num_floats)  written by backward  */
    float* exp_avg;                      /* (num_floats)  Adam m               */
    float* exp_avg_sq;                   /* (num_floats)  Adam v               */
    int32_t* counters;                   /* (MOPOE_NUM_COUNTERS) MOPOE_CTR_*      */

    const float* x[MOPOE_MAX_MODS];      /* (x_rows[m], d_m) input, ld = d_m; nothing
                                            past x_rows[m] * d_m floats is touched  */
    const int32_t* row_index[MOPOE_MAX_MODS]; /* optional (n) per modality: row of
                                            x[m] that holds batch row i (a batch
                                            is then a gather out of cohort arrays
                                            resident in HBM); NULL = identity.  An
                                            index outside [0, x_rows[m]) reads as a
                                            row of zeros                          */
    int32_t x_rows[MOPOE_MAX_MODS];      /* rows of x[m]: required with row_index[m]
                                            (the cohort's rows), else 0 or n       */

    float* hidden[MOPOE_MAX_MODS];       /* (n, 256)     relu(x W1^T + b1)     */
    float* heads[MOPOE_MAX_MODS];        /* (n, nh_m)    encoder outputs       */
    float* subsets_mu;                   /* (num_subsets, n, D)                */
    float* subsets_logvar;               /* (num_subsets, n, D)                */
    float* joint_mu;                     /* (n, D)                             */
    float* joint_logvar;                 /* (n, D)                             */
    float* z[MOPOE_MAX_MODS];            /* (R_m, ldz_m) [style | content]     */
    float* loc[MOPOE_MAX_MODS];          /* (R_m, d_m)   decoder mean          */
    float* stats;                        /* (MOPOE_NUM_STATS)                  */
    float* stats_host;                   /* optional: pinned HOST memory mapped
                                            into the device; the step's scalars
                                            are also written there by the kernel
                                            itself (a log without a D2H copy on
                                            the stream); NULL to skip           */
    int32_t* status_host;                /* optional: pinned HOST memory (4 int32):
                                            the last kernel of every training step
                                            writes {steps done, MOPOE_CTR_INVALID,
                                            MOPOE_CTR_FIRST_INVALID, 0} there, so the
                                            caller can notice an invalid step without
                                            synchronising                          */

    float* g_xhat[MOPOE_MAX_MODS];       /* (R_m, d_m)   d loss / d loc        */
    float* g_heads[MOPOE_MAX_MODS];      /* (n, nh_m)                          */
    float* g_pre[MOPOE_MAX_MODS];        /* (n, 256)     d loss / d pre-relu   */
    float* wfrag;                        /* optional, mopoe_wfrag_floats(model) floats:
                                            fragment-major copies of the head and decoder
                                            weights, read by the four-row form of the
                                            fused launch (NULL: that form is not used).
                                            Every update this library applies keeps them
                                            in step with `params`; after ANY other write
                                            to `params` (initialisation, a checkpoint, a
                                            broadcast) call mopoe_wfrag_refresh before
                                            the next step                            */
    float* partials;                     /* (mopoe_row_groups(model, step),
                                            mopoe_partials_stride(model)); ZERO it
                                            once after allocating: one word of a
                                            slab is the row group's hand-off flag
                                            in the fused launch, left at zero by
                                            every call that completes             */
    float* wgrad_scratch;                /* optional, mopoe_wgrad_scratch_floats(model,
                                            step) floats: partial 64 x 64 blocks of the
                                            weight gradients of a LARGE batch (the batch
                                            axis is split over workgroups, a second launch
                                            adds the parts in a fixed order) and, behind
                                            them, 64 pre-summed slabs of `partials`
                                            (mopoe_forward: those slabs alone);
                                            NULL, or a step whose count is 0: the one-launch
                                            form                                          */
    int64_t wgrad_scratch_floats;        /* floats `wgrad_scratch` holds (ABI 11).  The count a
                                            step needs depends on n AND on which modalities
                                            the batch holds: a call whose step needs more than
                                            this returns MOPOE_ERR_ARG and launches nothing
                                            (it used to write past the buffer)            */
} mopoe_buffers;

typedef struct mopoe_adam {
    float lr, beta1, beta2, eps;         /* experiment.py:268-271              */
} mopoe_adam;

/* Per-kernel timing for bench.py's roofline figure.  While enabled, every
 * launch is bracketed by a hipEvent pair recorded on the launch stream (not
 * legal during stream capture).  mopoe_profile_read waits for the recorded
 * events, returns per-kernel launch counts and summed milliseconds
 * (arrays of MOPOE_NUM_KERNELS) and clears the log. */
#define MOPOE_KERNEL_LINEAR 0
#define MOPOE_KERNEL_LATENT 1
#define MOPOE_KERNEL_WGRAD 2
#define MOPOE_KERNEL_ADAM 3
#define MOPOE_KERNEL_FINALIZE 4
#define MOPOE_KERNEL_FUSED 5   /* encoder layer + per-sample chain in one launch */
#define MOPOE_KERNEL_XGMI 6    /* gradient exchange over xGMI peer windows            */
#define MOPOE_KERNEL_RCCL 7    /* RCCL's all-reduce (mopoe_rccl_*), as enqueued       */
#define MOPOE_NUM_KERNELS 8
int mopoe_profile_enable(int enable);
int mopoe_profile_read(int32_t* count, float* total_ms);

int mopoe_abi_version(void);
/* The library reads its environment knobs (MOPOE_NO_FUSE, MOPOE_QUAD, ... -- test and
 * experiment switches between launch forms that compute the same bits) once, when it is
 * loaded: a training step makes no getenv call.  This re-reads them (tests only). */
int mopoe_reload_knobs(void);
const char* mopoe_last_error(void);
/* sizeof / offsetof probes so a binding can verify its struct mirrors:
 * 0 mopoe_model, 1 mopoe_step, 2 mopoe_buffers, 3 mopoe_adam (sizes);
 * 4 step.job_eps_content, 5 step.comp_w, 6 buffers.partials,
 * 7 model.num_floats, 8 buffers.status_host, 9 model.off_ctrl, 10 buffers.wfrag (offsets);
 * 11 mopoe_topology, 12 mopoe_gbuffers (sizes), 13 gbuffers.keep_enc (offset); -1 otherwise. */
int mopoe_sizeof(int which);

/* leading dimension (floats) of z[m]: round_up(zd_m, 4) */
int mopoe_ldz(const mopoe_model* model, int mod);
/* floats of mopoe_buffers.wfrag */
int mopoe_wfrag_floats(const mopoe_model* model);
/* rebuild buffers.wfrag from buffers.params (see mopoe_buffers.wfrag) */
int mopoe_wfrag_refresh(const mopoe_model* model, const mopoe_buffers* buf, void* stream);
/* floats of mopoe_buffers.wgrad_scratch this step would use (0: a training batch
 * below the size from which the split weight-gradient launches pay; a forward-only
 * step -- step->backward == 0 -- of fewer than 512 row groups) */
int64_t mopoe_wgrad_scratch_floats(const mopoe_model* model, const mopoe_step* step);
/* floats per row group in `partials` (the scalar partial sums, the d loss / d logvar column
 * sums per decoder pass and -- since the general topologies' likelihood runs in the output
 * layer's epilogue -- a word per decoder pass and tile of 16 output columns; always size
 * `partials` by this call) */
int mopoe_partials_stride(const mopoe_model* model);
/* row groups the fused per-sample kernel cuts the batch into for this step
 * (= the number of partial slabs the caller provides): ceil(n / rows), rows = 16
 * unless the LDS carve-up asks for fewer, step->rows_per_group pins it, or the step
 * runs in four-row groups (training batches of 4..1024 rows, <= 2 modalities) */
int mopoe_row_groups(const mopoe_model* model, const mopoe_step* step);
/* bytes of LDS the fused latent kernel needs for this model and step (<= 160 KiB
 * after the rows-per-group fallback; larger only if even one row does not fit) */
int mopoe_latent_lds_bytes(const mopoe_model* model, const mopoe_step* step);

/* Replaces BaseMMVae.forward / inference under torch.no_grad()
 * (utils/BaseMMVae.py:137-239) plus the scalar terms of
 * run_epochs.basic_routine_epoch (run_epochs.py:73-135): encoder, subset
 * fusion, joint latent, reparameterisation, decoder, NLL and KL terms ->
 * hidden, heads, subsets_*, joint_*, z, loc, stats. */
int mopoe_forward(const mopoe_model* model, const mopoe_step* step,
                  const mopoe_buffers* buf, void* stream);

/* Replaces one iteration of run_epochs.train (run_epochs.py:158-184):
 * basic_routine_epoch + total_loss.backward() [+ optimizer.step() when
 * `adam` is non-NULL; pass NULL to stop after the gradients, e.g. to
 * all-reduce buf->grads across ranks before mopoe_adam_step]. */
int mopoe_train_step(const mopoe_model* model, const mopoe_step* step,
                     const mopoe_buffers* buf, const mopoe_adam* adam,
                     void* stream);

/* Replaces torch.optim.Adam.step (experiment.py:256-279) on the flat buffer,
 * restricted to the segments of the modalities in present_mask (parameters
 * whose .grad is None are skipped by torch; their step count does not advance).
 * The step number of each modality's parameters lives on the device
 * (counters[MOPOE_CTR_ADAM_STEPS + m]) -- a captured graph replays correctly.
 * `world` > 1: buf->grads holds the SUM over `world` data-parallel ranks (after an
 * all-reduce): the gradient is multiplied by 1/world first, and the ranks' batches
 * are checked to have held the same modalities (the control words mopoe_train_step
 * leaves behind the last segment of buf->grads travel through the same
 * all-reduce); if not, nothing is updated and MOPOE_CTR_INVALID is raised. */
int mopoe_adam_step(const mopoe_model* model, int32_t present_mask,
                    const mopoe_buffers* buf, const mopoe_adam* adam,
                    int32_t world, void* stream);

/* ---------------------------------------------------------------------------
 * Data-parallel replicas on one node (the reference has no
 * distributed code, so there is nothing to cite beyond the optimizer step the
 * exchange feeds, experiment.py:256-279).  A communicator owns one device
 * "window" per rank (uncached device memory: two inboxes of `world` gradient
 * buffers + arrival flags), exported to the other ranks' processes with
 * hipIpcGetMemHandle; this is the one object of the library that allocates.
 *
 *   mopoe_comm_create   allocates the window and returns MOPOE_IPC_HANDLE_BYTES: its
 *                       64-byte IPC handle followed by the 16-byte UUID of the rank's
 *                       device; the caller exchanges the records out of band (e.g.
 *                       torch.distributed.all_gather_object) ...
 *   mopoe_comm_connect  ... and passes all `world` of them (rank order).  A peer whose
 *                       UUID is this rank's own shares the device (ranks time-slicing one
 *                       GPU: the one-GPU rehearsal of the node): other processes' grids on
 *                       the same compute units break the residency the fused launch's
 *                       in-kernel hand-off needs, so while such a communicator exists every
 *                       step of the process runs in separate launches (same bits).
 *   mopoe_comm_allreduce_adam
 *                       replaces `all_reduce(grads); mopoe_adam_step(1/world)`:
 *                       one launch pushes buf->grads to every peer over its xGMI
 *                       link, waits (bounded by timeout_ms) for the peers' pushes,
 *                       sums the copies in rank order (bit-identical on all ranks)
 *                       and leaves the sum in buf->grads; the Adam launch behind it
 *                       applies the mean -- or NOTHING when any block's exchange was
 *                       not good (a wait out of budget, a peer with other
 *                       modalities): MOPOE_CTR_INVALID is raised and parameters and
 *                       moments stay at the last complete step.  A time-out is seen
 *                       by the rank that waited, not necessarily by its peers: after
 *                       one, re-arm AND re-broadcast params / exp_avg / exp_avg_sq /
 *                       counters from one rank (every rank holds a whole step, the
 *                       ranks may be one step apart).  All ranks must call it the
 *                       same number of times.
 *   mopoe_comm_train_step
 *                       replaces `mopoe_train_step(adam = NULL); all_reduce(grads);
 *                       mopoe_adam_step(1/world)`: mopoe_train_step whose
 *                       weight-gradient launch exchanges every 32x32 gradient block
 *                       with the peers (push, flag, wait, rank-ordered sum) and
 *                       leaves the sum in buf->grads, followed by the Adam launch as
 *                       above (the whole step or none of it).  All ranks must step on
 *                       batches with the SAME present_mask and the same batch-size
 *                       class (n <= 512 or not): the blocks of the launch are matched
 *                       by index.
 *   mopoe_comm_allreduce
 *                       the same exchange without the update: data (num_floats) is
 *                       replaced by the rank-ordered sum.
 *   mopoe_comm_status   waits for nothing: copies the count of timed-out waits
 *                       (0 = every exchange so far was complete).
 *   mopoe_comm_destroy  unmaps and frees; call after a barrier over the ranks.
 * ------------------------------------------------------------------------- */
typedef struct mopoe_comm mopoe_comm;
int mopoe_comm_create(int32_t rank, int32_t world, int32_t num_floats, int32_t timeout_ms,
                      mopoe_comm** comm, void* handle_out);
int mopoe_comm_connect(mopoe_comm* comm, const void* handles);
int mopoe_comm_allreduce(mopoe_comm* comm, float* data, void* stream);
int mopoe_comm_allreduce_adam(mopoe_comm* comm, const mopoe_model* model,
                              int32_t present_mask, const mopoe_buffers* buf,
                              const mopoe_adam* adam, void* stream);
int mopoe_comm_train_step(mopoe_comm* comm, const mopoe_model* model,
                          const mopoe_step* step, const mopoe_buffers* buf,
                          const mopoe_adam* adam, void* stream);
int mopoe_comm_status(mopoe_comm* comm, int32_t* timeouts);
int mopoe_comm_destroy(mopoe_comm* comm);

/* ---------------------------------------------------------------------------
 * General topologies (ABI 9): the flags of workflow.train_exp beyond its defaults
 * (workflow.py:41-49 -> multimodal_cohort/networks/networks.py:16-20,51-59,66-77):
 *   enc_layers    num_hidden_layer_encoder: Linear(., 256) + ReLU + Dropout, 0..4 of them
 *                 (0: the heads read x directly)
 *   dec_layers    num_hidden_layer_decoder: the same stack between z and out_mu
 *   dropout       dropout_rate of every Dropout module (live in training steps only;
 *                 ATen's arithmetic x * (keep / (1 - p)))
 *   sample_scale  learn_output_sample_scale: decoders.<m>.logvar is a Linear head on the
 *                 decoder's last hidden layer (a (N, d_m) scale) instead of a parameter
 * The default topology {1, 0, 0, 0} is what mopoe_forward / mopoe_train_step implement (one
 * fused launch + the weight-gradient launch); every other one runs through the entry points
 * below as a chain of launches around the same kernels (csrc/mopoe_general.inc).
 *
 * mopoe_topology_layout fills the offsets of BOTH structs: per modality the encoder's
 * parameters in one run of the flat buffer -- hidden layers (shared_encoder.<3l>), then the
 * heads rows [style_mu | style_logvar | class_mu | class_logvar] of width 256 (or d_m) --
 * and the decoder's in another: hidden layers (shared_decoder.<3l>), out_mu (d_m, 256 or
 * zd_m), then logvar (1, d_m) or the head logvar.weight / logvar.bias.
 * ------------------------------------------------------------------------- */
typedef struct mopoe_topology {
    int32_t enc_layers, dec_layers;
    float dropout;
    int32_t sample_scale;
    /* filled by mopoe_topology_layout(): float offsets into the flat buffer */
    int32_t off_we[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];  /* encoder layer l weight (256, l ? 256 : d_m) */
    int32_t off_be[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];
    int32_t off_wg[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];  /* decoder layer l weight (256, l ? 256 : zd_m) */
    int32_t off_bg[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];
    int32_t off_wlv[MOPOE_MAX_MODS];                   /* logvar head (d_m, 256 or zd_m)             */
    int32_t off_blv[MOPOE_MAX_MODS];
} mopoe_topology;

int mopoe_topology_layout(mopoe_model* model, mopoe_topology* topo);

/* Buffers of a general topology, next to mopoe_buffers (whose `hidden`, `g_pre` and `wfrag`
 * it does not use).  Caller-owned, row-major float32.  EB = mopoe_general_enc_blocks():
 * encoder-side tensors (these, and heads / g_heads of mopoe_buffers) have EB * n rows -- 2
 * when a training step with dropout has decoder jobs fed by a subset of their own (method
 * poe's unimodal ELBOs re-run the model, run_epochs.py:104-128: with live Dropout modules
 * the second encoder pass draws new masks; its rows are the second row block).  Decoder-side
 * tensors have R_m rows like z / loc. */
typedef struct mopoe_gbuffers {
    float* enc_act[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];  /* (EB n, 256) output of encoder layer l   */
    float* enc_pre0[MOPOE_MAX_MODS];                   /* (n, 256) layer 0 before its Dropout
                                                          (training with dropout > 0 only)       */
    float* dec_act[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];  /* (R_m, 256) output of decoder layer l    */
    float* g_enc[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];    /* (EB n, 256) d loss / d pre-activation   */
    float* g_dec[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];    /* (R_m, 256)                              */
    float* lv[MOPOE_MAX_MODS];                         /* (R_m, d_m) per-sample logvar (sample_scale) */
    float* g_lv[MOPOE_MAX_MODS];                       /* its gradient                             */
    float* g_z[MOPOE_MAX_MODS];                        /* (R_m, ldz_m) d loss / d z                */
    /* injected dropout keep masks (0 / 1 floats, the shape of the activation) or NULL: the
     * kernels draw them (Philox, keyed by seed / step / layer / element) */
    const float* keep_enc[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];
    const float* keep_dec[MOPOE_MAX_MODS][MOPOE_MAX_LAYERS];
} mopoe_gbuffers;

/* row blocks of the encoder-side buffers for this step: 1 or 2 (see mopoe_gbuffers) */
int mopoe_general_enc_blocks(const mopoe_topology* topo, const mopoe_step* step, int train);
/* mopoe_forward for a general topology (evaluation: Dropout is the identity) */
int mopoe_general_forward(const mopoe_model* model, const mopoe_topology* topo,
                          const mopoe_step* step, const mopoe_buffers* buf,
                          const mopoe_gbuffers* gbuf, void* stream);
/* mopoe_train_step for a general topology.  `rccl` NULL: the one-rank step (Adam fused
 * into the weight-gradient launches when `adam` is given).  `rccl` non-NULL: the N-rank step
 * as in mopoe_rccl_train_step (gradients, ncclAllReduce, the Adam launch). */
struct mopoe_rccl;
int mopoe_general_train_step(const mopoe_model* model, const mopoe_topology* topo,
                             const mopoe_step* step, const mopoe_buffers* buf,
                             const mopoe_gbuffers* gbuf, const mopoe_adam* adam,
                             struct mopoe_rccl* rccl, void* stream);
/* mopoe_adam_step over the segments of a general topology */
int mopoe_general_adam_step(const mopoe_model* model, const mopoe_topology* topo,
                            int32_t present_mask, const mopoe_buffers* buf,
                            const mopoe_adam* adam, int32_t world, void* stream);

/* ---------------------------------------------------------------------------
 * The data-parallel step over RCCL as ONE call (ABI 9; SURVEY.md section 8e: "one
 * all-reduce (sum, then x 1/world) of the flat fp32 gradient buffer per step", feeding
 * the optimizer step of experiment.py:256-279).  librccl is bound at run time (dlopen;
 * inside a PyTorch process: the instance torch has loaded), the communicator is the
 * library's own:
 *
 *   mopoe_rccl_unique_id   ncclGetUniqueId on ONE rank; the caller hands the
 *                          MOPOE_RCCL_ID_BYTES to the other ranks out of band (e.g.
 *                          torch.distributed.broadcast) ...
 *   mopoe_rccl_create      ... and every rank calls this (ncclCommInitRank: collective,
 *                          the current HIP device is the rank's GPU).
 *   mopoe_rccl_train_step  replaces `mopoe_train_step(adam = NULL); all_reduce(grads);
 *                          mopoe_adam_step(world)` by one host call: forward + backward,
 *                          ncclAllReduce (sum) of buf->grads -- all num_floats, control
 *                          words included -- and the Adam launch with 1 / world, enqueued
 *                          back to back on `stream`.  No rank applies a step whose batch
 *                          held other modalities on some rank or that some rank could not
 *                          complete; every rank then raises MOPOE_CTR_INVALID at the same
 *                          step (MOPOE_CTR_FIRST_INVALID), so all ranks can re-arm and run
 *                          the same batches again.
 *   mopoe_rccl_allreduce   the plain in-place sum of `count` floats.
 *   mopoe_rccl_destroy     ncclCommDestroy.
 * ------------------------------------------------------------------------- */
typedef struct mopoe_rccl mopoe_rccl;
int mopoe_rccl_unique_id(void* id_out);
int mopoe_rccl_create(int32_t rank, int32_t world, const void* id, mopoe_rccl** comm);
int mopoe_rccl_train_step(mopoe_rccl* comm, const mopoe_model* model, const mopoe_step* step,
                          const mopoe_buffers* buf, const mopoe_adam* adam, void* stream);
int mopoe_rccl_allreduce(mopoe_rccl* comm, float* data, int64_t count, void* stream);
/* what RCCL itself says about the communicator (ncclCommUserRank / ncclCommCount), not what
 * the caller passed to mopoe_rccl_create: bench.py prints it next to n_gpus (ABI 11) */
int mopoe_rccl_info(mopoe_rccl* comm, int32_t* rank, int32_t* world);
int mopoe_rccl_destroy(mopoe_rccl* comm);

/* ---------------------------------------------------------------------------
 * HOST function (no device work, no stream): one epoch of the reference's
 * MissingModalitySampler (multimodal_cohort/dataset.py:296-354, non-stratified path) --
 * batches whose samples all have the same modality subset, drawn with
 * np.random.choice(rest, size, replace=False) until a subset is used up, the complete
 * batches then the incomplete ones, each group in an np.random.choice order -- bit for bit
 * the draws of numpy's legacy global RandomState:
 *   mt_key / mt_pos   in: np.random.get_state()[1] (624 uint32) and [2]; out: the state
 *                     after the epoch's draws (np.random.set_state it back)
 *   subset_begin      (num_subsets + 1) prefix offsets into subset_items, the sample
 *                     indices of every modality subset (dataset.idx_per_modality_subset)
 *   out_items         (total samples) the epoch, batch after batch
 *   out_begin         (number of batches + 1) offsets into out_items; the caller sizes it
 *                     sum(ceil(len_s / batch_size)) + 1
 *   out_subset        (number of batches) the modality subset each batch came from
 * ------------------------------------------------------------------------- */
int mopoe_sampler_epoch(uint32_t* mt_key, int32_t* mt_pos, int32_t num_subsets,
                        const int64_t* subset_begin, const int64_t* subset_items,
                        int64_t batch_size, int64_t* out_items, int64_t* out_begin,
                        int32_t* out_subset, int64_t* num_batches);
/* HOST function: the gather vectors of those batches over per-modality blocks resident in
 * HBM (mopoe_buffers.row_index).  rows[k] (subjects) = block row of a subject in modality
 * k's block or -1 (reference dataset.py:99-126: idx_per_mod); indices = the dataset's
 * optional subject subset (dataset.indices) or NULL; subset_has[s * num_mods + k] = modality
 * subset s holds modality k.  out_rows[k] receives the rows of every batch that holds k,
 * batch after batch (sized by the caller: the samples of those batches), out_start[k][b]
 * the offset of batch b in it or -1. */
int mopoe_sampler_rows(int32_t num_mods, int64_t num_batches, const int64_t* items,
                       const int64_t* batch_begin, const int32_t* batch_subset,
                       const uint8_t* subset_has, const int64_t* indices,
                       const int64_t* const* rows, int32_t* const* out_rows,
                       int64_t* const* out_start);

/* Free functions of section 8b, float32 device tensors. */

/* torch.nn.Linear (+ optional ReLU) as used by Encoder.forward / Decoder.forward
 * (multimodal_cohort/networks/networks.py:30-36,66-77):
 * y (n, ncols) = act(x (n, k) @ w (ncols, k)^T + b). */
int mopoe_linear(const float* x, int32_t n, int32_t k, const float* w,
                 const float* b, int32_t ncols, int32_t relu, float* y,
                 void* stream);

/* divergence_measures/mm_div.py:13-20  poe(mu, logvar, eps): (E,n,d)->(n,d) */
int mopoe_poe(const float* mu, const float* logvar, int32_t num_experts,
              int64_t numel, float eps, float* out_mu, float* out_logvar,
              void* stream);
/* divergence_measures/kl_div.py:7-14  calc_kl_divergence(mu0, logvar0,
 * norm_value): scalar; `scratch` holds >= 1024 floats. */
int mopoe_kl_divergence(const float* mu, const float* logvar, int64_t numel,
                        float norm_value, float* scratch, float* out,
                        void* stream);
/* utils/BaseMMVae.py:37-40  reparameterize: out = eps * exp(0.5*logvar) + mu */
int mopoe_reparameterize(const float* mu, const float* logvar,
                         const float* eps, int64_t numel, uint64_t seed,
                         uint64_t stream_id, float* out, void* stream);
/* utils/utils.py:63-85  mixture_component_selection with host-computed slice
 * starts (K+1 ints, device): (K,n,d) -> (n,d) */
int mopoe_mixture_select(const float* mus, const float* logvars,
                         int32_t num_comp, int32_t n, int32_t d,
                         const int32_t* bounds, float* out_mu,
                         float* out_logvar, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOPOE_HIP_H */
