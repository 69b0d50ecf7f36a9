// mopoe_kernels.hip -- hand-written gfx950 kernels of the MoPoE-VAE training
// step and the C ABI declared in include/mopoe_hip.h.
//
// One training step = TWO launches on the caller's stream while its grid fits the chip
// (batches of up to a few thousand rows):
//   k_fused<FORM>  producer workgroups: h_m = relu(x_m W1_m^T + b1_m) (MFMA, per row
//                  tile x hidden column block x modality), handed over through memory to
//                  the consumer workgroups of the SAME launch, one per row group (16 rows,
//                  or 4 in the four-row form of batches <= 256: FORM 4 / 5): everything
//                  that is per-sample -- encoder heads (MFMA), powerset-of-experts fusion
//                  + KL + mixture selection + reparameterisation (VALU, wave reductions),
//                  decoder + NLL (MFMA + epilogue) and the whole data-gradient chain back
//                  to the pre-ReLU gradient (latent_body, mopoe_latent.inc)
//   k_wgrad        all weight / bias gradients as reductions over the batch (MFMA), the
//                  Adam update fused into the epilogue when no exchange sits in between
// Larger batches (and MOPOE_NO_FUSE=1) keep the first launch in two: k_linear (or
// k_linear_big), then k_latent = latent_body on its own.  A data-parallel step adds the
// gradient exchange (RCCL: mopoe_rccl.inc; peer windows: mopoe_xgmi.inc) and k_adam.
// The math follows SURVEY.md Appendix A; reference file:line citations are in
// include/mopoe_hip.h and DESIGN.md.
#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "mopoe_common.h"

namespace {

constexpr int kLatentThreads = 1024;
constexpr int kLatentWaves = kLatentThreads / kWave;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr float kLog2 = 0.69314718055994530942f;
constexpr float kPoeEps = 1e-8f;
constexpr int kHandoffSpins = 1 << 18;  // bounded wait of a consumer group (~0.3 s)
constexpr int kStampWord = 80;          // (diagnostic build) unused words of a wave's LDS slot
constexpr int kStampBase = 128;         // (diagnostic build) stamps sit behind the stats
[[maybe_unused]] constexpr int kCtrStamp = MOPOE_NUM_COUNTERS;  // (diagnostic build) and behind the counters
static_assert(kStampWord >= kNumPart && kStampWord + 1 < kStatStride, "stamp words are free");
static_assert(kStampBase >= MOPOE_NUM_STATS, "stamps sit behind the stats");

// Diagnostic build only (-DMOPOE_STAMPS, libmopoe_hip_stamps.so): thread 0 of
// block 0 parks the low words of s_memrealtime [100 MHz] and s_memtime [shader
// clock] in unused LDS words at stage boundaries and copies them behind the
// stats at the end (tools/stage_stamps.py).  Going through LDS matters: a stamp
// that takes a pointer out of the argument block makes hipcc copy the block to
// scratch, and any private segment triples this kernel's time.
#ifdef MOPOE_STAMPS
#define STAMP(buf, i)                                                                  \
    do {                                                                               \
        if (stamp_blk && threadIdx.x == 0) {                                          \
            stamp_lds[(i) * kStatStride + kStampWord] =                                        \
                __uint_as_float((unsigned)__builtin_amdgcn_s_memrealtime());           \
            stamp_lds[(i) * kStatStride + kStampWord + 1] =                                        \
                __uint_as_float((unsigned)__builtin_amdgcn_s_memtime());               \
        }                                                                              \
    } while (0)
#define STAMPW(buf, i, w)                                                              \
    do {                                                                               \
        if (stamp_blk && threadIdx.x == (w) * 64) {                                   \
            stamp_lds[(i) * kStatStride + kStampWord] =                                        \
                __uint_as_float((unsigned)__builtin_amdgcn_s_memrealtime());           \
            stamp_lds[(i) * kStatStride + kStampWord + 1] =                                        \
                __uint_as_float((unsigned)__builtin_amdgcn_s_memtime());               \
        }                                                                              \
    } while (0)
#define STAMP_FLUSH(stats_ptr, n)                                                      \
    do {                                                                               \
        if (stamp_blk && threadIdx.x < 2 * (n))                                       \
            (stats_ptr)[kStampBase + threadIdx.x] =                                    \
                stamp_lds[(threadIdx.x >> 1) * kStatStride + kStampWord + (threadIdx.x & 1)]; \
    } while (0)
#else
#define STAMP(buf, i) \
    do {              \
    } while (0)
#define STAMPW(buf, i, w) \
    do {                  \
    } while (0)
#define STAMP_FLUSH(stats_ptr, n) \
    do {                          \
    } while (0)
#endif
// whole-step timeline (diagnostic build): 100 MHz realtime counter words written
// straight to global memory by one thread of one block
#ifdef MOPOE_STAMPS
#define GSTAMP(ptr, i, cond)                                                            \
    do {                                                                               \
        if (cond) ((volatile unsigned*)(ptr))[i] = (unsigned)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define GSTAMP(ptr, i, cond) \
    do {                     \
    } while (0)
#endif
struct KArgs {
    mopoe_model mdl;
    mopoe_step st;
    mopoe_buffers buf;
    LatentLds lds;  // carve-up of k_latent's LDS, computed on the host
};
static_assert(sizeof(KArgs) <= 7680 + 512, "kernel argument block too large");  // (12 KB launches fine: tools/kernarg_probe.hip)

DEV int src_row(const mopoe_buffers& buf, int m, int gn) {
    return buf.row_index[m] ? buf.row_index[m][gn] : gn;
}

// ---------------------------------------------------------------------------
// k_linear: grouped Y_g[n, j] = act(sum_k X_g[n, k] W_g[j, k] + b_g[j]).
// grid = (max column groups of 64, row tiles, groups), block = 256 (4 waves).
// The training step uses it once, grouped over the present modalities, for
// the encoder's hidden layer h_m = relu(x_m W1_m^T + b1_m).  The 16 x K input
// tile is staged in LDS with coalesced 16-byte loads over the (batch x
// feature) layout and shared by the four waves; each wave owns one 16 x 16
// output tile and streams its 16 rows of W straight from L2 into registers.
// ---------------------------------------------------------------------------
struct LinGroup {
    const float* X;        // (rows, K), row stride ldx
    const int32_t* rows;   // optional gather index (n) or nullptr
    const float* W;        // (ncols, K) row-major
    const float* b;        // (ncols) or nullptr
    float* Y;              // (n, ncols), row stride ldy
    int32_t K, ldx, ncols, ldy, relu;
    int32_t xrows;         // rows of X: nothing past xrows * ldx floats is read (the range
                           // check of the descriptor masks a 16-byte load over the tail)
    int32_t wslack;        // W stays readable 12 bytes past its end (a segment of the flat
                           // parameter buffer): rows whose length is not a multiple of 4
                           // are then read with 16-byte loads too
};

struct LinArgs {
    int32_t n;
    int32_t ngroups;
    int32_t* counters;     // training step: bump [0], publish Adam coefficients
    int32_t publish;       // adam is valid: publish its step coefficients
    int32_t ksplit;        // 1, 2 or 4: K parts per column tile (set by launch_linear)
    int32_t num_mods;      // modalities of the model (the Adam records of step_begin)
    int32_t spins;         // fused launch: polls of a row group before it gives up
    int32_t knock;         // (diagnostic build -DMOPOE_KNOCK: phases to leave out)
    int32_t bf16;          // mopoe_step.gemm_operands == MOPOE_OPERANDS_BF16 (k_linear_big only)
    int32_t xcd_order;     // k_linear_big: XCD-aware tile order (MOPOE_LIN_XCD, default 1)
    int32_t sig_n, sig_stride, sig_groups;   // fused launch: row groups per 16-row tile, words
                                             // between their flags, row groups in all
    mopoe_adam adam;
    LinGroup g[MOPOE_MAX_MODS];
};

// Dropout(p) behind a hidden layer of a general topology, applied in the layer's own epilogue
// (k_linear_drop; mopoe_general.inc: the keep decisions are g_dropout's -- the same Philox
// stream and element numbering, or the injected mask)
struct LinDropGroup {
    const float* keep;     // injected keep mask (n, ncols) of 0 / 1, or nullptr: Philox
    uint32_t stream;       // Philox stream of this (modality, stack, layer)
    uint32_t row0;         // first row's number inside the stream
    int32_t rows, pad;     // rows of this group's activation (the launch's n is the groups' maximum)
};
struct LinDrop {
    float p, scale;        // scale = 1 / (1 - p) in float32, as ATen's noise.div_(1 - p)
    uint64_t seed;
    const int32_t* counters;   // the step number: [BEGUN], or [DONE] + 1 in the launch that begins the step
    LinDropGroup g[MOPOE_MAX_MODS];
};
struct LinDropArgs {
    LinArgs la;
    LinDrop d;
};

// The likelihood of a general topology's training step in the OUTPUT layer's epilogue
// (k_linear_nll; mopoe_general.inc): the tile has loc in registers -- NLL terms, d loss / d loc
// and the column sums of d loss / d logvar are made there, the tile's NLL partial is left in
// the row group's slab (nll_tiles_off) for the next launch to add up.
constexpr int kNllSlots = 2;   // decoder passes of a modality (joint + unimodal): kLvoSlots
struct LinNllGroup {
    const float* x;            // (x_rows, ncols) input of the modality
    const int32_t* row_index;  // gather (n_step) or nullptr
    const float* lvo;          // (ncols) learnt logvar
    float* g_xhat;             // out (rows, ncols)
    const float* loc;          // sample scale: the loc head's output (the launch in front), (rows, ncols)
    float* g_lv;               // sample scale: out, d loss / d logvar (rows, ncols)
    int32_t x_rows, rows;      // rows = slots * n_step of this group
    int32_t lvo_part[kNllSlots], tile_off[kNllSlots];
    float coef[kNllSlots];     // nll_coef / n_step of the slot's job
};
struct LinNll {
    int32_t n_step, laplace, part_stride;   // n_step % 16 == 0 where a modality has two slots
    int32_t sample_scale;      // this launch is the LOGVAR head (learn_output_sample_scale): Y = logvar, loc is read
    float* partials;
    LinNllGroup g[MOPOE_MAX_MODS];
};
struct LinNllArgs {
    LinArgs la;
    LinNll q;
};

// Adam scalars of step t, torch.optim.Adam (_single_tensor_adam) semantics:
// python-double scalars applied to float32 tensors.
struct AdamCoef {
    float b2, one_m_b1, one_m_b2, step_size, bc2_sqrt, eps, pad;
};

// torch keeps the step count PER PARAMETER and skips parameters whose .grad is None, so
// every modality's parameters have their own count t_m (counters[MOPOE_CTR_ADAM_STEPS +
// m]) and their own bias corrections 1 - beta1^t, sqrt(1 - beta2^t) -- two pow() in
// double, ~1 us of one thread.  They are kept off every step's critical path: the ONE
// block that ends a step's Adam kernel computes the records of the NEXT step, for all
// modalities (the step of an absent one stays where it is), into slot (s + 1) & 1 of
// counters[MOPOE_CTR_BIAS ..]; the readers of step s use slot s & 1, so nobody reads a
// record while it is written.  A record is tagged with the step and the betas it was
// computed for; the step's first kernel recomputes the ones that do not match (first
// step ever, optimiser state loaded from outside, betas changed).
struct BiasRec {
    double bc1;      // 1 - beta1^t
    float bc2_sqrt;  // sqrt(1 - beta2^t)
    int32_t tag;     // global step the record belongs to
};
static_assert(sizeof(BiasRec) == 16, "four words per record");
constexpr int kCtrBeta = 10;   // counters[10], [11]: bits of the betas the records were made for
static_assert(MOPOE_CTR_BIAS + 2 * MOPOE_MAX_MODS * 4 <= MOPOE_NUM_COUNTERS, "records fit");

DEV BiasRec* bias_rec(int32_t* counters, int s, int m) {
    return reinterpret_cast<BiasRec*>(counters + MOPOE_CTR_BIAS) + ((s & 1) * MOPOE_MAX_MODS + m);
}
DEV BiasRec bias_of(const mopoe_adam& ad, int t, int tag) {
    BiasRec r;
    r.bc1 = 1.0 - pow((double)ad.beta1, (double)t);
    r.bc2_sqrt = (float)sqrt(1.0 - pow((double)ad.beta2, (double)t));
    r.tag = tag;
    return r;
}
DEV bool betas_match(const int32_t* counters, const mopoe_adam& ad) {
    return counters[kCtrBeta] == __builtin_bit_cast(int32_t, ad.beta1) &&
           counters[kCtrBeta + 1] == __builtin_bit_cast(int32_t, ad.beta2);
}
DEV AdamCoef adam_coef_of(const mopoe_adam& ad, double bc1, float bc2_sqrt) {
    AdamCoef c;
    c.b2 = ad.beta2;
    c.one_m_b1 = (float)(1.0 - (double)ad.beta1);
    c.one_m_b2 = (float)(1.0 - (double)ad.beta2);
    c.step_size = (float)((double)ad.lr / bc1);
    c.bc2_sqrt = bc2_sqrt;
    c.eps = ad.eps;
    c.pad = 0.f;
    return c;
}

// First kernel of a training step, ONE thread: the step number, and (when the step
// applies Adam) records that are valid for it.  Everything is read before anything is
// written: a store to the same buffer in between would order the loads behind it, one
// memory round trip each.
DEV void step_begin(int32_t* counters, int num_mods, const mopoe_adam* ad) {
    const int c0 = counters[MOPOE_CTR_STEPS_BEGUN];
    const int cb1 = counters[kCtrBeta], cb2 = counters[kCtrBeta + 1];
    int tag0[MOPOE_MAX_MODS], tag1[MOPOE_MAX_MODS], steps[MOPOE_MAX_MODS];
#pragma unroll
    for (int m = 0; m < MOPOE_MAX_MODS; ++m) {
        tag0[m] = counters[MOPOE_CTR_BIAS + m * 4 + 3];
        tag1[m] = counters[MOPOE_CTR_BIAS + (MOPOE_MAX_MODS + m) * 4 + 3];
        steps[m] = counters[MOPOE_CTR_ADAM_STEPS + m];
    }
    const int s = c0 + 1;
    counters[MOPOE_CTR_STEPS_BEGUN] = s;
    if (!ad) return;
    const int ab1 = __builtin_bit_cast(int32_t, ad->beta1), ab2 = __builtin_bit_cast(int32_t, ad->beta2);
    const bool same = cb1 == ab1 && cb2 == ab2;
#pragma unroll
    for (int m = 0; m < MOPOE_MAX_MODS; ++m)
        if (m < num_mods && (!same || ((s & 1) ? tag1[m] : tag0[m]) != s))
            *bias_rec(counters, s, m) = bias_of(*ad, steps[m] + 1, s);
    if (!same) {
        counters[kCtrBeta] = ab1;
        counters[kCtrBeta + 1] = ab2;
    }
}

// Last block of the kernel that applied (or, on an invalid step, withheld) the Adam
// update of step s, threads m < num_mods: advance the counts of the modalities that
// were updated, write every modality's record of step s + 1.
DEV void step_end(int32_t* counters, int s, int m, int num_mods, int present_mask, bool applied,
                  const mopoe_adam& ad) {
    if (m >= num_mods) return;
    int t = counters[MOPOE_CTR_ADAM_STEPS + m];
    if (applied && ((present_mask >> m) & 1)) counters[MOPOE_CTR_ADAM_STEPS + m] = ++t;
    *bias_rec(counters, s + 1, m) = bias_of(ad, t + 1, s + 1);
    if (m == 0) {
        counters[kCtrBeta] = __builtin_bit_cast(int32_t, ad.beta1);
        counters[kCtrBeta + 1] = __builtin_bit_cast(int32_t, ad.beta2);
    }
}

// A block's coefficients for modality m: both parity slots, the step number and the
// invalid word are requested together, early, next to the block's other operands; the
// selection sits where the values are first needed (a branch right after the request
// would hold every later load back one round trip).
struct AdamCoefRaw {
    int s, invalid, b1, b2;
    BiasRec r[2];
};
DEV AdamCoefRaw adam_coef_request(const int32_t* counters, int m) {
    AdamCoefRaw q;
    q.s = __builtin_nontemporal_load(counters + MOPOE_CTR_STEPS_BEGUN);
    q.invalid = __builtin_nontemporal_load(counters + MOPOE_CTR_INVALID);
    q.b1 = __builtin_nontemporal_load(counters + kCtrBeta);
    q.b2 = __builtin_nontemporal_load(counters + kCtrBeta + 1);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int32_t* w = counters + MOPOE_CTR_BIAS + (k * MOPOE_MAX_MODS + m) * 4;
        const uint32_t lo = (uint32_t)__builtin_nontemporal_load(w);
        const uint32_t hi = (uint32_t)__builtin_nontemporal_load(w + 1);
        q.r[k].bc1 = __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
        q.r[k].bc2_sqrt = __builtin_bit_cast(float, __builtin_nontemporal_load(w + 2));
        q.r[k].tag = __builtin_nontemporal_load(w + 3);
    }
    return q;
}
// false: the step must not be applied (MOPOE_CTR_INVALID is up, or there is no record
// for this step -- which the step's first kernel rules out)
DEV bool adam_coef_resolve(const AdamCoefRaw& q, const mopoe_adam& ad, AdamCoef& c) {
    const BiasRec r = (q.s & 1) ? q.r[1] : q.r[0];
    c = adam_coef_of(ad, r.bc1, r.bc2_sqrt);
    return q.invalid == 0 && r.tag == q.s && q.b1 == __builtin_bit_cast(int32_t, ad.beta1) &&
           q.b2 == __builtin_bit_cast(int32_t, ad.beta2);
}
// The Adam kernels that run on their own (k_adam, k_xgmi), one thread per block: the
// record if it is there, else the same numbers from the count (nobody writes the counts
// before the kernel's last block has seen every other block finish).
DEV AdamCoef adam_coef_load(const int32_t* counters, int m, const mopoe_adam& ad) {
    const int s = counters[MOPOE_CTR_STEPS_BEGUN];
    const BiasRec* r = bias_rec(const_cast<int32_t*>(counters), s, m);
    if (r->tag == s && betas_match(counters, ad)) return adam_coef_of(ad, r->bc1, r->bc2_sqrt);
    const BiasRec f = bias_of(ad, counters[MOPOE_CTR_ADAM_STEPS + m] + 1, s);
    return adam_coef_of(ad, f.bc1, f.bc2_sqrt);
}

DEV void adam_update(const AdamCoef& c, float g, float p, float m, float v, float* po,
                     float* mo, float* vo) {
    // every operation rounds by itself, like the separate tensor ops of torch's Adam --
    // and the three kernels that inline this (k_wgrad, k_adam, k_xgmi) then agree bit
    // for bit instead of each getting its own choice of fused multiply-adds
#pragma clang fp contract(off)
    const float m1 = m + c.one_m_b1 * (g - m);           // exp_avg.lerp_(g, 1-b1)
    const float v1 = v * c.b2 + (c.one_m_b2 * g) * g;     // mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = sqrtf(v1) / c.bc2_sqrt + c.eps;
    *mo = m1;
    *vo = v1;
    *po = p - c.step_size * (m1 / denom);                 // addcdiv_(m, denom, -step)
}

// Four consecutive floats of an input row starting at column k; columns >= Kc (the next
// row's first values when the row length is not a multiple of 4, which the 16-byte load
// picks up) become zeros.  One load, no branch -- with per-element loads the 7-column
// modality's encoder tiles were the LAST producers of a row tile, by 2.3 us.
DEV f32x4 ldg4_row(rsrc_t r, uint32_t base, bool ok, int k, int Kc) {
    f32x4 v = ldg4(r, guard(base, ok & (k < Kc)));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float f = v[e];
        v[e] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, f) &
                                             (k + e < Kc ? 0xFFFFFFFFu : 0u));
    }
    return v;
}

constexpr int kLinRedFloats = 4 * kWave * 4;  // partial tiles handed over through LDS

// One workgroup = 4 waves = one 16-row tile x (4 / KS) column tiles of 16, the K axis
// of a column tile cut in KS interleaved parts (fragment f of 16 k belongs to part
// f % KS), one wave per (column tile, part).  KS = 4 for small batches: the MFMA
// accumulation chain of a 16x16 tile over K = 444 is 111 issues of 32 cycles, and a
// small batch has too few tiles to keep the chip's SIMDs busy otherwise; KS = 1 for
// large ones, where a workgroup should cover as many columns per staged x tile as it can.
constexpr int kEpiPlain = 0, kEpiDrop = 1, kEpiNll = 2;
template <int KS, int EPI>
DEV void linear_tile(const LinArgs& a, const LinGroup& g, float* lds, const int* rowsel,
                     int tid, int lane, int wave, const void* ex, uint32_t step_no) {
    constexpr bool DROP = EPI == kEpiDrop;
    const LinDrop* dr = (const LinDrop*)ex;
    constexpr int CH = 8;      // W fragments per wave and batch
    constexpr int kStage = 8;  // float4 loads in flight per thread while staging
    const int N = a.n, K = g.K;
    const bool vec = K % 4 == 0 || g.wslack;   // W: 4-wide reads stay inside the buffer
    const int n0 = blockIdx.y * kRows;
    const int tile = wave / KS, part = wave % KS;
    const int j0 = (blockIdx.x * (4 / KS) + tile) * 16;
    const rsrc_t xr = make_rsrc(g.X, (size_t)g.xrows * g.ldx * sizeof(float));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    // Dropout: the tile's 16 x 16 keep factors are ONE Philox quad (four columns of a row) per
    // lane, drawn here by the tile's last K part -- whose wave has nothing to do at the end --
    // while the x tile is on its way, and left in the fourth slot of the partial tiles' LDS area
    // (slots t KS .. t KS + KS - 2 carry tile t's partials).  (Drawn in the epilogue by the one
    // wave that stores the tile, a quad per ELEMENT, the 32-bit multiplies of four Philox calls
    // were 1.1 us at the end of every block: 9.1 us per layer against 6.6 + 4.5 apart.)
    float* const dropk = lds + kRows * (round_up(min(K, kEncKChunk), 16) + 4) + (tile * KS + KS - 1) * kWave * 4;
    if (DROP && part == KS - 1) {
        const LinDropGroup dg = dr->g[blockIdx.z];
        const int pr = lane >> 2, pq = lane & 3;
        const int gn = n0 + pr, c0 = j0 + 4 * pq;
        f32x4 kp = {1.f, 1.f, 1.f, 1.f};
        if (gn < dg.rows && c0 < g.ncols) {   // (ncols % 4 == 0: hidden layers are 256 wide)
            if (dg.keep) {
                kp = *reinterpret_cast<const f32x4*>(dg.keep + (size_t)gn * g.ncols + c0);
            } else {
                const f32x4 u = philox_uniform4(dr->seed, step_no, dg.stream,
                                                (dg.row0 + (uint32_t)gn) * (uint32_t)(g.ncols >> 2) + (uint32_t)(c0 >> 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) kp[e] = u[e] >= dr->p ? 1.f : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(dropk + lane * 4) = kp;   // [row pr][column 4 pq ..]
    }
    for (int kc0 = 0; kc0 < K; kc0 += kEncKChunk) {
        const int Kc = min(kEncKChunk, K - kc0);
        const int Kp = round_up(Kc, 16);
        const int ldx = Kp + 4;
        const int q4 = Kp / 4;
        const int kend = j0 < g.ncols ? Kp : 0;
        const rsrc_t wr = make_rsrc(g.W + kc0, ((size_t)g.ncols * K + (g.wslack ? 3 : 0)) * sizeof(float));
        // fragment i of this wave starts at k = 16 * (part + KS * i)
        auto load_w = [&](int i0, f32x4 (&b)[CH]) __attribute__((always_inline)) {
            if (vec) {  // (a select of the two forms would issue BOTH sets of loads)
#pragma unroll
                for (int c = 0; c < CH; ++c)  // k >= Kc is out of range -> 0
                    b[c] = glb_b4_nt<true>(wr, K, Kc, j0, 16 * (part + KS * (i0 + c)), lane);
            } else {
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    b[c] = glb_b4_nt<false>(wr, K, Kc, j0, 16 * (part + KS * (i0 + c)), lane);
            }
        };
        f32x4 b[CH];
        if (kc0 > 0) __syncthreads();
        for (int s0 = 0; s0 < kRows * q4; s0 += kStage * 256) {
            f32x4 v[kStage];
#pragma unroll
            for (int i = 0; i < kStage; ++i) {
                const int s = s0 + i * 256 + tid;
                const int r = min(s / q4, kRows - 1), k = (s - (s / q4) * q4) * 4;
                const bool rv = (s < kRows * q4) & (n0 + r < N);
                const int row = rowsel ? rowsel[r] : min(n0 + r, N - 1);
                const uint32_t base = (uint32_t)(row * g.ldx + kc0 + k) * 4u;
                v[i] = ldg4_row(xr, base, rv, k, Kc);
            }
            // the wave's first batch of W fragments goes out BEHIND the x loads (loads
            // return in order: requested first, it would hold the x tile -- and the
            // barrier every wave waits at -- back until W has arrived too)
            if (s0 == 0) load_w(0, b);
#pragma unroll
            for (int i = 0; i < kStage; ++i) {
                const int s = s0 + i * 256 + tid;
                if (s < kRows * q4) {
                    const int r = s / q4, k = (s - r * q4) * 4;
                    *reinterpret_cast<f32x4*>(lds + r * ldx + k) = v[i];
                }
            }
        }
        __syncthreads();
        GSTAMP(a.counters, kCtrStamp + 14, a.counters && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == (a.ngroups > 1 ? 1 : 0) && tid == 0);
        for (int i0 = 0; 16 * (part + KS * i0) < kend; i0 += CH) {
            f32x4 bn[CH];
            const bool more = 16 * (part + KS * (i0 + CH)) < kend;  // wave-uniform
            if (more) load_w(i0 + CH, bn);
            // No branch per fragment: it would put every LDS read behind the previous
            // fragment's MFMAs.  A fragment past the end of K multiplies the (finite)
            // first fragment of the x tile by W values read out of range, i.e. zeros.
            f32x4 av[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int k = 16 * (part + KS * (i0 + c));
                av[c] = lds_a4(lds, ldx, k < kend ? k : 0, lane);
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                acc = mfma_16x16x4(av[c][0], b[c][0], acc);
                acc2 = mfma_16x16x4(av[c][1], b[c][1], acc2);
                acc = mfma_16x16x4(av[c][2], b[c][2], acc);
                acc2 = mfma_16x16x4(av[c][3], b[c][3], acc2);
            }
            if (more) {
#pragma unroll
                for (int c = 0; c < CH; ++c) b[c] = bn[c];
            }
        }
    }
    acc += acc2;
    GSTAMP(a.counters, kCtrStamp + 15, a.counters && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == (a.ngroups > 1 ? 1 : 0) && tid == 0);
    if (KS > 1) {  // parts 1.. hand their partial tiles over through LDS (behind the x tile)
        float* red = lds + kRows * (round_up(min(K, kEncKChunk), 16) + 4);
        if (part > 0) *reinterpret_cast<f32x4*>(red + ((wave - 1) * kWave + lane) * 4) = acc;
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int p = 1; p < KS; ++p)  // fixed order
            acc += *reinterpret_cast<const f32x4*>(red + ((wave + p - 1) * kWave + lane) * 4);
    }
    const int col = j0 + (lane & 15);
    if (EPI == kEpiNll) {
        // Modality.calc_log_prob (modalities/modality.py:42-45) of this tile: g_nll's arithmetic
        if (j0 >= g.ncols) return;   // (wave-uniform)
        const LinNll& nq = *(const LinNll*)ex;
        const LinNllGroup& ng = nq.g[blockIdx.z];
        if (n0 >= ng.rows) return;   // (the launch's rows are the groups' maximum: not a tile of this modality)
        const int slot = n0 / nq.n_step;          // (a tile lies inside one slot)
        const int g0 = n0 - slot * nq.n_step;     // its first row inside the slot
        float* part = nq.partials + (size_t)(g0 >> 4) * nq.part_stride;
        const float coef = slot == 0 ? ng.coef[0] : ng.coef[1];
        const bool laplace = nq.laplace != 0, cv = col < g.ncols;
        const int q = lane >> 4, dm = g.ncols;
        const float bias = (g.b && cv) ? g.b[col] : 0.f;
        const bool ss = nq.sample_scale != 0;   // (the logvar head's launch: this tile is logvar, loc is read)
        const float lcol = (cv && !ss) ? ng.lvo[col] : 0.f;
        float xv[4], lc[4];
        bool rv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gn = g0 + 4 * q + r;        // row inside the slot
            rv[r] = cv & (gn < nq.n_step) & (n0 + 4 * q + r < ng.rows);
            const int src = ng.row_index ? ng.row_index[rv[r] ? gn : 0] : gn;
            xv[r] = (rv[r] && (unsigned)src < (unsigned)ng.x_rows) ? ng.x[(size_t)src * dm + col] : 0.f;
            lc[r] = (ss && rv[r]) ? ng.loc[(size_t)(n0 + 4 * q + r) * dm + col] : 0.f;
        }
        const float inv_var_col = expf(laplace ? -0.5f * lcol : -lcol);
        float term = 0.f, glv = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t o = (size_t)(n0 + 4 * q + r) * dm + col;
            const float head = acc[r] + bias;
            const float loc = ss ? lc[r] : head;
            const float l = ss ? head : lcol;
            const float inv_var = ss ? expf(laplace ? -0.5f * l : -l) : inv_var_col;
            const float diff = xv[r] - loc;
            float qq, gg, gl, t;
            if (laplace) {   // scale b = e^(lv / 2): |diff| / b + lv / 2 + log 2
                qq = fabsf(diff) * inv_var;
                t = qq + 0.5f * l + kLog2;
                gl = 0.5f - 0.5f * qq;
                gg = -sign_of(diff) * inv_var * coef;
            } else {         // diff^2 / (2 var) + lv / 2 + log sqrt(2 pi)
                qq = 0.5f * diff * diff * inv_var;
                t = qq + 0.5f * l + kHalfLog2Pi;
                gl = 0.5f - qq;
                gg = -diff * inv_var * coef;
            }
            if (rv[r]) {
                g.Y[o] = head;   // (ldy == ncols)
                ng.g_xhat[o] = gg;
                if (ss) ng.g_lv[o] = gl * coef;
                term += t;
                glv += gl;
            }
        }
        // d loss / d logvar of the group's rows, per column (the learnt logvar vector): this lane's
        // four rows + the three other row quarters of the same column (lanes c + 16, + 32, + 48)
        glv += __shfl_xor(glv, 16);
        glv += __shfl_xor(glv, 32);
        if (q == 0 && cv && !ss) part[(slot == 0 ? ng.lvo_part[0] : ng.lvo_part[1]) + col] = glv * coef;
        const float ts = wave_sum(term);
        if (lane == 0) part[(slot == 0 ? ng.tile_off[0] : ng.tile_off[1]) + (j0 >> 4)] = ts;
        return;
    }
    if (col >= g.ncols) return;
    const float bias = g.b ? g.b[col] : 0.f;
    float keep[4] = {1.f, 1.f, 1.f, 1.f};
    if (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) keep[r] = dropk[(4 * (lane >> 4) + r) * 16 + (lane & 15)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gn = n0 + 4 * (lane >> 4) + r;
        if (gn < N) {
            float v = acc[r] + bias;
            v = g.relu ? fmaxf(v, 0.f) : v;
            if (DROP) v = v * (keep[r] * dr->scale);
            g.Y[(size_t)gn * g.ldy + col] = v;
        }
    }
}

// An instantiation per K split, at least MOPOE_LIN_MINW waves per SIMD: as ONE kernel with the
// three splits inside and no such bound the compiler spread over 255 VGPRs + 40 AGPRs -- one wave
// per SIMD, one workgroup per CU at a time, and every launch of 512 workgroups ran in two rounds
// (1,024 rows: 16.0 us; the layers of a general topology: 10 us each).
#ifndef MOPOE_LIN_MINW
#define MOPOE_LIN_MINW 2
#endif
template <int KS, int EPI>
DEV void linear_block(const LinArgs& a, const void* ex) {
    constexpr bool DROP = EPI == kEpiDrop;
    const LinDrop* dr = (const LinDrop*)ex;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave id as a provably wave-uniform scalar (guide T20): everything derived
    // from it stays in SGPRs and buffer descriptors need no waterfall loop
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    GSTAMP(a.counters, kCtrStamp + 11, a.counters && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == (a.ngroups > 1 ? 1 : 0) && tid == 0);
    // (the number of the step this launch belongs to, requested here and used in the epilogue: the
    //  launch that begins the step cannot wait for its block 0 -- latent_body's rule, [DONE] + 1)
    uint32_t step_no = 0;
    if (DROP)
        step_no = a.counters ? (uint32_t)__builtin_nontemporal_load(dr->counters + MOPOE_CTR_STEPS_DONE) + 1u
                             : (uint32_t)__builtin_nontemporal_load(dr->counters + MOPOE_CTR_STEPS_BEGUN);
    if (a.counters && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
        step_begin(a.counters, a.num_mods, a.publish ? &a.adam : nullptr);
    }
    const LinGroup& g = a.g[blockIdx.z];
    constexpr int ks = KS;
    if ((int)blockIdx.x * (64 / ks) >= g.ncols) return;
    // source row of each of the tile's 16 batch rows (a gather is resolved once)
    __shared__ int rowsel[kRows];
    const bool gather = g.rows != nullptr;
    if (gather) {
        if (tid < kRows) rowsel[tid] = g.rows[min((int)blockIdx.y * kRows + tid, a.n - 1)];
        __syncthreads();
    }
    GSTAMP(a.counters, kCtrStamp + 13, a.counters && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == (a.ngroups > 1 ? 1 : 0) && tid == 0);
    const int* rs = gather ? rowsel : nullptr;
    linear_tile<KS, EPI>(a, g, lds, rs, tid, lane, wave, ex, step_no);
    GSTAMP(a.counters, kCtrStamp + 12, a.counters && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == (a.ngroups > 1 ? 1 : 0) && tid == 0);
}

template <int KS>
__global__ __launch_bounds__(256, MOPOE_LIN_MINW) void k_linear(const LinArgs a_by_value) {
    (void)a_by_value;  // read in place (see k_latent)
    linear_block<KS, kEpiPlain>(*(const LinArgs*)__builtin_amdgcn_kernarg_segment_ptr(), nullptr);
}
// ... followed by Dropout(p) (a hidden layer of a general topology: mopoe_general.inc)
template <int KS>
__global__ __launch_bounds__(256, MOPOE_LIN_MINW) void k_linear_drop(const LinDropArgs a_by_value) {
    (void)a_by_value;
    const LinDropArgs& a = *(const LinDropArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    linear_block<KS, kEpiDrop>(a.la, &a.d);
}
// ... the output layer of a training step, followed by the likelihood (LinNll)
template <int KS>
__global__ __launch_bounds__(256, MOPOE_LIN_MINW) void k_linear_nll(const LinNllArgs a_by_value) {
    (void)a_by_value;
    const LinNllArgs& a = *(const LinNllArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    linear_block<KS, kEpiNll>(a.la, &a.q);
}

// ---------------------------------------------------------------------------
// k_linear_big: the same product for LARGE batches (>= kLinBigRows rows: folded DAA
// inference, big training batches).  A 16-row tile re-reads its 114 KB of W per
// workgroup (4 flop per byte of L2 traffic: L2-bound at ~35 TFLOP/s); here a
// workgroup owns 64 rows x 64 columns, stages x AND W chunks of 64 k through LDS
// (double buffered: the next chunk is in registers while the current one feeds the
// MFMAs), and each of its four waves accumulates a 16 x 64 strip -> 16 flop per byte.
// grid = (column groups of 64, row tiles of 64, groups), block = 256.
// ---------------------------------------------------------------------------
#ifndef MOPOE_BIGK_BF16
#define MOPOE_BIGK_BF16 64
#endif
#ifndef MOPOE_BIGK
#define MOPOE_BIGK 32
#endif
constexpr int kBigRows = 64, kBigCols = 64, kBigK = MOPOE_BIGK, kBigLd = kBigK + 4;
#ifndef MOPOE_BIG_BUFS
#define MOPOE_BIG_BUFS 1
#endif
constexpr int kBigBufs = MOPOE_BIG_BUFS;   // LDS buffers per operand
constexpr int kLinBigRows = 2048;  // batches from here on use the 64-row tiles

// ROWS = 64: four waves, a 16 x 64 strip each.  (Measured and dropped: ROWS = 32 for batches of
// ~1,000 rows -- twice the workgroups, but a workgroup's chain of fourteen load - park - barrier
// rounds takes 20 us whatever its size: 1,024 rows 20.4 us against 16.0 with the 16-row tiles.)
// BF16 (mopoe_step.gemm_operands, opt-in): x and W are rounded to bfloat16 (nearest even) as
// they are parked -- two k per LDS word, K chunks twice as deep in the same bytes -- and
// multiplied by v_mfma_f32_16x16x16_bf16 with float32 accumulation: one instruction for the
// four exact-f32 ones of a 16-deep K block (lane (c, q) supplies k = kb + 4 q .. + 3 of row /
// column c for both operands: tools/mfma_bf16_probe.hip).
DEV uint32_t bf16_pair(float lo, float hi) {   // two roundings to nearest even, packed (finite inputs)
    uint32_t a = __builtin_bit_cast(uint32_t, lo), b = __builtin_bit_cast(uint32_t, hi);
    a += 0x7FFFu + ((a >> 16) & 1u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xFFFF0000u);
}

#ifndef MOPOE_LB_MINW
#define MOPOE_LB_MINW 1
#endif
template <int ROWS, bool BF16>
__global__ __launch_bounds__(ROWS * 4, MOPOE_LB_MINW) void k_linear_big(const LinArgs a_by_value) {
    (void)a_by_value;  // read in place (see k_latent)
    const LinArgs& a = *(const LinArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int T = ROWS * 4;                  // threads: a wave per 16-row strip
    constexpr int KC = BF16 ? MOPOE_BIGK_BF16 : kBigK; // k per staged chunk
    constexpr int TPR = KC / 4;                  // staging threads per row (a float4 each)
    constexpr int RPP = T / TPR;                 // rows per staging pass
    constexpr int PA = ROWS / RPP, PB = kBigCols / RPP;   // passes over the x rows / the W rows
    static_assert(PA >= 1 && PB >= 1 && ROWS % RPP == 0 && kBigCols % RPP == 0, "staging split");
    constexpr int LD = BF16 ? KC / 2 + 2 : KC + 4;        // LDS words per row (bf16: two k per word)
    __shared__ __attribute__((aligned(16))) float As[kBigBufs][ROWS * LD];
    __shared__ __attribute__((aligned(16))) float Bs[kBigBufs][kBigCols * LD];
    __shared__ int rowsel[ROWS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, q = lane >> 4;
    const int N = a.n;
    if (a.counters && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
        step_begin(a.counters, a.num_mods, a.publish ? &a.adam : nullptr);
    }
    const LinGroup& g = a.g[blockIdx.z];
    // Workgroups reach the XCDs round robin in launch order (column tile fastest): the four
    // column tiles of a row tile would sit on four XCDs and each of their L2s would fetch the
    // same x rows (measured at 65,536 rows: 2.6 x the algorithmic bytes).  Tile p = (L % 8) *
    // (tiles / 8) + L / 8 of launch index L gives an XCD a run of consecutive tiles -- all
    // column tiles of its row tiles -- so an x tile is fetched into ONE L2.  (Speed only.)
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.xcd_order) {
        const int gx = gridDim.x, per = (gx * (int)gridDim.y) >> 3, l = bx + gx * by;
        if (l < 8 * per) {
            const int p = (l & 7) * per + (l >> 3);
            by = p / gx;
            bx = p - by * gx;
        }
    }
    const int j0 = bx * kBigCols, n0 = by * ROWS;
    if (j0 >= g.ncols) return;
    const int K = g.K;
    if (tid < ROWS) {
        const int gn = min(n0 + tid, N - 1);
        rowsel[tid] = g.rows ? g.rows[gn] : gn;
    }
    __syncthreads();
    const rsrc_t xr = make_rsrc(g.X, (size_t)g.xrows * g.ldx * sizeof(float));
    const rsrc_t wr = make_rsrc(g.W, (size_t)g.ncols * K * sizeof(float));
    const bool vec = K % 4 == 0;
    // staging: thread -> PA float4 of the x chunk and PB of the W chunk (rows r0 + RPP i)
    const int r0 = tid / TPR, k4 = (tid % TPR) * 4;
    f32x4 xa[PA], wb[PB];
    auto fetch = [&](int kc) __attribute__((always_inline)) {
        const int k = kc + k4;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = r0 + RPP * i;
            const uint32_t xo = (uint32_t)(rowsel[r] * g.ldx + k) * 4u;
            const bool rv = n0 + r < N;
            if (vec) {
                xa[i] = ldg4(xr, guard(xo, rv & (k < K)));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) xa[i][e] = ldg(xr, guard(xo + 4u * e, rv & (k + e < K)));
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int r = r0 + RPP * i;
            const uint32_t wo = (uint32_t)((j0 + r) * K + k) * 4u;   // row >= ncols: out of range
            if (vec) {
                wb[i] = ldg4(wr, guard(wo, k < K));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) wb[i][e] = ldg(wr, guard(wo + 4u * e, k + e < K));
            }
        }
    };
    auto park = [&](int buf) __attribute__((always_inline)) {
        if constexpr (BF16) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const u32x2 v = {bf16_pair(xa[i][0], xa[i][1]), bf16_pair(xa[i][2], xa[i][3])};
                *reinterpret_cast<u32x2*>(&As[buf][(r0 + RPP * i) * LD + k4 / 2]) = v;
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const u32x2 v = {bf16_pair(wb[i][0], wb[i][1]), bf16_pair(wb[i][2], wb[i][3])};
                *reinterpret_cast<u32x2*>(&Bs[buf][(r0 + RPP * i) * LD + k4 / 2]) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                *reinterpret_cast<f32x4*>(&As[buf][(r0 + RPP * i) * LD + k4]) = xa[i];
#pragma unroll
            for (int i = 0; i < PB; ++i)
                *reinterpret_cast<f32x4*>(&Bs[buf][(r0 + RPP * i) * LD + k4]) = wb[i];
        }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4] = {zero4, zero4, zero4, zero4};
    fetch(0);
    park(0);
    __syncthreads();
    const int nchunks = cdiv(K, KC);
    for (int c = 0; c < nchunks; ++c) {
        const int cur = kBigBufs == 1 ? 0 : c & 1;
        if (c + 1 < nchunks) fetch((c + 1) * KC);   // in flight under the MFMAs
        const float* Aw = &As[cur][(wave * 16) * LD];
        if constexpr (BF16) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int kb = 0; kb < KC; kb += 16) {
                const int o = c16 * LD + kb / 2 + 2 * q;   // (four consecutive k: two words)
                const s16x4 av = *reinterpret_cast<const s16x4*>(Aw + o);
                s16x4 bv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[t] = *reinterpret_cast<const s16x4*>(&Bs[cur][(16 * t) * LD + o]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv[t], acc[t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < KC; kb += 16) {
                const f32x4 av = lds_a4(Aw, LD, kb, lane);
                f32x4 bv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[t] = lds_a4(&Bs[cur][(16 * t) * LD], LD, kb, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = mfma_16x16x4(av[i], bv[t][i], acc[t]);
            }
        }
        if (c + 1 < nchunks) {
            if constexpr (kBigBufs == 1) {
                __syncthreads();   // (every wave is done reading the one buffer)
                park(0);
            } else {
                park(cur ^ 1);   // the other buffer was last read before the previous barrier
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = j0 + 16 * t + c16;
        if (col >= g.ncols) continue;
        const float bias = g.b ? g.b[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gn = n0 + wave * 16 + 4 * q + r;
            if (gn < N) {
                const float v = acc[t][r] + bias;
                g.Y[(size_t)gn * g.ldy + col] = g.relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
}

#include "mopoe_latent.inc"

// ---------------------------------------------------------------------------
// k_fused: encoder layer AND the per-sample chain in one launch, for small training
// batches.  Blocks [0, nlin) are the encoder layer (one 16-row x 64-column tile of one
// modality each, K cut over the four waves of a column tile), blocks [nlin, ..) are
// the row groups of k_latent.  A kernel boundary between the two costs ~2 us on the
// device plus the ~3.5 us the first loads of a fresh kernel take; here a row group
// has its weight fragments, x tiles and descriptors in flight while its rows of h are
// being computed, and picks them up one memory round trip after the last producer's
// flag.  Producers never wait for anything and sit at the LOW block indices, so the
// wait of a consumer cannot deadlock whatever the residency (and it is bounded).
// ---------------------------------------------------------------------------
// What a block needs to find its work sits in the FIRST 64-byte line of the argument
// block: one scalar-cache miss, not a chain of them, stands between a producer's entry
// and its first load.
struct alignas(64) FHead {
    int32_t nlin;        // encoder-layer blocks
    int32_t row_tiles;   // 16-row tiles = row groups
    int32_t ks;          // 4: K over four waves per column tile; 1: a wave per column tile
    int32_t producers;   // encoder-layer blocks per row tile (what a row group waits for)
    int32_t begin[MOPOE_MAX_MODS + 1];  // first block of encoder group z
    int32_t tiles[MOPOE_MAX_MODS];      // 16-column tiles per block of group z (ks == 4: 4, 2,
                                        // or 16 = one 256-column block, K <= 16 unsplit)
    int32_t gpt;         // row groups per 16-row tile: 1, or 4 (four-row groups)
};
static_assert(sizeof(FHead) == 64, "one scalar cache line");
struct FArgs {
    FHead hd;
    KArgs ka;
    LinArgs la;
};

// KS = 4: 64 columns per block, K over four waves (few row tiles); KS = 1: 256 columns
// per block, a wave per column tile (more row tiles than the grid could hold otherwise)
// A block's MFMA time is its column tiles x K: with 64 columns of a 444-wide modality a
// CU issues 444 MFMAs (1.5 us) while the blocks of a 7-wide one issue 8 -- so a wide
// modality gets blocks of kTiles = 2 tiles (32 columns): twice the blocks, half the
// chain, same K parts and the same summation order (the waves beyond KS * kTiles only
// help staging the x tile).
template <int KS, int kTiles = kLatentWaves / KS>
DEV void linear_block16(const LinArgs& a, const LinGroup& g, float* lds, int rt, int cg,
                        int tid, int lane, int wave, int32_t* flag, int slot) {
    constexpr int CH = 8, kStage = 2, kCols = 16 * kTiles;
    static_assert(KS * kTiles <= kLatentWaves, "a wave per (tile, K part)");
    const int N = a.n, K = g.K;
    // (W is a segment of the flat parameter buffer, its bias follows it: 4-wide reads of a
    //  row whose length is not a multiple of 4 stay inside the buffer -- g.wslack is set by
    //  the only caller; the per-element form of linear_tile and its sixteen precomputed
    //  offsets per fragment batch, which the register allocator spilled, are not needed here)
    const int n0 = rt * kRows;
    const bool busy = wave < KS * kTiles;      // (wave-uniform)
    const int tile = busy ? wave / KS : 0, part = wave % KS;
    const int j0 = (cg * kTiles + tile) * 16;
    int* rowsel = reinterpret_cast<int*>(lds);
    float* xt = lds + kRows;
    const bool gather = g.rows != nullptr;
    // (diagnostic build: the phases of one producer of the wide modality, row tile 0)
    const bool pstamp = a.counters && rt == 0 && cg == 0 && slot == a.ngroups - 1 && tid == 0;
    (void)pstamp;
    GSTAMP(a.counters, kCtrStamp + 32, pstamp);
    if (gather) {
        if (tid < kRows) rowsel[tid] = g.rows[min(n0 + tid, N - 1)];
        __syncthreads();
    }
    const rsrc_t xr = make_rsrc(g.X, (size_t)g.xrows * g.ldx * sizeof(float));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    for (int kc0 = 0; kc0 < K; kc0 += kEncKChunk) {
        const int Kc = min(kEncKChunk, K - kc0);
        const int Kp = round_up(Kc, 16);
        const int ldx = Kp + 4;
        const int kend = (busy & (j0 < g.ncols)) ? Kp : 0;
        const rsrc_t wr = make_rsrc(g.W + kc0, ((size_t)g.ncols * K + (g.wslack ? 3 : 0)) * sizeof(float));
        auto load_w = [&](int i0, f32x4 (&b)[CH]) __attribute__((always_inline)) {
            if (!busy) return;
#pragma unroll
            for (int c = 0; c < CH; ++c)  // k >= Kc is out of range -> 0
                b[c] = glb_b4_nt<true>(wr, K, Kc, j0, 16 * (part + KS * (i0 + c)), lane);
        };
        f32x4 b[CH];
        if (kc0 > 0) __syncthreads();
        // staging: wave w owns row w of the tile (16 waves = 16 rows), its lanes
        // consecutive float4 of it -- no index arithmetic before the loads go out,
        // one fully coalesced kilobyte per wave-instruction
        static_assert(kLatentWaves == kRows && kEncKChunk <= kStage * 4 * kWave, "one row per wave");
        {
            f32x4 v[kStage];
            const int r = wave;
            const bool rv = n0 + r < N;
            const int row = gather ? rowsel[r] : min(n0 + r, N - 1);
#pragma unroll
            for (int i = 0; i < kStage; ++i) {
                const int k = (i * kWave + lane) * 4;
                const uint32_t base = (uint32_t)(row * g.ldx + kc0 + k) * 4u;
                v[i] = ldg4_row(xr, base, rv, k, Kc);
            }
            load_w(0, b);  // behind the x loads (loads return in order)
            GSTAMP(a.counters, kCtrStamp + 33, pstamp);
#pragma unroll
            for (int i = 0; i < kStage; ++i) {
                const int k = (i * kWave + lane) * 4;
                if (k < Kp) *reinterpret_cast<f32x4*>(xt + r * ldx + k) = v[i];
            }
        }
        __syncthreads();
        GSTAMP(a.counters, kCtrStamp + 34, pstamp);
        for (int i0 = 0; 16 * (part + KS * i0) < kend; i0 += CH) {
            f32x4 bn[CH];
            const bool more = 16 * (part + KS * (i0 + CH)) < kend;  // wave-uniform
            if (more) load_w(i0 + CH, bn);
            f32x4 av[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int k = 16 * (part + KS * (i0 + c));
                av[c] = lds_a4(xt, ldx, k < kend ? k : 0, lane);
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                acc = mfma_16x16x4(av[c][0], b[c][0], acc);
                acc2 = mfma_16x16x4(av[c][1], b[c][1], acc2);
                acc = mfma_16x16x4(av[c][2], b[c][2], acc);
                acc2 = mfma_16x16x4(av[c][3], b[c][3], acc2);
            }
            if (more) {
#pragma unroll
                for (int c = 0; c < CH; ++c) b[c] = bn[c];
            }
        }
    }
    acc += acc2;
    GSTAMP(a.counters, kCtrStamp + 35, pstamp);
    // the bias of the four output columns this thread finishes: requested here, behind the
    // MFMAs (their registers are free now) and ahead of the partial tiles' trip through
    // LDS -- a load at the point of use would put a memory round trip in front of the
    // hand-off
    f32x4 bias4;
    {
        const int fcol = cg * kCols + (tid % (kCols / 4)) * 4;
        const rsrc_t br = make_rsrc(g.b, g.b ? (size_t)g.ncols * sizeof(float) : 0);
        bias4 = ldg4(br, guard((uint32_t)fcol * 4u, (tid < kRows * kCols / 4) & (fcol < g.ncols)));
    }
    // All four K parts park their partial tiles in LDS as [row][column] (behind the x
    // tile); 256 threads then add the parts in fixed order, four consecutive columns
    // each, and write h with ONE 16-byte write-through store (the consumer group reads
    // it from memory; 4-byte write-through stores cost several times more per byte).
    constexpr int kLdR = kCols + 4;
    float* red = xt + kRows * (round_up(min(K, kEncKChunk), 16) + 4);
    if (busy) {
        const int c16 = lane & 15, q = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            red[(part * kRows + 4 * q + r) * kLdR + 16 * tile + c16] = acc[r];
    }
    __syncthreads();
    GSTAMP(a.counters, kCtrStamp + 36, pstamp);
    if (tid < kRows * kCols / 4) {
        const int row = tid / (kCols / 4), c4 = (tid % (kCols / 4)) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(red + row * kLdR + c4);
#pragma unroll
        for (int p = 1; p < KS; ++p)  // fixed order
            v += *reinterpret_cast<const f32x4*>(red + (p * kRows + row) * kLdR + c4);
        const int col = cg * kCols + c4, gn = n0 + row;
        if (col < g.ncols && gn < N) {   // (ncols is a multiple of 4 here: 256)
            v += bias4;   // (zeros without a bias)
            if (g.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            const rsrc_t yr = make_rsrc(g.Y, (size_t)N * g.ldy * sizeof(float));
            stg4_wt(yr, (uint32_t)(gn * g.ldy + col) * 4u, v);
        }
    }
    // (diagnostic build: when each producer of row tile 0 has issued its stores;
    //  tools/fused_handoff_timeline.py -- needs a counters buffer of 32 words)
    GSTAMP(a.counters, kCtrStamp + 16 + slot * 4 + cg, a.counters && rt == 0 && tid == 0);
    // hand-off: every storing wave drains its stores, then ONE lane signals
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GSTAMP(a.counters, kCtrStamp + 37, pstamp);
    __syncthreads();
    // one add per row group that reads these rows (four-row groups: up to four of them,
    // their flags a slab apart)
    if (tid < a.sig_n && rt * a.sig_n + tid < a.sig_groups)
        __hip_atomic_fetch_add(flag + (size_t)tid * a.sig_stride, 1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    GSTAMP(a.counters, kCtrStamp + 38, pstamp);
}

// Rebuilds the fragment-major copies of the head and decoder weights (WFrag) from the
// parameters: a 16-byte piece W[r][4 k4 ..] moves to WF[r / 64][k4][r % 64].  The copies
// follow the parameters by themselves wherever THIS library rewrites them (the Adam
// epilogue of k_wgrad stores both; mopoe_adam_step and the exchanging update run this
// kernel behind theirs); after any other writer -- an initialisation, a checkpoint, a
// broadcast -- mopoe_wfrag_refresh runs it.
__global__ __launch_bounds__(256) void k_wfrag(const mopoe_model mdl, const float* __restrict__ params,
                                               float* __restrict__ wfrag) {
    const WFrag wf = wfrag_layout(mdl);
    const rsrc_t pr = make_rsrc(params, (size_t)mdl.num_floats * sizeof(float));
    const rsrc_t wr = make_rsrc(wfrag, (size_t)wf.total * sizeof(float));
    const int t0 = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;
    for (int m = 0; m < mdl.num_mods; ++m) {
        const int nh = heads_dim(mdl, m), zd = z_dim(mdl, m), dm = mdl.input_dim[m], k4d = wf.k4d[m];
        for (int p = t0; p < nh * (kHid / 4); p += nthr) {
            const int r = p >> 6, k4 = p & 63;
            const f32x4 v = ldg4(pr, (uint32_t)(mdl.off_wh[m] + r * kHid + 4 * k4) * 4u);
            stg4_wt(wr, (uint32_t)(wf.whf[m] + wfrag_piece(r, k4, kHid / 4)) * 4u, v);
        }
        for (int p = t0; p < dm * k4d; p += nthr) {
            const int r = p / k4d, k4 = p - r * k4d;
            f32x4 v = ldg4(pr, (uint32_t)(mdl.off_wd[m] + r * zd + 4 * k4) * 4u);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 4 * k4 + e < zd ? v[e] : 0.f;   // (the next row's start)
            stg4_wt(wr, (uint32_t)(wf.wdf[m] + wfrag_piece(r, k4, k4d)) * 4u, v);
        }
    }
}

template <int FORM>
__global__ __launch_bounds__(kLatentThreads) void k_fused(const FArgs f_by_value) {
    (void)f_by_value;  // read in place (see k_latent)
    const FArgs& f = *(const FArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x;
    const FHead& hd = f.hd;   // (read in place: one scalar-cache line)
    // the lines of the encoder layer's argument block, requested together with the head
    // line: a producer's group record is then a scalar-cache hit, not a second miss
    uint32_t karg_sink = 0;
    {
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        for (int line = w; line < (int)(sizeof(LinArgs) / 64); line += kLatentWaves)
            karg_sink |= ((const uint32_t*)&f.la)[line * 16];
    }
    if (b < hd.nlin) {
#ifdef MOPOE_KNOCK
        if ((f.la.knock >> 14) & 1) return;   // (diagnostic: producers leave at once)
#endif
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const LinArgs& a = f.la;
        // (the step number and the Adam records are looked after by row group 0 while it
        //  waits for its producers: step_begin has no business on a producer's path)
        const int z = find_seg<MOPOE_MAX_MODS>(hd.begin, b);
        const int local = b - hd.begin[z];
        const int stride = f.ka.lds.part_stride;
        const int tl = hd.tiles[z];   // 16-column tiles per block: 16 (one 256-column block), 8, 4 or 2
        if (tl == kLatentWaves) {     // K <= 16, or a grid that only fits with one block per row tile
            const int rt = local;
            int32_t* flag = reinterpret_cast<int32_t*>(f.ka.buf.partials + (size_t)rt * hd.gpt * stride + kHandoffWord);
            linear_block16<1>(a, a.g[z], lds, rt, 0, tid, lane, wave, flag, z);
        } else {
            const int ncg = kLatentWaves / tl;
            // (An XCD-aware map of the (row tile, column group) grid -- 4 x 2 over the classes
            //  b % 8 -- takes 0.45 MB of counted fetch off the launch and costs 0.15 us, the
            //  next kernel then finding its operands in other XCDs' L2s: measured, not kept.)
            const int cg = local % ncg, rt = local / ncg;
            int32_t* flag = reinterpret_cast<int32_t*>(f.ka.buf.partials + (size_t)rt * hd.gpt * stride + kHandoffWord);
            if (tl == 2)
                linear_block16<4, 2>(a, a.g[z], lds, rt, cg, tid, lane, wave, flag, z);
            else if (tl == 4)
                linear_block16<4, 4>(a, a.g[z], lds, rt, cg, tid, lane, wave, flag, z);
            else   // 128-column blocks, K over two waves (mid-size batches)
                linear_block16<2, 8>(a, a.g[z], lds, rt, cg, tid, lane, wave, flag, z);
        }
        asm volatile("" ::"s"(karg_sink));
        return;
    }
    asm volatile("" ::"s"(karg_sink));
#ifdef MOPOE_KNOCK
    if ((f.la.knock >> 15) & 1) return;       // (diagnostic: row groups leave at once)
#endif
    const int grp = b - hd.nlin;
    int32_t* flag = reinterpret_cast<int32_t*>(f.ka.buf.partials + (size_t)grp * f.ka.lds.part_stride + kHandoffWord);
    latent_body<true, FORM>(f.ka, lds, grp, flag, hd.producers, &f.la);
}

// Scalars of the step from the row tiles' partial sums, in a fixed order
// (run_epochs.py:89-128, mm_div.py:92-111, kl_div.py:7-14; utils/TBLogger.py:26-37 for
// the latent means).  One block; one thread per scalar -- a single thread walking the
// descriptor arrays pays one scalar-memory round trip per element and took ~11 us.
// sum of q[t * stride], t0 <= t < t1, in a fixed order: sixteen interleaved accumulators,
// sixteen loads in flight (a serial walk pays a trip to the L2 per element)
DEV float strided_sum(const float* __restrict__ q, int t0, int t1, size_t stride) {
    float s[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = 0.f;
    for (int t = t0; t < t1; t += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = q[(size_t)min(t + k, t1 - 1) * stride];
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] += t + k < t1 ? v[k] : 0.f;
    }
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int k = 0; k < w; ++k) s[k] += s[k + w];
    return s[0];
}

template <int THREADS>  // block size
DEV void finalize_stats(const KArgs& a, int tid) {
    constexpr int SLICES = THREADS / kStatStride;
    static_assert(SLICES >= 1, "a thread per partial index");
    __shared__ float slab[SLICES][kStatStride];
    __shared__ float kld[kStatStride];      // scalar values, by partial index
    __shared__ float contrib[kStatStride];  // their share of total_loss
    __shared__ float jdc[MOPOE_MAX_SUBSETS];
    const mopoe_buffers& buf = a.buf;
    const mopoe_step& st = a.st;
    const int tiles = a.lds.fold_tiles > 0 ? a.lds.fold_tiles : cdiv(st.n, a.lds.rows);
    const int stride = a.lds.part_stride;
    if (tid < SLICES * kStatStride) {
        // partial index p = tid % kStatStride, tile slice = tid / kStatStride; slices
        // summed in order, sixteen loads in flight per thread (a 50,000-row forward has
        // 3125 row groups to add up).
        const int p = tid % kStatStride, sl = tid / kStatStride;
        const int per = cdiv(tiles, SLICES);
        const int t1 = min((sl + 1) * per, tiles);
        slab[sl][p] = p < kNumPart ? strided_sum(buf.partials + p, sl * per, t1, (size_t)stride) : 0.f;
    }
    __syncthreads();
    const float fn = (float)st.n;
    if (tid < kNumPart) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < SLICES; ++k) sum += slab[k][tid];
        float v = 0.f, c = 0.f;
        int out = -1;
        if (tid < kPartKlStyle) {                       // KL of subset s
            const int s = tid;
            out = MOPOE_STAT_KLD_SUBSET + s;
            if (s < st.num_subsets && st.sub_avail[s]) {
                v = -0.5f * sum / fn;
                c = st.sub_kl_coef[s] * v;
            }
        } else if (tid < kPartNll) {                    // KL of style m
            const int m = tid - kPartKlStyle;
            out = MOPOE_STAT_KLD_STYLE + m;
            if (m < a.mdl.num_mods && ((st.present_mask >> m) & 1) && a.mdl.style_dim[m] > 0) {
                v = -0.5f * sum / fn;
                c = st.style_kl_coef[m] * v;
            }
        } else if (tid < kPartMean) {                   // NLL of decoder job j
            const int j = tid - kPartNll;
            out = MOPOE_STAT_NLL + j;
            if (j < st.num_jobs) {
                v = sum / fn;
                c = st.job_nll_coef[j] * v;
            }
        } else {                                        // mean of an encoder output
            const int m = (tid - kPartMean) >> 2, k = (tid - kPartMean) & 3;
            out = MOPOE_STAT_LATENT_MEAN + (tid - kPartMean);
            const int cols = k < 2 ? a.mdl.style_dim[m < a.mdl.num_mods ? m : 0] : a.mdl.class_dim;
            if (m < a.mdl.num_mods && ((st.present_mask >> m) & 1) && cols > 0)
                v = sum / (fn * (float)cols);           // (no share of the loss)
        }
        kld[tid] = v;
        contrib[tid] = c;
        buf.stats[out] = v;
        if (buf.stats_host) buf.stats_host[out] = v;
    }
    __syncthreads();
    if (tid < MOPOE_MAX_SUBSETS)
        jdc[tid] = tid < st.num_comp ? st.comp_w[tid] * kld[kPartKlSub + st.comp_sub[tid]] : 0.f;
    __syncthreads();
    if (tid == 0) {
        float total = 0.f, jd = 0.f;
#pragma unroll
        for (int i = 0; i < kPartMean; ++i) total += contrib[i];
#pragma unroll
        for (int k = 0; k < MOPOE_MAX_SUBSETS; ++k) jd += jdc[k];
        buf.stats[MOPOE_STAT_TOTAL_LOSS] = total;
        buf.stats[MOPOE_STAT_JOINT_DIV] = jd;
        if (buf.stats_host) {
            buf.stats_host[MOPOE_STAT_TOTAL_LOSS] = total;
            buf.stats_host[MOPOE_STAT_JOINT_DIV] = jd;
        }
    }
}

// ---------------------------------------------------------------------------
// k_wgrad: every weight / bias gradient is G^T X reduced over the batch rows.
//   W1_m: G = g_pre_m (n,256)      X = x_m   (n,d_m)   (+ bias column)
//   Wh_m: G = g_heads_m (n,nh_m)   X = h_m   (n,256)   (+ bias column)
//   Wd_m: G = g_xhat_m (R_m,d_m)   X = z_m   (R_m,zd)  (+ bias column)
// One workgroup (4 waves) per 16x16 output tile; the waves split the row
// range and are summed in fixed order through LDS (deterministic).  The bias
// gradient is the column of an implicit all-ones feature appended to X.
// Extra blocks reduce the decoder-logvar partials and finalise the scalars.
// With adam.lr != 0 the Adam update is applied in the epilogue.
// ---------------------------------------------------------------------------
struct WJob {
    const float* G;
    const float* X;
    const int32_t* xrows;  // gather index for X rows or nullptr
    int32_t ldg, gcols, ldx, xcols, R;
    int32_t off_w, off_b;
    int32_t tiles_j, tile_begin;
    int32_t mod;           // modality of the parameters (its own Adam step count)
    int32_t xtotal;        // rows of X (the descriptor covers exactly xtotal * ldx floats)
    int32_t wf_off, wf_k4; // fragment-major copy of this weight (WFrag): float offset or -1, K/4
    int32_t fold;          // the row length is a multiple of 32: the bias gradient (column sums of G)
                           // is added up by the lanes next to the MFMAs and parked in the spare
                           // column of the partial tiles, instead of costing a column tile of its
                           // own for the implicit ones column (24 of 277 blocks at configs[4])
    int32_t th;            // output rows per tile: 32, or 16 for a job with twice the batch rows of
                           // the others (method poe's decoder: joint + unimodal pass) -- a block's
                           // time is its MFMA count, rows x tile area, and the launch is as long as
                           // its longest block
};

struct WArgs {
    int32_t njobs;
    int32_t total_tiles;   // GEMM tiles
    int32_t lvo_blocks;    // blocks after the GEMM tiles
    int32_t fuse_adam;
    int32_t tile_begin[3 * MOPOE_MAX_MODS + 1];
    WJob jobs[3 * MOPOE_MAX_MODS];
    int32_t lvo_block_begin[MOPOE_MAX_MODS + 1];
    mopoe_adam adam;
    XgPeers xg;            // k_wgrad<.., true>: the exchange between backward and update
};

// One wave's share of a 32x32 block of G^T X: batch rows [rbeg, rend) in rounds
// of 64.  Both operands are read 8 bytes per lane (two adjacent columns of one
// batch row), all 32 loads of a round in flight together, from guarded offsets
// (no conditional loads).  Lane (c = lane&15, q = lane>>4), MFMA step s of a
// round covers batch rows rb+4s .. rb+4s+3 (row rb+4s+q in this lane):
//   A fragments  G[r][i0 + 2c + ti]   -> output rows    i = i0 + 2*(4q'+reg) + ti
//   B fragments  X[r][j0 + 2c + tj]   -> output columns j = j0 + 2c + tj
// (column-interleaved 16x16 tiles, as in k_latent).  The bias gradient is the
// column of an implicit all-ones feature at j == xcols.
// TI = 1: a 16-row tile -- one A fragment, G[r][i0 + c] -> output rows i = i0 + 4q' + reg.
// FOLD: also add up the column sums of G (WJob::fold) -- an instantiation of its own: two adds per
// MFMA step cost the jobs that do not need them 5 % at 65,536 rows.
// PIPE: a wave with more than one round (1,024 rows over 8 waves: two) requests round k + 1
// before it issues the MFMAs of round k, so one load round trip (~1.5-2 us under the launch's
// own traffic) instead of one per round sits in front of the MFMAs.  Every round's requests
// are guarded offsets, so the round behind the last one is requested too and reads nothing.
template <bool GATHER, int STEPS, int TI, bool FOLD, bool PIPE = false>  // STEPS MFMA steps (4 batch rows each) per round: 16 or 8
DEV void wgrad_rows(rsrc_t gr, rsrc_t xr, const int32_t* __restrict__ xrows, int ldg_, int ldx,
                    int gcols, int xcols, int i0, int j0, int rbeg, int rend, int lane,
                    f32x4 (&acc)[2][2], float (&bsum)[2]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int c = lane & 15, q = lane >> 4;
    const int ci = i0 + TI * c, cj = j0 + 2 * c;
    // per-element validity of the two columns this lane reads (bit masks)
    const uint32_t am0 = ci < gcols ? 0xFFFFFFFFu : 0u, am1 = ((TI == 2) & (ci + 1 < gcols)) ? 0xFFFFFFFFu : 0u;
    const uint32_t bm0 = cj < xcols ? 0xFFFFFFFFu : 0u, bm1 = cj + 1 < xcols ? 0xFFFFFFFFu : 0u;
    const float one0 = cj == xcols ? 1.f : 0.f, one1 = cj + 1 == xcols ? 1.f : 0.f;
    auto request = [&](int rb, f32x2 (&av)[STEPS], f32x2 (&bv)[STEPS]) {
        int xrow[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int r = rb + 4 * s + q;
            xrow[s] = GATHER ? xrows[min(r, rend - 1)] : r;
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int r = rb + 4 * s + q;
            const bool rv = r < rend;
            if constexpr (TI == 2) {
                av[s] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(
                            gr, guard((uint32_t)(r * ldg_ + ci) * 4u, rv & (ci < gcols)), 0, 0));
            } else {
                av[s][0] = ldg(gr, guard((uint32_t)(r * ldg_ + ci) * 4u, rv & (ci < gcols)));
                av[s][1] = 0.f;
            }
            bv[s] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        xr, guard((uint32_t)(xrow[s] * ldx + cj) * 4u, rv & (cj < xcols)), 0, 0));
        }
    };
    auto multiply = [&](int rb, const f32x2 (&av)[STEPS], const f32x2 (&bv)[STEPS]) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const bool rv = rb + 4 * s + q < rend;
            // columns past the end of a row hold the next row's values: mask them
            const float a0 = av[s][0], a1 = av[s][1], b0 = bv[s][0], b1 = bv[s][1];
            const float fa0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a0) & am0);
            const float fa1 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a1) & am1);
            const float fb0 =
                __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, b0) & bm0) + (rv ? one0 : 0.f);
            const float fb1 =
                __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, b1) & bm1) + (rv ? one1 : 0.f);
            if constexpr (FOLD) {
                bsum[0] += fa0;   // (column sums of G: the bias gradient of a folded job; rows past
                bsum[1] += fa1;   //  the end were read as zeros)
            }
            acc[0][0] = mfma_16x16x4(fa0, fb0, acc[0][0]);
            acc[0][1] = mfma_16x16x4(fa0, fb1, acc[0][1]);
            if constexpr (TI == 2) {
                acc[1][0] = mfma_16x16x4(fa1, fb0, acc[1][0]);
                acc[1][1] = mfma_16x16x4(fa1, fb1, acc[1][1]);
            }
        }
    };
    if constexpr (PIPE) {
        f32x2 a0[STEPS], b0[STEPS], a1[STEPS], b1[STEPS];
        request(rbeg, a0, b0);
        // two half rounds per trip, each buffer refilled right after its values went into the
        // MFMAs (no register copies: a copy out of a load's destination waits for the load).  An
        // odd count of half rounds multiplies one buffer of zeros at the end (guarded requests
        // past the last row return zeros; adding 0 x 0 leaves the sums as they are).
        for (int rb = rbeg; rb < rend; rb += 8 * STEPS) {
            request(rb + 4 * STEPS, a1, b1);
            multiply(rb, a0, b0);
            request(rb + 8 * STEPS, a0, b0);
            multiply(rb + 4 * STEPS, a1, b1);
        }
    } else {
        for (int rb = rbeg; rb < rend; rb += 4 * STEPS) {
            f32x2 av[STEPS], bv[STEPS];
            request(rb, av, bv);
            multiply(rb, av, bv);
        }
    }
}

#ifndef MOPOE_WGRAD_PIPE
#define MOPOE_WGRAD_PIPE 2
#endif
#ifndef MOPOE_WGRAD_W16
#define MOPOE_WGRAD_W16 1
#endif
constexpr int kWgLd = 36;  // leading dim of a 32x32 partial block in LDS

// kWgWaves waves split the batch rows of a block: 8 for large batches (twice as fast at
// N >= 1024) and from 256 rows on when every block has a CU to itself, 4 otherwise (the
// launch code decides).
//
// XG (data-parallel replicas, mopoe_comm_train_step): between the block's gradient and
// its Adam update sits the exchange of mopoe_xgmi.inc, per block: the 32x32 block (and
// its bias column) is pushed into every peer's inbox at its place in the flat buffer,
// the block's flag is raised at every peer, the peers' copies are awaited and added in
// rank order.  The N-rank step then has the same two launches as the one-rank step.
// LEAN (eight waves): half rounds only and at most 128 registers, so that TWO blocks share a CU
// -- for launches with more blocks than CUs (four modalities; the general chain's ten jobs),
// where the 166-register form needs a second round of blocks
template <int kWgWaves, bool XG, bool LEAN = false>
__global__ __launch_bounds__(kWgWaves * 64, LEAN ? 4 : 1) void k_wgrad(const KArgs a_by_value,
                                                                       const WArgs w_by_value) {
    __shared__ __attribute__((aligned(16))) float blk[kWgWaves][32 * kWgLd];
    (void)a_by_value;  // both argument blocks are read in place (see k_latent)
    (void)w_by_value;
    static_assert(sizeof(KArgs) % alignof(WArgs) == 0, "WArgs follows KArgs without padding");
    const char* kargs = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    const KArgs& a = *(const KArgs*)kargs;
    const WArgs& w = *(const WArgs*)(kargs + sizeof(KArgs));
    const mopoe_buffers& buf = a.buf;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform (T20)
    // the block's tables (tile -> job record -> pointers) into the scalar cache
    uint32_t karg_sink = 0;
    for (int line = wave; line < (int)(sizeof(WArgs) / 64); line += kWgWaves)
        karg_sink |= ((const uint32_t*)(kargs + sizeof(KArgs)))[line * 16];
    // (An XCD-aware order of the GEMM tiles -- an XCD's blocks a compact patch of a job's output,
    //  its operand panels in ONE L2 -- was built and measured in round 4: no faster here, and the
    //  fused launch behind it 1.3 us slower at 1,024 rows (the updated weights then sit in other
    //  XCDs' L2s than the ones its producers run on): profiles/r04_c_ab_k_wgrad_variants.txt.)
    const int b = blockIdx.x;
#if defined(MOPOE_FENCE_PROBE) && MOPOE_FENCE_PROBE >= 2
    // (diagnostic build: ... and what the matching acquire would cost every weight-gradient block)
    if (wave == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
#endif
    const bool fuse = w.fuse_adam != 0;
    const bool stamp_blk = b == 20 && tid == 0;  // a W1 block of the large modality
    (void)stamp_blk;
    GSTAMP(buf.stats, kStampBase + 40, stamp_blk);

    if (b < w.total_tiles) {
        // job of this block: compares against one contiguous table (a scan over
        // the job structs would be a chain of dependent scalar loads)
        int ji = 0;
#pragma unroll
        for (int k = 1; k < 3 * MOPOE_MAX_MODS; ++k) ji += (k < w.njobs) & (b >= w.tile_begin[k]);
        const WJob& job = w.jobs[ji];
        const int t = b - job.tile_begin;
        const int th = job.th;   // (wave-uniform: 32 or 16)
        const int i0 = (t / job.tiles_j) * th, j0 = (t % job.tiles_j) * 32;
        const int R = job.R, gcols = job.gcols, xcols = job.xcols;

        // epilogue ownership: thread -> output row i0 + tid/8, columns j0 + 4*(tid%8) ..+3
        const bool epi = tid < 256;  // the first four waves own the 32x32 outputs
        const int ei = i0 + ((tid & 255) >> 3), ej = j0 + 4 * (tid & 7);
        const bool erow = epi & (ei < gcols) & (ei < i0 + th);
        const int nvalid = erow ? min(xcols - ej, 4) : 0;   // weight columns
        // (the bias word: with the ones column in this thread's four -- or, folded job, with
        //  the thread that holds the row's last four weights)
        const bool fold = job.fold != 0;
        const bool has_b = erow & (fold ? ej + 4 == xcols : (xcols >= ej) & (xcols < ej + 4)) & (job.off_b >= 0);
        const int widx = job.off_w + ei * xcols + ej;
        const int bidx = job.off_b + ei;
        const size_t pbytes = (size_t)a.mdl.num_floats * sizeof(float);
        const rsrc_t rp = make_rsrc(buf.params, pbytes);
        const rsrc_t rm = make_rsrc(buf.exp_avg, pbytes);
        const rsrc_t rv = make_rsrc(buf.exp_avg_sq, pbytes);
        // Adam operands are requested first: they arrive while the GEMM runs.
        // (16-byte reads; elements past the row end are neighbours, never written back)
        f32x4 pp = {0.f, 0.f, 0.f, 0.f}, pm = pp, pv = pp;
        float bp = 0.f, bm = 0.f, bv = 0.f;
        AdamCoefRaw acr;
        if (fuse) {
            acr = adam_coef_request(buf.counters, job.mod);
            const uint32_t o = guard((uint32_t)widx * 4u, nvalid > 0);
            pp = ldg4(rp, o);
            pm = ldg4(rm, o);
            pv = ldg4(rv, o);
            const uint32_t ob = guard((uint32_t)bidx * 4u, has_b);
            bp = ldg(rp, ob);
            bm = ldg(rm, ob);
            bv = ldg(rv, ob);
        }

        GSTAMP(buf.stats, kStampBase + 41, stamp_blk);
        const int rq = round_up(cdiv(R, kWgWaves), 4);
        const int rbeg = wave * rq, rend = min(rbeg + rq, R);
        const rsrc_t gr = make_rsrc(job.G, (size_t)R * job.ldg * sizeof(float));
        const rsrc_t xr = make_rsrc(job.X, (size_t)job.xtotal * job.ldx * sizeof(float));
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc[2][2] = {{z4, z4}, {z4, z4}};
        float bsum[2] = {0.f, 0.f};
        // rounds of 64 batch rows (32 loads in flight), or of 32 when a wave's share is
        // no more than that (the padded steps of a long round would still issue MFMAs)
        const bool half = rq <= 32;
        // More than one round per wave: the pipelined instantiations -- half rounds of 8 steps in
        // two buffers (the 64 registers of one full round, 32 loads in flight as before; full
        // rounds in two buffers spilled: 256 VGPRs + 80 bytes of scratch).  A compile-time choice
        // (-DMOPOE_WGRAD_PIPE=0: full rounds, one buffer -- the A/B build): both sets of
        // instantiations in one translation unit crash this toolchain's register allocator.
#define WG_ROWS(G_, S_, T_, F_, P_) \
    wgrad_rows<G_, S_, T_, F_, P_>(gr, xr, G_ ? job.xrows : nullptr, job.ldg, job.ldx, gcols, xcols, i0, j0, rbeg, rend, lane, acc, bsum)
#define WG_FOLD(G_, S_, T_, P_) (fold ? WG_ROWS(G_, S_, T_, true, P_) : WG_ROWS(G_, S_, T_, false, P_))
        // MOPOE_WGRAD_PIPE: 0 never, 1 always, 2 four-wave blocks only (one wave per SIMD: no
        // partner wave whose MFMAs cover a wave's load round trip), 3 those + the 16-row tiles
        // of eight-wave blocks (a job over twice the batch rows: four rounds per wave)
#define WG_PIPED(G_, T_) WG_FOLD(G_, 8, T_, true)
#define WG_PLAIN(G_, T_) WG_FOLD(G_, 16, T_, false)
#if MOPOE_WGRAD_PIPE == 1
#define WG_LONG(G_, T_) WG_PIPED(G_, T_)
#define WG_LONG16(G_, T_) WG_PIPED(G_, T_)
#elif MOPOE_WGRAD_PIPE == 2
#define WG_LONG(G_, T_) (kWgWaves == 4 ? WG_PIPED(G_, T_) : WG_PLAIN(G_, T_))
#define WG_LONG16(G_, T_) WG_LONG(G_, T_)
#elif MOPOE_WGRAD_PIPE == 3
#define WG_LONG(G_, T_) (kWgWaves == 4 ? WG_PIPED(G_, T_) : WG_PLAIN(G_, T_))
#define WG_LONG16(G_, T_) WG_PIPED(G_, T_)
#else
#define WG_LONG(G_, T_) WG_PLAIN(G_, T_)
#define WG_LONG16(G_, T_) WG_PLAIN(G_, T_)
#endif
        // (sixteen waves: 128 registers each -- half rounds only, the other waves cover the trips)
        if (th == 16) {   // (only non-gathered operands: the decoder's z and g_xhat)
            if (half || kWgWaves == 16 || LEAN)
                WG_FOLD(false, 8, 1, false);
            else
                WG_LONG16(false, 1);
        } else if (job.xrows) {
            if (half || kWgWaves == 16 || LEAN)
                WG_FOLD(true, 8, 2, false);
            else
                WG_LONG(true, 2);
        } else {
            if (half || kWgWaves == 16 || LEAN)
                WG_FOLD(false, 8, 2, false);
            else
                WG_LONG(false, 2);
        }
#undef WG_LONG
#undef WG_LONG16
#undef WG_PIPED
#undef WG_PLAIN
#undef WG_FOLD
#undef WG_ROWS
        GSTAMP(buf.stats, kStampBase + 42, stamp_blk);
        // this wave's partial block -> LDS as [i][j]
        {
            const int c = lane & 15, q = lane >> 4;
            const int rs = th == 16 ? 1 : 2;   // (a 16-row tile: the first fragment's rows, packed)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ti < rs) blk[wave][(rs * (4 * q + r) + ti) * kWgLd + 2 * c + tj] = acc[ti][tj][r];
            if (fold) {   // this wave's column sums of G -> the spare column 32 of its partial tile
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    float v = bsum[ti];
                    v += __shfl_xor(v, 16, kWave);
                    v += __shfl_xor(v, 32, kWave);
                    if (q == 0 && ti < rs) blk[wave][(rs * c + ti) * kWgLd + 32] = v;
                }
            }
        }
        __syncthreads();
        // fixed-order sum of the four partials, 4 consecutive columns per thread
        asm volatile("" ::"s"(karg_sink));  // (keeps the prefetch loads alive)
        if (!XG && !epi) return;   // (XG: every wave stays for the exchange's barriers)
        const int li = (tid >> 3) & 31, lj = 4 * (tid & 7);
        f32x4 g = *reinterpret_cast<const f32x4*>(&blk[0][li * kWgLd + lj]);
#pragma unroll
        for (int k = 1; k < kWgWaves; ++k)  // fixed order
            g += *reinterpret_cast<const f32x4*>(&blk[k][li * kWgLd + lj]);
        GSTAMP(buf.stats, kStampBase + 43, stamp_blk);
        // (an invalid step -- a hand-off or an exchange that timed out -- leaves parameters
        //  and moments alone: run_epochs.py:180-182 never applies half a step)
        AdamCoef ac;
        bool apply = false;
        if (fuse) apply = adam_coef_resolve(acr, w.adam, ac);
        const int eb = xcols - ej;   // has_b: the bias column inside this thread's four
        float gb = eb == 0 ? g[0] : eb == 1 ? g[1] : eb == 2 ? g[2] : g[3];
        if (fold) {   // (fixed order over the waves, as the tile)
            gb = blk[0][li * kWgLd + 32];
#pragma unroll
            for (int k = 1; k < kWgWaves; ++k) gb += blk[k][li * kWgLd + 32];
        }
        float gscale = 1.f;
        if constexpr (XG) {
            const XgPeers& x = w.xg;
            const int W = x.world, me = x.rank;
            gscale = x.inv_world;
            // where this thread's values sit in the flat buffer: one 16-byte access for
            // a full group of four, single words at a row end, the bias word
            const uint32_t o16 = guard((uint32_t)widx * 4u, nvalid >= 4);
            uint32_t o1[3];
#pragma unroll
            for (int e = 0; e < 3; ++e)
                o1[e] = guard((uint32_t)(widx + e) * 4u, (nvalid < 4) & (e < nvalid));
            const uint32_t ob = guard((uint32_t)bidx * 4u, has_b);
#pragma unroll
            for (int r = 0; r < MOPOE_MAX_RANKS; ++r) {
                if (r >= W || r == me) continue;
                const rsrc_t rr = make_rsrc(static_cast<char*>(x.window[r]) + xg_inbox(x, me),
                                            pbytes);
                st16_sys(rr, o16, g);
#pragma unroll
                for (int e = 0; e < 3; ++e) st4_sys(rr, o1[e], g[e]);
                st4_sys(rr, ob, gb);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // pushes have landed
            __syncthreads();
            // (a failed exchange -- a wait out of budget, a peer with other modalities --
            //  withholds this block's update and raises the sticky invalid word)
            if (__syncthreads_or(xg_signal_and_wait(x, tid, b))) {
                apply = false;
                if (tid == 0)
                    __hip_atomic_fetch_add(buf.counters + MOPOE_CTR_INVALID, 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
            f32x4 in[MOPOE_MAX_RANKS];
            float inb[MOPOE_MAX_RANKS];
#pragma unroll
            for (int r = 0; r < MOPOE_MAX_RANKS; ++r) {
                in[r] = z4;
                inb[r] = 0.f;
                if (r >= W || r == me) continue;
                const rsrc_t rr = make_rsrc(static_cast<const char*>(x.window[me]) +
                                                xg_inbox(x, r), pbytes);
                const f32x4 v16 = ld16_sys(rr, o16);
                f32x4 v1 = z4;
#pragma unroll
                for (int e = 0; e < 3; ++e) v1[e] = ld4_sys(rr, o1[e]);
                in[r] = nvalid >= 4 ? v16 : v1;
                inb[r] = ld4_sys(rr, ob);
            }
            f32x4 s4 = me == 0 ? g : in[0];
            float sb = me == 0 ? gb : inb[0];
#pragma unroll
            for (int r = 1; r < MOPOE_MAX_RANKS; ++r) {   // rank order, as k_xgmi
                if (r >= W) continue;
                s4 += r == me ? g : in[r];
                sb += r == me ? gb : inb[r];
            }
            g = s4;
            gb = sb;
            if (!epi) return;
        }
        if (nvalid > 0) {
            f32x4 np = pp, nm = pm, nv = pv;
            if (apply) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float po, mo, vo;
                    adam_update(ac, XG ? __fmul_rn(g[e], gscale) : g[e], pp[e], pm[e], pv[e],
                                &po, &mo, &vo);
                    np[e] = po;
                    nm[e] = mo;
                    nv[e] = vo;
                }
            }
            if (apply && job.wf_off >= 0) {   // the fragment-major copy follows the weight
                f32x4 wv;
#pragma unroll
                for (int e = 0; e < 4; ++e) wv[e] = e < nvalid ? np[e] : 0.f;
                stg4_wt(make_rsrc(buf.wfrag, (size_t)a.lds.wf.total * sizeof(float)),
                        (uint32_t)(job.wf_off + wfrag_piece(ei, ej >> 2, job.wf_k4)) * 4u, wv);
            }
            if (nvalid >= 4) {
                const rsrc_t rg = make_rsrc(buf.grads, pbytes);
                stg4_wt(rg, (uint32_t)widx * 4u, g);
                if (apply) {
                    stg4_wt(rp, (uint32_t)widx * 4u, np);
                    stg4_wt(rm, (uint32_t)widx * 4u, nm);
                    stg4_wt(rv, (uint32_t)widx * 4u, nv);
                }
            } else {
                for (int e = 0; e < nvalid; ++e) {
                    buf.grads[widx + e] = g[e];
                    if (apply) {
                        buf.params[widx + e] = np[e];
                        buf.exp_avg[widx + e] = nm[e];
                        buf.exp_avg_sq[widx + e] = nv[e];
                    }
                }
            }
        }
        if (has_b) {
            buf.grads[bidx] = gb;
            if (apply)
                adam_update(ac, XG ? __fmul_rn(gb, gscale) : gb, bp, bm, bv, buf.params + bidx,
                            buf.exp_avg + bidx, buf.exp_avg_sq + bidx);
        }
        GSTAMP(buf.stats, kStampBase + 44, stamp_blk);
        return;
    }
    const int lb = b - w.total_tiles;
    if (lb < w.lvo_blocks) {
        GSTAMP(buf.stats, kStampBase + 62, lb == w.lvo_blocks - 1 && tid == 0);
        const int tiles = a.lds.fold_tiles > 0 ? a.lds.fold_tiles : cdiv(a.st.n, a.lds.rows);
        const int stride = a.lds.part_stride;
        // d loss / d decoders.<m>.logvar: sum of the row groups' partials.  A block
        // owns 64 columns; its waves take an equal share of the groups each (sixteen
        // loads in flight per thread) and are added in fixed order through LDS.
        int m = 0;
        while (lb >= w.lvo_block_begin[m + 1]) ++m;
        AdamCoef ac;
        bool apply = false;
        if (fuse) apply = adam_coef_resolve(adam_coef_request(buf.counters, m), w.adam, ac);
        const int col = (lb - w.lvo_block_begin[m]) * 64 + lane;
        const bool on = col < a.mdl.input_dim[m];
        float g = 0.f;
        if (on) {
            int slots = 0;
            for (int j = 0; j < a.st.num_jobs; ++j) slots += a.st.job_mod[j] == m;
            const int per = cdiv(tiles, kWgWaves);
            const int t0 = wave * per, t1 = min(t0 + per, tiles);
            for (int sl = 0; sl < slots; ++sl) {
                const float* p = buf.partials + a.lds.lvo_off[m] +
                                 sl * lvo_slot_stride(a.mdl, m) + col;
                g += strided_sum(p, t0, t1, (size_t)stride);
            }
        }
        blk[0][wave * 64 + lane] = g;
        __syncthreads();
        if (wave == 0) {
            g = 0.f;
#pragma unroll
            for (int k = 0; k < kWgWaves; ++k) g += blk[0][k * 64 + lane];  // fixed order
            const int idx = a.mdl.off_lvo[m] + col;
            float gscale = 1.f;
            if constexpr (XG) {
                if (a.mdl.learn_output_scale) {   // this wave alone exchanges its 64 columns
                    const XgPeers& x = w.xg;
                    const int W = x.world, me = x.rank;
                    gscale = x.inv_world;
                    const size_t pbytes = (size_t)a.mdl.num_floats * sizeof(float);
                    const uint32_t o = guard((uint32_t)idx * 4u, on);
#pragma unroll
                    for (int r = 0; r < MOPOE_MAX_RANKS; ++r) {
                        if (r >= W || r == me) continue;
                        const rsrc_t rr = make_rsrc(static_cast<char*>(x.window[r]) +
                                                        xg_inbox(x, me), pbytes);
                        st4_sys(rr, o, g);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (__any(xg_signal_and_wait(x, lane, b))) {
                        apply = false;
                        if (lane == 0)
                            __hip_atomic_fetch_add(buf.counters + MOPOE_CTR_INVALID, 1,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    float in[MOPOE_MAX_RANKS];
#pragma unroll
                    for (int r = 0; r < MOPOE_MAX_RANKS; ++r) {
                        in[r] = 0.f;
                        if (r >= W || r == me) continue;
                        const rsrc_t rr = make_rsrc(static_cast<const char*>(x.window[me]) +
                                                        xg_inbox(x, r), pbytes);
                        in[r] = ld4_sys(rr, o);
                    }
                    float s1 = me == 0 ? g : in[0];
#pragma unroll
                    for (int r = 1; r < MOPOE_MAX_RANKS; ++r) {
                        if (r >= W) continue;
                        s1 += r == me ? g : in[r];
                    }
                    g = s1;
                }
            }
            if (on) {
                if (a.mdl.learn_output_scale) {
                    buf.grads[idx] = g;
                    if (apply)
                        adam_update(ac, XG ? __fmul_rn(g, gscale) : g, buf.params[idx],
                                    buf.exp_avg[idx], buf.exp_avg_sq[idx], buf.params + idx,
                                    buf.exp_avg + idx, buf.exp_avg_sq + idx);
                } else {
                    buf.grads[idx] = 0.f;
                }
            }
        }
        GSTAMP(buf.stats, kStampBase + 63, lb == w.lvo_blocks - 1 && tid == 0);
        return;
    }
    // last block: scalars of the step (run_epochs.py:89-128) + step counter
    GSTAMP(buf.stats, kStampBase + 60, tid == 0);
    finalize_stats<kWgWaves * 64>(a, tid);
    GSTAMP(buf.stats, kStampBase + 61, tid == 0);
    // control words behind the last segment of the gradient buffer: which modalities this
    // rank's batch held (they ride through the ranks' all-reduce; mopoe_adam_step checks)
    if (tid < MOPOE_MAX_MODS)
        buf.grads[a.mdl.off_ctrl + tid] = (a.st.present_mask >> tid) & 1 ? 1.f : 0.f;
    if (a.st.backward) {
        const int s = buf.counters[MOPOE_CTR_STEPS_BEGUN];
        const int invalid = buf.counters[MOPOE_CTR_INVALID];
        // (control word MAX_MODS: this rank's backward was not completed -- adam_step_valid)
        if (tid == MOPOE_MAX_MODS) buf.grads[a.mdl.off_ctrl + MOPOE_MAX_MODS] = invalid ? 1.f : 0.f;
        // (the GEMM blocks read slot s & 1 and never the counts: nothing they read changes)
        if (fuse) step_end(buf.counters, s, tid, a.mdl.num_mods, a.st.present_mask, invalid == 0, w.adam);
        if (tid == 0) {
            const int done = buf.counters[MOPOE_CTR_STEPS_DONE] + 1;
            buf.counters[MOPOE_CTR_STEPS_DONE] = done;
            int first = buf.counters[MOPOE_CTR_FIRST_INVALID];
            if (invalid && first == 0) buf.counters[MOPOE_CTR_FIRST_INVALID] = first = s;
            if (buf.status_host) {
                buf.status_host[0] = done;
                buf.status_host[1] = invalid;
                buf.status_host[2] = first;
            }
        }
    }
}

// forward-only finalisation of the scalars
__global__ __launch_bounds__(1024) void k_finalize(const KArgs a_by_value) {
    (void)a_by_value;  // read in place (see k_latent)
    const KArgs& a = *(const KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    finalize_stats<1024>(a, threadIdx.x);
}

// ---------------------------------------------------------------------------
// k_adam: torch.optim.Adam on the flat buffer (used when the gradients pass
// through an all-reduce first).  grid.y = segment, grid-stride over floats.
// ---------------------------------------------------------------------------
struct AdamSegs {
    int32_t nseg;
    int32_t begin[2 * MOPOE_MAX_MODS];
    int32_t end[2 * MOPOE_MAX_MODS];
    int32_t seg_mod[2 * MOPOE_MAX_MODS];   // modality of the segment (its Adam step count)
    int32_t num_mods, present_mask;
    int32_t world;        // ranks whose gradients were summed into buf.grads (1: none)
    int32_t off_ctrl;     // control words of the gradient buffer (mopoe_model.off_ctrl)
    float grad_scale;     // 1 / world
    mopoe_adam adam;
    // the weight matrix inside segment k that has a fragment-major copy (WFrag: the heads
    // matrix of an encoder segment, the decoder matrix of a decoder segment): the update
    // stores the copy too, as the Adam epilogue of k_wgrad does -- no k_wfrag launch behind it
    int32_t wf_src[2 * MOPOE_MAX_MODS];    // first float of the matrix in the flat buffer, -1: no copy
    int32_t wf_count[2 * MOPOE_MAX_MODS];  // its floats
    int32_t wf_k[2 * MOPOE_MAX_MODS];      // row length K
    int32_t wf_k4[2 * MOPOE_MAX_MODS];     // pieces per row of the copy
    int32_t wf_dst[2 * MOPOE_MAX_MODS];    // the copy's first float in buffers.wfrag
    int32_t wf_total;                      // floats of buffers.wfrag
};

// Whether the step may be applied: no sticky invalid word, and (data-parallel) every
// rank's batch held this rank's modalities -- control word m of the summed gradient
// buffer is then `world` for a present modality and 0 for an absent one.
DEV bool adam_step_valid(const mopoe_buffers& buf, const AdamSegs& s, const float* ctrl) {
    bool ok = buf.counters[MOPOE_CTR_INVALID] == 0;
    if (s.world > 1) {
        for (int m = 0; m < s.num_mods; ++m)
            ok &= ctrl[m] == ((s.present_mask >> m) & 1 ? (float)s.world : 0.f);
        // control word MAX_MODS: how many ranks could not complete their backward (a hand-off
        // time-out raises MOPOE_CTR_INVALID on that rank alone): then NO rank applies the step,
        // every rank raises its sticky word below, and all of them notice together
        ok &= ctrl[MOPOE_MAX_MODS] == 0.f;
    }
    return ok;
}
// The block that finishes last ends the step: counts, the sticky word when the ranks
// disagreed.  (Every other block has read what it needs before it takes its ticket.)
// `records`: also the next step's Adam records (two pow() in double per modality, ~1.5 us
// of one thread) -- k_xgmi-style callers; k_adam has its first block compute them while the
// others stream (step_next_records), so the kernel's tail is a handful of integer stores.
DEV void adam_kernel_end(const mopoe_buffers& buf, const AdamSegs& s, bool valid, int total_blocks,
                         int tid, bool records = true) {
    __shared__ int last;
    __syncthreads();
    if (tid == 0)
        last = __hip_atomic_fetch_add(buf.counters + MOPOE_CTR_TICKET, 1, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT) == total_blocks - 1;
    __syncthreads();
    if (!last) return;
    const int step = buf.counters[MOPOE_CTR_STEPS_BEGUN];
    if (records) {
        step_end(buf.counters, step, tid, s.num_mods, s.present_mask, valid, s.adam);
    } else if (tid < s.num_mods) {
        if (valid && ((s.present_mask >> tid) & 1)) buf.counters[MOPOE_CTR_ADAM_STEPS + tid] += 1;
        if (tid == 0) {
            buf.counters[kCtrBeta] = __builtin_bit_cast(int32_t, s.adam.beta1);
            buf.counters[kCtrBeta + 1] = __builtin_bit_cast(int32_t, s.adam.beta2);
        }
    }
    if (tid == 0) {
        buf.counters[MOPOE_CTR_TICKET] = 0;
        if (!valid && buf.counters[MOPOE_CTR_INVALID] == 0) buf.counters[MOPOE_CTR_INVALID] = 1;
        if (!valid && buf.counters[MOPOE_CTR_FIRST_INVALID] == 0) buf.counters[MOPOE_CTR_FIRST_INVALID] = step;
        if (buf.status_host) {
            buf.status_host[1] = buf.counters[MOPOE_CTR_INVALID];
            buf.status_host[2] = buf.counters[MOPOE_CTR_FIRST_INVALID];
        }
    }
}

// k_adam: a block's operands are requested FIRST (they are on their way while one thread
// fetches the step's coefficients and the validity words in ONE round trip of independent
// loads), the next step's records are made by block (0, 0) under the others' streaming.
__global__ __launch_bounds__(256) void k_adam(const mopoe_buffers buf, const AdamSegs s) {
    __shared__ AdamCoef sc;
    __shared__ int ok;
    const int seg = blockIdx.y, tid = threadIdx.x;
    const int beg = s.begin[seg], end = s.end[seg];   // (beg: 256-byte aligned)
    const int wsrc = s.wf_src[seg], wcnt = s.wf_count[seg], wk = s.wf_k[seg];
    const int wk4 = s.wf_k4[seg], wdst = s.wf_dst[seg];
    const bool wcopy = buf.wfrag != nullptr && wsrc >= 0;
    const size_t pbytes = (size_t)end * sizeof(float);
    const rsrc_t rp = make_rsrc(buf.params, pbytes), rm = make_rsrc(buf.exp_avg, pbytes);
    const rsrc_t rv = make_rsrc(buf.exp_avg_sq, pbytes), rg = make_rsrc(buf.grads, pbytes);
    const rsrc_t rf = make_rsrc(buf.wfrag, wcopy ? (size_t)s.wf_total * sizeof(float) : 0);
    // four consecutive floats per thread (16-byte accesses; the words a load takes past
    // `end` come back as zeros and are not stored)
    const int stride = 4 * gridDim.x * blockDim.x;
    int i = beg + 4 * (blockIdx.x * blockDim.x + tid);
    f32x4 g4, p4, m4, v4;
    auto request = [&](int at) __attribute__((always_inline)) {
        const uint32_t o = guard((uint32_t)at * 4u, at < end);
        g4 = ldg4(rg, o);
        p4 = ldg4(rp, o);
        m4 = ldg4(rm, o);
        v4 = ldg4(rv, o);
    };
    request(i);
    if (tid == 0) {
        const int mod = s.seg_mod[seg];
        const AdamCoefRaw q = adam_coef_request(buf.counters, mod);
        bool good = q.invalid == 0;
        if (s.world > 1) {   // (independent loads: one round trip with the request above)
            const float* ctrl = buf.grads + s.off_ctrl;
            float cw[MOPOE_MAX_MODS + 1];
#pragma unroll
            for (int m = 0; m <= MOPOE_MAX_MODS; ++m) cw[m] = __builtin_nontemporal_load(ctrl + m);
#pragma unroll
            for (int m = 0; m < MOPOE_MAX_MODS; ++m)
                if (m < s.num_mods) good &= cw[m] == ((s.present_mask >> m) & 1 ? (float)s.world : 0.f);
            good &= cw[MOPOE_MAX_MODS] == 0.f;
        }
        AdamCoef c;
        if (!adam_coef_resolve(q, s.adam, c) && q.invalid == 0)   // no record for this step: from the count
            c = adam_coef_load(buf.counters, mod, s.adam);
        sc = c;
        ok = good;
    }
    __syncthreads();
    const AdamCoef ac = sc;
    const bool valid = ok != 0;
    if (valid) {
        for (; i < end; i += stride) {
            const uint32_t o = (uint32_t)i * 4u;
            f32x4 np, nm, nv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // a multiply of its own (never contracted into the update), as in k_xgmi
                float po, mo, vo;
                adam_update(ac, __fmul_rn(g4[e], s.grad_scale), p4[e], m4[e], v4[e], &po, &mo, &vo);
                np[e] = po;
                nm[e] = mo;
                nv[e] = vo;
            }
            if (i + 4 <= end) {
                stg4(rp, o, np);
                stg4(rm, o, nm);
                stg4(rv, o, nv);
            } else {
                for (int e = 0; i + e < end; ++e) {
                    buf.params[i + e] = np[e];
                    buf.exp_avg[i + e] = nm[e];
                    buf.exp_avg_sq[i + e] = nv[e];
                }
            }
            if (wcopy) {
                const int rel = i - wsrc;
                if ((wk & 3) == 0) {   // the four floats are one piece of the copy
                    if (rel >= 0 && rel < wcnt) {
                        const int r = rel / wk, k = rel - r * wk;
                        stg4(rf, (uint32_t)(wdst + wfrag_piece(r, k >> 2, wk4)) * 4u, np);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int q = rel + e;
                        const int r = q / wk, k = q - r * wk;
                        stg1(rf, guard((uint32_t)(wdst + wfrag_piece(r, k >> 2, wk4) + (k & 3)) * 4u,
                                       (q >= 0) & (q < wcnt)), np[e]);
                    }
                }
            }
            if (i + stride < end) request(i + stride);
        }
    }
    // the next step's records (slot (s + 1) & 1: nobody reads it during this step), by the
    // first block, from the counts as they will stand once the last block has advanced them
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid < s.num_mods) {
        const int step = buf.counters[MOPOE_CTR_STEPS_BEGUN];
        const int t = buf.counters[MOPOE_CTR_ADAM_STEPS + tid] + ((valid && ((s.present_mask >> tid) & 1)) ? 1 : 0);
        *bias_rec(buf.counters, step + 1, tid) = bias_of(s.adam, t + 1, step + 1);
    }
    adam_kernel_end(buf, s, valid, gridDim.x * gridDim.y, tid, false);
}

// ---------------------------------------------------------------------------
// free functions
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_poe(const float* __restrict__ mu,
                                             const float* __restrict__ lv, int E,
                                             long long numel, float eps,
                                             float* __restrict__ omu,
                                             float* __restrict__ olv) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel;
         i += (long long)gridDim.x * blockDim.x) {
        float musum = 0.f, tsum = 0.f;
        for (int e = 0; e < E; ++e) {
            const float T = 1.f / (expf(lv[e * numel + i]) + eps);
            musum += mu[e * numel + i] * T;
            tsum += T;
        }
        omu[i] = musum / tsum;
        olv[i] = logf(1.f / tsum);
    }
}

__global__ __launch_bounds__(256) void k_kl_partial(const float* __restrict__ mu,
                                                    const float* __restrict__ lv,
                                                    long long numel,
                                                    float* __restrict__ scratch) {
    __shared__ float wsum[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel;
         i += (long long)gridDim.x * blockDim.x)
        s += 1.f - expf(lv[i]) - mu[i] * mu[i] + lv[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) scratch[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(64) void k_kl_final(const float* __restrict__ scratch,
                                                 int nblocks, float norm,
                                                 float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) s += scratch[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        float k = -0.5f * s;
        if (norm > 0.f) k = k / norm;
        *out = k;
    }
}

__global__ __launch_bounds__(256) void k_reparam(const float* __restrict__ mu,
                                                 const float* __restrict__ lv,
                                                 const float* __restrict__ eps,
                                                 long long numel, uint64_t seed,
                                                 uint32_t stream_id,
                                                 float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel;
         i += (long long)gridDim.x * blockDim.x) {
        const float e = eps ? eps[i] : philox_normal(seed, 0u, stream_id, (uint32_t)i);
        out[i] = e * expf(0.5f * lv[i]) + mu[i];
    }
}

__global__ __launch_bounds__(256) void k_mix_select(const float* __restrict__ mus,
                                                    const float* __restrict__ lvs, int K,
                                                    int n, int d,
                                                    const int32_t* __restrict__ bounds,
                                                    float* __restrict__ omu,
                                                    float* __restrict__ olv) {
    const long long total = (long long)n * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / d);
        int k = 0;
        while (k + 1 < K && row >= bounds[k + 1]) ++k;
        omu[i] = mus[(size_t)k * total + i];
        olv[i] = lvs[(size_t)k * total + i];
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* arg = "") {
    snprintf(g_err, sizeof(g_err), fmt, arg);
    return code;
}

// ---- optional per-kernel timing with HIP events on the launch stream -------
struct ProfRec {
    int kernel;
    hipEvent_t beg, end;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;          // recorded, not yet read
std::vector<hipEvent_t> g_prof_pool;  // free events

hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {  // records an event pair around one launch when enabled
    hipStream_t s;
    ProfRec r;
    bool on;
    ProfScope(int kernel, hipStream_t st) : s(st), on(g_prof_on) {
        if (!on) return;
        r.kernel = kernel;
        r.beg = prof_event();
        r.end = prof_event();
        (void)hipEventRecord(r.beg, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.end, s);
        g_prof.push_back(r);
    }
};

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return MOPOE_ERR_HIP;
    }
    return 0;
}

int validate(const mopoe_model* mdl, const mopoe_step* st, const mopoe_buffers* buf,
             bool train) {
    if (!mdl || !st || !buf) return fail(MOPOE_ERR_ARG, "null descriptor%s");
    if (st && (st->gemm_operands != MOPOE_OPERANDS_F32 && st->gemm_operands != MOPOE_OPERANDS_BF16))
        return fail(MOPOE_ERR_ARG, "mopoe_step.gemm_operands: MOPOE_OPERANDS_F32 or MOPOE_OPERANDS_BF16%s");
    if (mdl->num_mods < 1 || mdl->num_mods > MOPOE_MAX_MODS)
        return fail(MOPOE_ERR_ARG, "num_mods out of range%s");
    if (mdl->class_dim < 1 || mdl->class_dim > 256)
        return fail(MOPOE_ERR_ARG, "class_dim out of range%s");
    if (st->n < 1) return fail(MOPOE_ERR_ARG, "empty batch%s");
    if (st->likelihood != MOPOE_LIK_NORMAL && st->likelihood != MOPOE_LIK_LAPLACE)
        return fail(MOPOE_ERR_ARG, "likelihood must be MOPOE_LIK_NORMAL or MOPOE_LIK_LAPLACE%s");
    if (st->present_mask <= 0 || st->present_mask >= (1 << mdl->num_mods))
        return fail(MOPOE_ERR_ARG, "present_mask out of range%s");
    if (st->num_subsets < 1 || st->num_subsets > MOPOE_MAX_SUBSETS)
        return fail(MOPOE_ERR_ARG, "num_subsets out of range%s");
    if (st->num_comp < 1 || st->num_comp > MOPOE_MAX_SUBSETS)
        return fail(MOPOE_ERR_ARG, "num_comp out of range%s");
    if (st->group_rows < 0 || (st->group_rows > 0 && st->n % st->group_rows != 0))
        return fail(MOPOE_ERR_ARG, "group_rows must be 0 or divide n%s");
    if (st->group_rows > 0 && st->backward)
        return fail(MOPOE_ERR_ARG, "group_rows is a forward-only feature%s");
    {
        const int r = st->rows_per_group;
        if (r != 0 && r != 1 && r != 2 && r != 4 && r != 8 && r != 16)
            return fail(MOPOE_ERR_ARG, "rows_per_group must be 0, 1, 2, 4, 8 or 16%s");
    }
    if (st->num_jobs < 1 || st->num_jobs > MOPOE_MAX_JOBS)
        return fail(MOPOE_ERR_ARG, "num_jobs out of range%s");
    for (int s = 0; s < st->num_subsets; ++s) {
        if (st->sub_mask[s] == 0 || st->sub_mask[s] >= (1 << mdl->num_mods))
            return fail(MOPOE_ERR_ARG, "subset mask out of range%s");
        if (st->sub_avail[s] && (st->sub_mask[s] & ~st->present_mask))
            return fail(MOPOE_ERR_ARG, "available subset has an absent member%s");
        if (st->sub_kind[s] > MOPOE_SUB_SLICES)
            return fail(MOPOE_ERR_ARG, "bad sub_kind%s");
        for (int j = 0; j < MOPOE_MAX_MODS; ++j)
            if (st->sub_members[s][j] >= mdl->num_mods)
                return fail(MOPOE_ERR_ARG, "bad sub_members%s");
    }
    for (int k = 0; k < st->num_comp; ++k)
        if (st->comp_sub[k] >= st->num_subsets || !st->sub_avail[st->comp_sub[k]])
            return fail(MOPOE_ERR_ARG, "mixture component is not an available subset%s");
    if (st->joint_mode == MOPOE_JOINT_EXPERT &&
        (st->expert_subset < 0 || st->expert_subset >= st->num_subsets ||
         !st->sub_avail[st->expert_subset]))
        return fail(MOPOE_ERR_ARG, "use_expert subset unavailable%s");
    if (st->joint_mode < 0 || st->joint_mode > MOPOE_JOINT_EXPERT)
        return fail(MOPOE_ERR_ARG, "bad joint_mode%s");
    if (train && st->joint_mode != MOPOE_JOINT_MIXTURE)
        return fail(MOPOE_ERR_ARG, "training requires joint_mode mixture%s");
    for (int j = 0; j < st->num_jobs; ++j) {
        const int m = st->job_mod[j];
        if (m >= mdl->num_mods || !((st->present_mask >> m) & 1))
            return fail(MOPOE_ERR_ARG, "decoder job for an absent modality%s");
        if (st->job_src[j] >= st->num_subsets ||
            (st->job_src[j] >= 0 && !st->sub_avail[(int)st->job_src[j]]))
            return fail(MOPOE_ERR_ARG, "decoder job source unavailable%s");
        if (j > 0 && st->job_stream[j] < st->job_stream[j - 1])
            return fail(MOPOE_ERR_ARG, "decoder jobs must be grouped by pass%s");
        // (the jobs of a slot share the decoder stages: one per modality, slots in order)
        if (j > 0 && st->job_slot[j] < st->job_slot[j - 1])
            return fail(MOPOE_ERR_ARG, "decoder jobs must be grouped by slot%s");
        for (int i = 0; i < j; ++i)
            if (st->job_slot[i] == st->job_slot[j] && st->job_mod[i] == m)
                return fail(MOPOE_ERR_ARG, "two decoder jobs of one modality in one slot%s");
    }
    for (int m = 0; m < mdl->num_mods; ++m) {
        if (mdl->input_dim[m] < 1 || mdl->style_dim[m] < 0)
            return fail(MOPOE_ERR_ARG, "bad modality dims%s");
        if (!((st->present_mask >> m) & 1)) continue;
        if (!buf->x[m] || !buf->hidden[m] || !buf->heads[m] || !buf->z[m] || !buf->loc[m])
            return fail(MOPOE_ERR_ARG, "null forward buffer%s");
        // x[m] is read through a descriptor of exactly x_rows[m] * d_m floats
        if (buf->x_rows[m] < 0 || (buf->row_index[m] && buf->x_rows[m] < 1))
            return fail(MOPOE_ERR_ARG, "x_rows[m] (rows of x[m]) is required with row_index[m]%s");
        if (!buf->row_index[m] && buf->x_rows[m] != 0 && buf->x_rows[m] < st->n)
            return fail(MOPOE_ERR_ARG, "x[m] has fewer rows than the batch%s");
        if ((long long)(buf->x_rows[m] ? buf->x_rows[m] : st->n) * mdl->input_dim[m] * 4 >= (1ll << 31))
            return fail(MOPOE_ERR_ARG, "x[m] too large for 32-bit row offsets%s");
        if (train && (!buf->g_xhat[m] || !buf->g_heads[m] || !buf->g_pre[m]))
            return fail(MOPOE_ERR_ARG, "null backward buffer%s");
        // the kernels address every per-row tensor with 32-bit byte offsets whose top bit
        // marks an invalid lane: (decoder passes) * n * widest row must stay below 2 GiB
        int njobs_m = 0;
        for (int j = 0; j < st->num_jobs; ++j) njobs_m += st->job_mod[j] == m;
        const long long widest = mdl->input_dim[m] > MOPOE_HIDDEN ? mdl->input_dim[m] : MOPOE_HIDDEN;
        if ((long long)(njobs_m > 0 ? njobs_m : 1) * st->n * widest * 4 >= (1ll << 31))
            return fail(MOPOE_ERR_ARG, "batch too large for 32-bit row offsets%s");
    }
    if ((long long)st->num_subsets * st->n * mdl->class_dim * 4 >= (1ll << 31))
        return fail(MOPOE_ERR_ARG, "batch too large for 32-bit row offsets%s");
    if (!buf->params || !buf->subsets_mu || !buf->subsets_logvar || !buf->joint_mu ||
        !buf->joint_logvar || !buf->stats || !buf->partials || !buf->counters)
        return fail(MOPOE_ERR_ARG, "null shared buffer%s");
    if (train && !buf->grads) return fail(MOPOE_ERR_ARG, "null grads%s");
    if (reinterpret_cast<uintptr_t>(buf->params) & 255)
        return fail(MOPOE_ERR_ARG, "params must be 256-byte aligned%s");
    return 0;
}

// the kernels' view of the caller's buffers: x_rows filled in for identity batches
void bind_buffers(KArgs& ka, const mopoe_buffers& buf) {
    ka.buf = buf;
    for (int m = 0; m < MOPOE_MAX_MODS; ++m)
        if (!ka.buf.x_rows[m]) ka.buf.x_rows[m] = ka.st.n;
}

// Four-row groups ("quad" form of the fused launch, latent_body FORM 4 / 5): a training step of
// <= 2 modalities with at most 256 rows is cut into groups of FOUR rows -- four times the row
// groups on four times the CUs, each issuing a quarter of the MFMAs.
// One decoder pass (FORM 4), or two whose jobs are the same modalities in the same order
// (FORM 5: method poe's joint + unimodal jobs).  MOPOE_QUAD=0 turns it off.
// The environment knobs (tests, experiments), read ONCE: a training step makes no getenv
// call -- a linear scan of the environment each -- and a setenv in mid-run cannot switch
// kernels between two steps of one run.  mopoe_reload_knobs() re-reads them (the tests that
// compare launch forms inside one process call it after changing the environment).
struct Knobs {
    int quad_max_n;      // MOPOE_QUAD_MAX_N: rows up to which the four-row form is used (1024 = one group per CU).
                         // Measured, us per step, sixteen-row / four-row groups (beyond 512 rows the launch has
                         // more blocks than CUs: launch_forward_part): joint_elbo 384 rows 38.1 / 32.9, 512:
                         // 39.2 / 37.1, 640: 43.2 / 42.0, 768: 46.5 / 42.1, 1024: 46.8 / 46.0; poe 512:
                         // 50.2 / 44.1, 640: 53.2 / 48.4, 768: 56.5 / 47.8, 1024: 58.0 / 53.4
    bool quad;           // MOPOE_QUAD=0 turns the four-row form off
    bool no_fuse;        // MOPOE_NO_FUSE: encoder layer and per-sample chain in two launches
    bool no_lean;        // MOPOE_NO_LEAN: the generic instantiation of the fused launch
    int quad_oversub;    // MOPOE_QUAD_OVERSUB: K parts of the producers of an over-subscribed four-row launch
                         // (-1: the launch code's rule, 0: never)
    int quad_max_n2;     // MOPOE_QUAD_MAX_N2: ... for steps with two decoder passes (method poe; follows MOPOE_QUAD_MAX_N)
    int q1_idle;         // MOPOE_Q1_IDLE: waves the four-row heads stage leaves free (2)
    int handoff_spins;   // MOPOE_TEST_HANDOFF_SPINS: polls of a row group before it gives up
    int fuse_blocks;     // MOPOE_FUSE_BLOCKS: largest grid the fused launch is used for; 0 (default): the
                         // compute units of the device the call runs on (256 on a whole MI355X)
    bool cu_mask;        // a CU mask is in force (HSA_CU_MASK / ROC_GLOBAL_CU_MASK / HSA_CU_MASK_SKIP_INIT in the
                         // environment): how many workgroups are resident at once is not known -- no fused launch
    int knock;           // MOPOE_KNOCK (diagnostic build): phases to leave out
    bool wgrad_nofold, wgrad_tall;   // MOPOE_WGRAD_NOFOLD / MOPOE_WGRAD_TALL: experiments
    int wb_min_rows;     // MOPOE_WB_MIN_ROWS: rows from which the weight gradients run as split 64 x 64 tiles (kWbMinRows)
    int lin_ks;          // MOPOE_LIN_KS: K parts of the separate encoder-layer launch (0: its own choice; experiments)
    bool enc0_chain;     // MOPOE_TOPOLOGY_CHAIN: every non-default topology through the general chain of launches (A/B, tests:
                         // an encoder without a hidden layer and the logvar head otherwise run in the row-group kernel)
    bool drop_apart;     // MOPOE_DROPOUT_APART: Dropout as a launch of its own behind every hidden layer (A/B)
    bool nll_apart;      // MOPOE_NLL_APART: the likelihood as a launch of its own in a training step (A/B)
    bool wgrad_lean8;    // MOPOE_WGRAD_LEAN8 (default 1): k_wgrad<8, lean> where the launch has 1-2 blocks per CU
    bool dec0_apart;     // MOPOE_DEC0_APART: the decoders' first hidden layer as a launch of its own (A/B)
    bool nn_wide;        // MOPOE_NN_WIDE: g_gemm_nn with one wave per tile at every batch size (A/B)
    bool uniform_ks;     // MOPOE_UNIFORM_KS: one K-part count for all wide modalities in the fused launch (A/B)
    int lin_xcd;         // MOPOE_LIN_XCD: k_linear_big's XCD-aware tile order (1)
    int lin_big_rows;    // MOPOE_LIN_BIG_ROWS: rows from which the encoder layer runs in 64 x 64 tiles (kLinBigRows)
    int xg_fail_slot;    // MOPOE_TEST_XG_FAIL_SLOT: the exchanging block that reports a failed wait (-1)
};
Knobs read_knobs() {
    auto num = [](const char* name, int dflt) {
        const char* v = getenv(name);
        return v ? (int)strtol(v, nullptr, 0) : dflt;
    };
    Knobs k;
    k.quad_oversub = num("MOPOE_QUAD_OVERSUB", -1);
    k.quad_max_n = num("MOPOE_QUAD_MAX_N", 1024);
    k.quad_max_n2 = num("MOPOE_QUAD_MAX_N2", getenv("MOPOE_QUAD_MAX_N") ? k.quad_max_n : 1024);
    k.quad = num("MOPOE_QUAD", 1) != 0;
    k.no_fuse = getenv("MOPOE_NO_FUSE") != nullptr;
    k.no_lean = getenv("MOPOE_NO_LEAN") != nullptr;
    k.q1_idle = num("MOPOE_Q1_IDLE", 2);
    k.lin_big_rows = num("MOPOE_LIN_BIG_ROWS", kLinBigRows);
    k.lin_ks = num("MOPOE_LIN_KS", 0);
    k.lin_xcd = num("MOPOE_LIN_XCD", 1);
    k.uniform_ks = getenv("MOPOE_UNIFORM_KS") != nullptr;
    k.nn_wide = getenv("MOPOE_NN_WIDE") != nullptr;
    k.dec0_apart = getenv("MOPOE_DEC0_APART") != nullptr;
    k.wgrad_lean8 = num("MOPOE_WGRAD_LEAN8", 1) != 0;
    k.nll_apart = getenv("MOPOE_NLL_APART") != nullptr;
    k.drop_apart = getenv("MOPOE_DROPOUT_APART") != nullptr;
    k.enc0_chain = getenv("MOPOE_TOPOLOGY_CHAIN") != nullptr;
    k.wb_min_rows = num("MOPOE_WB_MIN_ROWS", 4096);   // (= kWbMinRows, mopoe_wgrad_big.inc)
    k.handoff_spins = num("MOPOE_TEST_HANDOFF_SPINS", kHandoffSpins);
    k.fuse_blocks = num("MOPOE_FUSE_BLOCKS", 0);
    k.cu_mask = getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK") || getenv("HSA_CU_MASK_SKIP_INIT");
    k.knock = num("MOPOE_KNOCK", 0);
    k.wgrad_nofold = getenv("MOPOE_WGRAD_NOFOLD") != nullptr;
    k.wgrad_tall = getenv("MOPOE_WGRAD_TALL") != nullptr;
    k.xg_fail_slot = num("MOPOE_TEST_XG_FAIL_SLOT", -1);
    return k;
}
Knobs g_knobs = read_knobs();

// The fused launch's hand-off (row groups wait for producer workgroups of the SAME grid) is free
// of circular waits only while every producer holds a compute unit before a row group starts
// to wait.  What the launch code may assume about that:
//   device_cus()   the compute units of the device this thread's calls go to
//                  (hipDeviceAttributeMultiprocessorCount, asked once per device) -- NOT the
//                  constant 256: a partitioned MI355X (CPX: 32) or another part has fewer;
//   fuse_blocks()  the workgroups of 1,024 threads taken to be resident at once: that count,
//                  or MOPOE_FUSE_BLOCKS;
//   no_fuse()      MOPOE_NO_FUSE, a CU mask in the environment, or a connected peer-window
//                  communicator whose ranks share this device (g_shared_device_comms): other
//                  processes' grids on the same CUs break residency.
// launch_forward_part additionally holds the producers of an over-subscribed launch to
// hipOccupancyMaxActiveBlocksPerMultiprocessor x device_cus() (resident_blocks).
// HIP does not promise dispatch in block-index order; the order is what the hardware does on
// an exclusive device (soaked: profiles/r03_n_soak_oversubscribed_100k_steps.txt), the bounded
// wait + sticky invalid word + StepRetry turn a violation into a retried step, not a hang.
int device_cus() {
    static thread_local int cached_dev = -1, cached_cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;     // (no device: layout queries on a CPU host)
    if (dev != cached_dev) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) return 256;
        cached_dev = dev;
        cached_cus = v;
    }
    return cached_cus;
}
// connected peer-window communicators of this process whose ranks share this rank's device
// (mopoe_comm_connect / mopoe_comm_destroy): while there is one, EVERY step of the process runs in
// separate launches -- also the steps that do not pass the communicator (the exchange as a
// launch of its own behind mopoe_train_step)
std::atomic<int> g_shared_device_comms{0};
static bool no_fuse() { return g_knobs.no_fuse || g_knobs.cu_mask || g_shared_device_comms.load(std::memory_order_relaxed) > 0; }
static int fuse_blocks() { return g_knobs.fuse_blocks > 0 ? g_knobs.fuse_blocks : device_cus(); }

int quad_max_rows() { return g_knobs.quad_max_n; }
int quad_first_pass_jobs(const mopoe_step& st) {   // jobs of the first decoder pass (slot)
    int n0 = 0;
    while (n0 < st.num_jobs && st.job_slot[n0] == st.job_slot[0]) ++n0;
    return n0;
}
bool quad_step(const mopoe_model& mdl, const mopoe_step& st) {
    if (!g_knobs.quad || st.pad_) return false;   // (pad_: an encoder without a hidden layer -- generic body)
    if (!st.backward || !st.sample || st.joint_mode != MOPOE_JOINT_MIXTURE || st.group_rows != 0 ||
        st.rows_per_group != 0 || mdl.num_mods > 2 || st.num_jobs > 4 || st.n < 4 ||
        // (two decoder passes -- method poe -- gain more from four-row groups than one: measured,
        //  Knobs::quad_max_n)
        st.n > (st.num_jobs > quad_first_pass_jobs(st) ? g_knobs.quad_max_n2 : quad_max_rows()) ||
        cdiv(st.n, 4) > fuse_blocks() ||   // (every row group resident: one per CU)
        st.likelihood != MOPOE_LIK_NORMAL)
        return false;
    for (int k = 0; k < st.num_subsets; ++k)
        if (st.sub_kind[k] == MOPOE_SUB_SLICES) return false;
    {   // one pass, or two with the same modalities in the same order; the first pass samples
        // the joint latent with one noise stream, the second one feeds each job its own subset
        const int n0 = quad_first_pass_jobs(st);
        if (n0 > 2 || (st.num_jobs != n0 && st.num_jobs != 2 * n0)) return false;
        for (int j = 0; j < st.num_jobs; ++j) {
            if (j < n0 && (st.job_stream[j] != st.job_stream[0] || st.job_src[j] >= 0)) return false;
            if (j >= n0 && (st.job_mod[j] != st.job_mod[j - n0] || st.job_slot[j] != st.job_slot[n0] ||
                            st.job_src[j] < 0))
                return false;
        }
    }
    return !no_fuse() && !g_knobs.no_lean;
}
// THE layout of a step: every caller (launches, mopoe_row_groups, mopoe_latent_lds_bytes)
// goes through here, so they agree on the rows per group
// the work split of the four-row form's GEMM stages (one unit per wave); false: no fit
bool quad_tables(const mopoe_model& mdl, const mopoe_step& st, LatentLds& L) {
    // dL/dh: K = nh_m in parts of 8 (two 16-byte reads of the g_heads tile per wave)
    L.q6_kper = 8;
    int u = 0, pm = 0;
    for (int i = 0; i <= MOPOE_MAX_MODS; ++i) L.q6_begin[i] = L.q1_begin[i] = 0;
    for (int i = 0; i < mdl.num_mods; ++i)
        if ((st.present_mask >> i) & 1) {
            L.q6_begin[pm++] = u;
            u += cdiv(heads_dim(mdl, i), L.q6_kper);
        }
    for (int i = pm; i <= MOPOE_MAX_MODS; ++i) L.q6_begin[i] = u;
    if (u > kLatentWaves) return false;
    // dL/dz: K = d_m in parts of a multiple of 4, at most 32 (registers), one per wave.  (The
    // tables of the K parts describe ONE pass: a second pass has the same jobs, quad_step.)
    const int njobs0 = quad_first_pass_jobs(st);
    for (L.q4_kper = 4; L.q4_kper <= 32; L.q4_kper += 4) {
        u = 0;
        for (int j = 0; j < njobs0; ++j) u += cdiv(mdl.input_dim[st.job_mod[j]], L.q4_kper);
        if (u <= kLatentWaves) break;
    }
    if (L.q4_kper > 32) return false;
    u = 0;
    for (int j = 0; j <= MOPOE_MAX_JOBS; ++j) {
        L.q4_begin[j] = u;
        if (j < njobs0) {
            if (z_dim(mdl, st.job_mod[j]) > 64) return false;   // one 64-column tile of g_z
            u += cdiv(mdl.input_dim[st.job_mod[j]], L.q4_kper);
        }
    }
    // heads: (tile of 64 columns, K part) units per present modality; K = 256 = 64 groups
    // of 4, cut in as many parts as the waves allow (at most 16 groups per part: registers)
    L.wf = wfrag_layout(mdl);
    int tiles1 = 0;
    for (int i = 0; i < mdl.num_mods; ++i)
        if ((st.present_mask >> i) & 1) tiles1 += L.wf.t1[i];
    if (tiles1 < 1 || tiles1 > kLatentWaves) return false;
    {   // at least two waves of the heads stage stay free: they draw the step's noise meanwhile
        // (measured, configs[1]: 5 K parts and one free wave +0.27 us, 4 parts and four -0.3 us
        //  against the noise drawn in S0; MOPOE_Q1_IDLE: experiments)
        const int idle = g_knobs.q1_idle;
        L.q1_parts = (kLatentWaves - idle) / tiles1;
        if (L.q1_parts < 1) L.q1_parts = kLatentWaves / tiles1;
    }
    if (L.q1_parts > 8) L.q1_parts = 8;
    L.q1_kper = cdiv(kHid / 4, L.q1_parts);
    if (L.q1_kper > 16) return false;
    u = 0, pm = 0;
    for (int i = 0; i < mdl.num_mods; ++i)
        if ((st.present_mask >> i) & 1) {
            L.q1_begin[pm++] = u;
            u += L.wf.t1[i] * L.q1_parts;
            if (heads_dim(mdl, i) > 128) return false;   // (the reduce pass: 128 columns per modality)
        }
    for (int i = pm; i <= MOPOE_MAX_MODS; ++i) L.q1_begin[i] = u;
    // decoder: one 64-column tile per wave, K = z_dim <= 64 (units of all passes in one table)
    u = 0;
    for (int j = 0; j <= MOPOE_MAX_JOBS; ++j) {
        L.q3_begin[j] = u;
        if (j < st.num_jobs) u += L.wf.t3[st.job_mod[j]];
    }
    u = L.q3_begin[njobs0];                  // (units of a pass)
    if (u > kLatentWaves) return false;
    L.kl_first = u < kLatentWaves ? u : 0;   // the KL sums ride on the decoder stage's idle waves
    L.kl_pool = kLatentWaves - L.kl_first;
    L.qred = L.total;
    // the decoder weights' LDS copy (dL/dz) lies behind the 64-column partials of the heads
    // and dL/dz stages, inside / past the area of dL/dh's 256-column partials (used later)
    int off = L.qred + kLatentWaves * 4 * 64, units = 0;
    L.wd_units[0] = L.wd_units[1] = L.wd_valid[0] = L.wd_valid[1] = 0;
    pm = 0;
    for (int i = 0; i < mdl.num_mods; ++i) {
        L.wdl[i] = off;
        if (!((st.present_mask >> i) & 1)) continue;
        // (32 rows from the start of the last K part: dL/dz reads them unguarded, as zeros; the
        //  piece that holds the weights' last words is completed with the parameters behind
        //  them -- finite, and multiplied by the zero padding of the g_xhat tile)
        const int d = mdl.input_dim[i], zd = z_dim(mdl, i);
        const int nu = cdiv(((cdiv(d, L.q4_kper) - 1) * L.q4_kper + 32) * zd, 256);
        L.wd_valid[pm] = cdiv(d * zd, 4);
        L.wd_units[pm++] = nu;
        units += nu;
        off += 256 * nu;
    }
    for (int i = mdl.num_mods; i < MOPOE_MAX_MODS; ++i) L.wdl[i] = off;
    if (units > 6 * kLatentWaves) return false;   // (six units per wave)
    for (int w = 0; w < kLatentWaves; ++w) {
        LatentLds::Q4Unit& q = L.q4u[w];
        memset(&q, 0, sizeof(q));
        q.nk4 = -1;
        if (w >= L.q4_begin[njobs0]) continue;
        int j = 0;
        while (w >= L.q4_begin[j + 1]) ++j;
        const int m = st.job_mod[j], k0 = (w - L.q4_begin[j]) * L.q4_kper;
        const int left = round_up(mdl.input_dim[m], 16) - k0;   // columns of the g_xhat tile that are its own
        q.nk4 = left <= 0 ? 0 : cdiv(left, 4) < L.q4_kper / 4 ? cdiv(left, 4) : L.q4_kper / 4;
        q.ga = L.dj[j].gx + k0;
        q.wl = L.wdl[m] + k0 * z_dim(mdl, m);
        q.ldx_zd = L.dj[j].ldx << 8 | z_dim(mdl, m);   // (z_dim <= 64: checked above)
    }
    L.total = L.qred + kLatentWaves * 4 * 256;
    if (off > L.total) L.total = off;
    return L.total * 4 <= 160 * 1024;
}

void step_layout(const mopoe_model& mdl, const mopoe_step& st, LatentLds& L) {
    L.quad_ok = 0;
    L.fold_tiles = 0;
    if (quad_step(mdl, st)) {
        L.fits = latent_lds_layout_rows(mdl, st, kLatentWaves, 4, L);
        if (L.fits && L.xs_early && L.s3_nt == 2 && quad_tables(mdl, st, L)) {
            L.quad_ok = 1;
            return;
        }
    }
    latent_lds_layout(mdl, st, kLatentWaves, L);
    L.quad_ok = 0;
    L.wf = wfrag_layout(mdl);
}

int latent_lds_bytes(const mopoe_model& mdl, const mopoe_step& st) {
    LatentLds L;
    step_layout(mdl, st, L);
    return L.total * (int)sizeof(float);
}

// ks_hint: the K split of the fused launch for the same step (its sums must come out the
// same whichever form runs: tests/test_hip_fused.py), or 0 for this launch's own choice
// drop: Dropout(p) in the epilogue (k_linear_drop; the caller keeps to batches below
// Knobs::lin_big_rows, the 64-row tiles have no such epilogue)
int launch_linear(const LinArgs& la_in, int max_k, int max_cols, hipStream_t s, int ks_hint = 0,
                  const LinDrop* drop = nullptr, const LinNll* nll = nullptr) {
    LinArgs la = la_in;
    const int kp = round_up(max_k < kEncKChunk ? max_k : kEncKChunk, 16);
    const size_t lds = ((size_t)kRows * (kp + 4) + kLinRedFloats) * sizeof(float);
    if (la.n >= g_knobs.lin_big_rows) {
        if (drop || nll) return fail(MOPOE_ERR_ARG, "internal: no dropout / likelihood epilogue in the 64-row tiles%s");
        la.xcd_order = g_knobs.lin_xcd;
        ProfScope ps(MOPOE_KERNEL_LINEAR, s);
        if (la.bf16)
            hipLaunchKernelGGL((k_linear_big<64, true>), dim3(cdiv(max_cols, kBigCols), cdiv(la.n, 64), la.ngroups),
                               dim3(256), 0, s, la);
        else
            hipLaunchKernelGGL((k_linear_big<64, false>), dim3(cdiv(max_cols, kBigCols), cdiv(la.n, 64), la.ngroups),
                               dim3(256), 0, s, la);
        return check_launch("k_linear_big");
    }
    // K parts per column tile: as many as it takes to put ~2 workgroups on every CU
    const int tiles = cdiv(la.n, kRows) * la.ngroups;
    int ks = 1;
    while (ks < 4 && tiles * cdiv(max_cols, 64 / ks) < 2 * 256) ks *= 2;
    if (ks_hint) ks = ks_hint;
    if (g_knobs.lin_ks) ks = g_knobs.lin_ks;
    la.ksplit = ks;
    {
        ProfScope ps(MOPOE_KERNEL_LINEAR, s);
        const dim3 grid(cdiv(max_cols, 64 / ks), cdiv(la.n, kRows), la.ngroups);
        if (nll) {
            LinNllArgs na;
            na.la = la;
            na.q = *nll;
            if (ks == 4)
                hipLaunchKernelGGL(k_linear_nll<4>, grid, dim3(256), lds, s, na);
            else if (ks == 2)
                hipLaunchKernelGGL(k_linear_nll<2>, grid, dim3(256), lds, s, na);
            else
                hipLaunchKernelGGL(k_linear_nll<1>, grid, dim3(256), lds, s, na);
        } else if (drop) {
            LinDropArgs da;
            da.la = la;
            da.d = *drop;
            if (ks == 4)
                hipLaunchKernelGGL(k_linear_drop<4>, grid, dim3(256), lds, s, da);
            else if (ks == 2)
                hipLaunchKernelGGL(k_linear_drop<2>, grid, dim3(256), lds, s, da);
            else
                hipLaunchKernelGGL(k_linear_drop<1>, grid, dim3(256), lds, s, da);
        } else if (ks == 4)
            hipLaunchKernelGGL(k_linear<4>, grid, dim3(256), lds, s, la);
        else if (ks == 2)
            hipLaunchKernelGGL(k_linear<2>, grid, dim3(256), lds, s, la);
        else
            hipLaunchKernelGGL(k_linear<1>, grid, dim3(256), lds, s, la);
    }
    return check_launch("k_linear");
}

// MOPOE_NO_FUSE=1 keeps the encoder layer and the per-sample chain in two launches;
// MOPOE_FUSE_BLOCKS caps the grid the fused launch is used for (Knobs: read once; a test
// that compares the two forms inside one process calls mopoe_reload_knobs in between).
// MOPOE_TEST_HANDOFF_SPINS: a test knob -- with 0 every row group of the fused launch gives
// up without looking at its flag, which is how tests/test_hip_invalid.py drives the "step
// could not be completed" path (flags left non-zero by the producers included).
static int handoff_spins() { return g_knobs.handoff_spins; }

// Which instantiation of the fused launch serves this step (latent_body's FORM); 0 = the
// generic one.  MOPOE_NO_LEAN=1 forces the generic form (the forms are bit-identical:
// tests/test_hip_fused.py).
int launch_form(const KArgs& ka) {
    const mopoe_model& mdl = ka.mdl;
    const mopoe_step& st = ka.st;
    const LatentLds& L = ka.lds;
    if (g_knobs.no_lean) return 0;
    if (!st.backward || !st.sample || st.joint_mode != MOPOE_JOINT_MIXTURE || st.group_rows != 0 ||
        L.rows != kRows || st.likelihood != MOPOE_LIK_NORMAL)
        return 0;   // (four-row groups: the caller picks form 4)
    for (int k = 0; k < st.num_subsets; ++k)
        if (st.sub_kind[k] == MOPOE_SUB_SLICES) return 0;
    // (pad_ bit 1, the decoder's logvar head: form 6 = form 1 + the head, fused launch only; else generic)
    if (st.pad_ & 2)
        return (mdl.num_mods <= 2 && L.single_pass && L.s3_nt == 2 && L.xs_early && st.num_jobs <= 2 && !L.enc0) ? 6 : 0;
    if (mdl.num_mods <= 2 && L.single_pass && L.s3_nt == 2 && L.xs_early && st.num_jobs <= 2) return 1;
    if (mdl.num_mods <= 2 && !L.single_pass && L.s3_nt == 2 && L.xs_early && st.num_jobs <= 4) return 2;
    if (mdl.num_mods <= 4 && L.single_pass && L.s3_nt == 4 && !L.xs_early && st.num_jobs <= 4) return 3;
    return 0;
}

const void* fused_form_fn(int form) {
    switch (form) {
        case 1: return reinterpret_cast<const void*>(k_fused<1>);
        case 2: return reinterpret_cast<const void*>(k_fused<2>);
        case 3: return reinterpret_cast<const void*>(k_fused<3>);
        case 4: return reinterpret_cast<const void*>(k_fused<4>);
        case 5: return reinterpret_cast<const void*>(k_fused<5>);
        case 6: return reinterpret_cast<const void*>(k_fused<6>);
        default: return reinterpret_cast<const void*>(k_fused<0>);
    }
}
// Workgroups of the fused launch (instantiation `form`, `lds` bytes of dynamic LDS) that are
// resident at once on the current device: hipOccupancyMaxActiveBlocksPerMultiprocessor x its
// compute units.  Asked once per (device, form, LDS size) and thread.
int resident_blocks(int form, int lds) {
    struct Key { int dev, form, lds, blocks; };
    static thread_local Key cache[8] = {};
    static thread_local int used = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    for (int i = 0; i < used; ++i)
        if (cache[i].dev == dev && cache[i].form == form && cache[i].lds == lds) return cache[i].blocks;
    int per_cu = 0;
    if (lds > 64 * 1024)   // (the query honours the opt-in the launch makes as well)
        (void)hipFuncSetAttribute(fused_form_fn(form), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_form_fn(form), kLatentThreads, (size_t)lds) != hipSuccess)
        per_cu = 0;
    const int blocks = per_cu * device_cus();
    if (used < 8) cache[used++] = Key{dev, form, lds, blocks};
    return blocks;
}

int launch_forward_part(const KArgs& ka, const mopoe_adam* adam, hipStream_t s) {
    const mopoe_model& mdl = ka.mdl;
    LinArgs la;
    memset(&la, 0, sizeof(la));
    la.n = ka.st.n;
    la.counters = ka.st.backward ? ka.buf.counters : nullptr;
    la.publish = adam != nullptr;
    la.num_mods = mdl.num_mods;
    la.spins = handoff_spins();
    la.bf16 = ka.st.gemm_operands == MOPOE_OPERANDS_BF16;
#ifdef MOPOE_KNOCK
    la.knock = g_knobs.knock;
#endif
    if (adam) la.adam = *adam;
    int maxd = 1;
    for (int m = 0; m < mdl.num_mods; ++m) {
        if (!((ka.st.present_mask >> m) & 1)) continue;
        const int d = mdl.input_dim[m];
        maxd = d > maxd ? d : maxd;
        LinGroup& g = la.g[la.ngroups++];
        g.X = ka.buf.x[m];
        g.rows = ka.buf.row_index[m];
        g.W = ka.buf.params + mdl.off_w1[m];
        g.b = ka.buf.params + mdl.off_b1[m];
        g.Y = ka.buf.hidden[m];
        g.K = d;
        g.ldx = d;
        g.xrows = ka.buf.x_rows[m];
        g.ncols = kHid;
        g.ldy = kHid;
        g.relu = 1;
        g.wslack = 1;   // W1 is a segment of the flat buffer, b1 follows it
    }
    int lds = ka.lds.total * (int)sizeof(float);
    if (lds > 160 * 1024)
        return fail(MOPOE_ERR_ARG, "model exceeds the 160 KiB LDS tile budget%s");
    if (ka.lds.enc0) {
        // An encoder without a hidden layer: no encoder-layer launch, no producers -- the row
        // groups are the step's first launch (row group 0 begins the step) and read x themselves.
        static thread_local int lds_opted0 = 0;
        if (lds > 64 * 1024 && lds > lds_opted0) {
            const void* forms[] = {reinterpret_cast<const void*>(k_latent<0>), reinterpret_cast<const void*>(k_latent<1>),
                                   reinterpret_cast<const void*>(k_latent<2>), reinterpret_cast<const void*>(k_latent<3>)};
            for (const void* fn : forms) {
                hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) return fail(MOPOE_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            }
            lds_opted0 = lds;
        }
        {
            ProfScope ps(MOPOE_KERNEL_LATENT, s);
            const dim3 grid(cdiv(ka.st.n, ka.lds.rows)), block(kLatentThreads);
            switch (launch_form(ka)) {   // (the specialised bodies serve this topology too)
                case 1: hipLaunchKernelGGL(k_latent<1>, grid, block, (size_t)lds, s, ka); break;
                case 2: hipLaunchKernelGGL(k_latent<2>, grid, block, (size_t)lds, s, ka); break;
                case 3: hipLaunchKernelGGL(k_latent<3>, grid, block, (size_t)lds, s, ka); break;
                default: hipLaunchKernelGGL(k_latent<0>, grid, block, (size_t)lds, s, ka); break;
            }
        }
        return check_launch("k_latent");
    }
    // Small training batches: encoder layer and per-sample chain in ONE launch
    // (k_fused).  Needs full 16-row groups and the whole grid resident at once to pay.
    const int row_tiles = cdiv(ka.st.n, kRows);
    const bool quad = ka.lds.rows == 4 && ka.lds.quad_ok;
    if (quad && !ka.buf.wfrag)
        return fail(MOPOE_ERR_ARG, "the four-row form needs mopoe_buffers.wfrag%s");
    const int groups = cdiv(ka.st.n, quad ? 4 : kRows);   // row groups = consumer blocks
    FHead hd;
    memset(&hd, 0, sizeof(hd));
    hd.row_tiles = row_tiles;
    hd.ks = 4;
    hd.gpt = quad ? 4 : 1;
    // a modality of <= 16 columns has ONE K fragment: its K parts 1.. are zeros, the unsplit
    // form gives the same bits, and one 256-column block does for a row tile
    auto narrow = [&](int z) { return la.g[z].K <= 16; };
    // K parts (= column groups per row tile) of the wide modalities: as many as the chip
    // holds at once -- 4 x 64 columns, 2 x 128 (a block's MFMA chain is its tiles x K: twice
    // the blocks, half the chain), or one 256-column block
    auto nlin_for = [&](int ks) {
        int n = 0;
        for (int z = 0; z < la.ngroups; ++z) n += (narrow(z) ? 1 : ks) * row_tiles;
        return n;
    };
    // Four-row groups whose producers do not fit the chip beside them in 64-column blocks stay in
    // ONE launch of more blocks than CUs: the producers have the lower block numbers, so every
    // one of them holds a CU before the first row group starts to wait (no circular wait), and
    // the row groups take the CUs the producers leave.  K parts: as many as keep the launch
    // within 1.8 x the CUs (measured: method poe, 768 rows, 4 parts = 432 blocks 48.0 us, 2 parts
    // 51.3, the encoder layer as a launch of its own 52.6; 1,024 rows, 4 parts = 576 blocks 55.5,
    // 2 parts = 448 blocks 53.5, own launch 54.1).  MOPOE_QUAD_OVERSUB: -1 this rule, 0 never, k parts.
    int oversub_ks = 0;
    if (quad && groups <= fuse_blocks() && nlin_for(4) + groups > fuse_blocks()) {
        if (g_knobs.quad_oversub > 0) oversub_ks = g_knobs.quad_oversub;
        if (g_knobs.quad_oversub < 0)
            for (int k = 4; k >= 1 && !oversub_ks; k /= 2)
                if (5 * (nlin_for(k) + groups) <= 9 * fuse_blocks()) oversub_ks = k;
    }
    // (only where it buys more K parts than the launch that fits has)
    int fit_ks = 4;
    while (fit_ks > 1 && nlin_for(fit_ks) + groups > fuse_blocks()) fit_ks /= 2;
    // (two parts that fit are as good: 576 rows 46.9 / 47.5 us)
    bool oversub = oversub_ks > 0 && (nlin_for(fit_ks) + groups > fuse_blocks() || (fit_ks == 1 && oversub_ks > 1));
    if (oversub) {
        // ... and only while every PRODUCER is resident before the first row group waits: on the
        // device the call runs on, by the runtime's own occupancy figure for this kernel and
        // LDS size (not the constant 256).  Otherwise: the encoder layer as a launch of its own
        // in front of the row groups (quad_split below) -- same bits (tests/test_hip_fused.py)
        const int kp0 = round_up(maxd < kEncKChunk ? maxd : kEncKChunk, 16);
        const int lin_lds0 = (kRows + kRows * (kp0 + 4) + 4 * kRows * 68) * (int)sizeof(float);
        const int resident = resident_blocks(ka.lds.single_pass ? 4 : 5, lds > lin_lds0 ? lds : lin_lds0);
        const int cap = resident < fuse_blocks() ? resident : fuse_blocks();
        if (nlin_for(oversub_ks) > cap || groups > cap) oversub = false;
    }
    if (oversub) hd.ks = oversub_ks;
    while (hd.ks > 1 && !oversub && nlin_for(hd.ks) + groups > fuse_blocks()) hd.ks /= 2;
    int nlin = nlin_for(hd.ks);
    // K parts per modality.  One count for all is what fits when it is 4; with fewer, the parts
    // are dealt where the chains are (round 4): every modality starts with one 256-column block per
    // row tile, and the one with the longest chain K / parts doubles its parts while the grid
    // still fits -- configs[4] (7 / 444 / 128 / 64 columns, 32 row tiles): 4 parts for the
    // 444-column modality and 1 for the others = 224 producers, longest chain 128 columns' worth
    // instead of 222 with two parts for all three.  (Not for an over-subscribed launch: its rule
    // counts blocks against 1.8 x the CUs.)
    int ksz[MOPOE_MAX_MODS];
    for (int z = 0; z < MOPOE_MAX_MODS; ++z) ksz[z] = (z < la.ngroups && !narrow(z)) ? hd.ks : 1;
    if (!oversub && hd.ks < 4 && !g_knobs.uniform_ks) {
        int plan[MOPOE_MAX_MODS], blocks = la.ngroups * row_tiles;
        for (int z = 0; z < MOPOE_MAX_MODS; ++z) plan[z] = 1;
        for (;;) {
            int worst = -1, cost = 0;
            for (int z = 0; z < la.ngroups; ++z)
                if (!narrow(z) && plan[z] < 4 && la.g[z].K / plan[z] > cost) {
                    cost = la.g[z].K / plan[z];
                    worst = z;
                }
            if (worst < 0 || blocks + plan[worst] * row_tiles + groups > fuse_blocks()) break;
            // (a modality that is not the longest chain any more is not split further)
            int longest = 0;
            for (int z = 0; z < la.ngroups; ++z) longest = la.g[z].K / plan[z] > longest ? la.g[z].K / plan[z] : longest;
            if (cost < longest) break;
            blocks += plan[worst] * row_tiles;
            plan[worst] *= 2;
        }
        int chain_u = 0, chain_p = 0;
        for (int z = 0; z < la.ngroups; ++z) {
            chain_u = la.g[z].K / ksz[z] > chain_u ? la.g[z].K / ksz[z] : chain_u;
            chain_p = la.g[z].K / plan[z] > chain_p ? la.g[z].K / plan[z] : chain_p;
        }
        if (chain_p < chain_u && blocks + groups <= fuse_blocks()) {
            for (int z = 0; z < la.ngroups; ++z) ksz[z] = plan[z];
            nlin = blocks;
        }
    }
    int blocks_per_tile[MOPOE_MAX_MODS];
    for (int z = 0; z < MOPOE_MAX_MODS; ++z) {
        const bool one = ksz[z] == 1;
        hd.tiles[z] = one ? kLatentWaves : kLatentWaves / ksz[z];
        blocks_per_tile[z] = z < la.ngroups ? ksz[z] : 0;
    }
    // the widest modalities get 32-column blocks while the grid still fits the chip:
    // a block's MFMA chain is its tiles x K, and the row groups wait for the slowest
    if (hd.ks == 4)
        for (;;) {
            int worst = -1, cost = 0;
            for (int z = 0; z < la.ngroups; ++z)
                if (hd.tiles[z] <= 4 && la.g[z].K * hd.tiles[z] > cost) {
                    cost = la.g[z].K * hd.tiles[z];
                    worst = z;
                }
            if (worst < 0 || hd.tiles[worst] != 4 || la.g[worst].K < 64 ||
                oversub || nlin + 4 * row_tiles + groups > fuse_blocks())
                break;
            hd.tiles[worst] = 2;
            blocks_per_tile[worst] = 8;
            nlin += 4 * row_tiles;
        }
    for (int z = 0; z < MOPOE_MAX_MODS; ++z) {
        hd.begin[z + 1] = hd.begin[z] + blocks_per_tile[z] * row_tiles;
        hd.producers += blocks_per_tile[z];
    }
    hd.nlin = nlin;
    la.sig_n = hd.gpt;
    la.sig_stride = ka.lds.part_stride;
    la.sig_groups = groups;
    // Four-row groups whose producers would not fit the chip beside them (257..1024 rows): the
    // encoder layer runs as a launch of its own and the fused launch is row groups only --
    // all of them resident, one per CU, where sixteen-row groups would use a quarter of the CUs
    const bool quad_split = quad && !oversub && ka.st.group_rows == 0 && nlin + groups > fuse_blocks() &&
                            groups <= fuse_blocks() && !no_fuse();
    if (quad_split) {
        if (int rc = launch_linear(la, maxd, kHid, s, 0)) return rc;   // (bumps the step counter, publishes Adam's records)
        la.counters = nullptr;
        la.publish = 0;
        memset(hd.begin, 0, sizeof(hd.begin));
        memset(blocks_per_tile, 0, sizeof(blocks_per_tile));
        hd.producers = 0;
        hd.nlin = nlin = 0;
    }
    if ((ka.lds.rows == kRows || quad) && ka.st.group_rows == 0 && (nlin + groups <= fuse_blocks() || oversub) && !no_fuse()) {
        static thread_local int lds_opted_f = 0;
        const int kp = round_up(maxd < kEncKChunk ? maxd : kEncKChunk, 16);
        const int lin_lds = (kRows + kRows * (kp + 4) + 4 * kRows * 68) * (int)sizeof(float);   // (>= kRows * 260)
        if (lin_lds > lds) lds = lin_lds;
        if (lds > 64 * 1024 && lds > lds_opted_f) {
            const void* forms[] = {reinterpret_cast<const void*>(k_fused<0>), reinterpret_cast<const void*>(k_fused<1>),
                                   reinterpret_cast<const void*>(k_fused<2>), reinterpret_cast<const void*>(k_fused<3>),
                                   reinterpret_cast<const void*>(k_fused<4>), reinterpret_cast<const void*>(k_fused<5>),
                                   reinterpret_cast<const void*>(k_fused<6>)};
            for (const void* fn : forms) {
                hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) return fail(MOPOE_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            }
            lds_opted_f = lds;
        }
        FArgs fa;
        fa.hd = hd;
        fa.ka = ka;
        fa.la = la;
        {
            ProfScope ps(MOPOE_KERNEL_FUSED, s);
            const dim3 grid(nlin + groups), block(kLatentThreads);
            switch (quad ? (ka.lds.single_pass ? 4 : 5) : launch_form(ka)) {   // (the instantiations are described in latent_body)
                case 4: hipLaunchKernelGGL(k_fused<4>, grid, block, (size_t)lds, s, fa); break;
                case 5: hipLaunchKernelGGL(k_fused<5>, grid, block, (size_t)lds, s, fa); break;
                case 1: hipLaunchKernelGGL(k_fused<1>, grid, block, (size_t)lds, s, fa); break;
                case 2: hipLaunchKernelGGL(k_fused<2>, grid, block, (size_t)lds, s, fa); break;
                case 3: hipLaunchKernelGGL(k_fused<3>, grid, block, (size_t)lds, s, fa); break;
                case 6: hipLaunchKernelGGL(k_fused<6>, grid, block, (size_t)lds, s, fa); break;
                default: hipLaunchKernelGGL(k_fused<0>, grid, block, (size_t)lds, s, fa); break;
            }
        }
        return check_launch("k_fused");
    }
    // (a step the fused launch could take, kept in three launches: the fused form's K split)
    const bool fusable = (ka.lds.rows == kRows || quad) && ka.st.group_rows == 0 && nlin + groups <= fuse_blocks();
    {
        // (per-modality K parts: one encoder-layer launch per part count, so that every modality's
        //  sums come out as in the fused launch)
        bool uniform = true;
        for (int z = 0; z < la.ngroups; ++z) uniform &= ksz[z] == hd.ks || narrow(z);
        if (!fusable || uniform) {
            if (int rc = launch_linear(la, maxd, kHid, s, fusable ? hd.ks : 0)) return rc;
        } else {
            bool first = true;
            for (int k = 1; k <= 4; k *= 2) {
                LinArgs sub = la;
                sub.ngroups = 0;
                int kmax = 1;
                for (int z = 0; z < la.ngroups; ++z)
                    if (ksz[z] == k) {
                        sub.g[sub.ngroups++] = la.g[z];
                        kmax = la.g[z].K > kmax ? la.g[z].K : kmax;
                    }
                if (!sub.ngroups) continue;
                if (!first) {   // (the step begins once)
                    sub.counters = nullptr;
                    sub.publish = 0;
                }
                first = false;
                if (int rc = launch_linear(sub, kmax, kHid, s, k)) return rc;
            }
        }
    }

    static thread_local int lds_opted = 0;
    if (lds > 64 * 1024 && lds > lds_opted) {
        const void* forms[] = {reinterpret_cast<const void*>(k_latent<0>), reinterpret_cast<const void*>(k_latent<1>),
                               reinterpret_cast<const void*>(k_latent<2>), reinterpret_cast<const void*>(k_latent<3>)};
        for (const void* fn : forms) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return fail(MOPOE_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        lds_opted = lds;
    }
    {
        ProfScope ps(MOPOE_KERNEL_LATENT, s);
        const dim3 grid(cdiv(ka.st.n, ka.lds.rows)), block(kLatentThreads);
        switch (launch_form(ka)) {   // (the specialised bodies: training steps of the shapes that matter)
            case 1: hipLaunchKernelGGL(k_latent<1>, grid, block, (size_t)lds, s, ka); break;
            case 2: hipLaunchKernelGGL(k_latent<2>, grid, block, (size_t)lds, s, ka); break;
            case 3: hipLaunchKernelGGL(k_latent<3>, grid, block, (size_t)lds, s, ka); break;
            default: hipLaunchKernelGGL(k_latent<0>, grid, block, (size_t)lds, s, ka); break;
        }
    }
    return check_launch("k_latent");
}

void build_wargs(const KArgs& ka, const mopoe_adam* adam, WArgs& w, const XgPeers* xg = nullptr) {
    const mopoe_model& mdl = ka.mdl;
    const mopoe_step& st = ka.st;
    memset(&w, 0, sizeof(w));
    int tile = 0;
    for (int m = 0; m < mdl.num_mods; ++m) {
        if (!((st.present_mask >> m) & 1)) continue;
        int njobs_m = 0;
        for (int j = 0; j < st.num_jobs; ++j) njobs_m += st.job_mod[j] == m;
        const int d = mdl.input_dim[m], nh = heads_dim(mdl, m), zd = z_dim(mdl, m);
        WJob jobs[3] = {
            {ka.buf.g_pre[m], ka.buf.x[m], ka.buf.row_index[m], kHid, kHid, d, d, st.n,
             mdl.off_w1[m], mdl.off_b1[m], 0, 0, m, ka.buf.x_rows[m]},
            {ka.buf.g_heads[m], ka.buf.hidden[m], nullptr, nh, nh, kHid, kHid, st.n,
             mdl.off_wh[m], mdl.off_bh[m], 0, 0, m, st.n},
            {ka.buf.g_xhat[m], ka.buf.z[m], nullptr, d, d, ldz_glb(mdl, m), zd,
             njobs_m * st.n, mdl.off_wd[m], mdl.off_bd[m], 0, 0, m, njobs_m * st.n},
        };
        const WFrag wf = wfrag_layout(mdl);
        jobs[0].wf_off = jobs[1].wf_off = jobs[2].wf_off = -1;
        if (ka.buf.wfrag) {
            jobs[1].wf_off = wf.whf[m], jobs[1].wf_k4 = kHid / 4;
            jobs[2].wf_off = wf.wdf[m], jobs[2].wf_k4 = wf.k4d[m];
        }
        if (ka.lds.enc0) {   // the heads sit on x: Wh's gradient is g_heads^T x, there is no W1
            jobs[1].X = ka.buf.x[m];
            jobs[1].xrows = ka.buf.row_index[m];
            jobs[1].ldx = jobs[1].xcols = d;
            jobs[1].xtotal = ka.buf.x_rows[m];
            jobs[1].wf_off = -1;
        }
        WJob head = jobs[2];   // (ss: the logvar head's weights -- G = d loss / d logvar, X = z like the decoder's)
        if (ka.lds.ss) {
            int j0 = 0;
            while (j0 < st.num_jobs && st.job_mod[j0] != m) ++j0;
            if (j0 < st.num_jobs) {
                head.G = ka.lds.dj[j0].g_lv;
                head.off_w = ka.lds.dj[j0].off_wlv;
                head.off_b = ka.lds.dj[j0].off_blv;
                head.wf_off = -1;
            }
        }
        for (int k = ka.lds.enc0 ? 1 : 0; k < (ka.lds.ss && njobs_m > 0 ? 4 : 3); ++k) {
            WJob jb_local = k < 3 ? jobs[k] : head;
            WJob& jb = jb_local;
            // + the bias column, unless the rows are a whole number of tiles (WJob::fold)
            // (small batches only: there the count of blocks decides; with tens of thousands of
            //  rows every block is MFMA-bound and the extra adds make the folded ones the last)
            jb.fold = jb.xcols % 32 == 0 && st.n <= 2048 && !g_knobs.wgrad_nofold;
            jb.tiles_j = cdiv(jb.xcols + (jb.fold ? 0 : 1), 32);
            // (a job over at least twice the step's batch rows: 16-row tiles, see WJob::th)
            jb.th = (k == 2 && jb.R >= 2 * st.n && !g_knobs.wgrad_tall) ? 16 : 32;
            jb.tile_begin = tile;
            w.tile_begin[w.njobs] = tile;
            tile += cdiv(jb.gcols, jb.th) * jb.tiles_j;
            w.jobs[w.njobs++] = jb;
        }
    }
    w.total_tiles = tile;
    int lb = 0;
    for (int m = 0; m < MOPOE_MAX_MODS; ++m) {
        w.lvo_block_begin[m] = lb;
        // (with the logvar head there is no learnt logvar vector and no partial sums of its gradient)
        if (!ka.lds.ss && m < mdl.num_mods && ((st.present_mask >> m) & 1)) lb += cdiv(mdl.input_dim[m], 64);
    }
    w.lvo_block_begin[MOPOE_MAX_MODS] = lb;
    w.lvo_blocks = lb;
    w.fuse_adam = adam != nullptr;
    if (adam) w.adam = *adam;
    if (xg) w.xg = *xg;
}

#include "mopoe_wgrad_big.inc"

bool default_topology(const mopoe_topology* tp);

// `tp`: the model's topology (nullptr / the default one: the layout of mopoe_model_layout).
// A modality's encoder parameters are one contiguous run of the flat buffer and so are its
// decoder's (mopoe_topology_layout keeps it that way): two segments per modality.
int build_adam_segs(const mopoe_model& mdl, int32_t present_mask, const mopoe_adam& adam,
                    int32_t world, AdamSegs& sg, const mopoe_topology* tp = nullptr) {
    memset(&sg, 0, sizeof(sg));
    if (world < 1) return fail(MOPOE_ERR_ARG, "world must be >= 1%s");
    if (!default_topology(tp)) {
        for (int m = 0; m < mdl.num_mods; ++m) {
            if (!((present_mask >> m) & 1)) continue;
            const int d = mdl.input_dim[m];
            sg.seg_mod[sg.nseg] = m;
            sg.wf_src[sg.nseg] = -1;
            sg.begin[sg.nseg] = tp->enc_layers > 0 ? tp->off_we[m][0] : mdl.off_wh[m];
            sg.end[sg.nseg++] = mdl.off_bh[m] + heads_dim(mdl, m);
            sg.seg_mod[sg.nseg] = m;
            sg.wf_src[sg.nseg] = -1;
            sg.begin[sg.nseg] = tp->dec_layers > 0 ? tp->off_wg[m][0] : mdl.off_wd[m];
            sg.end[sg.nseg++] = tp->sample_scale ? tp->off_blv[m] + d
                                : mdl.learn_output_scale ? mdl.off_lvo[m] + d : mdl.off_bd[m] + d;
        }
        if (sg.nseg == 0) return fail(MOPOE_ERR_ARG, "empty present_mask%s");
        sg.num_mods = mdl.num_mods;
        sg.present_mask = present_mask;
        sg.world = world;
        sg.off_ctrl = mdl.off_ctrl;
        sg.grad_scale = 1.0f / (float)world;
        sg.adam = adam;
        return 0;
    }
    const WFrag wf = wfrag_layout(mdl);
    sg.wf_total = wf.total;
    for (int m = 0; m < mdl.num_mods; ++m) {
        if (!((present_mask >> m) & 1)) continue;
        sg.seg_mod[sg.nseg] = m;
        sg.begin[sg.nseg] = mdl.off_w1[m];
        sg.wf_src[sg.nseg] = mdl.off_wh[m];
        sg.wf_count[sg.nseg] = heads_dim(mdl, m) * kHid;
        sg.wf_k[sg.nseg] = kHid;
        sg.wf_k4[sg.nseg] = kHid / 4;
        sg.wf_dst[sg.nseg] = wf.whf[m];
        sg.end[sg.nseg++] = mdl.off_bh[m] + heads_dim(mdl, m);
        sg.seg_mod[sg.nseg] = m;
        sg.begin[sg.nseg] = mdl.off_wd[m];
        sg.wf_src[sg.nseg] = mdl.off_wd[m];
        sg.wf_count[sg.nseg] = mdl.input_dim[m] * z_dim(mdl, m);
        sg.wf_k[sg.nseg] = z_dim(mdl, m);
        sg.wf_k4[sg.nseg] = wf.k4d[m];
        sg.wf_dst[sg.nseg] = wf.wdf[m];
        sg.end[sg.nseg++] = mdl.learn_output_scale ? mdl.off_lvo[m] + mdl.input_dim[m]
                                                   : mdl.off_bd[m] + mdl.input_dim[m];
    }
    if (sg.nseg == 0) return fail(MOPOE_ERR_ARG, "empty present_mask%s");
    sg.num_mods = mdl.num_mods;
    sg.present_mask = present_mask;
    sg.world = world;
    sg.off_ctrl = mdl.off_ctrl;
    sg.grad_scale = 1.0f / (float)world;
    sg.adam = adam;
    return 0;
}

void comm_next(mopoe_comm* c, XgPeers& x);
int comm_flag_stride(const mopoe_comm* c);
int comm_check(const mopoe_comm* c, const mopoe_model* mdl);
int comm_world(const mopoe_comm* c);
int launch_wfrag(const mopoe_model& mdl, const mopoe_buffers& buf, hipStream_t s);

// k_adam on the flat buffers with the gradient scaled by 1 / world.  `ctrl_check`: the
// gradient buffer went through an all-reduce that also summed its control words (the ranks'
// modality masks and invalid flags: adam_step_valid); false when the exchange compared the
// masks itself (the xGMI forms: the masks travel in the arrival flags).
int launch_adam(const mopoe_model& mdl, int32_t present_mask, const mopoe_buffers& buf,
                const mopoe_adam& adam, int32_t world, bool ctrl_check, hipStream_t s,
                const mopoe_topology* tp = nullptr) {
    if (!buf.params || !buf.grads || !buf.exp_avg || !buf.exp_avg_sq || !buf.counters)
        return fail(MOPOE_ERR_ARG, "null optimiser buffer%s");
    AdamSegs sg;
    if (int rc = build_adam_segs(mdl, present_mask, adam, world, sg, tp)) return rc;
    if (!ctrl_check) sg.world = 1;   // (grad_scale stays 1 / world)
    {
        ProfScope ps(MOPOE_KERNEL_ADAM, s);
        hipLaunchKernelGGL(k_adam, dim3(64, sg.nseg), dim3(256), 0, s, buf, sg);
    }
    return check_launch("k_adam");   // (the fragment-major weight copies included)
}

}  // namespace
extern "C" int64_t mopoe_wgrad_scratch_floats(const mopoe_model* mdl, const mopoe_step* st);
namespace {

// `fuse_adam`: the weight-gradient launch applies the update itself (the one-rank step).
// Otherwise the launches stop at the gradients -- with `adam` non-NULL the step's first
// kernel still publishes its Adam records for the k_adam launch that follows an exchange.
// the logvar head's tensors and parameter offsets into the decoder jobs' records (LatentLds::ss)
int bind_logvar_head(KArgs& ka, const mopoe_topology* tp, const mopoe_gbuffers* gb, bool train) {
    if (!ka.lds.ss) return 0;
    if (!tp || !gb) return fail(MOPOE_ERR_ARG, "the logvar head needs its topology and buffers%s");
    for (int j = 0; j < ka.st.num_jobs; ++j) {
        const int m = ka.st.job_mod[j];
        if (!gb->lv[m] || (train && !gb->g_lv[m])) return fail(MOPOE_ERR_ARG, "null logvar head buffer%s");
        ka.lds.dj[j].lv = gb->lv[m];
        ka.lds.dj[j].g_lv = gb->g_lv[m];
        ka.lds.dj[j].off_wlv = tp->off_wlv[m];
        ka.lds.dj[j].off_blv = tp->off_blv[m];
    }
    return 0;
}

int train_step_impl(const mopoe_model* mdl, const mopoe_step* st, const mopoe_buffers* buf,
                    const mopoe_adam* adam, mopoe_comm* comm, bool fuse_adam, void* stream,
                    const mopoe_topology* tp = nullptr, const mopoe_gbuffers* gb = nullptr) {
    if (int rc = validate(mdl, st, buf, true)) return rc;
    if (adam && (!buf->exp_avg || !buf->exp_avg_sq))
        return fail(MOPOE_ERR_ARG, "null Adam state%s");
    if (comm) {
        if (!adam) return fail(MOPOE_ERR_ARG, "the exchanging step applies Adam: adam is NULL%s");
        if (int rc = comm_check(comm, mdl)) return rc;
    }
    KArgs ka;
    ka.mdl = *mdl;
    ka.st = *st;
    bind_buffers(ka, *buf);
    ka.st.backward = 1;
    ka.st.sample = 1;
    step_layout(ka.mdl, ka.st, ka.lds);
    latent_bind(ka.lds, ka.buf);
    if (int rc = bind_logvar_head(ka, tp, gb, true)) return rc;
    ka.lds.enc0_publish = adam != nullptr;
    if (adam) ka.lds.enc0_adam = *adam;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WArgs w;
    // (the exchanging launch does not apply the update itself: whether every block's exchange
    //  was good is known at the END of the launch, and a step is applied whole or not at all
    //  -- k_adam behind it decides for the whole step)
    build_wargs(ka, (comm || !fuse_adam) ? nullptr : adam, w);
    const dim3 grid(w.total_tiles + w.lvo_blocks + 1);
    // (checked before anything is launched: a refused call leaves no half step behind and
    //  does not advance the exchange's sequence number)
    if (comm && (int)grid.x > comm_flag_stride(comm))   // one arrival flag per exchanging workgroup
        return fail(MOPOE_ERR_ARG, "more weight-gradient blocks than the communicator has flags%s");
    // (the split weight-gradient launches write `need` floats of scratch: the count depends on
    //  the batch's modalities as well as on n, so it is checked here, before any launch)
    const bool big = !comm && wgrad_big_step(ka.st) && buf->wgrad_scratch;
    if (big) {
        const int64_t need = mopoe_wgrad_scratch_floats(mdl, st);
        if (buf->wgrad_scratch_floats < need)
            return fail(MOPOE_ERR_ARG, "mopoe_buffers.wgrad_scratch holds fewer floats than "
                                       "mopoe_wgrad_scratch_floats(model, step)%s");
    }
    if (int rc = launch_forward_part(ka, adam, s)) return rc;
    if (big) {
        // a large batch: 64 x 64 tiles over slices of the batch rows, the parts added in order
        // by a second launch (+ Adam), then the launch's tail alone -- decoder-logvar blocks,
        // the step's scalars and bookkeeping (mopoe_wgrad_big.inc)
        WbArgs wb;
        const int64_t wb_floats = build_wbargs(ka, (comm || !fuse_adam) ? nullptr : adam, wb);
        wb.scratch = buf->wgrad_scratch;
        // the row groups' partial slabs (thousands of them) are summed in kFoldSlices slices by
        // a launch of their own; the tail's few blocks then add kFoldSlices slabs instead of
        // walking all of them (65,536 rows: 47 -> ~10 us)
        KArgs kt = ka;
        const int groups = cdiv(ka.st.n, ka.lds.rows);
        const bool fold = groups >= 8 * kFoldSlices;
        if (fold) {
            kt.buf.partials = buf->wgrad_scratch + wb_floats;
            kt.lds.fold_tiles = kFoldSlices;
        }
        {
            ProfScope ps(MOPOE_KERNEL_WGRAD, s);
            hipLaunchKernelGGL(k_wgrad_big, dim3(round_up(wb.total_blocks, 8)), dim3(256), 0, s, wb);
            hipLaunchKernelGGL(k_wgrad_big_reduce, dim3(wb.total_tiles * kWbReduceParts), dim3(256), 0, s, ka.buf, wb);
            if (fold)
                hipLaunchKernelGGL(k_partials_fold, dim3(cdiv(ka.lds.part_stride, 256), kFoldSlices), dim3(256), 0, s,
                                   (const float*)buf->partials, kt.buf.partials, groups, ka.lds.part_stride);
            w.total_tiles = 0;
            w.njobs = 0;
            hipLaunchKernelGGL((k_wgrad<8, false>), dim3(w.lvo_blocks + 1), dim3(512), 0, s, kt, w);
        }
        if (int rc = check_launch("k_wgrad_big")) return rc;
        if (buf->wfrag && wb.fuse_adam) return launch_wfrag(*mdl, *buf, s);   // (the copies follow)
        return 0;
    }
    if (comm) {
        comm_next(comm, w.xg);
        w.xg.mask = ka.st.present_mask;
    }
    {
        ProfScope ps(MOPOE_KERNEL_WGRAD, s);
        // eight waves per block (half the MFMA chain per wave) for large batches, and from 256
        // rows on while every block still has a CU of its own (two 512-thread blocks do not
        // fit one: registers) -- measured: configs[1] -0.8 us, configs[4] (277 blocks) +2 us
#if MOPOE_WGRAD_W16
        // sixteen waves per block from 1,024 rows on while every block has a CU of its own: the
        // MFMA chain of a block is what it is, but four waves per SIMD cover one another's load
        // round trips (eight waves: two rounds of requests per wave, each waited for)
        if (!comm && ka.st.n >= 1024 && grid.x <= (unsigned)device_cus()) {
            hipLaunchKernelGGL((k_wgrad<16, false>), grid, dim3(1024), 0, s, ka, w);
        } else
#endif
        // ... and where the launch has MORE blocks than CUs (four modalities: 277; the general
        // chain's ten jobs: 441) the lean eight-wave form, two blocks to a CU: configs[4] 60.05 ->
        // 58.8 us per step (k_wgrad 14.5 -> 13.4), the chain at 2 + 1 hidden layers 87.9 -> 86.5
        if (!comm && g_knobs.wgrad_lean8 && ka.st.n >= 256 && grid.x > (unsigned)device_cus() &&
            grid.x <= 2u * (unsigned)device_cus()) {
            hipLaunchKernelGGL((k_wgrad<8, false, true>), grid, dim3(512), 0, s, ka, w);
        } else if (ka.st.n > 512 || (ka.st.n >= 256 && grid.x <= 256)) {
            if (comm)
                hipLaunchKernelGGL((k_wgrad<8, true>), grid, dim3(512), 0, s, ka, w);
            else
                hipLaunchKernelGGL((k_wgrad<8, false>), grid, dim3(512), 0, s, ka, w);
        } else {
            if (comm)
                hipLaunchKernelGGL((k_wgrad<4, true>), grid, dim3(256), 0, s, ka, w);
            else
                hipLaunchKernelGGL((k_wgrad<4, false>), grid, dim3(256), 0, s, ka, w);
        }
    }
    if (int rc = check_launch("k_wgrad")) return rc;
    if (comm) return launch_adam(*mdl, ka.st.present_mask, *buf, *adam, comm_world(comm), false, s);
    return 0;
}

int launch_wfrag(const mopoe_model& mdl, const mopoe_buffers& buf, hipStream_t s) {
    const int pieces = wfrag_layout(mdl).total / 4;
    hipLaunchKernelGGL(k_wfrag, dim3(cdiv(pieces, 256) < 256 ? cdiv(pieces, 256) : 256), dim3(256), 0, s, mdl,
                       buf.params, buf.wfrag);
    return check_launch("k_wfrag");
}

#include "mopoe_general.inc"

}  // namespace

#include "mopoe_xgmi.inc"

extern "C" {

int mopoe_abi_version(void) { return MOPOE_ABI_VERSION; }

int mopoe_reload_knobs(void) {
    g_knobs = read_knobs();
    return 0;
}

const char* mopoe_last_error(void) { return g_err; }

int mopoe_model_layout(mopoe_model* mdl) {
    if (!mdl) return fail(MOPOE_ERR_ARG, "null model%s");
    if (mdl->num_mods < 1 || mdl->num_mods > MOPOE_MAX_MODS)
        return fail(MOPOE_ERR_ARG, "num_mods out of range%s");
    int off = 0;
    auto seg = [&off](int count) {
        const int o = off;
        off += round_up(count, 64);
        return o;
    };
    for (int m = 0; m < mdl->num_mods; ++m) {
        const int d = mdl->input_dim[m], nh = heads_dim(*mdl, m), zd = z_dim(*mdl, m);
        if (d < 1 || mdl->style_dim[m] < 0) return fail(MOPOE_ERR_ARG, "bad modality dims%s");
        mdl->off_w1[m] = seg(kHid * d);
        mdl->off_b1[m] = seg(kHid);
        mdl->off_wh[m] = seg(nh * kHid);
        mdl->off_bh[m] = seg(nh);
        mdl->off_wd[m] = seg(d * zd);
        mdl->off_bd[m] = seg(d);
        mdl->off_lvo[m] = seg(d);
    }
    mdl->off_ctrl = seg(64);   // control words of the gradient buffer (no parameters)
    mdl->num_floats = off;
    return 0;
}

// Layout of a general topology (mopoe_topology): per modality the encoder's parameters in
// one run -- hidden layers, then the heads -- and the decoder's in another -- hidden layers,
// out_mu, then the logvar parameter or the logvar head.  The default topology gives exactly
// mopoe_model_layout's offsets.
int mopoe_topology_layout(mopoe_model* mdl, mopoe_topology* tp) {
    if (!mdl || !tp) return fail(MOPOE_ERR_ARG, "null model%s");
    if (mdl->num_mods < 1 || mdl->num_mods > MOPOE_MAX_MODS)
        return fail(MOPOE_ERR_ARG, "num_mods out of range%s");
    if (int rc = default_topology(tp) ? 0 : validate_topology(*mdl, *tp)) return rc;
    int off = 0;
    auto seg = [&off](int count) {
        const int o = off;
        off += round_up(count, 64);
        return o;
    };
    for (int m = 0; m < MOPOE_MAX_MODS; ++m)
        for (int l = 0; l < MOPOE_MAX_LAYERS; ++l)
            tp->off_we[m][l] = tp->off_be[m][l] = tp->off_wg[m][l] = tp->off_bg[m][l] = 0;
    for (int m = 0; m < mdl->num_mods; ++m) {
        const int d = mdl->input_dim[m], nh = heads_dim(*mdl, m), zd = z_dim(*mdl, m);
        if (d < 1 || mdl->style_dim[m] < 0) return fail(MOPOE_ERR_ARG, "bad modality dims%s");
        for (int l = 0; l < tp->enc_layers; ++l) {
            tp->off_we[m][l] = seg(kHid * (l == 0 ? d : kHid));
            tp->off_be[m][l] = seg(kHid);
        }
        mdl->off_w1[m] = tp->off_we[m][0];
        mdl->off_b1[m] = tp->off_be[m][0];
        mdl->off_wh[m] = seg(nh * (tp->enc_layers > 0 ? kHid : d));
        mdl->off_bh[m] = seg(nh);
        for (int l = 0; l < tp->dec_layers; ++l) {
            tp->off_wg[m][l] = seg(kHid * (l == 0 ? zd : kHid));
            tp->off_bg[m][l] = seg(kHid);
        }
        const int dw = tp->dec_layers > 0 ? kHid : zd;
        mdl->off_wd[m] = seg(d * dw);
        mdl->off_bd[m] = seg(d);
        mdl->off_lvo[m] = off;   // (with the logvar head: no parameter, the offset stays readable)
        tp->off_wlv[m] = tp->off_blv[m] = off;
        if (tp->sample_scale) {
            tp->off_wlv[m] = seg(d * dw);
            tp->off_blv[m] = seg(d);
        } else {
            mdl->off_lvo[m] = seg(d);
        }
    }
    mdl->off_ctrl = seg(64);
    mdl->num_floats = off;
    return 0;
}

int mopoe_general_enc_blocks(const mopoe_topology* tp, const mopoe_step* st, int train) {
    if (!tp || !st) return 1;
    return general_enc_blocks(*tp, *st, train != 0);
}

// Two topologies off the default run in the row-group kernel instead of the general chain of
// launches (round 4), alone or together:
//   * an encoder WITHOUT a hidden layer (the default topology minus its encoder layer: the heads
//     GEMM reads the x tile, K = d_m; no producers, no dL/dh stage) -- pad_ bit 0, LatentLds::enc0;
//   * the decoder's logvar HEAD (learn_output_sample_scale: a second output head, per-sample
//     scale; the default's fused launch with two sets of decoder accumulators, a second dL/dz
//     contribution and a fourth weight-gradient job per modality) -- pad_ bit 1, LatentLds::ss.
// Hidden decoder layers, more than one hidden encoder layer and dropout take the chain.  Returns
// true and fills (st2, b2) when the step takes the kernel: the step with its pad_ bits, the
// buffers with the hidden layer's tensors where the default topology has them (or stand-ins for
// the layer that does not exist: validate() wants them non-null, no kernel touches them).
static bool rowgroup_route(const mopoe_model* mdl, const mopoe_topology* tp, const mopoe_step* st,
                           const mopoe_buffers* buf, const mopoe_gbuffers* gb, bool train,
                           mopoe_step& st2, mopoe_buffers& b2) {
    if (!mdl || !tp || !st || !buf || !gb || g_knobs.enc0_chain) return false;
    if (tp->enc_layers > 1 || tp->dec_layers != 0 || tp->dropout != 0.f) return false;
    const int bits = (tp->enc_layers == 0 ? 1 : 0) | (tp->sample_scale ? 2 : 0);
    if (!bits) return false;
    int npres = 0;
    for (int m = 0; m < mdl->num_mods; ++m) npres += (st->present_mask >> m) & 1;
    if ((bits & 2) && 4 * npres > 3 * MOPOE_MAX_MODS) return false;   // (k_wgrad's job table)
    st2 = *st;
    st2.pad_ = bits;
    st2.rows_per_group = 0;
    st2.backward = train ? 1 : 0;
    LatentLds L;
    step_layout(*mdl, st2, L);
    if (!L.fits || L.rows != kRows) return false;   // (a general workspace has one slab per 16 rows)
    if ((bits & 2) && L.s3_nt != 2) return false;    // (the head's decoder unit exists in its two-tile form)
    b2 = *buf;
    for (int m = 0; m < MOPOE_MAX_MODS; ++m) {
        const bool layer = tp->enc_layers == 1 && m < mdl->num_mods;
        b2.hidden[m] = layer ? gb->enc_act[m][0] : buf->heads[m];
        b2.g_pre[m] = layer ? gb->g_enc[m][0] : buf->g_heads[m];
        if (layer && ((st->present_mask >> m) & 1) && (!b2.hidden[m] || (train && !b2.g_pre[m]))) return false;
    }
    b2.wgrad_scratch = nullptr;   // (the split weight-gradient launches know the default topology's jobs only)
    b2.wfrag = nullptr;
    return true;
}
int forward_impl(const mopoe_model* mdl, const mopoe_step* st, const mopoe_buffers* buf, void* stream,
                 const mopoe_topology* tp, const mopoe_gbuffers* gb);

int mopoe_general_forward(const mopoe_model* mdl, const mopoe_topology* tp, const mopoe_step* st,
                          const mopoe_buffers* buf, const mopoe_gbuffers* gb, void* stream) {
    {
        mopoe_step st2;
        mopoe_buffers b2;
        if (rowgroup_route(mdl, tp, st, buf, gb, false, st2, b2)) return forward_impl(mdl, &st2, &b2, stream, tp, gb);
    }
    GArgs ga;
    GeneralPlan gp;
    if (int rc = general_args(mdl, tp, st, buf, gb, false, ga, gp)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = general_forward_part(ga, gp, nullptr, s)) return rc;
    {
        ProfScope ps(MOPOE_KERNEL_FINALIZE, s);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, s, ga.ka);
    }
    return check_launch("k_finalize");
}

int mopoe_general_adam_step(const mopoe_model* mdl, const mopoe_topology* tp, int32_t present_mask,
                            const mopoe_buffers* buf, const mopoe_adam* adam, int32_t world, void* stream) {
    if (!mdl || !tp || !buf || !adam) return fail(MOPOE_ERR_ARG, "null descriptor%s");
    return launch_adam(*mdl, present_mask, *buf, *adam, world, true, static_cast<hipStream_t>(stream), tp);
}

int mopoe_profile_enable(int enable) {
    g_prof_on = enable != 0;
    return 0;
}

int mopoe_profile_read(int32_t* count, float* total_ms) {
    if (!count || !total_ms) return fail(MOPOE_ERR_ARG, "mopoe_profile_read: null%s");
    for (int k = 0; k < MOPOE_NUM_KERNELS; ++k) {
        count[k] = 0;
        total_ms[k] = 0.f;
    }
    for (const ProfRec& r : g_prof) {
        hipError_t e = hipEventSynchronize(r.end);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.beg, r.end);
        if (e != hipSuccess)
            return fail(MOPOE_ERR_HIP, "mopoe_profile_read: %s", hipGetErrorString(e));
        count[r.kernel] += 1;
        total_ms[r.kernel] += ms;
        g_prof_pool.push_back(r.beg);
        g_prof_pool.push_back(r.end);
    }
    g_prof.clear();
    return 0;
}

int mopoe_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(mopoe_model);
        case 1: return (int)sizeof(mopoe_step);
        case 2: return (int)sizeof(mopoe_buffers);
        case 3: return (int)sizeof(mopoe_adam);
        case 4: return (int)offsetof(mopoe_step, job_eps_content);
        case 5: return (int)offsetof(mopoe_step, comp_w);
        case 6: return (int)offsetof(mopoe_buffers, partials);
        case 8: return (int)offsetof(mopoe_buffers, status_host);
        case 9: return (int)offsetof(mopoe_model, off_ctrl);
        case 10: return (int)offsetof(mopoe_buffers, wfrag);
        case 11: return (int)sizeof(mopoe_topology);
        case 12: return (int)sizeof(mopoe_gbuffers);
        case 13: return (int)offsetof(mopoe_gbuffers, keep_enc);
        case 7: return (int)offsetof(mopoe_model, num_floats);
        default: return -1;
    }
}

int mopoe_ldz(const mopoe_model* mdl, int mod) { return ldz_glb(*mdl, mod); }

int mopoe_partials_stride(const mopoe_model* mdl) { return partials_stride(*mdl); }
int64_t mopoe_wgrad_scratch_floats(const mopoe_model* mdl, const mopoe_step* st) {
    if (!mdl || !st || st->n < 1) return 0;
    if (!st->backward) {   // mopoe_forward: kFoldSlices pre-summed slabs, from 512 row groups on
        LatentLds L;
        step_layout(*mdl, *st, L);
        return cdiv(st->n, L.rows) >= 8 * kFoldSlices ? (int64_t)kFoldSlices * partials_stride(*mdl) : 0;
    }
    if (!wgrad_big_step(*st)) return 0;
    KArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.mdl = *mdl;
    ka.st = *st;
    WbArgs wb;
    return build_wbargs(ka, nullptr, wb) + (int64_t)kFoldSlices * partials_stride(*mdl);
}
int mopoe_wfrag_floats(const mopoe_model* mdl) { return mdl ? wfrag_layout(*mdl).total : 0; }
int mopoe_row_groups(const mopoe_model* mdl, const mopoe_step* st) {
    if (!mdl || !st || st->n < 1) return 0;
    LatentLds L;
    step_layout(*mdl, *st, L);
    return cdiv(st->n, L.rows);
}

int mopoe_latent_lds_bytes(const mopoe_model* mdl, const mopoe_step* st) {
    return latent_lds_bytes(*mdl, *st);
}

int mopoe_forward(const mopoe_model* mdl, const mopoe_step* st, const mopoe_buffers* buf,
                  void* stream) {
    return forward_impl(mdl, st, buf, stream, nullptr, nullptr);
}

int forward_impl(const mopoe_model* mdl, const mopoe_step* st, const mopoe_buffers* buf, void* stream,
                 const mopoe_topology* tp, const mopoe_gbuffers* gb) {
    if (int rc = validate(mdl, st, buf, false)) return rc;
    KArgs ka;
    ka.mdl = *mdl;
    ka.st = *st;
    bind_buffers(ka, *buf);
    ka.st.backward = 0;
    step_layout(ka.mdl, ka.st, ka.lds);
    latent_bind(ka.lds, ka.buf);
    if (int rc = bind_logvar_head(ka, tp, gb, false)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (buf->wgrad_scratch && cdiv(ka.st.n, ka.lds.rows) >= 8 * kFoldSlices &&
        buf->wgrad_scratch_floats < (int64_t)kFoldSlices * ka.lds.part_stride)
        return fail(MOPOE_ERR_ARG, "mopoe_buffers.wgrad_scratch holds fewer floats than "
                                   "mopoe_wgrad_scratch_floats(model, step)%s");
    if (int rc = launch_forward_part(ka, nullptr, s)) return rc;
    {
        ProfScope ps(MOPOE_KERNEL_FINALIZE, s);
        // thousands of row groups (folded DAA inference): their slabs are summed in kFoldSlices (64)
        // slices by a launch of many blocks first -- the one finalising block walked 3,125
        // slabs in 27 us at 50,000 rows -- when the caller gave the scratch for it
        const int groups = cdiv(ka.st.n, ka.lds.rows);
        if (buf->wgrad_scratch && groups >= 8 * kFoldSlices) {
            hipLaunchKernelGGL(k_partials_fold, dim3(cdiv(kStatStride, 256), kFoldSlices), dim3(256), 0, s,
                               (const float*)buf->partials, buf->wgrad_scratch, groups, ka.lds.part_stride);
            ka.buf.partials = buf->wgrad_scratch;
            ka.lds.fold_tiles = kFoldSlices;
        }
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, s, ka);
    }
    return check_launch("k_finalize");
}

int mopoe_train_step(const mopoe_model* mdl, const mopoe_step* st, const mopoe_buffers* buf,
                     const mopoe_adam* adam, void* stream) {
    return train_step_impl(mdl, st, buf, adam, nullptr, true, stream);
}

int mopoe_adam_step(const mopoe_model* mdl, int32_t present_mask, const mopoe_buffers* buf,
                    const mopoe_adam* adam, int32_t world, void* stream) {
    if (!mdl || !buf || !adam) return fail(MOPOE_ERR_ARG, "null descriptor%s");
    return launch_adam(*mdl, present_mask, *buf, *adam, world, true, static_cast<hipStream_t>(stream));
}

int mopoe_wfrag_refresh(const mopoe_model* mdl, const mopoe_buffers* buf, void* stream) {
    if (!mdl || !buf || !buf->params || !buf->wfrag) return fail(MOPOE_ERR_ARG, "mopoe_wfrag_refresh: null argument%s");
    return launch_wfrag(*mdl, *buf, static_cast<hipStream_t>(stream));
}

int mopoe_linear(const float* x, int32_t n, int32_t k, const float* w, const float* b,
                 int32_t ncols, int32_t relu, float* y, void* stream) {
    if (!x || !w || !y || n < 1 || k < 1 || ncols < 1)
        return fail(MOPOE_ERR_ARG, "mopoe_linear: bad argument%s");
    LinArgs la;
    memset(&la, 0, sizeof(la));
    la.n = n;
    la.ngroups = 1;
    LinGroup& g = la.g[0];
    g.X = x;
    g.W = w;
    g.b = b;
    g.Y = y;
    g.K = k;
    g.ldx = k;
    g.xrows = n;
    g.ncols = ncols;
    g.ldy = ncols;
    g.relu = relu;
    return launch_linear(la, k, ncols, static_cast<hipStream_t>(stream));
}

int mopoe_poe(const float* mu, const float* logvar, int32_t num_experts, int64_t numel,
              float eps, float* out_mu, float* out_logvar, void* stream) {
    if (!mu || !logvar || !out_mu || !out_logvar || num_experts < 1 || numel < 1)
        return fail(MOPOE_ERR_ARG, "mopoe_poe: bad argument%s");
    const int blocks = (int)((numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_poe, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), mu,
                       logvar, num_experts, (long long)numel, eps, out_mu, out_logvar);
    return check_launch("k_poe");
}

int mopoe_kl_divergence(const float* mu, const float* logvar, int64_t numel, float norm_value,
                        float* scratch, float* out, void* stream) {
    if (!mu || !logvar || !scratch || !out || numel < 1)
        return fail(MOPOE_ERR_ARG, "mopoe_kl_divergence: bad argument%s");
    const int blocks = (int)((numel + 255) / 256 < 1024 ? (numel + 255) / 256 : 1024);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_kl_partial, dim3(blocks), dim3(256), 0, s, mu, logvar,
                       (long long)numel, scratch);
    if (int rc = check_launch("k_kl_partial")) return rc;
    hipLaunchKernelGGL(k_kl_final, dim3(1), dim3(64), 0, s, scratch, blocks, norm_value, out);
    return check_launch("k_kl_final");
}

int mopoe_reparameterize(const float* mu, const float* logvar, const float* eps, int64_t numel,
                         uint64_t seed, uint64_t stream_id, float* out, void* stream) {
    if (!mu || !logvar || !out || numel < 1)
        return fail(MOPOE_ERR_ARG, "mopoe_reparameterize: bad argument%s");
    const int blocks = (int)((numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_reparam, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       mu, logvar, eps, (long long)numel, seed, (uint32_t)stream_id, out);
    return check_launch("k_reparam");
}

int mopoe_mixture_select(const float* mus, const float* logvars, int32_t num_comp, int32_t n,
                         int32_t d, const int32_t* bounds, float* out_mu, float* out_logvar,
                         void* stream) {
    if (!mus || !logvars || !bounds || !out_mu || !out_logvar || num_comp < 1 || n < 1 || d < 1)
        return fail(MOPOE_ERR_ARG, "mopoe_mixture_select: bad argument%s");
    const long long total = (long long)n * d;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_mix_select, dim3(blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), mus, logvars, num_comp, n, d, bounds,
                       out_mu, out_logvar);
    return check_launch("k_mix_select");
}

// ---- data-parallel gradient exchange over xGMI peer windows (mopoe_xgmi.inc) ----------
int mopoe_comm_create(int32_t rank, int32_t world, int32_t num_floats, int32_t timeout_ms,
                      mopoe_comm** out, void* handle_out) {
    if (!out || !handle_out || world < 1 || world > MOPOE_MAX_RANKS || rank < 0 ||
        rank >= world || num_floats < 4 || num_floats % 4 != 0)
        return fail(MOPOE_ERR_ARG, "mopoe_comm_create: bad argument%s");
    static_assert(sizeof(hipIpcMemHandle_t) + 16 == MOPOE_IPC_HANDLE_BYTES, "handle size");
    mopoe_comm* c = new mopoe_comm();
    memset(c, 0, sizeof(*c));
    c->rank = rank;
    c->world = world;
    c->num_floats = num_floats;
    c->p4 = num_floats / 4;
    c->nchunks = cdiv(c->p4, kXgThreads);
    if (c->nchunks > kXgMaxChunks) {
        delete c;
        return fail(MOPOE_ERR_ARG, "mopoe_comm_create: buffer too large%s");
    }
    c->p4pad = (size_t)c->nchunks * kXgThreads;
    const size_t inbox = 2 * (size_t)world * c->p4pad * 16;
    c->flags_off = inbox;
    // one flag word per source rank and per exchanging workgroup: k_xgmi has nchunks of
    // them, the weight-gradient launch at most one per 64 parameters (a 32-row tile of
    // a one-column weight plus its bias)
    c->flag_stride = c->nchunks > num_floats / 64 + 64 ? c->nchunks : num_floats / 64 + 64;
    c->status_off = inbox + (((size_t)world * c->flag_stride * 4 + 255) / 256) * 256;
    c->bytes = c->status_off + 256;
    if (timeout_ms < 1) timeout_ms = 2000;
    if (timeout_ms > 20000) timeout_ms = 20000;
    c->timeout_ticks = (uint32_t)timeout_ms * 100000u;   // s_memrealtime: 100 MHz
    hipError_t e = hipExtMallocWithFlags(&c->window, c->bytes, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(c->window, 0, c->bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, c->window);
    if (e != hipSuccess) {
        if (c->window) (void)hipFree(c->window);
        delete c;
        return fail(MOPOE_ERR_HIP, "mopoe_comm_create: %s", hipGetErrorString(e));
    }
    memcpy(handle_out, &h, sizeof(h));
    {   // ... followed by the device's UUID: ranks that share a device find out in mopoe_comm_connect
        int dev = 0;
        hipUUID id;
        memset(&id, 0, sizeof(id));
        static_assert(sizeof(id.bytes) == 16, "uuid size");
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetUuid(&id, dev) == hipSuccess) memcpy(c->uuid, id.bytes, 16);
        memcpy(static_cast<char*>(handle_out) + sizeof(h), c->uuid, 16);
    }
    c->mapped[rank] = c->window;
    *out = c;
    return 0;
}

int mopoe_comm_connect(mopoe_comm* c, const void* handles) {
    if (!c || !handles) return fail(MOPOE_ERR_ARG, "mopoe_comm_connect: null argument%s");
    if (c->connected) return fail(MOPOE_ERR_ARG, "mopoe_comm_connect: already connected%s");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        const char* rec = static_cast<const char*>(handles) + (size_t)r * MOPOE_IPC_HANDLE_BYTES;
        memcpy(&h, rec, sizeof(h));
        static const char kNoUuid[16] = {0};
        if (memcmp(c->uuid, kNoUuid, 16) != 0 && memcmp(rec + sizeof(h), c->uuid, 16) == 0) c->shared_device = true;
        hipError_t e = hipIpcOpenMemHandle(&c->mapped[r], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int q = 0; q < r; ++q)
                if (q != c->rank && c->mapped[q]) {
                    (void)hipIpcCloseMemHandle(c->mapped[q]);
                    c->mapped[q] = nullptr;
                }
            return fail(MOPOE_ERR_HIP, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
        }
    }
    c->connected = true;
    if (c->shared_device) g_shared_device_comms.fetch_add(1);
    return 0;
}

int mopoe_comm_allreduce(mopoe_comm* c, float* data, void* stream) {
    if (!c || !data) return fail(MOPOE_ERR_ARG, "mopoe_comm_allreduce: null argument%s");
    return comm_launch(c, data, nullptr, 0, static_cast<hipStream_t>(stream));
}

int mopoe_comm_allreduce_adam(mopoe_comm* c, const mopoe_model* mdl, int32_t present_mask,
                              const mopoe_buffers* buf, const mopoe_adam* adam, void* stream) {
    if (!c || !mdl || !buf || !adam) return fail(MOPOE_ERR_ARG, "null descriptor%s");
    if (!buf->params || !buf->grads || !buf->exp_avg || !buf->exp_avg_sq || !buf->counters)
        return fail(MOPOE_ERR_ARG, "null optimiser buffer%s");
    if (mdl->num_floats != c->num_floats)
        return fail(MOPOE_ERR_ARG, "communicator was created for another buffer length%s");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (present_mask <= 0 || present_mask >= (1 << mdl->num_mods))
        return fail(MOPOE_ERR_ARG, "present_mask out of range%s");   // (before the exchange is launched)
    if (int rc = comm_launch(c, buf->grads, buf->counters, present_mask, s)) return rc;
    return launch_adam(*mdl, present_mask, *buf, *adam, c->world, false, s);
}

int mopoe_comm_train_step(mopoe_comm* c, const mopoe_model* mdl, const mopoe_step* st,
                          const mopoe_buffers* buf, const mopoe_adam* adam, void* stream) {
    if (!c) return fail(MOPOE_ERR_ARG, "mopoe_comm_train_step: null communicator%s");
    return train_step_impl(mdl, st, buf, adam, c, false, stream);
}

int mopoe_comm_status(mopoe_comm* c, int32_t* timeouts) {
    if (!c || !timeouts) return fail(MOPOE_ERR_ARG, "mopoe_comm_status: null argument%s");
    hipError_t e = hipMemcpy(timeouts, static_cast<char*>(c->window) + c->status_off, 4,
                             hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(MOPOE_ERR_HIP, "mopoe_comm_status: %s", hipGetErrorString(e));
    return 0;
}

int mopoe_comm_destroy(mopoe_comm* c) {
    if (!c) return 0;
    if (c->connected && c->shared_device) g_shared_device_comms.fetch_sub(1);
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->mapped[r]) (void)hipIpcCloseMemHandle(c->mapped[r]);
    if (c->window) (void)hipFree(c->window);
    delete c;
    return 0;
}

}  // extern "C"

#include "mopoe_rccl.inc"
#include "mopoe_sampler.inc"
