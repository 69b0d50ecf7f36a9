// mopoe_common.h -- shared host/device helpers of libmopoe_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mopoe_hip.h"

#define HD __host__ __device__ __forceinline__
#define DEV __device__ __forceinline__

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;            // CDNA4 wavefront
constexpr int kRows = MOPOE_ROWS;    // batch rows per tile = MFMA M
constexpr int kHid = MOPOE_HIDDEN;
constexpr int kLdH = kHid + 4;       // LDS leading dim of a hidden tile
constexpr int kStatStride = 96;      // floats reserved for scalar partials
constexpr int kGzChunks = 4;         // K-split of the decoder data-gradient GEMM
constexpr int kEncKChunk = 512;      // K chunk of the encoder input tile in LDS

// positions inside a row-tile's scalar partials
constexpr int kPartKlSub = 0;                         // + subset
constexpr int kPartKlStyle = MOPOE_MAX_SUBSETS;       // + modality
constexpr int kPartNll = kPartKlStyle + MOPOE_MAX_MODS;  // + job
constexpr int kPartMean = kPartNll + MOPOE_MAX_JOBS;  // + 4 * modality + {style mu, style lv, mu, lv}
constexpr int kNumPart = kPartMean + 4 * MOPOE_MAX_MODS;
constexpr int kHandoffWord = kStatStride - 1;  // word of a group's partial slab used as its flag
static_assert(kNumPart < kHandoffWord, "partials overflow their slot");

HD int round_up(int v, int m) { return (v + m - 1) / m * m; }
HD int cdiv(int a, int b) { return (a + b - 1) / b; }

HD int heads_dim(const mopoe_model& m, int i) {
    return 2 * m.style_dim[i] + 2 * m.class_dim;
}
HD int z_dim(const mopoe_model& m, int i) { return m.style_dim[i] + m.class_dim; }
HD int ldz_glb(const mopoe_model& m, int i) { return round_up(z_dim(m, i), 4); }
// LDS leading dims (+4 floats skews rows across banks, keeps 16 B alignment)
HD int ld_heads_lds(const mopoe_model& m, int i) { return round_up(heads_dim(m, i), 16) + 4; }
HD int ld_z_lds(const mopoe_model& m, int i) { return round_up(z_dim(m, i), 16) + 4; }
HD int ld_x_lds(const mopoe_model& m, int i) { return round_up(m.input_dim[i], 16) + 4; }

// offset of modality i's logvar-gradient partials inside a row tile's slot
constexpr int kLvoSlots = 2;  // decoder jobs per modality (joint + unimodal)
HD int lvo_part_off(const mopoe_model& m, int i) {
    int off = kStatStride;
    for (int k = 0; k < i; ++k) off += kLvoSlots * round_up(m.input_dim[k], 4);
    return off;
}
HD int lvo_slot_stride(const mopoe_model& m, int i) { return round_up(m.input_dim[i], 4); }
// behind them: a word per (modality, decoder job, tile of 16 output columns) -- the NLL partial of
// that tile of the row group, left by the output layer's epilogue of a general topology's
// training step and added up, in tile order, by the launch behind it (mopoe_general.inc)
HD int nll_tiles_off(const mopoe_model& m, int i) {
    int off = lvo_part_off(m, m.num_mods);
    for (int k = 0; k < i; ++k) off += kLvoSlots * cdiv(m.input_dim[k], 16);
    return off;
}
HD int nll_tiles_slot_stride(const mopoe_model& m, int i) { return cdiv(m.input_dim[i], 16); }
HD int partials_stride(const mopoe_model& m) { return round_up(nll_tiles_off(m, m.num_mods), 4); }

// Fragment-major weight copies of the four-row form (latent_body FORM 4).  The 4x4x1 MFMA
// wants, per K step, column l of a 64-column tile in lane l; with nn.Linear's (out, in)
// layout that is 64 different rows per wave-load.  WF[tile][k/4][lane][4] = W[64 tile +
// lane][k .. k+3] makes it one contiguous kilobyte for four K steps.  Per modality: the
// heads weights (K = 256) and the decoder weights (K = z_dim, padded to 4).
struct WFrag {
    int whf[MOPOE_MAX_MODS], wdf[MOPOE_MAX_MODS];   // float offsets into mopoe_buffers.wfrag
    int t1[MOPOE_MAX_MODS], t3[MOPOE_MAX_MODS];     // 64-column tiles: heads, decoder
    int k4d[MOPOE_MAX_MODS];                        // K/4 of the decoder (heads: 64)
    int total;
};
// float offset of the piece W[r][4 k4 .. 4 k4 + 3] inside a copy with K/4 = k4n
HD int wfrag_piece(int r, int k4, int k4n) { return (((r >> 6) * k4n + k4) * 64 + (r & 63)) * 4; }
HD WFrag wfrag_layout(const mopoe_model& m) {
    WFrag w;
    int off = 0;
    for (int i = 0; i < MOPOE_MAX_MODS; ++i) {
        const bool on = i < m.num_mods;
        w.t1[i] = on ? cdiv(heads_dim(m, i), 64) : 0;
        w.t3[i] = on ? cdiv(m.input_dim[i], 64) : 0;
        w.k4d[i] = on ? cdiv(z_dim(m, i), 4) : 0;
        w.whf[i] = off;
        off += w.t1[i] * (kHid / 4) * 256;
        w.wdf[i] = off;
        off += w.t3[i] * w.k4d[i] * 256;
    }
    w.total = off;
    return w;
}

// ---------------------------------------------------------------------------
// LDS carve-up of the fused latent kernel (floats), computed on the host and
// handed over in the kernel arguments.  Region R0 is time-shared: the hidden
// tiles (heads GEMM), then the input tiles x (NLL epilogue; turned into g_xhat
// in place when the modality is decoded once) + the K-split partials of g_z;
// the per-element KL terms overlay the partials' area until they are reduced.
// ---------------------------------------------------------------------------
// Packed descriptors of the element-wise stages (fusion forward / backward).  Those
// stages used to chase the public descriptor arrays field by field: ~90 scalar loads,
// most of them waited for on the spot, per wave and stage -- their latency WAS the
// stage.  Here everything a modality / job / subset contributes sits in one aligned
// record that one s_load_dwordx8/x16 fetches.
struct alignas(32) FuseMod {   // LDS offsets in floats
    int heads, gheads;   // tiles [R][ldh]: style mu | style logvar | content mu | content logvar
    int ldh, sd;         // leading dimension, style dim
    int tm, ev;          // [R*D] precision / exp(logvar) of the content posterior
    int klt_style;       // [R*sd] KL terms of the style posterior
    int present;
};
struct alignas(64) FuseJob {
    int mod, slot, src, stream;
    int zj, ldz, epsc, stdc;     // LDS: z tile and its ld, content eps / std [R*D]
    int epss, stds, gzj, gz_off; // LDS: style eps / std [R*sd], kept g_z, column in the slabs
    int sd, ldzg, chunks, pad1;  // style dim of the modality, ld of z[m] in HBM, g_z slabs
};
struct alignas(32) FuseSub {
    uint32_t desc;       // avail | kind << 1 | mask << 3 | E << 8 | members (3 bits each) << 11
                         // | is a mixture component << 26
    int lo, hi;          // rows (inside their logical batch) whose joint latent is this subset
    uint32_t src_jobs;   // decoder jobs fed by this subset's own distribution (method poe)
    int f;               // SLICES: rows per member slice
    float kl_coef;       // d loss / d KL(subset)
    int pad0, pad1;
};
constexpr uint32_t kSubComp = 1u << 26;
struct alignas(64) DecJob {   // decoder stages (x_hat and d loss / d z), per decoder job
    int mod, slot, dm, zd;
    int ldz, ldx, zj, xs;            // LDS: ld of the z / x tiles, z tile, x tile
    int gx, gz_off, off_wd, off_bd;  // LDS: g_xhat tile, column in the g_z slabs; parameters
    int off_lvo, lvo_part, nblk_z, nblk_d;  // lvo_part: this job's slot in a group's partials
    float nll_coef;
    int chunks, per, pad2;           // g_z slabs of this job, K blocks of 16 per slab
    float* loc;                      // buffers.loc[mod], buffers.g_xhat[mod]
    float* g_xhat;
    float* lv;                       // LatentLds::ss (the decoder's logvar HEAD, networks.py:57-59,73-75):
    float* g_lv;                     // gbuffers.lv[mod] / g_lv[mod], the head's parameters, and the LDS
    int off_wlv, off_blv, glv;       // tile [R][ldx] of d loss / d logvar this job's dL/dz stage reads
    int pad3[1];
};
struct alignas(64) EncMod {   // encoder-side stages (heads GEMM, d loss / d h), per modality
    int nh, ldh, off_wh, off_bh;
    int hs, heads, gheads, present;  // LDS: hidden tile, heads tile, its gradient
    int d, ldx, xs, nblk_h;
    const float* hidden;             // buffers.hidden / heads / g_pre / g_heads [mod]
    float* heads_out;
    float* g_pre;
    float* g_heads;
    int kin, a_off, lda;             // heads GEMM: K, the LDS tile it reads as A and its leading
                                     // dimension -- (256, hs, kLdH), or (d_m, xs, ld_x) for an encoder
                                     // without a hidden layer (LatentLds::enc0: the heads sit on x)
    int pad[9];
};

struct LatentLds {
    int hs[MOPOE_MAX_MODS];      // R0: hidden tile            [16][kLdH]
    int xs[MOPOE_MAX_MODS];      // R0: input tile             [16][ld_x]
    int gx[MOPOE_MAX_MODS];      // g_xhat tile (= xs when decoded once)
    int xs_early;                // x tiles have their own area and are loaded at kernel start
    int gzp, ld_gzp, gz_chunks;  // R0: g_z partials           [max chunks][16][ld_gzp]
    int job_chunks[MOPOE_MAX_JOBS], job_per[MOPOE_MAX_JOBS];  // slabs / blocks per slab
    int klt;                     // overlays gzp: KL terms     [subsets][16*D], then styles
    int klt_style[MOPOE_MAX_MODS];
    int heads[MOPOE_MAX_MODS];   // encoder outputs            [16][ld_heads]
    int gheads[MOPOE_MAX_MODS];  // their gradient (second K-half partial during S1)
    int tm[MOPOE_MAX_MODS];      // expert precision T_m       [16*D]
    int ev[MOPOE_MAX_MODS];      // exp(logvar_m)              [16*D]
    int zj[MOPOE_MAX_JOBS];      // decoder input              [16][ld_z]
    int gzj[MOPOE_MAX_JOBS];     // its gradient
    int epsc[MOPOE_MAX_JOBS], stdc[MOPOE_MAX_JOBS];  // content eps / std  [16*D]
    int epss[MOPOE_MAX_JOBS], stds[MOPOE_MAX_JOBS];  // style eps / std    [16*s_m]
    int red;                     // [waves][kStatStride]
    int total;
    // (not LDS) layout of a row tile's slot in `partials`, precomputed because a
    // runtime-bound loop over the model dims inside the kernel gets auto-
    // vectorised and then drags the whole argument block into scratch
    int part_stride;
    int lvo_off[MOPOE_MAX_MODS];
    int gz_off[MOPOE_MAX_JOBS];  // column of job j inside its pass's g_z partial slab
    int single_pass;             // all decoder jobs belong to one pass
    // Unit tables (prefix sums): a wave finds the modality / job of its unit with a
    // few scalar compares on one array instead of a scan of dependent scalar loads.
    // Entries past the last modality / job repeat the total.
    int s1_begin[MOPOE_MAX_MODS + 1];  // heads tiles (16 columns) of modality m
    int sl_begin[MOPOE_MAX_MODS + 1];  // element-wise slots: [0] content slots, then
                                       // the end of each modality's style slots
    int s3_begin[MOPOE_MAX_JOBS + 1];  // decoder units (64 columns), all jobs
    int s4_begin[MOPOE_MAX_JOBS + 1];  // g_z units (64 columns x the job's slabs), all jobs
    int pass_end[MOPOE_MAX_JOBS];      // one past the last job of the pass starting here
    int kl_first, kl_pool;             // waves [kl_first, kl_first + kl_pool) take the KL sums
    int pres_mod[MOPOE_MAX_MODS];      // k-th present modality
    int npres;
    int s3_nt;                         // 16-column tiles per decoder unit: 4, or 2 when the
                                       // 64-column units would leave half the waves idle
    FuseMod fm[MOPOE_MAX_MODS];
    FuseJob fj[MOPOE_MAX_JOBS];
    FuseSub fs[MOPOE_MAX_SUBSETS];
    uint32_t joint_jobs;               // decoder jobs fed by the joint latent
    DecJob dj[MOPOE_MAX_JOBS];
    EncMod em[MOPOE_MAX_MODS];
    // LDS tiles whose K padding the GEMM stages read (z, g_heads): zeroed whole, first
    // thing, one 16-byte store per thread (prefix table in float4 units)
    int zr_begin[MOPOE_MAX_JOBS + MOPOE_MAX_MODS + 1];
    int zr_off[MOPOE_MAX_JOBS + MOPOE_MAX_MODS];
    // four-row groups (latent_body FORM 4): K parts of the GEMM stages, one per wave
    int q1_begin[MOPOE_MAX_MODS + 1], q1_kper, q1_parts;   // heads: per present modality (tile, K part) units
    int q3_begin[MOPOE_MAX_JOBS + 1];            // decoder: 64-column tiles per job
    int q4_begin[MOPOE_MAX_JOBS + 1], q4_kper;   // dL/dz: K parts (of d_m) per job
    int q6_begin[MOPOE_MAX_MODS + 1], q6_kper;   // dL/dh: K parts (of nh_m) per present modality
    int qred;                                    // LDS: partial tiles [16 waves][4 rows][256]
    struct alignas(16) Q4Unit {                  // dL/dz: all a wave needs, in one scalar load
        int nk4;                                 // K steps of four that hold rows < round_up(d_m, 16); -1: no unit
        int ga;                                  // LDS: g_xhat tile + this part's first column
        int wl;                                  // LDS: Wd copy + this part's first row
        int ldx_zd;                              // ld of the tile << 8 | row length of the copy
    } q4u[16];
    int wdl[MOPOE_MAX_MODS];                     // LDS: copy of Wd_m (behind the 64-column partials)
    int wd_units[2];                             // 1 KB units (64 pieces of 16 bytes) of it, per present modality
    int wd_valid[2];                             // pieces that hold weights (the rest: zero rows)
    int quad_ok;
    int ss;                            // learn_output_sample_scale: the likelihood's log-variance is the output
                                       // of a second decoder head, per sample and feature (generic body only)
    int glv[MOPOE_MAX_JOBS];           // its gradient tiles [R][ld_x], per decoder job
    int enc0;                          // the encoder has NO hidden layer (networks.py:16-20 with
                                       // num_hidden_layers = 0: the four heads are Linear(d_m, .) on x):
                                       // no h tiles, no producers, no dL/dh stage; the heads GEMM
                                       // reads the x tile with K = d_m (generic body only)
    int enc0_publish;                  // enc0 training step: row group 0 begins the step (there is no
    mopoe_adam enc0_adam;              // encoder-layer launch to do it) and publishes these Adam records
    int fold_tiles;                    // > 0: `partials` holds this many pre-summed slabs (large batches: k_partials_fold)
    WFrag wf;                                    // fragment-major weight copies (mopoe_buffers.wfrag)
    int rows;                          // batch rows a group owns (16, 8, 4, 2 or 1)
    int rd;                            // round_up(rows * class_dim, 4): stride of a KL-term slab
    int fits;                          // the carve-up fits the 160 KiB budget
};


// Carve-up for groups of R rows (every tile R rows tall); returns false when even the
// most frugal option does not fit the 160 KiB budget.
HD bool latent_lds_layout_rows(const mopoe_model& m, const mopoe_step& st, int waves, int R,
                               LatentLds& L) {
    const int D = m.class_dim;
    const int RD = round_up(R * D, 4);  // element-wise arrays keep the tiles 16-byte aligned
    L.rows = R;
    L.rd = RD;
    // (pad_: set by the library on its own copy of the step -- mopoe_general_*: bit 0 enc0, bit 1 ss)
    L.enc0 = (st.pad_ & 1) != 0;
    L.ss = (st.pad_ & 2) != 0;
    int hsz = 0, xsz = 0, zcols = 0, klt = st.num_subsets * RD;
    int jobs_of[MOPOE_MAX_MODS] = {0, 0, 0, 0, 0};
    for (int j = 0; j < st.num_jobs; ++j) jobs_of[st.job_mod[j]]++;
    for (int i = 0; i < m.num_mods; ++i) {
        L.hs[i] = hsz;
        L.xs[i] = xsz;
        L.klt_style[i] = klt;
        if ((st.present_mask >> i) & 1) {
            hsz += L.enc0 ? 0 : R * kLdH;
            xsz += R * ld_x_lds(m, i);
            zcols += round_up(z_dim(m, i), 16);
            klt += round_up(R * m.style_dim[i], 4);
        }
    }
    L.ld_gzp = zcols + 4;
    // Decoder passes = jobs with the same row block (job_slot): a modality has one job per
    // slot, so the jobs of a slot can share the decoder stages -- method poe's unimodal jobs
    // (one noise stream EACH, run_epochs.py:104-128) are one pass, not one per modality.
    // (job_stream only says which jobs share their content noise.)
    L.single_pass = st.num_jobs > 0 && st.job_slot[st.num_jobs - 1] == st.job_slot[0];
    for (int j = 0, zo = 0; j < st.num_jobs; ++j) {
        if (j > 0 && st.job_slot[j] != st.job_slot[j - 1]) zo = 0;
        L.gz_off[j] = zo;
        zo += round_up(z_dim(m, st.job_mod[j]), 16);
    }
    L.part_stride = partials_stride(m);
    for (int i = 0; i < MOPOE_MAX_MODS; ++i) L.lvo_off[i] = i < m.num_mods ? lvo_part_off(m, i) : 0;
    // Preferred: x tiles in their own area (loaded at kernel start, no extra
    // round trip later) and the largest K-split; fall back until the kernel fits
    // the 160 KiB LDS budget.
    const int klt_rel = klt;
    bool fits = false;
    int xrel[MOPOE_MAX_MODS];
    for (int i = 0; i < MOPOE_MAX_MODS; ++i) xrel[i] = i < m.num_mods ? L.xs[i] : 0;
    for (int option = 0;; ++option) {
        // the dL/dz GEMM splits its reduction axis (d_m, in blocks of 16) into slabs
        // of >= 3 blocks (one round of loads per wave), at most `cap` slabs per job
        const int early_of[8] = {1, 1, 1, 0, 0, 0, 0, 0};
        const int cap_of[8] = {12, 8, 4, 12, 8, 4, 2, 1};
        L.xs_early = early_of[option];
        L.gz_chunks = 1;
        for (int j = 0; j < MOPOE_MAX_JOBS; ++j) {
            const int nb = j < st.num_jobs ? round_up(m.input_dim[st.job_mod[j]], 16) / 16 : 1;
            int c = cdiv(nb, 3);
            if (c > cap_of[option]) c = cap_of[option];
            L.job_chunks[j] = c;
            L.job_per[j] = cdiv(nb, c);
            if (c > L.gz_chunks) L.gz_chunks = c;
        }
        int gz = L.gz_chunks * R * L.ld_gzp;
        if (gz < klt_rel) gz = klt_rel;
        int off;
        if (L.xs_early) {  // [hs | ...] and [gzp/klt] share R0, xs follows
            L.gzp = 0;
            off = gz > hsz ? gz : hsz;
            for (int i = 0; i < m.num_mods; ++i) L.xs[i] = off + xrel[i];
            off += xsz;
        } else {           // xs overlays hs, gzp/klt follow xs
            for (int i = 0; i < m.num_mods; ++i) L.xs[i] = xrel[i];
            L.gzp = xsz;
            off = xsz + gz > hsz ? xsz + gz : hsz;
        }
        L.klt = L.gzp;
        {
            int ks = st.num_subsets * RD;
            for (int i = 0; i < m.num_mods; ++i) {
                L.klt_style[i] = L.klt + ks;
                if ((st.present_mask >> i) & 1) ks += round_up(R * m.style_dim[i], 4);
            }
        }
        for (int i = 0; i < m.num_mods; ++i) {
            L.gx[i] = L.xs[i];
            L.heads[i] = L.gheads[i] = L.tm[i] = L.ev[i] = off;
            if (!((st.present_mask >> i) & 1)) continue;
            if (jobs_of[i] > 1) {
                L.gx[i] = off;
                off += R * ld_x_lds(m, i);
            }
            L.heads[i] = off;
            off += R * ld_heads_lds(m, i);
            L.gheads[i] = off;
            off += R * ld_heads_lds(m, i);
            L.tm[i] = off;
            off += RD;
            L.ev[i] = off;
            off += RD;
        }
        for (int j = 0; j < MOPOE_MAX_JOBS; ++j) L.glv[j] = 0;
        for (int j = 0; j < st.num_jobs; ++j) {
            const int i = st.job_mod[j];
            L.zj[j] = off;
            off += R * ld_z_lds(m, i);
            L.gzj[j] = off;
            off += R * ld_z_lds(m, i);
            L.epsc[j] = off;
            off += RD;
            L.stdc[j] = off;
            off += RD;
            L.epss[j] = off;
            off += round_up(R * m.style_dim[i], 4);
            L.stds[j] = off;
            off += round_up(R * m.style_dim[i], 4);
            L.glv[j] = off;
            if (L.ss) off += R * ld_x_lds(m, i);
        }
        off = round_up(off, 4);
        L.red = off;
        off += waves * kStatStride;
        L.total = off;
        fits = off * 4 <= 160 * 1024;
        // (enc0: the heads GEMM reads the x tiles -- they must be in their own area from the start)
        if (fits || option == (L.enc0 ? 2 : 7)) break;
    }
    {
        int t = 0, sl = cdiv(R * D, kWave);
        L.npres = 0;
        for (int i = 0; i < MOPOE_MAX_MODS; ++i) {
            L.s1_begin[i] = t;
            L.sl_begin[i] = sl;
            L.pres_mod[i] = 0;
            if (i < m.num_mods && ((st.present_mask >> i) & 1)) {
                t += cdiv(heads_dim(m, i), 16);
                sl += cdiv(R * m.style_dim[i], kWave);
            }
        }
        for (int i = 0; i < m.num_mods; ++i)
            if ((st.present_mask >> i) & 1) L.pres_mod[L.npres++] = i;
        L.s1_begin[MOPOE_MAX_MODS] = t;
        L.sl_begin[MOPOE_MAX_MODS] = sl;
        int widest = 0;  // 64-column decoder units of the busiest pass
        for (int j = 0, u = 0; j < st.num_jobs; ++j) {
            if (j > 0 && st.job_slot[j] != st.job_slot[j - 1]) u = 0;
            u += cdiv(m.input_dim[st.job_mod[j]], 64);
            if (u > widest) widest = u;
        }
        L.s3_nt = 2 * widest <= waves ? 2 : 4;
        int u3 = 0, u4 = 0;
        for (int j = 0; j <= MOPOE_MAX_JOBS; ++j) {
            L.s3_begin[j] = u3;
            L.s4_begin[j] = u4;
            if (j < st.num_jobs) {
                u3 += cdiv(m.input_dim[st.job_mod[j]], 16 * L.s3_nt);
                u4 += cdiv(z_dim(m, st.job_mod[j]), 64) * L.job_chunks[j];
            }
        }
        for (int j = 0; j < MOPOE_MAX_JOBS; ++j) {
            int je = j + 1;
            while (je < st.num_jobs && st.job_slot[je] == st.job_slot[j]) ++je;
            L.pass_end[j] = j < st.num_jobs ? je : j + 1;
        }
        {   // the KL sums ride on the waves the first decoder pass leaves without a unit
            const int busy = L.s3_begin[L.pass_end[0]] < waves ? L.s3_begin[L.pass_end[0]] : waves;
            L.kl_first = busy < waves ? busy : 0;
            L.kl_pool = waves - L.kl_first;
        }
    }
    for (int i = 0; i < MOPOE_MAX_MODS; ++i) {
        FuseMod& f = L.fm[i];
        f.present = i < m.num_mods && ((st.present_mask >> i) & 1);
        f.heads = L.heads[i];
        f.gheads = L.gheads[i];
        f.ldh = i < m.num_mods ? ld_heads_lds(m, i) : 0;
        f.sd = i < m.num_mods ? m.style_dim[i] : 0;
        f.tm = L.tm[i];
        f.ev = L.ev[i];
        f.klt_style = L.klt_style[i];
    }
    L.joint_jobs = 0;
    for (int j = 0; j < MOPOE_MAX_JOBS; ++j) {
        FuseJob& f = L.fj[j];
        const bool on = j < st.num_jobs;
        const int i = on ? st.job_mod[j] : 0;
        f.mod = i;
        f.slot = on ? st.job_slot[j] : 0;
        f.src = on ? st.job_src[j] : 0;
        f.stream = on ? st.job_stream[j] : 0;
        f.zj = L.zj[j];
        f.ldz = ld_z_lds(m, i);
        f.epsc = L.epsc[j];
        f.stdc = L.stdc[j];
        f.epss = L.epss[j];
        f.stds = L.stds[j];
        f.gzj = L.gzj[j];
        f.gz_off = L.gz_off[j];
        f.sd = m.style_dim[i];
        f.ldzg = ldz_glb(m, i);
        f.chunks = L.job_chunks[j];
        f.pad1 = 0;
        if (on && st.job_src[j] < 0) L.joint_jobs |= 1u << j;
    }
    for (int k = 0; k < MOPOE_MAX_SUBSETS; ++k) {
        FuseSub& f = L.fs[k];
        const bool on = k < st.num_subsets;
        uint32_t desc = 0, nmem = 0;
        if (on) {
            for (int b = 0; b < MOPOE_MAX_MODS; ++b) nmem += (st.sub_mask[k] >> b) & 1;
            desc = (uint32_t)(st.sub_avail[k] != 0) | ((uint32_t)st.sub_kind[k] << 1) |
                   ((uint32_t)st.sub_mask[k] << 3) | (nmem << 8);
            for (int b = 0; b < MOPOE_MAX_MODS; ++b)
                desc |= ((uint32_t)st.sub_members[k][b] & 7u) << (11 + 3 * b);
        }
        f.desc = desc;
        f.lo = f.hi = 0;
        f.src_jobs = 0;
        f.f = on ? st.sub_f[k] : 0;
        f.kl_coef = on ? st.sub_kl_coef[k] : 0.f;
        f.pad0 = f.pad1 = 0;
        for (int j = 0; j < st.num_jobs; ++j)
            if (on && st.job_src[j] == k) f.src_jobs |= 1u << j;
    }
    for (int j = 0; j < MOPOE_MAX_JOBS; ++j) {
        DecJob& f = L.dj[j];
        const bool on = j < st.num_jobs;
        const int i = on ? st.job_mod[j] : 0;
        f.mod = i;
        f.slot = on ? st.job_slot[j] : 0;
        f.dm = m.input_dim[i];
        f.zd = z_dim(m, i);
        f.ldz = ld_z_lds(m, i);
        f.ldx = ld_x_lds(m, i);
        f.zj = L.zj[j];
        f.xs = L.xs[i];
        f.gx = L.gx[i];
        f.gz_off = L.gz_off[j];
        f.off_wd = m.off_wd[i];
        f.off_bd = m.off_bd[i];
        f.off_lvo = m.off_lvo[i];
        f.lvo_part = L.lvo_off[i] + f.slot * lvo_slot_stride(m, i);
        f.nblk_z = round_up(f.zd, 16) / 16;
        f.nblk_d = round_up(f.dm, 16) / 16;
        f.nll_coef = on ? st.job_nll_coef[j] : 0.f;
        f.chunks = L.job_chunks[j];
        f.per = L.job_per[j];
        f.pad2 = 0;
        f.loc = f.g_xhat = nullptr;   // latent_bind()
        f.lv = f.g_lv = nullptr;      // (ss: bound by the caller, with the head's offsets)
        f.off_wlv = f.off_blv = 0;
        f.glv = j < MOPOE_MAX_JOBS ? L.glv[j] : 0;
        f.pad3[0] = 0;
    }
    for (int i = 0; i < MOPOE_MAX_MODS; ++i) {
        EncMod& f = L.em[i];
        const bool on = i < m.num_mods;
        f.nh = on ? heads_dim(m, i) : 0;
        f.ldh = on ? ld_heads_lds(m, i) : 0;
        f.off_wh = on ? m.off_wh[i] : 0;
        f.off_bh = on ? m.off_bh[i] : 0;
        f.hs = L.hs[i];
        f.heads = L.heads[i];
        f.gheads = L.gheads[i];
        f.present = on && ((st.present_mask >> i) & 1);
        f.d = on ? m.input_dim[i] : 0;
        f.ldx = on ? ld_x_lds(m, i) : 0;
        f.xs = L.xs[i];
        f.nblk_h = round_up(f.nh, 16) / 16;
        f.hidden = nullptr;
        f.heads_out = f.g_pre = f.g_heads = nullptr;
        f.kin = L.enc0 ? f.d : kHid;
        f.a_off = L.enc0 ? f.xs : f.hs;
        f.lda = L.enc0 ? f.ldx : kLdH;
        for (int k = 0; k < 9; ++k) f.pad[k] = 0;
    }
    {
        int k = 0, q = 0;
        for (int j = 0; j < MOPOE_MAX_JOBS; ++j) {
            L.zr_begin[k] = q;
            L.zr_off[k++] = L.zj[j < st.num_jobs ? j : 0];
            if (j < st.num_jobs) q += R * ld_z_lds(m, st.job_mod[j]) / 4;
        }
        for (int i = 0; i < MOPOE_MAX_MODS; ++i) {
            L.zr_begin[k] = q;
            L.zr_off[k++] = L.gheads[i];
            if (i < m.num_mods && ((st.present_mask >> i) & 1)) q += R * ld_heads_lds(m, i) / 4;
        }
        L.zr_begin[k] = q;
    }
    if (st.joint_mode == MOPOE_JOINT_EXPERT) {
        L.fs[st.expert_subset].hi = 0x7fffffff;
    } else {
        for (int k = 0; k < st.num_comp; ++k) {
            FuseSub& f = L.fs[st.comp_sub[k]];
            f.desc |= kSubComp;
            if (st.joint_mode == MOPOE_JOINT_MIXTURE) {  // utils/utils.py:63-85
                f.lo = k * st.comp_f;
                f.hi = k == st.num_comp - 1 ? 0x7fffffff : (k + 1) * st.comp_f;
            }
        }
    }
    return fits;
}


// Rows per k_latent group: 16 (one full MFMA tile) whenever the carve-up fits the LDS,
// else the largest of 8, 4, 2, 1 that does (wide inputs, five modalities: the MFMA
// tiles then run partly empty -- a group of R < 16 rows issues as many MFMAs as one of
// 16, so the GEMM stages take the same time for fewer rows).
// mopoe_step.rows_per_group pins it (tests exercise the small-group path that way).
HD void latent_lds_layout(const mopoe_model& m, const mopoe_step& st, int waves,
                          LatentLds& L) {
    const int r0 = st.rows_per_group;
    if (r0 == 1 || r0 == 2 || r0 == 4 || r0 == 8 || r0 == 16) {
        L.fits = latent_lds_layout_rows(m, st, waves, r0, L);
        return;
    }
    for (int R = kRows; R >= 1; R /= 2) {
        L.fits = latent_lds_layout_rows(m, st, waves, R, L);
        if (L.fits) return;
    }
}
// the caller's buffer pointers into the per-job / per-modality records
HD void latent_bind(LatentLds& L, const mopoe_buffers& b) {
    for (int j = 0; j < MOPOE_MAX_JOBS; ++j) {
        L.dj[j].loc = b.loc[L.dj[j].mod];
        L.dj[j].g_xhat = b.g_xhat[L.dj[j].mod];
    }
    for (int i = 0; i < MOPOE_MAX_MODS; ++i) {
        L.em[i].hidden = b.hidden[i];
        L.em[i].heads_out = b.heads[i];
        L.em[i].g_pre = b.g_pre[i];
        L.em[i].g_heads = b.g_heads[i];
    }
}
HD int latent_groups(const mopoe_model& m, const mopoe_step& st, int waves) {
    LatentLds L;
    latent_lds_layout(m, st, waves, L);
    return cdiv(st.n, L.rows);
}

// ---------------------------------------------------------------------------
// device-only helpers
// ---------------------------------------------------------------------------
#if defined(__HIPCC__)

// v_mfma_f32_4x4x1_16B_f32: sixteen independent 4x4 blocks, K = 1, the same FLOP rate as
// the 16x16x4 form (8 cycles for 512 flops).  D[vgpr i][lane l] += A[lane 4*(l/4)+i] *
// B[lane l] (tools/mfma4_probe.hip): with A = X[row l%4][k] in every block and B =
// W[k][column l], lane l ends up with column l of FOUR batch rows.
DEV f32x4 mfma_4x4x1(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

DEV f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    // v_mfma_f32_16x16x4_f32: exact f32 fmaf chain (guide section 3).
    // A: lane l holds A[row l&15][k l>>4]; B: B[k l>>4][col l&15];
    // C/D: col = l&15, row = 4*(l>>4) + reg.
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// A packed descriptor record out of the argument block, fetched whole: 8-dword vector
// loads, which become s_load_dwordx8 (left to itself the compiler loads the fields one
// dword at a time, next to their uses, and waits for each).
// (a 16-byte record: one s_load_dwordx4)
template <class T>
DEV T load_rec16(const T& src) {
    static_assert(sizeof(T) == 16 && alignof(T) >= 16, "a 16-byte record");
    typedef int v4i __attribute__((ext_vector_type(4)));
    union U {
        T t;
        v4i v;
        DEV U() {}
    } u;
    u.v = *reinterpret_cast<const v4i*>(&src);
    return u.t;
}
template <class T>
DEV T load_rec(const T& src) {
    static_assert(sizeof(T) % 32 == 0 && alignof(T) >= 32, "records are 32-byte multiples");
    typedef int v8i __attribute__((ext_vector_type(8)));
    union U {
        T t;
        v8i v[sizeof(T) / 32];
        DEV U() {}
    } u;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 32; ++i) u.v[i] = reinterpret_cast<const v8i*>(&src)[i];
    return u.t;
}

// index of the segment of a prefix table (begin[0..NSEG]) that holds u
template <int NSEG>
DEV int find_seg(const int (&begin)[NSEG + 1], int u) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < NSEG; ++i) k += u >= begin[i];
    return k;
}
// The same with u >= begin[i] taken as the sign bit of begin[i] - u - 1: plain integer
// arithmetic stays on the scalar unit (the bool-to-int of a compare goes through v_cndmask +
// readfirstlane).  Used by the four-row form; in the 16-row forms it moved the register
// allocation to a slower place (+0.75 us, same box), so they keep the form above.
template <int NSEG>
DEV int find_seg_s(const int (&begin)[NSEG + 1], int u) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < NSEG; ++i) k += (int)((uint32_t)(begin[i] - u - 1) >> 31);
    return k;
}

// Sum over the 64 lanes, the same value in every lane.  Inside a 16-lane row with DPP
// adds (quad swaps, then the two row mirrors: an instruction each, no LDS crossbar --
// six dependent ds_bpermute round trips cost ~500 cycles of a wave's chain), then the
// four row sums through readlane in a fixed order.
// float add onto an LDS word (ds_add_f32, nothing returned)
DEV void lds_add(float* p, float v) {
    typedef __attribute__((address_space(3))) float lds_float;
    __hip_atomic_fetch_add((lds_float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// torch's sign(): 0 at 0 (the gradient torch.abs hands back)
DEV float sign_of(float v) { return v > 0.f ? 1.f : v < 0.f ? -1.f : 0.f; }

// A launch's argument block was written by the host a moment ago: the first touch of each of its
// 64-byte lines misses every cache, and a kernel that walks its tables (subset records, job
// records, buffer pointers) pays those misses one behind the other, a trip to memory each.
// Here the block's waves touch every line once, side by side, before the first table is
// needed: one trip, after which the scalar cache has the block.  (BYTES, NW compile-time.)
template <int BYTES, int NW>
DEV void warm_args(const void* base, int wave) {
    typedef __attribute__((address_space(4))) const int* cptr;
    cptr p = (cptr)(uintptr_t)base;
    int acc = 0;
#pragma unroll
    for (int i = 0; i < (BYTES / 64 + NW - 1) / NW; ++i) {
        const int line = wave + i * NW;
        acc += p[(line * 64 < BYTES ? line : 0) * 16];
    }
    __asm__ volatile("" ::"s"(acc));
}

DEV float wave_sum(float v) {
    auto dpp_add = [](float x, auto ctrl) __attribute__((always_inline)) {
        constexpr int kCtrl = decltype(ctrl)::value;
        const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), kCtrl, 0xF, 0xF, false);
        return x + __builtin_bit_cast(float, y);
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v = dpp_add(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v = dpp_add(v, std::integral_constant<int, 0x140>{});  // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

// A fragment: LDS tile [16][lda] (k contiguous).  Lane (r = l&15, q = l>>4)
// reads k = kb+4q .. kb+4q+3; the four values feed the four MFMA k-steps of a
// 16-deep block.  The k order inside the block is permuted (step i covers
// k = kb + 4q + i), identically for A and B, which leaves the sum unchanged.
DEV f32x4 lds_a4(const float* As, int lda, int kb, int lane) {
    return *reinterpret_cast<const f32x4*>(As + (lane & 15) * lda + kb + 4 * (lane >> 4));
}

// ---------------------------------------------------------------------------
// Global-memory reads of the hot loops go through buffer descriptors
// (`buffer_load ... offen`): the hardware range check returns 0 for an offset
// at or past num_records, so an out-of-range lane needs neither a branch nor a
// select on the loaded value -- it is simply given the offset kOOB.  This keeps
// every load of a batch in one basic block and in flight together.  (With plain
// pointer loads hipcc turns `cond ? *p : 0` into a branch around the load and
// drains the memory counter at every join: the batch serialises.)
// Descriptors are built from wave-uniform values only (kernel arguments and
// blockIdx-derived scalars), num_records is capped at 2 GiB - 1 so that kOOB
// is always out of range.
// ---------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOOB = 0x80000000u;

DEV rsrc_t make_rsrc(const void* p, size_t bytes) {
    // readfirstlane makes the uniformity provable to hipcc (guide T20); the
    // inputs ARE wave-uniform at every call site
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    uint32_t n = bytes > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)bytes;
    n = __builtin_amdgcn_readfirstlane(n);
    void* q = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, n, 0x00020000);
}
DEV rsrc_t make_rsrc_max(const void* p) { return make_rsrc(p, 0x7FFFFFFFull); }
// offset (< 2 GiB) of a lane, pushed out of range when the lane is invalid.
// Written as arithmetic on purpose: a `ok ? off : kOOB` select invites hipcc to
// split the two cases into branches again.
DEV uint32_t guard(uint32_t byte_off, bool ok) { return byte_off | (ok ? 0u : kOOB); }
DEV float ldg(rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
DEV f32x4 ldg4(rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
// 16-byte load past the L1 (sc1): for data another workgroup of the SAME launch wrote
// (write-through) -- this CU's L1 is never refreshed by other CUs' stores
DEV f32x4 ldg4_sc1(rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
// 16-byte WRITE-THROUGH store (sc1): the line does not stay dirty in the XCD's L2, so
// it is on its way to memory while the kernel still runs instead of being flushed when
// the kernel ends (the consumer is the next kernel, on any XCD, through memory anyway).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEV void stg4_wt(rsrc_t r, uint32_t byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}

// System-scope accesses (sc0 sc1): write-through / read-through past both cache levels,
// for memory another GPU writes or reads while the kernel runs (the xGMI peer windows).
DEV f32x4 ld16_sys(rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 17));
}
DEV float ld4_sys(rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 17));
}
DEV void st16_sys(rsrc_t r, uint32_t off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 17);
}
DEV void st4_sys(rsrc_t r, uint32_t off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, off, 0, 17);
}

// The exchange descriptor a kernel needs to reach the peers' windows (mopoe_xgmi.inc).
struct XgPeers {
    int32_t rank, world;
    uint32_t seq, parity, timeout_ticks;
    int32_t flag_stride;                 // flag words per source rank
    float inv_world;
    int32_t mask;                        // modalities in this rank's batch (0 when the
                                         // exchange is a plain all-reduce)
    int32_t fail_slot;                   // (test knob MOPOE_TEST_XG_FAIL_SLOT, else -1) the
                                         // exchanging block whose wait is reported as failed
    int32_t pad0;
    size_t p4pad;                        // float4 per (parity, source) inbox
    size_t flags_off, status_off;        // byte offsets inside a window
    void* window[MOPOE_MAX_RANKS];       // base of every rank's window as mapped HERE
};

// Publishes (seq, present_mask) in flags[me][slot] of every peer, then waits (bounded)
// until every peer's flags[peer][slot] HERE has reached seq.  Called by the threads
// tid < world of a workgroup (or wave) whose pushes have drained (s_waitcnt vmcnt(0) +
// barrier).  A flag word is seq << 8 | mask: the ranks' batches must hold the same
// modalities (the sum of gradients of different parameter sets is not a step of the
// reference), and a rank learns its peers' masks from the very word it waits for.
// Returns true when the exchange is NOT good: a wait ran out of its budget (also counted
// in the window's status word) or a peer's batch held other modalities.
DEV bool xg_signal_and_wait(const XgPeers& x, int tid, int slot) {
    if (slot == x.fail_slot) return tid == 0;   // (test knob: ONE block of the launch fails)
    if (tid >= x.world || tid == x.rank) return false;
    uint32_t* out = reinterpret_cast<uint32_t*>(static_cast<char*>(x.window[tid]) + x.flags_off) +
                    (size_t)x.rank * x.flag_stride + slot;
    const uint32_t want = x.seq << 8;
    __hip_atomic_store(out, want | ((uint32_t)x.mask & 0xFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* in = reinterpret_cast<const uint32_t*>(
                             static_cast<const char*>(x.window[x.rank]) + x.flags_off) +
                         (size_t)tid * x.flag_stride + slot;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const uint32_t got = __hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)((got & ~0xFFu) - want) >= 0)   // (a peer can be one exchange ahead)
            return (got & ~0xFFu) == want && (got & 0xFFu) != ((uint32_t)x.mask & 0xFFu);
        if (__builtin_amdgcn_s_memrealtime() - t0 > x.timeout_ticks) {
            int32_t* st = reinterpret_cast<int32_t*>(static_cast<char*>(x.window[x.rank]) +
                                                     x.status_off);
            __hip_atomic_fetch_add(st, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// byte offset of inbox[parity][src] inside a window
DEV size_t xg_inbox(const XgPeers& x, int src) {
    return ((size_t)x.parity * x.world + src) * x.p4pad * 16;
}

// Plain stores through a descriptor: with `guard`ed offsets an invalid lane's store is
// dropped by the hardware range check -- no branch around it, no 64-bit address
// arithmetic (hipcc turns `if (ok) *p = v` into exec-mask juggling around a flat store).
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEV void stg1(rsrc_t r, uint32_t off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, off, 0, 0);
}
DEV void stg2(rsrc_t r, uint32_t off, f32x2 v) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, off, 0, 0);
}
DEV void stg4(rsrc_t r, uint32_t off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}
// NT (2 or 4) consecutive floats at float offset `o` of the buffer behind r, of which
// `nvalid` (counted from the left, may be <= 0 or > NT) exist in the row; `ok`: the row
// exists.  One vector store when all NT exist, single words otherwise; every store is
// issued, the ones that do not apply go out of range.
template <int NT, class V>
DEV void stg_row(rsrc_t r, uint32_t o, V v, int nvalid, bool ok) {
    if constexpr (NT == 4)
        stg4(r, guard(o * 4u, ok & (nvalid >= 4)), v);
    else
        stg2(r, guard(o * 4u, ok & (nvalid >= 2)), v);
#pragma unroll
    for (int t = 0; t < NT - 1; ++t)
        stg1(r, guard((o + t) * 4u, ok & (nvalid < NT) & (t < nvalid)), v[t]);
}

// B fragment of Y = A * W^T: W is (ncols, K) row-major with row stride ldw;
// r covers ncols * ldw floats, so col >= ncols is out of range by itself.
// VEC: K % 4 == 0 (then a 4-wide read never crosses a row end).
template <bool VEC>
DEV f32x4 glb_b4_nt(rsrc_t r, int ldw, int K, int j0, int kb, int lane) {
    const int col = j0 + (lane & 15);
    const int k = kb + 4 * (lane >> 4);
    const uint32_t base = (uint32_t)(col * ldw + k) * 4u;
    f32x4 v;
    if (VEC) {
        v = ldg4(r, guard(base, k < K));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ldg(r, guard(base + 4u * i, k + i < K));
    }
    return v;
}

// Philox4x32-10 -> four uniforms in (0, 1): the dropout keep decisions (keep iff u >= p)
DEV f32x4 philox_uniform4(uint64_t seed, uint32_t step, uint32_t stream, uint32_t quad) {
    uint32_t c0 = quad, c1 = stream, c2 = step, c3 = 0x64726f70u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    f32x4 u;
    u[0] = ((c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    u[1] = ((c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    u[2] = ((c2 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    u[3] = ((c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return u;
}

// Philox4x32-10 -> FOUR standard normals per call (two Box-Muller pairs; the angle goes
// through v_sin_f32 / v_cos_f32, which take revolutions: no range reduction).  Counter =
// (quad, stream, step, tag), key = seed; element idx of a stream is component idx & 3 of
// quad idx >> 2, so a lane that owns four consecutive elements pays for one call.
DEV f32x4 philox_normal4(uint64_t seed, uint32_t step, uint32_t stream, uint32_t quad) {
    uint32_t c0 = quad, c1 = stream, c2 = step, c3 = 0x4d6f506fu;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float u0 = ((c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    const float u1 = ((c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((c2 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u3 = ((c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    // sqrt(-2 ln u) = sqrt(-2 ln2 * log2 u)
    const float ra = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u0));
    const float rb = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u2));
    f32x4 v;
    v[0] = ra * __builtin_amdgcn_cosf(u1);
    v[1] = ra * __builtin_amdgcn_sinf(u1);
    v[2] = rb * __builtin_amdgcn_cosf(u3);
    v[3] = rb * __builtin_amdgcn_sinf(u3);
    return v;
}
DEV float philox_normal(uint64_t seed, uint32_t step, uint32_t stream, uint32_t idx) {
    const f32x4 v = philox_normal4(seed, step, stream, idx >> 2);
    const uint32_t k = idx & 3u;
    return k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3];
}

#endif  // __HIPCC__
