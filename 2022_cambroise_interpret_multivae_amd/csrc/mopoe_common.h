// mopoe_common.h -- shared host/device helpers of libmopoe_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mopoe_hip.h"

#define HD __host__ __device__ __forceinline__
#define DEV __device__ __forceinline__

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;            // CDNA4 wavefront
constexpr int kRows = MOPOE_ROWS;    // batch rows per tile = MFMA M
constexpr int kHid = MOPOE_HIDDEN;
constexpr int kLdH = kHid + 4;       // LDS leading dim of a hidden tile
constexpr int kStatStride = 64;      // floats reserved for scalar partials
constexpr int kGzChunks = 4;         // K-split of the decoder data-gradient GEMM
constexpr int kEncKChunk = 1024;     // K chunk of the encoder input tile in LDS

// positions inside a row-tile's scalar partials
constexpr int kPartKlSub = 0;                         // + subset
constexpr int kPartKlStyle = MOPOE_MAX_SUBSETS;       // + modality
constexpr int kPartNll = kPartKlStyle + MOPOE_MAX_MODS;  // + job
constexpr int kNumPart = kPartNll + MOPOE_MAX_JOBS;
static_assert(kNumPart <= kStatStride, "partials overflow their slot");

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
HD int lvo_part_off(const mopoe_model& m, int i) {
    int off = kStatStride;
    for (int k = 0; k < i; ++k) off += round_up(m.input_dim[k], 4);
    return off;
}
HD int partials_stride(const mopoe_model& m) { return lvo_part_off(m, m.num_mods); }

// ---------------------------------------------------------------------------
// LDS carve-up of the fused latent kernel (floats).  Region R0 holds the
// hidden tiles while the heads GEMM runs and is re-used afterwards for the
// decoder-gradient tiles (g_xhat) and the K-split partials of g_z.
// ---------------------------------------------------------------------------
struct LatentLds {
    int hs[MOPOE_MAX_MODS];
    int gx[MOPOE_MAX_MODS];
    int gzp;      // [kGzChunks][kRows][ld_gzp]
    int ld_gzp;
    int heads[MOPOE_MAX_MODS];
    int gheads[MOPOE_MAX_MODS];
    int zj[MOPOE_MAX_JOBS];
    int gzj[MOPOE_MAX_JOBS];
    int red;      // [waves][kStatStride]
    int total;
};

HD void latent_lds_layout(const mopoe_model& m, const mopoe_step& st, int waves,
                          LatentLds& L) {
    int r0a = 0, r0b = 0, zcols = 0;
    for (int i = 0; i < m.num_mods; ++i) {
        L.hs[i] = r0a;
        L.gx[i] = r0b;
        if ((st.present_mask >> i) & 1) {
            r0a += kRows * kLdH;
            r0b += kRows * ld_x_lds(m, i);
            zcols += round_up(z_dim(m, i), 16);
        }
    }
    L.ld_gzp = zcols + 4;
    L.gzp = r0b;
    r0b += kGzChunks * kRows * L.ld_gzp;
    int off = r0a > r0b ? r0a : r0b;
    for (int i = 0; i < m.num_mods; ++i) {
        L.heads[i] = off;
        L.gheads[i] = off;
        if ((st.present_mask >> i) & 1) {
            off += kRows * ld_heads_lds(m, i);
            L.gheads[i] = off;
            off += kRows * ld_heads_lds(m, i);
        }
    }
    for (int j = 0; j < st.num_jobs; ++j) {
        int i = st.job_mod[j];
        L.zj[j] = off;
        off += kRows * ld_z_lds(m, i);
        L.gzj[j] = off;
        off += kRows * ld_z_lds(m, i);
    }
    L.red = off;
    off += waves * kStatStride;
    L.total = off;
}

// ---------------------------------------------------------------------------
// device-only helpers
// ---------------------------------------------------------------------------
#if defined(__HIPCC__)

DEV f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    // v_mfma_f32_16x16x4_f32: exact f32 fmaf chain (guide section 3).
    // A: lane l holds A[row l&15][k l>>4]; B: B[k l>>4][col l&15];
    // C/D: col = l&15, row = 4*(l>>4) + reg.
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

DEV float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// A fragment: LDS tile [16][lda] (k contiguous).  Lane (r = l&15, q = l>>4)
// reads k = kb+4q .. kb+4q+3; the four values feed the four MFMA k-steps of a
// 16-deep block.  The k order inside the block is permuted (step i covers
// k = kb + 4q + i), identically for A and B, which leaves the sum unchanged.
DEV f32x4 lds_a4(const float* As, int lda, int kb, int lane) {
    return *reinterpret_cast<const f32x4*>(As + (lane & 15) * lda + kb + 4 * (lane >> 4));
}

// B fragment of Y = A * W^T: W is (ncols, K) row-major with row stride ldw.
DEV f32x4 glb_b4_nt(const float* __restrict__ W, int ldw, int ncols, int K, int j0,
                    int kb, int lane, bool vec) {
    const int col = j0 + (lane & 15);
    const int k = kb + 4 * (lane >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < ncols) {
        const float* p = W + (size_t)col * ldw + k;
        if (vec && k + 3 < K) {
            v = *reinterpret_cast<const f32x4*>(p);
        } else {
            if (k < K) v[0] = p[0];
            if (k + 1 < K) v[1] = p[1];
            if (k + 2 < K) v[2] = p[2];
            if (k + 3 < K) v[3] = p[3];
        }
    }
    return v;
}

// B fragment of Y = A * B: B is (K, ncols) row-major with row stride ldb.
DEV f32x4 glb_b4_nn(const float* __restrict__ B, int ldb, int ncols, int K, int j0,
                    int kb, int lane) {
    const int col = j0 + (lane & 15);
    const int k = kb + 4 * (lane >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < ncols) {
        const float* p = B + (size_t)k * ldb + col;
        if (k < K) v[0] = p[0];
        if (k + 1 < K) v[1] = p[(size_t)ldb];
        if (k + 2 < K) v[2] = p[(size_t)2 * ldb];
        if (k + 3 < K) v[3] = p[(size_t)3 * ldb];
    }
    return v;
}

// One 16x16 output tile, A (16 x K, zero padded to a multiple of 16) in LDS,
// B streamed from global/L2 straight into registers (each B element is used by
// exactly one wave of the workgroup, so an LDS round trip would be pure
// overhead); next block's fragments are fetched ahead of the MFMAs.
template <bool NT>
DEV f32x4 tile_gemm(f32x4 acc, const float* As, int lda, const float* __restrict__ B,
                    int ldb, int ncols, int K, int j0, int kbeg, int kend, int lane,
                    bool vec) {
    if (kbeg >= kend) return acc;
    f32x4 a = lds_a4(As, lda, kbeg, lane);
    f32x4 b = NT ? glb_b4_nt(B, ldb, ncols, K, j0, kbeg, lane, vec)
                 : glb_b4_nn(B, ldb, ncols, K, j0, kbeg, lane);
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    for (int kb = kbeg; kb < kend; kb += 16) {
        f32x4 an = {0.f, 0.f, 0.f, 0.f}, bn = {0.f, 0.f, 0.f, 0.f};
        if (kb + 16 < kend) {
            an = lds_a4(As, lda, kb + 16, lane);
            bn = NT ? glb_b4_nt(B, ldb, ncols, K, j0, kb + 16, lane, vec)
                    : glb_b4_nn(B, ldb, ncols, K, j0, kb + 16, lane);
        }
        // two accumulators: the 16x16x4 f32 MFMA has a 40-cycle dependent
        // latency against a 32-cycle issue interval
        acc = mfma_16x16x4(a[0], b[0], acc);
        acc2 = mfma_16x16x4(a[1], b[1], acc2);
        acc = mfma_16x16x4(a[2], b[2], acc);
        acc2 = mfma_16x16x4(a[3], b[3], acc2);
        a = an;
        b = bn;
    }
    return acc + acc2;
}

// Philox4x32-10 -> one standard normal (Box-Muller).  Counter = (element,
// stream, step, tag), key = seed.
DEV float philox_normal(uint64_t seed, uint32_t step, uint32_t stream, uint32_t idx) {
    uint32_t c0 = idx, c1 = stream, c2 = step, c3 = 0x4d6f506fu;
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
    const float u1 = ((c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    const float u2 = ((c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

#endif  // __HIPCC__
