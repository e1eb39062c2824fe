// Fused dynamic-NGP radiance field for gfx950: Frequency encoding -> motion MLP -> normalise /
// selector -> multi-resolution hash gather -> [time encoding] -> mlp_base -> trunc_exp ->
// SH(2) -> mlp_head -> sigmoid, one kernel, nothing but the inputs and (rgb, sigma) touch HBM.
// Replaces DNGPradianceField.forward (cednerf/model.py:468-488; query_move :354-365,
// query_density :367-445, _query_rgb :447-466), i.e. five tiny-cuda-nn launches plus ~12 torch
// glue kernels per call, and the sigma_fn / rgb_sigma_fn closures of cednerf/utils.py:74-104,181-195.
//
// Execution shape (CDNA4): one 768-thread workgroup per CU (3 waves per SIMD; other launch geometries
// behind ced_set_option("field_variant")), persistent over 32-sample wave tiles.  All nine weight matrices (~86 KB fp32, pre-swizzled by the host into
// MFMA A-fragment order) are staged into LDS once per workgroup and read with conflict-free
// ds_read_b128.  The GEMMs run on v_mfma_f32_16x16x4_f32 with D^T = W * X^T (neurons on the
// accumulator rows, samples on lanes): exact fp32, an ascending-k fused-multiply-add chain,
// which is what makes sample counts downstream bit-exact against the CPU oracle.  Activations
// never leave registers, and they never move between lanes either: output rows are independent
// dot products, so the host packs every hidden layer's weight rows such that accumulator row
// 16nb + 4g + r (lane group g, register r) holds neuron 16nb + 4r + g -- which is exactly the element
// lane group g must supply as B operand of k-step 4nb + r of the next layer.  A layer's accumulator
// registers ARE the next layer's B operand (after a one-instruction ReLU), with the ascending-k
// summation order intact.  The hash gather lives in the same geometry: lane group g owns levels
// 4i + 2(g&1) + (g>>1), i = 0..3, of its 16 samples (128 gathers per sample spread over 4 lanes) and one
// v_permlane16_swap per level pair puts the features in operand order.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "ced_common.hpp"
#include "field_args.hpp"
#include "field_device.hpp"
#include "field_kernel.hpp"

namespace ced {


// ---- standalone hash-grid encode (one lane per point, all levels) ------------------------------
struct HashArgs {
    int64_t n;
    const float *x, *t;
    float *out;
    int n_levels, table_dtype, temporal;
    const void *table;
    float scale[CED_MAX_LEVELS];
    uint32_t res[CED_MAX_LEVELS], offset[CED_MAX_LEVELS], size[CED_MAX_LEVELS], hashed[CED_MAX_LEVELS];
};

template <bool F16, bool TEMPORAL>
__global__ __launch_bounds__(256) void hash_encode_kernel(HashArgs A)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    float x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) x[a] = __builtin_fminf(__builtin_fmaxf(A.x[3 * i + a], 0.0f), 1.0f);
    int k_lo = 0;
    float t_frac = 0.0f;
    if constexpr (TEMPORAL) temporal_keyframe(A.t ? A.t[i] : 0.0f, k_lo, t_frac);
    float *o = A.out + i * 2 * A.n_levels;
    // a lane owns one 8 * n_levels-byte output row: two levels' features leave as one 16-byte store when the row is
    // aligned (8-byte stores of 64 lanes in 64 rows wrote five times the bytes: 32-byte sectors written a quarter full)
    const bool vec = (A.n_levels & 1) == 0 && (reinterpret_cast<uintptr_t>(A.out) & 15) == 0;
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int l = 0; l < CED_MAX_LEVELS; ++l) {
        if (l < A.n_levels) {
            const LevelConst L = make_level(A.scale[l], A.res[l], A.offset[l], A.size[l], A.hashed[l],
                                            EntryBytes<F16, TEMPORAL>::value);
            float f0, f1;
            if (A.hashed[l]) hash_level<F16, TEMPORAL, 2>(L, A.table, x, k_lo, t_frac, f0, f1);      // level-uniform
            else hash_level<F16, TEMPORAL, 1>(L, A.table, x, k_lo, t_frac, f0, f1);
            if (!vec) {
                o[2 * l] = f0;
                o[2 * l + 1] = f1;
            } else if ((l & 1) == 0) {
                p0 = f0; p1 = f1;
            } else {
                typedef float of4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<of4 *>(o + 2 * (l - 1)) = of4{ p0, p1, f0, f1 };
            }
        }
    }
}

// ---- backward of the (non-temporal) hash encode: next row f2, training path ---------------------------
// One lane per (sample, level) as in hash_encoder_half.py:164-226.  Table gradient: one hardware fp32 atomic add
// per corner and feature (the reference's `hash_grad[index] += w * dL/dy`).  As in the reference the position
// gradient is w.r.t. the scaled position (no `scale` factor), and levels whose dL/dy is all zero are skipped.
struct HashBwdArgs {
    int64_t n;
    const float *x, *dy;
    const float *t;              // temporal tables: the sample times (hash_table_grad_temporal_kernel)
    float *grad_table, *dx;
    int n_levels, table_dtype, dx_scaled;
    const void *table;
    float scale[CED_MAX_LEVELS];
    uint32_t res[CED_MAX_LEVELS], offset[CED_MAX_LEVELS], size[CED_MAX_LEVELS], hashed[CED_MAX_LEVELS];
};

// Table gradient, second form (round 2): FOUR adjacent lanes = (x corner, feature) of one sample's entry pair, one level
// per blockIdx.y.  The two x corners of a (y, z) corner are adjacent table entries whenever the cell's x index is even
// (dense levels: always adjacent; hashed levels: (x ^ h) and ((x + 1) ^ h) differ in bit 0 only), so the four lanes add
// into 16 contiguous bytes and the memory side sees ONE request where the first form (hash_backward_kernel<., true>,
// one corner per instruction) sent two.  Runs of consecutive samples with the same entry are still summed across the
// wave first (segmented scan, stride 4).  The atomics are what bounds this kernel (requests, not bytes).
__global__ __launch_bounds__(256) void hash_table_grad_kernel(HashBwdArgs A)
{
  // grid-stride over the samples: the launch may be capped to a few workgroups per CU (ced_set_option
  // "hash_grad_blocks"), so that a caller can run it beside compute-bound kernels of another stream -- the atomics
  // are fire-and-forget and a handful of waves per CU keep the memory side's atomic units busy
  for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid - threadIdx.x < 4 * A.n;
       gid += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = gid >> 2;
    const int feat = (int)(gid & 1), cx = (int)((gid >> 1) & 1);
    const int l = (int)blockIdx.y;
    const int lane = threadIdx.x & 63;
    uint32_t pend_idx[4];
    float pend_val[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { pend_idx[c] = 0xffffffffu; pend_val[c] = 0.0f; }
    if (i < A.n) {
        const float g0 = A.dy[(i * A.n_levels + l) * 2], g1 = A.dy[(i * A.n_levels + l) * 2 + 1];
        if (g0 != 0.0f || g1 != 0.0f) {
            const float sc = A.scale[l];
            uint32_t g[3];
            float fr[3], om[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float xa = __builtin_fminf(__builtin_fmaxf(A.x[3 * i + a], 0.0f), 1.0f);
                const float pos = xa * sc + 0.5f;
                const float fl = __builtin_floorf(pos);
                g[a] = (uint32_t)fl;
                fr[a] = pos - fl;
                om[a] = 1.0f - fr[a];
            }
            const uint32_t res = A.res[l], size = A.size[l], off = A.offset[l];
            const bool hashed = A.hashed[l] != 0;
            const float gv = feat ? g1 : g0;
            const uint32_t px = g[0] + cx;
            const float wx = cx ? fr[0] : om[0];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t py = g[1] + (c & 1), pz = g[2] + (c >> 1);
                const float wy = (c & 1) ? fr[1] : om[1], wz = (c & 2) ? fr[2] : om[2];
                const float w = (wx * wy) * wz;                          // the forward's weight: (wx * wy) * wz
                uint32_t idx = hashed ? (px ^ (py * 2654435761u) ^ (pz * 805459861u)) : (px + py * res + pz * res * res);
                pend_idx[c] = off + idx % size;
                pend_val[c] = w * gv;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t idx = pend_idx[c];
        float v = pend_val[c];
        const uint32_t prev = __shfl_up(idx, 4, 64);
        int head = (lane < 4 || prev != idx) ? 1 : 0;
#pragma unroll
        for (int d = 4; d < 64; d <<= 1) {
            const float v_up = __shfl_up(v, d, 64);
            const int h_up = __shfl_up(head, d, 64);
            if (lane >= d && !head) { v += v_up; head |= h_up; }
        }
        const uint32_t next = __shfl_down(idx, 4, 64);
        const bool last = lane >= 60 || next != idx;
        if (last && idx != 0xffffffffu && v != 0.0f) unsafeAtomicAdd(A.grad_table + (size_t)idx * 2 + feat, v);
    }
  }
}

// Table gradient of the TEMPORAL table (entry = 4 key-frames x 2 features, hash_encoder_inter.py:202-275): a sample at time
// t interpolates the key-frames k and k + 1 (k = min(floor(3 t), 2), weight t_frac = 3 t - floor(3 t): the reference's
// own arithmetic, t = 1 included), so its gradient lands in FOUR contiguous floats of every corner's entry:
// [2k + f] += w dy_f (1 - t_frac), [2k + 2 + f] += w dy_f t_frac.  Four adjacent lanes = those four floats (one 16-byte
// request per corner), eight corners per lane, one level per blockIdx.y; runs of consecutive samples that meet the same
// entry AND the same key-frame pair are summed across the wave first, as in hash_table_grad_kernel.
__global__ __launch_bounds__(256) void hash_table_grad_temporal_kernel(HashBwdArgs A)
{
  for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid - threadIdx.x < 4 * A.n;
       gid += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = gid >> 2;
    const int feat = (int)(gid & 1), up = (int)((gid >> 1) & 1);
    const int l = (int)blockIdx.y;
    const int lane = threadIdx.x & 63;
    uint32_t pend_slot[8];          // float index of this lane's slot in the table: entry * 8 + 2 * (k + up) + feat
    float pend_val[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { pend_slot[c] = 0xffffffffu; pend_val[c] = 0.0f; }
    if (i < A.n) {
        const float g0 = A.dy[(i * A.n_levels + l) * 2], g1 = A.dy[(i * A.n_levels + l) * 2 + 1];
        if (g0 != 0.0f || g1 != 0.0f) {
            int k_lo;
            float t_frac;
            temporal_keyframe(A.t[i], k_lo, t_frac);
            const float tw = up ? t_frac : 1.0f - t_frac;
            const float sc = A.scale[l];
            uint32_t g[3];
            float fr[3], om[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float xa = __builtin_fminf(__builtin_fmaxf(A.x[3 * i + a], 0.0f), 1.0f);
                const float pos = xa * sc + 0.5f;
                const float fl = __builtin_floorf(pos);
                g[a] = (uint32_t)fl;
                fr[a] = pos - fl;
                om[a] = 1.0f - fr[a];
            }
            const uint32_t res = A.res[l], size = A.size[l], off = A.offset[l];
            const bool hashed = A.hashed[l] != 0;
            const float gv = feat ? g1 : g0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t px = g[0] + (c & 1), py = g[1] + ((c >> 1) & 1), pz = g[2] + (c >> 2);
                const float wx = (c & 1) ? fr[0] : om[0], wy = (c & 2) ? fr[1] : om[1], wz = (c & 4) ? fr[2] : om[2];
                const float w = (wx * wy) * wz;                          // the forward's weight
                const uint32_t idx = hashed ? (px ^ (py * 2654435761u) ^ (pz * 805459861u)) : (px + py * res + pz * res * res);
                pend_slot[c] = (off + idx % size) * 8u + 2u * (uint32_t)(k_lo + up) + (uint32_t)feat;
                pend_val[c] = (w * gv) * tw;                             // (w * dy) * (1 - t_frac | t_frac), :264-267
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t slot = pend_slot[c];
        float v = pend_val[c];
        const uint32_t prev = __shfl_up(slot, 4, 64);
        int head = (lane < 4 || prev != slot) ? 1 : 0;
#pragma unroll
        for (int d = 4; d < 64; d <<= 1) {
            const float v_up = __shfl_up(v, d, 64);
            const int h_up = __shfl_up(head, d, 64);
            if (lane >= d && !head) { v += v_up; head |= h_up; }
        }
        const uint32_t next = __shfl_down(slot, 4, 64);
        const bool last = lane >= 60 || next != slot;
        if (last && slot != 0xffffffffu && v != 0.0f) unsafeAtomicAdd(A.grad_table + (size_t)slot, v);
    }
  }
}

// LEVEL_MAJOR: blockIdx.y = level, consecutive lanes = consecutive samples -- a wave's atomics then fall into one
// level's region and, for ray-ordered samples, into few cache lines (used for the table gradient).  Otherwise 16
// consecutive lanes = the 16 levels of one sample, whose position-gradient contributions are summed inside the
// 16-lane group in a fixed order and stored without atomics (used for dx).
template <bool F16, bool LEVEL_MAJOR>
__global__ __launch_bounds__(256) void hash_backward_kernel(HashBwdArgs A)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // LEVEL_MAJOR: two adjacent lanes = the two features of one sample's entries, so one atomic instruction covers
    // 32 entries x 2 adjacent dwords (half the cache-line requests of 64 entries x 1 dword, twice)
    const int64_t i = LEVEL_MAJOR ? (gid >> 1) : (gid >> 4);
    const int feat = (int)(gid & 1);
    const int l = LEVEL_MAJOR ? (int)blockIdx.y : (int)(gid & 15);
    float gx[3] = { 0.0f, 0.0f, 0.0f };
    uint32_t pend_idx[8];
    float pend_val[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { pend_idx[c] = 0xffffffffu; pend_val[c] = 0.0f; }
    if (i < A.n && l < A.n_levels) {
        const float g0 = A.dy[(i * A.n_levels + l) * 2], g1 = A.dy[(i * A.n_levels + l) * 2 + 1];
        if (g0 != 0.0f || g1 != 0.0f) {
            const float sc = A.scale[l];
            uint32_t g[3];
            float fr[3], om[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float xa = __builtin_fminf(__builtin_fmaxf(A.x[3 * i + a], 0.0f), 1.0f);
                const float pos = xa * sc + 0.5f;
                const float fl = __builtin_floorf(pos);
                g[a] = (uint32_t)fl;
                fr[a] = pos - fl;
                om[a] = 1.0f - fr[a];
            }
            const uint32_t res = A.res[l], size = A.size[l], off = A.offset[l];
            const bool hashed = A.hashed[l] != 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t px = g[0] + (c & 1), py = g[1] + ((c >> 1) & 1), pz = g[2] + ((c >> 2) & 1);
                const float wx = (c & 1) ? fr[0] : om[0], wy = (c & 2) ? fr[1] : om[1], wz = (c & 4) ? fr[2] : om[2];
                const float w = (wx * wy) * wz;
                uint32_t idx = hashed ? (px ^ (py * 2654435761u) ^ (pz * 805459861u)) : (px + py * res + pz * res * res);
                idx = off + idx % size;
                if constexpr (!LEVEL_MAJOR) {
                    float f0, f1;
                    if constexpr (!F16) {
                        const float2 v = reinterpret_cast<const float2 *>(A.table)[idx];
                        f0 = v.x; f1 = v.y;
                    } else {
                        const uint32_t u = reinterpret_cast<const uint32_t *>(A.table)[idx];
                        f0 = half_bits_to_float((uint16_t)(u & 0xffffu)); f1 = half_bits_to_float((uint16_t)(u >> 16));
                    }
                    // d w / d pos_a = +-(product of the other two factors)
                    const float dot = (f0 * g0 + f1 * g1) * (A.dx_scaled ? sc : 1.0f);
                    gx[0] += dot * ((c & 1) ? (wy * wz) : -(wy * wz));
                    gx[1] += dot * ((c & 2) ? (wx * wz) : -(wx * wz));
                    gx[2] += dot * ((c & 4) ? (wx * wy) : -(wx * wy));
                } else {
                    pend_idx[c] = idx;
                    pend_val[c] = w * (feat ? g1 : g0);
                }
            }
        }
    }
    if constexpr (LEVEL_MAJOR) {
        // Consecutive samples of a ray sit in the same cell of the coarse and middle levels: sum the runs of equal
        // entries across the wave first (segmented scan over same-feature lanes, stride 2) and let the last lane of
        // a run issue ONE atomic -- the atomics are what bounds this kernel (one L2 request per cache line touched).
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t idx = pend_idx[c];
            float v = pend_val[c];
            const uint32_t prev = __shfl_up(idx, 2, 64);
            int head = (lane < 2 || prev != idx) ? 1 : 0;
#pragma unroll
            for (int d = 2; d < 64; d <<= 1) {
                const float v_up = __shfl_up(v, d, 64);
                const int h_up = __shfl_up(head, d, 64);
                if (lane >= d && !head) { v += v_up; head |= h_up; }
            }
            const uint32_t next = __shfl_down(idx, 2, 64);
            const bool last = lane >= 62 || next != idx;
            if (last && idx != 0xffffffffu && v != 0.0f) unsafeAtomicAdd(A.grad_table + (size_t)idx * 2 + feat, v);
        }
    }
    if constexpr (!LEVEL_MAJOR) {
        // sum over the 16 levels of the sample (lanes 16k..16k+15), fixed order
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) gx[a] += __shfl_down(gx[a], off, 16);
        }
        if (l == 0 && i < A.n) {
            A.dx[3 * i] = gx[0]; A.dx[3 * i + 1] = gx[1]; A.dx[3 * i + 2] = gx[2];
        }
    }
}

// Diagnostic knobs of ced_set_option: process-wide, read by concurrently rendering threads -> atomics (relaxed: each is
// an independent launch property; no setting changes a result).
std::atomic<int> g_field_stagger{ 0 };          // field kernel: start-up phase offset between the waves of a SIMD, in s_sleep(127) units
std::atomic<int> g_field_spread_tiles{ 2 };     // field kernels: tile -> wave mapping (field_device.hpp: field_tile_range)
std::atomic<int> g_march_early_out{ 1 };        // frame renderer: conservative brick-level early-out
std::atomic<int> g_march_two_pass{ -1 };        // frame renderer: first iteration as culling pass + marching of the rest (-1: when there are several grid levels)

// launch-geometry variant of the field kernel (ced_set_option("field_variant", v))
static std::atomic<int> g_hash_grad_blocks{ [] { const char *e = getenv("CED_HASH_GRAD_BLOCKS"); return e ? atoi(e) : 0; }() };
static std::atomic<int> g_hash_grad_form{ []  { const char *e = getenv("CED_HASH_GRAD_FORM"); return e ? atoi(e) : 1; }() };   // 0: one corner per instruction
static std::atomic<int> g_field_variant{ [] { const char *e = getenv("CED_FIELD_VARIANT"); return e ? atoi(e) : 3; }() };

static int validate_hash(const ced_hash_desc *h, const char *who)
{
    CED_REQUIRE(h != nullptr, "%s: null hash descriptor", who);
    CED_REQUIRE(h->n_levels >= 1 && h->n_levels <= CED_MAX_LEVELS, "%s: n_levels=%d out of range", who, h->n_levels);
    CED_REQUIRE(h->table_dtype == 0 || h->table_dtype == 1, "%s: table_dtype must be 0 (f32) or 1 (f16)", who);
    CED_REQUIRE(h->table != nullptr, "%s: null hash table", who);
    for (int l = 0; l < h->n_levels; ++l) {
        CED_REQUIRE(h->size[l] > 0, "%s: level %d has zero entries", who, l);
        if (h->hashed[l])
            CED_REQUIRE((h->size[l] & (h->size[l] - 1)) == 0, "%s: hashed level %d size %u is not a power of two", who, l,
                        h->size[l]);
        else
            CED_REQUIRE((uint64_t)h->res[l] * h->res[l] * h->res[l] <= h->size[l],
                        "%s: dense level %d smaller than res^3", who, l);
        CED_REQUIRE((uint64_t)h->offset[l] + h->size[l] <= h->total_entries, "%s: level %d exceeds the table", who, l);
    }
    return CED_OK;
}

int launch_field(const ced_field_desc *d, FieldArgs &A, void *stream)
{
    int rc = validate_hash(&d->hash, "field_forward");
    if (rc) return rc;
    if (d->hash.n_levels != 16) {
        set_error("field_forward: the fused kernel needs n_levels == 16 (got %d)", d->hash.n_levels);
        return CED_E_UNSUPPORTED;
    }
    CED_REQUIRE(d->time_mode >= 0 && d->time_mode <= 2, "field_forward: time_mode=%d", d->time_mode);
    CED_REQUIRE(d->packed_weights != nullptr, "field_forward: null packed_weights");
    CED_REQUIRE(d->mlp_precision >= CED_MLP_F32 && d->mlp_precision <= CED_MLP_F32_HEAD16X2, "field_forward: mlp_precision=%d",
                d->mlp_precision);
    CED_REQUIRE((int64_t)d->packed_floats == ced_packed_weight_words(d->use_div_offsets, d->time_mode, d->mlp_precision),
                "field_forward: packed_floats=%llu does not match this configuration (mlp_precision %d)",
                (unsigned long long)d->packed_floats, d->mlp_precision);
    for (int i = 0; i < 6; ++i) A.aabb[i] = d->aabb[i];
    A.moving_step = d->moving_step;
    A.use_div = d->use_div_offsets ? 1 : 0;
    A.time_mode = d->time_mode;
    A.weights = d->packed_weights;
    A.table_dtype = d->hash.table_dtype;
    A.temporal = d->hash.temporal ? 1 : 0;
    A.table = d->hash.table;
    for (int l = 0; l < CED_MAX_LEVELS; ++l) {
        A.scale[l] = d->hash.scale[l];
        A.res[l] = d->hash.res[l];
        A.offset[l] = d->hash.offset[l];
        A.size[l] = d->hash.size[l];
        A.hashed[l] = d->hash.hashed[l];
    }
    // per gather slot i: levels 4i..4i+3: 1 = all dense, 2 = all hashed, 0 = mixed, 3 = all dense with the x-corner pairs
    // as one load (hash_level MODE 3): non-temporal tables, and every level of the slot followed by another level of the
    // table -- the pair load of a level's last entry reads one entry past the level.
    A.level_mode = 0;
    for (int i = 0; i < 4; ++i) {
        int n_hashed = 0;
        bool pairs = !A.temporal;
        for (int g = 0; g < 4; ++g) {
            const int l = 4 * i + g;
            n_hashed += d->hash.hashed[l] ? 1 : 0;
            pairs = pairs && l + 1 < d->hash.n_levels &&
                    (uint64_t)d->hash.offset[l] + d->hash.size[l] < d->hash.total_entries;
        }
        A.level_mode |= (n_hashed == 0 ? (pairs ? 3 : 1) : (n_hashed == 4 ? 2 : 0)) << (2 * i);
    }
    // byte offsets are 32-bit
    CED_REQUIRE(d->hash.total_entries * (uint64_t)((A.table_dtype ? 4 : 8) * (A.temporal ? 4 : 1)) < (1ull << 32),
                "field_forward: hash table larger than 4 GiB");
    CED_REQUIRE(d->max_workgroups >= 0 && d->max_workgroups <= 65536, "field_forward: max_workgroups=%d", d->max_workgroups);
    A.max_blocks = d->max_workgroups;
    A.stagger = g_field_stagger;
    A.spread_tiles = g_field_spread_tiles;
    if (d->mlp_precision == CED_MLP_F32_HEAD16X2) return launch_field_mixed(A, d->time_mode, stream);
    if (d->mlp_precision != CED_MLP_F32) {
        // the half kernels gather level 4i + g in slot i: the same slot -> level-range mapping as above
        return launch_field_half(A, d->time_mode, d->mlp_precision, stream);
    }
    // 1024 threads (four waves per SIMD, 128 registers) is the default since the dense levels' pair loads of round 4 (the
    // kernels fit without scratch: C2 +2.4 %, C3 +2 % over 768 threads); the temporal-table kernels need 250-300 registers
    // and run two waves per SIMD without scratch instead.
    const int field_variant = g_field_variant.load(std::memory_order_relaxed);
    const int variant = (field_variant >= 2 && field_variant <= 3 && A.temporal) ? 1 : field_variant;
    auto launch = [&](auto kernel, int nt, int threads) {
        const int64_t n_tiles = (A.n + 16 * nt - 1) / (16 * nt);
        const int waves = threads / 64;
        int64_t blocks = A.spread_tiles ? (n_tiles + 3) / 4 : (n_tiles + waves - 1) / waves;
        const int cap = A.max_blocks > 0 ? A.max_blocks : kFieldBlocksDefault;
        if (blocks > cap) blocks = cap;                                   // one resident workgroup per CU, persistent over tiles
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, A);
    };
    const int sel = (d->time_mode ? 1 : 0) | (A.table_dtype ? 2 : 0) | (A.temporal ? 4 : 0);
#define CED_FIELD_CASE(NT_, TH_)                                                                                \
    switch (sel) {                                                                                              \
    case 0: launch(field_kernel<false, false, false, NT_, TH_>, NT_, TH_); break;                               \
    case 1: launch(field_kernel<true, false, false, NT_, TH_>, NT_, TH_); break;                                \
    case 2: launch(field_kernel<false, true, false, NT_, TH_>, NT_, TH_); break;                                \
    case 3: launch(field_kernel<true, true, false, NT_, TH_>, NT_, TH_); break;                                 \
    case 4: launch(field_kernel<false, false, true, NT_, TH_>, NT_, TH_); break;                                \
    case 5: launch(field_kernel<true, false, true, NT_, TH_>, NT_, TH_); break;                                 \
    case 6: launch(field_kernel<false, true, true, NT_, TH_>, NT_, TH_); break;                                 \
    default: launch(field_kernel<true, true, true, NT_, TH_>, NT_, TH_); break;                                 \
    }
    switch (variant) {
    case 1: CED_FIELD_CASE(2, 512) break;
    case 2: CED_FIELD_CASE(2, 768) break;
    case 3: CED_FIELD_CASE(2, 1024) break;
    case 4: CED_FIELD_CASE(1, 1024) break;
    default: CED_FIELD_CASE(4, 512) break;
    }
#undef CED_FIELD_CASE
    return check_launch("field_forward");
}

}  // namespace ced

extern "C" int ced_set_option(const char *key, int value)
{
    CED_REQUIRE(key != nullptr, "set_option: null key");
    if (strcmp(key, "field_stagger") == 0) {
        CED_REQUIRE(value >= 0 && value <= 64, "set_option: field_stagger must be 0..64");
        ced::g_field_stagger = value;
        return CED_OK;
    }
    if (strcmp(key, "field_spread_tiles") == 0) {
        CED_REQUIRE(value >= 0 && value <= 2, "set_option: field_spread_tiles must be 0, 1 or 2 (field_tile_range)");
        ced::g_field_spread_tiles = value;
        return CED_OK;
    }
    if (strcmp(key, "march_early_out") == 0) {
        ced::g_march_early_out = value != 0;
        return CED_OK;
    }
    if (strcmp(key, "march_two_pass") == 0) {
        CED_REQUIRE(value >= -1 && value <= 1, "set_option: march_two_pass must be -1 (automatic), 0 or 1");
        ced::g_march_two_pass = value;
        return CED_OK;
    }
    if (strcmp(key, "half_variant") == 0) {
        CED_REQUIRE(value >= 0 && value <= 4, "set_option: half_variant must be 0..3 (4: diagnostic builds only)");
        ced::set_half_variant(value);
        return CED_OK;
    }
    if (strcmp(key, "hash_grad_blocks") == 0) {
        CED_REQUIRE(value >= 0 && value <= (1 << 20), "set_option: hash_grad_blocks must be 0 (no cap) .. 2^20");
        ced::g_hash_grad_blocks = value;
        return CED_OK;
    }
    if (strcmp(key, "hash_grad_form") == 0) {
        CED_REQUIRE(value == 0 || value == 1, "set_option: hash_grad_form must be 0 or 1");
        ced::g_hash_grad_form = value;
        return CED_OK;
    }
    if (strcmp(key, "mixed_variant") == 0) {
        CED_REQUIRE(value >= 0 && value <= 2, "set_option: mixed_variant must be 0 (auto), 1 (512 threads) or 2 (768)");
        ced::set_mixed_variant(value);
        return CED_OK;
    }
    if (strcmp(key, "field_variant") == 0) {
        CED_REQUIRE(value >= 0 && value <= 4, "set_option: field_variant must be 0..4");
        ced::g_field_variant = value;
        return CED_OK;
    }
    ced::set_error("set_option: unknown key '%s'", key);
    return CED_E_INVALID;
}

extern "C" int64_t ced_packed_weight_floats(int use_div_offsets, int time_mode)
{
    (void)use_div_offsets;
    return time_mode ? ced::Blob<true>::TOTAL : ced::Blob<false>::TOTAL;
}

// Host-side reorder into MFMA A-fragment order: element (accumulator row p, input k) of a layer goes to
// [nb = p/16][q = (k/4)/4][lane = (k%4)*16 + p%16][s = (k/4)%4]; which neuron row p computes is the
// layer's placement (natural / hidden / base-out, below).
extern "C" int ced_pack_field_weights(int use_div_offsets, int time_mode, const float *m_w0, const float *m_w1,
                                      const float *m_w2, const float *m_w3, const float *b_w0, const float *b_w1,
                                      const float *h_w0, const float *h_w1, const float *h_w2, float *out)
{
    CED_REQUIRE(m_w0 && m_w1 && m_w2 && m_w3 && b_w0 && b_w1 && h_w0 && h_w1 && h_w2 && out,
                "pack_field_weights: null pointer");
    CED_REQUIRE(time_mode >= 0 && time_mode <= 2, "pack_field_weights: time_mode=%d", time_mode);
    const bool te = time_mode != 0;
    const int64_t total = ced_packed_weight_floats(use_div_offsets, time_mode);
    for (int64_t i = 0; i < total; ++i) out[i] = 0.0f;
    // placement: 0 natural, 1 hidden, 2 base-out, 3 one neuron per lane group (row 4a = neuron a)
    struct L { const float *w; int n_out, n_in, nb, ks, off; int placement; };
    const int base_in = te ? 41 : 32;
    const int n_mo = use_div_offsets ? 6 : 3;
    const int ksb0 = te ? 11 : 8;
    int offs[9];
    if (te) {
        using B = ced::Blob<true>;
        int o[9] = { B::M0, B::M1, B::M2, B::M3, B::B0, B::B1, B::H0, B::H1, B::H2 };
        for (int i = 0; i < 9; ++i) offs[i] = o[i];
    } else {
        using B = ced::Blob<false>;
        int o[9] = { B::M0, B::M1, B::M2, B::M3, B::B0, B::B1, B::H0, B::H1, B::H2 };
        for (int i = 0; i < 9; ++i) offs[i] = o[i];
    }
    const L layers[9] = {
        { m_w0, 64, 32, 4, 8, offs[0], 1 },      { m_w1, 64, 64, 4, 16, offs[1], 1 },
        { m_w2, 64, 64, 4, 16, offs[2], 1 },     { m_w3, n_mo, 64, 1, 16, offs[3], 0 },
        { b_w0, 64, base_in, 4, ksb0, offs[4], 1 }, { b_w1, 16, 64, 1, 16, offs[5], 2 },
        { h_w0, 64, 19, 4, 5, offs[6], 1 },      { h_w1, 64, 64, 4, 16, offs[7], 1 },
        { h_w2, 3, 64, 1, 16, offs[8], 3 },
    };
    for (const L &l : layers) {
        const int ks4 = ced::ks4_of(l.ks);
        for (int p = 0; p < l.nb * 16; ++p) {
            // which output neuron accumulator row p computes.  Hidden layers: row 16nb + 4g + r holds neuron
            // 16nb + 4r + g, so that the accumulator registers are the next layer's B operand as they are.
            int neuron = p;
            if (l.placement == 1) neuron = (p & ~15) | ((p & 3) << 2) | ((p >> 2) & 3);
            else if (l.placement == 2) neuron = ced::base_out_neuron(p);
            else if (l.placement == 3) neuron = (p & 3) ? l.n_out : (p >> 2);
            if (neuron >= l.n_out) continue;
            for (int k = 0; k < l.ks * 4; ++k) {
                if (k >= l.n_in) continue;
                const int S = k / 4, kk = k % 4;
                const int lane = kk * 16 + (p % 16);
                const int64_t idx = l.off + (((int64_t)(p / 16) * ks4 + S / 4) * 64 + lane) * 4 + (S % 4);
                out[idx] = l.w[(int64_t)neuron * l.n_in + k];
            }
        }
    }
    return CED_OK;
}

extern "C" int ced_field_forward(const ced_field_desc *desc, int64_t n, const float *positions, const float *t,
                                 const float *directions, float *rgb, float *sigma, float *geo, void *stream)
{
    CED_REQUIRE(desc != nullptr, "field_forward: null descriptor");
    CED_REQUIRE(n >= 0, "field_forward: n < 0");
    if (n == 0) return CED_OK;
    CED_REQUIRE(positions && t && sigma, "field_forward: null positions/t/sigma");
    CED_REQUIRE((directions != nullptr) == (rgb != nullptr), "field_forward: directions and rgb go together");
    ced::FieldArgs A{};
    A.n = n;
    A.pos = positions; A.t = t; A.dir = directions;
    A.rays_mode = 0; A.t_per_ray = 0; A.want_rgb = rgb ? 1 : 0;
    A.rgb = rgb; A.sigma = sigma; A.geo = geo;
    return ced::launch_field(desc, A, stream);
}

extern "C" int ced_field_forward_rays(const ced_field_desc *desc, int64_t n, const int64_t *n_dev,
                                      const float *rays_o, const float *rays_d, const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                      const float *timestamps, int32_t t_per_ray, int32_t want_rgb, float *rgb,
                                      float *sigma, void *stream)
{
    CED_REQUIRE(desc != nullptr, "field_forward_rays: null descriptor");
    CED_REQUIRE(n >= 0, "field_forward_rays: n < 0");
    if (n == 0) return CED_OK;
    CED_REQUIRE(rays_o && rays_d && ray_indices && t_starts && t_ends && timestamps && sigma,
                "field_forward_rays: null pointer");
    CED_REQUIRE(!want_rgb || rgb, "field_forward_rays: want_rgb without an rgb buffer");
    ced::FieldArgs A{};
    A.n = n;
    A.n_dev = n_dev;
    A.rays_o = rays_o; A.rays_d = rays_d; A.ray_idx = ray_indices;
    A.t0 = t_starts; A.t1 = t_ends; A.timestamps = timestamps;
    A.rays_mode = 1; A.t_per_ray = t_per_ray ? 1 : 0; A.want_rgb = want_rgb ? 1 : 0;
    A.rgb = rgb; A.sigma = sigma; A.geo = nullptr;
    return ced::launch_field(desc, A, stream);
}

extern "C" int ced_hash_encode(const ced_hash_desc *desc, int64_t n, const float *x, const float *t, float *out,
                               void *stream)
{
    int rc = ced::validate_hash(desc, "hash_encode");
    if (rc) return rc;
    CED_REQUIRE(n >= 0, "hash_encode: n < 0");
    if (n == 0) return CED_OK;
    CED_REQUIRE(x && out, "hash_encode: null pointer");
    ced::HashArgs A{};
    A.n = n; A.x = x; A.t = t; A.out = out;
    A.n_levels = desc->n_levels; A.table_dtype = desc->table_dtype; A.temporal = desc->temporal ? 1 : 0;
    A.table = desc->table;
    for (int l = 0; l < CED_MAX_LEVELS; ++l) {
        A.scale[l] = desc->scale[l]; A.res[l] = desc->res[l]; A.offset[l] = desc->offset[l];
        A.size[l] = desc->size[l]; A.hashed[l] = desc->hashed[l];
    }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const int sel = (A.table_dtype ? 1 : 0) | (A.temporal ? 2 : 0);
    switch (sel) {
    case 0: hipLaunchKernelGGL((ced::hash_encode_kernel<false, false>), grid, block, 0, (hipStream_t)stream, A); break;
    case 1: hipLaunchKernelGGL((ced::hash_encode_kernel<true, false>), grid, block, 0, (hipStream_t)stream, A); break;
    case 2: hipLaunchKernelGGL((ced::hash_encode_kernel<false, true>), grid, block, 0, (hipStream_t)stream, A); break;
    default: hipLaunchKernelGGL((ced::hash_encode_kernel<true, true>), grid, block, 0, (hipStream_t)stream, A); break;
    }
    return ced::check_launch("hash_encode");
}

extern "C" int ced_hash_encode_backward(const ced_hash_desc *desc, int64_t n, const float *x, const float *dy,
                                        float *grad_table, float *dx, int32_t dx_scaled, void *stream)
{
    int rc = ced::validate_hash(desc, "hash_encode_backward");
    if (rc) return rc;
    CED_REQUIRE(n >= 0, "hash_encode_backward: n < 0");
    CED_REQUIRE(!desc->temporal, "hash_encode_backward: the temporal table has no backward yet");
    CED_REQUIRE(desc->n_levels <= 16, "hash_encode_backward: n_levels > 16");
    if (n == 0) return CED_OK;
    CED_REQUIRE(x && dy && (grad_table || dx), "hash_encode_backward: null pointer");
    ced::HashBwdArgs A{};
    A.n = n; A.x = x; A.dy = dy; A.grad_table = grad_table; A.dx = dx;
    A.n_levels = desc->n_levels; A.table_dtype = desc->table_dtype; A.table = desc->table;
    A.dx_scaled = dx_scaled ? 1 : 0;
    for (int l = 0; l < CED_MAX_LEVELS; ++l) {
        A.scale[l] = desc->scale[l]; A.res[l] = desc->res[l]; A.offset[l] = desc->offset[l];
        A.size[l] = desc->size[l]; A.hashed[l] = desc->hashed[l];
    }
    const dim3 block(256);
    // table gradient: level-major; position gradient (optional): sample-major, no atomics
    if (!grad_table) {
        // position gradient only (the caller runs the table gradient elsewhere, e.g. on another stream)
    } else if (ced::g_hash_grad_form == 0) {
        const dim3 grid_t((unsigned)((2 * n + 255) / 256), (unsigned)desc->n_levels);
        hipLaunchKernelGGL((ced::hash_backward_kernel<false, true>), grid_t, block, 0, (hipStream_t)stream, A);
    } else {
        int64_t gx = (4 * n + 255) / 256;
        const int cap = ced::g_hash_grad_blocks;                // workgroups per level; <= 0: one per 64 samples
        if (cap > 0 && gx > cap) gx = cap;
        const dim3 grid_t((unsigned)gx, (unsigned)desc->n_levels);
        hipLaunchKernelGGL(ced::hash_table_grad_kernel, grid_t, block, 0, (hipStream_t)stream, A);
    }
    if (dx) {
        const dim3 grid_x((unsigned)((n * 16 + 255) / 256));
        if (A.table_dtype) hipLaunchKernelGGL((ced::hash_backward_kernel<true, false>), grid_x, block, 0, (hipStream_t)stream, A);
        else hipLaunchKernelGGL((ced::hash_backward_kernel<false, false>), grid_x, block, 0, (hipStream_t)stream, A);
    }
    return ced::check_launch("hash_encode_backward");
}

extern "C" int ced_hash_encode_backward_temporal(const ced_hash_desc *desc, int64_t n, const float *x, const float *t,
                                                 const float *dy, float *grad_table, void *stream)
{
    int rc = ced::validate_hash(desc, "hash_encode_backward_temporal");
    if (rc) return rc;
    CED_REQUIRE(n >= 0, "hash_encode_backward_temporal: n < 0");
    CED_REQUIRE(desc->temporal, "hash_encode_backward_temporal: the table is not temporal (use ced_hash_encode_backward)");
    CED_REQUIRE(desc->n_levels <= 16, "hash_encode_backward_temporal: n_levels > 16");
    CED_REQUIRE(desc->total_entries * 8ull < (1ull << 32), "hash_encode_backward_temporal: table too large for 32-bit slots");
    if (n == 0) return CED_OK;
    CED_REQUIRE(x && t && dy && grad_table, "hash_encode_backward_temporal: null pointer");
    ced::HashBwdArgs A{};
    A.n = n; A.x = x; A.t = t; A.dy = dy; A.grad_table = grad_table;
    A.n_levels = desc->n_levels; A.table_dtype = desc->table_dtype; A.table = desc->table;
    for (int l = 0; l < CED_MAX_LEVELS; ++l) {
        A.scale[l] = desc->scale[l]; A.res[l] = desc->res[l]; A.offset[l] = desc->offset[l];
        A.size[l] = desc->size[l]; A.hashed[l] = desc->hashed[l];
    }
    int64_t gx = (4 * n + 255) / 256;
    const int cap = ced::g_hash_grad_blocks;
    if (cap > 0 && gx > cap) gx = cap;
    hipLaunchKernelGGL(ced::hash_table_grad_temporal_kernel, dim3((unsigned)gx, (unsigned)desc->n_levels), dim3(256), 0,
                       (hipStream_t)stream, A);
    return ced::check_launch("hash_encode_backward_temporal");
}
